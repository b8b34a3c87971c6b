#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel (and per grid size) time per step."""
import csv, sys, collections, re
path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"s3d::", "", name)
    name = name.split("(")[0][:70]
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    key = (name, grid) if "conv_mfma" in name else (name, 0)
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0])
    a[0] += d; a[1] += 1
tot = sum(a[0] for a in agg.values())
print(f"total kernel time {tot/1e6:.2f} ms = {tot/1e6/steps:.3f} ms/step over {steps} steps")
for (name, grid), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if t / tot < 0.002: continue
    print(f"{name:72s} blocks={grid:6d} calls/step={n/steps:5.2f} avg_us={t/n/1e3:8.1f} ms/step={t/1e6/steps:7.4f} {100*t/tot:5.1f}%")
