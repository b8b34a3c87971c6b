#!/usr/bin/env python3
"""One step's kernel timeline from a rocprofv3 --kernel-trace CSV: start offset, duration and the idle gap before each
kernel, for the LAST complete step in the trace (a step ends with the output head, which applies the sampler update in the loops).

    python tools/trace_timeline.py <kernel_trace.csv> [marker-substring=k_out_head]
"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_out_head"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void |s3d::", "", r["Kernel_Name"]).split("(")[0][:60],
              int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])))
             for r in rows), key=lambda e: e[0])
ends = [i for i, e in enumerate(ev) if marker in e[2]]
if len(ends) < 3:
    sys.exit("not enough steps in the trace")
a, b = ends[-3] + 1, ends[-2] + 1
t0 = ev[a][0]
prev_end = ev[a - 1][1]
tot_gap = tot_busy = 0
print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel (blocks)")
for s, e, n, g in ev[a:b]:
    gap = (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {n} ({g})")
    tot_gap += max(gap, 0); tot_busy += (e - s) / 1e3
    prev_end = max(prev_end, e)
print(f"step span {(ev[b - 1][1] - ev[a - 1][1]) / 1e3:.1f} us, kernels {b - a}, sum of durations {tot_busy:.1f} us, sum of positive gaps {tot_gap:.1f} us")
