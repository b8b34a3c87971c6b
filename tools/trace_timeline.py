#!/usr/bin/env python3
"""One step's kernel timeline from a rocprofv3 --kernel-trace CSV: start offset, duration and the idle gap before each
kernel, for the LAST complete step in the trace (a step ends with the output head, which applies the sampler update in the loops).

    python tools/trace_timeline.py <kernel_trace.csv> [marker-substring=k_out_head] [avg-steps=0]

avg-steps = N > 0: after the single step, the MEAN duration and MEAN gap of every launch position over the last N complete steps
of equal launch count, and where the process's copy kernels (__amd_rocclr_copyBuffer) fall: inside those steps or before them.
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

navg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if navg > 0:
    L = b - a
    steps = [(ends[k] + 1, ends[k + 1] + 1) for k in range(len(ends) - 1) if ends[k + 1] - ends[k] == L][-navg - 1:-1]
    if steps:
        dur = [0.0] * L; gap = [0.0] * L; gmax = [0.0] * L
        for (x, y) in steps:
            pe = ev[x - 1][1]
            for j in range(L):
                s_, e_, _, _ = ev[x + j]
                g_ = (s_ - pe) / 1e3
                dur[j] += (e_ - s_) / 1e3; gap[j] += max(g_, 0.0); gmax[j] = max(gmax[j], g_)
                pe = max(pe, e_)
        n = len(steps)
        print(f"\nmean over the last {n} steps of {L} launches: position, mean dur_us, mean gap_us (max), kernel")
        for j in range(L):
            flag = "  <-- gap" if gap[j] / n > 1.0 else ""
            print(f"{j:3d} {dur[j] / n:8.1f} {gap[j] / n:7.2f} ({gmax[j]:5.1f})  {ev[a + j][2]}{flag}")
        print(f"mean step: sum of durations {sum(dur) / n:.1f} us, sum of positive gaps {sum(gap) / n:.2f} us")
        t_first = ev[steps[0][0]][0]
        copies = [e for e in ev if "copyBuffer" in e[2]]
        inside = [e for e in copies if e[0] >= t_first]
        print(f"copy kernels in the whole trace: {len(copies)}; inside those {n} steps: {len(inside)} (the rest precede them: parameter uploads at model load)")
