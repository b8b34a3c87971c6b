#!/usr/bin/env python3
"""One iteration's kernel timeline with the hardware queue of every launch (rocprofv3 --kernel-trace CSV of a multi-stream step: the
training tiers' side streams, independent sample chains): start, duration, queue, kernel — ordered by start — plus, per queue, busy time
and the idle time between its first and last launch of the iteration, and how long NO queue was running anything.

    python tools/trace_queues.py <kernel_trace.csv> [marker-substring=k_adamw] [min_us_to_print=0]
"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_adamw"
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void |s3d::", "", r["Kernel_Name"]).split("(")[0][:52], r.get("Queue_Id", "?"))
             for r in rows), key=lambda e: e[0])
ends = [i for i, e in enumerate(ev) if marker in e[2]]
if len(ends) < 3:
    sys.exit("not enough iterations in the trace")
a, b = ends[-3] + 1, ends[-2] + 1
it = ev[a:b]
t0 = it[0][0]
queues = sorted({e[3] for e in it})
col = {q: i for i, q in enumerate(queues)}
print(f"iteration span {(max(e[1] for e in it) - t0) / 1e3:.1f} us, {len(it)} launches on queues {queues}")
print(f"{'start_us':>9} {'dur_us':>8}  q  kernel")
for s, e, n, q in it:
    if (e - s) / 1e3 >= min_us:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {col[q]}  {'    ' * col[q]}{n}")
pts = sorted([(s, 1) for s, e, _, _ in it] + [(e, -1) for s, e, _, _ in it])
depth, last, idle = 0, pts[0][0], 0
for t, d in pts:
    if depth == 0:
        idle += t - last
    depth += d; last = t
print(f"no kernel in flight: {idle / 1e3:.1f} us of the iteration")
for q in queues:
    mine = [(s, e) for s, e, _, qq in it if qq == q]
    busy = sum(e - s for s, e in mine)
    print(f"queue {col[q]}: {len(mine)} launches, busy {busy / 1e3:.1f} us, first..last {(mine[0][0] - t0) / 1e3:.1f}..{(mine[-1][1] - t0) / 1e3:.1f} us")
