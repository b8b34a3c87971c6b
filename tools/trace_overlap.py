#!/usr/bin/env python3
"""Concurrency figures from a rocprofv3 --kernel-trace CSV of several chains on several streams (tools/two_chain_probe.py):
how much of the wall span has 0 / 1 / >= 2 kernels in flight, per-queue busy time, and per-kernel mean durations (to set beside a
single-chain trace: a kernel that shares the chip runs longer).

    python tools/trace_overlap.py <kernel_trace.csv> [skip_fraction=0.3]
"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void |s3d::", "", r["Kernel_Name"]).split("(")[0][:48],
             r.get("Queue_Id", "?")) for r in rows)
t_lo = ev[0][0] + int(skip * (ev[-1][1] - ev[0][0]))          # (the warm-up part of the trace)
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
pts = []
for s, e, _, _ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[min(depth, 3)] += t - last
    depth += d; last = t
print(f"kernels {len(ev)}, span {span / 1e6:.3f} ms, sum of durations {sum(e - s for s, e, _, _ in ev) / 1e6:.3f} ms")
for k in sorted(hist):
    print(f"  {k}{'+' if k == 3 else ' '} kernels in flight: {hist[k] / 1e6:8.3f} ms  {100 * hist[k] / span:5.1f} %")
q = collections.Counter()
for s, e, _, qid in ev:
    q[qid] += e - s
print("busy time per queue: " + ", ".join(f"q{k}: {v / 1e6:.3f} ms" for k, v in sorted(q.items())))
agg = collections.OrderedDict()
for s, e, n, _ in ev:
    a = agg.setdefault(n, [0, 0]); a[0] += e - s; a[1] += 1
print("kernel                                           calls   avg_us   total_ms")
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {n:48s} {c:5d} {t / c / 1e3:8.1f} {t / 1e6:9.3f}")
