#!/bin/bash
# per-kernel totals of the training step (tools/bench_train.py under rocprofv3 --kernel-trace --stats); PATTERN filters kernel names
ROOT=$GRAFT_REPO_ROOT; PAT=${1:-.}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_tr
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 20 --warmup 3 > /tmp/p_tr.log 2>&1
grep "^{" /tmp/p_tr.log | cut -c1-120
python3 - "$PAT" $(find /tmp/p_tr -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys, re
pat = re.compile(sys.argv[1]); tot = 0
rows = list(csv.DictReader(open(sys.argv[2])))
for r in rows: tot += float(r["TotalDurationNs"])
print(f"total kernel time per step {tot/23/1e3:.1f} us (23 steps)")
for r in rows:
    if pat.search(r["Name"]): print(f"  {r['Name'][:60]:60s} calls/step {int(r['Calls'])/23:6.1f} avg_us {float(r['AverageNs'])/1e3:8.1f} us/step {float(r['TotalDurationNs'])/23/1e3:8.1f}")
PY
