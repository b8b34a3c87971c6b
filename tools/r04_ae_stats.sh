#!/bin/bash
# per-kernel totals of the auto-encoder training iteration (tools/bench_ae_train.py under rocprofv3 --kernel-trace --stats)
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_ae
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_ae -o t --output-format csv -- python3 $ROOT/tools/bench_ae_train.py > /tmp/p_ae.log 2>&1
grep "^{" /tmp/p_ae.log | cut -c1-200
python3 - $(find /tmp/p_ae -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"  {r['Name'][:70]:70s} calls {int(r['Calls']):6d} avg_us {float(r['AverageNs'])/1e3:8.1f} share {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
