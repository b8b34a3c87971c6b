#!/bin/bash
# per-kernel average durations under environment settings: tools/r04_kernel_ab.sh CONFIG PATTERN "ENV=.." ...
ROOT=$GRAFT_REPO_ROOT; C=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for SET in "$@"; do
  rm -rf /tmp/pk; 
  env $SET timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o t --output-format csv -- python3 $ROOT/bench.py --config $C --steps 40 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off > /tmp/pk.log 2>&1
  echo "== [$SET] $C"; grep ms_per_step /tmp/pk.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],4))"
  python3 - "$PAT" $(find /tmp/pk -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys, re
pat = re.compile(sys.argv[1])
for r in csv.DictReader(open(sys.argv[2])):
    if pat.search(r["Name"]): print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
done
