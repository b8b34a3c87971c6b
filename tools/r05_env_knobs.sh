#!/bin/bash
# HIP runtime environment knobs against the headline step (bench.py --steps 300, no baselines), one box, in sequence
cd $GRAFT_REPO_ROOT
run() { echo "[$1] $(env $1 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --chains 0 --traffic off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"; }
run "S3D_NOOP=1"
for K in "$@"; do run "$K"; done
run "S3D_NOOP=1"
