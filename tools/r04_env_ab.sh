#!/bin/bash
# A/B of environment settings on one box: tools/r04_env_ab.sh CONFIG "ENV=.." "ENV=.." ...   ("" = default)
ROOT=$GRAFT_REPO_ROOT; cd $ROOT; C=$1; shift
for SET in "$@"; do
  for rep in 1 2; do
  env $SET timeout 600 python bench.py --config $C --steps 200 --warmup 5 --no-cpu-baseline --traffic off 2>/dev/null | tail -1 | python3 -c "
import json, sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$SET] $C ms/step', round(d['ms_per_step'],4), 'conv', r['conv3x3_ms_per_step'], 'rank1', r['rank1_ms_per_step'], '1x1', r['conv1x1_ms_per_step'])"
  done
done
