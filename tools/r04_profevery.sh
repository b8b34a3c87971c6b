#!/bin/bash
# what the HIP-event instrumentation of every n-th step costs the timed region (bench.py --profile-every)
ROOT=$GRAFT_REPO_ROOT; cd $ROOT
for PE in 8 0 32 8 0 32; do
  python bench.py --steps 1000 --warmup 20 --profile-every $PE --no-cpu-baseline --traffic off 2>/dev/null | tail -1 | python3 -c "
import json, sys
d=json.loads(sys.stdin.read()); print('profile-every $PE  1000 steps  ms/step', round(d['ms_per_step'],4))"
done
for PE in 8 0 8 0; do
  python bench.py --steps 20 --warmup 5 --profile-every $PE --no-cpu-baseline --traffic off 2>/dev/null | tail -1 | python3 -c "
import json, sys
d=json.loads(sys.stdin.read()); print('profile-every $PE  20 steps  ms/step', round(d['ms_per_step'],4))"
done
