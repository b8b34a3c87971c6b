#!/bin/bash
# the driver's flags (--steps 20 --warmup 5) under different untimed pre-warm lengths, three repeats each, one box
ROOT=$GRAFT_REPO_ROOT; cd $ROOT
for PW in 150 600 1500 150; do
  for rep in 1 2 3; do
    python bench.py --steps 20 --warmup 5 --prewarm $PW --no-cpu-baseline --traffic off 2>/dev/null | tail -1 | python3 -c "
import json, sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('prewarm $PW ms/step', round(d['ms_per_step'],4), 'conv', r['conv3x3_ms_per_step'], 'clk', r['clock']['gfxclk_mhz_mean'] if r.get('clock') else None)"
  done
done
python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --traffic off 2>/dev/null | tail -1 | python3 -c "
import json, sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('1000 steps ms/step', round(d['ms_per_step'],4), 'conv', r['conv3x3_ms_per_step'])"
