#!/bin/bash
# tuning helper: bench the forced 3x3 tile configs
mkdir -p gpurun_out
for c in "" 0 1 2 3 4; do
  S3D_CONV_CFG=$c python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg=$c', 'ms/step', round(d['ms_per_step'],3), 'conv3x3 TF', r['achieved'], 'conv3x3 ms', r['conv3x3_ms_per_step'], 'rank1', r['rank1_ms_per_step'])"
done
