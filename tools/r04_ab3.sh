#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; TAG=${1:-r04l}
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "rank1 or golden or leaf" 2>&1 | tail -6
for C in c2 c3 c5; do for SET in "S3D_RANK1_WIDE_MIN=0" "S3D_RANK1_WIDE_MIN=1"; do
  env $SET timeout 600 python bench.py --config $C --steps 300 --warmup 5 --no-cpu-baseline --traffic off > $OUT/${TAG}_bench_${C}_${SET}.json 2> $OUT/${TAG}_err.txt
  python3 -c "
import json
d=json.load(open('$OUT/${TAG}_bench_${C}_${SET}.json')); r=d['roofline']
print('$C [${SET}] ms/step', round(d['ms_per_step'],4), 'conv', r['conv3x3_ms_per_step'], 'rank1', r['rank1_ms_per_step'], '1x1', r['conv1x1_ms_per_step'])" || tail -5 $OUT/${TAG}_err.txt
done; done
for SET in "S3D_RANK1_WIDE_MIN=0" "S3D_RANK1_WIDE_MIN=1"; do echo "64-ch B=4 [$SET]"; env $SET timeout 300 python tools/bench_configs.py "64-ch" 2>&1 | tail -1 | cut -c1-160; done
