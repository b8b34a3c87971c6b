#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; TAG=${1:-r04k}
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $OUT/${TAG}_pytest.log; tail -6 $OUT/${TAG}_pytest.log
for C in c2 c3; do for SET in "S3D_CONV1X1_T=0" ""; do
  env $SET timeout 600 python bench.py --config $C --steps 300 --warmup 5 --no-cpu-baseline --traffic off > $OUT/${TAG}_bench_${C}_${SET:-default}.json 2> $OUT/${TAG}_err.txt
  python3 -c "
import json
d=json.load(open('$OUT/${TAG}_bench_${C}_${SET:-default}.json')); r=d['roofline']
print('$C [${SET:-default}] ms/step', round(d['ms_per_step'],4), 'conv', r['conv3x3_ms_per_step'], 'frac', r['frac'], 'rank1', r['rank1_ms_per_step'], '1x1', r['conv1x1_ms_per_step'])" || tail -5 $OUT/${TAG}_err.txt
done; done
for SET in "S3D_CONV1X1_T=0" ""; do echo "ae [$SET]"; env $SET timeout 600 python tools/bench_ae_train.py 2>&1 | tail -1 | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p1
timeout 600 rocprofv3 --kernel-trace -d /tmp/p1 -o t --output-format csv -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off > /tmp/p1.log 2>&1
F=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_timeline.py $F > $OUT/${TAG}_timeline.txt
grep -v "wino24\|gn_act \|means_fin\|rank1" $OUT/${TAG}_timeline.txt
