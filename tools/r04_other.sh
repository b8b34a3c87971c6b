#!/bin/bash
# GPU suite (rest), the other BASELINE configs through bench.py (own roofline each), training / AE step times
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; TAG=${1:-r04h}
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/${TAG}_pytest.log; tail -8 $OUT/${TAG}_pytest.log
for C in c3 c5; do
  timeout 600 python bench.py --config $C --steps 100 --warmup 5 --no-cpu-baseline --traffic off > $OUT/${TAG}_bench_$C.json 2> $OUT/${TAG}_bench_$C.err
  python3 -c "
import json
d=json.load(open('$OUT/${TAG}_bench_$C.json')); r=d['roofline']
print('$C ms/step', round(d['ms_per_step'],4), 'samples/s', round(d['value'],4), 'conv', r['conv3x3_ms_per_step'], 'frac', r['frac'], 'at clock', r['frac_at_measured_clock'], r['clock'], 'rank1', r['rank1_ms_per_step'], '1x1', r['conv1x1_ms_per_step'])" || tail -5 $OUT/${TAG}_bench_$C.err
done
timeout 600 python tools/bench_train.py --steps 30 > $OUT/${TAG}_train.json 2>&1; tail -1 $OUT/${TAG}_train.json | cut -c1-300
timeout 600 python tools/bench_ae_train.py > $OUT/${TAG}_ae.json 2>&1; tail -1 $OUT/${TAG}_ae.json | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver.json 2> $OUT/${TAG}_bench_driver.err; tail -c 2500 $OUT/${TAG}_bench_driver.json
