#!/bin/bash
# A/B of the wide 3x3 kernel inside the real steps (one box): bench.py configs and the training step under S3D_WINO24W=0 / default
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; TAG=${1:-r04i}
for C in c3 c5 c2; do for SET in "S3D_WINO24W=0" ""; do
  env $SET timeout 600 python bench.py --config $C --steps 200 --warmup 5 --no-cpu-baseline --traffic off > $OUT/${TAG}_bench_${C}_${SET:-default}.json 2> $OUT/${TAG}_err.txt
  python3 -c "
import json
d=json.load(open('$OUT/${TAG}_bench_${C}_${SET:-default}.json')); r=d['roofline']
print('$C [${SET:-default}] ms/step', round(d['ms_per_step'],4), 'conv', r['conv3x3_ms_per_step'], 'frac', r['frac'], 'at clock', r['frac_at_measured_clock'], (r['clock'] or {}).get('gfxclk_mhz_mean'), (r['clock'] or {}).get('socket_power_w_mean'), 'rank1', r['rank1_ms_per_step'], '1x1', r['conv1x1_ms_per_step'])" || tail -5 $OUT/${TAG}_err.txt
done; done
for SET in "S3D_WINO24W=0" "" "S3D_WINO24W=1"; do echo "train [$SET]"; env $SET timeout 600 python tools/bench_train.py --steps 50 2>&1 | tail -1 | cut -c1-200; done
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver.json 2> $OUT/${TAG}_bench_driver.err; python3 -c "
import json
d=json.load(open('$OUT/${TAG}_bench_driver.json')); r=d['roofline']
print('driver-style ms/step', d['ms_per_step'], 'value', d['value'], 'frac', r['frac'], r['frac_at_measured_clock'], r['clock'], r['traffic'], r['traffic_source'][:40])"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p1
timeout 600 rocprofv3 --kernel-trace -d /tmp/p1 -o t --output-format csv -- python3 $ROOT/bench.py --config c3 --steps 30 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off > /tmp/p1.log 2>&1
python3 $ROOT/tools/prof_summary.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 185 > $OUT/${TAG}_c3_kernel_summary.txt; head -30 $OUT/${TAG}_c3_kernel_summary.txt
