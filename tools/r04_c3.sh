#!/bin/bash
# parity subset, then config 3 (batch 8) and the headline through bench.py, then their kernel timelines
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=${1:-c3}; KEXPR=${2:-"golden or leaf or fused or sampler or head"}
cd $ROOT
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "$KEXPR" 2>&1 | tail -5
for C in c2 c3; do
  timeout 600 python bench.py --config $C --steps 200 --warmup 5 --no-cpu-baseline --traffic off > $OUT/${TAG}_bench_$C.json 2> $OUT/${TAG}_bench_$C.err
  python3 -c "
import json
d=json.load(open('$OUT/${TAG}_bench_$C.json')); r=d['roofline']
print('$C ms/step', round(d['ms_per_step'],4), 'samples/s', round(d['value'],4), 'conv', r['conv3x3_ms_per_step'], 'frac', r['frac'], 'rank1', r['rank1_ms_per_step'], '1x1', r['conv1x1_ms_per_step'])" || tail -5 $OUT/${TAG}_bench_$C.err
done
cd /tmp && export TMPDIR=/tmp
for C in c3 c2; do
rm -rf /tmp/p_$C
timeout 600 rocprofv3 --kernel-trace -d /tmp/p_$C -o t --output-format csv -- python3 $ROOT/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off > /tmp/p_$C.log 2>&1
F=$(find /tmp/p_$C -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_timeline.py $F > $OUT/${TAG}_timeline_$C.txt
done
grep -v "wino24\|k_gn_act \|means_fin" $OUT/${TAG}_timeline_c3.txt | head -40
grep "out_head\|step span" $OUT/${TAG}_timeline_c2.txt
