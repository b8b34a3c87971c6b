#!/bin/bash
# full GPU suite + headline bench + step timeline
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; TAG=${1:-r04d}
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $OUT/${TAG}_pytest.log; tail -12 $OUT/${TAG}_pytest.log
for SET in "" ; do
  env $SET timeout 600 python bench.py --steps 300 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
  python3 -c "
import json
d=json.load(open('$OUT/${TAG}_bench.json')); r=d['roofline']
print('ms/step', round(d['ms_per_step'],4), 'samples/s', round(d['value'],4), 'conv', r['conv3x3_ms_per_step'], 'frac', r['frac'], 'rank1', r['rank1_ms_per_step'], '1x1', r['conv1x1_ms_per_step'], r['kernel'][-80:])" || tail -5 $OUT/${TAG}_bench.err
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p1
timeout 600 rocprofv3 --kernel-trace -d /tmp/p1 -o t --output-format csv -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-every 0 > /tmp/p1.log 2>&1
F=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_timeline.py $F > $OUT/${TAG}_timeline.txt
python3 $ROOT/tools/prof_summary.py $F 185 > $OUT/${TAG}_kernel_summary.txt
cat $OUT/${TAG}_timeline.txt | head -60
