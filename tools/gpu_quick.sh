#!/bin/bash
# quick GPU check used while iterating on kernels: parity subset, headline bench, one step's kernel timeline.
#   gpurun --timeout 1500 -- 'tools/gpu_quick.sh TAG [pytest -k expr]'
TAG=${1:-quick}
KEXPR=${2:-"unet_forward or leaf or sampler_steps or trajectories or conv_edge"}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_hip_parity.py tests/test_hip_edge_cases.py -m gpu -q -x -k "$KEXPR" 2>&1 | tail -15 > $OUT/${TAG}_pytest.log
python bench.py --steps 300 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p1
rocprofv3 --kernel-trace -d /tmp/p1 -o t --output-format csv -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-every 0 > /tmp/p1.log 2>&1
python3 $ROOT/tools/trace_timeline.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_timeline.txt
python3 $ROOT/tools/prof_summary.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 185 > $OUT/${TAG}_kernel_summary.txt
tail -4 $OUT/${TAG}_pytest.log
python3 -c "
import json,sys
d=json.load(open('$OUT/${TAG}_bench.json'))
print('ms/step', round(d['ms_per_step'],4), 'samples/s', round(d['value'],4), 'conv', d['roofline']['conv3x3_ms_per_step'], 'frac', d['roofline']['frac'], 'rank1', d['roofline']['rank1_ms_per_step'], '1x1', d['roofline']['conv1x1_ms_per_step'])"
