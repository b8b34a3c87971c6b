#!/bin/bash
# quick GPU check of the TRAINING tier while iterating: gradient parity tests, training step time, per-kernel summary.
#   gpurun --timeout 1500 -- 'tools/gpu_quick_train.sh TAG'
TAG=${1:-train}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_hip_train.py -m gpu -q -x 2>&1 | tail -6 > $OUT/${TAG}_pytest.log
python tools/bench_train.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pt
rocprofv3 --kernel-trace -d /tmp/pt -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 20 --warmup 3 > /tmp/pt.log 2>&1
python3 $ROOT/tools/prof_summary.py $(find /tmp/pt -name "*kernel_trace.csv" | head -1) 23 > $OUT/${TAG}_summary.txt
python3 $ROOT/tools/trace_timeline.py $(find /tmp/pt -name "*kernel_trace.csv" | head -1) k_adamw_ema > $OUT/${TAG}_timeline.txt
tail -3 $OUT/${TAG}_pytest.log
cat $OUT/${TAG}_bench.json
head -24 $OUT/${TAG}_summary.txt
