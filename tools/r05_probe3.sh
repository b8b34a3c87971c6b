#!/bin/bash
# round 5, GPU call 3: training tier with the weight gradients on a side stream (A/B), chain tests again, training tests
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r05p3; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests/test_hip_train.py tests/test_hip_chains.py tests/test_hip_edge_cases.py -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest.log; cat $OUT/pytest.log
for S in 1 0 1 0; do
  echo "## S3D_BWD_SIDE=$S"; S3D_BWD_SIDE=$S timeout 600 python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{" | cut -c1-230
done > $OUT/train_ab.txt 2>&1
cat $OUT/train_ab.txt
python3 bench.py --config c4 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_c4.json; cut -c1-400 $OUT/bench_c4.json
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_tr
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 20 --warmup 3 > /tmp/p_tr.log 2>&1
{ grep "^{" /tmp/p_tr.log | cut -c1-200; python3 $ROOT/tools/prof_summary.py $(find /tmp/p_tr -name "*kernel_trace.csv" | head -1) 23; } > $OUT/train_kernel_summary.txt
python3 $ROOT/tools/trace_overlap.py $(find /tmp/p_tr -name "*kernel_trace.csv" | head -1) 0.6 > $OUT/train_overlap.txt 2>&1
head -12 $OUT/train_overlap.txt; head -30 $OUT/train_kernel_summary.txt
