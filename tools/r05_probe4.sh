#!/bin/bash
# round 5, GPU call 4: backward DAG (two side streams) A/B, training + option tests
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r05p4; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests/test_hip_train.py tests/test_hip_parity.py -m gpu -q -x -k "side_stream or marks or fast_path or forms_selected or losses_and_grads or optimizer" 2>&1 | tail -8 > $OUT/pytest.log; cat $OUT/pytest.log
for S in 1 2 0 1 2 0; do
  echo "## S3D_BWD_SIDE=$S"; S3D_BWD_SIDE=$S timeout 600 python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{" | cut -c1-160
done > $OUT/train_ab.txt 2>&1
cat $OUT/train_ab.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_tr
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 20 --warmup 3 > /tmp/p_tr.log 2>&1
python3 $ROOT/tools/trace_overlap.py $(find /tmp/p_tr -name "*kernel_trace.csv" | head -1) 0.6 > $OUT/train_overlap.txt 2>&1
head -14 $OUT/train_overlap.txt
