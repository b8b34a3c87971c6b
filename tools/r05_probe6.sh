#!/bin/bash
# round 5, GPU call: k_gn_bwd_fin folded into the apply launch (tests + A/B vs the committed .so is not possible: timing only), CU-mask experiment, full-size chain
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r05p6; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests/test_hip_train.py tests/test_hip_ae.py -m gpu -q -x 2>&1 | tail -4 > $OUT/pytest.log; cat $OUT/pytest.log
for M in "" 55555555 0000ffff ffff0000 "" ; do
  echo "## S3D_BWD_SIDE_MASK=$M"; S3D_BWD_SIDE_MASK=$M timeout 600 python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{" | cut -c1-160
done > $OUT/train_mask.txt 2>&1
{ echo "## S3D_BWD_SIDE=0"; S3D_BWD_SIDE=0 timeout 600 python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{" | cut -c1-160; python3 tools/bench_ae_train.py 2>/dev/null | grep "^{" | cut -c1-160; } >> $OUT/train_mask.txt
cat $OUT/train_mask.txt
timeout 1500 python3 tools/validate_full_size.py --steps 1000 --stride 100 > $OUT/r05_full_size_parity.txt 2>&1; tail -4 $OUT/r05_full_size_parity.txt
