#!/bin/bash
# round 5, GPU call 5: single side stream (final form) A/B for the denoiser and the auto-encoder, config 5 end to end, AE + train tests
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r05p5; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests/test_hip_train.py tests/test_hip_ae.py -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest.log; cat $OUT/pytest.log
for S in 1 0 1 0; do
  echo "## S3D_BWD_SIDE=$S"; S3D_BWD_SIDE=$S timeout 600 python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{" | cut -c1-160
  S3D_BWD_SIDE=$S timeout 600 python3 tools/bench_ae_train.py 2>/dev/null | grep "^{" | cut -c1-200
done > $OUT/train_ab.txt 2>&1
cat $OUT/train_ab.txt
timeout 900 python3 tools/bench_config5.py > $OUT/config5.json 2> $OUT/config5.err; cat $OUT/config5.json; tail -2 $OUT/config5.err
timeout 600 python3 tools/bench_decode.py --reso 512 --hwd 256 256 128 --aabb-scale 2 2 1 2>/dev/null | grep "^{" > $OUT/decode_512.json; cat $OUT/decode_512.json
