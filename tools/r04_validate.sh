#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
{ echo "# tools/validate_full_size.py --steps 1000 --stride 100 on one MI355X with the round-4 kernels (k_conv_wino24w on the 384->128 layer, one library call per step:"; echo "# the output head adds its GroupNorm statistics and applies the sampler update with reference-ordered unfused arithmetic), driven through p_sample_loop_progressive:"; echo "# the whole DDPM-1000 chain of BASELINE configs[1] (128-ch UNet, 128^3, batch 1), HIP path vs the CPU port with identical noise."; timeout 1500 python tools/validate_full_size.py --steps 1000 --stride 100 2>&1 | grep "^step"; } > $OUT/r04_full_size_parity.txt; cat $OUT/r04_full_size_parity.txt
