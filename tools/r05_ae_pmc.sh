#!/bin/bash
# SQ counters per kernel of the auto-encoder iteration, every launch on one stream (S3D_BWD_SIDE=0) -> gpurun_out/ae_pmc.txt
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_aep
export S3D_BWD_SIDE=0
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_aep -o t --output-format csv -- python3 $ROOT/tools/bench_ae_train.py --steps 6 --warmup 2 > /tmp/p_aep.log 2>&1
python3 $ROOT/tools/pmc_sq_summary.py $(find /tmp/p_aep -name "*counter_collection.csv" | head -1) > $ROOT/gpurun_out/ae_pmc.txt 2>&1
grep -E "wgrad|conv_mfma|relu|last" $ROOT/gpurun_out/ae_pmc.txt
