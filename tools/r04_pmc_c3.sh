#!/bin/bash
# SQ counters per kernel for config 3 (batch 8) and for the auto-encoder iteration: MFMA busy, wait shares, LDS bank conflicts
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $ROOT/bench.py --config c3 --steps 6 --warmup 2 --no-cpu-baseline --profile-every 0 --traffic off --prewarm 4"
rm -rf /tmp/q4 && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/q4 -o t --output-format csv -- $P > /tmp/q4.log 2>&1
{ echo "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --config c3 --steps 6 --warmup 2   (batch 8, DDIM-100)";
  echo "MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)";
  python3 $ROOT/tools/pmc_sq_summary.py $(find /tmp/q4 -name "*counter_collection.csv" | head -1); } > $OUT/r04_config3_pmc_sq_summary.txt
A="python3 $ROOT/tools/bench_ae_train.py --steps 4"
rm -rf /tmp/q5 && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/q5 -o t --output-format csv -- $A > /tmp/q5.log 2>&1
{ echo "the same counters, python3 tools/bench_ae_train.py --steps 4   (auto-encoder training iteration)";
  python3 $ROOT/tools/pmc_sq_summary.py $(find /tmp/q5 -name "*counter_collection.csv" | head -1); } > $OUT/r04_ae_pmc_sq_summary.txt
grep -v "at::native\|rocclr" $OUT/r04_config3_pmc_sq_summary.txt | head -30
