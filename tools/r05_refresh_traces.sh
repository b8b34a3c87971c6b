#!/bin/bash
# the single-chain traces of tools/refresh_profiles.sh again (the first r05 pass had bench.py's two-chain leg inside them)
R=r05; ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/refresh; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off --chains 0"
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t --output-format csv -- $B > /tmp/p1.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/${R}_kernel_stats.csv
python3 $ROOT/tools/prof_summary.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 215 > $OUT/${R}_kernel_summary.txt
python3 $ROOT/tools/trace_timeline.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) k_out_head 40 > $OUT/${R}_timeline.txt
P="python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-every 0 --traffic off --prewarm 4 --chains 0"
rm -rf /tmp/p2 && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o t --output-format csv -- $P > /tmp/p2.log 2>&1
rm -rf /tmp/p3 && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o t --output-format csv -- $P > /tmp/p3.log 2>&1
python3 $ROOT/tools/pmc_traffic.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $OUT/${R}_pmc_traffic.json k_conv_wino24
rm -rf /tmp/p4 && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p4 -o t --output-format csv -- $P > /tmp/p4.log 2>&1
{ echo "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 6 --warmup 2 --chains 0";
  echo "MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)";
  python3 $ROOT/tools/pmc_sq_summary.py $(find /tmp/p4 -name "*counter_collection.csv" | head -1); } > $OUT/${R}_pmc_sq_summary.txt
cd $ROOT
cp $OUT/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json
python3 bench.py 2>/dev/null | tail -1 > $OUT/${R}_bench.json
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${R}_bench_driver_flags.json
head -12 $OUT/${R}_kernel_summary.txt; tail -4 $OUT/${R}_timeline.txt
# the whole GPU suite and the full-size chain
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $OUT/${R}_pytest_gpu.log; cat $OUT/${R}_pytest_gpu.log
timeout 1200 python3 tools/validate_full_size.py > $OUT/${R}_full_size_parity.txt 2>&1; tail -5 $OUT/${R}_full_size_parity.txt
