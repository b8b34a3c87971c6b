#!/bin/bash
# the training-tier artefacts of tools/refresh_profiles.sh again (after the round's later training changes)
R=r05; ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/refresh; mkdir -p $OUT; cd $ROOT
{ echo "## default (weight gradients on the backward pass's side stream, GroupNorm-backward sums from the dgrad epilogue; the auto-encoder's two nets as two chains)"; python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{"; python3 tools/bench_ae_train.py 2>/dev/null | grep "^{";
  echo "## S3D_BWD_SIDE=0 (every launch on the caller's stream)"; S3D_BWD_SIDE=0 python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{"; S3D_BWD_SIDE=0 python3 tools/bench_ae_train.py 2>/dev/null | grep "^{";
  echo "## S3D_GNB_FUSED=0 (the GroupNorm backward's sums from their own read pass)"; S3D_GNB_FUSED=0 python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{";
  echo "## S3D_BWD_SIDE=0 S3D_GNB_FUSED=0 (round 4's launch structure with round 5's glue removal)"; S3D_BWD_SIDE=0 S3D_GNB_FUSED=0 python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{";
  echo "## bench.py --config c4"; python3 bench.py --config c4 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1;
  echo "## bench.py --config c4, S3D_BWD_SIDE=0: the 3x3 weight-gradient kernel alone (roofline.wgrad3x3)"; S3D_BWD_SIDE=0 python3 bench.py --config c4 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1; } > $OUT/${R}_train_step.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p7 && rocprofv3 --kernel-trace --stats -d /tmp/p7 -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 20 --warmup 3 > /tmp/p7.log 2>&1
{ grep "^{" /tmp/p7.log | cut -c1-220; python3 $ROOT/tools/prof_summary.py $(find /tmp/p7 -name "*kernel_trace.csv" | head -1) 23;
  echo; echo "(calls/step of __amd_rocclr_copyBuffer counts the parameter uploads at model load — 23 traced steps; inside a step: 2, the timestep / weight vectors)";
  echo; python3 $ROOT/tools/trace_overlap.py $(find /tmp/p7 -name "*kernel_trace.csv" | head -1) 0.6; } > $OUT/${R}_train_kernel_summary.txt
python3 $ROOT/tools/trace_timeline.py $(find /tmp/p7 -name "*kernel_trace.csv" | head -1) k_adamw > $OUT/${R}_train_timeline.txt 2>&1
rm -rf /tmp/p8 && S3D_BWD_SIDE=0 rocprofv3 --kernel-trace --stats -d /tmp/p8 -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 20 --warmup 3 > /tmp/p8.log 2>&1
{ echo "S3D_BWD_SIDE=0 (every launch on one stream: per-kernel times without sharing the chip)"; grep "^{" /tmp/p8.log | cut -c1-220; python3 $ROOT/tools/prof_summary.py $(find /tmp/p8 -name "*kernel_trace.csv" | head -1) 23; } > $OUT/${R}_train_kernel_summary_inline.txt
rm -rf /tmp/p9 && rocprofv3 --kernel-trace --stats -d /tmp/p9 -o t --output-format csv -- python3 $ROOT/tools/bench_ae_train.py > /tmp/p9.log 2>&1
{ grep "^{" /tmp/p9.log | cut -c1-220; python3 $ROOT/tools/prof_summary.py $(find /tmp/p9 -name "*kernel_trace.csv" | head -1) 23; echo; python3 $ROOT/tools/trace_overlap.py $(find /tmp/p9 -name "*kernel_trace.csv" | head -1) 0.6; } > $OUT/${R}_ae_train_kernel_summary.txt
cd $ROOT; cut -c1-260 $OUT/${R}_train_step.txt
