#!/bin/bash
# one step's kernel timeline of the other configs (c3: batch 8, c5: 256x256x128) and of the training step
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=${1:-tl}
cd /tmp && export TMPDIR=/tmp
for C in c3 c5; do
  rm -rf /tmp/p_$C
  timeout 600 rocprofv3 --kernel-trace -d /tmp/p_$C -o t --output-format csv -- python3 $ROOT/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off > /tmp/p_$C.log 2>&1
  F=$(find /tmp/p_$C -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/tools/trace_timeline.py $F > $OUT/${TAG}_timeline_$C.txt
  python3 $ROOT/tools/prof_summary.py $F 175 > $OUT/${TAG}_summary_$C.txt
done
rm -rf /tmp/p_tr
timeout 600 rocprofv3 --kernel-trace -d /tmp/p_tr -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 20 > /tmp/p_tr.log 2>&1
F=$(find /tmp/p_tr -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/prof_summary.py $F 20 > $OUT/${TAG}_summary_train.txt
tail -3 /tmp/p_tr.log
head -50 $OUT/${TAG}_timeline_c3.txt
