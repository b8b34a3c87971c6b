#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_tr
timeout 600 rocprofv3 --kernel-trace -d /tmp/p_tr -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 12 > /tmp/p_tr.log 2>&1
F=$(find /tmp/p_tr -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_timeline.py $F k_adamw_ema > $OUT/train_timeline.txt
tail -2 /tmp/p_tr.log | cut -c1-200
wc -l $OUT/train_timeline.txt
