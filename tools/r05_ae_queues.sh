#!/bin/bash
# one auto-encoder training iteration's launches per hardware queue (tools/trace_queues.py) -> gpurun_out/ae_queues.txt
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_aeq
timeout 900 rocprofv3 --kernel-trace -d /tmp/p_aeq -o t --output-format csv -- python3 $ROOT/tools/bench_ae_train.py --steps 12 > /tmp/p_aeq.log 2>&1
python3 $ROOT/tools/trace_queues.py $(find /tmp/p_aeq -name "*kernel_trace.csv" | head -1) k_adamw ${1:-0} > $ROOT/gpurun_out/ae_queues.txt 2>&1
tail -12 $ROOT/gpurun_out/ae_queues.txt
