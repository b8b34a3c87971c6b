#!/bin/bash
# rocprofv3 kernel trace of a short bench run, summarised per kernel (run on the GPU box via gpurun).
#   tools/prof_bench.sh [steps]  -> gpurun_out/prof_summary.txt
S=${1:-60}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace -d /tmp/prof -o trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps $S --warmup 5 --no-cpu-baseline --profile-every 0 > /tmp/prof_bench.log 2>&1
F=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $F $((S+5)) > $GRAFT_REPO_ROOT/gpurun_out/prof_summary.txt
cat $GRAFT_REPO_ROOT/gpurun_out/prof_summary.txt
