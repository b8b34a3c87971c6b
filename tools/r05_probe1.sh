#!/bin/bash
# round 5, GPU call 1: two-chain probe, the per-step copyBuffer (item 4), training-step kernel table (item 2a)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r05p1; mkdir -p $OUT; cd $ROOT
{ echo "## default queues"; timeout 600 python3 tools/two_chain_probe.py 2>&1 | grep "^{";
  echo "## GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 timeout 600 python3 tools/two_chain_probe.py --chains 1 2 --batches 2>&1 | grep "^{";
  echo "## GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 600 python3 tools/two_chain_probe.py --chains 1 2 4 --batches 2>&1 | grep "^{";
  echo "## no stagger"; timeout 600 python3 tools/two_chain_probe.py --chains 1 2 --batches --stagger 0 2>&1 | grep "^{";
} > $OUT/two_chains.txt 2>&1
cat $OUT/two_chains.txt
cd /tmp && export TMPDIR=/tmp
for N in 1 2; do
  rm -rf /tmp/tc$N && timeout 600 rocprofv3 --kernel-trace -d /tmp/tc$N -o t --output-format csv -- python3 $ROOT/tools/two_chain_probe.py --chains $N --trace-steps 60 > /tmp/tc$N.log 2>&1
  F=$(find /tmp/tc$N -name "*kernel_trace.csv" | head -1)
  { echo "## $N chain(s), 60 steps each, rocprofv3 --kernel-trace"; grep "^{" /tmp/tc$N.log; python3 $ROOT/tools/trace_overlap.py $F 0.4; } > $OUT/overlap_$N.txt 2>&1
  cat $OUT/overlap_$N.txt
  head -1 $F > $OUT/trace_head_$N.csv
done
# the per-step copy
rm -rf /tmp/mc && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d /tmp/mc -o t --output-format csv -- python3 $ROOT/bench.py --steps 30 --warmup 5 --prewarm 20 --no-cpu-baseline --profile-every 0 --traffic off > /tmp/mc.log 2>&1
ls /tmp/mc/* > $OUT/mc_files.txt 2>&1
F=$(find /tmp/mc -name "*memory_copy_trace.csv" | head -1)
[ -n "$F" ] && { head -1 $F; tail -40 $F; wc -l $F; } > $OUT/memcopy.txt
F=$(find /tmp/mc -name "*hip_api_trace.csv" | head -1)
[ -n "$F" ] && { head -1 $F; cut -d, -f1-4 $F | awk -F, '{c[$3]++} END {for (k in c) print c[k], k}' | sort -rn | head -30; } > $OUT/hipapi.txt
F=$(find /tmp/mc -name "*kernel_trace.csv" | head -1)
python3 - $F > $OUT/copy_context.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = 0
for i, r in enumerate(rows):
    if "copyBuffer" in r["Kernel_Name"] and i > len(rows) // 2:
        for j in range(max(0, i - 2), min(len(rows), i + 3)):
            q = rows[j]
            print(("-> " if j == i else "   ") + f"{(int(q['Start_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3:9.1f} us dur {(int(q['End_Timestamp']) - int(q['Start_Timestamp'])) / 1e3:6.1f} q={q.get('Queue_Id')} st={q.get('Stream_Id')} grid={q['Grid_Size_X']} {re.sub(r'^void |s3d::', '', q['Kernel_Name'])[:60]}")
        print()
        n += 1
        if n >= 4: break
PY
cat $OUT/memcopy.txt $OUT/hipapi.txt $OUT/copy_context.txt
# training step kernel table
cd /tmp; rm -rf /tmp/p_tr
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 20 --warmup 3 > /tmp/p_tr.log 2>&1
{ grep "^{" /tmp/p_tr.log | cut -c1-200; python3 $ROOT/tools/prof_summary.py $(find /tmp/p_tr -name "*kernel_trace.csv" | head -1) 23; } > $OUT/train_kernel_summary.txt
python3 $ROOT/tools/trace_timeline.py $(find /tmp/p_tr -name "*kernel_trace.csv" | head -1) k_adamw > $OUT/train_timeline.txt 2>&1
cat $OUT/train_kernel_summary.txt
