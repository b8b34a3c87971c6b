#!/bin/bash
# round 5, GPU call 2: chain tests, one-handle lanes probe, chains of batches, bench with chains2, averaged timeline
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r05p2; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_hip_chains.py -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest_chains.log; cat $OUT/pytest_chains.log
{ echo "## one handle, lanes"; timeout 600 python3 tools/two_chain_probe.py --one-handle --chains 1 2 3 --batches 2 4 --chains-batched 2 2>&1 | grep "^{"; } > $OUT/two_chains_lanes.txt 2>&1
python3 - $OUT/two_chains_lanes.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('#'): print(l.strip()); continue
    d = json.loads(l)
    print(d['chains'], d['batch_per_chain'], d['ms_per_sample_step'], d['per_chain_step_latency_ms'][0], d['host_issue_ms_per_step'], d.get('vs_one_chain'))
PY
timeout 900 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python3 - $OUT/bench_driver_flags.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"]); print("chains2", d["chains2"]); print(d["data"]); print(d["roofline"]["traffic_source"])
PY
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p1
timeout 600 rocprofv3 --kernel-trace -d /tmp/p1 -o t --output-format csv -- python3 $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off --chains 0 > /tmp/p1.log 2>&1
python3 $ROOT/tools/trace_timeline.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) k_out_head 40 > $OUT/timeline_avg.txt 2>&1
tail -60 $OUT/timeline_avg.txt
