#!/bin/bash
# A/B on one GPU box: parity subset, then the headline bench under each environment setting, then a kernel timeline of the
# default build.   gpurun --timeout 1500 -- 'tools/gpu_ab.sh TAG "pytest -k expr" "ENV1=.. ENV2=.." "ENV=.." ...'
# An empty setting ("") is the default configuration.  Everything lands in gpurun_out/TAG_*.
TAG=${1:-ab}; KEXPR=${2:-"unet_forward_golden or leaf or conv_edge or rank1_inline"}; shift 2
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_edge_cases.py -m gpu -q -x -k "$KEXPR" 2>&1 | tail -25 > $OUT/${TAG}_pytest.log
tail -5 $OUT/${TAG}_pytest.log
i=0
for SET in "$@"; do
  env $SET timeout 600 python bench.py --steps 300 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_$i.json 2> $OUT/${TAG}_bench_$i.err
  python3 - "$OUT/${TAG}_bench_$i.json" "$SET" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(f"[{sys.argv[2] or 'default'}] ms/step {d['ms_per_step']:.4f} samples/s {d['value']:.4f} conv3x3 {r['conv3x3_ms_per_step']} frac {r['frac']} rank1 {r['rank1_ms_per_step']} 1x1 {r['conv1x1_ms_per_step']}")
except Exception as e:
    print(f"[{sys.argv[2]}] FAILED {e!r}")
PY
  i=$((i+1))
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p1
timeout 600 rocprofv3 --kernel-trace -d /tmp/p1 -o t --output-format csv -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-every 0 > /tmp/p1.log 2>&1
F=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_timeline.py $F > $OUT/${TAG}_timeline.txt
python3 $ROOT/tools/prof_summary.py $F 185 > $OUT/${TAG}_kernel_summary.txt
head -60 $OUT/${TAG}_timeline.txt
