#!/bin/bash
# ONE parameterised same-box A/B driver (replaces the 35 single-use tools/r04_*.sh / r05_*.sh / r06_*.sh scripts of rounds 4-6, which
# live in the git history).  Everything runs inside one gpurun call = one box, one clock state; settings are interleaved REPS times.
#
#   gpurun --timeout 1500 -- 'tools/ab.sh TAG [options] -- "ENV=.. ENV2=.." "ENV=.." ...'      ("" = the default configuration)
#
# options (before --):
#   -c CONFIG     bench.py --config (c2 headline | c3 batch 8 | c4 training step | c5 big planes); default c2
#   -s STEPS      timed steps (default 300; c3 40, c4 100, c5 40)
#   -r REPS       repeats per setting, interleaved (default 2)
#   -k PATTERN    also: per-kernel average durations (rocprofv3 --kernel-trace) of kernels matching PATTERN, per setting
#   -t EXPR       first: pytest -m gpu -k EXPR over tests/ (parity gate of the forms under test)
#   -u "ARGS"     first: tools/ub_wino24 ARGS (the 3x3 kernels alone; UB_ONLY etc. from the environment) and tools/ub_wino24_t ARGS
#                 (phase stamps) — build both before the call (hipcc lines at the top of tools/wino24_ubench.hip)
#   -T            last: one step's kernel timeline + per-kernel summary of the default configuration
# Output: gpurun_out/TAG_*.txt (and on stdout).
TAG=${1:-ab}; shift
CONFIG=c2; STEPS=""; REPS=2; PATTERN=""; TESTS=""; UB=""; TIMELINE=0
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case $1 in -c) CONFIG=$2; shift 2;; -s) STEPS=$2; shift 2;; -r) REPS=$2; shift 2;; -k) PATTERN=$2; shift 2;; -t) TESTS=$2; shift 2;;
             -u) UB=$2; shift 2;; -T) TIMELINE=1; shift;; *) echo "unknown option $1"; exit 2;; esac
done
[ "$1" == "--" ] && shift
[ $# -eq 0 ] && set -- ""
if [ -z "$STEPS" ]; then case $CONFIG in c2) STEPS=300;; c4) STEPS=100;; *) STEPS=40;; esac; fi
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest tests -m gpu -q -x -k "$TESTS" 2>&1 | tail -6 | tee $OUT/${TAG}_pytest.log; fi
if [ -n "$UB" ]; then
  timeout 600 tools/ub_wino24 $UB > $OUT/${TAG}_ubench.txt 2>&1; grep "^wino\|MISMATCH" $OUT/${TAG}_ubench.txt | cut -c1-150
  timeout 400 tools/ub_wino24_t $UB 2>&1 | grep -v "block(s) per CU" | cut -c1-460 > $OUT/${TAG}_phases.txt
fi
: > $OUT/${TAG}_bench.txt
for rep in $(seq $REPS); do for SET in "$@"; do
  env $SET timeout 900 python bench.py --config $CONFIG --steps $STEPS --warmup 5 --no-cpu-baseline --traffic off --overlap off --chains 0 > $OUT/tmp_ab.json 2> $OUT/tmp_ab.err
  python3 - "$OUT/tmp_ab.json" "$CONFIG" "$SET" <<'PY' | tee -a $OUT/${TAG}_bench.txt
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    w = r.get("wgrad3x3")
    print(f"[{sys.argv[2]} {sys.argv[3] or 'default'}] ms/step {d['ms_per_step']:.4f} value {d['value']:.4f} conv3x3 {r['conv3x3_ms_per_step']} frac {r['frac']} "
          f"avg_launch_us {r['avg_launch_us']} rank1 {r['rank1_ms_per_step']} 1x1 {r['conv1x1_ms_per_step']} whole_step_mfma_frac {r['whole_step_mfma_frac']}"
          + (f" wgrad_frac {w['frac']}" if w else ""))
except Exception as e:
    print(f"[{sys.argv[2]} {sys.argv[3]}] FAILED {e!r}")
PY
done; done
if [ -n "$PATTERN" ]; then
  for SET in "$@"; do
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/abk
    env $SET timeout 600 rocprofv3 --kernel-trace -d /tmp/abk -o t --output-format csv -- python3 $ROOT/bench.py --config $CONFIG --steps 30 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off --overlap off --chains 0 > /tmp/abk.log 2>&1
    echo "## [${SET:-default}] kernels matching '$PATTERN'" | tee -a $OUT/${TAG}_kernels.txt
    python3 $ROOT/tools/prof_summary.py $(find /tmp/abk -name "*kernel_trace.csv" | head -1) 35 | grep -i "$PATTERN" | cut -c1-170 | tee -a $OUT/${TAG}_kernels.txt
    cd $ROOT
  done
fi
if [ $TIMELINE == 1 ]; then
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/abt
  timeout 600 rocprofv3 --kernel-trace -d /tmp/abt -o t --output-format csv -- python3 $ROOT/bench.py --config $CONFIG --steps 30 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off --overlap off --chains 0 > /tmp/abt.log 2>&1
  F=$(find /tmp/abt -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/tools/trace_timeline.py $F $([ $CONFIG == c4 ] && echo k_adamw) > $OUT/${TAG}_timeline.txt 2>&1
  python3 $ROOT/tools/prof_summary.py $F 35 > $OUT/${TAG}_kernel_summary.txt
  head -50 $OUT/${TAG}_timeline.txt; tail -2 $OUT/${TAG}_timeline.txt
fi
