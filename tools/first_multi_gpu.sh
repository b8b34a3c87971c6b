#!/bin/bash
# The first commands for a multi-GPU MI355X node (none has been available to this build: everything multi-GPU has run on gloo
# ranks and on one GPU only — DESIGN.md section 6).  Run from the repo root:
#     tools/first_multi_gpu.sh [MAX_GPUS=8]
# 1. the RCCL twins of the gloo tests (tests/test_multi_gpu.py; they skip below two devices);
# 2. bench.py --gpus N for N = 1, 2, 4, 8: BASELINE configs[1] (c2: independent batch-1 DDPM-1000 samples, no data-path collective),
#    configs[2] (c3: DDIM-100, 8 samples per GPU), configs[3]'s diffusion stage (c4: one flat-gradient all-reduce per step over RCCL,
#    exchange after the backward pass and overlapped with it) — every JSON line under profiles/scale_<config>_n<N>[_overlap].json.
# Nothing here computes an efficiency: the lines carry value / ms_per_step / per-rank times, the reader divides.
set -u
MAXN=${1:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p profiles
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "visible devices: $NDEV"
timeout 3000 python3 -m pytest tests/test_multi_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee profiles/scale_rccl_tests.txt
for N in 1 2 4 8; do
  [ "$N" -gt "$MAXN" ] && break
  [ "$N" -gt "$NDEV" ] && break
  for C in c2 c3 c4; do
    S=200; [ "$C" = c3 ] && S=100
    timeout 1800 python3 bench.py --gpus $N --config $C --steps $S --warmup 5 --no-cpu-baseline --traffic off 2> profiles/scale_${C}_n${N}.err | tail -1 > profiles/scale_${C}_n${N}.json
    python3 -c "import json,sys; d=json.load(open('profiles/scale_${C}_n${N}.json')); print('$C N=$N', 'value', round(d['value'],4), d['unit'], 'ms/step', round(d['ms_per_step'],4), 'per-rank', d['per_rank_ms'])" || tail -3 profiles/scale_${C}_n${N}.err
  done
  if [ "$N" -gt 1 ]; then
    S3D_OVERLAP_ALLREDUCE=1 timeout 1800 python3 bench.py --gpus $N --config c4 --steps 200 --warmup 5 --no-cpu-baseline --traffic off 2> profiles/scale_c4_n${N}_overlap.err | tail -1 > profiles/scale_c4_n${N}_overlap.json
    python3 -c "import json; d=json.load(open('profiles/scale_c4_n${N}_overlap.json')); print('c4 overlapped N=$N', 'value', round(d['value'],4), 'ms/step', round(d['ms_per_step'],4))" || tail -3 profiles/scale_c4_n${N}_overlap.err
  fi
done
find profiles -name "scale_*.err" -size 0 -delete
