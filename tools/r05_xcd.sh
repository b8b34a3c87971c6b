#!/bin/bash
# round 5, VERDICT r4 item 7: block -> XCD mapping of the mixed Winograd kernels on the batched config (C3: batch 8, DDIM-100).
# default: an XCD owns all output-channel blocks of a contiguous eighth of the pixel tiles; S3D_XCD_MAP=1: one HALF of the channel
# blocks of a QUARTER of the tiles.  Time, HBM traffic of the 3x3 launches (FETCH_SIZE x2 + WRITE_SIZE, separate passes), power.
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r05xcd; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "config3 or unet_forward_golden" 2>&1 | tail -3
for M in 0 1 0 1; do
  S3D_XCD_MAP=$M timeout 600 python3 bench.py --config c3 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/c3_map$M.json
  python3 - $OUT/c3_map$M.json $M <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; c = r.get("clock") or {}
print(f"XCD_MAP={sys.argv[2]}  ms/step {d['ms_per_step']:.4f}  conv3x3 {r['conv3x3_ms_per_step']} ms/step  avg launch {r['avg_launch_us']} us  frac {r['frac']}  clock {c.get('gfxclk_mhz_mean')} MHz  power {c.get('socket_power_w_mean')} W")
PY
done | tee $OUT/times.txt
cd /tmp && export TMPDIR=/tmp
for M in 0 1; do
  P="python3 $ROOT/bench.py --config c3 --steps 6 --warmup 2 --no-cpu-baseline --profile-every 0 --traffic off --prewarm 4"
  rm -rf /tmp/x2 /tmp/x3
  S3D_XCD_MAP=$M rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/x2 -o t --output-format csv -- $P > /tmp/x2.log 2>&1
  S3D_XCD_MAP=$M rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/x3 -o t --output-format csv -- $P > /tmp/x3.log 2>&1
  python3 $ROOT/tools/pmc_traffic.py $(find /tmp/x2 -name "*counter_collection.csv" | head -1) $(find /tmp/x3 -name "*counter_collection.csv" | head -1) $OUT/traffic_map$M.json k_conv_wino24 > /dev/null 2>&1
  python3 - $OUT/traffic_map$M.json $M <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"XCD_MAP={sys.argv[2]}  3x3 traffic per launch {d['dominant_traffic_bytes_per_launch'] / 1e6:.1f} MB;", {k: round(v / 1e6, 1) if isinstance(v, (int, float)) else v for k, v in d.items() if 'per_step' in k or 'dominant' in k})
PY
done | tee $OUT/traffic.txt
