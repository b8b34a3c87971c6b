#!/bin/bash
# k_conv_wino24g (LDS-DMA halo, persistent blocks) against the register-staged kernels, one box: parity, kernel alone, phases, in the step.
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_train.py -m gpu -q -x -k "switched_conv_forms or lds_dma_conv_form" 2>&1 | tail -5 > $OUT/r06_glds_pytest.log
cat $OUT/r06_glds_pytest.log
UB_ONLY=123 timeout 400 tools/ub_wino24 0 1 2 3 4 5 6 17 18 > $OUT/r06_glds_ubench.txt 2>&1
UB_ONLY=13 timeout 300 tools/ub_wino24_t 0 2 3 4 2>&1 | grep -v "per CU" > $OUT/r06_glds_phases.txt
for SET in "" "S3D_WINO24G=1" "" "S3D_WINO24G=1"; do
  env $SET timeout 600 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --traffic off --chains 0 > $OUT/tmp_bench.json 2> $OUT/tmp_bench.err
  python3 - "$OUT/tmp_bench.json" "$SET" <<'PY' | tee -a $OUT/r06_glds_in_step.txt
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(f"[{sys.argv[2] or 'default'}] ms/step {d['ms_per_step']:.4f} samples/s {d['value']:.4f} conv3x3 {r['conv3x3_ms_per_step']} frac {r['frac']} avg_launch_us {r['avg_launch_us']}")
except Exception as e:
    print(f"[{sys.argv[2]}] FAILED {e!r}")
PY
done
for SET in "" "S3D_WINO24G=1"; do
  env $SET timeout 600 python bench.py --config c3 --steps 40 --warmup 5 --no-cpu-baseline --traffic off > $OUT/tmp_bench.json 2> $OUT/tmp_bench.err
  python3 -c "
import json,sys
d=json.load(open('$OUT/tmp_bench.json')); r=d['roofline']; print('[c3 ${SET:-default}] ms/step', round(d['ms_per_step'],4), 'frac', r['frac'], 'avg_launch_us', r['avg_launch_us'])" | tee -a $OUT/r06_glds_in_step.txt
  env $SET timeout 600 python bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline --traffic off > $OUT/tmp_bench.json 2> $OUT/tmp_bench.err
  python3 -c "
import json,sys
d=json.load(open('$OUT/tmp_bench.json')); r=d['roofline']; print('[c4 ${SET:-default}] ms/step', round(d['ms_per_step'],4), 'frac', r['frac'], 'avg_launch_us', r['avg_launch_us'])" | tee -a $OUT/r06_glds_in_step.txt
done
