#!/bin/bash
# Regenerate the measured artefacts behind DESIGN.md on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'tools/refresh_profiles.sh r06'
# Everything lands in gpurun_out/refresh/ as <round>_*; copy what should be judged into profiles/.
R=${1:-r06}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/refresh
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off --chains 0"
# 1. kernel trace + stats, one step's timeline
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t --output-format csv -- $B > /tmp/p1.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/${R}_kernel_stats.csv
python3 $ROOT/tools/prof_summary.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 215 > $OUT/${R}_kernel_summary.txt
python3 $ROOT/tools/trace_timeline.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) > $OUT/${R}_timeline.txt
# 2. HBM traffic: two PMC passes (counters only with --kernel-trace)
P="python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-every 0 --traffic off --prewarm 4 --chains 0"
rm -rf /tmp/p2 && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o t --output-format csv -- $P > /tmp/p2.log 2>&1
rm -rf /tmp/p3 && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o t --output-format csv -- $P > /tmp/p3.log 2>&1
python3 $ROOT/tools/pmc_traffic.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $OUT/${R}_pmc_traffic.json k_conv_wino24
# 3. SQ counters
rm -rf /tmp/p4 && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p4 -o t --output-format csv -- $P > /tmp/p4.log 2>&1
{ echo "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 6 --warmup 2";
  echo "MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)";
  python3 $ROOT/tools/pmc_sq_summary.py $(find /tmp/p4 -name "*counter_collection.csv" | head -1); } > $OUT/${R}_pmc_sq_summary.txt
rm -rf /tmp/p5 && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/p5 -o t --output-format csv -- $P > /tmp/p5.log 2>&1
python3 - $(find /tmp/p5 -name "*counter_collection.csv" | head -1) >> $OUT/${R}_pmc_sq_summary.txt <<'PY'
import collections, csv, sys
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("s3d::", "").replace("void ", "").split("(")[0][:70]
    agg.setdefault(n, collections.Counter())[r["Counter_Name"]] += float(r["Counter_Value"])
print("LDS bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE):")
for n, c in agg.items():
    if c["SQ_LDS_IDX_ACTIVE"] > 0 and "conv" in n:
        print(f"  {n:60s} {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}")
PY
cd $ROOT
# 4. the headline line (bench.py measures roofline.traffic itself with two rocprofv3 --pmc children; the committed
#    profiles/<round>_pmc_traffic.json is only its fallback)
cp $OUT/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json
python3 bench.py 2>/dev/null | tail -1 > $OUT/${R}_bench.json
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${R}_bench_driver_flags.json
# 5. the other BASELINE configs through bench.py (each line with its own roofline), the small ones through tools/bench_configs.py,
#    training, decode
{ for C in c3 c5 c4; do python3 bench.py --config $C --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1; done
  python3 tools/bench_configs.py "C1 " 2>/dev/null | grep "^{"; python3 tools/bench_configs.py "64-ch" 2>/dev/null | grep "^{"; } > $OUT/${R}_other_configs.txt
{ echo "## default (weight gradients on the backward pass's side stream)"; python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{"; python3 tools/bench_ae_train.py 2>/dev/null | grep "^{";
  echo "## S3D_BWD_SIDE=0 (everything in line: round 4's launch structure)"; S3D_BWD_SIDE=0 python3 tools/bench_train.py --steps 100 --warmup 10 2>/dev/null | grep "^{"; S3D_BWD_SIDE=0 python3 tools/bench_ae_train.py 2>/dev/null | grep "^{";
  echo "## bench.py --config c4, S3D_BWD_SIDE=0: the 3x3 weight-gradient kernel alone (roofline.wgrad3x3)"; S3D_BWD_SIDE=0 python3 bench.py --config c4 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1; } > $OUT/${R}_train_step.txt
{ python3 tools/bench_decode.py 2>/dev/null | grep -v amdgpu.ids; python3 tools/bench_decode.py --reso 512 --hwd 256 256 128 --aabb-scale 2 2 1 2>/dev/null | grep -v amdgpu.ids;
  python3 tools/bench_config5.py 2>/dev/null | grep "^{"; python3 tools/bench_end_to_end.py 2>/dev/null | grep "^{"; } > $OUT/${R}_decode_and_isosurface.txt
# 6. the 3x3 kernels alone (steady state), with phase stamps and one / two / three blocks per CU; clock + power under them
[ -x tools/ub_wino24 ] && timeout 400 tools/ub_wino24 2>&1 | grep -v "wino4 " > $OUT/${R}_wino_ubench.txt
[ -x tools/ub_wino24_t ] && timeout 300 tools/ub_wino24_t 0 2 3 4 5 2>&1 | grep -v "wino4 " > $OUT/${R}_wino_ubench_phases.txt
[ -x tools/ub_clock ] && timeout 300 tools/ub_clock > $OUT/${R}_clock.txt 2>&1
# 7. config 3's kernel table
cd /tmp && rm -rf /tmp/p6 && rocprofv3 --kernel-trace -d /tmp/p6 -o t --output-format csv -- python3 $ROOT/bench.py --config c3 --steps 30 --warmup 5 --no-cpu-baseline --profile-every 0 --traffic off > /tmp/p6.log 2>&1
python3 $ROOT/tools/prof_summary.py $(find /tmp/p6 -name "*kernel_trace.csv" | head -1) 185 > $OUT/${R}_config3_kernel_summary.txt
# 8. the training step's kernel table, timeline and stream overlap (the only kernel table of the training step used to live in scratch)
cd /tmp && rm -rf /tmp/p7 && rocprofv3 --kernel-trace --stats -d /tmp/p7 -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 20 --warmup 3 > /tmp/p7.log 2>&1
{ grep "^{" /tmp/p7.log | cut -c1-220; python3 $ROOT/tools/prof_summary.py $(find /tmp/p7 -name "*kernel_trace.csv" | head -1) 23;
  echo; echo "(calls/step of __amd_rocclr_copyBuffer counts the parameter uploads at model load — 23 traced steps; inside a step: 2, the timestep / weight vectors)";
  echo; python3 $ROOT/tools/trace_overlap.py $(find /tmp/p7 -name "*kernel_trace.csv" | head -1) 0.6; } > $OUT/${R}_train_kernel_summary.txt
python3 $ROOT/tools/trace_timeline.py $(find /tmp/p7 -name "*kernel_trace.csv" | head -1) k_adamw > $OUT/${R}_train_timeline.txt 2>&1
rm -rf /tmp/p8 && S3D_BWD_SIDE=0 rocprofv3 --kernel-trace --stats -d /tmp/p8 -o t --output-format csv -- python3 $ROOT/tools/bench_train.py --steps 20 --warmup 3 > /tmp/p8.log 2>&1
{ echo "S3D_BWD_SIDE=0 (every launch on one stream: per-kernel times without sharing the chip)"; grep "^{" /tmp/p8.log | cut -c1-220; python3 $ROOT/tools/prof_summary.py $(find /tmp/p8 -name "*kernel_trace.csv" | head -1) 23; } > $OUT/${R}_train_kernel_summary_inline.txt
cd $ROOT
# 9. RCCL at world size 1: the same command with and without a forced process group, c4 (every step then contains
#    a real RCCL all-reduce of the 28-MB flat gradient + the start broadcast) and c2 (barrier / MAX all-reduce / object gather only)
{ echo "# bench.py --gpus 1 [--force-dist]: ms_per_step, rccl_world_size, dist_backend (same box, alternating)";
  for rep in 1 2; do for F in "" "--force-dist"; do for C in c4 c2; do
    python3 bench.py --gpus 1 --config $C $F --steps $([ $C == c4 ] && echo 200 || echo 300) --warmup 10 --no-cpu-baseline --traffic off --overlap off --chains 0 --profile-every 0 2>/dev/null | grep "^{" | tail -1 |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$C', '${F:-no process group}', 'ms_per_step', round(d['ms_per_step'],4), 'rccl_world_size', d['rccl_world_size'], 'dist_backend', d['dist_backend'])"
  done; done; done; } > $OUT/${R}_rccl_world1.txt
# 10. the 3x3 weight-gradient kernel alone: time, per-region stamps, SQ counters
[ -x tools/ub_wgrad ] && timeout 300 tools/ub_wgrad > $OUT/${R}_wgrad_ubench.txt 2>&1
[ -x tools/ub_wgrad_t ] && timeout 300 tools/ub_wgrad_t 2>&1 | cut -c1-420 > $OUT/${R}_wgrad_phases.txt
if [ -x tools/ub_wgrad ]; then
  cd /tmp && rm -rf /tmp/p10 && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p10 -o t --output-format csv -- $ROOT/tools/ub_wgrad > /tmp/p10.log 2>&1
  { echo "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -- tools/ub_wgrad (all six shapes together)";
    python3 $ROOT/tools/pmc_sq_summary.py $(find /tmp/p10 -name "*counter_collection.csv" | head -1); } > $OUT/${R}_wgrad_pmc.txt
  rm -rf /tmp/p11 && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/p11 -o t --output-format csv -- $ROOT/tools/ub_wgrad > /tmp/p11.log 2>&1
  python3 - $(find /tmp/p11 -name "*counter_collection.csv" | head -1) >> $OUT/${R}_wgrad_pmc.txt <<'PY'
import collections, csv, sys
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("s3d::", "").replace("void ", "").split("(")[0][:70]
    agg.setdefault(n, collections.Counter())[r["Counter_Name"]] += float(r["Counter_Value"])
print("LDS bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE):")
for n, c in agg.items():
    if c["SQ_LDS_IDX_ACTIVE"] > 0:
        print(f"  {n:60s} {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}")
PY
  cd $ROOT
fi
ls -la $OUT
