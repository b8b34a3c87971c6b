#!/bin/bash
# round 4, first GPU call: the suite after the experiment removal + the wide 3x3 kernel, its micro-benchmark with three weight-ring
# depths and phase stamps, the clock / power probe, and the headline bench.
#   gpurun --timeout 1500 -- 'tools/r04_wide.sh'
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $OUT/r04a_pytest.log; tail -5 $OUT/r04a_pytest.log
timeout 300 tools/ub_clock > $OUT/r04a_clock.txt 2>&1; cat $OUT/r04a_clock.txt
for v in "" _r6 _r12; do echo "== ring variant '$v'"; timeout 300 tools/ub_wino24$v 2>&1 | grep -v "wino4 " ; done > $OUT/r04a_wino_ubench.txt 2>&1; cat $OUT/r04a_wino_ubench.txt
timeout 300 tools/ub_wino24_t 2>&1 | grep -v "wino4 " > $OUT/r04a_wino_ubench_phases.txt; cat $OUT/r04a_wino_ubench_phases.txt
for SET in "" "S3D_WINO24W=1"; do
  env $SET timeout 600 python bench.py --steps 300 --warmup 5 --no-cpu-baseline > $OUT/r04a_bench_${SET:-default}.json 2> $OUT/r04a_bench_${SET:-default}.err
  tail -c 1500 $OUT/r04a_bench_${SET:-default}.json
done
