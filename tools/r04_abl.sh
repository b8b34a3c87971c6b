#!/bin/bash
# ablation of the wide 3x3 kernel's k-loop: which ingredient keeps a lone block from filling the matrix pipe?
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
for abl in 0 1 2 4 7; do echo "== W24W_ABL=$abl (1 no weight loads, 2 no halo loads, 4 no patch reads + transform)"; timeout 200 tools/ub_wino24_t_abl$abl 0 4 5 2>&1 | grep -v "wino4 "; done > $OUT/r04b_abl.txt 2>&1
cat $OUT/r04b_abl.txt | cut -c1-330
