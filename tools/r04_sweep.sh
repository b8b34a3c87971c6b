#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 600 tools/ub_wino24 2>&1 | grep -v "wino4 " > $OUT/r04c_sweep.txt; cat $OUT/r04c_sweep.txt
