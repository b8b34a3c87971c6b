#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 600 python tools/find_copies.py > $OUT/r04m_copies.txt 2>&1; grep -i "copy\|memcpy\|Name\|aten::" $OUT/r04m_copies.txt | head -30
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "switched or golden_other or fused" 2>&1 | tail -4
