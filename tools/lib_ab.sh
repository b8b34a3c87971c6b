#!/bin/bash
# same-box A/B of two builds of the library: tools/lib_ab.sh "<command>" ab/lib_a.so ab/lib_b.so ...   (each run three times, interleaved)
CMD=$1; shift
ROOT=$GRAFT_REPO_ROOT; cd $ROOT
cp sin3dm_amd/libsin3dm_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do
  for L in "$@"; do
    cp $L sin3dm_amd/libsin3dm_hip.so
    echo "[$L] $(bash -c "$CMD" 2>/dev/null | grep '^{' | tail -1 | cut -c1-${LIB_AB_CUT:-260})"
  done
done
cp /tmp/lib_keep.so sin3dm_amd/libsin3dm_hip.so
