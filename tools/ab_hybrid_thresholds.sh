#!/bin/bash
# usage (GPU box): tools/ab_hybrid_thresholds.sh -- where the hybrid forms start to pay: 16-bit groups (RSORT_HYBRID=1) and the narrow form (=2) against the plain loop (=0), 4- and 8-byte keys, pairs
cd $GRAFT_REPO_ROOT
L=radix-sort_amd/lib/librsort_hip.so
for lg in 25 26 27 28; do
  for h in 0 1; do
    echo -n "u64 2^$lg RSORT_HYBRID=$h: "
    RSORT_HYBRID=$h timeout -k 10 120 tools/rs_tune $L 8 $lg random 5 2>&1 | tail -1 | grep -o "total [0-9.]* ms" || exit 1
  done
done
for lg in 26 27 28; do
  for h in 0 1; do
    echo -n "u64+u32 pairs 2^$lg RSORT_HYBRID=$h: "
    RSORT_HYBRID=$h timeout -k 10 120 tools/rs_tune $L 8 $lg random 5 4 2>&1 | tail -1 | grep -o "total [0-9.]* ms" || exit 1
  done
done
for kb in 4 8; do
for n in 3000000 5000000 8000000 16777216 33554432 67108864 100000000; do
  for h in 0 2; do
    echo -n "kb=$kb n=$n RSORT_HYBRID=$h: "
    RSORT_HYBRID=$h RS_TUNE_WALL=1 timeout -k 10 120 tools/rs_tune $L $kb $n random 5 2>&1 | grep -o "wall us: median [0-9.]*" || exit 1
  done
done
done
for n in 10000000 16777216 33554432 67108864 100000000 134217728; do
  for h in 0 1; do
    echo -n "u32 n=$n RSORT_HYBRID=$h: "
    RSORT_HYBRID=$h RS_TUNE_WALL=1 timeout -k 10 120 tools/rs_tune $L 4 $n random 5 2>&1 | grep -o "wall us: median [0-9.]*" || exit 1
  done
done
