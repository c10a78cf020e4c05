#!/bin/bash
# A/B: 4-byte keys below the 16-bit-group threshold: the four-pass loop (as shipped) against the narrow form forced, group kernel in small workgroups; pairs of 8-byte keys below their threshold
cd "$(dirname "$0")/.."
L=radix-sort_amd/lib/librsort_hip.so
out=gpurun_out/narrow32.txt; : > $out
for n in 3000000 5000000 10000000 20000000 33554432 50000000 64000000 100000000; do
  for h in -1 2; do
    echo -n "u32 n=$n RSORT_HYBRID=$h: " >> $out
    RSORT_HYBRID=$h RS_TUNE_WALL=1 timeout -k 10 120 tools/rs_tune $L 4 $n random 9 2>&1 | grep -o "total [0-9.]* ms\|wall us: median [0-9.]*" | paste - - >> $out || { echo FAILED >> $out; exit 1; }
  done
done
for n in 10000000 20000000 30000000 45000000; do
  for h in 0 1; do
    echo -n "u64+u32 pairs n=$n RSORT_HYBRID=$h: " >> $out
    RSORT_HYBRID=$h timeout -k 10 120 tools/rs_tune $L 8 $n random 7 4 2>&1 | tail -1 | grep -o "total [0-9.]* ms" >> $out || { echo FAILED >> $out; exit 1; }
  done
done
