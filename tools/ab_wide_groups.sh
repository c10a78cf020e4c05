#!/bin/bash
# A/B: 8-byte keys and pairs: RSORT_WIDE_SMALL=0 (workgroups of 1024 threads, groups as before) against the shipped rules
cd "$(dirname "$0")/.."
L=radix-sort_amd/lib/librsort_hip.so
out=gpurun_out/narrow_wide2.txt; : > $out
for n in 3000000 5000000 10000000 20000000 30000000 50000000 67108864 100000000 115000000 134217728 200000000 268435456 400000000 536870912 1073741824; do
  for cfg in "0 100000000 12185" "1 - -"; do
    set -- $cfg
    echo -n "u64 n=$n small=$1: " >> $out
    if [ $1 = 0 ]; then export RSORT_WIDE_SMALL=0 RSORT_HYBRID_MIN_WIDE=$2 RSORT_NARROW_GROUP=$3; else unset RSORT_WIDE_SMALL RSORT_HYBRID_MIN_WIDE RSORT_NARROW_GROUP; fi
    timeout -k 10 120 tools/rs_tune $L 8 $n random 7 2>&1 | tail -1 | grep -o "total [0-9.]* ms.*max [0-9.]*" >> $out || { echo FAILED >> $out; exit 1; }
  done
done
for n in 67108864 100000000 200000000 268435456 400000000 536870912; do
  for sm in 0 1; do
    echo -n "u64+u32 pairs n=$n small=$sm: " >> $out
    RSORT_WIDE_SMALL=$sm timeout -k 10 120 tools/rs_tune $L 8 $n random 7 4 2>&1 | tail -1 | grep -o "total [0-9.]* ms.*max [0-9.]*" >> $out || { echo FAILED >> $out; exit 1; }
  done
done
