#!/bin/bash
# usage: tools/ab_keys.sh lib1.so lib2.so ...   (GPU box) -- keys-only rs_tune rows (u32 and u64, N = 2^30) for each library in turn
for lib in "$@"; do
  for cfg in "4 30 random 6 0" "8 30 random 4 0"; do
    timeout -k 10 120 tools/rs_tune "$lib" $cfg | tail -n 1 | awk '{print $1,$2,$9,$10,$11,$12,$13,$14,$15,$16,$17}' || exit 1
  done
done
