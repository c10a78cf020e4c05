#!/bin/bash
# usage: tools/ab_tune.sh lib1.so lib2.so ...   (GPU box) -- tools/rs_tune rows (keys and pairs, N = 2^30 / 2^29) for each library in turn
for lib in "$@"; do
  for cfg in "4 30 random 5 0" "8 30 random 5 0" "4 30 random 5 4" "4 29 random 5 8" "8 30 random 3 8"; do
    timeout -k 10 120 tools/rs_tune "$lib" $cfg | tail -n 1 || exit 1
  done
done
