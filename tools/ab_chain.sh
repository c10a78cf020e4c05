#!/bin/bash
# usage: tools/ab_chain.sh lib.so [shapes...]   (GPU box) -- chain mode off / on, uint32 and uint64 keys, N = 2^30
lib="$1"; shift
shapes="${@:-random}"
for shape in $shapes; do
  for cfg in "4 30" "8 30"; do
    for chain in 0 1; do
      echo -n "chain=$chain "
      RSORT_CHAIN=$chain timeout -k 10 120 tools/rs_tune "$lib" $cfg $shape 5 0 | tail -n 1 || exit 1
    done
  done
done
