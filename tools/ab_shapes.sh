#!/bin/bash
# usage: tools/ab_shapes.sh lib1.so lib2.so ...   (GPU box) -- u32 keys, every shape, N = 2^30, for each library in turn
for lib in "$@"; do
  for shape in random sorted reverse nearly dup; do
    timeout -k 10 120 tools/rs_tune "$lib" 4 30 $shape 4 0 | tail -n 1 | awk '{print $1,$2,$5,$9,$10,$11,$12,$13,$14,$15,$16,$17,$20}' || exit 1
  done
done
