#!/bin/bash
# usage: tools/ab_shapes.sh lib1.so lib2.so ...   (GPU box) -- uint32 and uint64 keys, every shape, N = 2^30, for each library in turn
for lib in "$@"; do
  for kb in 4 8; do
    for shape in random sorted reverse nearly dup; do
      timeout -k 10 120 tools/rs_tune "$lib" $kb 30 $shape 4 0 | tail -n 1 || exit 1
    done
  done
done
