#!/bin/bash
# usage: tools/ab_round.sh checked.so new.so head.so   (GPU box) -- a new kernel's first run: the range-checked build on ragged and
# full sizes first (a wrong rank becomes RS_ERR_KERNEL, not a wild store), then the shape rows of the candidate and the baseline
set -e
chk=$1; shift
for kb in 4 8; do
  for cfg in "100000007 random" "100000007 sorted" "100000007 dup" "33554433 nearly" "28 random" "28 dup" "28 sorted" "28 nearly"; do
    timeout -k 10 120 tools/rs_tune "$chk" $kb $cfg 2 0 | tail -n 1
  done
done
timeout -k 10 120 tools/rs_tune "$chk" 8 27 random 2 4 | tail -n 1
timeout -k 10 120 tools/rs_tune "$chk" 4 27 random 2 4 | tail -n 1
tools/ab_shapes.sh "$@"
