#!/bin/bash
# usage (GPU box): tools/ab_large_n.sh -- sorts beyond 2^30 keys, order checked on the device (profiles/r03_d_large_n.txt)
cd $GRAFT_REPO_ROOT
L=radix-sort_amd/lib/librsort_hip.so
for args in "4 31 random" "4 3000000019 random" "4 4294967296 random" "8 31 random" "4 31 dups" "8 31 dups" "4 31 nearly" "8 2500000000 random"; do
  timeout -k 10 120 tools/rs_tune $L $args 3 2>&1 | tail -1 || exit 1
done
