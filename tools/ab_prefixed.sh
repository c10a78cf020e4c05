#!/bin/bash
# usage (GPU box): tools/ab_prefixed.sh -- keys below a shared 3-bit prefix (a rank of eight after the exchange) and sorts beyond 2^30 keys, as shipped and with the plain loop
cd $GRAFT_REPO_ROOT
L=radix-sort_amd/lib/librsort_hip.so
for args in "4 30 prefix3" "8 30 prefix3" "4 1100000000 prefix3" "8 1100000000 prefix3" "4 1100000000 random" "4 1250000000 random"; do
  for h in -1 0; do
    echo -n "RSORT_HYBRID=$h: "
    RSORT_HYBRID=$h timeout -k 10 120 tools/rs_tune $L $args 3 2>&1 | tail -1 | cut -c42- || exit 1
  done
done
