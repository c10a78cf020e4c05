#!/bin/bash
# usage: tools/run_variants.sh "<kb> <n> <shape> [iters] [vb]" lib1.so lib2.so ...   (GPU box; each run time-limited)
args="$1"; shift
for lib in "$@"; do
  timeout -k 5 120 tools/rs_tune "$lib" $args || { echo "FAILED/TIMEOUT $lib rc=$?"; exit 1; }
done
