#!/bin/bash
# usage: tools/shift_sweep.sh lib.so   (GPU box) -- pass time against the relative placement of the ping-pong buffers
for rep in 1 2; do
for sh in 0 256 4096 65536 1048576 2101248 16777216 67108864; do
  RS_TUNE_SHIFT=$sh timeout -k 10 120 tools/rs_tune "$1" 4 30 random 4 4 | tail -n 2 | tr '\n' ' ' | awk '{print $1,$2,$4,$6,$8,$10, "|", $19,$20,$21,$22,$23,$24,$25,$26}' || exit 1
done
done
for sh in 0 4096 1048576 16777216; do
  RS_TUNE_SHIFT=$sh timeout -k 10 120 tools/rs_tune "$1" 4 30 random 4 0 | tail -n 2 | tr '\n' ' ' | awk '{print $1,$2,$4,$6, "|", $19,$20,$21,$22,$23,$24,$25,$26}' || exit 1
  RS_TUNE_SHIFT=$sh timeout -k 10 120 tools/rs_tune "$1" 8 30 random 3 0 | tail -n 2 | tr '\n' ' ' | awk '{print $1,$2,$4,$6, "|", $19,$20,$21,$22,$23,$24,$25,$26}' || exit 1
done
