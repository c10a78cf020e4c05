#!/bin/bash
for lib in tools/var/lbf12.so tools/var/lbf16.so tools/var/new.so tools/var/lbf12.so tools/var/lbf16.so tools/var/new.so; do
  for cfg in "4 30 random 4 0" "8 30 random 3 0" "4 30 nearly 3 0"; do
  timeout -k 10 120 tools/rs_tune $lib $cfg 2>&1 | tail -n 1
  done
done
