#!/bin/bash
for cfg in "4 100000007 sorted" "4 100000007 random" "8 100000007 random" "8 27 dup"; do
  timeout -k 10 120 tools/rs_tune tools/var/checked.so $cfg 2 0 2>&1 | tail -n 1
done
for lib in tools/var/et1.so tools/var/et0.so tools/var/et1.so tools/var/et0.so; do
  for cfg in "4 30 random" "8 30 random"; do
  timeout -k 10 120 tools/rs_tune $lib $cfg 4 0 2>&1 | tail -n 1
  done
done
