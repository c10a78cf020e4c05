#!/bin/bash
for lib in tools/var/keepdig.so tools/var/runs4.so tools/var/new.so tools/var/keepdig.so tools/var/runs4.so tools/var/new.so; do
  for cfg in "8 30 sorted 3 0" "8 30 dup 3 0" "8 30 random 3 0" "8 30 ramp32 3 0"; do
  timeout -k 10 120 tools/rs_tune $lib $cfg 2>&1 | tail -n 1
  done
done
