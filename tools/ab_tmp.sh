#!/bin/bash
for lib in tools/var/prio2_lbf8.so tools/var/prio2_lbf12.so tools/var/prio2.so tools/var/prio2_lbf8.so tools/var/prio2_lbf12.so tools/var/prio2.so; do
  for cfg in "4 30 random" "8 30 random"; do
  timeout -k 10 120 tools/rs_tune $lib $cfg 4 0 2>&1 | tail -n 1
  done
done
