#!/bin/bash
for lib in tools/var/fewchain.so tools/var/new.so tools/var/fewchain.so tools/var/new.so; do
  for cfg in "4 30 dup 4 0" "4 29 dup 3 4"; do
  timeout -k 10 120 tools/rs_tune $lib $cfg 2>&1 | tail -n 1
  done
done
