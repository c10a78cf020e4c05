#!/bin/bash
# usage: tools/ab_bench.sh "<bench.py args>" lib1.so lib2.so ...   (GPU box) -- bench.py rows with each library in turn
args="$1"; shift
for lib in "$@"; do
  cp "$lib" radix-sort_amd/lib/librsort_hip.so || exit 1
  echo "== $lib"
  timeout -k 10 300 python bench.py --no-cpu $args | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['rows']:
    f=r.get('full_sort') or {}
    print('  %-4s %-15s %8.2f ms %7.1f Gkeys/s  pass %s  full-sort %s'%(r['type'],r['shape'],r['ms'],r['gkeys_s'],('%.3f'%r['pass_ms']) if r['pass_ms'] else '-', ('%.2f ms pass %.3f'%(f['ms'],f['pass_ms'])) if f else '-'))
" || exit 1
done
