#!/usr/bin/env python3
"""usage: tools/pmc_summary.py gpurun_out/pmc_<tag> ... -- per-kernel mean of every counter in rocprofv3 --pmc csv output"""
import csv, glob, sys, collections, re
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70], r["Counter_Name"]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(f"{d}: {k:70s} {c:28s} mean {v / n:16.1f}  launches {n}")
