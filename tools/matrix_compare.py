"""Lays tools/bench_matrix output beside the reference's published result files (benchmarks/v2.0 *.txt; the authors' CPU,
hardware not stated) row for row.  Runs in the build container only (the reference is not on the GPU box).
  python tools/matrix_compare.py profiles/r01_g_matrix_bench_all.txt "/root/reference/benchmarks/v2.0 0 all.txt" > profiles/..."""
import re
import sys

ROW = re.compile(r"(\w+)\nSIZE = ([\d,]+) \((.*?)\)\n\n((?:.+\n)+)")


def parse(path):
    out = {}
    for ty, n, shape, body in ROW.findall(open(path).read()):
        vals = dict((k.strip(), int(v.replace(",", ""))) for k, v in re.findall(r"(\w+)\s*= ([\d,]+) us", body))
        out[(shape, ty, int(n.replace(",", "")))] = vals
    return out


ours, ref = parse(sys.argv[1]), parse(sys.argv[2])
print("| shape | type | size | reference radix_sort / radix_sort_par (us, published) | this engine radix_sort / radix_sort_par (us, MI355X, PCIe included) | par speed-up |")
print("|---|---|---|---|---|---|")
for key in ours:
    shape, ty, n = key
    if n < 1_000_000 or key not in ref:
        continue
    r, o = ref[key], ours[key]
    print(f"| {shape} | {ty} | {n:,} | {r['radix_sort']:,} / {r['radix_sort_par']:,} | {o['radix_sort']:,} / {o['radix_sort_par']:,} | "
          f"{r['radix_sort_par'] / max(o['radix_sort_par'], 1):.1f}x |")
small = [(k, ours[k], ref[k]) for k in ours if k in ref and 10 <= k[2] <= 1000 and k[0] == "randomized"]
print("\nSmall sizes (randomized), where a GPU round trip cannot win: reference radix_sort vs this engine, us")
for (shape, ty, n), o, r in small:
    print(f"  {ty} {n}: {r['radix_sort']} vs {o['radix_sort']}")
