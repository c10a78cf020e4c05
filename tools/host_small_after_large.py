"""Does a large staged transfer change the latency of later small host sorts on the same context?  Per-call trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radix_sort_amd as rs

def trace(eng, n, calls=400):
    v = np.random.default_rng(1).integers(0, 2**32, n, dtype=np.uint32)
    out = []
    t_begin = time.perf_counter()
    for _ in range(calls):
        t0 = time.perf_counter(); eng.sort_keys_host(v, 0); out.append((time.perf_counter() - t0) * 1e6)
    a = np.array(out)
    slow = np.nonzero(a > 2 * np.median(a))[0]
    return "median %.0f us, first 8: %s, calls slower than 2x median: %d (last at #%d, %.1f ms after start)" % (
        np.median(a), " ".join("%.0f" % x for x in a[:8]), slow.size, slow[-1] if slow.size else -1,
        a[:slow[-1] + 1].sum() / 1e3 if slow.size else 0.0)

eng = rs.Engine(0)
big = np.random.default_rng(2).integers(0, 2**32, 16_000_000, dtype=np.uint32)
print("fresh            1e4:", trace(eng, 10_000))
for lanes in (1, 2, 4):
    eng.set_host_threads(lanes); eng.sort_keys_host(big, 0)
    print("after %d-lane big 1e4:" % lanes, trace(eng, 10_000))
    eng.sort_keys_host(big, 0)
    print("after %d-lane big 1e5:" % lanes, trace(eng, 100_000))
eng.sort_keys_host(big, 0); time.sleep(0.5)
print("4-lane, 0.5 s idle, 1e4:", trace(eng, 10_000))
