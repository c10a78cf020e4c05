"""End-to-end latency of small sorts through the C-ABI (host buffers in and out): python tools/small_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radix_sort_amd as rs
from radix_sort_amd import shapes

eng = rs.Engine(0)
for name in ("u32", "u64", "f64"):
    dt, kb, xf = shapes.KEY_TYPES[name]
    for n in (100, 1000, 8192, 16384):
        v = shapes.generate(name, n, "randomized")
        for _ in range(20):
            eng.sort_keys_host(v, xf)
        t0 = time.perf_counter()
        for _ in range(200):
            eng.sort_keys_host(v, xf)
        dt_us = (time.perf_counter() - t0) / 200 * 1e6
        print(f"RSORT_SMALL={os.environ.get('RSORT_SMALL', '1')} {name} n={n}: {dt_us:.1f} us per rs_sort_keys call")
