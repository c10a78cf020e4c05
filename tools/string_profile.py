"""One rs_sort_strings call at n strings (uniform bytes 0..255, length 0..20), for rocprofv3 --kernel-trace: python tools/string_profile.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radix_sort_amd as rs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
rng = np.random.Generator(np.random.PCG64(69))
lens = rng.integers(0, 21, size=n).astype(np.uint64)
offs = np.zeros(n + 1, dtype=np.uint64); np.cumsum(lens, out=offs[1:])
blob = rng.integers(0, 256, size=int(offs[-1]), dtype=np.uint8)
eng = rs.Engine(0)
eng.set_host_threads(4)
for it in range(3):
    t0 = time.perf_counter()
    perm = eng.sort_strings_host(blob, offs, stable=True)
    dt = time.perf_counter() - t0
    print(f"n={n} blob={blob.size/1e6:.0f} MB  rs_sort_strings {dt*1e3:.1f} ms  ({n/dt/1e6:.1f} Mstrings/s)", flush=True)
eng.close()
