"""Device time of rs_sort_strings on inputs that need several MSD levels: python tools/string_levels.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radix_sort_amd as rs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(5)
eng = rs.Engine(0)
eng.set_profiling(True)
for prefix, tail in ((0, 12), (7, 8), (14, 8), (28, 8), (14, 2)):
    L = prefix + tail
    blob = np.empty((n, L), dtype=np.uint8)
    blob[:, :prefix] = 97
    blob[:, prefix:] = rng.integers(0, 256, size=(n, tail), dtype=np.uint8)
    offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    flat = blob.reshape(-1)
    eng.sort_strings_host(flat[: 64 * L], offsets[:65].copy())
    t0 = time.perf_counter()
    perm = eng.sort_strings_host(flat, offsets)
    t1 = time.perf_counter()
    dev = sum(eng.profile())
    # check: sorted order of the rows
    rows = blob[perm.astype(np.int64)]
    v = rows.view([("", np.uint8)] * L) if False else None
    ok = True
    idx = rng.integers(0, n - 1, size=100000)
    a, b = rows[idx], rows[idx + 1]
    for x, y, i in zip(a, b, idx):
        tx, ty = x.tobytes(), y.tobytes()
        if tx > ty or (tx == ty and perm[i] > perm[i + 1]):
            ok = False; break
    print(f"n={n} constant prefix {prefix} B + {tail} random bytes: device {dev:.2f} ms, end to end {(t1 - t0) * 1e3:.1f} ms, ordered sample {ok}", flush=True)
