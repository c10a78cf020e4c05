"""As shipped against RSORT_HYBRID=0 on 8-byte keys that take few thousand distinct values (beyond the few-distinct path):
the narrow form's bound on the largest group assumes independent digits.
python tools/clustered_bench.py [4|8]   (key bytes; default 8)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sort_amd as rs
eng = rs.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for n in ((10_000_000, 50_000_000, 110_000_000) if kb == 8 else (100_000_000, 300_000_000)):
    for distinct in (2000, 20000, 200000, 2000000):
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        pool = (torch.randint(-2**31, 2**31, (distinct,), dtype=torch.int64, device="cuda", generator=g) << 32) | torch.randint(0, 2**32, (distinct,), dtype=torch.int64, device="cuda", generator=g)
        if kb == 4:
            pool = (pool >> 32).to(torch.int32)
        src = pool[torch.randint(0, distinct, (n,), device="cuda", generator=g)]
        row = []
        for mode in (-1, 0):
            eng.set_hybrid(mode)
            ts = []
            for it in range(5):
                k = src.clone(); torch.cuda.synchronize()
                t0 = time.perf_counter(); eng.sort_keys_(k, 0); eng.synchronize(); ts.append(time.perf_counter() - t0)
            row.append((min(ts) * 1e3, int(eng.lib.rs_debug_last_hybrid(eng.h))))
        print(f"n={n} distinct={distinct}: shipped {row[0][0]:.3f} ms (form {row[0][1]}), plain loop {row[1][0]:.3f} ms", flush=True)
