"""Device-resident latency of mid-size sorts (rs_sort_keys_device + synchronise): python tools/mid_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sort_amd as rs

eng = rs.Engine(0)
wide = len(sys.argv) > 1 and sys.argv[1] == "u64"          # python tools/mid_latency.py u64: 8-byte keys
for pre in (True, False):
    eng.set_precheck(pre)
    for n in (20_000, 100_000, 1_000_000, 4_000_000, 16_000_000):   # (the first four are 'mid-size': at most 256 tiles)
        if wide and n > 4_000_000:
            continue
        src = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device="cuda")
        if wide:
            src = (src.to(torch.int64) << 32) | torch.randint(0, 2**32, (n,), dtype=torch.int64, device="cuda")
        work = torch.empty_like(src)
        ts = []
        for it in range(60):
            work.copy_(src); torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.sort_keys_ptr(work.data_ptr(), n, 8 if wide else 4, 0); eng.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[10:])
        print(f"precheck={pre} {'u64' if wide else 'u32'} n={n}: median {ts[len(ts)//2]*1e6:.0f} us  ({n/ts[len(ts)//2]/1e9:.2f} Gkeys/s)")
