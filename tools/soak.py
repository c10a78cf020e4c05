"""Soak: many large sorts of random size and type on one context (per-XCD tile order active: >= 1024 tiles), each checked on the
device for order, checksum and (pairs) payload consistency.  python tools/soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radix_sort_amd as rs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.Generator(np.random.PCG64(7))
eng = rs.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_precheck(False)
t_end = time.time() + budget
done = fails = 0
while time.time() < t_end:
    kb = int(rng.choice([4, 8]))
    xf = int(rng.choice([0, 1, 2]))
    pairs = bool(rng.integers(2))
    tile = 16384 if (kb == 4 and True) else 8192
    n = int(rng.integers(1024 * tile, 1024 * tile * 3)) + int(rng.integers(0, 64 * tile) if rng.integers(2) else 0)
    dt = torch.int32 if kb == 4 else torch.int64
    lo, hi = (-2**31, 2**31 - 1) if kb == 4 else (-2**63, 2**63 - 1)
    keys = torch.randint(lo, hi, (n,), dtype=dt, device="cuda")
    if rng.integers(3) == 0:
        keys = keys >> int(rng.integers(1, kb * 8 - 8))           # few significant bits / many duplicates
    ref_sum = keys.sum()
    sign = lo
    vals = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
    src = keys.clone() if pairs else None
    torch.cuda.synchronize()          # the engine runs on its own stream: torch must have finished writing the keys
    if pairs:
        eng.sort_pairs_(keys, vals, xf)
    else:
        eng.sort_keys_(keys, xf)
    eng.synchronize()
    okey = keys if xf == 1 else (keys ^ sign if xf == 0 else torch.where(keys < 0, ~keys ^ sign, keys))
    ok = bool((okey[1:] >= okey[:-1]).all().item()) and bool((keys.sum() == ref_sum).item())
    if pairs:
        ok = ok and bool((src[vals.long()] == keys).all().item())
        # stability: equal neighbours keep ascending payload
        eq = okey[1:] == okey[:-1]
        ok = ok and bool((vals[1:][eq] > vals[:-1][eq]).all().item())
    done += 1
    if not ok:
        fails += 1
        print("FAIL", dict(kb=kb, xf=xf, pairs=pairs, n=n))
    if done % 25 == 0:
        print(f"{done} sorts, {fails} failures", flush=True)
    del keys
print(f"soak: {done} sorts in {budget:.0f} s, failures: {fails}")
sys.exit(1 if fails else 0)
