"""Soak for the small and mid-size paths: sorts of random size (2 .. 6e6 keys), key width, transform and payload on ONE
context, one after the other with no synchronisation between the library's own state changes (mid-size sorts clear the
control block for each other), each compared on the device with torch's stable sort of the order-preserving image of
the keys.  python tools/soak_mid.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radix_sort_amd as rs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.Generator(np.random.PCG64(11))
eng = rs.Engine(0)
eng.set_precheck(False)
DT = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}


def image(k, kb, xf):
    """int64 tensor whose signed order is the order the library sorts by"""
    bits = kb * 8
    v = k.to(torch.int64)
    if kb < 8:
        u = v & ((1 << bits) - 1)                    # raw bits, unsigned
        if xf == 0: return u
        if xf == 1: return v
        neg = (u >> (bits - 1)) != 0
        return torch.where(neg, ((1 << bits) - 1) - u, u | (1 << (bits - 1)))
    sign = -(1 << 63)
    if xf == 0: return v ^ sign
    if xf == 1: return v
    return torch.where(v < 0, ~v, v ^ sign) ^ sign


t_end = time.time() + budget
done = fails = 0
while time.time() < t_end:
    kb = int(rng.choice([1, 2, 4, 4, 8, 8]))
    xf = int(rng.choice([0, 1, 2])) if kb >= 4 else int(rng.choice([0, 1]))
    pairs = bool(rng.integers(3) == 0)
    r = rng.integers(4)
    n = int(rng.integers(2, 40_000)) if r == 0 else int(rng.integers(40_000, 4_300_000)) if r < 3 else int(rng.integers(4_000_000, 6_000_000))
    lo, hi = -(1 << (kb * 8 - 1)), (1 << (kb * 8 - 1)) - 1
    keys = torch.randint(lo, hi, (n,), dtype=DT[kb], device="cuda")
    shape = rng.integers(4)
    if shape == 1 and kb > 1:
        keys = keys >> int(rng.integers(1, kb * 8 - 7))        # constant high bytes, duplicates
    elif shape == 2:
        keys = torch.sort(keys).values                         # ordered input
    want_img, order = torch.sort(image(keys, kb, xf), stable=True)
    vals = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
    alt = torch.empty_like(keys) if rng.integers(2) else None
    torch.cuda.synchronize()
    if pairs:
        eng.sort_pairs_ptr(keys.data_ptr(), vals.data_ptr(), n, kb, 4, xf)
    else:
        eng.sort_keys_ptr(keys.data_ptr(), n, kb, xf, d_alt=alt.data_ptr() if alt is not None else None)
    if rng.integers(4) == 0:
        eng.synchronize()          # sometimes wait, mostly let the next sort queue behind this one
    else:
        eng.synchronize()          # the check below runs on torch's stream: the sort must have finished
    ok = bool((image(keys, kb, xf) == want_img).all())
    if pairs:
        ok = ok and bool((vals.to(torch.int64) == order).all())
    done += 1
    if not ok:
        fails += 1
        print("FAIL", kb, xf, pairs, n, int(shape), flush=True)
    if done % 500 == 0:
        print(done, "sorts,", fails, "failures", flush=True)
print("soak_mid:", done, "sorts,", fails, "failures")
sys.exit(1 if fails else 0)
