"""Soak of the round-3 forms of large sorts -- hybrid (two sweeps + in-LDS group sort, forced on at these sizes), few
distinct keys, the groups' prefix passes + fix-up and its fall-back, oversize groups, stable pairs of 4- and 8-byte keys and a 4-byte
payload -- on one context, random size, width, transform and key distribution, every result compared element by element with
torch's (stable) sort of the same keys.
python tools/soak_forms.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radix_sort_amd as rs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.Generator(np.random.PCG64(11))
eng = rs.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_precheck(False)
t_end = time.time() + budget
done = fails = 0
forms = {"hybrid": 0, "narrow": 0, "prefixed": 0, "distinct": 0, "lsd": 0}


def rand_bits(n, kb):
    if kb == 4:
        return torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device="cuda").to(torch.int32)
    return (torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device="cuda") << 32) | torch.randint(0, 2**32, (n,), dtype=torch.int64, device="cuda")


while time.time() < t_end:
    kb = int(rng.choice([4, 8]))
    xf = int(rng.choice([0, 1, 2]))
    limit = int(rng.choice([0, 0, 0, 1, 30]))              # groups of more than 4 * limit keys go to the big-group kernel (0: what the window holds)
    n = int(rng.integers(4_300_000, 8_000_000 if limit else 40_000_000))
    dt = torch.int32 if kb == 4 else torch.int64
    bits = kb * 8
    kind = int(rng.integers(10))
    keys = rand_bits(n, kb)
    if kind == 1:                                            # few distinct keys (around the table's limit of 1024, too)
        pool = rand_bits(int(rng.choice([1, 2, 7, 256, 1000, 1024, 1025, 1500, 5000])), kb)
        if rng.integers(4) == 0:
            pool[0] = -1                                      # the all-ones key
        keys = pool[torch.randint(0, pool.numel(), (n,), device="cuda")]
    elif kind == 2:                                          # few top-16 values: large groups, some beyond the LDS window
        tops = int(rng.choice([3, 40, 300, 2000]))
        top = torch.randint(0, tops, (n,), dtype=torch.int64, device="cuda") * int(rng.integers(1, 65536 // tops + 1))
        keys = ((keys.to(torch.int64) & ((1 << (bits - 16)) - 1)) | (top << (bits - 16))).to(dt)
    elif kind == 3 and kb == 8:                              # 16-bit prefixes below the group bits taking few values: the fix-up's long runs
        pre = torch.randint(0, int(rng.choice([2, 50, 700, 5000])), (n,), dtype=torch.int64, device="cuda") * int(rng.integers(1, 13))
        keys = (keys & ~(0xFFFF << 32)) | ((pre & 0xFFFF) << 32)
    elif kind == 4:                                          # low bits of few values (equal whole keys inside the runs)
        keys = keys & ~int((1 << int(rng.integers(8, bits - 16))) - 1 - 5)
    elif kind == 5:                                          # nearly sorted
        keys = torch.sort(keys).values
        m = n // 40
        a = torch.randint(0, n, (m,), device="cuda"); b = torch.randint(0, n, (m,), device="cuda")
        tmp = keys[a].clone(); keys[a] = keys[b]; keys[b] = tmp
    elif kind == 7:                                          # leading bits shared by every key (a multi-GPU rank's keys): the prefixed form
        t = int(rng.integers(1, 9))
        keys = ((keys.to(torch.int64) & ((1 << (bits - t)) - 1)) | (int(rng.integers(0, 1 << t)) << (bits - t))).to(dt)
        if rng.integers(5) == 0:
            keys[int(rng.integers(0, n))] = 3                 # ... but for one key
    elif kind >= 8:                                          # a few thousand to a few hundred thousand distinct values: whole values in single groups
        pool = rand_bits(int(rng.choice([1100, 3000, 20000, 300000])), kb)
        if kind == 9:                                         # ... half of which share their top bits (groups of several values)
            pool[: pool.numel() // 2] &= ~((1 << 20) - 1) if kb == 4 else ~((1 << 40) - 1)
        keys = pool[torch.randint(0, pool.numel(), (n,), device="cuda")]
    hybrid = int(rng.choice([-1, -1, 1, 1, 2, 3, 3]))         # as shipped / 16-bit groups / narrow form / prefixed form forced where the shape allows
    eng.set_hybrid(hybrid)
    eng.lib.rs_debug_set_group_limit(eng.h, limit)
    sign = -(1 << (bits - 1))
    src = keys.clone()
    pairs = int(rng.integers(3)) == 0                        # (4-byte keys: windows of 4352 and 8704 pairs, the plain loop beyond)
    vals = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
    torch.cuda.synchronize()
    if pairs:
        eng.sort_pairs_(keys, vals, xf)
    else:
        eng.sort_keys_(keys, xf)
    eng.synchronize()
    took = int(eng.lib.rs_debug_last_hybrid(eng.h))
    forms["hybrid" if took == 1 else "narrow" if took == 2 else "prefixed" if took == 3 else "distinct" if eng.lib.rs_debug_last_distinct(eng.h) else "lsd"] += 1
    order = (lambda k: k if xf == 1 else (k ^ sign if xf == 0 else torch.where(k < 0, ~k ^ sign, k)))
    want, idx = torch.sort(order(src), stable=True)
    ok = bool((order(keys) == want).all().item())
    if pairs:
        ok = ok and bool((vals.to(torch.int64) == idx).all().item())
        forms["pairs"] = forms.get("pairs", 0) + 1
    done += 1
    if not ok:
        fails += 1
        print("FAIL", dict(kb=kb, xf=xf, n=n, kind=kind, hybrid=hybrid, limit=limit, pairs=pairs), flush=True)
    if done % 20 == 0:
        print(f"{done} sorts, {fails} failures, forms {forms}", flush=True)
    del keys, src, want, idx, vals
eng.lib.rs_debug_set_group_limit(eng.h, 0)
print(f"soak_forms: {done} sorts in {budget:.0f} s, failures: {fails}, forms {forms}")
sys.exit(1 if fails else 0)
