"""Stress: three host threads, each with its own context, run large sorts (2^28 uint32) at the same time with the per-XCD tile
order on.  A starved look-back would show as RS_ERR_KERNEL after 2 s (never a hang).  Observed on MI355X: no errors."""
import sys, os, threading, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import radix_sort_amd as rs
errors = []
def worker(seed):
    try:
        torch.cuda.set_device(0)
        eng = rs.Engine(0); eng.set_precheck(False); eng.set_tile_order(True)
        g = torch.Generator(device="cuda"); g.manual_seed(seed)
        src = torch.randint(-2**31, 2**31 - 1, (1 << 28,), dtype=torch.int32, device="cuda", generator=g)
        work = torch.empty_like(src)
        for it in range(6):
            work.copy_(src); torch.cuda.synchronize()
            eng.sort_keys_(work, 0); eng.synchronize()
            u = work ^ (-(1 << 31))
            if not bool((u[1:] >= u[:-1]).all().item()): errors.append((seed, it, "unsorted"))
        eng.close()
    except Exception as e:
        errors.append((seed, repr(e)))
t0 = time.time()
ts = [threading.Thread(target=worker, args=(s,)) for s in (1, 2, 3)]
[t.start() for t in ts]; [t.join() for t in ts]
print("errors:", errors, "elapsed %.1f s" % (time.time() - t0))
