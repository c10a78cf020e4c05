import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radix_sort_amd as rs
V = os.environ.get("V", "a")
n = 300_000
eng = rs.Engine(0)
eng.set_precheck(False)
side = torch.cuda.Stream()
if V == "b":      # torch work on the default stream before anything else
    junk = torch.sort(torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, device="cuda")).values
buf = torch.empty(n, dtype=torch.int32, device="cuda")
alt = torch.empty_like(buf)
eng.set_stream(side.cuda_stream)
with torch.cuda.stream(side):
    buf.random_(0, 1 << 30)
side.synchronize()
eng.sort_keys_ptr(buf.data_ptr(), n, 4, 0, d_alt=alt.data_ptr())
eng.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    eng.sort_keys_ptr(buf.data_ptr(), n, 4, 0, d_alt=alt.data_ptr())
rng = np.random.default_rng(1)
for it in range(5):
    v = rng.integers(0, 2**32, n, dtype=np.uint32)
    if V in "ce":
        d = torch.from_numpy(v.view(np.int32)).cuda(); buf.copy_(d)        # D2D copy into buf
    else:
        buf.copy_(torch.from_numpy(v.view(np.int32)))
    torch.cuda.synchronize()
    if V == "e":
        with torch.cuda.stream(side):
            g.replay()
    else:
        g.replay()
    torch.cuda.synchronize()
    if V == "g":
        dev = lambda: bool((buf[1:].to(torch.int64) & 0xFFFFFFFF >= buf[:-1].to(torch.int64) & 0xFFFFFFFF).all())
        d1 = dev()
        got = buf.cpu().numpy().view(np.uint32)
        d2 = dev()
        print(V, "replay", it, "device check", d1, "| host copy sorted", bool((got == np.sort(v)).all()), "host copy equals input", bool((got == v).all()), "| device check again", d2)
    elif V in "de":
        ok = bool((buf[1:].to(torch.int64) & 0xFFFFFFFF >= buf[:-1].to(torch.int64) & 0xFFFFFFFF).all())   # device-side check, no D2H of buf
        print(V, "replay", it, "sorted on device:", ok)
    else:
        got = buf.cpu().numpy().view(np.uint32)
        print(V, "replay", it, "ok", bool((got == np.sort(v)).all()), "equals input", bool((got == v).all()))
