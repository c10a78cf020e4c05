"""Host-clock latency of one mid-size device sort: the call as it is against a replay of the same call captured into a
HIP graph (python tools/graph_latency.py).  Keys are not refreshed between repeats: after the first one the input is sorted,
which the engine (pre-check off) sorts like any other input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sort_amd as rs

eng = rs.Engine(0)
eng.set_precheck(False)
side = torch.cuda.Stream()
eng.set_stream(side.cuda_stream)
for kb, dt in ((4, torch.int32), (8, torch.int64)):
    for n in (100_000, 1_000_000, 4_000_000 if kb == 4 else 2_000_000):
        src = torch.randint(0, 2**31 - 1, (n,), dtype=dt, device="cuda")       # non-negative: signed order = the unsigned order sorted by
        buf, alt = src.clone(), torch.empty_like(src)
        torch.cuda.synchronize()
        eng.sort_keys_ptr(buf.data_ptr(), n, kb, 0, d_alt=alt.data_ptr()); eng.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            eng.sort_keys_ptr(buf.data_ptr(), n, kb, 0, d_alt=alt.data_ptr())
        res = {}
        for mode in ("call", "graph"):
            ts = []
            for it in range(200):
                buf.copy_(src); torch.cuda.synchronize()
                t0 = time.perf_counter()
                if mode == "call":
                    eng.sort_keys_ptr(buf.data_ptr(), n, kb, 0, d_alt=alt.data_ptr()); eng.synchronize()
                else:
                    g.replay(); torch.cuda.current_stream().synchronize()      # (a replay runs on the current stream)
                ts.append((time.perf_counter() - t0) * 1e6)
            ts.sort()
            res[mode] = (ts[len(ts) // 2], ts[0])
        ok = bool((buf[1:] >= buf[:-1]).all())
        print("kb=%d n=%d  call median %.1f min %.1f us   graph median %.1f min %.1f us  %s" % (kb, n, *res["call"], *res["graph"], "OK" if ok else "WRONG"))
