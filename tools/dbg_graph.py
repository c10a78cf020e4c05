import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sort_amd as rs
eng = rs.Engine(0)
eng.set_precheck(False)
side = torch.cuda.Stream()
eng.set_stream(side.cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
src = torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, device="cuda")
want = torch.sort(src).values
src_host = src.cpu()
buf, alt = src.clone(), torch.empty_like(src)
torch.cuda.synchronize()
eng.sort_keys_ptr(buf.data_ptr(), n, 4, 0, d_alt=alt.data_ptr()); eng.synchronize()
print("eager ok", bool((buf == want).all()))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    eng.sort_keys_ptr(buf.data_ptr(), n, 4, 0, d_alt=alt.data_ptr())
for it in range(6):
    if os.environ.get("NEWDATA"):
        src = torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, device="cuda"); want = torch.sort(src).values; buf.copy_(src)
    elif os.environ.get("COPY") == "h2d": buf.copy_(src_host)
    else: buf.copy_(src)
    torch.cuda.synchronize()
    if os.environ.get("SLEEP"): time.sleep(float(os.environ["SLEEP"]))
    if os.environ.get("CHECKS") == "none" and it < 5:
        g.replay(); torch.cuda.synchronize(); continue
    t0 = time.perf_counter()
    g.replay()
    if it % 2 == 0: torch.cuda.current_stream().synchronize()
    else: torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e6
    torch.cuda.synchronize()
    print("replay", it, "current-stream sync" if it % 2 == 0 else "device sync", "%.1f us" % dt, "ok", bool((buf == want).all()), "mismatches", int((buf != want).sum()), "equals src:", int((buf == src).sum()), "sorted:", bool((buf[1:] >= buf[:-1]).all()),
          "alt==want", int((alt == want).sum()), "multiset ok:", bool((torch.sort(buf).values == want).all()), "by low 8/16/24 bits sorted:",
          [bool((((buf & m)[1:]) >= ((buf & m)[:-1])).all()) for m in (0xFF, 0xFFFF, 0xFFFFFF)])
try:
    eng.synchronize(); print("no fault")
except Exception as e:
    print("fault:", e)
