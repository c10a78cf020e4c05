"""Where the time of a mid-size device-resident sort goes: python tools/mid_breakdown.py   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sort_amd as rs

st = torch.cuda.Stream()
eng = rs.Engine(0, stream=st.cuda_stream)
eng.set_precheck(False)
for n in (100_000, 1_000_000, 4_000_000):
    src = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device="cuda")
    work = torch.empty_like(src)
    alt = torch.empty_like(src)
    t_call, t_done, t_sync, t_dev = [], [], [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(60):
        work.copy_(src); torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record(st)
        eng.sort_keys_ptr(work.data_ptr(), n, 4, 0, alt.data_ptr())
        e1.record(st)
        t1 = time.perf_counter()
        st.synchronize()
        t2 = time.perf_counter()
        eng.synchronize()
        t3 = time.perf_counter()
        t_call.append(t1 - t0); t_done.append(t2 - t0); t_sync.append(t3 - t2); t_dev.append(e0.elapsed_time(e1) * 1e-3)
    med = lambda v: sorted(v[10:])[len(v[10:]) // 2] * 1e6
    print(f"n={n}: queueing {med(t_call):.0f} us, queue+stream idle {med(t_done):.0f} us, device (events) {med(t_dev):.0f} us, rs_ctx_synchronize on an idle stream {med(t_sync):.0f} us")
