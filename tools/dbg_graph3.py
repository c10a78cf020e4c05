# control: a graph of plain torch work (no library of ours), replayed with torch kernels on the default stream in between
import torch
n = 300_000
side = torch.cuda.Stream()
buf = torch.zeros(n, dtype=torch.int32, device="cuda")
out = torch.zeros_like(buf)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    out.copy_(buf); out.add_(1)
side.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    out.copy_(buf); out.add_(1)
for it in range(5):
    d = torch.full((n,), it * 10, dtype=torch.int32, device="cuda"); buf.copy_(d)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print("torch-only graph replay", it, "ok", bool((out == it * 10 + 1).all()))
