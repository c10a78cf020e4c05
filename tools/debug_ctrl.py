import sys, os, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import radix_sort_amd as rs
eng = rs.Engine(0)
lib = eng.lib
lib.rs_debug_read_ctrl.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
n = 16384
rng = np.random.Generator(np.random.PCG64(1))
v = rng.integers(0, 1 << 32, n, dtype=np.uint32)
try:
    out = eng.sort_keys_host(v, 0)
except Exception as e:
    print("EXC", e)
chunks = 2
gh = np.zeros(4 * chunks * 256, np.uint32)
gb = np.zeros(4 * chunks * 256, np.uint64)
print(lib.rs_debug_read_ctrl(eng.h, gh.ctypes.data_as(C.c_void_p), 128, gh.nbytes))
print(lib.rs_debug_read_ctrl(eng.h, gb.ctypes.data_as(C.c_void_p), 128 + 8 * 256 * 256 * 4, gb.nbytes))
gh = gh.reshape(4, chunks, 256); gb = gb.reshape(4, chunks, 256)
for p in range(4):
    d = (v >> np.uint32(8 * p)) & np.uint32(255)
    want = np.stack([np.bincount(d[c * 8192:(c + 1) * 8192], minlength=256) for c in range(chunks)])
    tot = want.sum(0); db = np.cumsum(tot) - tot
    wb = np.stack([db + want[:c].sum(0) for c in range(chunks)])
    print("pass", p, "hist ok", np.array_equal(want, gh[p]), "base ok", np.array_equal(wb, gb[p].astype(np.int64)))
    if not np.array_equal(want, gh[p]):
        print(" hist sums", gh[p].sum(1), "want", want.sum(1)); print(gh[p][:, :8], want[:, :8])
    if not np.array_equal(wb, gb[p].astype(np.int64)):
        print(gb[p][:, :6], wb[:, :6])

dbg = np.zeros(4096, np.uint32)
lib.rs_debug_read_dbg.argtypes = [C.c_void_p]
print(lib.rs_debug_read_dbg(dbg.ctypes.data_as(C.c_void_p)))
for p in range(4):
    for t in range(3):
        print("pass", p, "ticket", t, dbg[p * 512 + t * 8: p * 512 + t * 8 + 8])
    d = (v >> np.uint32(8 * p)) & np.uint32(255)
    print("  want gbase[1][255]", int((d < 255).sum() + (d[:8192] == 255).sum()), "tile_start255 chunk1", int((d[8192:] < 255).sum()))
