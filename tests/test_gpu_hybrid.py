"""GPU parity of the hybrid form of 4-byte key sorts (rsort_groups.hpp: two sweeps on the top digits, every group of equal
top 16 bits finished in LDS) against the CPU oracle, bit for bit: forced on at sizes the oracle sorts in seconds, for every
shape and transform, with the one-chain and the chain-mode sweeps, with groups pushed to the big-group kernel, on
unaligned sub-ranges; at 2^30 keys (what it is for) through the size-independent properties of a sort."""
import os
import subprocess
import sys

import numpy as np
import pytest

from radix_sort_amd import shapes

pytestmark = pytest.mark.gpu

HYBRID_SIZES = [4_300_001, 5_000_000, (1 << 23) + 7]     # (more than 256 tiles of either geometry: mid-size sorts never take the hybrid form)


def _sort_on_device(engine, v, xf, offset=0):
    """v through rs_sort_keys_device at a key offset inside a larger allocation (16-byte alignment is not a given)."""
    import torch
    kb = v.itemsize
    buf = torch.empty(v.size * kb + 64, dtype=torch.uint8, device="cuda")
    view = buf[offset * kb: offset * kb + v.size * kb]
    view.copy_(torch.from_numpy(v.view(np.uint8).copy()))
    torch.cuda.synchronize()
    engine.sort_keys_ptr(view.data_ptr(), v.size, kb, xf)
    engine.synchronize()
    took_hybrid = bool(engine.lib.rs_debug_last_hybrid(engine.h))
    return view.cpu().numpy().view(v.dtype), took_hybrid


@pytest.mark.parametrize("name", ["u32", "i32", "f32", "u64", "i64", "f64"])
@pytest.mark.parametrize("shape", shapes.SHAPES)
def test_hybrid_matches_oracle(engine, oracle, name, shape):
    dt, kb, xf = shapes.KEY_TYPES[name]
    engine.set_precheck(False)
    engine.set_hybrid(1)
    try:
        for n in HYBRID_SIZES:
            v = shapes.generate(name, n, shape)
            got, hybrid = _sort_on_device(engine, v, xf)
            assert hybrid, "the hybrid form was not taken"
            assert got.tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), (name, shape, n)
    finally:
        engine.set_hybrid(-1)
        engine.set_precheck(True)


def test_hybrid_unaligned_and_big_groups(engine, oracle):
    """Key-aligned (not 16-byte aligned) buffers; groups larger than the LDS window go to the big-group kernel (forced at
    test size by a small window, and by keys that share their top 16 bits); low digits of two values (a constant digit's pass
    is skipped, and with fewer than four live passes the hybrid form is not taken)."""
    rng = np.random.Generator(np.random.PCG64(23))
    engine.set_precheck(False)
    engine.set_hybrid(1)
    try:
        n = 4_400_003
        cases = {
            "random": rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32),
            "eight top values": (lambda k: (k << np.uint32(29)) | (k << np.uint32(16)))(rng.integers(0, 8, n, dtype=np.uint64).astype(np.uint32))
                                | rng.integers(0, 1 << 16, n, dtype=np.uint64).astype(np.uint32),
            "low byte of two values": rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) & np.uint32(0xFFFFFF01) | np.uint32(0x5A),
            "second byte of two values": rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) & np.uint32(0xFFFF80FF),
        }
        for what, v in cases.items():
            want = np.sort(v)
            for offset in (0, 1, 3):
                got, hybrid = _sort_on_device(engine, v, 0, offset)
                assert hybrid and got.tobytes() == want.tobytes(), (what, offset)
        for limit in (1, 8, 40):                      # groups of more than 4 * limit keys: big-group kernel
            engine.lib.rs_debug_set_group_limit(engine.h, limit)
            for name in ("u32", "f32", "u64", "f64"):
                dt, kb, xf = shapes.KEY_TYPES[name]
                v = shapes.generate(name, n, "randomized", seed=limit)
                got, hybrid = _sort_on_device(engine, v, xf, 2)
                assert hybrid and got.tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), (name, limit)
    finally:
        engine.lib.rs_debug_set_group_limit(engine.h, 0)
        engine.set_hybrid(-1)
        engine.set_precheck(True)


def test_hybrid_over_chain_mode_sweeps(oracle):
    """The hybrid form over the chain-mode sweeps (per-XCD chains; the group table then adds the counts of the chains
    before a boundary): chain mode forced at test size in a child process (the switch is read once per process)."""
    code = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import radix_sort_amd as rs
from radix_sort_amd import shapes
from cpu_checkers import load_oracle
orc = load_oracle(); eng = rs.Engine(0); eng.set_precheck(False); eng.set_hybrid(1)
for name in ("u32", "i32", "f32", "u64", "f64"):
    dt, kb, xf = shapes.KEY_TYPES[name]
    for shape in shapes.SHAPES:
        for n in (4_300_001, (1 << 23) + 13):
            v = shapes.generate(name, n, shape)
            t = torch.from_numpy(v.view(np.uint8).copy()).cuda()
            torch.cuda.synchronize()
            eng.sort_keys_ptr(t.data_ptr(), n, kb, xf); eng.synchronize()
            assert eng.lib.rs_debug_last_hybrid(eng.h) == 1
            assert t.cpu().numpy().view(v.dtype).tobytes() == orc.sort_keys(v, xf, mt=True).tobytes(), (name, shape, n)
print("CHAIN HYBRID OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSORT_CHAIN_MIN_TILES="256")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CHAIN HYBRID OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_hybrid_at_full_size():
    """2^30 keys, as shipped (hybrid chosen by the library): order, key sum; randomized, a ramp with swaps (nearly sorted) and
    the 256-value duplicates shape (which must NOT take the hybrid form: its groups are 4 M keys each)."""
    import torch
    import radix_sort_amd as rs
    eng = rs.Engine(0)
    eng.set_precheck(False)
    try:
        n = 1 << 30
        for shape in ("randomized", "ramp", "duplicates", "randomized 8-byte"):
            g = torch.Generator(device="cuda"); g.manual_seed(11)
            if shape == "randomized 8-byte":
                keys = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
            elif shape == "randomized":
                keys = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device="cuda", generator=g)
            elif shape == "ramp":
                keys = (torch.arange(n, dtype=torch.int64, device="cuda") * 4 - 2**31).to(torch.int32)
                idx = torch.randint(0, n, (1 << 20,), device="cuda", generator=g)
                keys[idx] = keys[idx.flip(0)]
            else:
                vals = torch.randint(-2**31, 2**31 - 1, (256,), dtype=torch.int32, device="cuda", generator=g)
                keys = vals[torch.randint(0, 256, (n,), device="cuda", generator=g)]
            want_sum = int(keys.sum(dtype=torch.int64).item())          # (wraps for 8-byte keys, the same way before and after)
            torch.cuda.synchronize()
            eng.sort_keys_(keys, 1)
            eng.synchronize()
            assert bool(eng.lib.rs_debug_last_hybrid(eng.h)) == (shape != "duplicates"), shape
            assert int(keys.sum(dtype=torch.int64).item()) == want_sum, shape
            step = 1 << 27
            for lo in range(0, n - 1, step):
                hi = min(lo + step + 1, n)
                assert bool((keys[lo + 1:hi] >= keys[lo:hi - 1]).all().item()), (shape, lo)
            del keys
    finally:
        eng.close()
