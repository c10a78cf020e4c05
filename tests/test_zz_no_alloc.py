"""rs_sort_keys_device_noalloc (include/rsort_hip.h): the entry point that allocates nothing, through the C-ABI.

Its output must equal rs_sort_keys_device's bit for bit, on two calls in a row that share the static device workspace, the
second with a smaller input (what the first call left behind must not leak into it).  Collected last (file name), so a
failure here hides no other test."""
import numpy as np
import pytest

import radix_sort_amd as rs
from radix_sort_amd import shapes

pytestmark = pytest.mark.gpu


def _both(engine, torch, host, xf):
    """(result of the allocating entry point, result of the no-alloc one) for one input."""
    dev = torch.device("cuda:0")
    a = torch.from_numpy(host.copy()).to(dev)
    b = torch.from_numpy(host.copy()).to(dev)
    alt = torch.empty_like(b)
    torch.cuda.synchronize()
    engine.sort_keys_ptr(a.data_ptr(), a.numel(), a.element_size(), xf)
    engine.synchronize()
    engine.sort_keys_noalloc_ptr(b.data_ptr(), alt.data_ptr(), b.numel(), b.element_size(), xf)
    engine.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


@pytest.mark.parametrize("name", ["u32", "u64", "f32", "i64", "u16"])
def test_noalloc_matches_allocating_entry_point_twice(engine, name):
    import torch
    dt, kb, xf = shapes.KEY_TYPES[name]
    first = shapes.generate(name, 3_000_001, "randomized")
    second = shapes.generate(name, 70_003, "duplicates")          # smaller, other shape: stale descriptors / counts would show
    for host in (first, second):
        want, got = _both(engine, torch, host.view(np.dtype(dt)) if host.dtype != np.dtype(dt) else host, xf)
        assert got.tobytes() == want.tobytes(), (name, host.size)


def test_noalloc_at_its_size_limit(engine):
    """8-byte keys at n = 2^25, the largest input the static workspace serves: every descriptor row it holds is in use."""
    import torch
    host = shapes.generate("u64", 1 << 25, "randomized", seed=9)
    want, got = _both(engine, torch, host, 0)
    assert got.tobytes() == want.tobytes()
    assert bool(np.all(got[:-1] <= got[1:])) and int(got.sum(dtype=np.uint64)) == int(host.sum(dtype=np.uint64))


def test_noalloc_small_sizes_and_ordered_input(engine):
    import torch
    dt, kb, xf = shapes.KEY_TYPES["i32"]
    for n, shape in ((0, "randomized"), (1, "randomized"), (2, "reverse_sorted"), (1000, "sorted"), (20_000, "reverse_sorted"), (16_385, "nearly_sorted")):
        host = shapes.generate("i32", n, shape)
        if n == 0:
            continue
        want, got = _both(engine, torch, host, xf)
        assert got.tobytes() == want.tobytes(), (n, shape)


def test_noalloc_limits(engine):
    import ctypes as C
    import torch
    lib = engine.lib
    t = torch.zeros(16, dtype=torch.int32, device="cuda:0")
    # beyond the static workspace: refused before anything is touched
    rc = lib.rs_sort_keys_device_noalloc(engine.h, C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), (1 << 25) + 1, 4, 0, 0, 32)
    assert rc != 0 and b"at most" in lib.rs_last_error(engine.h)
    # no alternate buffer: refused (it allocates nothing)
    rc = lib.rs_sort_keys_device_noalloc(engine.h, C.c_void_p(t.data_ptr()), C.c_void_p(None), 16, 4, 0, 0, 32)
    assert rc != 0 and b"alternate" in lib.rs_last_error(engine.h)
    engine.synchronize()
