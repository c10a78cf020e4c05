"""GPU parity of the stable key+payload sort (projected / composite path) against the oracle."""
import numpy as np
import pytest

from radix_sort_amd import shapes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(shapes.KEY_TYPES))
@pytest.mark.parametrize("shape", ["randomized", "duplicates", "reverse_sorted", "nearly_sorted", "sorted"])
def test_pairs_match_oracle(engine, oracle, name, shape):
    dt, kb, xf = shapes.KEY_TYPES[name]
    for n in (0, 1, 2, 100, 101, 8193, 100000, 300001):
        v = shapes.generate(name, n, shape)
        wk, wp = oracle.sort_pairs(v, xf, mt=True)
        gk, gp = engine.sort_pairs_host(v, transform=xf)
        assert gk.tobytes() == wk.tobytes() and np.array_equal(gp, wp), (name, shape, n)


@pytest.mark.parametrize("vdtype", [np.uint32, np.uint64])
def test_pairs_value_widths(engine, oracle, vdtype):
    v = shapes.generate("u32", 200_000, "duplicates")
    wk, wp = oracle.sort_pairs(v, 0, mt=True)
    gk, gv = engine.sort_pairs_host(v, np.arange(v.size, dtype=vdtype), transform=0)
    assert gk.tobytes() == wk.tobytes() and np.array_equal(gv.astype(np.uint64), wp)


def test_bit_range(engine):
    # begin_bit/end_bit: stable sort on a digit sub-range only
    rng = np.random.Generator(np.random.PCG64(3))
    v = rng.integers(0, 1 << 32, 50_000, dtype=np.uint32)
    for b, e in ((0, 8), (8, 24), (4, 15), (0, 32), (27, 32)):
        out = engine.sort_keys_host(v, 0, b, e)
        field = (v >> np.uint32(b)) & np.uint32((1 << (e - b)) - 1)
        want = v[np.argsort(field, kind="stable")]
        assert out.tobytes() == want.tobytes(), (b, e)


@pytest.mark.parametrize("name", ["u8", "i16", "u32", "i64", "f32", "f64"])
def test_partition_device_is_one_stable_pass(engine, name):
    """rs_partition_device: out = in stably reordered by ONE digit of the transformed key, raw bits unchanged, values
    along, digit counts returned -- against numpy (stable argsort of that digit)."""
    import torch
    dt, kb, xf = shapes.KEY_TYPES[name]
    top = kb * 8
    for n, shape in ((1, "randomized"), (1000, "randomized"), (70_001, "duplicates"), (300_000, "nearly_sorted")):
        v = shapes.generate(name, n, shape, seed=n)
        u = v.view({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[kb]).astype(np.uint64)
        sign = np.uint64(1) << np.uint64(top - 1)
        mask = np.uint64((1 << top) - 1) if kb < 8 else np.uint64(0xFFFFFFFFFFFFFFFF)
        t_ = u ^ sign if xf == 1 else (np.where((u >> np.uint64(top - 1)).astype(bool), ~u & mask, u ^ sign) if xf == 2 else u)
        for begin, end in ((top - 8, top), (0, min(8, top)), (max(0, top - 11), top - 3) if kb > 1 else (2, 7)):
            digit = ((t_ >> np.uint64(begin)) & np.uint64((1 << (end - begin)) - 1)).astype(np.int64)
            order = np.argsort(digit, kind="stable")
            d_in = torch.from_numpy(v.view(np.uint8).copy()).cuda()
            d_out = torch.empty_like(d_in)
            vals = torch.arange(n, dtype=torch.int32).cuda()
            vout = torch.empty_like(vals)
            counts = engine.partition_ptr(d_in.data_ptr(), d_out.data_ptr(), n, kb, xf, begin, end,
                                          vals.data_ptr(), vout.data_ptr(), 4)
            engine.synchronize()
            assert d_out.cpu().numpy().tobytes() == v[order].tobytes(), (name, n, shape, begin, end)
            assert np.array_equal(vout.cpu().numpy(), order.astype(np.int32)), (name, n, shape, begin, end)
            assert np.array_equal(counts.astype(np.int64), np.bincount(digit, minlength=256)), (name, n, shape, begin, end)
            k_out = torch.empty_like(d_in)
            engine.partition_ptr(d_in.data_ptr(), k_out.data_ptr(), n, kb, xf, begin, end)       # keys only
            engine.synchronize()
            assert k_out.cpu().numpy().tobytes() == v[order].tobytes(), (name, n, shape, begin, end)
