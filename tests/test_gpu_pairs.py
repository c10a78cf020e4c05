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
