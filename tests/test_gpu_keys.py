"""GPU parity: the HIP sort (through the C-ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from radix_sort_amd import shapes

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 2, 10, 11, 63, 64, 65, 100, 101, 1000, 8191, 8192, 8193, 9999, 10001, 100000, 1 << 20]


@pytest.mark.parametrize("name", list(shapes.KEY_TYPES))
@pytest.mark.parametrize("shape", shapes.SHAPES)
def test_keys_match_oracle(engine, oracle, name, shape):
    dt, kb, xf = shapes.KEY_TYPES[name]
    for n in SIZES:
        v = shapes.generate(name, n, shape)
        want = oracle.sort_keys(v, xf, mt=True)
        got = engine.sort_keys_host(v, xf)
        assert got.tobytes() == want.tobytes(), (name, shape, n)


@pytest.mark.parametrize("name", ["u32", "u64", "f64", "i64", "f32"])
def test_keys_threshold_crossing(engine, oracle, name):
    # 2,000,001 crosses the reference's multi-threading threshold (radix_sort.hpp:114)
    dt, kb, xf = shapes.KEY_TYPES[name]
    v = shapes.generate(name, 2000001, "randomized")
    assert engine.sort_keys_host(v, xf).tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes()
