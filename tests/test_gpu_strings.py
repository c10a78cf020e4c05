"""GPU parity of the std::string key path against the oracle and the golden fixtures."""
import numpy as np
import pytest

import golden_cases as g
from radix_sort_amd import shapes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", shapes.SHAPES)
def test_strings_match_oracle(engine, oracle, shape):
    for n in (0, 1, 2, 10, 11, 100, 1000, 70000, 300000):
        blob, off = shapes.generate_strings(n, shape)
        perm = engine.sort_strings_host(blob, off, stable=True)
        want = oracle.sort_strings(blob, off, stable=True)
        assert np.array_equal(perm, want), (shape, n)
        strs = shapes.unpack_strings(blob, off)
        v = oracle.sort_strings(blob, off, stable=False)        # value path: same strings in the same order
        assert [strs[i] for i in perm.tolist()] == [strs[i] for i in v.tolist()]


def test_strings_golden(engine):
    for e in g.MANIFEST["strings"] + g.MANIFEST["edge_strings"]:
        blob, off = g.string_input(e)
        perm = engine.sort_strings_host(blob, off, stable=True)
        assert g.sha(perm) == e["perm_sha256"], e
        assert g.sorted_strings_sha(blob, off, perm) == e["sorted_sha256"], e


def test_long_and_tied_strings(engine, oracle):
    rng = np.random.Generator(np.random.PCG64(5))
    base = [bytes(rng.integers(0, 3, size=int(rng.integers(0, 60)), dtype=np.uint8)) for _ in range(4000)]
    strs = base + [b"\x00" * k for k in range(40)] + [b"prefix" * 9 + bytes([k % 7]) * (k % 5) for k in range(500)] + base[:1000]
    order = rng.permutation(len(strs))
    blob, off = shapes.pack_strings([strs[i] for i in order])
    perm = engine.sort_strings_host(blob, off, stable=True)
    assert np.array_equal(perm, oracle.sort_strings(blob, off, stable=True))
