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
        assert np.array_equal(engine.sort_strings_host(blob, off, stable=True, narrow=False), want), (shape, n, "8-byte offsets")
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


def test_strings_that_survive_many_levels(engine, oracle):
    """Inputs the MSD loop does not resolve at its first level: a long shared prefix, the reference's shortlex "sorted"
    shape (generators.hpp:319-347: not in std::string order, so it is really sorted), few distinct strings (every level's
    segments are large and final only when the strings end), strings that differ in their last byte only; both entry
    points (4- and 8-byte offsets); the number of levels is what the chunk widths say."""
    rng = np.random.Generator(np.random.PCG64(77))
    n = 60_000
    cases = {}
    tails = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 12)), dtype=np.uint8)) for _ in range(n)]
    cases["24-byte prefix"] = [b"p" * 24 + t for t in tails]
    cases["100-byte prefix"] = [b"\x00\xff" * 50 + t for t in tails]
    vocab = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8)) for _ in range(256)]
    cases["256 distinct strings"] = [vocab[int(i)] for i in rng.integers(0, 256, n)]
    cases["last byte differs"] = [b"x" * 37 + bytes([int(i)]) for i in rng.integers(0, 256, n)]
    for what, strs in cases.items():
        blob, off = shapes.pack_strings(strs)
        want = oracle.sort_strings(blob, off, stable=True)
        for narrow in (True, False):
            assert np.array_equal(engine.sort_strings_host(blob, off, stable=True, narrow=narrow), want), (what, narrow)
        assert engine.last_string_levels() >= 2, what
    blob, off = shapes.generate_strings(50_000, "sorted")
    assert np.array_equal(engine.sort_strings_host(blob, off), oracle.sort_strings(blob, off, stable=True))
