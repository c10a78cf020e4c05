"""CPU: the five shapes keep the properties the reference's generators define (generators.hpp)."""
import numpy as np
import pytest

from radix_sort_amd import shapes


@pytest.mark.parametrize("name", list(shapes.KEY_TYPES))
def test_shape_properties(name):
    dt, kb, xf = shapes.KEY_TYPES[name]
    n = 4096
    s = shapes.generate(name, n, "sorted")
    r = shapes.generate(name, n, "reverse_sorted")
    assert s.dtype == dt and s.size == n
    assert np.all(s[:-1] <= s[1:]) and np.array_equal(r, s[::-1])
    lim = np.finfo(dt) if xf == 2 else np.iinfo(dt)
    assert s[-1] == lim.max and s[0] == lim.min
    d = shapes.generate(name, n, "duplicates")
    assert np.unique(d.view(np.uint8).reshape(n, kb), axis=0).shape[0] <= 256
    ns = shapes.generate(name, n, "nearly_sorted")
    assert np.array_equal(np.sort(ns.view(s.dtype)), np.sort(s)) or xf == 2
    assert (ns != s).sum() <= 2 * int(np.ceil(n * 0.05))
    assert shapes.generate(name, 0, "randomized").size == 0 and shapes.generate(name, 1, "sorted")[0] == 0


def test_string_shapes():
    blob, off = shapes.generate_strings(1000, "sorted")
    strs = shapes.unpack_strings(blob, off)
    assert strs[0] == b"" and strs[1] == b"\x00" and strs[257] == b"\x00\x00" and len(set(strs)) == 1000
    blob, off = shapes.generate_strings(5000, "randomized")
    lens = np.diff(off.astype(np.int64))
    assert lens.min() == 0 and lens.max() == 20
    blob, off = shapes.generate_strings(5000, "duplicates")
    assert len(set(shapes.unpack_strings(blob, off))) <= 256
