"""GPU parity, randomised: sizes, key types, shapes, bit ranges and payload widths drawn from a fixed seed; keys against the
oracle, bit-range and pair sorts against numpy's stable argsort of the selected field of the transformed key."""
import numpy as np
import pytest

from radix_sort_amd import shapes

pytestmark = pytest.mark.gpu

_U = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}


def _order_bits(v, kb, xf):
    u = v.view(_U[kb]).astype(np.uint64)
    top = kb * 8
    sign = np.uint64(1) << np.uint64(top - 1)
    mask = np.uint64((1 << top) - 1) if kb < 8 else np.uint64(0xFFFFFFFFFFFFFFFF)
    if xf == 1:
        return u ^ sign
    if xf == 2:
        return np.where((u >> np.uint64(top - 1)).astype(bool), ~u & mask, u ^ sign)
    return u


@pytest.mark.parametrize("chunk", range(8))
def test_random_cases(engine, oracle, chunk):
    import torch
    rng = np.random.Generator(np.random.PCG64(1000 + chunk))
    names = list(shapes.KEY_TYPES)
    for case in range(25):
        name = names[rng.integers(len(names))]
        dt, kb, xf = shapes.KEY_TYPES[name]
        shape = shapes.SHAPES[rng.integers(len(shapes.SHAPES))]
        n = int(rng.choice([rng.integers(2, 200), rng.integers(200, 20_000), rng.integers(20_000, 400_000)]))
        v = shapes.generate(name, n, shape, seed=int(rng.integers(1 << 30)))
        where = (name, shape, n, chunk, case)
        # whole-key sort against the oracle
        assert engine.sort_keys_host(v, xf).tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), where
        # a bit range, keys and (sometimes 8-byte) payload, against the stable order of that field
        top = kb * 8
        begin = int(rng.integers(0, top - 1))
        end = int(rng.integers(begin + 1, top + 1))
        field = (_order_bits(v, kb, xf) >> np.uint64(begin)) & np.uint64((1 << (end - begin)) - 1)
        order = np.argsort(field, kind="stable")
        vb = int(rng.choice([4, 8]))
        k = torch.from_numpy(v.view(np.uint8).copy()).cuda()
        p = torch.arange(n, dtype=torch.int32 if vb == 4 else torch.int64).cuda()
        engine.sort_pairs_ptr(k.data_ptr(), p.data_ptr(), n, kb, vb, xf, None, None, begin, end)
        engine.synchronize()
        assert k.cpu().numpy().tobytes() == v[order].tobytes(), where + (begin, end, vb)
        assert np.array_equal(p.cpu().numpy().astype(np.int64), order.astype(np.int64)), where + (begin, end, vb)
