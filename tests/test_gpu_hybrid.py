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
    engine.lib.rs_debug_set_distinct(engine.h, 0)          # (the duplicates shape would otherwise be sorted by counting its 256 values)
    try:
        for n in HYBRID_SIZES:
            v = shapes.generate(name, n, shape)
            got, hybrid = _sort_on_device(engine, v, xf)
            assert hybrid, "the hybrid form was not taken"
            assert got.tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), (name, shape, n)
    finally:
        engine.lib.rs_debug_set_distinct(engine.h, 1)
        engine.set_hybrid(-1)
        engine.set_precheck(True)


def test_hybrid_unaligned_and_big_groups(engine, oracle):
    """Key-aligned (not 16-byte aligned) buffers; groups larger than the LDS window go to the big-group kernel (forced at
    test size by a small window, and by keys that share their top 16 bits); low digits of two values (a constant digit's pass
    is skipped, and with fewer than four live passes the hybrid form is not taken)."""
    rng = np.random.Generator(np.random.PCG64(23))
    engine.set_precheck(False)
    engine.set_hybrid(1)
    engine.lib.rs_debug_set_distinct(engine.h, 0)
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
        engine.lib.rs_debug_set_distinct(engine.h, 1)
        engine.set_hybrid(-1)
        engine.set_precheck(True)


def test_hybrid_prefix_runs_in_wide_groups(engine):
    """8-byte keys: each group's 48 low bits are sorted by two passes on their top 16 and a fix-up of the keys that agree on
    those (tile_sort_body's top_bit).  Groups of thousands of keys (512 top-16 values) whose 16-bit prefixes take few values
    (runs beyond the fix-up's reach: the passes run again over all the bits), about a key's worth each, or many with few
    low parts (equal whole keys inside the runs)."""
    rng = np.random.Generator(np.random.PCG64(31))
    engine.set_precheck(False)
    engine.set_hybrid(1)
    engine.lib.rs_debug_set_distinct(engine.h, 0)
    try:
        n = 4_500_001
        top = (rng.integers(0, 32, n, dtype=np.uint64) << np.uint64(59)) | (rng.integers(0, 16, n, dtype=np.uint64) << np.uint64(48))
        for prefixes, step, lows in ((4, 0x5501, 1 << 32), (600, 97, 1 << 32), (4000, 13, 1 << 32), (3000, 21, 2), (1 << 16, 1, 1 << 32)):
            pre = (rng.integers(0, prefixes, n, dtype=np.uint64) * np.uint64(step) & np.uint64(0xFFFF)) << np.uint64(32)
            low = rng.integers(0, lows, n, dtype=np.uint64)
            if lows == 2:
                low = np.where(low == 0, np.uint64(0x01020304), np.uint64(0xF1E2D3C4))      # (every digit takes two values: a constant digit is no sort for the hybrid form)
            v = top | pre | low
            got, hybrid = _sort_on_device(engine, v, 0, 1)
            assert hybrid and got.tobytes() == np.sort(v).tobytes(), (prefixes, lows)
    finally:
        engine.lib.rs_debug_set_distinct(engine.h, 1)
        engine.set_hybrid(-1)
        engine.set_precheck(True)


@pytest.mark.parametrize("name", ["u32", "i32", "f32", "u64", "i64", "f64"])
def test_narrow_hybrid_matches_oracle(engine, oracle, name):
    """The narrow hybrid form (sorts between the mid-size regime and 10^8 keys: two sweeps on the top 8 + b bits, b from the
    size, then whole keys of every group sorted in LDS): every shape against the oracle, key-aligned buffers.  Shapes whose
    groups cannot fit (256 values, float ramps) or whose swept bits are constant take the plain loop -- same result."""
    dt, kb, xf = shapes.KEY_TYPES[name]
    engine.set_precheck(False)
    engine.set_hybrid(2)
    engine.lib.rs_debug_set_distinct(engine.h, 0)
    try:
        for n in (4_300_001, 9_999_999, (1 << 24) + 5):
            for shape in shapes.SHAPES:
                v = shapes.generate(name, n, shape)
                got, _ = _sort_on_device(engine, v, xf, 1 if n & 1 else 0)
                form = engine.lib.rs_debug_last_hybrid(engine.h)
                assert got.tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), (name, shape, n)
                if shape == "randomized":
                    assert form == 2, (name, n, form)
                # (duplicates, 256 values: groups of n / 256 keys -- the plain loop when the digit counts show it, else the
                # big-group kernel behind the narrow form: either way the same result)
    finally:
        engine.lib.rs_debug_set_distinct(engine.h, 1)
        engine.set_hybrid(-1)
        engine.set_precheck(True)


def test_narrow_hybrid_prefixes(engine):
    """Keys whose swept bits are constant (small integers), take few values (groups beyond the window: the plain loop, or the
    big-group kernel when only some are), or whose low parts do (equal keys; 8-byte keys: long runs for the fix-up)."""
    rng = np.random.Generator(np.random.PCG64(41))
    engine.set_precheck(False)
    engine.set_hybrid(2)
    engine.lib.rs_debug_set_distinct(engine.h, 0)
    try:
        n = 6_000_003
        for dtype, bits in ((np.uint32, 32), (np.uint64, 64)):
            full = rng.integers(0, 1 << 63, n, dtype=np.uint64).astype(dtype) if bits == 32 else rng.integers(0, 1 << 64, n, dtype=np.uint64)
            cases = {
                "small integers": (full >> dtype(bits - 20)),
                "top bits of 3 values": (full >> dtype(2)) | (rng.integers(0, 3, n, dtype=np.uint64).astype(dtype) << dtype(bits - 2)),
                "one heavy prefix": np.where(rng.random(n) < 0.02, full >> dtype(12), full),
                "low part of 5 values": (full & ~dtype((1 << (bits - 16)) - 1)) | rng.integers(0, 5, n, dtype=np.uint64).astype(dtype),
                "mid prefix of 4 values": (full & ~dtype(0xFFFF << (bits - 32))) | (rng.integers(0, 4, n, dtype=np.uint64).astype(dtype) << dtype(bits - 30)),
            }
            for what, v in cases.items():
                got, _ = _sort_on_device(engine, v, 0)
                assert got.tobytes() == np.sort(v).tobytes(), (what, dtype)
            engine.lib.rs_debug_set_group_limit(engine.h, 100)           # groups of more than 400 keys: the big-group kernel
            got, _ = _sort_on_device(engine, full, 0)
            assert got.tobytes() == np.sort(full).tobytes() and engine.lib.rs_debug_last_hybrid(engine.h) == 2, dtype
            engine.lib.rs_debug_set_group_limit(engine.h, 0)
        # as shipped (form chosen by the library): 8-byte keys of this size take the narrow form when the bits it sweeps differ
        # among the keys (a sample spread over the array says so: random keys, a ramp), not when they hold small integers
        engine.set_hybrid(-1)
        engine.lib.rs_debug_set_distinct(engine.h, 1)
        full = rng.integers(0, 1 << 64, n, dtype=np.uint64)
        for what, v, form in (("random", full, 2), ("ramp", np.sort(full), 2), ("small integers", full >> np.uint64(28), 0),
                              ("4-byte keys", full.astype(np.uint32), 0)):
            got, _ = _sort_on_device(engine, v, 0)
            assert got.tobytes() == np.sort(v).tobytes() and engine.lib.rs_debug_last_hybrid(engine.h) == form, what
    finally:
        engine.lib.rs_debug_set_group_limit(engine.h, 0)
        engine.lib.rs_debug_set_distinct(engine.h, 1)
        engine.set_hybrid(-1)
        engine.set_precheck(True)


def test_prefixed_hybrid(engine, oracle):
    """Keys that share their leading t bits (a multi-GPU rank's keys after the exchange): two sweeps on the 16 bits below
    the prefix, groups finished in LDS, the prefix put back.  Every t, 4- and 8-byte keys, the transforms; a prefix the
    sample believes in but one key breaks (the histogram notices: the plain loop); skewed bits below the prefix."""
    rng = np.random.Generator(np.random.PCG64(43))
    engine.set_precheck(False)
    engine.set_hybrid(3)
    try:
        n = 4_400_001
        for dtype, bits in ((np.uint32, 32), (np.uint64, 64)):
            full = rng.integers(0, 1 << 63, n, dtype=np.uint64).astype(dtype) if bits == 32 else rng.integers(0, 1 << 64, n, dtype=np.uint64)
            for t in range(1, 8):
                prefix = dtype(int(rng.integers(0, 1 << t)) << (bits - t))
                v = (full >> dtype(t)) | prefix
                got, _ = _sort_on_device(engine, v, 0, t & 1)
                assert got.tobytes() == np.sort(v).tobytes(), (dtype, t)
                assert engine.lib.rs_debug_last_hybrid(engine.h) == 3, (dtype, t)
            v = (full >> dtype(3)) | dtype(5 << (bits - 3))
            v[n // 2] = dtype(1)                                       # one key outside the prefix, not in the sample
            got, _ = _sort_on_device(engine, v, 0)
            assert got.tobytes() == np.sort(v).tobytes() and engine.lib.rs_debug_last_hybrid(engine.h) == 0, dtype
            v = ((full >> dtype(3)) | dtype(2 << (bits - 3))) & ~dtype(0x3F << (bits - 11))      # eight bits below the prefix of four values
            got, _ = _sort_on_device(engine, v, 0)
            assert got.tobytes() == np.sort(v).tobytes(), dtype
        for name in ("i32", "f32", "i64", "f64"):
            dt, kb, xf = shapes.KEY_TYPES[name]
            v = shapes.generate(name, n, "randomized")
            if name[0] == "i":
                v = (v >> 2) + (1 << (kb * 8 - 3))                    # positive, two leading bits shared
            else:
                v = np.abs(v)                                          # sign bit shared (NaN payloads and all)
            got, _ = _sort_on_device(engine, v, xf)
            assert got.tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), name
    finally:
        engine.set_hybrid(-1)
        engine.set_precheck(True)


def _sort_pairs_on_device(engine, v, xf, offset=0):
    """(v, 0..n-1) through rs_sort_pairs_device, 4-byte payload: (sorted keys, payload = the stable permutation, took the hybrid form)."""
    import torch
    kb = v.itemsize
    buf = torch.empty(v.size * kb + 64, dtype=torch.uint8, device="cuda")
    kview = buf[offset * kb: offset * kb + v.size * kb]
    kview.copy_(torch.from_numpy(v.view(np.uint8).copy()))
    vals = torch.arange(v.size, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    engine.sort_pairs_ptr(kview.data_ptr(), vals.data_ptr(), v.size, kb, 4, xf)
    engine.synchronize()
    took_hybrid = bool(engine.lib.rs_debug_last_hybrid(engine.h))
    return kview.cpu().numpy().view(v.dtype), vals.cpu().numpy().astype(np.uint64), took_hybrid


@pytest.mark.parametrize("name", ["u32", "i32", "f32", "u64", "i64", "f64"])
def test_hybrid_pair_form_matches_oracle(engine, oracle, name):
    """Stable pairs of 4- or 8-byte keys and a 4-byte payload in the hybrid form (two sweeps, then every group of equal top 16 bits
    sorted in LDS with the payload beside the keys): keys and permutation against the oracle's stable pair sort, every shape.
    Shapes whose groups do not fit the window (256 values: 16 K pairs per group) are handed back to the plain loop after the
    sweeps -- same result, and the library says it did not take the hybrid form."""
    dt, kb, xf = shapes.KEY_TYPES[name]
    engine.set_precheck(False)
    engine.set_hybrid(1)
    try:
        for n in (4_300_001, (1 << 23) + 7):
            for shape in shapes.SHAPES:
                v = shapes.generate(name, n, shape)
                for offset in (0, 1):
                    keys, perm, hybrid = _sort_pairs_on_device(engine, v, xf, offset)
                    wk, wp = oracle.sort_pairs(v, xf, mt=True)
                    assert keys.tobytes() == wk.tobytes() and np.array_equal(perm, wp), (name, shape, n, offset)
                    # (float ramps: a quarter of the keys share their top nine bits -- groups far beyond the window, the plain loop)
                    # (float32 ramps at these sizes: the largest groups hold about n / 512 pairs, at the edge of the window -- either form)
                    if name != "f32" or shape in ("randomized", "duplicates"):
                        assert hybrid == (shape != "duplicates" and (name != "f64" or shape == "randomized")), (name, shape, n)
    finally:
        engine.set_hybrid(-1)
        engine.set_precheck(True)


def test_hybrid_pair_form_prefix_runs_and_windows(engine):
    """The pair form's fix-up keeps equal keys in their order and moves the payload with its key: 16-bit prefixes of few values
    (runs beyond its reach: every pass again), of many with few low parts (equal whole keys inside the runs), groups at and just
    beyond the window (12288 pairs), a window forced small (everything handed back to the plain loop)."""
    rng = np.random.Generator(np.random.PCG64(37))
    engine.set_precheck(False)
    engine.set_hybrid(1)
    try:
        n = 4_500_001
        top = (rng.integers(0, 32, n, dtype=np.uint64) << np.uint64(59)) | (rng.integers(0, 16, n, dtype=np.uint64) << np.uint64(48))
        for prefixes, step, lows in ((4, 0x5501, 1 << 32), (600, 97, 1 << 32), (3000, 21, 2), (1 << 16, 1, 1 << 32)):
            pre = (rng.integers(0, prefixes, n, dtype=np.uint64) * np.uint64(step) & np.uint64(0xFFFF)) << np.uint64(32)
            low = rng.integers(0, lows, n, dtype=np.uint64)
            if lows == 2:
                low = np.where(low == 0, np.uint64(0x01020304), np.uint64(0xF1E2D3C4))
            v = top | pre | low
            keys, perm, hybrid = _sort_pairs_on_device(engine, v, 0)
            order = np.argsort(v, kind="stable").astype(np.uint64)
            assert hybrid and keys.tobytes() == v[order].tobytes() and np.array_equal(perm, order), (prefixes, lows)
        # 366 groups of about 12288 pairs: some fit the window, some do not -> the plain loop; 420 groups: all fit;
        # 900 / 1800 groups: they fit the windows of the 512- / 256-thread workgroups (6144 / 3072 pairs)
        for groups, expect in ((366, False), (420, True), (900, True), (1800, True)):
            tops = rng.integers(0, groups, n, dtype=np.uint64)
            v = ((tops // np.uint64(16)) << np.uint64(56)) | ((tops % np.uint64(16)) << np.uint64(48)) | rng.integers(0, 1 << 48, n, dtype=np.uint64)
            keys, perm, hybrid = _sort_pairs_on_device(engine, v, 0)
            order = np.argsort(v, kind="stable").astype(np.uint64)
            assert keys.tobytes() == v[order].tobytes() and np.array_equal(perm, order), groups
            assert hybrid == expect, groups
        engine.lib.rs_debug_set_group_limit(engine.h, 2)
        v = rng.integers(0, 1 << 64, n, dtype=np.uint64)
        keys, perm, hybrid = _sort_pairs_on_device(engine, v, 0)
        order = np.argsort(v, kind="stable").astype(np.uint64)
        assert not hybrid and keys.tobytes() == v[order].tobytes() and np.array_equal(perm, order)
    finally:
        engine.lib.rs_debug_set_group_limit(engine.h, 0)
        engine.set_hybrid(-1)
        engine.set_precheck(True)


def test_hybrid_wide_groups_by_workgroup_size(engine):
    """8-byte keys only: the group kernel in workgroups of 256, 512 and 1024 threads (windows of 4352 / 8704 / 17408 keys, picked
    from the bound the digit counts give), groups around each window's edge (what does not fit goes to the big-group kernel),
    low parts that exercise the fix-up; both hybrid forms."""
    rng = np.random.Generator(np.random.PCG64(43))
    engine.set_precheck(False)
    engine.lib.rs_debug_set_distinct(engine.h, 0)
    try:
        n = 4_500_001
        for mode in (1, 2):
            engine.set_hybrid(mode)
            for groups in (1300, 1100, 700, 530, 400, 270):
                tops = rng.integers(0, groups, n, dtype=np.uint64)
                if mode == 1:
                    v = (tops * np.uint64(65536 // groups) << np.uint64(48)) | rng.integers(0, 1 << 48, n, dtype=np.uint64)
                else:      # narrow form: 2^11 groups at this size, of which `groups` * 2048 / 1300 ... hold keys
                    v = ((tops * np.uint64(2048 // groups + 1) & np.uint64(2047)) << np.uint64(53)) | rng.integers(0, 1 << 53, n, dtype=np.uint64)
                for xf in (0, 1):
                    keys, _ = _sort_on_device(engine, v, xf)
                    want = np.sort(v) if xf == 0 else np.sort(v.view(np.int64)).view(np.uint64)
                    assert keys.tobytes() == want.tobytes(), (mode, groups, xf)
                    assert int(engine.lib.rs_debug_last_hybrid(engine.h)) in (mode, 0), (mode, groups)
    finally:
        engine.lib.rs_debug_set_distinct(engine.h, 1)
        engine.set_hybrid(-1)
        engine.set_precheck(True)


@pytest.mark.parametrize("kb", [4, 8])
def test_hybrid_keys_of_few_thousand_values(engine, kb):
    """Keys that take a few hundred to a few thousand distinct values (beyond the few-distinct path, which is off here): whole
    values land in single groups far beyond any window.  A group that is one key repeated is left as it is (and rewritten when
    the keys are stored transformed), groups of two or three values go down the lists -- small workgroups, large ones, the
    big-group kernel; every form, every transform, as shipped too."""
    rng = np.random.Generator(np.random.PCG64(47))
    dt = np.uint32 if kb == 4 else np.uint64
    engine.set_precheck(False)
    engine.lib.rs_debug_set_distinct(engine.h, 0)
    try:
        n = 4_600_003
        for distinct in (300, 2500, 30000):
            pool = rng.integers(0, 1 << (8 * kb), distinct, dtype=np.uint64).astype(dt)
            pool[: distinct // 2] &= dt((1 << (8 * kb)) - 1 - 0xFFFF) if kb == 4 else dt(0xFFFFFFFF00000000)   # (values that share their top bits: groups of several values)
            v = pool[rng.integers(0, distinct, n)]
            for mode in (-1, 1, 2):
                engine.set_hybrid(mode)
                for xf in (0, 1, 2):
                    got, _ = _sort_on_device(engine, v, xf, 1)
                    bits = v.astype(np.uint64)
                    sign = np.uint64(1 << (8 * kb - 1))
                    allm = np.uint64((1 << (8 * kb)) - 1) if kb == 4 else np.uint64(0xFFFFFFFFFFFFFFFF)
                    rank = bits if xf == 0 else bits ^ sign if xf == 1 else np.where(bits & sign, bits ^ allm, bits | sign)
                    want = v[np.argsort(rank, kind="stable")]
                    assert got.tobytes() == want.tobytes(), (kb, distinct, mode, xf)
    finally:
        engine.lib.rs_debug_set_distinct(engine.h, 1)
        engine.set_hybrid(-1)
        engine.set_precheck(True)


def test_hybrid_pair_form_of_4_byte_keys_windows(engine):
    """Pairs of 4-byte keys: the group kernel in workgroups of 256 threads (largest group at most 4352 pairs), of 512 (8704),
    the plain loop beyond; low 16 bits of few values (equal keys inside every group: the payload says whether they kept their
    order), every transform, a key buffer that is only 4-byte aligned."""
    rng = np.random.Generator(np.random.PCG64(41))
    engine.set_precheck(False)
    engine.set_hybrid(1)
    try:
        n = 4_500_001
        for groups, lows, expect in ((1300, 1 << 16, True), (1300, 3, True), (700, 1 << 16, True), (700, 300, True), (540, 1 << 16, True), (400, 1 << 16, False)):
            tops = rng.integers(0, groups, n, dtype=np.uint32)
            low = rng.integers(0, lows, n, dtype=np.uint32) * np.uint32(0x9E37 if lows < (1 << 16) else 1) & np.uint32(0xFFFF)
            v = (tops * np.uint32(65536 // groups) << np.uint32(16)) | low
            for xf, offset in ((0, 0), (1, 1), (2, 3)):
                keys, perm, hybrid = _sort_pairs_on_device(engine, v, xf, offset)
                bits = v.astype(np.int64)
                rank = bits if xf == 0 else (bits ^ 0x80000000) if xf == 1 else np.where(bits & 0x80000000, bits ^ 0xFFFFFFFF, bits | 0x80000000)
                order = np.argsort(rank, kind="stable").astype(np.uint64)
                assert keys.tobytes() == v[order].tobytes() and np.array_equal(perm, order), (groups, lows, xf, offset)
                assert hybrid == expect, (groups, lows, xf)
    finally:
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
orc = load_oracle(); eng = rs.Engine(0); eng.set_precheck(False); eng.set_hybrid(1); eng.lib.rs_debug_set_distinct(eng.h, 0)
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


def test_hybrid_hands_back_to_plain_loop_when_groups_do_not_fit():
    """Sorts of 5 * 10^8 keys and more read their groups' real sizes back after the sweeps and, with more than 1/64 of the keys
    in groups beyond the window, hand the array -- a stable rearrangement of the input, keys transformed -- to the plain loop.
    Here at test size (the threshold is an environment switch read once per process: a child process), keys of 300 and
    2500 distinct values, every transform, both widths, as shipped; few and many values for the path that stays hybrid."""
    code = r'''
import numpy as np, torch
import radix_sort_amd as rs
eng = rs.Engine(0)
eng.set_precheck(False)
eng.lib.rs_debug_set_distinct(eng.h, 0)
rng = np.random.Generator(np.random.PCG64(53))
n = 4_700_001
back = stay = 0
for kb in (4, 8):
    dt = np.uint32 if kb == 4 else np.uint64
    for distinct in (300, 2500, 1 << 22):
        v = rng.integers(0, 1 << (8 * kb), distinct, dtype=np.uint64).astype(dt)[rng.integers(0, distinct, n)]
        for mode in (-1,):
            eng.set_hybrid(mode)
            for xf in (0, 1, 2):
                t = torch.from_numpy(v.view(np.uint8).copy()).cuda()
                torch.cuda.synchronize()
                eng.sort_keys_ptr(t.data_ptr(), n, kb, xf); eng.synchronize()
                form = int(eng.lib.rs_debug_last_hybrid(eng.h))
                bits = v.astype(np.uint64)
                sign = np.uint64(1 << (8 * kb - 1))
                allm = np.uint64((1 << (8 * kb)) - 1) if kb == 4 else np.uint64(0xFFFFFFFFFFFFFFFF)
                rank = bits if xf == 0 else bits ^ sign if xf == 1 else np.where(bits & sign, bits ^ allm, bits | sign)
                want = v[np.argsort(rank, kind="stable")]
                assert t.cpu().numpy().view(dt).tobytes() == want.tobytes(), (kb, distinct, xf, form)
                if kb == 8:                       # (4-byte keys of this size take the plain loop anyway)
                    back += form == 0 and distinct <= 2500
                    stay += form != 0 and distinct > 2500
assert back >= 3 and stay >= 3, (back, stay)
print("HAND BACK OK", back, stay)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSORT_GROUP_CHECK_MIN="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "HAND BACK OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


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
            assert bool(eng.lib.rs_debug_last_distinct(eng.h)) == (shape == "duplicates"), shape
            assert int(keys.sum(dtype=torch.int64).item()) == want_sum, shape
            step = 1 << 27
            for lo in range(0, n - 1, step):
                hi = min(lo + step + 1, n)
                assert bool((keys[lo + 1:hi] >= keys[lo:hi - 1]).all().item()), (shape, lo)
            del keys
    finally:
        eng.close()


def test_few_distinct_keys(engine, oracle):
    """Large keys-only sorts whose keys take few distinct values are sorted by counting them (rsd:: in rsort_groups.hpp: one
    read, one write): the reference's duplicates shape for every 4- and 8-byte type, one and two values, the most the
    table serves (1024) and one more (back to the usual sort), a few-valued head in front of random keys (the sample
    says yes, the full count says no), the key that is all ones (the table's "empty" mark), unaligned buffers."""
    rng = np.random.Generator(np.random.PCG64(99))
    engine.set_precheck(False)
    try:
        n = 5_000_011
        for name in ("u32", "i32", "f32", "u64", "i64", "f64"):
            dt, kb, xf = shapes.KEY_TYPES[name]
            v = shapes.generate(name, n, "duplicates")
            for offset in (0, 1):
                got, _ = _sort_on_device(engine, v, xf, offset)
                assert engine.lib.rs_debug_last_distinct(engine.h) == 1
                assert got.tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), (name, offset)
        for count, expect in ((1, 1), (2, 1), (1024, 1), (1025, 0), (5000, 0)):
            vals = rng.integers(0, 1 << 64, count, dtype=np.uint64)
            vals[0] = np.uint64(0xFFFFFFFFFFFFFFFF)                     # all ones: counted beside the table
            if count >= 2:
                vals[1] = np.uint64(0)
            for dtype in (np.uint32, np.uint64):
                pool = np.unique(vals.astype(dtype))
                while pool.size < count:                                  # (the narrowing may have merged values)
                    pool = np.unique(np.concatenate((pool, rng.integers(0, 1 << 32, count - pool.size, dtype=np.uint64).astype(dtype))))
                v = pool[rng.integers(0, pool.size, n)]
                v[:pool.size] = pool                                      # every value occurs
                if count == 1:
                    v[-1] = dtype(7)                                      # (one descent, or a constant array never reaches the sort engine's passes)
                got, _ = _sort_on_device(engine, v, 0)
                assert got.tobytes() == np.sort(v).tobytes(), (count, dtype)
                if count != 1:
                    assert engine.lib.rs_debug_last_distinct(engine.h) == expect, (count, dtype)
        v = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        v[:400_000] = v[:400_000] & np.uint32(3)                          # the sample sees four values
        got, _ = _sort_on_device(engine, v, 0)
        assert engine.lib.rs_debug_last_distinct(engine.h) == 0 and got.tobytes() == np.sort(v).tobytes()
    finally:
        engine.set_precheck(True)
