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


@pytest.mark.parametrize("name", ["u32", "i32", "f32", "u64", "f64", "u16"])
@pytest.mark.parametrize("shape", ["sorted", "reverse_sorted", "nearly_sorted"])
def test_ordered_shapes_without_the_precheck(engine, oracle, name, shape):
    """The order pre-check answers sorted / reverse inputs by itself; switch it off so the sort engine proper
    (histogram peel on skewed digits, look-back on one-bucket passes) is what gets compared."""
    dt, kb, xf = shapes.KEY_TYPES[name]
    engine.set_precheck(False)
    try:
        for n in (1000, 12288, 12289, 100000, 1 << 20):
            v = shapes.generate(name, n, shape)
            assert engine.sort_keys_host(v, xf).tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), (name, shape, n)
    finally:
        engine.set_precheck(True)


def test_precheck_paths(engine, oracle):
    """ascending -> untouched; descending -> reversed (isSortedBi, radix_sort.hpp:296-302); all equal -> untouched."""
    for name in ("u32", "i64", "f64"):
        dt, kb, xf = shapes.KEY_TYPES[name]
        for shape in ("sorted", "reverse_sorted"):
            v = shapes.generate(name, 300_000, shape)
            assert engine.sort_keys_host(v, xf).tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes()
    v = np.full(100_000, 7, dtype=np.uint32)
    assert engine.sort_keys_host(v, 0).tobytes() == v.tobytes()
    # non-strictly descending with ties, and NaNs of both signs in float order
    v = np.repeat(np.arange(5000, 0, -1, dtype=np.int32), 3)
    assert engine.sort_keys_host(v, 1).tobytes() == np.sort(v).tobytes()


@pytest.mark.parametrize("switch", ["RSORT_FORCE_WIDE=1", "RSORT_XCD=0", "RSORT_RANK=ballot",
                                    "RSORT_CHAIN_PROBE=fail,RSORT_CHAIN_MIN_TILES=16", "RSORT_CHAIN_MIN_TILES=16"])
def test_alternate_paths(oracle, switch):
    """Paths the product takes only in special situations, forced at test sizes (a child process each, because the
    switches are read once per process): RSORT_FORCE_WIDE=1 -- 8-byte look-back descriptors and offsets (n > 2^30);
    RSORT_XCD=0 -- one ticket counter instead of one per XCD (what a failed XCC-id probe selects);
    RSORT_RANK=ballot -- ballot/bitop3 ranking (what a failed LDS lane-order probe selects);
    RSORT_CHAIN_PROBE=fail -- chain-mode descriptors published with agent-scope stores (what a failed chain probe of
    rs_ctx_create selects), with chain mode itself forced at test sizes by RSORT_CHAIN_MIN_TILES; the same without the
    forced failure checks that the probe PASSES on this device (the context then reports workgroup-scope stores)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import radix_sort_amd as rs
from radix_sort_amd import shapes
from cpu_checkers import load_oracle
orc = load_oracle(); eng = rs.Engine(0); eng.set_precheck(False)
if "RSORT_CHAIN_MIN_TILES" in os.environ:
    assert eng.lib.rs_debug_desc_local(eng.h) == (0 if os.environ.get("RSORT_CHAIN_PROBE") == "fail" else 1), "chain probe"
for name in ("u32", "u64", "f64", "i16"):
    dt, kb, xf = shapes.KEY_TYPES[name]
    for shape in ("randomized", "duplicates", "sorted"):
        for n in (1000, 16385, 300001, 1 << 21):
            v = shapes.generate(name, n, shape)
            assert eng.sort_keys_host(v, xf).tobytes() == orc.sort_keys(v, xf, mt=True).tobytes(), (name, shape, n)
k = shapes.generate("i32", 200001, "duplicates")
gk, gp = eng.sort_pairs_host(k)
wk, wp = orc.sort_pairs(k, 1, mt=True)
assert gk.tobytes() == wk.tobytes() and (gp == wp).all()
print("WIDE OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **dict(kv.split("=") for kv in switch.split(",")))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WIDE OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_pass_skipping_keeps_results(engine, oracle):
    """Keys that agree on whole bytes (small values, one constant byte, a single distinct value): the passes
    whose digit is the same for every key are not launched; first/last-pass transforms must still be applied."""
    rng = np.random.Generator(np.random.PCG64(17))
    engine.set_precheck(False)
    try:
        cases = [
            ("u64", rng.integers(0, 1000, 200_000, dtype=np.uint64)),                       # only 2 live passes
            ("u64", (rng.integers(0, 256, 200_000, dtype=np.uint64) << np.uint64(40))),     # one live pass in the middle
            ("u32", rng.integers(0, 256, 100_000, dtype=np.uint32) | np.uint32(0xAB00CD00)),
            ("i64", rng.integers(-500, 500, 300_001, dtype=np.int64)),                      # sign flip on a skipped top byte? (top byte varies: 0x00/0xFF)
            ("i32", rng.integers(0, 100, 50_000, dtype=np.int32)),                          # top byte constant AFTER the sign flip
            ("f64", rng.integers(1, 200, 100_000).astype(np.float64)),
            ("f32", -rng.integers(1, 200, 100_000).astype(np.float32)),
            ("u32", np.full(70_000, 123456, dtype=np.uint32)),                              # no live pass at all
            ("f64", np.full(70_000, -2.5, dtype=np.float64)),
        ]
        for name, v in cases:
            dt, kb, xf = shapes.KEY_TYPES[name]
            assert engine.sort_keys_host(v, xf).tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), name
            wk, wp = oracle.sort_pairs(v, xf, mt=True)
            gk, gp = engine.sort_pairs_host(v, transform=xf)
            assert gk.tobytes() == wk.tobytes() and np.array_equal(gp, wp), name
    finally:
        engine.set_precheck(True)


@pytest.mark.parametrize("name", ["u8", "u16", "i32", "f64"])
def test_histogram_of_key_aligned_subranges(engine, name):
    """rs_digit_histogram_device on pointers that are key-aligned but not 16-byte aligned (the multi-GPU splitter
    passes such sub-ranges), against numpy digit counts of the transformed keys."""
    import torch
    dt, kb, xf = shapes.KEY_TYPES[name]
    v = shapes.generate(name, 50_000, "randomized", seed=5)
    u = v.view({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[kb]).astype(np.uint64)
    sign = np.uint64(1) << np.uint64(kb * 8 - 1)
    if xf == 1:
        u = u ^ sign
    elif xf == 2:
        neg = (u >> np.uint64(kb * 8 - 1)).astype(bool)
        u = np.where(neg, ~u & np.uint64((1 << (kb * 8)) - 1 if kb < 8 else 0xFFFFFFFFFFFFFFFF), u ^ sign)
    t = torch.from_numpy(v.view(np.uint8).copy()).cuda()
    for lo, hi in [(0, 50_000), (1, 50_000), (3, 17), (5, 40_001), (7, 8), (13, 4109)]:
        h = engine.digit_histogram_ptr(t.data_ptr() + lo * kb, hi - lo, kb, xf, 0, kb * 8)
        for p in range(kb):
            want = np.bincount(((u[lo:hi] >> np.uint64(8 * p)) & np.uint64(255)).astype(np.int64), minlength=256)
            assert (np.asarray(h[p]).astype(np.int64) == want).all(), (name, lo, hi, p)


def test_mid_size_sorts_between_other_calls(engine, oracle):
    """Mid-size device sorts launch no memset: the histogram kernel clears the control block for them and for the sort
    after them (two copies of the digit counts, used in turn).  One context, mid-size sorts interleaved with everything
    else that writes the control block -- large sorts, pair sorts, the histogram and partition entry points, the
    pre-check of the host-buffer calls -- every result against the oracle."""
    import torch
    rng = np.random.Generator(np.random.PCG64(23))
    engine.set_precheck(False)

    def mid(name, n, shape="randomized", seed=1):
        dt, kb, xf = shapes.KEY_TYPES[name]
        v = shapes.generate(name, n, shape, seed=seed)
        t = torch.from_numpy(v.view(np.uint8).copy()).cuda()
        engine.sort_keys_ptr(t.data_ptr(), n, kb, xf)
        engine.synchronize()
        assert t.cpu().numpy().tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), (name, n, shape)

    try:
        for k in range(3):
            mid("u32", 100_003 + k, seed=k)                # copies 0, 1, 0 of the counts
        mid("u64", 300_001)
        mid("u16", 70_000)                                  # 2 passes: the other passes' rows of the counts stay clear
        mid("f64", 250_000, "duplicates")
        mid("u32", 5_000_000)                               # not mid-size: memset, scan kernel, pass skipping read-back
        mid("u32", 4_000_000)                               # the largest mid-size sort (245 tiles)
        mid("i32", 33_333)
        v = shapes.generate("u32", 200_000, "randomized", seed=9)
        t = torch.from_numpy(v.view(np.uint8).copy()).cuda()
        h = engine.digit_histogram_ptr(t.data_ptr(), 200_000, 4, 0, 0, 32)
        assert (np.asarray(h[3]).astype(np.int64) == np.bincount((v >> 24).astype(np.int64), minlength=256)).all()
        mid("u32", 150_000, seed=4)
        mid("i64", 150_000, seed=5)
        out = torch.empty_like(t)
        engine.partition_ptr(t.data_ptr(), out.data_ptr(), 200_000, 4, 0, 24, 32)
        engine.synchronize()
        got = out.cpu().numpy().view(np.uint32)
        assert (got == v[np.argsort(v >> 24, kind="stable")]).all()
        mid("u32", 150_000, seed=6)
        mid("f32", 1_000_000, "nearly_sorted")
        # the short form of mid-size sorts (one sweep on the top byte, groups of equal top byte finished in LDS) and the
        # inputs that must NOT take it -- decided on the device, with the kernels of both forms queued: keys that share their
        # top byte (one group of everything), few top bytes (groups beyond the window), a window's worth exactly, every shape
        for name in ("u32", "i32", "f32", "u64", "i64", "f64"):
            for shape in shapes.SHAPES:
                mid(name, 777_777, shape, seed=3)
        for n in (20_000, 300_000, 2_000_000):
            dt = np.uint32
            small = rng.integers(0, 1 << 20, n, dtype=np.uint64).astype(dt)             # top byte constant
            four = (rng.integers(0, 4, n, dtype=np.uint64).astype(dt) << dt(30)) | rng.integers(0, 1 << 24, n, dtype=np.uint64).astype(dt)
            skew = np.where(rng.random(n) < 0.7, dt(0x7F000000), dt(0)) | rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(dt) >> dt(4)
            for v in (small, four, skew, small.astype(np.uint64) << np.uint64(20)):
                t = torch.from_numpy(v.view(np.uint8).copy()).cuda()
                engine.sort_keys_ptr(t.data_ptr(), n, v.itemsize, 0)
                engine.synchronize()
                assert t.cpu().numpy().tobytes() == np.sort(v).tobytes(), (n, v.dtype)
        # the groups' sort: two passes on the top 16 of the remaining bits, then the keys that agree on them put in order by
        # comparison (tile_sort_body's top_bit): prefixes of few values (runs beyond the fix-up's reach: the passes run again),
        # of many (runs of a few keys, equal whole keys among them), of one
        for n in (300_000, 1_000_000, 3_000_000):
            for dt, bits in ((np.uint32, 32), (np.uint64, 64)):
                top = rng.integers(0, 256, n, dtype=np.uint64) << np.uint64(bits - 8)
                low_w = bits - 24                                                    # bits below the 16-bit prefix
                for prefixes, lows in ((4, 1 << low_w), (1, 1 << low_w), (n // 1024, 1 << low_w), (n // 512, 3), (1 << 16, 1 << low_w)):
                    pre = rng.integers(0, max(prefixes, 1), n, dtype=np.uint64) * np.uint64(max((1 << 16) // max(prefixes, 1), 1)) << np.uint64(low_w)
                    v = (top | pre | rng.integers(0, lows, n, dtype=np.uint64)).astype(dt)
                    t = torch.from_numpy(v.view(np.uint8).copy()).cuda()
                    engine.sort_keys_ptr(t.data_ptr(), n, v.itemsize, 0)
                    engine.synchronize()
                    assert t.cpu().numpy().tobytes() == np.sort(v).tobytes(), (n, dt, prefixes, lows)
        # host-buffer calls: pre-check read-back before the histogram, pairs through the same control block
        engine.set_precheck(True)
        a = shapes.generate("u32", 120_000, "sorted", seed=1)
        assert engine.sort_keys_host(a, 0).tobytes() == a.tobytes()
        b = shapes.generate("u64", 120_000, "randomized", seed=2)
        wk, wp = oracle.sort_pairs(b, 0, mt=True)
        gk, gp = engine.sort_pairs_host(b, transform=0)
        assert gk.tobytes() == wk.tobytes() and np.array_equal(gp, wp)
        engine.set_precheck(False)
        mid("u32", 100_000, seed=7)
        mid("u64", 2_000_000, seed=8)
    finally:
        engine.set_precheck(True)


@pytest.mark.parametrize("entry", ["rs_sort_keys_device", "rs_sort_keys_device_noalloc"])
def test_mid_size_sort_replayed_from_a_graph(oracle, entry):
    """A mid-size device sort queues its kernels and returns, so it can be captured into a HIP graph once its buffers
    exist; the entry point that allocates nothing is meant for exactly that.  Replays must neither depend on host state
    of the capture (which copy of the digit counts was clear then) nor upset the ordinary sorts that run between them:
    replays and calls interleaved, with other kernels running on the device in between (with only copies in between,
    memset nodes of an earlier version replayed fine; with kernels they stopped clearing the tickets), every result
    against the oracle."""
    import torch
    import radix_sort_amd as rs
    n = 300_000
    eng = rs.Engine(0)
    if entry == "rs_sort_keys_device":
        def sort(d_keys, d_alt): eng.sort_keys_ptr(d_keys, n, 4, 0, d_alt=d_alt)
    else:
        def sort(d_keys, d_alt): eng.sort_keys_noalloc_ptr(d_keys, d_alt, n, 4, 0)
    try:
        eng.set_precheck(False)
        side = torch.cuda.Stream()
        buf = torch.empty(n, dtype=torch.int32, device="cuda")
        alt = torch.empty_like(buf)
        eng.set_stream(side.cuda_stream)
        with torch.cuda.stream(side):
            buf.random_(0, 1 << 30)
        side.synchronize()
        sort(buf.data_ptr(), alt.data_ptr())      # allocates the control block, loads the kernels
        eng.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            sort(buf.data_ptr(), alt.data_ptr())
        # replays and ordinary calls in every order: r r r c r c c r r c
        for step, how in enumerate("rrrcrccrrc"):
            v = shapes.generate("u32", n, "duplicates" if step % 4 == 3 else "randomized", seed=step)
            buf.copy_(torch.from_numpy(v.view(np.int32)))
            torch.cuda.synchronize()
            if how == "r":
                graph.replay()                      # (runs on torch's current stream)
                torch.cuda.synchronize()
            else:
                sort(buf.data_ptr(), alt.data_ptr())
                eng.synchronize()
            busy = torch.sort(buf.to(torch.int64)).values.sum().item()      # other kernels on the device between the steps
            assert buf.cpu().numpy().view(np.uint32).tobytes() == oracle.sort_keys(v, 0, mt=True).tobytes(), (step, how, busy)
        eng.synchronize()                           # no fault left behind by a replay
    finally:
        eng.close()


def test_small_inputs_down_the_general_path(oracle):
    """Inputs of at most one tile normally take rsk::small_sort_kernel (one workgroup, everything in LDS); with
    RSORT_SMALL=0 they go down the general path (histogram, tickets, look-back) instead, which must agree too.
    A child process, because the switch is read once per process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import radix_sort_amd as rs
from radix_sort_amd import shapes
from cpu_checkers import load_oracle
orc = load_oracle(); eng = rs.Engine(0)
for name in ("u8", "i16", "u32", "i64", "f32", "f64"):
    dt, kb, xf = shapes.KEY_TYPES[name]
    for shape in ("randomized", "duplicates", "reverse_sorted"):
        for n in (2, 3, 100, 101, 1000, 8191, 8192, 16384):
            v = shapes.generate(name, n, shape)
            assert eng.sort_keys_host(v, xf).tobytes() == orc.sort_keys(v, xf, mt=True).tobytes(), (name, shape, n)
    k = shapes.generate(name, 5000, "duplicates")
    gk, gp = eng.sort_pairs_host(k, transform=xf)
    wk, wp = orc.sort_pairs(k, xf, mt=True)
    assert gk.tobytes() == wk.tobytes() and (gp == wp).all(), name
print("GENERAL OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSORT_SMALL="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "GENERAL OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("name", list(shapes.KEY_TYPES))
def test_small_kernel_sizes_and_bit_ranges(engine, oracle, name):
    """The single-workgroup kernel at its edges: every size class up to one full tile, partial bit ranges, stable pairs."""
    import torch
    dt, kb, xf = shapes.KEY_TYPES[name]
    tile = 16384 if kb <= 4 else 8192
    for n in (2, 63, 64, 65, 1023, 1024, 1025, tile - 1, tile):
        for shape in ("randomized", "duplicates", "nearly_sorted"):
            v = shapes.generate(name, n, shape, seed=n)
            assert engine.sort_keys_host(v, xf).tobytes() == oracle.sort_keys(v, xf, mt=True).tobytes(), (name, shape, n)
        k = shapes.generate(name, n, "duplicates", seed=n + 1)
        gk, gp = engine.sort_pairs_host(k, transform=xf)
        wk, wp = oracle.sort_pairs(k, xf, mt=True)
        assert gk.tobytes() == wk.tobytes() and np.array_equal(gp, wp), (name, n)
    if xf == 0 and kb >= 2:                         # a bit range inside the key: stable on the selected field only
        v = shapes.generate(name, 5000, "randomized", seed=3)
        t = torch.from_numpy(v.view(np.uint8).copy()).cuda()
        engine.sort_keys_ptr(t.data_ptr(), v.size, kb, 0, None, 4, kb * 8 - 3)
        engine.synchronize()
        field = (v.astype(np.uint64) >> np.uint64(4)) & np.uint64((1 << (kb * 8 - 7)) - 1)
        want = v[np.argsort(field, kind="stable")]
        assert t.cpu().numpy().tobytes() == want.tobytes(), name


@pytest.mark.parametrize("name,n", [("u32", 1024 * 16384 + 12345), ("i64", 1030 * 8192 + 77), ("f32", 1087 * 16384)])
def test_per_xcd_tile_order_sizes(engine, oracle, name, n):
    """Sorts of at least 1024 tiles draw their tiles from one ticket counter per XCD (groups of 8 consecutive tiles);
    sizes that leave a partial last group make workgroups draw from other XCDs' counters.  Also with the order
    switched off through rs_ctx_set_tile_order."""
    dt, kb, xf = shapes.KEY_TYPES[name]
    v = shapes.generate(name, n, "randomized", seed=11)
    want = oracle.sort_keys(v, xf, mt=True).tobytes()
    engine.set_precheck(False)
    try:
        assert engine.sort_keys_host(v, xf).tobytes() == want
        engine.set_tile_order(False)
        assert engine.sort_keys_host(v, xf).tobytes() == want
    finally:
        engine.set_tile_order(True)
        engine.set_precheck(True)


def test_concurrent_sorts_on_two_contexts(oracle):
    """Two host threads, each with its own context (= its own stream), sorting at the same time on one GPU: the
    look-back of one sort must not depend on anything the other one holds.  Sizes below 1024 tiles (single ticket
    counter) and, with the per-XCD order switched off as rsort_hip.h asks of concurrent callers, above."""
    import threading
    import radix_sort_amd as rs
    errors = []

    def worker(seed, sizes, tile_order):
        try:
            eng = rs.Engine(0)
            eng.set_precheck(False)
            eng.set_tile_order(tile_order)
            for j, n in enumerate(sizes):
                v = shapes.generate("u32", n, "randomized", seed=seed * 100 + j)
                got = eng.sort_keys_host(v, 0)
                if got.tobytes() != np.sort(v).tobytes():
                    errors.append((seed, n))
            eng.close()
        except Exception as e:            # noqa: BLE001 -- reported through the assertion below
            errors.append((seed, repr(e)))

    for sizes, tile_order in (([3_000_000, 1_500_001, 5_000_000, 700_000] * 2, True), ([20_000_000, 18_000_001], False)):
        threads = [threading.Thread(target=worker, args=(s, sizes, tile_order)) for s in (1, 2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors


@pytest.mark.parametrize("switches", ["RSORT_CHAIN_MIN_TILES=2", "RSORT_CHAIN_MIN_TILES=2 RSORT_FEW_VALUED_BINS=0",
                                      "RSORT_CHAIN_MIN_TILES=2 RSORT_FORCE_WIDE=1"])
def test_chain_mode_at_test_sizes(oracle, switches):
    """The chain-mode sweeps (one chain of tiles per XCD, persistent workgroups, next-pass counting) serve sorts of at
    least 4096 tiles in the product -- 33 M four-byte keys, more than a parity test should move through the CPU oracle.
    RSORT_CHAIN_MIN_TILES lowers that bound (a child process: the switches are read once per process), so the same
    kernels run here on 2 to a few hundred tiles: chains of one tile, empty trailing chains, ragged last tiles, every key
    type that uses them, few-valued keys forced down them (RSORT_FEW_VALUED_BINS=0), sorted inputs whose values repeat
    (run-head ranking with aligned and unaligned runs), 8-byte descriptors, and the 8-byte-key / 4-byte-payload pairs."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import radix_sort_amd as rs
from radix_sort_amd import shapes
from cpu_checkers import load_oracle
orc = load_oracle(); eng = rs.Engine(0); eng.set_precheck(False)
sizes = (2 * 8192, 16385, 8 * 8192 + 5, 300001, (1 << 21) + 77)
for name in ("u32", "i32", "f32", "u64", "i64", "f64"):
    dt, kb, xf = shapes.KEY_TYPES[name]
    for shape in ("randomized", "duplicates", "sorted", "nearly_sorted", "reverse_sorted"):
        for n in sizes:
            v = shapes.generate(name, n, shape)
            assert eng.sort_keys_host(v, xf).tobytes() == orc.sort_keys(v, xf, mt=True).tobytes(), (name, shape, n)
    for rep in (32, 17, 3):                       # sorted, every value `rep` times: runs of equal digits in every pass
        for n in (300001, (1 << 21) + 77):
            base = np.sort(shapes.generate(name, n // rep + 1, "randomized"))
            v = np.repeat(base, rep)[:n].copy()
            if xf == 2:
                v = v[~np.isnan(v)]
                v = np.sort(v)
            got = eng.sort_keys_host(v, xf)
            assert got.tobytes() == orc.sort_keys(v, xf, mt=True).tobytes(), (name, "repeat", rep, n)
for kname in ("u64", "i64"):
    dt, kb, xf = shapes.KEY_TYPES[kname]
    for shape in ("randomized", "duplicates", "nearly_sorted"):
        for n in (16385, 300001, (1 << 20) + 3):
            k = shapes.generate(kname, n, shape)
            gk, gp = eng.sort_pairs_host(k, np.arange(n, dtype=np.uint32))      # 4-byte payload: the pair shape that runs in chain mode
            wk, wp = orc.sort_pairs(k, xf, mt=True)
            assert gk.tobytes() == wk.tobytes() and (gp == wp).all(), (kname, shape, n)
print("CHAIN OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for sw in switches.split():
        name, value = sw.split("=")
        env[name] = value
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CHAIN OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_more_than_2_30_keys_on_the_device():
    """Sorts of more than 2^30 keys: chain mode with 4-byte descriptors (a descriptor counts inside one chain), and -- few-valued
    keys -- the one-chain kernels with 8-byte descriptors.  Too large for the CPU oracle: checked on the device for order,
    key sum and the count of every value of the few-valued input (the size-independent properties of a sort)."""
    import torch
    import radix_sort_amd as rs
    eng = rs.Engine(0)
    eng.set_precheck(False)
    try:
        # (2^30 + 100 few-valued keys: the one-chain kernels' 8-byte descriptor rows, ceil(n / 16384) * 8 words per digit, are
        # more than the chain form's ceil(n / 8192) * 4 -- a multi-GPU rank holding a little over 2^30 keys after its exchange
        # was refused with "descriptor rows of the one-chain fallback" when only the latter were cleared)
        for few, n in ((False, (1 << 30) + 12345), (True, (1 << 30) + 12345), (True, (1 << 30) + 100)):
            g = torch.Generator(device="cuda"); g.manual_seed(3)
            keys = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device="cuda", generator=g)
            if few:
                keys = ((keys >> 25) & 127) * 0x01010101         # 128 values, and 128 digit values in every pass: the few-valued rule
            want_sum = int(keys.sum(dtype=torch.int64).item())
            want_hist = torch.bincount(((keys >> 24) + 128).to(torch.int64), minlength=256) if few else None
            torch.cuda.synchronize()
            eng.sort_keys_(keys, 1)                               # signed order
            eng.synchronize()
            assert int(keys.sum(dtype=torch.int64).item()) == want_sum
            step = 1 << 27
            for lo in range(0, n - 1, step):
                hi = min(lo + step + 1, n)
                assert bool((keys[lo + 1:hi] >= keys[lo:hi - 1]).all().item()), (few, lo)
            if few:
                assert bool((torch.bincount(((keys >> 24) + 128).to(torch.int64), minlength=256) == want_hist).all().item())
            del keys
    finally:
        eng.close()
