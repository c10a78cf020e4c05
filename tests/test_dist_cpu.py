"""CPU: the multi-GPU splitter (radix-sort_amd/dist.py) -- planning logic as unit tests, and the whole
histogram -> split -> all-to-all -> local sort path on 2 and 3 gloo ranks with a numpy local backend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from radix_sort_amd import dist as rsd
from radix_sort_amd import shapes


def test_plan_is_balanced_and_consistent():
    rng = np.random.Generator(np.random.PCG64(1))
    world = 4
    keys = [rng.integers(0, 1 << 32, 50_000, dtype=np.uint32) for _ in range(world)]
    l1 = np.stack([np.bincount(k >> 24, minlength=256) for k in keys]).astype(np.int64)
    refine = rsd.boundary_buckets(l1, world)
    l2 = {b: np.stack([np.bincount((k[(k >> 24) == b] >> 16) & 255, minlength=256) for k in keys]).astype(np.int64) for b in refine}
    split = rsd.plan_splits(l1, l2, world)
    assert split[0] == 0 and split[-1] == 65536 and np.all(np.diff(split) >= 0)
    counts = rsd.exchange_counts(l1, l2, split, world)
    allk = np.concatenate(keys)
    for r in range(world):
        want = int(((allk >> 16) >= split[r]).sum() - ((allk >> 16) >= split[r + 1]).sum())
        assert counts[:, r].sum() == want
        # 16-bit resolution, and a boundary may sit on a top-digit edge up to snap_tolerance keys from its quantile (each rank has two)
        assert abs(want - allk.size / world) <= allk.size / 65536 * 4 + 64 + 2 * rsd.snap_tolerance(allk.size, world)
    assert counts.sum() == allk.size and np.array_equal(counts.sum(axis=1), [k.size for k in keys])


def test_boundaries_snap_to_top_digit_edges_when_that_is_free():
    """Uniform keys over 8 ranks: every quantile lies within a few keys of a top-digit edge, so no bucket is refined, the
    boundaries are multiples of 256 * 32 and every rank's keys share their three leading bits (the local sort's prefixed
    hybrid form lives on that); a distribution whose quantiles fall inside buckets is refined as before."""
    rng = np.random.Generator(np.random.PCG64(5))
    world = 8
    keys = [rng.integers(0, 1 << 32, 400_000, dtype=np.uint32) for _ in range(world)]
    l1 = np.stack([np.bincount(k >> 24, minlength=256) for k in keys]).astype(np.int64)
    assert rsd.boundary_buckets(l1, world) == []
    split = rsd.plan_splits(l1, {}, world)
    assert list(split) == [r * 8192 for r in range(world + 1)]
    counts = rsd.exchange_counts(l1, {}, split, world)
    share = counts.sum(axis=0)
    assert share.max() / share.mean() <= 1.01
    skew = [k // np.uint32(3) + np.uint32(1 << 29) for k in keys]              # a third of the range: quantiles at 32 + 10.67 r, away from the edges
    l1s = np.stack([np.bincount(k >> 24, minlength=256) for k in skew]).astype(np.int64)
    assert len(rsd.boundary_buckets(l1s, world)) >= 4


def test_plan_degenerate_inputs():
    world = 3
    l1 = np.zeros((world, 256), dtype=np.int64)
    assert list(rsd.plan_splits(l1, {}, world)) == [0, 0, 0, 65536] or rsd.plan_splits(l1, {}, world)[-1] == 65536
    l1[:, 7] = 1000                                   # every key in one top-digit bucket
    refine = rsd.boundary_buckets(l1, world)
    assert refine == [7]
    l2 = {7: np.zeros((world, 256), dtype=np.int64)}
    l2[7][:, 3] = 1000                                # ... and one second digit: cannot be split further
    split = rsd.plan_splits(l1, l2, world)
    counts = rsd.exchange_counts(l1, l2, split, world)
    assert counts.sum() == 3000 and np.all(np.diff(split) >= 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, shape, n_per_rank, outdir, stages=1):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import radix_sort_amd  # noqa: F401
    from dist_cpu_backend import NumpyLocal
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dt, kb, xf = shapes.KEY_TYPES[name]
        full = shapes.generate(name, n_per_rank * world + 17, shape)
        bounds = np.linspace(0, full.size, world + 1).astype(int)
        bounds[1] += 5 if world > 1 else 0               # ragged shards
        mine = full[bounds[rank]:bounds[rank + 1]].copy()
        t = torch.from_numpy(mine.view({1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[kb]))
        out = rsd.distributed_sort(t, NumpyLocal(kb, xf), rsd.Comm(), stages=stages)
        np.save(os.path.join(outdir, f"out{rank}.npy"), out.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name,shape", [("u64", "randomized"), ("u64", "duplicates"), ("u32", "sorted"),
                                        ("i64", "nearly_sorted"), ("f64", "randomized"), ("u8", "randomized"),
                                        ("u32", "reverse_sorted")])
@pytest.mark.parametrize("stages", [1, 4])
def test_distributed_sort_gloo(tmp_path, oracle, world, name, shape, stages):
    """stages = 4: the pipelined exchange -- every destination's range in four sub-ranges, each exchanged and sorted on
    its own (RSORT_DIST_STAGES)."""
    n_per_rank = 20_000
    mp.spawn(_worker, args=(world, _free_port(), name, shape, n_per_rank, str(tmp_path), stages), nprocs=world, join=True)
    dt, kb, xf = shapes.KEY_TYPES[name]
    full = shapes.generate(name, n_per_rank * world + 17, shape)
    got = np.concatenate([np.load(tmp_path / f"out{r}.npy") for r in range(world)])
    want = oracle.sort_keys(full, xf, mt=True)
    assert got.tobytes() == want.tobytes()
    sizes = [np.load(tmp_path / f"out{r}.npy").size for r in range(world)]
    if shape == "randomized" and kb >= 4:
        assert max(sizes) - min(sizes) <= full.size // 100 + 64


def _pair_worker(rank, world, port, name, shape, n_per_rank, outdir, stages=1):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import radix_sort_amd  # noqa: F401
    from dist_cpu_backend import NumpyLocal
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dt, kb, xf = shapes.KEY_TYPES[name]
        full = shapes.generate(name, n_per_rank * world + 17, shape)
        bounds = np.linspace(0, full.size, world + 1).astype(int)
        bounds[1] += 5
        mine = full[bounds[rank]:bounds[rank + 1]].copy()
        t = torch.from_numpy(mine.view({1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[kb]))
        idx = torch.arange(bounds[rank], bounds[rank + 1], dtype=torch.int64)       # global positions as payload
        k, v = rsd.distributed_sort(t, NumpyLocal(kb, xf), rsd.Comm(), values=idx, stages=stages)
        np.save(os.path.join(outdir, f"k{rank}.npy"), k.numpy())
        np.save(os.path.join(outdir, f"v{rank}.npy"), v.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name,shape", [("u64", "duplicates"), ("i32", "duplicates"), ("f64", "randomized"), ("u16", "randomized")])
@pytest.mark.parametrize("stages", [1, 4])
def test_distributed_pairs_are_stable_gloo(tmp_path, oracle, world, name, shape, stages):
    """Keys + payload across ranks: the payload (global input positions) must come back as the reference's stable
    permutation (key path, radix_sort.hpp:1160-1206 / sortByIndices :1111-1154)."""
    n_per_rank = 15_000
    mp.spawn(_pair_worker, args=(world, _free_port(), name, shape, n_per_rank, str(tmp_path), stages), nprocs=world, join=True)
    dt, kb, xf = shapes.KEY_TYPES[name]
    full = shapes.generate(name, n_per_rank * world + 17, shape)
    gk = np.concatenate([np.load(tmp_path / f"k{r}.npy") for r in range(world)])
    gv = np.concatenate([np.load(tmp_path / f"v{r}.npy") for r in range(world)])
    wk, wp = oracle.sort_pairs(full, xf, mt=True)
    assert gk.tobytes() == wk.tobytes()
    assert np.array_equal(gv.astype(np.uint64), wp.astype(np.uint64))


@pytest.mark.parametrize("stages", [1, 4])
@pytest.mark.parametrize("shape", ["randomized", "sorted", "reverse_sorted", "nearly_sorted", "duplicates"])
def test_world8_all_shapes_sorted_and_balanced(tmp_path, oracle, shape, stages):
    """BASELINE configs[4] rehearsed on CPU: 8 ranks, uint64 keys, every generator shape -- the exact global order, and
    after the exchange no rank holds much more than its share (the 256-value duplicates shape included: buckets of equal
    keys are cut by count)."""
    world, n_per_rank = 8, 12_000
    mp.spawn(_worker, args=(world, _free_port(), "u64", shape, n_per_rank, str(tmp_path), stages), nprocs=world, join=True)
    full = shapes.generate("u64", n_per_rank * world + 17, shape)
    outs = [np.load(tmp_path / f"out{r}.npy") for r in range(world)]
    assert np.concatenate(outs).tobytes() == oracle.sort_keys(full, 0, mt=True).tobytes()
    sizes = np.array([o.size for o in outs], dtype=np.float64)
    assert sizes.max() / sizes.mean() <= (1.3 if shape == "duplicates" else 1.05), sizes


@pytest.mark.parametrize("stages", [1, 4])
def test_world8_stable_pairs_on_duplicates(tmp_path, oracle, stages):
    """Equal-key buckets cut by count must keep the global input order of the equal keys (payload = global position)."""
    world, n_per_rank = 8, 6_000
    mp.spawn(_pair_worker, args=(world, _free_port(), "u32", "duplicates", n_per_rank, str(tmp_path), stages), nprocs=world, join=True)
    full = shapes.generate("u32", n_per_rank * world + 17, "duplicates")
    gk = np.concatenate([np.load(tmp_path / f"k{r}.npy") for r in range(world)])
    gv = np.concatenate([np.load(tmp_path / f"v{r}.npy") for r in range(world)])
    wk, wp = oracle.sort_pairs(full, 0, mt=True)
    assert gk.tobytes() == wk.tobytes()
    assert np.array_equal(gv.astype(np.uint64), wp.astype(np.uint64))
    sizes = np.array([np.load(tmp_path / f"k{r}.npy").size for r in range(world)], dtype=np.float64)
    assert sizes.max() / sizes.mean() <= 1.3, sizes


def test_stage_pieces_tile_every_destination():
    """Planning only: the staged exchange's pieces are, for every source and destination, a partition of exactly the keys
    the unstaged exchange sends, in order; a destination spanning fewer top-digit values than stages gets empty pieces."""
    rng = np.random.Generator(np.random.PCG64(5))
    for world, stages in ((2, 4), (3, 4), (8, 4), (8, 64)):
        keys = [rng.integers(0, 1 << 32, 30_000, dtype=np.uint64).astype(np.uint32) for _ in range(world)]
        l1 = np.stack([np.bincount((k >> 24).astype(np.int64), minlength=256) for k in keys])
        refine = rsd.boundary_buckets(l1, world)
        l2 = {b: np.stack([np.bincount(((k[(k >> 24) == b] >> 16) & 255).astype(np.int64), minlength=256) for k in keys]) for b in refine}
        split = rsd.plan_splits(l1, l2, world)
        cuts = rsd.stage_cuts(l1, split, world, stages)
        assert (np.diff(cuts, axis=1) >= 0).all() and (cuts[:, 0] == split[:-1]).all() and (cuts[:, -1] == split[1:]).all()
        seg, off = rsd.stage_counts(l1, l2, split, cuts, world)
        dest = rsd.exchange_counts(l1, l2, split, world)
        assert (seg.sum(axis=2) == dest).all() and (seg >= 0).all()
        for i in range(world):
            assert (np.diff(off[i].reshape(-1)) == seg[i].reshape(-1)[:-1]).all() and off[i, 0, 0] == 0
            order = np.sort(keys[i] >> 16)                     # the source's keys by 16-bit prefix, as its partition passes leave them
            for r in range(world):
                for s in range(stages):
                    piece = order[off[i, r, s]: off[i, r, s] + seg[i, r, s]]
                    assert ((piece >= cuts[r, s]) & (piece < max(cuts[r, s + 1], cuts[r, s] + (piece.size > 0)))).all() or \
                           ((piece >= cuts[r, s]) & (piece <= cuts[r, s + 1])).all()


def test_equal_key_bucket_is_cut_by_count():
    """Planning only: every key equal (one 16-bit bucket holds everything) -- the inner cuts give each rank its share, taken
    from the sources in rank order."""
    world = 4
    l1 = np.zeros((world, 256), dtype=np.int64)
    l1[:, 9] = [1000, 10, 500, 490]
    l2 = {9: np.zeros((world, 256), dtype=np.int64)}
    l2[9][:, 200] = l1[:, 9]
    x16 = (9 << 8) + 200
    split = np.array([0, x16, x16, x16, 65536], dtype=np.int64)
    total = int(l1.sum())
    cuts = {r: (x16, r * total // world) for r in range(1, world)}
    counts = rsd.exchange_counts(l1, l2, split, world, cuts)
    assert counts.sum() == total and np.array_equal(counts.sum(axis=1), l1[:, 9])
    assert list(counts.sum(axis=0)) == [500, 500, 500, 500]
    assert list(counts[0]) == [500, 500, 0, 0] and list(counts[1]) == [0, 0, 10, 0]      # sources in rank order
