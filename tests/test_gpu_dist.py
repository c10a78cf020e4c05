"""GPU (one box, one card): two ranks share GPU 0, local compute through the C-ABI (HipLocal), exchange
over gloo staged through host memory -- exercises everything of the multi-GPU path except RCCL itself."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from radix_sort_amd import shapes

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, name, shape, n, outdir, stages=1):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import radix_sort_amd as rs
    from radix_sort_amd import dist as rsd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dt, kb, xf = shapes.KEY_TYPES[name]
        full = shapes.generate(name, n, shape)
        lo, hi = rank * n // world, (rank + 1) * n // world
        t = torch.from_numpy(full[lo:hi].view({4: np.int32, 8: np.int64}[kb]).copy()).cuda()
        eng = rs.Engine(0)
        out = rsd.distributed_sort(t, rsd.HipLocal(eng, kb, xf), rsd.Comm(), stages=stages)
        np.save(os.path.join(outdir, f"out{rank}.npy"), out.cpu().numpy())
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("stages", [1, 4])
@pytest.mark.parametrize("name,shape", [("u64", "randomized"), ("u64", "duplicates"), ("f64", "randomized"), ("u32", "sorted")])
def test_two_ranks_one_gpu(tmp_path, oracle, name, shape, stages):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n, world = 600_001, 2
    mp.spawn(_worker, args=(world, port, name, shape, n, str(tmp_path), stages), nprocs=world, join=True)
    dt, kb, xf = shapes.KEY_TYPES[name]
    got = np.concatenate([np.load(tmp_path / f"out{r}.npy") for r in range(world)])
    want = oracle.sort_keys(shapes.generate(name, n, shape), xf, mt=True)
    assert got.tobytes() == want.tobytes()


def _pair_worker(rank, world, port, name, shape, n, outdir, stages=1):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import radix_sort_amd as rs
    from radix_sort_amd import dist as rsd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dt, kb, xf = shapes.KEY_TYPES[name]
        full = shapes.generate(name, n, shape)
        lo, hi = rank * n // world, (rank + 1) * n // world
        t = torch.from_numpy(full[lo:hi].view({4: np.int32, 8: np.int64}[kb]).copy()).cuda()
        idx = torch.arange(lo, hi, dtype=torch.int64 if kb == 8 else torch.int32).cuda()
        eng = rs.Engine(0)
        k, v = rsd.distributed_sort(t, rsd.HipLocal(eng, kb, xf), rsd.Comm(), values=idx, stages=stages)
        np.save(os.path.join(outdir, f"k{rank}.npy"), k.cpu().numpy())
        np.save(os.path.join(outdir, f"v{rank}.npy"), v.cpu().numpy())
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("stages", [1, 4])
@pytest.mark.parametrize("name,shape", [("u64", "duplicates"), ("i32", "duplicates"), ("f64", "randomized")])
def test_two_ranks_one_gpu_stable_pairs(tmp_path, oracle, name, shape, stages):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n, world = 400_001, 2
    mp.spawn(_pair_worker, args=(world, port, name, shape, n, str(tmp_path), stages), nprocs=world, join=True)
    dt, kb, xf = shapes.KEY_TYPES[name]
    gk = np.concatenate([np.load(tmp_path / f"k{r}.npy") for r in range(world)])
    gv = np.concatenate([np.load(tmp_path / f"v{r}.npy") for r in range(world)])
    wk, wp = oracle.sort_pairs(shapes.generate(name, n, shape), xf, mt=True)
    assert gk.tobytes() == wk.tobytes()
    assert np.array_equal(gv.astype(np.uint64), wp.astype(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("stages", [1, 4])
def test_multi_gpu_bench_script_rehearsal(stages):
    """bench.py --gpus 2 end to end on ONE GPU (RSORT_DIST_REHEARSAL=1: both ranks on cuda:0, gloo in place of RCCL): the
    script the driver runs for the scaling table, its JSON line and its parity properties.  Timings are meaningless here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSORT_DIST_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", RSORT_DIST_STAGES=str(stages))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29641 + stages), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--keys", str(1 << 21)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["unit"] == "Gkeys/s" and line["scaling"] == "weak" and line["value"] > 0
    assert [row["type"] for row in line["rows"]] == ["u32", "u64"]
    for row in line["rows"]:
        assert all(row["parity"].values()), row
        assert sum(row["keys_per_rank_after_exchange"]) == 2 * (1 << 21)
        ph = row["phase_ms_max_over_ranks"]                    # the per-phase clocks of the extra, synchronised sort
        assert ph["partition_top_digit"] > 0 and row["stages"] == stages
        if stages == 1:
            assert ph["exchange"] > 0 and ph["local_sort"] > 0
        else:
            assert ph["exchange_and_local_sort_pipelined"] > 0 and len(row["stage_done_ms_rank0"]) == stages
        assert row["bytes_sent_all_ranks"] > 0 and row["peak_device_bytes_max_over_ranks"]["device_in_use_after_sort"] > 0
