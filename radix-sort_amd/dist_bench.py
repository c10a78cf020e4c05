"""bench.py's N > 1 leg: BASELINE config 5 -- uint64 keys, --keys keys per GPU (default 2^30, i.e. 2^33 on
8 GPUs), MSD splitter + RCCL all-to-all over xGMI + local LSD onesweep sort (radix-sort_amd/dist.py).

Launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.
One "step" = one complete distributed sort of the resident shards; the pristine shards are restored
device-to-device outside the timed region.  Scaling is weak (per-GPU keys fixed as N grows)."""
import json
import os
import time


def main(args):
    import torch
    import torch.distributed as dist

    import radix_sort_amd as rs
    from radix_sort_amd import dist as rsd
    from radix_sort_amd import shapes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    # RSORT_DIST_REHEARSAL=1: the same script on ONE GPU -- every rank on cuda:0, gloo instead of RCCL (the exchange is
    # staged through host memory by dist.Comm, the bookkeeping collectives run on CPU tensors).  It rehearses this
    # file's logic where only one GPU is at hand; its timings mean nothing.
    rehearsal = os.environ.get("RSORT_DIST_REHEARSAL") == "1"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    if rehearsal:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        # The first type is the headline (same key type as the N = 1 line, so the driver's scaling ratio compares
        # like with like); the others (uint64 = BASELINE configs[4]) are reported in `rows`.
        rows = [_run_type(args, name, rank, world, dev) for name in args.types]
        if rank == 0:
            head = rows[0]
            line = {
                "metric": "Gkeys/s (uint32/uint64, N=2^30) at 1/2/4/8 MI355X; % HBM roofline",
                "value": head["gkeys_s"], "unit": "Gkeys/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": head["ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": head["type"], "data": "synthetic",
                "config": {"workload": f"{head['type']} {head['shape']} N={args.n} keys per GPU ({args.n * world} total): MSD splitter "
                                       "+ RCCL all-to-all over xGMI + local LSD onesweep (BASELINE configs[4] pattern; "
                                       "the uint64 row of `rows` is configs[4] itself)"},
                "roofline": head.pop("roofline"),
                "rows": rows,
            }
            print(json.dumps(line))
    finally:
        dist.destroy_process_group()


def _run_type(args, name, rank, world, dev):
    import torch
    import torch.distributed as dist

    import radix_sort_amd as rs
    from radix_sort_amd import dist as rsd
    from radix_sort_amd import shapes

    dt, kb, xf = shapes.KEY_TYPES[name]
    n = args.n
    sign = -(1 << 31) if kb == 4 else -(1 << 63)
    eng = rs.Engine(dev.index, stream=torch.cuda.current_stream().cuda_stream)
    cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")      # where the bookkeeping collectives run
    eng.set_profiling(True)
    eng.reserve(n + n // 4, kb, 0)
    local = rsd.HipLocal(eng, kb, xf)
    comm = rsd.Comm()
    shape = args.shapes[0] if args.shapes else "randomized"
    # sorted shapes: every rank holds the same ramp, the worst case for the exchange (everyone sends everywhere)
    src = shapes.generate_device(name, n, shape, dev, seed=shapes.SEED + rank)
    work = torch.empty_like(src)

    stages = rsd.default_stages()                  # RSORT_DIST_STAGES: 1 = one all-to-all then one local sort; S > 1 = pipelined (dist.py)
    torch.cuda.reset_peak_memory_stats(dev)
    times, pass_ms = [], []
    out = None
    for it in range(args.warmup + args.steps):
        work.copy_(src)
        del out
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        out = rsd.distributed_sort(work, local, comm, stages=stages)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        if it >= args.warmup:
            times.append(t1 - t0)
            pass_ms.extend(eng.profile()[1:])
    # one more sort, outside the timed steps, with per-phase clocks (a device synchronisation at every phase end)
    work.copy_(src)
    del out
    torch.cuda.synchronize()
    dist.barrier()
    st = {}
    out = rsd.distributed_sort(work, local, comm, stats=st, stages=stages)
    torch.cuda.synchronize()
    free_b, total_b = torch.cuda.mem_get_info(dev)                          # device-wide: torch's tensors AND the library's own workspace
    mem = torch.tensor([float(torch.cuda.max_memory_allocated(dev)), float(total_b - free_b)], dtype=torch.float64, device=cdev)
    dist.all_reduce(mem, op=dist.ReduceOp.MAX)
    names = ["partition_top_digit", "gather_tables", "refine_second_digit", "equal_key_buckets", "exchange", "local_sort",
             "exchange_and_local_sort_pipelined"]
    ph = torch.tensor([st["phase_ms"].get(k, 0.0) for k in names] + [float(st["bytes_sent"])], dtype=torch.float64, device=cdev)
    ph_max = ph.clone()
    dist.all_reduce(ph_max, op=dist.ReduceOp.MAX)
    dist.all_reduce(ph, op=dist.ReduceOp.SUM)
    phase_ms = {k: round(float(ph_max[i].item()), 3) for i, k in enumerate(names)}
    bytes_sent_total = int(ph[len(names)].item())
    tsum = torch.tensor([sum(times)], dtype=torch.float64, device=cdev)     # max over ranks of the K timed steps
    dist.all_reduce(tsum, op=dist.ReduceOp.MAX)
    sec = float(tsum.item()) / len(times)

    # size-independent parity properties: locally sorted, rank boundaries ordered, count and checksum kept
    okey = (out ^ sign).to(torch.int64)                # unsigned order as signed
    ok_local = bool((okey[1:] >= okey[:-1]).all().item()) if out.numel() > 1 else True
    edge = (torch.stack([okey[0], okey[-1]]) if out.numel() else torch.tensor([2 ** 63 - 1, -(2 ** 63)], device=dev)).to(cdev)
    edges = torch.empty(2 * world, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(edges, edge)
    e = edges.cpu().view(world, 2)
    ok_edges = all(int(e[r][1]) <= int(e[r + 1][0]) for r in range(world - 1))
    stats = torch.stack([torch.tensor(out.numel(), dtype=torch.int64, device=dev), out.sum(),
                         torch.tensor(n, dtype=torch.int64, device=dev), src.sum(),
                         torch.tensor(int(ok_local), dtype=torch.int64, device=dev)]).to(cdev)
    dist.all_reduce(stats)
    sizes = torch.empty(world, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(sizes, torch.tensor([out.numel()], dtype=torch.int64, device=cdev))
    s = stats.cpu().tolist()
    total = n * world
    avg_pass = (sum(pass_ms) / len(pass_ms)) * 1e-3 if pass_ms else None
    achieved = float(sizes[rank].item()) * 2 * kb / avg_pass / 1e9 if avg_pass else None
    row = {"type": name, "shape": shape, "n_per_gpu": n, "ms": sec * 1e3, "gkeys_s": total / sec / 1e9,
           "hbm_frac_per_gpu": n * (kb + 2 * kb * kb) / sec / 1e9 / 8000.0,
           "parity": {"locally_sorted_all_ranks": s[4] == world, "rank_edges_ordered": ok_edges,
                      "count_kept": s[0] == s[2], "checksum_kept": s[1] == s[3]},
           "keys_per_rank_after_exchange": sizes.cpu().tolist(),
           "phase_ms_max_over_ranks": phase_ms,          # from the extra, synchronised sort: where a step's time goes
           "local_sort_form_rank0": {0: "lsd", 1: "hybrid", 2: "narrow hybrid", 3: "prefixed hybrid"}.get(int(eng.lib.rs_debug_last_hybrid(eng.h)), "?"),
           "stages": stages,                              # RSORT_DIST_STAGES (1 = unpipelined)
           "stage_done_ms_rank0": st.get("stage_done_ms"),   # pipelined form: when each stage was exchanged and sorted, from the first post
           "peak_device_bytes_max_over_ranks": {"torch_tensors": int(mem[0].item()), "device_in_use_after_sort": int(mem[1].item())},
           "bytes_sent_all_ranks": bytes_sent_total,
           "exchange_gbs_per_rank": (bytes_sent_total / world / (phase_ms["exchange"] * 1e-3) / 1e9) if phase_ms["exchange"] > 0 else None,   # (unpipelined form only)
           "roofline": {"bound": "hbm", "kernel": f"onesweep_kernel<{name}> (local sort, rank 0)", "achieved": achieved,
                        "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0 if achieved else None, "traffic": None}}
    del out, work, src
    eng.close()
    torch.cuda.empty_cache()
    return row
