"""Multi-GPU sort: MSD splitter + all-to-all + local LSD sort (one process per GPU).

This stands where the reference's CPU thread partitioner stands (radix_sort.hpp:112-123 thread
count, :165-190 per-thread counts + merge, :210-254 partitionRegions fanning MSD buckets out to
worker threads).  The GPU-node analogue, BASELINE config 5:

  1. every rank counts the TOP digit of its (transformed) keys          rs_digit_histogram_device
  2. the per-rank tables are all-gathered (256 x u64 per rank)          RCCL all_gather
  3. every rank derives the same G-1 split points at 16-bit resolution: the top-digit buckets that
     contain a quantile are refined with a second-digit histogram of just that bucket (one more
     tiny all_gather per such bucket)
  4. local MSD partition: one onesweep pass on the top digit (+ one on the second digit inside the
     few refined buckets), which makes every destination's keys contiguous   rs_sort_keys_device
  5. keys change ranks                                                   RCCL all_to_all_single (xGMI)
  6. each rank LSD-sorts what it received                                rs_sort_keys_device
The result is the concatenation of the ranks' slices in rank order.

Everything that touches keys goes through a `local` backend object; the product backend is
HipLocal (C-ABI on the rank's GPU).  tests/ plug in a numpy backend to exercise the splitter logic
and the exchange on CPU with gloo -- the product path never does.

Balance: split points fall on 16-bit prefixes, so a rank can exceed N/G by at most the largest
16-bit bucket -- unless that bucket holds ONE key value on every rank (few-valued inputs: the
reference's "duplicates" shape puts N/256 equal keys into a bucket).  Such a bucket is cut by count
at the exact quantile: each rank checks its part (all remaining digits constant), one more small
all-gather compares the values, and the cut takes the bucket's keys in (source rank, position)
order, so equal keys keep their global input order and the payload variant stays stable.

Overlap (stages > 1, or RSORT_DIST_STAGES=S): every destination's key range is cut into S sub-ranges at top-digit edges
(each source's keys of one destination and one sub-range are contiguous after the local MSD partition), the exchange
becomes S smaller all-to-alls posted back to back on the communication stream, and sub-range s is sorted locally
-- an independent sort: the sub-ranges are disjoint and ordered -- while sub-range s + 1 is still in flight.  The result is
the concatenation of the sorted sub-ranges.  Keys and payload of a stage travel in one grouped exchange.

`stats` (a dict) makes distributed_sort record per-phase wall times (with a device synchronisation
at each phase end, so only for diagnosis) and the bytes this rank sent; dist_bench.py runs one such
sort after the timed ones.
"""
from __future__ import annotations

import numpy as np

DIGIT = 8


# ---------------------------------------------------------------------------------------------
# split planning (pure numpy on small tables; identical on every rank)
# ---------------------------------------------------------------------------------------------
def snap_tolerance(total: int, world: int) -> int:
    """Keys a boundary may miss its quantile by for the sake of a top-digit edge: a quarter of a per cent of a rank's share.
    A boundary on a top-digit edge costs no second-digit pass over its bucket, and ranks whose ranges begin and end on
    such edges hold keys that share their leading bits (G = 8, uniform keys: three of them), which is what the local
    sort's prefixed hybrid form needs (rsort_api.hip)."""
    return total // world // 400


def boundary_buckets(level1_all: np.ndarray, world: int):
    """Top-digit buckets that contain an interior quantile r*N/G inside them, further than snap_tolerance from both edges
    (plan_splits puts a boundary of an unrefined bucket on the nearer edge).

    level1_all: [G, 256] per-rank counts of the top digit."""
    g = level1_all.sum(axis=0).astype(np.int64)
    total = int(g.sum())
    cum = np.cumsum(g)
    tol = snap_tolerance(total, world)
    out = []
    for r in range(1, world):
        q = (r * total) // world
        b = int(np.searchsorted(cum, q, side="right"))
        if b < 256 and (cum[b] - g[b]) + tol < q < cum[b] - tol:      # inside bucket b, away from its edges
            out.append(b)
    return sorted(set(out))


def plan_splits(level1_all: np.ndarray, level2_all: dict, world: int) -> np.ndarray:
    """split16[r] for r = 0..G: rank r receives the keys whose 16-bit prefix x satisfies
    split16[r] <= x < split16[r+1]  (split16[0] = 0, split16[G] = 65536, non-decreasing).

    level2_all: {bucket: [G, 256] counts of the second digit inside that top-digit bucket}."""
    g1 = level1_all.sum(axis=0).astype(np.int64)
    total = int(g1.sum())
    cum1 = np.cumsum(g1)
    split = np.zeros(world + 1, dtype=np.int64)
    split[world] = 65536
    for r in range(1, world):
        q = (r * total) // world
        b = int(np.searchsorted(cum1, q, side="right"))
        if b >= 256:
            split[r] = 65536
            continue
        base = int(cum1[b] - g1[b])
        if q == base:                                   # exactly on the lower edge of bucket b
            split[r] = b << 8
            continue
        if b not in level2_all:                         # bucket not refined (1-byte keys): nearer edge
            split[r] = (b << 8) if (q - base) <= (cum1[b] - q) else ((b + 1) << 8)
            continue
        g2 = level2_all[b].sum(axis=0).astype(np.int64)
        cum2 = base + np.cumsum(g2)
        s = int(np.searchsorted(cum2, q, side="right"))
        s = min(s, 255)
        lo_edge = int(cum2[s] - g2[s])
        # cut before or after second digit s, whichever is closer to the quantile
        split[r] = (b << 8) + (s if (q - lo_edge) <= (cum2[s] - q) else s + 1)
    # monotone by construction of quantiles, but equal quantiles / clamps can tie: enforce
    for r in range(1, world + 1):
        split[r] = max(split[r], split[r - 1])
    return split


def count_below(level1: np.ndarray, level2: dict, x16: int) -> int:
    """Keys of ONE rank with 16-bit prefix < x16 (x16 lies on a bucket edge or inside a refined bucket)."""
    b, s = x16 >> 8, x16 & 255
    n = int(level1[:b].sum())
    if s:
        n += int(level2[b][:s].sum())
    return n


def exchange_counts(level1_all, level2_all, split16, world, inner_cuts=None):
    """counts[i][r] = keys rank i sends to rank r.

    inner_cuts: {r: (x16, extra)} -- boundary r lies INSIDE the 16-bit bucket x16, whose keys are all equal: the first
    `extra` of them, counted over the sources in rank order, go below the boundary (split16[r] == x16 for those r)."""
    inner_cuts = inner_cuts or {}
    below = np.zeros((world, world + 1), dtype=np.int64)
    for i in range(world):
        l2 = {b: t[i] for b, t in level2_all.items()}
        below[i] = [count_below(level1_all[i], l2, int(split16[r])) for r in range(world + 1)]
    for r, (x16, extra) in inner_cuts.items():
        per_source = bucket_counts(level1_all, level2_all, x16)            # [G] keys of the bucket on each source
        before = np.concatenate(([0], np.cumsum(per_source)[:-1]))          # ... on the sources in front of it
        below[:, r] += np.clip(extra - before, 0, per_source)
    below = np.maximum.accumulate(below, axis=1)
    return np.diff(below, axis=1)


def stage_cuts(level1_all: np.ndarray, split16: np.ndarray, world: int, stages: int) -> np.ndarray:
    """cuts[r][0..S]: destination r's prefix range [split16[r], split16[r+1]) cut into S sub-ranges, interior cuts on
    top-digit edges (multiples of 256) nearest to the equal-count points of the destination's keys; non-decreasing, so a
    destination whose range spans fewer than S top-digit values gets empty sub-ranges."""
    g1 = level1_all.sum(axis=0).astype(np.int64)
    cum_edge = np.concatenate(([0], np.cumsum(g1)))                        # keys below top-digit edge e (e = 0..256)
    cuts = np.zeros((world, stages + 1), dtype=np.int64)
    for r in range(world):
        lo, hi = int(split16[r]), int(split16[r + 1])
        cuts[r, 0], cuts[r, stages] = lo, hi
        e_lo, e_hi = (lo + 255) >> 8, hi >> 8                              # top-digit edges inside [lo, hi]
        for s in range(1, stages):
            if e_lo > e_hi:
                cuts[r, s] = lo
                continue
            below_lo, below_hi = cum_edge[e_lo], cum_edge[e_hi]
            want = below_lo + (below_hi - below_lo) * s // stages
            e = int(np.clip(np.searchsorted(cum_edge, want, side="left"), e_lo, e_hi))
            if e > e_lo and abs(cum_edge[e - 1] - want) <= abs(cum_edge[e] - want):
                e -= 1
            cuts[r, s] = min(max(e << 8, lo), hi)
        cuts[r] = np.maximum.accumulate(cuts[r])
    return cuts


def stage_counts(level1_all, level2_all, split16, cuts, world, inner_cuts=None):
    """seg[i][r][s] = keys source i sends to destination r in stage s (their sum over s is exchange_counts[i][r]), and
    off[i][r][s] = where that piece starts in source i's partitioned array."""
    stages = cuts.shape[1] - 1
    dest = exchange_counts(level1_all, level2_all, split16, world, inner_cuts)     # [G, G]
    dest_off = np.concatenate((np.zeros((world, 1), dtype=np.int64), np.cumsum(dest, axis=1)), axis=1)   # [G, G + 1]
    seg = np.zeros((world, world, stages), dtype=np.int64)
    off = np.zeros((world, world, stages), dtype=np.int64)
    for i in range(world):
        l2 = {b: t[i] for b, t in level2_all.items()}
        for r in range(world):
            pts = [int(dest_off[i, r])]
            for s in range(1, stages):
                x = int(cuts[r, s])
                below = count_below(level1_all[i], l2, x) if (x & 255) == 0 or (x >> 8) in l2 else int(dest_off[i, r])
                pts.append(int(np.clip(below, dest_off[i, r], dest_off[i, r + 1])))
            pts.append(int(dest_off[i, r + 1]))
            pts = np.maximum.accumulate(np.array(pts, dtype=np.int64))
            seg[i, r] = np.diff(pts)
            off[i, r] = pts[:-1]
    return seg, off


def bucket_counts(level1_all, level2_all, x16):
    """[G] keys with 16-bit prefix x16 on every rank (x16 on a top-digit edge of an unrefined bucket = that whole bucket)."""
    b, s2 = x16 >> 8, x16 & 255
    if b in level2_all:
        return level2_all[b][:, s2].astype(np.int64)
    return level1_all[:, b].astype(np.int64)


def quantile_bucket(level1_all, level2_all, q, two_levels):
    """(x16, keys in front of the bucket, keys in it) for the 16-bit bucket that holds global position q, or None when
    the tables do not resolve it (its top-digit bucket was not refined although the keys have a second digit)."""
    g1 = level1_all.sum(axis=0).astype(np.int64)
    cum1 = np.cumsum(g1)
    b = int(np.searchsorted(cum1, q, side="right"))
    if b >= 256:
        return None
    base = int(cum1[b] - g1[b])
    if not two_levels:
        return (b << 8), base, int(g1[b])
    if b not in level2_all:
        return None
    g2 = level2_all[b].sum(axis=0).astype(np.int64)
    cum2 = base + np.cumsum(g2)
    s2 = min(int(np.searchsorted(cum2, q, side="right")), 255)
    return (b << 8) + s2, int(cum2[s2] - g2[s2]), int(g2[s2])


# ---------------------------------------------------------------------------------------------
# communication (torch.distributed; "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests)
# ---------------------------------------------------------------------------------------------
class Comm:
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def all_gather_table(self, table: np.ndarray, device) -> np.ndarray:
        """[G, *table.shape] from one small int64 table per rank."""
        import torch
        dev = device if self.backend == "nccl" else "cpu"
        t = torch.from_numpy(np.ascontiguousarray(table, dtype=np.int64).reshape(-1)).to(dev)
        out = torch.empty(self.world * t.numel(), dtype=torch.int64, device=dev)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().numpy().reshape((self.world,) + tuple(np.shape(table)))

    def all_to_all(self, send, send_counts, recv_counts, recv):
        """Variable all-to-all of a 1-D tensor; with gloo the payload is staged through host memory.
        Element types the backends do not take (16-bit integers) travel as bytes."""
        import torch
        ins, outs = [int(x) for x in send_counts], [int(x) for x in recv_counts]
        sb, rb = send, recv
        if send.dtype not in (torch.int8, torch.uint8, torch.int32, torch.int64):
            es = send.element_size()
            ins, outs = [x * es for x in ins], [x * es for x in outs]
            sb, rb = send.view(torch.uint8), recv.view(torch.uint8)
        if self.backend == "nccl" or not send.is_cuda:
            self.dist.all_to_all_single(rb, sb, output_split_sizes=outs, input_split_sizes=ins, group=self.group)
        else:
            r = rb.cpu()
            self.dist.all_to_all_single(r, sb.cpu(), output_split_sizes=outs, input_split_sizes=ins, group=self.group)
            rb.copy_(r)
        return recv


    def exchange_pieces(self, send_lists, recv_lists):
        """One stage of the staged exchange: for every array (keys, and the payload with them) a list of G send pieces and
        G receive pieces (views; piece r goes to / comes from rank r).  Returns an object whose wait() makes the pieces
        usable.  RCCL: all arrays of the stage in flight together on the communication stream, the call does not wait;
        gloo (CPU tests, one-GPU rehearsal): staged through host memory, done when the call returns."""
        import torch

        def as_bytes(t):
            return t if t.dtype in (torch.int8, torch.uint8, torch.int32, torch.int64) else t.view(torch.uint8)

        if self.backend == "nccl":
            works = [self.dist.all_to_all([as_bytes(t) for t in recv], [as_bytes(t) for t in send], group=self.group, async_op=True)
                     for send, recv in zip(send_lists, recv_lists)]

            class _Pending:
                def wait(self_inner):
                    for w in works:
                        w.wait()
            return _Pending()
        for send, recv in zip(send_lists, recv_lists):
            sb = torch.cat([as_bytes(t).reshape(-1).cpu() for t in send]) if send else torch.empty(0)
            ins = [as_bytes(t).numel() for t in send]
            outs = [as_bytes(t).numel() for t in recv]
            rb = torch.empty(sum(outs), dtype=sb.dtype)
            self.dist.all_to_all_single(rb, sb, output_split_sizes=outs, input_split_sizes=ins, group=self.group)
            at = 0
            for t, m in zip(recv, outs):
                as_bytes(t).reshape(-1).copy_(rb[at:at + m])
                at += m

        class _Done:
            def wait(self_inner):
                pass
        return _Done()


# ---------------------------------------------------------------------------------------------
# local compute through the C-ABI (product backend)
# ---------------------------------------------------------------------------------------------
class HipLocal:
    """Histogram / MSD partition / LSD sort of one rank's keys on its GPU via librsort_hip.so."""

    def __init__(self, engine, key_bytes: int, transform: int):
        self.eng = engine
        self.kb = key_bytes
        self.xf = transform
        self.top = key_bytes * 8

    def empty_like(self, t, n):
        import torch
        return torch.empty(n, dtype=t.dtype, device=t.device)

    def histogram(self, t, lo, hi, begin_bit, end_bit) -> np.ndarray:
        h = self.eng.digit_histogram_ptr(t.data_ptr() + lo * self.kb, hi - lo, self.kb, self.xf, begin_bit, end_bit)
        return h[0].astype(np.int64)

    def partition(self, t, lo, hi, begin_bit, end_bit, scratch, vals=None, vscratch=None):
        """Stable in-place partition of t[lo:hi] (and of vals[lo:hi] with it) by the digit at [begin_bit, end_bit)
        of the transformed key."""
        if hi - lo <= 1:
            return
        if vals is None:
            self.eng.sort_keys_ptr(t.data_ptr() + lo * self.kb, hi - lo, self.kb, self.xf,
                                   scratch.data_ptr() if scratch is not None else None, begin_bit, end_bit)
        else:
            vb = vals.element_size()
            self.eng.sort_pairs_ptr(t.data_ptr() + lo * self.kb, vals.data_ptr() + lo * vb, hi - lo, self.kb, vb, self.xf,
                                    scratch.data_ptr() if scratch is not None else None,
                                    vscratch.data_ptr() if vscratch is not None else None, begin_bit, end_bit)

    def partition_out(self, t, out, begin_bit, end_bit, vals=None, vout=None):
        """out = t stably reordered by the digit at [begin_bit, end_bit) (vout = vals likewise); returns the digit counts."""
        n = t.numel()
        if vals is None:
            h = self.eng.partition_ptr(t.data_ptr(), out.data_ptr(), n, self.kb, self.xf, begin_bit, end_bit)
        else:
            h = self.eng.partition_ptr(t.data_ptr(), out.data_ptr(), n, self.kb, self.xf, begin_bit, end_bit,
                                       vals.data_ptr(), vout.data_ptr(), vals.element_size())
        return h.astype(np.int64)

    def sort(self, t, scratch=None, vals=None, vscratch=None):
        n = t.numel()
        if n <= 1:
            return
        ks = scratch.data_ptr() if scratch is not None and scratch.numel() >= n else None
        if vals is None:
            self.eng.sort_keys_ptr(t.data_ptr(), n, self.kb, self.xf, ks)
        else:
            vs = vscratch.data_ptr() if vscratch is not None and vscratch.numel() >= n else None
            self.eng.sort_pairs_ptr(t.data_ptr(), vals.data_ptr(), n, self.kb, vals.element_size(), self.xf, ks, vs)

    def single_value(self, t, lo, hi, below_bit):
        """(1, v) when every key of t[lo:hi] has the same transformed bits below `below_bit` (v = those bits), (0, 0) for an
        empty range, (2, 0) otherwise: one histogram of the remaining digits, each with exactly one bin in use."""
        if hi <= lo:
            return 0, 0
        if below_bit <= 0:
            return 1, 0
        h = self.eng.digit_histogram_ptr(t.data_ptr() + lo * self.kb, hi - lo, self.kb, self.xf, 0, below_bit)
        v = 0
        for p in range(h.shape[0]):
            nz = np.flatnonzero(h[p])
            if nz.size != 1:
                return 2, 0
            v |= int(nz[0]) << (8 * p)
        return 1, v

    def synchronize(self):
        self.eng.synchronize()

    def wait_for_producers(self, t):
        """The engine sorts on ITS stream (a context given stream handle 0 -- torch's default stream -- uses its own, see
        rs_ctx_set_stream); whatever produced `t` on torch's streams (a copy, an RCCL collective) has to be finished
        before the engine reads it: sorting keys that are still being written breaks the histogram the scatter relies on."""
        import torch
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()


# ---------------------------------------------------------------------------------------------
# the distributed sort
# ---------------------------------------------------------------------------------------------
def default_stages() -> int:
    import os
    try:
        return max(1, int(os.environ.get("RSORT_DIST_STAGES", "1")))
    except ValueError:
        return 1


def distributed_sort(keys, local, comm: Comm, values=None, stats=None, stages=None):
    """keys: this rank's 1-D tensor of raw key bits (int32 / int64 ...).  Returns this rank's slice of
    the globally sorted sequence (a new tensor); `keys` (and `values`) are used as scratch and are clobbered.

    values: optional 1-D payload tensor (4- or 8-byte elements) that travels with the keys; the result is then
    (keys, values) and the sort is STABLE over the concatenation of the ranks' inputs in rank order -- equal keys keep
    (source rank, position) order: every partition pass is stable, a 16-bit split never separates equal keys unless
    the whole bucket is one value (then the cut follows (source rank, position) order), the exchange delivers the
    sources' pieces in rank order and the final local sort is stable.  With values = global indices this is the
    permutation the reference's projected sort applies (radix_sort.hpp:1111-1154).

    stats: optional dict; receives {"phase_ms": {...}, "bytes_sent": ..., "keys_received": ...} (adds a device
    synchronisation at the end of every phase).

    stages: sub-ranges per destination for the pipelined exchange (module docstring); None = RSORT_DIST_STAGES or 1 (one
    all-to-all, then one local sort -- the unpipelined form)."""
    stages = default_stages() if stages is None else max(1, int(stages))
    import time
    world = comm.world
    n = keys.numel()
    clock = [time.perf_counter()]
    phase_ms = {}

    def phase(name):
        if stats is not None:
            local.synchronize()
            local.wait_for_producers(keys)
            now = time.perf_counter()
            phase_ms[name] = phase_ms.get(name, 0.0) + (now - clock[0]) * 1e3
            clock[0] = now

    local.wait_for_producers(keys)
    if world == 1:
        local.sort(keys, None, values, None)
        local.synchronize()
        return keys if values is None else (keys, values)
    top = local.top
    two_levels = top >= 2 * DIGIT
    scratch = local.empty_like(keys, n)
    vscratch = local.empty_like(values, n) if values is not None else None

    # 1 + 4a. local MSD partition on the top digit, out of place (independent of the other ranks' counts); the pass
    # hands back the digit counts it needed anyway, and the two buffers swap roles instead of copying back
    h1 = local.partition_out(keys, scratch, top - DIGIT, top, values, vscratch)
    keys, scratch = scratch, keys
    if values is not None:
        values, vscratch = vscratch, values
    phase("partition_top_digit")

    # 2. the counts, all-gathered
    level1_all = comm.all_gather_table(h1, keys.device)                    # [G, 256]
    phase("gather_tables")

    # 3. refine the buckets that hold a quantile; 4b. partition those buckets on the second digit
    refine = boundary_buckets(level1_all, world) if two_levels else []
    level2_all = {}
    offs = np.concatenate(([0], np.cumsum(h1)))
    mine = np.zeros((len(refine), 256), dtype=np.int64)
    if refine:
        for k, b in enumerate(refine):
            lo, hi = int(offs[b]), int(offs[b + 1])
            mine[k] = local.histogram(keys, lo, hi, top - 2 * DIGIT, top - DIGIT)
            local.partition(keys, lo, hi, top - 2 * DIGIT, top - DIGIT, scratch, values, vscratch)
        phase("refine_second_digit")
        gathered = comm.all_gather_table(mine, keys.device)                # [G, len(refine), 256]
        for k, b in enumerate(refine):
            level2_all[b] = gathered[:, k, :]
        phase("gather_tables")
    split16 = plan_splits(level1_all, level2_all, world)

    # 3b. a boundary that the 16-bit resolution leaves far from its quantile: if the bucket holding the quantile is ONE key
    # value on every rank, it is cut by count (module docstring)
    total = int(level1_all.sum())
    tolerance = max(64, total // world // 200)
    candidates = {}                                                         # r -> (x16, keys in front, keys in the bucket)
    for r in range(1, world):
        q = (r * total) // world
        reached = sum(count_below(level1_all[i], {b: t[i] for b, t in level2_all.items()}, int(split16[r])) for i in range(world))
        if abs(reached - q) > tolerance:
            hit = quantile_bucket(level1_all, level2_all, q, two_levels)
            if hit is not None and hit[1] < q < hit[1] + hit[2]:
                candidates[r] = hit
    inner_cuts = {}
    if candidates:
        buckets = sorted({hit[0] for hit in candidates.values()})
        below_bit = top - 2 * DIGIT if two_levels else top - DIGIT
        report = np.zeros((len(buckets), 2), dtype=np.int64)
        for k, x16 in enumerate(buckets):
            b, s2 = x16 >> 8, x16 & 255
            lo = int(offs[b])
            if two_levels:
                row = mine[refine.index(b)]
                lo += int(row[:s2].sum())
                hi = lo + int(row[s2])
            else:
                hi = int(offs[b + 1])
            flag, v = local.single_value(keys, lo, hi, below_bit)
            report[k] = (flag, v)                                           # v < 2^48: at most the digits below the 16-bit prefix
        reports = comm.all_gather_table(report, keys.device)                # [G, buckets, 2]
        for r, (x16, front, _) in candidates.items():
            rep = reports[:, buckets.index(x16), :]
            held = rep[rep[:, 0] != 0]
            if held.size and np.all(held[:, 0] == 1) and np.all(held[:, 1] == held[0, 1]):
                inner_cuts[r] = (x16, (r * total) // world - front)
                split16[r] = x16
        for r in range(1, world + 1):
            split16[r] = max(split16[r], split16[r - 1])
        inner_cuts = {r: cut for r, cut in inner_cuts.items() if split16[r] == cut[0]}
        phase("equal_key_buckets")
    counts = exchange_counts(level1_all, level2_all, split16, world, inner_cuts)
    send_counts, recv_counts = counts[comm.rank], counts[:, comm.rank]

    if stages > 1 and two_levels:
        return _staged_exchange_and_sort(keys, values, scratch, vscratch, local, comm, level1_all, level2_all, split16, inner_cuts,
                                         stages, stats, phase, phase_ms)

    # 5. exchange
    local.synchronize()
    recv = local.empty_like(keys, int(recv_counts.sum()))
    comm.all_to_all(keys, send_counts, recv_counts, recv)
    vrecv = None
    if values is not None:
        vrecv = local.empty_like(values, int(recv_counts.sum()))
        comm.all_to_all(values, send_counts, recv_counts, vrecv)
    if stats is not None:
        local.wait_for_producers(recv)
    phase("exchange")

    # 6. local LSD sort of the received range (once the exchange has delivered it)
    local.wait_for_producers(recv)
    local.sort(recv, scratch, vrecv, vscratch)
    local.synchronize()
    phase("local_sort")
    if stats is not None:
        elem = keys.element_size() + (values.element_size() if values is not None else 0)
        stats["phase_ms"] = phase_ms
        stats["bytes_sent"] = int((send_counts.sum() - send_counts[comm.rank]) * elem)
        stats["keys_received"] = int(recv_counts.sum())
    return recv if values is None else (recv, vrecv)


def _staged_exchange_and_sort(keys, values, scratch, vscratch, local, comm, level1_all, level2_all, split16, inner_cuts, stages, stats,
                              phase, phase_ms):
    """Steps 5 and 6 of distributed_sort in S stages: the exchange of stage s + 1 is in flight while stage s is sorted."""
    import time
    world, me = comm.world, comm.rank
    cuts = stage_cuts(level1_all, split16, world, stages)
    seg, off = stage_counts(level1_all, level2_all, split16, cuts, world, inner_cuts)
    recv_sizes = seg[:, me, :]                                             # [source, stage]
    stage_total = recv_sizes.sum(axis=0)                                   # keys this rank receives per stage
    stage_start = np.concatenate(([0], np.cumsum(stage_total)))
    total_recv = int(stage_total.sum())
    recv = local.empty_like(keys, total_recv)
    vrecv = local.empty_like(values, total_recv) if values is not None else None

    def pieces(s):
        send_k = [keys.narrow(0, int(off[me, r, s]), int(seg[me, r, s])) for r in range(world)]
        at = int(stage_start[s]) + np.concatenate(([0], np.cumsum(recv_sizes[:, s])))
        recv_k = [recv.narrow(0, int(at[i]), int(recv_sizes[i, s])) for i in range(world)]
        if values is None:
            return [send_k], [recv_k]
        send_v = [values.narrow(0, int(off[me, r, s]), int(seg[me, r, s])) for r in range(world)]
        recv_v = [vrecv.narrow(0, int(at[i]), int(recv_sizes[i, s])) for i in range(world)]
        return [send_k, send_v], [recv_k, recv_v]

    local.synchronize()                                                    # the partition passes that wrote `keys` are done
    t_begin = time.perf_counter()
    stage_ms = []
    pending = comm.exchange_pieces(*pieces(0))
    for s in range(stages):
        nxt = comm.exchange_pieces(*pieces(s + 1)) if s + 1 < stages else None     # in flight while stage s is sorted
        pending.wait()
        local.wait_for_producers(recv)
        lo, m = int(stage_start[s]), int(stage_total[s])
        if m > 1:
            local.sort(recv.narrow(0, lo, m), scratch if scratch.numel() >= m else None,
                       vrecv.narrow(0, lo, m) if vrecv is not None else None,
                       vscratch if vscratch is not None and vscratch.numel() >= m else None)
        if stats is not None:
            local.synchronize()
            stage_ms.append(round((time.perf_counter() - t_begin) * 1e3, 3))      # when stage s was exchanged AND sorted
        pending = nxt
    local.synchronize()
    phase("exchange_and_local_sort_pipelined")
    if stats is not None:
        elem = keys.element_size() + (values.element_size() if values is not None else 0)
        stats["phase_ms"] = phase_ms
        stats["stage_done_ms"] = stage_ms
        stats["stages"] = stages
        stats["bytes_sent"] = int((seg[me].sum() - seg[me, me].sum()) * elem)
        stats["keys_received"] = total_recv
    return recv if values is None else (recv, vrecv)
