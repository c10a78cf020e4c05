#!/usr/bin/env python3
"""bench.py -- Gkeys/s of the MI355X sort engine on BASELINE.json's metric.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--keys KEYS] [--types u32,u64] [--shapes ...]

One "step" = one complete sort (histogram + scan + all onesweep passes) of N keys that are already
resident in HBM; the pristine input is restored device-to-device OUTSIDE the timed region, as the
reference's harness copies its vector outside the timer (benchmark.cpp:158-160).

N = 1 : workload = BASELINE configs[1]: uint32 and uint64, five generator shapes, N = 2^30, LSD
        onesweep on one MI355X.  `value` is the uint32 / randomized row (the case BASELINE's metric is
        quoted on first); every (type, shape) row is in `rows`.  Behind them, with fewer repetitions:
        configs[2] (int64 / float / double x randomized, duplicates), stable key+payload pair sorts
        (`pair_rows`), the string-key sort through the C-ABI (`string_rows`) and configs[3], the
        composite {std::string key; long long data} projected sort through include/radix_sort.hpp
        (`composite_row`, tools/frontend_bench).
N > 1 : BASELINE configs[4]: uint64 keys, N/GPU = --keys, MSD splitter + all-to-all + local LSD sort
        (radix-sort_amd/dist.py); `value` = total keys over all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def algorithmic_bytes_per_key(kbytes, vbytes=0):
    """canonical 8-bit-digit LSD traffic, B = K*(1 + 2P), P = K passes."""
    p = kbytes
    return kbytes + 2 * p * (kbytes + vbytes)


def cpu_baseline(n_sample):
    """The reference's own multi-threaded sort, timed on this node's host cores (rank 0, N = 1 only).

    kind "reference": oracle/_ref/libref_sort.so = radix_sort.hpp itself compiled in the build
    container.  Falls back to the serial restatement (kind "port") if that library is absent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from cpu_checkers import load_oracle, load_ref
    from radix_sort_amd import shapes

    v = shapes.generate("u32", n_sample, "randomized")
    ref = load_ref()
    if ref is not None:
        secs = []
        for _ in range(3):                                   # benchmark.cpp:185 averages 3 iterations
            out, s = ref.sort_keys(v, 0, mt=True)
            secs.append(s)
        assert bool(np.all(out[:-1] <= out[1:]))
        sec = sum(secs) / len(secs)
        return {"value": n_sample / sec / 1e9, "unit": "Gkeys/s", "cores": int(ref.num_threads(n_sample)),
                "kind": "reference",
                "sample": f"uint32 randomized N={n_sample} radix_sort::sort(v,{{}},true), mean of 3 (BASELINE config 1)",
                "host_cpus": os.cpu_count()}
    orc = load_oracle()
    n_small = min(n_sample, 20_000_000)
    v = v[:n_small]
    t0 = time.perf_counter()
    out = orc.sort_keys(v, 0, mt=True)
    sec = time.perf_counter() - t0
    return {"value": n_small / sec / 1e9, "unit": "Gkeys/s", "cores": 1, "kind": "port",
            "sample": f"uint32 randomized N={n_small}, serial restatement", "host_cpus": os.cpu_count()}


def bench_single(args):
    import torch
    import radix_sort_amd as rs
    from radix_sort_amd import shapes

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    eng = rs.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    eng.set_profiling(True)
    n = args.n
    rows = []
    headline = None
    dominant = None
    plan = [(name, args.shapes, args.steps, args.warmup) for name in args.types]
    plan += [(name, args.extra_shapes, args.extra_steps, 1) for name in args.extra_types if name not in args.types]
    for name, shape_list, steps, warmup in plan:
        dt, kb, xf = shapes.KEY_TYPES[name]
        eng.reserve(n, kb, 0)
        work = torch.empty(n, dtype=torch.int32 if kb == 4 else torch.int64, device=dev)
        alt = torch.empty_like(work)
        for shape in shape_list:
            src = shapes.generate_device(name, n, shape, dev)

            def timed(precheck):
                eng.set_precheck(precheck)
                times, pass_ms, hist_ms, group_ms = [], [], [], []
                hybrid = False
                for it in range(warmup + steps):
                    work.copy_(src)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    eng.sort_keys_(work, xf, alt=alt)
                    eng.synchronize()
                    t1 = time.perf_counter()
                    if it >= warmup:
                        times.append(t1 - t0)
                        prof = eng.profile()          # empty when the order pre-check answered the call
                        hybrid = bool(eng.lib.rs_debug_last_hybrid(eng.h))
                        if eng.lib.rs_debug_last_distinct(eng.h):
                            hybrid = "distinct"       # [counting read, read-back + host order of the distinct keys + fill]
                        if prof and hybrid is True:   # [histogram, sweep, sweep, group table + group sort, big-group kernel]
                            hist_ms.append(prof[0])
                            pass_ms.extend(prof[1:3])
                            group_ms.append(sum(prof[3:]))
                        elif prof:
                            hist_ms.append(prof[0])
                            pass_ms.extend(prof[1:])
                return times, pass_ms, hist_ms, group_ms, hybrid

            # product behaviour: order pre-check on (the reference's isSortedBi fast exits, radix_sort.hpp:269-305)
            times, pass_ms, hist_ms, group_ms, hybrid = timed(True)
            full = None
            if not pass_ms:                           # ordered input: also time the sort engine proper
                full = timed(False)
            # size-independent parity property at full size: sortedness + checksum (sum of bits)
            sign = -(1 << 31) if kb == 4 else -(1 << 63)
            if xf == 1:
                okey = work                                            # two's complement: signed compare as is
            elif xf == 0:
                okey = work ^ sign                                     # unsigned order via sign flip
            else:                                                      # IEEE-754 bits: negatives reverse (radix_sort.hpp:405)
                okey = torch.where(work < 0, ~work ^ sign, work)
            sorted_ok = bool((okey[1:] >= okey[:-1]).all().item())
            sum_ok = bool((work.sum() == src.sum()).item())
            del okey
            sec = sum(times) / len(times)
            b_key = algorithmic_bytes_per_key(kb)

            def form_of(hyb, group):
                """What ran, and the bytes per key it moves by design (SURVEY 8d asks for both: the fraction is priced
                against the canonical 8-bit LSD bytes whatever form ran).  Hybrid form of 4-byte keys: histogram read K,
                two sweeps 2 * 2K, one in-LDS group sort 2K = 7K = 28 bytes."""
                if not hyb:
                    return {"form": "lsd", "bytes_moved_per_key": b_key}
                if hyb == "distinct":     # hist_ms = the counting read, pass_ms = the fill (with the host step before it)
                    return {"form": "few distinct keys: counted in one read (hash table per workgroup), written back as runs",
                            "bytes_moved_per_key": 2 * kb}
                return {"form": "hybrid: 2 onesweep passes on the top digits + in-LDS sort of every 16-bit group",
                        "bytes_moved_per_key": 7 * kb, "group_ms": sum(group) / len(group) if group else None}

            row = {"type": name, "shape": shape, "n": n, "ms": sec * 1e3, "gkeys_s": n / sec / 1e9,
                   "hbm_frac": n * b_key / sec / 1e9 / HBM_PEAK_GBS if pass_ms else None,   # no sort traffic moved when the pre-check answers
                   "answered_by_precheck": not pass_ms, **(form_of(hybrid, group_ms) if pass_ms else {}),
                   "hist_ms": sum(hist_ms) / len(hist_ms) if hist_ms else None,
                   "pass_ms": sum(pass_ms) / len(pass_ms) if pass_ms else None,
                   "sorted": sorted_ok, "checksum": sum_ok}
            if full is not None:
                fsec = sum(full[0]) / len(full[0])
                row["full_sort"] = {"ms": fsec * 1e3, "gkeys_s": n / fsec / 1e9, "hbm_frac": n * b_key / fsec / 1e9 / HBM_PEAK_GBS,
                                    "hist_ms": sum(full[2]) / len(full[2]), "pass_ms": sum(full[1]) / len(full[1]),
                                    **form_of(full[4], full[3])}
            rows.append(row)
            if headline is None:
                headline = row
                avg_pass = (row["pass_ms"] if row["pass_ms"] else row["full_sort"]["pass_ms"]) * 1e-3
                achieved = n * 2 * kb / avg_pass / 1e9
                dominant = {"bound": "hbm", "kernel": f"onesweep_kernel<{name}> (all launches of one sort: the counting and the last-pass instantiation of the chain-mode sweep)", "achieved": achieved,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "traffic": load_traffic(name),
                            "traffic_source": "profiles/traffic.json (" + str(load_traffic(name, True)[1]) + "): rocprofv3 --pmc passes of the committed profile set, not collected in this run",
                            "bytes_per_launch": n * 2 * kb, "avg_launch_ms": avg_pass * 1e3,
                            "whole_sort_frac": row["hbm_frac"] if row["hbm_frac"] else row["full_sort"]["hbm_frac"],
                            "whole_sort_bytes_moved_per_key": row.get("bytes_moved_per_key") or row["full_sort"].get("bytes_moved_per_key")}
                gms = row.get("group_ms") or (row.get("full_sort") or {}).get("group_ms")
                if gms:     # the hybrid form's second kernel: one read + one write of the keys per launch
                    dominant["group_sort_kernel"] = {"bytes_per_launch": n * 2 * kb, "avg_launch_ms": gms,
                                                     "achieved": n * 2 * kb / (gms * 1e-3) / 1e9, "frac": n * 2 * kb / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del src
        del work, alt
        torch.cuda.empty_cache()
    pair_rows = [bench_pairs(eng, torch, dev, spec, args) for spec in args.pairs]
    eng.set_profiling(False)
    size_rows = [bench_size(eng, torch, dev, name, m) for m in args.sizes for name in ("u32", "u64")]
    string_rows = [bench_strings(eng, m) for m in args.strings]
    string_rows += [bench_strings(eng, args.strings[0], shape) for shape in args.string_shapes if args.strings]
    eng.close()
    return headline, rows, dominant, pair_rows, string_rows, size_rows


def bench_size(eng, torch, dev, name, n):
    """Device-resident keys-only sort at the sizes of the reference's own benchmark (main.cpp:19-21: up to 10^8; BASELINE
    configs[0] is the 10^8 uint32 row on the host CPU): randomized keys, order pre-check on, host clock around call + wait,
    median of 20.  Below the headline size other forms run (narrow hybrid for 8-byte keys, four-pass loop for 4-byte keys)."""
    from radix_sort_amd import shapes
    dt, kb, xf = shapes.KEY_TYPES[name]
    src = shapes.generate_device(name, n, "randomized", dev)
    work = torch.empty_like(src)
    alt = torch.empty_like(src)
    eng.set_precheck(True)
    ts = []
    for it in range(23):
        work.copy_(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sort_keys_(work, xf, alt=alt)
        eng.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[3:])
    sec = ts[len(ts) // 2]
    form = {0: "lsd", 1: "hybrid (groups of equal top 16 bits)", 2: "narrow hybrid (groups of equal top 9-15 bits)"}[int(eng.lib.rs_debug_last_hybrid(eng.h))]
    sign = -(1 << 31) if kb == 4 else -(1 << 63)
    okey = work ^ sign
    ok = bool((okey[1:] >= okey[:-1]).all().item()) and bool((work.sum() == src.sum()).item())
    b_key = algorithmic_bytes_per_key(kb)
    return {"type": name, "shape": "randomized", "n": n, "ms": sec * 1e3, "gkeys_s": n / sec / 1e9, "form": form,
            "hbm_frac": n * b_key / sec / 1e9 / HBM_PEAK_GBS, "sorted_and_checksum": ok}


def bench_pairs(eng, torch, dev, spec, args):
    """Stable (key, payload) sort, device-resident: keys randomized, payload = 0..n-1 (the permutation sortByIndices
    applies, radix_sort.hpp:1111-1154).  Canonical bytes B = K + 2P(K+V).  Checked at full size: keys ascending,
    keys_out == keys_in[payload] (the payload is the permutation), equal neighbours keep payload order (stability)."""
    from radix_sort_amd import shapes
    name, vb = spec.split("+")
    vb = int(vb)
    dt, kb, xf = shapes.KEY_TYPES[name]
    n = min(args.n, args.pairs_n)
    src = shapes.generate_device(name, n, "randomized", dev)
    kdt = torch.int32 if kb == 4 else torch.int64
    vdt = torch.int32 if vb == 4 else torch.int64
    keys = torch.empty(n, dtype=kdt, device=dev)
    vals = torch.empty(n, dtype=vdt, device=dev)
    alt_k, alt_v = torch.empty_like(keys), torch.empty_like(vals)
    iota = torch.arange(n, dtype=vdt, device=dev)
    eng.set_precheck(True)
    times, pass_ms, group_ms = [], [], []
    hybrid = False
    for it in range(1 + args.extra_steps):
        keys.copy_(src)
        vals.copy_(iota)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sort_pairs_(keys, vals, xf, alt_k=alt_k, alt_v=alt_v)
        eng.synchronize()
        t1 = time.perf_counter()
        if it >= 1:
            times.append(t1 - t0)
            prof = eng.profile()
            hybrid = bool(eng.lib.rs_debug_last_hybrid(eng.h))
            if hybrid:                                # [histogram, sweep, sweep, group table + group sort]
                pass_ms.extend(prof[1:3])
                group_ms.append(sum(prof[3:]))
            else:
                pass_ms.extend(prof[1:])
    del alt_k, alt_v, iota
    sign = -(1 << 31) if kb == 4 else -(1 << 63)
    okey = keys if xf == 1 else (keys ^ sign if xf == 0 else torch.where(keys < 0, ~keys ^ sign, keys))
    asc = okey[1:] >= okey[:-1]
    stable = ((okey[1:] != okey[:-1]) | (vals[1:] > vals[:-1]))
    ok_sorted = bool(asc.all().item())
    ok_stable = bool(stable.all().item())
    del asc, stable, okey
    ok_perm = bool((keys == src[vals.to(torch.int64)]).all().item())
    sec = sum(times) / len(times)
    b_key = algorithmic_bytes_per_key(kb, vb)
    return {"type": name, "value_bytes": vb, "shape": "randomized", "n": n, "ms": sec * 1e3, "gkeys_s": n / sec / 1e9,
            "bytes_per_key": b_key, "hbm_frac": n * b_key / sec / 1e9 / HBM_PEAK_GBS,
            "form": "hybrid: 2 onesweep passes on the top digits + stable in-LDS sort of every 16-bit group, payload beside the keys" if hybrid else "lsd",
            "bytes_moved_per_key": kb + 3 * 2 * (kb + vb) if hybrid else b_key,
            "pass_ms": sum(pass_ms) / len(pass_ms) if pass_ms else None,
            "group_ms": sum(group_ms) / len(group_ms) if group_ms else None,
            "sorted": ok_sorted, "stable": ok_stable, "checksum": ok_perm}


def string_workload(m, shape, rng):
    """(blob, offsets) of m strings, vectorised (the generators of radix-sort_amd/shapes.py loop in Python):
    randomized  0..20 uniform bytes (generators.hpp:264-291);
    prefix24    the same behind a 24-byte prefix every string shares (several MSD levels see one chunk value);
    shortlex    the reference's "sorted" string shape (generators.hpp:319-347): "", the 256 one-byte strings, the 65536
                two-byte strings, ... in counting order -- NOT std::string order, so it is really sorted;
    dup256      256 distinct strings of 0..20 bytes drawn uniformly (the "duplicates" shape: large segments at every level)."""
    import numpy as np
    if shape == "shortlex":
        idx = np.arange(m, dtype=np.int64)
        lens = np.where(idx < 1, 0, np.where(idx < 257, 1, np.where(idx < 65793, 2, 3)))
        offsets = np.zeros(m + 1, dtype=np.uint64)
        np.cumsum(lens, out=offsets[1:].view(np.int64))
        blob = np.empty(int(offsets[-1]), dtype=np.uint8)
        start = offsets[:-1].astype(np.int64)
        for length, first in ((1, 1), (2, 257), (3, 65793)):
            sel = np.flatnonzero(lens == length)
            v = idx[sel] - first
            for b in range(length):
                blob[start[sel] + b] = (v >> (8 * (length - 1 - b))) & 255
        return blob, offsets
    if shape == "dup256":
        ulen = rng.integers(0, 21, size=256)
        tab = rng.integers(0, 256, size=(256, 20), dtype=np.uint8)
        pick = rng.integers(0, 256, size=m)
        lens = ulen[pick]
        offsets = np.zeros(m + 1, dtype=np.uint64)
        np.cumsum(lens, out=offsets[1:].view(np.int64))
        blob = np.empty(int(offsets[-1]), dtype=np.uint8)
        step = 1_000_000
        for lo in range(0, m, step):                        # (in slices: the mask of a slice is 20 MB, not 200)
            hi = min(lo + step, m)
            rows = tab[pick[lo:hi]]
            mask = np.arange(20)[None, :] < lens[lo:hi, None]
            blob[int(offsets[lo]):int(offsets[hi])] = rows[mask]
        return blob, offsets
    pre = 24 if shape == "prefix24" else 0
    lens = rng.integers(0, 21, size=m, dtype=np.int64) + pre
    offsets = np.zeros(m + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:].view(np.int64))
    blob = rng.integers(0, 256, size=int(offsets[-1]), dtype=np.uint8)
    if pre:
        start = offsets[:-1].astype(np.int64)
        for b in range(pre):
            blob[start + b] = 65 + b
    return blob, offsets


def bench_strings(eng, m, shape="randomized"):
    """rs_sort_strings32 through the C-ABI: m strings (string_workload) in host memory, the permutation back in host
    memory (upload and download included: the entry point takes host buffers; 4-byte offsets and permutation, as the
    front-end passes them for blobs below 4 GiB).  Checked: the result is a permutation and 200,000 sampled neighbour
    pairs are in std::string order, ties by index."""
    import numpy as np
    rng = np.random.default_rng(69)
    blob, offsets = string_workload(m, shape, rng)
    off32 = offsets.astype(np.uint32)                                       # what the front-end packs for blobs below 4 GiB
    eng.sort_strings_host(blob[: 1 << 16], offsets[:1001].copy())          # warm-up (context, kernels)
    times, dev_ms = [], []
    eng.set_profiling(True)
    host_threads = min(12, os.cpu_count() or 1)     # what radix_sort::sort(v, proj, true) asks for (hostThreads, include/radix_sort.hpp)
    eng.set_host_threads(host_threads)
    for _ in range(3):
        t0 = time.perf_counter()
        perm = eng.sort_strings_host(blob, off32, stable=True, narrow=True, widen=False)
        times.append(time.perf_counter() - t0)
        dev_ms.append(sum(eng.profile()))       # HIP events around the device part (between upload and download)
    eng.set_profiling(False)
    eng.set_host_threads(1)
    ok_perm = bool((np.bincount(perm.astype(np.int64), minlength=m) == 1).all())
    idx = rng.integers(0, m - 1, size=200_000)
    ok_order = True
    raw = blob.tobytes()
    off = offsets.astype(np.int64)
    for i in idx:
        a, b = int(perm[i]), int(perm[i + 1])
        sa, sb = raw[off[a]:off[a + 1]], raw[off[b]:off[b + 1]]
        if sa > sb or (sa == sb and a > b):
            ok_order = False
            break
    sec = min(times)
    moved = int(offsets[-1]) + 4 * (m + 1) + 4 * m       # blob + 4-byte offsets up, 4-byte permutation down
    # Algorithmic HBM bytes of the first MSD level per string (every string takes part in it; later levels only the
    # unresolved ones): the chunk kernel reads the offsets (16) and <= 7 string bytes and writes an 8-byte key and a 4-byte
    # id (35); the stable pair sort of (8-byte chunk key, 4-byte id) is canonical K + 2*8*(K+V) = 200; heads, open flags,
    # two scans, the gathers and the write-back of the permutation are about 17 four-byte accesses (68): 303 bytes.
    level0_bytes = 303
    dev = min(dev_ms) if dev_ms and min(dev_ms) > 0 else None
    return {"strings": m, "shape": shape, "levels": eng.last_string_levels(), "ms": sec * 1e3, "mstrings_s": m / sec / 1e6, "host_bytes_per_string": moved / m,
            "includes": "H2D of blob+offsets, MSD sort, D2H of the permutation", "host_threads": host_threads,
            "device_ms": dev, "device_mstrings_s": m / dev / 1e3 if dev else None,
            "device_bytes_per_string_level0": level0_bytes,
            "device_hbm_frac_level0": m * level0_bytes / (dev * 1e-3) / 1e9 / HBM_PEAK_GBS if dev else None,
            "sorted": ok_order, "checksum": ok_perm}


def bench_composite(n):
    """BASELINE configs[3]: {std::string key; long long data}, projected-key sort through the drop-in header, host vectors in
    and out, checked against std::stable_sort element by element (tools/frontend_bench.cpp)."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "frontend_bench")
    if not os.path.exists(exe):
        return {"error": "tools/frontend_bench not built"}
    try:
        out = subprocess.run([exe, "1000", str(n)], capture_output=True, text=True, timeout=600).stdout
    except Exception as e:                                  # noqa: BLE001 -- a bench row must not take the line down
        return {"error": repr(e)}
    res = {}
    for line in out.splitlines():
        mt = re.search(r"^\{(.*)\} randomized N=(\d+) .*end-to-end: ([0-9.]+) ms = ([0-9.]+) Gkeys/s \(std::stable_sort ([0-9.]+) ms\)\s+(\w+)", line)
        if mt:
            row = {"workload": "{" + mt.group(1) + "} randomized, radix_sort::sort(v, &T::key, true), host vectors in and out",
                   "n": int(mt.group(2)), "ms": float(mt.group(3)), "gkeys_s": float(mt.group(4)),
                   "std_stable_sort_ms": float(mt.group(5)), "matches_std_stable_sort": mt.group(6) == "OK"}
            res["string_key" if "string" in mt.group(1) else "pod_int_key"] = row
    if res:
        return res
    return {"error": "no result line", "output": out[-400:]}


def load_traffic(name, with_source=False):
    """HBM bytes per onesweep launch from the committed PMC passes (profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE collected by tools/profile_round.sh, NOT measured in this run), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            d = json.load(f)
        return (d.get(name), d.get("_source")) if with_source else d.get(name)
    except (OSError, ValueError):
        return (None, None) if with_source else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--keys", dest="n", type=int, default=1 << 30, help="keys (per GPU when --gpus > 1)")
    ap.add_argument("--types", default="u32,u64")
    ap.add_argument("--shapes", default="randomized,sorted,reverse_sorted,nearly_sorted,duplicates")
    ap.add_argument("--extra-types", default="i64,f32,f64", help="BASELINE configs[2] rows (fewer repetitions)")
    ap.add_argument("--extra-shapes", default="randomized,duplicates,nearly_sorted")
    ap.add_argument("--extra-steps", type=int, default=3)
    ap.add_argument("--pairs", default="u32+4,u64+4", help="key type + payload bytes")
    ap.add_argument("--pairs-n", type=int, default=1 << 29)
    ap.add_argument("--strings", default="10000000,50000000", help="string counts for the rs_sort_strings rows")
    ap.add_argument("--string-shapes", default="prefix24,shortlex,dup256", help="harder string inputs, at the first --strings count")
    ap.add_argument("--sizes", default="10000000,100000000", help="device-resident randomized rows at the reference benchmark's own sizes")
    ap.add_argument("--composite-n", type=int, default=10_000_000, help="0 = skip the configs[3] row")
    ap.add_argument("--cpu-n", type=int, default=100_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    args.types = [t for t in args.types.split(",") if t]
    args.shapes = [s for s in args.shapes.split(",") if s]
    args.extra_types = [t for t in args.extra_types.split(",") if t]
    args.extra_shapes = [s for s in args.extra_shapes.split(",") if s]
    args.pairs = [p for p in args.pairs.split(",") if p]
    args.strings = [int(x) for x in args.strings.split(",") if x]
    args.sizes = [int(x) for x in args.sizes.split(",") if x]
    args.string_shapes = [x for x in args.string_shapes.split(",") if x]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.gpus > 1 or os.environ.get("RS_BENCH_FORCE_DIST"):
        from radix_sort_amd import dist_bench
        return dist_bench.main(args)

    headline, rows, roof, pair_rows, string_rows, size_rows = bench_single(args)
    out = {
        "metric": "Gkeys/s (uint32/uint64, N=2^30) at 1/2/4/8 MI355X; % HBM roofline",
        "value": headline["gkeys_s"], "unit": "Gkeys/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": headline["ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": headline["type"], "data": "synthetic",
        "config": {"workload": f"{headline['type']} {headline['shape']} N={headline['n']} keys-only sort "
                               f"({headline.get('form') or headline['full_sort'].get('form')}), "
                               "device-resident (BASELINE configs[1]); all (type, shape) rows in `rows`"},
        "roofline": roof,
        "rows": rows,
        "pair_rows": pair_rows,
        "string_rows": string_rows,
        "size_rows": size_rows,
    }
    if args.composite_n:
        out["composite_row"] = bench_composite(args.composite_n)
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.cpu_n)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
