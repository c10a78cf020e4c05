"""The five input shapes of the reference's generators, restated for numpy (host) and torch (device).

Semantics follow /root/reference/RadixSort/generators.hpp (cited per branch); the random stream does
not: the reference draws from std::mt19937_64(69) through std::uniform_int_distribution, whose output
is implementation-defined (MSVC != libstdc++), so no generator outside the author's machine can
reproduce the author's bytes.  Everything here is keyed by (seed, element index) through numpy's
PCG64 / torch's Philox instead, and parity tests always feed the SAME bytes to both sides.

These are measurement/test inputs only; nothing here is on the sort path.
"""
from __future__ import annotations

import math

import numpy as np

SEED = 69  # generators.hpp:56
SHAPES = ("randomized", "sorted", "reverse_sorted", "nearly_sorted", "duplicates")  # generators.hpp:39-46

# name -> (numpy dtype, key bytes, transform code of the C-ABI: 0 none, 1 signed, 2 IEEE-754)
KEY_TYPES = {
    "u8": (np.uint8, 1, 0), "i8": (np.int8, 1, 1),
    "u16": (np.uint16, 2, 0), "i16": (np.int16, 2, 1),
    "u32": (np.uint32, 4, 0), "i32": (np.int32, 4, 1),
    "u64": (np.uint64, 8, 0), "i64": (np.int64, 8, 1),
    "f32": (np.float32, 4, 2), "f64": (np.float64, 8, 2),
}


def _uint_of(bytes_):
    return {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[bytes_]


def _random_bits(rng, n, kbytes):
    """Uniform raw bit patterns: full-range integers (generators.hpp:82, :88-89) and, for floats,
    every bit pattern including NaN/inf/denormals (generators.hpp:168, :174-185)."""
    u = _uint_of(kbytes)
    return rng.integers(0, np.iinfo(u).max, size=n, dtype=u, endpoint=True)


def _sorted_ramp(name, n):
    """key[i] = min + i * step, last = max (generators.hpp:108-129 integers, :214-239 floats)."""
    dt, kbytes, xf = KEY_TYPES[name]
    i = np.arange(n, dtype=np.float64)
    if xf == 2:
        lo, hi = float(np.finfo(dt).min), float(np.finfo(dt).max)
        if kbytes == 4:
            v = (lo + i * ((hi - lo) / (n - 1))).astype(dt)                       # :220-226
        else:
            step = hi / (n - 1) - lo / (n - 1)                                     # :231-237
            with np.errstate(over="ignore", invalid="ignore"):
                v = lo + i * (step if math.isfinite(step) else hi)
            v = np.clip(np.nan_to_num(v, nan=lo), lo, hi).astype(dt)
        v[-1] = np.finfo(dt).max
        return v
    info = np.iinfo(dt)
    umax = (1 << (8 * kbytes)) - 1
    if kbytes < 8:
        acc = float(info.min) + i * (float(umax) / (n - 1))                        # :114-118
        acc = np.clip(acc, float(info.min), float(info.max))
        v = acc.astype(np.int64).astype(dt)
    else:
        k = np.arange(n, dtype=np.uint64) * np.uint64(umax // (n - 1))             # :123-127, wraps mod 2^64
        if xf == 1:
            k = k + np.uint64(1 << 63)                                             # + INT64_MIN, as bits
        v = k.view(dt).copy()
    v[-1] = info.max                                                               # :119, :128
    return v


def generate(name: str, n: int, shape: str = "randomized", seed: int = SEED) -> np.ndarray:
    """numpy array of key type `name` (see KEY_TYPES) with n elements of the given shape."""
    dt, kbytes, xf = KEY_TYPES[name]
    if shape not in SHAPES:
        raise ValueError(f"unknown shape {shape!r}")
    if n <= 0:
        return np.empty(0, dtype=dt)                                               # generators.hpp:395-396
    if n == 1:
        return np.zeros(1, dtype=dt)                                               # :397-398
    rng = np.random.Generator(np.random.PCG64(seed))
    if shape == "randomized":
        return _random_bits(rng, n, kbytes).view(dt)
    if shape == "duplicates":                                                      # :93-107, :189-213
        uniq = _random_bits(rng, 256, kbytes)
        return uniq[rng.integers(0, 256, size=n)].view(dt)
    v = _sorted_ramp(name, n)
    if shape == "reverse_sorted":                                                  # :131-132
        return v[::-1].copy()
    if shape == "nearly_sorted":                                                   # :133-144
        swaps = math.ceil(n * 0.05)
        ij = rng.integers(0, n, size=(swaps, 2))
        for a, b in ij.tolist():                                                   # sequential, as the reference
            v[a], v[b] = v[b], v[a]
    return v


def generate_strings(n: int, shape: str = "randomized", max_len: int = 20, seed: int = SEED):
    """(blob uint8[total], offsets uint64[n+1]) -- strings of length 0..max_len over bytes 0..255
    (generators.hpp:263-369).  "sorted" is the reference's shortlex enumeration (:319-347), which is
    NOT lexicographic order."""
    if shape not in SHAPES:
        raise ValueError(f"unknown shape {shape!r}")
    if n <= 0:
        return np.empty(0, np.uint8), np.zeros(1, np.uint64)
    if n == 1:
        return np.empty(0, np.uint8), np.zeros(2, np.uint64)
    rng = np.random.Generator(np.random.PCG64(seed))

    def rand_strings(m):
        lens = rng.integers(0, max_len, size=m, endpoint=True)
        return [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in lens]

    if shape == "randomized":
        strs = rand_strings(n)
    elif shape == "duplicates":
        uniq = rand_strings(256)
        strs = [uniq[i] for i in rng.integers(0, 256, size=n)]
    else:
        strs = []
        length, idx = 0, 0
        for _ in range(n):                                                         # shortlex counter
            strs.append(idx.to_bytes(length, "big") if length else b"")
            idx += 1
            if idx >= 256 ** length:
                length, idx = length + 1, 0
        if shape == "reverse_sorted":
            strs.reverse()
        elif shape == "nearly_sorted":
            swaps = math.ceil(n * 0.05)
            for a, b in rng.integers(0, n, size=(swaps, 2)).tolist():
                strs[a], strs[b] = strs[b], strs[a]
    return pack_strings(strs)


def pack_strings(strs):
    lens = np.fromiter((len(s) for s in strs), dtype=np.uint64, count=len(strs))
    offsets = np.zeros(len(strs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    blob = np.frombuffer(b"".join(strs), dtype=np.uint8).copy()
    return blob, offsets


def unpack_strings(blob, offsets):
    b = blob.tobytes()
    o = offsets.tolist()
    return [b[o[i]:o[i + 1]] for i in range(len(o) - 1)]


# ---------------------------------------------------------------------------------------------
# device-side generation for the bench sizes (N = 2^30 does not fit a host round trip cheaply)
# ---------------------------------------------------------------------------------------------
def generate_device(name: str, n: int, shape: str, device, seed: int = SEED):
    """torch tensor on `device` holding the raw key bits as int32 / int64 (torch has no unsigned
    arithmetic; the sort only ever sees bits).  Same shape definitions as generate()."""
    import torch

    dt, kbytes, xf = KEY_TYPES[name]
    if kbytes not in (4, 8):
        raise ValueError("device generation covers 4- and 8-byte keys")
    tdt = torch.int32 if kbytes == 4 else torch.int64
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lo, hi = (-(1 << 31), (1 << 31)) if kbytes == 4 else (-(1 << 63), (1 << 63) - 1)

    def rand_bits(m):
        if kbytes == 4:
            return torch.randint(lo, hi, (m,), dtype=torch.int64, device=device, generator=g).to(torch.int32)
        # randint's high is exclusive and capped at 2^63-1: draw two 32-bit halves instead
        a = torch.randint(0, 1 << 32, (m,), dtype=torch.int64, device=device, generator=g)
        b = torch.randint(0, 1 << 32, (m,), dtype=torch.int64, device=device, generator=g)
        return (a << 32) | b

    if n <= 1:
        return torch.zeros(max(n, 0), dtype=tdt, device=device)
    if shape == "randomized":
        return rand_bits(n)
    if shape == "duplicates":
        uniq = rand_bits(256)
        return uniq[torch.randint(0, 256, (n,), device=device, generator=g)]

    i = torch.arange(n, dtype=torch.int64, device=device)
    if xf == 2:
        fdt = torch.float32 if kbytes == 4 else torch.float64
        flo, fhi = torch.finfo(fdt).min, torch.finfo(fdt).max
        if kbytes == 4:
            v = (flo + i.to(torch.float64) * ((fhi - flo) / (n - 1))).to(fdt)
        else:
            v = (flo + i.to(torch.float64) * (fhi / (n - 1) - flo / (n - 1))).clamp(flo, fhi)
        v[-1] = fhi
        v = v.view(tdt)
    elif kbytes == 4:
        umax = float((1 << 32) - 1)
        base = 0.0 if xf == 0 else float(-(1 << 31))
        acc = (base + i.to(torch.float64) * (umax / (n - 1))).clamp(base, base + umax)
        v = acc.to(torch.int64).to(torch.int32)          # wraps uint32 values >= 2^31 to their int32 bits
        v[-1] = -1 if xf == 0 else (1 << 31) - 1
    else:
        step = ((1 << 64) - 1) // (n - 1)
        v = i * step                                      # int64 multiply wraps mod 2^64 == the unsigned ramp
        if xf == 1:
            v = v + (-(1 << 63))
        v[-1] = -1 if xf == 0 else (1 << 63) - 1
    del i
    if shape == "reverse_sorted":
        return v.flip(0).contiguous()
    if shape == "nearly_sorted":
        # ceil(5 % n) transpositions (generators.hpp:137-143).  The reference draws both ends
        # independently; here the ends are distinct by construction (one index per block of a
        # 2*swaps-way split, ends paired at random) so the vectorised swap stays a permutation.
        swaps = math.ceil(n * 0.05)
        blocks = 2 * swaps
        width = n // blocks
        if width >= 1:
            pick = torch.arange(blocks, dtype=torch.int64, device=device) * width + torch.randint(
                0, width, (blocks,), device=device, generator=g)
            pick = pick[torch.randperm(blocks, device=device, generator=g)]
            a, b = pick[:swaps], pick[swaps:]
            va, vb = v[a].clone(), v[b].clone()
            v[a], v[b] = vb, va
    return v
