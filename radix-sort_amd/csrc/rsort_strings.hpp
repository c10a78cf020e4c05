// rsort_strings.hpp -- device code of the std::string key path (MSD over fixed-width chunks).
//
// The reference sorts strings MSD, one byte per level, with bucket 256 = "past the end" placed
// first (radix_sort.hpp:107-110, :196-200), and hands a region to std::sort / std::stable_sort
// from depth 3 on (:559-569, :1241-1255).  The result of all that is std::string::operator<
// order (unsigned bytes, a proper prefix first), stable on the projected path.
//
// On the GPU the same order is produced MSD over CHUNKS of up to 7 bytes:
//   chunk key (u64) = segment id in the top sbits bits (level 0: none)
//                     | the next c bytes, big-endian, zero padded, in bits 8c+7..8
//                     | min(bytes remaining, c) in bits 7..0,          c = min(7, (56 - sbits) / 8)
// Zero padding alone would make "a" and "a\0" tie; the length byte breaks that tie the right way
// (shorter first), and it cannot outvote a payload byte because it is the least significant one.
// Two strings tie on a chunk key only if they are in the same segment (they tied on every earlier chunk),
// agree on those c bytes and both continue (length byte c) or both end at the same place.  Per level:
//   1. ONE stable LSD pair sort of the unresolved strings by chunk key (the onesweep path; passes whose digit
//      is the same for all keys are not launched) -- the segment id in the key's top bits keeps the segments
//      of the previous level together and in order, so no second sort regroups them (round 2 ran two);
//   2. neighbours that differ in the key start a new segment; a string is resolved when its segment has
//      one member or its chunk ended (length byte < c);
//   3. the unresolved strings are compacted and go to the next level (c bytes further).
// Every pass is stable and level 0 starts from the identity, so equal strings keep input order.
// The chunk's bytes come in through two aligned 8-byte loads and a funnel shift (the blob is padded), not byte by byte.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rss {

constexpr int kChunkBytes = 7;
constexpr int kBlock = 256;

// the 8 bytes at blob + begin, first byte in the TOP byte of the result (the blob is 8-byte aligned and readable 16 bytes past its end)
__device__ __forceinline__ uint64_t load8_be(const uint8_t* blob, uint64_t begin) {
  const uint64_t* p = reinterpret_cast<const uint64_t*>(blob + (begin & ~7ull));
  const uint64_t a = p[0], b = p[1];
  const unsigned sh = static_cast<unsigned>(begin & 7u) * 8u;
  const uint64_t v = sh ? (a >> sh) | (b << (64u - sh)) : a;      // byte i (little-endian) = blob[begin + i]
  return __builtin_bswap64(v);
}

// chunk key of string `id` at byte `depth`: c bytes (1..7), segment `seg` in the top sbits bits
template <typename OFF>
__device__ __forceinline__ uint64_t chunk_key(const uint8_t* blob, const OFF* offsets, uint32_t id, uint64_t depth, unsigned c,
                                              uint32_t seg, unsigned sbits) {
  const uint64_t begin = static_cast<uint64_t>(offsets[id]) + depth;
  const uint64_t end = static_cast<uint64_t>(offsets[id + 1]);
  const uint64_t rem = end > begin ? end - begin : 0;
  uint64_t chunk = 0;
  if (rem) {
    chunk = load8_be(blob, begin) >> (64u - 8u * c);
    if (rem < c) chunk &= ~((1ull << (8u * (c - static_cast<unsigned>(rem)))) - 1ull);     // bytes past the end read as 0
  }
  const uint64_t k = (chunk << 8) | (rem < c ? rem : c);
  return sbits ? k | (static_cast<uint64_t>(seg) << (64u - sbits)) : k;
}

__global__ void iota_kernel(uint32_t* a, uint32_t* b, uint32_t m) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) { a[i] = i; if (b) b[i] = i; }
}

// keys[j] = chunk key of string ids[j] (ids == null: string j) in segment seg[j] (seg == null: one segment)
template <typename OFF>
__global__ void chunk_keys_kernel(const uint8_t* blob, const OFF* offsets, const uint32_t* ids, const uint32_t* seg, uint32_t m,
                                  uint64_t depth, unsigned c, unsigned sbits, uint64_t* keys) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) keys[j] = chunk_key<OFF>(blob, offsets, ids ? ids[j] : j, depth, c, seg ? seg[j] : 0u, sbits);
}

__global__ void gather_u32_kernel(const uint32_t* in, const uint32_t* src, uint32_t m, uint32_t* out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) out[j] = in[src[j]];
}

// perm[pos[j]] = ids[j]: the subset occupies the same (ascending) positions before and after its sort
__global__ void write_back_kernel(const uint32_t* ids, const uint32_t* pos, uint32_t m, uint32_t* perm) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) perm[pos ? pos[j] : j] = ids[j];
}

// (the segment id is part of the key: a differing key is a differing (segment, chunk); c = the level's chunk bytes)
__global__ void heads_open_kernel(const uint64_t* keys, uint32_t m, unsigned c, uint32_t* head, uint32_t* open) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) {
    const uint64_t k = keys[j];
    const bool h = j == 0 || k != keys[j - 1];
    const bool next_h = j + 1 == m || k != keys[j + 1];
    head[j] = h ? 1u : 0u;
    open[j] = ((k & 0xFF) == c && !(h && next_h)) ? 1u : 0u;
  }
}

__global__ void compact_kernel(const uint32_t* open, const uint32_t* open_excl, const uint32_t* head_incl,
                               const uint32_t* ids, const uint32_t* pos, uint32_t m, uint32_t* ids_out,
                               uint32_t* pos_out, uint32_t* seg_out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m && open[j]) {
    uint32_t o = open_excl[j];
    ids_out[o] = ids[j];
    pos_out[o] = pos ? pos[j] : j;
    seg_out[o] = head_incl[j] - 1;
  }
}

__global__ void widen_kernel(const uint32_t* in, uint32_t m, uint64_t* out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) out[j] = in[j];
}

// ---- u32 prefix sums over m elements: per-block sums, one block scans them, blocks add the carry --
constexpr int kScanItems = 16;
constexpr int kScanTile = kBlock * kScanItems;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t& total) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t u = __shfl_up(inc, off, 64);
    if (lane >= static_cast<unsigned>(off)) inc += u;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t before = 0, sum = 0;
  for (unsigned w = 0; w < kBlock / 64; ++w) { if (w < wave) before += lds[w]; sum += lds[w]; }
  total = sum;
  __syncthreads();
  return before + inc - v;
}

__global__ __launch_bounds__(kBlock) void scan_block_sums_kernel(const uint32_t* in, uint32_t m, uint32_t* block_sums) {
  __shared__ uint32_t lds[kBlock / 64];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kScanTile;
  uint32_t s = 0;
  for (int i = 0; i < kScanItems; ++i) {
    uint64_t j = base + static_cast<uint64_t>(i) * kBlock + threadIdx.x;
    if (j < m) s += in[j];
  }
  uint32_t total;
  block_exclusive_scan(s, lds, total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// one block: exclusive scan of the block sums in place; total -> block_sums[blocks]
__global__ __launch_bounds__(kBlock) void scan_spine_kernel(uint32_t* block_sums, uint32_t blocks) {
  __shared__ uint32_t lds[kBlock / 64];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < blocks; base += kBlock) {
    uint32_t j = base + threadIdx.x;
    uint32_t v = j < blocks ? block_sums[j] : 0;
    uint32_t total;
    uint32_t ex = block_exclusive_scan(v, lds, total);
    if (j < blocks) block_sums[j] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) block_sums[blocks] = carry;
}

// out[j] = sum of in[0..j) (+ in[j] when INCLUSIVE); each thread owns kScanItems consecutive elements
template <bool INCLUSIVE>
__global__ __launch_bounds__(kBlock) void scan_apply_kernel(const uint32_t* in, uint32_t m, const uint32_t* block_sums,
                                                            uint32_t* out) {
  __shared__ uint32_t lds[kBlock / 64];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kScanTile + static_cast<uint64_t>(threadIdx.x) * kScanItems;
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) { v[i] = base + i < m ? in[base + i] : 0; s += v[i]; }
  uint32_t total;
  uint32_t run = block_sums[blockIdx.x] + block_exclusive_scan(s, lds, total);
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < m) out[base + i] = INCLUSIVE ? run + v[i] : run;
    run += v[i];
  }
}

}  // namespace rss
