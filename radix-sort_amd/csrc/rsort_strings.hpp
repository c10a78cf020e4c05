// rsort_strings.hpp -- device code of the std::string key path (MSD over fixed-width chunks).
//
// The reference sorts strings MSD, one byte per level, with bucket 256 = "past the end" placed
// first (radix_sort.hpp:107-110, :196-200), and hands a region to std::sort / std::stable_sort
// from depth 3 on (:559-569, :1241-1255).  The result of all that is std::string::operator<
// order (unsigned bytes, a proper prefix first), stable on the projected path.
//
// On the GPU the same order is produced MSD over 7-BYTE chunks:
//   chunk key (u64) = the next 7 bytes, big-endian, zero padded, in bits 63..8
//                     | min(bytes remaining, 7) in bits 7..0
// Zero padding alone would make "a" and "a\0" tie; the length byte breaks that tie the right way
// (shorter first), and it cannot outvote a payload byte because it is the least significant one.
// Two strings tie on a chunk key only if they agree on those 7 bytes and both continue (length
// byte 7) or both end at the same place.  Per level:
//   1. stable LSD pair sort of the unresolved strings by chunk key (the onesweep path, 8 passes),
//      then -- from level 1 on -- a stable sort by segment id, which regroups the segments of the
//      previous level (segmented sort without per-segment launches);
//   2. neighbours that differ in (segment, chunk key) start a new segment; a string is resolved
//      when its segment has one member or its chunk ended (length byte < 7);
//   3. the unresolved strings are compacted and go to the next level (7 bytes further).
// Every pass is stable and level 0 starts from the identity, so equal strings keep input order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rss {

constexpr int kChunkBytes = 7;
constexpr int kBlock = 256;

__device__ __forceinline__ uint64_t chunk_key(const uint8_t* blob, const uint64_t* offsets, uint32_t id, uint32_t level) {
  const uint64_t begin = offsets[id] + static_cast<uint64_t>(level) * kChunkBytes;
  const uint64_t end = offsets[id + 1];
  const uint64_t rem = end > begin ? end - begin : 0;
  uint64_t k = 0;
#pragma unroll
  for (int b = 0; b < kChunkBytes; ++b) k = (k << 8) | (static_cast<uint64_t>(b) < rem ? blob[begin + b] : 0u);
  return (k << 8) | (rem < kChunkBytes ? rem : kChunkBytes);
}

__global__ void iota_kernel(uint32_t* a, uint32_t* b, uint32_t m) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) { a[i] = i; if (b) b[i] = i; }
}

// keys[j] = chunk key of string ids[src ? src[j] : j]
__global__ void chunk_keys_kernel(const uint8_t* blob, const uint64_t* offsets, const uint32_t* ids,
                                  const uint32_t* src, uint32_t m, uint32_t level, uint64_t* keys) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) keys[j] = chunk_key(blob, offsets, ids[src ? src[j] : j], level);
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

// head[j] = 1 if (segment, key) differs from the left neighbour
__global__ void heads_kernel(const uint64_t* keys, const uint32_t* seg, uint32_t m, uint32_t* head) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) head[j] = (j == 0 || keys[j] != keys[j - 1] || (seg && seg[j] != seg[j - 1])) ? 1u : 0u;
}

// open[j] = 1 if the string must go to the next level: chunk full (it may continue) and not alone
__global__ void unresolved_kernel(const uint64_t* keys, const uint32_t* head, uint32_t m, uint32_t* open) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) {
    bool alone = head[j] && (j + 1 == m || head[j + 1]);
    open[j] = ((keys[j] & 0xFF) == kChunkBytes && !alone) ? 1u : 0u;
  }
}

// heads_kernel and unresolved_kernel in one launch: a thread looks at its key, its left and its right neighbour
__global__ void heads_open_kernel(const uint64_t* keys, const uint32_t* seg, uint32_t m, uint32_t* head, uint32_t* open) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) {
    const uint64_t k = keys[j];
    const uint32_t sg = seg ? seg[j] : 0u;
    const bool h = j == 0 || k != keys[j - 1] || (seg && sg != seg[j - 1]);
    const bool next_h = j + 1 == m || k != keys[j + 1] || (seg && sg != seg[j + 1]);
    head[j] = h ? 1u : 0u;
    open[j] = ((k & 0xFF) == kChunkBytes && !(h && next_h)) ? 1u : 0u;
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
