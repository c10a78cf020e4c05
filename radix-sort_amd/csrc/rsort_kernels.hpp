// rsort_kernels.hpp -- gfx950 (CDNA4, wave64) device code of the LSD sort path.
//
// Pipeline of one sort (P = number of 8-bit digit passes):
//   1. digit_histogram_kernel  one read of the keys -> P x 256 global digit counts
//                              (replaces getCountVectorThread, radix_sort.hpp:125-153, for all
//                               passes at once; LDS-privatised per workgroup, wave64 LDS atomics)
//   2. digit_scan_kernel       exclusive prefix of each pass' 256 counts
//                              (replaces getPrefixVector, radix_sort.hpp:193-208)
//   3. onesweep_kernel x P     rank-and-scatter, one read + one write of the keys per pass
//                              (replaces getSortedVector, radix_sort.hpp:447-470 / :1062-1109);
//                              the cross-tile prefix is a single-pass chained scan (decoupled
//                              look-back descriptors, see below)
//
// Look-back on MI355X.  The chip retires ~30 tiles per microsecond while a published word needs
// ~0.5-1 us to become visible to another CU, so a tile's nearest 16-32 predecessors are still
// "aggregate only" when it asks.  The walk fetches LOOKBACK predecessors per dependent round trip
// (independent loads, consumed in order).  Measured (stamps build, uint32, 16384-key tiles): median
// 24 rows walked in 4 round trips of ~1.5 us.  The descriptor reads (coherent, cache-bypassing) are
// the scarce resource, not the latency alone: spreading a digit's walk over 4 threads covered the
// 24 rows in 2 round trips and cut the per-tile look-back from 5.9 to 4.6 us, yet the pass got 60 %
// slower because every tile then read 64 rows.  Fewer, larger tiles are what helps.
//
// Memory-model notes (MI355X: 8 XCDs with private, mutually non-coherent L2s):
//   * look-back descriptors are single 4- or 8-byte words that carry status AND value, written
//     and polled with relaxed agent-scope atomics (sc1 stores / sc1 loads): the word is its own
//     flag, so no fence is needed around it;
//   * tile ids come from an atomic ticket, not blockIdx: a tile's predecessors have all started
//     before it exists, which is the forward-progress argument for the look-back spin (the spin
//     is bounded by a wall-clock timeout anyway);
//   * keys are written with plain stores and only read by the NEXT kernel on the same stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// Diagnostic builds only, never defined in the product:
//   -DRS_STAMPS   wave 0 / lane 0 of every 64th tile stores s_memrealtime (100 MHz) at phase
//                 boundaries into a buffer no other code reads;
#ifdef RS_STAMPS
__device__ unsigned long long rs_stamp_buf[16 * 4096];
#define RS_STAMP(slot)                                                                        \
  do {                                                                                        \
    if (RS_STAMP_ON && tid == 0 && (tile & 63) == 0 && (tile >> 6) < 4096)                    \
      rs_stamp_buf[(slot) * 4096 + (tile >> 6)] = wall_clock64();                             \
  } while (0)
#else
#define RS_STAMP(slot) do {} while (0)
#endif

namespace rsk {

constexpr int kRadixBits = 8;      // radix_sort.hpp:32  SHIFT_BITS
constexpr int kRadix = 256;        // radix_sort.hpp:33  BASE
constexpr int kWave = 64;

template <int BYTES> struct UIntOf;
template <> struct UIntOf<1> { using type = uint8_t;  using stype = int8_t; };
template <> struct UIntOf<2> { using type = uint16_t; using stype = int16_t; };
template <> struct UIntOf<4> { using type = uint32_t; using stype = int32_t; };
template <> struct UIntOf<8> { using type = uint64_t; using stype = int64_t; };

// Key transforms, branch-free.  fmask = all ones for IEEE keys else 0; smask = sign bit for signed
// and IEEE keys else 0.
//   forward:  IEEE  u>>63 ? ~u : u^SIGN  (radix_sort.hpp:405);  signed  u^SIGN  (:150)
//   inverse:  IEEE  u>>63 ? u^SIGN : ~u  (radix_sort.hpp:412)
template <typename K> __device__ __forceinline__ K xf_forward(K k, K fmask, K smask) {
  using S = typename UIntOf<sizeof(K)>::stype;
  K sign_fill = static_cast<K>(static_cast<S>(k) >> (sizeof(K) * 8 - 1));
  return static_cast<K>(k ^ ((sign_fill & fmask) | smask));
}
template <typename K> __device__ __forceinline__ K xf_inverse(K k, K fmask, K smask) {
  using S = typename UIntOf<sizeof(K)>::stype;
  K sign_fill = static_cast<K>(static_cast<S>(k) >> (sizeof(K) * 8 - 1));
  return static_cast<K>(k ^ ((static_cast<K>(~sign_fill) & fmask) | smask));
}

// Digit extraction.  `shift`/`bits` are wave-uniform.  For 8-byte keys the digit normally lies
// inside one 32-bit half (always, when begin_bit is a multiple of 8): one v_bfe_u32 on that half.
template <typename K> __device__ __forceinline__ unsigned digit_at(K k, int shift, int bits) {
  if constexpr (sizeof(K) == 8) {
    if ((shift & 31) + bits <= 32) {
      unsigned w = shift < 32 ? static_cast<unsigned>(k) : static_cast<unsigned>(k >> 32);
      return __builtin_amdgcn_ubfe(w, static_cast<unsigned>(shift & 31), static_cast<unsigned>(bits));
    }
    return static_cast<unsigned>(k >> shift) & ((1u << bits) - 1u);
  } else {
    return __builtin_amdgcn_ubfe(static_cast<unsigned>(k), static_cast<unsigned>(shift), static_cast<unsigned>(bits));
  }
}

__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// ------------------------------------------------------------------------------------------------
// 0. order pre-check (isSortedBi, radix_sort.hpp:269-305; key::isSorted :746-769)
// ------------------------------------------------------------------------------------------------
// flags bit 0: some neighbour pair descends (input is not ascending); bit 1: some pair ascends (not
// descending).  Blocks look at the flags before each chunk and stop once both are set, so the check
// costs microseconds on unordered inputs and one read of the keys on ordered ones -- where it
// replaces the whole sort, as it does in the reference.
struct OrderArgs {
  const void* keys;
  unsigned* flags;
  uint64_t n;
  uint64_t fmask, smask;
  int begin_bit;
  int bits;        // end_bit - begin_bit
};

template <typename K> __device__ __forceinline__ K order_field(K k, const OrderArgs& a) {
  K t = xf_forward(k, static_cast<K>(a.fmask), static_cast<K>(a.smask));
  if (a.bits == static_cast<int>(sizeof(K)) * 8) return t;
  return static_cast<K>((t >> a.begin_bit) & ((static_cast<K>(1) << a.bits) - 1));
}

template <typename K, int BLOCK>
__global__ __launch_bounds__(BLOCK) void order_check_kernel(OrderArgs a) {
  constexpr int VEC = 16 / sizeof(K);          // 16 bytes per lane per load
  constexpr int UNROLL = 4;                    // packs in flight per lane
  constexpr int CHECK_EVERY = 8;               // iterations between two looks at the flags (a coherent load each)
  const K* keys = static_cast<const K*>(a.keys);
  // neighbour pairs (j, j+1), j < n-1.  Packs of VEC keys start at the first 16-byte boundary; a lane checks the pairs
  // inside its pack and the pair that leaves it (one extra key: the first of the next pack, same cache line or the next).
  uint64_t head = ((16u - (reinterpret_cast<uintptr_t>(keys) & 15u)) & 15u) / sizeof(K);
  if (head > a.n) head = a.n;
  const uint64_t npack = (a.n - head) / VEC;
  struct alignas(16) Pack { K v[VEC]; };
  const Pack* packs = reinterpret_cast<const Pack*>(keys + head);
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * BLOCK;
  unsigned seen = __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (seen == 3u) return;                      // unordered input: the one-workgroup probe launch has already said so
  unsigned f = 0;
  auto pair = [&](K x, K y) { f |= (x > y ? 1u : 0u) | (x < y ? 2u : 0u); };
  auto flush = [&]() {
    // set a bit only if it was not seen set: the one flag word must not take an atomic per wave and iteration
    if (!(seen & 1u) && __ballot(f & 1u) && (threadIdx.x & 63) == 0) atomicOr(a.flags, 1u);
    if (!(seen & 2u) && __ballot(f & 2u) && (threadIdx.x & 63) == 0) atomicOr(a.flags, 2u);
  };
  if (blockIdx.x == 0) {                        // head and tail pairs, one per thread (fewer than 3 * VEC of them)
    const uint64_t tail0 = head + npack * VEC;  // first key after the packs
    uint64_t j = threadIdx.x;
    if (j < head && j + 1 < a.n) pair(order_field(keys[j], a), order_field(keys[j + 1], a));
    j = (tail0 > 0 ? tail0 - 1 : 0) + threadIdx.x;            // the pair leaving the last pack, then the tail
    if (j + 1 < a.n && j >= head) pair(order_field(keys[j], a), order_field(keys[j + 1], a));
  }
  int since = 0;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < npack; i += UNROLL * stride) {
    if (++since == CHECK_EVERY) {
      since = 0;
      flush();
      seen = __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen == 3u) return;
    }
    Pack pk[UNROLL];
    K nxt[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint64_t p = i + u * stride;
      if (p < npack) {
        pk[u] = packs[p];
        nxt[u] = p + 1 < npack ? packs[p + 1].v[0] : pk[u].v[VEC - 1];   // the last pack's outgoing pair belongs to block 0
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (i + u * stride < npack) {
        K prev = order_field(pk[u].v[0], a);
#pragma unroll
        for (int e = 1; e < VEC; ++e) { K cur = order_field(pk[u].v[e], a); pair(prev, cur); prev = cur; }
        pair(prev, order_field(nxt[u], a));
      }
    }
  }
  flush();
}

// in-place reversal (std::reverse, radix_sort.hpp:300)
template <typename K>
__global__ void reverse_kernel(K* keys, uint64_t n) {
  uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (; i < n / 2; i += stride) {
    K x = keys[i], y = keys[n - 1 - i];
    keys[i] = y;
    keys[n - 1 - i] = x;
  }
}

// ------------------------------------------------------------------------------------------------
// 1. all-pass digit histogram
// ------------------------------------------------------------------------------------------------
struct HistArgs {
  const void* keys;
  unsigned long long* ghist;   // [passes][256]
  uint64_t n;
  uint64_t fmask, smask;
  int begin_bit;
  int end_bit;
  int passes;
  // chain mode (see "XCD chains" at the sweep kernel): workgroup b counts inside chain (b & 7)'s key range
  // [chain * chain_keys, (chain + 1) * chain_keys) and adds to chain_hist[chain][pass][digit] (rows of kChainHistStride);
  // chain_keys == 0: one range, counts go to ghist.  Needs 16-byte aligned keys and chain_keys % (16 / sizeof(K)) == 0.
  unsigned long long* chain_hist;
  uint64_t chain_keys;
  // Mid-size sorts (rsort_api.hip, sort_device): the kernel also clears up to two regions of the control block, 16 bytes
  // per store, spread over the grid -- what a memset launched before it would have cleared (nothing in this kernel reads them;
  // the sweeps that do come after it on the stream).  zero_vecs counts 16-byte units; 0 = nothing to clear.
  void* zero[2];
  unsigned zero_vecs[2];
};
constexpr unsigned kChainHistStride = 8 * 256;   // words between two chains' tables in chain_hist

// LDS layout of the per-workgroup histogram: [pass][digit][kHistCols] 32-bit counters, a lane adds
// to column (lane & 15).  The 32 lanes the LDS serves together then touch 16 different banks with
// at most two lanes per bank whatever their digits are, which an LDS add absorbs at its issue
// rate (MI355X_MICROARCH.md, LDS: 2-way costs nothing extra on a 4-cycle store-class instruction).
// A single [pass][digit] table took ~11 cycles per wave instruction on random digits (bank
// conflicts) and serialised 64 ways on sorted / few-valued inputs; this layout needs no special
// case for either.  16 KB per pass: 64 KB for 4-byte keys (two workgroups per CU), 128 KB for 8-byte.
#ifndef RS_HIST_COLS
#define RS_HIST_COLS 16
#endif
constexpr int kHistCols = RS_HIST_COLS;

// WHOLE = the sort covers the whole key in byte-aligned digits (begin_bit 0, end_bit 8*sizeof(K)):
// every shift is a compile-time constant (one v_bfe_u32 with immediates per digit).
template <typename K, int MAX_PASSES, bool WHOLE>
__device__ __forceinline__ void hist_add_key(K k, unsigned* col, const HistArgs& a) {
#ifdef RS_HIST_TIMING_ONE_DIGIT    // timing experiment only (wrong results): what would a histogram of the first digit alone cost?
  constexpr int COUNTED = 1;
#else
  constexpr int COUNTED = MAX_PASSES;
#endif
#pragma unroll
  for (int p = 0; p < COUNTED; ++p) {
    if constexpr (WHOLE) {
      unsigned w = sizeof(K) == 8 && p >= 4 ? static_cast<unsigned>(static_cast<uint64_t>(k) >> 32) : static_cast<unsigned>(k);
      atomicAdd(col + (p * kRadix + ((w >> ((p & 3) * 8)) & 0xFFu)) * kHistCols, 1u);
    } else if (p < a.passes) {
      int shift = a.begin_bit + p * kRadixBits;
      int bits = a.end_bit - shift < kRadixBits ? a.end_bit - shift : kRadixBits;
      atomicAdd(col + (p * kRadix + digit_at(k, shift, bits)) * kHistCols, 1u);
    }
  }
}

template <typename K, int BLOCK, bool WHOLE>
__global__ __launch_bounds__(BLOCK) void digit_histogram_kernel(HistArgs a) {
  constexpr int MAX_PASSES = sizeof(K);
  constexpr int VEC = 16 / sizeof(K);            // 16 bytes per lane per load
#ifndef RS_HIST_UNROLL
#define RS_HIST_UNROLL 4
#endif
  constexpr int UNROLL = RS_HIST_UNROLL;         // loads in flight per lane
  __shared__ unsigned lh[MAX_PASSES * kRadix * kHistCols];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    uint4* z = static_cast<uint4*>(a.zero[r]);
    for (unsigned i = blockIdx.x * BLOCK + threadIdx.x; i < a.zero_vecs[r]; i += gridDim.x * BLOCK) z[i] = uint4{0u, 0u, 0u, 0u};
  }
  for (int i = threadIdx.x; i < MAX_PASSES * kRadix * kHistCols; i += BLOCK) lh[i] = 0;
  __syncthreads();
  unsigned* col = lh + (threadIdx.x & (kHistCols - 1));

  const K* keys = static_cast<const K*>(a.keys);
  const K fm = static_cast<K>(a.fmask), sm = static_cast<K>(a.smask);
  // the key range [lo, hi) this workgroup counts in, as worker `wb` of `nwb`
  const bool chained = a.chain_keys != 0;
  const unsigned chain = chained ? blockIdx.x & 7u : 0u;
  const uint64_t wb = chained ? blockIdx.x >> 3 : blockIdx.x, nwb = chained ? gridDim.x >> 3 : gridDim.x;
  uint64_t lo = 0, hi = a.n;
  if (chained) {
    lo = chain * a.chain_keys < a.n ? chain * a.chain_keys : a.n;
    hi = lo + a.chain_keys < a.n ? lo + a.chain_keys : a.n;
  }
  // head: keys before the first 16-byte boundary (the multi-GPU splitter passes sub-range pointers that are only
  // key-aligned; never in chain mode); tail: what is left after the last whole 16-byte pack
  uint64_t head = chained ? 0 : ((16u - (reinterpret_cast<uintptr_t>(keys) & 15u)) & 15u) / sizeof(K);
  if (head > a.n) head = a.n;
  const uint64_t vlo = chained ? lo / VEC : 0, nvec = chained ? hi / VEC : (a.n - head) / VEC;   // packs [vlo, nvec)
  struct alignas(16) Pack { K v[VEC]; };
  const Pack* vkeys = reinterpret_cast<const Pack*>(keys + head);
  const uint64_t stride = nwb * BLOCK;
  uint64_t i = vlo + wb * BLOCK + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
    Pack pk[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) pk[u] = vkeys[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int j = 0; j < VEC; ++j) hist_add_key<K, MAX_PASSES, WHOLE>(xf_forward(pk[u].v[j], fm, sm), col, a);
  }
  for (; i < nvec; i += stride) {
    Pack pk = vkeys[i];
#pragma unroll
    for (int j = 0; j < VEC; ++j) hist_add_key<K, MAX_PASSES, WHOLE>(xf_forward(pk.v[j], fm, sm), col, a);
  }
  if (wb == 0 && (!chained || (hi == a.n && lo < hi))) {   // fewer than 2 * VEC keys: head, and the tail after the last pack
    if (threadIdx.x < head) hist_add_key<K, MAX_PASSES, WHOLE>(xf_forward(keys[threadIdx.x], fm, sm), col, a);
    uint64_t t = head + nvec * VEC + threadIdx.x;
    if (t < a.n) hist_add_key<K, MAX_PASSES, WHOLE>(xf_forward(keys[t], fm, sm), col, a);
  }
  __syncthreads();
  unsigned long long* dst = chained ? a.chain_hist + static_cast<size_t>(chain) * kChainHistStride : a.ghist;
  for (int b = threadIdx.x; b < a.passes * kRadix; b += BLOCK) {
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < kHistCols; ++j) c += lh[b * kHistCols + ((j + b) & (kHistCols - 1))];
    if (c) atomicAdd(&dst[b], static_cast<unsigned long long>(c));
  }
}

// Clears `vecs` 16-byte units at p.  What hipMemsetAsync does, as a kernel of this library: sorts that are being captured
// into a HIP graph use it, because memset NODES did not reliably run again when such a graph was replayed (rsort_api.hip,
// clear_async).
__global__ __launch_bounds__(256) void zero_kernel(uint4* p, unsigned vecs) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < vecs; i += gridDim.x * 256u) p[i] = uint4{0u, 0u, 0u, 0u};
}

// ------------------------------------------------------------------------------------------------
// 2. exclusive scan of each pass' digit counts (grid = passes, block = 256)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_inclusive_scan(unsigned long long v) {
  unsigned lane = lane_id();
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    unsigned long long t = __shfl_up(v, off, kWave);
    if (lane >= static_cast<unsigned>(off)) v += t;
  }
  return v;
}

// chain_hist != null: the counts arrive per chain (HistArgs::chain_hist); their sums are written to ghist first
__global__ __launch_bounds__(kRadix) void digit_scan_kernel(unsigned long long* ghist, unsigned long long* gbase,
                                                           const unsigned long long* chain_hist) {
  __shared__ unsigned long long wsum[kRadix / kWave];
  const int p = blockIdx.x, d = threadIdx.x;
  unsigned long long c;
  if (chain_hist) {
    c = 0;
    for (unsigned x = 0; x < 8; ++x) c += chain_hist[x * kChainHistStride + p * kRadix + d];
    ghist[p * kRadix + d] = c;
  } else {
    c = ghist[p * kRadix + d];
  }
  unsigned long long inc = wave_inclusive_scan(c);
  if ((d & (kWave - 1)) == kWave - 1) wsum[d / kWave] = inc;
  __syncthreads();
  unsigned long long before = 0;
  for (int w = 0; w < d / kWave; ++w) before += wsum[w];
  gbase[p * kRadix + d] = before + inc - c;
}


// ------------------------------------------------------------------------------------------------
// 3. onesweep rank-and-scatter pass
// ------------------------------------------------------------------------------------------------
// Descriptor word: status in the top two bits, value below.
//   0                 not written yet in this sort
//   1  (AGG)          value = this tile's own count of the digit
//   2 or 3 (INCL)     value = count of the digit in tiles 0..this, for passes of even/odd parity:
//                     pass p accepts only code 2 + (p & 1), so the words a finished pass leaves
//                     behind (all INCL of the other parity) read as "not written" to the next
//                     pass and one memset per sort suffices.
template <typename D> struct Desc;
template <> struct Desc<uint32_t> {
  static constexpr int kShift = 30;
  static constexpr uint32_t kValueMask = (1u << 30) - 1u;   // counts up to 2^30 - 1: n <= 2^30 for one chain over all tiles; n < 2^32 in chain
                                                            // mode, where a word counts inside one of eight chains (see host dispatch)
};
template <> struct Desc<uint64_t> {
  static constexpr int kShift = 62;
  static constexpr uint64_t kValueMask = (1ull << 62) - 1ull;
};
constexpr unsigned kCodeAgg = 1;

// Tile order.  MI355X has 8 XCDs with private L2s, and workgroups are dealt to them round-robin.  The runs two
// consecutive tiles write for one digit are adjacent in memory and almost never end on a 128-byte line, so the line
// at every run boundary is written half by one tile and half by the next.  If those two tiles run on different XCDs,
// each L2 holds a partial line and both write it back separately; tools/scatter_roofline.hip measures what that
// costs without any sorting work: 256 streams x 256-byte runs, runs shifted by 64 bytes: 3.99 TB/s; the same with
// the tiles dealt to the XCDs in groups of 8 consecutive tiles: 4.83 TB/s (aligned runs: 5.17 TB/s).
// Tiles are therefore drawn from one ticket counter PER XCD: the k-th ticket of XCD x is tile
//     (k / kXcdGroup) * 8 * kXcdGroup + x * kXcdGroup + k % kXcdGroup,
// i.e. every XCD takes groups of kXcdGroup consecutive tiles and the partial lines inside a group meet in one L2.
// A workgroup whose own XCD has run out draws from the others' counters.  Each counter hands out its tiles in
// increasing order and every tile that has been drawn is running, so the lowest unfinished tile is either running or
// next on an XCD whose earlier tiles are all finished, i.e. one with free workgroup slots (forward progress of the
// look-back; the wait is bounded by a wall-clock timeout in any case).
constexpr unsigned kXcds = 8;
#ifndef RS_XCD_GROUP
#define RS_XCD_GROUP 8
#endif
constexpr unsigned kXcdGroup = RS_XCD_GROUP;
constexpr unsigned kTicketWords = kXcds * 32;      // one counter per XCD, 128 bytes apart
constexpr unsigned kNoTile = 0xFFFFFFFFu;

__device__ __forceinline__ unsigned xcc_id() {
  // s_getreg_b32 HW_REG_XCC_ID (id 20 on gfx94x/gfx950), field XCC_ID = bits [3:0]
  return static_cast<unsigned>(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20)) & (kXcds - 1);
}

// Probe for the tile order: every workgroup counts itself under the XCC id it reads.  The host enables the per-XCD
// order only if all kXcds ids show up with similar counts (a partitioned device or a different register layout would
// put everything on one id, and the per-XCD order would then starve the look-back).
__global__ __launch_bounds__(64) void xcc_probe_kernel(unsigned* counts) {
  if (threadIdx.x == 0) atomicAdd(&counts[xcc_id()], 1u);
}

__device__ __forceinline__ unsigned draw_tile(unsigned* tickets, unsigned tiles, bool per_xcd) {
  if (!per_xcd) {                                 // one counter: tile = ticket
    const unsigned k = atomicAdd(&tickets[0], 1u);
    return k < tiles ? k : kNoTile;
  }
  const unsigned home = xcc_id();
  for (unsigned step = 0; step < kXcds; ++step) {
    const unsigned x = (home + step) & (kXcds - 1);
    const unsigned k = atomicAdd(&tickets[x * 32], 1u);
    const unsigned long long t = static_cast<unsigned long long>(k / kXcdGroup) * (kXcds * kXcdGroup) + x * kXcdGroup + k % kXcdGroup;
    if (t < tiles) return static_cast<unsigned>(t);      // a counter's tiles increase with k: the first miss means it is exhausted
  }
  return kNoTile;
}

struct SweepArgs {
  const void* keys_in;
  void* keys_out;
  const void* vals_in;
  void* vals_out;
  void* desc;                       // [tiles][256] words of D
  unsigned* ticket;                 // kTicketWords counters for this pass (one per XCD), zero at launch
  const unsigned long long* gbase;  // [256] for this pass: where each digit's run starts in the output
  const unsigned long long* counts; // non-null (mid-size sorts, 4-byte descriptors, not chain mode): this pass' 256 digit counts;
                                    // every tile scans them itself and gbase is not read -- one launch less per sort
  unsigned* error_flag;             // set non-zero on a look-back timeout
  uint64_t n;
  uint64_t fmask_in, smask_in;      // non-zero only for the first pass
  uint64_t fmask_out, smask_out;    // non-zero only for the last pass
  int shift;
  int bits;
  unsigned incl_code;               // 2 + (pass & 1)
  int per_xcd_order;                // tiles from one ticket counter per XCD (rs_ctx_create probes whether that is safe)
  // chain mode only (CHAIN kernels, see "XCD chains" below)
  const unsigned long long* chain_cnt;   // [kXcds] rows of 256: this pass' digit counts over each chain's input range
  unsigned chain_cnt_stride;             // words between two rows of chain_cnt
  unsigned chain_tiles;                  // tiles per chain (the last chain may hold fewer)
  unsigned long long* next_cnt;          // [kXcds][256] counts of the NEXT launched pass' digit over each chain's range of this
                                         // pass' output (accumulated here); null in the last pass
  int next_shift, next_bits;             // the next launched pass' digit
  unsigned chunk_shift, chunk_mul;       // chain of output position o = (mul_u24(o >> chunk_shift, chunk_mul)) >> 16
  int desc_local;                        // descriptors published with workgroup-scope stores (chain_probe_kernel found them visible to the other
                                         // workgroups of the XCD); 0: agent-scope stores, as the one-chain kernels use
  // Mid-size keys-only sorts (rsort_api.hip): the host queues every kernel of BOTH forms without looking at a count, and
  // each kernel decides for itself, from the top digit's 256 counts, which form is running -- the same answer in all of
  // them.  When every group of equal top digit fits the group kernel's LDS window (rsk::group_sort_mid_kernel), the sort is
  // one sweep on the top digit plus that kernel; otherwise it is the LSD loop as before.
  const unsigned long long* mid_counts;  // non-null: this launch has a role in such a sort; the top digit's counts
  unsigned mid_cap;                      // the window: groups of at most this many keys
  int mid_role;                          // 1 = a lower digit's pass: leaves at once when the groups fit;
                                         // 2 = the top digit's pass: last LSD pass as its other arguments say when they do not fit, FIRST
                                         //     pass of the short form when they do, with the operands below and no inverse transform
  const void* mid_keys_in;
  void* mid_keys_out;
  uint64_t mid_fmask_in, mid_smask_in;
};

// Lanes of the wave whose digit DIFFERS from this lane's, as two 32-bit halves: per bit, the
// ballot XOR the lane's own bit spread over the word (0 / ~0), ORed in (one v_bitop3 per half).
__device__ __forceinline__ void differ_mask(unsigned d, unsigned& lo, unsigned& hi) {
  lo = 0; hi = 0;
#pragma unroll
  for (int b = 0; b < kRadixBits; ++b) {
    int sb = static_cast<int>(d << (31 - b)) >> 31;        // v_bfe_i32: 0 or -1
    unsigned long long v = __builtin_amdgcn_sicmp(sb, 0, 40 /* ICMP_SLT */);
    lo |= static_cast<unsigned>(v) ^ static_cast<unsigned>(sb);
    hi |= static_cast<unsigned>(v >> 32) ^ static_cast<unsigned>(sb);
  }
}

template <typename D> __device__ __forceinline__ D desc_load(const D* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// LOCAL: every reader runs on the writer's XCD (chain mode) -- a store that stays in that XCD's L2 (no write-through)
template <bool LOCAL = false, typename D> __device__ __forceinline__ void desc_store(D* p, D v) {
  if constexpr (LOCAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool C, typename A, typename B> struct Pick { using type = A; };
template <typename A, typename B> struct Pick<false, A, B> { using type = B; };

// Probe for what chain mode assumes of the memory system: a descriptor published with a WORKGROUP-scope store (desc_store<true>:
// the line stays in the XCD's L2, no write-through) is seen by an agent-scope load (desc_load) of ANOTHER workgroup on the same
// XCD.  True on gfx950 (one L2 per XCD, the vector L1 writes through to it), but nothing in the HSA memory model promises it.
// One-wave workgroups take a ticket from their XCD's counter; ticket k waits for word k - 1 of its XCD's chain (bounded:
// ~200 us of wall clock) and then publishes word k with exactly the pair the sweep uses.  A chain that does not get
// through -- any wait that expires -- is counted in failures, and the context then publishes with agent-scope stores.
__global__ __launch_bounds__(64) void chain_probe_kernel(unsigned* tickets /* [kXcds * 32] */, uint32_t* words /* [kXcds][links] */,
                                                         unsigned links, unsigned* failures) {
  if (threadIdx.x != 0) return;
  const unsigned x = xcc_id();
  const unsigned k = atomicAdd(&tickets[x * 32], 1u);
  if (k >= links) return;
  uint32_t* chain = words + static_cast<size_t>(x) * links;
  if (k > 0) {
    const long long t0 = static_cast<long long>(wall_clock64());
    while (desc_load(&chain[k - 1]) != (0x40000000u | (k - 1))) {
      __builtin_amdgcn_s_sleep(1);
      if (static_cast<long long>(wall_clock64()) - t0 > 20000ll) { atomicAdd(failures, 1u); break; }      // 100 MHz ticks
    }
  }
  desc_store<true>(&chain[k], 0x40000000u | k);       // (published even after a failed wait: the workgroups behind must not all time out in turn)
}

// Bounded wait helper: call once per unsuccessful poll; true = give up (timeout, or another
// workgroup already reported a fault).  ~2 s of wall clock (s_memrealtime ticks at 100 MHz).
struct SpinGuard {
  unsigned spins = 0;
  long long t0 = 0;
  __device__ __forceinline__ bool expired(unsigned* error_flag) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 255u) != 0) return false;
    long long now = static_cast<long long>(wall_clock64());
    if (t0 == 0) t0 = now;
    return now - t0 > 200000000ll ||
           __hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  }
};

// ---- ranking by returning LDS adds: the forms every sweep kernel, the single-workgroup kernel AND the probe use ----
// cells: the wave's 256 digit counters.  Two layouts:
//   WIDE = false  16-bit cells, two to a 32-bit word (cell d = half (d & 1) of word d >> 1): 512 bytes per wave, what lets
//                 two 16384-key workgroups of 16 waves share a CU;
//   WIDE = true   one 32-bit word per digit (1 KB per wave; the 512-thread chain kernels, which have the LDS for it): the
//                 address is one shift-add, the add is the constant 1 and the answer needs no extraction -- 2 instead of 7
//                 vector instructions per key around the LDS operation.
// cell_add: one returning add per lane; lanes of one wave instruction that hit the same word (the same digit, or -- 16-bit
// cells -- the two digits that share it) must be served in lane order for the ranks to be stable: what
// lds_order_probe_kernel checks, for both layouts.
template <bool WIDE> __device__ __forceinline__ unsigned cell_add(unsigned* cells, unsigned d, unsigned amount) {
  if constexpr (WIDE) return atomicAdd(&cells[d], amount);
  else return (atomicAdd(&cells[d >> 1], amount << ((d & 1u) * 16)) >> ((d & 1u) * 16)) & 0xFFFFu;
}
template <bool WIDE> __device__ __forceinline__ unsigned cell_load(const unsigned* cells, unsigned d) {
  if constexpr (WIDE) return __hip_atomic_load(&cells[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else return (__hip_atomic_load(&cells[d >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> ((d & 1u) * 16)) & 0xFFFFu;
}
template <bool WIDE = false> __device__ __forceinline__ unsigned rank_add(unsigned* cells, unsigned d) {
  return cell_add<WIDE>(cells, d, 1u);
}
// rank_add_peeled: when at least 16 active lanes hold the first active lane's digit, that lane adds their number once
// and the others take its answer plus their position among them; the remaining lanes add one each as above.
// Call with the active lanes only (EXEC = the lanes that hold a key).
template <bool WIDE = false> __device__ __forceinline__ unsigned rank_add_peeled(unsigned* cells, unsigned d) {
  const unsigned d0 = __builtin_amdgcn_readfirstlane(d);
  const bool mine = d == d0;
  const unsigned long long same = __ballot(mine);
  if (__popcll(same) >= 16) {
    const unsigned below = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(same >> 32),
                                                     __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(same), 0u));
    unsigned ret = 0;
    if (!mine || below == 0) ret = cell_add<WIDE>(cells, d, mine ? static_cast<unsigned>(__popcll(same)) : 1u);
    const unsigned base0 = __builtin_amdgcn_readfirstlane(ret);   // the first active lane holds d0 and is its leader
    return mine ? base0 + below : ret;
  }
  return rank_add<WIDE>(cells, d);
}

// rank_add_runs: for rounds whose lanes hold few, CONTIGUOUS runs of equal digits (ordered inputs whose neighbours share a
// digit: a float ramp repeats every value 16-128 times, so a round holds 2-4 runs of adjacent digits, two of them to a
// word, and only one run can be peeled).  The first lane of every run adds the run's length; the others take what it
// got plus their distance from it.  Two runs of one digit in a round (A B A) are two adds on one cell in one instruction,
// served in lane order like any other pair.  Correct for any digits -- with 64 runs it is rank_add with more work.
// Call with all 64 lanes active (full tiles; the probe's run-head rounds).
// The digit of the lane below across the whole wave (DPP wave_shr:1, one instruction, no LDS); lane 0 reads ~d, i.e. a head.
__device__ __forceinline__ unsigned digit_of_lane_below(unsigned d) {
  return static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(~d), static_cast<int>(d), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}
// N successive rounds at once, each phase over all of them in straight-line code: the head masks (vector and scalar
// arithmetic only), then the N returning adds of the heads, then the N broadcasts of the heads' answers -- N LDS
// operations in flight per phase instead of three dependent LDS round trips per round.  All 64 lanes must be active.
template <bool WIDE, int N>
__device__ __forceinline__ void rank_add_runs_n(unsigned* cells, const unsigned (&d)[N], unsigned (&r)[N]) {
  const unsigned lane = lane_id();
  unsigned head_lane[N], got[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const unsigned long long heads = __ballot(digit_of_lane_below(d[j]) != d[j]);      // lane 0 always
    const unsigned long long upto = heads & (~0ull >> (63u - lane));                   // heads at or below this lane: never empty
    head_lane[j] = 63u - static_cast<unsigned>(__clzll(static_cast<long long>(upto)));
    const unsigned long long above = lane == 63u ? 0ull : heads >> (lane + 1u);
    const unsigned run_end = above ? lane + static_cast<unsigned>(__ffsll(static_cast<long long>(above))) : 64u;   // next head, or one past the wave
    got[j] = run_end - lane;                                                           // the run's length, where this lane is a head
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const unsigned len = got[j];
    got[j] = 0;
    if (head_lane[j] == lane) got[j] = cell_add<WIDE>(cells, d[j], len);
  }
#pragma unroll
  for (int j = 0; j < N; ++j)
    r[j] = static_cast<unsigned>(__shfl(static_cast<int>(got[j]), static_cast<int>(head_lane[j]), kWave)) + (lane - head_lane[j]);
}
template <bool WIDE = false> __device__ __forceinline__ unsigned rank_add_runs(unsigned* cells, unsigned d) {
  const unsigned dd[1] = {d};
  unsigned rr[1];
  rank_add_runs_n<WIDE, 1>(cells, dd, rr);
  return rr[0];
}

// Probe for the ranking shortcut, on exactly the forms the kernels run: both cell layouts (WIDE = false in 1024-thread
// workgroups, two per CU, as the one-chain kernels; WIDE = true in 512-thread workgroups as the chain kernels), one
// histogram per wave, rank_add, rank_add_peeled and rank_add_runs as the kernels call them (the first two also with EXEC
// restricted to the active lanes).  Every lane checks the rank it got against the ballot form's answer: (cell before the
// instruction) + (lower active lanes with the same digit).  Digit patterns: uniform, four values, one value (the peeled
// leader adds the population), neighbouring digits 2k / 2k+1 (different cells of ONE word when cells are 16 bits), runs
// of four lanes, a 37-value mix, and a quarter of the lanes inactive.  512 workgroups x 1024 threads x 192 rounds = 1.0e8
// lane-operations for the packed layout and half as many for the wide one, about 100 us each.
template <bool WIDE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void lds_order_probe_kernel(unsigned* violations, int rounds) {
  constexpr int kCellWords = WIDE ? kRadix : kRadix / 2;
  __shared__ unsigned hist[BLOCK / kWave][kCellWords];
  const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  unsigned* cells = hist[wave];
  for (unsigned i = lane; i < kCellWords; i += kWave) cells[i] = 0;
  __syncthreads();
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    unsigned long long h = (static_cast<unsigned long long>(blockIdx.x) << 40) ^ (static_cast<unsigned long long>(r) << 16) ^ threadIdx.x;
    h += 0x9E3779B97F4A7C15ull; h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull; h = (h ^ (h >> 27)) * 0x94D049BB133111EBull; h ^= h >> 31;
    const int mode = r % 8;
    const unsigned d = mode == 0 ? (h & 255u) : mode == 1 ? (h & 3u) : mode == 2 ? 7u : mode == 3 ? 2u * ((r >> 3) & 127u) + (h & 1u)
                     : mode == 4 ? (lane >> 2) + (h & 1u) : mode == 5 ? static_cast<unsigned>((h >> 8) % 37u) : (h >> 12) & 255u;
    const bool active = mode < 6 || ((h >> 20) & 3u) != 0;
    // the ballot form's answer
    const unsigned before = cell_load<WIDE>(cells, d);
    unsigned dlo, dhi;
    differ_mask(d, dlo, dhi);
    const unsigned long long alive = __ballot(active);
    const unsigned below = __builtin_amdgcn_mbcnt_hi(~dhi & static_cast<unsigned>(alive >> 32),
                                                     __builtin_amdgcn_mbcnt_lo(~dlo & static_cast<unsigned>(alive), 0u));
    __builtin_amdgcn_wave_barrier();
    unsigned got = 0;
    if (active) {
      if (mode < 6 && (r & 16)) got = rank_add_runs<WIDE>(cells, d);          // (needs a prefix of lanes: not with the random inactive lanes)
      else got = (mode == 2 || mode == 7 || (r & 8)) ? rank_add_peeled<WIDE>(cells, d) : rank_add<WIDE>(cells, d);
    }
    __builtin_amdgcn_wave_barrier();
    if (active && got != before + below) ++bad;
    if ((r & 63) == 63) {                      // keep 16-bit cells far from overflow
      for (unsigned i = lane; i < kCellWords; i += kWave) cells[i] = 0;
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (bad) atomicAdd(violations, bad);
}

struct NoVal {};

// test hook (rs_debug_inject_fault): what a faulting sweep does to the error word
__global__ void inject_fault_kernel(unsigned* error_flag, unsigned code) {
  __hip_atomic_fetch_or(error_flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Staging-area swizzle.  A key staged at tile-local sorted position p lives at element p ^ ((p >> 6) & 31).  On
// ordered inputs every digit's run in a tile has (nearly) the same power-of-two length, so the lanes of one staging
// store -- same rank, different digits -- would otherwise all fall on one LDS bank (measured: 4.4x the bank-conflict
// cycles of random input, staging 5.3 us instead of 1.4).  The XOR only permutes positions inside aligned groups of
// 32, so the write-out, which reads aligned runs of consecutive positions, stays conflict-free.
__device__ __forceinline__ unsigned stage_swizzle(unsigned p) { return p ^ ((p >> 6) & 31u); }

// -DRS_CHECKED: every key store of every sweep kernel is range-checked and a store that would leave the array sets the
// error word instead (RS_ERR_KERNEL).  For trying out new ranking / staging code on the GPU: a wrong rank then shows up as
// an error status, not as a wild store and a GPU fault.  Costs about 2 % of a uint32 pass; off in the shipped build, whose
// per-run check (below) covers everything but bugs in the kernel's own ranking.
#ifdef RS_CHECKED
constexpr bool kCheckedStores = true;
#else
constexpr bool kCheckedStores = false;
#endif

template <typename K, typename V, typename D, int BLOCK, int ITEMS, bool WIDE_CELLS>
struct SweepSmem {
  using Cell = typename Pick<WIDE_CELLS, unsigned, unsigned short>::type;
  static constexpr int WAVES = BLOCK / kWave;
  static constexpr int TILE = BLOCK * ITEMS;
  static constexpr bool HAS_VAL = !__is_same(V, NoVal);
  static constexpr size_t kElem = HAS_VAL ? (sizeof(V) > sizeof(K) ? sizeof(V) : sizeof(K)) : sizeof(K);
  alignas(16) unsigned char stage[TILE * kElem];
  // per wave: running digit counts, later tile_start + wave prefix.  16-bit cells (a wave holds at most 64 x ITEMS
  // keys, a tile fewer than 65536), updated two to a 32-bit LDS word: cell d lives in half (d & 1) of word d >> 1;
  // or one word per digit (WIDE_CELLS, see cell_add).
  Cell whist[WAVES][kRadix];
  static_assert(TILE < 65536, "16-bit tile-local positions");
  unsigned tile_start[kRadix];      // exclusive scan of the tile's digit counts
  D gofs[kRadix];                   // global offset of the digit's run minus tile_start
  unsigned wave_total[kRadix / kWave];
  unsigned gwave_total[kRadix / kWave];   // the same for the scan of the whole array's digit counts (SweepArgs::counts)
  unsigned tile_id;
  unsigned out_of_range;            // some digit's run of this tile would end past the array (see the write-out)
};

// XCD chains (CHAIN kernels).  The tiles of a pass are cut into kXcds contiguous ranges ("chains") and chain x is swept
// by the workgroups that run on XCD x, front to back, each with its own ticket counter and its own chained scan:
//   * a tile only ever waits for tiles of its own chain, all of which run on the same XCD, so the look-back
//     descriptors never leave that XCD's L2: they are published with plain stores (the line stays in L2) and polled
//     with L1-bypassing loads that L2 serves -- a few hundred cycles per round trip instead of the 1.5 us of a
//     write-through store plus a fabric read under load, and walks of a few rows instead of 24, because only an
//     eighth of the chip's tiles are queued on each chain;
//   * the runs that consecutive tiles write for one digit are adjacent; with all of a chain on one XCD every
//     partial line at a run boundary is completed in one L2 before it is written back;
//   * where chain x's output starts inside each digit's run needs this pass' digit counts over every chain's
//     input range.  For the first launched pass they come from the histogram kernel (each of its workgroups
//     counts inside one chain's range); for later passes the previous sweep counts them: in its write-out every
//     key adds to an LDS table (chain of its output position, next digit), summed in registers over the
//     workgroup's tiles and added to global memory once at the end.
// Workgroups are persistent (2 per CU) and take the chain of the XCD they find themselves on (XCC_ID), so the
// placement the dispatcher chooses does not matter for correctness as long as every XCD gets a workgroup, which
// rs_ctx_create probes.
template <typename D, bool COUNT>
struct ChainSmem {
  // 16-bit cells [chain >> 1][next digit][chain & 1], two to a word: cell counts of ONE tile (< 65536)
  unsigned ccnt[COUNT ? kXcds / 2 * kRadix : 1];
  D cbase[kRadix];                  // where this chain's output starts inside each digit's run
};
struct NoChainSmem {};

template <typename K, typename V, typename D, int BLOCK, int ITEMS, int LOOKBACK = 8, int MIN_BLOCKS = 1, bool RANK_BY_LDS_ATOMIC = true, bool CHAIN = false, bool COUNT = false>
__global__ __launch_bounds__(BLOCK, (MIN_BLOCKS * BLOCK + 255) / 256) void onesweep_kernel(SweepArgs a_in) {
  SweepArgs a = a_in;
  if constexpr (!CHAIN && sizeof(D) == 4) {
    if (a.mid_counts != nullptr) {               // (kernel-uniform) see SweepArgs
      const bool over = threadIdx.x < kRadix && a.mid_counts[threadIdx.x] > a.mid_cap;
      if (!__syncthreads_or(over)) {             // every group fits: the short form runs
        if (a.mid_role == 1) return;
        a.keys_in = a.mid_keys_in; a.keys_out = a.mid_keys_out;
        a.fmask_in = a.mid_fmask_in; a.smask_in = a.mid_smask_in;
        a.fmask_out = 0; a.smask_out = 0;
      }
    }
  }
#ifdef RS_NO_WIDE_CELLS
  constexpr bool WIDE_CELLS = false;
#else
  // (the 512-thread chain kernels have the LDS for one word per digit and wave -- except 8-byte keys with 8-byte offsets,
  // n > 2^30: their offset tables are 2 KB larger, and with 4 KB more of cells two workgroups no longer fit a CU)
  constexpr bool WIDE_CELLS = CHAIN && BLOCK <= 512 && !(sizeof(K) == 8 && sizeof(D) == 8);
#endif
  using Smem = SweepSmem<K, V, D, BLOCK, ITEMS, WIDE_CELLS>;
  using Cell = typename Smem::Cell;
  constexpr int WAVES = Smem::WAVES;
  constexpr int TILE = Smem::TILE;
  constexpr bool HAS_VAL = Smem::HAS_VAL;
  static_assert(BLOCK >= kRadix, "one thread per digit");
  __shared__ Smem s;
  __shared__ typename Pick<CHAIN, ChainSmem<D, COUNT>, NoChainSmem>::type cs;

  const unsigned tid0 = threadIdx.x;
  const unsigned tiles_total = static_cast<unsigned>((a.n + TILE - 1) / TILE);

  // chain mode: this workgroup's chain is the XCD it runs on; it sweeps tiles of that chain until none is left
  constexpr int CW = CHAIN ? (kXcds / 2 * kRadix + BLOCK - 1) / BLOCK : 1;   // words of cs.ccnt per thread
  unsigned chain = 0, chain_first = 0, chain_end = 0;
  unsigned acc[2 * CW];                    // (chain of the output position, next digit) counts over this workgroup's tiles
  static_assert(CHAIN || !COUNT, "only chain-mode kernels count for the next pass");
#ifdef RS_STAMPS
#ifdef RS_STAMP_COUNT_PASSES     // stamps from the counting passes only (the last pass would overwrite them)
  constexpr bool RS_STAMP_ON = COUNT;
#else
  constexpr bool RS_STAMP_ON = true;
#endif
#endif
  if constexpr (CHAIN) {
    chain = xcc_id();
    const unsigned long long first = static_cast<unsigned long long>(chain) * a.chain_tiles;
    chain_first = first < tiles_total ? static_cast<unsigned>(first) : tiles_total;
    chain_end = first + a.chain_tiles < tiles_total ? static_cast<unsigned>(first + a.chain_tiles) : tiles_total;
    if (tid0 < kRadix) {
      unsigned long long cbase = a.gbase[tid0];
      for (unsigned c = 0; c < chain; ++c) cbase += a.chain_cnt[static_cast<size_t>(c) * a.chain_cnt_stride + tid0];
      cs.cbase[tid0] = static_cast<D>(cbase);
    }
#pragma unroll
    for (int e = 0; e < CW; ++e) {
      acc[2 * e] = acc[2 * e + 1] = 0;
      if (COUNT && tid0 + e * BLOCK < kXcds / 2 * kRadix) cs.ccnt[tid0 + e * BLOCK] = 0;
    }
  }

  bool have_ticket = false;
  for (;;) {   // one tile per trip; kernels without CHAIN leave after the first
  // In the persistent (CHAIN) form everything derived from the thread id is loop-invariant, and the compiler would keep
  // all of it (addresses, segment offsets) in registers across the whole tile: the id is passed through an empty asm
  // so that each trip derives what it needs where it needs it, as the one-tile kernel does.
  unsigned tid = tid0;
  if constexpr (CHAIN) asm volatile("" : "+v"(tid));
  const unsigned lane = tid & (kWave - 1);
  const unsigned wave = tid / kWave;

#ifdef RS_STAMPS
  unsigned long long stamp_begin = wall_clock64();
#endif
  if (tid == 0) {
    if constexpr (CHAIN) {
      if (!have_ticket) {      // (after the first tile the ticket was drawn at the end of the tile before, see there)
        const unsigned k = atomicAdd(&a.ticket[chain * 32], 1u);
        s.tile_id = k < chain_end - chain_first ? chain_first + k : kNoTile;
      }
    } else {
      s.tile_id = draw_tile(a.ticket, tiles_total, a.per_xcd_order != 0);
    }
    s.out_of_range = 0;
  }
#pragma unroll
  for (int i = 0; i < static_cast<int>(sizeof(Cell)) * kRadix / 4 / kWave; ++i) reinterpret_cast<unsigned*>(s.whist[wave])[i * kWave + lane] = 0;
  __syncthreads();

  if (s.tile_id == kNoTile) {                     // workgroup-uniform
    // Without CHAIN there is one workgroup per tile: a ticket beyond the last tile means the tickets were not clear when
    // the pass started.  Reported (RS_ERR_KERNEL), not passed over: the pass would otherwise silently move nothing.
    if (!CHAIN && tid == 0) __hip_atomic_fetch_or(a.error_flag, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    break;
  }
  const uint64_t tile = s.tile_id;
  const uint64_t first_tile = CHAIN ? chain_first : 0u;   // the tile whose aggregate is its inclusive prefix
#ifdef RS_STAMPS
  if (RS_STAMP_ON && tid == 0 && (tile & 63) == 0 && (tile >> 6) < 4096) rs_stamp_buf[tile >> 6] = stamp_begin;
#endif
  RS_STAMP(1);
  // A tile's way to its aggregate goes first.  The tiles behind it in the chain cannot finish their look-back before this
  // tile has published its digit counts, so a wave that is loading or ranking is given priority at the CU's instruction
  // arbiter over waves that are staging or writing out (s_setprio; back to 0 once the aggregate is out).  What this buys is
  // less spread in the ticket-to-aggregate time, which is what successors wait for: uint64 pass 3.73 -> 3.53 ms, uint32
  // 2.05 -> 2.01 ms at N = 2^30.  RS_PRIO_PRE_AGG=0 builds without it.
#ifndef RS_PRIO_PRE_AGG
#define RS_PRIO_PRE_AGG 2
#endif
#if RS_PRIO_PRE_AGG
  __builtin_amdgcn_s_setprio(RS_PRIO_PRE_AGG);
#endif
  const uint64_t tile_base = tile * TILE;
  // Keys in this tile: TILE except in the last tile of the pass.  In 32-bit arithmetic on the tile number, on purpose: the
  // 64-bit form `n - tile_base < TILE ? n - tile_base : TILE` followed by the uniform branch on `tile_count == TILE` below
  // was miscompiled by hipcc 7.2.0 (the compare moved to the vector unit, v_cmp_lt_u64 -> VCC, while the s_cselect_b32
  // that picks the count still read SCC, left over from an unrelated s_or_b64: the ragged tile of untransformed keys then
  // ran with tile_count = TILE).
  const unsigned last_count = static_cast<unsigned>(a.n - static_cast<uint64_t>(tiles_total - 1) * TILE);   // 1 .. TILE
  const unsigned tile_count = s.tile_id == tiles_total - 1 ? last_count : static_cast<unsigned>(TILE);

  // The rest of the tile exists twice: once for full tiles (all but the last tile of a pass), where no key needs a
  // bounds test and every phase is straight-line code -- the sixteen returning adds of the ranking, the sixteen
  // staging-area reads of the write-out are issued back to back and waited for once -- and once for the ragged last
  // tile.  With the test `full || index < tile_count` evaluated at run time each key sat in its own EXEC-masked block
  // and the compiler had to wait for every LDS answer before the next request (one s_waitcnt lgkmcnt(0) per key).
  // Returns false when the workgroup must stop (a run of this tile would leave the array).
  auto sweep_tile = [&](auto full_tag) -> bool {
  constexpr bool full = decltype(full_tag)::value;

  const K* kin = static_cast<const K*>(a.keys_in) + tile_base;
  const K fmi = static_cast<K>(a.fmask_in), smi = static_cast<K>(a.smask_in);
  const bool xf_in = (a.fmask_in | a.smask_in) != 0;

  // Mid-size sorts (SweepArgs::counts): a digit's thread fetches the whole array's count of its digit now and scans it
  // next to the tile's own counts below (fewer than 2^30 keys: 32 bits hold it).
  constexpr bool IN_TILE_BASE = !CHAIN && sizeof(D) == 4;
  unsigned gcount = 0, gexcl = 0;
  if constexpr (IN_TILE_BASE) {
    if (a.counts != nullptr && tid < kRadix) gcount = static_cast<unsigned>(a.counts[tid]);
  }

  // ---- load: wave-striped, so (item, lane) order == key order inside the wave's segment ----------
  K key[ITEMS];
  const unsigned seg = wave * (kWave * ITEMS) + lane;
  if constexpr (full && CHAIN) {
    // Chain mode (16-byte aligned buffers): the wave's segment comes in through LDS-DMA, 1 KB per instruction, into the
    // staging area (idle at this point), and is read back striped.  A 4-byte-per-lane load costs the CU's address unit as
    // much as a 16-byte one, and with single keys in both directions that unit, not HBM, bounds the pass (TA/TD 70-90 %
    // busy, profiles/r02_a_pmc_*): this quarters its work for the loads.  Each wave reads only what it requested itself.
    constexpr unsigned kSegBytes = kWave * ITEMS * sizeof(K);
    unsigned char* land = s.stage + wave * kSegBytes;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(kin) + wave * kSegBytes + lane * 16;
#pragma unroll
    for (unsigned b = 0; b < kSegBytes; b += kWave * 16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b),
                                       (__attribute__((address_space(3))) void*)(land + b), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const K* lk = reinterpret_cast<const K*>(land);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) key[i] = lk[i * kWave + lane];
  } else if (full) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) key[i] = kin[seg + i * kWave];
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      unsigned idx = seg + i * kWave;
      key[i] = idx < tile_count ? kin[idx] : static_cast<K>(0);
    }
  }
  if (xf_in) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) key[i] = xf_forward(key[i], fmi, smi);
  }
#ifdef RS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RS_STAMP(2);

  // ---- rank inside the wave -----------------------------------------------------------------------
  // After round i, whist[wave][d] = number of digit-d keys in rounds 0..i; rank = (count before
  // the round) + (peers in lower lanes).  Lane order inside a round and round order inside the
  // wave are both key order, so equal digits keep their input order (stability).
  // The LDS executes one wave's operations in order, so round i+1's read of whist sees round i's
  // add without any wait in the instruction stream: 2 x ITEMS LDS operations stay in flight and no
  // register dependency links the rounds.
  unsigned short rank[ITEMS];
  // 8-byte kernels keep every key's digit from the ranking to the staging (they have the registers); 4-byte kernels
  // extract it again there: one instruction against a register per key, which the 64-register kernels do not have
  constexpr bool KEEP_DIG = sizeof(K) == 8;
  unsigned char dig[KEEP_DIG ? ITEMS : 1];
  unsigned* wwords = reinterpret_cast<unsigned*>(s.whist[wave]);   // the wave's cells as 32-bit words
  static_assert(RANK_BY_LDS_ATOMIC || !CHAIN, "the ballot form knows the 16-bit cells only");
  if constexpr (RANK_BY_LDS_ATOMIC) {
    // rank = the value a returning LDS add hands back.  When several lanes of ONE wave instruction hit the same
    // word, gfx950 serves them in lane order (lds_order_probe_kernel below checks exactly this when a context
    // is created; the ballot form is used if it ever fails), so equal digits keep lane = key order, and
    // successive rounds are successive instructions of one wave, which the LDS executes in order.
    // A digit held by 16 or more lanes is peeled: its first holder adds the population once and the others
    // derive their rank with mbcnt, so sorted / few-valued inputs do not serialise 64 ways on one word.
    // The ranking is issue-bound (8 waves per SIMD run this loop together), so the test "is this round worth peeling"
    // is made once per group of four rounds, on the group's first round: random digits then pay 9 instructions per
    // key instead of 15.  A group that is not peeled but should have been is still ranked correctly, only slower.
    static_assert(ITEMS % 4 == 0, "rounds are ranked in groups of four");
    // A wave whose rounds are a few CONTIGUOUS runs of equal digits (an ordered input whose neighbours share a digit: a
    // float ramp repeats every value 8-128 times) ranks all its rounds by run heads; peeling would take out one run only,
    // and often none (no run reaches 16 lanes).  Looked at once per tile, on the wave's first round, so that random
    // digits pay a third of an instruction per key for it: one DPP move (the digit of the lane below, across the wave),
    // one compare, one count.  One run over the whole wave reads 1 -- that case stays with the peel, its cheapest form --
    // and random digits read about 64.  (The first version looked inside rows of 16 lanes and counted the row starts as
    // heads: two ALIGNED runs of 32 then read 4 like one run, were peeled, and the second run's 32 lanes queued on one
    // word -- a sorted input with every value 32 times swept at 2.65 ms per pass against 1.88 with distinct values.)
    bool runs_group = false;
    bool special = false;      // some group of four rounds wants the peeled or the run-head form (wave-uniform)
    bool peel_group[ITEMS / 4];
    if constexpr (full) {      // (the ragged last tile ranks by plain adds: correct for any digits, and it is one tile)
      const unsigned d = digit_at(key[0], a.shift, a.bits);
      const unsigned heads = static_cast<unsigned>(__popcll(__ballot(digit_of_lane_below(d) != d)));   // runs in the wave's first round
#ifndef RS_RANK_RUNS_MAX
#define RS_RANK_RUNS_MAX 20
#endif
      runs_group = heads > 1 && heads <= RS_RANK_RUNS_MAX;
      special = runs_group;
#pragma unroll
      for (int g = 0; g < ITEMS; g += 4) {
        const unsigned dg = digit_at(key[g], a.shift, a.bits);
        peel_group[g / 4] = __popcll(__ballot(dg == static_cast<unsigned>(__builtin_amdgcn_readfirstlane(dg)))) >= 16;
        special |= peel_group[g / 4];
      }
    }
#ifdef RS_NO_RANK16
    if constexpr (full) special = true;
#endif
    if (!special) {
      // the common case (and the ragged tile): one returning add per key, all ITEMS of them in flight before the first
      // answer is looked at
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        unsigned d = digit_at(key[i], a.shift, a.bits);
        if constexpr (KEEP_DIG) dig[i] = static_cast<unsigned char>(d);
        unsigned r = 0;
        if (full || (seg + i * kWave) < tile_count) r = rank_add<WIDE_CELLS>(wwords, d);
        rank[i] = static_cast<unsigned short>(r);
      }
    } else {
#pragma unroll
      for (int g = 0; g < ITEMS; g += 4) {
        if (runs_group) {
          // (four rounds at a time where the registers allow it; the 1024-thread kernels, held to 64 registers, take two)
          constexpr int RB = BLOCK >= 1024 ? 2 : 4;
#pragma unroll
          for (int h = g; h < g + 4; h += RB) {
            unsigned dd[RB], rr[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
              dd[j] = digit_at(key[h + j], a.shift, a.bits);
              if constexpr (KEEP_DIG) dig[h + j] = static_cast<unsigned char>(dd[j]);
            }
            rank_add_runs_n<WIDE_CELLS, RB>(wwords, dd, rr);
#pragma unroll
            for (int j = 0; j < RB; ++j) rank[h + j] = static_cast<unsigned short>(rr[j]);
          }
        } else if (peel_group[g / 4]) {
#pragma unroll
          for (int i = g; i < g + 4; ++i) {
            unsigned d = digit_at(key[i], a.shift, a.bits);
            if constexpr (KEEP_DIG) dig[i] = static_cast<unsigned char>(d);
            rank[i] = static_cast<unsigned short>(rank_add_peeled<WIDE_CELLS>(wwords, d));
          }
        } else {
#pragma unroll
          for (int i = g; i < g + 4; ++i) {
            unsigned d = digit_at(key[i], a.shift, a.bits);
            if constexpr (KEEP_DIG) dig[i] = static_cast<unsigned char>(d);
            rank[i] = static_cast<unsigned short>(rank_add<WIDE_CELLS>(wwords, d));
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      unsigned d = digit_at(key[i], a.shift, a.bits);
      if constexpr (KEEP_DIG) dig[i] = static_cast<unsigned char>(d);
      unsigned dlo, dhi;
      differ_mask(d, dlo, dhi);
      unsigned plo = ~dlo, phi = ~dhi;                   // peers: lanes holding the same digit
      bool ok = true;
      if (!full) {
        ok = (seg + i * kWave) < tile_count;
        unsigned long long alive = __ballot(ok);
        plo &= static_cast<unsigned>(alive);
        phi &= static_cast<unsigned>(alive >> 32);
      }
      unsigned below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
      // (read through the same word pointer the adds use, as an atomic load: a plain 16-bit load of the cell would alias
      // the previous round's add through an unrelated type and could legally be hoisted above it)
      unsigned before = (__hip_atomic_load(&wwords[d >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> ((d & 1u) * 16)) & 0xFFFFu;
      if (ok && below == 0) atomicAdd(&wwords[d >> 1], static_cast<unsigned>(__popc(plo) + __popc(phi)) << ((d & 1u) * 16));
      rank[i] = static_cast<unsigned short>(before + below);
    }
  }
  __syncthreads();
  RS_STAMP(3);
#if RS_PRIO_PRE_AGG
  if (tid >= kRadix) __builtin_amdgcn_s_setprio(0);
#endif

  // ---- per digit: tile count, publish the aggregate, scan over digits -----------------------------
  D* desc = static_cast<D*>(a.desc);
  unsigned my_count = 0;
  unsigned inc = 0;
  if (tid < kRadix) {
    unsigned sum = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) sum += s.whist[w][tid];
    my_count = sum;
    D word = (static_cast<D>(tile == first_tile ? a.incl_code : kCodeAgg) << Desc<D>::kShift) | static_cast<D>(sum);
    if (CHAIN && a.desc_local) desc_store<true>(&desc[tile * kRadix + tid], word); else desc_store<false>(&desc[tile * kRadix + tid], word);
    unsigned v = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      unsigned u = __shfl_up(v, off, kWave);
      if (lane >= static_cast<unsigned>(off)) v += u;
    }
    inc = v;
    if (lane == kWave - 1) s.wave_total[wave] = v;
    if constexpr (IN_TILE_BASE) {
      if (a.counts != nullptr) {
        unsigned g = gcount;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
          unsigned u = __shfl_up(g, off, kWave);
          if (lane >= static_cast<unsigned>(off)) g += u;
        }
        gexcl = g - gcount;
        if (lane == kWave - 1) s.gwave_total[wave] = g;
      }
    }
#if RS_PRIO_PRE_AGG
    __builtin_amdgcn_s_setprio(0);
#endif
  }
  __syncthreads();
  if (tid < kRadix) {
    if constexpr (IN_TILE_BASE) {
      if (a.counts != nullptr)
        for (unsigned w = 0; w < wave; ++w) gexcl += s.gwave_total[w];
    }
    unsigned before = 0;
    for (unsigned w = 0; w < wave; ++w) before += s.wave_total[w];
    unsigned start = before + inc - my_count;
    s.tile_start[tid] = start;
    unsigned sum = start;               // fold tile_start into the wave prefix: one LDS read per key later
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      unsigned c = s.whist[w][tid];
      s.whist[w][tid] = static_cast<Cell>(sum);
      sum += c;
    }
  }
  __syncthreads();
  RS_STAMP(4);

  // ---- stage keys in LDS in digit order -------------------------------------------------------------
  // (two loops: all the reads of the wave's prefixes first, in flight together; in one loop every staging store, which the
  // compiler must assume to alias the next read, waited for the read before it)
  K* stage_k = reinterpret_cast<K*>(s.stage);
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) rank[i] = static_cast<unsigned short>(s.whist[wave][KEEP_DIG ? static_cast<unsigned>(dig[i]) : digit_at(key[i], a.shift, a.bits)] + rank[i]);
  auto stage_keys = [&](auto from_tag, auto to_tag) {
#pragma unroll
    for (int i = decltype(from_tag)::value; i < decltype(to_tag)::value; ++i)
      if (full || (seg + i * kWave) < tile_count) stage_k[stage_swizzle(rank[i])] = key[i];
  };
#ifndef RS_LOOKBACK_FIRST
#define RS_LOOKBACK_FIRST 0
#endif
  // The digit waves stage only their first ITEMS - kLate keys now and the others after the look-back, so that the
  // inclusive prefix -- what the tiles behind this one wait for -- is published that much earlier.
#ifndef RS_LOOKBACK_FIRST_LAST   // the last launched pass (no counting): all of them late -- uint32 last pass 1.93 -> 1.84 ms at N = 2^30;
#define RS_LOOKBACK_FIRST_LAST 16 // the counting kernels spill with more than 12 and show no gain with 8 or 12
#endif
  constexpr int kLateWanted = !CHAIN ? 0 : COUNT ? RS_LOOKBACK_FIRST : RS_LOOKBACK_FIRST_LAST;
  constexpr int kLate = kLateWanted < ITEMS ? kLateWanted : ITEMS;
  static_assert(kLate >= 0, "keys staged after the look-back");
  if (kLate == 0 || tid >= kRadix) stage_keys(std::integral_constant<int, 0>{}, std::integral_constant<int, ITEMS>{});
  else stage_keys(std::integral_constant<int, 0>{}, std::integral_constant<int, ITEMS - kLate>{});
  RS_STAMP(5);

  // ---- decoupled look-back: one thread per digit, LOOKBACK predecessors per round trip --------------
  if (tid < kRadix) {
    unsigned long long excl = 0;
    if (tile > first_tile) {
      long long t = static_cast<long long>(tile) - 1;     // next predecessor to consume
      SpinGuard guard;
      bool failed = false;
#ifdef RS_STAMPS
      unsigned trips = 0, empty_trips = 0;
#endif
      while (true) {
#ifdef RS_STAMPS
        ++trips;
#endif
        D w[LOOKBACK];
#pragma unroll
        for (int j = 0; j < LOOKBACK; ++j) {
          long long tt = t - j;
          w[j] = desc_load(&desc[static_cast<uint64_t>(tt < static_cast<long long>(first_tile) ? first_tile : tt) * kRadix + tid]);
        }
        bool done = false;
        int consumed = 0;
#pragma unroll
        for (int j = 0; j < LOOKBACK; ++j) {
          if (!done && consumed == j) {
            unsigned code = static_cast<unsigned>(w[j] >> Desc<D>::kShift);
            if (code == a.incl_code) { excl += static_cast<unsigned long long>(w[j] & Desc<D>::kValueMask); done = true; }
            else if (code == kCodeAgg) { excl += static_cast<unsigned long long>(w[j] & Desc<D>::kValueMask); ++consumed; }
          }
        }
        if (done) break;
        t -= consumed;
#ifdef RS_STAMPS
        if (consumed == 0) ++empty_trips;
#endif
        if (consumed == 0 && guard.expired(a.error_flag)) { failed = true; break; }
      }
#ifdef RS_STAMPS
      if (RS_STAMP_ON && tid == 0 && (tile & 63) == 0 && (tile >> 6) < 4096) {
        rs_stamp_buf[8 * 4096 + (tile >> 6)] = trips;
        rs_stamp_buf[9 * 4096 + (tile >> 6)] = empty_trips;
        rs_stamp_buf[10 * 4096 + (tile >> 6)] = static_cast<unsigned long long>(static_cast<long long>(tile) - 1 - t);
      }
#endif
      if (failed) __hip_atomic_fetch_or(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      D word = (static_cast<D>(a.incl_code) << Desc<D>::kShift) |
               (static_cast<D>(excl + my_count) & Desc<D>::kValueMask);
      if (CHAIN && a.desc_local) desc_store<true>(&desc[tile * kRadix + tid], word); else desc_store<false>(&desc[tile * kRadix + tid], word);
    }
    unsigned long long run_begin = excl;
    if constexpr (CHAIN) run_begin += cs.cbase[tid];
    else if (IN_TILE_BASE && a.counts != nullptr) run_begin += gexcl;
    else run_begin += a.gbase[tid];
    s.gofs[tid] = static_cast<D>(run_begin - s.tile_start[tid]);
    // Offsets come from the up-front histogram; keys that change while the sort runs (a caller that did not wait for
    // whatever produces them) make the two disagree.  A tile with a run that would end past the array reports that
    // and writes nothing: one compare per digit here instead of one per key in the write-out.
    if (my_count != 0 && run_begin + my_count > a.n) s.out_of_range = 1;
    if constexpr (kLate > 0) stage_keys(std::integral_constant<int, ITEMS - kLate>{}, std::integral_constant<int, ITEMS>{});
  }
  __syncthreads();
  RS_STAMP(6);
  if (s.out_of_range) {                            // workgroup-uniform
    if (tid == 0) __hip_atomic_fetch_or(a.error_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
  }

  // ---- write out: consecutive threads write consecutive addresses inside each digit run -------------
  // In phases over all ITEMS rounds, each a straight line: read the staged keys, read their runs' offsets, count for the
  // next pass, transform, store.  The LDS reads of a phase are independent and all in flight together; written as one loop
  // over rounds the compiler kept one or two in flight (the counting adds are atomics, nothing moves across them).
  K* kout = static_cast<K*>(a.keys_out);
  const K fmo = static_cast<K>(a.fmask_out), smo = static_cast<K>(a.smask_out);
  const bool xf_out = (a.fmask_out | a.smask_out) != 0;
  D dst[HAS_VAL ? ITEMS : 1];
  // staged element j = i * BLOCK + tid sits at stage_swizzle(j) = j ^ ((j >> 6) & 31).  BLOCK / 64 is a power of two of at
  // most 32, so (j >> 6) & 31 = ((i * BLOCK / 64) & 31) | (tid >> 6) with no carry between the two parts, and the swizzled
  // index is a per-round constant XOR a per-thread constant: one instruction per key instead of three.
  static_assert((BLOCK / kWave) <= 32 && ((BLOCK / kWave) & (BLOCK / kWave - 1)) == 0, "swizzle split");
  const unsigned sw_t = tid ^ (tid >> 6);
  K wk[ITEMS];
  D wo[ITEMS];
  // Counting kernels add every key to its (chain of the output position, next digit) cell.  RUNS selects how: false = one
  // LDS add per key, straight-line code; true = by run heads: the staged keys are in digit order, so on ordered inputs the
  // 64 lanes of a round hold a few CONTIGUOUS runs of keys that share their cell (a sorted ramp: one run; nearly sorted: the
  // ramp's run, cut here and there by a displaced key), and 64 single adds would queue on one LDS word.  The first lane of
  // every run adds the run's length and the others add nothing -- the run-head form of the ranking (rank_add_runs_n)
  // without its broadcast.  Correct for any cells (64 runs of one: the plain form with more instructions).  The form is
  // chosen per batch of rounds from the batch's SECOND round (one DPP move, one compare, one count): a wave's rounds are
  // 512 staged positions apart, and the first and the last round of a tile of a nearly sorted input hold its displaced keys,
  // one cell each, while the rounds between are one run.  Measured, uint32 N = 2^30, first sweep of the hybrid form on the
  // nearly sorted shape: 4.15 ms with the earlier two-leader peel chosen per tile from its first round, 2.85 ms with run
  // heads chosen per batch from its first round; a choice per round costs random keys 0.12 ms per pass (2.11 -> 2.23) and a
  // group-leader form for few interleaved values did not move the 256-value duplicates shape (2.5-2.9 ms either way).
  // Needs all 64 lanes (full tiles).
#ifndef RS_COUNT_RUNS_MAX
#define RS_COUNT_RUNS_MAX 16
#endif
  auto cell_of = [&](int i) -> unsigned {
    const unsigned dn = digit_at(wk[i], a.next_shift, a.next_bits);
    const unsigned c = __umul24(static_cast<unsigned>(wo[i] >> a.chunk_shift), a.chunk_mul) >> 16;
    return (c << 8) | dn;
  };
  auto count_round = [&](auto runs_tag, int i) {
    constexpr bool RUNS = decltype(runs_tag)::value;
    if constexpr (CHAIN && COUNT) if (full || i * BLOCK + tid < tile_count) {
      const unsigned cell = cell_of(i);
      const unsigned c = cell >> 8, dn = cell & 0xFFu;
      if constexpr (!RUNS) {
        atomicAdd(&cs.ccnt[((c >> 1) << 8) | dn], 1u << ((c & 1u) << 4));
      } else {
        const unsigned long long heads = __ballot(digit_of_lane_below(cell) != cell);      // lane 0 always
        const unsigned long long above = lane == 63u ? 0ull : heads >> (lane + 1u);
        const unsigned run_end = above ? lane + static_cast<unsigned>(__ffsll(static_cast<long long>(above))) : 64u;   // next head, or one past the wave
        if ((heads >> lane) & 1ull) atomicAdd(&cs.ccnt[((c >> 1) << 8) | dn], (run_end - lane) << ((c & 1u) << 4));
      }
    }
  };
  // The rounds go in batches of WB: within a batch every phase covers all its rounds before the next phase starts.
#ifndef RS_WOUT_BATCH
#define RS_WOUT_BATCH 4
#endif
  constexpr int WB = RS_WOUT_BATCH < ITEMS ? RS_WOUT_BATCH : ITEMS;
  static_assert(ITEMS % WB == 0, "write-out batches");
#ifndef RS_COUNT_RUNS_MAX
#define RS_COUNT_RUNS_MAX 16
#endif
#pragma unroll
  for (int b0 = 0; b0 < ITEMS; b0 += WB) {
#pragma unroll
    for (int i = b0; i < b0 + WB; ++i) {
      const unsigned sw_i = static_cast<unsigned>(i * BLOCK) ^ (static_cast<unsigned>(i * (BLOCK / kWave)) & 31u);
      if (full || i * BLOCK + tid < tile_count) wk[i] = stage_k[sw_i ^ sw_t];
    }
#pragma unroll
    for (int i = b0; i < b0 + WB; ++i) {
      const unsigned j = i * BLOCK + tid;
      if (full || j < tile_count) wo[i] = s.gofs[digit_at(wk[i], a.shift, a.bits)] + j;
    }
    if constexpr (CHAIN && COUNT) {
      bool runs = false;                 // (wave-uniform)
      if constexpr (full && WB >= 2) {
        const unsigned cell = cell_of(b0 + 1);
        runs = __popcll(__ballot(digit_of_lane_below(cell) != cell)) <= RS_COUNT_RUNS_MAX;
      }
      if (runs) {
#pragma unroll
        for (int i = b0; i < b0 + WB; ++i) count_round(std::true_type{}, i);
      } else {
#pragma unroll
        for (int i = b0; i < b0 + WB; ++i) count_round(std::false_type{}, i);
      }
    }
    if (xf_out) {                                      // workgroup-uniform: the last launched pass only
#pragma unroll
      for (int i = b0; i < b0 + WB; ++i) wk[i] = xf_inverse(wk[i], fmo, smo);
    }
#pragma unroll
    for (int i = b0; i < b0 + WB; ++i) {
      if (full || i * BLOCK + tid < tile_count) {
        if constexpr (HAS_VAL || kCheckedStores) {
          // (always true after the per-run check above; kept because the pair kernels schedule measurably better with the
          // store under this predicate: uint32+uint32 pass 3.6-3.7 ms with it, 3.9-4.0 ms without, same results)
          const bool inside = static_cast<uint64_t>(wo[i]) < a.n;
          if (inside) kout[wo[i]] = wk[i];
          if constexpr (HAS_VAL) dst[i] = inside ? wo[i] : ~static_cast<D>(0);
          if constexpr (kCheckedStores) { if (!inside) __hip_atomic_fetch_or(a.error_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        } else {
          kout[wo[i]] = wk[i];
        }
      }
    }
  }
#ifdef RS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RS_STAMP(7);

  if constexpr (HAS_VAL) {
    const V* vin = static_cast<const V*>(a.vals_in) + tile_base;
    V* vout = static_cast<V*>(a.vals_out);
    V val[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      unsigned idx = seg + i * kWave;
      if (full || idx < tile_count) val[i] = vin[idx];
    }
    __syncthreads();
    V* stage_v = reinterpret_cast<V*>(s.stage);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
      if (full || (seg + i * kWave) < tile_count) stage_v[rank[i]] = val[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      unsigned j = i * BLOCK + tid;
      if ((full || j < tile_count) && dst[i] != ~static_cast<D>(0)) vout[dst[i]] = stage_v[j];
    }
  }

#ifndef RS_EARLY_TICKET
#define RS_EARLY_TICKET 0
#endif
  if constexpr (CHAIN && RS_EARLY_TICKET) {
    // Measured and left off: the next tile's ticket drawn now that this tile's stores are issued, so that the round trip of
    // the atomic (about 0.5 us) overlaps the end of the tile instead of opening the next one.  uint32 N = 2^30: 8.77 ->
    // 8.90 ms -- a ticket that exists before its workgroup is free lengthens the wait of every tile behind it, as the
    // prefetching variants did.  (tile_id was last read before this tile's barriers.)
    if (tid == 0) {
      const unsigned k = atomicAdd(&a.ticket[chain * 32], 1u);
      s.tile_id = k < chain_end - chain_first ? chain_first + k : kNoTile;
    }
  }
  return true;
  };   // sweep_tile
  const bool go_on = tile_count == TILE ? sweep_tile(std::true_type{}) : sweep_tile(std::false_type{});
  if (!go_on) break;
  have_ticket = RS_EARLY_TICKET != 0;

  if constexpr (!CHAIN) break;
  else {
    // fold this tile's (chain, next digit) cells into the registers and clear them for the next tile
    __syncthreads();
    if constexpr (COUNT) {
#pragma unroll
      for (int e = 0; e < CW; ++e) {
        const unsigned wi = tid + e * BLOCK;
        if (wi < kXcds / 2 * kRadix) {
          const unsigned w = cs.ccnt[wi];
          cs.ccnt[wi] = 0;
          acc[2 * e] += w & 0xFFFFu;
          acc[2 * e + 1] += w >> 16;
        }
      }
    }
  }
  }   // tile loop

  if constexpr (CHAIN) {
    if constexpr (COUNT) {
#pragma unroll
      for (int e = 0; e < CW; ++e) {
        const unsigned wi = tid0 + e * BLOCK;
        if (wi < kXcds / 2 * kRadix) {
          const unsigned c2 = wi >> 8, dn = wi & (kRadix - 1);
          if (acc[2 * e]) atomicAdd(&a.next_cnt[(2 * c2) * kRadix + dn], static_cast<unsigned long long>(acc[2 * e]));
          if (acc[2 * e + 1]) atomicAdd(&a.next_cnt[(2 * c2 + 1) * kRadix + dn], static_cast<unsigned long long>(acc[2 * e + 1]));
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// 3b. permutation gather of fixed-size rows (sortByIndices, radix_sort.hpp:1111-1154, for trivially copyable objects)
// ------------------------------------------------------------------------------------------------
// out row i = in row perm[i]; rows are `words` 32-bit words long.  Thread per word: the lanes that share a row read
// consecutive words of it, rows land at random (one 32..256-byte piece per row), writes are contiguous.
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t* in, const uint32_t* perm, uint32_t* out,
                                                          uint64_t n, unsigned words) {
  const uint64_t total = n * words;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    const uint64_t row = i / words;
    const unsigned w = static_cast<unsigned>(i - row * words);
    out[i] = in[static_cast<uint64_t>(perm[row]) * words + w];
  }
}
__global__ __launch_bounds__(256) void iota_u32_kernel(uint32_t* a, uint64_t n) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = static_cast<uint32_t>(i);
}

// ------------------------------------------------------------------------------------------------
// 4. small inputs: the whole sort in one workgroup
// ------------------------------------------------------------------------------------------------
// n <= one tile.  One launch does every pass: keys (and payload) stay in registers / LDS between passes, there is no
// histogram kernel, no descriptor, no ticket and nothing for the host to wait for.  Each pass is the tile part of
// onesweep_kernel (rank by returning LDS adds in key order -> scan over digits -> scatter to the staging area), then
// the keys are read back wave-striped for the next pass.  A pass whose digit is the same for all keys is skipped
// (getMaxLength's effect, radix_sort.hpp:307-370).  Stable, so it serves the pair sort as well.
struct SmallArgs {
  void* keys;          // sorted in place
  void* vals;          // payload, or null
  unsigned n;
  uint64_t fmask, smask;
  int begin_bit, end_bit;
};

template <typename K, typename V, int BLOCK, int ITEMS>
struct TileSortSmem {
  static constexpr int WAVES = BLOCK / kWave;
  static constexpr int TILE = BLOCK * ITEMS;
  static constexpr bool HAS_VAL = !__is_same(V, NoVal);
  alignas(16) K stage_k[TILE];
  alignas(16) V stage_v[HAS_VAL ? TILE : 1];
  alignas(16) unsigned short whist[WAVES][kRadix];   // (updated two cells at a time through 32-bit LDS atomics: word-aligned, whatever precedes it)
  unsigned wave_total[kRadix / kWave];
};

// The whole sort of at most BLOCK * ITEMS keys (and payload) by one workgroup, keys read at gin and stored at gk (the same
// array, or two of one size), payload in place at gv: every pass in registers and LDS.  fm_in / sm_in: transform applied to the keys as they are read (zero when they are transformed already);
// fm_out / sm_out: inverse transform applied as they are stored.  Used by small_sort_kernel (one array) and by
// group_sort_wide_kernel (rsort_groups.hpp: one group of a larger array per call).  Ends with every thread past its last
// LDS access of the call only after a barrier of its own, so calls can follow each other in one kernel.
// ITEMS = keys per thread of THIS call: at most what the shared-memory block was sized for (a kernel that sorts groups of
// different sizes instantiates the body for several and picks the smallest that holds the group: every phase is a loop
// over ITEMS rounds whether they hold keys or not).
//
// TOP (0 = off; needs begin_bit == 0 and keys that agree on every bit from end_bit up, as the keys of one group do; stable,
// so it serves pairs too: equal keys are never counted as larger or smaller and keep their places): sort on the bits [TOP, end_bit) alone, then put the keys that agree on those bits -- they are neighbours now -- in
// order among themselves by comparing whole keys: every key counts the larger keys of its run before it and the smaller
// ones behind it, and moves by the difference.  Two passes and the fix-up instead of six or seven passes for the groups of
// 8-byte keys, whose runs of equal 16-bit prefix are one or two keys long.  A run that reaches more than kFixRunMax
// places to either side of one of its members calls the fix-up off and the passes run again, over all the bits.
constexpr unsigned kFixRunMax = 16;
template <typename K, typename V, int BLOCK, int ITEMS, int TOP = 0, typename Smem>
__device__ __forceinline__ void tile_sort_body(Smem& s, const K* gin, K* gk, V* gv, unsigned n, K fm_in, K sm_in,
                                               K fm_out, K sm_out, int begin_bit, int end_bit) {
  constexpr int WAVES = BLOCK / kWave;
  constexpr int TILE = BLOCK * ITEMS;
  static_assert(TILE <= Smem::TILE && BLOCK == Smem::WAVES * kWave, "the staging area holds the call's keys");
  constexpr bool HAS_VAL = !__is_same(V, NoVal);
  static_assert(TILE < 65536 && BLOCK >= kRadix, "16-bit positions; one thread per digit");
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const unsigned seg = wave * (kWave * ITEMS) + lane;

  K key[ITEMS];
  V val[HAS_VAL ? ITEMS : 1];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const unsigned idx = seg + i * kWave;
    key[i] = idx < n ? xf_forward(gin[idx], fm_in, sm_in) : static_cast<K>(0);
    if constexpr (HAS_VAL) { if (idx < n) val[i] = gv[idx]; }
  }

  unsigned* wwords = reinterpret_cast<unsigned*>(s.whist[wave]);
  constexpr int top_bit = TOP;
  bool fix = TOP > 0 && begin_bit == 0 && top_bit < end_bit && n > 1;      // (workgroup-uniform)
  bool staged = false;                                     // the staging area holds the keys in their current order
  unsigned short moved_to[TOP > 0 ? ITEMS : 1];            // where the fix-up sends each key (else it stays where the passes left it)
  if constexpr (TOP > 0) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) moved_to[i] = static_cast<unsigned short>(seg + i * kWave);
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
  for (int shift = fix ? top_bit : begin_bit; shift < end_bit; shift += kRadixBits) {
    const int bits = end_bit - shift < kRadixBits ? end_bit - shift : kRadixBits;
#pragma unroll
    for (int i = 0; i < kRadix / 2 / kWave; ++i) wwords[i * kWave + lane] = 0;
    // rank inside the wave (see onesweep_kernel: lane order inside one returning LDS add, round order inside the wave)
    unsigned short rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const unsigned d = digit_at(key[i], shift, bits);
      unsigned r = 0;
      if (seg + i * kWave < n) r = rank_add_peeled(wwords, d);
      rank[i] = static_cast<unsigned short>(r);
    }
    __syncthreads();
    // per digit: count over the waves; a digit held by every key makes the pass the identity
    unsigned my_count = 0, inc = 0;
    if (tid < kRadix) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) my_count += s.whist[w][tid];
      unsigned v = my_count;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        unsigned u = __shfl_up(v, off, kWave);
        if (lane >= static_cast<unsigned>(off)) v += u;
      }
      inc = v;
      if (lane == kWave - 1) s.wave_total[wave] = v;
    }
    if (__syncthreads_or(tid < kRadix && my_count == n)) continue;     // workgroup-uniform
    if (tid < kRadix) {
      unsigned before = 0;
      for (unsigned w = 0; w < wave; ++w) before += s.wave_total[w];
      unsigned sum = before + inc - my_count;       // start of the digit's run, then + the waves before each wave
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        unsigned c = s.whist[w][tid];
        s.whist[w][tid] = static_cast<unsigned short>(sum);
        sum += c;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (seg + i * kWave < n) {
        const unsigned pos = s.whist[wave][digit_at(key[i], shift, bits)] + rank[i];
        s.stage_k[pos] = key[i];
        if constexpr (HAS_VAL) s.stage_v[pos] = val[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const unsigned idx = seg + i * kWave;
      if (idx < n) {
        key[i] = s.stage_k[idx];
        if constexpr (HAS_VAL) val[i] = s.stage_v[idx];
      }
    }
    staged = true;
    // no barrier: the next pass writes the staging area only behind its own two barriers, and each wave zeroes and
    // fills only its own histogram row before the first of them
  }
  if (!fix) break;
  // the fix-up.  The staging area may not hold the keys at all (every pass skipped: one prefix for the whole tile); it is
  // only read from here on, and the keys go from the registers straight to their places in the output.
  if constexpr (TOP > 0) {
  if (!staged) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
      if (seg + i * kWave < n) s.stage_k[seg + i * kWave] = key[i];
    __syncthreads();
    staged = true;
  }
  // Round r looks r places to either side.  Rounds 1 and 2 for all the keys of the thread at once, straight-line, the reads
  // in flight together (a loop per key over its own run costs five counting passes in waits); from round 3 on only
  // for the few keys whose run reaches that far, and a wave stops when it has none.
  unsigned alive_l = 0, alive_r = 0;                       // bit i: key i's run may reach further on that side
#pragma unroll
  for (int i = 0; i < ITEMS; ++i)
    if (seg + i * kWave < n) { alive_l |= 1u << i; alive_r |= 1u << i; }
  // (8-byte keys split at bit 32: one 32-bit compare each for "same prefix" and, among keys of one prefix, "larger")
  constexpr bool kSplitWords = sizeof(K) == 8 && TOP == 32;
  auto same_prefix = [](K x, K y) {
    if constexpr (kSplitWords) return static_cast<unsigned>(static_cast<unsigned long long>(x) >> 32) == static_cast<unsigned>(static_cast<unsigned long long>(y) >> 32);
    else return ((x ^ y) >> top_bit) == 0;
  };
  auto low_part = [](K x) {
    if constexpr (kSplitWords) return static_cast<unsigned>(x);
    else return x;
  };
  unsigned seen_larger = 0;                                // this thread has met a larger key of a key's run before that key (an inversion inside a run)
  auto look_left = [&](int i, unsigned r, K o) {
    const bool same = ((alive_l >> i) & 1u) && seg + i * kWave >= r && same_prefix(o, key[i]);
    if (same && low_part(o) > low_part(key[i])) { --moved_to[i]; seen_larger = 1u; }       // a larger key before this one comes to stand behind it
    if (!same) alive_l &= ~(1u << i);
  };
  auto look_right = [&](int i, unsigned r, K o) {
    const bool same = ((alive_r >> i) & 1u) && seg + i * kWave + r < n && same_prefix(o, key[i]);
    if (same && low_part(o) < low_part(key[i])) { ++moved_to[i]; seen_larger = 1u; }       // a smaller key behind this one comes to stand before it
    if (!same) alive_r &= ~(1u << i);
  };
  // One round for all the keys of the thread.  Only the LEFT neighbour is read and compared: "key p has a smaller key of its
  // run r places behind it" is "key p + r has a larger key of its run r places before it", which the thread r lanes further
  // on (or, past lane 63, the next row of the same thread set) has just worked out -- two cross-lane reads of bit masks
  // per round instead of a second LDS read and compare per key.  Only the wave's last row has its partners in another wave
  // and looks to the right itself.
  auto round_all = [&](unsigned r) {
    constexpr int CH = ITEMS < 8 ? ITEMS : 8;              // (keys in flight per batch: registers)
    unsigned larger_before = 0;                            // bit i: the key r places before key i is of its run and larger
#pragma unroll
    for (int base = 0; base < ITEMS; base += CH) {
      K other[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (base + c < ITEMS) { const unsigned idx = seg + (base + c) * kWave; other[c] = s.stage_k[idx >= r ? idx - r : 0u]; }
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (base + c < ITEMS) {
          const int i = base + c;
          const bool same = ((alive_l >> i) & 1u) && seg + i * kWave >= r && same_prefix(other[c], key[i]);
          const bool larger = same && low_part(other[c]) > low_part(key[i]);
          if (larger) { --moved_to[i]; larger_before |= 1u << i; }
          if (!same) alive_l &= ~(1u << i);
        }
    }
    // (alive_l now says: the key r places before is of this key's run)
    seen_larger |= larger_before;
    const bool wrapped = lane + r >= kWave;                // the partner r places on sits in the thread set's next row
    unsigned run_on = __shfl(alive_l, static_cast<int>((lane + r) & (kWave - 1)), kWave);
    unsigned smaller_behind = __shfl(larger_before, static_cast<int>((lane + r) & (kWave - 1)), kWave);
    if (wrapped) { run_on >>= 1; smaller_behind >>= 1; }   // its bit i + 1 speaks of this thread's key i
    constexpr unsigned last = 1u << (ITEMS - 1);
    const unsigned keep = wrapped ? alive_r & last : 0u;   // the last row's partners are in the next wave: looked at below
    run_on &= alive_r;
    smaller_behind &= run_on;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) moved_to[i] = static_cast<unsigned short>(moved_to[i] + ((smaller_behind >> i) & 1u));
    alive_r = (wrapped ? run_on & ~last : run_on) | keep;
    if (wrapped) {
      const unsigned idx = seg + (ITEMS - 1) * kWave;
      look_right(ITEMS - 1, r, s.stage_k[((alive_r >> (ITEMS - 1)) & 1u) && idx + r < n ? idx + r : 0u]);
    }
  };
  bool too_long = false;
  round_all(1u);
  const bool inverted = seen_larger != 0;                  // a key of this thread stands behind a larger left neighbour of its run
  if (__any((alive_l | alive_r) != 0)) round_all(2u);
  for (unsigned r = 3; __any((alive_l | alive_r) != 0); ++r) {
    // (a wave whose runs go on although none of its keys has met an inversion yet: runs of one key repeated, most likely --
    // keys of few distinct values; it stops looking, and the workgroup finds out below whether that was all)
    if (r > kFixRunMax || (r > 3 && !__any(seen_larger != 0))) { too_long = true; break; }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const unsigned idx = seg + i * kWave;
      if (__any((alive_l >> i) & 1u)) look_left(i, r, s.stage_k[((alive_l >> i) & 1u) && idx >= r ? idx - r : 0u]);
      if (__any((alive_r >> i) & 1u)) look_right(i, r, s.stage_k[((alive_r >> i) & 1u) && idx + r < n ? idx + r : 0u]);
    }
  }
  bool again = __syncthreads_or(too_long);                 // (workgroup-uniform; also the last LDS access of the call)
  // No key of the group stands behind a larger left neighbour of its run (round 1 compared them all): every run is in order
  // already -- one key repeated, when the keys take a few thousand to a few million distinct values, far fewer than there
  // are prefixes -- and nothing has moved or has to, however long the runs are.  (Without this, runs of 17 and more equal
  // keys sent every group through all the passes again.)
  if (again && !__syncthreads_or(inverted)) again = false;
  if (again) {                                             // the passes again, over all the bits
#ifdef RS_FIX_STATS
    if (tid == 0) atomicAdd(RS_FIX_STATS, 1u);
#endif
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) moved_to[i] = static_cast<unsigned short>(seg + i * kWave);
    fix = false;
    continue;
  }
  // The keys that move change places through the staging area (every reader of it is past the barrier above): the places
  // of a run are permuted among its keys, so a key that stays is not displaced, only the threads whose key moved read
  // their place again, and the store below writes whole lines, each thread its own places.
  // -DRS_FIX_DIRECT_STORE stores the moved keys straight from the registers, a place or two beside the thread's own:
  // two barriers less and 5 % less time (34.2 against 36.0 us per group of 16384 8-byte keys), but lines with a moved key
  // are written in parts -- PMC at N = 2^30: 1.52 x the algorithmic bytes against 1.00 x (profiles/r03_c_pmc.txt had it).
#ifndef RS_FIX_DIRECT_STORE
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const unsigned idx = seg + i * kWave;
    if (idx < n && moved_to[i] != idx) {
      s.stage_k[moved_to[i]] = key[i];
      if constexpr (HAS_VAL) s.stage_v[moved_to[i]] = val[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const unsigned idx = seg + i * kWave;
    if (idx < n && moved_to[i] != idx) {
      key[i] = s.stage_k[idx];
      if constexpr (HAS_VAL) val[i] = s.stage_v[idx];
      moved_to[i] = static_cast<unsigned short>(idx);
    }
  }
  __syncthreads();                                         // (the caller may start another call on this staging area)
#endif
  }
  break;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const unsigned idx = seg + i * kWave;
    if (idx < n) {
      unsigned to = idx;
      if constexpr (TOP > 0) to = moved_to[i];
      gk[to] = xf_inverse(key[i], fm_out, sm_out);
      if constexpr (HAS_VAL) gv[to] = val[i];
    }
  }
}

template <typename K, typename V, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void small_sort_kernel(SmallArgs a) {
  __shared__ TileSortSmem<K, V, BLOCK, ITEMS> s;
  const K fm = static_cast<K>(a.fmask), sm = static_cast<K>(a.smask);
  tile_sort_body<K, V, BLOCK, ITEMS>(s, static_cast<const K*>(a.keys), static_cast<K*>(a.keys), static_cast<V*>(a.vals), a.n, fm, sm, fm, sm, a.begin_bit, a.end_bit);
}

}  // namespace rsk
