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
//                              the cross-tile prefix is a decoupled look-back chained scan
//
// Memory-model notes (MI355X: 8 XCDs with private, mutually non-coherent L2s):
//   * look-back descriptors are single 4- or 8-byte words that carry status AND value, written
//     and polled with relaxed agent-scope atomics (sc1 stores / sc1 loads): the word is its own
//     flag, so no fence is needed around it;
//   * tile ids come from an atomic ticket, not blockIdx: a tile's predecessors have all started
//     before it exists, which is the forward-progress argument for the look-back spin;
//   * keys are written with plain stores and only read by the NEXT kernel on the same stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

template <typename K> __device__ __forceinline__ unsigned digit_at(K k, int shift, unsigned mask) {
  return static_cast<unsigned>(k >> shift) & mask;
}

__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
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
};

template <typename K, int MAX_PASSES>
__device__ __forceinline__ void hist_add_key(K k, unsigned* lh, const HistArgs& a) {
#pragma unroll
  for (int p = 0; p < MAX_PASSES; ++p) {
    if (p < a.passes) {
      int shift = a.begin_bit + p * kRadixBits;
      int bits = a.end_bit - shift;
      unsigned mask = bits >= kRadixBits ? 0xFFu : ((1u << bits) - 1u);
      unsigned d = digit_at(k, shift, mask);
      // Skewed inputs (sorted shapes: every lane of a wave holds the same high digit) would
      // serialise 64 ways on one LDS word: one lane adds the wave's population instead.
      unsigned d0 = __builtin_amdgcn_readfirstlane(d);
      unsigned long long same = __ballot(d == d0);
      unsigned long long act = __ballot(1);
      if (same == act) {
        if (lane_id() == static_cast<unsigned>(__ffsll(static_cast<long long>(act)) - 1))
          atomicAdd(&lh[p * kRadix + d0], static_cast<unsigned>(__popcll(act)));
      } else {
        atomicAdd(&lh[p * kRadix + d], 1u);
      }
    }
  }
}

template <typename K, int BLOCK>
__global__ __launch_bounds__(BLOCK) void digit_histogram_kernel(HistArgs a) {
  constexpr int MAX_PASSES = sizeof(K);
  constexpr int VEC = 16 / sizeof(K);            // 16 bytes per lane per load
  __shared__ unsigned lh[MAX_PASSES * kRadix];
  for (int i = threadIdx.x; i < MAX_PASSES * kRadix; i += BLOCK) lh[i] = 0;
  __syncthreads();

  const K* keys = static_cast<const K*>(a.keys);
  const K fm = static_cast<K>(a.fmask), sm = static_cast<K>(a.smask);
  const uint64_t nvec = a.n / VEC;
  struct alignas(16) Pack { K v[VEC]; };
  const Pack* vkeys = reinterpret_cast<const Pack*>(keys);
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < nvec;
       i += static_cast<uint64_t>(gridDim.x) * BLOCK) {
    Pack pk = vkeys[i];
#pragma unroll
    for (int j = 0; j < VEC; ++j) hist_add_key<K, MAX_PASSES>(xf_forward(pk.v[j], fm, sm), lh, a);
  }
  if (blockIdx.x == 0) {                          // tail (n not a multiple of VEC)
    uint64_t i = nvec * VEC + threadIdx.x;
    if (i < a.n) hist_add_key<K, MAX_PASSES>(xf_forward(keys[i], fm, sm), lh, a);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.passes * kRadix; i += BLOCK) {
    unsigned c = lh[i];
    if (c) atomicAdd(&a.ghist[i], static_cast<unsigned long long>(c));
  }
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

__global__ __launch_bounds__(kRadix) void digit_scan_kernel(const unsigned long long* ghist,
                                                           unsigned long long* gbase) {
  __shared__ unsigned long long wsum[kRadix / kWave];
  const int p = blockIdx.x, d = threadIdx.x;
  unsigned long long c = ghist[p * kRadix + d];
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
  static constexpr uint32_t kValueMask = (1u << 30) - 1u;   // n <= 2^30 (see host dispatch)
};
template <> struct Desc<uint64_t> {
  static constexpr int kShift = 62;
  static constexpr uint64_t kValueMask = (1ull << 62) - 1ull;
};
constexpr unsigned kCodeAgg = 1;

struct SweepArgs {
  const void* keys_in;
  void* keys_out;
  const void* vals_in;
  void* vals_out;
  void* desc;                       // [tiles][256] words of D
  unsigned* ticket;                 // one counter for this pass, zero at launch
  const unsigned long long* gbase;  // [256] for this pass
  unsigned* error_flag;             // set non-zero on a look-back timeout
  uint64_t n;
  uint64_t fmask_in, smask_in;      // non-zero only for the first pass
  uint64_t fmask_out, smask_out;    // non-zero only for the last pass
  int shift;
  unsigned mask;
  unsigned incl_code;               // 2 + (pass & 1)
};

// Peers of this lane: the lanes of the wave whose digit equals this lane's, as a 64-bit mask
// (8 ballots; each halves the candidate set).  `alive` removes padding lanes of a ragged tile.
__device__ __forceinline__ unsigned long long match_digit(unsigned d, unsigned long long alive) {
  unsigned long long m = alive;
#pragma unroll
  for (int b = 0; b < kRadixBits; ++b) {
    bool bit = (d >> b) & 1u;
    unsigned long long vote = __ballot(bit);
    m &= bit ? vote : ~vote;
  }
  return m;
}

__device__ __forceinline__ unsigned count_below(unsigned long long m) {   // popc(m & lanemask_lt)
  return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u));
}

template <typename D> __device__ __forceinline__ D desc_load(const D* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename D> __device__ __forceinline__ void desc_store(D* p, D v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename V> struct ValHolder { V v; };
struct NoVal {};

template <typename K, typename V, typename D, int BLOCK, int ITEMS>
struct SweepSmem {
  static constexpr int WAVES = BLOCK / kWave;
  static constexpr int TILE = BLOCK * ITEMS;
  static constexpr bool HAS_VAL = !__is_same(V, NoVal);
  static constexpr size_t kElem = HAS_VAL ? (sizeof(V) > sizeof(K) ? sizeof(V) : sizeof(K)) : sizeof(K);
  alignas(16) unsigned char stage[TILE * kElem];
  unsigned whist[WAVES][kRadix];
  unsigned tile_start[kRadix];
  D gofs[kRadix];
  unsigned wave_total[kRadix / kWave];
  unsigned tile_id;
};

template <typename K, typename V, typename D, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void onesweep_kernel(SweepArgs a) {
  using Smem = SweepSmem<K, V, D, BLOCK, ITEMS>;
  constexpr int WAVES = Smem::WAVES;
  constexpr int TILE = Smem::TILE;
  constexpr bool HAS_VAL = Smem::HAS_VAL;
  __shared__ Smem s;

  const unsigned tid = threadIdx.x;
  const unsigned lane = tid & (kWave - 1);
  const unsigned wave = tid / kWave;

  if (tid == 0) s.tile_id = atomicAdd(a.ticket, 1u);
#pragma unroll
  for (int i = 0; i < kRadix / kWave; ++i) s.whist[wave][i * kWave + lane] = 0;
  __syncthreads();

  const uint64_t tile = s.tile_id;
  const uint64_t tile_base = tile * TILE;
  const unsigned tile_count = (a.n - tile_base) < static_cast<uint64_t>(TILE)
                                  ? static_cast<unsigned>(a.n - tile_base) : static_cast<unsigned>(TILE);
  const bool full = tile_count == TILE;

  const K* kin = static_cast<const K*>(a.keys_in) + tile_base;
  const K fmi = static_cast<K>(a.fmask_in), smi = static_cast<K>(a.smask_in);

  // ---- load: wave-striped, so (item, lane) order == key order inside the wave's segment ----------
  K key[ITEMS];
  const unsigned seg = wave * (kWave * ITEMS) + lane;
  if (full) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) key[i] = xf_forward(kin[seg + i * kWave], fmi, smi);
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      unsigned idx = seg + i * kWave;
      key[i] = idx < tile_count ? xf_forward(kin[idx], fmi, smi) : static_cast<K>(~static_cast<K>(0));
    }
  }

  // ---- rank inside the wave -----------------------------------------------------------------------
  // After round i, whist[wave][d] = number of digit-d keys in rounds 0..i; rank = (count before
  // the round) + (peers in lower lanes).  Lane order inside a round and round order inside the
  // wave are both key order, so equal digits keep their input order (stability).
  unsigned rank[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    unsigned d = digit_at(key[i], a.shift, a.mask);
    bool ok = full || (seg + i * kWave) < tile_count;
    unsigned long long alive = full ? ~0ull : __ballot(ok);
    unsigned long long peers = match_digit(d, alive);
    unsigned below = count_below(peers);
    unsigned before = s.whist[wave][d];
    __builtin_amdgcn_wave_barrier();
    if (ok && below == 0) s.whist[wave][d] = before + static_cast<unsigned>(__popcll(peers));
    __builtin_amdgcn_wave_barrier();
    rank[i] = before + below;
  }
  __syncthreads();

  // ---- per-digit: counts over waves, publish the aggregate, scan over digits ---------------------
  D* desc = static_cast<D*>(a.desc);
  unsigned my_count = 0;
  unsigned inc = 0;
  if (tid < kRadix) {
    unsigned sum = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      unsigned c = s.whist[w][tid];
      s.whist[w][tid] = sum;          // becomes: digit-d keys in lower waves
      sum += c;
    }
    my_count = sum;
    D word = (static_cast<D>(tile == 0 ? a.incl_code : kCodeAgg) << Desc<D>::kShift) | static_cast<D>(sum);
    desc_store(&desc[tile * kRadix + tid], word);
    unsigned v = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      unsigned t = __shfl_up(v, off, kWave);
      if (lane >= static_cast<unsigned>(off)) v += t;
    }
    inc = v;
    if (lane == kWave - 1) s.wave_total[wave] = v;
  }
  __syncthreads();
  if (tid < kRadix) {
    unsigned before = 0;
    for (unsigned w = 0; w < wave; ++w) before += s.wave_total[w];
    s.tile_start[tid] = before + inc - my_count;
  }
  __syncthreads();

  // ---- stage keys in LDS in digit order -------------------------------------------------------------
  K* stage_k = reinterpret_cast<K*>(s.stage);
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    unsigned d = digit_at(key[i], a.shift, a.mask);
    unsigned pos = s.tile_start[d] + s.whist[wave][d] + rank[i];
    rank[i] = pos;
    if (full || (seg + i * kWave) < tile_count) stage_k[pos] = key[i];
  }

  // ---- decoupled look-back: one thread per digit -----------------------------------------------------
  if (tid < kRadix) {
    unsigned long long excl = 0;
    if (tile > 0) {
      uint64_t t = tile - 1;
      unsigned spins = 0;
      long long t0 = 0;
      bool failed = false;
      while (true) {
        D w = desc_load(&desc[t * kRadix + tid]);
        unsigned code = static_cast<unsigned>(w >> Desc<D>::kShift);
        if (code == a.incl_code) { excl += static_cast<unsigned long long>(w & Desc<D>::kValueMask); break; }
        if (code == kCodeAgg) { excl += static_cast<unsigned long long>(w & Desc<D>::kValueMask); --t; spins = 0; continue; }
        // predecessor has not published yet: back off; give up after ~2 s of wall clock so a
        // broken invariant ends the kernel instead of hanging the device.
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0) {
          long long now = static_cast<long long>(wall_clock64());
          if (t0 == 0) t0 = now;
          if (now - t0 > 200000000ll ||
              __hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            failed = true;
            break;
          }
        }
      }
      if (failed) __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      D word = (static_cast<D>(a.incl_code) << Desc<D>::kShift) |
               (static_cast<D>(excl + my_count) & Desc<D>::kValueMask);
      desc_store(&desc[tile * kRadix + tid], word);
    }
    s.gofs[tid] = static_cast<D>(a.gbase[tid] + excl - s.tile_start[tid]);
  }
  __syncthreads();

  // ---- write out: consecutive threads write consecutive addresses inside each digit run -------------
  K* kout = static_cast<K*>(a.keys_out);
  const K fmo = static_cast<K>(a.fmask_out), smo = static_cast<K>(a.smask_out);
  D dst[HAS_VAL ? ITEMS : 1];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    unsigned j = i * BLOCK + tid;
    if (full || j < tile_count) {
      K k = stage_k[j];
      D o = s.gofs[digit_at(k, a.shift, a.mask)] + j;
      kout[o] = xf_inverse(k, fmo, smo);
      if constexpr (HAS_VAL) dst[i] = o;
    }
  }

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
      if (full || j < tile_count) vout[dst[i]] = stage_v[j];
    }
  }
}

}  // namespace rsk
