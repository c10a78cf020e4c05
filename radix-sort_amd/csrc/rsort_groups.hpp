// rsort_groups.hpp -- the "finish in LDS" half of the hybrid sort of 4-byte keys (keys only).
//
// The LSD loop of the reference (sortLsd, radix_sort.hpp:506-526) moves every key once per digit: 4 reads + 4 writes for
// 4-byte keys.  Here the two TOP digits are sorted first by two onesweep passes (bits 16-23, then 24-31: stable, so the
// array is then grouped by its top 16 bits, 65536 groups of n / 65536 keys on uniform keys, each group in input order), and
// the two LOW digits are sorted by ONE kernel in which a workgroup takes a whole group into LDS, sorts it there with two
// 8-bit passes and writes it back in place: one read and one write of the keys instead of two of each (28 instead of the
// canonical 36 bytes per key for the whole sort).  Same result: the array ends up ordered by (top 16 bits, low 16 bits).
// Keys only: equal keys are bit-identical, so nothing here has to be stable with respect to the INPUT order, only the
// second LDS pass with respect to the first.
//
//   group_table_kernel   where each of the 65536 groups starts, from what the second sweep left behind (its look-back
//                        descriptors hold, per tile, the inclusive digit counts): no pass over the keys, no atomics
//   group_sort_kernel    one workgroup per group: load (16 bytes per lane), LDS pass on bits 0-7 (order-free: shared
//                        counters), LDS pass on bits 8-15 (stable: per-wave counters, as the sweep ranks), store (16 bytes
//                        per lane).  Groups that do not fit the LDS window are left alone and put on a list
//   group_big_kernel     the groups on that list, one workgroup each, by ranges of values through the alternate buffer
//   group_sort_wide_kernel  whole keys (and a 4-byte payload) of a group in LDS: 8-byte keys, the narrow and the pair forms;
//                        workgroups of 256 / 512 / 1024 threads by the size of the groups, a LISTED instantiation for the
//                        groups the small workgroups passed on (GroupArgs::mid_list), group_oversize for what fits none
//   group_max_kernel     the real size of the largest group, for the hosts' checks (pair forms; sorts of 5 * 10^8 keys and more)
#pragma once
#include "rsort_kernels.hpp"

namespace rsk {

// -DRS_GROUP_ABL=bits: timing experiments only (results wrong): 1 no counting adds, 2 no cursor adds in the first scatter,
// 4 no second pass, 8 no stores, 16 no loads
#ifndef RS_GROUP_ABL
#define RS_GROUP_ABL 0
#endif

struct GroupArgs {
  void* keys;                  // grouped by the bits above low_bits; each group is sorted in place
  void* vals;                  // payload that moves with the keys (pair form of the wide kernel only), else null
  const unsigned* start;       // [groups + 1] first element of each group; start[groups] = n
  unsigned groups;
  int low_bits;                // bits still to sort, 1..16 (values are staged as 16-bit words)
  uint64_t fmask_out, smask_out;   // inverse key transform, applied in the store
  uint64_t top_or;             // key bits above the group's number that every key shares (prefixed form; 16-bit staging kernel: put back with it), else 0
  unsigned* big_list;          // groups too large for the LDS window: [0] = how many, [1..] = group numbers
  unsigned big_cap;            // room in big_list (entries)
  unsigned* mid_list;          // not null: groups beyond THIS launch's window are listed here instead (same layout, room for every group) and
                               // a second launch, of the workgroups with the largest window, takes them from there (from_list)
  const unsigned* from_list;   // not null: the groups to sort are the listed ones, each workgroup every gridDim.x-th of them
  unsigned max_vec;            // groups whose window has more 16-byte vectors than this are left to the big-group kernel (<= BLOCK * MAXV)
  unsigned n;                  // keys in the array: a group that does not lie inside it is reported (error word bit 3) and left alone
  unsigned* error_flag;
};

// A group's range as the table gives it; false (and the error word set) when it cannot be a range of the array -- the table
// is derived from counts taken earlier, and keys that changed since (a caller that did not wait for their producer) make it
// inconsistent: nothing is read or written through such a range.
__device__ __forceinline__ bool group_range_ok(unsigned start, unsigned end, unsigned n, unsigned* error_flag) {
  if (start <= end && end <= n) return true;
  if (threadIdx.x == 0) __hip_atomic_fetch_or(error_flag, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// A group beyond the launch's window.  With a mid_list: listed there.  Else, keys only: a group whose keys are all the same
// key needs no sort (keys of a few thousand distinct values put whole values into single groups: without this the big-group
// kernel, meant for the odd oversize group, did most of such a sort) -- one read of the group, and one write when the keys
// are stored transformed; anything else goes on the big list.  Workgroup-uniform; every thread of the workgroup calls it.
template <typename K, bool KEYS_ONLY>
__device__ __forceinline__ void group_oversize(const GroupArgs& a, unsigned g, unsigned start, unsigned len) {
  if (a.mid_list != nullptr) {
    if (threadIdx.x == 0) a.mid_list[1u + atomicAdd(&a.mid_list[0], 1u)] = g;       // (room for every group)
    return;
  }
  if constexpr (KEYS_ONLY) {
    K* keys = static_cast<K*>(a.keys) + start;
    const K first = keys[0];
    bool differs = false;
    for (unsigned i = threadIdx.x; i < len; i += blockDim.x) differs |= keys[i] != first;
    if (!__syncthreads_or(differs)) {
      const K fmo = static_cast<K>(a.fmask_out), smo = static_cast<K>(a.smask_out);
      if (fmo | smo) {
        const K out = xf_inverse(first, fmo, smo);
        for (unsigned i = threadIdx.x; i < len; i += blockDim.x) keys[i] = out;
      }
      return;
    }
  }
  if (threadIdx.x == 0) {
    const unsigned slot = atomicAdd(&a.big_list[0], 1u);
    if (KEYS_ONLY && slot < a.big_cap) a.big_list[1 + slot] = g;
  }
}

template <int BLOCK, int MAXV>
struct GroupSmem {
  static constexpr int WAVES = BLOCK / kWave;
  static constexpr int CAP = BLOCK * MAXV * 4;          // elements of the largest window
  alignas(16) unsigned short val[CAP + 8];   // (+8: pads of a window that starts up to 3 elements before its group land behind CAP)
  alignas(16) unsigned hist[kRadix];         // pass 1: digit counts, then the cursors of the scatter
  unsigned whist[WAVES][kRadix]; // pass 2: per-wave counters, then start + prefix over the waves
  unsigned wave_total[kRadix / kWave];
};

// One group through LDS.  IT = 16-byte vectors per thread (compile time: the rounds of every phase are straight-line code).
// The window is the 16-byte aligned range of the array that covers the group; window elements outside the group and the
// elements of vectors beyond the window are "pads": they take the largest value, sort behind every real key and are never
// stored.  Real key of final rank j lands at window element head + j, so the store phase writes whole aligned vectors.
// Staging-area swizzle: value number p lives at element p ^ ((((p >> 6) ^ (p >> 10)) & 15) << 2), i.e. inside every aligned
// block of 64 elements (128 bytes) the 8-byte pieces are permuted by a hash of the block's number (bits 6-9 for runs of 64,
// the second pass of a ramp; bits 10-13 as well for runs of 256, its first pass).  On a ramp (nearly sorted input: a group is
// 16384 consecutive multiples of 4) every digit's run has the same power-of-two length, so the lanes of one scatter store
// -- same rank, neighbouring digits -- would all fall on one LDS bank (measured: 3.7 ms for the ramp against 2.4 ms for
// random keys at N = 2^30).  Aligned groups of four elements stay together, so the store phase still reads 8 bytes per lane.
__device__ __forceinline__ unsigned group_swizzle(unsigned p) { return p ^ ((((p >> 6) ^ (p >> 10)) & 15u) << 2); }

// Cell of digit d in the shared counter table of the first pass: d ^ (d >> 4).  A bijection on 0..255 that spreads digits
// a multiple of 16 apart (what the lanes of one instruction hold when the group is a ramp of step 4) over the banks.
__device__ __forceinline__ unsigned group_cell(unsigned d) { return d ^ (d >> 4); }

template <typename K, int BLOCK, int MAXV, int IT>
__device__ __forceinline__ void group_sort_body(GroupSmem<BLOCK, MAXV>& s, K* win, unsigned head, unsigned len, unsigned nvec,
                                                K top, int low_bits, K fmo, K smo) {
  static_assert(sizeof(K) == 4, "4-byte keys");
  constexpr int WAVES = BLOCK / kWave;
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const unsigned lowmask = (1u << low_bits) - 1u;
  const bool two = low_bits > kRadixBits;

  // ---- load: 16 bytes per lane, consecutive lanes consecutive vectors --------------------------------
  // Vector k of a thread is window vector k * BLOCK + tid.  With IT = ceil(nvec / BLOCK), the vectors of rounds 1 .. IT - 2
  // lie wholly inside the group; only round 0 (the window's first vector may begin before the group) and round IT - 1 (the
  // window ends inside it) hold elements that do not belong to the group.  Those are simply left out of the sort.
  const uint4* wv = reinterpret_cast<const uint4*>(win);
  uint4 x[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const unsigned v = k * BLOCK + tid;
    x[k] = uint4{0u, 0u, 0u, 0u};
    if (!(RS_GROUP_ABL & 16) && (k < IT - 1 || v < nvec)) x[k] = wv[v];
    if (RS_GROUP_ABL & 16) x[k] = uint4{v * 2654435761u, v * 40503u, v ^ 0x5bd1e995u, v * 7u};
  }
  if (tid < kRadix) s.hist[tid] = 0;
#pragma unroll
  for (int i = 0; i < kRadix / kWave; ++i) s.whist[wave][i * kWave + lane] = 0;
  __syncthreads();

  // ---- pass 1, bits 0-7: count (any order will do) ---------------------------------------------------
  // A lane takes the four keys of a vector in an order rotated by its row of 16 lanes: when the group is a ramp whose
  // low digit repeats every 16 lanes, the four lanes that would share a counter in one instruction then hold four different digits.
  const unsigned rot = (lane >> 4) & 3u;
  unsigned lo[2 * IT];          // two 16-bit values per register
  unsigned ok[IT];              // bit c: element c of the vector (in the rotated order) belongs to the group; rounds 0 and IT - 1 only
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const unsigned a = x[k].x & lowmask, b = x[k].y & lowmask, c = x[k].z & lowmask, d = x[k].w & lowmask;
    // rotate (a, b, c, d) left by rot
    const unsigned r0 = rot & 1u ? b : a, r1 = rot & 1u ? c : b, r2 = rot & 1u ? d : c, r3 = rot & 1u ? a : d;
    const unsigned a0 = rot & 2u ? r2 : r0, a1 = rot & 2u ? r3 : r1, a2 = rot & 2u ? r0 : r2, a3 = rot & 2u ? r1 : r3;
    lo[2 * k] = a0 | (a1 << 16);
    lo[2 * k + 1] = a2 | (a3 << 16);
    if (k == 0 || k == IT - 1) {
      const unsigned e = 4u * (k * BLOCK + tid) - head;     // rank of x[k].x inside the group if it belongs to it (wraps below head)
      unsigned m = 0;
#pragma unroll
      for (unsigned j = 0; j < 4; ++j) m |= (e + ((j + rot) & 3u) < len ? 1u : 0u) << j;
      ok[k] = m;
      if (!(RS_GROUP_ABL & 1)) {
        if (m & 1u) atomicAdd(&s.hist[group_cell(a0 & 255u)], 1u);
        if (m & 2u) atomicAdd(&s.hist[group_cell(a1 & 255u)], 1u);
        if (m & 4u) atomicAdd(&s.hist[group_cell(a2 & 255u)], 1u);
        if (m & 8u) atomicAdd(&s.hist[group_cell(a3 & 255u)], 1u);
      }
    } else if (!(RS_GROUP_ABL & 1)) {
      atomicAdd(&s.hist[group_cell(a0 & 255u)], 1u);
      atomicAdd(&s.hist[group_cell(a1 & 255u)], 1u);
      atomicAdd(&s.hist[group_cell(a2 & 255u)], 1u);
      atomicAdd(&s.hist[group_cell(a3 & 255u)], 1u);
    }
  }
  __syncthreads();
  // exclusive scan of the 256 counts in digit order by one wave, four digits per lane.  Digits 4 * lane .. 4 * lane + 3 share
  // their high nibble, so their cells are the aligned group of four at (4 * lane) ^ (high nibble & 12), permuted inside it by
  // the high nibble's low two bits.
  if (wave == 0) {
    const unsigned hn = lane >> 2;                         // high nibble of the lane's digits
    uint4 c = reinterpret_cast<const uint4*>(s.hist)[lane ^ (hn >> 2)];
    unsigned q[4] = {c.x, c.y, c.z, c.w};
    unsigned cnt[4];
#pragma unroll
    for (unsigned j = 0; j < 4; ++j) {                     // count of digit 4 * lane + j sits at component j ^ (hn & 3)
      const unsigned i = j ^ (hn & 3u);
      cnt[j] = i == 0 ? q[0] : i == 1 ? q[1] : i == 2 ? q[2] : q[3];
    }
    const unsigned sum = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    unsigned inc = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned u = __shfl_up(inc, off, kWave);
      if (lane >= static_cast<unsigned>(off)) inc += u;
    }
    unsigned b = inc - sum + (two ? 0u : head);            // (a single pass scatters straight to the store layout)
    unsigned o[4];
#pragma unroll
    for (unsigned j = 0; j < 4; ++j) { o[j] = b; b += cnt[j]; }
    unsigned w[4];
#pragma unroll
    for (unsigned i = 0; i < 4; ++i) {                     // component i holds digit 4 * lane + (i ^ (hn & 3))
      const unsigned j = i ^ (hn & 3u);
      w[i] = j == 0 ? o[0] : j == 1 ? o[1] : j == 2 ? o[2] : o[3];
    }
    reinterpret_cast<uint4*>(s.hist)[lane ^ (hn >> 2)] = uint4{w[0], w[1], w[2], w[3]};
  }
  __syncthreads();
  // scatter: a returning add on the digit's cursor hands out the slots of its run (eight keys at a time: registers)
  {
    auto scatter = [&](auto first_tag, auto count_tag, auto pred_tag) {
      constexpr int B0 = decltype(first_tag)::value, NB = decltype(count_tag)::value;
      constexpr bool PRED = decltype(pred_tag)::value;
      unsigned pos[2 * NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const unsigned m = PRED ? ok[(B0 + k) / 2] >> (2 * ((B0 + k) & 1)) : 3u;
        pos[2 * k] = pos[2 * k + 1] = 0;
        if (RS_GROUP_ABL & 2) { pos[2 * k] = 4u * ((B0 + k) / 2 * BLOCK + tid) + 2 * ((B0 + k) & 1); pos[2 * k + 1] = pos[2 * k] + 1; continue; }
        if (!PRED || (m & 1u)) pos[2 * k] = atomicAdd(&s.hist[group_cell(lo[B0 + k] & 255u)], 1u);
        if (!PRED || (m & 2u)) pos[2 * k + 1] = atomicAdd(&s.hist[group_cell((lo[B0 + k] >> 16) & 255u)], 1u);
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const unsigned m = PRED ? ok[(B0 + k) / 2] >> (2 * ((B0 + k) & 1)) : 3u;
        if (!PRED || (m & 1u)) s.val[group_swizzle(pos[2 * k])] = static_cast<unsigned short>(lo[B0 + k]);
        if (!PRED || (m & 2u)) s.val[group_swizzle(pos[2 * k + 1])] = static_cast<unsigned short>(lo[B0 + k] >> 16);
      }
    };
    // round 0, then the inner rounds in batches of two vectors, then round IT - 1
    scatter(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::true_type{});
    if constexpr (IT > 2) {
      constexpr int INNER = 2 * (IT - 2);                 // registers of lo[]
      auto inner = [&](auto self, auto at_tag) {
        constexpr int AT = decltype(at_tag)::value;
        if constexpr (AT < 2 + INNER) {
          constexpr int NB = 2 + INNER - AT >= 4 ? 4 : 2;
          scatter(std::integral_constant<int, AT>{}, std::integral_constant<int, NB>{}, std::false_type{});
          self(self, std::integral_constant<int, AT + NB>{});
        }
      };
      inner(inner, std::integral_constant<int, 2>{});
    }
    if constexpr (IT > 1) scatter(std::integral_constant<int, 2 * (IT - 1)>{}, std::integral_constant<int, 2>{}, std::true_type{});
  }
  __syncthreads();

  // ---- pass 2, bits 8-15: stable with respect to the order pass 1 left --------------------------------
  if (two && !(RS_GROUP_ABL & 4)) {                           // workgroup-uniform
    constexpr int R = 4 * IT;                                 // rounds of 64 per wave
    const unsigned segfirst = wave * (R * kWave);             // the wave's segment of the order pass 1 left: [segfirst, segfirst + R * 64)
    const unsigned segbase = segfirst + lane;
    unsigned kr[R];                                           // value | rank << 16
    const bool whole = segfirst + R * kWave <= len;           // (wave-uniform) every position of the segment holds a key
    if (whole) {
#pragma unroll
      for (int r = 0; r < R; ++r) kr[r] = s.val[group_swizzle(segbase + r * kWave)];
#pragma unroll
      for (int r = 0; r < R; ++r) kr[r] |= rank_add<true>(s.whist[wave], kr[r] >> 8) << 16;
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        kr[r] = 0;
        if (segbase + r * kWave < len) {
          kr[r] = s.val[group_swizzle(segbase + r * kWave)];
          kr[r] |= rank_add<true>(s.whist[wave], kr[r] >> 8) << 16;
        }
      }
    }
    __syncthreads();
    unsigned my_count = 0, inc = 0;
    if (tid < kRadix) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) my_count += s.whist[w][tid];
      unsigned v = my_count;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const unsigned u = __shfl_up(v, off, kWave);
        if (lane >= static_cast<unsigned>(off)) v += u;
      }
      inc = v;
      if (lane == kWave - 1) s.wave_total[wave] = v;
    }
    __syncthreads();
    if (tid < kRadix) {
      unsigned sum = head + inc - my_count;
      for (unsigned w = 0; w < wave; ++w) sum += s.wave_total[w];
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const unsigned c = s.whist[w][tid];
        s.whist[w][tid] = sum;
        sum += c;
      }
    }
    __syncthreads();
    if (whole) {
#pragma unroll
      for (int b = 0; b < R; b += 4) {
        unsigned pos[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pos[r] = s.whist[wave][(kr[b + r] >> 8) & 255u] + (kr[b + r] >> 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) s.val[group_swizzle(pos[r])] = static_cast<unsigned short>(kr[b + r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (segbase + r * kWave < len)
          s.val[group_swizzle(s.whist[wave][(kr[r] >> 8) & 255u] + (kr[r] >> 16))] = static_cast<unsigned short>(kr[r]);
    }
    __syncthreads();
  }

  // ---- store: whole aligned vectors; the first and the last vector of the window element by element ----
  const bool xf_out = (fmo | smo) != 0;
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const unsigned v = k * BLOCK + tid;
    if ((k < IT - 1 || v < nvec) && !(RS_GROUP_ABL & 8)) {
      const uint2 p = reinterpret_cast<const uint2*>(s.val)[v ^ (((v >> 4) ^ (v >> 8)) & 15u)];      // group_swizzle(4 * v) / 4
      K o0 = top | (p.x & 0xFFFFu), o1 = top | (p.x >> 16), o2 = top | (p.y & 0xFFFFu), o3 = top | (p.y >> 16);
      if (xf_out) { o0 = xf_inverse(o0, fmo, smo); o1 = xf_inverse(o1, fmo, smo); o2 = xf_inverse(o2, fmo, smo); o3 = xf_inverse(o3, fmo, smo); }
      const unsigned e = 4u * v - head;
      if ((k > 0 && k < IT - 1) || (4u * v >= head && e + 3u < len)) {
        reinterpret_cast<uint4*>(win)[v] = uint4{o0, o1, o2, o3};
      } else {
        if (e < len) win[4u * v] = o0;
        if (e + 1u < len) win[4u * v + 1u] = o1;
        if (e + 2u < len) win[4u * v + 2u] = o2;
        if (e + 3u < len) win[4u * v + 3u] = o3;
      }
    }
  }
}

template <typename K, int BLOCK, int MAXV>
__device__ __forceinline__ void group_sort_one(GroupSmem<BLOCK, MAXV>& s, const GroupArgs& a, unsigned g) {
  const unsigned start = a.start[g], end = a.start[g + 1];
  if (!group_range_ok(start, end, a.n, a.error_flag)) return;
  if (end <= start + 1u) {                      // empty, or one key: in place already -- but the store applies the transform
    if (end == start || !(a.fmask_out | a.smask_out)) return;
  }
  const unsigned len = end - start;
  K* keys = static_cast<K*>(a.keys);
  const unsigned head = static_cast<unsigned>((reinterpret_cast<uintptr_t>(keys + start) & 15u) / sizeof(K));
  K* win = keys + start - head;
  const unsigned nvec = (head + len + 3u) / 4u;
  if (nvec > a.max_vec) {   // workgroup-uniform: does not fit this window
    group_oversize<K, true>(a, g, start, len);
    return;
  }
  const K top = (static_cast<K>(g) << a.low_bits) | static_cast<K>(a.top_or);
  const K fmo = static_cast<K>(a.fmask_out), smo = static_cast<K>(a.smask_out);
  const unsigned it = (nvec + BLOCK - 1) / BLOCK;
  switch (it) {      // (one instantiation per vector count: a group of 16384 +- 130 keys takes 8 or 9 of 512 threads' vectors)
#define RS_GROUP_CASE(N) case N: if constexpr (N <= MAXV) group_sort_body<K, BLOCK, MAXV, N>(s, win, head, len, nvec, top, a.low_bits, fmo, smo); break;
    RS_GROUP_CASE(1) RS_GROUP_CASE(2) RS_GROUP_CASE(3) RS_GROUP_CASE(4) RS_GROUP_CASE(5) RS_GROUP_CASE(6)
    RS_GROUP_CASE(7) RS_GROUP_CASE(8) RS_GROUP_CASE(9) RS_GROUP_CASE(10) RS_GROUP_CASE(11) RS_GROUP_CASE(12)
#undef RS_GROUP_CASE
    default: break;
  }
}

// (LISTED: the groups of a.from_list, every gridDim.x-th one, instead of group blockIdx.x -- an instantiation of its own, see
// group_sort_wide_kernel)
template <typename K, int BLOCK, int MAXV, int MIN_BLOCKS, bool LISTED = false>
__global__ __launch_bounds__(BLOCK, (MIN_BLOCKS * BLOCK + 255) / 256) void group_sort_kernel(GroupArgs a) {
  __shared__ GroupSmem<BLOCK, MAXV> s;
  if constexpr (!LISTED) {
    group_sort_one<K, BLOCK, MAXV>(s, a, blockIdx.x);
  } else {
    const unsigned listed = min(a.from_list[0], a.groups);
    for (unsigned i = blockIdx.x; i < listed; i += gridDim.x) {
      const unsigned g = a.from_list[1u + i];
      if (g < a.groups) group_sort_one<K, BLOCK, MAXV>(s, a, g);
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Groups of wider keys, or of more low bits
// ------------------------------------------------------------------------------------------------
// The same finish for keys whose low part does not fit 16 bits -- 8-byte keys after the two sweeps on their top digits
// (48 low bits), or 4-byte keys grouped by one top digit only (24 low bits; mid-size sorts): one workgroup takes a group's
// whole keys into LDS and runs the single-workgroup sort's loop over the low bits (tile_sort_body, rsort_kernels.hpp:
// stable LDS passes of 8 bits, digits that are the same for the whole group skipped), one read and one write of the keys
// whatever the number of passes.  8-byte keys at N = 2^30: 8 + 2 * 16 + 16 = 56 bytes per key instead of the canonical 136.
// tile_sort_body for a group of len keys, instantiated for the smallest of 1, 2, 4, 8, 12, 16, ITEMS keys per thread that holds it
// TOP: where the groups' prefix passes start (tile_sort_body): a constant of the calling kernel, so that the fix-up's compares
// fold to one word; 0 = passes over all the low bits.  low_bits: the bits the group is sorted on (run time: the narrow form
// leaves 48 to 55 of them, or 16 to 23 of a 4-byte key).
#ifndef RS_FIX_PREFIX_BITS
#define RS_FIX_PREFIX_BITS 16     // the passes cover at least the top 16 of the low bits, the fix-up the rest (0: passes over all the low bits)
#endif
template <typename K, typename V, int BLOCK, int ITEMS, int TOP, typename Smem>
__device__ __forceinline__ void group_tile_sort(Smem& s, const K* gin, K* gout, V* gv, unsigned len, K fmo, K smo, int low_bits) {
  const unsigned per = (len + BLOCK - 1) / BLOCK;        // (workgroup-uniform)
  auto run = [&](auto it_tag) {
    constexpr int IT = decltype(it_tag)::value < ITEMS ? decltype(it_tag)::value : ITEMS;
    tile_sort_body<K, V, BLOCK, IT, TOP>(s, gin, gout, gv, len, static_cast<K>(0), static_cast<K>(0), fmo, smo, 0, low_bits);
  };
  if (per <= 1) run(std::integral_constant<int, 1>{});
  else if (per <= 2) run(std::integral_constant<int, 2>{});
  else if (per <= 4) run(std::integral_constant<int, 4>{});
  else if (per <= 8) run(std::integral_constant<int, 8>{});
  else if (per <= 12) run(std::integral_constant<int, 12>{});
  else if (per <= 16) run(std::integral_constant<int, 16>{});
  else run(std::integral_constant<int, ITEMS>{});
}
// (measured, 16384 keys per group: the fix-up costs what one and a half passes cost, so it pays from five passes on -- 8-byte
// keys: 48 low bits 41.9 -> 29.2 us per group; the low bits of 4-byte keys are two or three passes and stay that)
template <typename K> constexpr int group_fix_top(int low) {
  return RS_FIX_PREFIX_BITS > 0 && sizeof(K) == 8 && low >= RS_FIX_PREFIX_BITS + 24 ? low - RS_FIX_PREFIX_BITS : 0;
}

// V = NoVal: keys only.  V = a 4-byte payload (a.vals): the stable pair form -- every pass and the fix-up keep equal keys in
// their order, the payload rides through the staging area beside its key; groups beyond the (smaller) window are only
// counted in big_list[0] and the host sorts the whole array again, every pass (the big-group kernel is keys only).
// Pairs of 4-byte keys run it in workgroups of 512 or 256 threads x 17 pairs (two or four to a CU: two plain passes over
// the 16 low bits, and another workgroup's passes to hide the load and the store behind).
// TOP_FROM: where the groups' passes start when it is not the default (bit 32 for 8-byte keys).  The prefixed form leaves
// 41 to 47 low bits: from bit 32 that is a prefix of 9 to 15 bits and runs of two or three keys for the fix-up (11.1 ms at
// N = 2^30); from bit 24, one pass more and runs of one key.
// (four waves per SIMD whatever the workgroup's size -- one workgroup of 1024 threads, two of 512, four of 256 per CU: left to
// itself the compiler takes 141-144 registers for the smaller ones and only one or three of them fit)
template <typename K, typename V, int BLOCK, int ITEMS, int TOP_FROM = -1>
__device__ __forceinline__ void group_sort_wide_one(TileSortSmem<K, V, BLOCK, ITEMS>& s, const GroupArgs& a, unsigned g) {
  constexpr bool HAS_VAL = !__is_same(V, NoVal);
  const unsigned start = a.start[g], end = a.start[g + 1];
  if (!group_range_ok(start, end, a.n, a.error_flag)) return;
  if (end <= start + 1u) {
    if (end == start || !(a.fmask_out | a.smask_out)) return;
  }
  const unsigned len = end - start;
  if (len > a.max_vec) {          // (here max_vec counts keys) workgroup-uniform: does not fit this window
    group_oversize<K, !HAS_VAL>(a, g, start, len);
    return;
  }
  V* gv = nullptr;
  if constexpr (HAS_VAL) gv = static_cast<V*>(a.vals) + start;
  // (8-byte keys: passes from bit 32 up -- 16 bits of prefix for the 48 low bits of the 16-bit-group form, up to 23 for the
  // narrow form's 55 -- then the fix-up; 4-byte keys: passes over all the low bits)
  group_tile_sort<K, V, BLOCK, ITEMS, (TOP_FROM >= 0 ? TOP_FROM : group_fix_top<K>((static_cast<int>(sizeof(K)) - 2) * 8))>(s, static_cast<const K*>(a.keys) + start, static_cast<K*>(a.keys) + start, gv, len,
                                                                                               static_cast<K>(a.fmask_out), static_cast<K>(a.smask_out), a.low_bits);
}
// LISTED: the groups of a.from_list, every gridDim.x-th one, instead of group blockIdx.x.  An instantiation of its own: with
// the loop around it the body spills (256 bytes of scratch per lane against 16, and the 2^30-key sort loses 7 % of its group
// kernel); the listed groups are the few that the small workgroups could not hold.
template <typename K, typename V, int BLOCK, int ITEMS, int TOP_FROM = -1, bool LISTED = false>
__global__ __launch_bounds__(BLOCK, 4) void group_sort_wide_kernel(GroupArgs a) {
  __shared__ TileSortSmem<K, V, BLOCK, ITEMS> s;
  if constexpr (!LISTED) {
    group_sort_wide_one<K, V, BLOCK, ITEMS, TOP_FROM>(s, a, blockIdx.x);
  } else {
    const unsigned listed = min(a.from_list[0], a.groups);
    for (unsigned i = blockIdx.x; i < listed; i += gridDim.x) {
      const unsigned g = a.from_list[1u + i];
      if (g < a.groups) group_sort_wide_one<K, V, BLOCK, ITEMS, TOP_FROM>(s, a, g);
      __syncthreads();
    }
  }
}

// The largest group of a start table, and how many keys lie in groups beyond `window`, asked before anything is sorted:
// out[0] = max, out[1] = keys in oversize groups (both must be zero).  The host's bound on the largest group comes from digit
// counts and assumes independent digits; keys of a few thousand distinct values break it.
__global__ __launch_bounds__(kRadix) void group_max_kernel(const unsigned* start, unsigned* out, unsigned groups, unsigned window) {
  const unsigned g = blockIdx.x * kRadix + threadIdx.x;
  unsigned len = 0;
  if (g < groups) {
    const unsigned lo = start[g], hi = start[g + 1];
    len = hi >= lo ? hi - lo : 0xFFFFFFFFu;              // (an inconsistent table: larger than any window)
  }
  unsigned over = len > window ? len : 0u;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    len = max(len, static_cast<unsigned>(__shfl_down(len, off, kWave)));
    const unsigned o = static_cast<unsigned>(__shfl_down(over, off, kWave));
    over = over + o < over ? 0xFFFFFFFFu : over + o;
  }
  if ((threadIdx.x & (kWave - 1)) == 0) {
    atomicMax(&out[0], len);
    if (over) atomicAdd(&out[1], over);                  // (the sum of all groups is n < 2^32; an inconsistent table saturates one wave's share, which is then over any threshold)
  }
}

// Mid-size sorts (at most 256 tiles; SweepArgs::mid_counts has the protocol): after ONE sweep on the top digit, workgroup g
// sorts the keys whose top digit is g on their remaining bits, from the sweep's output buffer into the caller's.  No table
// kernel: the 256 counts ARE the group sizes, and every workgroup sums the ones before its own.  Leaves at once when some
// group does not fit (then the LSD passes queued beside it are doing the sort).
struct MidArgs {
  const void* keys_in;                   // the top-digit sweep's output
  void* keys_out;                        // the caller's array
  const unsigned long long* counts;      // the top digit's 256 counts
  unsigned cap;                          // <= BLOCK * ITEMS
  int low_bits;
  uint64_t fmask_out, smask_out;
};

template <typename K, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void group_sort_mid_kernel(MidArgs a) {
  __shared__ TileSortSmem<K, NoVal, BLOCK, ITEMS> s;
  __shared__ unsigned long long part[kRadix / kWave];
  const unsigned g = blockIdx.x, tid = threadIdx.x;
  const unsigned long long c = tid < kRadix ? a.counts[tid] : 0ull;
  if (__syncthreads_or(c > a.cap)) return;              // (kernel-uniform)
  if (tid < kRadix) {
    unsigned long long v = tid < g ? c : 0ull;          // keys of the groups before this one
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    if ((tid & (kWave - 1)) == 0) part[tid / kWave] = v;
  }
  __syncthreads();
  const unsigned start = static_cast<unsigned>(part[0] + part[1] + part[2] + part[3]);
  const unsigned len = static_cast<unsigned>(a.counts[g]);
  if (len == 0) return;
  group_tile_sort<K, NoVal, BLOCK, ITEMS, group_fix_top<K>((static_cast<int>(sizeof(K)) - 1) * 8)>(s, static_cast<const K*>(a.keys_in) + start, static_cast<K*>(a.keys_out) + start,
                                                                                                   static_cast<NoVal*>(nullptr), len, static_cast<K>(a.fmask_out), static_cast<K>(a.smask_out), a.low_bits);
}

// ------------------------------------------------------------------------------------------------
// Where the groups start
// ------------------------------------------------------------------------------------------------
// After the sweep on the upper top digit (hi) of an array that the sweep before it had ordered by the lower top digit (lo),
// group (hi, lo) starts at
//     gbase_hi[hi] + #{keys with upper digit hi among the sweep's input positions [0, gbase_lo[lo])},
// because a stable pass hands out a digit's output slots in input order.  The count up to an input position is what the
// sweep's look-back descriptors hold for whole tiles (every tile's word ends as the inclusive count of its digit over the
// tiles of its chain up to itself), plus the counts of the chains before, plus a partial tile that is counted here from the
// sweep's input, which is still intact in the other buffer.  One workgroup per value of the lower digit (the boundary it
// looks at), one thread per value of the upper digit; at most one tile of keys is read per workgroup.
struct TableArgs {
  const uint32_t* desc;                  // [tiles][256] 4-byte descriptors left by the sweep on the upper digit
  const void* keys_in;                   // that sweep's input (keys of K bytes, transformed already)
  const unsigned long long* gbase_lo;    // [256] where each value of the lower digit starts in keys_in
  const unsigned long long* gbase_hi;    // [256] where each value of the upper digit starts in the sweep's output
  const unsigned long long* chain_cnt;   // chain mode: the upper digit's counts over each chain's input range, row c at c * chain_cnt_stride
  unsigned chain_cnt_stride;
  unsigned chain_tiles;                  // tiles per chain; 0 = one chain over all tiles
  unsigned tile;                         // keys per tile of that sweep
  unsigned n;
  int shift_hi;                          // bit position of the upper digit
  unsigned* start;                       // out: [65537], group g = hi * 256 + lo; start[65536] = n
};

template <typename K>
__global__ __launch_bounds__(kRadix) void group_table_kernel(TableArgs a) {
  __shared__ unsigned part[kRadix];
  const unsigned lo = blockIdx.x, hi = threadIdx.x;
  const unsigned long long pos64 = a.gbase_lo[lo];
  part[hi] = 0;
  __syncthreads();
  unsigned result;
  if (pos64 >= a.n) {                                   // (workgroup-uniform) no key at or behind this boundary: the group is empty
    result = hi == kRadix - 1 ? a.n : static_cast<unsigned>(a.gbase_hi[hi + 1]);
  } else {
    const unsigned pos = static_cast<unsigned>(pos64);
    const unsigned t = pos / a.tile, r = pos - t * a.tile;
    const K* kin = static_cast<const K*>(a.keys_in) + static_cast<size_t>(t) * a.tile;
    for (unsigned i = hi; i < r; i += kRadix) atomicAdd(&part[static_cast<unsigned>(kin[i] >> a.shift_hi) & 0xFFu], 1u);
    __syncthreads();
    const unsigned chain = a.chain_tiles ? t / a.chain_tiles : 0u;
    const unsigned first = chain * a.chain_tiles;
    unsigned long long c = a.gbase_hi[hi] + part[hi];
    for (unsigned x = 0; x < chain; ++x) c += a.chain_cnt[static_cast<size_t>(x) * a.chain_cnt_stride + hi];
    if (t > first) c += a.desc[static_cast<size_t>(t - 1) * kRadix + hi] & Desc<uint32_t>::kValueMask;
    result = static_cast<unsigned>(c);
  }
  a.start[hi * kRadix + lo] = result;
  if (lo == 0 && hi == 0) a.start[kRadix * kRadix] = a.n;
}

// ------------------------------------------------------------------------------------------------
// Groups that do not fit the LDS window
// ------------------------------------------------------------------------------------------------
// One workgroup per listed group: an LSD sort of the group's low bits through global memory, tile after tile (the tile
// part of the sweep: rank inside the wave by returning adds, prefix over the waves, scatter; the running digit offsets
// stay in LDS, so there is no look-back).  Passes go keys -> alt -> keys; an odd number of passes ends with a copy back.
// What one pass wrote is read by the next through L1-bypassing loads (another workgroup on this CU may have cached a line
// that this group shares with its neighbour before this workgroup wrote its part of it).
struct BigArgs {
  void* keys;
  void* alt;
  const unsigned* start;
  const unsigned* big_list;      // [0] = count, [1..] = group numbers
  unsigned big_cap;
  int low_bits;
  uint64_t fmask_out, smask_out;
  unsigned n;
  unsigned* error_flag;
};

template <typename K, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void group_big_kernel(BigArgs a) {
  constexpr int WAVES = BLOCK / kWave;
  constexpr unsigned TILE = BLOCK * ITEMS;
  __shared__ alignas(16) unsigned base[kRadix];
  __shared__ unsigned whist[WAVES][kRadix];
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const K fmo = static_cast<K>(a.fmask_out), smo = static_cast<K>(a.smask_out);
  const unsigned listed = min(a.big_list[0], a.big_cap);
  const int passes = (a.low_bits + kRadixBits - 1) / kRadixBits;
  auto load = [](const K* p, bool coherent) -> K {
    return coherent ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
  };
  for (unsigned e = blockIdx.x; e < listed; e += gridDim.x) {
    const unsigned g = a.big_list[1 + e];
    if (g >= 65536u || !group_range_ok(a.start[g], a.start[g + 1], a.n, a.error_flag)) continue;      // (workgroup-uniform)
    const unsigned gs = a.start[g], len = a.start[g + 1] - gs;
    K* src = static_cast<K*>(a.keys) + gs;
    K* dst = static_cast<K*>(a.alt) + gs;
    for (int p = 0; p < passes; ++p) {
      const int shift = p * kRadixBits;
      const int bits = a.low_bits - shift < kRadixBits ? a.low_bits - shift : kRadixBits;
      const bool coherent = p > 0;
      const bool last = p == passes - 1 && (passes & 1) == 0;      // this pass writes the final place
      if (tid < kRadix) base[tid] = 0;
      __syncthreads();
      for (unsigned i = tid; i < len; i += BLOCK) atomicAdd(&base[digit_at(load(src + i, coherent), shift, bits)], 1u);
      __syncthreads();
      if (wave == 0) {                        // exclusive scan, four digits per lane
        const uint4 c = reinterpret_cast<const uint4*>(base)[lane];
        const unsigned sum = c.x + c.y + c.z + c.w;
        unsigned inc = sum;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
          const unsigned u = __shfl_up(inc, off, kWave);
          if (lane >= static_cast<unsigned>(off)) inc += u;
        }
        unsigned b = inc - sum;
        uint4 o;
        o.x = b; b += c.x; o.y = b; b += c.y; o.z = b; b += c.z; o.w = b;
        reinterpret_cast<uint4*>(base)[lane] = o;
      }
      __syncthreads();
      for (unsigned t0 = 0; t0 < len; t0 += TILE) {
#pragma unroll
        for (int i = 0; i < kRadix / kWave; ++i) whist[wave][i * kWave + lane] = 0;
        K key[ITEMS];
        unsigned rank[ITEMS];
        const unsigned seg = t0 + wave * (kWave * ITEMS) + lane;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
          const unsigned idx = seg + i * kWave;
          key[i] = idx < len ? load(src + idx, coherent) : static_cast<K>(0);
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
          rank[i] = 0;
          if (seg + i * kWave < len) rank[i] = rank_add<true>(whist[wave], digit_at(key[i], shift, bits));
        }
        __syncthreads();
        if (tid < kRadix) {
          unsigned sum = base[tid];
#pragma unroll
          for (int w = 0; w < WAVES; ++w) {
            const unsigned c = whist[w][tid];
            whist[w][tid] = sum;
            sum += c;
          }
          base[tid] = sum;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
          if (seg + i * kWave < len) {
            const unsigned o = whist[wave][digit_at(key[i], shift, bits)] + rank[i];
            if (o < len) dst[o] = last ? xf_inverse(key[i], fmo, smo) : key[i];
          }
        }
        __syncthreads();
      }
      __threadfence();
      __syncthreads();
      K* t = src; src = dst; dst = t;
    }
    if (passes & 1) {                          // the result sits in alt: back to keys (src and dst have swapped roles)
      for (unsigned i = tid; i < len; i += BLOCK) dst[i] = xf_inverse(load(src + i, true), fmo, smo);
    }
    __syncthreads();
  }
}

}  // namespace rsk

// ------------------------------------------------------------------------------------------------
// Few distinct keys (keys only, whole key)
// ------------------------------------------------------------------------------------------------
// An array that holds only a handful of different keys -- the reference harness' "duplicates" shape draws every key from 256
// values (generators.hpp:93-107) -- is sorted once its distinct keys and their counts are known: the output is each key,
// in order, as often as it occurs.  One read of the keys (counting them in a small hash table per workgroup, merged into
// a global one) and one write (filling the runs) replace the K passes of the LSD loop: 2 * K bytes per key instead of
// K * (1 + 2 * K).  Keys only: equal keys are bit-identical, so there is no order among them to keep.
//   distinct_collect_kernel   probe (a sample, one workgroup: how many distinct keys does it see?) and full count
//   fill_runs_kernel          writes key i of the sorted distinct keys to positions [starts[i], starts[i + 1])
// The table holds at most kDistinctMax keys; an array with more sets the overflow word and is sorted the usual way.
namespace rsd {
constexpr unsigned kDistinctMax = 1024;          // distinct keys this path serves
constexpr unsigned kLocalSlots = 4096;           // per-workgroup hash table: open addressing from the first slot of the key's bucket of four on, at most a quarter full
constexpr unsigned kGlobalSlots = 4096;          // the merged table
constexpr unsigned kCountCols = 2;               // counters per slot: lane & 1 picks one, so equal keys in one wave queue on two words, not one

struct CollectArgs {
  const void* keys;
  uint64_t n;                 // keys to look at
  void* gkeys;                // [kGlobalSlots] keys of the merged table; empty = all ones
  unsigned long long* gcnt;   // [kGlobalSlots]
  unsigned long long* ones;   // occurrences of the key that is all ones (it is the table's "empty" mark)
  unsigned* used;             // [0] distinct keys in the merged table, [1] overflow (more distinct keys than the path serves)
  int probe;                  // 1: one workgroup's own table only; used[0] = distinct keys it saw (kDistinctMax + 1: too many)
};

// Multiplicative hash, top `bits` bits: one 32-bit multiply per key half (a full 64-bit product costs four).
template <typename K> __device__ __forceinline__ unsigned distinct_hash(K k, unsigned bits) {
  unsigned h = static_cast<unsigned>(k) * 0x9E3779B1u;
  if constexpr (sizeof(K) == 8) h ^= static_cast<unsigned>(static_cast<unsigned long long>(k) >> 32) * 0x85EBCA77u + 0x27D4EB2Fu;
  return h >> (32u - bits);
}
constexpr unsigned kLocalBits = 12, kGlobalBits = 12;
static_assert((1u << kLocalBits) == kLocalSlots && (1u << kGlobalBits) == kGlobalSlots, "table sizes");

template <typename K, int BLOCK>
__global__ __launch_bounds__(BLOCK) void distinct_collect_kernel(CollectArgs a) {
  constexpr K kEmpty = static_cast<K>(~static_cast<K>(0));
  constexpr int VEC = 16 / sizeof(K);
  __shared__ alignas(16) K lkeys[kLocalSlots];
  __shared__ unsigned lcnt[kLocalSlots * kCountCols];
  __shared__ unsigned lused, lover, lones;
  struct alignas(16) Bucket { K s[4]; };
  const unsigned tid = threadIdx.x, col = tid & (kCountCols - 1u);
  for (unsigned i = tid; i < kLocalSlots; i += BLOCK) lkeys[i] = kEmpty;
  for (unsigned i = tid; i < kLocalSlots * kCountCols; i += BLOCK) lcnt[i] = 0;
  if (tid == 0) { lused = 0; lover = 0; lones = 0; }
  __syncthreads();

  // the general case: probe from the first slot of the key's bucket on, claim the first empty one
  auto add_key = [&](K k) {
    if (k == kEmpty) { atomicAdd(&lones, 1u); return; }
    unsigned slot = distinct_hash(k, kLocalBits - 2u) * 4u;
#pragma nounroll
    for (unsigned probes = 0; probes < kLocalSlots; ++probes) {
      K cur = __hip_atomic_load(&lkeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (cur == kEmpty) {
        if (__hip_atomic_load(&lused, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= kDistinctMax) { __hip_atomic_store(&lover, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return; }
        K expected = kEmpty;
        if (__hip_atomic_compare_exchange_strong(&lkeys[slot], &expected, k, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
          atomicAdd(&lused, 1u);
          cur = k;
        } else {
          cur = expected;
        }
      }
      if (cur == k) { atomicAdd(&lcnt[slot * kCountCols + col], 1u); return; }
      slot = (slot + 1u) & (kLocalSlots - 1u);
    }
    __hip_atomic_store(&lover, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  const K* keys = static_cast<const K*>(a.keys);
  // 16-byte packs from the first 16-byte boundary on; the keys before it and after the last whole pack one by one (workgroup 0)
  uint64_t head = ((16u - (reinterpret_cast<uintptr_t>(keys) & 15u)) & 15u) / sizeof(K);
  if (head > a.n) head = a.n;
  const uint64_t nvec = (a.n - head) / VEC;
  struct alignas(16) Pack { K v[VEC]; };
  const Pack* packs = reinterpret_cast<const Pack*>(keys + head);
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * BLOCK;
  // two packs per turn, each fetched while the other is counted
  auto count_pack = [&](const Pack& pk) -> bool {
    // the usual case, straight-line: the key sits in one of the four slots of its bucket -- one 16-byte read per key (two
    // for 8-byte keys), all the look-ups of the pack in flight together (13 instructions per key: at 2^30 keys the kernel
    // has about 29 per 4-byte key before it, not the memory, sets the time).  Anything else -- first sight of a key, a
    // bucket that has overflowed into the next, the all-ones key -- takes the general path.  A read that races with another
    // wave's insert sees the slot empty or taken, both of which the general path handles.
    unsigned bucket[VEC];
    Bucket cur[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) bucket[j] = distinct_hash(pk.v[j], kLocalBits - 2u);
    __asm__ volatile("" ::: "memory");                     // (the table changes under the loop: read it again every time round)
#pragma unroll
    for (int j = 0; j < VEC; ++j) cur[j] = *reinterpret_cast<const Bucket*>(&lkeys[bucket[j] * 4u]);
    const unsigned over = __hip_atomic_load(&lover, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned slow = 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      unsigned idx = 4;
#pragma unroll
      for (int t = 3; t >= 0; --t) idx = cur[j].s[t] == pk.v[j] ? static_cast<unsigned>(t) : idx;
      if (idx < 4u && pk.v[j] != kEmpty) atomicAdd(&lcnt[(bucket[j] * 4u + idx) * kCountCols + col], 1u);
      else slow |= 1u << j;
    }
    if (over) return true;                                 // too many distinct keys: nothing more to learn
    if (slow) {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if ((slow >> j) & 1u) add_key(pk.v[j]);
    }
    return false;
  };
  uint64_t i = static_cast<uint64_t>(blockIdx.x) * BLOCK + tid;
  Pack pa{}, pb{};
  if (i < nvec) pa = packs[i];
  while (i < nvec) {
    if (i + stride < nvec) pb = packs[i + stride];
    if (count_pack(pa)) break;
    if (i + stride >= nvec) break;
    if (i + 2 * stride < nvec) pa = packs[i + 2 * stride];
    if (count_pack(pb)) break;
    i += 2 * stride;
  }
  if (blockIdx.x == 0) {
    if (tid < head) add_key(keys[tid]);
    const uint64_t t = head + nvec * VEC + tid;
    if (t < a.n) add_key(keys[t]);
  }
  __syncthreads();
  if (a.probe) {
    if (tid == 0) a.used[0] = lover ? kDistinctMax + 1u : lused;
    return;
  }
  if (lover) {                                             // (workgroup-uniform)
    if (tid == 0) a.used[1] = 1u;
    return;
  }
  if (tid == 0 && lones) atomicAdd(a.ones, static_cast<unsigned long long>(lones));
  K* gkeys = static_cast<K*>(a.gkeys);
  for (unsigned i2 = tid; i2 < kLocalSlots; i2 += BLOCK) {
    const K k = lkeys[i2];
    if (k == kEmpty) continue;
    unsigned long long c = 0;
#pragma unroll
    for (unsigned j = 0; j < kCountCols; ++j) c += lcnt[i2 * kCountCols + j];
    unsigned slot = distinct_hash(k, kGlobalBits);
    bool placed = false;
    for (unsigned probes = 0; probes < kGlobalSlots && !placed; ++probes) {
      K cur = __hip_atomic_load(&gkeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == kEmpty) {
        K expected = kEmpty;
        if (__hip_atomic_compare_exchange_strong(&gkeys[slot], &expected, k, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          if (atomicAdd(&a.used[0], 1u) >= kDistinctMax) a.used[1] = 1u;
          cur = k;
        } else {
          cur = expected;
        }
      }
      if (cur == k) { atomicAdd(&a.gcnt[slot], c); placed = true; }
      slot = (slot + 1u) & (kGlobalSlots - 1u);
    }
    if (!placed) a.used[1] = 1u;
  }
}

// Which bits differ among the keys?  AND and OR over 4096 keys spread evenly over the array (one workgroup, microseconds):
// a hint for the narrow hybrid form, which is pointless when the bits it would sweep are the same for every key (small
// integers in 8-byte keys).  out[0] must be all ones, out[1] zero.
// out[2] (must be zero): how many of the sampled keys equal a sampled key before them -- S^2 / 2D on average for keys drawn
// from D values, 0.002 for random 4-byte keys: the only sign, before anything is sorted, of keys that take thousands to
// millions of distinct values (a value's copies all fall into one group of the hybrid forms).
template <typename K>
__global__ __launch_bounds__(1024) void bit_span_kernel(const K* keys, uint64_t n, unsigned long long* out) {
  constexpr unsigned kSlots = 8192;
  __shared__ unsigned long long seen[kSlots];
  for (unsigned i = threadIdx.x; i < kSlots; i += 1024u) seen[i] = ~0ull;
  __syncthreads();
  const uint64_t stride = n / 4096u ? n / 4096u : 1u;
  unsigned long long a = ~0ull, o = 0ull;
  unsigned dups = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t i = (static_cast<uint64_t>(j) * 1024u + threadIdx.x) * stride;
    if (i < n) {
      const unsigned long long k = static_cast<unsigned long long>(keys[i]);
      a &= k; o |= k;
      if (k != ~0ull) {                                      // (the table's "empty" mark)
        unsigned slot = static_cast<unsigned>((k * 0x9E3779B97F4A7C15ull) >> 51);
        for (unsigned probes = 0; probes < kSlots; ++probes) {
          const unsigned long long old = atomicCAS(&seen[slot], ~0ull, k);
          if (old == ~0ull) break;
          if (old == k) { ++dups; break; }
          slot = (slot + 1u) & (kSlots - 1u);
        }
      }
    }
  }
  if (dups) atomicAdd(&out[2], static_cast<unsigned long long>(dups));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a &= static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(a), off, 64));
    o |= static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(o), off, 64));
  }
  if ((threadIdx.x & 63u) == 0) { atomicAnd(&out[0], a); atomicOr(&out[1], o); }
}

struct FillArgs {
  void* out;
  uint64_t n;
  const void* sorted_keys;            // [distinct] raw key bits, in output order
  const unsigned long long* starts;   // [distinct + 1], starts[distinct] = n
  unsigned distinct;
};

template <typename K, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fill_runs_kernel(FillArgs a) {
  constexpr int VEC = 16 / sizeof(K);
  __shared__ unsigned long long ls[kDistinctMax + 2];
  __shared__ K lk[kDistinctMax + 1];
  const unsigned tid = threadIdx.x;
  for (unsigned i = tid; i <= a.distinct; i += BLOCK) ls[i] = a.starts[i];
  for (unsigned i = tid; i < a.distinct; i += BLOCK) lk[i] = static_cast<const K*>(a.sorted_keys)[i];
  __syncthreads();
  K* out = static_cast<K*>(a.out);
  auto run_of = [&](uint64_t p) {                 // the run that holds position p: last i with starts[i] <= p
    unsigned lo = 0, hi = a.distinct;             // invariant: starts[lo] <= p < starts[hi]
    while (hi - lo > 1) {
      const unsigned mid = (lo + hi) >> 1;
      if (ls[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
  };
  uint64_t head = ((16u - (reinterpret_cast<uintptr_t>(out) & 15u)) & 15u) / sizeof(K);
  if (head > a.n) head = a.n;
  const uint64_t nvec = (a.n - head) / VEC;
  struct alignas(16) Pack { K v[VEC]; };
  Pack* packs = reinterpret_cast<Pack*>(out + head);
  // every workgroup fills its own contiguous share front to back: the run of a thread's next pack is the run of its
  // last one or a later one, so one search at the start and a step forward now and then is all the looking-up there is
  const uint64_t per = (nvec + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = per * blockIdx.x, hi = lo + per < nvec ? lo + per : nvec;
  uint64_t i = lo + tid;
  if (i < hi) {
    unsigned r = run_of(head + i * VEC);
    unsigned long long next = ls[r + 1];
    for (; i < hi; i += BLOCK) {
      const uint64_t p = head + i * VEC;
      while (p >= next) next = ls[++r + 1];                // (p < n = starts[distinct]: r stays below distinct)
      const K kv = lk[r];
      Pack pk;
      if (p + VEC <= next) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) pk.v[j] = kv;
      } else {                                             // a pack that straddles run boundaries
        unsigned rr = r;
        unsigned long long nn = next;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          while (p + j >= nn) nn = ls[++rr + 1];           // (empty runs do not exist: every listed key occurs)
          pk.v[j] = lk[rr];
        }
      }
      packs[i] = pk;
    }
  }
  if (blockIdx.x == 0) {
    if (tid < head) out[tid] = lk[run_of(tid)];
    const uint64_t t = head + nvec * VEC + tid;
    if (t < a.n) out[t] = lk[run_of(t)];
  }
}
}  // namespace rsd
