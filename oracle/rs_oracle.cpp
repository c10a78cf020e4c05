// rs_oracle.cpp -- CPU restatement of the reference sort path.  TEST INFRASTRUCTURE ONLY.
//
// This file is the *checker* for the HIP path: only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load it.  Nothing under radix-sort_amd/ or include/ links,
// loads or calls it; the product path fails loudly when the HIP library is missing.
//
// It restates, serially and with plain arrays, the algorithm of the reference header
// RadixSort/radix_sort.hpp (cited per function as "ref :LINE").  The reference's multithreading
// changes which strategy is selected (LSD vs MSD) but never the result, so the `mt` flag here walks
// the same strategy branches while executing on one thread.
//
// Pinning: oracle/_ref (the reference header itself, compiled by oracle/Makefile from
// /root/reference) is compared against this file in tests/test_oracle_vs_ref.py for every key type
// and shape; the golden fixtures in tests/golden/ were produced by oracle/_ref
// (tests/golden/make_golden.py) and are re-checked against this file on every CPU test run.
//
// Build: g++ -O2 -std=c++17 -shared -fPIC oracle/rs_oracle.cpp -o oracle/librs_oracle.so
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string_view>
#include <thread>
#include <vector>

namespace {

using Index = std::size_t;

// ref :32-50 -- digit width and the thresholds that steer the strategy.
constexpr Index kDigitBits = 8;
constexpr Index kBins = 256;
constexpr Index kStrBins = 257;            // bucket 256 = "past the end" (ref :38, :107-110)
constexpr Index kInsertionStr = 10;        // ref :42
constexpr Index kInsertionAll = 100;       // ref :43
constexpr Index kStrMaxDepth = 3;          // ref :44
constexpr Index kSwThreads = 12;           // ref :47
constexpr Index kMtThreshold = 1000000;    // ref :48

enum Transform { XF_NONE = 0, XF_SIGNED = 1, XF_IEEE = 2 };

struct Region { Index l, r, len, pos; };   // ref :52-59

// ref :112-123 -- thread-count policy; only used to decide LSD vs MSD (ref :688-695).
Index num_threads(Index n, Index hw) {
  if (n < (kMtThreshold << 1)) return 1;
  Index ratio = n / kMtThreshold, t = 1;
  while ((t << 1) <= ratio) t <<= 1;
  return std::min({t, hw, kSwThreads});
}

template <class U> constexpr U sign_mask() { return U(1) << (sizeof(U) * 8 - 1); }

// ref :405 / :412 -- IEEE-754 bit flip to an order-preserving unsigned key, and back.
template <class U> U ieee_fwd(U u) { return (u >> (sizeof(U) * 8 - 1)) ? U(~u) : U(u ^ sign_mask<U>()); }
template <class U> U ieee_inv(U u) { return (u >> (sizeof(U) * 8 - 1)) ? U(u ^ sign_mask<U>()) : U(~u); }

// Order key of a raw element: the comparison the reference applies to T (`<` for integers,
// std::strong_order for floats, ref :275-281, :375-377) equals `<` on this unsigned image.
template <class U> U order_key(U raw, int xf) {
  if (xf == XF_SIGNED) return U(raw ^ sign_mask<U>());
  if (xf == XF_IEEE) return ieee_fwd(raw);
  return raw;
}

// ref :136, :145, :150 -- digit of a raw integer at `shift`; signed top byte is ^0x80.
template <class U> Index digit_of(U raw, Index shift, bool is_signed) {
  Index d = Index(raw >> shift) & 0xFF;
  if (is_signed && shift == (sizeof(U) - 1) * 8) d ^= 0x80;
  return d;
}

// ref :269-305 -- ascending -> done; descending (strictly non-ascending) -> reverse in place.
template <class U> bool sorted_or_reversed(U* v, Index n, int xf) {
  if (n < 2) return true;
  bool asc = true, desc = true;
  for (Index i = 0; i + 1 < n; ++i) {
    U a = order_key(v[i], xf), b = order_key(v[i + 1], xf);
    if (asc && a > b) asc = false;
    if (desc && a < b) desc = false;
    if (!asc && !desc) break;
  }
  if (asc) return true;
  if (desc) { std::reverse(v, v + n); return true; }
  return false;
}

// ref :307-370 -- number of significant bytes of the maximum (any negative -> all bytes).
template <class U> Index max_len(const U* v, Index n, int xf) {
  constexpr Index full = sizeof(U);
  if (full == 1 || xf == XF_IEEE) return full;
  Index len = 0;
  U best = 0;
  // The running maximum starts at numeric_limits<T>::min(): 0 for unsigned (ref :342), so only
  // x > 0 moves it; INT_MIN for signed (ref :319), so the first non-negative x always does.
  bool have = (xf != XF_SIGNED);
  for (Index i = 0; i < n; ++i) {
    U x = v[i];
    if (xf == XF_SIGNED && (x & sign_mask<U>())) return full;
    if (!have || x > best) {
      best = x; have = true;
      while (len != full) {
        if ((best >> (kDigitBits * len)) == 0) break;
        if (++len == full) return full;
      }
    }
  }
  return len;
}

// ref :372-392 -- stable insertion sort on [l, r).
template <class U> void insertion(U* v, Index l, Index r, int xf) {
  for (Index i = l + 1; i < r; ++i) {
    U val = v[i];
    U kv = order_key(val, xf);
    Index j = i;
    while (j > l && kv < order_key(v[j - 1], xf)) { v[j] = v[j - 1]; --j; }
    v[j] = val;
  }
}

// ref :125-153 + :193-208 -- 256-bin count of one digit and its exclusive prefix from `l`.
template <class U> void count_digit(const U* v, Index l, Index r, Index shift, bool sg, Index* count) {
  std::fill(count, count + kBins, Index(0));
  for (Index i = l; i < r; ++i) ++count[digit_of(v[i], shift, sg)];
}
void prefix_from(const Index* count, Index l, Index* prefix) {
  prefix[0] = l;
  for (Index d = 1; d < kBins; ++d) prefix[d] = prefix[d - 1] + count[d - 1];
}

// ref :506-526 -- LSD: `len` stable counting passes from the least significant byte.
template <class U> void lsd(U* v, Index n, Index len, bool sg) {
  std::vector<U> tmp(n);
  U* a = v; U* b = tmp.data();
  Index count[kBins], prefix[kBins];
  for (Index p = 0, shift = 0; p < len; ++p, shift += kDigitBits) {
    count_digit(a, 0, n, shift, sg, count);
    prefix_from(count, 0, prefix);
    for (Index i = 0; i < n; ++i) b[prefix[digit_of(a[i], shift, sg)]++] = a[i];   // ref :447-470
    std::swap(a, b);
  }
  if (a != v) std::memcpy(v, a, n * sizeof(U));
}

// ref :528-586 (+ :210-254) -- MSD: LIFO of regions; each region is counted, scattered to tmp,
// copied back, then split into child regions for every bucket holding more than one key.
template <class U> void msd(U* v, Index n, Index len, bool sg, int xf) {
  std::vector<U> tmp(n);
  std::vector<Region> stack;
  stack.push_back({0, n, len, (len - 1) * 8});
  Index count[kBins], prefix[kBins];
  while (!stack.empty()) {
    Region g = stack.back(); stack.pop_back();
    if (g.r - g.l < 2 || g.len == 0) continue;
    if (g.r - g.l <= kInsertionAll) { insertion(v, g.l, g.r, xf); continue; }
    count_digit(v, g.l, g.r, g.pos, sg, count);
    prefix_from(count, g.l, prefix);
    for (Index i = g.l; i < g.r; ++i) tmp[prefix[digit_of(v[i], g.pos, sg)]++] = v[i];  // ref :480-498
    std::memcpy(v + g.l, tmp.data() + g.l, (g.r - g.l) * sizeof(U));                    // ref :578
    if (g.len - 1 == 0) continue;
    Index start = g.l;
    for (Index d = 0; d < kBins; ++d) {                                                 // ref :229-234
      if (count[d] > 1) stack.push_back({start, start + count[d], g.len - 1, g.pos - kDigitBits});
      start += count[d];
    }
  }
}

// ref :684-696 -- 8-byte keys (len>1) -> MSD; 1-byte keys, len<=1, no threads -> LSD; else MSD.
template <class U> void select_strategy(U* v, Index n, bool sg, int xf_cmp, bool mt, Index hw) {
  Index len = max_len(v, n, sg ? XF_SIGNED : XF_NONE);
  Index threads = num_threads(n, hw);
  bool large = sizeof(U) >= 8, small = sizeof(U) <= 1;
  if (large && len > 1) msd(v, n, len, sg, xf_cmp);
  else if (small || len <= 1 || !mt || threads <= 1) lsd(v, n, len, sg);
  else msd(v, n, len, sg, xf_cmp);
}

// ref :698-729 -- value-path dispatcher.
template <class U> void sort_value(U* v, Index n, int xf, bool mt, Index hw) {
  if (sorted_or_reversed(v, n, xf)) return;
  if (n <= kInsertionAll) { insertion(v, 0, n, xf); return; }
  if (xf != XF_IEEE) { select_strategy(v, n, xf == XF_SIGNED, xf, mt, hw); return; }
  for (Index i = 0; i < n; ++i) v[i] = ieee_fwd(v[i]);     // ref :394-407, :724-725
  select_strategy(v, n, false, XF_NONE, mt, hw);            // sorted as plain unsigned (ref :726)
  for (Index i = 0; i < n; ++i) v[i] = ieee_inv(v[i]);     // ref :408-415, :727
}

// ---------------------------------------------------------------------------------------------
// key path (stable): objects are represented by their original index; `k` is the key column.
// ---------------------------------------------------------------------------------------------

// ref :746-769 -- ascending only; no reversal on the key path (keeps stability).
template <class U> bool keys_sorted(const U* k, Index n, int xf) {
  for (Index i = 0; i + 1 < n; ++i)
    if (order_key(k[i], xf) > order_key(k[i + 1], xf)) return false;
  return true;
}

// ref :867-890 -- stable insertion sort of (index, key) pairs.
template <class U> void insertion_pairs(uint64_t* idx, U* k, Index l, Index r, int xf) {
  for (Index i = l + 1; i < r; ++i) {
    uint64_t o = idx[i]; U key = k[i]; U kv = order_key(key, xf);
    Index j = i;
    while (j > l && kv < order_key(k[j - 1], xf)) { idx[j] = idx[j - 1]; k[j] = k[j - 1]; --j; }
    idx[j] = o; k[j] = key;
  }
}

// ref :1184-1206 -- pair LSD.
template <class U> void lsd_pairs(uint64_t* idx, U* k, Index n, Index len, bool sg) {
  std::vector<uint64_t> ti(n); std::vector<U> tk(n);
  uint64_t* ia = idx; uint64_t* ib = ti.data(); U* ka = k; U* kb = tk.data();
  Index count[kBins], prefix[kBins];
  for (Index p = 0, shift = 0; p < len; ++p, shift += kDigitBits) {
    count_digit(ka, 0, n, shift, sg, count);
    prefix_from(count, 0, prefix);
    for (Index i = 0; i < n; ++i) {                          // ref :1062-1109
      Index pos = prefix[digit_of(ka[i], shift, sg)]++;
      ib[pos] = ia[i]; kb[pos] = ka[i];
    }
    std::swap(ia, ib); std::swap(ka, kb);
  }
  if (ia != idx) { std::memcpy(idx, ia, n * sizeof(uint64_t)); std::memcpy(k, ka, n * sizeof(U)); }
}

// ref :1274-1322 -- pair MSD.
template <class U> void msd_pairs(uint64_t* idx, U* k, Index n, Index len, bool sg, int xf) {
  std::vector<uint64_t> ti(n); std::vector<U> tk(n);
  std::vector<Region> stack;
  stack.push_back({0, n, len, (len - 1) * 8});
  Index count[kBins], prefix[kBins];
  while (!stack.empty()) {
    Region g = stack.back(); stack.pop_back();
    if (g.r - g.l < 2 || g.len == 0) continue;
    if (g.r - g.l <= kInsertionAll) { insertion_pairs(idx, k, g.l, g.r, xf); continue; }
    count_digit(k, g.l, g.r, g.pos, sg, count);
    prefix_from(count, g.l, prefix);
    for (Index i = g.l; i < g.r; ++i) {
      Index pos = prefix[digit_of(k[i], g.pos, sg)]++;
      ti[pos] = idx[i]; tk[pos] = k[i];
    }
    std::memcpy(idx + g.l, ti.data() + g.l, (g.r - g.l) * sizeof(uint64_t));
    std::memcpy(k + g.l, tk.data() + g.l, (g.r - g.l) * sizeof(U));
    if (g.len - 1 == 0) continue;
    Index start = g.l;
    for (Index d = 0; d < kBins; ++d) {
      if (count[d] > 1) stack.push_back({start, start + count[d], g.len - 1, g.pos - kDigitBits});
      start += count[d];
    }
  }
}

// ref :1530-1541, :1543-1592, :1594-1620 -- key dispatcher for numeric keys.  Whether the
// reference sorts the objects directly (sizeof(T) <= 8 or len <= 1, ref :1565) or sorts
// (index, key) pairs and gathers (ref :1571-1590), both are the same stable counting sort, so the
// permutation is identical; this restatement always carries the index.
template <class U> void sort_pairs(U* k, uint64_t* idx, Index n, int xf, bool mt, Index hw) {
  for (Index i = 0; i < n; ++i) idx[i] = i;                 // ref :1553-1554, :1571-1572
  if (keys_sorted(k, n, xf)) return;                        // ref :1608
  if (n <= kInsertionAll) { insertion_pairs(idx, k, 0, n, xf); return; }   // ref :1613-1617
  bool sg = xf == XF_SIGNED;
  Index len = max_len(k, n, xf);                            // ref :1548 (before the flip)
  int xf_cmp = xf;
  if (xf == XF_IEEE) {                                      // ref :1550-1558
    for (Index i = 0; i < n; ++i) k[i] = ieee_fwd(k[i]);
    sg = false; xf_cmp = XF_NONE;
  }
  Index threads = num_threads(n, hw);
  bool large = sizeof(U) >= 8, small = sizeof(U) <= 1;
  if (large && len > 1) msd_pairs(idx, k, n, len, sg, xf_cmp);
  else if (small || len <= 1 || !mt || threads <= 1) lsd_pairs(idx, k, n, len, sg);
  else msd_pairs(idx, k, n, len, sg, xf_cmp);
  if (xf == XF_IEEE) for (Index i = 0; i < n; ++i) k[i] = ieee_inv(k[i]);   // hand keys back raw
}

// ---------------------------------------------------------------------------------------------
// strings: a string is a byte span; the sort moves indices (the reference moves std::string
// objects, ref :478, or indices through a projection, ref :1574-1578 -- same order either way).
// ---------------------------------------------------------------------------------------------
struct StrTable {
  const uint8_t* blob; const uint64_t* off;
  std::string_view at(uint64_t i) const {
    return std::string_view(reinterpret_cast<const char*>(blob) + off[i], off[i + 1] - off[i]);
  }
  Index chr(uint64_t i, Index pos) const {                  // ref :107-110
    Index len = off[i + 1] - off[i];
    return pos < len ? Index(blob[off[i] + pos]) : 256;
  }
  // std::string::operator< compares bytes as unsigned char (char_traits<char>::lt), shorter prefix
  // first; string_view::compare does the same.
  bool less(uint64_t a, uint64_t b) const { return at(a).compare(at(b)) < 0; }
};

void insertion_str(const StrTable& t, uint64_t* p, Index l, Index r) {   // ref :372-392, :842-865
  for (Index i = l + 1; i < r; ++i) {
    uint64_t o = p[i]; Index j = i;
    while (j > l && t.less(o, p[j - 1])) { p[j] = p[j - 1]; --j; }
    p[j] = o;
  }
}

// ref :528-586 / :1208-1272 -- string MSD over byte positions with the depth-3 fallback to
// std::sort (value path, ref :559-569) or std::stable_sort (key path, ref :1241-1255).
void msd_strings(const StrTable& t, uint64_t* p, Index n, Index len, bool stable) {
  std::vector<uint64_t> tmp(n);
  std::vector<Region> stack;
  stack.push_back({0, n, len, 0});
  std::vector<Index> count(kStrBins), prefix(kStrBins);
  auto lt = [&t](uint64_t a, uint64_t b) { return t.less(a, b); };
  while (!stack.empty()) {
    Region g = stack.back(); stack.pop_back();
    if (g.r - g.l < 2 || g.len == 0) continue;
    if (g.r - g.l <= kInsertionStr) { insertion_str(t, p, g.l, g.r); continue; }
    if (g.pos >= kStrMaxDepth) {
      if (stable) std::stable_sort(p + g.l, p + g.r, lt); else std::sort(p + g.l, p + g.r, lt);
      continue;
    }
    std::fill(count.begin(), count.end(), Index(0));
    for (Index i = g.l; i < g.r; ++i) ++count[t.chr(p[i], g.pos)];
    prefix[256] = g.l;                                       // ref :196-200: end-of-string first
    prefix[0] = g.l + count[256];
    for (Index d = 1; d < kBins; ++d) prefix[d] = prefix[d - 1] + count[d - 1];
    for (Index i = g.l; i < g.r; ++i) tmp[prefix[t.chr(p[i], g.pos)]++] = p[i];
    std::memcpy(p + g.l, tmp.data() + g.l, (g.r - g.l) * sizeof(uint64_t));
    if (g.len - 1 == 0) continue;
    Index start = g.l + count[256];                          // ref :216-220: bucket 256 is final
    for (Index d = 0; d < kBins; ++d) {
      if (count[d] > 1) stack.push_back({start, start + count[d], g.len - 1, g.pos + 1});
      start += count[d];
    }
  }
}

void sort_strings(const StrTable& t, uint64_t* p, Index n, bool stable) {
  for (Index i = 0; i < n; ++i) p[i] = i;
  if (n < 2) return;
  // value path: ref :269-305 (ascending -> done, strictly descending -> reverse);
  // key path: ref :746-769 (ascending only).
  bool asc = true, desc = true;
  for (Index i = 0; i + 1 < n; ++i) {
    int c = t.at(i).compare(t.at(i + 1));
    if (c > 0) asc = false;
    if (c < 0) desc = false;
    if (!asc && (!desc || stable)) break;
  }
  if (asc) return;
  if (!stable && desc) { std::reverse(p, p + n); return; }
  if (n <= kInsertionStr) { insertion_str(t, p, 0, n); return; }   // ref :713-718, :1612-1617
  Index len = 0;                                             // ref :361-367, :829-836
  for (Index i = 0; i < n; ++i) len = std::max<Index>(len, t.off[i + 1] - t.off[i]);
  msd_strings(t, p, n, len, stable);                         // ref :690-691, :1522-1523
}

template <class F> int by_width(int key_bytes, F&& f) {
  switch (key_bytes) {
    case 1: f(uint8_t{}); return 0;
    case 2: f(uint16_t{}); return 0;
    case 4: f(uint32_t{}); return 0;
    case 8: f(uint64_t{}); return 0;
  }
  return 1;
}

Index hw_threads() { Index h = std::thread::hardware_concurrency(); return h ? h : 1; }   // ref :46

}  // namespace

extern "C" {

// radix_sort::sort(v, {}, mt) for a numeric T (ref :1628-1638).  `keys` holds the raw bytes of T.
int orc_sort_keys(void* keys, size_t n, int key_bytes, int transform, int mt) {
  if (transform < 0 || transform > 2 || (transform == XF_IEEE && key_bytes < 4)) return 2;
  return by_width(key_bytes, [&](auto tag) {
    using U = decltype(tag);
    sort_value(static_cast<U*>(keys), n, transform, mt != 0, hw_threads());
  });
}

// radix_sort::sort(v, proj, mt) for a numeric key (ref :1640-1647): on return `perm[i]` is the
// original index of the object now at position i and `keys` is the key column in sorted order.
int orc_sort_pairs(void* keys, uint64_t* perm, size_t n, int key_bytes, int transform, int mt) {
  if (transform < 0 || transform > 2 || (transform == XF_IEEE && key_bytes < 4)) return 2;
  return by_width(key_bytes, [&](auto tag) {
    using U = decltype(tag);
    sort_pairs(static_cast<U*>(keys), perm, n, transform, mt != 0, hw_threads());
  });
}

// std::string keys: string i is blob[offsets[i] .. offsets[i+1]).  stable=0 is the value path
// (sort(v)), stable=1 the projected path (sort(v, &T::name)).  perm as above.
int orc_sort_strings(const uint8_t* blob, const uint64_t* offsets, size_t n, uint64_t* perm, int stable) {
  StrTable t{blob, offsets};
  sort_strings(t, perm, n, stable != 0);
  return 0;
}

// Which strategy the reference would take (for tests of the dispatcher): 0 = presorted exit,
// 1 = insertion sort, 2 = LSD, 3 = MSD.  hw = hardware threads assumed.
int orc_strategy(const void* keys, size_t n, int key_bytes, int transform, int mt, int hw) {
  int out = -1;
  by_width(key_bytes, [&](auto tag) {
    using U = decltype(tag);
    std::vector<U> v(static_cast<const U*>(keys), static_cast<const U*>(keys) + n);
    if (sorted_or_reversed(v.data(), n, transform)) { out = 0; return; }
    if (n <= kInsertionAll) { out = 1; return; }
    Index len = max_len(v.data(), n, transform);
    Index threads = num_threads(n, Index(hw));
    bool large = sizeof(U) >= 8, small = sizeof(U) <= 1;
    if (large && len > 1) out = 3;
    else if (small || len <= 1 || !mt || threads <= 1) out = 2;
    else out = 3;
  });
  return out;
}

}  // extern "C"
