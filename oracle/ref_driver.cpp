// ref_driver.cpp -- C entry points around the UNMODIFIED reference header.  TEST INFRASTRUCTURE.
//
// Compiled only in the build container (where /root/reference exists) by oracle/Makefile into
// oracle/_ref/libref_sort.so; the reference header is included from where it lies
// (-I/root/reference/RadixSort), never copied.  The built .so travels to the GPU box and is used
// (a) to pin oracle/rs_oracle.cpp, (b) to generate tests/golden/, (c) as bench.py's
// cpu_baseline (kind "reference").  Nothing in the product path loads it.
//
// Two gaps of g++ 11 / libstdc++ 11 and how this driver deals with them WITHOUT stand-ins:
//  * the header calls std::memcpy without <cstring> (radix_sort.hpp:404, :413, :900): this
//    driver includes the real <cstring> first, like any translation unit of a user could.
//  * libstdc++ 11 has no floating-point std::strong_order, so isSortedBi / insertionSort /
//    key::isSorted (radix_sort.hpp:276, :376, :754) cannot be instantiated for float/double.
//    No replacement is written.  For float keys the driver calls the reference's own internal
//    pipeline that follows those pre-checks -- getUnsignedVector -> selectSortStrategy ->
//    getUnsignedVector (radix_sort.hpp:724-727), resp. key::selectProjStrategy (:1619) -- and
//    refuses (status 3) the inputs the pre-checks would have taken (n <= 100).  A presorted or
//    reverse-sorted float input is still sorted correctly by that pipeline; only the fast exit
//    is skipped, and the result of a total-order sort is unique.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "radix_sort.hpp"

namespace {

using clk = std::chrono::steady_clock;

template <class T> struct Wide { T key; long long data; long long pad; };   // sizeof > 8: index path (:1565)
template <class T> struct Slim { T key; uint32_t data; };                  // sizeof <= 8 for 4-byte keys: direct path

template <class T> double run_value(void* keys, size_t n, bool mt) {
  T* p = static_cast<T*>(keys);
  std::vector<T> v(p, p + n);
  auto t0 = clk::now();
  radix_sort::sort(v, {}, mt);
  auto t1 = clk::now();
  std::memcpy(p, v.data(), n * sizeof(T));
  return std::chrono::duration<double>(t1 - t0).count();
}

template <class T> double run_float_value(void* keys, size_t n, bool mt) {
  namespace rv = radix_sort::internal::value;
  using U = radix_sort::internal::shared::t2u<T>;
  T* p = static_cast<T*>(keys);
  std::vector<T> v(p, p + n);
  auto t0 = clk::now();
  std::vector<U> vu(n);                       // radix_sort.hpp:724-727, verbatim call sequence
  rv::getUnsignedVector(v, vu, false, mt);
  rv::selectSortStrategy(vu, mt);
  rv::getUnsignedVector(v, vu, true, mt);
  auto t1 = clk::now();
  std::memcpy(p, v.data(), n * sizeof(T));
  return std::chrono::duration<double>(t1 - t0).count();
}

template <class Obj, class T, bool IsFloat> double run_pairs(void* keys, uint64_t* perm, size_t n, bool mt) {
  T* p = static_cast<T*>(keys);
  std::vector<Obj> v(n);
  for (size_t i = 0; i < n; ++i) { v[i].key = p[i]; v[i].data = static_cast<decltype(v[i].data)>(i); }
  auto t0 = clk::now();
  if constexpr (!IsFloat) radix_sort::sort(v, &Obj::key, mt);
  else radix_sort::internal::key::selectProjStrategy(v, &Obj::key, mt);   // radix_sort.hpp:1619
  auto t1 = clk::now();
  for (size_t i = 0; i < n; ++i) { p[i] = v[i].key; perm[i] = static_cast<uint64_t>(v[i].data); }
  return std::chrono::duration<double>(t1 - t0).count();
}

struct StrObj { std::string key; long long data; };   // BASELINE config 4: {std::string key; long long data}

}  // namespace

extern "C" {

// status: 0 ok, 2 bad arguments, 3 input not covered by this build (see header comment).
int ref_sort_keys(void* keys, size_t n, int key_bytes, int transform, int mt, double* seconds) {
  double s = 0;
  bool m = mt != 0;
  if (transform == 0) {
    switch (key_bytes) {
      case 1: s = run_value<uint8_t>(keys, n, m); break;
      case 2: s = run_value<uint16_t>(keys, n, m); break;
      case 4: s = run_value<uint32_t>(keys, n, m); break;
      case 8: s = run_value<uint64_t>(keys, n, m); break;
      default: return 2;
    }
  } else if (transform == 1) {
    switch (key_bytes) {
      case 1: s = run_value<int8_t>(keys, n, m); break;
      case 2: s = run_value<int16_t>(keys, n, m); break;
      case 4: s = run_value<int32_t>(keys, n, m); break;
      case 8: s = run_value<long long>(keys, n, m); break;
      default: return 2;
    }
  } else if (transform == 2) {
    if (n <= 100) return 3;
    switch (key_bytes) {
      case 4: s = run_float_value<float>(keys, n, m); break;
      case 8: s = run_float_value<double>(keys, n, m); break;
      default: return 2;
    }
  } else return 2;
  if (seconds) *seconds = s;
  return 0;
}

// layout 0: object wider than 8 bytes (index path), 1: 8-byte object (direct path; 4-byte keys only).
int ref_sort_pairs(void* keys, uint64_t* perm, size_t n, int key_bytes, int transform, int mt, int layout,
                   double* seconds) {
  double s = 0;
  bool m = mt != 0;
  if (layout == 1 && key_bytes != 4) return 2;
  if (transform == 2 && n <= 100) return 3;
  if (transform == 0) {
    if (key_bytes == 4) s = layout ? run_pairs<Slim<uint32_t>, uint32_t, false>(keys, perm, n, m)
                                   : run_pairs<Wide<uint32_t>, uint32_t, false>(keys, perm, n, m);
    else if (key_bytes == 8) s = run_pairs<Wide<uint64_t>, uint64_t, false>(keys, perm, n, m);
    else if (key_bytes == 2) s = run_pairs<Wide<uint16_t>, uint16_t, false>(keys, perm, n, m);
    else if (key_bytes == 1) s = run_pairs<Wide<uint8_t>, uint8_t, false>(keys, perm, n, m);
    else return 2;
  } else if (transform == 1) {
    if (key_bytes == 4) s = layout ? run_pairs<Slim<int32_t>, int32_t, false>(keys, perm, n, m)
                                   : run_pairs<Wide<int32_t>, int32_t, false>(keys, perm, n, m);
    else if (key_bytes == 8) s = run_pairs<Wide<long long>, long long, false>(keys, perm, n, m);
    else if (key_bytes == 2) s = run_pairs<Wide<int16_t>, int16_t, false>(keys, perm, n, m);
    else if (key_bytes == 1) s = run_pairs<Wide<int8_t>, int8_t, false>(keys, perm, n, m);
    else return 2;
  } else if (transform == 2) {
    if (key_bytes == 4) s = layout ? run_pairs<Slim<float>, float, true>(keys, perm, n, m)
                                   : run_pairs<Wide<float>, float, true>(keys, perm, n, m);
    else if (key_bytes == 8) s = run_pairs<Wide<double>, double, true>(keys, perm, n, m);
    else return 2;
  } else return 2;
  if (seconds) *seconds = s;
  return 0;
}

// String i is blob[offsets[i] .. offsets[i+1]).  stable=0: sort(vector<string>) -- the sorted
// strings are written back as out_blob/out_offsets (n+1 entries) since the value path moves the
// strings themselves.  stable=1: sort(vector<{string key; long long data}>, &key) -- perm[i] is
// the original index now at position i.
int ref_sort_strings(const uint8_t* blob, const uint64_t* offsets, size_t n, int stable, int mt,
                     uint8_t* out_blob, uint64_t* out_offsets, uint64_t* perm, double* seconds) {
  bool m = mt != 0;
  const char* b = reinterpret_cast<const char*>(blob);
  double s = 0;
  if (!stable) {
    std::vector<std::string> v(n);
    for (size_t i = 0; i < n; ++i) v[i].assign(b + offsets[i], offsets[i + 1] - offsets[i]);
    auto t0 = clk::now();
    radix_sort::sort(v, {}, m);
    s = std::chrono::duration<double>(clk::now() - t0).count();
    uint64_t pos = 0;
    for (size_t i = 0; i < n; ++i) {
      out_offsets[i] = pos;
      std::memcpy(out_blob + pos, v[i].data(), v[i].size());
      pos += v[i].size();
    }
    out_offsets[n] = pos;
  } else {
    std::vector<StrObj> v(n);
    for (size_t i = 0; i < n; ++i) {
      v[i].key.assign(b + offsets[i], offsets[i + 1] - offsets[i]);
      v[i].data = static_cast<long long>(i);
    }
    auto t0 = clk::now();
    radix_sort::sort(v, &StrObj::key, m);
    s = std::chrono::duration<double>(clk::now() - t0).count();
    for (size_t i = 0; i < n; ++i) perm[i] = static_cast<uint64_t>(v[i].data);
  }
  if (seconds) *seconds = s;
  return 0;
}

// Threads the reference would use for n elements on this machine (radix_sort.hpp:112-123).
size_t ref_num_threads(size_t n) { return radix_sort::internal::shared::getNumOfThreads(n); }

}  // extern "C"
