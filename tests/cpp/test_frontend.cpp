// test_frontend.cpp -- the reference's own check (benchmark::testing, benchmark.cpp:394-577),
// restated for include/radix_sort.hpp: sort with radix_sort::sort, sort a copy with
// std::sort / std::stable_sort, compare element-wise.  Composite types carry a payload that
// differs between equal keys, so the comparison pins stability (as Employee does in the reference).
// Build: g++ -std=c++20 -O2 -Iinclude tests/cpp/test_frontend.cpp -Lradix-sort_amd/lib -lrsort_hip
#include <algorithm>
#include <atomic>
#include <bit>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "radix_sort.hpp"

static std::atomic<int> failures{0};
static std::atomic<int> cases{0};

template <class U> static U flipped(U u) {           // IEEE-754 totalOrder as an unsigned image
  constexpr U S = U(1) << (sizeof(U) * 8 - 1);
  return (u & S) ? U(~u) : U(u ^ S);
}
template <class T> static bool total_less(T a, T b) {
  if constexpr (std::floating_point<T>) {
    using U = std::conditional_t<sizeof(T) == 4, uint32_t, uint64_t>;
    return flipped(std::bit_cast<U>(a)) < flipped(std::bit_cast<U>(b));
  } else return a < b;
}
template <class T> static bool same_bits(const T& a, const T& b) {
  if constexpr (std::is_trivially_copyable_v<T>) return std::memcmp(&a, &b, sizeof(T)) == 0;
  else return a == b;
}

template <class T> static std::vector<T> make_keys(size_t n, int shape, std::mt19937_64& gen) {
  std::vector<T> v(n);
  if constexpr (std::same_as<T, std::string>) {
    std::uniform_int_distribution<int> len(0, 20), ch(0, 255);
    std::vector<std::string> uniq(256);
    for (auto& s : uniq) { s.resize(len(gen)); for (auto& c : s) c = static_cast<char>(ch(gen)); }
    for (auto& s : v) {
      if (shape == 1) s = uniq[gen() & 255];
      else { s.resize(len(gen)); for (auto& c : s) c = static_cast<char>(ch(gen)); }
    }
  } else {
    using U = std::conditional_t<sizeof(T) == 1, uint8_t, std::conditional_t<sizeof(T) == 2, uint16_t,
              std::conditional_t<sizeof(T) == 4, uint32_t, uint64_t>>>;
    U uniq[256];
    for (auto& u : uniq) u = static_cast<U>(gen());
    for (auto& x : v) { U u = shape == 1 ? uniq[gen() & 255] : static_cast<U>(gen()); std::memcpy(&x, &u, sizeof(T)); }
  }
  if (shape == 2) std::sort(v.begin(), v.end(), [](const T& a, const T& b) { return total_less(b, a); });   // reverse sorted
  if (shape == 3) std::sort(v.begin(), v.end(), [](const T& a, const T& b) { return total_less(a, b); });   // sorted
  return v;
}

template <class T> static void check_value(const char* name, size_t n, int shape) {
  std::mt19937_64 gen(69 + n + shape);
  std::vector<T> v = make_keys<T>(n, shape, gen), want = v;
  std::sort(want.begin(), want.end(), [](const T& a, const T& b) { return total_less(a, b); });
  radix_sort::sort(v, {}, true);
  bool ok = v.size() == want.size();
  for (size_t i = 0; ok && i < n; i++) ok = same_bits(v[i], want[i]);
  ++cases;
  if (!ok) { ++failures; std::printf("WRONG SORT value %s n=%zu shape=%d\n", name, n, shape); }
}

template <class K> struct Composite { K key; long long data; std::string tag; };   // non-trivial, wider than 8 bytes

template <class K> static void check_key(const char* name, size_t n, int shape) {
  std::mt19937_64 gen(1069 + n + shape);
  std::vector<K> keys = make_keys<K>(n, shape, gen);
  std::vector<Composite<K>> v(n);
  for (size_t i = 0; i < n; i++) v[i] = {keys[i], static_cast<long long>(i), "t" + std::to_string(i % 7)};
  auto want = v;
  std::stable_sort(want.begin(), want.end(), [](const auto& a, const auto& b) { return total_less(a.key, b.key); });
  if (shape % 2 == 0) radix_sort::sort(v, &Composite<K>::key, true);
  else radix_sort::sort(v, [](const Composite<K>& c) -> const K& { return c.key; });
  bool ok = true;
  for (size_t i = 0; ok && i < n; i++) ok = same_bits(v[i].key, want[i].key) && v[i].data == want[i].data && v[i].tag == want[i].tag;
  ++cases;
  if (!ok) { ++failures; std::printf("WRONG SORT key %s n=%zu shape=%d\n", name, n, shape); }
}

// Trivially copyable objects wider than 8 bytes: the front-end sends them to the GPU with their keys and the permutation is
// applied there (rs_sort_rows); the result must still be the stable order, payload for payload.
template <class K> struct Pod { K key; unsigned seq; long long data; };
template <class K> static void check_pod(const char* name, size_t n, int shape) {
  static_assert(std::is_trivially_copyable_v<Pod<K>> && sizeof(Pod<K>) % 4 == 0);
  std::mt19937_64 gen(2069 + n + shape);
  std::vector<K> keys = make_keys<K>(n, shape, gen);
  std::vector<Pod<K>> v(n);
  for (size_t i = 0; i < n; i++) v[i] = {keys[i], static_cast<unsigned>(i), static_cast<long long>(i) * 3 - 7};
  auto want = v;
  std::stable_sort(want.begin(), want.end(), [](const auto& a, const auto& b) { return total_less(a.key, b.key); });
  radix_sort::sort(v, &Pod<K>::key, shape % 2 == 0);
  bool ok = true;
  for (size_t i = 0; ok && i < n; i++) ok = same_bits(v[i].key, want[i].key) && v[i].seq == want[i].seq && v[i].data == want[i].data;
  ++cases;
  if (!ok) { ++failures; std::printf("WRONG SORT pod %s n=%zu shape=%d\n", name, n, shape); }
}

// A projection that returns bool (std::integral, so `supported`): two buckets, stable inside each.
static void check_bool_projection(size_t n) {
  struct Flagged { bool flag; std::string tag; long long data; };
  std::mt19937_64 gen(3069 + n);
  std::vector<Flagged> v(n);
  for (size_t i = 0; i < n; i++) v[i] = {static_cast<bool>(gen() & 1), "f" + std::to_string(i % 5), static_cast<long long>(i)};
  auto want = v;
  std::stable_sort(want.begin(), want.end(), [](const Flagged& a, const Flagged& b) { return a.flag < b.flag; });
  radix_sort::sort(v, &Flagged::flag, true);
  bool ok = true;
  for (size_t i = 0; ok && i < n; i++) ok = v[i].flag == want[i].flag && v[i].data == want[i].data && v[i].tag == want[i].tag;
  ++cases;
  if (!ok) { ++failures; std::printf("WRONG SORT bool projection n=%zu\n", n); }
}

int main() {
  const size_t sizes[] = {0, 1, 2, 10, 100, 101, 1000, 10000, 100000, 1000000};
  for (size_t n : sizes)
    for (int shape = 0; shape < 3; shape++) {
      check_value<char>("char", n, shape);
      check_value<unsigned char>("uchar", n, shape);
      check_value<short>("short", n, shape);
      check_value<unsigned short>("ushort", n, shape);
      check_value<int>("int", n, shape);
      check_value<unsigned int>("uint", n, shape);
      check_value<long long>("ll", n, shape);
      check_value<unsigned long long>("ull", n, shape);
      check_value<float>("float", n, shape);
      check_value<double>("double", n, shape);
      if (n <= 100000) check_value<std::string>("string", n, shape);
      if (n <= 100000) {
        check_key<int>("complex_i32", n, shape);
        check_key<long long>("complex_ll", n, shape);
        check_key<float>("complex_float", n, shape);
        check_key<double>("complex_double", n, shape);
        check_key<unsigned char>("complex_u8", n, shape);
        check_key<std::string>("complex_str", n, shape);
      }
      check_pod<int>("pod_i32", n, shape);
      check_pod<double>("pod_double", n, shape);
      check_pod<unsigned short>("pod_u16", n, shape);
    }
  check_pod<long long>("pod_ll_threaded", 2100001, 0);
  check_pod<float>("pod_float_sorted", 100000, 3);
  for (size_t n : {size_t(2), size_t(1000), size_t(300000)}) check_bool_projection(n);
  // above 2,000,000 elements with the third argument true the host-side steps (key column, string packing, moving
  // the objects) run on several threads, by the reference's own rule (getNumOfThreads, radix_sort.hpp:112-123)
  check_key<int>("complex_i32_threaded", 2100001, 0);
  check_key<double>("complex_double_threaded", 2100001, 1);     // shape 1 calls sort(v, proj) without the flag: serial host path
  check_key<std::string>("complex_str_threaded", 2100001, 0);
  check_value<std::string>("string_threaded", 2100001, 0);
  // ordered inputs: the fast exits (device pre-check for numeric keys, host checks for strings and projected keys)
  for (size_t n : {size_t(2), size_t(1000), size_t(100000), size_t(2100001)}) {
    check_value<unsigned int>("uint_sorted", n, 3);
    check_value<double>("double_sorted", n, 3);
    check_value<std::string>("string_sorted", n, 3);
    check_value<std::string>("string_reverse", n, 2);
    check_key<int>("complex_i32_sorted", n, 3);        // shape 3 is odd: lambda projection, serial host path
    check_key<std::string>("complex_str_sorted", n, 3);
  }
  // The reference is re-entrant (no mutable globals): several caller threads sort at the same time.  Here each
  // caller thread gets its own device context (thread_local), so their transfers and kernels overlap on the one GPU.
  {
    std::vector<std::thread> callers;
    for (int t = 0; t < 4; t++)
      callers.emplace_back([t] {
        check_value<unsigned int>("uint_concurrent", 3000000 + 1000 * t, 0);     // staged transfers, threaded host side
        check_key<int>("complex_i32_concurrent", 300000 + t, t % 3);
        check_value<double>("double_concurrent", 50000 + t, 1);
        check_key<std::string>("complex_str_concurrent", 20000 + t, 0);
        check_value<unsigned short>("ushort_concurrent", 1000 + t, 0);           // single-workgroup kernel
      });
    for (auto& th : callers) th.join();
  }
  std::printf("%d cases, WRONG SORTS: %d\n", cases.load(), failures.load());
  return failures ? 1 : 0;
}
