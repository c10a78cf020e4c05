// bench_matrix.cpp -- the reference harness' type x shape x size matrix, run through include/radix_sort.hpp.
//
// What it mirrors (reference file:line, RadixSort/):
//   benchmark.cpp:147-186   one row = copy of the input made OUTSIDE the timer, the call timed with steady_clock,
//                           mean of ITERATIONS in whole microseconds
//   benchmark.cpp:200-343   loops shape -> type -> size; block text "TYPE\nSIZE = 1,000 (shape)\n\nmethod = t us"
//   benchmark.cpp:394-577   testing mode: expected = std::sort (std::stable_sort for projected keys), "(correct)" /
//                           "(wrong, different outputs)", final "WRONG SORTS = k"
//   main.cpp:19-21          sizes 0..1e8 (keys), 0..5e7 (strings), 0..1e7 (Employee); ITERATIONS = 3
//   generators.hpp:75-387   input shapes (restated below in this file's own words; seed 69)
// so the text lines up row for row with the reference's benchmarks/*.txt.  The rows `sort_par` / `stable_sort_par`
// need a parallel STL back-end this image does not have and are not printed; `sort` / `stable_sort` are opt-in
// (--with-std) because they take minutes at the top sizes.  Every radix row runs on the GPU, host vector in, host
// vector out (PCIe included) -- there is no CPU sort behind radix_sort::sort here.
//
//   g++ -std=c++20 -O2 -pthread -Iinclude tools/bench_matrix.cpp -Lradix-sort_amd/lib -lrsort_hip -o tools/bench_matrix
//   tools/bench_matrix [--testing] [--types INT,UINT,...] [--shapes randomized,...] [--max-size N] [--iterations K]
//                      [--with-std]
#include <algorithm>
#include <bit>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <random>
#include <string>
#include <type_traits>
#include <vector>

#include "radix_sort.hpp"

namespace {

struct Employee {                       // the reference's composite (generators.hpp:11-35): five sortable members
  std::int32_t age = 0;
  long long id = 0;
  float salary_f = 0.f;
  double salary = 0.0;
  std::string name;
};

enum Shape { RANDOMIZED, SORTED, REVERSE_SORTED, NEARLY_SORTED, DUPLICATES, SHAPES };
const char* const kShapeName[SHAPES] = {"randomized", "sorted", "reverse sorted", "nearly sorted", "duplicates"};
constexpr unsigned kSeed = 69;          // generators.hpp:56
constexpr std::size_t kDistinct = 256;  // generators.hpp:95

// ---- total order on IEEE bit patterns (what std::strong_order gives; libstdc++ 11 has no floating strong_order) ----
template <class F> auto orderedBits(F x) {
  using U = std::conditional_t<sizeof(F) == 4, std::uint32_t, std::uint64_t>;
  U u = std::bit_cast<U>(x);
  constexpr U top = U(1) << (sizeof(U) * 8 - 1);
  return (u & top) ? U(~u) : U(u | top);
}
template <class K> bool keyLess(const K& a, const K& b) {
  if constexpr (std::is_floating_point_v<K>) return orderedBits(a) < orderedBits(b);
  else return a < b;
}
template <class K> bool keySame(const K& a, const K& b) {
  if constexpr (std::is_floating_point_v<K>) return orderedBits(a) == orderedBits(b);
  else return a == b;
}

// ---- shapes -------------------------------------------------------------------------------------------------------
// sorted ramp + its two derivatives, shared by every element type (generators.hpp:131-144, :241-254, :349-362)
template <class T> void finishRamp(std::vector<T>& v, Shape shape, std::mt19937_64& gen) {
  if (shape == REVERSE_SORTED) std::reverse(v.begin(), v.end());
  if (shape == NEARLY_SORTED) {
    std::uniform_int_distribution<std::size_t> pick(0, v.size() - 1);
    for (std::size_t swaps = static_cast<std::size_t>(std::ceil(v.size() * 0.05)); swaps; --swaps) {
      std::size_t i = pick(gen), j = pick(gen);
      std::swap(v[i], v[j]);
    }
  }
}
// 256 distinct values drawn first, then n picks among them (generators.hpp:93-107)
template <class T, class Draw> std::vector<T> fewValued(std::size_t n, std::mt19937_64& gen, Draw&& draw) {
  std::vector<T> pool(kDistinct), v(n);
  for (auto& x : pool) x = draw();
  std::uniform_int_distribution<std::size_t> pick(0, kDistinct - 1);
  for (auto& x : v) x = pool[pick(gen)];
  return v;
}

template <class T> std::vector<T> makeIntegers(std::size_t n, Shape shape) {      // generators.hpp:75-151
  using Lim = std::numeric_limits<T>;
  using Wide = std::conditional_t<(sizeof(T) < 4), std::int32_t, T>;                // 1- and 2-byte types draw as int32
  std::mt19937_64 gen(kSeed);
  std::uniform_int_distribution<Wide> full(Lim::min(), Lim::max());
  auto draw = [&] { return static_cast<T>(full(gen)); };
  if (shape == RANDOMIZED) { std::vector<T> v(n); for (auto& x : v) x = draw(); return v; }
  if (shape == DUPLICATES) return fewValued<T>(n, gen, draw);
  std::vector<T> v(n);
  using U = std::make_unsigned_t<T>;
  if constexpr (sizeof(T) < 8) {                                                     // ramp in double arithmetic
    const double step = static_cast<double>(std::numeric_limits<U>::max()) / static_cast<double>(n - 1);
    double at = static_cast<double>(Lim::min());
    for (std::size_t i = 0; i < n; ++i, at += step) v[i] = static_cast<T>(at);
  } else {                                                                           // 8-byte: exact integer steps
    const U step = std::numeric_limits<U>::max() / (n - 1);
    U at = static_cast<U>(Lim::min());
    for (std::size_t i = 0; i < n; ++i, at += step) v[i] = static_cast<T>(at);
  }
  v.back() = Lim::max();
  finishRamp(v, shape, gen);
  return v;
}

template <class F> std::vector<F> makeFloats(std::size_t n, Shape shape) {         // generators.hpp:153-261
  using U = std::conditional_t<sizeof(F) == 4, std::uint32_t, std::uint64_t>;
  std::mt19937_64 gen(kSeed);
  std::uniform_int_distribution<U> bits(0, std::numeric_limits<U>::max());            // every pattern, NaN and inf included
  auto draw = [&] { return std::bit_cast<F>(bits(gen)); };
  if (shape == RANDOMIZED) { std::vector<F> v(n); for (auto& x : v) x = draw(); return v; }
  if (shape == DUPLICATES) return fewValued<F>(n, gen, draw);
  std::vector<F> v(n);
  constexpr F lo = std::numeric_limits<F>::lowest(), hi = std::numeric_limits<F>::max();
  if constexpr (sizeof(F) < 8) {
    const double step = (static_cast<double>(hi) - static_cast<double>(lo)) / static_cast<double>(n - 1);
    double at = static_cast<double>(lo);
    for (std::size_t i = 0; i < n; ++i, at += step) v[i] = static_cast<F>(at);
  } else {
    const F step = hi / static_cast<F>(n - 1) - lo / static_cast<F>(n - 1);
    F at = lo;
    for (std::size_t i = 0; i < n; ++i, at += step) v[i] = at;
  }
  v.back() = hi;
  finishRamp(v, shape, gen);
  return v;
}

std::vector<std::string> makeStrings(std::size_t n, Shape shape) {                  // generators.hpp:263-369
  std::mt19937_64 gen(kSeed);
  std::uniform_int_distribution<std::size_t> length(0, 20);
  std::uniform_int_distribution<int> byte(0, 255);
  auto draw = [&] {
    std::string s(length(gen), '\0');
    for (auto& c : s) c = static_cast<char>(byte(gen));
    return s;
  };
  if (shape == RANDOMIZED) { std::vector<std::string> v(n); for (auto& s : v) s = draw(); return v; }
  if (shape == DUPLICATES) return fewValued<std::string>(n, gen, draw);
  // the "sorted" strings count upwards in base 256, shortest first: "", "\0", "\1", ..., "\xff", "\0\0", ...
  std::vector<std::string> v(n);
  std::string cur;
  for (std::size_t i = 0; i < n; ++i) {
    v[i] = cur;
    std::size_t k = cur.size();
    while (k && static_cast<unsigned char>(cur[k - 1]) == 255) cur[--k] = '\0';
    if (k) cur[k - 1] = static_cast<char>(static_cast<unsigned char>(cur[k - 1]) + 1);
    else cur.assign(cur.size() + 1, '\0');
  }
  finishRamp(v, shape, gen);
  return v;
}

template <class T> std::vector<T> makeInput(std::size_t n, Shape shape) {           // generators.hpp:390-401
  if (n == 0) return {};
  if (n == 1) return std::vector<T>(1);
  if constexpr (std::is_same_v<T, Employee>) {
    auto a = makeIntegers<std::int32_t>(n, shape);
    auto b = makeIntegers<long long>(n, shape);
    auto c = makeFloats<float>(n, shape);
    auto d = makeFloats<double>(n, shape);
    auto e = makeStrings(n, shape);
    std::vector<Employee> v(n);
    for (std::size_t i = 0; i < n; ++i) v[i] = Employee{a[i], b[i], c[i], d[i], std::move(e[i])};
    return v;
  } else if constexpr (std::is_same_v<T, std::string>) return makeStrings(n, shape);
  else if constexpr (std::is_floating_point_v<T>) return makeFloats<T>(n, shape);
  else return makeIntegers<T>(n, shape);
}

// ---- text ---------------------------------------------------------------------------------------------------------
std::string grouped(long long x) {                                                   // 1234567 -> "1,234,567" ({:L}, en_US)
  std::string d = std::to_string(x), out;
  for (std::size_t i = 0; i < d.size(); ++i) {
    out += d[i];
    std::size_t left = d.size() - 1 - i;
    if (left && left % 3 == 0 && d[i] != '-') out += ',';
  }
  return out;
}
const char kRule[] = "===============================\n";

struct Options {
  bool testing = false, withStd = false;
  int iterations = 3;
  std::size_t maxSize = 100000000;
  std::vector<std::string> types, shapes;
};
bool wanted(const std::vector<std::string>& list, const std::string& name) {
  return list.empty() || std::find(list.begin(), list.end(), name) != list.end();
}

template <class F> long long timedUs(F&& f) {
  auto t0 = std::chrono::steady_clock::now();
  f();
  return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
}

// One (type, member) of the matrix.  Proj is std::identity for plain keys, a member pointer for Employee rows.
template <class T, class Proj> struct Row {
  Proj proj;
  bool less(const T& a, const T& b) const { return keyLess(std::invoke(proj, a), std::invoke(proj, b)); }

  template <class Call> void timeMethod(const char* label, const std::vector<T>& input, int iterations, Call&& call) const {
    long long total = 0;
    for (int i = 0; i < iterations; ++i) {
      std::vector<T> work(input);                                                    // the copy is not timed
      total += timedUs([&] { call(work); });
    }
    std::printf("%s = %s us\n", label, grouped(total / iterations).c_str());
  }

  void bench(const std::vector<T>& input, const Options& o) const {
    auto cmp = [this](const T& a, const T& b) { return less(a, b); };
    if (o.withStd) {
      timeMethod("sort           ", input, o.iterations, [&](std::vector<T>& w) { std::sort(w.begin(), w.end(), cmp); });
      timeMethod("stable_sort    ", input, o.iterations, [&](std::vector<T>& w) { std::stable_sort(w.begin(), w.end(), cmp); });
    }
    timeMethod("radix_sort     ", input, o.iterations, [&](std::vector<T>& w) { radix_sort::sort(w, proj, false); });
    timeMethod("radix_sort_par ", input, o.iterations, [&](std::vector<T>& w) { radix_sort::sort(w, proj, true); });
  }

  // element-wise comparison of two results: keys by total order; projected rows compare every member (= same permutation)
  static bool sameElement(const T& a, const T& b) {
    if constexpr (std::is_same_v<T, Employee>)
      return a.age == b.age && a.id == b.id && keySame(a.salary_f, b.salary_f) && keySame(a.salary, b.salary) && a.name == b.name;
    else return keySame(a, b);
  }
  static bool sameVector(const std::vector<T>& a, const std::vector<T>& b) {
    return a.size() == b.size() && std::equal(a.begin(), a.end(), b.begin(), sameElement);
  }

  bool test(std::vector<T>& input) const {                                            // benchmark.cpp:394-429
    auto cmp = [this](const T& a, const T& b) { return less(a, b); };
    constexpr bool projected = std::is_same_v<T, Employee>;
    std::vector<T> expected(input), got(input);
    long long us = timedUs([&] {
      if constexpr (projected) std::stable_sort(expected.begin(), expected.end(), cmp);
      else std::sort(expected.begin(), expected.end(), cmp);
    });
    std::printf("%s = %s us\n", projected ? "stable_sort    " : "sort           ", grouped(us).c_str());
    bool ok = true;
    us = timedUs([&] { radix_sort::sort(got, proj, false); });
    bool same = sameVector(got, expected);
    std::printf("radix_sort      = %s us \t%s\n", grouped(us).c_str(), same ? "(correct)" : "(wrong, different outputs)");
    ok &= same;
    got = std::move(input);
    us = timedUs([&] { radix_sort::sort(got, proj, true); });
    same = sameVector(got, expected);
    std::printf("radix_sort_par  = %s us \t%s\n", grouped(us).c_str(), same ? "(correct)" : "(wrong, different outputs)");
    return ok && same;
  }
};

struct Matrix {
  Options o;
  std::size_t wrong = 0, blocks = 0;

  template <class T, class Proj = std::identity>
  void run(const char* typeName, Shape shape, const std::vector<std::size_t>& sizes, Proj proj = {}) {
    if (!wanted(o.types, typeName)) return;
    for (std::size_t n : sizes) {
      if (n > o.maxSize) continue;
      std::printf("%s\nSIZE = %s (%s)\n\n", typeName, grouped(static_cast<long long>(n)).c_str(), kShapeName[shape]);
      std::vector<T> input = makeInput<T>(n, shape);
      Row<T, Proj> row{proj};
      if (o.testing) wrong += !row.test(input);
      else row.bench(input, o);
      std::printf("\n%s\n", kRule);
      std::fflush(stdout);
      ++blocks;
    }
  }

  void all() {
    const std::vector<std::size_t> keys = {0, 1, 10, 100, 1000, 10000, 100000, 1000000, 10000000, 100000000};       // main.cpp:19
    const std::vector<std::size_t> strs = {0, 1, 10, 100, 1000, 10000, 100000, 1000000, 10000000, 50000000};        // main.cpp:20
    const std::vector<std::size_t> clx = {0, 1, 10, 100, 1000, 10000, 100000, 1000000, 10000000};                   // main.cpp:21
    for (int s = 0; s < SHAPES; ++s) {
      if (!wanted(o.shapes, kShapeName[s])) continue;
      Shape shape = static_cast<Shape>(s);
      run<char>("CHAR", shape, keys);
      run<unsigned char>("UCHAR", shape, keys);
      run<short>("SHORT", shape, keys);
      run<unsigned short>("USHORT", shape, keys);
      run<int>("INT", shape, keys);
      run<unsigned int>("UINT", shape, keys);
      run<long long>("LL", shape, keys);
      run<unsigned long long>("ULL", shape, keys);
      run<float>("FLOAT", shape, keys);
      run<double>("DOUBLE", shape, keys);
      run<std::string>("STRING", shape, strs);
      run<Employee>("COMPLEX_INT", shape, clx, &Employee::age);
      run<Employee>("COMPLEX_LL", shape, clx, &Employee::id);
      run<Employee>("COMPLEX_FLOAT", shape, clx, &Employee::salary_f);
      run<Employee>("COMPLEX_DOUBLE", shape, clx, &Employee::salary);
      run<Employee>("COMPLEX_STR", shape, clx, &Employee::name);
    }
  }
};

std::vector<std::string> splitList(const char* arg) {
  std::vector<std::string> out;
  std::string cur;
  for (const char* p = arg;; ++p) {
    if (*p == ',' || *p == '\0') { if (!cur.empty()) out.push_back(cur); cur.clear(); if (!*p) break; }
    else cur += *p;
  }
  return out;
}

}  // namespace

int main(int argc, char** argv) {
  Matrix m;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto value = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
    if (a == "--testing") m.o.testing = true;
    else if (a == "--with-std") m.o.withStd = true;
    else if (a == "--iterations") m.o.iterations = std::max(1, std::atoi(value()));
    else if (a == "--max-size") m.o.maxSize = std::strtoull(value(), nullptr, 10);
    else if (a == "--types") m.o.types = splitList(value());
    else if (a == "--shapes") m.o.shapes = splitList(value());
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  {  // the first calls of a thread create its GPU context (probes, workspace) and the transfer lanes' pinned buffers;
     // keep that out of the first timed rows
    std::vector<unsigned> warm(8u << 20);
    for (std::size_t i = 0; i < warm.size(); ++i) warm[i] = static_cast<unsigned>((i * 2654435761u) >> 3);
    radix_sort::sort(warm, {}, true);
    radix_sort::sort(warm, {}, false);
  }
  const char* what = m.o.testing ? "--------TESTING STARTED--------" : "-------BENCHMARK STARTED-------";
  std::printf("%s%s\n%s\n%s\n", kRule, what, kRule, kRule);
  m.all();
  if (m.o.testing) std::printf("WRONG SORTS = %zu\n\n", m.wrong);
  what = m.o.testing ? "--------TESTING STOPPED--------" : "-------BENCHMARK STOPPED-------";
  std::printf("%s%s\n%s\n", kRule, what, kRule);
  return m.wrong ? 1 : 0;
}
