// frontend_bench.cpp -- end-to-end timing of include/radix_sort.hpp the way the reference's harness times a
// row (benchmark.cpp:147-186: copy outside the timer, time the call, mean of the iterations), host vectors
// in, host vectors out (PCIe included).  Rows: BASELINE config 1 (uint32 1e8) and config 4
// ({std::string key; long long data}, 1e7, projected key), beside std::sort / std::stable_sort.
//   g++ -std=c++20 -O2 -Iinclude tools/frontend_bench.cpp -Lradix-sort_amd/lib -lrsort_hip -o tools/frontend_bench
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <optional>
#include <random>
#include <string>
#include <vector>

#include "radix_sort.hpp"

struct Composite { std::string key; long long data; };
struct PodRow { int key; int a; long long b; };      // trivially copyable, 16 bytes: gathered on the device (rs_sort_rows)

template <class F> static double mean_ms(int iters, F&& f) {
  double total = 0;
  for (int i = 0; i < iters; i++) total += f();
  return total / iters;
}
template <class F> static double timed(F&& f) {
  auto t0 = std::chrono::steady_clock::now();
  f();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char** argv) {
  const size_t n_u32 = argc > 1 ? std::stoull(argv[1]) : 100000000ull;
  const size_t n_str = argc > 2 ? std::stoull(argv[2]) : 10000000ull;
  std::mt19937_64 gen(69);
  {
    std::vector<unsigned> src(n_u32);
    for (auto& x : src) x = static_cast<unsigned>(gen());
    std::vector<unsigned> v = src;
    radix_sort::sort(v, {}, true);                                   // warm-up: context, workspace
    double ms = mean_ms(3, [&] { v = src; return timed([&] { radix_sort::sort(v, {}, true); }); });
    bool ok = std::is_sorted(v.begin(), v.end());
    std::printf("uint32 randomized N=%zu radix_sort::sort(v,{},true) end-to-end: %.1f ms = %.3f Gkeys/s  %s\n", n_u32, ms,
                n_u32 / ms / 1e6, ok ? "OK" : "WRONG");
  }
  {
    std::uniform_int_distribution<int> len(0, 20), ch(0, 255);
    std::vector<Composite> src(n_str);
    for (size_t i = 0; i < n_str; i++) {
      src[i].key.resize(len(gen));
      for (auto& c : src[i].key) c = static_cast<char>(ch(gen));
      src[i].data = static_cast<long long>(i);
    }
    std::vector<Composite> v = src;
    radix_sort::sort(v, &Composite::key, true);
    double ms = mean_ms(3, [&] { v = src; return timed([&] { radix_sort::sort(v, &Composite::key, true); }); });
    std::vector<Composite> w = src;
    double ms_std = timed([&] { std::stable_sort(w.begin(), w.end(), [](const Composite& a, const Composite& b) { return a.key < b.key; }); });
    bool ok = true;
    for (size_t i = 0; ok && i < n_str; i++) ok = v[i].data == w[i].data;
    std::printf("{string key; long long data} randomized N=%zu sort(v,&key,true) end-to-end: %.1f ms = %.4f Gkeys/s (std::stable_sort %.1f ms)  %s\n",
                n_str, ms, n_str / ms / 1e6, ms_std, ok ? "OK" : "WRONG");
    if (std::getenv("FRONTEND_PHASES")) {   // the same call taken apart: where do the milliseconds go?
      namespace dev = radix_sort::internal::device;
      for (int it = 0; it < 2; it++) {
        v = src;
        auto proj = &Composite::key;
        const std::size_t threads = dev::helperThreads(v.size(), true);
        double t_asc = timed([&] { (void)dev::isAscending(v, proj); });
        std::optional<dev::PackedStrings<std::uint32_t>> p;      // (10^7 strings of at most 20 bytes: well below 4 GiB)
        double t_pack = timed([&] { p.emplace(dev::packStrings<std::uint32_t>(v, proj, threads, dev::stringRangeBytes(v, proj, threads))); });
        dev::Uninitialized<std::uint32_t> perm(v.size());
        rs_ctx* ctx = dev::context();
        rs_ctx_set_host_threads(ctx, static_cast<int>(dev::hostThreads(v.size(), true)));
        double t_sort = timed([&] { dev::check(ctx, rs_sort_strings32(ctx, p->blob.data(), p->offsets.data(), v.size(), perm.data(), 1)); });
        double t_apply = timed([&] { dev::applyPermutation(v, perm.data(), threads); });
        double t_free = timed([&] { p.reset(); });
        std::printf("   phases (%zu host threads): order check %.1f  pack %.1f  rs_sort_strings %.1f  apply permutation %.1f  free %.1f ms\n",
                    threads, t_asc, t_pack, t_sort, t_apply, t_free);
      }
    }
  }
  {
    std::vector<PodRow> src(n_str);
    for (size_t i = 0; i < n_str; i++) src[i] = {static_cast<int>(gen()), static_cast<int>(i), static_cast<long long>(i) * 5};
    std::vector<PodRow> v = src;
    radix_sort::sort(v, &PodRow::key, true);
    double ms = mean_ms(3, [&] { v = src; return timed([&] { radix_sort::sort(v, &PodRow::key, true); }); });
    std::vector<PodRow> w = src;
    double ms_std = timed([&] { std::stable_sort(w.begin(), w.end(), [](const PodRow& a, const PodRow& b) { return a.key < b.key; }); });
    bool ok = true;
    for (size_t i = 0; ok && i < n_str; i++) ok = v[i].key == w[i].key && v[i].a == w[i].a && v[i].b == w[i].b;
    std::printf("{int key; int a; long long b} randomized N=%zu sort(v,&key,true) end-to-end: %.1f ms = %.4f Gkeys/s (std::stable_sort %.1f ms)  %s\n",
                n_str, ms, n_str / ms / 1e6, ms_std, ok ? "OK" : "WRONG");
  }
  return 0;
}
