// rs_tune.hip -- developer harness: times one build of librsort_hip.so (path given) on synthetic
// keys, device-resident, and verifies order on the device.  Not part of the product or the tests.
//   hipcc -O2 --offload-arch=gfx950 tools/rs_tune.hip -o tools/rs_tune -ldl
//   tools/rs_tune <lib.so> <key_bytes> <log2n|n> <shape: random|sorted|reverse|nearly|dups|ramp32|prefix3> [iters] [value_bytes]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/rsort_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

__device__ inline uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
template <typename K> __global__ void gen(K* k, uint64_t n, int shape) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t umax = sizeof(K) == 8 ? ~0ull : 0xFFFFFFFFull;
  const uint64_t step = umax / (n - 1);
  for (; i < n; i += stride) {
    uint64_t v;
    if (shape == 0) v = mix(i);
    else if (shape == 4) v = mix(mix(i) & 255);
    else if (shape == 5) v = (i >> 5) * (step << 5);      // sorted, every value 32 times (what a float ramp looks like)
    else if (shape == 6) v = ((mix(i) & umax) >> 3) | (5ull << (sizeof(K) * 8 - 3));   // random below a shared 3-bit prefix (a rank of 8 after the exchange)
    else {
      uint64_t j = i;
      if (shape == 2) j = n - 1 - i;
      if (shape == 3 && (mix(i * 7 + 1) % 20) == 0) { uint64_t o = mix(i * 3 + 5) % n; j = o; }  // ~5% displaced
      v = j * step;
    }
    k[i] = (K)v;
  }
}
template <typename K> __global__ void check(const K* k, uint64_t n, unsigned* bad, unsigned long long* sum) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long s = 0; unsigned b = 0;
  for (; i < n; i += stride) { s += (unsigned long long)k[i]; if (i + 1 < n && k[i] > k[i + 1]) b++; }
  if (b) atomicAdd(bad, b);
  atomicAdd(sum, s);
}

int main(int argc, char** argv) {
  if (argc < 5) { printf("usage\n"); return 1; }
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
  auto f_create = (int (*)(rs_ctx**, int))dlsym(h, "rs_ctx_create");
  auto f_sort = (int (*)(rs_ctx*, void*, void*, size_t, int, int, int, int))dlsym(h, "rs_sort_keys_device");
  auto f_pairs = (int (*)(rs_ctx*, void*, void*, void*, void*, size_t, int, int, int, int, int))dlsym(h, "rs_sort_pairs_device");
  auto f_sync = (int (*)(rs_ctx*))dlsym(h, "rs_ctx_synchronize");
  auto f_prof = (int (*)(rs_ctx*, int))dlsym(h, "rs_ctx_set_profiling");
  auto f_get = (int (*)(rs_ctx*, float*, int, int*))dlsym(h, "rs_ctx_get_profile");
  auto f_err = (const char* (*)(const rs_ctx*))dlsym(h, "rs_last_error");
  int kb = atoi(argv[2]);
  uint64_t n = strtoull(argv[3], 0, 10); if (n <= 40) n = 1ull << n;
  int shape = !strcmp(argv[4], "random") ? 0 : !strcmp(argv[4], "sorted") ? 1 : !strcmp(argv[4], "reverse") ? 2 : !strcmp(argv[4], "nearly") ? 3 : !strcmp(argv[4], "ramp32") ? 5 : !strcmp(argv[4], "prefix3") ? 6 : 4;
  int iters = argc > 5 ? atoi(argv[5]) : 5;
  int vb = argc > 6 ? atoi(argv[6]) : 0;
  rs_ctx* c = nullptr;
  if (f_create(&c, 0)) { printf("ctx create failed\n"); return 1; }
  void *src, *work, *alt, *vals = nullptr, *valt = nullptr; unsigned* bad; unsigned long long* sums;
  // RS_TUNE_SHIFT=<bytes>: work / alt / vals / valt start 0, 1, 2, 3 shifts into their (over-sized) allocations -- does the
  // relative placement of the ping-pong buffers matter?
  const size_t shift = getenv("RS_TUNE_SHIFT") ? strtoull(getenv("RS_TUNE_SHIFT"), 0, 0) : 0;
  CK(hipMalloc(&src, n * kb)); CK(hipMalloc(&work, n * kb + 4 * shift)); CK(hipMalloc(&alt, n * kb + 4 * shift));
  alt = (char*)alt + shift;
  if (vb) { CK(hipMalloc(&vals, n * vb + 4 * shift)); CK(hipMalloc(&valt, n * vb + 4 * shift)); vals = (char*)vals + 2 * shift; valt = (char*)valt + 3 * shift; CK(hipMemset(vals, 0, n * vb)); }
  if (getenv("RS_TUNE_SHIFT")) printf("shift %zu: work %p alt %p vals %p valt %p\n", shift, work, alt, vals, valt);
  CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sums, 16)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(sums, 0, 16));
  if (kb == 4) gen<uint32_t><<<2048, 256>>>((uint32_t*)src, n, shape); else gen<uint64_t><<<2048, 256>>>((uint64_t*)src, n, shape);
  if (kb == 4) check<uint32_t><<<2048, 256>>>((uint32_t*)src, n, bad, sums); else check<uint64_t><<<2048, 256>>>((uint64_t*)src, n, bad, sums);
  CK(hipDeviceSynchronize());
  f_prof(c, 1);
  auto f_pre = (int (*)(rs_ctx*, int))dlsym(h, "rs_ctx_set_precheck");
  if (f_pre) f_pre(c, getenv("RS_TUNE_PRECHECK") ? 1 : 0);   // time the sort engine proper unless asked otherwise
  std::vector<float> tot, hist, pass;
  for (int it = 0; it < iters + 2; ++it) {
    CK(hipMemcpy(work, src, n * kb, hipMemcpyDeviceToDevice));
    CK(hipDeviceSynchronize());
    int rc = vb ? f_pairs(c, work, vals, alt, valt, n, kb, vb, 0, 0, kb * 8) : f_sort(c, work, alt, n, kb, 0, 0, kb * 8);
    if (rc) { printf("sort failed: %s\n", f_err(c)); return 1; }
    if (f_sync(c)) { printf("sync failed: %s\n", f_err(c)); return 1; }
    float ms[16]; int cnt = 0; f_get(c, ms, 16, &cnt);
    if (it >= 2) { float t = 0; for (int i = 0; i < cnt; ++i) t += ms[i]; tot.push_back(t); hist.push_back(ms[0]); for (int i = 1; i < cnt; ++i) pass.push_back(ms[i]); }
  }
  if (getenv("RS_TUNE_WALL")) {   // host-clock latency of one device-resident sort (call + wait), profiling off: median and minimum
    f_prof(c, 0);
    std::vector<double> w;
    for (int it = 0; it < 300; ++it) {
      CK(hipMemcpy(work, src, n * kb, hipMemcpyDeviceToDevice));
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      int rc = vb ? f_pairs(c, work, vals, alt, valt, n, kb, vb, 0, 0, kb * 8) : f_sort(c, work, alt, n, kb, 0, 0, kb * 8);
      if (rc || f_sync(c)) { printf("sort failed: %s\n", f_err(c)); return 1; }
      w.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(w.begin(), w.end());
    printf("wall us: median %.1f min %.1f   ", w[w.size() / 2], w.front());
  }
  CK(hipMemset(bad, 0, 4));
  if (kb == 4) check<uint32_t><<<2048, 256>>>((uint32_t*)work, n, bad, sums + 1); else check<uint64_t><<<2048, 256>>>((uint64_t*)work, n, bad, sums + 1);
  unsigned hb; unsigned long long hs[2];
  CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, sums, 16, hipMemcpyDeviceToHost));
  // a sort answered by the order pre-check, or one whose passes were all skipped, leaves no per-pass times: indexing the
  // empty vectors below was this harness' segmentation fault of round 1 (gpurun_out/tune16.log, rc 139)
  if (tot.empty()) tot.push_back(0.f);
  if (hist.empty()) hist.push_back(0.f);
  if (pass.empty()) pass.push_back(0.f);
  std::sort(tot.begin(), tot.end()); std::sort(hist.begin(), hist.end()); std::sort(pass.begin(), pass.end());
  double t = tot[tot.size() / 2];
  double canon = (double)n * (kb + 2.0 * kb * (kb + vb));
  printf("%-40s kb=%d vb=%d n=%llu %-7s total %.3f ms  hist %.3f  pass(med) %.3f (min %.3f max %.3f)  %.1f Gkeys/s  roofline %.1f%%  %s\n",
         strrchr(argv[1], '/') ? strrchr(argv[1], '/') + 1 : argv[1], kb, vb, (unsigned long long)n, argv[4], t, hist[hist.size() / 2],
         pass[pass.size() / 2], pass.front(), pass.back(), n / t / 1e6, canon / (t * 1e-3) / 8e12 * 100,
         (hb == 0 && hs[0] == hs[1]) ? "OK" : "WRONG");
  auto f_st = (int (*)(unsigned long long*))dlsym(h, "rs_debug_read_stamps");
  if (f_st) {
    std::vector<unsigned long long> st(16 * 4096);
    f_st(st.data());
    size_t tile_keys = getenv("RS_TUNE_TILE") ? atoi(getenv("RS_TUNE_TILE")) : ((kb <= 4 && vb <= 4) ? 16384 : 8192); size_t tiles = (n + tile_keys - 1) / tile_keys, cnt = std::min<size_t>(tiles / 64, 4096);
    const char* names[] = {"begin->ticket", "ticket->counted", "counted->ranked", "ranked->scanned", "scanned->staged", "staged->lookedback", "lookedback->written"};
    for (int k = 0; k < 7; ++k) {
      std::vector<double> d;
      for (size_t i = 8; i < cnt; ++i) d.push_back((double)(long long)(st[(k + 1) * 4096 + i] - st[k * 4096 + i]) * 0.01);
      std::sort(d.begin(), d.end());
      if (!d.empty()) printf("   %-22s median %.2f us  p10 %.2f  p90 %.2f\n", names[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
  }
  if (f_st && getenv("RS_TUNE_STAMP_ALL")) {   // few tiles (mid-size sorts): every stamped tile of the last pass, phase by phase
    std::vector<unsigned long long> st(16 * 4096);
    f_st(st.data());
    size_t tile_keys = getenv("RS_TUNE_TILE") ? atoi(getenv("RS_TUNE_TILE")) : ((kb <= 4 && vb <= 4) ? 16384 : 8192); size_t tiles = (n + tile_keys - 1) / tile_keys, cnt = std::min<size_t>((tiles + 63) / 64, 4096);
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < cnt; ++i) t0 = std::min(t0, st[i]);
    for (size_t i = 0; i < cnt; ++i) {
      printf("   tile %4zu begins %6.2f us:", i * 64, (st[i] - t0) * 0.01);
      for (int k = 0; k < 7; ++k) printf(" %6.2f", (double)(long long)(st[(k + 1) * 4096 + i] - st[k * 4096 + i]) * 0.01);
      printf("   trips %llu empty %llu walked %llu\n", st[8 * 4096 + i], st[9 * 4096 + i], st[10 * 4096 + i]);
    }
  }
  if (f_st && getenv("RS_TUNE_CHAINS")) {   // chain mode: when did each XCD's chain start and finish (stamped tiles only, last stamped pass)
    std::vector<unsigned long long> st(16 * 4096);
    f_st(st.data());
    size_t tile_keys = getenv("RS_TUNE_TILE") ? atoi(getenv("RS_TUNE_TILE")) : 8192; size_t tiles = (n + tile_keys - 1) / tile_keys, cnt = std::min<size_t>(tiles / 64, 4096);
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < cnt; ++i) t0 = std::min(t0, st[i]);
    const size_t per_chain = (tiles + 7) / 8;
    for (int c = 0; c < 8; ++c) {
      unsigned long long lo = ~0ull, hi = 0; std::vector<double> lat;
      for (size_t i = 0; i < cnt; ++i) {
        const size_t tile = i * 64;
        if (tile / per_chain != (size_t)c) continue;
        lo = std::min(lo, st[i]); hi = std::max(hi, st[7 * 4096 + i]);
        lat.push_back((double)(long long)(st[7 * 4096 + i] - st[i]) * 0.01);
      }
      std::sort(lat.begin(), lat.end());
      double mean = 0; for (double x : lat) mean += x; if (!lat.empty()) mean /= lat.size();
      printf("   chain %d: first tile begins %.1f us, last stamped tile written %.1f us; tile latency mean %.2f median %.2f p99 %.2f us\n", c, (lo - t0) * 0.01, (hi - t0) * 0.01,
             mean, lat.empty() ? 0. : lat[lat.size() / 2], lat.empty() ? 0. : lat[lat.size() * 99 / 100]);
    }
  }
  if (f_st) {
    std::vector<unsigned long long> st(16 * 4096);
    f_st(st.data());
    size_t tile_keys = getenv("RS_TUNE_TILE") ? atoi(getenv("RS_TUNE_TILE")) : ((kb <= 4 && vb <= 4) ? 16384 : 8192); size_t tiles = (n + tile_keys - 1) / tile_keys, cnt = std::min<size_t>(tiles / 64, 4096);
    const char* nm[] = {"lookback trips", "empty trips", "tiles walked"};
    for (int k = 0; k < 3; ++k) {
      std::vector<double> d;
      for (size_t i = 8; i < cnt; ++i) d.push_back((double)st[(8 + k) * 4096 + i]);
      std::sort(d.begin(), d.end());
      if (!d.empty()) printf("   %-22s median %.0f  p10 %.0f  p90 %.0f max %.0f\n", nm[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10], d.back());
    }
  }
  return (hb == 0 && hs[0] == hs[1]) ? 0 : 3;
}
