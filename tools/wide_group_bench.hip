// wide_group_bench.hip -- the 8-byte group kernel on its own: G groups of L random 48-bit keys each (top 16 bits = group).
//   hipcc -O3 --offload-arch=gfx950 -Iradix-sort_amd/csrc tools/wide_group_bench.hip -o tools/abl/wide_group_bench [-DRS_FIX_TOP=0]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <random>
#include <type_traits>
#include <vector>
__device__ unsigned g_fix_fallbacks;
#define RS_FIX_STATS (&g_fix_fallbacks)
#include "rsort_kernels.hpp"
#include "rsort_groups.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

int main(int argc, char** argv) {
  const unsigned G = argc > 1 ? atoi(argv[1]) : 8192, L = argc > 2 ? atoi(argv[2]) : 16384;
  const size_t n = size_t(G) * L;
  std::vector<uint64_t> h(n);
  std::mt19937_64 rng(5);
  for (size_t i = 0; i < n; ++i) h[i] = (uint64_t(i / L) << 48) | (rng() >> 16);
  std::vector<unsigned> start(65537);
  for (unsigned g = 0; g <= 65536; ++g) start[g] = static_cast<unsigned>(std::min<size_t>(size_t(g) * L, n));
  uint64_t *d, *src; unsigned *ds, *flag, *big;
  CK(hipMalloc(&d, n * 8)); CK(hipMalloc(&src, n * 8)); CK(hipMalloc(&ds, 65537 * 4)); CK(hipMalloc(&flag, 64)); CK(hipMalloc(&big, 4096));
  CK(hipMemcpy(src, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, start.data(), 65537 * 4, hipMemcpyHostToDevice));
  CK(hipMemset(flag, 0, 64)); CK(hipMemset(big, 0, 4096));
  rsk::GroupArgs a{};
  a.keys = d; a.start = ds; a.groups = 65536; a.low_bits = argc > 3 ? atoi(argv[3]) : 48; a.big_list = big; a.big_cap = 100; a.max_vec = 1024 * 17; a.n = static_cast<unsigned>(n); a.error_flag = flag;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CK(hipMemcpy(d, src, n * 8, hipMemcpyDeviceToDevice));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint64_t, rsk::NoVal, 1024, 17>), dim3(G), dim3(1024), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  std::vector<uint64_t> out(n);
  CK(hipMemcpy(out.data(), d, n * 8, hipMemcpyDeviceToHost));
  bool ok = std::is_sorted(out.begin(), out.end());
  unsigned stats[16]; CK(hipMemcpy(stats, flag, 64, hipMemcpyDeviceToHost));
  CK(hipMemcpyFromSymbol(&stats[4], HIP_SYMBOL(g_fix_fallbacks), 4));
  printf("G=%u L=%u  %.3f ms  %.2f us/group/CU  %.2f TB/s  sorted=%d  flag=%u fallbacks=%u\n", G, L, best, best * 1e3 * 256 / G, n * 16 / best / 1e9, int(ok), stats[0], stats[4]);
  return ok || argc > 3 ? 0 : 1;
}
