// group_sort_bench.hip -- prices rsk::group_sort_kernel (rsort_groups.hpp) on its own: an array already grouped by its top
// 16 bits (what two onesweep passes leave), groups of n / 65536 keys +- jitter, random low 16 bits; checks the result
// (ascending, same sum) and times the launch.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iradix-sort_amd/csrc -Iinclude tools/group_sort_bench.hip -o tools/group_sort_bench
//   tools/group_sort_bench [log2 n = 30] [jitter = 400]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "rsort_groups.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned long long h) {
  h += 0x9E3779B97F4A7C15ull; h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull; h = (h ^ (h >> 27)) * 0x94D049BB133111EBull; h ^= h >> 31;
  return static_cast<unsigned>(h);
}
__global__ void fill_kernel(unsigned* keys, const unsigned* start, int mode) {
  const unsigned g = blockIdx.x, s = start[g], e = start[g + 1];
  for (unsigned i = s + threadIdx.x; i < e; i += blockDim.x) {
    unsigned lo = mix(i) & 0xFFFFu;
    if (mode == 1) lo = ((i - s) * 4u) & 0xFFFFu;            // a ramp inside the group
    if (mode == 2) lo = 0xFFFFu - (((i - s) * 4u) & 0xFFFFu);
    keys[i] = (g << 16) | lo;
  }
}
__global__ void check_kernel(const unsigned* keys, size_t n, unsigned long long* out /* [0] descents, [1] sum */) {
  unsigned long long sum = 0, bad = 0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    sum += keys[i];
    if (i + 1 < n && keys[i] > keys[i + 1]) ++bad;
  }
  atomicAdd(&out[1], sum);
  if (bad) atomicAdd(&out[0], bad);
}

template <int BLOCK, int MAXV, int MINB>
static void run(const char* name, unsigned* keys, unsigned* pristine, const unsigned* d_start, unsigned groups, size_t n, unsigned* big, unsigned long long* chk) {
  rsk::GroupArgs a{};
  a.keys = keys; a.start = d_start; a.groups = groups; a.low_bits = 16; a.big_list = big; a.big_cap = 65536; a.max_vec = BLOCK * MAXV; a.n = static_cast<unsigned>(n); a.error_flag = big + 65536;
  unsigned long long want[2], got[2];
  CK(hipMemset(chk, 0, 16));
  check_kernel<<<4096, 256>>>(pristine, n, chk);
  CK(hipMemcpy(want, chk, 16, hipMemcpyDeviceToHost));
  std::vector<float> t;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 7; ++it) {
    CK(hipMemcpy(keys, pristine, n * 4, hipMemcpyDeviceToDevice));
    CK(hipMemset(big, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rsk::group_sort_kernel<unsigned, BLOCK, MAXV, MINB>), dim3(groups), dim3(BLOCK), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 2) t.push_back(ms);
  }
  CK(hipGetLastError());
  std::sort(t.begin(), t.end());
  CK(hipMemset(chk, 0, 16));
  check_kernel<<<4096, 256>>>(keys, n, chk);
  CK(hipMemcpy(got, chk, 16, hipMemcpyDeviceToHost));
  unsigned nbig; CK(hipMemcpy(&nbig, big, 4, hipMemcpyDeviceToHost));
  printf("%-28s %.3f ms  %.2f TB/s  descents %llu  sum %s  left to the big-group kernel %u\n", name, t[t.size() / 2], 2.0 * n * 4 / t[t.size() / 2] / 1e9,
         got[0], got[1] == want[1] ? "same" : "DIFFERENT", nbig);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  const int jitter = argc > 2 ? atoi(argv[2]) : 400;
  const int mode = argc > 3 ? atoi(argv[3]) : 0;
  const size_t n = size_t(1) << lg;
  const unsigned groups = 65536;
  std::vector<unsigned> start(groups + 1);
  const size_t avg = n / groups;
  unsigned long long h = 12345;
  for (unsigned g = 0; g <= groups; ++g) {
    h = h * 6364136223846793005ull + 1442695040888963407ull;
    long long j = g == 0 || g == groups || jitter == 0 ? 0 : static_cast<long long>((h >> 33) % (2 * jitter + 1)) - jitter;
    long long s = static_cast<long long>(g * avg) + j;
    start[g] = static_cast<unsigned>(std::max<long long>(0, std::min<long long>(s, n)));
  }
  start[groups] = static_cast<unsigned>(n == (size_t(1) << 32) ? 0xFFFFFFFFu : n);
  unsigned *keys, *pristine, *d_start, *big; unsigned long long* chk;
  CK(hipMalloc(&keys, n * 4)); CK(hipMalloc(&pristine, n * 4)); CK(hipMalloc(&d_start, (groups + 1) * 4)); CK(hipMalloc(&big, (65536 + 1) * 4)); CK(hipMalloc(&chk, 16));
  CK(hipMemcpy(d_start, start.data(), (groups + 1) * 4, hipMemcpyHostToDevice));
  fill_kernel<<<groups, 256>>>(pristine, d_start, mode);
  CK(hipDeviceSynchronize());
  printf("n = 2^%d, %u groups of %zu +- %d keys, mode %d\n", lg, groups, avg, jitter, mode);
#if RS_GROUP_ABL
  printf("ablation %d: ", RS_GROUP_ABL);
#else
  run<512, 10, 3>("512 x 10 vectors, 3 per CU", keys, pristine, d_start, groups, n, big, chk);
  run<1024, 5, 2>("1024 x 5 vectors, 2 per CU", keys, pristine, d_start, groups, n, big, chk);
#endif
  run<512, 10, 2>("512 x 10 vectors, 2 per CU", keys, pristine, d_start, groups, n, big, chk);
  return 0;
}
