// rocprim_yardstick.hip -- the vendor library's keys-only radix sort (rocPRIM, what hipCUB and torch.sort call) on the same
// inputs and clock as tools/rs_tune, as an outside yardstick for the sweep.  Not part of the product and not a baseline of the
// reference (which is a CPU library).
//   hipcc -O3 --offload-arch=gfx950 tools/rocprim_yardstick.hip -o tools/rocprim_yardstick
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static __device__ __host__ inline uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
template <typename K> __global__ void gen(K* k, uint64_t n) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) k[i] = (K)mix(i);
}

template <typename K> int run(uint64_t n) {
  K *src, *in, *out; void* tmp = nullptr; size_t tmp_bytes = 0;
  CK(hipMalloc(&src, n * sizeof(K))); CK(hipMalloc(&in, n * sizeof(K))); CK(hipMalloc(&out, n * sizeof(K)));
  gen<K><<<2048, 256>>>(src, n);
  CK(rocprim::radix_sort_keys(nullptr, tmp_bytes, in, out, n));
  CK(hipMalloc(&tmp, tmp_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int it = 0; it < 7; ++it) {
    CK(hipMemcpy(in, src, n * sizeof(K), hipMemcpyDeviceToDevice));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    CK(rocprim::radix_sort_keys(tmp, tmp_bytes, in, out, n));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float m; CK(hipEventElapsedTime(&m, e0, e1)); if (it >= 2) ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  printf("rocprim::radix_sort_keys %zu-byte keys n=%llu random: %.3f ms  %.1f Gkeys/s  (temporary storage %.0f MB)\n", sizeof(K),
         (unsigned long long)n, ms[ms.size() / 2], n / ms[ms.size() / 2] / 1e6, tmp_bytes / 1e6);
  hipFree(src); hipFree(in); hipFree(out); hipFree(tmp);
  return 0;
}

int main(int argc, char** argv) {
  uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 30; if (n <= 40) n = 1ull << n;
  if (run<uint32_t>(n)) return 1;
  if (run<uint64_t>(n)) return 1;
  return 0;
}
