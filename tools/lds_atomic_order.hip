// lds_atomic_order.hip -- does ds_add_rtn_u32 hand out return values in LANE ORDER when several lanes of one
// wave instruction hit the same LDS word?  (Undocumented; probed here because a lane-ordered return would
// replace the 8-ballot match in the ranking loop.)   hipcc -O2 --offload-arch=gfx950 tools/lds_atomic_order.hip -o tools/lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void probe(unsigned long long* violations, unsigned long long* checked, int rounds, int mode) {
  __shared__ unsigned hist[8][256];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = lane; i < 256; i += 64) hist[wave][i] = 0;
  __syncthreads();
  unsigned long long bad = 0, cnt = 0;
  for (int r = 0; r < rounds; ++r) {
    uint64_t h = mix(((uint64_t)blockIdx.x << 40) ^ ((uint64_t)r << 16) ^ threadIdx.x);
    unsigned d;
    switch (mode) {
      case 0: d = h & 255; break;                // uniform digits
      case 1: d = h & 3; break;                  // 4 distinct: 16-way conflicts
      case 2: d = 7; break;                      // all equal: 64-way
      case 3: d = (lane >> 2) + (h & 1); break;  // neighbouring lanes collide
      default: d = (h >> 8) % 37; break;
    }
    bool active = mode != 5 || ((h >> 20) & 3) != 0;   // mode 5: random inactive lanes
    unsigned before = hist[wave][d];
    unsigned long long peers = 0;
    for (int l = 0; l < 64; ++l) { unsigned dl = __shfl(d, l); bool al = __shfl((int)active, l); if (al && dl == d && l < (int)lane) peers |= 1ull << l; }
    __builtin_amdgcn_wave_barrier();
    unsigned got = 0;
    if (active) got = atomicAdd(&hist[wave][d], 1u);
    __builtin_amdgcn_wave_barrier();
    if (active) { ++cnt; if (got != before + (unsigned)__popcll(peers)) ++bad; }
  }
  atomicAdd(violations, bad); atomicAdd(checked, cnt);
}
int main() {
  unsigned long long *v, *c; hipMalloc(&v, 8); hipMalloc(&c, 8);
  for (int mode = 0; mode < 6; ++mode) {
    hipMemset(v, 0, 8); hipMemset(c, 0, 8);
    probe<<<2048, 512>>>(v, c, 2000, mode);
    unsigned long long hv, hc; hipMemcpy(&hv, v, 8, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    printf("mode %d: checked %llu lane-ops, violations %llu\n", mode, hc, hv);
  }
  return 0;
}
