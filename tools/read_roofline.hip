// read_roofline.hip -- how fast can a kernel READ an array on this GPU, and how fast can one COPY it?  The yardstick for the digit histogram kernel
// (one read of the keys, nothing written): the same 16-byte loads, persistent and one-shot grids, several loads in flight.
//   hipcc -O3 --offload-arch=gfx950 tools/read_roofline.hip -o tools/read_roofline && tools/read_roofline [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void read_kernel(const uint4* in, size_t vecs, unsigned* sink) {
  const size_t stride = static_cast<size_t>(gridDim.x) * BLOCK;
  size_t i = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
  unsigned acc = 0;
  for (; i + (UNROLL - 1) * stride < vecs; i += UNROLL * stride) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < vecs; i += stride) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;      // (keeps the loads alive)
}

// each workgroup reads one contiguous slab (what a tile-per-workgroup kernel does)
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void read_slab_kernel(const uint4* in, size_t vecs, unsigned* sink) {
  const size_t base = static_cast<size_t>(blockIdx.x) * BLOCK * UNROLL + threadIdx.x;
  unsigned acc = 0;
  uint4 v[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) v[u] = base + u * BLOCK < vecs ? in[base + u * BLOCK] : uint4{0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  if (acc == 0x12345678u) *sink = acc;
}

// persistent, but each workgroup streams through its own contiguous share of the array
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void read_range_kernel(const uint4* in, size_t vecs, unsigned* sink) {
  const size_t per = (vecs + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = lo + per < vecs ? lo + per : vecs;
  unsigned acc = 0;
  size_t i = lo + threadIdx.x;
  for (; i + (UNROLL - 1) * BLOCK < hi; i += UNROLL * BLOCK) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = in[i + u * BLOCK];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < hi; i += BLOCK) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}

// copy: what a kernel that reads an array once and writes it once reaches (the yardstick for the sweeps and the group sort,
// whose algorithmic bytes are one read + one write of the keys)
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void copy_kernel(const uint4* in, uint4* out, size_t vecs) {
  const size_t stride = static_cast<size_t>(gridDim.x) * BLOCK;
  size_t i = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < vecs; i += UNROLL * stride) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) out[i + u * stride] = v[u];
  }
  for (; i < vecs; i += stride) out[i] = in[i];
}
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void copy_slab_kernel(const uint4* in, uint4* out, size_t vecs) {
  const size_t base = static_cast<size_t>(blockIdx.x) * BLOCK * UNROLL + threadIdx.x;
  uint4 v[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) if (base + u * BLOCK < vecs) v[u] = in[base + u * BLOCK];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) if (base + u * BLOCK < vecs) out[base + u * BLOCK] = v[u];
}

template <typename F> static double time_ms(F&& launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> t;
  for (int it = 0; it < 7; ++it) {
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 4096ull) << 20;
  const size_t vecs = bytes / 16;
  uint4* in; unsigned* sink;
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(in, 1, bytes)); CK(hipDeviceSynchronize());
  auto report = [&](const char* what, double ms) { printf("%-58s %.3f ms  %.2f TB/s\n", what, ms, bytes / ms / 1e9); };
  report("persistent 512 x 1024 threads, 4 loads in flight", time_ms([&] { read_kernel<1024, 4><<<512, 1024>>>(in, vecs, sink); }));
  report("persistent 256 x 1024 threads, 4 loads in flight", time_ms([&] { read_kernel<1024, 4><<<256, 1024>>>(in, vecs, sink); }));
  report("persistent 512 x 1024 threads, 8 loads in flight", time_ms([&] { read_kernel<1024, 8><<<512, 1024>>>(in, vecs, sink); }));
  report("persistent 2048 x 256 threads, 4 loads in flight", time_ms([&] { read_kernel<256, 4><<<2048, 256>>>(in, vecs, sink); }));
  report("persistent 4096 x 256 threads, 2 loads in flight", time_ms([&] { read_kernel<256, 2><<<4096, 256>>>(in, vecs, sink); }));
  report("persistent 1024 x 512 threads, 4 loads in flight", time_ms([&] { read_kernel<512, 4><<<1024, 512>>>(in, vecs, sink); }));
  report("persistent 512 x 1024, own contiguous range, 4 loads", time_ms([&] { read_range_kernel<1024, 4><<<512, 1024>>>(in, vecs, sink); }));
  report("persistent 256 x 1024, own contiguous range, 4 loads", time_ms([&] { read_range_kernel<1024, 4><<<256, 1024>>>(in, vecs, sink); }));
  report("persistent 2048 x 256, own contiguous range, 4 loads", time_ms([&] { read_range_kernel<256, 4><<<2048, 256>>>(in, vecs, sink); }));
  report("persistent 512 x 1024, own contiguous range, 2 loads", time_ms([&] { read_range_kernel<1024, 2><<<512, 1024>>>(in, vecs, sink); }));
  report("one slab of 64 KB per workgroup (1024 threads x 4 loads)", time_ms([&] { read_slab_kernel<1024, 4><<<static_cast<unsigned>((vecs + 4095) / 4096), 1024>>>(in, vecs, sink); }));
  report("one slab of 32 KB per workgroup (512 threads x 4 loads)", time_ms([&] { read_slab_kernel<512, 4><<<static_cast<unsigned>((vecs + 2047) / 2048), 512>>>(in, vecs, sink); }));
  report("one slab of 16 KB per workgroup (256 threads x 4 loads)", time_ms([&] { read_slab_kernel<256, 4><<<static_cast<unsigned>((vecs + 1023) / 1024), 256>>>(in, vecs, sink); }));
  // ---- copies: bytes = read + write ----
  uint4* out; CK(hipMalloc(&out, bytes));
  auto report2 = [&](const char* what, double ms) { printf("%-58s %.3f ms  %.2f TB/s (read + write)\n", what, ms, 2.0 * bytes / ms / 1e9); };
  report2("hipMemcpyAsync device to device", time_ms([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }));
  report2("copy, persistent 512 x 1024 threads, 4 in flight", time_ms([&] { copy_kernel<1024, 4><<<512, 1024>>>(in, out, vecs); }));
  report2("copy, persistent 1024 x 512 threads, 4 in flight", time_ms([&] { copy_kernel<512, 4><<<1024, 512>>>(in, out, vecs); }));
  report2("copy, persistent 2048 x 256 threads, 4 in flight", time_ms([&] { copy_kernel<256, 4><<<2048, 256>>>(in, out, vecs); }));
  report2("copy, persistent 768 x 512 threads, 8 in flight", time_ms([&] { copy_kernel<512, 8><<<768, 512>>>(in, out, vecs); }));
  report2("copy, one 32 KB slab per workgroup (512 x 4)", time_ms([&] { copy_slab_kernel<512, 4><<<static_cast<unsigned>((vecs + 2047) / 2048), 512>>>(in, out, vecs); }));
  report2("copy, one 64 KB slab per workgroup (1024 x 4)", time_ms([&] { copy_slab_kernel<1024, 4><<<static_cast<unsigned>((vecs + 4095) / 4096), 1024>>>(in, out, vecs); }));
  report2("copy, one 16 KB slab per workgroup (256 x 4)", time_ms([&] { copy_slab_kernel<256, 4><<<static_cast<unsigned>((vecs + 1023) / 1024), 256>>>(in, out, vecs); }));
  return 0;
}
