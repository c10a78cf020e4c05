// scatter_roofline.hip -- what HBM delivers for the onesweep access pattern without any sorting work.
//   hipcc -O3 --offload-arch=gfx950 tools/scatter_roofline.hip -o tools/scatter_roofline
// One workgroup per 64 KB tile: reads the tile front to back, writes it as `streams` runs of `run` bytes, run r of
// tile t going to stream (r % streams) at the position the t-th tile of a uniform-digit sort would write it.
// streams = 1 is a plain copy; streams = 256, run = 256 is a uint32 onesweep pass on uniform random digits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// shift16: every run is moved by this many 16-byte units, so that runs no longer start on a 128-byte line and the
//          line at each run boundary is shared by two tiles;
// group:   tiles are dealt to the XCDs in groups of `group` consecutive tiles (workgroup b runs on XCD b % 8), so
//          that the two tiles sharing a line mostly run on the same XCD and meet in one L2.
__global__ __launch_bounds__(1024) void scatter_kernel(const uint4* in, uint4* out, unsigned tiles, unsigned streams, unsigned run16 /* 16-byte units per run */,
                                                       unsigned shift16, unsigned group) {
  unsigned t = blockIdx.x;
  if (group > 1) {
    const unsigned xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    t = (slot / group) * (8 * group) + xcd * group + slot % group;
    if (t >= tiles) return;
  }
  const unsigned tile16 = 65536 / 16;                       // 4096 uint4 per tile
  const uint4* src = in + static_cast<size_t>(t) * tile16;
  const size_t stream16 = static_cast<size_t>(tiles) * tile16 / streams;   // uint4 per stream
  const unsigned runs_per_stream = tile16 / run16 / streams;                // runs of this tile per stream
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned j = i * 1024 + threadIdx.x;             // uint4 index in the tile
    const uint4 v = src[j];
    const unsigned r = j / run16, o = j % run16;
    const unsigned s = r % streams, k = r / streams;       // stream and the run's index among this tile's runs of that stream
    out[static_cast<size_t>(s) * stream16 + (static_cast<size_t>(t) * runs_per_stream + k) * run16 + o + shift16] = v;
  }
}

// The same pattern with 4-byte loads and stores per lane (what a sort kernel that moves single keys issues): thread t moves
// dwords t, t + 1024, ... of the tile, 16 per thread.
__global__ __launch_bounds__(1024) void scatter_kernel_dword(const unsigned* in, unsigned* out, unsigned tiles, unsigned streams, unsigned run4,
                                                             unsigned shift4, unsigned group) {
  unsigned t = blockIdx.x;
  if (group > 1) {
    const unsigned xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    t = (slot / group) * (8 * group) + xcd * group + slot % group;
    if (t >= tiles) return;
  }
  const unsigned tile4 = 65536 / 4;
  const unsigned* src = in + static_cast<size_t>(t) * tile4;
  const size_t stream4 = static_cast<size_t>(tiles) * tile4 / streams;
  const unsigned runs_per_stream = tile4 / run4 / streams;
  unsigned v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = src[i * 1024 + threadIdx.x];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned j = i * 1024 + threadIdx.x;
    const unsigned r = j / run4, o = j % run4;
    const unsigned s = r % streams, k = r / streams;
    out[static_cast<size_t>(s) * stream4 + (static_cast<size_t>(t) * runs_per_stream + k) * run4 + o + shift4] = v[i];
  }
}

// Chain order: persistent workgroups, chain = XCC id, tiles of a chain drawn front to back from the chain's own counter (the
// order of the chain-mode sweep): every run boundary is shared by two tiles of one XCD.
__device__ inline unsigned xcc_id_() { return static_cast<unsigned>(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20)) & 7u; }
// The shipped chain-mode sweep's data path with the sorting work taken out: persistent workgroups of BLOCK threads, chain =
// XCC id, tiles of TILE_BYTES drawn from the chain's counter; every wave brings its segment in by LDS-DMA (16 bytes per lane),
// the workgroup meets at a barrier, and thread t writes staged elements t, t + BLOCK, ... with one T-sized store each to where
// a uniform-digit sort would put them (256 streams).  No ranking, no scan, no look-back: what is left is what the memory
// system gives this access pattern at the sweep's residency (LDS footprint = staging area + `lds` bytes of ballast).
template <typename T, unsigned BLOCK, unsigned TILE_BYTES, unsigned WGS>
__global__ __launch_bounds__(BLOCK, (WGS * BLOCK + 255) / 256) void sweep_shape_kernel(const T* in, T* out, unsigned tiles, unsigned run_e, unsigned shift_e, unsigned* tickets) {
  __shared__ unsigned s_tile;
  __shared__ __attribute__((aligned(16))) unsigned char stage[TILE_BYTES];
  extern __shared__ unsigned char lds_ballast[];
  if (run_e == 0xFFFFFFFFu) lds_ballast[threadIdx.x] = 0;      // (never true: keeps the dynamic LDS allocated)
  constexpr unsigned tile_e = TILE_BYTES / sizeof(T), per_thread = tile_e / BLOCK, streams = 256;
  constexpr unsigned seg_bytes = TILE_BYTES / (BLOCK / 64);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x / 64;
  const unsigned chain = xcc_id_(), chain_tiles = (tiles + 7) / 8;
  const unsigned first = chain * chain_tiles, end = min(first + chain_tiles, tiles);
  const unsigned stream_e = static_cast<unsigned>(static_cast<size_t>(tiles) * tile_e / streams);
  const unsigned runs_per_stream = tile_e / run_e / streams;
  const unsigned lg_run = 31 - __builtin_clz(run_e);
  for (;;) {
    if (threadIdx.x == 0) s_tile = first + atomicAdd(&tickets[chain * 32], 1u);
    __syncthreads();
    const unsigned t = s_tile;
    if (t >= end) break;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(in) + static_cast<size_t>(t) * TILE_BYTES + wave * seg_bytes + lane * 16;
    unsigned char* land = stage + wave * seg_bytes;
#pragma unroll
    for (unsigned b = 0; b < seg_bytes; b += 1024)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b), (__attribute__((address_space(3))) void*)(land + b), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const T* st = reinterpret_cast<const T*>(stage);
#pragma unroll
    for (unsigned i = 0; i < per_thread; ++i) {
      const unsigned j = i * BLOCK + threadIdx.x;
      const unsigned r = j >> lg_run, o = j & (run_e - 1);
      const unsigned sid = r & (streams - 1), k = r >> 8;
      out[sid * stream_e + ((t * runs_per_stream + k) << lg_run) + o + shift_e] = st[j];
    }
    __syncthreads();
  }
}

// The same with FOUR consecutive staged 4-byte elements per lane and ONE 16-byte store at a 4-byte aligned address (runs are
// multiples of four elements here, so a lane's four elements always share a run): does the write side of a uint32 sweep
// get cheaper when the address unit sees a quarter of the store instructions?
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <unsigned BLOCK, unsigned TILE_BYTES, unsigned WGS>
__global__ __launch_bounds__(BLOCK, (WGS * BLOCK + 255) / 256) void sweep_shape_kernel_x4(const unsigned* in, unsigned* out, unsigned tiles, unsigned run_e, unsigned shift_e, unsigned* tickets) {
  __shared__ unsigned s_tile;
  __shared__ __attribute__((aligned(16))) unsigned char stage[TILE_BYTES];
  extern __shared__ unsigned char lds_ballast[];
  if (run_e == 0xFFFFFFFFu) lds_ballast[threadIdx.x] = 0;
  constexpr unsigned tile_e = TILE_BYTES / 4, per_thread = tile_e / BLOCK / 4, streams = 256;
  constexpr unsigned seg_bytes = TILE_BYTES / (BLOCK / 64);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x / 64;
  const unsigned chain = xcc_id_(), chain_tiles = (tiles + 7) / 8;
  const unsigned first = chain * chain_tiles, end = min(first + chain_tiles, tiles);
  const unsigned stream_e = static_cast<unsigned>(static_cast<size_t>(tiles) * tile_e / streams);
  const unsigned runs_per_stream = tile_e / run_e / streams;
  const unsigned lg_run = 31 - __builtin_clz(run_e);
  for (;;) {
    if (threadIdx.x == 0) s_tile = first + atomicAdd(&tickets[chain * 32], 1u);
    __syncthreads();
    const unsigned t = s_tile;
    if (t >= end) break;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(in) + static_cast<size_t>(t) * TILE_BYTES + wave * seg_bytes + lane * 16;
    unsigned char* land = stage + wave * seg_bytes;
#pragma unroll
    for (unsigned b = 0; b < seg_bytes; b += 1024)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b), (__attribute__((address_space(3))) void*)(land + b), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const u32x4* st = reinterpret_cast<const u32x4*>(stage);
#pragma unroll
    for (unsigned i = 0; i < per_thread; ++i) {
      const unsigned q = i * BLOCK + threadIdx.x, j = q * 4;
      const unsigned r = j >> lg_run, o = j & (run_e - 1);
      const unsigned sid = r & (streams - 1), k = r >> 8;
      unsigned* dst = out + (sid * stream_e + ((t * runs_per_stream + k) << lg_run) + o + shift_e);
      const u32x4 v = st[q];
      asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
    }
    __syncthreads();
  }
}

// BLOCK x TILE_BYTES: 1024 x 64 KB is the one-chain sweep's tile; 512 x 32 KB (uint32) / 512 x 64 KB (uint64) with `lds` bytes of
// dynamic LDS that hold the residency at 3 / 2 workgroups per CU are the shipped chain-mode sweeps' shape.
template <typename T, unsigned BLOCK = 1024, unsigned TILE_BYTES = 65536, unsigned WGS = 1>   // T = unsigned (4-byte lanes) or uint2 (8-byte lanes)
__global__ __launch_bounds__(BLOCK, (WGS * BLOCK + 255) / 256) void scatter_kernel_chain(const T* in, T* out, unsigned tiles, unsigned streams, unsigned run_e /* elements per run */,
                                                             unsigned shift_e, unsigned* tickets) {
  __shared__ unsigned s_tile;
  extern __shared__ unsigned char lds_ballast[];
  if (run_e == 0xFFFFFFFFu) lds_ballast[threadIdx.x] = 0;      // (never true: keeps the dynamic LDS allocated)
  constexpr unsigned tile_e = TILE_BYTES / sizeof(T);
  constexpr unsigned per_thread = tile_e / BLOCK;
  const unsigned chain = xcc_id_(), chain_tiles = (tiles + 7) / 8;
  const unsigned first = chain * chain_tiles, end = min(first + chain_tiles, tiles);
  const size_t stream_e = static_cast<size_t>(tiles) * tile_e / streams;
  const unsigned runs_per_stream = tile_e / run_e / streams;
  const unsigned lg_run = 31 - __builtin_clz(run_e), lg_streams = 31 - __builtin_clz(streams);
  for (;;) {
    if (threadIdx.x == 0) s_tile = first + atomicAdd(&tickets[chain * 32], 1u);
    __syncthreads();
    const unsigned t = s_tile;
    __syncthreads();
    if (t >= end) break;
    const T* src = in + static_cast<size_t>(t) * tile_e;
    T v[per_thread];
#pragma unroll
    for (unsigned i = 0; i < per_thread; ++i) v[i] = src[i * BLOCK + threadIdx.x];
#pragma unroll
    for (unsigned i = 0; i < per_thread; ++i) {
      const unsigned j = i * BLOCK + threadIdx.x;                  // (run_e and streams are powers of two)
      const unsigned r = j >> lg_run, o = j & (run_e - 1);
      const unsigned st = r & (streams - 1), k = r >> lg_streams;
      // (32-bit element index: the buffers hold at most 2^32 bytes, and a 64-bit address per pending store would spill)
      const unsigned idx = st * static_cast<unsigned>(stream_e) + ((t * runs_per_stream + k) << lg_run) + o + shift_e;
      out[idx] = v[i];
    }
  }
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? strtoull(argv[1], 0, 10) << 20 : 4096ull << 20;
  const unsigned tiles = bytes / 65536;
  uint4 *in, *out;
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes + 4096));
  CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned cfg[][4] = {{1, 65536, 0, 1}, {16, 4096, 0, 1}, {64, 1024, 0, 1}, {128, 512, 0, 1}, {256, 256, 0, 1}, {512, 128, 0, 1}, {1024, 64, 0, 1},
                             {256, 256, 1, 1}, {256, 256, 4, 1}, {256, 256, 4, 8}, {256, 256, 4, 32}, {256, 256, 0, 8}, {1, 65536, 4, 1}};
  for (auto& c : cfg) {
    std::vector<float> ms;
    for (int it = 0; it < 7; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(scatter_kernel, dim3(tiles), dim3(1024), 0, 0, in, out, tiles, c[0], c[1] / 16, c[2], c[3]);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float m; CK(hipEventElapsedTime(&m, e0, e1)); if (it >= 2) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    printf("streams %4u run %6u B shift %3u B xcd-group %2u : %.3f ms  %.2f TB/s (read+write)\n", c[0], c[1], c[2] * 16, c[3], ms[ms.size() / 2], 2.0 * bytes / ms[ms.size() / 2] / 1e9);
  }
  const unsigned cfg4[][4] = {{1, 65536, 0, 1}, {256, 256, 0, 1}, {256, 256, 64, 1}, {256, 256, 64, 8}, {256, 256, 20, 8}, {256, 256, 0, 8}};
  for (auto& c : cfg4) {
    std::vector<float> ms;
    for (int it = 0; it < 7; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(scatter_kernel_dword, dim3(tiles), dim3(1024), 0, 0, reinterpret_cast<const unsigned*>(in), reinterpret_cast<unsigned*>(out), tiles, c[0], c[1] / 4, c[2] / 4, c[3]);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float m; CK(hipEventElapsedTime(&m, e0, e1)); if (it >= 2) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    printf("4-byte lanes: streams %4u run %6u B shift %3u B xcd-group %2u : %.3f ms  %.2f TB/s (read+write)\n", c[0], c[1], c[2], c[3], ms[ms.size() / 2], 2.0 * bytes / ms[ms.size() / 2] / 1e9);
  }
  unsigned* tickets; CK(hipMalloc(&tickets, 8 * 32 * 4));
  const unsigned cfgc[][3] = {{256, 256, 0}, {256, 256, 64}, {256, 256, 20}, {512, 128, 20}, {128, 512, 20}};
  for (int wide = 0; wide < 2; ++wide)
    for (auto& c : cfgc) {
      std::vector<float> ms;
      for (int it = 0; it < 7; ++it) {
        CK(hipMemset(tickets, 0, 8 * 32 * 4));
        CK(hipEventRecord(e0));
        if (wide) hipLaunchKernelGGL(scatter_kernel_chain<uint2>, dim3(512), dim3(1024), 0, 0, reinterpret_cast<const uint2*>(in), reinterpret_cast<uint2*>(out), tiles, c[0], c[1] / 8, (c[2] + 7) / 8, tickets);
        else hipLaunchKernelGGL(scatter_kernel_chain<unsigned>, dim3(512), dim3(1024), 0, 0, reinterpret_cast<const unsigned*>(in), reinterpret_cast<unsigned*>(out), tiles, c[0], c[1] / 4, c[2] / 4, tickets);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); if (it >= 2) ms.push_back(m);
      }
      std::sort(ms.begin(), ms.end());
      printf("chain order, %d-byte lanes: streams %4u run %6u B shift %3u B : %.3f ms  %.2f TB/s (read+write)\n", wide ? 8 : 4, c[0], c[1], wide ? (c[2] + 7) / 8 * 8 : c[2], ms[ms.size() / 2], 2.0 * bytes / ms[ms.size() / 2] / 1e9);
    }
  // the shipped chain-mode sweeps' shape: 512 threads, 8192-key tiles, 3 (uint32) / 2 (uint64) workgroups per CU held by LDS ballast
  {
    const unsigned cus = 256;
    for (int wide = 0; wide < 2; ++wide) {
      const unsigned tile_bytes = wide ? 65536 : 32768, tiles_s = bytes / tile_bytes;
      const unsigned wgs = wide ? 2 : 3, lds = wide ? 8 * 1024 : 11 * 1024;   // + the staging area: 72 / 43 KB per workgroup, as the sweeps
      const unsigned run_b = tile_bytes / 256;       // 256 streams: the run a uniform digit gets in one tile
      for (unsigned shift : {0u, 20u}) {
        std::vector<float> ms;
        for (int it = 0; it < 7; ++it) {
          CK(hipMemset(tickets, 0, 8 * 32 * 4));
          CK(hipEventRecord(e0));
          if (!wide && getenv("SCATTER_X4")) hipLaunchKernelGGL((sweep_shape_kernel_x4<512, 32768, 3>), dim3(cus * wgs), dim3(512), lds, 0, reinterpret_cast<const unsigned*>(in), reinterpret_cast<unsigned*>(out), tiles_s, run_b / 4, shift / 4, tickets);
          else if (wide) hipLaunchKernelGGL((sweep_shape_kernel<uint2, 512, 65536, 2>), dim3(cus * wgs), dim3(512), lds, 0, reinterpret_cast<const uint2*>(in), reinterpret_cast<uint2*>(out), tiles_s, run_b / 8, (shift + 7) / 8, tickets);
          else hipLaunchKernelGGL((sweep_shape_kernel<unsigned, 512, 32768, 3>), dim3(cus * wgs), dim3(512), lds, 0, reinterpret_cast<const unsigned*>(in), reinterpret_cast<unsigned*>(out), tiles_s, run_b / 4, shift / 4, tickets);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float m; CK(hipEventElapsedTime(&m, e0, e1)); if (it >= 2) ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        printf("the sweep's data path alone (512 threads x %u B tiles, %u workgroups per CU, %u KB LDS each; LDS-DMA in, one store per element out), %d-byte lanes: 256 streams run %u B shift %u B : %.3f ms  %.2f TB/s (read+write)\n",
               tile_bytes, wgs, (lds + tile_bytes) / 1024, wide ? 8 : 4, run_b, wide ? (shift + 7) / 8 * 8 : shift, ms[ms.size() / 2], 2.0 * bytes / ms[ms.size() / 2] / 1e9);
      }
    }
  }
  return 0;
}
