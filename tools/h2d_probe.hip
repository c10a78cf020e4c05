// h2d_probe.hip -- how fast can a pageable host vector reach HBM and come back?  Prices the three candidates for the host
// legs of rs_sort_keys (today: one pageable hipMemcpyAsync each way):
//   1. pageable hipMemcpy (the runtime's own staging)
//   2. hipHostRegister the caller's pages, DMA straight from them, unregister
//   3. T host threads, each copying its slice through two pinned bounce buffers on its own stream
//   hipcc -O2 --offload-arch=gfx950 -pthread tools/h2d_probe.hip -o tools/h2d_probe ; tools/h2d_probe [MiB]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Lane {                       // one host thread's private pipeline
  hipStream_t stream;
  char* pinned[2];
  hipEvent_t done[2];
};

static void staged(char* host, char* dev, size_t bytes, int threads, size_t chunk, bool to_device, std::vector<Lane>& lanes) {
  std::vector<std::thread> pool;
  size_t per = (bytes / threads + 255) & ~size_t(255);
  for (int t = 0; t < threads; t++) {
    pool.emplace_back([=, &lanes] {
      CK(hipSetDevice(0));
      Lane& L = lanes[t];
      size_t lo = std::min(bytes, per * t), hi = std::min(bytes, lo + per);
      int b = 0;
      if (to_device) {
        for (size_t at = lo; at < hi; at += chunk, b ^= 1) {
          size_t len = std::min(chunk, hi - at);
          CK(hipEventSynchronize(L.done[b]));                         // the DMA that last read this bounce buffer
          std::memcpy(L.pinned[b], host + at, len);
          CK(hipMemcpyAsync(dev + at, L.pinned[b], len, hipMemcpyHostToDevice, L.stream));
          CK(hipEventRecord(L.done[b], L.stream));
        }
        CK(hipStreamSynchronize(L.stream));
      } else {
        size_t pend_at[2] = {0, 0}, pend_len[2] = {0, 0};
        for (size_t at = lo; at < hi || pend_len[0] || pend_len[1]; b ^= 1) {
          if (pend_len[b]) {                                           // drain the copy issued two steps ago
            CK(hipEventSynchronize(L.done[b]));
            std::memcpy(host + pend_at[b], L.pinned[b], pend_len[b]);
            pend_len[b] = 0;
          }
          if (at < hi) {
            size_t len = std::min(chunk, hi - at);
            CK(hipMemcpyAsync(L.pinned[b], dev + at, len, hipMemcpyDeviceToHost, L.stream));
            CK(hipEventRecord(L.done[b], L.stream));
            pend_at[b] = at; pend_len[b] = len;
            at += len;
          }
        }
      }
    });
  }
  for (auto& th : pool) th.join();
}

int main(int argc, char** argv) {
  size_t mib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 400;
  size_t bytes = mib << 20;
  CK(hipSetDevice(0));
  char* dev;
  CK(hipMalloc(&dev, bytes));
  char* host = static_cast<char*>(std::malloc(bytes));
  for (size_t i = 0; i < bytes; i += 64) host[i] = static_cast<char>(i >> 6);
  char* back = static_cast<char*>(std::malloc(bytes));
  std::memset(back, 1, bytes);
  auto report = [&](const char* what, double ms) { std::printf("%-58s %8.2f ms  %6.1f GB/s\n", what, ms, bytes / ms / 1e6); std::fflush(stdout); };

  for (int rep = 0; rep < 2; rep++) {
    double t0 = now_ms(); CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice)); report("pageable hipMemcpy H2D", now_ms() - t0);
    t0 = now_ms(); CK(hipMemcpy(back, dev, bytes, hipMemcpyDeviceToHost)); report("pageable hipMemcpy D2H", now_ms() - t0);
  }
  for (int rep = 0; rep < 2; rep++) {
    char* al = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(host) + 4095) & ~uintptr_t(4095));   // whole pages only
    size_t alen = (bytes - (al - host)) & ~size_t(4095);
    double t0 = now_ms();
    hipError_t e = hipHostRegister(al, alen, hipHostRegisterDefault);
    double t1 = now_ms();
    if (e != hipSuccess) { std::printf("hipHostRegister of caller pages: %s (not available here)\n", hipGetErrorString(e)); (void)hipGetLastError(); break; }
    report("hipHostRegister (caller pages)", t1 - t0);
    CK(hipMemcpy(dev, al, alen, hipMemcpyHostToDevice)); double t2 = now_ms(); report("  registered H2D", t2 - t1);
    CK(hipMemcpy(al, dev, alen, hipMemcpyDeviceToHost)); double t3 = now_ms(); report("  registered D2H", t3 - t2);
    CK(hipHostUnregister(al)); report("  hipHostUnregister", now_ms() - t3);
  }
  const size_t chunks[] = {size_t(1) << 20, size_t(4) << 20, size_t(16) << 20};
  const int tcounts[] = {1, 2, 4, 8, 12};
  std::vector<Lane> lanes(12);
  for (auto& L : lanes) {
    CK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++) { CK(hipHostMalloc(reinterpret_cast<void**>(&L.pinned[b]), size_t(16) << 20, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&L.done[b], hipEventDisableTiming)); }
  }
  for (size_t chunk : chunks)
    for (int t : tcounts) {
      char label[96];
      double best_up = 1e30, best_dn = 1e30;
      for (int rep = 0; rep < 3; rep++) {
        double t0 = now_ms(); staged(host, dev, bytes, t, chunk, true, lanes); best_up = std::min(best_up, now_ms() - t0);
        t0 = now_ms(); staged(back, dev, bytes, t, chunk, false, lanes); best_dn = std::min(best_dn, now_ms() - t0);
      }
      std::snprintf(label, sizeof label, "staged H2D  %2d threads, %2zu MiB chunks", t, chunk >> 20); report(label, best_up);
      std::snprintf(label, sizeof label, "staged D2H  %2d threads, %2zu MiB chunks", t, chunk >> 20); report(label, best_dn);
      if (std::memcmp(host, back, bytes) != 0) { std::printf("MISMATCH\n"); return 1; }
    }
  std::printf("round trip verified\n");
  return 0;
}
