// rsort_transfer.hpp -- the host legs of rs_sort_keys / rs_sort_pairs / rs_sort_strings: caller's pageable memory <-> HBM.
//
// Host code only.  A pageable hipMemcpy on this platform reaches 29 GB/s up and 17 GB/s down (the runtime stages through
// one pinned buffer on one thread); registering the caller's pages is refused here.  Each lane below is one host thread
// copying its slice of the array through two pinned 4 MiB bounce buffers on its own stream, so the memcpy of chunk i+1
// overlaps the DMA of chunk i: 29 / 28 GB/s with one lane, 50 / 50 GB/s with two or more (tools/h2d_probe.hip,
// profiles/r01_g_host_transfer_probe.txt).  How many lanes run is the caller's choice (rs_ctx_set_host_threads): the
// front-end passes the reference's own thread rule (radix_sort.hpp:112-123, one thread unless the caller asked for
// multi-threading and has >= 2,000,000 elements).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstring>
#include <thread>
#include <vector>

namespace rst {

constexpr size_t kChunk = size_t(4) << 20;        // bounce buffer size = largest piece
constexpr size_t kMinPiece = size_t(256) << 10;   // smallest piece: short arrays still get a pipeline of several steps
constexpr int kMaxLanes = 4;                       // two already saturate the link; four cover slower host memory
constexpr size_t kStagedMin = size_t(1) << 20;     // below this one pageable copy on the sort's stream is the quicker way

struct Lane {
  hipStream_t stream = nullptr;
  char* pinned[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
};

struct Mover {
  Lane lanes[kMaxLanes];
  int host_threads = 1;

  hipError_t prepare(Lane& L) {
    if (L.stream) return hipSuccess;
    hipError_t e = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
      e = hipHostMalloc(reinterpret_cast<void**>(&L.pinned[b]), kChunk, hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&L.done[b], hipEventDisableTiming);
    }
    if (e != hipSuccess) release(L);
    return e;
  }
  static void release(Lane& L) {
    for (int b = 0; b < 2; ++b) {
      if (L.done[b]) (void)hipEventDestroy(L.done[b]);
      if (L.pinned[b]) (void)hipHostFree(L.pinned[b]);
      L.done[b] = nullptr; L.pinned[b] = nullptr;
    }
    if (L.stream) (void)hipStreamDestroy(L.stream);
    L.stream = nullptr;
  }
  void release_all() { for (auto& L : lanes) release(L); }

  // bytes [lo, hi) of the array, host -> device
  static hipError_t up(Lane& L, const char* host, char* dev, size_t lo, size_t hi, size_t piece) {
    hipError_t e = hipSuccess;
    int b = 0;
    for (size_t at = lo; at < hi && e == hipSuccess; at += piece, b ^= 1) {
      const size_t len = std::min(piece, hi - at);
      e = hipEventSynchronize(L.done[b]);                          // the DMA that last read this bounce buffer
      if (e != hipSuccess) break;
      std::memcpy(L.pinned[b], host + at, len);
      e = hipMemcpyAsync(dev + at, L.pinned[b], len, hipMemcpyHostToDevice, L.stream);
      if (e == hipSuccess) e = hipEventRecord(L.done[b], L.stream);
    }
    hipError_t s = hipStreamSynchronize(L.stream);
    return e != hipSuccess ? e : s;
  }
  // bytes [lo, hi) of the array, device -> host: the DMA of one chunk runs while the previous one is copied out
  static hipError_t down(Lane& L, char* host, const char* dev, size_t lo, size_t hi, size_t piece) {
    hipError_t e = hipSuccess;
    size_t pend_at[2] = {0, 0}, pend_len[2] = {0, 0};
    int b = 0;
    for (size_t at = lo; (at < hi || pend_len[0] || pend_len[1]) && e == hipSuccess; b ^= 1) {
      if (pend_len[b]) {
        e = hipEventSynchronize(L.done[b]);
        if (e != hipSuccess) break;
        std::memcpy(host + pend_at[b], L.pinned[b], pend_len[b]);
        pend_len[b] = 0;
      }
      if (at < hi) {
        const size_t len = std::min(piece, hi - at);
        e = hipMemcpyAsync(L.pinned[b], dev + at, len, hipMemcpyDeviceToHost, L.stream);
        if (e == hipSuccess) e = hipEventRecord(L.done[b], L.stream);
        pend_at[b] = at; pend_len[b] = len;
        at += len;
      }
    }
    hipError_t s = hipStreamSynchronize(L.stream);
    return e != hipSuccess ? e : s;
  }

  // Moves the whole array and returns when it has arrived.  The caller has made sure nothing queued elsewhere still
  // uses the device range (to_device) or still writes it (from device).
  hipError_t move(int device, char* host, char* dev, size_t bytes, bool to_device) {
    const size_t most = (bytes + kMinPiece - 1) / kMinPiece;                     // no lane for less than one small piece
    const int lanes_used = static_cast<int>(std::min<size_t>({static_cast<size_t>(std::max(host_threads, 1)), size_t(kMaxLanes), most}));
    for (int t = 0; t < lanes_used; ++t) {
      hipError_t e = prepare(lanes[t]);
      if (e != hipSuccess) return e;
    }
    // each lane's share in about eight pieces, between 256 KiB and the bounce buffer size, in whole 64 KiB units
    const size_t per = (((bytes + lanes_used - 1) / lanes_used) + 65535) & ~size_t(65535);
    const size_t piece = std::min(kChunk, std::max(kMinPiece, ((per / 8) + 65535) & ~size_t(65535)));
    std::vector<hipError_t> status(lanes_used, hipSuccess);
    auto work = [&](int t) {
      const size_t lo = std::min(bytes, per * t), hi = std::min(bytes, lo + per);
      status[t] = to_device ? up(lanes[t], host, dev, lo, hi, piece) : down(lanes[t], host, dev, lo, hi, piece);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < lanes_used; ++t)
      pool.emplace_back([&, t] { status[t] = hipSetDevice(device); if (status[t] == hipSuccess) work(t); });
    work(0);                                                                    // lane 0 is the calling thread
    for (auto& th : pool) th.join();
    for (hipError_t e : status) if (e != hipSuccess) return e;
    return hipSuccess;
  }
};

}  // namespace rst
