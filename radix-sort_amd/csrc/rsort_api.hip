// rsort_api.hip -- host side of librsort_hip.so: the C-ABI declared in include/rsort_hip.h.
//
// Build (see __graft_entry__.build):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Iinclude \
//         radix-sort_amd/csrc/rsort_api.hip -o radix-sort_amd/lib/librsort_hip.so
#include "rsort_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "rsort_transfer.hpp"
#include "rsort_kernels.hpp"
#include "rsort_groups.hpp"
#include "rsort_strings.hpp"

// see Workspace below; named only in rs_ctx_create (hipGetSymbolAddress), everything else uses the address kept in the context
static __device__ __align__(256) unsigned char rs_static_scratch[(size_t(1) << 22) + (size_t(1) << 19)];

namespace {

constexpr int kMaxPasses = 8;
// Tile geometry of the onesweep pass (overridable at build time for tuning runs).  Narrow = keys and
// payload of at most 4 bytes each: 1024 threads x 16 keys = 16384-key tiles; 64 KB staging + 8 KB of 16-bit
// wave histograms leave room for two workgroups = 32 waves per CU, and the kernel fits 64 VGPRs.  Measured on
// uint32, N = 2^30, per pass: 512x16 (3 WG/CU) 3.05 ms, 768x16 (2 WG/CU) 2.59 ms, 1024x16 (2 WG/CU) 2.26 ms --
// fewer, larger tiles mean fewer look-back walks per key.  Wide (8-byte keys or payload): 512 x 16, where the
// 64 KB staging area already limits a CU to two workgroups.
#ifndef RS_SWEEP_BLOCK
#define RS_SWEEP_BLOCK 1024
#endif
#ifndef RS_SWEEP_BLOCK_WIDE
#define RS_SWEEP_BLOCK_WIDE 512
#endif
#ifndef RS_SWEEP_ITEMS
#define RS_SWEEP_ITEMS 16
#endif
#ifndef RS_LOOKBACK
#define RS_LOOKBACK 8
#endif
#ifndef RS_LOOKBACK_CHAIN   // look-back window of the chain-mode kernels (short walks: an eighth of the tiles per chain)
#define RS_LOOKBACK_CHAIN 4
#endif
// Chain-mode geometry for narrow keys: 512 threads x 16 keys = 8192-key tiles, three workgroups per CU.  With the look-back
// inside one XCD's L2 the walk no longer argues for large tiles, and 24 waves per CU leave 80 registers per lane: the
// counting kernels (next-pass digit counts in the write-out) then need no scratch, whose loads and stores were a fifth of
// the 1024 x 16 kernel's vector-memory instructions (uint32 N = 2^30, pass with counting: 2.28 ms at 1024 x 16, 2.11 here).
#ifndef RS_CHAIN_BLOCK
#define RS_CHAIN_BLOCK 512
#endif
#ifndef RS_CHAIN_MIN_BLOCKS
#define RS_CHAIN_MIN_BLOCKS 3
#endif
#ifndef RS_MIN_BLOCKS   // workgroups per CU the register allocator must leave room for
#define RS_MIN_BLOCKS 2
#endif
#ifndef RS_MIN_BLOCKS_WIDE
#define RS_MIN_BLOCKS_WIDE 1
#endif
#ifndef RS_SWEEP_ITEMS_WIDE
#define RS_SWEEP_ITEMS_WIDE 16
#endif
#ifndef RS_CHAIN_ITEMS     // keys per thread of the narrow chain-mode kernels
#define RS_CHAIN_ITEMS 16
#endif
constexpr int sweep_items(int key_bytes, int value_bytes, bool chain = false) {
  return (key_bytes <= 4 && value_bytes <= 4) ? (chain ? RS_CHAIN_ITEMS : RS_SWEEP_ITEMS) : RS_SWEEP_ITEMS_WIDE;
}
constexpr bool narrow_geometry(int key_bytes, int value_bytes) { return key_bytes <= 4 && value_bytes <= 4; }
constexpr int sweep_block(int key_bytes, int value_bytes, bool chain = false) {
  return narrow_geometry(key_bytes, value_bytes) ? (chain ? RS_CHAIN_BLOCK : RS_SWEEP_BLOCK) : RS_SWEEP_BLOCK_WIDE;
}
constexpr int kHistBlock = 1024;
constexpr size_t kSmallDescLimit = size_t(1) << 30;   // 30 value bits in a 4-byte descriptor
// The per-XCD tile order (rsort_kernels.hpp) lets a tile wait for a tile whose workgroup has not been launched yet.  One
// sort alone always gets that workgroup launched (each XCD has free slots whenever it is behind), but several LARGE sorts
// running on one GPU at the same time could fill each other's slots with waiting workgroups.  So the order is used only
// where it pays -- sorts of at least this many tiles (16 M four-byte keys), which saturate the GPU on their own -- and
// rs_ctx_set_tile_order(ctx, 0) turns it off for callers that do run such sorts concurrently on one device.
constexpr size_t kPerXcdMinTiles = 1024;
constexpr size_t kHybridMinKeys = 64000000;    // hybrid form of 4-byte key sorts from here on (four-wave group kernel; 2^26 keys 655 against 674 us, 10^8 875 against 970; 2^25 429 against 385 -- below, the fixed cost of 65536 workgroups outweighs the two sweeps saved)
constexpr size_t kHybridMinKeysWide = 120000000;   // the same for 8-byte keys (two LDS passes + fix-up per group; 2^27 keys: 3.21 against 3.90 ms, 2^28: 5.0 against 7.6; 2^26: 2.36 against 2.06)
// Narrow hybrid form (sort_device): 8-byte keys from the end of the mid-size regime up to kHybridMinKeysWide -- 5 / 16.7 / 67 M keys
// 219 / 404 / 1358 us against 342 / 596 / 2075 us with eight passes.  4-byte keys: never by itself (whole keys in LDS, three
// LDS passes: 16.7 M keys 308 against 241 us with four passes); rs_ctx_set_hybrid(2) forces it for tests.
constexpr size_t kNarrowMinKeys = kHybridMinKeys, kNarrowMinKeysWide = 0;
constexpr size_t kHybridMinPairsWide = 14000000;    // and with a 4-byte payload (1 / 2 / 3 / 4.5 * 10^7 pairs: 0.53 / 0.77 / 0.95 / 1.24 against 0.51 / 1.03 / 1.55 / 2.07 ms; 2^26: 1.64 against 2.88, 2^28: 5.2 against 11.5)
constexpr size_t kAlwaysSweepTiles = 256;   // up to here (4 M four-byte keys) neither pass skipping nor the order pre-check is worth a host synchronisation

// Hybrid sort of 4-byte keys (rsort_groups.hpp): two sweeps on the top digits, then every group of equal top 16 bits is
// finished in LDS by one workgroup.  Geometry of that kernel: 512 threads, up to 10 vectors of four keys each per thread.
constexpr int kGroupBlock = 512;
constexpr int kGroupMaxVec = 10;
constexpr size_t kGroupCap = size_t(kGroupBlock) * kGroupMaxVec * 4 - 3;   // keys of the largest group the LDS window holds at any alignment
// 8-byte keys: whole keys in LDS (rsk::group_sort_wide_kernel), 1024 threads x 17 keys = 17408 keys = 136 KB, one workgroup per CU
constexpr int kWideGroupBlock = 1024;
constexpr int kWideGroupItems = 17;
constexpr size_t kWideGroupCap = size_t(kWideGroupBlock) * kWideGroupItems;
// mid-size sorts of 4-byte keys: groups of equal top byte, whole keys in LDS (rsk::group_sort_mid_kernel), 1024 x 20 keys = 80 KB
constexpr int kMidGroupBlock = 1024;
constexpr int kMidGroupItems = 20;
constexpr int kSmallGroupBlock = 256;          // 4-byte keys, groups of at most 256 x 10 x 4 - 3 = 10237 keys (the kernel wants one thread per digit: not fewer)
constexpr size_t kSmallGroupMeanMax = 6500;    // mean group size below which the four-wave workgroups sort the groups (n below 4.26e8; measured 1.3e8 / 2.7e8 / 4e8 keys: 1030 / 1747 / 2525 against 1164 / 1913 / 2681 us)
constexpr int kWidePairGroupItems = 12;   // 8-byte keys with a 4-byte payload: 1024 x 12 = 12288 pairs = 144 KB
constexpr size_t kWidePairGroupCap = size_t(1024) * kWidePairGroupItems;
// 4-byte keys with a 4-byte payload: the same kernel in smaller workgroups, so that several share a CU and one's load and
// store hide behind the others' passes -- 512 x 17 = 8704 pairs = 78 KB with the counters, two to a CU (2^29 pairs: group sort
// 2.07 ms, whole sort 7.66 -> 6.16 ms; one workgroup of 1024 x 12 per CU: 3.4 ms, no gain over the two sweeps it replaces);
// when the largest group fits 256 x 17 = 4352 pairs, four to a CU (1 / 1.34 / 2 * 10^8 pairs: 1.50 / 1.80 / 2.49 ms against
// 1.68 / 2.12 / 2.74 with 512 threads and 1.61 / 1.96 / 2.97 for the four-pass loop; 6.7 * 10^7: 1.11 against 1.02, hence the threshold)
constexpr int kPair32Block = 512, kPair32SmallBlock = 256, kPair32Items = 17;
constexpr size_t kPair32Cap = size_t(kPair32Block) * kPair32Items, kPair32SmallCap = size_t(kPair32SmallBlock) * kPair32Items;
constexpr size_t kHybridMinPairs = 90000000;
// 8-byte groups well inside smaller windows run in smaller workgroups, several to a CU (one's load and store behind the
// others' passes): 256 threads x 17 = 4352 keys, four per CU; 512 x 17 = 8704, two per CU.  The narrow form aims its groups at
// the smallest window.  (2^26 / 10^8 / 2 * 10^8 / 2^28 / 4 * 10^8 keys: 1.34 / 2.88 / 3.98 / 5.21 / 6.91 -> 1.21 / 1.65 / 2.79 / 4.10 / 5.68 ms)
constexpr double kNarrowGroupTarget = 3700;
constexpr unsigned long long kClusteredRepeats = 3, kClusteredRepeatsWide = 1500;   // repeats among 4096 sampled keys from which on a keys-only sort is left to the plain loop (sort_device)
constexpr unsigned kGroupCount = 65536;
constexpr size_t kGroupsBytes = (size_t(kGroupCount) + 1) * 3 * sizeof(unsigned);   // start table + list of oversize groups + list of the groups beyond the small workgroups' window
constexpr size_t kGroupCheckMinKeys = 500000000;   // sorts from here on look at their groups' real sizes before the LDS finish (one read-back: 40-60 us with the launches it holds up)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

struct rs_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  DevBuf alt_keys, alt_vals, desc, ctrl, strbuf, groups, distinct;
  // ctrl layout: kCtrl* below
  bool rank_by_lds_atomic = false;   // set by the probe in rs_ctx_create
  bool per_xcd_order = false;        // set by the XCC-id probe in rs_ctx_create (and cleared by rs_ctx_set_tile_order)
  bool per_xcd_probe_ok = false;     // what the probe found
  bool desc_local = false;           // chain-mode descriptors may be published with workgroup-scope stores (chain_probe_kernel in rs_ctx_create)
  int cus = 256;                     // compute units of the device (persistent grids: 2 workgroups each)
  char* static_scratch = nullptr;    // this device's copy of rs_static_scratch (rs_sort_keys_device_noalloc)
  bool static_used = false;          // this context has queued a sort on it (or captured one into a graph that may replay at any time): its error word
                                     // is read at every synchronisation from then on
  int hist_copy = 0;                 // the copy of the digit counts in `ctrl` that is zero now (kCtrlHist1)
  unsigned* pinned_flags = nullptr;  // two pinned host words the error words are fetched into (a pageable 4-byte copy costs 20 us)
  bool host_call = false;            // sort_device runs inside a host-buffer entry point (which waits for the device anyway)
  bool profiling = false;
  bool precheck = true;    // order pre-check before sorting (rs_ctx_set_precheck)
  bool last_unchanged = false;   // the last sort_device found its input already in order and moved nothing
  int hybrid_mode = -1;          // hybrid forms of keys-only sorts: -1 chosen per sort, 0 never, 1 the 16-bit-group form whenever the shape allows, 2 the narrow form whenever it does (rs_ctx_set_hybrid; RSORT_HYBRID at creation)
  unsigned group_max_vec = 0;    // test hook: groups of more vectors than this go to the big-group kernel (0 = what the LDS window holds)
  int last_hybrid = 0;          // the last sort_device took a hybrid form (1: groups of equal top 16 bits, 2: the narrow form)
  int last_string_levels = 0;    // MSD levels the last string sort took
  bool last_distinct = false;    // the last sort_device sorted by counting the few distinct keys (rsd::, rsort_groups.hpp)
  bool distinct_path = true;     // RSORT_DISTINCT=0 at creation switches that path off (tests, A/B timing)
  bool last_all_equal = false;   // the last sort_device read its digit counts and found every key equal on the sorted bits (nothing launched)
  hipEvent_t ev[kMaxPasses + 4] = {};
  hipEvent_t small_up = nullptr;     // after the last small host-to-device copy through lane 0's bounce buffer (to_device)
  bool small_up_pending = false;
  int ev_count = 0;
  rst::Mover mover;        // host <-> HBM legs of the host-buffer entry points (rsort_transfer.hpp)
  struct Fetched { void* host; size_t bytes; } fetched[2] = {{nullptr, 0}, {nullptr, 0}};   // small results waiting in pinned memory
  std::string err;
};

namespace {

// Control block: [0, 64) error word (+0) and order-check flags (+32), never part of a per-sort memset; then what every
// sort clears with ONE memset: digit counts u64[8][256], their exclusive prefixes u64[8][256], per pass one ticket word per
// XCD a cache line apart, and -- directly behind, so that the same memset reaches them -- the look-back descriptors of
// sorts of at most kAlwaysSweepTiles tiles (256 KB); larger sorts use the separate descriptor buffer.
constexpr size_t kCtrlError = 0;
constexpr size_t kCtrlHist = 64;
// A second copy of the digit counts, for mid-size sorts on the context's own block: they launch no memset.  The histogram
// kernel adds into the copy rs_ctx::hist_copy names, which is zero when a sort starts, and clears what the memset would
// have: the OTHER copy (for the next sort), the tickets and the descriptor rows of this one.  Every path that does launch
// the memset clears both copies, uses copy 0 and leaves hist_copy = 1.
constexpr size_t kCtrlHist1 = kCtrlHist + kMaxPasses * 256 * 8;
constexpr size_t kCtrlBase = kCtrlHist1 + kMaxPasses * 256 * 8;
constexpr size_t kCtrlTicket = kCtrlBase + kMaxPasses * 256 * 8;
constexpr size_t kCtrlMidDesc = kCtrlTicket + kMaxPasses * rsk::kTicketWords * sizeof(unsigned);
constexpr size_t kMidDescTiles = 256;
constexpr size_t kCtrlBytes = kCtrlMidDesc + kMidDescTiles * 256 * 4;
static_assert(kCtrlHist % 16 == 0 && kCtrlHist1 % 16 == 0 && kCtrlTicket % 16 == 0 && kCtrlMidDesc % 16 == 0, "cleared 16 bytes per store");
// chain mode only (rsort_kernels.hpp, "XCD chains"), zeroed only by sorts that use it:
constexpr size_t kCtrlChainHist = (kCtrlBytes + 255) & ~size_t(255);                       // u64 [8 chains][8 passes][256] from the histogram kernel
constexpr size_t kCtrlNextCnt = kCtrlChainHist + rsk::kXcds * rsk::kChainHistStride * 8;   // u64 [8 passes][8 chains][256] from the sweeps
constexpr size_t kCtrlBytesChain = kCtrlNextCnt + kMaxPasses * rsk::kXcds * 256 * 8;

// Workspace of the entry point that allocates nothing (rs_sort_keys_device_noalloc): control block and look-back
// descriptors in a static device array instead of hipMalloc'ed memory.  Every GPU the library runs on holds its own copy
// until the process ends; all calls on one GPU share it.
constexpr size_t kNoAllocMaxKeys = size_t(1) << 25;
constexpr size_t kNoAllocCtrlBytes = (kCtrlBytes + 255) & ~size_t(255);
constexpr size_t kNoAllocDescBytes = (kNoAllocMaxKeys / 8192) * 256 * 4;     // smallest tile of any geometry, 4-byte descriptors
constexpr size_t kNoAllocBytes = kNoAllocCtrlBytes + kNoAllocDescBytes;       // 4.3 MiB (the array is declared with 4.5 MiB)
struct Workspace {        // non-null: sort_device uses these and neither allocates nor waits for the device
  char* ctrl;
  void* desc;
};

struct HostCall {         // marks the context for the duration of a host-buffer entry point
  rs_ctx* c;
  explicit HostCall(rs_ctx* ctx) : c(ctx) { c->host_call = true; }
  ~HostCall() { c->host_call = false; }
};

int fail(rs_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

#define RS_HIP(c, call)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail((c), RS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));      \
  } while (0)

int ensure(rs_ctx* c, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes) return RS_OK;
  if (b.p) {
    RS_HIP(c, hipStreamSynchronize(c->stream));
    RS_HIP(c, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) {
    b.p = nullptr;
    return fail(c, RS_ERR_NOMEM, "hipMalloc(" + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
  }
  b.bytes = bytes;
  if (&b == &c->ctrl) RS_HIP(c, hipMemsetAsync(b.p, 0, bytes, c->stream));   // the error word starts clear and is only cleared again once read
  return RS_OK;
}

// Zero `bytes` at p on the stream.  by_kernel: with a kernel of this library instead of hipMemsetAsync -- for sorts that are
// being captured into a HIP graph.  Measured (ROCm 7.0, tools/dbg_graph2.py): a captured sort with memset nodes replayed
// correctly as long as only copies ran between the replays; once other kernels had run on the device in between, the
// memset nodes no longer cleared the tickets and every later replay moved nothing.  Kernel nodes replay as captured.
int clear_async(rs_ctx* c, void* p, size_t bytes, hipStream_t s, bool by_kernel) {
  if (bytes == 0) return RS_OK;
  if (by_kernel && (bytes & 15u) == 0 && (reinterpret_cast<uintptr_t>(p) & 15u) == 0 && (bytes >> 4) <= 0xFFFFFFFFull) {
    const unsigned vecs = static_cast<unsigned>(bytes >> 4);
    const unsigned grid = std::min<unsigned>((vecs + 255u) / 256u, 2048u);
    hipLaunchKernelGGL(rsk::zero_kernel, dim3(grid), dim3(256), 0, s, static_cast<uint4*>(p), vecs);
    return RS_OK;
  }
  RS_HIP(c, hipMemsetAsync(p, 0, bytes, s));
  return RS_OK;
}

size_t tile_keys(int key_bytes, int value_bytes, bool chain = false) {
  return static_cast<size_t>(sweep_block(key_bytes, value_bytes, chain)) * sweep_items(key_bytes, value_bytes, chain);
}
size_t tiles_for(size_t n, int key_bytes, int value_bytes, bool chain = false) {
  const size_t t = tile_keys(key_bytes, value_bytes, chain);
  return (n + t - 1) / t;
}
// sized for the smaller (wide) tile so that one reservation serves every geometry
size_t desc_bytes_for(size_t n) { return tiles_for(n, 8, 8) * 256 * 8; }   // sized for 8-byte descriptors

int check_args(rs_ctx* c, const void* keys, size_t n, int key_bytes, int value_bytes, int transform,
               int begin_bit, int end_bit) {
  if (!c) return RS_ERR_ARG;
  if (key_bytes != 1 && key_bytes != 2 && key_bytes != 4 && key_bytes != 8)
    return fail(c, RS_ERR_ARG, "key_bytes must be 1, 2, 4 or 8");
  if (value_bytes != 0 && value_bytes != 4 && value_bytes != 8)
    return fail(c, RS_ERR_ARG, "value_bytes must be 4 or 8");
  if (transform < RS_XF_NONE || transform > RS_XF_IEEE) return fail(c, RS_ERR_ARG, "unknown transform");
  if (transform == RS_XF_IEEE && key_bytes < 4)
    return fail(c, RS_ERR_ARG, "RS_XF_IEEE needs 4- or 8-byte keys (radix_sort.hpp:94-98 has no other float width)");
  if (begin_bit < 0 || end_bit > key_bytes * 8 || begin_bit >= end_bit)
    return fail(c, RS_ERR_ARG, "need 0 <= begin_bit < end_bit <= 8*key_bytes");
  if (n > 0 && !keys) return fail(c, RS_ERR_ARG, "null key pointer");
  if (n >= (size_t(1) << 62)) return fail(c, RS_ERR_ARG, "n too large");
  return RS_OK;
}

template <typename K, typename V, typename D>
void launch_sweep(const rsk::SweepArgs& a, size_t tiles, bool atomic_rank, bool chain, unsigned chain_grid, hipStream_t s) {
  constexpr int kVal = std::is_same<V, rsk::NoVal>::value ? 0 : static_cast<int>(sizeof(V));
  constexpr int kBlock = sweep_block(static_cast<int>(sizeof(K)), kVal);
  constexpr int kItems = sweep_items(static_cast<int>(sizeof(K)), kVal);
  constexpr int kMinBlocks = narrow_geometry(static_cast<int>(sizeof(K)), kVal) ? RS_MIN_BLOCKS : RS_MIN_BLOCKS_WIDE;
  if constexpr (kVal == 0 ? sizeof(K) >= 4 : (sizeof(K) == 8 && kVal == 4)) {   // the shapes sort_device selects chain mode for
    if (chain) {   // persistent workgroups, one chain of tiles per XCD (only with the returning-add ranking)
      constexpr int kBlockC = sweep_block(static_cast<int>(sizeof(K)), kVal, true);
      constexpr int kItemsC = sweep_items(static_cast<int>(sizeof(K)), kVal, true);
      // (wide: two workgroups per CU by LDS, so the register allocator is held to the 128 registers that go with it;
      // the pair kernel with counting took 144 when left alone, i.e. one workgroup per CU)
#ifndef RS_CHAIN_MIN_BLOCKS_WIDE
#define RS_CHAIN_MIN_BLOCKS_WIDE 2
#endif
      constexpr int kMinBlocksC = narrow_geometry(static_cast<int>(sizeof(K)), kVal) ? RS_CHAIN_MIN_BLOCKS : RS_CHAIN_MIN_BLOCKS_WIDE;
      auto launch = [&](auto kernel) {
        static const int per_cu = [&] {      // resident workgroups per CU (registers, LDS, wave slots)
          int nb = 0;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlockC, 0) != hipSuccess || nb < 1) nb = 1;
          return nb;
        }();
        hipLaunchKernelGGL(kernel, dim3(chain_grid * per_cu), dim3(kBlockC), 0, s, a);
      };
      // the last launched pass has no next digit to count for
      if (a.next_cnt) launch(rsk::onesweep_kernel<K, V, D, kBlockC, kItemsC, RS_LOOKBACK_CHAIN, kMinBlocksC, true, true, true>);
      else launch(rsk::onesweep_kernel<K, V, D, kBlockC, kItemsC, RS_LOOKBACK_CHAIN, kMinBlocksC, true, true, false>);
      return;
    }
  }
  if (atomic_rank)
    hipLaunchKernelGGL((rsk::onesweep_kernel<K, V, D, kBlock, kItems, RS_LOOKBACK, kMinBlocks, true>),
                       dim3(static_cast<unsigned>(tiles)), dim3(kBlock), 0, s, a);
  else
    hipLaunchKernelGGL((rsk::onesweep_kernel<K, V, D, kBlock, kItems, RS_LOOKBACK, kMinBlocks, false>),
                       dim3(static_cast<unsigned>(tiles)), dim3(kBlock), 0, s, a);
}

struct SweepLaunch {
  size_t tiles; int key_bytes; int value_bytes; bool wide; bool atomic_rank; bool chain; unsigned chain_grid /* compute units */; hipStream_t s;
};

template <typename K, typename V>
void launch_sweep_d(const rsk::SweepArgs& a, const SweepLaunch& l) {
  if (l.wide) launch_sweep<K, V, uint64_t>(a, l.tiles, l.atomic_rank, l.chain, l.chain_grid, l.s);
  else launch_sweep<K, V, uint32_t>(a, l.tiles, l.atomic_rank, l.chain, l.chain_grid, l.s);
}

template <typename K>
void launch_sweep_v(const rsk::SweepArgs& a, const SweepLaunch& l) {
  if (l.value_bytes == 0) launch_sweep_d<K, rsk::NoVal>(a, l);
  else if (l.value_bytes == 4) launch_sweep_d<K, uint32_t>(a, l);
  else launch_sweep_d<K, uint64_t>(a, l);
}

void launch_sweep_k(const rsk::SweepArgs& a, const SweepLaunch& l) {
  switch (l.key_bytes) {
    case 1: launch_sweep_v<uint8_t>(a, l); break;
    case 2: launch_sweep_v<uint16_t>(a, l); break;
    case 4: launch_sweep_v<uint32_t>(a, l); break;
    default: launch_sweep_v<uint64_t>(a, l); break;
  }
}

// Persistent grid for the histogram: its LDS table is key_bytes x 16 KB (rsk::kHistCols), so one
// workgroup per CU for 8-byte keys, two otherwise; each workgroup strides over the keys.
constexpr unsigned hist_wgs_per_cu(int key_bytes) {      // LDS table = key_bytes x 256 x kHistCols x 4 bytes
  return static_cast<size_t>(key_bytes) * 256 * rsk::kHistCols * 4 <= 80 * 1024 ? 2u : 1u;
}
unsigned hist_grid(size_t n, int key_bytes) {
  const size_t per_block = static_cast<size_t>(kHistBlock) * 16 / key_bytes * 4;
  const size_t resident = 256 * hist_wgs_per_cu(key_bytes);
  return static_cast<unsigned>(std::max<size_t>(std::min<size_t>((n + per_block - 1) / per_block, resident), 1));
}

// Single-workgroup sort of at most one tile (rsk::small_sort_kernel); same geometry table as the sweep.
template <typename K, typename V>
void launch_small_kv(const rsk::SmallArgs& a, hipStream_t s) {
  constexpr int kVal = std::is_same<V, rsk::NoVal>::value ? 0 : static_cast<int>(sizeof(V));
  constexpr int kBlock = sweep_block(static_cast<int>(sizeof(K)), kVal);
  constexpr int kItems = sweep_items(static_cast<int>(sizeof(K)), kVal);
  hipLaunchKernelGGL((rsk::small_sort_kernel<K, V, kBlock, kItems>), dim3(1), dim3(kBlock), 0, s, a);
}
template <typename K>
void launch_small_k(const rsk::SmallArgs& a, int value_bytes, hipStream_t s) {
  if (value_bytes == 0) launch_small_kv<K, rsk::NoVal>(a, s);
  else if (value_bytes == 4) launch_small_kv<K, uint32_t>(a, s);
  else launch_small_kv<K, uint64_t>(a, s);
}
void launch_small(const rsk::SmallArgs& a, int key_bytes, int value_bytes, hipStream_t s) {
  switch (key_bytes) {
    case 1: launch_small_k<uint8_t>(a, value_bytes, s); break;
    case 2: launch_small_k<uint16_t>(a, value_bytes, s); break;
    case 4: launch_small_k<uint32_t>(a, value_bytes, s); break;
    default: launch_small_k<uint64_t>(a, value_bytes, s); break;
  }
}

template <typename K>
void launch_hist_k(const rsk::HistArgs& a, unsigned grid, hipStream_t s) {
  const bool whole = a.begin_bit == 0 && a.end_bit == static_cast<int>(sizeof(K)) * 8;
  if (whole) hipLaunchKernelGGL((rsk::digit_histogram_kernel<K, kHistBlock, true>), dim3(grid), dim3(kHistBlock), 0, s, a);
  else hipLaunchKernelGGL((rsk::digit_histogram_kernel<K, kHistBlock, false>), dim3(grid), dim3(kHistBlock), 0, s, a);
}

void launch_hist(const rsk::HistArgs& a, int key_bytes, unsigned grid, hipStream_t s) {
  switch (key_bytes) {
    case 1: launch_hist_k<uint8_t>(a, grid, s); break;
    case 2: launch_hist_k<uint16_t>(a, grid, s); break;
    case 4: launch_hist_k<uint32_t>(a, grid, s); break;
    default: launch_hist_k<uint64_t>(a, grid, s); break;
  }
}

template <typename K>
void launch_order_k(const rsk::OrderArgs& a, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((rsk::order_check_kernel<K, 256>), dim3(grid), dim3(256), 0, s, a);
}
template <typename K>
void launch_reverse_k(void* keys, size_t n, hipStream_t s) {
  unsigned grid = static_cast<unsigned>(std::min<size_t>((n / 2 + 255) / 256, 256 * 16));
  hipLaunchKernelGGL((rsk::reverse_kernel<K>), dim3(std::max(grid, 1u)), dim3(256), 0, s, static_cast<K*>(keys), static_cast<uint64_t>(n));
}

// Sort by counting the distinct keys (see the call in sort_device).  *done = 1 when the array has been sorted that way.
int sort_few_distinct(rs_ctx* c, void* keys, size_t n, int key_bytes, uint64_t fmask, uint64_t smask, int* done, unsigned long long* span) {
  *done = 0;
  hipStream_t s = c->stream;
  constexpr size_t G = rsd::kGlobalSlots, D = rsd::kDistinctMax;
  constexpr size_t kOffCnt = G * 8, kOffOnes = kOffCnt + G * 8, kOffUsed = kOffOnes + 8, kOffSorted = kOffUsed + 8,
                   kOffStarts = kOffSorted + (D + 1) * 8, kOffSpan = kOffStarts + (D + 2) * 8, kBytes = kOffSpan + 24;
  int rc;
  if ((rc = ensure(c, c->distinct, kBytes)) != RS_OK) return rc;
  char* d = static_cast<char*>(c->distinct.p);
  rsd::CollectArgs a{};
  a.keys = keys; a.gkeys = d; a.gcnt = reinterpret_cast<unsigned long long*>(d + kOffCnt);
  a.ones = reinterpret_cast<unsigned long long*>(d + kOffOnes); a.used = reinterpret_cast<unsigned*>(d + kOffUsed);
  auto launch = [&](unsigned grid) {
    if (key_bytes == 4) hipLaunchKernelGGL((rsd::distinct_collect_kernel<uint32_t, 1024>), dim3(grid), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL((rsd::distinct_collect_kernel<uint64_t, 1024>), dim3(grid), dim3(1024), 0, s, a);
  };
  // 1. the sample: the first 2^18 keys, one workgroup
  unsigned used[2] = {0, 0};
  RS_HIP(c, hipMemsetAsync(a.used, 0, 8, s));
  a.n = std::min<size_t>(n, size_t(1) << 18); a.probe = 1;
  launch(1);
  // (riding on the same wait: which key bits differ at all, over a sample spread across the array -- sort_device's narrow form asks)
  auto* d_span = reinterpret_cast<unsigned long long*>(d + kOffSpan);
  RS_HIP(c, hipMemsetAsync(d_span, 0xFF, 8, s));
  RS_HIP(c, hipMemsetAsync(d_span + 1, 0, 16, s));
  if (key_bytes == 4) hipLaunchKernelGGL(rsd::bit_span_kernel<uint32_t>, dim3(1), dim3(1024), 0, s, static_cast<const uint32_t*>(keys), static_cast<uint64_t>(n), d_span);
  else hipLaunchKernelGGL(rsd::bit_span_kernel<uint64_t>, dim3(1), dim3(1024), 0, s, static_cast<const uint64_t*>(keys), static_cast<uint64_t>(n), d_span);
  RS_HIP(c, hipMemcpyAsync(used, a.used, 8, hipMemcpyDeviceToHost, s));
  unsigned long long got[3] = {0, 0, 0};
  RS_HIP(c, hipMemcpyAsync(got, d_span, 24, hipMemcpyDeviceToHost, s));
  RS_HIP(c, hipStreamSynchronize(s));
  span[0] = got[0]; span[1] = got[1]; span[3] = got[2];
  span[2] = 1;                                              // known
  if (used[0] > D) return RS_OK;
  // 2. the whole array
  int ev = 0;
  if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
  RS_HIP(c, hipMemsetAsync(d, 0xFF, G * 8, s));
  RS_HIP(c, hipMemsetAsync(d + kOffCnt, 0, kOffSorted - kOffCnt, s));
  a.n = n; a.probe = 0;
  launch(static_cast<unsigned>(c->cus) * 2u);
  if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
  std::vector<unsigned char> h(kOffSorted);
  RS_HIP(c, hipMemcpyAsync(h.data(), d, kOffSorted, hipMemcpyDeviceToHost, s));
  RS_HIP(c, hipStreamSynchronize(s));
  std::memcpy(used, h.data() + kOffUsed, 8);
  if (used[1] || used[0] > D) return RS_OK;                 // more distinct keys than the path serves: the usual sort
  struct Entry { uint64_t order, raw, count; };
  std::vector<Entry> e;
  const uint64_t all = key_bytes == 8 ? ~0ull : 0xFFFFFFFFull, sign = 1ull << (key_bytes * 8 - 1);
  auto forward = [&](uint64_t raw) {                        // xf_forward (rsort_kernels.hpp): the order the keys sort in
    const uint64_t fill = (raw & sign) ? all : 0;
    return (raw ^ ((fill & fmask) | smask)) & all;
  };
  uint64_t total = 0;
  for (size_t i = 0; i < G; ++i) {
    uint64_t raw = 0, cnt = 0;
    std::memcpy(&raw, h.data() + i * key_bytes, key_bytes);
    std::memcpy(&cnt, h.data() + kOffCnt + i * 8, 8);
    if (raw == all) continue;
    e.push_back({forward(raw), raw, cnt});
    total += cnt;
  }
  uint64_t ones = 0;
  std::memcpy(&ones, h.data() + kOffOnes, 8);
  if (ones) { e.push_back({forward(all), all, ones}); total += ones; }
  if (total != n || e.empty() || e.size() > D) return RS_OK;    // (keys that changed under the count: leave it to the usual sort and its checks)
  std::sort(e.begin(), e.end(), [](const Entry& x, const Entry& y) { return x.order < y.order; });
  std::vector<unsigned char> up((D + 1) * 8 + (D + 2) * 8, 0);
  uint64_t at = 0;
  for (size_t i = 0; i < e.size(); ++i) {
    std::memcpy(up.data() + i * key_bytes, &e[i].raw, key_bytes);
    std::memcpy(up.data() + (D + 1) * 8 + i * 8, &at, 8);
    at += e[i].count;
  }
  std::memcpy(up.data() + (D + 1) * 8 + e.size() * 8, &at, 8);
  RS_HIP(c, hipMemcpyAsync(d + kOffSorted, up.data(), up.size(), hipMemcpyHostToDevice, s));
  RS_HIP(c, hipStreamSynchronize(s));                       // (the staging vector goes out of scope)
  rsd::FillArgs f{};
  f.out = keys; f.n = n; f.sorted_keys = d + kOffSorted; f.starts = reinterpret_cast<const unsigned long long*>(d + kOffStarts);
  f.distinct = static_cast<unsigned>(e.size());
  const unsigned grid = static_cast<unsigned>(std::min<size_t>((n * key_bytes / 16 + 255) / 256, size_t(c->cus) * 16));
  if (key_bytes == 4) hipLaunchKernelGGL((rsd::fill_runs_kernel<uint32_t, 256>), dim3(std::max(grid, 1u)), dim3(256), 0, s, f);
  else hipLaunchKernelGGL((rsd::fill_runs_kernel<uint64_t, 256>), dim3(std::max(grid, 1u)), dim3(256), 0, s, f);
  RS_HIP(c, hipGetLastError());
  if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
  c->ev_count = c->profiling ? ev : 0;
  c->last_distinct = true;
  *done = 1;
  return RS_OK;
}

size_t hybrid_min_keys_wide() {     // RSORT_HYBRID_MIN_WIDE (A/B timing): 8-byte keys-only sorts below this take the narrow form, from it on groups of equal top 16 bits
  static const size_t v = [] { const char* e = std::getenv("RSORT_HYBRID_MIN_WIDE"); return e && e[0] ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : kHybridMinKeysWide; }();
  return v;
}
double narrow_group_target() {     // RSORT_NARROW_GROUP (A/B timing): keys per group the narrow form aims at
  static const double v = [] { const char* e = std::getenv("RSORT_NARROW_GROUP"); return e && e[0] ? std::strtod(e, nullptr) : kNarrowGroupTarget; }();
  return v;
}
int wide_small_blocks() {     // RSORT_WIDE_SMALL=0 (A/B timing): 8-byte groups always in workgroups of 1024 threads
  static const int v = [] { const char* e = std::getenv("RSORT_WIDE_SMALL"); return e && e[0] ? std::atoi(e) : 1; }();
  return v;
}
size_t group_check_min_keys() {     // RSORT_GROUP_CHECK_MIN (A/B timing)
  static const size_t v = [] { const char* e = std::getenv("RSORT_GROUP_CHECK_MIN"); return e && e[0] ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : kGroupCheckMinKeys; }();
  return v;
}
size_t small_group_mean_max() {     // RSORT_SMALL_GROUP_MEAN (A/B timing): mean group size below which the four-wave workgroups sort the groups
  static const size_t v = [] { const char* e = std::getenv("RSORT_SMALL_GROUP_MEAN"); return e ? static_cast<size_t>(std::atol(e)) : kSmallGroupMeanMax; }();
  return v;
}

// The device sort proper.  keys/vals are sorted in place; alt_* are same-size scratch.
int sort_device(rs_ctx* c, void* keys, void* vals, void* alt_keys, void* alt_vals, size_t n, int key_bytes,
                int value_bytes, int transform, int begin_bit, int end_bit, const Workspace* ws = nullptr, bool forward_done = false) {
  c->last_unchanged = false;
  c->last_hybrid = false;
  c->last_all_equal = false;
  c->last_distinct = false;
  if (n < 2) { c->ev_count = 0; return RS_OK; }
  hipStream_t s = c->stream;
  int passes = (end_bit - begin_bit + rsk::kRadixBits - 1) / rsk::kRadixBits;
  // RSORT_FORCE_WIDE=1 (tests, A/B timing): 8-byte descriptors and offsets regardless of n
  static const bool force_wide = [] { const char* e = std::getenv("RSORT_FORCE_WIDE"); return e && e[0] == '1'; }();
  const bool wide = n > kSmallDescLimit || (force_wide && !ws);   // (the static workspace holds 4-byte descriptor rows only, and its n <= 2^25 never needs more)
  // Chain mode (rsort_kernels.hpp, "XCD chains"): one chain of tiles per XCD, swept by persistent workgroups.  For sorts
  // large enough to keep every XCD busy with its own chain, on devices where the XCC-id probe found all eight XCDs, with
  // the returning-add ranking, 4- and 8-byte keys, 16-byte aligned buffers (the histogram kernel counts per chain in
  // 16-byte packs; the sweeps fetch their keys by 16-byte LDS-DMA).
  // From how many tiles on: measured per key width (tools/ab_tmp.sh sweep, profiles/r02_g_chain_threshold.txt, randomized and
  // ordered shapes).  8-byte keys gain from about 1200 tiles (12 M keys: 0.479 against 0.505 ms; 33 M: 1.16 against 1.25 ms)
  // and are level at 1024; 4-byte keys lose below about 12000 tiles (33 M keys: 0.388 against 0.352 ms with the one-chain
  // kernels, 67 M: 0.69 against 0.67) and are level from 10^8 keys on; 8-byte keys with a 4-byte payload are level
  // everywhere below 8192 tiles.  RSORT_CHAIN_MIN_TILES overrides all three (tests, A/B timing).
  static const long chain_min_env = [] { const char* e = std::getenv("RSORT_CHAIN_MIN_TILES"); return e ? std::atol(e) : -1l; }();
  const long chain_min_tiles = chain_min_env >= 0 ? chain_min_env : value_bytes ? 4096l : key_bytes == 8 ? 1024l : 12288l;
  static const bool chain_env = [] { const char* e = std::getenv("RSORT_CHAIN"); return !(e && e[0] == '0'); }();
  // Pairs: measured at N = 2^30 / 2^29, only 8-byte keys with a 4-byte payload gain (5.07 against 6.00 ms per pass; the
  // other three pair kernels run out of registers in the persistent form and lose), so only those use it.
  const bool chain_shape = value_bytes == 0 ? key_bytes >= 4 : (key_bytes == 8 && value_bytes == 4);
  const bool chain = !ws && chain_env && c->per_xcd_order && c->rank_by_lds_atomic && chain_shape &&
                     static_cast<long>(tiles_for(n, key_bytes, value_bytes, true)) >= chain_min_tiles &&
                     (reinterpret_cast<uintptr_t>(keys) & 15u) == 0 && (reinterpret_cast<uintptr_t>(alt_keys) & 15u) == 0 &&
                     n < (size_t(1) << 35);
  const size_t tiles = tiles_for(n, key_bytes, value_bytes, chain);
  if (tiles > 0x7FFFFFFFull) return fail(c, RS_ERR_ARG, "n too large for one launch");

  const uint64_t sign = 1ull << (key_bytes * 8 - 1);
  const uint64_t all = key_bytes == 8 ? ~0ull : ((1ull << (key_bytes * 8)) - 1ull);
  const uint64_t fmask = transform == RS_XF_IEEE ? all : 0;
  const uint64_t smask = transform == RS_XF_NONE ? 0 : sign;
  // forward_done: the keys are in their transformed form already (the pair form of the hybrid sort hands an array back to
  // the plain loop after its two sweeps, below): nothing is applied as they are read, the inverse still as they are stored
  const uint64_t fmask_rd = forward_done ? 0 : fmask, smask_rd = forward_done ? 0 : smask;

  // Small inputs (at most one tile): one launch of one workgroup does the whole sort in LDS -- no histogram, no
  // descriptors, no pre-check and no host synchronisation (the reference's own small-input shortcut is the insertion
  // sort at n <= 100, radix_sort.hpp:372-392; the result of either is the one stable ascending order).
  // RSORT_SMALL=0 (tests, A/B timing) sends small inputs down the general path instead.
  static const bool small_path = [] { const char* e = std::getenv("RSORT_SMALL"); return !(e && e[0] == '0'); }();
  if (small_path && tiles == 1 && c->rank_by_lds_atomic && !forward_done) {
    int ev = 0;
    if (c->profiling) { RS_HIP(c, hipEventRecord(c->ev[ev++], s)); RS_HIP(c, hipEventRecord(c->ev[ev++], s)); }
    rsk::SmallArgs sa{};
    sa.keys = keys; sa.vals = vals; sa.n = static_cast<unsigned>(n); sa.fmask = fmask; sa.smask = smask;
    sa.begin_bit = begin_bit; sa.end_bit = end_bit;
    launch_small(sa, key_bytes, value_bytes, s);
    RS_HIP(c, hipGetLastError());
    if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
    c->ev_count = c->profiling ? ev : 0;
    return RS_OK;
  }

  // Mid-size sorts (at most kAlwaysSweepTiles tiles) keep their descriptors inside the control block: one memset, no
  // descriptor buffer, and -- below -- no host synchronisation at all.
  const bool mid = tiles <= kAlwaysSweepTiles && !wide;
  int rc;
  unsigned long long span[4] = {0, 0, 0, 0};     // AND and OR of a sample of the keys, [2] = known, [3] = repeats among the 4096 sampled keys (sort_few_distinct)
  if (!ws) {
    if ((rc = ensure(c, c->ctrl, kCtrlBytesChain)) != RS_OK) return rc;
    if (!mid && (rc = ensure(c, c->desc, desc_bytes_for(n))) != RS_OK) return rc;
  } else {
    c->static_used = true;
  }
  if (!alt_keys) {
    if ((rc = ensure(c, c->alt_keys, n * key_bytes)) != RS_OK) return rc;
    alt_keys = c->alt_keys.p;
  }
  if (value_bytes && !alt_vals) {
    if ((rc = ensure(c, c->alt_vals, n * value_bytes)) != RS_OK) return rc;
    alt_vals = c->alt_vals.p;
  }

  char* ctrl = ws ? ws->ctrl : static_cast<char*>(c->ctrl.p);
  void* const desc = mid ? static_cast<void*>(ctrl + kCtrlMidDesc) : ws ? ws->desc : c->desc.p;
  // Mid-size sorts on the context's own control block launch neither the memset nor the scan kernel: the histogram
  // kernel clears what they need (kCtrlHist1) and every sweep scans the 256 digit counts of its pass itself
  // (SweepArgs::counts).  Two launches less in a chain of seven.
  // Not while the stream is being captured into a graph: which copy of the counts is zero is host state, and a replay runs
  // when the host is not looking.  A captured sort keeps its memset (both copies) and clears the counts again when it is
  // done, so that whichever copy the next ordinary sort takes for clear is clear, however replays and calls interleave.
  const bool in_tile_scan = mid && !chain;      // (chain mode at mid sizes happens only under RSORT_CHAIN_MIN_TILES, in tests)
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &capture) != hipSuccess) { (void)hipGetLastError(); capture = hipStreamCaptureStatusNone; }
  const bool capturing = capture != hipStreamCaptureStatusNone;     // (only sorts that never wait for the device can be: mid-size ones and the workspace form)
  const bool self_clearing = in_tile_scan && !ws && !capturing;
  const int hist_copy = self_clearing ? c->hist_copy : 0;
  auto* ghist = reinterpret_cast<unsigned long long*>(ctrl + (hist_copy ? kCtrlHist1 : kCtrlHist));
  auto* gbase = reinterpret_cast<unsigned long long*>(ctrl + kCtrlBase);
  auto* tickets = reinterpret_cast<unsigned*>(ctrl + kCtrlTicket);
  auto* error_flag = reinterpret_cast<unsigned*>(ctrl + kCtrlError);

  // Order pre-check (isSortedBi :269-305 / key::isSorted :746-769): ascending on the sorted field -> nothing to do
  // (a stable sort of ordered input is the identity); descending, whole key, no payload -> reverse in place.
  // Not for mid-size device-resident sorts: there the two launches and the read-back cost as much as the sort
  // (the host-buffer entry points, which wait for the device anyway and save a download when the input is in order, ask for it).
  if (c->precheck && !ws && (!mid || c->host_call) && !forward_done) {
    rsk::OrderArgs o{};
    o.keys = keys; o.flags = error_flag + 8; o.n = n; o.fmask = fmask_rd; o.smask = smask_rd;
    o.begin_bit = begin_bit; o.bits = end_bit - begin_bit;
    RS_HIP(c, hipMemsetAsync(o.flags, 0, sizeof(unsigned), s));
    // A one-workgroup probe over the head of the array first: on unordered input it sets both flags and the
    // full-size launch exits at once; without it every workgroup of that launch would start on clear flags and
    // queue its atomics on the one flag word (measured 0.3 ms at N = 2^30).
    auto launch_order = [&](const rsk::OrderArgs& oa, unsigned grid) {
      switch (key_bytes) {
        case 1: launch_order_k<uint8_t>(oa, grid, s); break;
        case 2: launch_order_k<uint16_t>(oa, grid, s); break;
        case 4: launch_order_k<uint32_t>(oa, grid, s); break;
        default: launch_order_k<uint64_t>(oa, grid, s); break;
      }
    };
    rsk::OrderArgs probe = o;
    probe.n = std::min<size_t>(n, 65536);
    launch_order(probe, 1);
    if (n > probe.n) launch_order(o, static_cast<unsigned>(std::min<size_t>((n + 2047) / 2048, 256 * 8)));
    unsigned flags = 3;
    RS_HIP(c, hipMemcpyAsync(&flags, o.flags, sizeof(flags), hipMemcpyDeviceToHost, s));
    RS_HIP(c, hipStreamSynchronize(s));
    const bool whole = begin_bit == 0 && end_bit == key_bytes * 8;
    if ((flags & 1u) == 0) { c->ev_count = 0; c->last_unchanged = true; return RS_OK; }
    if ((flags & 2u) == 0 && whole && value_bytes == 0) {
      switch (key_bytes) {
        case 1: launch_reverse_k<uint8_t>(keys, n, s); break;
        case 2: launch_reverse_k<uint16_t>(keys, n, s); break;
        case 4: launch_reverse_k<uint32_t>(keys, n, s); break;
        default: launch_reverse_k<uint64_t>(keys, n, s); break;
      }
      RS_HIP(c, hipGetLastError());
      c->ev_count = 0;
      return RS_OK;
    }
  }

  // Few distinct keys (rsd::, rsort_groups.hpp): large keys-only sorts of whole 4- and 8-byte keys first let one workgroup
  // count the distinct keys of a sample; when they are few, the whole array is counted the same way and the result is
  // written as runs -- one read and one write of the keys.  More distinct keys than the table serves anywhere in the array:
  // the overflow word says so and the sort goes on as usual (the counting read was then wasted: about a tenth of a sort).
  if (c->distinct_path && !ws && !mid && value_bytes == 0 && (key_bytes == 4 || key_bytes == 8) && begin_bit == 0 && end_bit == key_bytes * 8 &&
      (reinterpret_cast<uintptr_t>(keys) & static_cast<uintptr_t>(key_bytes - 1)) == 0) {
    int done = 0;
    if ((rc = sort_few_distinct(c, keys, n, key_bytes, fmask, smask, &done, span)) != RS_OK) return rc;
    if (done) return RS_OK;
  }

  // Narrow hybrid form (keys-only sorts between the mid-size regime and the sizes where 65536 groups pay): the same two
  // sweeps and in-LDS finish, but on the top 8 + b bits only, b chosen so that the 2^(8+b) groups hold at most kNarrowGroupTarget keys
  // each (the window of the smallest workgroups of the group kernel, four of which share a CU) -- the sort is then a sort on the bit range [sort_begin, end_bit) (two digits: 8 bits and, on top, b bits) followed
  // by the wide group kernel on the bits below.  Decided from n alone, before the histogram (which then covers only the two
  // swept digits); whether the groups really fit is known from those counts, and the plain loop takes over when they do not.
  // Keys of a few thousand to a few million distinct values (the sample's repeats say so, when the few-distinct probe ran):
  // every value's copies share one group, whose low digits then take one or two values -- the group sorts' LDS counters
  // queue, groups outgrow their windows -- and the plain loop is the faster form (4-byte keys: 3 * 10^8 keys of 2 * 10^4 /
  // 2 * 10^5 values 7.3 / 3.9 ms against 2.9; at 2 * 10^6 values, 4 repeats expected, even).  8-byte keys cope with all but
  // the smallest pools, so their limit is a few thousand values.  As shipped only: a forced form stays forced.
  const unsigned long long sample_repeats = span[2] ? span[3] : 0;
  const bool clustered = c->hybrid_mode == -1 && sample_repeats >= (key_bytes == 4 ? kClusteredRepeats : kClusteredRepeatsWide);
  int sort_begin = begin_bit, narrow_bits = 0;
  if ((c->hybrid_mode == -1 || c->hybrid_mode == 2) && !ws && !mid && !wide && !forward_done && value_bytes == 0 && (key_bytes == 4 || key_bytes == 8) &&
      begin_bit == 0 && end_bit == key_bytes * 8 && c->rank_by_lds_atomic && (reinterpret_cast<uintptr_t>(keys) & static_cast<uintptr_t>(key_bytes - 1)) == 0 &&
      n < (key_bytes == 4 ? kHybridMinKeys : hybrid_min_keys_wide()) && (c->hybrid_mode == 2 || n >= (key_bytes == 4 ? kNarrowMinKeys : kNarrowMinKeysWide))) {
    int g = 9;
    while (g < 16 && static_cast<double>(n) / static_cast<double>(size_t(1) << g) > narrow_group_target()) ++g;
    // (the sample of the few-distinct probe, when it ran: bits that are the same in every sampled key.  The top g bits must
    // all differ somewhere, or the groups cannot be balanced -- 8-byte keys that hold small integers -- and the plain loop,
    // which skips constant digits, is the better start.  A hint only: the histogram below still decides.)
    const unsigned long long need = ((1ull << g) - 1ull) << (end_bit - g);
    if (clustered) {
      // (the plain loop)
    } else if (!span[2] || (((span[0] ^ span[1]) & need) == need) || c->hybrid_mode == 2) {
      narrow_bits = g - 8;
      sort_begin = end_bit - g;
      passes = 2;
    }
  }
  // Prefixed hybrid form: keys whose leading t bits (1..7) are the same in every key -- what a multi-GPU rank holds after the
  // exchange when the split points fall on top-digit edges (dist.py snaps them there when that costs no balance), or plain
  // data of a narrower range.  Groups of equal top 16 bits would be 2^t times the usual size; the 16 bits BELOW the prefix
  // make groups of the usual size.  So the sort is a sort on the bit range [end - t - 16, end): two full digits, which are
  // swept, and on top a partial digit of t bits that must come out constant -- the histogram says so exactly, the sample
  // (span) only suggests it -- followed by the in-LDS finish of the end - t - 16 bits below.
  int prefix_bits = 0;
  if (!narrow_bits && !clustered && (c->hybrid_mode == -1 || c->hybrid_mode == 3) && span[2] && !ws && !mid && !forward_done && value_bytes == 0 &&
      (key_bytes == 4 || key_bytes == 8) && begin_bit == 0 && end_bit == key_bytes * 8 && c->rank_by_lds_atomic &&
      (reinterpret_cast<uintptr_t>(keys) & static_cast<uintptr_t>(key_bytes - 1)) == 0 && n < (size_t(1) << 32) &&
      (c->hybrid_mode == 3 || n >= (key_bytes == 4 ? kHybridMinKeys : hybrid_min_keys_wide()))) {
    const unsigned long long vary = (span[0] ^ span[1]) & all;
    if (vary) {
      const int t = end_bit - (64 - __builtin_clzll(vary));       // leading bits the sampled keys share
      if (t >= 1 && t <= 7) { prefix_bits = t; sort_begin = end_bit - t - 16; passes = 3; }
    }
  }

  unsigned chain_tiles = 0, chunk_shift = 0, chunk_mul = 0;
  if (chain) {
    // chains of U8 << r tiles, r such that positions >> (log2(tile) + r) stay below 256: the chain of an output position is
    // then one shift, one 24-bit multiply and one shift (checked here for every value the kernel can meet)
    // (tile = odd * 2^q keys: positions are counted in units of 2^(q + r) keys, a chain is u8 units with u8 a multiple of
    // `odd`, so that every chain is a whole number of tiles; odd = 1 for the power-of-two tiles of the shipped geometry)
    const size_t tk = tile_keys(key_bytes, value_bytes, true);
    unsigned q = 0;
    while (((tk >> q) & 1u) == 0) ++q;
    const unsigned odd = static_cast<unsigned>(tk >> q);
    const size_t units0 = (n + (size_t(1) << q) - 1) >> q;
    unsigned r = 0;
    while (((units0 + (size_t(1) << r) - 1) >> r) > 255 - rsk::kXcds * odd) ++r;
    const unsigned u_max = static_cast<unsigned>((units0 + (size_t(1) << r) - 1) >> r);
    unsigned u8 = (u_max + rsk::kXcds - 1) / rsk::kXcds;
    u8 = (u8 + odd - 1) / odd * odd;
    if ((static_cast<size_t>(u8) << r) % odd != 0) return fail(c, RS_ERR_KERNEL, "chain split arithmetic");
    chain_tiles = static_cast<unsigned>((static_cast<size_t>(u8) << r) / odd);
    chunk_shift = q + r;
    chunk_mul = (65536u + u8 - 1) / u8;
    for (unsigned u = 0; u < u_max; ++u)
      if (((u * chunk_mul) >> 16) != u / u8 || u / u8 >= rsk::kXcds) return fail(c, RS_ERR_KERNEL, "chain split arithmetic");
  }

  int ev = 0;
  if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
  // counts, bases and tickets; NOT the error word behind them: it keeps a fault until check_kernel_error has read it
  if (!self_clearing) {
    if ((rc = clear_async(c, ctrl + kCtrlHist, kCtrlMidDesc - kCtrlHist + (mid ? tiles * 256 * 4 : 0), s, capturing)) != RS_OK) return rc;
    if (!ws && !capturing) c->hist_copy = 1;
  }
  if (chain) RS_HIP(c, hipMemsetAsync(ctrl + kCtrlChainHist, 0, kCtrlBytesChain - kCtrlChainHist, s));
  // Chain mode keeps 4-byte descriptors and offsets up to 2^32 keys: a descriptor holds a count inside ONE chain (at most an
  // eighth of the keys plus a tile, below 2^30), and output offsets below 2^32 fit the 4-byte tables.  The one-chain
  // kernels' descriptors hold counts over the whole array and need the 8-byte form from 2^30 keys on.
  const bool wide_chain = (force_wide && !ws) || n >= (size_t(1) << 32);
  // (the rows this sort uses, not the whole reservation.  Should the few-valued rule below send a chain-mode sort down the
  // one-chain kernels, those use their own tile size and, beyond 2^30 keys, 8-byte rows: about the same bytes as the 4-byte
  // rows of the 8192-key tiles, but rounded up separately -- ceil(n / 16384) * 8 can exceed ceil(n / 8192) * 4 by one row
  // pair -- so the larger of the two is cleared.  The first version cleared the chain form's bytes only and refused the
  // sort when they fell short: a multi-GPU rank that held a little more than 2^30 four-byte keys after the exchange, whose
  // top digit takes few values by construction, hit exactly that.)
  size_t desc_clear = tiles * 256 * ((chain ? wide_chain : wide) ? 8 : 4);
  if (chain) desc_clear = std::max(desc_clear, tiles_for(n, key_bytes, value_bytes, false) * 256 * (wide ? size_t(8) : size_t(4)));
  if (ws && !mid && desc_clear > kNoAllocDescBytes) return fail(c, RS_ERR_ARG, "descriptor rows exceed the static workspace");   // (cannot happen: n <= kNoAllocMaxKeys, 4-byte rows)
  if (!mid && (rc = clear_async(c, desc, desc_clear, s, capturing)) != RS_OK) return rc;

  auto* chain_hist = reinterpret_cast<unsigned long long*>(ctrl + kCtrlChainHist);
  auto* next_cnt = reinterpret_cast<unsigned long long*>(ctrl + kCtrlNextCnt);
  rsk::HistArgs h{};
  h.keys = keys; h.ghist = ghist; h.n = n; h.fmask = fmask_rd; h.smask = smask_rd;
  h.begin_bit = sort_begin; h.end_bit = end_bit; h.passes = passes;
  h.chain_hist = chain ? chain_hist : nullptr;
  h.chain_keys = chain ? static_cast<uint64_t>(chain_tiles) * tile_keys(key_bytes, value_bytes, true) : 0;
  if (self_clearing) {
    h.zero[0] = ctrl + (hist_copy ? kCtrlHist : kCtrlHist1);
    h.zero_vecs[0] = static_cast<unsigned>(kMaxPasses * 256 * 8 / 16);
    h.zero[1] = ctrl + kCtrlTicket;
    h.zero_vecs[1] = static_cast<unsigned>((kCtrlMidDesc - kCtrlTicket + tiles * 256 * 4) / 16);
  }
  launch_hist(h, key_bytes, chain ? 256u * hist_wgs_per_cu(key_bytes) : hist_grid(n, key_bytes), s);
  RS_HIP(c, hipGetLastError());         // (a histogram that was not launched has not cleared the other copy of the counts)
  if (self_clearing) c->hist_copy ^= 1;
  if (!in_tile_scan)
    hipLaunchKernelGGL(rsk::digit_scan_kernel, dim3(passes), dim3(rsk::kRadix), 0, s, ghist, gbase,
                       static_cast<const unsigned long long*>(chain ? chain_hist : nullptr));
  if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));

  // Pass skipping (the device-side counterpart of getMaxLength, radix_sort.hpp:307-370, which lets the reference
  // drop high bytes that are zero): a pass whose digit is the same for every key is the identity permutation, so it
  // is not launched.  Needs the counts on the host: 2 KB per pass and one synchronisation.
  // Short sorts of few passes do not ask: the synchronisation (about 25 us) costs more than the passes it could save
  // (under 10 us each below kAlwaysSweepTiles tiles).
  int live[kMaxPasses];     // the passes that are launched, in order
  int n_live = 0;
  int bins_of[kMaxPasses];  // digit values in use, per launched pass (known only with the counts on the host)
  unsigned long long h_hist_keep[kMaxPasses * 256];   // the counts themselves (hybrid form, below)
  if (ws || mid) {
    for (int p = 0; p < passes; ++p) live[n_live++] = p;
  } else {
    unsigned long long h_hist[kMaxPasses * 256];
    RS_HIP(c, hipMemcpyAsync(h_hist, ghist, sizeof(unsigned long long) * 256 * passes, hipMemcpyDeviceToHost, s));
    RS_HIP(c, hipStreamSynchronize(s));
    std::memcpy(h_hist_keep, h_hist, sizeof(unsigned long long) * 256 * passes);
    for (int p = 0; p < passes; ++p) {
      bool constant = false;
      for (int d = 0; d < 256; ++d)
        if (h_hist[p * 256 + d] == n) { constant = true; break; }
      if (!constant) {
        int bins = 0;
        for (int d = 0; d < 256; ++d) bins += h_hist[p * 256 + d] != 0;
        bins_of[n_live] = bins;
        live[n_live++] = p;
      }
    }
  }
  c->last_all_equal = !(ws || mid) && n_live == 0;
  // Few-valued 4-byte keys (the 256-value "duplicates" shape: about 160 of 256 bins in use per pass) go down the
  // one-chain kernels: staged in digit order, their equal keys sit next to each other, and the chain kernels' counting adds
  // then cost more than the chained look-back saves (N = 2^30: 11.1 ms against 9.7 ms; 8-byte keys gain either way).
  // The histogram and the zeroed descriptor rows serve both forms.  "Few-valued" = most of the launched passes see fewer
  // than RS_FEW_VALUED_BINS digit values, not just one of them: a multi-GPU rank's keys share their top bits by construction
  // (top digit: 256 / ranks values) and are otherwise as random as before the exchange.
  bool sweep_chain = chain;
  size_t sweep_tiles = tiles;
#ifndef RS_FEW_VALUED_BINS
#define RS_FEW_VALUED_BINS 200
#endif
  static const int few_bins = [] { const char* e = std::getenv("RSORT_FEW_VALUED_BINS"); return e ? std::atoi(e) : RS_FEW_VALUED_BINS; }();   // (A/B timing)
  int few_passes = 0;
  if (!(ws || mid))
    for (int i = 0; i < n_live; ++i) few_passes += bins_of[i] < few_bins;
  if (chain && key_bytes == 4 && value_bytes <= 4 && 2 * few_passes > n_live) {
    sweep_chain = false;
    sweep_tiles = tiles_for(n, key_bytes, value_bytes, false);
  }

  // Hybrid form for whole 4-byte keys without payload (rsort_groups.hpp): only the two top digits are swept; the sweeps
  // leave the array grouped by its top 16 bits, and each group's low 16 bits are sorted in LDS by one workgroup -- one read
  // and one write of the keys instead of two of each.  Same result (keys only: equal keys are bit-identical).
  // Chosen when the groups will (almost all) fit the LDS window: sizes where the mean group does, top digits whose
  // counts say so (the product of the two largest digit shares bounds the largest group when the digits are independent;
  // groups that turn out larger anyway are sorted by rsk::group_big_kernel, correct at any size, only slower), and low digits
  // that are not few-valued (their LDS counters would queue; the four-pass form skips or peels such digits).
  // rs_ctx_set_hybrid / RSORT_HYBRID: 0 never, 1 whenever the shape allows it (tests: any size, any distribution).
  const int hybrid_env = c->hybrid_mode;
  bool hybrid = false;
  double group_bound = 0;          // the largest group, as the digit counts bound it (0: not known)
  const int top_lo = prefix_bits ? 0 : passes - 2, top_hi = prefix_bits ? 1 : passes - 1;   // the two swept digits (whole-key sorts: passes = key bytes; the narrow and the prefixed form: 0 and 1)
  // Pair form (4- or 8-byte keys, 4-byte payload; stable): the same two sweeps, the wide group kernel with the payload beside the
  // keys and a smaller window; a group beyond it sends the array -- as the sweeps left it: a stable rearrangement of the
  // input, keys transformed -- through the plain loop instead (there is no big-group kernel for pairs).
  const bool pair_form = value_bytes == 4 && (key_bytes == 8 || key_bytes == 4) && vals != nullptr;
  const size_t pair_cap = key_bytes == 4 ? kPair32Cap : kWidePairGroupCap;
  const size_t group_cap = pair_form ? pair_cap : key_bytes == 4 ? (n / kGroupCount < small_group_mean_max() ? size_t(kSmallGroupBlock) * kGroupMaxVec * 4 - 3 : kGroupCap) : kWideGroupCap;
  if (hybrid_env != 0 && hybrid_env != 2 && hybrid_env != 3 && !narrow_bits && !prefix_bits && !ws && !mid && !(sweep_chain ? wide_chain : wide) && !forward_done && (value_bytes == 0 || pair_form) && (key_bytes == 4 || key_bytes == 8) && begin_bit == 0 &&
      end_bit == key_bytes * 8 && n_live == key_bytes && n < (size_t(1) << 32) && c->rank_by_lds_atomic &&
      (reinterpret_cast<uintptr_t>(keys) & static_cast<uintptr_t>(key_bytes - 1)) == 0) {
    unsigned long long mx[kMaxPasses] = {};
    for (int p = 0; p < key_bytes; ++p)
      for (int d = 0; d < 256; ++d) mx[p] = std::max(mx[p], h_hist_keep[p * 256 + d]);
    const double bound = static_cast<double>(mx[top_hi]) * static_cast<double>(mx[top_lo]) / static_cast<double>(n);   // largest group if the top digits are independent
    group_bound = bound;
    if (hybrid_env == 1) {
      hybrid = true;
    } else {
      hybrid = !clustered && n >= (pair_form ? (key_bytes == 4 ? kHybridMinPairs : kHybridMinPairsWide) : key_bytes == 4 ? kHybridMinKeys : hybrid_min_keys_wide()) && bound <= (key_bytes == 8 && !pair_form ? 0.97 : 0.95) * group_cap;
      for (int p = 0; p < top_lo; ++p) hybrid = hybrid && mx[p] <= n / 16;
    }
  }
  uint64_t shared_prefix = 0;      // prefixed form: the bits above the swept digits, in place
  if (narrow_bits || prefix_bits) {
    unsigned long long mx[2] = {0, 0};
    for (int p = 0; p < 2; ++p)
      for (int d = 0; d < 256; ++d) mx[p] = std::max(mx[p], h_hist_keep[p * 256 + d]);
    const double bound = static_cast<double>(mx[1]) * static_cast<double>(mx[0]) / static_cast<double>(n);
    group_bound = bound;
    if (narrow_bits) {
      hybrid = n_live == 2 && bound <= 0.95 * kWideGroupCap;
    } else {
      int only = -1;               // the one value of the partial digit on top, if it is one
      for (int d = 0; d < 256; ++d)
        if (h_hist_keep[2 * 256 + d] == n) only = d;
      // (0.97 of the window for 8-byte keys: a rank of eight holds 2^30 keys give or take half a per cent, groups of 16384 +- 128 keys in a window of 17408)
      hybrid = only >= 0 && n_live == 2 && live[0] == 0 && live[1] == 1 && !(sweep_chain ? wide_chain : wide) && bound <= (key_bytes == 8 ? 0.97 : 0.95) * group_cap;
      if (c->hybrid_mode == -1)
        for (int p = 0; p < 2; ++p) hybrid = hybrid && mx[p] <= n / 16;
      if (hybrid) shared_prefix = static_cast<uint64_t>(only) << (end_bit - prefix_bits);
    }
    if (!hybrid) {     // a constant or skewed prefix: the plain loop, from the start (nothing has been moved; the checks that ran need not run again)
      const int keep = c->hybrid_mode;
      const bool keep_pre = c->precheck, keep_distinct = c->distinct_path;
      c->hybrid_mode = 0; c->precheck = false; c->distinct_path = false;
      rc = sort_device(c, keys, vals, alt_keys, alt_vals, n, key_bytes, value_bytes, transform, begin_bit, end_bit, ws, false);
      c->hybrid_mode = keep; c->precheck = keep_pre; c->distinct_path = keep_distinct;
      return rc;
    }
  }
  if (hybrid) {
    c->last_hybrid = narrow_bits ? 2 : prefix_bits ? 3 : 1;
    if ((rc = ensure(c, c->groups, kGroupsBytes)) != RS_OK) return rc;
    live[0] = top_lo; live[1] = top_hi; n_live = 2;
  }

  // Mid-size keys-only sorts of whole 4- and 8-byte keys: the short form -- ONE sweep on the top digit, then one launch in
  // which workgroup g finishes the keys of top digit g in LDS (rsk::group_sort_mid_kernel) -- whenever every group fits its
  // window.  Whether they do is only known on the device and a mid-size sort does not wait for it, so the kernels of both
  // forms are queued and decide for themselves (rsk::SweepArgs::mid_counts): a sort that takes the short form pays K - 1
  // launches that leave at once; one that cannot (keys that share their top byte: small integers) runs the LSD loop as before.
  const bool mid_short = mid && in_tile_scan && !ws && !forward_done && c->hybrid_mode != 0 && value_bytes == 0 && (key_bytes == 4 || key_bytes == 8) &&
                         begin_bit == 0 && end_bit == key_bytes * 8 && c->rank_by_lds_atomic && n_live == key_bytes;
  const unsigned mid_cap = key_bytes == 4 ? kMidGroupBlock * kMidGroupItems : static_cast<unsigned>(kWideGroupCap);

  void* kin = keys; void* kout = alt_keys;
  void* vin = vals; void* vout = alt_vals;
  if (chain && !sweep_chain && sweep_tiles * 256 * (wide ? size_t(8) : size_t(4)) > desc_clear)
    return fail(c, RS_ERR_KERNEL, "descriptor rows of the one-chain fallback");      // (cannot happen: desc_clear covers both forms)
  const SweepLaunch launch{sweep_tiles, key_bytes, value_bytes, sweep_chain ? wide_chain : wide, c->rank_by_lds_atomic, sweep_chain, static_cast<unsigned>(c->cus), s};
  rsk::SweepArgs last_sweep{};
  for (int done = 0; done < n_live; ++done) {
    const int p = live[done];
    rsk::SweepArgs a{};
    a.keys_in = kin; a.keys_out = kout; a.vals_in = vin; a.vals_out = vout;
    a.desc = desc; a.ticket = tickets + p * rsk::kTicketWords; a.gbase = gbase + p * rsk::kRadix; a.error_flag = error_flag;
    a.counts = in_tile_scan ? ghist + p * rsk::kRadix : nullptr;
    a.n = n;
    a.fmask_in = done == 0 ? fmask_rd : 0; a.smask_in = done == 0 ? smask_rd : 0;              // first launched pass
    const bool stores_result = done == n_live - 1 && !hybrid;                                  // last launched pass (the hybrid form's group kernel stores the result)
    a.fmask_out = stores_result ? fmask : 0; a.smask_out = stores_result ? smask : 0;
    a.shift = sort_begin + p * rsk::kRadixBits;
    a.bits = std::min(rsk::kRadixBits, end_bit - a.shift);
    a.per_xcd_order = c->per_xcd_order && sweep_tiles >= kPerXcdMinTiles ? 1 : 0;
    a.incl_code = 2u + (done & 1);          // alternates over the LAUNCHED passes (descriptor parity, rsort_kernels.hpp)
    if (sweep_chain) {
      // this pass' digit counts per chain: from the histogram kernel for the first launched pass, from the previous sweep after that
      a.chain_cnt = done == 0 ? chain_hist + p * rsk::kRadix : next_cnt + static_cast<size_t>(done - 1) * rsk::kXcds * rsk::kRadix;
      a.chain_cnt_stride = done == 0 ? rsk::kChainHistStride : rsk::kRadix;
      a.chain_tiles = chain_tiles;
      a.chunk_shift = chunk_shift; a.chunk_mul = chunk_mul;
      a.desc_local = c->desc_local ? 1 : 0;
      if (done + 1 < n_live) {
        a.next_cnt = next_cnt + static_cast<size_t>(done) * rsk::kXcds * rsk::kRadix;
        a.next_shift = sort_begin + live[done + 1] * rsk::kRadixBits;
        a.next_bits = std::min(rsk::kRadixBits, end_bit - a.next_shift);
      }
    }
    if (mid_short) {
      a.mid_counts = ghist + top_hi * rsk::kRadix;
      a.mid_cap = mid_cap;
      a.mid_role = p == top_hi ? 2 : 1;
      a.mid_keys_in = keys; a.mid_keys_out = alt_keys;            // the short form's only sweep: caller's keys -> alternate buffer
      a.mid_fmask_in = fmask_rd; a.mid_smask_in = smask_rd;
    }
    launch_sweep_k(a, launch);
    if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
    std::swap(kin, kout);
    std::swap(vin, vout);
    last_sweep = a;
  }
  if (mid_short) {
    rsk::MidArgs m{};
    m.keys_in = alt_keys; m.keys_out = keys; m.counts = ghist + top_hi * rsk::kRadix; m.cap = mid_cap;
    m.low_bits = top_hi * rsk::kRadixBits; m.fmask_out = fmask; m.smask_out = smask;
    if (key_bytes == 4)
      hipLaunchKernelGGL((rsk::group_sort_mid_kernel<uint32_t, kMidGroupBlock, kMidGroupItems>), dim3(rsk::kRadix), dim3(kMidGroupBlock), 0, s, m);
    else
      hipLaunchKernelGGL((rsk::group_sort_mid_kernel<uint64_t, kWideGroupBlock, kWideGroupItems>), dim3(rsk::kRadix), dim3(kWideGroupBlock), 0, s, m);
    if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
  }
  if (hybrid) {
    // (two sweeps: the grouped keys are back in `keys`, the second sweep's input is intact in alt_keys)
    auto* start = static_cast<unsigned*>(c->groups.p);
    unsigned* big_list = start + kGroupCount + 1;
    unsigned* mid_list = big_list + kGroupCount + 1;
    RS_HIP(c, hipMemsetAsync(big_list, 0, 2 * sizeof(unsigned), s));
    RS_HIP(c, hipMemsetAsync(mid_list, 0, sizeof(unsigned), s));
    rsk::TableArgs t{};
    t.desc = static_cast<const uint32_t*>(desc);
    t.keys_in = last_sweep.keys_in;
    t.gbase_lo = gbase + top_lo * rsk::kRadix; t.gbase_hi = gbase + top_hi * rsk::kRadix;
    t.chain_cnt = sweep_chain ? last_sweep.chain_cnt : nullptr;
    t.chain_cnt_stride = last_sweep.chain_cnt_stride;
    t.chain_tiles = sweep_chain ? chain_tiles : 0u;
    t.tile = static_cast<unsigned>(tile_keys(key_bytes, value_bytes, sweep_chain));
    t.n = static_cast<unsigned>(n);
    t.shift_hi = sort_begin + top_hi * rsk::kRadixBits;
    t.start = start;
    rsk::GroupArgs ga{};
    ga.keys = keys; ga.start = start; ga.groups = narrow_bits ? 256u << narrow_bits : kGroupCount; ga.low_bits = sort_begin + top_lo * rsk::kRadixBits;
    ga.fmask_out = fmask; ga.smask_out = smask; ga.top_or = shared_prefix;
    ga.big_list = big_list; ga.big_cap = kGroupCount;
    ga.n = static_cast<unsigned>(n); ga.error_flag = error_flag;
    rsk::BigArgs ba{};
    ba.keys = keys; ba.alt = alt_keys; ba.start = start; ba.big_list = big_list; ba.big_cap = kGroupCount; ba.low_bits = ga.low_bits;
    ba.fmask_out = fmask; ba.smask_out = smask;
    ba.n = static_cast<unsigned>(n); ba.error_flag = error_flag;
    // As shipped (no form forced, no test window): the groups' real sizes are looked at before anything is sorted in LDS.  The
    // bound that let the sort come this way assumes independent digits; keys of a few thousand distinct values (beyond the
    // few-distinct path) put whole values' worth of keys into single groups, and the big-group kernel, meant for the odd
    // oversize group, would then do most of the sort (5 * 10^7 8-byte keys of 2000 values: 7.3 ms against 1.8 for the plain
    // loop).  More than 1/64 of the keys in groups beyond the window: the plain loop takes over from what the sweeps left, as
    // in the pair form.  One 8-byte read-back -- which holds up the launches behind it, 40-60 us in all: sorts of at least
    // kGroupCheckMinKeys keys only.  Smaller sorts rely on what the group kernels do by themselves: a group beyond the window
    // whose keys are all one key is left as it is (rsk::group_oversize), and the small workgroups pass what they cannot hold
    // on to the large ones (mid_list).
    int fit_rc = RS_OK;
    unsigned largest_group = 0;          // as measured (0: not measured -- a form forced, a test window)
    auto groups_fit = [&](unsigned groups, size_t window) -> bool {
      if (c->hybrid_mode != -1 || c->group_max_vec || n < group_check_min_keys()) return true;
      unsigned seen[2] = {0, 0};
      hipLaunchKernelGGL(rsk::group_max_kernel, dim3((groups + rsk::kRadix - 1) / rsk::kRadix), dim3(rsk::kRadix), 0, s, static_cast<const unsigned*>(start), big_list, groups,
                         static_cast<unsigned>(std::min<size_t>(window, 0xFFFFFFFFu)));
      if (hipMemcpyAsync(seen, big_list, sizeof(seen), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess ||
          hipMemsetAsync(big_list, 0, 2 * sizeof(unsigned), s) != hipSuccess) {
        fit_rc = fail(c, RS_ERR_KERNEL, "reading the largest group of a hybrid sort");
        return false;
      }
      largest_group = seen[0];
      if (seen[1] <= n / 64) return true;
      c->last_hybrid = 0;
      const int keep = c->hybrid_mode;
      const bool keep_pre = c->precheck, keep_distinct = c->distinct_path;
      c->hybrid_mode = 0; c->precheck = false; c->distinct_path = false;
      fit_rc = sort_device(c, keys, vals, alt_keys, alt_vals, n, key_bytes, value_bytes, transform, begin_bit, end_bit, ws, true);
      c->hybrid_mode = keep; c->precheck = keep_pre; c->distinct_path = keep_distinct;
      return false;
    };
    if (key_bytes == 4 && narrow_bits && !pair_form) {    // whole keys in LDS (the low part is wider than the 16-bit staging of the kernel below)
      hipLaunchKernelGGL(rsk::group_table_kernel<uint32_t>, dim3(rsk::kRadix), dim3(rsk::kRadix), 0, s, t);
      const double gb = group_bound > 0 ? group_bound : 1.1 * static_cast<double>(n) / static_cast<double>(256u << narrow_bits);
      if (!groups_fit(256u << narrow_bits, kWideGroupCap)) return fit_rc;
      const int wg = !wide_small_blocks() ? 1024 : largest_group ? (largest_group <= 4352 ? 256 : largest_group <= 8704 ? 512 : 1024) : gb <= 0.9 * 4352 ? 256 : gb <= 0.95 * 8704 ? 512 : 1024;
      const size_t wcap = size_t(wg) * kWideGroupItems;
      ga.max_vec = c->group_max_vec ? static_cast<unsigned>(std::min<size_t>(size_t(c->group_max_vec) * 4u, wcap)) : static_cast<unsigned>(wcap);   // (keys, for this kernel)
      if (wg < 1024) ga.mid_list = mid_list;
      if (wg == 256)
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint32_t, rsk::NoVal, 256, kWideGroupItems>), dim3(256u << narrow_bits), dim3(256), 0, s, ga);
      else if (wg == 512)
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint32_t, rsk::NoVal, 512, kWideGroupItems>), dim3(256u << narrow_bits), dim3(512), 0, s, ga);
      else
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint32_t, rsk::NoVal, kWideGroupBlock, kWideGroupItems>), dim3(256u << narrow_bits), dim3(kWideGroupBlock), 0, s, ga);
      if (wg < 1024) {     // what the small workgroups could not hold
        rsk::GroupArgs gm = ga;
        gm.mid_list = nullptr; gm.from_list = mid_list;
        gm.max_vec = c->group_max_vec ? static_cast<unsigned>(std::min<size_t>(size_t(c->group_max_vec) * 4u, kWideGroupCap)) : static_cast<unsigned>(kWideGroupCap);
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint32_t, rsk::NoVal, kWideGroupBlock, kWideGroupItems, -1, true>), dim3(static_cast<unsigned>(c->cus)), dim3(kWideGroupBlock), 0, s, gm);
      }
      if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
      hipLaunchKernelGGL((rsk::group_big_kernel<uint32_t, 512, 8>), dim3(static_cast<unsigned>(c->cus) * 2u), dim3(512), 0, s, ba);
    } else if (key_bytes == 4 && !pair_form) {
      hipLaunchKernelGGL(rsk::group_table_kernel<uint32_t>, dim3(rsk::kRadix), dim3(rsk::kRadix), 0, s, t);
      // Small groups (sorts of some 10^7 to 10^8 keys: a thousand keys per group) go to workgroups of four waves, six of
      // which share a CU, instead of leaving most of 512 threads idle; the few groups beyond their window (10237 keys) to the
      // big-group kernel as ever.
      const bool small_groups = n / kGroupCount < small_group_mean_max();
      const unsigned window_vec = small_groups ? kSmallGroupBlock * kGroupMaxVec : kGroupBlock * kGroupMaxVec;
      ga.max_vec = c->group_max_vec ? std::min<unsigned>(c->group_max_vec, window_vec) : window_vec;
      if (!groups_fit(kGroupCount, kGroupCap)) return fit_rc;
      if (small_groups) {
        ga.mid_list = mid_list;
        hipLaunchKernelGGL((rsk::group_sort_kernel<uint32_t, kSmallGroupBlock, kGroupMaxVec, 6>), dim3(kGroupCount), dim3(kSmallGroupBlock), 0, s, ga);
        rsk::GroupArgs gm = ga;          // what the small workgroups could not hold
        gm.mid_list = nullptr; gm.from_list = mid_list;
        gm.max_vec = c->group_max_vec ? std::min<unsigned>(c->group_max_vec, kGroupBlock * kGroupMaxVec) : kGroupBlock * kGroupMaxVec;
        hipLaunchKernelGGL((rsk::group_sort_kernel<uint32_t, kGroupBlock, kGroupMaxVec, 3, true>), dim3(static_cast<unsigned>(c->cus) * 3u), dim3(kGroupBlock), 0, s, gm);
      } else {
        hipLaunchKernelGGL((rsk::group_sort_kernel<uint32_t, kGroupBlock, kGroupMaxVec, 3>), dim3(kGroupCount), dim3(kGroupBlock), 0, s, ga);
      }
      if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
      hipLaunchKernelGGL((rsk::group_big_kernel<uint32_t, 512, 8>), dim3(static_cast<unsigned>(c->cus) * 2u), dim3(512), 0, s, ba);
    } else if (pair_form) {
      if (key_bytes == 4) hipLaunchKernelGGL(rsk::group_table_kernel<uint32_t>, dim3(rsk::kRadix), dim3(rsk::kRadix), 0, s, t);
      else hipLaunchKernelGGL(rsk::group_table_kernel<uint64_t>, dim3(rsk::kRadix), dim3(rsk::kRadix), 0, s, t);
      ga.vals = vals;
      ga.max_vec = c->group_max_vec ? static_cast<unsigned>(std::min<size_t>(size_t(c->group_max_vec) * 4u, pair_cap)) : static_cast<unsigned>(pair_cap);
      // the largest group, before anything is sorted: big_list[0] doubles as the word the maximum is gathered in
      RS_HIP(c, hipMemsetAsync(big_list, 0, 2 * sizeof(unsigned), s));
      hipLaunchKernelGGL(rsk::group_max_kernel, dim3(rsk::kRadix), dim3(rsk::kRadix), 0, s, static_cast<const unsigned*>(start), big_list, kGroupCount, ga.max_vec);
      unsigned largest = 0;
      RS_HIP(c, hipMemcpyAsync(&largest, big_list, sizeof(largest), hipMemcpyDeviceToHost, s));
      RS_HIP(c, hipStreamSynchronize(s));
      if (largest > ga.max_vec) {
        c->last_hybrid = false;
        RS_HIP(c, hipGetLastError());
        const int keep = c->hybrid_mode;
        c->hybrid_mode = 0;
        rc = sort_device(c, keys, vals, alt_keys, alt_vals, n, key_bytes, value_bytes, transform, begin_bit, end_bit, ws, true);
        c->hybrid_mode = keep;
        return rc;
      }
      RS_HIP(c, hipMemsetAsync(big_list, 0, 2 * sizeof(unsigned), s));
      if (key_bytes == 4 && largest <= kPair32SmallCap)
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint32_t, uint32_t, kPair32SmallBlock, kPair32Items>), dim3(kGroupCount), dim3(kPair32SmallBlock), 0, s, ga);
      else if (key_bytes == 4)
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint32_t, uint32_t, kPair32Block, kPair32Items>), dim3(kGroupCount), dim3(kPair32Block), 0, s, ga);
      else if (largest <= 256u * kWidePairGroupItems && wide_small_blocks())
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint64_t, uint32_t, 256, kWidePairGroupItems>), dim3(kGroupCount), dim3(256), 0, s, ga);
      else if (largest <= 512u * kWidePairGroupItems && wide_small_blocks())
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint64_t, uint32_t, 512, kWidePairGroupItems>), dim3(kGroupCount), dim3(512), 0, s, ga);
      else
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint64_t, uint32_t, kWideGroupBlock, kWidePairGroupItems>), dim3(kGroupCount), dim3(kWideGroupBlock), 0, s, ga);
      if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
    } else {
      hipLaunchKernelGGL(rsk::group_table_kernel<uint64_t>, dim3(rsk::kRadix), dim3(rsk::kRadix), 0, s, t);
      // groups well below the window of 1024 threads go to workgroups of 512 or 256 threads (8704 / 4352 keys), two or four to a CU
      const double gb = group_bound > 0 ? group_bound : 1.1 * static_cast<double>(n) / static_cast<double>(narrow_bits ? 256u << narrow_bits : kGroupCount);
      const dim3 ggrid(narrow_bits ? 256u << narrow_bits : kGroupCount);
      if (!groups_fit(ggrid.x, kWideGroupCap)) return fit_rc;
      const int wg = prefix_bits || !wide_small_blocks() ? 1024 : largest_group ? (largest_group <= 4352 ? 256 : largest_group <= 8704 ? 512 : 1024) : gb <= 0.9 * 4352 ? 256 : gb <= 0.95 * 8704 ? 512 : 1024;
      const size_t wcap = size_t(wg) * kWideGroupItems;
      ga.max_vec = c->group_max_vec ? static_cast<unsigned>(std::min<size_t>(size_t(c->group_max_vec) * 4u, wcap)) : static_cast<unsigned>(wcap);   // (keys, for this kernel)
      if (wg < 1024) ga.mid_list = mid_list;
      if (wg == 256)
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint64_t, rsk::NoVal, 256, kWideGroupItems>), ggrid, dim3(256), 0, s, ga);
      else if (wg == 512)
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint64_t, rsk::NoVal, 512, kWideGroupItems>), ggrid, dim3(512), 0, s, ga);
      else if (prefix_bits)     // (41 to 47 low bits: passes from bit 24 up, so that the prefix the fix-up works below stays 17 bits or more)
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint64_t, rsk::NoVal, kWideGroupBlock, kWideGroupItems, 24>), dim3(kGroupCount), dim3(kWideGroupBlock), 0, s, ga);
      else
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint64_t, rsk::NoVal, kWideGroupBlock, kWideGroupItems>), ggrid, dim3(kWideGroupBlock), 0, s, ga);
      if (wg < 1024) {     // what the small workgroups could not hold
        rsk::GroupArgs gm = ga;
        gm.mid_list = nullptr; gm.from_list = mid_list;
        gm.max_vec = c->group_max_vec ? static_cast<unsigned>(std::min<size_t>(size_t(c->group_max_vec) * 4u, kWideGroupCap)) : static_cast<unsigned>(kWideGroupCap);
        hipLaunchKernelGGL((rsk::group_sort_wide_kernel<uint64_t, rsk::NoVal, kWideGroupBlock, kWideGroupItems, -1, true>), dim3(static_cast<unsigned>(c->cus)), dim3(kWideGroupBlock), 0, s, gm);
      }
      if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
      hipLaunchKernelGGL((rsk::group_big_kernel<uint64_t, 512, 8>), dim3(static_cast<unsigned>(c->cus) * 2u), dim3(512), 0, s, ba);
    }
    if (c->profiling) RS_HIP(c, hipEventRecord(c->ev[ev++], s));
  }
  RS_HIP(c, hipGetLastError());
  if (capturing && !ws && (rc = clear_async(c, ctrl + kCtrlHist, kCtrlBase - kCtrlHist, s, true)) != RS_OK) return rc;   // (see self_clearing)
  if (n_live & 1) {   // result sits in the alternate buffers
    RS_HIP(c, hipMemcpyAsync(keys, alt_keys, n * key_bytes, hipMemcpyDeviceToDevice, s));
    if (value_bytes) RS_HIP(c, hipMemcpyAsync(vals, alt_vals, n * value_bytes, hipMemcpyDeviceToDevice, s));
  }
  c->ev_count = c->profiling ? ev : 0;
  return RS_OK;
}

// Waits for the stream, then reports (and clears) a fault that any sort queued since the last call left in an error word:
// the word is not part of the per-sort memset and kernels only OR into it, so a fault is neither lost to a later sort
// nor reported twice.
int check_kernel_error(rs_ctx* c) {
  char* ctrl = static_cast<char*>(c->ctrl.p);
  unsigned* f = c->pinned_flags;
  f[0] = f[1] = 0;
  if (ctrl) RS_HIP(c, hipMemcpyAsync(&f[0], ctrl + kCtrlError, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  if (c->static_used) RS_HIP(c, hipMemcpyAsync(&f[1], c->static_scratch + kCtrlError, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  RS_HIP(c, hipStreamSynchronize(c->stream));      // (also the wait when only single-workgroup sorts ran: no error word yet)
  unsigned flag = f[0], sflag = f[1];
  if (flag) RS_HIP(c, hipMemsetAsync(ctrl + kCtrlError, 0, sizeof(flag), c->stream));
  if (sflag) RS_HIP(c, hipMemsetAsync(c->static_scratch + kCtrlError, 0, sizeof(sflag), c->stream));
  flag |= sflag;
  if (flag & 2u)
    return fail(c, RS_ERR_KERNEL, "an output offset fell outside the array: the keys changed while the sort was running "
                                  "(the stream that produces them was not waited for?); the result is not a sort of the input");
  if (flag & 8u)
    return fail(c, RS_ERR_KERNEL, "the group table of a hybrid sort names a range outside the array: the keys changed while the sort was running "
                                  "(the stream that produces them was not waited for?); the result is not a sort of the input");
  if (flag & 4u)
    return fail(c, RS_ERR_KERNEL, "a pass found its tile tickets already drawn (the control block was not clear when it started): nothing was moved");
  if (flag) return fail(c, RS_ERR_KERNEL, "onesweep look-back timed out (descriptor chain broken)");
  return RS_OK;
}

// Host array -> HBM before a sort.  Small arrays: one pageable copy on the sort's stream.  Large ones: the mover's lanes,
// once everything queued on the stream (which may still use the workspace) has drained.
bool staged_transfers() {                  // RSORT_STAGED=0: plain pageable copies on the sort's stream (A/B timing, tests)
  static const bool on = [] { const char* e = std::getenv("RSORT_STAGED"); return !(e && e[0] == '0'); }();
  return on;
}
// Host array -> HBM before a sort; `slot` (0 or 1) names the bounce buffer an array of less than 1 MiB goes through, so
// the two arrays of one call (keys and payload, blob and offsets) use different ones.
//   small arrays: memcpy into lane 0's pinned buffer, then one DMA on the sort's own stream (nothing to wait for);
//   large arrays: the mover's lanes, once everything queued on the stream (which may still use the workspace) has drained.
int to_device(rs_ctx* c, void* d, const void* h, size_t bytes, int slot) {
  if (!staged_transfers()) {
    RS_HIP(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    return RS_OK;
  }
  if (bytes < rst::kStagedMin) {
    rst::Lane& L = c->mover.lanes[0];
    RS_HIP(c, c->mover.prepare(L));
    std::memcpy(L.pinned[slot], h, bytes);
    RS_HIP(c, hipMemcpyAsync(d, L.pinned[slot], bytes, hipMemcpyHostToDevice, c->stream));
    RS_HIP(c, hipEventRecord(c->small_up, c->stream));      // lane 0's bounce buffer is busy until this copy has read it
    c->small_up_pending = true;
    return RS_OK;
  }
  RS_HIP(c, hipStreamSynchronize(c->stream));
  c->small_up_pending = false;
  RS_HIP(c, c->mover.move(c->device, static_cast<char*>(const_cast<void*>(h)), static_cast<char*>(d), bytes, true));
  return RS_OK;
}
// The same for a destination that nothing queued on the stream touches: a large array moves through the lanes WHILE the
// stream's kernels run (rs_sort_rows: the rows travel during the pair sort); a small one is queued on the stream.
int to_device_beside_stream(rs_ctx* c, void* d, const void* h, size_t bytes, int slot) {
  if (!staged_transfers() || bytes < rst::kStagedMin) return to_device(c, d, h, bytes, slot);
  // The lanes fill the same pinned buffers a small array of this call went through (to_device above), and the stream is
  // NOT waited for here: wait for that one copy (rs_sort_rows: keys of less than 1 MiB, then rows of more -- the key
  // DMA could otherwise still be reading the buffer while row bytes are written into it, and the sort would run on those).
  if (c->small_up_pending) {
    RS_HIP(c, hipEventSynchronize(c->small_up));
    c->small_up_pending = false;
  }
  RS_HIP(c, c->mover.move(c->device, static_cast<char*>(const_cast<void*>(h)), static_cast<char*>(d), bytes, true));
  return RS_OK;
}
// what to_host fetched into pinned memory goes to the caller's arrays (the stream has been waited for)
void hand_over(rs_ctx* c) {
  for (int slot = 0; slot < 2; ++slot) {
    if (c->fetched[slot].host) std::memcpy(c->fetched[slot].host, c->mover.lanes[0].pinned[slot], c->fetched[slot].bytes);
    c->fetched[slot] = {nullptr, 0};
  }
}
// HBM -> host array after a sort.  Small arrays are fetched into the pinned buffer on the sort's stream and handed to the
// caller by finish_host_call; large ones are moved here, after the sort has finished and finished well.
int to_host(rs_ctx* c, void* h, const void* d, size_t bytes, int slot) {
  if (!staged_transfers()) {
    RS_HIP(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    return RS_OK;
  }
  if (bytes < rst::kStagedMin) {
    rst::Lane& L = c->mover.lanes[0];
    RS_HIP(c, c->mover.prepare(L));
    RS_HIP(c, hipMemcpyAsync(L.pinned[slot], d, bytes, hipMemcpyDeviceToHost, c->stream));
    c->fetched[slot] = {h, bytes};
    return RS_OK;
  }
  int rc = check_kernel_error(c);
  if (rc != RS_OK) return rc;
  hand_over(c);                            // a small array of the same call sits in the buffers the lanes are about to reuse
  RS_HIP(c, c->mover.move(c->device, static_cast<char*>(h), static_cast<char*>(const_cast<void*>(d)), bytes, false));
  return RS_OK;
}
// Last step of a host-buffer entry point: wait, report a kernel fault, hand over what to_host fetched into pinned memory.
int finish_host_call(rs_ctx* c) {
  int rc = check_kernel_error(c);
  if (rc == RS_OK) hand_over(c);
  c->fetched[0] = c->fetched[1] = {nullptr, 0};
  return rc;
}


// rs_sort_strings / rs_sort_strings32: OFF = type of the offsets AND of the permutation (8 bytes: any blob; 4 bytes: blobs and
// string counts below 2^32, a third less to move over PCIe, which is 9/10 of the call).
template <typename OFF>
int sort_strings_impl(rs_ctx* c, const uint8_t* h_blob, const OFF* h_offsets, size_t n, OFF* h_perm, int stable) {
  (void)stable;   // every pass is stable and level 0 starts from the identity: one code path serves both
  if (!c) return RS_ERR_ARG;
  if (n > 0 && (!h_offsets || !h_perm)) return fail(c, RS_ERR_ARG, "null offsets/perm pointer");
  if (n >= 0xFFFFFFFFull) return fail(c, RS_ERR_ARG, "string sort takes fewer than 2^32 - 1 strings");
  if (n == 0) return RS_OK;
  if (n == 1) { h_perm[0] = 0; return RS_OK; }
  const uint64_t total = h_offsets[n];
  if (total > 0 && !h_blob) return fail(c, RS_ERR_ARG, "null blob pointer");
  HostCall host_call(c);
  c->fetched[0] = c->fetched[1] = {nullptr, 0};   // nothing left over from a call that failed half-way
  RS_HIP(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;

  // carve one device allocation
  auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t m0 = n;
  const size_t blocks0 = (m0 + rss::kScanTile - 1) / rss::kScanTile;
  size_t need = up(total + 16) + up((n + 1) * sizeof(OFF)) + 2 * up(m0 * 8) /*keys, keys_alt (also the widened perm)*/ +
                11 * up(m0 * 4) + up((blocks0 + 2) * 4);
  int rc;
  if ((rc = ensure(c, c->strbuf, need)) != RS_OK) return rc;
  char* p = static_cast<char*>(c->strbuf.p);
  auto take = [&](size_t b) { char* q = p; p += up(b); return q; };
  auto* d_blob = reinterpret_cast<uint8_t*>(take(total + 16));
  auto* d_off = reinterpret_cast<OFF*>(take((n + 1) * sizeof(OFF)));
  auto* keys = reinterpret_cast<uint64_t*>(take(m0 * 8));
  auto* keys_alt = reinterpret_cast<uint64_t*>(take(m0 * 8));
  uint32_t* u[11];
  for (auto& q : u) q = reinterpret_cast<uint32_t*>(take(m0 * 4));
  uint32_t *perm = u[0], *val = u[1], *val_alt = u[2], *seg = u[3], *ids = u[4], *ids_sorted_buf = u[5], *pos = u[6], *pos2 = u[7],
           *head = u[8], *scan_a = u[9], *scan_b = u[10];
  auto* block_sums = reinterpret_cast<uint32_t*>(take((blocks0 + 2) * 4));

  if (total && (rc = to_device(c, d_blob, h_blob, total, 0)) != RS_OK) return rc;
  if ((rc = to_device(c, d_off, h_offsets, (n + 1) * sizeof(OFF), 1)) != RS_OK) return rc;

  auto grid = [](size_t m) { return dim3(static_cast<unsigned>((m + rss::kBlock - 1) / rss::kBlock)); };
  auto scan = [&](const uint32_t* in, size_t m, uint32_t* out, bool inclusive) {
    unsigned blocks = static_cast<unsigned>((m + rss::kScanTile - 1) / rss::kScanTile);
    hipLaunchKernelGGL(rss::scan_block_sums_kernel, dim3(blocks), dim3(rss::kBlock), 0, s, in, static_cast<uint32_t>(m), block_sums);
    hipLaunchKernelGGL(rss::scan_spine_kernel, dim3(1), dim3(rss::kBlock), 0, s, block_sums, blocks);
    if (inclusive) hipLaunchKernelGGL(rss::scan_apply_kernel<true>, dim3(blocks), dim3(rss::kBlock), 0, s, in, static_cast<uint32_t>(m), block_sums, out);
    else hipLaunchKernelGGL(rss::scan_apply_kernel<false>, dim3(blocks), dim3(rss::kBlock), 0, s, in, static_cast<uint32_t>(m), block_sums, out);
    return blocks;
  };

  const bool prof_strings = c->profiling;
  if (prof_strings) RS_HIP(c, hipEventRecord(c->ev[0], s));     // the device part: everything between the upload and the download
  size_t m = n;
  uint32_t* cur_pos = nullptr;      // level 0: position j is slot j
  uint32_t* cur_ids = nullptr;      // level 0: slot j is string j; later: `ids`, the compacted ids of the level's slots
  uint32_t* cur_seg = nullptr;      // level 0: one segment
  size_t max_seg = 0;               // segment numbers of the level are < max_seg
  uint64_t depth = 0;               // bytes of every unresolved string already compared
  const bool prof = c->profiling;
  const bool precheck = c->precheck;
  c->profiling = false;             // the per-kernel event slots belong to single sorts
  c->precheck = false;              // (chunk keys of unordered strings: the order pre-check of each level's sort would only cost its launches)
  auto restore = [&] { c->profiling = prof; c->precheck = precheck; };
  uint32_t level = 0;
  for (;; ++level) {
    const uint32_t mm = static_cast<uint32_t>(m);
    unsigned sbits = 0;
    if (cur_seg)
      while (sbits < 32 && (size_t(1) << sbits) < max_seg) ++sbits;        // bits that hold every segment number of this level
    const unsigned cbytes = std::min<unsigned>(rss::kChunkBytes, (56u - sbits) / 8u);
    hipLaunchKernelGGL(rss::iota_kernel, grid(m), dim3(rss::kBlock), 0, s, val, static_cast<uint32_t*>(nullptr), mm);
    hipLaunchKernelGGL(rss::chunk_keys_kernel<OFF>, grid(m), dim3(rss::kBlock), 0, s, d_blob, d_off, cur_ids, cur_seg, mm, depth, cbytes, sbits, keys);
    if ((rc = sort_device(c, keys, val, keys_alt, val_alt, m, 8, 4, RS_XF_NONE, 0, 64)) != RS_OK) { restore(); return rc; }
    if (c->last_all_equal) {
      // Every unresolved string has the same chunk key (a prefix all of them share): if the chunk is full, nothing about
      // them changes at this level -- same set, same order, same segments -- and the next level starts c bytes further.
      // (Otherwise they all END here, equal: the general path below writes them out in their order and finishes.)
      uint64_t k0 = 0;
      RS_HIP(c, hipMemcpyAsync(&k0, keys, sizeof(k0), hipMemcpyDeviceToHost, s));
      RS_HIP(c, hipStreamSynchronize(s));
      if ((k0 & 0xFF) == cbytes) { depth += cbytes; continue; }
    }
    // the string ids in sorted order: while the ids are the identity (no compaction yet) the sorted payload is already that
    const uint32_t* ids_sorted = val;
    if (cur_ids) { hipLaunchKernelGGL(rss::gather_u32_kernel, grid(m), dim3(rss::kBlock), 0, s, cur_ids, val, mm, ids_sorted_buf); ids_sorted = ids_sorted_buf; }
    // (positions stay where they are: the subset occupies the same ascending slots of the permutation before and after its sort)
    hipLaunchKernelGGL(rss::write_back_kernel, grid(m), dim3(rss::kBlock), 0, s, ids_sorted, cur_pos, mm, perm);
    hipLaunchKernelGGL(rss::heads_open_kernel, grid(m), dim3(rss::kBlock), 0, s, keys, mm, cbytes, head, val_alt /* open */);
    scan(head, m, scan_a, true);                       // segment numbers (1-based)
    unsigned blocks = scan(val_alt, m, scan_b, false); // slots of the unresolved
    uint32_t open_count = 0;
    RS_HIP(c, hipMemcpyAsync(&open_count, block_sums + blocks, 4, hipMemcpyDeviceToHost, s));
    RS_HIP(c, hipStreamSynchronize(s));
    if (open_count == 0) break;
    uint32_t* next_pos = cur_pos == pos ? pos2 : pos;
    // (the compacted ids go to `ids`: what it held, this level's ids before the sort, was last read by the gather above)
    hipLaunchKernelGGL(rss::compact_kernel, grid(m), dim3(rss::kBlock), 0, s, val_alt, scan_b, scan_a, ids_sorted, cur_pos, mm,
                       ids, next_pos, seg);
    cur_pos = next_pos;
    cur_ids = ids;
    cur_seg = seg;
    max_seg = m;            // segment numbers are < m
    m = open_count;
    depth += cbytes;
    if (level > (1u << 28)) { restore(); return fail(c, RS_ERR_KERNEL, "string sort did not converge"); }
  }
  restore();
  c->last_string_levels = static_cast<int>(level) + 1;
  RS_HIP(c, hipGetLastError());
  if constexpr (sizeof(OFF) == 8) {
    hipLaunchKernelGGL(rss::widen_kernel, grid(n), dim3(rss::kBlock), 0, s, perm, static_cast<uint32_t>(n), keys);
    if (prof_strings) { RS_HIP(c, hipEventRecord(c->ev[1], s)); c->ev_count = 2; }   // rs_ctx_get_profile: ms[0] = device time of the string sort
    if ((rc = to_host(c, h_perm, keys, n * 8, 0)) != RS_OK) return rc;
  } else {
    if (prof_strings) { RS_HIP(c, hipEventRecord(c->ev[1], s)); c->ev_count = 2; }
    if ((rc = to_host(c, h_perm, perm, n * 4, 0)) != RS_OK) return rc;
  }
  return finish_host_call(c);
}

}  // namespace

extern "C" {

int rs_version(void) { return 100; }

#ifdef RS_STAMPS
int rs_debug_read_stamps(unsigned long long* dst) {   // diagnostic builds only; not in rsort_hip.h
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(rs_stamp_buf), sizeof(unsigned long long) * 16 * 4096) == hipSuccess ? 0 : 2;
}
#endif

// Test hook, not in rsort_hip.h: ORs `code` into the context's error word on its stream, as a faulting sweep kernel would
// (tests/test_gpu_errors.py checks that the fault is reported once, by the next synchronisation, whatever was queued behind it).
int rs_debug_inject_fault(rs_ctx* c, unsigned code) {
  if (!c) return RS_ERR_ARG;
  RS_HIP(c, hipSetDevice(c->device));
  int rc = ensure(c, c->ctrl, kCtrlBytesChain);
  if (rc != RS_OK) return rc;
  hipLaunchKernelGGL(rsk::inject_fault_kernel, dim3(1), dim3(1), 0, c->stream,
                     reinterpret_cast<unsigned*>(static_cast<char*>(c->ctrl.p) + kCtrlError), code);
  RS_HIP(c, hipGetLastError());
  return RS_OK;
}

int rs_ctx_create(rs_ctx** out, int device) {
  if (!out) return RS_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return RS_ERR_HIP;
  rs_ctx* c = new (std::nothrow) rs_ctx();
  if (!c) return RS_ERR_NOMEM;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return RS_ERR_HIP;
  }
  c->stream = c->own_stream;
  if (const char* e = std::getenv("RSORT_DISTINCT")) c->distinct_path = !(e[0] == '0');
  if (const char* e = std::getenv("RSORT_HYBRID")) c->hybrid_mode = std::atoi(e) < 0 ? -1 : std::atoi(e) > 3 ? 1 : std::atoi(e);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->cus = prop.multiProcessorCount;
  }
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) { rs_ctx_destroy(c); return RS_ERR_HIP; }
  if (hipEventCreateWithFlags(&c->small_up, hipEventDisableTiming) != hipSuccess) { c->small_up = nullptr; rs_ctx_destroy(c); return RS_ERR_HIP; }
  if (hipHostMalloc(reinterpret_cast<void**>(&c->pinned_flags), 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) {
    c->pinned_flags = nullptr;
    rs_ctx_destroy(c);
    return RS_ERR_HIP;
  }
  // this device's copy of the static workspace (the array's name is only a host-side handle: kernels and memsets get
  // the address looked up here, with the device current)
  static_assert(sizeof(rs_static_scratch) >= kNoAllocBytes, "static workspace too small");
  {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(rs_static_scratch)) != hipSuccess || !p) { rs_ctx_destroy(c); return RS_ERR_HIP; }
    c->static_scratch = static_cast<char*>(p);
  }
  // Ranking shortcut: usable only if returning LDS adds of one wave instruction are served in lane order.
  // Probed here on this device with the very functions the kernels rank with (rsk::rank_add / rank_add_peeled /
  // rank_add_runs, on both cell layouts: 16-bit cells in 1024-thread workgroups two to a CU, 32-bit cells in 512-thread
  // workgroups; 1.5e8 lane-operations, eight collision patterns); any violation, or any failure of the probe itself,
  // keeps the ballot-based ranking.
  {
    const char* env = std::getenv("RSORT_RANK");      // "ballot" forces the ballot form (tests, A/B timing)
    unsigned* d_bad = nullptr;
    unsigned h_bad = 1;
    if (!(env && std::strcmp(env, "ballot") == 0) && hipMalloc(&d_bad, sizeof(unsigned)) == hipSuccess) {
      if (hipMemsetAsync(d_bad, 0, sizeof(unsigned), c->stream) == hipSuccess) {
        hipLaunchKernelGGL((rsk::lds_order_probe_kernel<false, 1024>), dim3(512), dim3(1024), 0, c->stream, d_bad, 192);
        hipLaunchKernelGGL((rsk::lds_order_probe_kernel<true, 512>), dim3(512), dim3(512), 0, c->stream, d_bad, 192);
        if (hipMemcpyAsync(&h_bad, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
          h_bad = 1;
      }
      (void)hipFree(d_bad);
    }
    c->rank_by_lds_atomic = h_bad == 0;
  }

  // Tile order: one ticket counter per XCD needs workgroups on every XCD, under the XCC ids the kernel reads.
  // Probed here: 1024 one-wave workgroups count themselves per id; the order is enabled only if all 8 ids appear with
  // at least half their fair share.  RSORT_XCD=0 keeps the single counter (tests, A/B timing).
  {
    const char* env = std::getenv("RSORT_XCD");
    unsigned* d_cnt = nullptr;
    unsigned h_cnt[rsk::kXcds] = {};
    if (!(env && env[0] == '0') && hipMalloc(&d_cnt, sizeof(h_cnt)) == hipSuccess) {
      bool ok = hipMemsetAsync(d_cnt, 0, sizeof(h_cnt), c->stream) == hipSuccess;
      if (ok) {
        hipLaunchKernelGGL(rsk::xcc_probe_kernel, dim3(1024), dim3(64), 0, c->stream, d_cnt);
        ok = hipMemcpyAsync(h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
             hipStreamSynchronize(c->stream) == hipSuccess;
      }
      (void)hipFree(d_cnt);
      for (unsigned x = 0; ok && x < rsk::kXcds; ++x) ok = h_cnt[x] >= 1024 / rsk::kXcds / 2;
      c->per_xcd_order = ok;
      c->per_xcd_probe_ok = ok;
    }
  }
  // Chain mode publishes its look-back descriptors with workgroup-scope stores and reads them from other workgroups of the
  // same XCD (rsk::chain_probe_kernel has the reasoning): probed here with the very store / load pair, 128 links per XCD.
  // Any failure, of a wait or of the probe itself, selects agent-scope stores (correct under the memory model, slower).
  // RSORT_CHAIN_PROBE=fail forces that (tests, A/B timing).
  if (c->per_xcd_probe_ok) {
    const char* env = std::getenv("RSORT_CHAIN_PROBE");
    constexpr unsigned kLinks = 128;
    constexpr size_t kBytes = rsk::kXcds * 32 * sizeof(unsigned) + rsk::kXcds * kLinks * sizeof(uint32_t) + sizeof(unsigned);
    char* d = nullptr;
    unsigned failures = 1;
    if (!(env && std::strcmp(env, "fail") == 0) && hipMalloc(reinterpret_cast<void**>(&d), kBytes) == hipSuccess) {
      if (hipMemsetAsync(d, 0, kBytes, c->stream) == hipSuccess) {
        auto* tickets = reinterpret_cast<unsigned*>(d);
        auto* words = reinterpret_cast<uint32_t*>(d + rsk::kXcds * 32 * sizeof(unsigned));
        auto* fails = reinterpret_cast<unsigned*>(d + kBytes - sizeof(unsigned));
        hipLaunchKernelGGL(rsk::chain_probe_kernel, dim3(rsk::kXcds * kLinks), dim3(64), 0, c->stream, tickets, words, kLinks, fails);
        std::vector<uint32_t> h_words(rsk::kXcds * kLinks);
        if (hipMemcpyAsync(&failures, fails, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(h_words.data(), words, h_words.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
          failures = 1;
        // every link of every chain must have been published (a chain cut short -- an XCD that got fewer workgroups than
        // links -- proves nothing about the links behind the cut: at least half of each chain is asked for)
        for (unsigned x = 0; failures == 0 && x < rsk::kXcds; ++x)
          for (unsigned k = 0; k < kLinks / 2; ++k)
            if (h_words[x * kLinks + k] != (0x40000000u | k)) { failures = 1; break; }
      }
      (void)hipFree(d);
    }
    c->desc_local = failures == 0;
  }
  *out = c;
  return RS_OK;
}

int rs_ctx_destroy(rs_ctx* c) {
  if (!c) return RS_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (DevBuf* b : {&c->alt_keys, &c->alt_vals, &c->desc, &c->ctrl, &c->strbuf, &c->groups, &c->distinct})
    if (b->p) (void)hipFree(b->p);
  for (auto& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->small_up) (void)hipEventDestroy(c->small_up);
  c->mover.release_all();
  if (c->pinned_flags) (void)hipHostFree(c->pinned_flags);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return RS_OK;
}

const char* rs_last_error(const rs_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rs_ctx_set_stream(rs_ctx* c, void* hip_stream) {
  if (!c) return RS_ERR_ARG;
  RS_HIP(c, hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  return RS_OK;
}

int rs_ctx_reserve(rs_ctx* c, size_t n, int key_bytes, int value_bytes) {
  int rc = check_args(c, reinterpret_cast<void*>(1), n, key_bytes, value_bytes, RS_XF_NONE, 0, key_bytes * 8);
  if (rc != RS_OK) return rc;
  RS_HIP(c, hipSetDevice(c->device));
  if ((rc = ensure(c, c->ctrl, kCtrlBytesChain)) != RS_OK) return rc;
  if ((rc = ensure(c, c->desc, desc_bytes_for(n))) != RS_OK) return rc;
  if ((rc = ensure(c, c->alt_keys, n * key_bytes)) != RS_OK) return rc;
  if (value_bytes && (rc = ensure(c, c->alt_vals, n * value_bytes)) != RS_OK) return rc;
  return RS_OK;
}

int rs_ctx_synchronize(rs_ctx* c) {
  if (!c) return RS_ERR_ARG;
  return check_kernel_error(c);   // synchronises, then reports a look-back fault of any queued sort
}

int rs_ctx_set_precheck(rs_ctx* c, int enabled) {
  if (!c) return RS_ERR_ARG;
  c->precheck = enabled != 0;
  return RS_OK;
}

int rs_ctx_set_hybrid(rs_ctx* c, int mode) {
  if (!c) return RS_ERR_ARG;
  if (mode < -1 || mode > 3) return fail(c, RS_ERR_ARG, "hybrid mode must be -1 (per sort), 0 (never), 1 (16-bit groups whenever the shape allows), 2 (narrow form whenever it does) or 3 (prefixed form whenever it does)");
  c->hybrid_mode = mode;
  return RS_OK;
}

// Test hooks, not in rsort_hip.h: the largest group (in 16-byte vectors) the LDS kernel of the hybrid form takes (smaller
// values push groups to the big-group kernel at test sizes), and whether the last device sort took the hybrid form.
int rs_debug_set_group_limit(rs_ctx* c, unsigned max_vec) {
  if (!c) return RS_ERR_ARG;
  c->group_max_vec = max_vec;
  return RS_OK;
}
int rs_debug_last_hybrid(const rs_ctx* c) { return c ? c->last_hybrid : 0; }   // 1: groups of equal top 16 bits, 2: the narrow form
int rs_debug_last_distinct(const rs_ctx* c) { return c && c->last_distinct ? 1 : 0; }
int rs_debug_set_distinct(rs_ctx* c, int on) { if (!c) return RS_ERR_ARG; c->distinct_path = on != 0; return RS_OK; }   // test hook: the few-distinct-keys path on / off
int rs_debug_desc_local(const rs_ctx* c) { return c && c->desc_local ? 1 : 0; }      // what the chain probe of rs_ctx_create found

int rs_ctx_set_tile_order(rs_ctx* c, int per_xcd) {
  if (!c) return RS_ERR_ARG;
  c->per_xcd_order = per_xcd != 0 && c->per_xcd_probe_ok;   // never on a device the probe rejected
  return RS_OK;
}

int rs_ctx_set_host_threads(rs_ctx* c, int threads) {
  if (!c) return RS_ERR_ARG;
  if (threads < 1) return fail(c, RS_ERR_ARG, "host threads must be >= 1");
  c->mover.host_threads = threads;      // the mover uses at most rst::kMaxLanes of them
  return RS_OK;
}

int rs_ctx_set_profiling(rs_ctx* c, int enabled) {
  if (!c) return RS_ERR_ARG;
  c->profiling = enabled != 0;
  c->ev_count = 0;
  return RS_OK;
}

int rs_ctx_get_profile(rs_ctx* c, float* ms, int capacity, int* count) {
  if (!c || !ms || !count) return RS_ERR_ARG;
  *count = 0;
  if (c->ev_count < 2) return RS_OK;
  RS_HIP(c, hipEventSynchronize(c->ev[c->ev_count - 1]));
  int k = std::min(capacity, c->ev_count - 1);
  for (int i = 0; i < k; ++i) RS_HIP(c, hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
  *count = k;
  return RS_OK;
}

int rs_sort_keys_device(rs_ctx* c, void* d_keys, void* d_alt, size_t n, int key_bytes, int transform,
                        int begin_bit, int end_bit) {
  int rc = check_args(c, d_keys, n, key_bytes, 0, transform, begin_bit, end_bit);
  if (rc != RS_OK) return rc;
  RS_HIP(c, hipSetDevice(c->device));
  return sort_device(c, d_keys, nullptr, d_alt, nullptr, n, key_bytes, 0, transform, begin_bit, end_bit);
}

int rs_sort_keys_device_noalloc(rs_ctx* c, void* d_keys, void* d_alt, size_t n, int key_bytes, int transform,
                                int begin_bit, int end_bit) {
  int rc = check_args(c, d_keys, n, key_bytes, 0, transform, begin_bit, end_bit);
  if (rc != RS_OK) return rc;
  if (n > kNoAllocMaxKeys)
    return fail(c, RS_ERR_ARG, "rs_sort_keys_device_noalloc sorts at most " + std::to_string(kNoAllocMaxKeys) +
                               " keys (its workspace is a fixed static device array); use rs_sort_keys_device");
  if (n > 1 && !d_alt) return fail(c, RS_ERR_ARG, "rs_sort_keys_device_noalloc needs the caller's alternate buffer (it allocates nothing)");
  RS_HIP(c, hipSetDevice(c->device));
  const Workspace ws{c->static_scratch, c->static_scratch + kNoAllocCtrlBytes};
  return sort_device(c, d_keys, nullptr, d_alt, nullptr, n, key_bytes, 0, transform, begin_bit, end_bit, &ws);
}

int rs_sort_pairs_device(rs_ctx* c, void* d_keys, void* d_values, void* d_alt_keys, void* d_alt_values,
                         size_t n, int key_bytes, int value_bytes, int transform, int begin_bit, int end_bit) {
  int rc = check_args(c, d_keys, n, key_bytes, value_bytes, transform, begin_bit, end_bit);
  if (rc != RS_OK) return rc;
  if (value_bytes == 0) return fail(c, RS_ERR_ARG, "value_bytes must be 4 or 8");
  if (n > 0 && !d_values) return fail(c, RS_ERR_ARG, "null value pointer");
  RS_HIP(c, hipSetDevice(c->device));
  return sort_device(c, d_keys, d_values, d_alt_keys, d_alt_values, n, key_bytes, value_bytes, transform,
                     begin_bit, end_bit);
}

int rs_partition_device(rs_ctx* c, const void* d_keys_in, void* d_keys_out, const void* d_values_in, void* d_values_out,
                        size_t n, int key_bytes, int value_bytes, int transform, int begin_bit, int end_bit,
                        uint64_t* h_counts) {
  int rc = check_args(c, d_keys_in, n, key_bytes, value_bytes, transform, begin_bit, end_bit);
  if (rc != RS_OK) return rc;
  if (end_bit - begin_bit > rsk::kRadixBits) return fail(c, RS_ERR_ARG, "rs_partition_device moves keys by one digit: end_bit - begin_bit <= 8");
  if (n > 0 && !d_keys_out) return fail(c, RS_ERR_ARG, "null output pointer");
  if (value_bytes && n > 0 && (!d_values_in || !d_values_out)) return fail(c, RS_ERR_ARG, "null value pointer");
  if (h_counts) std::memset(h_counts, 0, sizeof(uint64_t) * 256);
  if (n == 0) return RS_OK;
  RS_HIP(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t tiles = tiles_for(n, key_bytes, value_bytes);
  static const bool force_wide = [] { const char* e = std::getenv("RSORT_FORCE_WIDE"); return e && e[0] == '1'; }();
  const bool wide = n > kSmallDescLimit || force_wide;
  if (tiles > 0x7FFFFFFFull) return fail(c, RS_ERR_ARG, "n too large for one launch");
  if ((rc = ensure(c, c->ctrl, kCtrlBytesChain)) != RS_OK) return rc;
  if ((rc = ensure(c, c->desc, desc_bytes_for(n))) != RS_OK) return rc;
  char* ctrl = static_cast<char*>(c->ctrl.p);
  auto* ghist = reinterpret_cast<unsigned long long*>(ctrl + kCtrlHist);
  auto* gbase = reinterpret_cast<unsigned long long*>(ctrl + kCtrlBase);
  auto* tickets = reinterpret_cast<unsigned*>(ctrl + kCtrlTicket);
  auto* error_flag = reinterpret_cast<unsigned*>(ctrl + kCtrlError);
  const uint64_t sign = 1ull << (key_bytes * 8 - 1);
  const uint64_t all = key_bytes == 8 ? ~0ull : ((1ull << (key_bytes * 8)) - 1ull);
  const uint64_t fmask = transform == RS_XF_IEEE ? all : 0;
  const uint64_t smask = transform == RS_XF_NONE ? 0 : sign;

  RS_HIP(c, hipMemsetAsync(ctrl + kCtrlHist, 0, kCtrlMidDesc - kCtrlHist, s));       // (not the error word: see sort_device)
  c->hist_copy = 1;                                                                   // (both copies cleared, copy 0 used: kCtrlHist1)
  RS_HIP(c, hipMemsetAsync(c->desc.p, 0, tiles * 256 * (wide ? 8 : 4), s));
  rsk::HistArgs h{};
  h.keys = d_keys_in; h.ghist = ghist; h.n = n; h.fmask = fmask; h.smask = smask;
  h.begin_bit = begin_bit; h.end_bit = end_bit; h.passes = 1;
  launch_hist(h, key_bytes, hist_grid(n, key_bytes), s);
  hipLaunchKernelGGL(rsk::digit_scan_kernel, dim3(1), dim3(rsk::kRadix), 0, s, ghist, gbase, static_cast<const unsigned long long*>(nullptr));
  unsigned long long counts[256];
  RS_HIP(c, hipMemcpyAsync(counts, ghist, sizeof(counts), hipMemcpyDeviceToHost, s));
  RS_HIP(c, hipStreamSynchronize(s));
  bool constant = false;
  for (int d = 0; d < 256; ++d) {
    if (h_counts) h_counts[d] = counts[d];
    constant = constant || counts[d] == n;
  }
  if (constant) {            // one digit for all keys: the partition is the identity
    RS_HIP(c, hipMemcpyAsync(d_keys_out, d_keys_in, n * key_bytes, hipMemcpyDeviceToDevice, s));
    if (value_bytes) RS_HIP(c, hipMemcpyAsync(d_values_out, d_values_in, n * value_bytes, hipMemcpyDeviceToDevice, s));
    return RS_OK;
  }
  rsk::SweepArgs a{};
  a.keys_in = d_keys_in; a.keys_out = d_keys_out; a.vals_in = d_values_in; a.vals_out = d_values_out;
  a.desc = c->desc.p; a.ticket = tickets; a.gbase = gbase; a.error_flag = error_flag;
  a.n = n;
  a.fmask_in = fmask; a.smask_in = smask; a.fmask_out = fmask; a.smask_out = smask;   // ordered on the transformed digit, stored as they came
  a.shift = begin_bit; a.bits = end_bit - begin_bit;
  a.per_xcd_order = c->per_xcd_order && tiles >= kPerXcdMinTiles ? 1 : 0;
  a.incl_code = 2u;
  launch_sweep_k(a, SweepLaunch{tiles, key_bytes, value_bytes, wide, c->rank_by_lds_atomic, false, 0, s});
  RS_HIP(c, hipGetLastError());
  return RS_OK;
}

int rs_sort_keys(rs_ctx* c, void* h_keys, size_t n, int key_bytes, int transform, int begin_bit, int end_bit) {
  int rc = check_args(c, h_keys, n, key_bytes, 0, transform, begin_bit, end_bit);
  if (rc != RS_OK) return rc;
  HostCall host_call(c);
  if (n < 2) return RS_OK;
  c->fetched[0] = c->fetched[1] = {nullptr, 0};   // nothing left over from a call that failed half-way
  RS_HIP(c, hipSetDevice(c->device));
  // device layout: [keys | alt] in the workspace's alt_keys buffer
  const size_t bytes = n * key_bytes;
  const size_t padded = (bytes + 255) & ~size_t(255);
  if ((rc = ensure(c, c->alt_keys, 2 * padded)) != RS_OK) return rc;
  char* d = static_cast<char*>(c->alt_keys.p);
  if ((rc = to_device(c, d, h_keys, bytes, 0)) != RS_OK) return rc;
  if ((rc = sort_device(c, d, nullptr, d + padded, nullptr, n, key_bytes, 0, transform, begin_bit, end_bit)) != RS_OK)
    return rc;
  if (!c->last_unchanged && (rc = to_host(c, h_keys, d, bytes, 0)) != RS_OK) return rc;   // in order already: the caller's array is the result
  return finish_host_call(c);
}

int rs_sort_pairs(rs_ctx* c, void* h_keys, void* h_values, size_t n, int key_bytes, int value_bytes, int transform,
                  int begin_bit, int end_bit) {
  int rc = check_args(c, h_keys, n, key_bytes, value_bytes, transform, begin_bit, end_bit);
  if (rc != RS_OK) return rc;
  HostCall host_call(c);
  if (value_bytes == 0) return fail(c, RS_ERR_ARG, "value_bytes must be 4 or 8");
  if (n > 0 && !h_values) return fail(c, RS_ERR_ARG, "null value pointer");
  if (n < 2) return RS_OK;
  c->fetched[0] = c->fetched[1] = {nullptr, 0};   // nothing left over from a call that failed half-way
  RS_HIP(c, hipSetDevice(c->device));
  const size_t kb = n * key_bytes, vb = n * value_bytes;
  const size_t kpad = (kb + 255) & ~size_t(255), vpad = (vb + 255) & ~size_t(255);
  if ((rc = ensure(c, c->alt_keys, 2 * kpad)) != RS_OK) return rc;
  if ((rc = ensure(c, c->alt_vals, 2 * vpad)) != RS_OK) return rc;
  char* dk = static_cast<char*>(c->alt_keys.p);
  char* dv = static_cast<char*>(c->alt_vals.p);
  if ((rc = to_device(c, dk, h_keys, kb, 0)) != RS_OK) return rc;
  if ((rc = to_device(c, dv, h_values, vb, 1)) != RS_OK) return rc;
  if ((rc = sort_device(c, dk, dv, dk + kpad, dv + vpad, n, key_bytes, value_bytes, transform, begin_bit, end_bit)) != RS_OK)
    return rc;
  if (!c->last_unchanged) {                 // keys in order already: a stable sort moved neither column
    if ((rc = to_host(c, h_keys, dk, kb, 0)) != RS_OK) return rc;
    if ((rc = to_host(c, h_values, dv, vb, 1)) != RS_OK) return rc;
  }
  return finish_host_call(c);
}


int rs_sort_rows(rs_ctx* c, void* h_rows, size_t n, int row_bytes, const void* h_keys, int key_bytes, int transform) {
  int rc = check_args(c, h_keys, n, key_bytes, 4, transform, 0, key_bytes * 8);
  if (rc != RS_OK) return rc;
  HostCall host_call(c);
  if (row_bytes < 4 || row_bytes > 256 || (row_bytes & 3)) return fail(c, RS_ERR_ARG, "row_bytes must be a multiple of 4 between 4 and 256");
  if (n > 0 && !h_rows) return fail(c, RS_ERR_ARG, "null row pointer");
  if (n >= 0xFFFFFFFFull) return fail(c, RS_ERR_ARG, "rs_sort_rows takes fewer than 2^32 - 1 rows (4-byte row indices)");
  if (n < 2) return RS_OK;
  c->fetched[0] = c->fetched[1] = {nullptr, 0};
  RS_HIP(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t kb = up(n * key_bytes), ib = up(n * 4), rb = up(n * static_cast<size_t>(row_bytes));
  if ((rc = ensure(c, c->strbuf, 2 * kb + 2 * ib + 2 * rb)) != RS_OK) return rc;
  char* p = static_cast<char*>(c->strbuf.p);
  char* d_keys = p; char* d_keys_alt = p + kb;
  auto* d_idx = reinterpret_cast<uint32_t*>(p + 2 * kb);
  auto* d_idx_alt = reinterpret_cast<uint32_t*>(p + 2 * kb + ib);
  char* d_rows = p + 2 * kb + 2 * ib; char* d_rows_out = d_rows + rb;
  // keys up, the pair sort queued; the rows (the bulk of the bytes) travel while it runs
  if ((rc = to_device(c, d_keys, h_keys, n * key_bytes, 0)) != RS_OK) return rc;
  const unsigned grid = static_cast<unsigned>(std::min<size_t>((n + 255) / 256, size_t(c->cus) * 16));
  hipLaunchKernelGGL(rsk::iota_u32_kernel, dim3(grid), dim3(256), 0, s, d_idx, static_cast<uint64_t>(n));
  if ((rc = sort_device(c, d_keys, d_idx, d_keys_alt, d_idx_alt, n, key_bytes, 4, transform, 0, key_bytes * 8)) != RS_OK) return rc;
  if (c->last_unchanged) return finish_host_call(c);            // keys in order already: a stable sort moves no row
  if ((rc = to_device_beside_stream(c, d_rows, h_rows, n * static_cast<size_t>(row_bytes), 1)) != RS_OK) return rc;
  const unsigned words = static_cast<unsigned>(row_bytes / 4);
  const unsigned ggrid = static_cast<unsigned>(std::min<size_t>((n * words + 255) / 256, size_t(c->cus) * 32));
  hipLaunchKernelGGL(rsk::gather_rows_kernel, dim3(ggrid), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(d_rows), d_idx,
                     reinterpret_cast<uint32_t*>(d_rows_out), static_cast<uint64_t>(n), words);
  RS_HIP(c, hipGetLastError());
  if ((rc = to_host(c, h_rows, d_rows_out, n * static_cast<size_t>(row_bytes), 0)) != RS_OK) return rc;
  return finish_host_call(c);
}

int rs_digit_histogram_device(rs_ctx* c, const void* d_keys, size_t n, int key_bytes, int transform, int begin_bit,
                              int end_bit, uint64_t* h_counts) {
  int rc = check_args(c, d_keys, n, key_bytes, 0, transform, begin_bit, end_bit);
  if (rc != RS_OK) return rc;
  if (!h_counts) return fail(c, RS_ERR_ARG, "null output pointer");
  const int passes = (end_bit - begin_bit + rsk::kRadixBits - 1) / rsk::kRadixBits;
  std::memset(h_counts, 0, sizeof(uint64_t) * 256 * passes);
  if (n == 0) return RS_OK;
  RS_HIP(c, hipSetDevice(c->device));
  if ((rc = ensure(c, c->ctrl, kCtrlBytesChain)) != RS_OK) return rc;
  hipStream_t s = c->stream;
  char* ctrl = static_cast<char*>(c->ctrl.p);
  auto* ghist = reinterpret_cast<unsigned long long*>(ctrl + kCtrlHist);
  RS_HIP(c, hipMemsetAsync(ghist, 0, kCtrlBase - kCtrlHist, s));       // both copies of the counts; copy 0 is used (kCtrlHist1)
  c->hist_copy = 1;
  const uint64_t sign = 1ull << (key_bytes * 8 - 1);
  const uint64_t all = key_bytes == 8 ? ~0ull : ((1ull << (key_bytes * 8)) - 1ull);
  rsk::HistArgs h{};
  h.keys = d_keys; h.ghist = ghist; h.n = n;
  h.fmask = transform == RS_XF_IEEE ? all : 0;
  h.smask = transform == RS_XF_NONE ? 0 : sign;
  h.begin_bit = begin_bit; h.end_bit = end_bit; h.passes = passes;
  launch_hist(h, key_bytes, hist_grid(n, key_bytes), s);
  RS_HIP(c, hipGetLastError());
  RS_HIP(c, hipMemcpyAsync(h_counts, ghist, sizeof(uint64_t) * 256 * passes, hipMemcpyDeviceToHost, s));
  RS_HIP(c, hipStreamSynchronize(s));
  return RS_OK;
}

int rs_sort_strings(rs_ctx* c, const uint8_t* h_blob, const uint64_t* h_offsets, size_t n, uint64_t* h_perm, int stable) {
  return sort_strings_impl<uint64_t>(c, h_blob, h_offsets, n, h_perm, stable);
}

int rs_sort_strings32(rs_ctx* c, const uint8_t* h_blob, const uint32_t* h_offsets, size_t n, uint32_t* h_perm, int stable) {
  return sort_strings_impl<uint32_t>(c, h_blob, h_offsets, n, h_perm, stable);
}

int rs_debug_last_string_levels(const rs_ctx* c) { return c ? c->last_string_levels : 0; }   // test / bench hook: MSD levels of the last string sort

}  // extern "C"
