/* rsort_hip.h -- C-ABI of librsort_hip.so, the MI355X (gfx950) sort engine.
 *
 * The reference (Devil11Assassin/radix-sort) is a header-only C++20 template and has NO FFI; its
 * only boundary is
 *     template <typename T, typename Proj = std::identity>
 *     void radix_sort::sort(std::vector<T>& v, Proj proj = {}, bool enableMultiThreading = false);
 *                                                              (RadixSort/radix_sort.hpp:1628-1649)
 * include/radix_sort.hpp in this repository keeps that signature and lowers every instantiation to
 * the entry points below; each entry point names the reference code whose RESULT it reproduces.
 * Plain pointers and sizes only; no C++ or torch types cross this boundary; no exceptions; the
 * caller owns every buffer it passes.
 *
 * Status codes: 0 = ok; anything else = failure, rs_last_error(ctx) describes it.  There is no CPU
 * fallback behind any of these calls: without a usable GPU they fail.
 */
#ifndef RSORT_HIP_H
#define RSORT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rs_ctx rs_ctx;

enum rs_status {
  RS_OK = 0,
  RS_ERR_ARG = 1,      /* bad argument (width, transform, bit range, null pointer) */
  RS_ERR_HIP = 2,      /* a HIP runtime call failed */
  RS_ERR_KERNEL = 3,   /* a kernel reported an internal fault (look-back timeout) */
  RS_ERR_NOMEM = 4     /* workspace allocation failed */
};

/* Key transform: how the raw bits of a key map to an order-preserving unsigned digit string.
 *   RS_XF_NONE   unsigned integers, bool, char8: bits as they are
 *                (radix_sort.hpp:133-137 count, :450-454 scatter)
 *   RS_XF_SIGNED signed integers: top digit ^ 0x80  (radix_sort.hpp:138-152, :455-469)
 *   RS_XF_IEEE   float/double: u>>63 ? ~u : u^SIGN on load, inverse on store
 *                (radix_sort.hpp:394-416 getUnsignedVectorThread, :724-727)
 * The transform is fused into the first read and the last write; no extra pass touches HBM. */
enum rs_transform { RS_XF_NONE = 0, RS_XF_SIGNED = 1, RS_XF_IEEE = 2 };

/* Context: one per host thread and device; owns a HIP stream (unless one is attached) and the
 * device workspace (alternate key/value buffers, digit histograms, look-back descriptors), which
 * grows on demand and is reused across calls.  Calls on one context are serialised on its stream.
 * Replaces: the per-call std::vector temporaries of the reference (radix_sort.hpp:510, :646). */
int rs_ctx_create(rs_ctx** out, int device);
int rs_ctx_destroy(rs_ctx* ctx);
const char* rs_last_error(const rs_ctx* ctx);

/* Attach a caller-owned hipStream_t (passed as void*; NULL = the context's own stream -- HIP's default stream also has
 * handle NULL, so it cannot be attached: wait for work queued there before sorting data it produces).  Keys must not
 * change while a sort runs on them; if they do the call reports RS_ERR_KERNEL at the next synchronisation. */
int rs_ctx_set_stream(rs_ctx* ctx, void* hip_stream);
/* Pre-size the workspace for n keys of key_bytes with value_bytes payload (0 = keys only), so that
 * later sorts up to that size do not allocate. */
int rs_ctx_reserve(rs_ctx* ctx, size_t n, int key_bytes, int value_bytes);
/* Block until everything queued on the context's stream has finished; returns RS_ERR_KERNEL if a
 * device-resident sort queued since the last call reported an internal fault. */
int rs_ctx_synchronize(rs_ctx* ctx);

/* Order pre-check (default on): before sorting, one early-exit pass tests whether the keys are already
 * ascending on the sorted field (then the call returns: isSortedBi radix_sort.hpp:296-297, key::isSorted
 * :746-769) or -- whole key, no payload -- descending (then they are reversed in place: :298-302).
 * Costs microseconds on unordered input (plus one host synchronisation).  bench.py times the sort
 * engine with the check both on and off. */
int rs_ctx_set_precheck(rs_ctx* ctx, int enabled);

/* Tile order of the sort pass (default: per XCD when the device probe at rs_ctx_create allows it).  MI355X deals
 * workgroups to 8 XCDs with private L2s; handing each XCD groups of 8 consecutive tiles lets the partial cache lines at
 * the boundaries of the runs that consecutive tiles write meet in one L2 (7 % per pass at N = 2^30).  With that order a
 * tile may wait for a tile whose workgroup is not launched yet; a sort running alone always gets it launched, and sorts
 * of fewer than 1024 tiles use the single-counter order anyway.  Callers that run several LARGE sorts on one device at
 * the same time (several contexts or processes) should pass 0: waiting workgroups of one sort could otherwise hold
 * the slots another sort needs (the wait is bounded: such a sort would fail with RS_ERR_KERNEL after 2 s, not hang).
 * No counterpart in the reference (a CPU library). */
int rs_ctx_set_tile_order(rs_ctx* ctx, int per_xcd);

/* Hybrid form of the sort of whole 4-byte keys without payload (rs_sort_keys*, key_bytes 4, bits 0..32).  The reference's
 * LSD loop (sortLsd radix_sort.hpp:506-526) moves every key once per digit; the hybrid form runs the two passes on the top
 * digits only (count :125-153, prefix :193-208, scatter :447-470), after which the array is grouped by its top 16 bits,
 * and finishes every group's low 16 bits inside LDS, one workgroup per group: 28 bytes of HBM traffic per key instead of 36.
 * Same result, bit for bit (equal keys are identical, so no order among them exists to keep).
 * 8-byte keys take the same form (56 instead of 136 bytes per key; with a 4-byte payload, stable: 80 instead of 200; 4-byte
 * keys with a 4-byte payload: 52 instead of 68), and between about 2*10^6 and 1.2*10^8 keys a narrow one: sweeps on the top 9
 * to 15 bits, whole keys of every group finished in LDS.
 *   mode -1 (default): chosen per sort, from its size and its digit counts (groups must fit the LDS window);
 *   mode  0: never;   mode 1: groups of equal top 16 bits whenever the key shape allows it, at any size and distribution;
 *   mode  2: the narrow form whenever the key shape allows it;   mode 3: the prefixed form (keys that share their leading
 *            1 to 7 bits: sweeps on the 16 bits below the prefix) whenever it does (1, 2 and 3: tests, A/B timing).
 * The environment variable RSORT_HYBRID gives the default of new contexts.  No counterpart in the reference. */
int rs_ctx_set_hybrid(rs_ctx* ctx, int mode);

/* Host threads the entry points that take HOST buffers (rs_sort_keys, rs_sort_pairs, rs_sort_strings) may use to move
 * the arrays to the GPU and back (default 1 = the calling thread only; at most 4 are used).  Arrays of 1 MiB and more
 * travel through pinned bounce buffers, one pipeline per thread; the front-end passes the reference's own thread rule
 * here (getNumOfThreads radix_sort.hpp:112-123: 1 unless the caller asked for multi-threading and n >= 2,000,000). */
int rs_ctx_set_host_threads(rs_ctx* ctx, int threads);

/* Per-kernel timing with HIP events on the context's stream (for bench.py's roofline block).
 * After a sort with profiling on, rs_ctx_get_profile fills ms[0] = histogram+scan,
 * ms[1..passes] = the onesweep passes in order; *count = 1 + passes.  Synchronises the stream. */
int rs_ctx_set_profiling(rs_ctx* ctx, int enabled);
int rs_ctx_get_profile(rs_ctx* ctx, float* ms, int capacity, int* count);

/* ---- device-resident entry points (pointers are device memory on the context's device) ---------
 *
 * rs_sort_keys_device: ascending sort of n keys of key_bytes (1, 2, 4, 8) in place.
 *   Result == radix_sort::sort(v, {}, mt) for the matching T (value path,
 *   radix_sort.hpp:698-729): LSD over the digits [begin_bit, end_bit) in 8-bit steps
 *   (sortLsd :506-526; count :125-153, prefix :193-208, scatter :447-470), transform fused.
 *   begin_bit = 0, end_bit = 8*key_bytes sorts on the whole key.
 *   d_alt: optional caller scratch of n*key_bytes bytes (NULL = workspace). */
int rs_sort_keys_device(rs_ctx* ctx, void* d_keys, void* d_alt, size_t n, int key_bytes,
                        int transform, int begin_bit, int end_bit);

/* rs_sort_keys_device_noalloc: rs_sort_keys_device for callers that cannot allocate device memory -- code under HIP
 *   graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference (a CPU library
 *   whose every call allocates its temporaries, radix_sort.hpp:510, :646).  Same result as rs_sort_keys_device, bit for bit.
 *   It calls no hipMalloc / hipFree and no allocating library routine, and it does not wait for the device: memsets and
 *   kernels are queued on the context's stream and the call returns.  Consequently: d_alt (n*key_bytes bytes) is
 *   required; there is no order pre-check and no pass skipping (both need counts on the host); n <= RS_NOALLOC_MAX_KEYS,
 *   larger n is rejected with RS_ERR_ARG.  Histograms, tickets and look-back descriptors live in a static device array of
 *   4.5 MiB that every GPU the library runs on holds until the process ends.  All calls on one GPU share that array:
 *   calls that overlap on one GPU (two streams, two contexts, two threads) race and must be serialised by the caller;
 *   each call clears what it needs at its start (on a stream that is being captured into a graph: with a kernel of the
 *   library rather than memset nodes, which did not reliably clear again on replay).  A fault is reported by the next
 *   rs_ctx_synchronize, as for the other device entry points. */
#define RS_NOALLOC_MAX_KEYS ((size_t)1 << 25)
int rs_sort_keys_device_noalloc(rs_ctx* ctx, void* d_keys, void* d_alt, size_t n, int key_bytes,
                                int transform, int begin_bit, int end_bit);

/* rs_sort_pairs_device: STABLE ascending sort of (key, value) pairs in place; value_bytes 4 or 8.
 *   Result == the key path of the reference (radix_sort.hpp:1160-1206 pair sortLsd,
 *   :1062-1109 pair scatter): with values = 0..n-1 the value column is the permutation that
 *   sortByIndices (:1111-1154) applies. */
int rs_sort_pairs_device(rs_ctx* ctx, void* d_keys, void* d_values, void* d_alt_keys,
                         void* d_alt_values, size_t n, int key_bytes, int value_bytes,
                         int transform, int begin_bit, int end_bit);

/* rs_digit_histogram_device: counts of the 8-bit digits of the (transformed) keys over
 *   [begin_bit, end_bit): h_counts[p*256 + d] for pass p = digit at begin_bit + 8p.  Synchronises.
 *   Result == getCountVector (radix_sort.hpp:155-191) for each digit position.  Used by the
 *   multi-GPU splitter, where the all-reduced top-digit histogram stands in for the reference's
 *   partitionRegions fan-out (:210-254). */
int rs_digit_histogram_device(rs_ctx* ctx, const void* d_keys, size_t n, int key_bytes, int transform,
                              int begin_bit, int end_bit, uint64_t* h_counts);

/* rs_partition_device: ONE stable pass, out of place: d_keys_out = d_keys_in reordered by the digit at
 *   [begin_bit, end_bit) (at most 8 bits) of the transformed key, raw key bits unchanged; values (value_bytes 4 or 8, or
 *   0 with null pointers) move with their keys; h_counts[256] (optional) receives the digit counts.  Synchronises
 *   (the counts come back before the pass is launched); the pass itself is only queued.
 *   Result == one iteration of the LSD loop (count :125-153, prefix :193-208, scatter :447-470 / :1062-1109) = the
 *   bucket layout partitionRegions (:210-254) produces for one MSD level.  The multi-GPU splitter uses it to make every
 *   destination's keys contiguous without the copy back that an in-place call needs after an odd number of passes. */
int rs_partition_device(rs_ctx* ctx, const void* d_keys_in, void* d_keys_out, const void* d_values_in,
                        void* d_values_out, size_t n, int key_bytes, int value_bytes, int transform,
                        int begin_bit, int end_bit, uint64_t* h_counts);

/* ---- host-buffer entry points (what include/radix_sort.hpp calls) --------------------------------
 * Same contracts; buffers are host memory; H2D, sort, D2H run on the context's stream and the call
 * returns after the data is back. */
int rs_sort_keys(rs_ctx* ctx, void* h_keys, size_t n, int key_bytes, int transform,
                 int begin_bit, int end_bit);
int rs_sort_pairs(rs_ctx* ctx, void* h_keys, void* h_values, size_t n, int key_bytes,
                  int value_bytes, int transform, int begin_bit, int end_bit);

/* rs_sort_rows: STABLE sort of n fixed-size rows (row_bytes: a multiple of 4, at most 256) by a separate key column;
 *   h_rows is reordered in place, h_keys is only read.  Result == the projected path of the reference for trivially
 *   copyable objects: key copy (selectProjStrategy radix_sort.hpp:1581-1585), pair sort of (key, index) (:1160-1206,
 *   :1062-1109), then sortByIndices (:1111-1154) -- whose random gather runs on the device here: the keys go up first, the
 *   rows follow while the pair sort runs, one kernel moves row perm[i] to place i, and only the rows come back.
 *   n < 2^32 - 1. */
int rs_sort_rows(rs_ctx* ctx, void* h_rows, size_t n, int row_bytes, const void* h_keys, int key_bytes, int transform);

/* rs_sort_strings: order of n byte strings; string i is h_blob[h_offsets[i] .. h_offsets[i+1]).
 *   On return h_perm[i] = index of the string that belongs at position i; ascending by unsigned
 *   byte, a proper prefix first (std::string::operator<), ties in input order.
 *   Result == radix_sort::sort on std::string keys: the value path (radix_sort.hpp:528-586 string
 *   branch, getChar :107-110, bucket 256 first :196-200, std::sort from depth 3 :559-569) for
 *   stable = 0, the projected path (:1208-1272, std::stable_sort from depth 3 :1241-1255) for
 *   stable = 1 -- the permutation that sortByIndices (:1111-1154) applies.
 *   n < 2^32 - 1.  Runs MSD over chunks of up to 7 bytes, one stable pair sort per level with the segment id in the key's
 *   top bits (radix-sort_amd/csrc/rsort_strings.hpp). */
int rs_sort_strings(rs_ctx* ctx, const uint8_t* h_blob, const uint64_t* h_offsets, size_t n,
                    uint64_t* h_perm, int stable);
/* rs_sort_strings32: the same with 4-byte offsets and a 4-byte permutation, for blobs below 4 GiB (h_offsets[n] < 2^32):
 *   18 instead of 26 bytes per string cross PCIe (blob + offsets up, permutation down), which is nine tenths of the call. */
int rs_sort_strings32(rs_ctx* ctx, const uint8_t* h_blob, const uint32_t* h_offsets, size_t n,
                      uint32_t* h_perm, int stable);

/* Library/ABI version: major*10000 + minor*100 + patch. */
int rs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RSORT_HIP_H */
