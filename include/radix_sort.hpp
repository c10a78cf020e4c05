// radix_sort.hpp -- drop-in front-end for the reference's public API, executed on MI355X.
//
//     template <typename T, typename Proj = std::identity>
//     void radix_sort::sort(std::vector<T>& v, Proj proj = {}, bool enableMultiThreading = false);
//
// Same name, signature, accepted types and compile-time diagnostics as the reference
// (RadixSort/radix_sort.hpp:1628-1649; concepts :65-92); same results: ascending order for
// integers, IEEE-754 totalOrder for float/double, std::string::operator< order for strings, and
// the unique STABLE permutation for projected keys (README.md:62-67).  Every instantiation lowers
// to the C-ABI of librsort_hip.so (include/rsort_hip.h); there is no CPU sort in this header --
// without the library or a GPU the call throws std::runtime_error.
//
// How each reference strategy maps (radix_sort.hpp line numbers):
//   value path, numeric T   (:698-729)    -> rs_sort_keys, transform fused on the device
//   value path, std::string (:528-586)    -> pack to blob+offsets, rs_sort_strings, move by permutation
//   key path, numeric key   (:1543-1592)  -> key column + index payload, rs_sort_pairs (stable),
//                                            then the moves of sortByIndices (:1111-1133); trivially copyable
//                                            objects go to the GPU with their keys and are gathered there (rs_sort_rows)
//   key path, string key    (:1574-1578)  -> pack projected strings, rs_sort_strings (stable), moves
// The third argument keeps its meaning for the steps that stay on the host (key column, string packing, moving the
// objects: threaded by the reference's own rule when it is true); the device path is always parallel.
// The reference's pre-checks (already sorted -> return, descending -> reverse, :269-305 / :746-769) are reproduced: on the
// device for numeric keys (rs_ctx_set_precheck), on the host for strings and projected keys (below).  Its insertion-sort
// thresholds (:42-43) change its running time, never its result; the device's counterpart is a single-workgroup sort.
//
// Link with:  -L<repo>/radix-sort_amd/lib -lrsort_hip
#pragma once
#include <algorithm>
#include <concepts>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "rsort_hip.h"

namespace radix_sort
{
	namespace internal
	{
		namespace shared
		{
			template <typename T>
			concept is_string = std::same_as<T, std::string>;

			template <typename T>
			concept supported = std::integral<T> || std::floating_point<T> || is_string<T>;

			template <typename T, typename Proj>
			using invoke_result = std::invoke_result_t<Proj&, const T&>;

			template <typename T, typename Proj>
			using sort_key = std::remove_cvref_t<std::invoke_result_t<Proj&, const T&>>;

			template <typename T, typename Proj>
			concept sortable = requires(Proj proj, const T& t) {
				{ std::invoke(proj, t) } -> std::same_as<invoke_result<T, Proj>>;
			} && supported<sort_key<T, Proj>>;
		}

		namespace device
		{
			struct CtxDeleter { void operator()(rs_ctx* c) const { rs_ctx_destroy(c); } };

			// One context per host thread (rsort_hip.h: calls on a context are serialised on its stream).
			inline rs_ctx* context()
			{
				thread_local std::unique_ptr<rs_ctx, CtxDeleter> ctx;
				if (!ctx)
				{
					const char* dev = std::getenv("RSORT_DEVICE");
					rs_ctx* raw = nullptr;
					if (rs_ctx_create(&raw, dev ? std::atoi(dev) : 0) != RS_OK)
						throw std::runtime_error("radix_sort::sort: no usable GPU (rs_ctx_create failed)");
					ctx.reset(raw);
				}
				return ctx.get();
			}

			inline void check(rs_ctx* ctx, int status)
			{
				if (status != RS_OK)
					throw std::runtime_error(std::string("radix_sort::sort: ") + rs_last_error(ctx));
			}

			template <typename Key>
			constexpr int transformOf()
			{
				if constexpr (std::floating_point<Key>) return RS_XF_IEEE;
				else if constexpr (std::signed_integral<Key>) return RS_XF_SIGNED;
				else return RS_XF_NONE;
			}

			template <typename Key>
			constexpr void checkKeyType()
			{
				static_assert(sizeof(Key) == 1 || sizeof(Key) == 2 || sizeof(Key) == 4 || sizeof(Key) == 8,
					"ERROR: Unable to sort vector! CAUSE: Unsupported key width!");   // t2u has no other case (:94-98)
				if constexpr (std::floating_point<Key>)
					static_assert(std::numeric_limits<Key>::is_iec559,
						"ERROR: generators.hpp requires IEEE-754/IEC-559 compliance.\n");   // :701-707
			}

			// key::isSorted (:746-769): ascending on the projected key (IEEE keys by total order), left at the first descent.
			// An ordered input is returned as it came -- no key column, no transfer, no gather.
			template <typename T, typename Proj>
			inline bool isAscending(const std::vector<T>& v, Proj& proj)
			{
				using Key = shared::sort_key<T, Proj>;
				auto after = [](const Key& a, const Key& b) {
					if constexpr (std::floating_point<Key>)
					{
						using U = std::conditional_t<sizeof(Key) == 4, std::uint32_t, std::uint64_t>;
						constexpr U TOP = U(1) << (sizeof(U) * 8 - 1);
						auto rank = [](Key x) { U u; std::memcpy(&u, &x, sizeof(U)); return (u & TOP) ? U(~u) : U(u | TOP); };
						return rank(a) > rank(b);
					}
					else
						return a > b;
				};
				for (std::size_t i = 0, last = v.size() - 1; i < last; i++)
					if (after(std::invoke(proj, v[i]), std::invoke(proj, v[i + 1])))
						return false;
				return true;
			}

			// Host threads.  Two users: the library's transfer lanes (rs_ctx_set_host_threads) and the steps that stay on the
			// CPU here (key column, string packing, moving the objects).  Both only when the caller asked for multi-threading,
			// as sortByIndices does (:1121-1153), and both capped like the reference's getNumOfThreads (:112-123): at most 12,
			// at most what the hardware has.  Not the reference's "largest power of two <= n / 1,000,000" below the cap: its
			// threads each own a power-of-two share of the buckets, these only cut [0, n) into ranges, and what they do waits on
			// memory (10^7 {string; long long} objects: 82 ms end to end with 8 threads, 61 ms with 12).
			//   transfers: from 16 MiB on (hostThreads below; strings: from 2,000,000 strings, the reference's element count, at 8 bytes each);
			//   packing / gathering strings and objects: from 250,000 on -- starting a dozen threads costs well under a
			//   millisecond, one thread needs 13-20 ms for 10^6 strings or objects (STRING 10^6: 19.5 -> 7.8 ms).
			inline std::size_t hostThreadsFrom(std::size_t n, bool enableMultiThreading, std::size_t threshold)
			{
				if (!enableMultiThreading || n < threshold)
					return 1;
				const std::size_t hw = std::max<std::size_t>(1, std::thread::hardware_concurrency());
				return std::min(hw, std::size_t(12));
			}
			// transfer lanes, by the BYTES the largest array of the call moves: the reference's 2,000,000-element threshold is
			// 8 MB of its headline type; several lanes pay from about 16 MiB on (10^7 one-byte keys: 1.07 ms over one lane,
			// 3.0 ms over four; 10^7 two-byte keys: 1.85 against 1.52 ms; 10^7 four-byte keys: 3.2 against 2.4 ms)
			inline std::size_t hostThreads(std::size_t n, bool enableMultiThreading, std::size_t bytesPerElement = 8)
			{
				return hostThreadsFrom(n * bytesPerElement, enableMultiThreading, std::size_t(16) << 20);
			}
			inline std::size_t helperThreads(std::size_t n, bool enableMultiThreading)     // packing, gathering
			{
				return hostThreadsFrom(n, enableMultiThreading, 250'000);
			}

			// body(lo, hi) over [0, n) cut into `threads` contiguous ranges (the last one takes the remainder)
			template <typename Body>
			inline void forRanges(std::size_t n, std::size_t threads, Body&& body)
			{
				if (threads <= 1)
				{
					body(std::size_t(0), n, std::size_t(0));
					return;
				}
				std::vector<std::thread> pool;
				pool.reserve(threads);
				const std::size_t step = n / threads;
				for (std::size_t t = 0; t < threads; t++)
					pool.emplace_back([&body, t, step, n, threads] { body(t * step, t + 1 == threads ? n : (t + 1) * step, t); });
				for (auto& th : pool)
					th.join();
			}

			// n elements of a trivial type, NOT value-initialised.  std::vector<T>(n) writes n zeros on the calling thread before
			// the worker threads (or the download) overwrite every one of them: for 10^7 strings, offsets + permutation + blob are
			// 200 MB of zeros written and paged in by one thread.  Here a page is first touched by whoever fills it.
			template <typename T>
			requires std::is_trivially_default_constructible_v<T> && std::is_trivially_destructible_v<T>
			class Uninitialized
			{
				std::unique_ptr<T[]> mem;
				std::size_t count;
			public:
				explicit Uninitialized(std::size_t n) : mem(new T[n]), count(n) {}
				T* data() noexcept { return mem.get(); }
				const T* data() const noexcept { return mem.get(); }
				std::size_t size() const noexcept { return count; }
				T& operator[](std::size_t i) noexcept { return mem[i]; }
				const T& operator[](std::size_t i) const noexcept { return mem[i]; }
			};

			// the cache line(s) of an object that is about to be moved from (read, then left in its moved-from state)
			template <typename T>
			inline void prefetchForMove(const T* obj) noexcept
			{
				const char* p = reinterpret_cast<const char*>(obj);
				__builtin_prefetch(p, 1, 0);
				if constexpr (sizeof(T) > 8)
					__builtin_prefetch(p + sizeof(T) - 1, 1, 0);      // an object that straddles a line boundary
			}

			// tmp[i] = move(v[perm[i]]), then the objects go back: sortByIndices (:1111-1154), threaded like it when asked to.
			// The threaded form gathers into raw storage (move-construction, first touched by the thread that fills it) and
			// moves the objects back in order, instead of value-initialising a second vector on one thread first: at 10^7
			// 56-byte objects that single-threaded construction alone cost more than the gather.
			template <typename T, typename I>
			inline void applyPermutation(std::vector<T>& v, const I* perm, std::size_t threads)
			{
				if constexpr (std::is_nothrow_move_constructible_v<T> && std::is_nothrow_move_assignable_v<T>)
				{
					if (threads > 1)
					{
						std::allocator<T> alloc;
						const std::size_t n = v.size();
						T* raw = alloc.allocate(n);
						// (the sources are scattered over the whole vector: one cache miss per object unless it is asked for ahead
						// of time -- perm is read in order, so the object `kAhead` steps on is known now)
						constexpr std::size_t kAhead = 24;
						forRanges(n, threads, [&](std::size_t lo, std::size_t hi, std::size_t) {
							for (std::size_t i = lo; i < hi; i++)
							{
								if (i + kAhead < hi)
									prefetchForMove(&v[static_cast<std::size_t>(perm[i + kAhead])]);
								std::construct_at(raw + i, std::move(v[static_cast<std::size_t>(perm[i])]));
							}
						});
						forRanges(n, threads, [&](std::size_t lo, std::size_t hi, std::size_t) {
							for (std::size_t i = lo; i < hi; i++)
							{
								v[i] = std::move(raw[i]);
								std::destroy_at(raw + i);
							}
						});
						alloc.deallocate(raw, n);
						return;
					}
				}
				std::vector<T> tmp;
				tmp.reserve(v.size());
				for (std::size_t i = 0, n = v.size(); i < n; i++)
				{
					if (i + 24 < n)
						prefetchForMove(&v[static_cast<std::size_t>(perm[i + 24])]);
					tmp.emplace_back(std::move(v[static_cast<std::size_t>(perm[i])]));
				}
				std::swap(tmp, v);
			}

			// OFF = std::uint32_t when the bytes of all strings stay below 4 GiB (rs_sort_strings32: 18 instead of 26 bytes per
			// string cross PCIe), std::uint64_t otherwise
			template <typename OFF>
			struct PackedStrings
			{
				Uninitialized<std::uint8_t> blob;
				Uninitialized<OFF> offsets;
			};

			// bytes of the strings of each of `threads` ranges of v, as running totals: [0] = 0, [threads] = all of them
			template <typename T, typename Proj>
			inline std::vector<std::uint64_t> stringRangeBytes(const std::vector<T>& v, Proj& proj, std::size_t threads)
			{
				std::vector<std::uint64_t> rangeBytes(std::max<std::size_t>(threads, 1) + 1, 0);
				forRanges(v.size(), threads, [&](std::size_t lo, std::size_t hi, std::size_t t) {
					std::uint64_t total = 0;
					for (std::size_t i = lo; i < hi; i++)
						total += std::invoke(proj, v[i]).size();
					rangeBytes[t + 1] = total;
				});
				for (std::size_t t = 1; t < rangeBytes.size(); t++)
					rangeBytes[t] += rangeBytes[t - 1];
				return rangeBytes;
			}

			template <typename OFF, typename T, typename Proj>
			inline PackedStrings<OFF> packStrings(const std::vector<T>& v, Proj& proj, std::size_t threads, const std::vector<std::uint64_t>& rangeBytes)
			{
				const std::size_t n = v.size();
				PackedStrings<OFF> p{Uninitialized<std::uint8_t>(rangeBytes.back()), Uninitialized<OFF>(n + 1)};
				p.offsets[n] = static_cast<OFF>(rangeBytes.back());
				forRanges(n, threads, [&](std::size_t lo, std::size_t hi, std::size_t t) {
					std::uint64_t at = rangeBytes[t];
					constexpr std::size_t kAhead = 16;     // the bytes of a string past the small-string size live somewhere on the heap
					for (std::size_t i = lo; i < hi; i++)
					{
						// (only when the projection hands out the object's own string: one returned by value would be built and
						// destroyed here just to be asked for its address, and would have left that address behind by the copy)
						if constexpr (std::is_lvalue_reference_v<std::invoke_result_t<Proj&, const T&>>)
							if (i + kAhead < hi)
								__builtin_prefetch(std::invoke(proj, v[i + kAhead]).data(), 0, 0);
						const std::string& s = std::invoke(proj, v[i]);
						p.offsets[i] = static_cast<OFF>(at);
						std::copy(s.begin(), s.end(), p.blob.data() + at);
						at += s.size();
					}
				});
				return p;
			}

			template <typename T, typename Proj>
			inline void sortByStringKey(std::vector<T>& v, Proj& proj, bool stable, bool enableMultiThreading)
			{
				const std::size_t threads = helperThreads(v.size(), enableMultiThreading);
				const std::vector<std::uint64_t> rangeBytes = stringRangeBytes(v, proj, threads);
				rs_ctx* ctx = context();
				check(ctx, rs_ctx_set_host_threads(ctx, static_cast<int>(hostThreads(v.size(), enableMultiThreading))));
				if (rangeBytes.back() < (std::uint64_t(1) << 32) && v.size() < 0xFFFFFFFFull)
				{
					PackedStrings<std::uint32_t> p = packStrings<std::uint32_t>(v, proj, threads, rangeBytes);
					Uninitialized<std::uint32_t> perm(v.size());
					check(ctx, rs_sort_strings32(ctx, p.blob.data(), p.offsets.data(), v.size(), perm.data(), stable ? 1 : 0));
					applyPermutation(v, perm.data(), threads);
				}
				else
				{
					PackedStrings<std::uint64_t> p = packStrings<std::uint64_t>(v, proj, threads, rangeBytes);
					Uninitialized<std::uint64_t> perm(v.size());
					check(ctx, rs_sort_strings(ctx, p.blob.data(), p.offsets.data(), v.size(), perm.data(), stable ? 1 : 0));
					applyPermutation(v, perm.data(), threads);
				}
			}
		}

		namespace value
		{
			template <shared::supported T>
			inline void sortDispatcher(std::vector<T>& v, bool enableMultiThreading)
			{
				if (v.size() < 2)
					return;

				if constexpr (shared::is_string<T>)
				{
					// isSortedBi (:269-305) on the host, as in the reference: ascending -> done, descending -> reversed in
					// place; left at the first pair that rules both out.  (Numeric keys get the same check on the device.)
					bool ascending = true, descending = true;
					for (std::size_t i = 0, last = v.size() - 1; i < last && (ascending || descending); i++)
					{
						const int order = v[i].compare(v[i + 1]);
						if (order > 0) ascending = false;
						if (order < 0) descending = false;
					}
					if (ascending)
						return;
					if (descending)
					{
						std::reverse(v.begin(), v.end());
						return;
					}
					std::identity id;
					device::sortByStringKey(v, id, false, enableMultiThreading);
				}
				else
				{
					static_assert(!std::same_as<T, bool>,
						"ERROR: Unable to sort vector! CAUSE: std::vector<bool> has no element storage!");
					device::checkKeyType<T>();
					rs_ctx* ctx = device::context();
					device::check(ctx, rs_ctx_set_host_threads(ctx, static_cast<int>(device::hostThreads(v.size(), enableMultiThreading, sizeof(T)))));
					device::check(ctx, rs_sort_keys(ctx, v.data(), v.size(), static_cast<int>(sizeof(T)),
						device::transformOf<T>(), 0, static_cast<int>(sizeof(T)) * 8));
				}
			}
		}

		namespace key
		{
			template <typename T, typename Proj>
			requires shared::sortable<T, Proj>
			inline void sortDispatcher(std::vector<T>& v, Proj proj, bool enableMultiThreading)
			{
				using Key = shared::sort_key<T, Proj>;
				const std::size_t SIZE = v.size();
				if (SIZE < 2)
					return;
				if (device::isAscending(v, proj))          // :1607-1608
					return;

				if constexpr (shared::is_string<Key>)
				{
					device::sortByStringKey(v, proj, true, enableMultiThreading);
				}
				else
				{
					device::checkKeyType<Key>();

					const std::size_t threads = device::helperThreads(SIZE, enableMultiThreading);
					// the key column (:1581-1585); a bool key is stored as a byte (std::vector<bool> has no element storage)
					using Stored = std::conditional_t<std::same_as<Key, bool>, std::uint8_t, Key>;
					device::Uninitialized<Stored> k(SIZE);
					device::forRanges(SIZE, threads, [&](std::size_t lo, std::size_t hi, std::size_t) {
						for (std::size_t i = lo; i < hi; i++)
							k[i] = static_cast<Stored>(std::invoke(proj, v[i]));
					});

					rs_ctx* ctx = device::context();
					device::check(ctx, rs_ctx_set_host_threads(ctx, static_cast<int>(device::hostThreads(SIZE, enableMultiThreading, std::max(sizeof(Stored), sizeof(T))))));
					constexpr int XF = device::transformOf<Key>();
					constexpr int BITS = static_cast<int>(sizeof(Key)) * 8;

					// Trivially copyable objects are plain bytes: they follow their keys to the GPU and sortByIndices' random
					// gather (:1111-1154) runs there, at HBM speed, instead of on the host (10^7 16-byte objects: the
					// host gather alone took longer than upload, sort, gather and download together).
					if constexpr (std::is_trivially_copyable_v<T> && sizeof(T) % 4 == 0 && sizeof(T) <= 256 && alignof(T) >= 4)
					{
						if (SIZE < std::numeric_limits<std::uint32_t>::max())
						{
							device::check(ctx, rs_sort_rows(ctx, static_cast<void*>(v.data()), SIZE, static_cast<int>(sizeof(T)),
								k.data(), static_cast<int>(sizeof(Key)), XF));
							return;
						}
					}

					if (SIZE <= std::numeric_limits<std::uint32_t>::max())
					{
						device::Uninitialized<std::uint32_t> indices(SIZE);   // iota (:1571-1572), 4-byte payload
						device::forRanges(SIZE, threads, [&](std::size_t lo, std::size_t hi, std::size_t) {
							for (std::size_t i = lo; i < hi; i++)
								indices[i] = static_cast<std::uint32_t>(i);
						});
						device::check(ctx, rs_sort_pairs(ctx, k.data(), indices.data(), SIZE, static_cast<int>(sizeof(Key)), 4, XF, 0, BITS));
						device::applyPermutation(v, indices.data(), threads);
					}
					else
					{
						device::Uninitialized<std::uint64_t> indices(SIZE);
						for (std::size_t i = 0; i < SIZE; i++)
							indices[i] = i;
						device::check(ctx, rs_sort_pairs(ctx, k.data(), indices.data(), SIZE, static_cast<int>(sizeof(Key)), 8, XF, 0, BITS));
						device::applyPermutation(v, indices.data(), threads);
					}
				}
			}
		}
	}

	// =============
	// -----API-----
	// =============

	template <typename T, typename Proj = std::identity>
	inline void sort(std::vector<T>& v, Proj proj = {}, bool enableMultiThreading = false)
	{
		if constexpr (std::same_as<std::remove_cvref_t<Proj>, std::identity>)
		{
			static_assert(
				internal::shared::supported<T>,
				"ERROR: Unable to sort vector! CAUSE: Unsupported type!"
			);

			internal::value::sortDispatcher(v, enableMultiThreading);
		}
		else
		{
			static_assert(
				internal::shared::sortable<T, Proj>,
				"ERROR: Unable to sort vector! CAUSE: Key extractor is invalid or returns an unsupported type!"
			);

			internal::key::sortDispatcher(v, proj, enableMultiThreading);
		}
	}
};
