"""Host-side handle on the sort engine: a thin, checked wrapper over the C-ABI.

torch is used only for device memory and streams; every sort goes through librsort_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from .shapes import KEY_TYPES

XF_NONE, XF_SIGNED, XF_IEEE = 0, 1, 2


class Engine:
    """One rs_ctx (one device, one stream, one reusable workspace)."""

    def __init__(self, device: int = 0, stream=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.rs_ctx_create(C.byref(h), device)
        if rc != 0:
            raise _lib.RsortError(rc, f"rs_ctx_create(device={device}) failed (no usable GPU?)")
        self.h = h
        self.device = device
        self._own_stream = True
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "h", None):
            self.lib.rs_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, rc):
        if rc != 0:
            raise _lib.RsortError(rc, self.lib.rs_last_error(self.h).decode())

    def set_stream(self, stream):
        """stream: a raw hipStream_t integer (e.g. torch.cuda.current_stream().cuda_stream) or None.
        Handle 0 / None -- which is also what torch's DEFAULT stream reports -- selects the context's own stream, so work
        queued by torch on its default stream is NOT ordered before the sort: wait for it first
        (torch.cuda.current_stream().synchronize()), as bench.py and dist.py do."""
        self._ck(self.lib.rs_ctx_set_stream(self.h, C.c_void_p(stream or None)))
        self._own_stream = not stream

    def _wait_for_torch(self, t):
        """Tensor entry points only: when the sort runs on the context's own stream, what torch queued for this tensor
        must have finished (a sort of keys that are still being written scatters to offsets its histogram never saw)."""
        if self._own_stream:
            import torch
            torch.cuda.current_stream(t.device).synchronize()

    def _finish_for_torch(self):
        """Tensor entry points only: on the context's own stream nothing orders torch's work behind the sort, so the
        call waits for it (and surfaces RS_ERR_KERNEL); with an attached stream the sort is ordered on that stream and
        the call returns once it is queued.  The raw *_ptr variants never wait: follow them with synchronize()."""
        if self._own_stream:
            self.synchronize()

    def reserve(self, n, key_bytes, value_bytes=0):
        self._ck(self.lib.rs_ctx_reserve(self.h, n, key_bytes, value_bytes))

    def synchronize(self):
        self._ck(self.lib.rs_ctx_synchronize(self.h))

    def set_precheck(self, on: bool):
        self._ck(self.lib.rs_ctx_set_precheck(self.h, int(on)))

    def set_tile_order(self, per_xcd: bool):
        self._ck(self.lib.rs_ctx_set_tile_order(self.h, int(per_xcd)))

    def set_hybrid(self, mode: int):
        """Hybrid forms of keys-only sorts: -1 chosen per sort (default), 0 never, 1 the 16-bit-group form whenever the key
        shape allows, 2 the narrow form (sizes below 10^8) whenever it does."""
        self._ck(self.lib.rs_ctx_set_hybrid(self.h, int(mode)))

    def set_host_threads(self, threads: int):
        """Threads the host-buffer entry points may use for the transfers (default 1; at most 4 are used)."""
        self._ck(self.lib.rs_ctx_set_host_threads(self.h, int(threads)))

    def set_profiling(self, on: bool):
        self._ck(self.lib.rs_ctx_set_profiling(self.h, int(on)))

    def profile(self):
        """[ms histogram+scan, ms pass 0, ms pass 1, ...] of the last sort (profiling must be on)."""
        buf = (C.c_float * 16)()
        cnt = C.c_int(0)
        self._ck(self.lib.rs_ctx_get_profile(self.h, buf, 16, C.byref(cnt)))
        return [buf[i] for i in range(cnt.value)]

    # ---- device-resident (raw pointers) -----------------------------------------------------------
    def sort_keys_ptr(self, d_keys, n, key_bytes, transform, d_alt=None, begin_bit=0, end_bit=None):
        end_bit = key_bytes * 8 if end_bit is None else end_bit
        self._ck(self.lib.rs_sort_keys_device(self.h, C.c_void_p(d_keys), C.c_void_p(d_alt or None), n,
                                              key_bytes, transform, begin_bit, end_bit))

    def sort_keys_noalloc_ptr(self, d_keys, d_alt, n, key_bytes, transform, begin_bit=0, end_bit=None):
        """rs_sort_keys_device_noalloc: static device workspace, nothing allocated, nothing waited for (n <= 2^25)."""
        end_bit = key_bytes * 8 if end_bit is None else end_bit
        self._ck(self.lib.rs_sort_keys_device_noalloc(self.h, C.c_void_p(d_keys), C.c_void_p(d_alt or None), n,
                                                      key_bytes, transform, begin_bit, end_bit))

    def sort_pairs_ptr(self, d_keys, d_vals, n, key_bytes, value_bytes, transform, d_alt_keys=None,
                       d_alt_vals=None, begin_bit=0, end_bit=None):
        end_bit = key_bytes * 8 if end_bit is None else end_bit
        self._ck(self.lib.rs_sort_pairs_device(self.h, C.c_void_p(d_keys), C.c_void_p(d_vals),
                                               C.c_void_p(d_alt_keys or None), C.c_void_p(d_alt_vals or None), n,
                                               key_bytes, value_bytes, transform, begin_bit, end_bit))

    def digit_histogram_ptr(self, d_keys, n, key_bytes, transform, begin_bit, end_bit):
        """numpy uint64 [passes, 256]: digit counts of the transformed keys over [begin_bit, end_bit)."""
        passes = (end_bit - begin_bit + 7) // 8
        out = np.zeros((passes, 256), dtype=np.uint64)
        self._ck(self.lib.rs_digit_histogram_device(self.h, C.c_void_p(d_keys or None), n, key_bytes, transform, begin_bit,
                                                    end_bit, out.ctypes.data_as(C.c_void_p)))
        return out

    def partition_ptr(self, d_keys_in, d_keys_out, n, key_bytes, transform, begin_bit, end_bit, d_vals_in=None,
                      d_vals_out=None, value_bytes=0):
        """One stable out-of-place pass on the digit [begin_bit, end_bit) (<= 8 bits); returns its 256 digit counts."""
        counts = np.zeros(256, dtype=np.uint64)
        self._ck(self.lib.rs_partition_device(self.h, C.c_void_p(d_keys_in or None), C.c_void_p(d_keys_out or None),
                                              C.c_void_p(d_vals_in or None), C.c_void_p(d_vals_out or None), n, key_bytes,
                                              value_bytes, transform, begin_bit, end_bit, counts.ctypes.data_as(C.c_void_p)))
        return counts

    # ---- torch tensors on the engine's device ------------------------------------------------------
    def sort_keys_(self, t, transform, alt=None, begin_bit=0, end_bit=None):
        """In-place sort of a contiguous 1-D device tensor whose element size is the key width."""
        assert t.is_cuda and t.is_contiguous() and t.dim() == 1
        self._wait_for_torch(t)
        self.sort_keys_ptr(t.data_ptr(), t.numel(), t.element_size(), transform,
                           alt.data_ptr() if alt is not None else None, begin_bit, end_bit)
        self._finish_for_torch()
        return t

    def sort_pairs_(self, k, v, transform, alt_k=None, alt_v=None, begin_bit=0, end_bit=None):
        assert k.is_cuda and v.is_cuda and k.is_contiguous() and v.is_contiguous() and k.numel() == v.numel()
        self._wait_for_torch(k)
        self.sort_pairs_ptr(k.data_ptr(), v.data_ptr(), k.numel(), k.element_size(), v.element_size(), transform,
                            alt_k.data_ptr() if alt_k is not None else None,
                            alt_v.data_ptr() if alt_v is not None else None, begin_bit, end_bit)
        self._finish_for_torch()
        return k, v

    # ---- host numpy arrays (H2D + sort + D2H inside the library) -----------------------------------
    def sort_keys_host(self, a: np.ndarray, transform=None, begin_bit=0, end_bit=None):
        """Sorted copy of a numpy key array; transform defaults to the array dtype's."""
        out = np.ascontiguousarray(a).copy()
        xf = _transform_of(out.dtype) if transform is None else transform
        end_bit = out.itemsize * 8 if end_bit is None else end_bit
        self._ck(self.lib.rs_sort_keys(self.h, out.ctypes.data_as(C.c_void_p), out.size, out.itemsize, xf,
                                       begin_bit, end_bit))
        return out

    def sort_pairs_host(self, keys: np.ndarray, values: np.ndarray = None, transform=None):
        """(sorted keys, values carried along); values default to 0..n-1 as uint64 (the permutation)."""
        k = np.ascontiguousarray(keys).copy()
        v = np.arange(k.size, dtype=np.uint64) if values is None else np.ascontiguousarray(values).copy()
        xf = _transform_of(k.dtype) if transform is None else transform
        self._ck(self.lib.rs_sort_pairs(self.h, k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), k.size,
                                        k.itemsize, v.itemsize, xf, 0, k.itemsize * 8))
        return k, v


    def sort_strings_host(self, blob: np.ndarray, offsets: np.ndarray, stable: bool = True, narrow=None, widen=True):
        """perm (uint64) such that string perm[i] belongs at position i; string i = blob[offsets[i]:offsets[i+1]].
        narrow: use rs_sort_strings32 (4-byte offsets and permutation; blobs below 4 GiB) -- None = whenever it applies
        (offsets that are uint32 already are passed as they are; widen=False returns its uint32 permutation as it came)."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        n = np.asarray(offsets).size - 1
        last = int(np.asarray(offsets)[-1]) if n >= 0 else 0
        if narrow is None:
            narrow = last < (1 << 32)
        if narrow:
            off32 = np.ascontiguousarray(offsets, dtype=np.uint32)
            perm32 = np.empty(n, dtype=np.uint32)
            self._ck(self.lib.rs_sort_strings32(self.h, blob.ctypes.data_as(C.c_void_p), off32.ctypes.data_as(C.c_void_p), n,
                                                perm32.ctypes.data_as(C.c_void_p), int(stable)))
            return perm32.astype(np.uint64) if widen else perm32
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        perm = np.empty(n, dtype=np.uint64)
        self._ck(self.lib.rs_sort_strings(self.h, blob.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p), n,
                                          perm.ctypes.data_as(C.c_void_p), int(stable)))
        return perm

    def last_string_levels(self):
        """MSD levels the last string sort took (diagnostic)."""
        return int(self.lib.rs_debug_last_string_levels(self.h))


def _transform_of(dtype):
    kind = np.dtype(dtype).kind
    if kind == "f":
        return XF_IEEE
    if kind == "i":
        return XF_SIGNED
    if kind in "ub":
        return XF_NONE
    raise TypeError(f"unsupported key dtype {dtype}")


def transform_of_name(name):
    return KEY_TYPES[name][2]
