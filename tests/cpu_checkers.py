"""ctypes bindings to the two CPU checkers (TEST INFRASTRUCTURE; see oracle/rs_oracle.cpp):

  oracle  -- oracle/librs_oracle.so, the restatement, always present after `make -C oracle`
  ref     -- oracle/_ref/libref_sort.so, the reference header itself compiled in the build container
             (absent => `ref` is None and the tests that need it skip)
"""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = os.path.join(_ROOT, "oracle", "librs_oracle.so")
_REF = os.path.join(_ROOT, "oracle", "_ref", "libref_sort.so")

_vp, _sz, _i, _u64p, _dp, _u8p = C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, path=_ORACLE):
        self.lib = C.CDLL(path)
        self.lib.orc_sort_keys.argtypes = [_vp, _sz, _i, _i, _i]
        self.lib.orc_sort_pairs.argtypes = [_vp, _vp, _sz, _i, _i, _i]
        self.lib.orc_sort_strings.argtypes = [_vp, _vp, _sz, _vp, _i]
        self.lib.orc_strategy.argtypes = [_vp, _sz, _i, _i, _i, _i]

    def sort_keys(self, keys, transform, mt=False):
        """sorted copy of a numpy key array (raw bits interpreted through `transform`)."""
        out = np.ascontiguousarray(keys).copy()
        rc = self.lib.orc_sort_keys(_ptr(out), out.size, out.itemsize, transform, int(mt))
        assert rc == 0, rc
        return out

    def sort_pairs(self, keys, transform, mt=False):
        """(sorted keys, perm) with perm[i] = original index of the element now at i (stable)."""
        k = np.ascontiguousarray(keys).copy()
        perm = np.empty(k.size, dtype=np.uint64)
        rc = self.lib.orc_sort_pairs(_ptr(k), _ptr(perm), k.size, k.itemsize, transform, int(mt))
        assert rc == 0, rc
        return k, perm

    def sort_strings(self, blob, offsets, stable):
        n = offsets.size - 1
        perm = np.empty(n, dtype=np.uint64)
        blob = np.ascontiguousarray(blob)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        rc = self.lib.orc_sort_strings(_ptr(blob), _ptr(offsets), n, _ptr(perm), int(stable))
        assert rc == 0, rc
        return perm

    def strategy(self, keys, transform, mt, hw):
        k = np.ascontiguousarray(keys)
        return self.lib.orc_strategy(_ptr(k), k.size, k.itemsize, transform, int(mt), hw)


class Ref:
    NOT_COVERED = 3

    def __init__(self, path=_REF):
        self.lib = C.CDLL(path)
        self.lib.ref_sort_keys.argtypes = [_vp, _sz, _i, _i, _i, _dp]
        self.lib.ref_sort_pairs.argtypes = [_vp, _vp, _sz, _i, _i, _i, _i, _dp]
        self.lib.ref_sort_strings.argtypes = [_vp, _vp, _sz, _i, _i, _vp, _vp, _vp, _dp]
        self.lib.ref_num_threads.argtypes = [_sz]
        self.lib.ref_num_threads.restype = _sz

    def sort_keys(self, keys, transform, mt=False):
        """(sorted copy | None if the build does not cover the input, seconds inside sort())."""
        out = np.ascontiguousarray(keys).copy()
        sec = C.c_double(0)
        rc = self.lib.ref_sort_keys(_ptr(out), out.size, out.itemsize, transform, int(mt), C.byref(sec))
        if rc == self.NOT_COVERED:
            return None, 0.0
        assert rc == 0, rc
        return out, sec.value

    def sort_pairs(self, keys, transform, mt=False, layout=0):
        k = np.ascontiguousarray(keys).copy()
        perm = np.empty(k.size, dtype=np.uint64)
        sec = C.c_double(0)
        rc = self.lib.ref_sort_pairs(_ptr(k), _ptr(perm), k.size, k.itemsize, transform, int(mt), layout, C.byref(sec))
        if rc == self.NOT_COVERED:
            return None, None, 0.0
        assert rc == 0, rc
        return k, perm, sec.value

    def sort_strings(self, blob, offsets, stable, mt=False):
        """stable=False -> (sorted blob, sorted offsets, seconds); stable=True -> (perm, seconds)."""
        n = offsets.size - 1
        blob = np.ascontiguousarray(blob)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        sec = C.c_double(0)
        if not stable:
            ob = np.empty_like(blob)
            oo = np.empty_like(offsets)
            rc = self.lib.ref_sort_strings(_ptr(blob), _ptr(offsets), n, 0, int(mt), _ptr(ob), _ptr(oo), None, C.byref(sec))
            assert rc == 0, rc
            return ob, oo, sec.value
        perm = np.empty(n, dtype=np.uint64)
        rc = self.lib.ref_sort_strings(_ptr(blob), _ptr(offsets), n, 1, int(mt), None, None, _ptr(perm), C.byref(sec))
        assert rc == 0, rc
        return perm, sec.value

    def num_threads(self, n):
        return self.lib.ref_num_threads(n)


def load_oracle():
    if not os.path.exists(_ORACLE):
        raise RuntimeError(f"{_ORACLE} missing: run `make -C oracle` (or __graft_entry__.build())")
    return Oracle()


def load_ref():
    return Ref() if os.path.exists(_REF) else None
