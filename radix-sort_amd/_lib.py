"""ctypes binding to librsort_hip.so (the C-ABI in include/rsort_hip.h).

Loading fails loudly when the library is missing: there is no CPU fallback on the product path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librsort_hip.so")

# every symbol include/rsort_hip.h declares: (name, restype, argtypes)
_vp, _sz, _i = C.c_void_p, C.c_size_t, C.c_int
SYMBOLS = [
    ("rs_version", _i, []),
    ("rs_ctx_create", _i, [C.POINTER(_vp), _i]),
    ("rs_ctx_destroy", _i, [_vp]),
    ("rs_last_error", C.c_char_p, [_vp]),
    ("rs_ctx_set_stream", _i, [_vp, _vp]),
    ("rs_ctx_reserve", _i, [_vp, _sz, _i, _i]),
    ("rs_ctx_synchronize", _i, [_vp]),
    ("rs_ctx_set_precheck", _i, [_vp, _i]),
    ("rs_ctx_set_tile_order", _i, [_vp, _i]),
    ("rs_ctx_set_hybrid", _i, [_vp, _i]),
    ("rs_ctx_set_host_threads", _i, [_vp, _i]),
    ("rs_ctx_set_profiling", _i, [_vp, _i]),
    ("rs_ctx_get_profile", _i, [_vp, C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    ("rs_sort_keys_device", _i, [_vp, _vp, _vp, _sz, _i, _i, _i, _i]),
    ("rs_sort_keys_device_noalloc", _i, [_vp, _vp, _vp, _sz, _i, _i, _i, _i]),
    ("rs_sort_pairs_device", _i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i]),
    ("rs_digit_histogram_device", _i, [_vp, _vp, _sz, _i, _i, _i, _i, _vp]),
    ("rs_partition_device", _i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    ("rs_sort_keys", _i, [_vp, _vp, _sz, _i, _i, _i, _i]),
    ("rs_sort_pairs", _i, [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i]),
    ("rs_sort_rows", _i, [_vp, _vp, _sz, _i, _vp, _i, _i]),
    ("rs_sort_strings", _i, [_vp, _vp, _vp, _sz, _vp, _i]),
    ("rs_sort_strings32", _i, [_vp, _vp, _vp, _sz, _vp, _i]),
]

_lib = None


class RsortError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"librsort_hip status {code}: {message}")
        self.code = code


def load():
    """The loaded library with argtypes set; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "There is no CPU fallback for the sort path.")
        # torch ships a HIP/HSA runtime of its own (torch/lib).  Whichever runtime a process loads first is the one that sees
        # the GPU: with /opt/rocm's loaded first (through this library), a later `import torch` finds no device.  So when
        # torch is installed it goes first, and this library binds to the runtime torch brought.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
