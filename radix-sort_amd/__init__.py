"""radix-sort_amd: MI355X-native sort engine behind the reference's radix_sort::sort boundary.

Only what the sort path needs lives here: csrc/ (HIP kernels + C-ABI), the ctypes binding to the
C-ABI, the host-side handle on it, the multi-GPU splitter and the input shapes.
"""
from . import shapes  # noqa: F401
from ._lib import LIB_PATH, RsortError, load as load_library  # noqa: F401
from .engine import XF_IEEE, XF_NONE, XF_SIGNED, Engine  # noqa: F401
