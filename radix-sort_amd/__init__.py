"""radix-sort_amd: MI355X-native sort engine behind the reference's radix_sort::sort boundary.

Only what the sort path needs lives here: csrc/ (HIP kernels + C-ABI), the ctypes binding to the
C-ABI, the host-side mirror of the reference interface, the multi-GPU splitter and the input shapes.
"""
from . import shapes  # noqa: F401
