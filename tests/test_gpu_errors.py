"""The C-ABI's error behaviour on a GPU box: bad arguments come back as RS_ERR_ARG with a message, nothing is sorted,
nothing crashes, and the context stays usable."""
import ctypes as C

import numpy as np
import pytest

import radix_sort_amd as rs

pytestmark = pytest.mark.gpu

RS_OK, RS_ERR_ARG = 0, 1


def test_bad_arguments_are_rejected_with_a_message():
    import torch
    lib = rs.load_library()
    eng = rs.Engine(0)
    t = torch.arange(100, dtype=torch.int32, device="cuda")
    alt = torch.empty_like(t)
    p, q = C.c_void_p(t.data_ptr()), C.c_void_p(alt.data_ptr())
    cases = [
        lambda: lib.rs_sort_keys_device(eng.h, p, q, 100, 3, 0, 0, 24),        # key width
        lambda: lib.rs_sort_keys_device(eng.h, p, q, 100, 4, 7, 0, 32),        # transform
        lambda: lib.rs_sort_keys_device(eng.h, p, q, 100, 4, 0, 8, 8),         # empty bit range
        lambda: lib.rs_sort_keys_device(eng.h, p, q, 100, 4, 0, 0, 40),        # bit range past the key
        lambda: lib.rs_sort_keys_device(eng.h, None, q, 100, 4, 0, 0, 32),     # null keys
        lambda: lib.rs_sort_keys_device(eng.h, p, q, 100, 2, 2, 0, 16),        # IEEE needs 4 or 8 bytes
        lambda: lib.rs_sort_pairs_device(eng.h, p, q, None, None, 100, 4, 2, 0, 0, 32),    # payload width
        lambda: lib.rs_sort_pairs_device(eng.h, p, None, None, None, 100, 4, 4, 0, 0, 32),  # null payload
        lambda: lib.rs_partition_device(eng.h, p, q, None, None, 100, 4, 0, 0, 0, 16, None),  # more than one digit
        lambda: lib.rs_partition_device(eng.h, p, None, None, None, 100, 4, 0, 0, 0, 8, None),  # null output
        lambda: lib.rs_digit_histogram_device(eng.h, p, 100, 4, 0, 0, 32, None),            # null counts
    ]
    for i, call in enumerate(cases):
        assert call() == RS_ERR_ARG, i
        assert len(lib.rs_last_error(eng.h)) > 0, i
    assert torch.equal(t.cpu(), torch.arange(100, dtype=torch.int32)), "a rejected call must not touch the keys"
    # the context still works
    v = np.array([5, 3, 9, 1], dtype=np.uint32)
    assert eng.sort_keys_host(v, 0).tolist() == [1, 3, 5, 9]
    # n = 0 and n = 1 are valid and do nothing
    assert lib.rs_sort_keys_device(eng.h, p, q, 0, 4, 0, 0, 32) == RS_OK
    assert lib.rs_sort_keys_device(eng.h, p, q, 1, 4, 0, 0, 32) == RS_OK
    eng.close()


def test_kernel_fault_is_reported_once_and_not_lost():
    """The device error word (include/rsort_hip.h: rs_ctx_synchronize 'reports a fault of any sort queued since the last
    call'): a fault survives sorts queued behind it, is reported by the next synchronisation, and does not stick to later
    calls.  The fault is injected by the library's test hook (what a sweep kernel does when an output offset leaves the
    array), so the test does not depend on winning a race against a running sort."""
    import torch
    lib = rs.load_library()
    lib.rs_debug_inject_fault.restype = C.c_int
    lib.rs_debug_inject_fault.argtypes = [C.c_void_p, C.c_uint]
    eng = rs.Engine(0)
    eng.set_precheck(False)
    RS_ERR_KERNEL = 3
    big = torch.randint(0, 2**31 - 1, (3_000_000,), dtype=torch.int32, device="cuda")
    small = torch.randint(0, 2**31 - 1, (500,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    # a fault, then two more sorts queued behind it (one multi-tile, whose start-of-sort memsets must not wipe it; one single-tile)
    assert lib.rs_debug_inject_fault(eng.h, 2) == RS_OK
    eng.sort_keys_ptr(big.data_ptr(), big.numel(), 4, 1)
    eng.sort_keys_ptr(small.data_ptr(), small.numel(), 4, 1)
    rc = lib.rs_ctx_synchronize(eng.h)
    assert rc == RS_ERR_KERNEL, rc
    assert b"outside the array" in lib.rs_last_error(eng.h)
    # reported once: the next calls are clean, small and large, device and host entry points
    assert lib.rs_ctx_synchronize(eng.h) == RS_OK
    eng.sort_keys_ptr(small.data_ptr(), small.numel(), 4, 1)
    assert lib.rs_ctx_synchronize(eng.h) == RS_OK
    v = np.array([5, 3, 9, 1], dtype=np.uint32)
    assert eng.sort_keys_host(v, 0).tolist() == [1, 3, 5, 9]
    big2 = torch.randint(0, 2**31 - 1, (3_000_000,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.sort_keys_ptr(big2.data_ptr(), big2.numel(), 4, 1)
    assert lib.rs_ctx_synchronize(eng.h) == RS_OK
    assert bool((big2[1:] >= big2[:-1]).all())
    # a look-back timeout code (1) is reported with its own message
    assert lib.rs_debug_inject_fault(eng.h, 1) == RS_OK
    assert lib.rs_ctx_synchronize(eng.h) == RS_ERR_KERNEL
    assert b"look-back" in lib.rs_last_error(eng.h)
    assert lib.rs_ctx_synchronize(eng.h) == RS_OK
    eng.close()
