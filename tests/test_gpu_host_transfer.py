"""GPU: the host-buffer entry points move large arrays through the staged, optionally multi-threaded transfer
(radix-sort_amd/csrc/rsort_transfer.hpp).  Sizes straddle the 1 MiB switch-over and the 4 MiB chunk / lane boundaries;
the result must not depend on the thread count (rs_ctx_set_host_threads)."""
import numpy as np
import pytest

import radix_sort_amd as rs

CHUNK = 4 << 20


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 2, 3, 4, 9])
def test_keys_any_thread_count(threads):
    eng = rs.Engine(0)
    eng.set_host_threads(threads)
    rng = np.random.default_rng(threads)
    for nbytes in [(1 << 20) - 4, 1 << 20, (1 << 20) + 4, CHUNK - 4, CHUNK + 4, 2 * CHUNK, 5 * CHUNK + 12, 9 * CHUNK - 8, 37 * CHUNK + 4]:
        v = rng.integers(0, 2**32, nbytes // 4, dtype=np.uint32)
        got = eng.sort_keys_host(v)
        assert got.tobytes() == np.sort(v).tobytes(), (threads, nbytes)
    d = rng.standard_normal(3 * CHUNK // 8 + 5)
    assert eng.sort_keys_host(d).tobytes() == np.sort(d).tobytes()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 4])
def test_pairs_and_back_to_back_calls(threads):
    eng = rs.Engine(0)
    eng.set_host_threads(threads)
    rng = np.random.default_rng(5)
    for n in [3_000_000, 200_000, 5_000_001, 17]:                 # large, small, large, tiny on one context
        k = rng.integers(0, 1000, n, dtype=np.int32)             # many ties: the payload order shows stability
        gk, gp = eng.sort_pairs_host(k)
        order = np.argsort(k, kind="stable")
        assert gk.tobytes() == k[order].tobytes() and (gp == order.astype(np.uint64)).all(), (threads, n)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [120_000, 600_000])      # blob staged + offsets/permutation through the small path; all staged
def test_strings_large_blob(n):
    eng = rs.Engine(0)
    eng.set_host_threads(4)
    rng = np.random.default_rng(11)
    lens = rng.integers(0, 21, n)
    offsets = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    blob = rng.integers(0, 256, int(offsets[-1]), dtype=np.uint8)      # ~6 MB blob, 4.8 MB offsets, 4.8 MB permutation
    perm = eng.sort_strings_host(blob, offsets, stable=True)
    raw = blob.tobytes()
    strs = [raw[int(offsets[i]):int(offsets[i + 1])] for i in range(n)]
    want = sorted(range(n), key=lambda i: strs[i])
    assert perm.tolist() == want
    eng.close()


@pytest.mark.gpu
def test_bad_thread_count_is_an_argument_error():
    eng = rs.Engine(0)
    with pytest.raises(rs.RsortError):
        eng.set_host_threads(0)
    eng.close()


@pytest.mark.gpu
def test_ordered_input_is_returned_without_a_download():
    """Ascending keys: the pre-check answers, nothing is fetched back, the caller's arrays are the result (large and small
    arrays, keys and pairs); descending keys-only input is reversed on the device and does come back."""
    eng = rs.Engine(0)
    eng.set_host_threads(2)
    for n in [50_000, 3_000_000]:
        asc = np.arange(n, dtype=np.uint32) * 3
        assert eng.sort_keys_host(asc).tobytes() == asc.tobytes()
        assert eng.sort_keys_host(asc[::-1].copy()).tobytes() == asc.tobytes()
        k = (np.arange(n, dtype=np.int64) // 4)                    # ascending with ties
        gk, gp = eng.sort_pairs_host(k)
        assert gk.tobytes() == k.tobytes() and (gp == np.arange(n, dtype=np.uint64)).all()
        mixed = asc.copy(); mixed[n // 2] = 0                       # one descent: a real sort, result fetched
        assert eng.sort_keys_host(mixed).tobytes() == np.sort(mixed).tobytes()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("precheck", [True, False])
def test_rows_small_key_column_large_rows(precheck):
    """rs_sort_rows with a key column that travels through the small-array bounce buffer (< 1 MiB) and rows that travel
    through the transfer lanes (>= 1 MiB) WHILE the pair sort runs: the lanes reuse the bounce buffer, so the key copy must
    have read it first (one-workgroup sorts and sorts without the pre-check never wait for the stream in between)."""
    import ctypes as C
    eng = rs.Engine(0)
    eng.set_precheck(precheck)
    eng.set_host_threads(4)
    rng = np.random.default_rng(31)
    for n, row_bytes in [(8192, 256), (16384, 64), (5000, 256), (200_000, 16)]:
        for rep in range(3):
            keys = rng.integers(0, 50, n, dtype=np.int32)                       # many ties: stability shows in the rows
            rows = rng.integers(0, 2**32, (n, row_bytes // 4), dtype=np.uint32)
            rows[:, 0] = np.arange(n, dtype=np.uint32)                          # the row's original position
            got = rows.copy()
            rc = eng.lib.rs_sort_rows(eng.h, got.ctypes.data_as(C.c_void_p), n, row_bytes, keys.ctypes.data_as(C.c_void_p), 4, 1)
            assert rc == 0, eng.lib.rs_last_error(eng.h)
            order = np.argsort(keys, kind="stable")
            assert got.tobytes() == rows[order].tobytes(), (n, row_bytes, rep)
    eng.close()
