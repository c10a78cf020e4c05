"""GPU: the C++20 drop-in front-end (include/radix_sort.hpp) against std::sort / std::stable_sort,
the same differential check the reference's benchmark::testing does (benchmark.cpp:394-577)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_frontend")


@pytest.mark.gpu
def test_cpp_frontend_matches_std_sort():
    assert os.path.exists(BIN), "tests/cpp/test_frontend missing: run __graft_entry__.build() in the container"
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WRONG SORTS: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


MATRIX = os.path.join(ROOT, "tools", "bench_matrix")


@pytest.mark.gpu
def test_reference_matrix_in_testing_mode():
    """tools/bench_matrix --testing: the reference harness' 16 types x 5 shapes x sizes (main.cpp:19-21, capped at 1e5
    here; the full-size run is kept in profiles/), each row compared with std::sort / std::stable_sort element by
    element exactly as benchmark::testing does."""
    assert os.path.exists(MATRIX), "tools/bench_matrix missing: run __graft_entry__.build() in the container"
    r = subprocess.run([MATRIX, "--testing", "--max-size", "100000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WRONG SORTS = 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("(correct)") == 2 * 5 * 16 * 7 and "wrong" not in r.stdout
