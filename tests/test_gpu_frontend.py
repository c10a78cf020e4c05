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
