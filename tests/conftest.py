import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import radix_sort_amd  # noqa: E402,F401  (registers the radix-sort_amd/ directory as a package)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from cpu_checkers import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def ref():
    from cpu_checkers import load_ref
    r = load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libref_sort.so not built (reference tree absent)")
    return r


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    eng = radix_sort_amd.Engine(0)
    yield eng
    eng.close()
