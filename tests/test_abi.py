"""CPU: librsort_hip.so loads and exports every symbol include/rsort_hip.h declares; argument checks
that need no GPU behave; the product package never touches oracle/."""
import ctypes as C
import os
import re

import pytest

import radix_sort_amd as rs
from radix_sort_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rsort_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(name for name, _, _ in _lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(rs.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert rs.load_library().rs_version() >= 100


def test_null_context_is_rejected_without_gpu():
    lib = rs.load_library()
    assert lib.rs_ctx_synchronize(None) != 0
    assert lib.rs_sort_keys(None, None, 0, 4, 0, 0, 32) != 0
    assert lib.rs_last_error(None) == b"null context"


def test_no_gpu_means_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rs.RsortError):
        rs.Engine(0)


def test_product_path_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "radix-sort_amd")
    offenders = []
    for base in (pkg, os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"rs_oracle|libref_sort|cpu_checkers|orc_sort|ref_sort_", text):
                        offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
