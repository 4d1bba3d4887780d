"""CPU suite: the C-ABI library loads and exports every symbol include/rcx.h declares; without a GPU it
refuses to create a context (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "rcx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rcx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rust_compress_amd import _native
    lib = _native.lib()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "librcx.so does not export %s" % n
    assert set(_native.EXPORTS) <= set(names)
    assert lib.rcx_version() >= 1


def test_bounds_and_strings():
    from rust_compress_amd import _native
    lib = _native.lib()
    assert lib.rcx_lz4_compression_bound(0) == 20 and lib.rcx_lz4_compression_bound(65536) == 65536 + 257 + 20
    assert lib.rcx_lz4_compression_bound(0x7E000001) == 0          # lz4.rs:176-177 None
    assert lib.rcx_status_string(1) == b"unexpected end of file"    # lib.rs:56-59
    assert lib.rcx_status_string(17) == b"not enough bits" and lib.rcx_status_string(30) == b"Overly long run"
    assert lib.rcx_status_string(24) == b"invalid checksum on zlib stream"


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import rust_compress_amd as R
    with pytest.raises(R.RcxError):
        R.Context()


def test_product_never_imports_oracle():
    """no import / dlopen / include of anything under oracle/ from the product package"""
    pkg = os.path.join(ROOT, "rust_compress_amd")
    pat = re.compile(r"import\s+oracle|from\s+oracle|oracle_py|liboracle|#include\s*[\"<][^\n]*oracle")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                s = open(os.path.join(dp, f)).read()
                assert not pat.search(s), f
