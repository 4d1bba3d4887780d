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


def test_host_register_argument_checks():
    """rcx_host_register / rcx_host_unregister (page-locking a caller's buffers, include/rcx.h): null or empty ranges are the caller's
    error before any device is looked at; without a device a real range is an error too, never a crash."""
    import numpy as np
    import torch
    from rust_compress_amd import _native
    lib = _native.lib()
    assert lib.rcx_host_register(None, 4096) == _native.RC_BAD_ARG
    buf = np.zeros(1 << 16, np.uint8)
    assert lib.rcx_host_register(buf.ctypes.data, 0) == _native.RC_BAD_ARG
    assert lib.rcx_host_unregister(None) == _native.RC_BAD_ARG
    if not torch.cuda.is_available():
        assert lib.rcx_host_register(buf.ctypes.data, buf.size) != _native.RC_OK
        assert lib.rcx_host_unregister(buf.ctypes.data) != _native.RC_OK


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


def test_partition_export_is_dist_partition():
    """rcx_partition (the C-ABI's block ranges for more than one device) == rust_compress_amd.dist.partition, the ranges the
    Python ranks use: contiguous, in order, balanced by weight, bounds[0] = 0, bounds[parts] = n."""
    import ctypes as C
    import numpy as np
    from rust_compress_amd import _native as N, dist
    L = C.CDLL(N.LIB_PATH)                       # (pure host arithmetic: no device needed)
    L.rcx_partition.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.rcx_partition.restype = None
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 7, 64, 1000):
        for parts in (1, 2, 3, 8, 13):
            for w in (np.ones(n, np.uint64), rng.integers(0, 70000, n).astype(np.uint64), np.zeros(n, np.uint64)):
                b = np.zeros(parts + 1, np.uint32)
                L.rcx_partition(w.ctypes.data, n, parts, b.ctypes.data)
                assert b[0] == 0 and b[parts] == n and (np.diff(b.astype(np.int64)) >= 0).all()
                if w.sum():
                    assert b.tolist() == dist.partition(w, parts).tolist(), (n, parts)

