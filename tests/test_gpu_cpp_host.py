"""GPU: the C++ host mirror (rust_compress_amd/host/compress.hpp) runs the reference's tests over the C-ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "rust_compress_amd", "host")


def _build():
    exe = os.path.join(HOST, "test_compress")
    cmd = ["g++", "-std=c++17", "-O1", os.path.join(HOST, "test_compress.cpp"), "-L" + os.path.join(ROOT, "rust_compress_amd", "csrc"),
           "-lrcx", "-Wl,-rpath," + os.path.join(ROOT, "rust_compress_amd", "csrc"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cpp_host_compiles():
    """CPU: the header and its test program compile and link against librcx.so"""
    from rust_compress_amd import _native
    _native.lib()
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_cpp_host_reference_tests():
    exe = _build()
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "CPP_HOST_OK" in p.stdout, p.stdout + p.stderr
