"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (rust_compress_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)
szp = C.POINTER(C.c_size_t)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.o_lz4_compression_bound.restype = C.c_uint64
        _LIB.o_lz4_compression_bound.argtypes = [C.c_uint64]
        _LIB.o_ari_byte_encode_bound.restype = C.c_uint64
        _LIB.o_ari_byte_encode_bound.argtypes = [C.c_uint64]
        _LIB.o_rle_encode_bound.restype = C.c_uint64
        _LIB.o_rle_encode_bound.argtypes = [C.c_uint64]
        _LIB.o_adler32.restype = C.c_uint32
        _LIB.o_adler32.argtypes = [C.c_void_p, C.c_size_t]
        _LIB.o_batch_run.restype = C.c_double
        _LIB.o_batch_run.argtypes = [C.c_int] + [C.c_void_p] * 11 + [C.c_uint32, C.c_int]
    return _LIB


class OracleError(Exception):
    def __init__(self, status, partial=b""):
        super().__init__("oracle status %d" % status)
        self.status = status
        self.partial = partial


def _buf(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    if a.size == 0:
        a = np.zeros(1, dtype=np.uint8)[:0]
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _call_bytes(fn, data, cap, extra_in=(), want_used=False, want_flags=False, raise_on_error=True):
    """fn(in, n, [extra], out, cap, &out_len, [&in_used], [&flags])"""
    a = _buf(data)
    out = np.empty(max(int(cap), 1), dtype=np.uint8)
    olen = C.c_size_t(0)
    used = C.c_size_t(0)
    flags = C.c_uint32(0)
    args = [_ptr(a), C.c_size_t(a.size)] + list(extra_in) + [_ptr(out), C.c_size_t(int(cap)), C.byref(olen)]
    if want_used:
        args.append(C.byref(used))
    if want_flags:
        args.append(C.byref(flags))
    st = fn(*args)
    res = out[: olen.value].tobytes()
    if st != 0 and raise_on_error:
        raise OracleError(st, res)
    ret = [res]
    if want_used:
        ret.append(used.value)
    if want_flags:
        ret.append(flags.value)
    if not raise_on_error:
        ret.append(st)
    return ret[0] if len(ret) == 1 else tuple(ret)


# ---- lz4 ----
def lz4_decode_block(data, cap=None, **kw):
    cap = cap if cap is not None else max(64, 256 * len(data) + 64)
    return _call_bytes(lib().o_lz4_decode_block, data, cap, **kw)


def lz4_compression_bound(n):
    return lib().o_lz4_compression_bound(n)


def lz4_encode_block(data, cap=None, **kw):
    cap = cap if cap is not None else lz4_compression_bound(len(data))
    return _call_bytes(lib().o_lz4_encode_block, data, cap, **kw)


def lz4_frame_decode(data, cap=None, **kw):
    cap = cap if cap is not None else max(64, 256 * len(data) + 64)
    return _call_bytes(lib().o_lz4_frame_decode, data, cap, want_used=True, **kw)


def lz4_frame_encode(data, **kw):
    return _call_bytes(lib().o_lz4_frame_encode, data, len(data) + 4 * (len(data) // (256 * 1024) + 1) + 32, **kw)


# ---- flate / zlib / adler ----
def inflate(data, cap=None, **kw):
    cap = cap if cap is not None else max(1 << 16, 1100 * len(data))
    return _call_bytes(lib().o_inflate, data, cap, want_used=True, want_flags=True, **kw)


def zlib_decode(data, cap=None, **kw):
    cap = cap if cap is not None else max(1 << 16, 1100 * len(data))
    return _call_bytes(lib().o_zlib_decode, data, cap, want_used=True, want_flags=True, **kw)


def adler32(data):
    a = _buf(data)
    return lib().o_adler32(_ptr(a), a.size)


# ---- bwt / mtf / dc ----
def bwt_encode(data):
    a = _buf(data)
    out = np.empty(max(a.size, 1), dtype=np.uint8)
    origin = C.c_uint32(0)
    st = lib().o_bwt_encode(_ptr(a), C.c_size_t(a.size), _ptr(out), C.byref(origin))
    if st:
        raise OracleError(st)
    return out[: a.size].tobytes(), origin.value


def bwt_suffixes(data):
    a = _buf(data)
    sa = np.empty(max(a.size, 1), dtype=np.uint32)
    lib().o_bwt_compute_suffixes(_ptr(a), C.c_size_t(a.size), _ptr(sa))
    return sa[: a.size].copy()


def bwt_inversion_table(L, origin):
    a = _buf(L)
    t = np.empty(max(a.size, 1), dtype=np.uint32)
    st = lib().o_bwt_inversion_table(_ptr(a), C.c_size_t(a.size), C.c_uint32(origin), _ptr(t))
    if st:
        raise OracleError(st)
    return t[: a.size].copy()


def bwt_decode(L, origin, minimal=False):
    a = _buf(L)
    out = np.empty(max(a.size, 1), dtype=np.uint8)
    fn = lib().o_bwt_decode_minimal if minimal else lib().o_bwt_decode
    st = fn(_ptr(a), C.c_size_t(a.size), C.c_uint32(origin), _ptr(out))
    if st:
        raise OracleError(st)
    return out[: a.size].tobytes()


def bwt_stream_encode(data, block_size):
    n = len(data)
    nblk = (n + block_size - 1) // block_size if block_size else 0
    return _call_bytes(lib().o_bwt_stream_encode, data, 4 + n + 8 * nblk + 8, extra_in=[C.c_uint32(block_size)])


def bwt_stream_decode(data, cap=None, **kw):
    cap = cap if cap is not None else len(data) + 16
    return _call_bytes(lib().o_bwt_stream_decode, data, cap, **kw)


def mtf_encode(data):
    a = _buf(data)
    out = np.empty(max(a.size, 1), dtype=np.uint8)
    lib().o_mtf_encode(_ptr(a), C.c_size_t(a.size), _ptr(out))
    return out[: a.size].tobytes()


def mtf_decode(data):
    a = _buf(data)
    out = np.empty(max(a.size, 1), dtype=np.uint8)
    lib().o_mtf_decode(_ptr(a), C.c_size_t(a.size), _ptr(out))
    return out[: a.size].tobytes()


class _DcCtx(C.Structure):
    _fields_ = [("symbol", C.c_uint8), ("last_rank", C.c_uint8), ("distance_limit", C.c_uint32)]


def dc_encode(data, with_ctx=False):
    """-> np.uint32 words (256 init + k distances) [, list of (symbol, last_rank, distance_limit)]"""
    a = _buf(data)
    words = np.empty(256 + a.size + 1, dtype=np.uint32)
    nw = C.c_size_t(0)
    ctx = (_DcCtx * (a.size + 1))() if with_ctx else None
    st = lib().o_dc_encode(_ptr(a), C.c_size_t(a.size), _ptr(words), C.c_size_t(words.size), C.byref(nw), ctx)
    if st:
        raise OracleError(st)
    w = words[: nw.value].copy()
    if with_ctx:
        k = nw.value - 256
        return w, [(ctx[i].symbol, ctx[i].last_rank, ctx[i].distance_limit) for i in range(k)]
    return w


def dc_decode(words, n, with_ctx=False):
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.empty(max(n, 1), dtype=np.uint8)
    cons = C.c_size_t(0)
    ctx = (_DcCtx * (n + 2))() if with_ctx else None
    st = lib().o_dc_decode(_ptr(w), C.c_size_t(w.size), C.c_size_t(n), _ptr(out), C.byref(cons), ctx)
    if st:
        raise OracleError(st)
    if with_ctx:
        return out[:n].tobytes(), cons.value, [(ctx[i].symbol, ctx[i].last_rank, ctx[i].distance_limit) for i in range(cons.value)]
    return out[:n].tobytes(), cons.value


# ---- ari ----
def ari_byte_encode(data, **kw):
    return _call_bytes(lib().o_ari_byte_encode, data, 2 * len(data) + 16, **kw)


def ari_byte_decode(data, cap=None, **kw):
    cap = cap if cap is not None else max(1 << 16, 64 * len(data))
    return _call_bytes(lib().o_ari_byte_decode, data, cap, want_used=True, **kw)


def ari_binary_encode(data, rate):
    return _call_bytes(lib().o_ari_binary_encode, data, 2 * len(data) + 16, extra_in=[C.c_uint32(rate)])


def ari_binary_decode(data, rate, nbytes):
    a = _buf(data)
    out = np.empty(max(nbytes, 1), dtype=np.uint8)
    st = lib().o_ari_binary_decode(_ptr(a), C.c_size_t(a.size), C.c_uint32(rate), _ptr(out), C.c_size_t(nbytes))
    if st:
        raise OracleError(st)
    return out[:nbytes].tobytes()


def ari_apm_encode(data, raise_on_error=True):
    return _call_bytes(lib().o_ari_apm_encode, data, 2 * len(data) + 16, raise_on_error=raise_on_error)


def ari_apm_decode(data, nbytes):
    a = _buf(data)
    out = np.empty(max(nbytes, 1), dtype=np.uint8)
    st = lib().o_ari_apm_decode(_ptr(a), C.c_size_t(a.size), _ptr(out), C.c_size_t(nbytes))
    if st:
        raise OracleError(st)
    return out[:nbytes].tobytes()


def apm_tables():
    stretch = np.zeros(4096, dtype=np.int16); gate = np.zeros(17, dtype=np.uint16)
    lib().o_apm_tables(stretch.ctypes.data_as(C.c_void_p), gate.ctypes.data_as(C.c_void_p))
    return stretch, gate


def ari_proxy_encode(data):
    return _call_bytes(lib().o_ari_proxy_encode, data, 2 * len(data) + 16)


def ari_proxy_decode(data, nbytes):
    a = _buf(data)
    out = np.empty(max(nbytes, 1), dtype=np.uint8)
    st = lib().o_ari_proxy_decode(_ptr(a), C.c_size_t(a.size), _ptr(out), C.c_size_t(nbytes))
    if st:
        raise OracleError(st)
    return out[:nbytes].tobytes()


# ---- rle ----
def rle_encode(data, **kw):
    return _call_bytes(lib().o_rle_encode, data, lib().o_rle_encode_bound(len(data)), **kw)


def rle_decode(data, cap=None, **kw):
    cap = cap if cap is not None else max(1 << 16, 1024 * len(data))
    return _call_bytes(lib().o_rle_decode, data, cap, **kw)


# ---- batch driver (CPU baseline + bulk parity) ----
def batch_run(codec, in_base, in_off, in_len, out_base, out_off, out_cap, aux=None, n_out=None, threads=1):
    """All arrays numpy; returns (seconds, out_len, in_used, status)."""
    n = len(in_off)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint64)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint64)
    out_len = np.zeros(n, dtype=np.uint64)
    in_used = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)
    p = lambda a: None if a is None else a.ctypes.data
    if n_out is not None:
        n_out = np.ascontiguousarray(n_out, dtype=np.uint64)
    secs = lib().o_batch_run(int(codec), p(in_base), p(in_off), p(in_len), p(out_base), p(out_off), p(out_cap),
                             p(out_len), p(in_used), p(status), p(aux), p(n_out), n, int(threads))
    return secs, out_len, in_used, status
