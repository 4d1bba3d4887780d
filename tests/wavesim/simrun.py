"""Python driver for the wave64 simulator build of the kernels (TEST INFRASTRUCTURE)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
from rust_compress_amd import batch as B  # noqa: E402
import build as _build  # noqa: E402


class KArgs(C.Structure):
    _fields_ = [("in_base", C.c_void_p), ("in_off", C.c_void_p), ("in_len", C.c_void_p),
                ("out_base", C.c_void_p), ("out_off", C.c_void_p), ("out_cap", C.c_void_p),
                ("out_len", C.c_void_p), ("in_used", C.c_void_p), ("status", C.c_void_p),
                ("aux", C.c_void_p), ("n_out", C.c_void_p), ("scratch", C.c_void_p),
                ("scratch_bytes", C.c_uint64), ("nblocks", C.c_uint32),
                ("out_mirror", C.c_void_p), ("gate", C.c_void_p), ("gate_host", C.c_void_p),
                ("gate_seq", C.c_uint32), ("gate_ticks", C.c_uint32), ("gate_all", C.c_uint32), ("gate_bnd", C.c_uint32 * 15)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
        _lib.sim_launch.argtypes = [C.c_int, C.c_int, C.POINTER(KArgs)]
    return _lib


def run(codec, variant, blobs, caps, aux=None, n_out=None, in_misalign=0, out_misalign=0, scratch_bytes=0, scratch_init=None, scratch_out=None,
        mirror=False, gate_bnd=None, gates_open=True):
    """-> (outputs list[bytes], out_len, in_used, status, aux)
    mirror (LZ4 decode variant 60, Lz4V4 MIRROR): a second output buffer at the same address mod 256, which must hold the same bytes in
    every block's decoded range and nothing else; gate_bnd: first blocks of ranges 1.. whose blocks wait at a gate (open, or closed:
    they give up with RCX_ST_GATE once the simulator's clock has passed the limit)"""
    n = len(blobs)
    base, off, lens = B.pack(blobs)
    if in_misalign:
        base = np.concatenate([np.zeros(in_misalign, np.uint8), base])
        off = off + np.uint64(in_misalign)
    total, ooff, ocap = B.layout(caps)
    ooff = ooff + np.uint64(out_misalign)
    out = np.full(total + out_misalign + 64, 0xEE, dtype=np.uint8)
    out_len = np.zeros(n, np.uint64); in_used = np.zeros(n, np.uint64); status = np.full(n, -99, np.int32)
    if aux is None:
        aux = np.zeros(max(n, 1), np.uint32)
    scratch = np.zeros(max(scratch_bytes, 8), np.uint8)
    if scratch_init is not None:
        scratch = np.frombuffer(bytes(scratch_init) + bytes(64), dtype=np.uint8).copy()
    p = lambda a: a.ctypes.data
    k = KArgs(p(base), p(off), p(lens), p(out), p(ooff), p(ocap), p(out_len), p(in_used), p(status), p(aux),
              p(n_out) if n_out is not None else None, p(scratch), scratch.size, n)
    out2 = None
    if mirror:
        raw2 = np.full(out.size + 512, 0xEE, dtype=np.uint8)
        sh = (p(out) - p(raw2)) % 256
        out2 = raw2[sh: sh + out.size]
        assert (p(out2) - p(out)) % 256 == 0
        k.out_mirror = p(out2)
        if gate_bnd is not None:
            gw = np.zeros(16, np.uint32); gh = np.zeros(16, np.uint32)
            k.gate, k.gate_host, k.gate_seq, k.gate_ticks = p(gw), p(gh), 7, (100000 if gates_open else 0)    # (shut: the first look at the clock is past the limit)
            if gates_open:
                gh[:] = 7
                k.gate_all = 1
            for i in range(15):
                k.gate_bnd[i] = gate_bnd[i] if i < len(gate_bnd) else 0xffffffff
    rc = lib().sim_launch(codec, variant, C.byref(k))
    assert rc == 0
    if mirror:
        m2 = np.ones(out2.size, bool)
        for o, l, st_ in zip(ooff, out_len, status):
            if st_ == 0:
                assert (out2[int(o): int(o) + int(l)] == out[int(o): int(o) + int(l)]).all(), "the mirror differs from the output"
                m2[int(o): int(o) + int(l)] = False
        for o, c, st_ in zip(ooff, ocap, status):
            if st_ != 0:
                m2[int(o): int(o) + int(c)] = False           # (a failed block's slot holds whatever was produced)
        assert (out2[m2] == 0xEE).all(), "the mirror was written outside a block's decoded bytes"
    if scratch_out is not None:
        scratch_out.append(scratch)
    outs = [bytes(out[int(o):int(o) + int(l)]) for o, l in zip(ooff, out_len)]
    # guard: nothing outside [off, off+cap) may be touched
    mask = np.ones(out.size, bool)
    for o, c in zip(ooff, ocap):
        mask[int(o):int(o) + int(c)] = False
    assert (out[mask] == 0xEE).all(), "kernel wrote outside its output slots"
    return outs, out_len, in_used, status, aux
