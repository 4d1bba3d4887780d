"""GPU tests of the gzip / CRC-32 extension through the C-ABI.  The reference crate has no gzip
code, so the checkers are Python's zlib.crc32 and gzip module (independent implementations): CRC values and decoded
bytes must match them exactly, framing errors must map to the documented statuses, and whatever corrupted member
Python's decoder accepts or rejects, the GPU path must never return wrong bytes with status 0."""
import gzip
import io
import zlib

import numpy as np
import pytest

import corpus
from rust_compress_amd import _native as N
from test_wavesim_codecs import _gzip_members

pytestmark = pytest.mark.gpu


def test_crc32_matches_zlib(ctx, golden):
    rng = np.random.default_rng(3)
    raws = corpus.small_corpus(sizes=(17, 63, 64, 65, 1000, 262144)) + [golden("test.txt"), b"", b"a", b"x" * 1000003,
                                                                       rng.integers(0, 256, 4 << 20, dtype=np.uint8).tobytes()]
    res = ctx.crc32(raws).check()
    assert [int(x) for x in res.aux] == [zlib.crc32(r) for r in raws]


def test_gzip_members_match_python_gzip(ctx, golden):
    raws = corpus.small_corpus(sizes=(17, 1000, 40000, 300000)) + [golden("test.txt"), b""]
    gz = _gzip_members(raws)
    res = ctx.gzip_decode([g + b"next" for g in gz], [len(r) for r in raws]).check()
    assert res.outputs == raws and [int(u) for u in res.in_used] == [len(g) for g in gz]
    for g, r in zip(gz, raws):
        assert gzip.decompress(g) == r


def test_gzip_errors_and_fuzz(ctx):
    raws = corpus.small_corpus(sizes=(1000, 5000), with_empty=False)[:6]
    gz = _gzip_members(raws)
    g0 = gz[0]
    bad = [b"\x1f\x8c" + g0[2:], g0[:2] + b"\x07" + g0[3:], g0[:3] + b"\x20" + g0[4:], g0[:-8] + bytes(4) + g0[-4:],
           g0[:-4] + bytes(4), g0[:-3], g0[:5], b""]
    exp = [N.E_GZIP_MAGIC, N.E_GZIP_METHOD, N.E_GZIP_FLAGS, N.E_GZIP_CRC, N.E_GZIP_ISIZE, N.E_EOF, N.E_EOF, N.E_EOF]
    res = ctx.gzip_decode(bad, [len(raws[0])] * len(bad))
    assert [int(s) for s in res.status] == exp
    # random corruption: status 0 only with the bytes Python's gzip produces for the same member
    blobs, caps = corpus.mutate(gz, 600, 4, [200000])
    res = ctx.gzip_decode(blobs, caps)
    ok = 0
    for b_, s, out, used in zip(blobs, res.status, res.outputs, res.in_used):
        if int(s) == 0:
            ok += 1
            assert gzip.GzipFile(fileobj=io.BytesIO(bytes(b_[: int(used)]))).read() == out
    assert ok < len(blobs)                                       # the CRC catches nearly every payload corruption


def test_gzip_reader_api(ctx):
    from rust_compress_amd import compress as cz
    cz.set_context(ctx)
    parts = [b"hello, ", b"gzip " * 1000, b"world"]
    stream = b"".join(gzip.compress(p, mtime=0) for p in parts)              # concatenated members, RFC 1952 2.2
    d = cz.gzip.Decoder(io.BytesIO(stream))
    assert d.read_to_end() == b"".join(parts) and d.members == 3 and d.consumed == len(stream)
    c = cz.Crc32(); c.feed(b"abc"); c.feed(b"def")
    assert c.result() == zlib.crc32(b"abcdef")


def test_gzip_host_path_into_page_locked_output(ctx, golden):
    """rcx_gzip_decode_batch with RCX_MEM_HOST and a PAGE-LOCKED output buffer (rcx_api.hip: the inflate kernel stores the decoded bytes
    in the caller's buffer itself; members its first pass hands back are copied out behind the second pass): statuses, lengths,
    consumed counts and bytes of the plain copies (rcx_ctx_set_param(ctx, RCX_GZIP_DECODE, 1)), and Python's gzip for the good ones."""
    import ctypes as C
    import torch
    from rust_compress_amd import batch as B, synth
    L = N.lib()
    rng = np.random.default_rng(12)
    raws = [synth.gen(("text", "runs", "rand")[i % 3], int(rng.choice([0, 1, 900, 16384, 50000])), 900 + i).tobytes() for i in range(300)] + [golden("test.txt")]
    gz = _gzip_members(raws)
    for step in (47, 2):                                        # a few corrupted members / every other one
        blobs = list(gz)
        for i in range(0, len(blobs), step):
            b = bytearray(blobs[i])
            if len(b) > 30: b[int(rng.integers(12, len(b) - 8))] ^= 0x21
            blobs[i] = bytes(b)
        n = len(blobs)
        base, off, lens = B.pack(blobs)
        total, ooff, ocap = B.layout([len(r) + 5 for r in raws])
        inb = torch.from_numpy(base).pin_memory()
        res = {}
        for plain in (True, False):
            outb = torch.full((int(total) + 64,), 0xAA, dtype=torch.uint8).pin_memory()
            out_len, in_used, status, flags = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.full(n, -9, np.int32), np.zeros(n, np.uint32)
            p = lambda a: a.ctypes.data
            b = N.Batch(inb.data_ptr(), p(off), p(lens), outb.data_ptr(), p(ooff), p(ocap), p(out_len), p(in_used), p(status), n, N.MEM_HOST)
            assert L.rcx_ctx_set_param(ctx._h, N.GZIP_DECODE, 1 if plain else 0) == 0
            try:
                assert L.rcx_gzip_decode_batch(ctx._h, C.byref(b), C.c_void_p(p(flags))) == 0, L.rcx_last_error(ctx._h)
            finally:
                L.rcx_ctx_set_param(ctx._h, N.GZIP_DECODE, 0)
            got = outb.numpy()
            res[plain] = (status.copy(), out_len.copy(), in_used.copy(), [bytes(got[int(o): int(o) + int(l)]) for o, l in zip(ooff, out_len)])
        for k in range(3):
            assert (res[True][k] == res[False][k]).all(), (step, k)
        assert res[True][3] == res[False][3], step
        good = 0
        for i in range(n):
            if res[False][0][i] == 0:
                good += 1
                assert res[False][3][i] == gzip.decompress(blobs[i])
        assert good >= n - (n + step - 1) // step
