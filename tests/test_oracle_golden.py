"""CPU suite: pins the oracle (oracle/*.c) to the reference's own fixtures and known answers.

Reference-owned vectors: src/data/test.txt <-> test.z.0-9 / test.z.go / test.lz4.1-9 (flate.rs:528-542,
zlib.rs:151-164, lz4.rs:647-659), RLE KATs (rle.rs:320-352).  Independent second source: Python's zlib.
Derived vectors: SURVEY.md Appendix B (independent transliteration of the cited lines).
"""
import hashlib
import zlib

import numpy as np
import pytest

from rust_compress_amd import synth


def test_flate_zlib_fixtures(oracle, golden):
    txt = golden("test.txt")
    assert hashlib.sha256(txt).hexdigest() == "768ec5c6935e81ed3f50bc422f3d259b7dd905ee2d96e4d674c6f1a88aaa733d"
    for i in range(10):
        d = golden("test.z.%d" % i)
        out, used, _ = oracle.zlib_decode(d)
        assert out == txt and used == len(d)
        out, used, _ = oracle.inflate(d[2:-4])          # flate.rs:504-506 fixup
        assert out == txt and used == len(d) - 6
    out, used, flags = oracle.inflate(golden("test.z.go"))
    assert out == txt and flags == 1                     # two empty stored blocks mid-stream (A.1 quirk flag)


def test_lz4_frame_fixtures(oracle, golden):
    txt = golden("test.txt")
    for i in range(1, 10):
        d = golden("test.lz4.%d" % i)
        out, used = oracle.lz4_frame_decode(d)
        assert out == txt and used == len(d) - 4         # content checksum is never read (lz4.rs:384)


def test_rle_kats(oracle):
    enc, dec = oracle.rle_encode, oracle.rle_decode
    assert enc(b"") == b"" and enc(b"a") == b"a" and enc(b"abca123") == b"abca123"      # rle.rs:320-327
    assert enc(bytes([20] * 5 + [15])) == bytes([20, 20, 5 - 2 + 128, 15])
    assert enc(bytes([0, 0])) == bytes([0, 0, 128])
    assert enc(bytes([5] * 129)) == bytes([5, 5, 255])                                   # rle.rs:330-336
    data = bytes([1, 3, 4, 4]) + bytes([100] * (2 + 52 + 128))
    assert enc(data) == bytes([1, 3, 4, 4, 128, 100, 100, 52, 129])
    assert dec(b"") == b"" and dec(b"a") == b"a" and dec(b"abca123") == b"abca123"      # rle.rs:339-345
    assert dec(bytes([20, 20, 131, 15])) == bytes([20] * 5 + [15])
    assert dec(bytes([0, 0, 128])) == bytes([0, 0])
    assert dec(bytes([1, 3, 4, 4, 128, 100, 100, 52, 129])) == data                      # rle.rs:348-352
    assert enc(b"Helloooo world!!").hex() == "48656c6c806f6f8220776f726c64212180"        # rustdoc rle.rs:20-30
    with pytest.raises(oracle.OracleError) as e:
        dec(b"aa" + bytes(10))                           # 10 length bytes -> "Overly long run"
    assert e.value.status == 30
    assert dec(b"aa") == b"aa" and dec(b"aab") == b"a" * (2 + 0x62)   # EOF inside a header flushes what it has


def test_appendix_b_vectors(oracle, golden):
    txt = golden("test.txt")
    o = oracle
    assert o.bwt_encode(b"abracadabra") == (b"rdarcaaaabb", 2)
    assert o.bwt_encode(b"banana") == (b"nnbaaa", 3)
    assert o.bwt_encode(b"some text") == (b"emtostx e", 5)
    L, origin = o.bwt_encode(txt)
    assert origin == 829 and hashlib.sha256(L).hexdigest().startswith("14e3facfe883794b")
    assert o.bwt_stream_encode(b"abracadabra", 4 << 20).hex() == "000040000b000000726461726361616161626202000000"
    s = o.bwt_stream_encode(txt, 1024)
    assert len(s) == 3078 and hashlib.sha256(s).hexdigest().startswith("01f5c9a701d8f08b")
    assert list(o.mtf_encode(b"abracadabra")) == [97, 98, 114, 2, 100, 1, 101, 1, 4, 4, 2]
    w = o.dc_encode(b"teeesst_dc")
    assert {chr(i): int(w[i]) for i in range(256) if w[i] != 10} == {"t": 0, "e": 1, "s": 4, "_": 7, "d": 8, "c": 9}
    assert list(map(int, w[256:])) == [3, 1, 0, 0, 0, 0, 0]
    w = o.dc_encode(b"abracadabra")
    assert list(map(int, w[256:])) == [0, 2, 2, 0, 2, 0, 1, 0, 0, 0, 0]
    w = o.dc_encode(b"aaaa")
    assert list(map(int, w[256:])) == [0] and o.dc_decode(w, 4) == (b"aaaa", 0)   # decoder consumes none (A.6)
    assert o.ari_byte_encode(b"").hex() == "ff00ff0000"
    assert o.ari_byte_encode(b"abracadabra").hex() == "6101aba17aa9d5cc68d39733f600"
    assert o.ari_byte_encode(b"some text").hex() == "72fb93041016a77256f24b6000"
    e = o.ari_byte_encode(txt)
    assert len(e) == 1861 and hashlib.sha256(e).hexdigest().startswith("2589cf8a9f1fd353")
    a, b = o.ari_byte_encode(b"abra"), o.ari_byte_encode(b"cadabra")
    assert (len(a), len(b)) == (8, 11)
    out, used = o.ari_byte_decode(a + b)
    assert out == b"abra" and used == 8                  # test.rs:52-89: the second decoder starts at byte 8
    assert o.ari_byte_decode((a + b)[used:])[0] == b"cadabra"
    assert o.lz4_encode_block(b"").hex() == "00" and o.lz4_encode_block(b"a").hex() == "1061"
    assert o.lz4_encode_block(b"a" * 54).hex() == "1f6101001d506161616161"
    assert o.lz4_encode_block(b"abcd" * 9).hex() == "4f61626364040008506461626364"
    e = o.lz4_encode_block(txt)
    assert len(e) == 2724 and hashlib.sha256(e).hexdigest().startswith("92921c4321ae45b3")
    assert o.lz4_frame_encode(b"test").hex() == "04224d1860500004000080746573740000000000000000"
    e = o.rle_encode(txt)
    assert len(e) == 3084 and hashlib.sha256(e).hexdigest().startswith("370bea93f99d0996")
    assert o.adler32(b"abracadabra") == 0x19F20455 and o.adler32(txt) == 0xFB4FCFA6 == zlib.adler32(txt)


def _corpus():
    out = [b"", b"a", b"ab", b"aaaa", b"abracadabra", b"banana", b"test", b"some text", bytes(range(256)) * 3]
    for i, k in enumerate(("text", "runs", "rand", "dna4")):
        for n in (1, 17, 1000, 20000):
            out.append(synth.gen(k, n, 100 + i).tobytes())
    return out


def test_roundtrip_properties(oracle, golden):
    """the reference's own round-trip tests: lz4.rs:661-726, bwt/mod.rs:528-551, mtf.rs:179-197,
    dc.rs:259-302 (incl. context equality), ari/test.rs:185-212, rle.rs:354-361"""
    o = oracle
    for d in _corpus() + [golden("test.txt")]:
        assert o.lz4_decode_block(o.lz4_encode_block(d), cap=len(d)) == d
        assert o.lz4_frame_decode(o.lz4_frame_encode(d), cap=len(d) + 1)[0] == d
        L, origin = o.bwt_encode(d)
        assert o.bwt_decode(L, origin) == d
        assert sorted(L) == sorted(d)
        if d:
            assert o.bwt_stream_decode(o.bwt_stream_encode(d, 1024)) == d
        assert o.mtf_decode(o.mtf_encode(d)) == d
        w, ectx = o.dc_encode(d, with_ctx=True)
        dec, cons, dctx = o.dc_decode(w, len(d), with_ctx=True)
        assert dec == d
        if len(set(d)) > 1:
            assert cons == len(w) - 256 and dctx == ectx      # dc.rs:268-289 roundtrip_ctx
        e = o.ari_byte_encode(d)
        out, used = o.ari_byte_decode(e, cap=len(d) + 1)
        assert out == d and used == len(e)
        assert o.rle_decode(o.rle_encode(d), cap=len(d) + 1) == d
        for rate in (1, 5):
            e = o.ari_binary_encode(d[:2000], rate)
            assert o.ari_binary_decode(e, rate, len(d[:2000])) == d[:2000]
        e = o.ari_proxy_encode(d[:2000])
        assert o.ari_proxy_decode(e, len(d[:2000])) == d[:2000]
        e, st = o.ari_apm_encode(d[:2000], raise_on_error=False)          # test.rs:150-182 (status 3: the reference panics)
        assert st in (0, 3) and (st or o.ari_apm_decode(e, len(d[:2000])) == d[:2000])
    assert o.bwt_decode(*o.bwt_encode(b"abracadabra"), minimal=True) == b"abracadabra"   # bwt/mod.rs:549-551
    assert o.bwt_decode(*o.bwt_encode(b"test"), minimal=True) != b"test"                 # A.4: decode_minimal is wrong here


def test_suffix_order_is_sentinel_order(oracle):
    for d in _corpus():
        if not d:
            continue
        sa = oracle.bwt_suffixes(d)
        ref = sorted(range(len(d)), key=lambda i: d[i:])
        assert list(map(int, sa)) == ref


def test_inflate_vs_python_zlib(oracle):
    rng = np.random.default_rng(3)
    for i, d in enumerate(_corpus()):
        for level in (0, 1, 6, 9):
            z = zlib.compress(d, level)
            out, used, _ = oracle.zlib_decode(z, cap=len(d) + 8)
            assert out == d and used == len(z)
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)     # BTYPE=1: no reference fixture has one
        z = c.compress(d) + c.flush()
        out, used, _ = oracle.inflate(z, cap=len(d) + 8)
        assert out == d and used == len(z)
    # config 1 plumbing: one 1 MiB raw DEFLATE stream
    d = synth.gen("text", 1 << 20, 42).tobytes()
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    z = c.compress(d) + c.flush()
    out, used, _ = oracle.inflate(z, cap=len(d))
    assert out == d == zlib.decompress(z, -15) and used == len(z)


def test_zlib_header_errors(oracle):
    z = bytearray(zlib.compress(b"hello world hello world"))
    def st(b):
        return oracle.zlib_decode(bytes(b), raise_on_error=False)[-1]
    assert st(z) == 0
    b = bytearray(z); b[0] = 0x79; assert st(b) == 20          # CM != 8
    b = bytearray(zlib.compress(b"x")); b[0] = 0x68; assert st(b) == 21   # CINFO != 7 (valid small window rejected)
    b = bytearray(z); b[1] |= 0x20; assert st(b) == 22          # FDICT
    b = bytearray(z); b[1] ^= 1; assert st(b) == 23             # FCHECK
    b = bytearray(z); b[-1] ^= 1; assert st(b) == 24            # Adler-32
    assert st(z[:-2]) == 1 and st(z[:1]) == 1                   # truncated trailer / header
    assert oracle.inflate(b"\x07", raise_on_error=False)[-1] == 11   # BTYPE=3 "invalid block code"


def test_large_fixture_if_reference_present(oracle):
    import os
    p = "/root/reference/src/data/test.large.z.5"
    if not os.path.exists(p):
        pytest.skip("reference tree not mounted (GPU box)")
    big = open(p, "rb").read()
    out, used, _ = oracle.zlib_decode(big, cap=6100000)
    assert len(out) == 6100000 and used == len(big)
    assert hashlib.sha256(out).hexdigest() == "94d8990947a4b4d878afa2509e1f6b08fb52906fd688204139c6b814e4b97014"
    assert oracle.adler32(out) == 0x82E12107


def test_oracle_vs_derived_golden(oracle):
    """The encoder side of the oracle (no reference-held vector exists for it) against the committed output of a SECOND,
    independent restatement: tests/gen_derived_golden.py, plain Python written from SURVEY Appendix A, which imports nothing
    from oracle/.  36 inputs (six generators x sizes 0 .. 262144) x LZ4 block encode, MTF, dc::encode_simple words and
    contexts, ByteEncoder, the binary and the SumProxy coders."""
    import derived
    n = 0
    for rec, data in derived.records():
        derived.check(rec, "lz4_encode", oracle.lz4_encode_block(data))
        derived.check(rec, "mtf_encode", oracle.mtf_encode(data))
        words, ctx = oracle.dc_encode(data, with_ctx=True)
        derived.check(rec, "dc_words", words.astype("<u4").tobytes())
        derived.check(rec, "dc_ctx", derived.ctx_bytes(ctx))
        derived.check(rec, "ari_byte", oracle.ari_byte_encode(data))
        derived.check(rec, "ari_bin5", oracle.ari_binary_encode(data, 5))
        derived.check(rec, "ari_proxy", oracle.ari_proxy_encode(data))
        n += 1
    assert n == 36

