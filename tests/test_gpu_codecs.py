"""GPU parity through the C-ABI vs the oracle, bit-exact, for every codec besides LZ4:
inflate / zlib / Adler-32, BWT forward + inverse, MTF, DC, adaptive byte range coder, RLE."""
import zlib

import numpy as np
import pytest

import corpus
from rust_compress_amd import _native as N

pytestmark = pytest.mark.gpu


def _raws(oracle):
    raws = corpus.small_corpus(sizes=(17, 1000, 20000, 262144))
    return raws + [oracle.bwt_encode(r)[0] for r in raws[:20]]


def test_inflate_zlib_fixtures_and_python_zlib(ctx, oracle, golden):
    txt = golden("test.txt")
    raws = corpus.small_corpus(sizes=(17, 1000, 40000, 300000))
    zs, exp = [], []
    for r in raws:
        for lvl in (0, 1, 6, 9):
            zs.append(zlib.compress(r, lvl)); exp.append(r)
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        zs.append(c.compress(r) + c.flush()); exp.append(r)
    for i in range(10):
        zs.append(golden("test.z.%d" % i)); exp.append(txt)                 # zlib.rs:151-164
    for variant in N.INFLATE_VARIANTS:          # auto (wave per stream + exact fallback here), lane per stream, wave per stream, v1
        ctx.set_variant(N.ZLIB_DECODE, variant)
        res = ctx.zlib_decode(zs, [len(e) for e in exp]).check()
        assert res.outputs == exp and list(res.in_used) == [len(z) for z in zs], variant
    ctx.set_variant(N.ZLIB_DECODE, 0)
    raw = [z[2:-4] for z in zs] + [golden("test.z.go")]                      # flate.rs:528-542
    res = ctx.inflate(raw, [len(e) for e in exp] + [len(txt)]).check()
    assert res.outputs == exp + [txt] and res.aux[-1] == 1


def test_inflate_malformed_statuses(ctx, oracle, golden):
    zs = [zlib.compress(r, 6) for r in corpus.small_corpus(sizes=(17, 1000, 5000))] + [golden("test.z.5")]
    blobs, caps = corpus.mutate(zs, 1500, 2, [50, 3000, 50000])
    res = ctx.zlib_decode(blobs, caps)
    for i, (b, c) in enumerate(zip(blobs, caps)):
        eo, eu, _, es = oracle.zlib_decode(b, cap=c, raise_on_error=False)
        assert es == res.status[i] and eu == res.in_used[i], (i, es, res.status[i])
        if es == 0:
            assert eo == res.outputs[i]
    res = ctx.inflate(blobs, caps)
    for i, (b, c) in enumerate(zip(blobs, caps)):
        eo, eu, _, es = oracle.inflate(b, cap=c, raise_on_error=False)
        assert es == res.status[i] and eu == res.in_used[i] and eo == res.outputs[i], i


def test_adler32(ctx, oracle):
    raws = corpus.small_corpus(sizes=(17, 1000, 70000, 1 << 20))
    res = ctx.adler32(raws).check()
    assert list(res.aux) == [oracle.adler32(r) for r in raws] == [zlib.adler32(r) for r in raws]


def test_bwt_suffixes_and_inversion_table(ctx, oracle):
    """compute_suffixes / compute_inversion_table (bwt/mod.rs:136-166, 223-239, both `pub`) as exports of their own: the suffix
    array and the jump table word for word the oracle's -- the small corpus and one 256 KiB block each of text / dna4 / runs."""
    from rust_compress_amd import synth, compress
    raws = corpus.small_corpus(sizes=(17, 1000, 20000))
    raws += [synth.gen(k, 262144, 40 + i).tobytes() for i, k in enumerate(("text", "dna4", "runs"))]
    res = ctx.bwt_suffixes(raws).check()
    assert list(res.out_len) == [4 * len(r) for r in raws]
    Ls, orgs = [], []
    for r, sa, og in zip(raws, res.outputs, res.aux):
        assert np.array_equal(np.frombuffer(sa, dtype="<u4"), oracle.bwt_suffixes(r)), len(r)
        eL, eo = oracle.bwt_encode(r)
        assert not r or int(og) == eo
        if r:
            Ls.append(eL); orgs.append(eo)
    tb = ctx.bwt_inversion_table(Ls, orgs).check()
    for L, og, t in zip(Ls, orgs, tb.outputs):
        assert np.array_equal(np.frombuffer(t, dtype="<u4"), oracle.bwt_inversion_table(L, og)), len(L)
    # arbitrary (L, origin), the index panic, a short slot; the table drives the reference's InverseIterator (:266-281)
    rng = np.random.default_rng(8)
    arb = [bytes(rng.integers(0, 7, n, dtype=np.uint8)) for n in (1, 2, 64, 65, 4097, 70001)]
    ao = [int(rng.integers(0, len(a))) for a in arb]
    tb = ctx.bwt_inversion_table(arb + [b"abc", b""], ao + [3, 0])
    assert list(tb.status) == [0] * len(arb) + [3, 3]
    for L, og, t in zip(arb, ao, tb.outputs):
        assert np.array_equal(np.frombuffer(t, dtype="<u4"), oracle.bwt_inversion_table(L, og))
    # the mirrors (compress.py): a caller-provided array is filled in place
    sa = [0] * 11
    assert compress.bwt.compute_suffixes(b"abracadabra", sa) == sa == list(oracle.bwt_suffixes(b"abracadabra"))
    L, og = oracle.bwt_encode(b"abracadabra")
    tab = [0] * 11
    compress.bwt.compute_inversion_table(L, og, tab)
    assert tab == list(oracle.bwt_inversion_table(L, og))
    with pytest.raises(AssertionError):
        compress.bwt.compute_inversion_table(L, og, [0] * 10)


def test_bwt_forward_and_inverse(ctx, oracle):
    from rust_compress_amd import synth
    raws = corpus.small_corpus(sizes=(17, 1000, 20000, 262144))
    raws.append(synth.gen("text", 1 << 20, 9).tobytes())                 # 16 slots per marked node: chains of several 16-byte groups
    fw = ctx.bwt_forward(raws).check()
    for r, L, og in zip(raws, fw.outputs, fw.aux):
        eL, eo = oracle.bwt_encode(r)
        assert L == eL and (not r or og == eo)
    nz = [i for i, r in enumerate(raws) if r]
    for variant in (0, 1, 2, 3, 4, 5, 0x10, 0x41, 0x60):    # bit 0: short parking (second chases), bit 1: the scattered-table kernel, bit 2: one workgroup per block, bits 4..7: chase geometry
        ctx.set_variant(N.BWT_INVERSE, variant)
        inv = ctx.bwt_inverse([fw.outputs[i] for i in nz], [int(fw.aux[i]) for i in nz]).check()
        assert inv.outputs == [raws[i] for i in nz]
    ctx.set_variant(N.BWT_INVERSE, 0)
    bad = ctx.bwt_inverse([b"abc"], [3])                    # origin >= n: bwt/mod.rs:230 panics
    assert bad.status[0] == 3


def test_bwt_inverse_minimal(ctx, oracle):
    """decode_minimal (src/bwt/mod.rs:298-315), what bwt::Decoder runs with extra_mem = false: the reference's answer, which is
    the text for some inputs ("abracadabra", its only test :549-551) and not for others ("test": SURVEY.md A.4)."""
    from rust_compress_amd import synth
    rng = np.random.default_rng(11)
    raws = [b"abracadabra", b"banana", b"test", b"bab", b"some text"] + corpus.small_corpus(sizes=(17, 1000, 20000), with_empty=False)
    raws += [synth.gen("text", 100000, 3).tobytes(), synth.gen("dna4", 70000, 4).tobytes()]         # stride 7 and 5: parked chains, reversed copies
    pairs = [oracle.bwt_encode(r) for r in raws]
    for n, alpha in ((1, 2), (2, 2), (9, 2), (64, 3), (1000, 4), (20000, 256), (40000, 7), (60000, 2)):   # not a BWT: short cycles, periodic output
        L = rng.integers(0, alpha, n, dtype=np.uint8).tobytes()
        pairs += [(L, int(rng.integers(0, n))), (L, n - 1), (L, 0)]
    pairs += [(b"abc", 3), (b"abc", 7), (b"", 0), (b"", 1)]              # origin >= n is an error; n == 0 is Ok only with origin 0 (:300-302)
    Ls, orgs = zip(*pairs)
    exp = []
    for L, og in pairs:
        try:
            exp.append((0, oracle.bwt_decode(L, og, minimal=True)))
        except Exception as e:                                            # oracle_py.OracleError
            exp.append((e.status, b""))
    for variant in (0, 1):                                                # 1: park 16 bytes per walker at most, the rest by second chases
        ctx.set_variant(N.BWT_INVERSE_MINIMAL, variant)
        res = ctx.bwt_inverse_minimal(list(Ls), list(orgs))
        for i, (est, eout) in enumerate(exp):
            assert int(res.status[i]) == est and (est or res.outputs[i] == eout), (variant, i, len(Ls[i]), orgs[i])
    ctx.set_variant(N.BWT_INVERSE_MINIMAL, 0)
    assert exp[0][1] == raws[0] and exp[2][1] != raws[2]                  # the reference's function: right on its own test, wrong on "test"


def test_bwt_forward_key_layouts(ctx, oracle):
    """The first sort key adapts to the batch (alphabet compaction, symbols per key) and big batches are sorted 1024 blocks
    at a time: every layout must give the reference's (L, origin)."""
    from rust_compress_amd import synth
    rng = np.random.default_rng(5)
    def check(raws):
        fw = ctx.bwt_forward(raws).check()
        for r, L, og in zip(raws, fw.outputs, fw.aux):
            eL, eo = oracle.bwt_encode(r)
            assert L == eL and (not r or og == eo)
    check([synth.gen("dna4", 50000, 1).tobytes(), b"ab" * 9000, bytes(5000), b"abracadabra" * 700])      # 1-4 symbols: 16+ per key
    check([synth.gen("text", 60000, 2).tobytes(), synth.gen("words", 30000, 3).tobytes()])                # ~6 bits: 8 per key
    skew = bytes(rng.choice(256, 40000, p=np.r_[[0.7], np.full(255, 0.3 / 255)]).astype(np.uint8))
    check([skew + bytes(range(256))])                                                                     # all 256 bytes occur, low entropy: plain bytes
    check([bytes(rng.integers(0, 255, 40000, dtype=np.uint8))])                                           # 255 symbols, high entropy: plain bytes
    check([bytes(rng.choice(200, 30000, p=np.r_[[0.5], np.full(199, 0.5 / 199)]).astype(np.uint8))])      # 200 symbols, 8-bit codes
    # the symbol count of the first key follows the alphabet (k_bwt.hip: the count whose bits fill the LSD passes): 2-3 symbols -> 14 per
    # key, 4-7 -> 12, 8-15 -> 15, 16-31 -> 12, 32-63 -> 10, 64-127 -> 9; each with long repeats (ties carried into the doubling rounds)
    for alpha in (2, 3, 5, 11, 20, 50, 100):
        body = rng.integers(0, alpha, 30000, dtype=np.uint8)
        rep = np.concatenate([body[:9000], body[2000:9000], body[:9000]])                                  # LCPs of thousands
        check([bytes(body + 7), bytes(rep + 1)])
    small = [synth.gen(("text", "runs", "dna4", "rand")[i % 4], int(rng.integers(1, 400)), 100 + i).tobytes() for i in range(2500)]
    ctx.set_variant(N.BWT_FORWARD, 700)                                                                   # at most 700 blocks per sorting pass: four passes
    check(small)
    ctx.set_variant(N.BWT_FORWARD, 0)
    check(small)


def test_mtf_dc_ari_rle(ctx, oracle):
    raws = _raws(oracle)
    lens = [len(r) for r in raws]
    e = ctx.mtf_encode(raws).check()
    assert e.outputs == [oracle.mtf_encode(r) for r in raws]
    assert ctx.mtf_decode(e.outputs).check().outputs == raws
    e = ctx.rle_encode(raws).check()
    assert e.outputs == [oracle.rle_encode(r) for r in raws]
    assert ctx.rle_decode(e.outputs, lens).check().outputs == raws
    want = [oracle.ari_byte_encode(r) for r in raws]
    for variant in (0, 1, 2, 3):                       # auto, a lane / a wave / a quad of lanes per stream
        ctx.set_variant(N.ARI_BYTE_ENCODE, variant); ctx.set_variant(N.ARI_BYTE_DECODE, variant)
        e = ctx.ari_byte_encode(raws).check()
        assert e.outputs == want, variant
        d = ctx.ari_byte_decode([x + b"tail" for x in e.outputs], lens).check()
        assert d.outputs == raws and list(d.in_used) == [len(x) for x in e.outputs], variant
        # mutated streams and short slots: status, bytes and consumed count as the oracle's
        blobs, caps = corpus.mutate(want[:12], 300, 5 + variant, [10, 300, 30000])
        res = ctx.ari_byte_decode(blobs, caps)
        for i, (b_, c_) in enumerate(zip(blobs, caps)):
            eo = oracle.ari_byte_decode(b_, cap=c_, raise_on_error=False)
            assert eo[-1] == res.status[i], (variant, i, eo[-1], res.status[i])
            if eo[-1] == 0:
                assert eo[0] == res.outputs[i]
    ctx.set_variant(N.ARI_BYTE_ENCODE, 0); ctx.set_variant(N.ARI_BYTE_DECODE, 0)
    e = ctx.dc_encode(raws).check()
    assert e.outputs == [oracle.dc_encode(r).tobytes() for r in raws]
    assert ctx.dc_decode(e.outputs, lens).check().outputs == raws
    # dc::Context (dc.rs:40-58) from the device: next to every distance the encoder yields (:88-103) and for every call of
    # the decoder's distance callback (:199-229); both == the oracle's, and == each other wherever the reference's test says so
    ex = ctx.dc_encode_ctx(raws).check()
    dx = ctx.dc_decode_ctx(e.outputs, lens).check()
    for r, w, x, y in zip(raws, e.outputs, ex.outputs, dx.outputs):
        n, k = len(r), len(w) // 4 - 256
        assert len(x) == 4 * (256 + n) + 8 * k and x[: len(w)] == w
        got = np.frombuffer(x[4 * (256 + n):], dtype="<u4").reshape(-1, 2)
        assert [(int(a) & 255, (int(a) >> 8) & 255, int(b_)) for a, b_ in got] == oracle.dc_encode(r, with_ctx=True)[1]
        co = (n + 7) & ~7
        want = oracle.dc_decode(np.frombuffer(w, dtype="<u4"), n, with_ctx=True)[2]
        assert y[:n] == r and len(y) == co + 8 * len(want)
        gotd = np.frombuffer(y[co:], dtype="<u4").reshape(-1, 2)
        assert [(int(a) & 255, (int(a) >> 8) & 255, int(b_)) for a, b_ in gotd] == want


def test_dc_encode_lane_per_chunk(ctx, oracle):
    """The default DC encoder for blocks of >= 8 KiB over <= 64 symbols (k_dcx_prep + k_dcx_main, a lane per chunk) and the
    wave-per-block kernel (variant 1; also what takes the blocks the first refuses): both == the oracle."""
    from rust_compress_amd import synth
    rng = np.random.default_rng(12)
    srcs = [synth.gen("text", 262144, 1).tobytes(), synth.gen("dna4", 100000, 2).tobytes(), synth.gen("runs", 70000, 3).tobytes(),
            synth.gen("words", 9000, 4).tobytes(), synth.gen("text", 1 << 20, 7).tobytes()]
    raws = [oracle.bwt_encode(x)[0] for x in srcs] + srcs[:2]
    raws += [b"a" * 10000, b"ab" * 5000, bytes(rng.integers(0, 64, 8192, dtype=np.uint8)), bytes(rng.integers(0, 64, 8193, dtype=np.uint8)),
             bytes(rng.integers(0, 3, 12345, dtype=np.uint8)) + bytes(range(3, 64))]
    raws += [bytes(rng.integers(0, 65, 9000, dtype=np.uint8)), synth.gen("rand", 10000, 5).tobytes(), synth.gen("text", 5000, 6).tobytes(), b""]
    raws += [synth.gen("text", 4189000, 8).tobytes(), synth.gen("runs", 4200000, 9).tobytes()]        # chunks of 64 K positions
    want = [oracle.dc_encode(r).tobytes() for r in raws]
    for variant in (0, 1):
        ctx.set_variant(N.DC_ENCODE, variant)
        e = ctx.dc_encode(raws).check()
        assert e.outputs == want, variant
    ctx.set_variant(N.DC_ENCODE, 0)
    assert ctx.dc_decode(want, [len(r) for r in raws]).check().outputs == raws


def test_ari_binary_and_proxy_models(ctx, oracle):
    """bin::Model (every rate the reference's tests use, test.rs:52-89) and the SumProxy pair (test.rs:91-148)."""
    raws = [r[:20000] for r in corpus.small_corpus()] + [bytes(range(256)) * 8]
    lens = [len(r) for r in raws]
    for rate in (1, 2, 3, 4, 5, 6, 7):
        e = ctx.ari_binary_encode(raws, rate).check()
        assert e.outputs == [oracle.ari_binary_encode(r, rate) for r in raws], rate
        assert ctx.ari_binary_decode(e.outputs, rate, lens).check().outputs == raws
    e = ctx.ari_proxy_encode(raws).check()
    assert e.outputs == [oracle.ari_proxy_encode(r) for r in raws]
    assert ctx.ari_proxy_decode(e.outputs, lens).check().outputs == raws
    res = ctx.ari_proxy_decode([x[:-5] for x in e.outputs], [n + 4 for n in lens])       # runs off the end of the input
    assert all(s == N.E_MALFORMED for s in res.status)
    with pytest.raises(Exception):
        ctx.ari_binary_encode(raws, 0)                                                    # rate out of range
    # apm::Bit + apm::Gate (test.rs:150-182); all-equal bits drive the gate index out of range: the reference panics
    apm_in = [r for r in raws if len(r)] + [b"", bytes(300), b"\xff" * 300]
    exp = [oracle.ari_apm_encode(r, raise_on_error=False) for r in apm_in]
    e = ctx.ari_apm_encode(apm_in)
    assert [int(x) for x in e.status] == [s for _, s in exp] and N.E_MALFORMED in list(e.status)
    good = [i for i, (_, s) in enumerate(exp) if s == 0]
    assert len(good) > 5 and all(e.outputs[i] == exp[i][0] for i in good)
    d = ctx.ari_apm_decode([e.outputs[i] for i in good], [len(apm_in[i]) for i in good]).check()
    assert d.outputs == [apm_in[i] for i in good]


def test_rle_ari_malformed(ctx, oracle):
    rng = np.random.default_rng(2)
    arb = [bytes(rng.integers(0, 4, rng.integers(0, 200), dtype=np.uint8) * rng.integers(1, 100)) for _ in range(200)]
    arb += [b"aa", b"aab", b"aa" + bytes(10), b"a", b"aaa\x80b"]
    res = ctx.rle_decode(arb, [5000] * len(arb))
    for i, b in enumerate(arb):
        eo, es = oracle.rle_decode(b, cap=5000, raise_on_error=False)
        assert es == res.status[i] and (es != 0 or eo == res.outputs[i])
    enc = [oracle.ari_byte_encode(r) for r in corpus.small_corpus(sizes=(17, 1000))]
    blobs, caps = corpus.mutate(enc, 400, 9, [10, 2000])
    res = ctx.ari_byte_decode(blobs, caps)
    for i, (b, c) in enumerate(zip(blobs, caps)):
        eo, eu, es = oracle.ari_byte_decode(b, cap=c, raise_on_error=False)
        assert es == res.status[i], (i, es, res.status[i])
        if es == 0:
            assert eo == res.outputs[i] and eu == res.in_used[i]


def test_inflate_large_and_ragged_streams(ctx):
    """A few long streams (the wave-per-stream decoder's home ground): every level, fixed codes, stored blocks, 1 B .. 3 MiB,
    several dynamic blocks per stream, long matches (runs) and far ones (dna), against Python's zlib."""
    from rust_compress_amd import synth
    rng = np.random.default_rng(11)
    raws = []
    for i, kind in enumerate(("text", "runs", "dna4", "rand", "words", "mix")):
        raws.append(synth.gen(kind, int(rng.integers(1 << 20, 3 << 20)), 100 + i).tobytes())
    raws += [b"", b"a", b"ab" * 70000, bytes(200000), synth.gen("text", 70001, 3).tobytes()]
    zs, exp = [], []
    for i, r in enumerate(raws):
        for lvl in (0, 1, 4, 9):
            zs.append(zlib.compress(r, lvl)); exp.append(r)
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        zs.append(c.compress(r) + c.flush()); exp.append(r)
        c = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_DEFAULT_STRATEGY)            # flush points: empty stored blocks mid-stream
        zs.append(c.compress(r[: len(r) // 2]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(r[len(r) // 2:]) + c.flush()); exp.append(r)
    for variant in (0, 9):
        ctx.set_variant(N.ZLIB_DECODE, variant)
        res = ctx.zlib_decode(zs, [len(e) for e in exp]).check()
        assert res.outputs == exp and [int(u) for u in res.in_used] == [len(z) for z in zs], variant
    ctx.set_variant(N.ZLIB_DECODE, 0)
    # short output slots: the status (and nothing else) must be "output buffer too small", as the lane-per-stream kernel reports it
    res0 = ctx.zlib_decode(zs[:12], [max(0, len(e) - 1) for e in exp[:12]])
    ctx.set_variant(N.ZLIB_DECODE, 9)
    res9 = ctx.zlib_decode(zs[:12], [max(0, len(e) - 1) for e in exp[:12]])
    ctx.set_variant(N.ZLIB_DECODE, 0)
    assert [int(s) for s in res0.status] == [int(s) for s in res9.status]


def test_hip_vs_derived_golden(ctx):
    """The encoder-side kernels against tests/golden/derived directly -- expected bytes that come from neither the device nor
    oracle/*.c but from an independent plain-Python restatement of SURVEY Appendix A (tests/gen_derived_golden.py): LZ4 block
    encode, MTF, dc::encode_simple words + contexts, ByteEncoder, the binary (rate 5) and the SumProxy coders, 36 inputs each."""
    import derived
    recs = list(derived.records())
    raws = [d for _, d in recs]
    assert len(recs) == 36
    lz = ctx.lz4_encode_blocks(raws).check()
    mt = ctx.mtf_encode(raws).check()
    dw = ctx.dc_encode(raws).check()
    dx = ctx.dc_encode_ctx(raws).check()
    ab = ctx.ari_byte_encode(raws).check()
    bn = ctx.ari_binary_encode(raws, 5).check()
    px = ctx.ari_proxy_encode(raws).check()
    for i, (rec, data) in enumerate(recs):
        derived.check(rec, "lz4_encode", lz.outputs[i])
        derived.check(rec, "mtf_encode", mt.outputs[i])
        derived.check(rec, "dc_words", dw.outputs[i])
        n = len(data)
        assert dx.outputs[i][: len(dw.outputs[i])] == dw.outputs[i]
        derived.check(rec, "dc_ctx", dx.outputs[i][4 * (256 + n):])
        derived.check(rec, "ari_byte", ab.outputs[i])
        derived.check(rec, "ari_bin5", bn.outputs[i])
        derived.check(rec, "ari_proxy", px.outputs[i])



def test_inflate_host_path_into_page_locked_output(ctx, oracle, golden):
    """rcx_zlib_decode_batch / rcx_inflate_batch with RCX_MEM_HOST and a PAGE-LOCKED output buffer (rcx_api.hip: the wave-per-stream
    decoder stores what leaves its window straight into the caller's buffer; streams its first pass hands back -- every corrupted or
    unusual one -- are copied out behind the second pass, one by one when they are few, as one span when they are many): the bytes,
    lengths, consumed counts, flags and statuses of the plain copies (rcx_ctx_set_param(ctx, codec, 1)) and of the oracle."""
    import ctypes as C
    import torch
    from rust_compress_amd import batch as B, synth
    L = N.lib()
    rng = np.random.default_rng(77)
    raws = [synth.gen(("text", "runs", "rand", "dna4")[i % 4], int(rng.choice([0, 1, 300, 5000, 16384, 70000])), 500 + i).tobytes() for i in range(400)]
    good = []
    for i, r in enumerate(raws):
        co = zlib.compressobj([1, 6, 9, 6][i % 4], zlib.DEFLATED, 15, 8, zlib.Z_FIXED if i % 5 == 4 else zlib.Z_DEFAULT_STRATEGY)
        good.append(co.compress(r) + co.flush())
    good.append(golden("test.z.5"))
    raws.append(zlib.decompress(good[-1]))
    for few in (True, False):
        blobs, caps = list(good), [len(r) + 9 for r in raws]
        step = 41 if few else 3                                # a few corrupted streams among the good ones / a third of the batch
        for i in range(0, len(blobs), step):
            b = bytearray(blobs[i])
            if len(b) > 12: b[int(rng.integers(2, len(b)))] ^= 0x11
            blobs[i] = bytes(b)
        n = len(blobs)
        for codec, fn, strip in ((N.ZLIB_DECODE, "rcx_zlib_decode_batch", 0), (N.INFLATE, "rcx_inflate_batch", 2)):
            bl = [x[strip:] for x in blobs]
            base, off, lens = B.pack(bl)
            total, ooff, ocap = B.layout(caps)
            inb = torch.from_numpy(base).pin_memory()
            res = {}
            for plain in (True, False):
                outb = torch.full((int(total) + 64,), 0xAA, dtype=torch.uint8).pin_memory()
                out_len, in_used, status, flags = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.full(n, -9, np.int32), np.zeros(n, np.uint32)
                p = lambda a: a.ctypes.data
                b = N.Batch(inb.data_ptr(), p(off), p(lens), outb.data_ptr(), p(ooff), p(ocap), p(out_len), p(in_used), p(status), n, N.MEM_HOST)
                assert L.rcx_ctx_set_param(ctx._h, codec, 1 if plain else 0) == 0
                try:
                    assert getattr(L, fn)(ctx._h, C.byref(b), C.c_void_p(p(flags))) == 0, L.rcx_last_error(ctx._h)
                finally:
                    L.rcx_ctx_set_param(ctx._h, codec, 0)
                got = outb.numpy()
                res[plain] = (status.copy(), out_len.copy(), in_used.copy(), flags.copy(),
                              [bytes(got[int(o): int(o) + int(l)]) for o, l in zip(ooff, out_len)])
            for k in range(4):
                assert (res[True][k] == res[False][k]).all(), (few, codec, k)
            assert res[True][4] == res[False][4], (few, codec)
            dec = oracle.zlib_decode if codec == N.ZLIB_DECODE else oracle.inflate
            for i in range(n):
                eo, eu, _, es = dec(bl[i], cap=caps[i], raise_on_error=False)
                assert es == res[False][0][i] and eu == res[False][2][i], (few, codec, i)
                if es == 0:
                    assert eo == res[False][4][i]


def test_bwt_block_over_the_sorters_limit_gets_its_own_status(ctx):
    """bwt::Encoder::new(w, block_size) takes any usize (bwt/mod.rs:451); the suffix sorter keeps four flag bits beside a suffix index, so a
    block of 2^28 bytes or more cannot be sorted here.  It must cost exactly that block: RCX_E_BWT_BLOCK_TOO_LARGE on it (include/rcx.h),
    the other blocks of the batch transformed as ever."""
    import torch
    import rust_compress_amd as R
    from rust_compress_amd import synth
    dev = torch.device("cuda", 0)
    small = synth.gen("text", 5000, 3).tobytes()
    big = 1 << 28
    inb = torch.zeros(big + 8192 + 64, dtype=torch.uint8, device=dev)
    inb[:5000] = torch.frombuffer(bytearray(small), dtype=torch.uint8).to(dev)
    inb[big + 8192 - 5000: big + 8192] = inb[:5000]
    i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
    # block 0: 5000 bytes; block 1: 2^28 bytes (zeros, never read); block 2: the 5000 bytes again
    db = R.DeviceBatch(inb, i64([0, 5008, big + 8192 - 5000]), i64([5000, big, 5000]),
                       torch.zeros(3 * 8192, dtype=torch.uint8, device=dev), i64([0, 8192, 16384]), i64([8192, 8192, 8192]))
    sc = torch.empty(ctx.scratch_bytes(N.BWT_FORWARD, 3, 8192) + 256, dtype=torch.uint8, device=dev)
    ctx.launch_dev(N.BWT_FORWARD, db, sc)
    torch.cuda.synchronize()
    one = ctx.bwt_forward([small]).check()
    assert db.status.tolist() == [0, N.E_BWT_BLOCK_TOO_LARGE, 0] and db.out_len.tolist() == [5000, 0, 5000]
    got = db.out_base.cpu().numpy()
    assert bytes(got[:5000]) == one.outputs[0] == bytes(got[16384: 16384 + 5000])
    assert int(db.aux[0]) == int(one.aux[0]) == int(db.aux[2])
    assert N.lib().rcx_status_string(N.E_BWT_BLOCK_TOO_LARGE).decode() == "bwt block of 2^28 bytes or more"
