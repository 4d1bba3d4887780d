"""CPU suite: the unmodified .hip kernels of every codec on the wave64 simulator vs the oracle
(kernel-logic debugging aid; the real parity tests are the -m gpu ones)."""
import zlib

import numpy as np

import corpus
from rust_compress_amd import _native as N


def _raws(oracle):
    raws = corpus.small_corpus()
    return raws + [oracle.bwt_encode(r)[0] for r in raws]     # BWT outputs: long runs


def test_lz4_encode(oracle):
    import simrun
    raws = _raws(oracle)
    from rust_compress_amd import synth
    raws += [synth.gen(k, 70000, 3).tobytes() for k in ("text", "rand", "runs", "dna4")]   # skip acceleration, back-tracking
    exp = [oracle.lz4_encode_block(r) for r in raws]
    for variant in (0, 2, 1):                    # windowed probe (8 lanes widening / always 64), serial probe chain
        outs, _, _, st, _ = simrun.run(N.LZ4_ENCODE, variant, raws, [oracle.lz4_compression_bound(len(r)) for r in raws],
                                       scratch_bytes=len(raws) * (1 << 19))
        assert not st.any() and outs == exp, variant


def test_mtf_rle_ari_dc(oracle):
    import simrun
    raws = _raws(oracle)
    lens = [len(r) for r in raws]
    enc, _, _, st, _ = simrun.run(N.MTF_ENCODE, 0, raws, lens)
    assert not st.any() and enc == [oracle.mtf_encode(r) for r in raws]
    dec, _, _, st, _ = simrun.run(N.MTF_DECODE, 0, enc, lens)
    assert not st.any() and dec == raws

    enc, _, _, st, _ = simrun.run(N.RLE_ENCODE, 0, raws, [oracle.lib().o_rle_encode_bound(n) for n in lens])
    assert not st.any() and enc == [oracle.rle_encode(r) for r in raws]
    dec, _, _, st, _ = simrun.run(N.RLE_DECODE, 0, enc, lens)
    assert not st.any() and dec == raws
    rng = np.random.default_rng(2)
    arb = [bytes(rng.integers(0, 4, rng.integers(0, 200), dtype=np.uint8) * rng.integers(1, 100)) for _ in range(40)]
    arb += [b"aa", b"aab", b"aa" + bytes(10), b"a", b"aaa\x80b"]
    exp = [oracle.rle_decode(b, cap=5000, raise_on_error=False) for b in arb]
    outs, _, _, st, _ = simrun.run(N.RLE_DECODE, 0, arb, [5000] * len(arb))
    for (eo, es), s, o_ in zip(exp, st, outs):
        assert es == s and (s != 0 or eo == o_)

    for variant in (1, 2, 3):                      # one lane per stream, one wave per stream, a quad of lanes per stream
        enc, _, _, st, _ = simrun.run(N.ARI_BYTE_ENCODE, variant, raws, [2 * n + 16 for n in lens])
        assert not st.any() and enc == [oracle.ari_byte_encode(r) for r in raws], variant
        dec, _, used, st, _ = simrun.run(N.ARI_BYTE_DECODE, variant, [e + b"xyz" for e in enc], lens)
        assert not st.any() and dec == raws and list(used) == [len(e) for e in enc], variant   # stops exactly at the stream end
        # truncated / corrupted streams and short output slots: statuses as the oracle's
        bad = [e[: max(0, len(e) - k)] for e in enc[:6] for k in (1, 3, 5)] + [bytes([255] * 40), b"", b"\x00\x01"]
        caps = [len(r) // 2 + 1 for r in raws[:6] for _ in range(3)] + [100, 10, 10]
        exp = [oracle.ari_byte_decode(b_, cap=c, raise_on_error=False) for b_, c in zip(bad, caps)]
        outs, _, used, st, _ = simrun.run(N.ARI_BYTE_DECODE, variant, bad, caps)
        for i, e in enumerate(exp):
            assert e[-1] == st[i], (variant, i, e[-1], st[i])

    enc, _, _, st, _ = simrun.run(N.DC_ENCODE, 0, raws, [4 * (256 + n) for n in lens])
    assert not st.any() and enc == [oracle.dc_encode(r).tobytes() for r in raws]
    dec, _, _, st, _ = simrun.run(N.DC_DECODE, 0, enc, lens, n_out=np.array(lens, dtype=np.uint64))
    assert not st.any() and dec == raws
    # the same with the coding contexts (dc.rs:40-58): encoder and decoder see the same ones, and the oracle's (dc.rs:268-289)
    encx, _, _, st, _ = simrun.run(N.DC_ENCODE, 1, raws, [4 * (256 + n) + 8 * n for n in lens])
    assert not st.any()
    decx, _, _, st, _ = simrun.run(N.DC_DECODE, 1, enc, [((n + 7) & ~7) + 8 * (len(e) // 4 - 256) for n, e in zip(lens, enc)], n_out=np.array(lens, dtype=np.uint64))
    assert not st.any()
    for r, e, x, y in zip(raws, enc, encx, decx):
        n, k = len(r), len(e) // 4 - 256
        want = [(s_, rk, dl) for s_, rk, dl in oracle.dc_encode(r, with_ctx=True)[1]]
        assert len(x) == 4 * (256 + n) + 8 * k and x[: len(e)] == e
        got = np.frombuffer(x[4 * (256 + n):], dtype="<u4").reshape(-1, 2)
        assert [(int(a) & 255, (int(a) >> 8) & 255, int(b_)) for a, b_ in got] == want
        co = (n + 7) & ~7
        assert y[:n] == r and len(y) == co + 8 * len(oracle.dc_decode(np.frombuffer(e, dtype="<u4"), n, with_ctx=True)[2])
        gotd = np.frombuffer(y[co:], dtype="<u4").reshape(-1, 2)
        assert [(int(a) & 255, (int(a) >> 8) & 255, int(b_)) for a, b_ in gotd] == oracle.dc_decode(np.frombuffer(e, dtype="<u4"), n, with_ctx=True)[2]
    # distance streams cut short, with contexts: the step that finds no distance ends in EOF and must not write its Context past
    # the slot (capacity exactly what rcx.h asks for: coff + 8 * (nwords - 256)); simrun's guard checks every byte outside the slots
    cut, cn = [], []
    for r, e in zip(raws, enc):
        k = len(e) // 4 - 256
        for drop in (1, 2, k // 2, k):
            if 0 < drop <= k:
                cut.append(e[: len(e) - 4 * drop]); cn.append(len(r))
    caps = [((n + 7) & ~7) + 8 * (len(e) // 4 - 256) for n, e in zip(cn, cut)]
    _, _, _, st, _ = simrun.run(N.DC_DECODE, 1, cut, caps, n_out=np.array(cn, dtype=np.uint64))
    want = [_oracle_status(oracle.dc_decode, np.frombuffer(e, dtype="<u4"), n) for e, n in zip(cut, cn)]
    assert list(st) == want and any(want)


def test_dc_encode_lane_per_chunk(oracle):
    """k_dcx_prep + k_dcx_main (blocks of >= 8 KiB over <= 64 symbols, scratch given): the same words as the oracle; the blocks
    that path refuses (short, or a larger alphabet) come from the wave-per-block kernel in the same launch."""
    import simrun
    from rust_compress_amd import synth
    rng = np.random.default_rng(12)
    srcs = [synth.gen("text", 70000, 1).tobytes(), synth.gen("dna4", 20000, 2).tobytes(), synth.gen("runs", 30000, 3).tobytes(),
            synth.gen("words", 9000, 4).tobytes()]
    raws = [oracle.bwt_encode(x)[0] for x in srcs] + srcs[:2]                  # after the BWT (the pipeline's input) and plain
    raws += [b"a" * 10000, b"ab" * 5000, bytes(rng.integers(0, 64, 8192, dtype=np.uint8)), bytes(rng.integers(0, 64, 8193, dtype=np.uint8)),
             bytes(rng.integers(0, 3, 12345, dtype=np.uint8)) + bytes(range(3, 64))]            # one run / two symbols / exactly 64 symbols / late first occurrences
    raws += [bytes(rng.integers(0, 65, 9000, dtype=np.uint8)), synth.gen("rand", 10000, 5).tobytes(), synth.gen("text", 5000, 6).tobytes(), b""]   # refused: 65 symbols, 256, short, empty
    lens = [len(r) for r in raws]
    sc = []
    enc, _, used, st, _ = simrun.run(N.DC_ENCODE, 0, raws, [4 * (256 + n) for n in lens], scratch_bytes=len(raws) * 37632 + 256, scratch_out=sc)
    assert not st.any() and list(used) == lens
    took = [int(sc[0][i * 37632: i * 37632 + 4].view("<u4")[0]) for i in range(len(raws))]           # the slot's first word: 1 = the chunk kernels encoded the block
    assert took == [1] * (len(raws) - 4) + [0] * 4
    for i, r in enumerate(raws):
        assert enc[i] == oracle.dc_encode(r).tobytes(), (i, len(r))


def _oracle_status(fn, *args):
    try:
        fn(*args)
        return 0
    except Exception as e:                       # oracle_py.OracleError
        return e.status


def test_ari_binary_and_proxy_models(oracle):
    """bin::Model and the two SumProxy models, driven as src/entropy/ari/test.rs drives them."""
    import simrun
    raws = [r[:3000] for r in _raws(oracle)] + [bytes(range(256)) * 4, b"\xff" * 500]
    lens = [len(r) for r in raws]
    caps = [2 * n + 16 for n in lens]
    for rate in (1, 3, 5, 9):
        enc, _, _, st, _ = simrun.run(N.ARI_BINARY_ENCODE, rate, raws, caps)
        assert not st.any() and enc == [oracle.ari_binary_encode(r, rate) for r in raws], rate
        dec, _, _, st, _ = simrun.run(N.ARI_BINARY_DECODE, rate, enc, lens)
        assert not st.any() and dec == raws, rate
    enc, _, _, st, _ = simrun.run(N.ARI_PROXY_ENCODE, 0, raws, caps)
    assert not st.any() and enc == [oracle.ari_proxy_encode(r) for r in raws]
    dec, _, _, st, _ = simrun.run(N.ARI_PROXY_DECODE, 0, enc, lens)
    assert not st.any() and dec == raws
    # truncated streams / asking for more bytes than were coded: same status as the oracle
    bad = [e[: max(0, len(e) - k)] for e in enc[:8] for k in (2, 6)] + [b"", bytes([255] * 30)]
    want = [n + 3 for n in lens[:8] for _ in range(2)] + [1, 50]
    _, _, _, st, _ = simrun.run(N.ARI_PROXY_DECODE, 0, bad, want)
    assert list(st) == [_oracle_status(oracle.ari_proxy_decode, b_, w) for b_, w in zip(bad, want)]
    _, _, _, st, _ = simrun.run(N.ARI_BINARY_DECODE, 5, bad, want)
    assert list(st) == [_oracle_status(oracle.ari_binary_decode, b_, 5, w) for b_, w in zip(bad, want)]
    # apm::Bit + apm::Gate: the stretch / gate tables come from the host (libm); a skewed history makes the reference panic
    stretch, gate = oracle.apm_tables()
    tabs = stretch.tobytes() + gate.tobytes()
    apm_in = [r for r in raws if len(r)] + [b"", bytes(300), b"\xff" * 300]
    exp = [oracle.ari_apm_encode(r, raise_on_error=False) for r in apm_in]
    assert any(s for _, s in exp) and any(s == 0 and len(o) > 10 for o, s in exp)
    enc, _, _, st, _ = simrun.run(N.ARI_APM_ENCODE, 0, apm_in, [2 * len(r) + 16 for r in apm_in], scratch_init=tabs)
    assert [int(x) for x in st] == [s for _, s in exp]
    good = [i for i, (_, s) in enumerate(exp) if s == 0]
    assert all(enc[i] == exp[i][0] for i in good)
    dec, _, _, st, _ = simrun.run(N.ARI_APM_DECODE, 0, [enc[i] for i in good], [len(apm_in[i]) for i in good], scratch_init=tabs)
    assert not st.any() and dec == [apm_in[i] for i in good]
    _, _, _, st, _ = simrun.run(N.ARI_APM_DECODE, 0, [enc[i][:-6] for i in good[:6]], [len(apm_in[i]) + 2 for i in good[:6]], scratch_init=tabs)
    assert list(st) == [_oracle_status(oracle.ari_apm_decode, enc[i][:-6], len(apm_in[i]) + 2) for i in good[:6]]
    _, _, _, st, _ = simrun.run(N.ARI_BINARY_ENCODE, 5, raws[:4], [3] * 4)     # short output slots
    assert all(int(x) == N.E_OUTPUT_TOO_SMALL for x in st)


def synth_dna(n):
    from rust_compress_amd import synth
    return synth.gen("dna4", n, 5).tobytes()


def test_zlib_first_pass_sums_its_own_adler():
    """The wave-per-stream decoder sums the Adler-32 of a zlib stream while the bytes leave (window drain, stored blocks, wave-wide
    copies); a wrong sum would only send the stream to the exact second pass -- slower, still right, invisible to a parity test.
    So the first pass ALONE (variants 12 / 10): every valid stream must come back OK from it, at any output alignment."""
    import simrun
    raws = corpus.small_corpus(sizes=(17, 1000, 40000)) + [synth_dna(120000), b"", b"x", b"ab" * 40000, bytes(100000)]
    zs, exp = [], []
    for r in raws:
        for lvl in (0, 1, 6):                                   # level 0: stored blocks (the wave-wide literal copy), > 64 KiB of them
            zs.append(zlib.compress(r, lvl)); exp.append(r)
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        zs.append(c.compress(r) + c.flush()); exp.append(r)
    sb = 12 * (len(zs) + 1) + 256
    for variant, mis in ((12, 0), (12, 5), (10, 11)):
        outs, _, used, st, _ = simrun.run(N.ZLIB_DECODE, variant, zs, [len(e) for e in exp], scratch_bytes=sb, out_misalign=mis)
        assert not st.any() and outs == exp and list(used) == [len(z) for z in zs], (variant, mis)


def test_inflate_zlib_adler(oracle, golden):
    import simrun
    txt = golden("test.txt")
    raws = corpus.small_corpus(sizes=(17, 1000, 40000))
    raws.append(synth_dna(120000))                  # far matches, long distance codes, several dynamic blocks
    zs, exp = [], []
    for r in raws:
        for lvl in (0, 1, 6, 9):
            zs.append(zlib.compress(r, lvl)); exp.append(r)
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        zs.append(c.compress(r) + c.flush()); exp.append(r)
    for i in range(10):
        zs.append(golden("test.z.%d" % i)); exp.append(txt)
    sb = 12 * (len(zs) + 1) + 256                  # variant 0 = wave-per-stream kernel + exact fallback: wants scratch
    for variant in (0, 11, 9, 2, 3, 4, 1):          # ..., speculative pass, lane-per-stream auto (8 streams per wave here), 64, 32, 16, first kernel
        outs, _, used, st, _ = simrun.run(N.ZLIB_DECODE, variant, zs, [len(e) for e in exp], scratch_bytes=sb)
        assert not st.any() and outs == exp and list(used) == [len(z) for z in zs], variant
    raw = [z[2:-4] for z in zs] + [golden("test.z.go")]
    outs, _, _, st, aux = simrun.run(N.INFLATE, 0, raw, [len(e) for e in exp] + [len(txt)], scratch_bytes=sb)
    assert not st.any() and outs == exp + [txt] and aux[len(raw) - 1] == 1
    blobs, caps = corpus.mutate(zs, 300, 2, [50, 3000, 50000])
    ex = [oracle.zlib_decode(b, cap=c, raise_on_error=False) for b, c in zip(blobs, caps)]
    outs, _, used, st, _ = simrun.run(N.ZLIB_DECODE, 0, blobs, caps, scratch_bytes=12 * len(blobs) + 256)
    for i, e in enumerate(ex):
        assert e[-1] == st[i] and e[1] == used[i] and (st[i] != 0 or e[0] == outs[i]), i
    ex = [oracle.inflate(b, cap=c, raise_on_error=False) for b, c in zip(blobs, caps)]
    outs, _, used, st, _ = simrun.run(N.INFLATE, 0, blobs, caps, scratch_bytes=12 * len(blobs) + 256)
    for i, e in enumerate(ex):
        assert e[-1] == st[i] and e[1] == used[i] and e[0] == outs[i], i
    big = raws + [b"x" * 70000]
    _, _, _, _, aux = simrun.run(N.ADLER32, 0, big, [0] * len(big))
    assert list(aux[: len(big)]) == [oracle.adler32(r) for r in big]


def test_bwt_inverse(oracle):
    import simrun
    from rust_compress_amd import synth
    raws = corpus.small_corpus(sizes=(17, 1000, 20000, 70000), with_empty=False)
    raws.append(synth.gen("text", 200000, 5).tobytes())                  # 4 slots per marked node: ~500 chains use the second park array, a few the long list
    Ls, orgs = zip(*[oracle.bwt_encode(r) for r in raws])
    maxn = max(len(r) for r in raws)
    # variant bit 0: walkers park at most 16 bytes, so most chains of the larger blocks take the second chase; bit 1: the forward chase
    # over the scattered jump table instead of the backward walk over place()
    rng = np.random.default_rng(8)
    bad = [(bytes(rng.integers(0, 4, 3000, dtype=np.uint8)), 17), (Ls[2], (orgs[2] + 1) % len(Ls[2])), (b"abc", 3)]     # not a BWT / wrong origin / origin >= n
    # bit 2: one workgroup per block from start to end (the default is three launches: table, chase by (block, slice), rank + copy); bits 4..7: chase geometry
    for variant in (0, 1, 2, 3, 4, 5, 0x10, 0x41, 0x60):
        outs, _, _, st, _ = simrun.run(N.BWT_INVERSE, variant, list(Ls) + [b[0] for b in bad], [len(r) for r in raws] + [len(b[0]) for b in bad],
                                       aux=np.array(list(orgs) + [b[1] for b in bad], dtype=np.uint32),
                                       scratch_bytes=(len(raws) + len(bad)) * (24 * maxn + (4 << 20)) + 256)     # jump table (4n) + parked first-chase bytes (16n + slack) + node records
        assert not st[: len(raws)].any() and outs[: len(raws)] == raws, variant
        assert list(st[len(raws):]) == [_oracle_status(oracle.bwt_decode, *b) for b in bad], variant


def test_bwt_inverse_minimal(oracle):
    """decode_minimal (src/bwt/mod.rs:298-315) as the reference computes it: on real BWT outputs (where it is sometimes not the
    inverse), and on arbitrary (L, origin) pairs, where the LF walk closes cycles shorter than n and the output is periodic."""
    import simrun
    rng = np.random.default_rng(5)
    raws = corpus.small_corpus(sizes=(17, 1000, 20000), with_empty=False) + [b"abracadabra", b"test", b"bab", b"some text", b"banana"]
    pairs = [oracle.bwt_encode(r) for r in raws]
    for n, alpha in ((1, 2), (2, 2), (9, 2), (64, 3), (1000, 4), (5000, 256), (20000, 2), (20000, 256), (40000, 7)):
        L = rng.integers(0, alpha, n, dtype=np.uint8).tobytes()
        pairs += [(L, int(rng.integers(0, n))), (L, n - 1), (L, 0)]
    pairs += [(b"abc", 3), (b"abc", 7), (b"", 0), (b"", 1)]              # origin >= n -> error; n == 0 is Ok only with origin 0
    Ls, orgs = zip(*pairs)
    maxn = max(len(L) for L in Ls)
    wrong = 0
    for variant in (0, 1):                                                # 1: park at most 16 bytes per walker (second chases)
        outs, olen, _, st, _ = simrun.run(N.BWT_INVERSE_MINIMAL, variant, list(Ls), [len(L) for L in Ls], aux=np.array(orgs, dtype=np.uint32),
                                          scratch_bytes=len(Ls) * (24 * maxn + (4 << 20)) + 256)
        for i, (L, og) in enumerate(pairs):
            try:
                exp, est = oracle.bwt_decode(L, og, minimal=True), 0
            except Exception as e:                                        # oracle_py.OracleError
                exp, est = b"", e.status
            assert int(st[i]) == est and (est or outs[i] == exp), (variant, i, len(L), og)
            wrong += i < len(raws) and exp != raws[i]
    assert wrong >= 2                                                     # "test", "bab", "some text": the reference's function is not an inverse there


def test_bwt_forward(oracle):
    """The hand-written suffix sorter (k_bwt.hip + k_bwt_sort.hip: first level from the text, radix levels, LDS local sorts, dense
    passes, list queues) on the simulator: (L, origin) against the oracle for every key layout a batch can take."""
    import simrun
    from rust_compress_amd import synth
    def check(raws):
        total = sum(len(r) for r in raws)
        outs, olen, _, st, aux = simrun.run(N.BWT_FORWARD, 0, raws, [len(r) for r in raws], scratch_bytes=64 * total + (8 << 20))
        assert not st.any()
        for r, L, og in zip(raws, outs, aux):
            eL, eo = oracle.bwt_encode(r)
            assert L == eL and (not r or int(og) == eo), len(r)
    check(corpus.small_corpus(sizes=(17, 1000, 9000)))                                   # text, runs, rand, dna: 9-bit plain-byte keys (all 256 bytes occur)
    check([synth.gen("dna4", 6000, 1).tobytes(), b"ab" * 2500, bytes(3000), b"abracadabra" * 300])     # 1-4 symbols: 16+ per key, long shared prefixes
    check([synth.gen("text", 30000, 2).tobytes(), synth.gen("words", 5000, 3).tobytes(), b"", b"x"])   # ~6 bits per symbol, groups of every size class
    rng = np.random.default_rng(12)
    for alpha in (2, 5, 11, 20, 100):                                                    # 14 / 12 / 15 / 12 / 9 symbols per key (the count follows the alphabet)
        body = rng.integers(0, alpha, 5000, dtype=np.uint8)
        check([bytes(np.concatenate([body[:1500], body[300:1500], body]) + 3)])


def test_bwt_suffixes_and_inversion_table(oracle):
    """The reference's two public helpers as exports of their own (mod.rs:136-166, 223-239): the suffix array the sorter holds
    and the scattered jump table, word for word the oracle's."""
    import simrun
    from rust_compress_amd import synth
    raws = corpus.small_corpus(sizes=(17, 1000)) + [synth.gen("text", 9000, 2).tobytes(), synth.gen("dna4", 3000, 1).tobytes(),
                                                   b"abracadabra" * 50, bytes(700), b"x", b""]
    total = sum(len(r) for r in raws)
    outs, olen, _, st, aux = simrun.run(N.BWT_SUFFIXES, 0, raws, [4 * len(r) for r in raws], scratch_bytes=64 * total + (8 << 20))
    assert not st.any() and list(olen) == [4 * len(r) for r in raws]
    Ls, orgs = [], []
    for r, sa, og in zip(raws, outs, aux):
        want = oracle.bwt_suffixes(r)
        assert np.array_equal(np.frombuffer(sa, dtype="<u4"), want), len(r)
        eL, eo = oracle.bwt_encode(r)
        assert not r or int(og) == eo
        Ls.append(eL); orgs.append(eo)
    outs, olen, _, st, _ = simrun.run(N.BWT_INVERSION_TABLE, 0, Ls, [4 * len(r) for r in Ls], aux=np.array(orgs, dtype=np.uint32))
    for L, og, t, s_ in zip(Ls, orgs, outs, st):
        if not L:
            assert s_ == 3                                        # input[origin] panics on the empty block (:230)
            continue
        assert s_ == 0 and np.array_equal(np.frombuffer(t, dtype="<u4"), oracle.bwt_inversion_table(L, og))
    # arbitrary (L, origin) pairs, origin out of range, short slots
    rng = np.random.default_rng(5)
    arb = [bytes(rng.integers(0, 5, n, dtype=np.uint8)) for n in (1, 2, 63, 64, 65, 1500, 5000)]
    ao = [int(rng.integers(0, len(a))) for a in arb]
    arb += [b"abc", b"abcd"]; ao += [3, 1]
    caps = [4 * len(a) for a in arb]; caps[-1] = 15
    outs, _, _, st, _ = simrun.run(N.BWT_INVERSION_TABLE, 0, arb, caps, aux=np.array(ao, dtype=np.uint32))
    assert list(st[-2:]) == [3, 2]
    for L, og, t in list(zip(arb, ao, outs))[:-2]:
        assert np.array_equal(np.frombuffer(t, dtype="<u4"), oracle.bwt_inversion_table(L, og))


def _gzip_members(raws):
    """gzip members with every optional header field (RFC 1952), made with Python's gzip/zlib (the checker here:
    the reference crate has no gzip code, see include/rcx.h)."""
    import gzip, struct
    out = []
    for i, r in enumerate(raws):
        plain = gzip.compress(r, compresslevel=(1, 6, 9)[i % 3], mtime=0)
        if i % 4 == 0:
            out.append(plain)
            continue
        body = plain[10:]                                        # deflate stream + CRC32 + ISIZE
        flg, hdr = 0, b""
        if i % 4 in (1, 3):
            flg |= 4; x = b"ab" + struct.pack("<H", 3) + b"xyz"; hdr += struct.pack("<H", len(x)) + x      # FEXTRA
        if i % 4 in (2, 3):
            flg |= 8 | 16; hdr += b"name.txt\0" + b"a comment\0"                                          # FNAME, FCOMMENT
        if i % 4 == 3:
            flg |= 2; hdr += b"\x12\x34"                                                                   # FHCRC (skipped)
        out.append(b"\x1f\x8b\x08" + bytes([flg]) + plain[4:10] + hdr + body)
    return out


def test_crc32_gzip(golden):
    import gzip
    import simrun
    raws = corpus.small_corpus(sizes=(17, 63, 64, 65, 1000, 40000)) + [golden("test.txt"), b"x" * 70000]
    _, _, _, st, aux = simrun.run(N.CRC32, 0, raws, [0] * len(raws))
    assert not st.any() and list(aux[: len(raws)]) == [zlib.crc32(r) for r in raws]
    gz = _gzip_members(raws)
    for g, r in zip(gz, raws):
        assert gzip.decompress(g) == r                           # the members are valid for the independent decoder
    outs, _, used, st, _ = simrun.run(N.GZIP_DECODE, 0, [g + b"tail" for g in gz], [len(r) for r in raws],
                                      scratch_bytes=48 * len(gz) + 256)
    assert not st.any() and outs == raws and list(used) == [len(g) for g in gz]
    # framing errors
    g0 = gz[4]
    bad = [b"\x1f\x8c" + g0[2:], g0[:2] + b"\x07" + g0[3:], g0[:3] + b"\x20" + g0[4:], g0[:-8] + bytes(4) + g0[-4:],
           g0[:-4] + bytes(4), g0[:-3], g0[:5], b""]
    exp = [N.E_GZIP_MAGIC, N.E_GZIP_METHOD, N.E_GZIP_FLAGS, N.E_GZIP_CRC, N.E_GZIP_ISIZE, N.E_EOF, N.E_EOF, N.E_EOF]
    _, _, _, st, _ = simrun.run(N.GZIP_DECODE, 0, bad, [len(raws[4])] * len(bad), scratch_bytes=48 * len(bad) + 256)
    assert list(st) == exp


def test_zlib_mirror_and_gates(oracle):
    """The wave-per-stream inflate kernel with the caller's page-locked buffer as a second destination (k_inflate3<.., MIRROR>,
    rcx_api.hip): valid streams come back OK from the FIRST pass with the second buffer holding exactly their decoded bytes (stored
    blocks and long matches leave by the wave-wide copies, the rest through the window's drain in whole 256-byte lines), at three output
    alignments; behind open gates the same; behind a gate that stays shut a stream gives up with the internal status that sends the
    batch round again."""
    import simrun
    raws = corpus.small_corpus(sizes=(17, 1000, 40000)) + [synth_dna(70000), b"", b"x", b"ab" * 30000, bytes(70000)]
    zs, exp = [], []
    for r in raws:
        for lvl in (0, 1, 6):
            zs.append(zlib.compress(r, lvl)); exp.append(r)
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        zs.append(c.compress(r) + c.flush()); exp.append(r)
    n = len(zs)
    sb = 13 * (n + 1) + 512
    for mis in (0, 5, 250):
        outs, _, used, st, _ = simrun.run(N.ZLIB_DECODE, 12, zs, [len(e) + 3 for e in exp], scratch_bytes=sb, out_misalign=mis, mirror=True)
        assert not st.any() and outs == exp and list(used) == [len(z) for z in zs], mis
    outs, _, _, st, _ = simrun.run(N.ZLIB_DECODE, 12, zs, [len(e) for e in exp], scratch_bytes=sb, mirror=True, gate_bnd=[2, n // 2], gates_open=True)
    assert not st.any() and outs == exp
    outs, _, _, st, _ = simrun.run(N.INFLATE, 12, [z[2:-4] for z in zs], [len(e) for e in exp], scratch_bytes=sb, mirror=True, gate_bnd=[n // 2], gates_open=False)
    assert not st[: n // 2].any() and outs[: n // 2] == exp[: n // 2] and (st[n // 2:] == 0x7ff00003).all()
    # corrupted streams with the mirror and both passes (variant 0): the reference's statuses; what the first pass handed back is marked
    blobs, caps = corpus.mutate(zs[:40], 120, 2, [50, 3000, 50000])
    ex = [oracle.zlib_decode(b, cap=c, raise_on_error=False) for b, c in zip(blobs, caps)]
    keep = []
    outs, _, used, st, _ = simrun.run(N.ZLIB_DECODE, 0, blobs, caps, scratch_bytes=13 * len(blobs) + 512, mirror=True, scratch_out=keep)
    for i, e in enumerate(ex):
        assert e[-1] == st[i] and e[1] == used[i] and (st[i] != 0 or e[0] == outs[i]), i
    marks = keep[0][(12 * len(blobs) + 256 + 63) & ~63:]
    nfb = int(np.frombuffer(marks[:4].tobytes(), np.uint32)[0])
    assert nfb == int(marks[64: 64 + len(blobs)].sum()) and nfb >= sum(1 for e in ex if e[-1] != 0)
