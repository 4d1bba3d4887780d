"""GPU: the reference crate's OWN tests, re-stated against the host-side mirror of its Reader/Writer surface
(rust_compress_amd.compress).  Each test cites the Rust test it restates."""
import io
import random

import pytest

import corpus
from rust_compress_amd import compress as C

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _use_ctx(ctx):
    C.set_context(ctx)


def _one_byte_at_a_time(d):
    out = bytearray()
    assert not d.eof()
    while True:
        b = d.read(1)
        if not b:
            break
        out += b
    assert d.eof()
    return bytes(out)


def _random_lengths(d):
    out = bytearray()
    while True:
        b = d.read(1 + random.randrange(40))
        if not b:
            break
        out += b
    return bytes(out)


def test_lz4_decode_fixtures(golden):                      # lz4.rs:647-659 decode
    ref = golden("test.txt")
    for i in range(1, 10):
        assert C.lz4.Decoder(io.BytesIO(golden("test.lz4.%d" % i))).read_to_end() == ref


def test_lz4_raw_encode_block(golden):                     # lz4.rs:661-672
    data = golden("test.txt")
    enc, dec = bytearray(), bytearray()
    C.lz4.encode_block(data, enc)
    C.lz4.decode_block(enc, dec)
    assert bytes(dec) == data


def test_lz4_streaming(golden):                            # lz4.rs:674-706
    assert _one_byte_at_a_time(C.lz4.Decoder(io.BytesIO(golden("test.lz4.1")))) == golden("test.txt")
    assert _random_lengths(C.lz4.Decoder(io.BytesIO(golden("test.lz4.1")))) == golden("test.txt")


def test_lz4_frame_roundtrips(golden, oracle):             # lz4.rs:708-726 some_roundtrips
    for data in (b"test", b"", golden("test.txt"), corpus.small_corpus()[-1] * 20):
        w = io.BytesIO()
        e = C.lz4.Encoder(w)
        e.write(data)
        e.finish()
        assert w.getvalue() == oracle.lz4_frame_encode(data)
        assert C.lz4.Decoder(io.BytesIO(w.getvalue())).read_to_end() == data
    assert C.lz4.compression_bound(0x7E000001) is None and C.lz4.compression_bound(100) == 120
    with pytest.raises(C.InvalidInput):
        C.lz4.Decoder(io.BytesIO(b"\x00\x01\x02\x03rest")).read(1)          # bad magic, lz4.rs:366
    with pytest.raises(C.UnexpectedEof):
        C.lz4.Decoder(io.BytesIO(golden("test.lz4.1")[:100])).read(1)


def test_flate_fixtures_and_streaming(golden):             # flate.rs:528-582
    ref = golden("test.txt")
    fix = lambda b: b[2:-4]                                 # flate.rs:504-506
    for i in range(10):
        assert C.flate.Decoder(io.BytesIO(fix(golden("test.z.%d" % i)))).read_to_end() == ref
    assert C.flate.Decoder(io.BytesIO(golden("test.z.go"))).read_to_end() == ref
    assert _one_byte_at_a_time(C.flate.Decoder(io.BytesIO(fix(golden("test.z.1"))))) == ref
    assert _random_lengths(C.flate.Decoder(io.BytesIO(fix(golden("test.z.1"))))) == ref
    with pytest.raises(C.InvalidInput) as e:
        C.flate.Decoder(io.BytesIO(b"\x07")).read(1)
    assert str(e.value) == "invalid block code"             # flate.rs:58


def test_zlib_fixtures_and_errors(golden):                 # zlib.rs:151-203
    ref = golden("test.txt")
    for i in range(10):
        assert C.zlib.Decoder(io.BytesIO(golden("test.z.%d" % i))).read_to_end() == ref
    assert _one_byte_at_a_time(C.zlib.Decoder(io.BytesIO(golden("test.z.1")))) == ref
    bad = bytearray(golden("test.z.1")); bad[-1] ^= 1
    with pytest.raises(C.InvalidInput) as e:
        C.zlib.Decoder(io.BytesIO(bytes(bad))).read_to_end()
    assert str(e.value) == "invalid checksum on zlib stream"   # zlib.rs:111-114
    a = C.Adler32(); a.feed(b"abra"); a.feed(b"cadabra")
    assert a.result() == 0x19F20455


def test_bwt_mtf_dc_roundtrips(golden, oracle):            # bwt/mod.rs:528-551, mtf.rs:179-197, dc.rs:259-302
    for data in (b"abracadabra", golden("test.txt")):
        w = io.BytesIO()
        e = C.bwt.Encoder(w, 1024)
        e.write(data)
        e.finish()
        assert w.getvalue() == oracle.bwt_stream_encode(data, 1024)
        assert C.bwt.Decoder(io.BytesIO(w.getvalue()), True).read_to_end() == data
        L, origin = C.bwt.encode_simple(data)
        assert (L, origin) == oracle.bwt_encode(data) and C.bwt.decode_simple(L, origin) == data
        w = io.BytesIO()
        m = C.bwt.mtf.Encoder(w); m.write(data); m.finish()
        assert w.getvalue() == oracle.mtf_encode(data)
        assert C.bwt.mtf.Decoder(io.BytesIO(w.getvalue())).read_to_end() == data
    for data in (b"teeesst_dc", b"", golden("test.txt")):
        d = C.bwt.dc.encode_simple(data)
        assert d == list(map(int, oracle.dc_encode(data)))
        assert C.bwt.dc.decode_simple(len(data), d) == data
    assert C.bwt.encode_simple(b"abracadabra") == (b"rdarcaaaabb", 2)          # doctest bwt/mod.rs:26-43
    with pytest.raises(C.UnexpectedEof):
        C.bwt.Decoder(io.BytesIO(b"\x00\x04")).read(1)


def test_ari_roundtrips(golden, oracle):                   # ari/test.rs:8-20, 52-89, 185-212
    for data in (b"abracadabra", b"", golden("test.txt")):
        w = io.BytesIO()
        e = C.entropy.ari.ByteEncoder(w); e.write(data); e.finish()
        assert w.getvalue() == oracle.ari_byte_encode(data)
        assert C.entropy.ari.ByteDecoder(io.BytesIO(w.getvalue())).read_to_end() == data
    w = io.BytesIO()                                        # roundtrip_term: two terminated streams back to back
    for part in (b"abra", b"cadabra"):
        e = C.entropy.ari.ByteEncoder(w); e.write(part); e.finish()
    d1 = C.entropy.ari.ByteDecoder(io.BytesIO(w.getvalue()))
    assert d1.read_to_end() == b"abra"
    rest = d1.finish()
    assert C.entropy.ari.ByteDecoder(rest).read_to_end() == b"cadabra"


def test_rle_known_answers():                              # rle.rs:320-361
    def enc(b):
        w = io.BytesIO(); e = C.rle.Encoder(w); e.write_all(b); e.finish(); return w.getvalue()
    dec = lambda b: C.rle.Decoder(io.BytesIO(b)).read_to_end()
    assert enc(b"") == b"" and enc(b"a") == b"a" and enc(b"abca123") == b"abca123"
    assert enc(bytes([20] * 5 + [15])) == bytes([20, 20, 131, 15]) and enc(bytes([0, 0])) == bytes([0, 0, 128])
    assert enc(bytes([5] * 129)) == bytes([5, 5, 255])
    data = bytes([1, 3, 4, 4]) + bytes([100] * 182)
    assert enc(data) == bytes([1, 3, 4, 4, 128, 100, 100, 52, 129]) and dec(enc(data)) == data
    assert dec(bytes([20, 20, 131, 15])) == bytes([20] * 5 + [15]) and dec(bytes([0, 0, 128])) == bytes([0, 0])
    rng = random.Random(7)
    for _ in range(20):
        buf = bytes(rng.randrange(256) for _ in range(13579))
        assert dec(enc(buf)) == buf
    with pytest.raises(C.CompressError) as e:
        dec(b"aa" + bytes(10))
    assert str(e.value) == "Overly long run"


def test_readers_are_left_exactly_after_the_stream(golden, oracle):
    """The reference's decoders never read past their stream (flate.rs:250-260 byte-wise reads, ari/mod.rs:289-292 finish,
    ari/test.rs:52-89): whatever follows a stream must still be readable from the same reader afterwards."""
    import zlib as pyz
    txt = golden("test.txt")
    tail = b"TAIL-BYTES-" + bytes(range(40))
    # zlib: the Adler-32 trailer belongs to the stream, the tail does not (zlib.rs:99-127)
    r = io.BytesIO(pyz.compress(txt, 6) + tail)
    d = C.zlib.Decoder(r)
    assert d.read_to_end() == txt and d.unwrap().read(-1) == tail
    # raw DEFLATE followed by bytes (what zlib::Decoder relies on to find its trailer)
    raw = pyz.compress(txt, 9)[2:-4]
    d = C.flate.Decoder(io.BytesIO(raw + tail))
    assert d.read_to_end() == txt and d.r.read(5) == tail[:5] and d.r.read(-1) == tail[5:]
    # three Ari streams back to back, each decoder picking up the previous one's reader
    w = io.BytesIO()
    parts = [b"abra", txt[:1500], b""]
    for part in parts:
        e = C.entropy.ari.ByteEncoder(w); e.write(part); e.finish()
    r = io.BytesIO(w.getvalue() + tail)
    for part in parts:
        d = C.entropy.ari.ByteDecoder(r)
        assert d.read_to_end() == part
        r = d.finish()
    assert r.read(-1) == tail
    # an LZ4 frame ends at its end mark (the content checksum is never read, lz4.rs:384)
    d = C.lz4.Decoder(io.BytesIO(golden("test.lz4.1")[:-4] + tail))       # fixture = frame + 4-byte content checksum
    assert d.read_to_end() == txt and d.r.read(-1) == tail
    # two gzip members then a tail: the members decode, the tail is not a member -> error, exactly like a second header check
    import gzip as pyg
    g = pyg.compress(txt[:3000], mtime=0) + pyg.compress(txt[3000:], mtime=0)
    assert C.gzip.Decoder(io.BytesIO(g)).read_to_end() == txt


def test_bwt_decoder_extra_mem_flag(golden, oracle):
    """`extra_mem = false` selects the reference's decode_minimal (bwt/mod.rs:298-315, :397-399), which is not an inverse in
    general: the decoder returns what the reference's returns, block for block."""
    import struct
    txt = golden("test.txt")
    w = io.BytesIO(); e = C.bwt.Encoder(w, 1 << 10); e.write(txt); e.finish()
    assert C.bwt.Decoder(io.BytesIO(w.getvalue()), extra_mem=True).read_to_end() == txt
    want = b"".join(oracle.bwt_decode(*oracle.bwt_encode(txt[i:i + 1024]), minimal=True) for i in range(0, len(txt), 1024))
    assert C.bwt.Decoder(io.BytesIO(w.getvalue()), extra_mem=False).read_to_end() == want
    w = io.BytesIO(); e = C.bwt.Encoder(w, 64); e.write(b"abracadabra"); e.finish()            # the reference's own case, :549-551
    assert C.bwt.Decoder(io.BytesIO(w.getvalue()), extra_mem=False).read_to_end() == b"abracadabra"
    empty = struct.pack("<III", 16, 0, 0)                                                       # an empty block: :230 panics, :300-302 does not
    assert C.bwt.Decoder(io.BytesIO(empty), extra_mem=False).read_to_end() == b""
    with pytest.raises(C.Malformed):
        C.bwt.Decoder(io.BytesIO(empty), extra_mem=True).read_to_end()


def test_bwt_encoder_flush_makes_the_reference_block_boundary(golden, oracle):
    """bwt/mod.rs:511-518: flush() encodes the pending partial block as a block of its own; a stream flushed mid-way has that
    boundary (and still decodes to the text)."""
    import struct
    txt = golden("test.txt")[:3000]
    w = io.BytesIO(); e = C.bwt.Encoder(w, 1024); e.write(txt[:1500]); e.flush(); e.write(txt[1500:]); e.finish()
    want = struct.pack("<I", 1024)
    for part in (txt[:1024], txt[1024:1500], txt[1500:2524], txt[2524:]):                      # the reference's blocks for this call sequence
        L, origin = oracle.bwt_encode(part)
        want += struct.pack("<I", len(part)) + L + struct.pack("<I", origin)
    assert w.getvalue() == want
    assert C.bwt.Decoder(io.BytesIO(w.getvalue()), extra_mem=True).read_to_end() == txt


def _ref_rle_stream(calls):
    """The reference's streaming rle::Encoder (rle.rs:82-143) restated as a state machine (TEST INFRASTRUCTURE): `calls` is a
    list of ("w", bytes) / ("f",) ; returns what its writer receives, finish() included."""
    out = bytearray()
    st = {"byte": 0, "reps": 0, "in_run": False}

    def flush():
        if st["reps"] == 1:
            out.append(st["byte"])
        elif st["reps"] > 1:
            out.extend([st["byte"], st["byte"]])
            v = st["reps"] - 2
            while True:
                x = v & 0x7F
                v >>= 7
                if v == 0:
                    out.append(x | 0x80)
                    break
                out.append(x)

    for c in calls + [("f",)]:
        if c[0] == "f":
            flush()
            continue
        buf = c[1]
        if not st["in_run"] and buf:
            st["byte"], st["reps"], st["in_run"] = buf[0], 1, True
        for b in buf[1:]:
            if st["byte"] == b:
                st["reps"] += 1
            else:
                flush()
                st["reps"], st["byte"] = 1, b
    return bytes(out)


def test_rle_encoder_streaming_writes_like_the_reference():
    """rle.rs:100-143: several write() calls and flushes in between come out as the reference writes them, oddities included
    (a later write loses its first byte; a flush inside a run writes the run again when it ends)."""
    rng = random.Random(5)
    for trial in range(30):
        calls = []
        for _ in range(rng.randrange(1, 6)):
            n = rng.randrange(0, 40)
            calls.append(("w", bytes(rng.choice(b"ab") if rng.random() < 0.8 else rng.randrange(256) for _ in range(n))))
            if rng.random() < 0.4:
                calls.append(("f",))
        w = io.BytesIO()
        e = C.rle.Encoder(w)
        for c in calls:
            e.write(c[1]) if c[0] == "w" else e.flush()
        e.finish()
        assert w.getvalue() == _ref_rle_stream(calls), (trial, calls)


def test_decode_many_is_the_single_decoders_together(golden, oracle):
    """decode_many (round 6): the frames / streams / members of many readers through ONE batch call -- every decoder must then read what it
    would have read alone (the reference's fixtures lz4.rs:647-659, flate.rs:528-582, zlib.rs:151-203 all at once), leave its reader
    exactly behind its stream, and the first bad stream must raise what its own Decoder raises."""
    import zlib as pz
    ref = golden("test.txt")
    tail = b"TAIL-BYTES"
    # lz4: the nine reference frames + a frame of stored blocks, each with bytes behind it
    w = io.BytesIO(); e = C.lz4.Encoder(w); e.write(ref * 3); e.finish()
    frames = [golden("test.lz4.%d" % i) for i in range(1, 10)] + [w.getvalue()]
    readers = [io.BytesIO(f + tail) for f in frames]
    decs = C.lz4.decode_many(readers)
    assert [d.read_to_end() for d in decs] == [ref] * 9 + [ref * 3]
    for d, f in zip(decs, frames):                       # (the reference never reads a frame's content checksum: it stays in the reader)
        alone = C.lz4.Decoder(io.BytesIO(f + tail))
        alone.read_to_end()
        left = d.finish().read(-1)
        assert left == alone.finish().read(-1) and left.endswith(tail)
    assert _one_byte_at_a_time(C.lz4.decode_many([io.BytesIO(frames[0])])[0]) == ref
    # zlib: the ten reference members; flate: raw streams of several sizes (one of them larger than the first slot)
    zs = [golden("test.z.%d" % i) for i in range(10)]
    decs = C.zlib.decode_many([io.BytesIO(z + tail) for z in zs])
    assert [d.read_to_end() for d in decs] == [ref] * 10 and all(d.finish().read(-1) == tail for d in decs)
    raws = [ref, ref * 40, b"", b"x" * 300000, bytes(range(256)) * 9]
    def raw_deflate(r, lvl):
        c = pz.compressobj(lvl, pz.DEFLATED, -15)
        return c.compress(r) + c.flush()
    streams = [raw_deflate(r, (1, 6, 9)[i % 3]) for i, r in enumerate(raws)]
    decs = C.flate.decode_many([io.BytesIO(s + tail) for s in streams])
    assert [d.read_to_end() for d in decs] == raws and all(d.finish().read(-1) == tail for d in decs)
    assert [d.read_to_end() for d in C.flate.decode_many([])] == []
    # a bad member among good ones raises like its own decoder
    bad = bytearray(zs[3]); bad[-1] ^= 0x55
    with pytest.raises(C.CompressError) as one:
        C.zlib.Decoder(io.BytesIO(bytes(bad))).read_to_end()
    with pytest.raises(C.CompressError) as many:
        C.zlib.decode_many([io.BytesIO(zs[0]), io.BytesIO(bytes(bad)), io.BytesIO(zs[1])])
    assert type(many.value) is type(one.value) and many.value.status == one.value.status
