#!/usr/bin/env python3
"""Second, independent pin for the ENCODER side of the oracle (TEST INFRASTRUCTURE; run by hand, its output is committed).

The reference (rusty-shell/rust-compress 0.2.1) holds golden vectors for DEFLATE / zlib / LZ4 *decode* and RLE only; the bytes
of `lz4::encode_block`, `mtf::Encoder`, `dc::encode_simple` (+ the per-distance `Context`), `ari::table::ByteEncoder` and the
`bin::Model` / `SumProxy` coders are pinned by the algorithm text alone.  `oracle/*.c` restates that text in C; THIS file restates
it a second time, in plain Python, from SURVEY.md Appendix A (A.3, A.5, A.6, A.7) and the driving loops of the reference's own
tests (src/entropy/ari/test.rs:22-50 encode_binary, :91-148 roundtrip_proxy) -- it imports nothing from oracle/, is not
imported by the product, and runs nowhere but here.  It writes

    tests/golden/derived/manifest.json          one record per (input, codec): length + sha256 of the expected bytes
    tests/golden/derived/<input>.<codec>.bin    the expected bytes themselves for the small inputs (<= 1000 bytes)

for 36 inputs: the six synthetic generators x sizes 0, 1, 13, 1000, 70000, 262144.  tests/test_oracle_golden.py compares the C
oracle with these files; tests/test_gpu_codecs.py compares the HIP path with them directly.  Reference lines cited per function.

    python tests/gen_derived_golden.py            # ~10 min of pure Python; rewrites the directory
    python tests/gen_derived_golden.py --self     # only the Appendix-B known answers (seconds)
    python tests/gen_derived_golden.py --bins     # the small inputs' .bin files again, checked against the committed manifest (seconds)
"""
import hashlib
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "derived")
M32 = 0xFFFFFFFF

KINDS = ("text", "words", "runs", "rand", "dna4", "mix")
SIZES = (0, 1, 13, 1000, 70000, 262144)
SMALL = 1000                      # inputs up to this size have their expected bytes committed in full


# ------------------------------------------------------------------------------------------------ LZ4 block encode
def _lz4_len_ext(out, v):
    """length extension bytes for v >= 0: 255-runs, then the remainder (lz4.rs:205-215, 293-304)"""
    while v >= 255:
        out.append(255)
        v -= 255
    out.append(v)


def lz4_encode_block(src):
    """BlockEncoder::encode, src/lz4.rs:226-310, release-profile (wrapping) arithmetic."""
    n = len(src)
    out = bytearray()
    table = {}                                     # 2^17 entries, zero-filled: absent key == 0
    pos = anchor = 0
    step, limit = 1, 128
    BIAS = 0x88888888
    while True:
        if pos + 12 > n:                           # (1) the last literals
            ln = n - anchor
            out.append(min(ln, 15) << 4)
            if ln >= 15:
                _lz4_len_ext(out, ln - 15)
            out += src[anchor:n]
            return bytes(out)
        seq = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16) | (src[pos + 3] << 24)      # (2)
        h = ((seq * 2654435761) & M32) >> 15
        r = (table.get(h, 0) + BIAS) & M32
        table[h] = (pos - BIAS) & M32
        miss = (((pos - r) & M32) >> 16) != 0      # (3)
        if not miss:
            miss = seq != (src[r] | (src[r + 1] << 8) | (src[r + 2] << 16) | (src[r + 3] << 24))
        if miss:
            if pos - anchor > limit:
                limit <<= 1
                step += 1 + (step >> 2)
            pos += step
            continue
        if step > 1:                               # (4) back-track (u32 subtraction that wraps in a release build)
            table[h] = (r - BIAS) & M32
            pos -= step - 1
            step = 1
            continue
        limit = 128                                # (5)
        ln = pos - anchor
        back = pos - r
        pos += 4
        r += 4
        a2 = pos
        while pos < n - 5 and src[pos] == src[r]:
            pos += 1
            r += 1
        ml = pos - a2
        out.append((min(ln, 15) << 4) | min(ml, 15))          # (6)
        if ln >= 15:
            _lz4_len_ext(out, ln - 15)
        out += src[anchor:anchor + ln]
        out += struct.pack("<H", back)
        if ml >= 15:
            _lz4_len_ext(out, ml - 15)
        anchor = pos


# ------------------------------------------------------------------------------------------------ MTF
class MTF:
    """src/bwt/mtf.rs:44-91"""

    def __init__(self, alphabetical=False):
        self.symbols = list(range(256)) if alphabetical else [0] * 256

    def encode(self, sym):
        s = self.symbols
        rank = s.index(sym)                        # the first position holding sym (mtf.rs:63-79: the swap loop stops there)
        if rank:
            del s[rank]
            s.insert(0, sym)
        return rank


def mtf_encode(data):
    """mtf::Encoder over a whole stream: identity start list, a rank per byte (mtf.rs:103-104, 118-124)"""
    m = MTF(alphabetical=True)
    return bytes(m.encode(b) for b in data)


# ------------------------------------------------------------------------------------------------ DC
def dc_encode_simple(data, with_ctx=False):
    """dc::encode + EncodeIterator, src/bwt/dc.rs:62-159.  -> u32 LE words: 256 init values, then the distances in
    position order [, the Context (symbol, last_rank, distance_limit) of every distance]."""
    n = len(data)
    last = [n] * 256
    init = [n] * 256
    dist = [n] * n                                 # n = "no distance here"
    mtf = MTF()
    nu = 0
    for i, sym in enumerate(data):
        base = last[sym]
        last[sym] = i
        if base == n:
            mtf.symbols[nu] = sym
            mtf.encode(sym)
            init[sym] = i
            nu += 1
        else:
            r = mtf.encode(sym)
            if r > 0:
                dist[base] = i - base - r - 1
    for rank, sym in enumerate(mtf.symbols[:nu]):
        dist[last[sym]] = n - last[sym] - rank - 1
    words = list(init)
    ctx = []
    pos = list(init)
    last_active = 0
    for i, sym in enumerate(data):                 # the iterator: dc.rs:88-104
        d = dist[i]
        if d == n:
            continue
        rank = last_active - pos[sym]
        assert 0 <= rank < 256
        last_active = i + 1
        pos[sym] = i + 1 + d
        words.append(d)
        ctx.append((sym, rank, n - i))
    raw = struct.pack("<%dI" % len(words), *words)
    return (raw, ctx) if with_ctx else raw


def dc_ctx_bytes(ctx):
    """contexts as the C-ABI lays them out (include/rcx.h: symbol u8, last_rank u8, 2 pad bytes, distance_limit u32 LE)"""
    return b"".join(struct.pack("<BBHI", s, r, 0, lim) for s, r, lim in ctx)


# ------------------------------------------------------------------------------------------------ range coder
class RangeCoder:
    """RangeEncoder + Encoder glue, src/entropy/ari/mod.rs:67-150, 208-237"""

    def __init__(self):
        self.low, self.hai, self.out = 0, M32, bytearray()

    def encode(self, total, frm, to):
        rng = ((self.hai - self.low) & M32) // total
        lo = (self.low + rng * frm) & M32
        hi = (self.low + rng * to) & M32
        while True:
            if (lo ^ hi) & 0xFF000000:
                if ((hi - lo) & M32) > (1 << 14):
                    break
                lim = hi & 0xFF000000
                if ((hi - lim) & M32) >= ((lim - lo) & M32):
                    lo = lim
                else:
                    hi = (lim - 1) & M32
            self.out.append(lo >> 24)
            lo = (lo << 8) & M32
            hi = (hi << 8) & M32
        self.low, self.hai = lo, hi

    def finish(self):
        self.out += struct.pack(">I", self.low)
        return bytes(self.out)


class Table:
    """table::Model, src/entropy/ari/table.rs:20-122"""

    def __init__(self, nsym, cut):
        self.f, self.total, self.cut = [1] * nsym, nsym, cut
        while self.total >= cut:
            self._down()

    def _down(self):
        self.f = [(x + 1) >> 1 for x in self.f]
        self.total = sum(self.f)

    def update(self, v, add_log, add_const):
        add = (self.total >> add_log) + add_const
        self.f[v] = (self.f[v] + add) & 0xFFFF
        self.total += add
        if self.total >= self.cut:
            self._down()

    def range(self, v):
        lo = sum(self.f[:v])
        return lo, lo + self.f[v]


def ari_byte_encode(data):
    """table::ByteEncoder, src/entropy/ari/table.rs:185-224: 257 symbols, cut 4096, update(v, 10, 1), EOF = 256 at finish"""
    rc = RangeCoder()
    t = Table(257, (1 << 14) >> 2)
    for b in data:
        lo, hi = t.range(b)
        rc.encode(t.total, lo, hi)
        t.update(b, 10, 1)
    lo, hi = t.range(256)
    rc.encode(t.total, lo, hi)
    return rc.finish()


class Bin:
    """bin::Model, src/entropy/ari/bin.rs:17-82"""

    def __init__(self, total, rate):
        self.zero, self.total, self.rate = total >> 1, total, rate

    def update(self, bit):
        if bit:
            self.zero -= self.zero >> self.rate
        else:
            self.zero += (self.total - self.zero) >> self.rate


def ari_binary_encode(data, rate):
    """encode_binary, src/entropy/ari/test.rs:22-35 with the model of roundtrip_binary (:37-39): bin::Model::new_flat(2048, rate)"""
    rc = RangeCoder()
    m = Bin((1 << 14) >> 3, rate)
    for b in data:
        for i in range(8):
            bit = (b >> i) & 1
            if bit:
                rc.encode(m.total, m.zero, m.total)
            else:
                rc.encode(m.total, 0, m.zero)
            m.update(bit)
    return rc.finish()


def ari_proxy_encode(data):
    """the encoder half of roundtrip_proxy, src/entropy/ari/test.rs:91-122: high nibble through table::SumProxy(2, t0, 1, t1, 0),
    the four low bits through bin::SumProxy(1, b0, 1, b1, 1)  (table.rs:127-180, bin.rs:112-167)"""
    thr = (1 << 14) >> 3
    t0, t1 = Table(16, thr), Table(16, thr)
    b0, b1 = Bin(thr, 3), Bin(thr, 5)
    rc = RangeCoder()
    for byte in data:
        high = byte >> 4
        lo0, hi0 = t0.range(high)
        lo1, hi1 = t1.range(high)
        rc.encode((2 * t0.total + t1.total) >> 0, (2 * lo0 + lo1) >> 0, (2 * hi0 + hi1) >> 0)
        t0.update(high, 10, 1)
        t1.update(high, 5, 1)
        for i in range(4):
            bit = (byte >> i) & 1
            zero = (b0.zero + b1.zero) >> 1
            total = (b0.total + b1.total) >> 1
            if bit:
                rc.encode(total, zero, total)
            else:
                rc.encode(total, 0, zero)
            b0.update(bit)
            b1.update(bit)
    return rc.finish()


CODECS = {
    "lz4_encode": lz4_encode_block,
    "mtf_encode": mtf_encode,
    "dc_words": dc_encode_simple,
    "dc_ctx": lambda d: dc_ctx_bytes(dc_encode_simple(d, with_ctx=True)[1]),
    "ari_byte": ari_byte_encode,
    "ari_bin5": lambda d: ari_binary_encode(d, 5),
    "ari_proxy": ari_proxy_encode,
}


# ------------------------------------------------------------------------------------------------ self check + driver
def self_check():
    """SURVEY Appendix B's derived known answers (an earlier, separate transliteration): this one must land on the same bytes"""
    H = bytes.fromhex
    assert lz4_encode_block(b"") == H("00") and lz4_encode_block(b"a") == H("1061")
    assert lz4_encode_block(b"a" * 54) == H("1f6101001d506161616161")
    assert lz4_encode_block(b"abcd" * 9) == H("4f61626364040008506461626364")
    assert list(mtf_encode(b"abracadabra")) == [97, 98, 114, 2, 100, 1, 101, 1, 4, 4, 2]
    w = struct.unpack("<263I", dc_encode_simple(b"teeesst_dc"))
    assert [w[ord(c)] for c in "tes_dc"] == [0, 1, 4, 7, 8, 9] and sorted(set(w[:256])) == [0, 1, 4, 7, 8, 9, 10] and list(w[256:]) == [3, 1, 0, 0, 0, 0, 0]
    w = struct.unpack("<%dI" % (256 + 11), dc_encode_simple(b"abracadabra"))
    assert list(w[256:]) == [0, 2, 2, 0, 2, 0, 1, 0, 0, 0, 0]
    assert list(struct.unpack("<257I", dc_encode_simple(b"aaaa"))[256:]) == [0]
    assert ari_byte_encode(b"") == H("ff00ff0000")
    assert ari_byte_encode(b"abracadabra") == H("6101aba17aa9d5cc68d39733f600")
    assert ari_byte_encode(b"some text") == H("72fb93041016a77256f24b6000")
    txt = os.path.join(ROOT, "tests", "golden", "test.txt")
    if os.path.exists(txt):
        t = open(txt, "rb").read()
        e = lz4_encode_block(t)
        assert len(e) == 2724 and hashlib.sha256(e).hexdigest().startswith("92921c4321ae45b3")
        a = ari_byte_encode(t)
        assert len(a) == 1861 and hashlib.sha256(a).hexdigest().startswith("2589cf8a9f1fd353")
    print("Appendix-B known answers: ok")


def inputs():
    sys.path.insert(0, ROOT)
    from rust_compress_amd import synth          # the synthetic generators (numpy, deterministic per (kind, n, seed)) -- not the oracle
    for kind in KINDS:
        for n in SIZES:
            seed = 0xD0 + len(kind) + n % 251
            yield "%s_%d" % (kind, n), kind, n, seed, synth.gen(kind, n, seed).tobytes()


def bins_only():
    """the expected bytes of the small inputs once more (seconds), each checked against the committed manifest before it is written"""
    man = json.load(open(os.path.join(OUT, "manifest.json")))
    recs = {r["input"]: r for r in man["records"]}
    wrote = 0
    for name, kind, n, seed, data in inputs():
        if n > SMALL:
            continue
        assert hashlib.sha256(data).hexdigest() == recs[name]["input_sha256"], name
        for cname, fn in CODECS.items():
            out = fn(data)
            e = recs[name]["expect"][cname]
            assert len(out) == e["len"] and hashlib.sha256(out).hexdigest() == e["sha256"], (name, cname)
            with open(os.path.join(OUT, "%s.%s.bin" % (name, cname)), "wb") as fh:
                fh.write(out)
            wrote += 1
    print("wrote %d vectors to %s" % (wrote, OUT))


def main():
    self_check()
    if "--self" in sys.argv:
        return
    if "--bins" in sys.argv:
        return bins_only()
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        os.remove(os.path.join(OUT, f))
    man = {"what": "expected encoder-side bytes from tests/gen_derived_golden.py (a plain-Python transliteration of SURVEY Appendix A; independent of oracle/*.c)",
           "codecs": sorted(CODECS), "records": []}
    for name, kind, n, seed, data in inputs():
        rec = {"input": name, "kind": kind, "n": n, "seed": seed, "input_sha256": hashlib.sha256(data).hexdigest(), "expect": {}}
        for cname, fn in CODECS.items():
            out = fn(data)
            rec["expect"][cname] = {"len": len(out), "sha256": hashlib.sha256(out).hexdigest()}
            if n <= SMALL:
                with open(os.path.join(OUT, "%s.%s.bin" % (name, cname)), "wb") as fh:
                    fh.write(out)
        man["records"].append(rec)
        print(name, {k: v["len"] for k, v in rec["expect"].items()}, flush=True)
    with open(os.path.join(OUT, "manifest.json"), "w") as fh:
        json.dump(man, fh, indent=1)
    print("wrote %d records to %s" % (len(man["records"]), OUT))


if __name__ == "__main__":
    main()
