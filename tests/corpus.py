"""Shared test inputs (synthetic distributions + edge cases + the reference's text fixture)."""
import os

import numpy as np

from rust_compress_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def txt():
    return open(os.path.join(GOLDEN, "test.txt"), "rb").read()


def small_corpus(sizes=(17, 1000, 20000), with_empty=True):
    raws = [b"a", b"ab", b"aaaa", b"abracadabra", b"banana", b"teeesst_dc", txt(), bytes(range(256)) * 3,
            b"\0" * 1000, b"ab" * 700 + b"c" * 300]
    if with_empty:
        raws = [b""] + raws
    for i, k in enumerate(("text", "runs", "rand", "dna4")):
        for n in sizes:
            raws.append(synth.gen(k, n, 100 + i).tobytes())
    return raws


def mutate(blobs, count, seed, caps_choices):
    """corrupt / truncate / extend / randomise compressed blobs -> (blobs, caps)"""
    rng = np.random.default_rng(seed)
    out, caps = [], []
    for it in range(count):
        b = bytearray(blobs[(it * 7) % len(blobs)])
        mode = it % 5
        if mode == 0 and len(b):
            for _ in range(rng.integers(1, 4)):
                b[rng.integers(0, len(b))] = rng.integers(0, 256)
        elif mode == 1:
            b = b[: rng.integers(0, len(b) + 1)]
        elif mode == 2:
            b = b + bytes(rng.integers(0, 256, rng.integers(1, 40), dtype=np.uint8))
        elif mode == 3:
            b = bytearray(rng.integers(0, 256, rng.integers(0, 300), dtype=np.uint8).tobytes())
        out.append(bytes(b))
        caps.append(int(rng.choice(caps_choices)))
    return out, caps



def lz4_stream(seqs, tail):
    """an LZ4 block from (literal bytes, match length, offset) triples and the last sequence's literals (lz4.rs:67-140's format)"""
    out = bytearray()

    def ext(v):
        while v >= 255:
            out.append(255); v -= 255
        out.append(v)
    for lit, m, off in seqs:
        L, M = len(lit), m - 4
        out.append((min(L, 15) << 4) | min(M, 15))
        if L >= 15: ext(L - 15)
        out += lit
        out += bytes((off & 255, off >> 8))
        if M >= 15: ext(M - 15)
    L = len(tail)
    out.append(min(L, 15) << 4)
    if L >= 15: ext(L - 15)
    out += tail
    return bytes(out)


def lz4_edge_streams(oracle, count, seed, max_out=180000):
    """-> (blocks, their decoded bytes by the oracle): hand-built streams whose length extensions sit on both sides of every
    boundary a decoder has -- literal runs and matches of 14..16, 269..271, 524..526 and 1000+ bytes among short tokens"""
    import numpy as np
    rng = np.random.default_rng(seed)
    edge_L = [0, 0, 0, 1, 2, 5, 12, 13, 14, 15, 16, 17, 31, 32, 33, 254, 255, 268, 269, 270, 271, 272, 300, 524, 525, 526, 1000, 4100]
    edge_M = [4, 5, 8, 16, 17, 18, 19, 20, 32, 33, 64, 65, 272, 273, 274, 275, 276, 528, 529, 530, 2000]
    blobs, raws = [], []
    for it in range(count):
        seqs, produced = [], 0
        nseq = int(rng.integers(1, 1500))
        for k in range(nseq):
            rare = rng.random() < 0.08
            L = int(rng.choice(edge_L)) if rare else int(rng.choice([0, 0, 0, 0, 1, 2, 3, 6]))
            if produced == 0 and L == 0: L = 1
            lit = rng.integers(0, 256, L, dtype=np.uint8).tobytes()
            produced += L
            M = int(rng.choice(edge_M)) if rng.random() < 0.08 else int(rng.integers(4, 19))
            off = int(rng.integers(1, min(produced, 65535) + 1))
            if rng.random() < 0.1: off = min(produced, int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33])))
            seqs.append((lit, M, off)); produced += M
            if produced > max_out: break
        tail = rng.integers(0, 256, int(rng.choice([0, 1, 5, 12, 15, 19, 20, 21, 40, 270, 300])), dtype=np.uint8).tobytes()
        b = lz4_stream(seqs, tail)
        blobs.append(b); raws.append(oracle.lz4_decode_block(b, cap=produced + len(tail)))
        assert len(raws[-1]) == produced + len(tail)
    return blobs, raws


def lz4_run_streams(oracle, seed=5):
    """-> (blocks, their decoded bytes by the oracle): runs -- matches that overlap themselves -- of every offset 1..15 (and 16, 17: not
    runs) with lengths on both sides of the parser's piece boundaries (k_lz4_decode_v8.hip, RCX_RUNSPLIT: 32 / 64 / ... / 255, and the
    256+ that leave the batch), with 0..3 literals between them, back to back and at the start of a block."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lens = [4, 15, 16, 17, 31, 32, 33, 34, 47, 48, 63, 64, 65, 66, 95, 96, 97, 127, 128, 129, 191, 192, 193, 223, 224, 225, 254, 255, 256, 257, 300, 1100]
    blobs, raws = [], []
    for off0 in range(1, 18):
        for variant in range(3):
            seqs, produced = [], 0
            for M in (lens if variant < 2 else list(rng.permutation(lens))):
                L = off0 if not seqs else int(rng.integers(0, 4)) if variant else 0
                lit = rng.integers(0, 256, L, dtype=np.uint8).tobytes()
                produced += L
                off = min(off0 if variant != 1 else int(rng.integers(1, 18)), produced)
                seqs.append((lit, int(M), off)); produced += int(M)
            b = lz4_stream(seqs, rng.integers(0, 256, 7, dtype=np.uint8).tobytes())
            blobs.append(b); raws.append(oracle.lz4_decode_block(b, cap=produced + 7))
            assert len(raws[-1]) == produced + 7
    return blobs, raws
