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
