"""Struct-of-arrays batch packing shared by the host API, the tests and the bench."""
import numpy as np


def pack(blobs, align=16):
    """list of bytes-like -> (base uint8[], off uint64[n], len uint64[n]); each blob starts `align`-aligned."""
    n = len(blobs)
    lens = np.array([len(b) for b in blobs], dtype=np.uint64)
    padded = (lens + np.uint64(align - 1)) // np.uint64(align) * np.uint64(align)
    off = np.zeros(n, dtype=np.uint64)
    if n > 1:
        off[1:] = np.cumsum(padded)[:-1]
    total = int(padded.sum()) if n else 0
    base = np.zeros(max(total, align), dtype=np.uint8)
    for i, b in enumerate(blobs):
        if lens[i]:
            base[int(off[i]):int(off[i]) + int(lens[i])] = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    return base, off, lens


def layout(caps, align=16):
    """output capacities -> (total bytes, off uint64[n], cap uint64[n])"""
    caps = np.asarray(caps, dtype=np.uint64)
    padded = (caps + np.uint64(align - 1)) // np.uint64(align) * np.uint64(align)
    off = np.zeros(len(caps), dtype=np.uint64)
    if len(caps) > 1:
        off[1:] = np.cumsum(padded)[:-1]
    total = int(padded.sum()) if len(caps) else 0
    return max(total, align), off, caps


def unpack(base, off, lens):
    return [bytes(base[int(o):int(o) + int(l)]) for o, l in zip(off, lens)]
