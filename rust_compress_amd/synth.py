"""Synthetic byte-stream generators, numpy, deterministic per (kind, n, seed).

G-text : Zipf(1.1)-ranked pseudo-words from a fixed 4096-word vocabulary, joined by spaces / newlines
         (LZ4 ratio ~2x, DEFLATE ~3x: "text-like").
G-runs : runs of geometric length (mean 24) over 16 symbols.
G-rand : incompressible bytes.
G-dna4 : uniform over ACGT.
G-mix  : block i uses kind (text, runs, rand)[i % 3].
"""
import numpy as np

_VOCAB = None


def _vocab(nwords=4096):
    global _VOCAB
    if _VOCAB is None:
        rng = np.random.default_rng(0x5EED)
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        lw = 1.0 / (np.arange(26) + 2.0)
        lw /= lw.sum()
        lens = np.clip(rng.poisson(4.6, nwords) + 1, 1, 14)
        total = int(lens.sum() + nwords)
        blob = np.empty(total, dtype=np.uint8)
        offs = np.zeros(nwords, dtype=np.int64)
        p = 0
        for i in range(nwords):
            offs[i] = p
            blob[p:p + lens[i]] = letters[rng.choice(26, lens[i], p=lw)]
            blob[p + lens[i]] = 32
            p += lens[i] + 1
        w = 1.0 / (np.arange(nwords) + 1.0) ** 1.1
        w /= w.sum()
        _VOCAB = (blob, offs, lens.astype(np.int64) + 1, np.cumsum(w))
    return _VOCAB


def g_text(n, seed):
    blob, offs, lens, cdf = _vocab()
    rng = np.random.default_rng(seed)
    nw = n // 4 + 16
    ranks = np.searchsorted(cdf, rng.random(nw), side="right").clip(0, len(offs) - 1)
    wl = lens[ranks]
    ends = np.cumsum(wl)
    total = int(ends[-1])
    starts = ends - wl
    idx = np.repeat(offs[ranks] - starts, wl) + np.arange(total)
    out = blob[idx]
    nl = rng.random(nw) < (1.0 / 12.0)
    out[ends[nl] - 1] = 10
    assert total >= n
    return out[:n].copy()


def g_runs(n, seed):
    rng = np.random.default_rng(seed)
    nr = n // 12 + 16
    rl = rng.geometric(1.0 / 24.0, nr)
    sym = (rng.integers(0, 16, nr) * 13 + 65).astype(np.uint8)
    out = np.repeat(sym, rl)
    while out.size < n:
        out = np.concatenate([out, out])
    return out[:n].copy()


def g_rand(n, seed):
    return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8)


def g_dna4(n, seed):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(seed).integers(0, 4, n)]


KINDS = {"text": g_text, "runs": g_runs, "rand": g_rand, "dna4": g_dna4}


def gen(kind, n, seed):
    if kind == "mix":
        return KINDS[("text", "runs", "rand")[seed % 3]](n, seed)
    return KINDS[kind](n, seed)


def gen_blocks(kind, nblocks, block_bytes, base_seed):
    """-> uint8 array [nblocks*block_bytes]; block i = gen(kind, block_bytes, base_seed ^ i)"""
    out = np.empty(nblocks * block_bytes, dtype=np.uint8)
    for i in range(nblocks):
        out[i * block_bytes:(i + 1) * block_bytes] = gen(kind, block_bytes, (base_seed ^ i) if kind != "mix" else (base_seed + i))
    return out
