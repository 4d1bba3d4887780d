"""Synthetic byte-stream generators, numpy, deterministic per (kind, n, seed).

G-text : text-like ("LZ4 ratio ~ 2-3.5x"): Zipf(1.1)-ranked pseudo-words from a fixed 4096-word
         vocabulary, 45 % of the tokens drawn from a fixed bank of 1024 multi-word phrases (natural text repeats
         phrases, not just words), 8 % novel alphanumeric tokens (names, numbers: literals), spaces / newlines.
         Measured with the reference's compressor: LZ4 ratio 2.7x, 11.8 bytes per sequence (mean literal run
         1.2, mean match 10.6); zlib -6 ratio 4.5x.
G-words: the same vocabulary with NO phrase structure (independent Zipf words): the harshest text-like case,
         LZ4 ratio 1.87x, 7.5 bytes per sequence.
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


def g_words(n, seed):
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


_PHR = None


def _phrases(nphr=1024):
    """item table = the 4096 words followed by 1024 phrases of 3-8 Zipf words; every item ends with a space"""
    global _PHR
    if _PHR is None:
        blob, offs, lens, cdf = _vocab()
        rng = np.random.default_rng(0xF4A5E)
        nw = len(offs)
        words = [blob[offs[i]:offs[i] + lens[i] - 1].tobytes() for i in range(nw)]
        items = list(words)
        for _ in range(nphr):
            ids = np.searchsorted(cdf, rng.random(int(rng.integers(3, 9)))).clip(0, nw - 1)
            items.append(b" ".join(words[j] for j in ids))
        ib = bytearray()
        io, il = [], []
        for it in items:
            io.append(len(ib)); ib += it + b" "; il.append(len(it) + 1)
        pw = 1.0 / (np.arange(nphr) + 1.0) ** 0.9
        pw /= pw.sum()
        _PHR = (np.frombuffer(bytes(ib), dtype=np.uint8), np.array(io, dtype=np.int64), np.array(il, dtype=np.int64),
                cdf, np.cumsum(pw), nw)
    return _PHR


def g_text(n, seed, p_phrase=0.45, p_novel=0.08):
    blob, offs, ilen, cdf, pcdf, nw = _phrases()
    rng = np.random.default_rng(seed)
    nt = n // 10 + 64                                            # mean item length is ~20 bytes
    u = rng.random(nt)
    ids = np.searchsorted(cdf, rng.random(nt), side="right").clip(0, nw - 1)
    ph = nw + np.searchsorted(pcdf, rng.random(nt), side="right").clip(0, len(pcdf) - 1)
    ids = np.where(u < p_phrase, ph, ids)
    wl = ilen[ids]
    ends = np.cumsum(wl)
    total = int(ends[-1])
    starts = ends - wl
    idx = np.repeat(offs[ids] - starts, wl) + np.arange(total)
    out = blob[idx].copy()
    nov = np.nonzero((u > 1.0 - p_novel) & (ids < nw))[0]             # novel tokens: random alphanumerics
    if nov.size:
        alnum = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)
        ln = wl[nov] - 1
        pos = np.repeat(starts[nov], ln) + (np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln))
        out[pos] = alnum[rng.integers(0, 36, pos.size)]
    nl = rng.random(nt) < (1.0 / 12.0)
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


KINDS = {"text": g_text, "words": g_words, "runs": g_runs, "rand": g_rand, "dna4": g_dna4}


def gen(kind, n, seed):
    if kind == "mix":
        return KINDS[("text", "runs", "rand")[seed % 3]](n, seed)
    return KINDS[kind](n, seed)


def gen_blocks(kind, nblocks, block_bytes, base_seed):
    """-> uint8 array [nblocks*block_bytes].  "mix": block i = gen(kind, block_bytes, base_seed + i); the other
    kinds are generated 4 MiB at a time (chunk c seeded base_seed ^ c) and cut into blocks."""
    out = np.empty(nblocks * block_bytes, dtype=np.uint8)
    if kind == "mix":
        for i in range(nblocks):
            out[i * block_bytes:(i + 1) * block_bytes] = gen(kind, block_bytes, base_seed + i)
        return out
    per = max(1, (4 << 20) // block_bytes)
    for c, b0 in enumerate(range(0, nblocks, per)):
        nb = min(per, nblocks - b0)
        out[b0 * block_bytes:(b0 + nb) * block_bytes] = gen(kind, nb * block_bytes, base_seed ^ c)
    return out
