#!/usr/bin/env python3
"""A model of emit5's copy rounds (k_lz4_decode_v5.hip) on the headline's text blocks: how many rounds a batch takes and what
would change it.  Plain Python on the CPU; the compressor is the survey-derived one of tests/gen_derived_golden.py (bit-exact with
the reference's, tests/golden/derived).  python benchmarks/models/lz4_rounds_model.py [nblocks]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rust_compress_amd import synth
import gen_derived_golden as G

H, TCAP, LCAP, MCAP, SPLIT = 768, 1024, 32, 64, 32


def sequences(comp):
    p, n, out = 0, len(comp), []
    while p < n:
        t = comp[p]; p += 1
        L = t >> 4
        if L == 15:
            while True:
                x = comp[p]; p += 1; L += x
                if x != 255: break
        p += L
        if p >= n:
            out.append((L, 0, 0)); break
        off = comp[p] | (comp[p + 1] << 8); p += 2
        M = t & 15
        if M == 15:
            while True:
                x = comp[p]; p += 1; M += x
                if x != 255: break
        out.append((L, M + 4, off))
    return out


def batches(seqs):
    """the parser's batching: entries of one token each, two for a match longer than SPLIT, 64 a batch, cut at a long sequence"""
    cur, pos = [], 0
    for (L, M, off) in seqs:
        longrun = M > MCAP and M <= 255 and 0 < off < 16
        if L > LCAP or (M > MCAP and not longrun):
            if cur: yield pos, cur
            pos += sum(l + m for l, m, _ in cur) + L + M; cur = []
            continue
        ent = [(L, SPLIT, off), (0, M - SPLIT, off)] if (M > SPLIT and off >= 16) else [(L, M, off)]
        if len(cur) + len(ent) > 64 or sum(l + m for l, m, _ in cur) + sum(l + m for l, m, _ in ent) > TCAP:
            yield pos, cur
            pos += sum(l + m for l, m, _ in cur); cur = []
        cur += ent
    if cur: yield pos, cur


def rounds(oend0, ent, RR=2, progress=False, cap=16):
    ns = len(ent)
    L = np.array([e[0] for e in ent]); M = np.array([e[1] for e in ent]); off = np.array([e[2] for e in ent])
    ln = L + M
    ostart = oend0 + np.cumsum(ln) - ln
    mdst = ostart + L
    re = max(0, ((oend0 - H) & ~15))
    slo = mdst - off; shi = np.minimum(slo + M, mdst)
    isfar = (M > 0) & (slo < re)
    inb = (M > 0) & ~isfar & (shi > oend0)
    def entry_of(x):
        return int(np.searchsorted(ostart, x, side="right") - 1)
    ka = np.array([entry_of(max(slo[i], oend0)) for i in range(ns)]); kb = np.array([entry_of(max(shi[i] - 1, oend0)) for i in range(ns)])
    pmd = np.where(isfar | (off < M), 1 << 40, mdst)
    prod = np.where(inb & (ka == kb) & (slo >= pmd[ka]) & (off >= M), ka, 64)
    S = off.copy()
    for _ in range(RR):
        S2, prod2, ka2, kb2, inb2 = S.copy(), prod.copy(), ka.copy(), kb.copy(), inb.copy()
        for i in range(ns):
            j = prod[i]
            if j < 64:
                if mdst[i] - S[i] - S[j] >= re and S[i] + S[j] <= mdst[i]:
                    S2[i] = S[i] + S[j]; prod2[i] = prod[j]; ka2[i] = ka[j]; kb2[i] = kb[j]; inb2[i] = inb[j]
                else:
                    prod2[i] = 64
        S, prod, ka, kb, inb = S2, prod2, ka2, kb2, inb2
    src = mdst - S
    far16 = isfar
    Mc = np.where(far16, np.minimum(M, 16), M)
    pend = (M > 0)
    prog = np.zeros(ns, dtype=np.int64)
    r = 0
    while pend.any():
        r += 1
        ready = np.zeros(ns, bool)
        for i in range(ns):
            if not pend[i]: continue
            if far16[i] or not inb[i]:
                ready[i] = True; continue
            ok = True
            for k in range(ka[i], min(kb[i], i - 1) + 1):
                if k >= i: break
                if pend[k]:
                    if progress:
                        # bytes of k's match already copied cover what lane i reads this round?
                        need_hi = src[i] + prog[i] + min(cap, Mc[i] - prog[i])
                        have = mdst[k] + prog[k]
                        if need_hi <= have and not (k < kb[i] and need_hi > ostart[k] + ln[k]): continue
                    ok = False; break
            ready[i] = ok
        if not ready.any():
            ready[np.argmax(pend)] = True           # (self-overlapping lanes: the other loop)
        for i in np.nonzero(ready)[0]:
            nv = min(cap, Mc[i] - prog[i]); prog[i] += nv
            if prog[i] >= Mc[i]: pend[i] = False
    return r


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    raw = synth.gen_blocks("text", nb, 65536, 0x4C5A3401)
    tot = {}
    nbatch = 0
    hist = {}
    for b in range(nb):
        comp = G.lz4_encode_block(bytes(raw[b * 65536:(b + 1) * 65536]))
        seqs = sequences(comp)
        for oend0, ent in batches(seqs):
            nbatch += 1
            for name, kw in (("RR=2 (the kernel)", dict(RR=2)), ("RR=0", dict(RR=0)), ("RR=1", dict(RR=1)), ("RR=3", dict(RR=3)), ("RR=4", dict(RR=4)),
                             ("RR=2, a consumer starts when the bytes of its round stand", dict(RR=2, progress=True)),
                             ("RR=2, 32 bytes a round", dict(RR=2, cap=32))):
                r = rounds(oend0, ent, **kw)
                tot[name] = tot.get(name, 0) + r
                if name.startswith("RR=2 (the"): hist[r] = hist.get(r, 0) + 1
    print("%d blocks, %d batches (%.1f a block)" % (nb, nbatch, nbatch / nb))
    for k, v in tot.items():
        print("  %-62s %.2f rounds a batch" % (k, v / nbatch))
    print("  rounds of the kernel's setting, histogram:", dict(sorted(hist.items())))


if __name__ == "__main__":
    main()
