#!/usr/bin/env python3
"""Inverse BWT on arbitrary (L, origin) pairs -- most are not the transform of anything -- against the oracle: status for
status, bytes wherever the oracle succeeds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth, batch as B
import oracle_py as O


def main(count=6000, seed=4, ctx=None, minimal=False, variant=0):
    """minimal: the reference's decode_minimal (src/bwt/mod.rs:298-315) instead -- every pair with origin < n has an answer there,
    mostly a periodic one (the walk closes a short cycle); the oracle's restatement is O(n^2), so the blocks are smaller."""
    rng = np.random.default_rng(seed)
    ctx = ctx or R.Context(0)
    Ls, origins = [], []
    big = 30000 if minimal else 70000
    for it in range(count):
        n = int(rng.integers(1, 40)) if it % 3 == 0 else int(rng.integers(1, 3000)) if it % 3 == 1 else int(rng.integers(1, big))
        if not minimal and it % 40 == 7: n = int(rng.integers(70000, 700000))           # several table slots per marked node: chains of more than one step, park groups, the long list
        k = ("text", "runs", "dna4", "rand")[it % 4]
        src = synth.gen(k, n, int(rng.integers(1 << 30))).tobytes()
        L, og = O.bwt_encode(src)
        m = it % 4
        if m == 1:
            b = bytearray(L); b[int(rng.integers(n))] = int(rng.integers(256)); L = bytes(b)
        elif m == 2:
            og = int(rng.integers(0, n + 2))
        elif m == 3:
            L = bytes(rng.permutation(np.frombuffer(L, np.uint8)))
        Ls.append(L); origins.append(og)
    base, off, lens = B.pack(Ls)
    total, ooff, ocap = B.layout([len(x) for x in Ls])
    out = np.zeros(total + 64, np.uint8)
    aux = np.asarray(origins, np.uint32)
    _, olen, used, st = O.batch_run(N.BWT_INVERSE_MINIMAL if minimal else N.BWT_INVERSE, base, off, lens, out, ooff, ocap, aux=aux.copy(), threads=64)
    ctx.set_variant(N.BWT_INVERSE_MINIMAL if minimal else N.BWT_INVERSE, variant)       # bit 0: short parking, bit 1: the scattered-table kernel, bit 2: one workgroup per block
    res = ctx.bwt_inverse_minimal(Ls, origins) if minimal else ctx.bwt_inverse(Ls, origins)
    ctx.set_variant(N.BWT_INVERSE_MINIMAL if minimal else N.BWT_INVERSE, 0)
    bad = 0
    for i in range(len(Ls)):
        ok = int(res.status[i]) == int(st[i])
        if ok and st[i] == 0:
            ok = res.outputs[i] == out[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes()
        if not ok:
            bad += 1
            if bad < 5: print("MISMATCH", i, res.status[i], st[i], len(Ls[i]), origins[i])
    print("bwt inverse (minimal) fuzz:" if minimal else "bwt inverse fuzz:", len(Ls), "pairs,", int((st == 0).sum()), "ok status,", bad, "mismatches")
    return bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    sys.exit(1 if main(6000, seed) + main(3000, seed + 2, variant=1) + main(3000, seed + 3, variant=2) + main(3000, seed + 4, variant=4) + main(3000, seed + 1, minimal=True) else 0)
