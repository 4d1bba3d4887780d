"""DC decode on mutated distance-coded streams (wrong lengths, spliced words, truncation) against the oracle: status for status,
bytes wherever the oracle succeeds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth, batch as B
import oracle_py as O


def main(count=20000, seed=3, ctx=None):
    rng = np.random.default_rng(seed)
    ctx = ctx or R.Context(0)
    src = []
    for kind in ("text", "runs", "dna4", "rand", "mix"):
        for sz in (1, 7, 300, 5000, 40000):
            src.append(synth.gen(kind, sz, int(rng.integers(1 << 30))).tobytes())
    L = [O.bwt_encode(s)[0] for s in src]
    enc = [O.dc_encode(x).tobytes() for x in L]
    blobs, nouts = [], []
    for it in range(count):
        j = int(rng.integers(len(enc)))
        b = bytearray(enc[j]); n = len(L[j])
        m = it % 5
        if m == 0 and len(b):
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(len(b))); b[p] = int(rng.integers(256))
        elif m == 1:
            b = b[: (int(rng.integers(len(b) + 1)) // 4) * 4]
        elif m == 2 and len(b) >= 8:
            p = (int(rng.integers(len(b) // 4)) ) * 4; b[p:p+4] = int(rng.integers(0, n + 5)).to_bytes(4, "little")
        elif m == 3:
            n = max(0, n + int(rng.integers(-3, 4)))
        blobs.append(bytes(b)); nouts.append(n)
    base, off, lens = B.pack(blobs)
    total, ooff, ocap = B.layout(nouts)
    out = np.zeros(total + 64, np.uint8)
    _, olen, used, st = O.batch_run(N.DC_DECODE, base, off, lens, out, ooff, ocap, n_out=np.asarray(nouts, np.uint64), threads=64)
    res = ctx.dc_decode(blobs, nouts)
    bad = 0
    for i in range(len(blobs)):
        ok = int(res.status[i]) == int(st[i])
        if ok and st[i] == 0:
            ok = res.outputs[i] == out[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes()
        if not ok:
            bad += 1
            if bad < 5: print("MISMATCH", i, res.status[i], st[i], len(blobs[i]), nouts[i])
    print("dc decode fuzz:", len(blobs), "streams,", int((st == 0).sum()), "ok status,", bad, "mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(20000, int(sys.argv[1]) if len(sys.argv) > 1 else 3) else 0)
