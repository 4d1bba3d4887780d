"""Block codec stand-in for `bench.py --dry-gloo` (TEST INFRASTRUCTURE): the oracle's LZ4 block functions, so that the
benchmark's launcher, rank bookkeeping and scatter / gather legs can run on CPU ranks.  bench.py loads it by name from
RCX_BENCH_DRY_CODEC; nothing in the product imports it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402
from rust_compress_amd import synth  # noqa: E402

O.build()


def make_blocks(kind, nblocks, block, seed):
    raws = [synth.gen(kind, block, seed + i).tobytes() for i in range(nblocks)]
    return [O.lz4_encode_block(r) for r in raws], raws


def decode(blob, cap):
    return O.lz4_decode_block(blob, cap=cap)


# ---- BASELINE config 5 on CPU: the RCXQ container of rust_compress_amd.pipeline written / read with the oracle's stages
# (BWT -> DC -> block record -> PARTS word-aligned pieces -> one range-coded stream each), so that the sharded pipeline
# leg of bench.py (--dry-gloo) runs on gloo ranks and a GPU test can compare the device's container with this one byte for byte.
def pipe_encode(data, block_size, parts=16):
    import numpy as np
    from rust_compress_amd import pipeline as P
    data = bytes(data)
    lens, praw, clen, payload = [], [], [], []
    for i in range(0, len(data), block_size):
        blk = data[i:i + block_size]
        L, origin = O.bwt_encode(blk)
        words = O.dc_encode(L)
        rec = np.concatenate([np.array([len(blk), origin, len(words) - 256], dtype="<u4"), words.astype("<u4")]).tobytes()
        cuts = [(len(rec) * j // parts) & ~3 for j in range(parts)] + [len(rec)]
        lens.append(len(blk))
        for j in range(parts):
            piece = rec[cuts[j]:cuts[j + 1]]
            coded = O.ari_byte_encode(piece)
            praw.append(len(piece)); clen.append(len(coded)); payload.append(coded)
    return P.build_container(block_size, parts, lens, praw, clen, b"".join(payload))


def pipe_decode(blob):
    import numpy as np
    from rust_compress_amd import pipeline as P
    block_size, parts, lens, praw, clen, p = P.parse_container(blob)
    out = []
    for b in range(len(lens)):
        rec = b""
        for j in range(parts):
            cl, rl = int(clen[b, j]), int(praw[b, j])
            rec += O.ari_byte_decode(blob[p:p + cl], cap=rl)[0]
            p += cl
        w = np.frombuffer(rec, dtype="<u4")
        n, origin, k = int(w[0]), int(w[1]), int(w[2])
        assert n == int(lens[b]) and len(w) == 3 + 256 + k
        L, _ = O.dc_decode(w[3:], n)
        out.append(O.bwt_decode(L, origin))
    return b"".join(out)
