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
