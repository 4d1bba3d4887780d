#!/usr/bin/env python3
"""Boundary sizes (powers of two and the wave / tile / window sizes of the kernels, each -1 / 0 / +1) through every codec, against the
oracle: encoders byte for byte, decoders on the oracle's streams.  python benchmarks/edge_sizes.py"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rust_compress_amd as R
from rust_compress_amd import synth
import oracle_py as O


def sizes(limit):
    out = set()
    for p in (0, 1, 2, 3, 5, 6, 8, 12, 13, 14, 15, 16, 17, 18, 20):
        for d in (-1, 0, 1):
            v = (1 << p) + d
            if 0 <= v <= limit:
                out.add(v)
    for v in (32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 2047, 2048, 2049, 2575, 2576, 4623, 4624, 65535 + 2048):
        if v <= limit:
            out.update((v - 1, v, v + 1))
    return sorted(out)


def main(ctx=None, limit=(1 << 20) + 1, kinds=("text", "runs", "dna4", "rand")):
    ctx = ctx or R.Context(0)
    bad = 0
    def check(name, got, want):
        nonlocal bad
        for i, (g, w) in enumerate(zip(got, want)):
            if g != w:
                bad += 1
                if bad < 10: print("MISMATCH", name, "input", i, len(raws[i]))
    for kind in kinds:
        raws = [synth.gen(kind, n, 1000 + n % 97).tobytes() if n else b"" for n in sizes(limit)]
        # transforms and serial coders
        fw = ctx.bwt_forward(raws).check()
        exp = [O.bwt_encode(r) for r in raws]
        check("bwt L " + kind, fw.outputs, [e[0] for e in exp])
        check("bwt origin " + kind, [int(a) for r, a in zip(raws, fw.aux) if r], [e[1] for r, e in zip(raws, exp) if r])
        nz = [i for i, r in enumerate(raws) if r]
        inv = ctx.bwt_inverse([exp[i][0] for i in nz], [exp[i][1] for i in nz]).check()
        check("bwt inverse " + kind, inv.outputs, [raws[i] for i in nz])
        sm = [i for i in nz if len(raws[i]) <= 70000]                                 # the O(n^2) restatement of decode_minimal
        mn = ctx.bwt_inverse_minimal([exp[i][0] for i in sm], [exp[i][1] for i in sm]).check()
        check("bwt inverse (minimal) " + kind, mn.outputs, [O.bwt_decode(exp[i][0], exp[i][1], minimal=True) for i in sm])
        Ls = [e[0] for e in exp]
        check("mtf enc " + kind, ctx.mtf_encode(Ls).check().outputs, [O.mtf_encode(x) for x in Ls])
        check("mtf dec " + kind, ctx.mtf_decode([O.mtf_encode(x) for x in Ls]).check().outputs, Ls)
        dce = [O.dc_encode(x).tobytes() for x in Ls]
        check("dc enc " + kind, ctx.dc_encode(Ls).check().outputs, dce)
        check("dc dec " + kind, ctx.dc_decode(dce, [len(x) for x in Ls]).check().outputs, Ls)
        check("rle enc " + kind, ctx.rle_encode(raws).check().outputs, [O.rle_encode(r) for r in raws])
        check("rle dec " + kind, ctx.rle_decode([O.rle_encode(r) for r in raws], [len(r) for r in raws]).check().outputs, raws)
        small = [r for r in raws if len(r) <= 70000]                      # one serial chain per stream: keep these short
        ae = [O.ari_byte_encode(r) for r in small]
        check("ari enc " + kind, ctx.ari_byte_encode(small).check().outputs, ae)
        check("ari dec " + kind, ctx.ari_byte_decode(ae, [len(r) for r in small]).check().outputs, small)
        # LZ4 blocks and zlib members
        le = [O.lz4_encode_block(r) for r in raws]
        check("lz4 enc " + kind, ctx.lz4_encode_blocks(raws).check().outputs, le)
        check("lz4 dec " + kind, ctx.lz4_decode_blocks(le, [len(r) for r in raws]).check().outputs, raws)
        zs = [zlib.compress(r, (1, 6, 9)[i % 3]) for i, r in enumerate(raws)]
        check("zlib dec " + kind, ctx.zlib_decode(zs, [len(r) for r in raws]).check().outputs, raws)
        print(kind, len(raws), "sizes up to", max(len(r) for r in raws), "mismatches so far", bad, flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
