#!/usr/bin/env python3
"""Large randomized parity run (not part of the suite: minutes of CPU oracle time): mutated compressed streams through
the GPU decoders and the oracle; statuses must agree everywhere, bytes and consumed counts wherever the oracle succeeds.
    python benchmarks/fuzz_gpu.py [count] [seed]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth, batch as B
import oracle_py as O



def sources(rng):
    out = []
    for kind in ("text", "words", "runs", "dna4", "mix", "rand"):
        for sz in (0, 1, 13, 200, 3000, 20000, 70000):
            out.append(synth.gen(kind, sz, int(rng.integers(1 << 30))).tobytes())
    out += [bytes(70000), b"ab" * 30000, bytes(range(256)) * 64]
    return out


def mutate(rng, valid, count):
    blobs, caps = [], []
    for it in range(count):
        b = bytearray(valid[int(rng.integers(len(valid)))])
        mode = it % 7
        if mode == 0 and len(b):
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(len(b)))] = int(rng.integers(256))
        elif mode == 1 and len(b):
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(len(b)))] ^= 1 << int(rng.integers(8))
        elif mode == 2:
            b = b[: int(rng.integers(len(b) + 1))]
        elif mode == 3:
            b = b + bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        elif mode == 4:
            o = valid[int(rng.integers(len(valid)))]
            b = b[: int(rng.integers(len(b) + 1))] + o[int(rng.integers(len(o) + 1)):]
        elif mode == 5:
            b = bytearray(rng.integers(0, 256, int(rng.integers(0, 400)), dtype=np.uint8).tobytes())
        blobs.append(bytes(b))
        caps.append(int(rng.choice([0, 7, 200, 3000, 20000, 70000, 70001, 140000])))
    return blobs, caps


def oracle_batch(codec, blobs, caps):
    base, off, lens = B.pack(blobs)
    total, ooff, ocap = B.layout(caps)
    out = np.zeros(total + 64, np.uint8)
    aux = np.zeros(len(blobs), np.uint32)
    _, out_len, in_used, status = O.batch_run(codec, base, off, lens, out, ooff, ocap, aux=aux, threads=os.cpu_count() or 8)
    return out, ooff, out_len, in_used, status


def check(ctx, name, codec, gpu_fn, blobs, caps, variants, cmp_used, cmp_partial):
    out, ooff, olen, used, st = oracle_batch(codec, blobs, caps)
    total = 0
    for v in variants:
        ctx.set_variant(codec, v)
        res = gpu_fn(blobs, caps)
        bad = 0
        for i in range(len(blobs)):
            exp = out[int(ooff[i]): int(ooff[i]) + int(olen[i])].tobytes()
            ok = int(res.status[i]) == int(st[i])
            if ok and (st[i] == 0 or cmp_partial):
                ok = res.outputs[i] == exp
            if ok and cmp_used and st[i] == 0:
                ok = int(res.in_used[i]) == int(used[i])
            if not ok:
                bad += 1
                if bad <= 5:
                    print("  MISMATCH %s v%d #%d: status gpu %d oracle %d, len gpu %d oracle %d, in %d bytes cap %d" % (
                        name, v, i, res.status[i], st[i], len(res.outputs[i]), olen[i], len(blobs[i]), caps[i]))
                    open(os.path.join(ROOT, "gpurun_out", "fuzz_%s_v%d_%d.bin" % (name, v, i)), "wb").write(blobs[i])
        print("%-12s variant %2d: %6d streams, %5d ok status, %d mismatches" % (name, v, len(blobs), int((st == 0).sum()), bad), flush=True)
        ctx.set_variant(codec, 0)
        total += bad
    return total



def main(count=20000, seed=1, ctx=None):
    rng = np.random.default_rng(seed)
    ctx = ctx or R.Context(0)
    bad = 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    src = sources(rng)
    lz4 = [O.lz4_encode_block(s) for s in src]
    blobs, caps = mutate(rng, lz4, count)
    bad += check(ctx, "lz4", N.LZ4_DECODE, ctx.lz4_decode_blocks, blobs, caps, (0, 15), False, False)
    zs = [zlib.compress(s, int(rng.integers(0, 10))) for s in src]
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    zs.append(c.compress(src[10]) + c.flush())
    blobs, caps = mutate(rng, zs, count // 2)
    bad += check(ctx, "zlib", N.ZLIB_DECODE, ctx.zlib_decode, blobs, caps, (0, 9), True, False)
    raw = [z[2:-4] for z in zs]
    blobs, caps = mutate(rng, raw, count // 2)
    bad += check(ctx, "inflate", N.INFLATE, ctx.inflate, blobs, caps, (0, 9), True, True)
    rle = [O.rle_encode(s) for s in src]
    blobs, caps = mutate(rng, rle, count // 4)
    bad += check(ctx, "rle", N.RLE_DECODE, ctx.rle_decode, blobs, caps, (0,), False, False)
    ari = [O.ari_byte_encode(s) for s in src if len(s) <= 20000]
    blobs, caps = mutate(rng, ari, count // 8)
    bad += check(ctx, "ari", N.ARI_BYTE_DECODE, ctx.ari_byte_decode, blobs, caps, (1, 2, 3), True, False)
    print("done")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
