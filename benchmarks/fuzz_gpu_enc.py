#!/usr/bin/env python3
"""Bulk parity of the ENCODE side and the transforms (outside the suite): random inputs of every synthetic distribution,
sizes 0 .. 70 000 with the small sizes (token / end-of-block edge cases) over-represented, every kernel variant, compared
byte for byte with the oracle; decoders are then run on the GPU's own outputs (round trip).
    python benchmarks/fuzz_gpu_enc.py [count] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth, batch as B
import oracle_py as O

COUNT = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(SEED)
ctx = R.Context(0)
KINDS = ("text", "words", "runs", "dna4", "mix", "rand")


def inputs(count, maxn):
    out = []
    for i in range(count):
        m = i % 4
        n = int(rng.integers(0, 40)) if m == 0 else int(rng.integers(0, 600)) if m == 1 else int(rng.integers(0, 9000)) if m == 2 else int(rng.integers(0, maxn))
        k = KINDS[int(rng.integers(len(KINDS)))]
        b = synth.gen(k, n, int(rng.integers(1 << 30))).tobytes() if n else b""
        if i % 11 == 0 and n:
            b = bytes([b[0]]) * n                       # one long run
        if i % 13 == 0 and n > 8:
            p = int(rng.integers(1, 9)); b = (b[:p] * (n // p + 1))[:n]      # short period
        out.append(b)
    return out


def oracle_batch(codec, blobs, caps, aux=None, n_out=None):
    base, off, lens = B.pack(blobs)
    total, ooff, ocap = B.layout(caps)
    out = np.zeros(total + 64, np.uint8)
    a = np.zeros(max(len(blobs), 1), np.uint32) if aux is None else np.ascontiguousarray(aux, dtype=np.uint32)
    _, out_len, in_used, status = O.batch_run(codec, base, off, lens, out, ooff, ocap, aux=a, n_out=n_out, threads=os.cpu_count() or 8)
    return [out[int(o): int(o) + int(l)].tobytes() for o, l in zip(ooff, out_len)], status, a


def report(name, v, n, bad):
    print("%-14s variant %2d: %6d blocks, %d mismatches" % (name, v, n, bad), flush=True)


def cmp(name, v, got, exp, extra_ok=True):
    bad = sum(1 for g, e in zip(got, exp) if g != e) + (0 if extra_ok else 1)
    report(name, v, len(exp), bad)


raws = inputs(COUNT, 70000)
lens = [len(r) for r in raws]
exp, st, _ = oracle_batch(N.LZ4_ENCODE, raws, [int(N.lib().rcx_lz4_compression_bound(n)) or 1 for n in lens])
for v in (0, 1, 2):
    ctx.set_variant(N.LZ4_ENCODE, v)
    res = ctx.lz4_encode_blocks(raws)
    cmp("lz4 encode", v, res.outputs, exp, not res.status.any())
ctx.set_variant(N.LZ4_ENCODE, 0)
for v in (0, 15):
    ctx.set_variant(N.LZ4_DECODE, v)
    cmp("lz4 roundtrip", v, ctx.lz4_decode_blocks(exp, lens).outputs, raws)
ctx.set_variant(N.LZ4_DECODE, 0)

exp, _, _ = oracle_batch(N.RLE_ENCODE, raws, [int(N.lib().rcx_rle_encode_bound(n)) for n in lens])
res = ctx.rle_encode(raws); cmp("rle encode", 0, res.outputs, exp, not res.status.any())
cmp("rle roundtrip", 0, ctx.rle_decode(exp, lens).outputs, raws)
exp, _, _ = oracle_batch(N.MTF_ENCODE, raws, lens)
res = ctx.mtf_encode(raws); cmp("mtf encode", 0, res.outputs, exp, not res.status.any())
cmp("mtf roundtrip", 0, ctx.mtf_decode(exp).outputs, raws)
exp, _, _ = oracle_batch(N.DC_ENCODE, raws, [4 * (256 + n) for n in lens])
res = ctx.dc_encode(raws); cmp("dc encode", 0, res.outputs, exp, not res.status.any())
cmp("dc roundtrip", 0, ctx.dc_decode(exp, lens).outputs, raws)

small = [r[:20000] for r in raws]
slens = [len(r) for r in small]
exp, _, _ = oracle_batch(N.ARI_BYTE_ENCODE, small, [2 * n + 16 for n in slens])
for v in (1, 2, 3):
    ctx.set_variant(N.ARI_BYTE_ENCODE, v); ctx.set_variant(N.ARI_BYTE_DECODE, v)
    res = ctx.ari_byte_encode(small); cmp("ari encode", v, res.outputs, exp, not res.status.any())
    cmp("ari roundtrip", v, ctx.ari_byte_decode(exp, slens).outputs, small)
ctx.set_variant(N.ARI_BYTE_ENCODE, 0); ctx.set_variant(N.ARI_BYTE_DECODE, 0)
few = small[: max(COUNT // 8, 50)]
for rate in (1, 4, 7):
    e = [O.ari_binary_encode(r, rate) for r in few]
    res = ctx.ari_binary_encode(few, rate); cmp("ari binary r%d" % rate, 0, res.outputs, e, not res.status.any())
    cmp("  roundtrip", 0, ctx.ari_binary_decode(e, rate, [len(r) for r in few]).outputs, few)
e = [O.ari_proxy_encode(r) for r in few]
res = ctx.ari_proxy_encode(few); cmp("ari proxy", 0, res.outputs, e, not res.status.any())
cmp("  roundtrip", 0, ctx.ari_proxy_decode(e, [len(r) for r in few]).outputs, few)

exp, st, origin = oracle_batch(N.BWT_FORWARD, raws, lens)
res = ctx.bwt_forward(raws)
bad = sum(1 for g, e_, r, a, o in zip(res.outputs, exp, raws, res.aux, origin) if g != e_ or (len(r) and int(a) != int(o)))
report("bwt forward", 0, len(raws), bad + int(res.status.any()))
cmp("bwt roundtrip", 0, ctx.bwt_inverse(exp, origin[: len(raws)]).outputs, raws)
print("done")
