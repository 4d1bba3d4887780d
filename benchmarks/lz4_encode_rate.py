#!/usr/bin/env python3
"""LZ4 block ENCODE rate: 4096 x 64 KiB per distribution, windowed probe (variant 0) against the serial
probe chain (variant 1); the two outputs are compared byte for byte, and a sample of blocks with the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth
import oracle_py as O

VARIANTS = [int(x) for x in os.environ.get("VARIANTS", "0,1").split(",")]
BLOCK, NB = 65536, int(os.environ.get("NB", "4096"))
dev = torch.device("cuda", 0)
ctx = R.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)
for kind in sys.argv[1:] or ["text", "words", "runs", "rand", "mix"]:
    raw_h = synth.gen_blocks(kind, NB, BLOCK, 0x4C5A)
    raw = torch.from_numpy(raw_h).to(dev)
    bound = int(N.lib().rcx_lz4_compression_bound(BLOCK))
    slot = (bound + 15) & ~15
    ar = np.arange(NB)
    scratch = torch.empty(ctx.scratch_bytes(N.LZ4_ENCODE, NB, BLOCK) + 64, dtype=torch.uint8, device=dev)
    outs = {}
    for v in VARIANTS:
        enc = R.DeviceBatch(raw, i64(ar * BLOCK), i64(np.full(NB, BLOCK)), torch.zeros(NB * slot + 64, dtype=torch.uint8, device=dev),
                            i64(ar * slot), i64(np.full(NB, slot)))
        ctx.set_variant(N.LZ4_ENCODE, v)
        ctx.launch_dev(N.LZ4_ENCODE, enc, scratch); torch.cuda.synchronize()
        assert int(enc.status.abs().max()) == 0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for i in range(3):
            ev[i].record(); ctx.launch_dev(N.LZ4_ENCODE, enc, scratch)
        ev[3].record(); torch.cuda.synchronize()
        ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(3))
        outs[v] = (enc.out_base.cpu().numpy(), enc.out_len[:NB].cpu().numpy())
        print("%-6s variant %d: %8.3f ms  %7.2f GiB/s in   ratio %.3f" % (kind, v, ms, NB * BLOCK / ms / 1e-3 / 2**30,
                                                                          NB * BLOCK / float(outs[v][1].sum())), flush=True)
    v0 = VARIANTS[0]
    same = all(np.array_equal(outs[v0][1], outs[v][1]) and all(
        np.array_equal(outs[v0][0][b * slot: b * slot + int(outs[v0][1][b])], outs[v][0][b * slot: b * slot + int(outs[v0][1][b])]) for b in range(NB))
        for v in VARIANTS[1:])
    samp = [0, 1, NB // 2, NB - 1]
    okr = all(outs[v0][0][b * slot: b * slot + int(outs[v0][1][b])].tobytes() == O.lz4_encode_block(raw_h[b * BLOCK:(b + 1) * BLOCK].tobytes()) for b in samp)
    print("%-6s variants identical: %s   oracle sample identical: %s" % (kind, same, okr), flush=True)
    del raw, scratch, enc
