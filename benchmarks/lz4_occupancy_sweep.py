#!/usr/bin/env python3
"""LZ4 decode time as a function of the number of concurrent blocks (one wave each): latency- or issue-bound?
Also prints the PROF phase breakdown at each size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rust_compress_amd as R
from rust_compress_amd import _native as N
import bench
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
variants = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 10)
dev = torch.device("cuda", 0)
ctx = R.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
names = ["parse", "room", "head", "far+lit", "dep", "copy", "flush", "stage", "solo", "wide"]
for nb in (256, 1024, 2048, 4096, 8192, 16384):
    dec, raw, cb, ob = bench.make_workload(R, ctx, torch, dev, kind, nb, 0x4C5A3401)
    for variant in variants:
        ctx.set_variant(N.LZ4_DECODE, variant)
        sc = torch.zeros(nb * 16 * 8 + 64, dtype=torch.uint8, device=dev)
        for _ in range(2):
            ctx.launch_dev(N.LZ4_DECODE, dec, sc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ctx.launch_dev(N.LZ4_DECODE, dec, sc)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        line = "blocks %6d variant %d  %.3f ms  %.1f GiB/s" % (nb, variant, ms, nb * 65536 / ms / 1e-3 / 2**30)
        if variant == 99:
            p = sc[: nb * 128].view(torch.int64).view(nb, 16).cpu().numpy().astype(np.float64)
            tot = p[:, :10].sum(axis=1).mean()
            line += "  cyc/block %.0fK: " % (tot / 1e3) + " ".join("%s %.0fK" % (names[i], p[:, i].mean() / 1e3) for i in range(10))
        print(line, flush=True)
