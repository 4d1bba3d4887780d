#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_lz4_decode_v4 (PROF build, variant 9): s_memtime deltas per block."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rust_compress_amd as R
from rust_compress_amd import _native as N
import bench
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
dev = torch.device("cuda", 0)
ctx = R.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dec, raw, cb, ob = bench.make_workload(R, ctx, torch, dev, kind, 4096, 0x4C5A3401)
ctx.set_variant(N.LZ4_DECODE, 9)
sc = torch.zeros(4096 * 16 * 8 + 64, dtype=torch.uint8, device=dev)
ctx.launch_dev(N.LZ4_DECODE, dec, sc); torch.cuda.synchronize()
ctx.launch_dev(N.LZ4_DECODE, dec, sc); torch.cuda.synchronize()
p = sc[: 4096 * 128].view(torch.int64).view(4096, 16).cpu().numpy().astype(np.float64)
names = ["parse", "make_room", "head+scan+valid", "far+literals", "dep masks", "copy rounds", "flush", "stage", "solo", "wide", "#batches", "#entries"]
tot = p[:, :10].sum(axis=1).mean()
print("kind", kind, "mean cycles/block (sum of phases)", int(tot), "batches/block", p[:, 10].mean(), "entries/batch", p[:, 11].mean() / max(p[:, 10].mean(), 1))
for i in range(10):
    print("%-18s %10.0f cycles/block  %5.1f%%" % (names[i], p[:, i].mean(), 100 * p[:, i].mean() / tot))
print("copy iterations/batch %.2f   parse windows/batch %.2f" % (p[:, 12].mean() / p[:, 10].mean(), p[:, 13].mean() / p[:, 10].mean()))
print("parse split: window setup %.0f  walk %.0f  (cycles/block)" % (p[:, 14].mean(), p[:, 15].mean()))
