#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_lz4_decode_v6 (PROF build, variant 19): s_memtime deltas of wave 0 per block."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rust_compress_amd as R
from rust_compress_amd import _native as N
import bench
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
ctx = R.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dec, raw, cb, ob = bench.make_workload(R, ctx, torch, dev, kind, nb, 0x4C5A3401)
ctx.set_variant(N.LZ4_DECODE, 19)
sc = torch.zeros(nb * 16 * 8 + 64, dtype=torch.uint8, device=dev)
ctx.launch_dev(N.LZ4_DECODE, dec, sc); torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); ctx.launch_dev(N.LZ4_DECODE, dec, sc); ev1.record(); torch.cuda.synchronize()
p = sc[: nb * 128].view(torch.int64).view(nb, 16).cpu().numpy().astype(np.float64)
names = ["stage", "hops+doubling+compose", "chain walk", "tokens (lift+fields)", "descs+bitmap", "classify", "resolve", "drain"]
tot = p[:, :8].sum(axis=1).mean()
print("kind", kind, "blocks", nb, "kernel ms %.3f" % ev0.elapsed_time(ev1), "mean cycles/block", int(tot), "rounds/block %.1f" % p[:, 8].mean(), "batches/block %.1f" % p[:, 9].mean(),
      "bailed", int((dec.status[:nb] != 0).sum()))
for i in range(8):
    print("%-24s %10.0f cycles/block  %5.1f%%   per round %7.0f" % (names[i], p[:, i].mean(), 100 * p[:, i].mean() / tot, p[:, i].mean() / max(p[:, 8].mean(), 1)))
