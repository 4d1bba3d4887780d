#!/usr/bin/env python3
"""How even are the blocks of the headline launch?  Per-block wall time of the executor wave (the phase-timer build, A/B variant 24)
as percentiles of the slowest block's, and by XCD / dispatch order.  RCX_AB=1 python benchmarks/r4_lz4_tail.py [kind] [nblocks]"""
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
ctx.set_variant(N.LZ4_DECODE, 24)
sc = torch.zeros(nb * 256 + 64, dtype=torch.uint8, device=dev)
for _ in range(3):
    sc.zero_()
    ctx.launch_dev(N.LZ4_DECODE, dec, sc); torch.cuda.synchronize()
p = sc[: nb * 256].view(torch.int64).view(nb, 32).cpu().numpy().astype(np.float64)
tot = np.maximum(p[:, 10], p[:, 11])
batches = p[:, 9]
clen = dec.in_len[:nb].cpu().numpy().astype(np.float64)
mx = tot.max()
print("kind %s, %d blocks: per-block wall (cycles of the timer): min %.0f  p10 %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f" % (
    kind, nb, tot.min(), np.percentile(tot, 10), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), mx))
print("  mean / max = %.3f (the share of the launch an average block's slot is busy)" % (tot.mean() / mx))
print("  batches per block: min %.0f median %.0f max %.0f; corr(wall, batches) %.2f, corr(wall, compressed bytes) %.2f" % (
    batches.min(), np.median(batches), batches.max(), np.corrcoef(tot, batches)[0, 1], np.corrcoef(tot, clen)[0, 1]))
for x in range(8):
    t = tot[x::8]
    print("  blocks %d mod 8 (one XCD if workgroups go round-robin): mean %.0f max %.0f" % (x, t.mean(), t.max()))
q = nb // 8
print("  by dispatch order, eighths: " + "  ".join("%.0f" % tot[i * q:(i + 1) * q].mean() for i in range(8)))
slot = p[:, 12].astype(np.int64) & 15
print("  by the executor wave's slot on its SIMD (HW_ID[3:0]): " + "  ".join("%d: %.0f (%d)" % (k, tot[slot == k].mean(), int((slot == k).sum())) for k in range(16) if (slot == k).any()))

