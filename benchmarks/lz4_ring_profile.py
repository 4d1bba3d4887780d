#!/usr/bin/env python3
"""Who waits for whom in the two-wave LZ4 decoder (variant 14): cycles each wave spends polling the descriptor ring."""
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
ctx.set_variant(N.LZ4_DECODE, 14)
sc = torch.zeros(4096 * 64 + 64, dtype=torch.uint8, device=dev)
for _ in range(2):
    ctx.launch_dev(N.LZ4_DECODE, dec, sc); torch.cuda.synchronize()
p = sc[: 4096 * 64].view(torch.int64).view(4096, 8).cpu().numpy().astype(np.float64)
print("kind", kind)
print("parser  : total %9.0f cycles, waiting for a free slot %9.0f (%4.1f%%), posts %.1f" % (p[:, 1].mean(), p[:, 0].mean(), 100 * p[:, 0].mean() / p[:, 1].mean(), p[:, 2].mean()))
print("executor: total %9.0f cycles, waiting for a batch     %9.0f (%4.1f%%), batches %.1f" % (p[:, 5].mean(), p[:, 4].mean(), 100 * p[:, 4].mean() / p[:, 5].mean(), p[:, 6].mean()))
