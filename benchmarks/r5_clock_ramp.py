#!/usr/bin/env python3
"""Does the headline launch get faster once the GPU has been busy for a while?  Per-launch event times of 600 back-to-back
launches of the headline workload (4096 x 64 KiB G-text LZ4 decode), in groups of 20."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_compress_amd as R
from rust_compress_amd import _native as N
import bench
dev = torch.device("cuda", 0); ctx = R.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dec, raw, cb, ob = bench.make_workload(R, ctx, torch, dev, "text", 4096, 0x4C5A3401)
torch.cuda.synchronize()
import time; time.sleep(2.0)                       # the GPU idle, as before a benchmark's first launch
n = 600
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    ctx.launch_dev(N.LZ4_DECODE, dec); ev[i + 1].record()
torch.cuda.synchronize()
ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n)])
for g in range(0, n, 20):
    print("launches %3d..%3d: mean %.4f ms  min %.4f  max %.4f" % (g, g + 19, ms[g:g + 20].mean(), ms[g:g + 20].min(), ms[g:g + 20].max()))
