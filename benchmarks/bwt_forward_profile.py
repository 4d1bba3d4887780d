#!/usr/bin/env python3
"""BWT forward of 1024 x 256 KiB for rocprofv3 --kernel-trace --stats: python bwt_forward_profile.py [kind] [nblocks]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
BLOCK = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
dev = torch.device("cuda", 0); ctx = R.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
raw = torch.from_numpy(synth.gen_blocks(kind, nb, BLOCK, 0xB77)).to(dev)
ar = np.arange(nb, dtype=np.int64)
fw = R.DeviceBatch(raw, i64(ar * BLOCK), i64(np.full(nb, BLOCK)), torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK), i64(np.full(nb, BLOCK)))
sc = torch.empty(ctx.scratch_bytes(N.BWT_FORWARD, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
for _ in range(2):
    ctx.launch_dev(N.BWT_FORWARD, fw, sc)
torch.cuda.synchronize()
assert int(fw.status[:nb].abs().max()) == 0
print("ok")
