#!/usr/bin/env python3
"""zlib decode of N members x 16 KiB for rocprofv3 (no multiprocessing): python inflate_profile.py [members] [variant]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth, batch as B
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
BLOCK = 16384
dev = torch.device("cuda", 0); ctx = R.Context(0)
raw_np = synth.gen_blocks("text", nb, BLOCK, 0x5A11)
members = [zlib.compress(raw_np[i * BLOCK:(i + 1) * BLOCK].tobytes(), (1, 6, 9)[i % 3]) for i in range(nb)]
base, off, lens = B.pack(members)
ar = np.arange(nb, dtype=np.int64)
db = R.DeviceBatch.from_host(base, off, lens, nb * BLOCK, (ar * BLOCK).astype(np.uint64), np.full(nb, BLOCK, dtype=np.uint64), dev)
sc = torch.empty(ctx.scratch_bytes(N.ZLIB_DECODE, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
ctx.set_variant(N.ZLIB_DECODE, variant)
for _ in range(3):
    ctx.launch_dev(N.ZLIB_DECODE, db, sc)
torch.cuda.synchronize()
assert int(db.status[:nb].abs().max()) == 0 and torch.equal(db.out_base[: nb * BLOCK].cpu(), torch.from_numpy(raw_np))
print("ok")
