import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth
dev = torch.device("cuda", 0); ctx = R.Context(0)
nb, BS = 1024, 262144
raw = torch.from_numpy(synth.gen_blocks("text", nb, BS, 0xB7)).to(dev)
ar = np.arange(nb, dtype=np.int64)
i64 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int64), device=dev)
db = R.DeviceBatch(raw, i64(ar*BS), i64(np.full(nb, BS)), torch.empty(nb*BS+64, dtype=torch.uint8, device=dev), i64(ar*BS), i64(np.full(nb, BS)))
sc = torch.empty(ctx.scratch_bytes(N.BWT_FORWARD, nb, BS) + 256, dtype=torch.uint8, device=dev)
for _ in range(2):
    ctx.launch_dev(N.BWT_FORWARD, db, sc); torch.cuda.synchronize()
