import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth
BLOCK = 262144; nb = 1024
dev = torch.device("cuda", 0); ctx = R.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
for kind in ("text", "dna4"):
    raw = torch.from_numpy(synth.gen_blocks(kind, nb, BLOCK, 0xB77)).to(dev)
    ar = np.arange(nb, dtype=np.int64)
    fw = R.DeviceBatch(raw, i64(ar * BLOCK), i64(np.full(nb, BLOCK)), torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK), i64(np.full(nb, BLOCK)))
    sc = torch.empty(ctx.scratch_bytes(N.BWT_FORWARD, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
    ref = None
    for v in (0, 512, 256, 128, 64, 32, 16):
        ctx.set_variant(N.BWT_FORWARD, v)
        ctx.launch_dev(N.BWT_FORWARD, fw, sc); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): ctx.launch_dev(N.BWT_FORWARD, fw, sc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        out = fw.out_base.clone() if hasattr(fw, "out_base") else None
        print(kind, "blocks per pass", v or 1024, "%.2f ms" % (dt * 1e3), flush=True)
