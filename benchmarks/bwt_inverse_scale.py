import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth
BLOCK = 262144
dev = torch.device("cuda", 0); ctx = R.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
for nb in (8, 32, 64, 128, 256, 512, 1024):
    ar = np.arange(nb, dtype=np.int64)
    raw = torch.from_numpy(synth.gen_blocks("text", nb, BLOCK, 0xB77)).to(dev)
    fw = R.DeviceBatch(raw, i64(ar * BLOCK), i64(np.full(nb, BLOCK)), torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK), i64(np.full(nb, BLOCK)))
    sc = torch.empty(ctx.scratch_bytes(N.BWT_FORWARD, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
    ctx.launch_dev(N.BWT_FORWARD, fw, sc); torch.cuda.synchronize(); del sc
    inv = R.DeviceBatch(fw.out_base, fw.out_off, fw.out_len, torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK), i64(np.full(nb, BLOCK)), aux=fw.aux)
    sc = torch.empty(ctx.scratch_bytes(N.BWT_INVERSE, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
    for variant in (0, 2):
        ctx.set_variant(N.BWT_INVERSE, variant)
        ctx.launch_dev(N.BWT_INVERSE, inv, sc); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): ctx.launch_dev(N.BWT_INVERSE, inv, sc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("nb %4d variant %d  %.3f ms  (%.2f ms per wave of 256 blocks)" % (nb, variant, dt * 1e3, dt * 1e3 / max(1, nb / 256)), flush=True)
