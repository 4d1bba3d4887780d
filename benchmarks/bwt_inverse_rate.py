#!/usr/bin/env python3
"""Inverse BWT of 1024 x 256 KiB, device resident: the table inverse (bwt/mod.rs:223-294) and decode_minimal (:298-315)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth
nb, BLOCK = 1024, 262144
dev = torch.device("cuda", 0); ctx = R.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
ar = np.arange(nb, dtype=np.int64)
for kind in ("text", "dna4"):
    raw = torch.from_numpy(synth.gen_blocks(kind, nb, BLOCK, 0xB77)).to(dev)
    fw = R.DeviceBatch(raw, i64(ar * BLOCK), i64(np.full(nb, BLOCK)), torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK), i64(np.full(nb, BLOCK)))
    sc = torch.empty(ctx.scratch_bytes(N.BWT_FORWARD, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
    ctx.launch_dev(N.BWT_FORWARD, fw, sc); torch.cuda.synchronize(); del sc
    inv = R.DeviceBatch(fw.out_base, fw.out_off, fw.out_len, torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK), i64(np.full(nb, BLOCK)), aux=fw.aux)
    sc = torch.empty(ctx.scratch_bytes(N.BWT_INVERSE, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
    extra = [(N.BWT_INVERSE, "chase geometry %d" % g, g << 4) for g in range(1, 10)] if "--sweep" in sys.argv else []
    extra += [(N.BWT_INVERSE, "contract geometry %d" % g, g << 8) for g in range(1, 7)] if "--sweep2" in sys.argv else []
    for codec, name, variant in [(N.BWT_INVERSE, "inverse", 0)] + extra + [(N.BWT_INVERSE, "inverse (one workgroup per block)", 4), (N.BWT_INVERSE, "inverse (scattered table)", 2), (N.BWT_INVERSE_MINIMAL, "decode_minimal", 0)]:
        ctx.set_variant(codec, variant)
        inv.out_base.zero_()
        ctx.launch_dev(codec, inv, sc); torch.cuda.synchronize()
        if codec == N.BWT_INVERSE: assert torch.equal(inv.out_base[: nb * BLOCK], raw)
        t0 = time.perf_counter()
        for _ in range(5): ctx.launch_dev(codec, inv, sc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("%-5s %-34s %.2f ms  %.1f GiB/s" % (kind, name, dt * 1e3, nb * BLOCK / dt / 2**30), flush=True)
