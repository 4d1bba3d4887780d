#!/usr/bin/env python3
"""Adaptive byte range coder: the three kernels (1 = a lane, 2 = a wave, 3 = a quad of lanes per stream) over the number of
streams -- where the automatic choice should switch.  python benchmarks/ari_variant_sweep.py [symbols per stream]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
ctx = R.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
i64 = lambda a: torch.tensor(np.asarray(a, dtype=np.int64), dtype=torch.int64, device=dev)
for n in (16, 256, 1024, 4096, 12288, 16384, 32768, 65536, 131072, 262144):
    raw_np = synth.gen_blocks("text", n, S, 77)
    # a DC-like symbol distribution: mostly small values
    raw_np = (np.minimum(raw_np, 250) // 16).astype(np.uint8) if os.environ.get("SMALLSYM") else raw_np
    raw = torch.from_numpy(raw_np).to(dev)
    ar = np.arange(n, dtype=np.int64)
    slot = 2 * S + 64
    row = []
    for v in (1, 2, 3):
        if v == 2 and n > 32768:
            row.append("   -      -   "); continue
        enc = R.DeviceBatch(raw, i64(ar * S), i64(np.full(n, S)), torch.zeros(n * slot + 64, dtype=torch.uint8, device=dev), i64(ar * slot), i64(np.full(n, slot)))
        ctx.set_variant(N.ARI_BYTE_ENCODE, v); ctx.set_variant(N.ARI_BYTE_DECODE, v)
        t = []
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.launch_dev(N.ARI_BYTE_ENCODE, enc); torch.cuda.synchronize(); te = time.perf_counter() - t0
        assert int(enc.status.abs().max()) == 0
        dec = R.DeviceBatch(enc.out_base, enc.out_off, enc.out_len[:n].clone(), torch.zeros(n * S + 64, dtype=torch.uint8, device=dev), i64(ar * S), i64(np.full(n, S)))
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.launch_dev(N.ARI_BYTE_DECODE, dec); torch.cuda.synchronize(); td = time.perf_counter() - t0
        assert int(dec.status.abs().max()) == 0 and torch.equal(dec.out_base[: n * S], raw)
        row.append("%6.2f %6.2f" % (te * 1e3, td * 1e3))
    print("n %7d  (enc dec ms)  lane %s | wave %s | quad %s" % (n, row[0], row[1], row[2]))
