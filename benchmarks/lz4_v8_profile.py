#!/usr/bin/env python3
"""Phase timers of the segment-parallel parser wave (k_lz4_decode_v8, A/B variant 24): RCX_AB=1 python benchmarks/lz4_v8_profile.py [kind] [nblocks]"""
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
for _ in range(2):
    ctx.launch_dev(N.LZ4_DECODE, dec, sc); torch.cuda.synchronize()
allp = sc[: nb * 256].view(torch.int64).view(nb, 32).cpu().numpy().astype(np.float64)
p = allp.mean(axis=0)
if allp[:, 13].min() > 0:      # when each block started and ended (s_memrealtime, 10 ns ticks), against the first start
    t0 = allp[:, 13].min(); st = (allp[:, 13] - t0) / 100.0; en = (allp[:, 14] - t0) / 100.0
    q = lambda a: "  ".join("%d%% %.1f" % (k, np.percentile(a, k)) for k in (0, 10, 50, 90, 99, 100))
    print("block start, us after the first:", q(st)); print("block end,   us after the first start:", q(en)); print("block life, us:", q(en - st))
    raw = sc[: nb * 256].view(torch.int64).view(nb, 32).cpu().numpy()
    hw = raw[:, 15] & 0xffffffff; xcc = (raw[:, 15] >> 32) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)      # CU_ID[11:8], SH_ID[12], SE_ID[15:13], XCC
    ids = np.unique(cu); print("CUs seen:", len(ids), " blocks a CU: min %d max %d" % (min((cu == i).sum() for i in ids), max((cu == i).sum() for i in ids)))
    cend = np.array([en[cu == i].max() for i in ids]); cfirst = np.array([en[cu == i].min() for i in ids])
    print("a CU's LAST block ends, us:", q(cend)); print("a CU's FIRST block ends, us:", q(cfirst))
    rank = (hw & 0xf) >> 1
    ilen = dec.in_len.cpu().numpy()[:nb].astype(np.float64); life = en - st; nbat = allp[:, 9]
    print("compressed bytes a block: min %d mean %d max %d; batches a block: min %d mean %.1f max %d" % (ilen.min(), ilen.mean(), ilen.max(), nbat.min(), nbat.mean(), nbat.max()))
    print("corr(block life, compressed bytes) %.3f  corr(block life, batches) %.3f  corr(executor cycles, batches) %.3f" % (np.corrcoef(life, ilen)[0, 1], np.corrcoef(life, nbat)[0, 1], np.corrcoef(allp[:, 11], nbat)[0, 1]))
    cb = np.array([nbat[cu == i].sum() for i in ids]); print("batches a CU: min %d mean %.0f max %d; corr(CU's last end, its batches) %.3f" % (cb.min(), cb.mean(), cb.max(), np.corrcoef(cend, cb)[0, 1]))
    wg = np.arange(nb); print("blocks of CU", ids[0], ":", wg[cu == ids[0]][:16], " of CU", ids[1], ":", wg[cu == ids[1]][:16])
    np.save(os.path.join(ROOT, "gpurun_out", "lz4_block_cu.npy"), cu)
    print("block end by the executor's age rank on its SIMD (wave slot >> 1), us: " + "  ".join("%d: %.1f (n %d)" % (r, en[rank == r].mean(), (rank == r).sum()) for r in np.unique(rank)))
    xe = (allp[:, 11]); print("executor cycles per block: min %.0fK mean %.0fK max %.0fK" % (xe.min() / 1e3, xe.mean() / 1e3, xe.max() / 1e3))
names = ["walk", "link", "list", "fields", "post(wait)", "-", "tiles", "walk steps", "repairs", "batches"]
print("kind %s, %d blocks: parser total %.0fK cycles, executor total %.0fK" % (kind, nb, p[10] / 1e3, p[11] / 1e3))
print("  cycles: " + "  ".join("%s %.0fK" % (names[i], p[i] / 1e3) for i in range(5)))
print("  counts: " + "  ".join("%s %.1f" % (names[i], p[i]) for i in range(6, 10)))
e = p[16:28]
print("  executor: header+scan %.0fK  make_room %.0fK  after_batch %.0fK  | of the waiting, for the first batch: %.0fK" % (e[1] / 1e3, e[11] / 1e3, e[3] / 1e3, e[9] / 1e3))
print("  executor: waiting for a batch %.0fK | scan+validate %.0fK  loads+chains %.0fK  lit/gather stores %.0fK  copy rounds %.0fK  flush %.0fK | rounds %.1f  emit calls %.1f  batches %.1f" % (e[0] / 1e3, e[4] / 1e3, e[5] / 1e3, e[6] / 1e3, e[7] / 1e3, e[8] / 1e3, e[9], e[10], e[2]))
