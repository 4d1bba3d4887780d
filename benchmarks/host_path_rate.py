#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-memory entry point (rcx_lz4_decode_batch with RCX_MEM_HOST) on the headline
workload: descriptors and data on the host, one H2D of the compressed bytes + one D2H of the decoded bytes inside
the call.  bench.py's `value` is the HBM-resident rate; this is the number DESIGN.md quotes beside it."""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rust_compress_amd as R
from rust_compress_amd import _native as N
import bench

dev = torch.device("cuda", 0)
ctx = R.Context(0)
nb = 4096
dec, raw, cb, ob = bench.make_workload(R, ctx, torch, dev, "text", nb, 0x4C5A3401)
u64 = lambda t: np.ascontiguousarray(t.cpu().numpy().astype(np.uint64))
slots = np.ascontiguousarray(dec.in_base.cpu().numpy())
so, in_len, out_off, out_cap = u64(dec.in_off), u64(dec.in_len), u64(dec.out_off), u64(dec.out_cap)
# the encoder wrote one block per compression_bound-sized slot; a caller hands over packed blocks (as an LZ4 frame holds them)
in_off = np.zeros(nb, np.uint64)
pos = 0
for i in range(nb):
    in_off[i] = pos; pos += (int(in_len[i]) + 15) & ~15
in_base = np.zeros(pos + 64, np.uint8)
for i in range(nb):
    in_base[int(in_off[i]): int(in_off[i]) + int(in_len[i])] = slots[int(so[i]): int(so[i]) + int(in_len[i])]
# RCX_HP_PIECES: block ranges of the page-locked call (rcx_ctx_set_param(ctx, RCX_LZ4_DECODE, pieces << 8); 0 = the library's choice), a list to sweep
sweep = [int(x, 0) for x in os.environ.get("RCX_HP_PIECES", "0").split(",")]      # (values above 255: first-range divisor << 8 and growth in quarters << 16 ride along, rcx_api.hip)
for pinned in [False] + [True] * len(sweep):
    if pinned:
        N.lib().rcx_ctx_set_param(ctx._h, N.LZ4_DECODE, sweep[0] << 8)
        print("pieces", sweep.pop(0), end=": ")
    if pinned:
        inb = torch.from_numpy(in_base).pin_memory(); outb = torch.empty(nb * bench.BLOCK + 64, dtype=torch.uint8).pin_memory()
        ip, op = inb.data_ptr(), outb.data_ptr()
    else:
        outn = np.zeros(nb * bench.BLOCK + 64, dtype=np.uint8)
        ip, op = in_base.ctypes.data, outn.ctypes.data
    out_len, in_used, status = np.zeros(nb, np.uint64), np.zeros(nb, np.uint64), np.zeros(nb, np.int32)
    p = lambda a: a.ctypes.data
    b = N.Batch(ip, p(in_off), p(in_len), op, p(out_off), p(out_cap), p(out_len), p(in_used), p(status), nb, N.MEM_HOST)
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        rc = N.lib().rcx_lz4_decode_batch(ctx._h, C.byref(b))
        ts.append(time.perf_counter() - t0)
    assert rc == 0 and not status.any()
    got = outb.numpy() if pinned else outn
    assert np.array_equal(got[: nb * bench.BLOCK], raw.cpu().numpy()[: nb * bench.BLOCK])
    t = float(np.median(ts[1:]))
    print("host path (%s host buffers): %.2f ms per 4096 x 64 KiB  -> %.1f GiB/s decoded, PCIe-inclusive (%.0f MB in, %.0f MB out)"
          % ("pinned" if pinned else "pageable", t * 1e3, ob / t / 2**30, cb / 1e6, ob / 1e6))

# the same blocks through rcx_multi_batch over the ONE device listed k times: k contexts, k host threads, each range's copy in, decode
# and copy out under the others' (include/rcx.h)
L = N.lib()
inb = torch.from_numpy(in_base).pin_memory(); outb = torch.empty(nb * bench.BLOCK + 64, dtype=torch.uint8).pin_memory()
for k in (() if os.environ.get("RCX_HP_ONLY") else (1, 2, 3, 4)):      # (RCX_HP_ONLY: benchmarks/r5_hostpath_trace.sh -- the trace ends with the page-locked single-context call)
    devs = (C.c_int * k)(*([0] * k))
    h = C.c_void_p()
    assert L.rcx_multi_create(devs, k, C.byref(h)) == 0
    out_len, in_used, status = np.zeros(nb, np.uint64), np.zeros(nb, np.uint64), np.zeros(nb, np.int32)
    b = N.Batch(inb.data_ptr(), p(in_off), p(in_len), outb.data_ptr(), p(out_off), p(out_cap), p(out_len), p(in_used), p(status), nb, N.MEM_HOST)
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        rc = L.rcx_multi_batch(h, N.LZ4_DECODE, C.byref(b), None, None, None)
        ts.append(time.perf_counter() - t0)
    assert rc == 0 and not status.any()
    assert np.array_equal(outb.numpy()[: nb * bench.BLOCK], raw.cpu().numpy()[: nb * bench.BLOCK])
    t = float(np.median(ts[1:]))
    print("rcx_multi_batch, device 0 listed %d x (pinned): %.2f ms -> %.1f GiB/s decoded, PCIe-inclusive" % (k, t * 1e3, ob / t / 2**30))
    L.rcx_multi_destroy(h)

