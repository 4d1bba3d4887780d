#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: GiB/s of decoded output over batched independent blocks.

Workload at N=1 (BASELINE.json configs[1]): LZ4 decode of 4096 independent 64 KiB blocks on one MI355X.
A "step" is one decode pass over the whole batch with inputs (compressed blocks + descriptors) and outputs
resident in HBM.  For N>1 every rank owns its own 4096 blocks (weak scaling, no data-path collective:
blocks are independent -- SURVEY.md 8e).  One JSON line is printed by rank 0.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself under
`torch.distributed.run` with one rank per GPU; launched by the driver under torch.distributed.run it reads
RANK / LOCAL_RANK / WORLD_SIZE as given.  Besides the kernel-only headline value the line carries
  * `end_to_end`: the same job when the compressed blocks live on rank 0 -- root scatter of the compressed ranges,
    decode, root gather of the decoded ranges (RCCL grouped send/recv, rust_compress_amd/dist.py), timed per phase;
  * `other_configs` (N=1): BASELINE configs 3, 4, 5 from the same process (benchmarks/bench_configs.py);
  * `hbm_ceiling_measured`: a device-to-device copy (the library's own 16-bytes-a-lane kernel), next to the 8 TB/s spec peak the roofline uses;
  * `single_stream` (N=1): one LZ4 block / one 1 MiB DEFLATE stream / one BWT block alone on the GPU, batches of 1..4096, break-even sizes;
  * `cpu_baseline` (N=1): the oracle on the host cores.
Inputs are synthetic (rust_compress_amd.synth) and are compressed on the GPU by the product's own bit-exact LZ4 encoder;
the oracle is used only for the cpu_baseline leg (and its parity check).

`--dry-gloo` (CPU test of the launcher / rank / scatter / gather logic only): gloo backend, CPU tensors, a tiny workload,
and the block codec taken from the module named by RCX_BENCH_DRY_CODEC (the test suite supplies it; this file never
imports the oracle outside cpu_baseline).
"""
import argparse
import ctypes as C
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BLOCK = 65536
NBLOCKS = 4096
HEADLINE_KERNEL_SOURCES = ["k_lz4_decode_v4.hip", "k_lz4_decode_v5.hip", "k_lz4_emit6.hip", "k_lz4_decode_v8.hip", "rcx_dev.h"]


def kernel_source_hash():
    """sha256 over the sources of the headline kernel: a PMC traffic figure is only quoted for the code it was measured on."""
    h = hashlib.sha256()
    for f in HEADLINE_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "rust_compress_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------ engines
class GpuEngine:
    """The product: HIP kernels through the C-ABI, tensors in HBM."""
    name = "gpu"

    def __init__(self, args, local_rank):
        import torch
        import rust_compress_amd as R
        from rust_compress_amd import _native as N
        self.torch, self.R, self.N = torch, R, N
        # RCX_BENCH_SHARE_GPU=1: the ranks share the GPUs there are (a check of the N-rank path on a one-GPU box; with
        # RCX_BENCH_BACKEND=gloo, since RCCL refuses two ranks on one device) -- its rates are not a scaling measurement
        share = bool(os.environ.get("RCX_BENCH_SHARE_GPU"))
        torch.cuda.set_device(local_rank % torch.cuda.device_count() if share else local_rank)
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.ctx = R.Context(torch.cuda.current_device())
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.ctx.set_variant(N.LZ4_DECODE, args.variant)
        self.backend = os.environ.get("RCX_BENCH_BACKEND", "nccl")
        self.shared_gpu = share

    def sync(self):
        self.torch.cuda.synchronize()

    def make_workload(self, kind, nblocks, seed):
        dec, raw, cb, ob = make_workload(self.R, self.ctx, self.torch, self.dev, kind, nblocks, seed)
        return {"dec": dec, "raw": raw, "comp_bytes": cb, "out_bytes": ob, "nblocks": nblocks}

    def check(self, wl, reps=12):
        """Parity, untimed: the decoded bytes == the synthetic source -- `reps` times over, each into an output buffer refilled
        with a different byte first.  The decoder is two waves per block talking through an LDS ring under issue priorities:
        a race would show as a run that differs, a stale byte as one that only passes over the previous run's output."""
        torch, dec, nb = self.torch, wl["dec"], wl["nblocks"]
        for r in range(reps):
            dec.out_base.fill_((0x5A + 37 * r) & 0xFF)
            dec.status.fill_(-1)
            self.ctx.launch_dev(self.N.LZ4_DECODE, dec)
            torch.cuda.synchronize()
            assert int(dec.status[:nb].abs().max()) == 0, "decode status != OK (run %d)" % r
            assert bool((dec.out_len[:nb] == BLOCK).all())
            assert torch.equal(dec.out_base[: nb * BLOCK], wl["raw"][: nb * BLOCK]), "GPU decode != source (run %d)" % r

    def decode(self, wl):
        self.ctx.launch_dev(self.N.LZ4_DECODE, wl["dec"])

    def events(self, n):
        return [self.torch.cuda.Event(enable_timing=True) for _ in range(n)]

    def compressed(self, wl):
        dec = wl["dec"]
        return dec.in_base, dec.in_off.cpu().numpy(), dec.in_len.cpu().numpy()

    def decode_packed(self, local, loff, llen, nblk, desc=None):
        """decode blocks that arrived packed (end-to-end leg) -> (out tensor, offsets, lengths as numpy).  `desc`: the
        descriptors as they arrived on the device ([offsets | lengths], int64) -- the kernel reads them where they are; the
        result arrays are persistent, so a call is the launch, two small read-backs (statuses, lengths) and one sync."""
        torch, N = self.torch, self.N
        ar = np.arange(nblk, dtype=np.int64)
        if nblk == 0:
            return torch.zeros(64, dtype=torch.uint8, device=self.dev), ar, np.zeros(0, np.int64)
        st = getattr(self, "_e2e", None)
        if st is None or st["n"] != nblk:
            i64 = lambda a: torch.tensor(np.asarray(a, dtype=np.int64), dtype=torch.int64, device=self.dev)
            st = {"n": nblk, "out": torch.zeros(nblk * BLOCK + 64, dtype=torch.uint8, device=self.dev),
                  "ooff": i64(ar * BLOCK), "ocap": i64(np.full(nblk, BLOCK)),
                  "olen": torch.zeros(nblk, dtype=torch.int64, device=self.dev), "used": torch.zeros(nblk, dtype=torch.int64, device=self.dev),
                  "stat": torch.full((nblk,), -1, dtype=torch.int32, device=self.dev), "aux": torch.zeros(nblk, dtype=torch.int32, device=self.dev),
                  "h_olen": torch.zeros(nblk, dtype=torch.int64).pin_memory(), "h_stat": torch.zeros(nblk, dtype=torch.int32).pin_memory()}
            self._e2e = st
        if desc is None:
            desc = torch.from_numpy(np.concatenate([np.asarray(loff, dtype=np.int64), np.asarray(llen, dtype=np.int64)])).to(self.dev)
        st["desc"] = desc                                       # (kept alive while the kernel reads it)
        # (the kernels read a block's last bytes with 16-byte loads only inside the block: no padding of the packed bytes)
        db = N.DevBatch(local.data_ptr(), desc.data_ptr(), desc.data_ptr() + 8 * nblk, st["out"].data_ptr(), st["ooff"].data_ptr(), st["ocap"].data_ptr(),
                        st["olen"].data_ptr(), st["used"].data_ptr(), st["stat"].data_ptr(), st["aux"].data_ptr(), nblk)
        self.ctx._chk(N.lib().rcx_launch_dev(self.ctx._h, N.LZ4_DECODE, C.byref(db), C.c_void_p(None), 0))
        st["h_stat"].copy_(st["stat"], non_blocking=True)
        st["h_olen"].copy_(st["olen"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        assert not st["h_stat"].numpy().any()
        return st["out"], ar * BLOCK, st["h_olen"].numpy().copy()

    def block_crcs(self, base, offs, lens):
        """CRC-32 of every block (the product's k_crc32): what the end-to-end leg compares, block by block"""
        torch, R = self.torch, self.R
        n = len(lens)
        if n == 0:
            return np.zeros(0, np.uint32)
        i64 = lambda a: torch.tensor(np.asarray(a, dtype=np.int64), dtype=torch.int64, device=self.dev)
        dummy = torch.zeros(64, dtype=torch.uint8, device=self.dev)
        db = R.DeviceBatch(base, i64(offs), i64(lens), dummy, i64(np.zeros(n)), i64(np.zeros(n)))
        self.ctx.launch_dev(self.N.CRC32, db)
        torch.cuda.synchronize()
        return db.aux[:n].cpu().numpy().astype(np.uint32)

    def close(self):
        self.ctx.close()


class DryEngine:
    """CPU stand-in for the test of the multi-rank plumbing (--dry-gloo): the codec comes from RCX_BENCH_DRY_CODEC."""
    name = "dry"

    def __init__(self, args, local_rank):
        import torch
        self.torch = torch
        self.dev = torch.device("cpu")
        self.backend = "gloo"
        self.codec = importlib.import_module(os.environ["RCX_BENCH_DRY_CODEC"])

    def sync(self):
        pass

    def make_workload(self, kind, nblocks, seed):
        from rust_compress_amd import batch as B
        blobs, raws = self.codec.make_blocks(kind, nblocks, BLOCK, seed)
        base, off, lens = B.pack(blobs)
        return {"blobs": blobs, "raws": raws, "base": self.torch.from_numpy(base), "off": off, "lens": lens,
                "comp_bytes": int(sum(map(len, blobs))), "out_bytes": int(sum(map(len, raws))), "nblocks": nblocks, "out": None}

    def check(self, wl):
        assert [self.codec.decode(b, BLOCK) for b in wl["blobs"]] == wl["raws"]

    def decode(self, wl):
        wl["out"] = [self.codec.decode(b, BLOCK) for b in wl["blobs"]]

    def events(self, n):
        class Ev:
            def record(s): s.t = time.perf_counter()
            def elapsed_time(s, o): return (o.t - s.t) * 1e3
        return [Ev() for _ in range(n)]

    def compressed(self, wl):
        return wl["base"], wl["off"], wl["lens"]

    def decode_packed(self, local, loff, llen, nblk, desc=None):
        buf = local.numpy()
        outs = [self.codec.decode(buf[int(o):int(o) + int(l)].tobytes(), BLOCK) for o, l in zip(loff, llen)]
        lens = np.array([len(o) for o in outs], dtype=np.int64)
        out = self.torch.from_numpy(np.frombuffer(b"".join(outs) + b"\0", dtype=np.uint8).copy())
        return out, np.concatenate([[0], np.cumsum(lens)[:-1]]) if nblk else np.zeros(0, np.int64), lens

    def block_crcs(self, base, offs, lens):
        import zlib
        buf = base.numpy()
        return np.array([zlib.crc32(buf[int(o):int(o) + int(l)].tobytes()) for o, l in zip(offs, lens)], dtype=np.uint32)

    def close(self):
        pass


def make_workload(R, ctx, torch, dev, kind, nblocks, seed):
    """-> (decode DeviceBatch, raw tensor, comp_bytes, out_bytes)"""
    from rust_compress_amd import synth, _native as N
    raw_np = synth.gen_blocks(kind, nblocks, BLOCK, seed)
    raw = torch.from_numpy(raw_np).to(dev)
    bound = int(N.lib().rcx_lz4_compression_bound(BLOCK))
    slot = (bound + 63) // 64 * 64
    i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
    ar = np.arange(nblocks, dtype=np.int64)
    enc = R.DeviceBatch(raw, i64(ar * BLOCK), i64(np.full(nblocks, BLOCK)),
                        torch.zeros(nblocks * slot + 64, dtype=torch.uint8, device=dev), i64(ar * slot),
                        i64(np.full(nblocks, slot)))
    scratch = torch.empty(ctx.scratch_bytes(N.LZ4_ENCODE, nblocks, BLOCK) + 64, dtype=torch.uint8, device=dev)
    ctx.launch_dev(N.LZ4_ENCODE, enc, scratch)
    torch.cuda.synchronize()
    assert int(enc.status.abs().max()) == 0, "lz4 encode failed"
    del scratch
    comp_len = enc.out_len[:nblocks].clone()
    dec = R.DeviceBatch(enc.out_base, enc.out_off, comp_len,
                        torch.zeros(nblocks * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK),
                        i64(np.full(nblocks, BLOCK)))
    return dec, raw, int(comp_len.sum()), nblocks * BLOCK


def time_steps(eng, wl, steps, warmup, dist):
    """W untimed steps, then exactly K steps between barrier + device sync on both sides; per-step device time from events
    recorded on the stream the kernels are launched on."""
    # (a barrier among ONE rank waits for nobody and costs a collective's launch and host sync -- 0.1-0.45 ms inside a timed region of
    # 10 ms, 1-4 % of ms_per_step at K = 20 -- so with world 1 the device syncs alone bracket the region)
    peers = dist is not None and dist.get_world_size() > 1
    for _ in range(warmup):
        eng.decode(wl)
    eng.sync()
    if peers:
        dist.barrier()
    eng.sync()
    evs = eng.events(steps + 1)
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        eng.decode(wl)
        evs[i + 1].record()
    eng.sync()
    if peers:
        dist.barrier()
    eng.sync()
    wall = time.perf_counter() - t0
    kern_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    return wall, float(np.mean(kern_ms)), float(np.median(kern_ms))


def cpu_baseline(dec, raw, torch, nblocks, budget_s=10.0):
    """The oracle (a line-faithful port of the reference's CPU decoder) on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from rust_compress_amd import _native as N
    O.build()
    in_base = dec.in_base.cpu().numpy()
    in_off = dec.in_off.cpu().numpy().astype(np.uint64)
    in_len = dec.in_len.cpu().numpy().astype(np.uint64)
    out_off = dec.out_off.cpu().numpy().astype(np.uint64)
    out_cap = dec.out_cap.cpu().numpy().astype(np.uint64)
    out = np.zeros(nblocks * BLOCK + 64, dtype=np.uint8)
    cores = os.cpu_count() or 1
    total_s, reps, nbytes = 0.0, 0, 0
    while total_s < budget_s and reps < 2000:                   # ~10 s of CPU work on the workload's own blocks (the contract's bounded sample)
        secs, out_len, _, status = O.batch_run(N.LZ4_DECODE, in_base, in_off, in_len, out, out_off, out_cap, threads=cores)
        assert not status.any()
        total_s += secs
        reps += 1
        nbytes += int(out_len.sum())
    ok = bool(np.array_equal(out[: nblocks * BLOCK], raw.cpu().numpy()[: nblocks * BLOCK]))
    secs1, out_len1, _, _ = O.batch_run(N.LZ4_DECODE, in_base, in_off[:256], in_len[:256], out, out_off[:256], out_cap[:256], threads=1)
    return {"value": round(nbytes / total_s / 2**30, 3), "unit": "GiB/s", "cores": cores, "kind": "port",
            "sample": "all %d blocks x %d passes (%.1f s) on %d threads; 1-thread rate on 256 blocks: %.3f GiB/s; "
                      "oracle output == GPU input data: %s" % (nblocks, reps, total_s, cores,
                                                                 float(out_len1.sum()) / secs1 / 2**30, ok)}


# ------------------------------------------------------------------------------------------------ end-to-end leg
def end_to_end(eng, wl, dist, rank, world, reps=3):
    """Root holds every rank's compressed blocks; timed: scatter (compressed) -> decode -> gather (decoded)."""
    import rust_compress_amd.dist as D
    torch = eng.torch
    base, off, lens = eng.compressed(wl)
    nb = wl["nblocks"]
    # untimed set-up: collect every rank's compressed blocks on the root, packed
    packed, clens = D.gather_blocks(base, off, lens, np.arange(world + 1, dtype=np.int64) * nb, root=0)
    # what must come back, block by block: CRC-32 of every rank's source blocks, in global block order (all_gather of small arrays)
    if eng.name == "gpu":
        mine = eng.block_crcs(wl["raw"], np.arange(nb, dtype=np.int64) * BLOCK, np.full(nb, BLOCK, dtype=np.int64))
    else:
        import zlib
        mine = np.array([zlib.crc32(r) for r in wl["raws"]], dtype=np.uint32)
    crc_all = [torch.zeros(nb, dtype=torch.int64, device=eng.dev) for _ in range(world)]
    dist.all_gather(crc_all, torch.from_numpy(mine.astype(np.int64)).to(eng.dev))
    want = torch.cat(crc_all).cpu().numpy().astype(np.uint32)
    roff = bounds = None
    if rank == 0:
        roff = np.concatenate([[0], np.cumsum(clens)[:-1]])
        bounds = D.partition(np.full(world * nb, BLOCK), world)
    best = None
    def fence():                                               # phase boundary: device idle on every rank (one rank: nobody to wait for)
        eng.sync()
        if world > 1:
            dist.barrier()
    for rep in range(reps + 1):
        times = []
        fence(); t0 = time.perf_counter()
        local, loff, llen, bnd, ddesc = D.scatter_blocks(packed, roff, clens, bounds, root=0, device=eng.dev, with_desc=True)
        fence(); t1 = time.perf_counter()
        nblk = int(bnd[rank + 1] - bnd[rank])
        out, ooff, olen = eng.decode_packed(local, loff, llen, nblk, ddesc)
        fence(); t2 = time.perf_counter()
        got, glens = D.gather_blocks(out, ooff, olen, bnd, root=0)
        fence(); t3 = time.perf_counter()
        times = [t1 - t0, t2 - t1, t3 - t2]
        if rep and (best is None or sum(times) < sum(best)):      # the first pass pays allocations and RCCL channel set-up
            best = times
    tt = torch.tensor(best, dtype=torch.float64, device=eng.dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    res = None
    if rank == 0:
        goff = np.concatenate([[0], np.cumsum(glens)[:-1]]).astype(np.int64)
        ok = bool(len(glens) == world * nb and (glens == BLOCK).all() and np.array_equal(eng.block_crcs(got, goff, glens), want))   # every block, in its place
        sc, de, ga = [float(x) for x in tt.tolist()]
        res = {"scatter_ms": round(sc * 1e3, 3), "decode_ms": round(de * 1e3, 3), "gather_ms": round(ga * 1e3, 3),
               "value": round(float(glens.sum()) / (sc + de + ga) / 2**30, 3), "unit": "GiB/s decoded, root scatter + decode + root gather",
               "bytes_scattered": int(clens.sum()) - int(clens[:nb].sum()), "bytes_gathered": int(glens.sum()) - nb * BLOCK,
               "verified": ok, "transport": "%s grouped isend/irecv (batch_isend_irecv), %d ranks" % (eng.backend, world)}
    return res


def rccl_self_sendrecv(eng, dist, rank, nbytes=64 << 20):
    """One grouped send+recv of a device tensor to this rank itself: exercises the RCCL point-to-point path with device
    tensors even on a one-GPU box (world 1 has no peer to talk to)."""
    torch = eng.torch
    try:
        a = torch.arange(nbytes // 8, dtype=torch.int64, device=eng.dev)
        b = torch.zeros_like(a)
        for rep in range(3):
            eng.sync(); t0 = time.perf_counter()
            for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, rank), dist.P2POp(dist.irecv, b, rank)]):
                r.wait()
            eng.sync(); t = time.perf_counter() - t0
        return {"bytes": nbytes, "GB/s": round(nbytes / t / 1e9, 2), "ok": bool(torch.equal(a, b))}
    except Exception as e:                                     # pragma: no cover (backend dependent)
        return {"error": str(e)[:200]}


def host_path(eng, wl, reps=6):
    """The PCIe-inclusive rate of the same workload through the host-memory entry point (rcx_lz4_decode_batch, RCX_MEM_HOST): what
    a caller of compress::lz4::decode_block (src/lz4.rs:602-611) with its blocks in host memory gets.  The compressed blocks are
    packed as an LZ4 frame holds them, both buffers page-locked: the decoder then stores what leaves its window straight into the
    caller's buffer (no device-to-host copy behind the launch) and the compressed bytes travel in as block ranges while the launch
    already decodes the ranges before them (rcx_api.hip run_batch; `ms_plain_copies`: one copy each way around the launch, what a
    pageable buffer gets).  The output buffer is wiped before the last, verified repetition.  Reported beside
    `value`, never as `value` (the measurement contract: inputs resident in HBM)."""
    torch, N = eng.torch, eng.N
    dec, nb = wl["dec"], wl["nblocks"]
    u64 = lambda t: np.ascontiguousarray(t.cpu().numpy().astype(np.uint64))
    slots = dec.in_base.cpu().numpy()
    so, in_len, out_off, out_cap = u64(dec.in_off), u64(dec.in_len), u64(dec.out_off), u64(dec.out_cap)
    in_off = np.zeros(nb, np.uint64)
    in_off[1:] = np.cumsum((in_len[:-1].astype(np.int64) + 15) & ~15).astype(np.uint64)
    total_in = int(in_off[-1]) + int(in_len[-1])
    inb = torch.zeros(total_in + 64, dtype=torch.uint8).pin_memory()
    ib = inb.numpy()
    for i in range(nb):
        ib[int(in_off[i]): int(in_off[i]) + int(in_len[i])] = slots[int(so[i]): int(so[i]) + int(in_len[i])]
    outb = torch.zeros(nb * BLOCK + 64, dtype=torch.uint8).pin_memory()
    out_len, in_used, status = np.zeros(nb, np.uint64), np.zeros(nb, np.uint64), np.zeros(nb, np.int32)
    p = lambda a: a.ctypes.data
    b = N.Batch(inb.data_ptr(), p(in_off), p(in_len), outb.data_ptr(), p(out_off), p(out_cap), p(out_len), p(in_used), p(status), nb, N.MEM_HOST)
    def timed(plain):
        N.lib().rcx_ctx_set_param(eng.ctx._h, N.LZ4_DECODE, 1 if plain else 0)
        ts = []
        try:
            for it in range(reps):
                if it == reps - 1:
                    outb.zero_(); status[:] = -9
                t0 = time.perf_counter()
                rc = N.lib().rcx_lz4_decode_batch(eng.ctx._h, C.byref(b))
                ts.append(time.perf_counter() - t0)
                assert rc == 0 and not status.any()
        finally:
            N.lib().rcx_ctx_set_param(eng.ctx._h, N.LZ4_DECODE, 0)
        return float(np.median(ts[1:])), bool(np.array_equal(outb.numpy()[: nb * BLOCK], wl["raw"].cpu().numpy()[: nb * BLOCK]))
    t_plain, ok_plain = timed(True)
    t, ok = timed(False)
    return {"GiB/s": round(wl["out_bytes"] / t / 2**30, 2), "ms": round(t * 1e3, 3), "ms_plain_copies": round(t_plain * 1e3, 3), "bytes_in": total_in,
            "bytes_out": int(wl["out_bytes"]), "verified": ok and ok_plain,
            "what": "rcx_lz4_decode_batch, RCX_MEM_HOST, page-locked host buffers: decoded bytes stored straight into the caller's buffer by the launch, compressed bytes arriving in ranges under it"}


def sustained(eng, wl, seconds=2.0):
    """The headline launch back to back for ~2 s: the rate once clocks and power have settled (after an idle spell the first ~20
    launches run 5 % slower: benchmarks/r5_clock_ramp.py), and two seconds of GPU activity an outside sampler can see.  Events
    bracket groups of 50 launches; reported beside `value`, which stays the contract's W warm-up + K timed steps."""
    torch = eng.torch
    group, ms = 50, []
    eng.sync()
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        e0, e1 = eng.events(2)
        e0.record()
        for _ in range(group):
            eng.decode(wl)
        e1.record()
        eng.sync()
        ms.append(e0.elapsed_time(e1) / group)
    ms = np.array(ms[1:] if len(ms) > 1 else ms)
    alg = wl["comp_bytes"] + wl["out_bytes"]
    return {"launches": int(group * (len(ms) + 1)), "kernel_ms_avg": round(float(ms.mean()), 4), "kernel_ms_min_group": round(float(ms.min()), 4),
            "GiB/s": round(wl["out_bytes"] / (float(ms.mean()) * 1e-3) / 2**30, 2), "roofline_frac": round(alg / (float(ms.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}


def hbm_ceiling(torch, dev, ctx=None, nbytes=1 << 30, reps=10):
    """What a plain copy reaches on this device, two ways: the library's own kernel (rcx_hbm_copy_probe: 16 bytes a thread, a
    workgroup per 4 KiB, one launch) -- the figure quoted as "achievable" -- and torch's copy_, which reads ~15 % lower."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev).fill_(1)
    b = torch.empty_like(a)
    b.copy_(a); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    del a, b
    torch.cuda.empty_cache()
    res = {"GB/s": round(2 * nbytes / t / 1e9, 1), "what": "device-to-device copy of 1 GiB (read + write bytes / time), torch copy_", "spec_peak_GB/s": HBM_PEAK_GBS}
    if ctx is not None:
        from rust_compress_amd import _native as N
        g = C.c_double(0)
        rc = N.lib().rcx_hbm_copy_probe(ctx._h, nbytes, reps, C.byref(g))
        if rc == 0:
            res["torch_copy_GB/s"] = res["GB/s"]
            res["GB/s"] = round(g.value, 1)
            res["what"] = "device-to-device copy of 1 GiB (read + write bytes / time): rcx_hbm_copy_probe, the library's own 16-bytes-a-lane kernel; torch copy_ beside it"
    return res


def dry_sharded_pipeline(eng, dist, rank, world, nblocks=11, block=4096):
    """BASELINE config 5's sharding on CPU ranks (--dry-gloo): ONE stream, its RCXQ container split by block ranges
    (dist.partition + pipeline.split_container), the shards scattered, decoded per rank, the decoded ranges gathered and
    compared with the source byte for byte; and the other way round -- raw ranges scattered, encoded per rank, the shards'
    containers gathered and joined: byte for byte the container one device writes.  The block codec is the test suite's."""
    import rust_compress_amd.dist as D
    from rust_compress_amd import pipeline as P, synth
    torch = eng.torch
    codec = eng.codec
    total = nblocks * block - 1234                              # a ragged last block
    lens = np.array([min(block, total - i) for i in range(0, total, block)], dtype=np.int64)
    bounds = D.partition(lens, world)
    data = whole = None
    t8 = lambda b: torch.from_numpy(np.frombuffer(bytes(b) + b"\0", dtype=np.uint8).copy())
    if dist is None:                                            # (one rank, no process group: the same calls without transport)
        data = synth.gen("text", total, 0xC5).tobytes()
        whole = codec.pipe_encode(data, block)
        ok = P.join_containers(P.split_container(whole, bounds)) == whole and codec.pipe_decode(whole) == data
        return {"config": 5, "n_gpus": 1, "sharded_container_verified": bool(ok), "joined_equals_single_device": bool(ok), "blocks": int(len(lens))}
    # 1. decode side: container shards root -> ranks, decoded ranges ranks -> root
    shards, slen = None, None
    if rank == 0:
        data = synth.gen("text", total, 0xC5).tobytes()
        whole = codec.pipe_encode(data, block)
        parts = P.split_container(whole, bounds)
        slen = np.array([len(x) for x in parts], dtype=np.int64)
        shards = t8(b"".join(parts))
    sb = np.arange(world + 1, dtype=np.int64)                   # one "block" per rank: its shard
    soff = None if slen is None else np.concatenate([[0], np.cumsum(slen)[:-1]])
    local, loff, llen, _ = D.scatter_blocks(shards, soff, slen, sb, root=0, device=eng.dev)
    mine = local.numpy()[: int(llen[0])].tobytes() if len(llen) else P.build_container(block, 16, [], [], [], b"")
    dec = codec.pipe_decode(mine)
    a, b = int(bounds[rank]), int(bounds[rank + 1])
    got, glens = D.gather_blocks(t8(dec), np.concatenate([[0], np.cumsum(lens[a:b])[:-1]]) if b > a else np.zeros(0, np.int64), lens[a:b], bounds, root=0)
    # 2. encode side: raw ranges root -> ranks, containers ranks -> root, joined
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    lraw, lo_, ll_, _ = D.scatter_blocks(t8(data) if rank == 0 else None, offs, lens, bounds, root=0, device=eng.dev)
    myc = codec.pipe_encode(lraw.numpy()[: int(ll_.sum())].tobytes(), block) if len(ll_) else P.build_container(block, 16, [], [], [], b"")
    cg, clens = D.gather_blocks(t8(myc), np.zeros(1, np.int64), np.array([len(myc)], dtype=np.int64), sb, root=0)
    if rank != 0:
        return None
    ok_dec = bool(np.array_equal(glens, lens) and got.numpy()[: total].tobytes() == data)
    buf = cg.numpy()
    co = np.concatenate([[0], np.cumsum(clens)[:-1]])
    joined = P.join_containers([buf[int(o):int(o) + int(l)].tobytes() for o, l in zip(co, clens)])
    return {"config": 5, "n_gpus": world, "sharded_container_verified": ok_dec, "joined_equals_single_device": bool(joined == whole),
            "blocks": int(len(lens)), "block_ranges": np.diff(bounds).tolist(), "container_bytes": len(whole)}


def summary_of(res):
    """The line's key numbers once more, compact, as its last key (a log that keeps the last 2000 characters of the job's output
    still holds every config's time): headline, the other distributions, configs 3 / 3g / 4 / 5, host path, CPU baseline."""
    g = lambda d, *ks: (d.get(ks[0]) if len(ks) == 1 else g(d.get(ks[0]) or {}, *ks[1:])) if isinstance(d, dict) else None
    out = {"lz4_decode_ms": res.get("ms_per_step"), "lz4_decode_GiB/s": res.get("value"), "roofline_frac": g(res, "roofline", "frac"),
           "sustained_ms": g(res, "sustained", "kernel_ms_avg"), "runs_ms": g(res, "per_distribution", "G-runs", "ms_per_step"),
           "rand_ms": g(res, "per_distribution", "G-rand", "ms_per_step"), "e2e_GiB/s": g(res, "end_to_end", "value"),
           "host_path_GiB/s": g(res, "host_path", "GiB/s"), "cpu_GiB/s": g(res, "cpu_baseline", "value"),
           "one_lz4_block_us": g(res, "single_stream", "lz4_block_64KiB", "host_us"), "one_deflate_MiB_us": g(res, "single_stream", "deflate_stream_1MiB", "host_us"),
           "one_bwt_block_fwd_us": g(res, "single_stream", "bwt_block_256KiB", "forward_host_us"), "hbm_copy_GB/s": g(res, "hbm_ceiling_measured", "GB/s")}
    for o in res.get("other_configs") or []:
        if not isinstance(o, dict):
            continue
        c = o.get("config")
        if "error" in o:
            out["cfg_%s" % o.get("leg", c)] = "error"
        elif c in (3, "3g"):
            out["cfg%s_ms" % c] = o.get("ms"); out["cfg%s_frac" % c] = g(o, "roofline", "frac")
            if g(o, "host_path", "GiB/s") is not None:
                out["cfg%s_host_GiB/s" % c] = g(o, "host_path", "GiB/s")
        elif c == 4:
            k = "text" if "G-text" in str(o.get("workload")) else "dna4"
            out["cfg4_%s_fwd_ms" % k] = o.get("forward_ms"); out["cfg4_%s_inv_ms" % k] = o.get("inverse_ms")
        elif c == 5:
            out["cfg5_enc_s"] = o.get("encode_s"); out["cfg5_dec_s"] = o.get("decode_s")
    return {k: v for k, v in out.items() if v is not None}


# ------------------------------------------------------------------------------------------------ side legs
class SideLegs:
    """Everything bench.py measures AFTER the headline (other distributions, end_to_end, configs 3-5, CPU baselines) runs
    through here, so that none of it can cost the line the driver reads:

    * `run(name, fn)`: every rank catches its own exception; with peers, ONE all_reduce(MAX) of an error flag then decides
      together whether the leg counts as done.  A leg that failed on any rank is reported as {"error": ...}, and the legs
      after it that need the process group are skipped (a rank that left a scatter half way has unmatched sends behind it:
      the group's state is unknown) -- local legs still run.
    * the process group carries a timeout (RCX_BENCH_PG_TIMEOUT seconds), so a peer that never shows up is an exception on
      gloo; on RCCL, where a stuck collective is a stuck device, the WATCHDOG thread is what ends it: RCX_BENCH_SIDE_TIMEOUT
      seconds after the headline it prints the stashed line (rank 0) with the legs finished so far and exits the process.
    * `finish()`: the one JSON line, printed once (the watchdog and the main thread share a lock)."""

    def __init__(self, eng, dist, rank, world, res):
        import threading
        self.eng, self.dist, self.rank, self.world, self.res = eng, dist, rank, world, res
        self.broken = None            # why the process group is not used any more
        self.failed = []
        self.lock = threading.Lock()
        self.printed = False
        self.current = None

    def run(self, name, fn, collective=True):
        multi = self.world > 1 and self.dist is not None
        if collective and multi and self.broken:
            return {"error": "skipped: " + self.broken}
        self.current = name
        out, err = None, None
        try:
            out = fn()
        except Exception as e:                                  # (KeyboardInterrupt / SystemExit pass)
            err = "%s: %s" % (type(e).__name__, str(e)[:300])
        if collective and multi:
            torch = self.eng.torch
            try:
                flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float32, device=self.eng.dev)
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
                if float(flag.item()) > 0 and err is None:
                    err = "another rank failed in this leg"
                if float(flag.item()) > 0:
                    self.broken = "the leg '%s' failed on some rank; the process group is not used after that" % name
            except Exception as e:
                self.broken = "process group unusable after the leg '%s' (%s)" % (name, str(e)[:160])
                err = err or self.broken
        self.current = None
        if err is not None:
            self.failed.append(name)
            print("bench.py: rank %d: side leg %s failed: %s" % (self.rank, name, err), file=sys.stderr, flush=True)
            return {"error": err}
        return out

    def _emit(self, note=None):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            if self.res is not None:
                if self.failed:
                    self.res["side_legs_failed"] = list(self.failed)
                if note:
                    self.res["side_legs_note"] = note
                self.res["summary"] = summary_of(self.res)      # the LAST key: a record that keeps only the line's tail keeps this
                print(json.dumps(self.res), flush=True)

    def start_watchdog(self, seconds):
        import threading

        def fire():
            self._emit("side legs did not finish within %.0f s (running: %s): the line carries what was finished" % (seconds, self.current))
            try:
                sys.stdout.flush(); sys.stderr.flush()
            finally:
                os._exit(0 if self.res is not None or self.rank != 0 else 1)
        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def finish(self):
        # RCCL writes its version banner to stdout through C stdio, which is flushed at exit -- after anything Python prints.
        # Every rank pushes it out first; rank 0 prints the JSON line after a barrier, as the last line of the job's stdout.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        multi = self.dist is not None
        if multi and not self.broken:
            try:
                self.dist.barrier()
            except Exception as e:
                self.broken = "final barrier: %s" % str(e)[:120]
        self._emit()
        if getattr(self, "timer", None) is not None:
            self.timer.cancel()
        if multi and not self.broken:
            try:
                self.dist.barrier()
                self.dist.destroy_process_group()
            except Exception:
                pass
        elif multi:
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)                                         # (a broken group's destructor may wait for peers that are gone)


# ------------------------------------------------------------------------------------------------ launcher
def self_spawn(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start one rank per GPU and pass their output through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=25)     # (after the parity check's idle gaps the first ~20 launches run ~2-5 % slower: benchmarks/r5_clock_ramp.py)
    ap.add_argument("--kind", default="text", help="synthetic distribution: text|runs|rand|mix")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (A/B)")
    ap.add_argument("--nblocks", type=int, default=NBLOCKS)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the scatter / decode / gather leg")
    ap.add_argument("--no-others", action="store_true", help="skip BASELINE configs 3-5 and the HBM ceiling")
    ap.add_argument("--no-dists", action="store_true", help="skip config 2's other distributions (G-runs, G-rand)")
    ap.add_argument("--others-scale", type=float, default=1.0, help="fraction of the full size of configs 3-5")
    ap.add_argument("--extras", action="store_true", help="also time the other distributions / variants")
    ap.add_argument("--dry-gloo", action="store_true", help="CPU test of the multi-rank plumbing (see the module docstring)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    eng = (DryEngine if args.dry_gloo else GpuEngine)(args, local_rank)
    torch = eng.torch
    dist = None
    want_dist = world > 1 or not args.no_e2e or os.environ.get("RCX_BENCH_FORCE_DIST")
    if want_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        try:
            import datetime
            # a peer that never arrives is an exception after this long, not a hang (SideLegs); RCCL's own watchdog would end the
            # process at this point, so it lies BEHIND the side legs' watchdog, which prints the line first
            pg_to = datetime.timedelta(seconds=float(os.environ.get("RCX_BENCH_PG_TIMEOUT", "1500")))
            if eng.backend == "nccl":
                dist.init_process_group("nccl", device_id=eng.dev, timeout=pg_to)
            else:
                dist.init_process_group("gloo", timeout=pg_to)
        except Exception as e:
            if world > 1:
                raise
            print("bench.py: process group unavailable (%s): end_to_end skipped" % str(e)[:120], file=sys.stderr)
            dist = None

    wl = eng.make_workload(args.kind, args.nblocks, 0x4C5A3401 + 7919 * rank)
    if os.environ.get("RCX_BENCH_EXPERIMENT_NOCHECK") and (args.variant in (21, 22, 41, 42, 43, 44, 45, 48, 49) or "RCX_XCUT" in os.environ.get("RCX_EXTRA_FLAGS", "")):
        print("bench.py: EXPERIMENT variant %d (part of the kernel disabled): output not checked, not a result" % args.variant, file=sys.stderr)
    else:
        eng.check(wl)                                           # parity (untimed): decoded bytes == the synthetic source on this rank
    comp_bytes, out_bytes = wl["comp_bytes"], wl["out_bytes"]

    wall, kern_ms, kern_med = time_steps(eng, wl, args.steps, args.warmup, dist)
    t = torch.tensor([wall], dtype=torch.float64, device=eng.dev)
    tot = torch.tensor([float(out_bytes), float(comp_bytes)], dtype=torch.float64, device=eng.dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    wall = float(t.item())
    total_out = float(tot[0].item())

    # ---- the headline is complete here.  Rank 0 stashes the line now; everything below is a SIDE LEG that runs under
    # SideLegs.run (every rank catches, one all_reduce(MAX) of an error flag decides together whether the ranks can go on
    # using the process group) and under a watchdog that prints the stashed line if the legs do not come back in time:
    # no failure or hang after this point can cost the headline value.
    comp_total = float(tot[1].item())
    res = None
    if rank == 0:
        alg_bytes = comp_bytes + out_bytes                     # per launch on this rank
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = traffic_src = None
        pmc = os.path.join(ROOT, "profiles", "pmc_lz4_decode.json")
        if os.path.exists(pmc) and not args.dry_gloo:           # only a figure measured on exactly these kernel sources is quoted
            try:
                j = json.load(open(pmc))
                if j.get("kernel_source_hash") == kernel_source_hash() and args.variant == 0 and args.kind == "text":
                    traffic = j.get("hbm_bytes_per_launch")
                    traffic_src = {"file": "profiles/pmc_lz4_decode.json", "kernel_source_hash": j.get("kernel_source_hash"), "date": j.get("date")}
            except Exception:
                traffic = None
        res = {
            "metric": "GiB/s decoded output (batched blocks)",
            "value": round(total_out * args.steps / wall / 2**30, 3),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic (dry run on CPU: plumbing test, not a measurement)" if args.dry_gloo else "synthetic",
            "config": {"workload": "LZ4 block decode, %d independent 64 KiB blocks per GPU (BASELINE configs[1])" % args.nblocks,
                       "distribution": "G-%s" % args.kind, "lz4_ratio": round(out_bytes / comp_bytes, 3),
                       "kernel_variant": args.variant, "parallelism": "blocks sharded, %d per rank, no collective" % args.nblocks},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_avg": round(kern_ms, 4),
                         "kernel_ms_median": round(kern_med, 4)},
            "per_distribution": {"G-" + args.kind: {"GiB/s": round(total_out * args.steps / wall / 2**30, 3), "ms_per_step": round(wall / args.steps * 1e3, 4),
                                                    "kernel_ms_avg": round(kern_ms, 4), "lz4_ratio": round(out_bytes / comp_bytes, 3),
                                                    "roofline_frac": round(achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": alg_bytes}},
            **({"shared_gpu_check": "the %d ranks share the box's GPUs over %s (RCX_BENCH_SHARE_GPU): a check of the N-rank path, not a scaling measurement"
                                    % (world, eng.backend)} if getattr(eng, "shared_gpu", False) else {}),
            "end_to_end": None,
        }
    legs = SideLegs(eng, dist, rank, world, res)
    legs.start_watchdog(float(os.environ.get("RCX_BENCH_SIDE_TIMEOUT", "900")))

    # ---- config 2's other distributions (reported separately), same timing discipline, every rank
    def leg_dists():
        per_dist = {}
        for kind in ("runs", "rand"):
            w2 = eng.make_workload(kind, args.nblocks, 0x4C5A3401 + 7919 * rank + len(kind))
            eng.check(w2)
            wall2, km2, _ = time_steps(eng, w2, max(5, args.steps // 2), 2, dist if world > 1 else None)
            tt2 = torch.tensor([wall2, float(w2["out_bytes"]), float(w2["comp_bytes"])], dtype=torch.float64, device=eng.dev)
            if dist is not None:
                mx = tt2.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
                sm = tt2.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
                wall2, tot_out2 = float(mx[0]), float(sm[1])
            else:
                tot_out2 = float(w2["out_bytes"])
            st2 = max(5, args.steps // 2)
            alg2 = w2["comp_bytes"] + w2["out_bytes"]
            per_dist["G-" + kind] = {"GiB/s": round(tot_out2 * st2 / wall2 / 2**30, 2), "ms_per_step": round(wall2 / st2 * 1e3, 4), "kernel_ms_avg": round(km2, 4),
                                     "lz4_ratio": round(w2["out_bytes"] / w2["comp_bytes"], 3),
                                     "roofline_frac": round(alg2 / (km2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": alg2}
            del w2
            torch.cuda.empty_cache()
        return per_dist
    if not args.no_dists and not args.dry_gloo:
        pdist = legs.run("per_distribution", leg_dists)
        if rank == 0:
            res["per_distribution"].update(pdist if "error" not in pdist else {"others": pdist})

    if dist is not None and not args.no_e2e:
        e2e = legs.run("end_to_end", lambda: end_to_end(eng, wl, dist, rank, world))
        if rank == 0:
            res["end_to_end"] = e2e
        if eng.backend == "nccl":
            sp = legs.run("rccl_self_sendrecv", lambda: rccl_self_sendrecv(eng, dist, rank), collective=False)
            if rank == 0:
                res["rccl_self_sendrecv"] = sp

    if not args.dry_gloo and not args.no_dists:
        sus = legs.run("sustained", lambda: sustained(eng, wl), collective=False)
        if rank == 0:
            res["sustained"] = sus
    if rank == 0 and world == 1 and not args.no_others and not args.dry_gloo:
        res["host_path"] = legs.run("host_path", lambda: host_path(eng, wl), collective=False)
    if rank == 0 and world == 1 and not args.no_cpu and not args.dry_gloo:
        res["cpu_baseline"] = legs.run("cpu_baseline", lambda: cpu_baseline(wl["dec"], wl["raw"], torch, args.nblocks), collective=False)
    if rank == 0 and world == 1 and not args.no_others and not args.dry_gloo:
        # batch of ONE and the batch-size sweep (benchmarks/single_stream.py): what a caller with a single stream gets, and from which
        # batch size the GPU path beats the host -- its oracle timings are a cpu_baseline leg like the one above (skipped with --no-cpu)
        def leg_single():
            sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
            import single_stream as SS
            return SS.single_stream(eng.ctx, torch, eng.dev, cpu=not args.no_cpu)
        res["single_stream"] = legs.run("single_stream", leg_single, collective=False)
    # ---- BASELINE configs 3, 4, 5 on the same ranks: 3 and 4 weak (every rank its own members / blocks), 5 ONE stream sharded
    if not args.no_others:
        others = []
        if args.dry_gloo:
            r5 = legs.run("config5_dry", lambda: dry_sharded_pipeline(eng, dist, rank, world))
            if r5 is not None:
                others.append(r5)
        else:
            sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
            import bench_configs as BC
            del wl
            torch.cuda.empty_cache()
            cpu_legs = world == 1 and not args.no_cpu
            kw = dict(rank=rank, world=world, dist=dist if world > 1 else None)
            for name, fn in (("config3", lambda: [BC.config3(eng.ctx, torch, eng.dev, args.others_scale, cpu=cpu_legs, **kw)]),
                             ("config3_gzip", lambda: [BC.config3(eng.ctx, torch, eng.dev, args.others_scale, cpu=cpu_legs, gzip_framing=True, **kw)]),
                             ("config4", lambda: BC.config4(eng.ctx, torch, eng.dev, args.others_scale, cpu=cpu_legs, **kw)),
                             ("config5", lambda: [BC.config5(eng.ctx, torch, eng.dev, args.others_scale, reps=3, cpu=cpu_legs, **kw)])):
                r = legs.run(name, fn)                          # a failing side config must not lose the headline line
                if isinstance(r, dict):                         # ({"error": ...})
                    others.append(dict(r, leg=name))
                elif r:
                    others += [x for x in r if x is not None]
                torch.cuda.empty_cache()
            wl = None
        if rank == 0:
            res["other_configs"] = others

    if rank == 0:
        if world == 1 and not args.no_others and not args.dry_gloo:
            res["hbm_ceiling_measured"] = legs.run("hbm_ceiling", lambda: hbm_ceiling(torch, eng.dev, eng.ctx), collective=False)
        if args.extras and world == 1 and not args.dry_gloo:
            def leg_extras():
                N = eng.N
                extras = {}
                for kind in ("text", "words", "runs", "rand", "mix"):
                    w2 = eng.make_workload(kind, args.nblocks, 0x77 + len(kind))
                    for v in N.LZ4_DECODE_VARIANTS:
                        eng.ctx.set_variant(N.LZ4_DECODE, v)
                        eng.decode(w2); eng.sync()
                        d2 = w2["dec"]
                        ok = torch.equal(d2.out_base[: args.nblocks * BLOCK], w2["raw"][: args.nblocks * BLOCK]) and int(d2.status.abs().max()) == 0
                        _, km, _ = time_steps(eng, w2, 5, 1, None)
                        extras["%s/v%d" % (kind, v)] = {"GiB/s": round(w2["out_bytes"] / (km * 1e-3) / 2**30, 2), "ratio": round(w2["out_bytes"] / w2["comp_bytes"], 2),
                                                        "hbm_frac": round((w2["comp_bytes"] + w2["out_bytes"]) / (km * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ok": bool(ok)}
                    del w2
                eng.ctx.set_variant(N.LZ4_DECODE, args.variant)
                return extras
            res["extras"] = legs.run("extras", leg_extras, collective=False)
    legs.finish()
    eng.close()


if __name__ == "__main__":
    main()
