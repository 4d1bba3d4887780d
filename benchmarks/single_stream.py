#!/usr/bin/env python3
"""Batch of ONE, and how large a batch has to be: what a `compress::*` caller who hands over a single stream gets from the GPU path.

`single_stream(ctx, torch, dev, cpu)` -> dict for bench.py's line (also a script: prints the dict as JSON):
  * one 64 KiB LZ4 block, BASELINE config 1's one 1 MiB RFC-1951 stream, one 256 KiB BWT block (forward and inverse): microseconds and
    MB/s of a device-resident launch (`rcx_launch_dev` + a sync) and of the host-memory entry point (`rcx_*_batch`, pageable buffers:
    what the Reader / Writer mirrors call), with the oracle on ONE host thread beside each;
  * LZ4 blocks and zlib members in batches of 1 / 8 / 64 / 512 / 4096: time a call, rate, and the oracle on one thread and on every core;
  * the break-even batch sizes that follow (the smallest measured batch from which the host-memory call beats the CPU).
The oracle is the CPU baseline here as everywhere (bench.py's cpu_baseline rule): it is only run when `cpu` is set."""
import ctypes as C
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SWEEP = (1, 8, 64, 512, 4096)


def _med(fn, sync, reps=9, warm=2):
    for _ in range(warm):
        fn()
    sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); sync(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def _host_call(N, ctx, fn_name, base, off, lens, out_total, ooff, ocap, extra=None):
    """one rcx_*_batch call over pageable host buffers (numpy): -> (seconds a call, output array, statuses)"""
    n = len(off)
    out = np.zeros(int(out_total) + 64, dtype=np.uint8)
    out_len, in_used, status = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.int32)
    p = lambda a: a.ctypes.data
    hb = N.Batch(p(base), p(off), p(lens), p(out), p(ooff), p(ocap), p(out_len), p(in_used), p(status), n, N.MEM_HOST)
    fn = getattr(N.lib(), fn_name)
    args = [ctx._h, C.byref(hb)] + ([] if extra is None else [C.c_void_p(p(extra))])
    def call():
        rc = fn(*args)
        assert rc == 0, N.lib().rcx_last_error(ctx._h)
    t = _med(call, lambda: None, reps=7, warm=2)
    assert not status.any(), status[status != 0][:4]
    return t, out, out_len


def _oracle_t(O, codec, base, off, lens, out_total, ooff, ocap, threads, aux=None, reps=3):
    out = np.zeros(int(out_total) + 64, dtype=np.uint8)
    ts = []
    for _ in range(reps):
        t, _, _, st = O.batch_run(codec, base, off, lens, out, ooff, ocap, aux=aux, threads=threads)
        assert not st.any()
        ts.append(t)
    return float(np.median(ts)), out


def single_stream(ctx, torch, dev, cpu=True):
    import rust_compress_amd as R
    from rust_compress_amd import _native as N, synth, batch as B
    sync = torch.cuda.synchronize
    O = None
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as O_
        O_.build(); O = O_
    cores = os.cpu_count() or 1
    u64 = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
    res = {"what": "one stream alone on the GPU, and batches of 1..4096: microseconds a call (median), device-resident launch + sync ('dev') and host-memory entry point with pageable buffers ('host'), the oracle on one host thread beside it"}

    def leg(name, codec, fn_name, blobs, caps, raws, extra=None, aux_np=None):
        base, off, lens = B.pack(blobs)
        n = len(blobs)
        total, ooff, ocap = B.layout(caps)
        out_bytes = int(sum(len(r) for r in raws))
        db = R.DeviceBatch.from_host(base, off, lens, total, ooff, ocap, dev)
        if aux_np is not None:
            db.aux[:n] = torch.from_numpy(aux_np.astype(np.int32)).to(dev)
        sc = torch.empty(ctx.scratch_bytes(codec, n, max(max(caps), max(len(b) for b in blobs))) + 256, dtype=torch.uint8, device=dev)
        t_dev = _med(lambda: ctx.launch_dev(codec, db, sc), sync)
        assert int(db.status[:n].abs().max()) == 0
        got = db.out_base.cpu().numpy()
        for o, c, r in zip(ooff, caps, raws):
            assert bytes(got[int(o): int(o) + len(r)]) == r, name
        t_host, _, _ = _host_call(N, ctx, fn_name, base, off, lens, total, ooff, ocap, extra=extra if extra is not None else aux_np)
        r_ = {"n": n, "out_bytes": out_bytes, "dev_us": round(t_dev * 1e6, 1), "dev_MB/s": round(out_bytes / t_dev / 1e6, 1),
              "host_us": round(t_host * 1e6, 1), "host_MB/s": round(out_bytes / t_host / 1e6, 1)}
        if O is not None:
            aux_o = None if aux_np is None else aux_np.copy()
            if codec in (N.BWT_FORWARD,):
                aux_o = np.zeros(n, dtype=np.uint32)
            t1, _ = _oracle_t(O, codec, base, u64(off), u64(lens), total, u64(ooff), u64(ocap), 1, aux=aux_o)
            r_["cpu_1thread_us"] = round(t1 * 1e6, 1); r_["cpu_1thread_MB/s"] = round(out_bytes / t1 / 1e6, 1)
            if n >= 8:
                th = min(cores, n)
                tn, _ = _oracle_t(O, codec, base, u64(off), u64(lens), total, u64(ooff), u64(ocap), th, aux=aux_o)
                r_["cpu_all_cores_us"] = round(tn * 1e6, 1); r_["cpu_threads"] = th
        del db, sc
        return r_

    # ---- one stream each
    raw = synth.gen("text", 65536, 0x51).tobytes()
    blob = ctx.lz4_encode_blocks([raw]).check().outputs[0]
    res["lz4_block_64KiB"] = leg("lz4", N.LZ4_DECODE, "rcx_lz4_decode_batch", [blob], [65536], [raw])
    raw1m = synth.gen("text", 1 << 20, 0x52).tobytes()
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    stream = co.compress(raw1m) + co.flush()
    flags = np.zeros(1, np.uint32)
    res["deflate_stream_1MiB"] = dict(leg("inflate", N.INFLATE, "rcx_inflate_batch", [stream], [1 << 20], [raw1m], extra=flags),
                                      what="BASELINE configs[0]: one RFC-1951 stream of 1 MiB (zlib level 6, raw)")
    rawb = synth.gen("text", 262144, 0x53).tobytes()
    # (the forward transform's output is L, not the input: its own small leg)
    base, off, lens = B.pack([rawb]); total, ooff, ocap = B.layout([262144])
    db = R.DeviceBatch.from_host(base, off, lens, total, ooff, ocap, dev)
    sc = torch.empty(ctx.scratch_bytes(N.BWT_FORWARD, 1, 262144) + 256, dtype=torch.uint8, device=dev)
    t_f = _med(lambda: ctx.launch_dev(N.BWT_FORWARD, db, sc), sync, reps=5, warm=1)
    L = db.out_base[:262144].cpu().numpy().tobytes(); org = db.aux[:1].cpu().numpy().view(np.uint32).copy()
    t_fh, _, _ = _host_call(N, ctx, "rcx_bwt_forward_batch", base, off, lens, total, ooff, ocap, extra=np.zeros(1, np.uint32))
    bw = {"out_bytes": 262144, "forward_dev_us": round(t_f * 1e6, 1), "forward_dev_MB/s": round(262144 / t_f / 1e6, 1), "forward_host_us": round(t_fh * 1e6, 1)}
    del db, sc
    inv = leg("bwt inverse", N.BWT_INVERSE, "rcx_bwt_inverse_batch", [L], [262144], [rawb], aux_np=org)
    bw.update({"inverse_dev_us": inv["dev_us"], "inverse_dev_MB/s": inv["dev_MB/s"], "inverse_host_us": inv["host_us"]})
    if O is not None:
        t1f, _ = _oracle_t(O, N.BWT_FORWARD, base, u64(off), u64(lens), total, u64(ooff), u64(ocap), 1, aux=np.zeros(1, np.uint32), reps=1)
        bw.update({"forward_cpu_1thread_us": round(t1f * 1e6, 1), "inverse_cpu_1thread_us": inv.get("cpu_1thread_us")})
    res["bwt_block_256KiB"] = bw

    # ---- how large a batch has to be
    nmax = max(SWEEP)
    raws = [synth.gen("text", 65536, 0x600 + i).tobytes() for i in range(64)]
    blobs = ctx.lz4_encode_blocks(raws).check().outputs
    sweep = {"lz4_64KiB_blocks": [], "zlib_16KiB_members": []}
    for n in SWEEP:
        sweep["lz4_64KiB_blocks"].append(leg("lz4 x%d" % n, N.LZ4_DECODE, "rcx_lz4_decode_batch", [blobs[i % 64] for i in range(n)], [65536] * n, [raws[i % 64] for i in range(n)]))
    zraws = [synth.gen("text", 16384, 0x700 + i).tobytes() for i in range(64)]
    zm = [zlib.compress(r, (1, 6, 9)[i % 3]) for i, r in enumerate(zraws)]
    zflags = np.zeros(nmax, np.uint32)
    for n in SWEEP:
        sweep["zlib_16KiB_members"].append(leg("zlib x%d" % n, N.ZLIB_DECODE, "rcx_zlib_decode_batch", [zm[i % 64] for i in range(n)], [16384] * n, [zraws[i % 64] for i in range(n)], extra=zflags))
    res["sweep"] = sweep
    if O is not None:
        be = {}
        for k, rows in sweep.items():
            one = next((r["n"] for r in rows if r["host_us"] < r["cpu_1thread_us"]), None)
            allc = next((r["n"] for r in rows if "cpu_all_cores_us" in r and r["host_us"] < r["cpu_all_cores_us"]), None)
            be[k] = {"beats_one_host_thread_from": one, "beats_every_host_core_from": allc, "host_cores": cores,
                     "note": "smallest measured batch (1 / 8 / 64 / 512 / 4096) whose host-memory call is faster than the oracle on one thread / on min(batch, cores) threads; null: not within the sweep"}
        res["break_even"] = be
    return res


if __name__ == "__main__":
    import torch
    import rust_compress_amd as R
    dev = torch.device("cuda", 0)
    ctx = R.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    print(json.dumps(single_stream(ctx, torch, dev, cpu="--no-cpu" not in sys.argv)))
