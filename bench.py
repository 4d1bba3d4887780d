#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: GiB/s of decoded output over batched independent blocks.

Workload at N=1 (BASELINE.json configs[1]): LZ4 decode of 4096 independent 64 KiB blocks on one MI355X.
A "step" is one decode pass over the whole batch with inputs (compressed blocks + descriptors) and outputs
resident in HBM.  For N>1 every rank owns its own 4096 blocks (weak scaling, no data-path collective:
blocks are independent -- SURVEY.md 8e).  One JSON line is printed by rank 0.

Inputs are synthetic (rust_compress_amd.synth) and are compressed on the GPU by the product's own
bit-exact LZ4 encoder; the oracle is used only for the cpu_baseline leg (and its parity check).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BLOCK = 65536
NBLOCKS = 4096


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


def time_steps(ctx, torch, dec, steps, warmup, codec, dist=None):
    for _ in range(warmup):
        ctx.launch_dev(codec, dec)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        ctx.launch_dev(codec, dec)
        evs[i + 1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
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
    while total_s < budget_s and reps < 64:
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kind", default="text", help="synthetic distribution: text|runs|rand|mix")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (A/B)")
    ap.add_argument("--nblocks", type=int, default=NBLOCKS)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also time the other distributions / variants")
    args = ap.parse_args()

    import torch
    import rust_compress_amd as R
    from rust_compress_amd import _native as N

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("RCX_BENCH_FORCE_DIST"):      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    ctx = R.Context(torch.cuda.current_device())
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_variant(N.LZ4_DECODE, args.variant)

    dec, raw, comp_bytes, out_bytes = make_workload(R, ctx, torch, dev, args.kind, args.nblocks, 0x4C5A3401 + 7919 * rank)
    # parity (untimed): decoded bytes == the synthetic source on this rank
    ctx.launch_dev(N.LZ4_DECODE, dec)
    torch.cuda.synchronize()
    assert int(dec.status.abs().max()) == 0, "decode status != OK"
    assert bool((dec.out_len[: args.nblocks] == BLOCK).all())
    assert torch.equal(dec.out_base[: args.nblocks * BLOCK], raw[: args.nblocks * BLOCK]), "GPU decode != source"

    wall, kern_ms, kern_med = time_steps(ctx, torch, dec, args.steps, args.warmup, N.LZ4_DECODE, dist)
    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(out_bytes), float(comp_bytes)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    wall = float(t.item())
    total_out = float(tot[0].item())

    if rank == 0:
        alg_bytes = comp_bytes + out_bytes                     # per launch on this rank
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_lz4_decode.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
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
            "data": "synthetic",
            "config": {"workload": "LZ4 block decode, %d independent 64 KiB blocks per GPU (BASELINE configs[1])" % args.nblocks,
                       "distribution": "G-%s" % args.kind, "lz4_ratio": round(out_bytes / comp_bytes, 3),
                       "kernel_variant": args.variant, "parallelism": "blocks sharded, %d per rank, no collective" % args.nblocks},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_avg": round(kern_ms, 4),
                         "kernel_ms_median": round(kern_med, 4)},
        }
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(dec, raw, torch, args.nblocks)
        if args.extras and world == 1:
            extras = {}
            for kind in ("text", "words", "runs", "rand", "mix"):
                d2, r2, cb, ob = make_workload(R, ctx, torch, dev, kind, args.nblocks, 0x77 + len(kind))
                for v in N.LZ4_DECODE_VARIANTS:
                    ctx.set_variant(N.LZ4_DECODE, v)
                    ctx.launch_dev(N.LZ4_DECODE, d2)
                    torch.cuda.synchronize()
                    ok = torch.equal(d2.out_base[: args.nblocks * BLOCK], r2[: args.nblocks * BLOCK]) and int(d2.status.abs().max()) == 0
                    w, km, _ = time_steps(ctx, torch, d2, 5, 1, N.LZ4_DECODE)
                    extras["%s/v%d" % (kind, v)] = {"GiB/s": round(ob / (km * 1e-3) / 2**30, 2), "ratio": round(ob / cb, 2),
                                                    "hbm_frac": round((cb + ob) / (km * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ok": bool(ok)}
                del d2, r2
            res["extras"] = extras
        line = json.dumps(res)
    else:
        line = None
    # RCCL writes its version banner to stdout through C stdio, which is flushed at exit -- after anything Python prints.
    # Every rank pushes it out first; rank 0 prints the JSON line after a barrier, as the last line of the job's stdout.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if dist is not None:
        dist.barrier()
    if line is not None:
        print(line, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
