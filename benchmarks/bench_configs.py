#!/usr/bin/env python3
"""Timing of the other BASELINE.json configs on one MI355X (3: zlib members, 4: BWT fwd+inv, 5: BWT->DC->Ari).
Each `config*` function returns a dict (decoded/processed GiB/s, algorithmic bytes, HBM-roofline fraction); bench.py
at the repo root calls them for its `other_configs` list, and run as a script this file prints one JSON line per config."""
import argparse
import json
import os
import sys
import time
import zlib
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 8000.0


def _zmember(args):
    i, data = args
    return zlib.compress(data, (1, 6, 9)[i % 3])


def _gzmember(args):
    import gzip
    i, data = args
    return gzip.compress(data, compresslevel=(1, 6, 9)[i % 3], mtime=0)


def timeit(fn, torch, reps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def _libz_rate(members, nbytes, budget_s=6.0):
    """'Strong CPU' line for DEFLATE: libz inflate through Python's zlib (releases the GIL) on every host core,
    and on one thread over a bounded sample."""
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    sample = members[: max(1024, min(len(members), 16384))]
    frac = len(sample) / len(members)
    def work(chunk):
        n = 0
        for m in chunk:
            n += len(zlib.decompress(m))
        return n
    chunks = [sample[i::cores] for i in range(cores)]
    t0 = time.perf_counter(); reps = 0; got = 0
    with ThreadPoolExecutor(cores) as ex:
        while time.perf_counter() - t0 < budget_s / 2 and reps < 8:
            got += sum(ex.map(work, chunks)); reps += 1
    tm = time.perf_counter() - t0
    one = sample[:1024]
    t0 = time.perf_counter(); g1 = work(one); t1 = time.perf_counter() - t0
    return {"value": round(got / tm / 2**30, 3), "unit": "GiB/s", "cores": cores, "kind": "libz (zlib.decompress, %d threads)" % cores,
            "sample": "%d of %d members x %d passes; 1 thread on 1024 members: %.3f GiB/s" % (len(sample), len(members), reps, g1 / t1 / 2**30)}


def config3(ctx, torch, dev, scale=1.0, gzip_framing=False, cpu=True):
    import rust_compress_amd as R
    from rust_compress_amd import _native as N, synth, batch as B
    nb, BLOCK = int(65536 * scale), 16384
    raw_np = synth.gen_blocks("text", nb, BLOCK, 0x5A11)
    with Pool(min(32, os.cpu_count() or 1)) as pool:
        members = pool.map(_gzmember if gzip_framing else _zmember, [(i, raw_np[i * BLOCK:(i + 1) * BLOCK].tobytes()) for i in range(nb)], chunksize=512)
    base, off, lens = B.pack(members)
    ar = np.arange(nb, dtype=np.int64)
    db = R.DeviceBatch.from_host(base, off, lens, nb * BLOCK, (ar * BLOCK).astype(np.uint64), np.full(nb, BLOCK, dtype=np.uint64), dev)
    codec = N.GZIP_DECODE if gzip_framing else N.ZLIB_DECODE
    sc = torch.empty(ctx.scratch_bytes(codec, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
    t = timeit(lambda: ctx.launch_dev(codec, db, sc), torch)
    assert int(db.status[:nb].abs().max()) == 0 and torch.equal(db.out_base[: nb * BLOCK].cpu(), torch.from_numpy(raw_np))
    if gzip_framing:
        assert bool((db.in_used[:nb].cpu() == torch.from_numpy(lens.astype(np.int64))).all())
    alg = int(lens.sum()) + nb * BLOCK
    res = {"config": "3g" if gzip_framing else 3,
           "workload": "%s decode, %d members x 16 KiB (G-text, levels 1/6/9)" % ("gzip (header + DEFLATE + CRC-32/ISIZE check)" if gzip_framing else "zlib", nb),
           "GiB/s": round(nb * BLOCK / t / 2**30, 2), "ms": round(t * 1e3, 3), "ratio": round(nb * BLOCK / lens.sum(), 2),
           "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 2), "peak": PEAK, "unit": "GB/s", "frac": round(alg / t / 1e9 / PEAK, 5),
                        "algorithmic_bytes_per_launch": alg}}
    if cpu and not gzip_framing:
        res["cpu_baseline_libz"] = _libz_rate(members, nb * BLOCK)
    return res


def config4(ctx, torch, dev, scale=1.0, kinds=("text", "dna4")):
    import rust_compress_amd as R
    from rust_compress_amd import _native as N, synth
    i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
    nb, BLOCK = int(1024 * scale), 262144
    out = []
    for kind in kinds:
        raw = torch.from_numpy(synth.gen_blocks(kind, nb, BLOCK, 0xB77)).to(dev)
        ar = np.arange(nb, dtype=np.int64)
        fw = R.DeviceBatch(raw, i64(ar * BLOCK), i64(np.full(nb, BLOCK)), torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK), i64(np.full(nb, BLOCK)))
        sc = torch.empty(ctx.scratch_bytes(N.BWT_FORWARD, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
        tf = timeit(lambda: ctx.launch_dev(N.BWT_FORWARD, fw, sc), torch, reps=3)
        del sc
        inv = R.DeviceBatch(fw.out_base, fw.out_off, fw.out_len, torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK), i64(np.full(nb, BLOCK)), aux=fw.aux)
        sc = torch.empty(ctx.scratch_bytes(N.BWT_INVERSE, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
        ti = timeit(lambda: ctx.launch_dev(N.BWT_INVERSE, inv, sc), torch, reps=3)
        del sc
        assert torch.equal(inv.out_base[: nb * BLOCK], raw)
        tot = nb * BLOCK
        alg = 2 * tot + 4 * nb
        out.append({"config": 4, "workload": "BWT %d x 256 KiB G-%s" % (nb, kind), "forward_GiB/s": round(tot / tf / 2**30, 3), "forward_ms": round(tf * 1e3, 2),
                    "inverse_GiB/s": round(tot / ti / 2**30, 3), "inverse_ms": round(ti * 1e3, 2),
                    "forward_roofline": {"bound": "hbm", "achieved": round(alg / tf / 1e9, 2), "peak": PEAK, "unit": "GB/s", "frac": round(alg / tf / 1e9 / PEAK, 6), "algorithmic_bytes_per_launch": alg},
                    "inverse_roofline": {"bound": "hbm", "achieved": round(alg / ti / 1e9, 2), "peak": PEAK, "unit": "GB/s", "frac": round(alg / ti / 1e9 / PEAK, 6), "algorithmic_bytes_per_launch": alg}})
        del raw, fw, inv
    return out


def config5(ctx, torch, dev, scale=1.0, reps=4):
    from rust_compress_amd import synth, pipeline as P
    BLOCK = 262144
    total = int(1e9 * scale)
    lens = [BLOCK] * (total // BLOCK) + ([total % BLOCK] if total % BLOCK else [])
    data = np.concatenate([synth.gen("text", min(BLOCK * 256, total - s), 0xC0 + s) for s in range(0, total, BLOCK * 256)])[:total]
    raw = torch.from_numpy(data).to(dev)
    pipe = P.BwtDcAri(ctx, dev)
    te = td = 1e9
    for rep in range(reps):                  # the first pass pays the one-off scratch / output allocations; best of the rest
        t0 = time.perf_counter(); comp, coff, clen, praw, _ = pipe.encode(raw, lens); torch.cuda.synchronize(); e_ = time.perf_counter() - t0
        t0 = time.perf_counter(); back = pipe.decode(comp, coff, clen, praw, lens); torch.cuda.synchronize(); d_ = time.perf_counter() - t0
        if rep:
            te, td = min(te, e_), min(td, d_)
    assert torch.equal(back, raw)
    csum = int(clen.sum())
    return {"config": 5, "workload": "BWT->DC->Ari, %d bytes in %d blocks of 256 KiB" % (total, len(lens)), "compressed_ratio": round(total / csum, 3),
            "encode_GiB/s": round(total / te / 2**30, 3), "decode_GiB/s": round(total / td / 2**30, 3), "encode_s": round(te, 3), "decode_s": round(td, 3),
            "decode_roofline": {"bound": "hbm", "achieved": round((total + csum) / td / 1e9, 2), "peak": PEAK, "unit": "GB/s", "frac": round((total + csum) / td / 1e9 / PEAK, 6),
                                "algorithmic_bytes_per_launch": total + csum}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="3,4,5")
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the full config size")
    args = ap.parse_args()
    import torch
    import rust_compress_amd as R
    dev = torch.device("cuda", 0)
    ctx = R.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for cfg in args.configs.split(","):
        if cfg == "3":
            print(json.dumps(config3(ctx, torch, dev, args.scale)), flush=True)
        elif cfg == "3g":                                   # the same members in gzip framing (extension)
            print(json.dumps(config3(ctx, torch, dev, args.scale, gzip_framing=True)), flush=True)
        elif cfg == "4":
            for r in config4(ctx, torch, dev, args.scale):
                print(json.dumps(r), flush=True)
        elif cfg == "5":
            print(json.dumps(config5(ctx, torch, dev, args.scale)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
