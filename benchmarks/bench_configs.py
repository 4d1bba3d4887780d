#!/usr/bin/env python3
"""Timing of the other BASELINE.json configs on one MI355X (3: zlib members, 4: BWT fwd+inv, 5: BWT->DC->Ari).
Prints one JSON line per config: decoded/processed GiB/s, algorithmic bytes, HBM-roofline fraction.
(bench.py at the repo root stays the headline config-1/2 benchmark the driver runs.)"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="3,4,5")
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the full config size")
    args = ap.parse_args()
    import torch
    import rust_compress_amd as R
    from rust_compress_amd import _native as N, synth, batch as B, pipeline as P
    dev = torch.device("cuda", 0)
    ctx = R.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
    for cfg in args.configs.split(","):
        if cfg == "3":
            nb, BLOCK = int(65536 * args.scale), 16384
            raw_np = synth.gen_blocks("text", nb, BLOCK, 0x5A11)
            with Pool(32) as pool:
                members = pool.map(_zmember, [(i, raw_np[i * BLOCK:(i + 1) * BLOCK].tobytes()) for i in range(nb)], chunksize=512)
            base, off, lens = B.pack(members)
            ar = np.arange(nb, dtype=np.int64)
            db = R.DeviceBatch.from_host(base, off, lens, nb * BLOCK, (ar * BLOCK).astype(np.uint64), np.full(nb, BLOCK, dtype=np.uint64), dev)
            sc = torch.empty(ctx.scratch_bytes(N.ZLIB_DECODE, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
            t = timeit(lambda: ctx.launch_dev(N.ZLIB_DECODE, db, sc), torch)
            assert int(db.status[:nb].abs().max()) == 0 and torch.equal(db.out_base[: nb * BLOCK].cpu(), torch.from_numpy(raw_np))
            alg = int(lens.sum()) + nb * BLOCK
            print(json.dumps({"config": 3, "workload": "zlib decode, %d members x 16 KiB (G-text, levels 1/6/9)" % nb, "GiB/s": round(nb * BLOCK / t / 2**30, 2),
                              "ms": round(t * 1e3, 3), "ratio": round(nb * BLOCK / lens.sum(), 2), "roofline_frac": round(alg / t / 1e9 / PEAK, 5)}), flush=True)
        elif cfg == "3g":                                   # the same members in gzip framing (extension)
            nb, BLOCK = int(65536 * args.scale), 16384
            raw_np = synth.gen_blocks("text", nb, BLOCK, 0x5A11)
            with Pool(32) as pool:
                members = pool.map(_gzmember, [(i, raw_np[i * BLOCK:(i + 1) * BLOCK].tobytes()) for i in range(nb)], chunksize=512)
            base, off, lens = B.pack(members)
            ar = np.arange(nb, dtype=np.int64)
            db = R.DeviceBatch.from_host(base, off, lens, nb * BLOCK, (ar * BLOCK).astype(np.uint64), np.full(nb, BLOCK, dtype=np.uint64), dev)
            sc = torch.empty(ctx.scratch_bytes(N.GZIP_DECODE, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
            t = timeit(lambda: ctx.launch_dev(N.GZIP_DECODE, db, sc), torch)
            assert int(db.status[:nb].abs().max()) == 0 and torch.equal(db.out_base[: nb * BLOCK].cpu(), torch.from_numpy(raw_np))
            assert bool((db.in_used[:nb].cpu() == torch.from_numpy(lens.astype(np.int64))).all())
            alg = int(lens.sum()) + nb * BLOCK
            print(json.dumps({"config": "3g", "workload": "gzip decode (header + DEFLATE + CRC-32/ISIZE check), %d members x 16 KiB (G-text, levels 1/6/9)" % nb,
                              "GiB/s": round(nb * BLOCK / t / 2**30, 2), "ms": round(t * 1e3, 3), "ratio": round(nb * BLOCK / lens.sum(), 2),
                              "roofline_frac": round(alg / t / 1e9 / PEAK, 5)}), flush=True)
        elif cfg == "4":
            nb, BLOCK = int(1024 * args.scale), 262144
            for kind in ("text", "dna4"):
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
                print(json.dumps({"config": 4, "workload": "BWT %d x 256 KiB G-%s" % (nb, kind), "forward_GiB/s": round(tot / tf / 2**30, 3), "forward_ms": round(tf * 1e3, 2),
                                  "inverse_GiB/s": round(tot / ti / 2**30, 3), "inverse_ms": round(ti * 1e3, 2),
                                  "forward_roofline_frac": round((2 * tot + 4 * nb) / tf / 1e9 / PEAK, 6), "inverse_roofline_frac": round((2 * tot + 4 * nb) / ti / 1e9 / PEAK, 6)}), flush=True)
        elif cfg == "5":
            BLOCK = 262144
            total = int(1e9 * args.scale)
            lens = [BLOCK] * (total // BLOCK) + ([total % BLOCK] if total % BLOCK else [])
            data = np.concatenate([synth.gen("text", min(BLOCK * 256, total - s), 0xC0 + s) for s in range(0, total, BLOCK * 256)])[:total]
            raw = torch.from_numpy(data).to(dev)
            pipe = P.BwtDcAri(ctx, dev)
            te = td = 1e9
            for rep in range(4):                  # the first pass pays the one-off scratch / output allocations; best of the rest
                t0 = time.perf_counter(); comp, coff, clen, praw, _ = pipe.encode(raw, lens); torch.cuda.synchronize(); e_ = time.perf_counter() - t0
                t0 = time.perf_counter(); back = pipe.decode(comp, coff, clen, praw, lens); torch.cuda.synchronize(); d_ = time.perf_counter() - t0
                if rep:
                    te, td = min(te, e_), min(td, d_)
            assert torch.equal(back, raw)
            print(json.dumps({"config": 5, "workload": "BWT->DC->Ari, %d bytes in %d blocks of 256 KiB" % (total, len(lens)), "compressed_ratio": round(total / clen.sum(), 3),
                              "encode_GiB/s": round(total / te / 2**30, 3), "decode_GiB/s": round(total / td / 2**30, 3), "encode_s": round(te, 3), "decode_s": round(td, 3)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
