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
    """zlib members at levels 1 / 6 / 9 and, every fourth one, fixed-Huffman blocks (Z_FIXED: flate.rs:397-450's `statik` path)"""
    i, data = args
    if i % 4 == 3:
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        return c.compress(data) + c.flush()
    return zlib.compress(data, (1, 6, 9)[i % 4])


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


def timeit_ranks(fn, torch, dist, reps=5, warm=1):
    """timeit on every rank between barriers; the MAX over the ranks of the per-rank median (seconds)"""
    if dist is None:
        return timeit(fn, torch, reps, warm)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); dist.barrier()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = torch.tensor([float(np.median(ts))], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


CONFIG_SOURCES = {      # the device sources a config's kernels are compiled from (the translation units of rust_compress_amd/csrc/tu_*.hip)
    "3": ["k_inflate.hip", "k_inflate2.hip", "k_inflate3.hip", "k_lz4_decode_v4.hip", "k_lz4_decode_v5.hip", "k_crc32.hip", "k_gzip.hip", "tu_inflate.hip", "rcx_dev.h"],
    "4": ["k_bwt.hip", "k_bwt_sort.hip", "k_bwt_inverse.hip", "tu_bwt.hip", "rcx_dev.h"],
    "5": ["k_serial.hip", "tu_serial.hip", "k_bwt.hip", "k_bwt_sort.hip", "k_bwt_inverse.hip", "tu_bwt.hip", "rcx_dev.h"],
}


def source_hash(cfg=None):
    """sha256 over the device sources of a config's kernels (all of them when cfg is None; the host-side files -- rcx_api.hip, the C-ABI
    and its staging, and rcx_tu.h -- hold no device code): a PMC traffic figure is only quoted for the code it was measured on"""
    import hashlib
    d = os.path.join(ROOT, "rust_compress_amd", "csrc")
    h = hashlib.sha256()
    files = CONFIG_SOURCES.get(str(cfg)[:1]) if cfg is not None else None
    if files is None:
        files = [f for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h")) and f not in ("rcx_api.hip", "rcx_tu.h")]
    for f in sorted(files):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(cfg, key=None):
    """HBM bytes per launch from profiles/pmc_cfg<cfg>.json (separate FETCH_SIZE / WRITE_SIZE passes, the guide's gfx950
    correction), or None when the kernels have changed since it was measured."""
    f = os.path.join(ROOT, "profiles", "pmc_cfg%s.json" % cfg)
    try:
        j = json.load(open(f))
        if j.get("kernel_source_hash") != source_hash(cfg):
            return None
        j = j[key] if key else j
        return {"hbm_bytes_per_launch": j["hbm_bytes_per_launch"], "file": "profiles/pmc_cfg%s.json" % cfg}
    except Exception:
        return None


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build()
    return O


def oracle_rate(codec, in_base, in_off, in_len, out_total, out_off, out_cap, bytes_per_block, what, aux=None, n_out=None, budget_s=4.0, check=None):
    """The oracle (the line-cited CPU restatement of the reference) on every host core over a BOUNDED sample: as many passes
    over the sample as fit the budget, plus one block-sized probe on one thread.  -> cpu_baseline dict (kind "port")."""
    O = _oracle()
    cores = os.cpu_count() or 1
    n = len(in_off)
    out = np.zeros(int(out_total) + 64, dtype=np.uint8)
    secs, reps, st = 0.0, 0, None
    while secs < budget_s and reps < 32:
        t, out_len, _, st = O.batch_run(codec, in_base, in_off, in_len, out, out_off, out_cap, aux=aux, n_out=n_out, threads=cores)
        out_len = out_len.astype(np.int64)
        assert not st.any(), "oracle status %s" % st[st != 0][:4]
        secs += t; reps += 1
    ok = bool(check(out)) if check is not None else None
    k1 = max(1, min(n, 4))
    t1, _, _, _ = O.batch_run(codec, in_base, in_off[:k1], in_len[:k1], out, out_off[:k1], out_cap[:k1], aux=aux, n_out=None if n_out is None else n_out[:k1], threads=1)
    return {"value": round(n * bytes_per_block * reps / secs / 2**30, 3), "unit": "GiB/s", "cores": cores, "kind": "port",
            "sample": "%s: %d blocks x %d passes (%.1f s) on %d threads; 1 thread on %d blocks: %.3f GiB/s%s" % (
                what, n, reps, secs, cores, k1, k1 * bytes_per_block / t1 / 2**30, "" if ok is None else "; oracle output == the GPU's: %s" % ok),
            "_out_len": out_len}, out


def _libz_rate(members, nbytes, budget_s=6.0, wbits=15):
    """'Strong CPU' line for DEFLATE: libz inflate through Python's zlib (releases the GIL) on every host core,
    and on one thread over a bounded sample.  wbits 31: gzip members (header, CRC-32 and ISIZE checked by libz)."""
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    sample = members[: max(1024, min(len(members), 16384))]
    frac = len(sample) / len(members)
    def work(chunk):
        n = 0
        for m in chunk:
            n += len(zlib.decompress(m, wbits))
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


def _roof(alg, t, traffic=None, digits=5):
    r = {"bound": "hbm", "achieved": round(alg / t / 1e9, 2), "peak": PEAK, "unit": "GB/s", "frac": round(alg / t / 1e9 / PEAK, digits),
         "algorithmic_bytes_per_launch": alg, "traffic": None, "traffic_source": None}
    if traffic:
        r["traffic"], r["traffic_source"] = traffic["hbm_bytes_per_launch"], traffic["file"]
    return r


def config3(ctx, torch, dev, scale=1.0, gzip_framing=False, cpu=True, rank=0, world=1, dist=None, once=False):
    """zlib members, weak scaling: every rank decodes its own 65 536 x scale members (no collective on the data path)."""
    import rust_compress_amd as R
    from rust_compress_amd import _native as N, synth, batch as B
    nb, BLOCK = int(65536 * scale), 16384
    raw_np = synth.gen_blocks("text", nb, BLOCK, 0x5A11 + 7919 * rank)
    with Pool(min(32, max(1, (os.cpu_count() or 1) // world))) as pool:
        members = pool.map(_gzmember if gzip_framing else _zmember, [(i, raw_np[i * BLOCK:(i + 1) * BLOCK].tobytes()) for i in range(nb)], chunksize=512)
    base, off, lens = B.pack(members)
    ar = np.arange(nb, dtype=np.int64)
    db = R.DeviceBatch.from_host(base, off, lens, nb * BLOCK, (ar * BLOCK).astype(np.uint64), np.full(nb, BLOCK, dtype=np.uint64), dev)
    codec = N.GZIP_DECODE if gzip_framing else N.ZLIB_DECODE
    if os.environ.get("RCX_INFLATE_VARIANT"):                  # A/B of the DEFLATE kernels (see launch_inflate)
        ctx.set_variant(codec, int(os.environ["RCX_INFLATE_VARIANT"]))
    sc = torch.empty(ctx.scratch_bytes(codec, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
    t = timeit_ranks(lambda: ctx.launch_dev(codec, db, sc), torch, dist if world > 1 else None, reps=1 if once else 5, warm=0 if once else 1)
    nocheck = bool(os.environ.get("RCX_CFG_NOCHECK"))           # attribution builds (part of a kernel cut out): time only
    if os.environ.get("RCX_INF3_PROF"):                         # a -DINF3_PROF build: phase cycle totals in the scratch's last 128 bytes
        tail = (sc.numel() - 128) & ~7
        sc[tail: tail + 64] = 0
        ctx.launch_dev(codec, db, sc); torch.cuda.synchronize()
        pf = sc[tail: tail + 64].view(torch.int64).cpu().numpy().astype(np.float64) / nb
        names = ["emit5", "wide copies", "staging", "headers + tables", "symbol passes", "  of them tile builds", "chunks", "symbols booked"]
        print("k_inflate3 per member (cycles / counts): " + "  ".join("%s %.0f" % (n_, v_) for n_, v_ in zip(names, pf)), file=sys.stderr)
    if nocheck:
        st_, cn_ = torch.unique(db.status[:nb], return_counts=True)
        print("statuses:", dict(zip(st_.tolist(), cn_.tolist())), "out_len sum", int(db.out_len[:nb].sum()), file=sys.stderr)
    assert nocheck or (int(db.status[:nb].abs().max()) == 0 and torch.equal(db.out_base[: nb * BLOCK].cpu(), torch.from_numpy(raw_np)))
    if gzip_framing:
        assert bool((db.in_used[:nb].cpu() == torch.from_numpy(lens.astype(np.int64))).all())
    alg = int(lens.sum()) + nb * BLOCK                       # per rank and launch
    cliff = None
    if not gzip_framing and not once and rank == 0 and world == 1 and not os.environ.get("RCX_INFLATE_VARIANT") and not nocheck:
        # what a batch costs when EVERY member is handed to the exact lane-per-stream kernel (k_inflate2, the second pass of
        # the default path: statuses, odd codes, overruns): the same members with that kernel alone (variant 9)
        ctx.set_variant(codec, 9)
        t9 = timeit(lambda: ctx.launch_dev(codec, db, sc), torch, reps=3)
        ctx.set_variant(codec, 0)
        assert int(db.status[:nb].abs().max()) == 0 and torch.equal(db.out_base[: nb * BLOCK].cpu(), torch.from_numpy(raw_np))
        cliff = {"ms": round(t9 * 1e3, 3), "GiB/s": round(nb * BLOCK / t9 / 2**30, 2), "times_the_default_path": round(t9 / t, 2),
                 "what": "every member decoded by the fallback kernel k_inflate2 (lane per stream) instead of k_inflate3 (wave per stream)"}
    res = {"config": "3g" if gzip_framing else 3, "n_gpus": world, "scaling": "weak",
           "workload": "%s decode, %d members x 16 KiB per GPU (G-text; %s)" % (
               "gzip (header + DEFLATE + CRC-32/ISIZE check)" if gzip_framing else "zlib", nb,
               "levels 1/6/9" if gzip_framing else "levels 1/6/9 and Z_FIXED, a quarter each"),
           "GiB/s": round(world * nb * BLOCK / t / 2**30, 2), "ms": round(t * 1e3, 3), "ratio": round(nb * BLOCK / lens.sum(), 2),
           "roofline": _roof(alg, t, None if gzip_framing else pmc_traffic(3))}
    if cliff is not None:
        res["fallback_cliff"] = cliff
    if not once and rank == 0 and world == 1 and not nocheck and not os.environ.get("RCX_INFLATE_VARIANT"):
        # PCIe-inclusive: the same members through rcx_zlib_decode_batch / rcx_gzip_decode_batch with page-locked host buffers -- the decoder stores what leaves
        # its window straight into the caller's buffer (rcx_api.hip) -- beside one copy each way around the launch
        import ctypes as C
        inb = torch.from_numpy(base).pin_memory()
        outb = torch.zeros(nb * BLOCK + 64, dtype=torch.uint8).pin_memory()
        ooff, ocap = (ar * BLOCK).astype(np.uint64), np.full(nb, BLOCK, dtype=np.uint64)
        out_len, in_used, status, flags = np.zeros(nb, np.uint64), np.zeros(nb, np.uint64), np.zeros(nb, np.int32), np.zeros(nb, np.uint32)
        p = lambda a: a.ctypes.data
        hb = N.Batch(inb.data_ptr(), p(off), p(lens), outb.data_ptr(), p(ooff), p(ocap), p(out_len), p(in_used), p(status), nb, N.MEM_HOST)
        hp = {}
        for plain in (True, False):
            N.lib().rcx_ctx_set_param(ctx._h, codec, 1 if plain else 0)
            ts = []
            try:
                for it in range(4):
                    if it == 3:
                        outb.zero_(); status[:] = -9
                    t0 = time.perf_counter()
                    rc = (N.lib().rcx_gzip_decode_batch if gzip_framing else N.lib().rcx_zlib_decode_batch)(ctx._h, C.byref(hb), C.c_void_p(p(flags)))
                    ts.append(time.perf_counter() - t0)
                    assert rc == 0 and not status.any()
            finally:
                N.lib().rcx_ctx_set_param(ctx._h, codec, 0)
            assert np.array_equal(outb.numpy()[: nb * BLOCK], raw_np)
            hp[plain] = float(np.median(ts[1:]))
        res["host_path"] = {"GiB/s": round(nb * BLOCK / hp[False] / 2**30, 2), "ms": round(hp[False] * 1e3, 2), "ms_plain_copies": round(hp[True] * 1e3, 2),
                            "bytes_in": int(lens.sum()), "bytes_out": nb * BLOCK, "verified": True,
                            "what": "%s, RCX_MEM_HOST, page-locked host buffers: decoded bytes stored straight into the caller's buffer by the launch" % ("rcx_gzip_decode_batch" if gzip_framing else "rcx_zlib_decode_batch")}
    if cpu and gzip_framing and rank == 0 and world == 1:
        # the reference crate has no gzip reader (an extension), so there is no oracle leg for this framing:
        # the CPU line is libz itself -- inflate + CRC-32 + ISIZE per member, every host core
        res["cpu_baseline"] = _libz_rate(members, nb * BLOCK, wbits=31)
        res["cpu_baseline"]["kind"] = "libz gzip members (zlib.decompress(wbits=31), %d threads)" % (os.cpu_count() or 1)
    if cpu and not gzip_framing and rank == 0 and world == 1:
        res["cpu_baseline_libz"] = _libz_rate(members, nb * BLOCK)
        ns = min(nb, 4096)                                  # the oracle walks its Huffman trees bit by bit: a bounded sample
        res["cpu_baseline"], _ = oracle_rate(N.ZLIB_DECODE, base, off[:ns], lens[:ns], ns * BLOCK, ar[:ns] * BLOCK, np.full(ns, BLOCK), BLOCK,
                                             "oracle zlib decode, the first %d of the %d members" % (ns, nb),
                                             check=lambda o: np.array_equal(o[: ns * BLOCK], raw_np[: ns * BLOCK]))
        res["cpu_baseline"].pop("_out_len", None)
    return res


def config4(ctx, torch, dev, scale=1.0, kinds=("text", "dna4"), cpu=False, rank=0, world=1, dist=None, once=False):
    """BWT forward + inverse, weak scaling: every rank transforms its own 1024 x scale blocks of 256 KiB."""
    import rust_compress_amd as R
    from rust_compress_amd import _native as N, synth
    i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
    nb, BLOCK = int(1024 * scale), 262144
    dd = dist if world > 1 else None
    reps, warm = (1, 0) if once else (3, 1)
    out = []
    for kind in kinds:
        raw = torch.from_numpy(synth.gen_blocks(kind, nb, BLOCK, 0xB77 + 7919 * rank)).to(dev)
        ar = np.arange(nb, dtype=np.int64)
        fw = R.DeviceBatch(raw, i64(ar * BLOCK), i64(np.full(nb, BLOCK)), torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK), i64(np.full(nb, BLOCK)))
        sc = torch.empty(ctx.scratch_bytes(N.BWT_FORWARD, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
        tf = timeit_ranks(lambda: ctx.launch_dev(N.BWT_FORWARD, fw, sc), torch, dd, reps=reps, warm=warm)
        del sc
        inv = R.DeviceBatch(fw.out_base, fw.out_off, fw.out_len, torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev), i64(ar * BLOCK), i64(np.full(nb, BLOCK)), aux=fw.aux)
        sc = torch.empty(ctx.scratch_bytes(N.BWT_INVERSE, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
        ti = timeit_ranks(lambda: ctx.launch_dev(N.BWT_INVERSE, inv, sc), torch, dd, reps=reps, warm=warm)
        del sc
        assert torch.equal(inv.out_base[: nb * BLOCK], raw)
        tot = nb * BLOCK
        alg = 2 * tot + 4 * nb
        r = {"config": 4, "n_gpus": world, "scaling": "weak", "workload": "BWT %d x 256 KiB per GPU, G-%s" % (nb, kind),
             "forward_GiB/s": round(world * tot / tf / 2**30, 3), "forward_ms": round(tf * 1e3, 2),
             "inverse_GiB/s": round(world * tot / ti / 2**30, 3), "inverse_ms": round(ti * 1e3, 2),
             "forward_roofline": _roof(alg, tf, pmc_traffic(4, "forward_" + kind), 6),
             "inverse_roofline": _roof(alg, ti, pmc_traffic(4, "inverse_" + kind), 6)}
        if cpu and rank == 0 and world == 1:
            ns = min(nb, 256)
            raw_h, L_h = raw[: ns * BLOCK].cpu().numpy(), fw.out_base[: ns * BLOCK].cpu().numpy()
            aux_h = fw.aux[:ns].cpu().numpy().astype(np.uint32)
            offs, caps = ar[:ns] * BLOCK, np.full(ns, BLOCK)
            aux_o = np.zeros(ns, dtype=np.uint32)
            r["cpu_baseline_forward"], Lo = oracle_rate(N.BWT_FORWARD, raw_h, offs, caps, ns * BLOCK, offs, caps, BLOCK,
                                                        "oracle bwt::encode (comparison-sorted suffixes), %d of the %d blocks" % (ns, nb), aux=aux_o,
                                                        check=lambda o: np.array_equal(o[: ns * BLOCK], L_h) and np.array_equal(aux_o, aux_h))
            r["cpu_baseline_inverse"], _ = oracle_rate(N.BWT_INVERSE, L_h, offs, caps, ns * BLOCK, offs, caps, BLOCK,
                                                       "oracle bwt::decode (inversion table + pointer chase), %d of the %d blocks" % (ns, nb), aux=aux_h,
                                                       check=lambda o: np.array_equal(o[: ns * BLOCK], raw_h))
            r["cpu_baseline_forward"].pop("_out_len", None); r["cpu_baseline_inverse"].pop("_out_len", None)
        out.append(r)
        del raw, fw, inv
    return out


def _pipeline_cpu(torch, pipe, raw, lens, stages, ns):
    """The oracle's BWT -> DC -> Ari and back over the first `ns` blocks (every host core), checked against the device's stages."""
    from rust_compress_amd import _native as N
    from rust_compress_amd import pipeline as P
    BLOCK = int(lens[0])
    S = pipe.S
    ar = np.arange(ns, dtype=np.int64)
    raw_h = raw[: ns * BLOCK].cpu().numpy()
    offs, caps = ar * BLOCK, np.full(ns, BLOCK)
    aux_o = np.zeros(ns, dtype=np.uint32)
    enc = {}
    enc["bwt"], L_h = oracle_rate(N.BWT_FORWARD, raw_h, offs, caps, ns * BLOCK, offs, caps, BLOCK, "bwt", aux=aux_o, budget_s=2.0)
    slot = 4 * (256 + BLOCK) + 64
    enc["dc"], dcw = oracle_rate(N.DC_ENCODE, L_h, offs, caps, ns * slot, ar * slot, np.full(ns, slot), BLOCK, "dc", budget_s=1.0)
    # the device's records and pieces for the same blocks: the coder's input / output
    roff = np.asarray(stages["rec_off"][:ns], dtype=np.int64)
    slot_dev = int(stages["rec_off"][1]) if len(stages["rec_off"]) > 1 else int(stages["rec"].numel())
    rec = stages["rec"][: ns * slot_dev + 64].cpu().numpy()
    rec_len = np.asarray(stages["rec_len"][:ns].cpu().numpy(), dtype=np.int64)
    cuts = stages["cuts"][:ns].cpu().numpy().astype(np.int64)
    pin = (roff[:, None] + cuts[:, :S]).reshape(-1)
    plen = (cuts[:, 1:] - cuts[:, :-1]).reshape(-1)
    same = all(np.array_equal(rec[int(roff[b]) + 12: int(roff[b]) + int(rec_len[b])], dcw[int(b * slot): int(b * slot) + int(rec_len[b]) - 12]) for b in range(min(ns, 8)))
    cslot = 2 * (int(plen.max()) + 8) + 64
    co = np.arange(ns * S, dtype=np.int64) * cslot
    enc["ari"], coded = oracle_rate(N.ARI_BYTE_ENCODE, rec, pin, plen, ns * S * cslot, co, np.full(ns * S, cslot), BLOCK / S, "ari", budget_s=2.0)
    ar_dev = stages["ari"]
    dev_len = ar_dev.out_len[: ns * S].cpu().numpy().astype(np.int64)
    dev_off = ar_dev.out_off[: ns * S].cpu().numpy().astype(np.int64)
    dev_out = ar_dev.out_base[: int(dev_off[-1] + dev_len[-1])].cpu().numpy()
    clen_o = enc["ari"]["_out_len"]
    same = same and bool(np.array_equal(clen_o, dev_len)) and all(
        np.array_equal(coded[int(co[i]): int(co[i]) + int(dev_len[i])], dev_out[int(dev_off[i]): int(dev_off[i]) + int(dev_len[i])]) for i in range(0, ns * S, max(1, ns * S // 64)))
    dec = {}
    dec["ari"], rec_o = oracle_rate(N.ARI_BYTE_DECODE, coded, co, clen_o, len(rec), pin, plen, BLOCK / S, "ari", budget_s=2.0)
    k = (rec_len - 12) // 4 - 256
    dec["dc"], L2 = oracle_rate(N.DC_DECODE, rec, roff + 12, 4 * (256 + k), ns * BLOCK, offs, caps, BLOCK, "dc", n_out=caps.astype(np.uint64), budget_s=1.0)
    dec["bwt"], back = oracle_rate(N.BWT_INVERSE, L_h, offs, caps, ns * BLOCK, offs, caps, BLOCK, "bwt", aux=aux_o, budget_s=2.0)
    ok = same and bool(np.array_equal(back[: ns * BLOCK], raw_h)) and bool(np.array_equal(L2[: ns * BLOCK], L_h[: ns * BLOCK]))
    for d in (enc, dec):
        for v in d.values():
            v.pop("_out_len", None)
    def total(d):
        return 1.0 / sum(1.0 / d[s]["value"] for s in d)
    cores = os.cpu_count() or 1
    note = "oracle stages one after another over the first %d of %d blocks, %d threads; per stage GiB/s %s; DC words / Ari bytes / round trip == the device's: %s"
    return ({"value": round(total(dec), 3), "unit": "GiB/s decoded", "cores": cores, "kind": "port",
             "sample": note % (ns, len(lens), cores, {s: dec[s]["value"] for s in dec}, ok)},
            {"value": round(total(enc), 3), "unit": "GiB/s encoded", "cores": cores, "kind": "port",
             "sample": note % (ns, len(lens), cores, {s: enc[s]["value"] for s in enc}, ok)})


def config5(ctx, torch, dev, scale=1.0, reps=4, cpu=False, rank=0, world=1, dist=None, once=False):
    """ONE stream of 10^9 x scale bytes, sharded by block ranges (dist.partition over its 3815 blocks): the root scatters the raw
    ranges, every rank encodes and decodes its own, the root gathers the decoded ranges and checks decode(encode(x)) == x."""
    from rust_compress_amd import synth, pipeline as P, dist as D
    BLOCK = 262144
    total = int(1e9 * scale)
    lens = np.array([BLOCK] * (total // BLOCK) + ([total % BLOCK] if total % BLOCK else []), dtype=np.int64)
    sharded = world > 1 and dist is not None
    raw = None
    if rank == 0:
        data = np.concatenate([synth.gen("text", min(BLOCK * 256, total - s), 0xC0 + s) for s in range(0, total, BLOCK * 256)])[:total]
        raw = torch.from_numpy(data).to(dev)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    bounds = D.partition(lens, world)
    ts = tg = 0.0
    if sharded:
        torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
        local, loff, llen, bnd = D.scatter_blocks(raw, offs, lens, bounds, root=0, device=dev)
        torch.cuda.synchronize(); dist.barrier(); ts = time.perf_counter() - t0
        llen = llen.astype(np.int64)
    else:
        local, llen, bnd = raw, lens, bounds
    # two lanes (pipeline.PipelineLanes: a host thread, a context and a stream each, half the blocks each): the stages' dependent
    # chains run under each other -- 10^9 bytes: decode 0.050 -> 0.044 s, encode 0.123 -> 0.117 (benchmarks/r4_pipe_lanes.py)
    lanes = int(os.environ.get("RCX_PIPE_LANES", "2"))
    pipe = P.PipelineLanes(dev, lanes=lanes) if lanes > 1 and not once else P.BwtDcAri(ctx, dev)
    te = td = 1e9
    n_rep = 1 if once else reps
    stages = None
    for rep in range(n_rep):                 # the first pass pays the one-off scratch / output allocations; best of the rest
        if sharded:
            torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter(); comp, coff, clen, praw, stages = pipe.encode(local, llen, keep_stages=(cpu and rep == n_rep - 1)); torch.cuda.synchronize(); e_ = time.perf_counter() - t0
        if sharded:
            dist.barrier()
        t0 = time.perf_counter(); back = pipe.decode(comp, coff, clen, praw, llen); torch.cuda.synchronize(); d_ = time.perf_counter() - t0
        if os.environ.get("RCX_CFG_VERBOSE"):
            print("config 5 rep %d: encode %.4f s decode %.4f s" % (rep, e_, d_), file=sys.stderr)
        if rep or n_rep == 1:
            te, td = min(te, e_), min(td, d_)
    csum = int(clen.sum())
    if sharded:
        tt = torch.tensor([te, td, float(csum)], dtype=torch.float64, device=dev)
        mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        te, td, csum = float(mx[0]), float(mx[1]), int(sm[2])
        ooff = np.concatenate([[0], np.cumsum(llen)[:-1]]) if len(llen) else np.zeros(0, np.int64)
        torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
        got, glens = D.gather_blocks(back, ooff, llen, bnd, root=0)
        torch.cuda.synchronize(); dist.barrier(); tg = time.perf_counter() - t0
        if rank == 0:
            assert torch.equal(got, raw), "sharded pipeline: decode(encode(x)) != x on the root"
    else:
        assert torch.equal(back, raw)
    if rank != 0:
        return None
    res = {"config": 5, "n_gpus": world, "scaling": "strong (one stream, sharded by block ranges)",
           "workload": "BWT->DC->Ari, %d bytes in %d blocks of 256 KiB; block ranges per rank %s" % (total, len(lens), np.diff(bounds).tolist()),
           "lanes_per_gpu": lanes if lanes > 1 and not once else 1,
           "compressed_ratio": round(total / csum, 3),
           "encode_GiB/s": round(total / te / 2**30, 3), "decode_GiB/s": round(total / td / 2**30, 3), "encode_s": round(te, 3), "decode_s": round(td, 3),
           "scatter_raw_s": round(ts, 3), "gather_decoded_s": round(tg, 3),
           "end_to_end_GiB/s": round(total / (ts + te + td + tg) / 2**30, 3) if sharded else None,
           "decode_roofline": _roof(total + csum, td, pmc_traffic(5, "decode"), 6)}
    if cpu and world == 1 and stages is not None:
        ns = min(len(lens), 256, int(stages.get("nblocks", len(lens))))      # (with lanes the stages kept are the first group's)
        res["cpu_baseline"], res["cpu_baseline_encode"] = _pipeline_cpu(torch, pipe, raw, lens, stages, ns)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="3,4,5")
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the full config size")
    ap.add_argument("--kinds", default="text,dna4", help="config 4: the distributions")
    ap.add_argument("--once", action="store_true", help="every launch exactly once, no warm-up (PMC passes: counters per launch)")
    ap.add_argument("--cpu", action="store_true", help="also time the oracle (CPU port of the reference) on a bounded sample")
    args = ap.parse_args()
    import torch
    import rust_compress_amd as R
    dev = torch.device("cuda", 0)
    ctx = R.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for cfg in args.configs.split(","):
        if cfg == "3":
            print(json.dumps(config3(ctx, torch, dev, args.scale, cpu=args.cpu, once=args.once)), flush=True)
        elif cfg == "3g":                                   # the same members in gzip framing (extension)
            print(json.dumps(config3(ctx, torch, dev, args.scale, gzip_framing=True)), flush=True)
        elif cfg == "4":
            for r in config4(ctx, torch, dev, args.scale, kinds=tuple(args.kinds.split(",")), cpu=args.cpu, once=args.once):
                print(json.dumps(r), flush=True)
        elif cfg == "5":
            print(json.dumps(config5(ctx, torch, dev, args.scale, cpu=args.cpu, once=args.once)), flush=True)
        if os.environ.get("RCX_CFG_KEEP_CACHE") is None:    # as bench.py does between its side configs: the next one starts from an empty allocator cache
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
