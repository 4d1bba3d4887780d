"""GPU parity: LZ4 block decode/encode through the C-ABI vs the oracle (bit-exact) -- fixtures, synthetic
distributions, ragged / empty / misaligned blocks, malformed inputs, and BASELINE's full size
(4096 x 64 KiB) through round-trip properties."""
import numpy as np
import pytest

from rust_compress_amd import _native as N
from rust_compress_amd import synth

pytestmark = pytest.mark.gpu


def _raws(golden):
    rng = np.random.default_rng(1)
    raws = [b"", b"a", b"a" * 54, b"abcd" * 9, golden("test.txt")]
    for kind in ("text", "runs", "rand", "dna4"):
        for sz in (65536, 5000, 70001, 262144):
            raws.append(synth.gen(kind, sz, 7).tobytes())
    raws += [b"\0" * 65536, b"ab" * 30000, bytes(range(256)) * 100,
             b"x" * 100 + bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) + b"x" * 5000,
             (b"abcdefghijklmnopqrstuvwxyz0123456789" * 3 + b"Q") * 500]
    return raws


@pytest.mark.parametrize("variant", [0, 1, 2])     # windowed probe (8 lanes widening), serial probe chain, always 64 lanes
def test_encode_bit_exact_vs_oracle(ctx, oracle, golden, variant):
    raws = _raws(golden)
    raws += [synth.gen(k, 300000, 11).tobytes() for k in ("rand", "mix", "words")]    # skip acceleration, back-tracking
    ctx.set_variant(N.LZ4_ENCODE, variant)
    try:
        res = ctx.lz4_encode_blocks(raws).check()
    finally:
        ctx.set_variant(N.LZ4_ENCODE, 0)
    for r, e in zip(raws, res.outputs):
        assert e == oracle.lz4_encode_block(r)
    assert list(res.in_used) == [len(r) for r in raws]


@pytest.mark.parametrize("variant", N.LZ4_DECODE_VARIANTS)
def test_decode_bit_exact_vs_oracle(ctx, oracle, golden, variant):
    ctx.set_variant(N.LZ4_DECODE, variant)
    raws = _raws(golden)
    blobs = [oracle.lz4_encode_block(r) for r in raws]
    blobs.append(golden("test.lz4.1")[11:11 + 2722])      # the reference's own compressed blocks (lz4.rs:647-659)
    blobs.append(golden("test.lz4.9")[11:11 + 2664])
    raws += [golden("test.txt")] * 2
    res = ctx.lz4_decode_blocks(blobs, [len(r) for r in raws]).check()
    assert res.outputs == raws
    assert list(res.in_used) == [len(b) for b in blobs]
    for b, r in zip(blobs, raws):
        assert oracle.lz4_decode_block(b, cap=len(r)) == r
    ctx.set_variant(N.LZ4_DECODE, 0)


@pytest.mark.parametrize("variant", N.LZ4_DECODE_VARIANTS[:6])
def test_decode_malformed_statuses(ctx, oracle, variant):
    ctx.set_variant(N.LZ4_DECODE, variant)
    rng = np.random.default_rng(5)
    base = [oracle.lz4_encode_block(synth.gen(k, 3000 + 500 * i, i).tobytes())
            for i, k in enumerate(("text", "runs", "rand", "text", "runs", "dna4"))]
    blobs, caps = [], []
    for it in range(600):
        b = bytearray(base[it % len(base)])
        mode = it % 5
        if mode == 0:
            for _ in range(rng.integers(1, 4)):
                b[rng.integers(0, len(b))] = rng.integers(0, 256)
        elif mode == 1:
            b = b[: rng.integers(0, len(b))]
        elif mode == 2:
            b = b + bytes(rng.integers(0, 256, rng.integers(1, 40), dtype=np.uint8))
        elif mode == 3:
            b = bytearray(rng.integers(0, 256, rng.integers(0, 300), dtype=np.uint8).tobytes())
        blobs.append(bytes(b))
        caps.append(int(rng.choice([100, 3000, 5000, 200000])))
    res = ctx.lz4_decode_blocks(blobs, caps)
    for i, (b, c) in enumerate(zip(blobs, caps)):
        eo, es = oracle.lz4_decode_block(b, cap=c, raise_on_error=False)
        assert es == res.status[i], (i, es, res.status[i])
        if es == 0:
            assert eo == res.outputs[i]
    ctx.set_variant(N.LZ4_DECODE, 0)


@pytest.mark.parametrize("variant", N.LZ4_DECODE_VARIANTS[:2])
def test_decode_runs_of_every_offset_and_length(ctx, oracle, variant):
    """Runs (matches that overlap themselves) of every offset 1..15 and lengths on both sides of the parser's piece boundaries
    (tests/corpus.py: lz4_run_streams): the parser hands a run over as pieces whose offsets are whole periods back (k_lz4_decode_v8.hip,
    RCX_RUNSPLIT); whole, cut short and with too little room -- bytes and statuses as the oracle's."""
    import corpus
    rng = np.random.default_rng(3)
    blobs, raws = corpus.lz4_run_streams(oracle)
    ctx.set_variant(N.LZ4_DECODE, variant)
    try:
        for rot in (0, 1, 2):
            bl, rw = blobs[rot:] + blobs[:rot], raws[rot:] + raws[:rot]
            res = ctx.lz4_decode_blocks(bl, [len(r) for r in rw]).check()
            assert res.outputs == rw and list(res.in_used) == [len(b) for b in bl], rot
        cut = [b[: int(rng.integers(0, len(b)))] for b in blobs]
        caps = [int(rng.integers(0, len(r) + 1)) for r in raws]
        for bl, cp in ((cut, [len(r) for r in raws]), (blobs, caps)):
            res = ctx.lz4_decode_blocks(bl, cp)
            for i, (b, c) in enumerate(zip(bl, cp)):
                eo, es = oracle.lz4_decode_block(b, cap=c, raise_on_error=False)
                assert es == res.status[i], (i, es, res.status[i])
                if es == 0:
                    assert eo == res.outputs[i]
    finally:
        ctx.set_variant(N.LZ4_DECODE, 0)


@pytest.mark.parametrize("variant", N.LZ4_DECODE_VARIANTS[:2])
def test_decode_long_runs_and_chunk_edges(ctx, oracle, variant):
    """Length extensions on both sides of every boundary the parser has (k_lz4_decode_v8.hip, next_tok_c / fields): literal runs and
    matches of 14..16, 269..271 (one extension byte, 254 / 255 / 255+0), 524..526 and 1000+ bytes, a text of short tokens
    between them so that the long ones fall at every phase of the 4 KiB chunks and 64-byte segments, at many input alignments,
    and within the last 20 bytes of the block (tests/corpus.py: lz4_edge_streams)."""
    import corpus
    rng = np.random.default_rng(78)
    blobs, raws = corpus.lz4_edge_streams(oracle, 160, 77)
    ctx.set_variant(N.LZ4_DECODE, variant)
    try:
        for rot in (0, 1, 2, 3):                       # (blocks are packed back to back: every rotation gives every block another input alignment)
            bl, rw = blobs[rot:] + blobs[:rot], raws[rot:] + raws[:rot]
            res = ctx.lz4_decode_blocks(bl, [len(r) for r in rw])
            res.check()
            assert res.outputs == rw, rot
            assert list(res.in_used) == [len(b) for b in bl]
        # the same streams cut short and with too little room: statuses as the reference's
        cut = [b[: int(rng.integers(0, len(b)))] for b in blobs]
        caps = [int(rng.integers(0, len(r) + 1)) for r in raws]
        for bl, cp in ((cut, [len(r) for r in raws]), (blobs, caps)):
            res = ctx.lz4_decode_blocks(bl, cp)
            for i, (b, c) in enumerate(zip(bl, cp)):
                eo, es = oracle.lz4_decode_block(b, cap=c, raise_on_error=False)
                assert es == res.status[i], (i, es, res.status[i])
                if es == 0:
                    assert eo == res.outputs[i]
    finally:
        ctx.set_variant(N.LZ4_DECODE, 0)


@pytest.mark.parametrize("kind", ["text", "mix", "runs", "rand"])
def test_full_size_roundtrip_device_resident(ctx, oracle, kind):
    """BASELINE configs[1]: 4096 x 64 KiB, device-resident: decode(encode(x)) == x for every block, and a
    sample of blocks is compared with the oracle's encoder/decoder bytes."""
    import torch
    import rust_compress_amd as R
    nb, BLOCK = 4096, 65536
    dev = torch.device("cuda", 0)
    raw_np = synth.gen_blocks(kind, nb, BLOCK, 0x4C5A3401)
    raw = torch.from_numpy(raw_np).to(dev)
    slot = (int(N.lib().rcx_lz4_compression_bound(BLOCK)) + 63) // 64 * 64
    i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
    ar = np.arange(nb, dtype=np.int64)
    enc = R.DeviceBatch(raw, i64(ar * BLOCK), i64(np.full(nb, BLOCK)),
                        torch.zeros(nb * slot + 64, dtype=torch.uint8, device=dev), i64(ar * slot), i64(np.full(nb, slot)))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    scratch = torch.empty(ctx.scratch_bytes(N.LZ4_ENCODE, nb, BLOCK) + 64, dtype=torch.uint8, device=dev)
    ctx.launch_dev(N.LZ4_ENCODE, enc, scratch)
    torch.cuda.synchronize()
    assert int(enc.status.abs().max()) == 0
    clen = enc.out_len[:nb].clone()
    for variant in N.LZ4_DECODE_VARIANTS[:6]:
        ctx.set_variant(N.LZ4_DECODE, variant)
        dec = R.DeviceBatch(enc.out_base, enc.out_off, clen, torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev),
                            i64(ar * BLOCK), i64(np.full(nb, BLOCK)))
        ctx.launch_dev(N.LZ4_DECODE, dec)
        torch.cuda.synchronize()
        assert int(dec.status.abs().max()) == 0
        assert bool((dec.out_len[:nb] == BLOCK).all())
        assert torch.equal(dec.out_base[: nb * BLOCK], raw)
    ctx.set_variant(N.LZ4_DECODE, 0)
    comp = enc.out_base.cpu().numpy()
    cl = clen.cpu().numpy()
    for i in range(0, nb, 257):
        blob = comp[i * slot: i * slot + int(cl[i])].tobytes()
        src = raw_np[i * BLOCK:(i + 1) * BLOCK].tobytes()
        assert blob == oracle.lz4_encode_block(src)
        assert oracle.lz4_decode_block(blob, cap=BLOCK) == src
    ctx.set_stream(0)


def test_multi_device_entry_points(ctx, oracle):
    """include/rcx.h rcx_multi_*: a host-memory batch sharded over "devices" (the one GPU listed three times: three contexts,
    three host threads, three ranges) returns block for block what one context returns -- LZ4 both ways, and BWT forward /
    inverse for the aux arrays that travel with a range; device-resident ranges through rcx_multi_launch_dev."""
    import ctypes as C
    import torch
    import rust_compress_amd as R
    from rust_compress_amd import batch as B
    L = N.lib()
    raws = [synth.gen(("text", "runs", "rand")[i % 3], 1000 + 997 * (i % 23), 70 + i).tobytes() for i in range(101)] + [b"", b"x"]
    devs = (C.c_int * 3)(0, 0, 0)
    h = C.c_void_p()
    assert L.rcx_multi_create(devs, 3, C.byref(h)) == 0 and L.rcx_multi_count(h) == 3 and L.rcx_multi_ctx(h, 2)
    try:
        def multi(codec, blobs, caps, aux_in=None, want_aux=False):
            n = len(blobs)
            base, off, lens = B.pack(blobs)
            total, ooff, ocap = B.layout(caps)
            out = np.zeros(total + 64, np.uint8)
            out_len, in_used, status = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.full(n, -9, np.int32)
            aux = np.zeros(n, np.uint32)
            p = lambda a: a.ctypes.data
            b = N.Batch(p(base), p(off), p(lens), p(out), p(ooff), p(ocap), p(out_len), p(in_used), p(status), n, N.MEM_HOST)
            ai = np.ascontiguousarray(aux_in, dtype=np.uint32) if aux_in is not None else None
            rc = L.rcx_multi_batch(h, codec, C.byref(b), p(ai) if ai is not None else None, p(aux) if want_aux else None, None)
            assert rc == 0, L.rcx_multi_last_error(h)
            assert not status.any()
            return [bytes(out[int(o): int(o) + int(l)]) for o, l in zip(ooff, out_len)], aux
        enc, _ = multi(N.LZ4_ENCODE, raws, [int(L.rcx_lz4_compression_bound(len(r))) for r in raws])
        assert enc == ctx.lz4_encode_blocks(raws).check().outputs == [oracle.lz4_encode_block(r) for r in raws]
        dec, _ = multi(N.LZ4_DECODE, enc, [len(r) for r in raws])
        assert dec == raws
        nz = [r for r in raws if r]
        fw, origin = multi(N.BWT_FORWARD, nz, [len(r) for r in nz], want_aux=True)
        one = ctx.bwt_forward(nz).check()
        assert fw == one.outputs and origin.tolist() == [int(x) for x in one.aux]
        back, _ = multi(N.BWT_INVERSE, fw, [len(r) for r in nz], aux_in=origin)
        assert back == nz
        # device-resident ranges: two DeviceBatches (both on device 0 here), enqueued on their contexts' streams, one sync
        dev = torch.device("cuda", 0)
        halves = [enc[:50], enc[50:]]
        dbs = []
        for part, rr in zip(halves, (raws[:50], raws[50:])):
            base, off, lens = B.pack(part)
            total, ooff, ocap = B.layout([len(r) for r in rr])
            dbs.append(R.DeviceBatch.from_host(base, off, lens, total, ooff, ocap, dev))
        torch.cuda.synchronize()
        arr = (C.POINTER(N.DevBatch) * 3)(C.pointer(dbs[0].c), C.pointer(dbs[1].c), None)
        assert L.rcx_multi_launch_dev(h, N.LZ4_DECODE, arr, None, None) == 0 and L.rcx_multi_sync(h) == 0
        for db, rr in zip(dbs, (raws[:50], raws[50:])):
            assert int(db.status[: len(rr)].abs().max()) == 0
            got = db.out_base.cpu().numpy()
            oo, ol = db.out_off.cpu().numpy(), db.out_len.cpu().numpy()
            assert [bytes(got[int(o): int(o) + int(l)]) for o, l in zip(oo, ol)][: len(rr)] == rr
    finally:
        L.rcx_multi_destroy(h)


def test_multi_device_to_device_shard_path(ctx, oracle):
    """rcx_multi_scatter_dev / rcx_multi_gather_dev (include/rcx.h): the whole batch sits on ONE device; the compressed ranges travel to
    the devices that decode them and the decoded ranges come back, device to device.  On this box the set lists the one GPU three
    times (three contexts, three streams; transport "peer": RCCL wants one rank per device) -- the result must be the one-context
    result, block for block, with nothing waited for between scatter, launch and gather.  And a set of ONE device with
    RCX_MULTI_TRANSPORT=rccl sends its range to itself through a grouped ncclSend / ncclRecv: librccl loaded, communicator built,
    group issued."""
    import ctypes as C
    import os
    import torch
    import rust_compress_amd as R
    from rust_compress_amd import batch as B
    L = N.lib()
    dev = torch.device("cuda", 0)
    raws = [synth.gen(("text", "runs", "rand", "mix")[i % 4], 3000 + 1013 * (i % 29), 170 + i).tobytes() for i in range(150)]
    enc = ctx.lz4_encode_blocks(raws).check().outputs
    base, off, lens = B.pack(enc)
    caps = [len(r) for r in raws]
    total, ooff, ocap = B.layout(caps)
    root_in = torch.from_numpy(base).to(dev)
    root_out = torch.zeros(total + 64, dtype=torch.uint8, device=dev)

    def run(devices, env):
        G = len(devices)
        old = os.environ.get("RCX_MULTI_TRANSPORT")
        if env is None:
            os.environ.pop("RCX_MULTI_TRANSPORT", None)
        else:
            os.environ["RCX_MULTI_TRANSPORT"] = env
        h = C.c_void_p()
        assert L.rcx_multi_create((C.c_int * G)(*devices), G, C.byref(h)) == 0
        try:
            bounds = (C.c_uint32 * (G + 1))()
            w = np.ascontiguousarray(ocap, dtype=np.uint64)
            L.rcx_partition(w.ctypes.data, len(enc), G, bounds)
            bnd = list(bounds)
            in_end = [int(off[b]) if b < len(enc) else int(off[-1] + ((lens[-1] + 15) // 16) * 16) for b in bnd]
            out_end = [int(ooff[b]) if b < len(enc) else total for b in bnd]
            r_in = (C.c_uint64 * (G + 1))(*in_end); r_out = (C.c_uint64 * (G + 1))(*out_end)
            # every range gets buffers of its own on "its" device (range-relative), the root's range too: the copy is exercised for all
            pin = [torch.zeros(max(in_end[g + 1] - in_end[g], 16) + 64, dtype=torch.uint8, device=dev) for g in range(G)]
            pout = [torch.zeros(max(out_end[g + 1] - out_end[g], 16) + 64, dtype=torch.uint8, device=dev) for g in range(G)]
            dbs = []
            i64 = lambda a: torch.tensor(np.asarray(a, dtype=np.int64), dtype=torch.int64, device=dev)
            for g in range(G):
                a0, a1 = bnd[g], bnd[g + 1]
                dbs.append(R.DeviceBatch(pin[g], i64(off[a0:a1].astype(np.int64) - in_end[g]), i64(lens[a0:a1].astype(np.int64)),
                                         pout[g], i64(ooff[a0:a1].astype(np.int64) - out_end[g]), i64(ocap[a0:a1].astype(np.int64))) if a1 > a0 else None)
            root_out.zero_()
            torch.cuda.synchronize()
            pi = (C.c_void_p * G)(*[t.data_ptr() for t in pin]); po = (C.c_void_p * G)(*[t.data_ptr() for t in pout])
            arr = (C.POINTER(N.DevBatch) * G)(*[C.pointer(d.c) if d is not None else None for d in dbs])
            assert L.rcx_multi_scatter_dev(h, 0, root_in.data_ptr(), r_in, pi) == 0, L.rcx_multi_last_error(h)
            assert L.rcx_multi_launch_dev(h, N.LZ4_DECODE, arr, None, None) == 0, L.rcx_multi_last_error(h)
            assert L.rcx_multi_gather_dev(h, 0, root_out.data_ptr(), r_out, po) == 0, L.rcx_multi_last_error(h)
            assert L.rcx_multi_sync(h) == 0
            name = L.rcx_multi_transport(h).decode()
            for d in dbs:
                if d is not None:
                    assert int(d.status[: d.n].abs().max()) == 0
            got = root_out.cpu().numpy()
            assert [bytes(got[int(o): int(o) + c]) for o, c in zip(ooff, caps)] == raws
            return name
        finally:
            L.rcx_multi_destroy(h)
            if old is None:
                os.environ.pop("RCX_MULTI_TRANSPORT", None)
            else:
                os.environ["RCX_MULTI_TRANSPORT"] = old

    assert run([0, 0, 0], None) == "peer"
    assert run([0, 0], "peer") == "peer"
    assert run([0], "rccl") == "rccl"                    # (a send to oneself inside a group: the transport an N-device node takes)



def test_host_path_into_page_locked_output(ctx, oracle):
    """rcx_lz4_decode_batch with RCX_MEM_HOST and a PAGE-LOCKED output buffer (rcx_api.hip: the decoder stores what leaves its window
    straight into the caller's buffer, the compressed bytes travel in as block ranges under the launches before them) returns what
    the plain copies return (rcx_ctx_set_param(ctx, RCX_LZ4_DECODE, 1)) and what the oracle returns: ragged and empty blocks, the
    wave-wide paths (incompressible blocks, long runs), the reference's statuses on corrupted blocks among the good ones, output
    slots at odd addresses with room to spare (not a byte beyond a block's decoded length is touched), blocks listed in reverse
    input order (one range), and enough blocks for several ranges."""
    import ctypes as C
    import torch
    import corpus
    from rust_compress_amd import batch as B
    L = N.lib()
    rng = np.random.default_rng(2026)
    kinds = ("text", "runs", "rand", "dna4", "mix")
    raws = [synth.gen(kinds[i % 5], int(rng.choice([0, 1, 13, 700, 4096, 20000, 65536, 100000])) if i % 7 else 65536, 300 + i).tobytes() for i in range(700)]
    blobs = [oracle.lz4_encode_block(r) for r in raws]
    eb, er = corpus.lz4_edge_streams(oracle, 60, 91)
    blobs += eb; raws += er
    caps = [len(r) + int(rng.integers(0, 40)) for r in raws]
    for i in range(0, len(blobs), 37):                    # corrupted / short of room among the good ones
        b = bytearray(blobs[i])
        if len(b) > 8 and i % 2: b[int(rng.integers(0, len(b)))] ^= 0x5a
        elif len(b) > 8: b = b[: int(rng.integers(1, len(b)))]
        blobs[i] = bytes(b)
        if i % 3 == 0 and caps[i] > 10: caps[i] = caps[i] // 2
    n = len(blobs)
    want = [oracle.lz4_decode_block(b, cap=c, raise_on_error=False) for b, c in zip(blobs, caps)]

    def run(order, plain, out_mis):
        bl = [blobs[i] for i in order]
        base, off, lens = B.pack(bl)
        total, ooff, ocap = B.layout([caps[i] for i in order])
        ooff = ooff + np.uint64(out_mis)
        inb = torch.from_numpy(base).pin_memory()
        outb = torch.full((int(total) + 256,), 0xAA, dtype=torch.uint8).pin_memory()
        out_len, in_used, status = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.full(n, -9, np.int32)
        p = lambda a: a.ctypes.data
        b = N.Batch(inb.data_ptr(), p(off), p(lens), outb.data_ptr() + 3, p(ooff), p(ocap), p(out_len), p(in_used), p(status), n, N.MEM_HOST)
        assert L.rcx_ctx_set_param(ctx._h, N.LZ4_DECODE, 1 if plain else 0) == 0
        try:
            assert L.rcx_lz4_decode_batch(ctx._h, C.byref(b)) == 0, L.rcx_last_error(ctx._h)
        finally:
            L.rcx_ctx_set_param(ctx._h, N.LZ4_DECODE, 0)
        got = outb.numpy()[3:]
        for j, i in enumerate(order):
            eo, es = want[i]
            assert es == status[j], (plain, i, es, status[j])
            o = int(ooff[j])
            if es == 0:
                assert int(out_len[j]) == len(eo) and bytes(got[o: o + len(eo)]) == eo, (plain, i)
                assert int(in_used[j]) == len(blobs[i])
                if not plain:                             # (the plain path copies one span back: the room between the blocks comes with it)
                    assert (got[o + len(eo): o + int(ocap[j])] == 0xAA).all(), (i, "bytes beyond the decoded length were written")
        return status.copy(), out_len.copy()

    fwd = list(range(n))
    s0, l0 = run(fwd, True, 0)
    s1, l1 = run(fwd, False, 0)
    assert (s0 == s1).all() and (l0 == l1).all()
    run(fwd, False, 5)
    # blocks listed in reverse input order: the ranges' spans would overlap, the call takes one range
    bl = [blobs[i] for i in fwd]
    base, off, lens = B.pack(bl)
    total, ooff, ocap = B.layout(caps)
    inb = torch.from_numpy(base).pin_memory()
    outb = torch.full((int(total) + 64,), 0xAA, dtype=torch.uint8).pin_memory()
    rev = np.arange(n)[::-1].copy()
    off_r, lens_r, ooff_r, ocap_r = (np.ascontiguousarray(a[rev]) for a in (off, lens, ooff, ocap))
    out_len, in_used, status = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.full(n, -9, np.int32)
    p = lambda a: a.ctypes.data
    b = N.Batch(inb.data_ptr(), p(off_r), p(lens_r), outb.data_ptr(), p(ooff_r), p(ocap_r), p(out_len), p(in_used), p(status), n, N.MEM_HOST)
    assert L.rcx_lz4_decode_batch(ctx._h, C.byref(b)) == 0
    got = outb.numpy()
    for j in range(n):
        i = int(rev[j]); eo, es = want[i]
        assert es == status[j]
        if es == 0:
            assert bytes(got[int(ooff[i]): int(ooff[i]) + len(eo)]) == eo
    # the same page-locked buffers through rcx_multi_batch (two contexts on the one device: two gated launches side by side)
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert L.rcx_multi_create(devs, 2, C.byref(h)) == 0
    try:
        outb.fill_(0xAA)
        out_len[:] = 0; in_used[:] = 0; status[:] = -9
        b = N.Batch(inb.data_ptr(), p(off), p(lens), outb.data_ptr(), p(ooff), p(ocap), p(out_len), p(in_used), p(status), n, N.MEM_HOST)
        for rep in range(2):
            assert L.rcx_multi_batch(h, N.LZ4_DECODE, C.byref(b), None, None, None) == 0, L.rcx_multi_last_error(h)
        got = outb.numpy()
        for i in range(n):
            eo, es = want[i]
            assert es == status[i]
            if es == 0:
                assert int(out_len[i]) == len(eo) and bytes(got[int(ooff[i]): int(ooff[i]) + len(eo)]) == eo
    finally:
        L.rcx_multi_destroy(h)
    # ordinary (pageable) allocations page-locked through the C-ABI, as a host language without the HIP headers pins a Vec's memory
    # (include/rcx.h rcx_host_register): the mirrored path again, the same bytes
    inn = np.ascontiguousarray(base)
    outn = np.full(int(total) + 64, 0xAA, np.uint8)
    assert L.rcx_host_register(inn.ctypes.data, inn.size) == 0 and L.rcx_host_register(outn.ctypes.data, outn.size) == 0
    try:
        out_len[:] = 0; in_used[:] = 0; status[:] = -9
        b = N.Batch(p(inn), p(off), p(lens), p(outn), p(ooff), p(ocap), p(out_len), p(in_used), p(status), n, N.MEM_HOST)
        assert L.rcx_lz4_decode_batch(ctx._h, C.byref(b)) == 0, L.rcx_last_error(ctx._h)
        for i in range(n):
            eo, es = want[i]
            assert es == status[i]
            if es == 0:
                o = int(ooff[i])
                assert int(out_len[i]) == len(eo) and bytes(outn[o: o + len(eo)]) == eo
                assert (outn[o + len(eo): o + int(ocap[i])] == 0xAA).all()
    finally:
        assert L.rcx_host_unregister(outn.ctypes.data) == 0 and L.rcx_host_unregister(inn.ctypes.data) == 0


def test_host_path_ranges_see_fresh_input(ctx, oracle):
    """The gated launch reads input that lands in the staging buffer WHILE it runs: a line of that buffer still cached from the call
    before would be a stale read nobody notices as long as every call carries the same bytes.  So: two different sets of blocks of
    the same total size, alternately through one context (the same staging buffer, the same offsets, other bytes), page-locked
    in and out, three rounds -- LZ4 blocks and zlib members."""
    import ctypes as C
    import zlib
    import torch
    from rust_compress_amd import batch as B
    L = N.lib()
    sets = []
    for s in (0, 1):
        raws = [synth.gen(("text", "runs", "dna4")[(i + s) % 3], 20000 + 64 * ((i * 7 + s * 13) % 50), 4000 + 1000 * s + i).tobytes() for i in range(600)]
        sets.append(raws)
    for codec, fn, enc in ((N.LZ4_DECODE, "rcx_lz4_decode_batch", oracle.lz4_encode_block), (N.ZLIB_DECODE, "rcx_zlib_decode_batch", lambda r: zlib.compress(r, 1))):
        packed = []
        for raws in sets:
            blobs = [enc(r) for r in raws]
            base, off, lens = B.pack(blobs)
            total, ooff, ocap = B.layout([len(r) for r in raws])
            packed.append((torch.from_numpy(base).pin_memory(), off, lens, int(total), ooff, ocap, raws))
        n = 600
        outb = torch.zeros(max(p[3] for p in packed) + 64, dtype=torch.uint8).pin_memory()
        for rnd in range(3):
            for inb, off, lens, total, ooff, ocap, raws in packed:
                outb.fill_(0x33 + rnd)
                out_len, in_used, status, flags = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.full(n, -9, np.int32), np.zeros(n, np.uint32)
                p = lambda a: a.ctypes.data
                b = N.Batch(inb.data_ptr(), p(off), p(lens), outb.data_ptr(), p(ooff), p(ocap), p(out_len), p(in_used), p(status), n, N.MEM_HOST)
                args = (ctx._h, C.byref(b)) if codec == N.LZ4_DECODE else (ctx._h, C.byref(b), C.c_void_p(p(flags)))
                assert getattr(L, fn)(*args) == 0, L.rcx_last_error(ctx._h)
                assert not status.any(), (codec, rnd, np.nonzero(status)[0][:5], status[status != 0][:5])
                got = outb.numpy()
                for i in range(n):
                    assert bytes(got[int(ooff[i]): int(ooff[i]) + int(out_len[i])]) == raws[i], (codec, rnd, i)
