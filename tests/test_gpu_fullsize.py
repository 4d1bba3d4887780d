"""GPU: BASELINE.json configs 3, 4, 5 at (or near) full size, device resident, checked through
size-independent properties (decode(x) == source, inverse(forward(x)) == x) plus oracle parity on samples."""
import zlib
from multiprocessing import Pool

import numpy as np
import pytest

from rust_compress_amd import _native as N
from rust_compress_amd import synth

pytestmark = pytest.mark.gpu


def _zmember(args):
    i, data = args
    if i % 16 == 15:
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)          # fixed-Huffman blocks too
        return c.compress(data) + c.flush()
    return zlib.compress(data, (1, 6, 9, 0)[i % 4] if i % 64 else 0)


def test_config1_one_mebibyte_deflate_stream(ctx, oracle):
    """BASELINE configs[0] names the reference's own CPU case, one 1 MiB RFC-1951 stream (tests/test_oracle_golden.py decodes it
    with the oracle); the same bytes through rcx_inflate_batch, every kernel variant: output, consumed count and flags == the oracle's."""
    d = synth.gen("text", 1 << 20, 42).tobytes()
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    z = c.compress(d) + c.flush()
    want, used, flags = oracle.inflate(z, cap=len(d))
    assert want == d
    for variant in N.INFLATE_VARIANTS:
        ctx.set_variant(N.INFLATE, variant)
        res = ctx.inflate([z], [len(d)]).check()
        assert res.outputs[0] == d and int(res.in_used[0]) == used == len(z) and int(res.aux[0]) == flags, variant
    ctx.set_variant(N.INFLATE, 0)


def test_config3_zlib_65536_members(ctx, oracle):
    """65536 independent zlib members of 16 KiB (levels 0/1/6/9 + Z_FIXED), Adler-32 verified on the GPU."""
    import torch
    import rust_compress_amd as R
    nb, BLOCK = 65536, 16384
    dev = torch.device("cuda", 0)
    raw_np = np.concatenate([synth.gen_blocks(k, nb // 4, BLOCK, 0x5A11 + j) for j, k in enumerate(("text", "text", "runs", "dna4"))])
    blocks = [raw_np[i * BLOCK:(i + 1) * BLOCK].tobytes() for i in range(nb)]
    with Pool(32) as pool:
        members = pool.map(_zmember, list(enumerate(blocks)), chunksize=512)
    from rust_compress_amd import batch as B
    base, off, lens = B.pack(members)
    ar = np.arange(nb, dtype=np.int64)
    db = R.DeviceBatch.from_host(base, off, lens, nb * BLOCK, (ar * BLOCK).astype(np.uint64), np.full(nb, BLOCK, dtype=np.uint64), dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.launch_dev(N.ZLIB_DECODE, db)
    torch.cuda.synchronize()
    assert int(db.status[:nb].abs().max()) == 0
    assert bool((db.out_len[:nb] == BLOCK).all())
    assert bool((db.in_used[:nb].cpu() == torch.from_numpy(lens.astype(np.int64))).all())
    assert torch.equal(db.out_base[: nb * BLOCK].cpu(), torch.from_numpy(raw_np))
    for i in range(0, nb, 4099):
        out, used, _ = oracle.zlib_decode(members[i], cap=BLOCK)
        assert out == blocks[i] and used == len(members[i])
    ctx.set_stream(0)


def _gzmember(args):
    import gzip
    i, data = args
    if i % 64 == 63:                                                             # a member with FNAME + FEXTRA + FHCRC set (RFC 1952 2.3)
        import struct
        body = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = body.compress(data) + body.flush()
        head = b"\x1f\x8b\x08" + bytes([0x02 | 0x04 | 0x08]) + b"\0\0\0\0\0\xff" + struct.pack("<H", 5) + b"extra" + b"name.txt\0"
        head += struct.pack("<H", zlib.crc32(head) & 0xffff)
        return head + raw + struct.pack("<II", zlib.crc32(data), len(data) & 0xffffffff)
    return gzip.compress(data, compresslevel=(1, 6, 9)[i % 3], mtime=0)


def test_config3_gzip_65536_members(ctx):
    """BASELINE configs[2] as it is worded: 65536 independent GZIP members of 16 KiB.  Header parsed, DEFLATE body decoded,
    CRC-32 and ISIZE verified on the device (status 0 means they matched); bytes == the source; consumed == the member's
    length; a sample against Python's gzip module (the reference has no gzip code: an independent implementation is the checker);
    and a member with a flipped payload bit must come back with the CRC status."""
    import gzip
    import torch
    import rust_compress_amd as R
    nb, BLOCK = 65536, 16384
    dev = torch.device("cuda", 0)
    raw_np = np.concatenate([synth.gen_blocks(k, nb // 4, BLOCK, 0x6A11 + j) for j, k in enumerate(("text", "text", "runs", "dna4"))])
    blocks = [raw_np[i * BLOCK:(i + 1) * BLOCK].tobytes() for i in range(nb)]
    with Pool(32) as pool:
        members = pool.map(_gzmember, list(enumerate(blocks)), chunksize=512)
    bad = 12345                                                                  # one corrupted member among them: flip a bit in the literal-heavy middle
    mb = bytearray(members[bad]); mb[len(mb) // 2] ^= 0x10; members[bad] = bytes(mb)
    from rust_compress_amd import batch as B
    base, off, lens = B.pack(members)
    ar = np.arange(nb, dtype=np.int64)
    db = R.DeviceBatch.from_host(base, off, lens, nb * BLOCK, (ar * BLOCK).astype(np.uint64), np.full(nb, BLOCK, dtype=np.uint64), dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    sc = torch.empty(ctx.scratch_bytes(N.GZIP_DECODE, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
    ctx.launch_dev(N.GZIP_DECODE, db, sc)
    torch.cuda.synchronize()
    st = db.status[:nb].cpu().numpy()
    good = np.ones(nb, dtype=bool); good[bad] = False
    assert not st[good].any()
    assert int(st[bad]) != 0                                                      # (whatever the flipped bit turned into: never status 0 with wrong bytes)
    try:
        py = gzip.decompress(members[bad])
    except Exception:
        py = None
    assert py is None or py != blocks[bad]
    assert bool((db.out_len[:nb].cpu().numpy()[good] == BLOCK).all())
    assert bool((db.in_used[:nb].cpu().numpy()[good] == lens.astype(np.int64)[good]).all())
    out = db.out_base[: nb * BLOCK].cpu().numpy().reshape(nb, BLOCK)
    assert np.array_equal(out[good], raw_np.reshape(nb, BLOCK)[good])
    for i in list(range(0, nb, 4099)) + [63, 127]:
        if i != bad:
            assert gzip.decompress(members[i]) == blocks[i] == out[i].tobytes()
    ctx.set_stream(0)


def test_config4_bwt_1024x256k(ctx, oracle):
    """1024 blocks x 256 KiB: forward then inverse; inverse(forward(x)) == x; samples == oracle (L, origin)."""
    import torch
    import rust_compress_amd as R
    nb, BLOCK = 1024, 262144
    dev = torch.device("cuda", 0)
    raw_np = np.concatenate([synth.gen_blocks("text", nb // 2, BLOCK, 0xB77), synth.gen_blocks("dna4", nb // 2, BLOCK, 0xB78)])
    raw = torch.from_numpy(raw_np).to(dev)
    i64 = lambda a: torch.tensor(a, dtype=torch.int64, device=dev)
    ar = np.arange(nb, dtype=np.int64)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    fw = R.DeviceBatch(raw, i64(ar * BLOCK), i64(np.full(nb, BLOCK)), torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev),
                       i64(ar * BLOCK), i64(np.full(nb, BLOCK)))
    sc = torch.empty(ctx.scratch_bytes(N.BWT_FORWARD, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
    ctx.launch_dev(N.BWT_FORWARD, fw, sc)
    torch.cuda.synchronize()
    del sc
    assert int(fw.status[:nb].abs().max()) == 0
    inv = R.DeviceBatch(fw.out_base, fw.out_off, fw.out_len, torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device=dev),
                        i64(ar * BLOCK), i64(np.full(nb, BLOCK)), aux=fw.aux)
    sc = torch.empty(ctx.scratch_bytes(N.BWT_INVERSE, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
    ctx.launch_dev(N.BWT_INVERSE, inv, sc)
    torch.cuda.synchronize()
    assert int(inv.status[:nb].abs().max()) == 0
    assert torch.equal(inv.out_base[: nb * BLOCK], raw)
    L = fw.out_base.cpu().numpy()
    og = fw.aux.cpu().numpy()
    for i in (0, 511, 512, 1023):
        eL, eo = oracle.bwt_encode(raw_np[i * BLOCK:(i + 1) * BLOCK].tobytes())
        assert L[i * BLOCK:(i + 1) * BLOCK].tobytes() == eL and int(og[i]) == eo
    # decode_minimal (bwt/mod.rs:298-315) over the same 1024 blocks; two of them against the O(n^2) restatement (7 s of CPU each)
    inv.out_base.zero_()
    ctx.launch_dev(N.BWT_INVERSE_MINIMAL, inv, sc)
    torch.cuda.synchronize()
    assert int(inv.status[:nb].abs().max()) == 0 and bool((inv.out_len[:nb] == BLOCK).all())
    mn = inv.out_base.cpu().numpy()
    for i in (3, 1020):
        want = oracle.bwt_decode(L[i * BLOCK:(i + 1) * BLOCK].tobytes(), int(og[i]), minimal=True)
        assert mn[i * BLOCK:(i + 1) * BLOCK].tobytes() == want
    ctx.set_stream(0)


def test_bwt_single_block_of_2_24_bytes_and_more(ctx, oracle):
    """Ranks of a block this long do not fit the 24 bits the 32-bit key path gives them (the sorter switches to 64-bit keys), and the
    inverse transform's table entries no longer carry the byte (index + 1 needs more than 24 bits)."""
    r = synth.gen("text", (1 << 24) + 200_001, 77).tobytes()
    fw = ctx.bwt_forward([r]).check()
    eL, eo = oracle.bwt_encode(r)
    assert fw.outputs[0] == eL and int(fw.aux[0]) == eo
    inv = ctx.bwt_inverse([eL], [eo]).check()
    assert inv.outputs[0] == r


def test_config5_pipeline_roundtrip_and_stage_parity(ctx, oracle):
    """BWT -> DC -> Ari over 96 blocks x 256 KiB (+ a ragged tail): decode(encode(x)) == x, per-stage parity on samples."""
    import struct
    import torch
    from rust_compress_amd import pipeline as P
    BLOCK = 262144
    dev = torch.device("cuda", 0)
    data = synth.gen("text", 96 * BLOCK + 182784, 0xC0FFEE)
    lens = [BLOCK] * 96 + [182784]
    pipe = P.BwtDcAri(ctx, dev)
    raw = torch.from_numpy(data).to(dev)
    comp, coff, clen, praw, st = pipe.encode(raw, lens, keep_stages=True)
    back = pipe.decode(comp, coff, clen, praw, lens)
    assert torch.equal(back, raw)
    assert clen.sum() < 0.45 * data.size                               # it does compress text
    Lall = st["bwt"].out_base.cpu().numpy()
    rec = st["rec"].cpu().numpy()
    compn = comp.cpu().numpy()
    for i in (0, 50, 96):
        src = data[i * BLOCK:i * BLOCK + lens[i]].tobytes()
        eL, eo = oracle.bwt_encode(src)
        assert Lall[i * BLOCK:i * BLOCK + lens[i]].tobytes() == eL
        words = oracle.dc_encode(eL)
        record = struct.pack("<III", lens[i], eo, len(words) - 256) + words.tobytes()
        o = int(st["rec_off"][i])
        assert rec[o:o + len(record)].tobytes() == record
        assert int(praw[i].sum()) == len(record)
        cut = 0
        for s_ in range(praw.shape[1]):                                  # every piece of the record is its own Ari stream
            piece = record[cut:cut + int(praw[i, s_])]; cut += int(praw[i, s_])
            assert compn[int(coff[i, s_]):int(coff[i, s_]) + int(clen[i, s_])].tobytes() == oracle.ari_byte_encode(piece)
    blob = P.encode_stream(ctx, data[: 3 * BLOCK + 17].tobytes())
    assert P.decode_stream(ctx, blob) == data[: 3 * BLOCK + 17].tobytes()
    for parts in (1, 2, 8):                                            # the container records how many pieces a block record has
        blob = P.encode_stream(ctx, data[: BLOCK + 5].tobytes(), parts=parts)
        assert P.decode_stream(ctx, blob) == data[: BLOCK + 5].tobytes()
    assert P.decode_stream(ctx, P.encode_stream(ctx, b"")) == b"" and P.decode_stream(ctx, P.encode_stream(ctx, b"x")) == b"x"
    ctx.set_stream(0)


def test_config5_pipeline_lanes_same_bytes(ctx, oracle):
    """pipeline.PipelineLanes (the block range cut into groups that host threads with a context and a stream of their own
    work through side by side) writes, piece for piece, the bytes the one-lane pipeline writes, and decodes them back; ragged
    block lengths, more groups than lanes, more lanes than blocks."""
    import torch
    from rust_compress_amd import pipeline as P
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    lens = np.concatenate([rng.integers(1, 70000, size=37), [262144, 5, 131072]]).astype(np.int64)
    data = synth.gen("text", int(lens.sum()), 0x1A9E5)
    raw = torch.from_numpy(data).to(dev)
    one = P.BwtDcAri(ctx, dev)
    comp0, coff0, clen0, praw0, _ = one.encode(raw, lens)
    pieces0 = [comp0[int(o):int(o) + int(n)].cpu().numpy().tobytes() for o, n in zip(coff0.reshape(-1), clen0.reshape(-1))]
    for lanes, groups, ll in ((2, 2, lens), (2, 5, lens), (3, 3, lens), (4, 4, lens[:2]), (2, 2, lens[:1])):
        nb = len(ll)
        nbytes = int(ll.sum())
        pipe = P.PipelineLanes(dev, lanes=lanes, groups=groups)
        comp, coff, clen, praw, _ = pipe.encode(raw[:nbytes], ll)
        assert np.array_equal(clen, clen0[:nb]) and np.array_equal(praw, praw0[:nb])
        pieces = [comp[int(o):int(o) + int(n)].cpu().numpy().tobytes() for o, n in zip(coff.reshape(-1), clen.reshape(-1))]
        assert pieces == pieces0[: nb * praw0.shape[1]]
        assert torch.equal(pipe.decode(comp, coff, clen, praw, ll), raw[:nbytes])
        assert torch.equal(pipe.decode(comp0, coff0[:nb], clen0[:nb], praw0[:nb], ll), raw[:nbytes])     # and the one-lane pipeline's buffer
        pipe.close()
    ctx.set_stream(0)


def test_config5_sharded_container_on_the_device(ctx, oracle):
    """BASELINE config 5's sharding on the GPU: the container the device writes for a stream is, byte for byte, the one the
    oracle's stages write on the CPU; split by block ranges (dist.partition) every shard decodes on its own, the decoded ranges
    in rank order are the stream, and the shards encoded separately and joined are the single-device container again."""
    import os, sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _dry_codec as D
    from rust_compress_amd import pipeline as P, dist
    BS = 8192
    data = synth.gen("text", 13 * BS + 4321, 0x5EA).tobytes()
    whole = P.encode_stream(ctx, data, block_size=BS)
    assert whole == D.pipe_encode(data, BS)
    _, _, lens, _, _, _ = P.parse_container(whole)
    for world in (2, 3, 8):
        bounds = dist.partition(lens, world)
        shards = P.split_container(whole, bounds)
        assert b"".join(P.decode_stream(ctx, x) for x in shards) == data
        cuts = np.concatenate([[0], np.cumsum(lens)])[bounds]
        enc = [P.encode_stream(ctx, data[int(cuts[g]):int(cuts[g + 1])], block_size=BS) for g in range(world)]
        assert P.join_containers(enc) == whole
    ctx.set_stream(0)


def test_config5_pipeline_1e9_bytes(ctx, oracle):
    """BASELINE configs[4] at its full size: 10^9 bytes of G-text in 3815 blocks of 256 KiB through BWT -> DC -> Ari and back,
    decode(encode(x)) == x over the whole gigabyte, and three sampled blocks compared with the oracle stage by stage."""
    import struct
    import torch
    from rust_compress_amd import pipeline as P
    BLOCK, total = 262144, 1000000000
    dev = torch.device("cuda", 0)
    lens = [BLOCK] * (total // BLOCK) + [total % BLOCK]
    data = np.concatenate([synth.gen("text", min(BLOCK * 256, total - s), 0xC0 + s) for s in range(0, total, BLOCK * 256)])[:total]
    raw = torch.from_numpy(data).to(dev)
    pipe = P.BwtDcAri(ctx, dev)
    comp, coff, clen, praw, st = pipe.encode(raw, lens, keep_stages=True)
    back = pipe.decode(comp, coff, clen, praw, lens)
    assert torch.equal(back, raw)
    assert int(clen.sum()) < 0.3 * total
    Lall, rec, compn = st["bwt"].out_base, st["rec"], comp
    for i in (7, 1900, len(lens) - 1):                                  # a full block early, one in the middle, the ragged last one
        o0 = i * BLOCK
        src = data[o0:o0 + lens[i]].tobytes()
        eL, eo = oracle.bwt_encode(src)
        assert Lall[o0:o0 + lens[i]].cpu().numpy().tobytes() == eL
        words = oracle.dc_encode(eL)
        record = struct.pack("<III", lens[i], eo, len(words) - 256) + words.tobytes()
        o = int(st["rec_off"][i])
        assert rec[o:o + len(record)].cpu().numpy().tobytes() == record
        cut = 0
        for s_ in range(praw.shape[1]):
            piece = record[cut:cut + int(praw[i, s_])]; cut += int(praw[i, s_])
            a = int(coff[i, s_])
            assert compn[a:a + int(clen[i, s_])].cpu().numpy().tobytes() == oracle.ari_byte_encode(piece)
    ctx.set_stream(0)
