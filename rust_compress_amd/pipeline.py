"""BWT -> DC -> adaptive range coder pipeline (BASELINE config 5, "bzip-like"), device resident.

The reference provides the three STAGES (bwt::encode, dc::encode_simple order, ari::ByteEncoder) but no
wiring and no byte serialisation of the DC output (hard part 8), so this module defines the
container; parity is per stage:
    (L, origin)                      == bwt::encode_simple               (src/bwt/mod.rs:214-219)
    (init[256], distances[k])        == dc::encode_simple::<u32> order   (src/bwt/dc.rs:153-159)
    Ari bytes                        == ari::ByteEncoder over the block record below
Block record (little-endian u32 words):  n, origin, k, init[256], dist[k].  The record is cut into PARTS contiguous
pieces (word aligned, equal length to within a word) and every piece is range coded on its own: the adaptive coder is a
serial chain per stream, so the kernel's time is the length of a stream, not the amount of data.  Measured per 10^9 bytes
(decode / encode of the coder, compression ratio): 1 piece 175 / 122 ms, 4.80; 4 pieces 82 / 63, 4.78; 16 pieces (default)
21 / 16, 4.70; 32 pieces 20 / 15, 4.60 (from 15 000 streams on the library runs one LANE per stream).
Stream container:  b"RCXQ" u32 block_size u32 nblocks u32 parts, then per block u32 n and parts x (u32 raw_len,
u32 comp_len), then the payloads.  Blocks are independent in every stage, so a stream shards across GPUs by block
ranges (dist.partition).
"""
import struct

import numpy as np

from . import _native as N
from .api import DeviceBatch

MAGIC = b"RCXQ"
HDR_WORDS = 3
PARTS = 16


class ContainerError(ValueError):
    """A corrupt or truncated RCXQ container (raised, never asserted: the checks must survive python -O)."""


def _need(cond, msg):
    if not cond:
        raise ContainerError(msg)


class BwtDcAri:
    def __init__(self, ctx, device, parts=PARTS):
        import torch
        self.ctx, self.dev, self.torch, self.S = ctx, device, torch, int(parts)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def _sync(self):
        """Wait for this pipeline's stream only: several pipelines may run side by side on streams of their own (PipelineLanes)."""
        self.torch.cuda.current_stream().synchronize()

    def _i64(self, a):
        return self.torch.as_tensor(np.asarray(a, dtype=np.int64), device=self.dev)

    def _scratch(self, codec, nb, maxn):
        """One scratch buffer per pipeline object, grown on demand: the suffix sort wants ~29 B per suffix of a pass (at
        most 2^27 suffixes), 3.9 GB for 256 KiB blocks, and a fresh hipMalloc of that size per call is slow."""
        need = self.ctx.scratch_bytes(codec, nb, maxn) + 256
        if getattr(self, "_sc", None) is None or self._sc.numel() < need:
            self._sc = None
            self._sc = self.torch.empty(need, dtype=self.torch.uint8, device=self.dev)
        return self._sc

    def encode(self, raw, lens, keep_stages=False, maxn=None, comp=None):
        """raw: uint8 tensor holding the blocks back to back; lens: block lengths (numpy).
        -> (comp tensor, comp_off[nb, S], comp_len[nb, S], raw_part_len[nb, S] numpy, stages dict)
        maxn / comp: the slot geometry's block length and the buffer the coded pieces go to, when this call is one lane's
        share of a larger stream (PipelineLanes); default: this call's own."""
        torch = self.torch
        lens = np.asarray(lens, dtype=np.int64)
        nb = len(lens)
        maxn = int(maxn) if maxn is not None else (int(lens.max()) if nb else 0)
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]) if nb else np.zeros(0, np.int64)
        # 1. BWT
        bw = DeviceBatch(raw, self._i64(off), self._i64(lens), torch.empty(int(lens.sum()) + 64, dtype=torch.uint8, device=self.dev),
                         self._i64(off), self._i64(lens))
        sc = self._scratch(N.BWT_FORWARD, nb, maxn)          # the library sorts at most 2^27 suffixes per pass (3.9 GB of scratch for 256 KiB blocks)
        self.ctx.launch_dev(N.BWT_FORWARD, bw, sc)
        # 2. DC into the record slot, 12 bytes in (n, origin, k go in front)
        slot = (4 * (HDR_WORDS + 256 + maxn) + 63) // 64 * 64
        rec = torch.empty(nb * slot + 64, dtype=torch.uint8, device=self.dev)       # (every byte the range coder reads is written below: the slots are
        roff = np.arange(nb, dtype=np.int64) * slot                                  # worst-case sized, 4 GB for 10^9 bytes, and zeroing them was 1 ms)
        dc = DeviceBatch(bw.out_base, bw.out_off, bw.out_len, rec, self._i64(roff + 4 * HDR_WORDS),
                         self._i64(np.full(nb, slot - 4 * HDR_WORDS)))
        self.ctx.launch_dev(N.DC_ENCODE, dc, self._scratch(N.DC_ENCODE, nb, maxn))       # (37 KiB a block: the lane-per-chunk encoder's chunk states)
        # 3. header words n, origin, k
        k = (dc.out_len[:nb] // 4 - 256).to(torch.int32)
        hdr = torch.stack([self._i64(lens).to(torch.int32), bw.aux[:nb].to(torch.int32), k], dim=1).contiguous()
        rec32 = rec[: nb * slot].view(torch.int32).view(nb, slot // 4)
        rec32[:, :HDR_WORDS] = hdr
        rec_len = dc.out_len[:nb] + 4 * HDR_WORDS
        # 4. range coder, S pieces per record
        S = self.S
        cuts = torch.stack([(rec_len * i // S) & ~3 for i in range(S)] + [rec_len], dim=1)        # [nb, S + 1]
        plen = (cuts[:, 1:] - cuts[:, :-1]).contiguous()
        pin = (self._i64(roff)[:, None] + cuts[:, :S]).contiguous()
        cslot = (2 * (slot // S + 8) + 16 + 63) // 64 * 64
        coff = np.arange(nb * S, dtype=np.int64) * cslot
        if comp is None:
            comp = torch.empty(nb * S * cslot + 64, dtype=torch.uint8, device=self.dev)
        ar = DeviceBatch(rec, pin.view(-1), plen.view(-1).clone(), comp, self._i64(coff), self._i64(np.full(nb * S, cslot)))
        self.ctx.launch_dev(N.ARI_BYTE_ENCODE, ar)
        self._sync()
        assert int(bw.status[:nb].abs().max()) == 0 and int(dc.status[:nb].abs().max()) == 0 and int(ar.status[: nb * S].abs().max()) == 0, \
            "pipeline stage failed"
        stages = {"bwt": bw, "dc": dc, "rec": rec, "rec_off": roff, "rec_len": rec_len, "ari": ar, "cuts": cuts, "nblocks": nb} if keep_stages else None
        return (ar.out_base, coff.reshape(nb, S), ar.out_len[: nb * S].cpu().numpy().astype(np.int64).reshape(nb, S),
                plen.cpu().numpy().astype(np.int64), stages)

    def decode(self, comp, comp_off, comp_len, praw, lens, out=None):
        """comp_off / comp_len / praw: [nb, S] (offset and length of every coded piece, its decoded length)
        -> uint8 tensor with the blocks back to back (written to `out` when given: >= sum(lens) + 64 bytes, or exactly the
        blocks' bytes when more blocks follow in the same buffer)"""
        torch = self.torch
        lens = np.asarray(lens, dtype=np.int64)
        nb = len(lens)
        maxn = int(lens.max()) if nb else 0
        slot = (4 * (HDR_WORDS + 256 + maxn) + 63) // 64 * 64
        roff = np.arange(nb, dtype=np.int64) * slot
        praw = np.asarray(praw, dtype=np.int64).reshape(nb, -1)
        S = praw.shape[1]
        _need(nb == 0 or ((praw >= 0).all() and (praw.sum(axis=1) <= slot).all()), "container piece lengths exceed the block record")
        pstart = np.concatenate([np.zeros((nb, 1), np.int64), np.cumsum(praw, axis=1)[:, :-1]], axis=1)
        ar = DeviceBatch(comp, self._i64(np.asarray(comp_off).reshape(-1)), self._i64(np.asarray(comp_len).reshape(-1)),
                         torch.empty(nb * slot + 64, dtype=torch.uint8, device=self.dev),          # (what is read of a record is what its pieces decode to)
                         self._i64((roff[:, None] + pstart).reshape(-1)), self._i64(praw.reshape(-1)))
        self.ctx.launch_dev(N.ARI_BYTE_DECODE, ar)
        self._sync()
        _need(nb == 0 or int(ar.status[: nb * S].abs().max()) == 0, "ari decode failed")
        _need(bool((ar.out_len[: nb * S].cpu().numpy().astype(np.int64) == praw.reshape(-1)).all()), "container piece length mismatch")
        # the record buffer is not zeroed: a record's header may only be looked at when its first piece really covers it
        _need(nb == 0 or bool((praw[:, 0] >= 4 * HDR_WORDS).all()), "container record: first piece shorter than the record header")
        rec32 = ar.out_base[: nb * slot].view(torch.int32).view(nb, slot // 4)
        hdr = rec32[:, :HDR_WORDS].cpu().numpy()
        n, origin, k = hdr[:, 0].astype(np.int64), hdr[:, 1].astype(np.uint32), hdr[:, 2].astype(np.int64)
        _need(bool((n == lens).all()), "container length mismatch")
        # k (the number of DC distances) comes from the decoded, untrusted record: it must be what the record's length says
        _need(bool(((k >= 0) & (k <= n) & (4 * (HDR_WORDS + 256 + k) == praw.sum(axis=1))).all()), "container record: distance count does not match the record length")
        ooff = np.concatenate([[0], np.cumsum(lens)[:-1]]) if nb else np.zeros(0, np.int64)
        total = int(lens.sum())
        # DC decode through the host-descriptor entry point (it needs n_out)
        import ctypes as C
        L = torch.empty(total + 64, dtype=torch.uint8, device=self.dev)
        u64 = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
        in_off, in_len = u64(roff + 4 * HDR_WORDS), u64(4 * (256 + k))
        out_off, out_cap, n_out = u64(ooff), u64(lens), u64(lens)
        out_len, in_used, status = np.zeros(nb, np.uint64), np.zeros(nb, np.uint64), np.zeros(nb, np.int32)
        p = lambda a: a.ctypes.data
        b = N.Batch(ar.out_base.data_ptr(), p(in_off), p(in_len), L.data_ptr(), p(out_off), p(out_cap), p(out_len), p(in_used),
                    p(status), nb, N.MEM_DEVICE)
        self.ctx._chk(N.lib().rcx_dc_decode_batch(self.ctx._h, C.byref(b), C.c_void_p(p(n_out))))
        _need(not status.any(), "dc decode failed")
        if out is None:
            out = torch.empty(total + 64, dtype=torch.uint8, device=self.dev)
        inv = DeviceBatch(L, self._i64(ooff), self._i64(lens), out,
                          self._i64(ooff), self._i64(lens), aux=torch.as_tensor(origin.astype(np.int32), device=self.dev))
        sc = self._scratch(N.BWT_INVERSE, nb, maxn)
        self.ctx.launch_dev(N.BWT_INVERSE, inv, sc)
        self._sync()
        _need(nb == 0 or int(inv.status[:nb].abs().max()) == 0, "bwt inverse failed")
        return inv.out_base[:total]


class PipelineLanes:
    """The same pipeline with the block range cut into contiguous GROUPS that LANES work through side by side: a lane is a host
    thread with an rcx_ctx and a HIP stream of its own.  Why: every stage of the decode is one wave per block or per sixteen
    streams (3.7 waves a SIMD for 10^9 bytes) and bound by its own dependent chain, and the forward transform waits for the
    host once a round; with two lanes the chains of one stage run under those of another and one lane's round trip hides
    behind the other's kernels.  Measured, 10^9 bytes of text (benchmarks/r4_pipe_lanes.py): 1 lane 0.1226 s encode /
    0.0501 s decode, 2 lanes 0.1173 / 0.0443, 3 lanes 0.1186 / 0.0446, 4 lanes 0.137 / 0.082 (the host threads' round trips
    queue up behind each other).  Same bytes as one lane: blocks are independent in every stage and the slot geometry is the
    whole stream's."""

    def __init__(self, device, parts=PARTS, lanes=2, groups=None):
        import torch
        from concurrent.futures import ThreadPoolExecutor
        from .api import Context
        self.torch, self.dev, self.S, self.lanes = torch, device, int(parts), int(lanes)
        self.groups = int(groups) if groups else self.lanes
        idx = device.index if device.index is not None else torch.cuda.current_device()
        # Lanes must not share a hardware queue: HIP hands its few queues (four by default) to the streams of a process in turn, and two
        # lanes that land on one queue run one after the other (measured: a process in five decodes 10^9 bytes in 0.064 s instead of
        # 0.044 -- slower than one lane).  Streams of different priority never share a queue, so the lanes alternate between the two
        # levels every device has; a third and fourth lane take their chances within a level.
        self.streams = [torch.cuda.Stream(device=device, priority=(-1 if l % 2 else 0)) for l in range(self.lanes)]
        self.pipes = []
        for st in self.streams:
            with torch.cuda.stream(st):
                self.pipes.append(BwtDcAri(Context(idx), device, parts))
        self.pool = ThreadPoolExecutor(self.lanes)

    def _ranges(self, lens):
        """The groups' block ranges [a, b): contiguous, balanced by bytes, none empty."""
        from .dist import partition
        bounds = partition(lens, max(1, min(self.groups, len(lens))))
        return [(int(a), int(b)) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]

    def _run(self, jobs):
        """jobs[g](pipe) for every group; lane l takes groups l, l + lanes, ... in order, on its own stream."""
        torch = self.torch
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())                   # what the caller enqueued (the input) is in place first
        res = [None] * len(jobs)

        def lane(l):
            torch.cuda.set_device(self.dev)                              # (the current device is a property of the thread)
            with torch.cuda.stream(self.streams[l]):
                self.streams[l].wait_event(ready)
                for g in range(l, len(jobs), self.lanes):
                    res[g] = jobs[g](self.pipes[l])
                self.streams[l].synchronize()
        # every lane is waited for before an exception of one of them travels on: the lanes share the caller's buffers (safe
        # across streams only because a lane synchronises its stream before it returns), and none may still be writing then
        futs = [self.pool.submit(lane, l) for l in range(self.lanes)]
        errs = []
        for f in futs:
            try:
                f.result()
            except BaseException as e:          # noqa: BLE001 (re-raised below)
                errs.append(e)
        if errs:
            raise errs[0]
        return res

    def encode(self, raw, lens, keep_stages=False):
        """-> what BwtDcAri.encode returns; with keep_stages the stages of the FIRST group (blocks [0, stages["nblocks"]))."""
        torch = self.torch
        lens = np.asarray(lens, dtype=np.int64)
        nb, S = len(lens), self.S
        if nb == 0:
            return self.pipes[0].encode(raw, lens, keep_stages)
        maxn = int(lens.max())
        slot = (4 * (HDR_WORDS + 256 + maxn) + 63) // 64 * 64
        cslot = (2 * (slot // S + 8) + 16 + 63) // 64 * 64
        comp = torch.empty(nb * S * cslot + 64, dtype=torch.uint8, device=self.dev)
        offs = np.concatenate([[0], np.cumsum(lens)])

        def job(a, b):
            return lambda pipe: pipe.encode(raw[int(offs[a]):int(offs[b])], lens[a:b], keep_stages and a == 0, maxn=maxn,
                                            comp=comp[a * S * cslot:])
        res = self._run([job(a, b) for a, b in self._ranges(lens)])
        coff = np.arange(nb * S, dtype=np.int64).reshape(nb, S) * cslot
        return comp, coff, np.concatenate([r[2] for r in res]), np.concatenate([r[3] for r in res]), res[0][4]

    def decode(self, comp, comp_off, comp_len, praw, lens):
        torch = self.torch
        lens = np.asarray(lens, dtype=np.int64)
        nb = len(lens)
        if nb == 0:
            return self.pipes[0].decode(comp, comp_off, comp_len, praw, lens)
        comp_off, comp_len, praw = (np.asarray(x).reshape(nb, -1) for x in (comp_off, comp_len, praw))
        total = int(lens.sum())
        out = torch.empty(total + 64, dtype=torch.uint8, device=self.dev)
        offs = np.concatenate([[0], np.cumsum(lens)])

        def job(a, b):
            return lambda pipe: pipe.decode(comp, comp_off[a:b], comp_len[a:b], praw[a:b], lens[a:b], out=out[int(offs[a]):])
        self._run([job(a, b) for a, b in self._ranges(lens)])
        return out[:total]

    def close(self):
        self.pool.shutdown()
        for p in self.pipes:
            p.ctx.close()


# ---------------------------------------------------------------------------------------------- container (host side, no device)
def parse_container(blob):
    """-> (block_size, parts, lens[nb], praw[nb, parts], clen[nb, parts], payload offset).  Raises ContainerError."""
    _need(len(blob) >= 16 and bytes(blob[:4]) == MAGIC, "not an RCXQ container")
    block_size, nb, parts = struct.unpack_from("<III", blob, 4)
    _need(parts >= 1 and block_size >= 1, "container header: bad block size / piece count")
    _need(16 + nb * (4 + 8 * parts) <= len(blob), "container descriptor table is truncated")
    tab = np.frombuffer(blob, dtype="<u4", count=nb * (1 + 2 * parts), offset=16).reshape(nb, 1 + 2 * parts).astype(np.int64)
    lens = tab[:, 0]
    _need(bool((lens <= block_size).all()), "container block longer than the block size")
    praw, clen = tab[:, 1::2], tab[:, 2::2]
    p = 16 + nb * (4 + 8 * parts)
    _need(p + int(clen.sum()) <= len(blob), "container payload is truncated")
    return block_size, parts, lens, praw, clen, p


def build_container(block_size, parts, lens, praw, clen, payload):
    """The container of blocks with the given descriptors; `payload` = the coded pieces back to back (bytes)."""
    lens = np.asarray(lens, dtype=np.int64)
    nb = len(lens)
    tab = np.zeros((nb, 1 + 2 * parts), dtype="<u4")
    if nb:
        tab[:, 0] = lens
        tab[:, 1::2] = np.asarray(praw).reshape(nb, parts)
        tab[:, 2::2] = np.asarray(clen).reshape(nb, parts)
    return b"".join([MAGIC, struct.pack("<III", block_size, nb, parts), tab.tobytes(), bytes(payload)])


def split_container(blob, bounds):
    """One container per block range [bounds[g], bounds[g + 1]) -- each a valid container on its own: blocks are independent
    in every stage (bwt/mod.rs:373-401), so a stream shards across GPUs by block ranges (dist.partition over the decoded sizes)."""
    block_size, parts, lens, praw, clen, p = parse_container(blob)
    ends = p + np.concatenate([[0], np.cumsum(clen.sum(axis=1))]).astype(np.int64)
    out = []
    for g in range(len(bounds) - 1):
        a, b = int(bounds[g]), int(bounds[g + 1])
        out.append(build_container(block_size, parts, lens[a:b], praw[a:b], clen[a:b], blob[int(ends[a]):int(ends[b])]))
    return out


def join_containers(blobs):
    """Inverse of split_container: the shards' containers, in rank order, as one stream (byte for byte what one device writes)."""
    metas = [parse_container(b) for b in blobs]
    _need(len(metas) > 0, "nothing to join")
    bs, parts = metas[0][0], metas[0][1]
    _need(all(m[0] == bs and m[1] == parts for m in metas), "shards disagree on block size / piece count")
    lens = np.concatenate([m[2] for m in metas])
    praw = np.concatenate([m[3].reshape(-1, parts) for m in metas])
    clen = np.concatenate([m[4].reshape(-1, parts) for m in metas])
    payload = b"".join(bytes(b[m[5]: m[5] + int(m[4].sum())]) for b, m in zip(blobs, metas))
    return build_container(bs, parts, lens, praw, clen, payload)


def encode_stream(ctx, data, block_size=256 * 1024, device=None, parts=PARTS):
    """bytes -> container bytes"""
    import torch
    dev = device or torch.device("cuda", torch.cuda.current_device())
    data = bytes(data)
    lens = [min(block_size, len(data) - i) for i in range(0, len(data), block_size)]
    if not lens:
        return build_container(block_size, parts, [], [], [], b"")
    raw = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    comp, coff, clen, praw, _ = BwtDcAri(ctx, dev, parts).encode(raw, lens)
    comp = comp.cpu().numpy()
    payload = b"".join(comp[int(o):int(o) + int(cl)].tobytes() for o, cl in zip(coff.reshape(-1), clen.reshape(-1)))
    return build_container(block_size, parts, lens, praw, clen, payload)


def decode_stream(ctx, blob, device=None):
    import torch
    dev = device or torch.device("cuda", torch.cuda.current_device())
    block_size, parts, lens, praw, clen, p = parse_container(blob)
    nb = len(lens)
    if not nb:
        return b""
    clen = clen.reshape(-1)
    coff = np.concatenate([[0], np.cumsum(clen)[:-1]]).astype(np.int64)
    comp = torch.frombuffer(bytearray(blob[p:p + int(clen.sum())] + b"\0" * 64), dtype=torch.uint8).to(dev)
    out = BwtDcAri(ctx, dev, parts).decode(comp, coff.reshape(nb, parts), clen.reshape(nb, parts), praw.reshape(nb, parts), lens)
    return out.cpu().numpy().tobytes()
