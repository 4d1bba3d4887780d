"""Batch API over the C-ABI (include/rcx.h): host-memory batches (lists of bytes) and device-resident
batches (torch uint8 tensors, used by bench.py).  No CPU fallback, no oracle import."""
import ctypes as C

import numpy as np

from . import _native as N
from . import batch as B


class RcxError(RuntimeError):
    pass


class BlockError(Exception):
    """One block failed: mirrors the reference's io::Error for that block."""

    def __init__(self, status, index=0):
        self.status = int(status)
        self.index = index
        msg = N.lib().rcx_status_string(int(status)).decode()
        super().__init__("block %d: status %d (%s)" % (index, status, msg))


class Result:
    __slots__ = ("outputs", "out_len", "in_used", "status", "aux")

    def __init__(self, outputs, out_len, in_used, status, aux):
        self.outputs, self.out_len, self.in_used, self.status, self.aux = outputs, out_len, in_used, status, aux

    def check(self):
        bad = np.nonzero(self.status)[0]
        if bad.size:
            raise BlockError(self.status[bad[0]], int(bad[0]))
        return self


class Context:
    """One rcx_ctx (one HIP device + stream).  Raises if there is no device: there is no CPU path."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        rc = N.lib().rcx_ctx_create(device, C.byref(self._h))
        if rc != N.RC_OK:
            self._h = None
            raise RcxError("rcx_ctx_create failed rc=%d (%s)" % (rc, "no HIP device: this library has no CPU fallback"
                                                                 if rc == N.RC_NO_DEVICE else "HIP error"))

    def close(self):
        if getattr(self, "_h", None):
            N.lib().rcx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:           # interpreter shutdown: the module globals may already be gone
            pass

    def _chk(self, rc):
        if rc != N.RC_OK:
            raise RcxError("rcx call failed rc=%d: %s" % (rc, N.lib().rcx_last_error(self._h).decode()))

    def set_variant(self, codec, variant):
        self._chk(N.lib().rcx_ctx_set_variant(self._h, codec, variant))

    def set_stream(self, stream_ptr):
        self._chk(N.lib().rcx_ctx_set_stream(self._h, C.c_void_p(stream_ptr)))

    # ---------------- host-memory batches ----------------
    def _run_host(self, fn_name, blobs, caps, extra_in=None, extra_out=False, n_out=None, needs_out=True, scalar=None):
        n = len(blobs)
        base, off, lens = B.pack(blobs)
        total, ooff, ocap = B.layout(caps if needs_out else [0] * n)
        out = np.zeros(total, dtype=np.uint8)
        out_len = np.zeros(max(n, 1), np.uint64)
        in_used = np.zeros(max(n, 1), np.uint64)
        status = np.zeros(max(n, 1), np.int32)
        p = lambda a: a.ctypes.data
        b = N.Batch(p(base), p(off), p(lens), p(out), p(ooff), p(ocap), p(out_len), p(in_used), p(status), n, N.MEM_HOST)
        fn = getattr(N.lib(), fn_name)
        aux = None
        if scalar is not None:
            self._chk(fn(self._h, C.byref(b), scalar))
        elif extra_in is not None:
            aux = np.ascontiguousarray(extra_in, dtype=np.uint32)
            self._chk(fn(self._h, C.byref(b), C.c_void_p(p(aux))))
        elif n_out is not None:
            no = np.ascontiguousarray(n_out, dtype=np.uint64)
            self._chk(fn(self._h, C.byref(b), C.c_void_p(p(no))))
        elif extra_out:
            aux = np.zeros(max(n, 1), np.uint32)
            self._chk(fn(self._h, C.byref(b), C.c_void_p(p(aux))))
        else:
            self._chk(fn(self._h, C.byref(b)))
        outs = B.unpack(out, ooff, out_len[:n]) if needs_out else [b""] * n
        return Result(outs, out_len[:n], in_used[:n], status[:n], aux[:n] if aux is not None else None)

    def lz4_decode_blocks(self, blobs, caps):
        return self._run_host("rcx_lz4_decode_batch", blobs, caps)

    def lz4_encode_blocks(self, blobs):
        return self._run_host("rcx_lz4_encode_batch", blobs, [max(int(N.lib().rcx_lz4_compression_bound(len(b))), 1) for b in blobs])

    def inflate(self, blobs, caps):
        return self._run_host("rcx_inflate_batch", blobs, caps, extra_out=True)

    def zlib_decode(self, blobs, caps):
        return self._run_host("rcx_zlib_decode_batch", blobs, caps, extra_out=True)

    def adler32(self, blobs):
        return self._run_host("rcx_adler32_batch", blobs, None, extra_out=True, needs_out=False)

    def crc32(self, blobs):
        """CRC-32 as in the gzip trailer (extension beyond the reference)."""
        return self._run_host("rcx_crc32_batch", blobs, None, extra_out=True, needs_out=False)

    def gzip_decode(self, blobs, caps):
        """One gzip member (RFC 1952) per blob: header, DEFLATE, CRC32 + ISIZE (extension)."""
        return self._run_host("rcx_gzip_decode_batch", blobs, caps, extra_out=True)

    def bwt_forward(self, blobs):
        return self._run_host("rcx_bwt_forward_batch", blobs, [len(b) for b in blobs], extra_out=True)

    def bwt_suffixes(self, blobs):
        """compute_suffixes (src/bwt/mod.rs:136-166): outputs[i] = the block's suffix array, n little-endian u32; extra[i] = origin."""
        return self._run_host("rcx_bwt_suffixes_batch", blobs, [4 * len(b) for b in blobs], extra_out=True)

    def bwt_inversion_table(self, blobs, origins):
        """compute_inversion_table (src/bwt/mod.rs:223-239) of L = blobs[i] with origins[i]: n little-endian u32 entries."""
        return self._run_host("rcx_bwt_inversion_table_batch", blobs, [4 * len(b) for b in blobs], extra_in=origins)

    def bwt_inverse(self, blobs, origins):
        return self._run_host("rcx_bwt_inverse_batch", blobs, [len(b) for b in blobs], extra_in=origins)

    def bwt_inverse_minimal(self, blobs, origins):
        """The reference's decode_minimal (src/bwt/mod.rs:298-315), reproduced as it computes -- not the inverse of bwt_forward in general."""
        return self._run_host("rcx_bwt_inverse_minimal_batch", blobs, [len(b) for b in blobs], extra_in=origins)

    def mtf_encode(self, blobs):
        return self._run_host("rcx_mtf_encode_batch", blobs, [len(b) for b in blobs])

    def mtf_decode(self, blobs):
        return self._run_host("rcx_mtf_decode_batch", blobs, [len(b) for b in blobs])

    def dc_encode(self, blobs):
        return self._run_host("rcx_dc_encode_batch", blobs, [4 * (256 + len(b)) for b in blobs])

    def dc_decode(self, blobs, n_out):
        return self._run_host("rcx_dc_decode_batch", blobs, list(n_out), n_out=n_out)

    def dc_encode_ctx(self, blobs):
        """-> result whose outputs[i] = the words (first 4 * (256 + k) bytes), a gap, then k 8-byte contexts from byte
        4 * (256 + n) on (include/rcx.h); `dc_split_ctx` cuts it up"""
        return self._run_host("rcx_dc_encode_ctx_batch", blobs, [4 * (256 + len(b)) + 8 * len(b) for b in blobs])

    def dc_decode_ctx(self, blobs, n_out):
        return self._run_host("rcx_dc_decode_ctx_batch", blobs, [((n + 7) & ~7) + 8 * max(0, len(b) // 4 - 256) for b, n in zip(blobs, n_out)], n_out=n_out)

    def ari_byte_encode(self, blobs):
        return self._run_host("rcx_ari_byte_encode_batch", blobs, [int(N.lib().rcx_ari_byte_encode_bound(len(b))) for b in blobs])

    def ari_byte_decode(self, blobs, caps):
        return self._run_host("rcx_ari_byte_decode_batch", blobs, caps)

    def ari_binary_encode(self, blobs, rate):
        """bin::Model, 8 decisions per byte (src/entropy/ari/test.rs:22-50)."""
        return self._run_host("rcx_ari_binary_encode_batch", blobs, [int(N.lib().rcx_ari_byte_encode_bound(len(b))) for b in blobs], scalar=rate)

    def ari_binary_decode(self, blobs, rate, nbytes):
        """nbytes[i] = bytes to decode from stream i (the coding has no end marker)."""
        return self._run_host("rcx_ari_binary_decode_batch", blobs, list(nbytes), scalar=rate)

    def ari_proxy_encode(self, blobs):
        """table::SumProxy + bin::SumProxy (src/entropy/ari/test.rs:91-148)."""
        return self._run_host("rcx_ari_proxy_encode_batch", blobs, [int(N.lib().rcx_ari_byte_encode_bound(len(b))) for b in blobs])

    def ari_proxy_decode(self, blobs, nbytes):
        return self._run_host("rcx_ari_proxy_decode_batch", blobs, list(nbytes))

    def ari_apm_encode(self, blobs):
        """apm::Bit through apm::Gate (src/entropy/ari/test.rs:150-182); status E_MALFORMED = the reference panics on this input."""
        return self._run_host("rcx_ari_apm_encode_batch", blobs, [int(N.lib().rcx_ari_byte_encode_bound(len(b))) for b in blobs])

    def ari_apm_decode(self, blobs, nbytes):
        return self._run_host("rcx_ari_apm_decode_batch", blobs, list(nbytes))

    def rle_encode(self, blobs):
        return self._run_host("rcx_rle_encode_batch", blobs, [int(N.lib().rcx_rle_encode_bound(len(b))) for b in blobs])

    def rle_decode(self, blobs, caps):
        return self._run_host("rcx_rle_decode_batch", blobs, caps)

    # ---------------- device-resident batches (torch tensors) ----------------
    def launch_dev(self, codec, db, scratch=None):
        """db: DeviceBatch. Enqueues on the ctx stream and returns (no sync)."""
        sp = scratch.data_ptr() if scratch is not None else None
        sb = scratch.numel() if scratch is not None else 0
        self._chk(N.lib().rcx_launch_dev(self._h, codec, C.byref(db.c), C.c_void_p(sp), sb))

    def scratch_bytes(self, codec, nblocks, max_block):
        return int(N.lib().rcx_scratch_bytes(codec, nblocks, max_block))


class DeviceBatch:
    """Struct-of-arrays batch whose every array is a torch tensor in HBM."""

    def __init__(self, in_base, in_off, in_len, out_base, out_off, out_cap, aux=None):
        import torch
        dev = in_base.device
        n = in_off.numel()
        self.n = n
        self.in_base, self.in_off, self.in_len = in_base, in_off, in_len
        self.out_base, self.out_off, self.out_cap = out_base, out_off, out_cap
        self.out_len = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
        self.in_used = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
        self.status = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
        self.aux = aux if aux is not None else torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        self.c = N.DevBatch(in_base.data_ptr(), in_off.data_ptr(), in_len.data_ptr(), out_base.data_ptr(),
                            out_off.data_ptr(), out_cap.data_ptr(), self.out_len.data_ptr(), self.in_used.data_ptr(),
                            self.status.data_ptr(), self.aux.data_ptr(), n)

    def sub(self, lo, hi):
        """Blocks [lo, hi) of this batch as a batch of its own: the same tensors, sliced (results land in this batch's arrays)."""
        v = object.__new__(DeviceBatch)
        v.n = hi - lo
        v.in_base, v.out_base = self.in_base, self.out_base
        for name in ("in_off", "in_len", "out_off", "out_cap", "out_len", "in_used", "status", "aux"):
            setattr(v, name, getattr(self, name)[lo:hi])
        v.c = N.DevBatch(v.in_base.data_ptr(), v.in_off.data_ptr(), v.in_len.data_ptr(), v.out_base.data_ptr(),
                         v.out_off.data_ptr(), v.out_cap.data_ptr(), v.out_len.data_ptr(), v.in_used.data_ptr(),
                         v.status.data_ptr(), v.aux.data_ptr(), v.n)
        return v

    @staticmethod
    def from_host(blobs_base, off, lens, out_total, out_off, out_cap, device):
        import torch
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(device)
        return DeviceBatch(t(blobs_base, np.uint8), t(off.astype(np.uint64), np.int64), t(lens.astype(np.uint64), np.int64),
                           torch.zeros(out_total + 64, dtype=torch.uint8, device=device),
                           t(out_off.astype(np.uint64), np.int64), t(out_cap.astype(np.uint64), np.int64))
