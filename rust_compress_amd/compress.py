"""Host-side mirror of the reference crate's per-algorithm Reader/Writer surface (`compress::*`), over the
batch C-ABI.  Same names, argument meaning and error behaviour as the Rust types so the parity tests read
like the reference's own tests:

    compress::lz4::{Decoder, Encoder, decode_block, encode_block, compression_bound}   src/lz4.rs
    compress::flate::Decoder, compress::zlib::Decoder                                 src/flate.rs, src/zlib.rs
    compress::bwt::{Encoder, Decoder, encode_simple, decode_simple}                   src/bwt/mod.rs
    compress::bwt::mtf::{Encoder, Decoder}, compress::bwt::dc::{encode_simple, decode_simple}
    compress::entropy::ari::{ByteEncoder, ByteDecoder}                                src/entropy/ari/table.rs
    compress::entropy::ari::{RangeEncoder, Model, Encoder, Decoder, table, bin, apm}  src/entropy/ari/*.rs (per symbol: host code, ari_symbol.py)
    compress::rle::{Encoder, Decoder}                                                 src/rle.rs
    compress::Adler32                                                                 src/checksum/adler.rs

A `Decoder(r)` wraps any object with `.read(n) -> bytes` (io.BytesIO, a file, ...) and itself has
`.read(n)` returning at most n bytes (chunk-size independent, like std::io::Read); an `Encoder(w)` wraps
any object with `.write(bytes)` and has `.write(buf)` / `.finish()`.  Framing (LZ4 frame, BWT stream,
zlib header/trailer position) is parsed on the host; every block kernel call is ONE batch FFI call.
io::Error kinds map to Python exceptions: InvalidInput -> InvalidInput, "unexpected end of file" ->
UnexpectedEof, a reference panic -> Malformed.
"""
import io
import struct

from . import _native as N
from .api import BlockError, Context

_ctx = None


def context():
    global _ctx
    if _ctx is None:
        _ctx = Context()
    return _ctx


def set_context(ctx):
    global _ctx
    _ctx = ctx


class CompressError(IOError):
    def __init__(self, status, msg=None):
        self.status = int(status)
        super().__init__(msg if msg is not None else N.lib().rcx_status_string(int(status)).decode())


class InvalidInput(CompressError):       # io::ErrorKind::InvalidInput
    pass


class UnexpectedEof(CompressError):      # io::ErrorKind::Other("unexpected end of file") / UnexpectedEof
    pass


class Malformed(CompressError):          # the reference panics here
    pass


def _raise(status):
    if status == 0:
        return
    if status == 1:
        raise UnexpectedEof(status)
    if status in (2, 3):
        raise Malformed(status)
    raise InvalidInput(status)


def _check(res):
    for s in res.status:
        _raise(int(s))
    return res


def _read_all(r):
    chunks = []
    while True:
        c = r.read(1 << 20)
        if not c:
            break
        chunks.append(c)
    return b"".join(chunks)


class TailReader:
    """A reader that takes bytes back.  The batch decoders have to read ahead (a stream's end is only known once it is
    decoded); the reference's decoders stop reading exactly at the end of their stream (flate.rs:250-260 reads byte by
    byte, ari/mod.rs:289-292 `finish`), and its tests rely on the reader being left there (ari/test.rs:52-89: two streams
    back to back).  Every Decoder here wraps its reader in a TailReader and hands the bytes behind its stream back to
    it, so `decoder.r` / `finish()` / `unwrap()` is a reader positioned exactly after the stream."""

    def __init__(self, r):
        self.inner = r
        self._tail = b""

    def read(self, n=-1):
        if n is None or n < 0:
            out, self._tail = self._tail + _read_all(self.inner), b""
            return out
        if self._tail:
            out, self._tail = self._tail[:n], self._tail[n:]
            if len(out) < n:                           # like BufRead: top up from the inner reader
                out += self.inner.read(n - len(out))
            return out
        return self.inner.read(n)

    def unread(self, data):
        if data:
            self._tail = bytes(data) + self._tail


MAX_BLOCK = 0xFFFFFFFF          # the kernels index a block with 32 bits (run_batch rejects a larger slot with RCX_RC_BAD_ARG)


def _grow_caps(call, first_cap, limit=MAX_BLOCK):
    """Run `call(cap)` with growing output slots until the block fits.  A kernel stops at a full slot, so a failed attempt
    costs its (8x smaller) slot: all the retries together cost a seventh of the decode that fits.  The last attempt is the
    largest slot a block may have (2^32 - 1 bytes); past that the answer is the block's own RCX_E_OUTPUT_TOO_SMALL status,
    not a batch-level error."""
    cap = min(first_cap, limit)
    while True:
        res = call(cap)
        if res.status[0] == N.E_OUTPUT_TOO_SMALL and cap < limit:
            cap = min(cap * 8, limit)
            continue
        return res


class _BufferedDecoder:
    """Serves self._out through read(n); subclasses fill it in _decode_all() on first use and set self.consumed when
    their stream ends before the input does (the rest goes back to the reader)."""

    def __init__(self, r):
        self.r = r if isinstance(r, TailReader) else TailReader(r)
        self._out = None
        self._pos = 0

    def _ensure(self):
        if self._out is None:
            self.consumed = None
            self._raw = self.r.read(-1)
            self._out = self._decode_all(self._raw)
            self._pos = 0
            if self.consumed is not None:
                self.r.unread(self._raw[self.consumed:])
                self._raw = self._raw[:self.consumed]

    def read(self, n=-1):
        self._ensure()
        if n is None or n < 0:
            n = len(self._out) - self._pos
        chunk = self._out[self._pos:self._pos + n]
        self._pos += len(chunk)
        return chunk

    def read_to_end(self):
        return self.read(-1)

    def eof(self):
        self._ensure()
        return self._pos == len(self._out)

    def reset(self):
        self._out = None
        self._pos = 0

    def finish(self):
        """-> the reader, positioned exactly after this decoder's stream"""
        self._ensure()
        return self.r

    unwrap = finish


def _decode_many(cls, readers, call, first_cap):
    """Several streams of one kind through ONE batch call (and a second one for the slots that were too small): what a caller with many
    `Decoder`s uses instead of reading them one by one -- a stream alone on the GPU takes longer than on one host thread, a batch of
    eight or more does not ("how large a batch has to be").  `readers`: objects with .read(); -> the decoders, each
    already decoded (read() / finish() serve from memory), in order.  The first stream that fails raises what its Decoder would raise."""
    decs = [cls(r) for r in readers]
    raws = []
    for d in decs:
        d.consumed = None
        d._raw = d.r.read(-1)
        raws.append(d._raw)
    if not decs:
        return decs
    caps = [first_cap(x) for x in raws]
    res = call(raws, caps)
    outs, status, used, aux = list(res.outputs), [int(x) for x in res.status], [int(x) for x in res.in_used], list(res.aux) if res.aux is not None else [0] * len(raws)
    redo = [i for i, st in enumerate(status) if st == N.E_OUTPUT_TOO_SMALL]
    cap = max(caps) if caps else 0
    while redo and cap < MAX_BLOCK:                    # the slots that did not fit, eight times larger, together
        cap = min(cap * 8, MAX_BLOCK)
        r2 = call([raws[i] for i in redo], [cap] * len(redo))
        for j, i in enumerate(redo):
            outs[i], status[i], used[i] = r2.outputs[j], int(r2.status[j]), int(r2.in_used[j])
            if r2.aux is not None:
                aux[i] = r2.aux[j]
        redo = [i for i in redo if status[i] == N.E_OUTPUT_TOO_SMALL]
    for d, o, st, u, a in zip(decs, outs, status, used, aux):
        _raise(st)
        d._out, d._pos, d.consumed, d.flags = o, 0, u, int(a)
        d.r.unread(d._raw[u:])
        d._raw = d._raw[:u]
    return decs


# ------------------------------------------------------------------------------------------------ lz4
class lz4:
    MAGIC = 0x184D2204

    @staticmethod
    def compression_bound(size):                      # lz4.rs:175-181 -> None | int
        v = int(N.lib().rcx_lz4_compression_bound(size))
        return v if v else None

    @staticmethod
    def decode_block(input, output, max_output=None):  # lz4.rs:602-611: appends to `output`, returns the count
        if max_output is not None:
            res = _check(context().lz4_decode_blocks([bytes(input)], [max_output]))
        else:                                          # the reference grows the Vec as the block decodes (:148-161)
            res = _check(_grow_caps(lambda cap: context().lz4_decode_blocks([bytes(input)], [cap]), max(1 << 16, 8 * len(input))))
        output += res.outputs[0]
        return len(res.outputs[0])

    @staticmethod
    def encode_block(input, output):                  # lz4.rs:616-627
        res = context().lz4_encode_blocks([bytes(input)])
        if res.status[0] == 42:                       # compression_bound() == None -> encode returns 0
            return 0
        _check(res)
        output += res.outputs[0]
        return len(res.outputs[0])

    class Decoder(_BufferedDecoder):                  # lz4.rs:316-500 (frame reader)
        MAX_SIZES = [0, 0, 0, 0, 64 << 10, 256 << 10, 1 << 20, 4 << 20]

        def _parse(self, data):
            """host framing: -> [(stored?, payload)], self.consumed, self.max_block_size"""
            p, n = 0, len(data)
            if n - p < 4:
                raise UnexpectedEof(1)
            if struct.unpack_from("<I", data, p)[0] != lz4.MAGIC:            # :365-367
                raise InvalidInput(40, "")
            p += 4
            bits = data[p:p + 2] + b"\0\0"                                    # :369-372 short read tolerated
            p = min(p + 2, n)
            flg, bd = bits[0], bits[1]
            if (flg >> 6) != 1:                                               # :375-377
                raise InvalidInput(41, "")
            blk_checksum, stream_size, preset = bool(flg & 0x10), bool(flg & 0x08), bool(flg & 0x01)
            self.max_block_size = self.MAX_SIZES[(bd >> 4) & 7]
            if stream_size:
                if n - p < 8:
                    raise UnexpectedEof(1)
                p += 8
            if preset:                                                         # :407 assert!
                raise Malformed(3, "preset dictionaries not supported yet")
            if p >= n:
                raise UnexpectedEof(1)
            p += 1                                                             # header checksum, ignored :417
            parts = []                                                         # (kind, payload)
            while True:
                if n - p < 4:
                    raise UnexpectedEof(1)
                v = struct.unpack_from("<I", data, p)[0]
                p += 4
                if v == 0:
                    break
                amt = v & 0x7FFFFFFF
                if n - p < amt:
                    raise UnexpectedEof(1)
                parts.append((bool(v & 0x80000000), data[p:p + amt]))
                p += amt
                if blk_checksum:
                    if n - p < 4:
                        raise UnexpectedEof(1)
                    p += 4
            self.consumed = p                                                  # content checksum is never read
            return parts

        @staticmethod
        def _decode_blocks(comp, caps):
            """every compressed block of one or MANY frames: ONE batch call; a conforming frame's blocks decode to at most max_block_size
            bytes, only the blocks that do not fit get a second, larger slot (the reference would simply grow its Vec, :148-161)"""
            if not comp:
                return []
            res = context().lz4_decode_blocks(comp, caps)
            outl = list(res.outputs)
            redo = [i for i, st in enumerate(res.status) if st == N.E_OUTPUT_TOO_SMALL]
            for i in redo:
                outl[i] = _check(_grow_caps(lambda cap, d=comp[i]: context().lz4_decode_blocks([d], [cap]), 8 * caps[i])).outputs[0]
            for i, st in enumerate(res.status):
                if i not in redo:
                    _raise(int(st))
            return outl

        def _decode_all(self, data):
            parts = self._parse(data)
            comp = [d for stored, d in parts if not stored]
            outs = iter(self._decode_blocks(comp, [max(self.max_block_size, 1 << 16)] * len(comp)))
            return b"".join(d if stored else next(outs) for stored, d in parts)

    @staticmethod
    def decode_many(readers):
        """-> [lz4.Decoder]: the frames of all readers parsed on the host, EVERY compressed block of EVERY frame in one batch call
        (one 64 KiB block alone on the GPU takes six times what one host thread needs; eight or more together take less: INTEGRATION.md)."""
        decs = [lz4.Decoder(r) for r in readers]
        frames, comp, caps = [], [], []
        for d in decs:
            d.consumed = None
            d._raw = d.r.read(-1)
            parts = d._parse(d._raw)
            frames.append(parts)
            for stored, blk in parts:
                if not stored:
                    comp.append(blk); caps.append(max(d.max_block_size, 1 << 16))
        outs = iter(lz4.Decoder._decode_blocks(comp, caps))
        for d, parts in zip(decs, frames):
            d._out = b"".join(blk if stored else next(outs) for stored, blk in parts)
            d._pos = 0
            d.r.unread(d._raw[d.consumed:])
            d._raw = d._raw[:d.consumed]
        return decs

    class Encoder:                                     # lz4.rs:505-597: stored blocks only (compress() is false)
        def __init__(self, w):
            self.w = w
            self.buf = bytearray()
            self.wrote_header = False
            self.limit = 256 * 1024

        def _encode_block(self):
            self.w.write(struct.pack("<I", len(self.buf) | 0x80000000))
            self.w.write(bytes(self.buf))
            self.buf.clear()

        def write(self, buf):
            if not self.wrote_header:
                self.w.write(struct.pack("<I", lz4.MAGIC) + bytes([0b01100000, 0b01010000, 0]))
                self.wrote_header = True
            buf = memoryview(bytes(buf))
            while len(buf):
                amt = min(self.limit - len(self.buf), len(buf))
                self.buf += buf[:amt]
                if len(self.buf) == self.limit:
                    self._encode_block()
                buf = buf[amt:]
            return 0                                   # the reference returns Ok(buf.len()) of the EMPTIED slice (:588)

        def flush(self):
            if self.buf:
                self._encode_block()

        def finish(self):
            self.flush()
            self.w.write(b"\0" * 8)
            return self.w


# ------------------------------------------------------------------------------------------------ flate / zlib
class flate:
    class Decoder(_BufferedDecoder):                   # flate.rs:164-488; batch semantics: decoded to BFINAL
        def _decode_all(self, data):
            res = _check(_grow_caps(lambda cap: context().inflate([data], [cap]), max(1 << 16, 4 * len(data))))
            self.consumed = int(res.in_used[0])
            self.flags = int(res.aux[0])
            return res.outputs[0]

    @staticmethod
    def decode_many(readers):
        """-> [flate.Decoder], every stream decoded by ONE batch call (the reference decodes one deflate block per read(), flate.rs:468-488:
        one stream is one wave's work here -- hand over many)."""
        return _decode_many(flate.Decoder, readers, lambda raws, caps: context().inflate(raws, caps), lambda x: max(1 << 16, 4 * len(x)))


class zlib:
    class Decoder(_BufferedDecoder):                   # zlib.rs:32-127
        def _decode_all(self, data):
            res = _check(_grow_caps(lambda cap: context().zlib_decode([data], [cap]), max(1 << 16, 4 * len(data))))
            self.consumed = int(res.in_used[0])
            return res.outputs[0]

    @staticmethod
    def decode_many(readers):
        """-> [zlib.Decoder], every member decoded (and its Adler-32 checked) by ONE batch call"""
        return _decode_many(zlib.Decoder, readers, lambda raws, caps: context().zlib_decode(raws, caps), lambda x: max(1 << 16, 4 * len(x)))


class gzip:
    """RFC 1952 members on top of the same DEFLATE kernel.  EXTENSION: the reference has zlib framing only
    (SURVEY.md 8f rank 3); the shape follows zlib::Decoder.  A stream of concatenated members decodes member by member
    (the ISIZE trailer sizes each output exactly; CRC-32 is verified on the device)."""

    class Decoder(_BufferedDecoder):
        def _decode_all(self, data):
            out, pos = [], 0
            self.members = 0
            view = memoryview(data)
            window = 1 << 20                           # bytes handed to one call: a member's end is only known once it is decoded
            while pos < len(data):
                # ISIZE (the last four bytes, if the input is exactly one member) sizes the slot; otherwise it grows
                hint = int.from_bytes(data[-4:], "little") if len(data) >= 18 else 0
                first = hint if 0 < hint <= 1032 * len(data) and not out else max(1 << 16, 4 * min(window, len(data) - pos))
                while True:
                    member = bytes(view[pos:pos + window])
                    res = _grow_caps(lambda cap: context().gzip_decode([member], [cap]), max(first, 1))
                    short = pos + window < len(data)
                    if short and res.status[0] in (N.E_EOF, 17, N.E_MALFORMED, N.E_GZIP_CRC, N.E_GZIP_ISIZE):
                        window *= 4                    # the window cut the member short: take more input
                        continue
                    break
                _check(res)
                out.append(res.outputs[0])
                pos += int(res.in_used[0])
                self.members += 1
            self.consumed = pos
            return b"".join(out)


class Crc32:                                           # extension, mirrors Adler32 below
    def __init__(self):
        self.reset()

    def feed(self, buf):
        self._data += bytes(buf)

    def result(self):
        return int(context().crc32([bytes(self._data)]).aux[0])

    def reset(self):
        self._data = bytearray()


class Adler32:                                         # checksum/adler.rs:22-51
    def __init__(self):
        self.reset()

    def feed(self, buf):
        self._data += bytes(buf)

    def result(self):
        return int(context().adler32([bytes(self._data)]).aux[0])

    def reset(self):
        self._data = bytearray()


# ------------------------------------------------------------------------------------------------ bwt
class _mtf:
    class Encoder:                                     # mtf.rs:95-129
        def __init__(self, w):
            self.w, self._buf = w, bytearray()

        def write(self, buf):
            self._buf += bytes(buf)
            return len(buf)

        def finish(self):
            self.w.write(_check(context().mtf_encode([bytes(self._buf)])).outputs[0])
            return self.w

    class Decoder(_BufferedDecoder):                   # mtf.rs:133-169
        def _decode_all(self, data):
            return _check(context().mtf_decode([data])).outputs[0]


class _dc:
    @staticmethod
    def encode_simple(input):                          # dc.rs:153-159 -> list of ints (256 init + distances)
        out = _check(context().dc_encode([bytes(input)])).outputs[0]
        return list(struct.unpack("<%dI" % (len(out) // 4), out))

    @staticmethod
    def decode_simple(n, distances):                   # dc.rs:236-252
        blob = struct.pack("<%dI" % len(distances), *distances)
        return _check(context().dc_decode([blob], [n])).outputs[0]

    @staticmethod
    def encode(input):
        """dc.rs:110-149 in batch-backed form -> (init[256], [(distance, (symbol, last_rank, distance_limit)), ...]): the
        initial positions and the (distance, Context) pairs the reference's EncodeIterator yields (:88-103), from ONE kernel call"""
        input = bytes(input)
        n = len(input)
        out = _check(context().dc_encode_ctx([input])).outputs[0]
        k = (len(out) - 4 * (256 + n)) // 8
        words = struct.unpack("<%dI" % (256 + k), out[: 4 * (256 + k)])
        cw = struct.unpack("<%dI" % (2 * k), out[4 * (256 + n): 4 * (256 + n) + 8 * k])
        return list(words[:256]), [(words[256 + j], (cw[2 * j] & 255, (cw[2 * j] >> 8) & 255, cw[2 * j + 1])) for j in range(k)]

    @staticmethod
    def decode(init, distances, n):
        """dc.rs:162-233 in batch-backed form -> (bytes, contexts): the Context handed to the distance callback before each
        distance is read (:208), in call order"""
        blob = struct.pack("<%dI" % (256 + len(distances)), *(list(init) + list(distances)))
        out = _check(context().dc_decode_ctx([blob], [n])).outputs[0]
        co = (n + 7) & ~7
        cw = struct.unpack("<%dI" % ((len(out) - co) // 4), out[co:])
        return out[:n], [(cw[2 * j] & 255, (cw[2 * j] >> 8) & 255, cw[2 * j + 1]) for j in range(len(cw) // 2)]


class bwt:
    mtf = _mtf
    dc = _dc

    @staticmethod
    def encode_simple(input):                          # bwt/mod.rs:214-219 -> (L, origin)
        res = _check(context().bwt_forward([bytes(input)]))
        return res.outputs[0], int(res.aux[0])

    @staticmethod
    def compute_suffixes(input, suf_array=None):       # bwt/mod.rs:136-166
        """The sorted suffix array of `input`.  The reference fills a caller-provided slice (`suf_array: &mut [SUF]`): pass a
        mutable sequence of len(input) to have it filled in place; the list of indices is returned either way."""
        input = bytes(input)
        if suf_array is not None and len(suf_array) < len(input):
            raise IndexError("suf_array is shorter than the input")          # suf_array[p] index panic, :146
        out = _check(context().bwt_suffixes([input])).outputs[0]
        sa = list(struct.unpack("<%dI" % len(input), out))
        if suf_array is not None:
            suf_array[:len(sa)] = sa
        return sa

    @staticmethod
    def compute_inversion_table(input, origin, table=None):   # bwt/mod.rs:223-239
        """The inversion jump table of the transformed block `input` with `origin`; fills `table` (len(input) entries, the
        reference asserts the lengths are equal, :224) when one is given and returns the list."""
        input = bytes(input)
        if table is not None and len(table) != len(input):
            raise AssertionError("input.len() != table.len()")               # assert_eq!, :224
        if origin >= len(input):
            raise IndexError("origin out of range")                          # input[origin], :230
        out = _check(context().bwt_inversion_table([input], [origin])).outputs[0]
        t = list(struct.unpack("<%dI" % len(input), out))
        if table is not None:
            table[:] = t
        return t

    @staticmethod
    def decode_simple(input, origin):                  # bwt/mod.rs:291-294
        if len(input) == 0:
            return b""
        return _check(context().bwt_inverse([bytes(input)], [origin])).outputs[0]

    class Encoder:                                     # bwt/mod.rs:437-518
        def __init__(self, w, block_size):
            self.w, self.block_size, self._buf, self.wrote_header = w, block_size, bytearray(), False

        def write(self, buf):
            if not self.wrote_header:
                self.w.write(struct.pack("<I", self.block_size & 0xFFFFFFFF))
                self.wrote_header = True
            self._buf += bytes(buf)
            return 0                                   # same Ok(0) quirk as lz4 (:507)

        def flush(self):
            """bwt/mod.rs:511-518: what is buffered -- whole blocks and the partial one -- is encoded (ONE batch call), then the
            writer is flushed: a caller that flushes mid-stream gets the block boundary the reference gives it."""
            data, bs = bytes(self._buf), max(self.block_size, 1)
            blocks = [data[i:i + bs] for i in range(0, len(data), bs)]
            if blocks:
                res = _check(context().bwt_forward(blocks))
                for blk, L, origin in zip(blocks, res.outputs, res.aux):
                    self.w.write(struct.pack("<I", len(blk)) + L + struct.pack("<I", int(origin)))
            self._buf.clear()
            if hasattr(self.w, "flush"):
                self.w.flush()

        def finish(self):                              # :485-489: flush, then the writer back
            self.flush()
            return self.w

    class Decoder(_BufferedDecoder):                   # bwt/mod.rs:321-432
        def __init__(self, r, extra_mem=True):
            # extra_mem = False selects the reference's `decode_minimal` (bwt/mod.rs:298-315, called at :397-399), which is NOT
            # the inverse of the encoder in general and is unused by the reference's own application
            # (main.rs:90).  It is reproduced as the reference computes it (rcx_bwt_inverse_minimal_batch): a drop-in returns
            # what the reference returns.  Use extra_mem = True to get the text back.
            super().__init__(r)
            self.extra_memory = extra_mem

        def _decode_all(self, data):
            p, n = 0, len(data)
            if n - p < 4:
                raise UnexpectedEof(1)                 # :369
            self.max_block_size = struct.unpack_from("<I", data, p)[0]
            p += 4
            Ls, origins = [], []
            while n - p >= 4:                          # EOF at a block boundary ends the stream, :374-377
                bn = struct.unpack_from("<I", data, p)[0]
                p += 4
                if n - p < bn:
                    raise UnexpectedEof(1)
                L = data[p:p + bn]
                p += bn
                if n - p < 4:
                    raise UnexpectedEof(1)
                origins.append(struct.unpack_from("<I", data, p)[0])
                p += 4
                if bn == 0 and self.extra_memory:
                    raise Malformed(3)                 # input[origin] panics, :230 (decode_minimal: only if origin != 0, :300-302)
                Ls.append(L)
            if not Ls:
                return b""
            inv = context().bwt_inverse if self.extra_memory else context().bwt_inverse_minimal
            return b"".join(_check(inv(Ls, origins)).outputs)


# ------------------------------------------------------------------------------------------------ ari
class _ari:
    class ByteEncoder:                                 # table.rs:185-224
        def __init__(self, w):
            self.w, self._buf = w, bytearray()

        def write(self, buf):
            self._buf += bytes(buf)
            return len(buf)

        def finish(self):
            self.w.write(_check(context().ari_byte_encode([bytes(self._buf)])).outputs[0])
            return self.w

    class ByteDecoder(_BufferedDecoder):               # table.rs:229-273; stops exactly at the stream's end
        def _decode_all(self, data):
            res = _check(_grow_caps(lambda cap: context().ari_byte_decode([data], [cap]), max(1 << 12, 4 * len(data))))
            self.consumed = int(res.in_used[0])        # mod.rs:289-292: the reader ends exactly after this stream
            return res.outputs[0]


# the per-symbol surface (RangeEncoder, the Model trait, the generic Encoder / Decoder, table / bin / apm models): host
# integer code, ari_symbol.py -- one decision per call against a caller-owned model has no batch to give the device
from . import ari_symbol as _sym   # noqa: E402
for _n in ("RangeEncoder", "Model", "Encoder", "Decoder", "table", "bin", "apm", "RANGE_DEFAULT_THRESHOLD", "PanicError"):
    setattr(_ari, _n, getattr(_sym, _n))


class entropy:
    ari = _ari


# ------------------------------------------------------------------------------------------------ rle
class rle:
    class Encoder:
        """rle.rs:40-143.  Buffers, and codes on flush / finish with one batch call; writes what the reference's streaming
        `write` (:100-114) and `flush` (:116-143) write for the same calls, including their two oddities: every write after
        the first drops its first byte (the loop starts at buf[1..]; only the very first call seeds the run with buf[0]),
        and a flush in the middle of a run writes the run without closing it, so it is written again, with its full count,
        when it ends."""

        def __init__(self, w):
            self.w, self._buf, self._in_run = w, bytearray(), False

        def write(self, buf):
            buf = bytes(buf)
            if not self._in_run and buf:
                self._buf.append(buf[0])
                self._in_run = True
            self._buf += buf[1:]
            return len(buf)

        write_all = write

        def flush(self):
            if self._buf:
                self.w.write(_check(context().rle_encode([bytes(self._buf)])).outputs[0])
                last = self._buf[-1]
                run = len(self._buf) - len(self._buf.rstrip(bytes([last])))
                del self._buf[: len(self._buf) - run]      # the open run stays open (:116-143 never reset it)

        def finish(self):                              # :62-66
            self.flush()
            return self.w

    class Decoder(_BufferedDecoder):                   # rle.rs:176-281
        def _decode_all(self, data):
            res = _grow_caps(lambda cap: context().rle_decode([data], [cap]), max(1 << 12, 16 * len(data)))
            if res.status[0] == 30:
                raise CompressError(30)                # io::ErrorKind::Other "Overly long run"
            return _check(res).outputs[0]
