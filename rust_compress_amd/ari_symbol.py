"""The reference's PER-SYMBOL range-coder surface as host code (the Python twin of host/ari_symbol.hpp):

    RangeEncoder                         src/entropy/ari/mod.rs:67-169
    Model (get_range / find_value / get_denominator + encode / decode)   mod.rs:174-204
    Encoder(w) / Decoder(r)              mod.rs:208-293
    table.Model, table.SumProxy          src/entropy/ari/table.rs:20-180
    bin.Model, bin.SumProxy              src/entropy/ari/bin.rs:17-167
    apm.Bit, apm.Gate                    src/entropy/ari/apm.rs:36-198

This surface codes ONE decision per call against a model the caller owns and updates between calls (test.rs:22-50,
91-148): there is no batch to give the device, so it is integer host code.  The whole-stream codecs over the same
arithmetic (compress.entropy.ari.ByteEncoder / ByteDecoder) run on the GPU; tests/test_ari_symbol_host.py checks that the
bytes agree with the oracle's streams.  A Rust panic is a PanicError.
"""
import struct

M32 = 0xFFFFFFFF
RANGE_DEFAULT_THRESHOLD = 1 << 14
_TOP = 0xFF000000


class PanicError(AssertionError):
    pass


class RangeEncoder:
    def __init__(self, max_range):
        self.threshold, self.low, self.hai = max_range, 0, M32

    def reset(self):
        self.low, self.hai = 0, M32

    def process(self, total, frm, to):
        """[frm/total, to/total) of the current interval -> the bytes that leave"""
        # mod.rs:118-122: the reference's asserts; a zero-width interval would otherwise ship bytes for ever (the reference's
        # release build dies on output[4]: the slice has BORDER_BYTES = 4 entries)
        if total == 0:
            raise PanicError("attempt to divide by zero (RangeEncoder.process: total == 0)")
        if not (frm < to <= total):
            raise PanicError("assertion failed: from<to && to<=total")
        width = ((self.hai - self.low) & M32) // total
        if width == 0:
            raise PanicError("RangeCoder range is too narrow for the total")
        a, b = (self.low + width * frm) & M32, (self.low + width * to) & M32
        out = bytearray()
        while True:
            if (a ^ b) & _TOP:
                if ((b - a) & M32) > self.threshold:
                    break
                edge = b & _TOP
                if ((b - edge) & M32) >= ((edge - a) & M32):
                    a = edge
                else:
                    b = (edge - 1) & M32
            if len(out) == 4:
                raise PanicError("index out of bounds: the len is 4 but the index is 4 (RangeEncoder.process)")
            out.append(a >> 24)
            a, b = (a << 8) & M32, (b << 8) & M32
        self.low, self.hai = a, b
        return bytes(out)

    def query(self, total, code):
        return ((code - self.low) & M32) // (((self.hai - self.low) & M32) // total)

    def get_code_tail(self):
        t, self.low, self.hai = self.low, 0, 0
        return t


class Model:
    """the trait: subclasses give get_range(value), find_value(offset), get_denominator()"""

    def encode(self, value, re):
        lo, hi = self.get_range(value)
        return re.process(self.get_denominator(), lo, hi)

    def decode(self, code, re):
        total = self.get_denominator()
        value, lo, hi = self.find_value(re.query(total, code))
        return value, len(re.process(total, lo, hi))


class Encoder:
    def __init__(self, w):
        self.stream, self.range = w, RangeEncoder(RANGE_DEFAULT_THRESHOLD)

    def encode(self, value, model):
        out = model.encode(value, self.range)
        if out:
            self.stream.write(out)

    def finish(self):
        self.stream.write(struct.pack(">I", self.range.get_code_tail()))
        if hasattr(self.stream, "flush"):
            self.stream.flush()
        return self.stream

    def flush(self):
        if hasattr(self.stream, "flush"):
            self.stream.flush()


class Decoder:
    def __init__(self, r):
        self.stream, self.range, self.code, self.bytes_pending = r, RangeEncoder(RANGE_DEFAULT_THRESHOLD), 0, 4

    def _feed(self):
        while self.bytes_pending:
            b = self.stream.read(1)
            if len(b) != 1:
                return False
            self.code = ((self.code << 8) + b[0]) & M32
            self.bytes_pending -= 1
        return True

    def decode(self, model):
        if not self._feed():
            raise PanicError("feed().unwrap(): the stream ended inside the code")
        value, self.bytes_pending = model.decode(self.code, self.range)
        return value

    def finish(self):
        """-> the reader, standing right behind the stream (mod.rs:289-292)"""
        self._feed()
        return self.stream


def _bad_offset(offset, total):
    return PanicError("Invalid frequency offset %d requested under total %d" % (offset, total))


class table:
    class Model(Model):
        def __init__(self, freq, threshold):
            self.table, self.cut_threshold, self.cut_shift = [f & 0xFFFF for f in freq], threshold, 1
            self.total = sum(self.table) & M32
            while self.total >= threshold:
                self.downscale()

        @classmethod
        def new_custom(cls, num_values, threshold, fn_init):
            return cls([fn_init(i) for i in range(num_values)], threshold)

        @classmethod
        def new_flat(cls, num_values, threshold):
            return cls([1] * num_values, threshold)

        def reset_flat(self):
            self.table = [1] * len(self.table)
            self.total = len(self.table)

        def update(self, value, add_log, add_const):
            add = (self.total >> add_log) + add_const
            if not add < 2 * self.cut_threshold:
                raise PanicError("add < 2 * cut_threshold")
            self.table[value] = (self.table[value] + add) & 0xFFFF
            self.total += add
            if self.total >= self.cut_threshold:
                self.downscale()
                if not self.total < self.cut_threshold:
                    raise PanicError("total < cut_threshold")

        def downscale(self):
            up = (1 << self.cut_shift) - 1
            self.table = [((f + up) & 0xFFFF) >> self.cut_shift for f in self.table]
            self.total = sum(self.table)

        def get_frequencies(self):
            return self.table

        def get_range(self, value):
            if not 0 <= value < len(self.table):
                raise PanicError("index out of bounds")
            lo = sum(self.table[:value])
            return lo, lo + self.table[value]

        def find_value(self, offset):
            if not offset < self.total:
                raise _bad_offset(offset, self.total)
            lo = 0
            for v, f in enumerate(self.table):
                if lo + f > offset:
                    return v, lo, lo + f
                lo += f
            raise PanicError("index out of bounds")

        def get_denominator(self):
            return self.total

    class SumProxy(Model):
        def __init__(self, wa, fa, wb, fb, shift):
            if len(fa.get_frequencies()) != len(fb.get_frequencies()):
                raise PanicError("assert_eq!(fa.len(), fb.len())")
            self.a, self.b, self.wa, self.wb, self.ws = fa, fb, wa, wb, shift

        def get_range(self, value):
            (l0, h0), (l1, h1) = self.a.get_range(value), self.b.get_range(value)
            return ((self.wa * l0 + self.wb * l1) & M32) >> self.ws, ((self.wa * h0 + self.wb * h1) & M32) >> self.ws

        def find_value(self, offset):
            total = self.get_denominator()
            if not offset < total:
                raise _bad_offset(offset, total)
            lo = 0
            for v, (fa, fb) in enumerate(zip(self.a.get_frequencies(), self.b.get_frequencies())):
                hi = lo + (((self.wa * fa + self.wb * fb) & M32) >> self.ws)
                if hi > offset:
                    return v, lo, hi
                lo = hi
            raise PanicError("index out of bounds")

        def get_denominator(self):
            return ((self.wa * self.a.get_denominator() + self.wb * self.b.get_denominator()) & M32) >> self.ws


class bin:  # noqa: A001  (the reference's module name)
    class Model(Model):
        def __init__(self, zero, total, rate):
            self.zero, self.total, self.rate = zero, total, rate

        @classmethod
        def new_flat(cls, threshold, rate):
            return cls(threshold >> 1, threshold, rate)

        @classmethod
        def new_custom(cls, zero_percent, threshold, rate):
            if threshold < 100:
                raise PanicError("threshold >= 100")
            return cls(zero_percent * threshold // 100, threshold, rate)

        def reset_flat(self):
            self.zero = self.total >> 1

        def get_probability_zero(self):
            return self.zero

        def get_probability_one(self):
            return self.total - self.zero

        def update_zero(self):
            self.zero += (self.total - self.zero) >> self.rate

        def update_one(self):
            self.zero -= self.zero >> self.rate

        def update(self, value):
            if value:
                self.update_one()
            else:
                self.update_zero()

        def get_range(self, value):
            return (self.zero, self.total) if value else (0, self.zero)

        def find_value(self, offset):
            if not offset < self.total:
                raise _bad_offset(offset, self.total)
            return (False, 0, self.zero) if offset < self.zero else (True, self.zero, self.total)

        def get_denominator(self):
            return self.total

    class SumProxy(Model):
        def __init__(self, wa, first, wb, second, shift):
            self.a, self.b, self.wa, self.wb, self.ws = first, second, wa, wb, shift

        def _zero(self):
            return ((self.wa * self.a.get_probability_zero() + self.wb * self.b.get_probability_zero()) & M32) >> self.ws

        def get_range(self, value):
            z = self._zero()
            return (z, self.get_denominator()) if value else (0, z)

        def find_value(self, offset):
            z, total = self._zero(), self.get_denominator()
            if not offset < total:
                raise _bad_offset(offset, total)
            return (False, 0, z) if offset < z else (True, z, total)

        def get_denominator(self):
            return ((self.wa * self.a.get_denominator() + self.wb * self.b.get_denominator()) & M32) >> self.ws


class apm:
    """12-bit "flat" probabilities and their stretched form; the f32 ln / exp come from this host's libm (numpy float32),
    which is what Rust's f32::ln / f32::exp call."""
    FLAT_TOTAL, WIDE_OFFSET, PORTAL_OFFSET, PORTAL_BINS = 1 << 12, 1 << 11, 8, 17

    class Bit(Model):
        def __init__(self, fp):
            self.fp = fp & 0xFFFF

        @classmethod
        def new_equal(cls):
            return cls(apm.FLAT_TOTAL >> 1)

        @classmethod
        def from_flat(cls, fp):
            return cls(fp)

        @classmethod
        def from_wide(cls, wp):
            import numpy as np
            d = np.float32(wp) / np.float32(apm.WIDE_OFFSET)
            p = np.float32(1.0) / (np.float32(1.0) + np.exp(-d, dtype=np.float32))
            x = float(p * np.float32(apm.FLAT_TOTAL))
            if not (-1.0 < x < 65536.0):
                raise PanicError("to_u16().unwrap()")
            return cls(int(x))

        def to_flat(self):
            return self.fp

        def to_wide(self):
            import numpy as np
            with np.errstate(divide="ignore", invalid="ignore"):
                p = np.float32(self.fp) / np.float32(apm.FLAT_TOTAL)
                d = np.log(p / (np.float32(1.0) - p), dtype=np.float32)
                x = float(d * np.float32(apm.WIDE_OFFSET))
            if not (-32769.0 < x < 32768.0):
                raise PanicError("to_i16().unwrap()")
            return int(x)

        def update_zero(self, rate, bias):
            self.fp = (self.fp + ((apm.FLAT_TOTAL - bias - self.fp) >> rate)) & 0xFFFF

        def update_one(self, rate, bias):
            self.fp = (self.fp - ((self.fp - bias) >> rate)) & 0xFFFF

        def update(self, value, rate, bias):
            if value:
                self.update_one(rate, bias)
            else:
                self.update_zero(rate, bias)

        def get_range(self, value):
            return (self.fp, apm.FLAT_TOTAL) if value else (0, self.fp)

        def find_value(self, offset):
            if not offset < apm.FLAT_TOTAL:
                raise PanicError("Invalid bit offset %d requested" % offset)
            return (False, 0, self.fp) if offset < self.fp else (True, self.fp, apm.FLAT_TOTAL)

        def get_denominator(self):
            return apm.FLAT_TOTAL

    class Gate:
        def __init__(self):
            import numpy as np
            self.map = []
            for i in range(apm.PORTAL_BINS):
                rp = np.float32(i) / np.float32(apm.PORTAL_OFFSET) - np.float32(1.0)
                self.map.append(apm.Bit.from_wide(int(rp * np.float32(apm.WIDE_OFFSET))))

        def pass_(self, bit):
            """`Gate::pass` (a Python keyword): -> (Bit, (index, weight))"""
            fp, coords = self.pass_wide(bit.to_wide())
            return apm.Bit.from_flat(fp), coords

        def pass_wide(self, wp):
            index = (wp + apm.WIDE_OFFSET) >> 8
            if index < 0 or index + 1 >= apm.PORTAL_BINS:
                raise PanicError("index out of bounds")
            weight = wp & 255
            s = self.map[index].to_flat() * (256 - weight) + self.map[index + 1].to_flat() * weight
            return (s >> 8) & 0xFFFF, (index, weight)

        def update_zero(self, bc, rate, bias):
            self.map[bc[0]].update_zero(rate, bias)
            self.map[bc[0] + 1].update_zero(rate, bias)

        def update_one(self, bc, rate, bias):
            self.map[bc[0]].update_one(rate, bias)
            self.map[bc[0] + 1].update_one(rate, bias)

        def update(self, value, bc, rate, bias):
            if value:
                self.update_one(bc, rate, bias)
            else:
                self.update_zero(bc, rate, bias)
