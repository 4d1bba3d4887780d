//! Adaptive order-0 byte range coder (reference: src/entropy/ari/table.rs:185-273 over mod.rs:67-293, table.rs:20-122).
pub mod apm;
pub mod bin;
pub mod table;

use crate::rcx_sys::*;
use crate::{grow_decode, run_batch, Buffered, TailReader};
use std::io::{self, Read, Write};

// Two layers, as in the crate.  PER SYMBOL: `RangeEncoder`, the `Model<V>` trait and the generic `Encoder<W>` / `Decoder<R>`
// (mod.rs:67-293) code one decision per call against a model the caller owns and updates between calls -- no batch exists to
// hand a device, so they are host code here as well (integer arithmetic; the C++ twin is rust_compress_amd/host/ari_symbol.hpp,
// checked symbol by symbol against the oracle's streams).  PER STREAM: `ByteEncoder` / `ByteDecoder` below and the
// `encode_bytes` / `decode_bytes` forms in `table`, `bin`, `apm` run on the GPU, one kernel call per stream.

pub type Symbol = u8;
pub type Border = u32;
const BORDER_BYTES: usize = 4;
const TOP_BYTE: Border = 0xff00_0000;
pub const RANGE_DEFAULT_THRESHOLD: Border = 1 << 14;

/// mod.rs:67-169.  The interval [low, hai) lives in 32 bits; a symbol narrows it to its share, every leading byte both ends
/// agree on is shipped, and an interval that straddles a byte boundary while narrower than `threshold` is cut at the boundary
/// (the larger side survives) so that a byte can leave.
pub struct RangeEncoder {
    low: Border,
    hai: Border,
    pub threshold: Border,
}

impl RangeEncoder {
    pub fn new(max_range: Border) -> RangeEncoder {
        RangeEncoder { low: 0, hai: !0, threshold: max_range }
    }
    pub fn reset(&mut self) {
        self.low = 0;
        self.hai = !0;
    }
    /// [from/total, to/total) of the current interval; returns how many bytes left into `output`
    pub fn process(&mut self, total: Border, from: Border, to: Border, output: &mut [Symbol]) -> usize {
        let width = self.hai.wrapping_sub(self.low) / total;
        let mut a = self.low.wrapping_add(width.wrapping_mul(from));
        let mut b = self.low.wrapping_add(width.wrapping_mul(to));
        let mut shipped = 0;
        loop {
            if (a ^ b) & TOP_BYTE != 0 {
                if b.wrapping_sub(a) > self.threshold {
                    break;
                }
                let edge = b & TOP_BYTE;
                if b.wrapping_sub(edge) >= edge.wrapping_sub(a) {
                    a = edge;
                } else {
                    b = edge.wrapping_sub(1);
                }
            }
            output[shipped] = (a >> 24) as Symbol;
            shipped += 1;
            a <<= 8;
            b <<= 8;
        }
        self.low = a;
        self.hai = b;
        shipped
    }
    /// the offset in [0, total) that `code` stands for
    pub fn query(&self, total: Border, code: Border) -> Border {
        code.wrapping_sub(self.low) / (self.hai.wrapping_sub(self.low) / total)
    }
    pub fn get_code_tail(&mut self) -> Border {
        let tail = self.low;
        self.low = 0;
        self.hai = 0;
        tail
    }
}

/// mod.rs:174-204: a model hands out probability ranges; `encode` / `decode` are the trait's provided methods.
pub trait Model<V: Copy> {
    fn get_range(&self, value: V) -> (Border, Border);
    fn find_value(&self, offset: Border) -> (V, Border, Border);
    fn get_denominator(&self) -> Border;

    fn encode(&self, value: V, re: &mut RangeEncoder, out: &mut [Symbol]) -> usize {
        let (lo, hi) = self.get_range(value);
        re.process(self.get_denominator(), lo, hi, out)
    }
    fn decode(&self, code: Border, re: &mut RangeEncoder) -> (V, usize) {
        let total = self.get_denominator();
        let (value, lo, hi) = self.find_value(re.query(total, code));
        let mut scratch = [0 as Symbol; BORDER_BYTES + 4];
        let shift = re.process(total, lo, hi, &mut scratch[..]);
        (value, shift)
    }
}

/// mod.rs:208-251
pub struct Encoder<W> {
    stream: W,
    range: RangeEncoder,
}

impl<W: Write> Encoder<W> {
    pub fn new(w: W) -> Encoder<W> {
        Encoder { stream: w, range: RangeEncoder::new(RANGE_DEFAULT_THRESHOLD) }
    }
    pub fn encode<V: Copy, M: Model<V>>(&mut self, value: V, model: &M) -> io::Result<()> {
        let mut buf = [0 as Symbol; BORDER_BYTES + 4];
        let n = model.encode(value, &mut self.range, &mut buf[..]);
        self.stream.write(&buf[..n]).map(|_| ())
    }
    /// the code tail, big-endian, then the writer back
    pub fn finish(mut self) -> (W, io::Result<()>) {
        let tail = self.range.get_code_tail().to_be_bytes();
        let res = self.stream.write_all(&tail).and_then(|_| self.stream.flush());
        (self.stream, res)
    }
    pub fn flush(&mut self) -> io::Result<()> {
        self.stream.flush()
    }
}

/// mod.rs:254-293
pub struct Decoder<R> {
    stream: R,
    range: RangeEncoder,
    code: Border,
    bytes_pending: usize,
}

impl<R: Read> Decoder<R> {
    pub fn new(r: R) -> Decoder<R> {
        Decoder { stream: r, range: RangeEncoder::new(RANGE_DEFAULT_THRESHOLD), code: 0, bytes_pending: BORDER_BYTES }
    }
    fn feed(&mut self) -> io::Result<()> {
        while self.bytes_pending != 0 {
            let mut b = [0u8; 1];
            self.stream.read_exact(&mut b)?;
            self.code = (self.code << 8).wrapping_add(b[0] as Border);
            self.bytes_pending -= 1;
        }
        Ok(())
    }
    pub fn decode<V: Copy, M: Model<V>>(&mut self, model: &M) -> io::Result<V> {
        self.feed().unwrap(); // :277, a panic in the reference as well
        let (value, shift) = model.decode(self.code, &mut self.range);
        self.bytes_pending = shift;
        Ok(value)
    }
    /// reads what the last symbol left pending: the reader comes back right behind the stream (:289-292)
    pub fn finish(mut self) -> (R, io::Result<()>) {
        let err = self.feed();
        (self.stream, err)
    }
}

/// table.rs:185-224
pub struct ByteEncoder<W: Write> {
    w: W,
    buf: Vec<u8>,
}

impl<W: Write> ByteEncoder<W> {
    pub fn new(w: W) -> ByteEncoder<W> {
        ByteEncoder { w, buf: Vec::new() }
    }
    /// table.rs:203-208: codes the EOF symbol and the 4-byte tail, returns the writer
    pub fn finish(mut self) -> (W, io::Result<()>) {
        let cap = unsafe { rcx_ari_byte_encode_bound(self.buf.len() as u64) };
        let res = match run_batch(&[&self.buf[..]], &[cap], |c, b, _| unsafe { rcx_ari_byte_encode_batch(c, b) }).check() {
            Ok(r) => self.w.write_all(&r.out[0]),
            Err(e) => Err(e),
        };
        (self.w, res)
    }
}

impl<W: Write> Write for ByteEncoder<W> {
    fn write(&mut self, buf: &[u8]) -> io::Result<usize> {
        self.buf.extend_from_slice(buf);
        Ok(buf.len())
    }
    fn flush(&mut self) -> io::Result<()> {
        self.w.flush()
    }
}

/// table.rs:229-273.  Stops exactly at the stream's end: `finish()` returns the reader positioned after it
/// (mod.rs:289-292; test.rs:52-89 decodes two streams back to back from one reader).
pub struct ByteDecoder<R: Read> {
    pub r: TailReader<R>,
    buf: Buffered,
}

impl<R: Read> ByteDecoder<R> {
    pub fn new(r: R) -> ByteDecoder<R> {
        ByteDecoder { r: TailReader::new(r), buf: Buffered::new() }
    }
    pub fn finish(mut self) -> (TailReader<R>, io::Result<()>) {
        let res = self.fill();
        (self.r, res)
    }
    fn fill(&mut self) -> io::Result<()> {
        self.buf.ensure(&mut self.r, |raw| {
            let r = grow_decode(raw, 4 * raw.len() as u64, |c, b, _| unsafe { rcx_ari_byte_decode_batch(c, b) })?;
            Ok((r.out[0].clone(), Some(r.in_used[0] as usize)))
        })
    }
}

impl<R: Read> Read for ByteDecoder<R> {
    fn read(&mut self, dst: &mut [u8]) -> io::Result<usize> {
        self.fill()?;
        Ok(self.buf.serve(dst))
    }
}
