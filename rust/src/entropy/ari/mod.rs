//! Adaptive order-0 byte range coder (reference: src/entropy/ari/table.rs:185-273 over mod.rs:67-293, table.rs:20-122).
pub mod apm;
pub mod bin;
pub mod table;

use crate::rcx_sys::*;
use crate::{grow_decode, run_batch, Buffered, TailReader};
use std::io::{self, Read, Write};

// What is NOT here, and why: `RangeEncoder` (mod.rs:67-165), the `Model<V>` trait (:170-196) and the generic `Encoder<W>` /
// `Decoder<R>` (:200-293) code ONE symbol per call against a caller-supplied model -- host-side control flow that cannot
// cross an FFI made of batch calls.  A port keeps using the crate's own host code for those; the device offers every model
// the crate ships as whole-stream codecs: `table::Model` through `ByteEncoder` / `ByteDecoder` below, `bin::Model`,
// `table::SumProxy` + `bin::SumProxy`, `apm::Bit` + `apm::Gate` through the `encode_bytes` / `decode_bytes` of their modules.

/// The stream coder over `table::Model` is the crate's `ari::Encoder<W>` / `ari::Decoder<R>` as its byte codecs use them.
pub type Encoder<W> = ByteEncoder<W>;
pub type Decoder<R> = ByteDecoder<R>;

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
