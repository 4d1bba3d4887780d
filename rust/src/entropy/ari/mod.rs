//! Adaptive order-0 byte range coder (reference: src/entropy/ari/table.rs:185-273 over mod.rs:67-293, table.rs:20-122).
use crate::rcx_sys::*;
use crate::{grow_decode, run_batch, Buffered, TailReader};
use std::io::{self, Read, Write};

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
    r: TailReader<R>,
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
