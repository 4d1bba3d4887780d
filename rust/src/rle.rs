//! Run-length codecs (reference: src/rle.rs:40-123, 176-281).  The encoder has one-shot semantics (`write_all` + `finish`,
//! what the reference's tests use, :320-352).
use crate::rcx_sys::*;
use crate::{grow_decode, run_batch, Buffered, TailReader};
use std::io::{self, Read, Write};

pub struct Encoder<W: Write> {
    w: W,
    buf: Vec<u8>,
}

impl<W: Write> Encoder<W> {
    pub fn new(w: W) -> Encoder<W> {
        Encoder { w, buf: Vec::new() }
    }
    /// rle.rs:62-66
    pub fn finish(mut self) -> (W, io::Result<()>) {
        let cap = unsafe { rcx_rle_encode_bound(self.buf.len() as u64) };
        let res = match run_batch(&[&self.buf[..]], &[cap], |c, b, _| unsafe { rcx_rle_encode_batch(c, b) }).check() {
            Ok(r) => self.w.write_all(&r.out[0]),
            Err(e) => Err(e),
        };
        (self.w, res)
    }
}

impl<W: Write> Write for Encoder<W> {
    fn write(&mut self, buf: &[u8]) -> io::Result<usize> {
        self.buf.extend_from_slice(buf);
        Ok(buf.len())
    }
    fn flush(&mut self) -> io::Result<()> {
        Ok(())
    }
}

pub struct Decoder<R: Read> {
    r: TailReader<R>,
    buf: Buffered,
}

impl<R: Read> Decoder<R> {
    pub fn new(r: R) -> Decoder<R> {
        Decoder { r: TailReader::new(r), buf: Buffered::new() }
    }
}

impl<R: Read> Read for Decoder<R> {
    /// "Overly long run" (rle.rs:152-154) comes back as io::ErrorKind::Other
    fn read(&mut self, dst: &mut [u8]) -> io::Result<usize> {
        self.buf.ensure(&mut self.r, |raw| {
            let r = grow_decode(raw, 16 * raw.len() as u64, |c, b, _| unsafe { rcx_rle_decode_batch(c, b) })?;
            Ok((r.out[0].clone(), None))
        })?;
        Ok(self.buf.serve(dst))
    }
}
