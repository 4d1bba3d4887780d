//! Run-length codecs (reference: src/rle.rs:40-123, 176-281).  The encoder buffers and codes on `flush` / `finish` with ONE
//! batch call; what it writes is what the reference's streaming `write` (:100-114) and `flush` (:116-143) write for the same
//! calls, including their two oddities: every `write` after the first drops its first byte (the loop starts at `buf[1..]`
//! and only the very first call seeds the run with `buf[0]`), and a `flush` in the middle of a run writes the run without
//! closing it, so the run is written again -- with its full count -- when it ends.
use crate::rcx_sys::*;
use crate::{grow_decode, run_batch, Buffered, TailReader};
use std::io::{self, Read, Write};

pub struct Encoder<W: Write> {
    w: W,
    buf: Vec<u8>,
    in_run: bool,
}

impl<W: Write> Encoder<W> {
    pub fn new(w: W) -> Encoder<W> {
        Encoder { w, buf: Vec::new(), in_run: false }
    }
    /// rle.rs:62-66: `flush`, then the writer back.
    pub fn finish(mut self) -> (W, io::Result<()>) {
        let res = self.flush();
        (self.w, res)
    }
}

impl<W: Write> Write for Encoder<W> {
    /// rle.rs:100-114: the first call seeds the run with `buf[0]`; every call feeds `buf[1..]`.
    fn write(&mut self, buf: &[u8]) -> io::Result<usize> {
        if !self.in_run && !buf.is_empty() {
            self.buf.push(buf[0]);
            self.in_run = true;
        }
        if buf.len() > 1 {
            self.buf.extend_from_slice(&buf[1..]);
        }
        Ok(buf.len())
    }
    /// rle.rs:116-143: everything buffered is coded (one batch call) and written; the open run stays open -- the buffer is
    /// re-seeded with it, so it comes out again, with whatever is added to it, at the next flush.
    fn flush(&mut self) -> io::Result<()> {
        if self.buf.is_empty() {
            return Ok(());
        }
        let cap = unsafe { rcx_rle_encode_bound(self.buf.len() as u64) };
        let r = run_batch(&[&self.buf[..]], &[cap], |c, b, _| unsafe { rcx_rle_encode_batch(c, b) }).check()?;
        self.w.write_all(&r.out[0])?;
        let last = *self.buf.last().unwrap();
        let run = self.buf.iter().rev().take_while(|&&x| x == last).count();
        let start = self.buf.len() - run;
        self.buf.drain(..start);
        Ok(())
    }
}

pub struct Decoder<R: Read> {
    pub r: TailReader<R>,
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
