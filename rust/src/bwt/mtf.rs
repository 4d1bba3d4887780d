//! Move-to-front stream codecs (reference: src/bwt/mtf.rs:95-169; list initialised 0..255, :103-104, 141-142).
use crate::rcx_sys::*;
use crate::{run_batch, Buffered, TailReader};
use std::io::{self, Read, Write};

pub struct Encoder<W: Write> {
    w: W,
    buf: Vec<u8>,
}

impl<W: Write> Encoder<W> {
    pub fn new(w: W) -> Encoder<W> {
        Encoder { w, buf: Vec::new() }
    }
    /// mtf.rs:112-115 (returns the writer; the ranks are written here, in one batch call)
    pub fn finish(mut self) -> W {
        let r = run_batch(&[&self.buf[..]], &[self.buf.len() as u64], |c, b, _| unsafe { rcx_mtf_encode_batch(c, b) }).check().unwrap();
        self.w.write_all(&r.out[0]).unwrap();
        self.w
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
    pub r: TailReader<R>,
    buf: Buffered,
}

impl<R: Read> Decoder<R> {
    pub fn new(r: R) -> Decoder<R> {
        Decoder { r: TailReader::new(r), buf: Buffered::new() }
    }
    /// mtf.rs:150-153
    pub fn finish(self) -> TailReader<R> {
        self.r
    }
}

impl<R: Read> Read for Decoder<R> {
    fn read(&mut self, dst: &mut [u8]) -> io::Result<usize> {
        self.buf.ensure(&mut self.r, |raw| {
            let r = run_batch(&[raw], &[raw.len() as u64], |c, b, _| unsafe { rcx_mtf_decode_batch(c, b) }).check()?;
            Ok((r.out[0].clone(), None))
        })?;
        Ok(self.buf.serve(dst))
    }
}

/// mtf.rs:44-91 in batch-backed form.  The crate's `MTF` moves one symbol per call (`encode(sym) -> rank`, `decode(rank) ->
/// sym`); a whole block through the same list, started as the stream codecs start it (identity, :103-104), is one kernel call.
pub struct MTF;

impl MTF {
    pub fn encode_block(input: &[u8]) -> Vec<u8> {
        run_batch(&[input], &[input.len() as u64], |c, b, _| unsafe { rcx_mtf_encode_batch(c, b) }).check().unwrap().out[0].clone()
    }
    pub fn decode_block(ranks: &[u8]) -> Vec<u8> {
        run_batch(&[ranks], &[ranks.len() as u64], |c, b, _| unsafe { rcx_mtf_decode_batch(c, b) }).check().unwrap().out[0].clone()
    }
}
