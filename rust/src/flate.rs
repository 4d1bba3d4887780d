//! RFC 1951 decoder (reference: src/flate.rs:164-193, 453-488).  Batch semantics: the stream is decoded to BFINAL in one
//! call; `flags & RCX_W_EMPTY_BLOCK_MIDSTREAM` reports the reference's `Ok(0)` quirk (:474-476) instead of acting it out.
use crate::rcx_sys::*;
use crate::{decode_many_with, grow_decode, Buffered, TailReader};
use std::io::{self, Read};

pub struct Decoder<R: Read> {
    /// `pub r: R` (flate.rs:166), left exactly after the DEFLATE stream once it is decoded (:250-260 reads byte by byte)
    pub r: TailReader<R>,
    buf: Buffered,
    pub flags: u32,
}

impl<R: Read> Decoder<R> {
    pub fn new(r: R) -> Decoder<R> {
        Decoder { r: TailReader::new(r), buf: Buffered::new(), flags: 0 }
    }
    /// :453-458: the final block has been served completely
    pub fn eof(&mut self) -> bool {
        self.buf.eof()
    }
    /// :460-465
    pub fn reset(&mut self) {
        self.buf.reset()
    }
}

impl<R: Read> Read for Decoder<R> {
    fn read(&mut self, dst: &mut [u8]) -> io::Result<usize> {
        let flags = &mut self.flags;
        self.buf.ensure(&mut self.r, |raw| {
            let r = grow_decode(raw, 4 * raw.len() as u64, |c, b, f| unsafe { rcx_inflate_batch(c, b, f) })?;
            *flags = r.aux[0];
            Ok((r.out[0].clone(), Some(r.in_used[0] as usize)))
        })?;
        Ok(self.buf.serve(dst))
    }
}

/// Many raw DEFLATE streams through ONE batch call (the reference decodes one deflate block per `read()`, flate.rs:468-488: on the
/// GPU a stream is one wave's work, so a caller with many streams hands them over together).  -> per stream (decoded bytes, input
/// bytes used, flags); the first stream that fails returns what its `Decoder` would.
pub fn decode_many(streams: &[&[u8]]) -> io::Result<Vec<(Vec<u8>, usize, u32)>> {
    decode_many_with(streams, |c, b, f| unsafe { rcx_inflate_batch(c, b, f) })
}
