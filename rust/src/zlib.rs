//! RFC 1950 decoder (reference: src/zlib.rs:32-127): CMF/FLG checks, DEFLATE, Adler-32 trailer -- all on the device.
use crate::rcx_sys::*;
use crate::{decode_many_with, grow_decode, Buffered, TailReader};
use std::io::{self, Read};

pub struct Decoder<R: Read> {
    r: TailReader<R>,
    buf: Buffered,
}

impl<R: Read> Decoder<R> {
    /// zlib.rs:42-49
    pub fn new(r: R) -> Decoder<R> {
        Decoder { r: TailReader::new(r), buf: Buffered::new() }
    }
    /// zlib.rs:51-53: the reader, positioned exactly after the 4-byte Adler-32 trailer
    pub fn unwrap(self) -> TailReader<R> {
        self.r
    }
}

impl<R: Read> Read for Decoder<R> {
    fn read(&mut self, dst: &mut [u8]) -> io::Result<usize> {
        self.buf.ensure(&mut self.r, |raw| {
            let r = grow_decode(raw, 4 * raw.len() as u64, |c, b, f| unsafe { rcx_zlib_decode_batch(c, b, f) })?;
            Ok((r.out[0].clone(), Some(r.in_used[0] as usize)))
        })?;
        Ok(self.buf.serve(dst))
    }
}

/// Many zlib members through ONE batch call, every member's Adler-32 checked on the device.  -> per member (decoded bytes, input
/// bytes used); the first member that fails returns what its `Decoder` would.
pub fn decode_many(members: &[&[u8]]) -> io::Result<Vec<(Vec<u8>, usize)>> {
    Ok(decode_many_with(members, |c, b, f| unsafe { rcx_zlib_decode_batch(c, b, f) })?.into_iter().map(|(o, u, _)| (o, u)).collect())
}
