//! BWT stream codecs and block functions (reference: src/bwt/mod.rs).
pub mod dc;
pub mod mtf;

use crate::rcx_sys::*;
use crate::{eof_error, le32, run_batch, Buffered, TailReader};
use std::io::{self, Read, Write};

/// bwt/mod.rs:214-219 -> (L, origin)
pub fn encode_simple(input: &[u8]) -> (Vec<u8>, usize) {
    let r = run_batch(&[input], &[input.len() as u64], |c, b, o| unsafe { rcx_bwt_forward_batch(c, b, o) }).check().unwrap();
    (r.out[0].clone(), r.aux[0] as usize)
}

/// bwt/mod.rs:291-294
pub fn decode_simple(input: &[u8], origin: usize) -> Vec<u8> {
    if input.is_empty() {
        return Vec::new();
    }
    let og = [origin as u32];
    let r = run_batch(&[input], &[input.len() as u64], |c, b, _| unsafe { rcx_bwt_inverse_batch(c, b, og.as_ptr()) }).check().unwrap();
    r.out[0].clone()
}

/// bwt/mod.rs:136-166: the sorted suffix array of `input` into `suf_array` (one kernel call; the reference's generic `SUF`
/// is any integer type a suffix index fits: `usize`, `u32`, ... through `TryFrom<u32>`).  Panics, as the reference does, when
/// `suf_array` is shorter than the input (:146) or an index does not fit `SUF` (`NumCast::from(i).unwrap()`).
pub fn compute_suffixes<SUF: TryFrom<u32>>(input: &[u8], suf_array: &mut [SUF]) {
    assert!(suf_array.len() >= input.len());
    let r = run_batch(&[input], &[4 * input.len() as u64], |c, b, o| unsafe { rcx_bwt_suffixes_batch(c, b, o) }).check().unwrap();
    for (dst, w) in suf_array.iter_mut().zip(r.out[0].chunks_exact(4)) {
        *dst = SUF::try_from(le32(w)).ok().unwrap();
    }
}

/// bwt/mod.rs:223-239: the inversion jump table of the transformed block `input` with `origin` into `table`.
pub fn compute_inversion_table<SUF: TryFrom<u32>>(input: &[u8], origin: usize, table: &mut [SUF]) {
    assert_eq!(input.len(), table.len()); // :224
    assert!(origin < input.len()); // input[origin], :230
    let og = [origin as u32];
    let r = run_batch(&[input], &[4 * input.len() as u64], |c, b, _| unsafe { rcx_bwt_inversion_table_batch(c, b, og.as_ptr()) }).check().unwrap();
    for (dst, w) in table.iter_mut().zip(r.out[0].chunks_exact(4)) {
        *dst = SUF::try_from(le32(w)).ok().unwrap();
    }
}

/// bwt/mod.rs:136-210 in batch-backed form: `encode(input, suf)` returns an iterator over the transformed bytes with the
/// origin behind it; here the block is transformed by ONE kernel call and the iterator serves the result (there is no suffix
/// array on the host to hand in: the device sorts in its own scratch).
pub struct TransformIterator {
    bytes: std::vec::IntoIter<u8>,
    origin: usize,
}

impl TransformIterator {
    /// :192-195
    pub fn get_origin(&self) -> usize {
        self.origin
    }
}

impl Iterator for TransformIterator {
    type Item = u8;
    fn next(&mut self) -> Option<u8> {
        self.bytes.next()
    }
}

pub fn encode(input: &[u8]) -> TransformIterator {
    let (l, origin) = encode_simple(input);
    TransformIterator { bytes: l.into_iter(), origin }
}

/// bwt/mod.rs:223-288 in batch-backed form: `decode(input, origin, table)` returns an iterator over the original bytes.
pub struct InverseIterator {
    bytes: std::vec::IntoIter<u8>,
}

impl Iterator for InverseIterator {
    type Item = u8;
    fn next(&mut self) -> Option<u8> {
        self.bytes.next()
    }
}

pub fn decode(input: &[u8], origin: usize) -> InverseIterator {
    InverseIterator { bytes: decode_simple(input, origin).into_iter() }
}

/// bwt/mod.rs:437-518: `u32 LE block_size`, then per block `u32 LE n`, n bytes of L, `u32 LE origin`.
pub struct Encoder<W: Write> {
    w: W,
    buf: Vec<u8>,
    block_size: usize,
    wrote_header: bool,
}

impl<W: Write> Encoder<W> {
    pub fn new(w: W, block_size: usize) -> Encoder<W> {
        Encoder { w, buf: Vec::new(), block_size, wrote_header: false }
    }
    /// Everything buffered since the last flush, cut into blocks of `block_size` (the last one may be shorter), in ONE batch
    /// call -- the blocks the reference writes one by one as its buffer fills (:497-505) plus the partial block its `flush`
    /// encodes (:511-516).
    fn encode_pending(&mut self) -> io::Result<()> {
        if self.buf.is_empty() {
            return Ok(());
        }
        let buf = std::mem::take(&mut self.buf);
        let blocks: Vec<&[u8]> = buf.chunks(self.block_size.max(1)).collect();
        let caps: Vec<u64> = blocks.iter().map(|b| b.len() as u64).collect();
        let r = run_batch(&blocks, &caps, |c, b, o| unsafe { rcx_bwt_forward_batch(c, b, o) }).check()?;
        for i in 0..blocks.len() {
            self.w.write_all(&(blocks[i].len() as u32).to_le_bytes())?;
            self.w.write_all(&r.out[i])?;
            self.w.write_all(&r.aux[i].to_le_bytes())?;
        }
        Ok(())
    }
    /// :485-489: `flush`, then the writer back.
    pub fn finish(mut self) -> (W, io::Result<()>) {
        let res = self.flush();
        (self.w, res)
    }
}

impl<W: Write> Write for Encoder<W> {
    fn write(&mut self, buf: &[u8]) -> io::Result<usize> {
        if !self.wrote_header {
            self.w.write_all(&(self.block_size as u32).to_le_bytes())?; // :493-496
            self.wrote_header = true;
        }
        self.buf.extend_from_slice(buf);
        Ok(0) // the reference's Ok(0) quirk (:507): callers use write_all-free loops, kept for byte-for-byte behaviour
    }
    /// :511-518: the pending (partial) block is encoded as a block of its own, then the writer is flushed -- a caller that
    /// flushes mid-stream gets the block boundary the reference gives it.
    fn flush(&mut self) -> io::Result<()> {
        let ret = self.encode_pending();
        ret.and(self.w.flush())
    }
}

/// bwt/mod.rs:321-432.  `extra_mem = false` selects the reference's `decode_minimal` (:298-315, called at :397-399), which is not
/// the inverse of the encoder in general; it is reproduced as the reference computes it
/// (`rcx_bwt_inverse_minimal_batch`), so this type returns what the reference's returns for either setting.
pub struct Decoder<R: Read> {
    pub r: TailReader<R>,
    buf: Buffered,
    pub extra_memory: bool,
    pub max_block_size: usize,
}

impl<R: Read> Decoder<R> {
    pub fn new(r: R, extra_mem: bool) -> Decoder<R> {
        Decoder { r: TailReader::new(r), buf: Buffered::new(), extra_memory: extra_mem, max_block_size: 0 }
    }
}

impl<R: Read> Read for Decoder<R> {
    fn read(&mut self, dst: &mut [u8]) -> io::Result<usize> {
        let mbs = &mut self.max_block_size;
        let extra = self.extra_memory;
        self.buf.ensure(&mut self.r, |d| {
            let n = d.len();
            let mut p = 0usize;
            if n - p < 4 {
                return Err(eof_error()); // :369
            }
            *mbs = le32(&d[p..]) as usize;
            p += 4;
            let (mut ls, mut origins): (Vec<&[u8]>, Vec<u32>) = (Vec::new(), Vec::new());
            while n - p >= 4 {
                // a clean EOF at a block boundary ends the stream (:374-377)
                let bn = le32(&d[p..]) as usize;
                p += 4;
                if n - p < bn {
                    return Err(eof_error());
                }
                let l = &d[p..p + bn];
                p += bn;
                if n - p < 4 {
                    return Err(eof_error());
                }
                origins.push(le32(&d[p..]));
                p += 4;
                assert!(bn != 0 || !extra, "index out of bounds"); // input[origin] panics (:230); decode_minimal asserts origin == 0 (:300-302)
                ls.push(l);
            }
            if ls.is_empty() {
                return Ok((Vec::new(), None));
            }
            let caps: Vec<u64> = ls.iter().map(|l| l.len() as u64).collect();
            let r = run_batch(&ls, &caps, |c, b, _| unsafe {
                if extra { rcx_bwt_inverse_batch(c, b, origins.as_ptr()) } else { rcx_bwt_inverse_minimal_batch(c, b, origins.as_ptr()) }
            })
            .check()?;
            Ok((r.out.concat(), None)) // the format runs to the reader's end
        })?;
        Ok(self.buf.serve(dst))
    }
}
