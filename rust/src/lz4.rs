//! LZ4 frame reader / writer and the pure block functions (reference: src/lz4.rs).
use crate::rcx_sys::*;
use crate::{eof_error, grow_decode, le32, run_batch, Buffered, TailReader};
use std::io::{self, Read, Write};

const MAGIC: u32 = 0x184d2204;

/// lz4.rs:175-181
pub fn compression_bound(size: u32) -> Option<u32> {
    match unsafe { rcx_lz4_compression_bound(size as u64) } {
        0 => None,
        v => Some(v as u32),
    }
}

/// lz4.rs:602-611: appends the decoded block to `output`, returns its length.  Panics where the reference panics.
pub fn decode_block(input: &[u8], output: &mut Vec<u8>) -> usize {
    let r = grow_decode(input, 8 * input.len() as u64, |c, b, _| unsafe { rcx_lz4_decode_batch(c, b) }).unwrap();
    output.extend_from_slice(&r.out[0]);
    r.out[0].len()
}

/// lz4.rs:616-627: 0 when `compression_bound` is None.
pub fn encode_block(input: &[u8], output: &mut Vec<u8>) -> usize {
    let cap = unsafe { rcx_lz4_compression_bound(input.len() as u64) } + 1;
    let r = run_batch(&[input], &[cap], |c, b, _| unsafe { rcx_lz4_encode_batch(c, b) });
    if r.status[0] == RCX_E_LZ4_INPUT_TOO_LARGE {
        return 0;
    }
    let r = r.check().unwrap();
    output.extend_from_slice(&r.out[0]);
    r.out[0].len()
}

/// lz4.rs:316-500.  `r` is left exactly behind the frame's end mark (the content checksum is never read, :384).
pub struct Decoder<R: Read> {
    pub r: TailReader<R>,
    buf: Buffered,
    pub max_block_size: usize,
}

impl<R: Read> Decoder<R> {
    pub fn new(r: R) -> Decoder<R> {
        Decoder { r: TailReader::new(r), buf: Buffered::new(), max_block_size: 0 }
    }
    pub fn reset(&mut self) {
        self.buf.reset()
    }
    pub fn eof(&mut self) -> bool {
        self.buf.eof()
    }
}

/// the host framing of lz4.rs:316-500: -> ((stored?, payload) parts, bytes the frame used)
fn parse_frame<'a>(d: &'a [u8], max_block_size: &mut usize) -> io::Result<(Vec<(bool, &'a [u8])>, usize)> {
    let n = d.len();
    let mut p = 0usize;
    if n - p < 4 {
        return Err(eof_error());
    }
    if le32(&d[p..]) != MAGIC {
        return Err(io::Error::new(io::ErrorKind::InvalidInput, "")); // :365-367
    }
    p += 4;
    let flg = if p < n { d[p] } else { 0 }; // :369-372, a short read is tolerated
    let bd = if p + 1 < n { d[p + 1] } else { 0 };
    p = (p + 2).min(n);
    if (flg >> 6) != 1 {
        return Err(io::Error::new(io::ErrorKind::InvalidInput, "")); // :375-377
    }
    let (blk_checksum, stream_size, preset) = (flg & 0x10 != 0, flg & 0x08 != 0, flg & 0x01 != 0);
    const MAXS: [usize; 8] = [0, 0, 0, 0, 64 << 10, 256 << 10, 1 << 20, 4 << 20];
    *max_block_size = MAXS[((bd >> 4) & 7) as usize];
    if stream_size {
        if n - p < 8 {
            return Err(eof_error());
        }
        p += 8;
    }
    assert!(!preset, "preset dictionaries not supported yet"); // :407
    if p >= n {
        return Err(eof_error());
    }
    p += 1; // header checksum, ignored (:417)
    let mut parts: Vec<(bool, &[u8])> = Vec::new();
    loop {
        if n - p < 4 {
            return Err(eof_error());
        }
        let v = le32(&d[p..]);
        p += 4;
        if v == 0 {
            break;
        }
        let amt = (v & 0x7fff_ffff) as usize;
        if n - p < amt {
            return Err(eof_error());
        }
        parts.push((v & 0x8000_0000 != 0, &d[p..p + amt]));
        p += amt;
        if blk_checksum {
            if n - p < 4 {
                return Err(eof_error());
            }
            p += 4;
        }
    }
    Ok((parts, p))
}

/// ONE batch call for every compressed block of one or MANY frames; a conforming frame's blocks decode to at most max_block_size
/// bytes, and only a block that does not fit is decoded again with a larger slot (the reference would grow its Vec, :148-161)
fn decode_frames(frames: &[(Vec<(bool, &[u8])>, usize)]) -> io::Result<Vec<Vec<u8>>> {
    let mut comp: Vec<&[u8]> = Vec::new();
    let mut caps: Vec<u64> = Vec::new();
    for (parts, mbs) in frames {
        for part in parts.iter().filter(|p| !p.0) {
            comp.push(part.1);
            caps.push((*mbs).max(1 << 16) as u64);
        }
    }
    let mut outs: Vec<Vec<u8>> = Vec::new();
    if !comp.is_empty() {
        let mut r = run_batch(&comp, &caps, |c, b, _| unsafe { rcx_lz4_decode_batch(c, b) });
        for i in 0..comp.len() {
            if r.status[i] == RCX_E_OUTPUT_TOO_SMALL {
                let mut o = Vec::new();
                decode_block(comp[i], &mut o);
                r.out[i] = o;
                r.status[i] = RCX_OK;
            }
        }
        outs = r.check()?.out;
    }
    let mut ci = 0;
    let mut res = Vec::with_capacity(frames.len());
    for (parts, _) in frames {
        let mut out = Vec::new();
        for (stored, data) in parts {
            if *stored {
                out.extend_from_slice(data);
            } else {
                out.extend_from_slice(&outs[ci]);
                ci += 1;
            }
        }
        res.push(out);
    }
    Ok(res)
}

fn decode_frame(d: &[u8], max_block_size: &mut usize) -> io::Result<(Vec<u8>, Option<usize>)> {
    let (parts, used) = parse_frame(d, max_block_size)?;
    let mut outs = decode_frames(&[(parts, *max_block_size)])?;
    Ok((outs.remove(0), Some(used)))
}

/// Many frames, EVERY compressed block of EVERY frame in one batch call (one 64 KiB block alone on the GPU takes six times what one
/// host thread needs; eight or more together take less: INTEGRATION.md).  -> per frame (decoded bytes, bytes of the input it used).
pub fn decode_many(frames: &[&[u8]]) -> io::Result<Vec<(Vec<u8>, usize)>> {
    let mut parsed = Vec::with_capacity(frames.len());
    let mut used = Vec::with_capacity(frames.len());
    for d in frames {
        let mut mbs = 0usize;
        let (parts, u) = parse_frame(d, &mut mbs)?;
        parsed.push((parts, mbs));
        used.push(u);
    }
    Ok(decode_frames(&parsed)?.into_iter().zip(used).collect())
}

impl<R: Read> Read for Decoder<R> {
    fn read(&mut self, dst: &mut [u8]) -> io::Result<usize> {
        let mbs = &mut self.max_block_size;
        self.buf.ensure(&mut self.r, |raw| decode_frame(raw, mbs))?;
        Ok(self.buf.serve(dst))
    }
}

/// lz4.rs:505-597: blocks of <= 256 KiB, always stored (`compress()` returns false, :543-545).
pub struct Encoder<W: Write> {
    w: W,
    buf: Vec<u8>,
    wrote_header: bool,
    limit: usize,
}

impl<W: Write> Encoder<W> {
    pub fn new(w: W) -> Encoder<W> {
        Encoder { w, buf: Vec::with_capacity(1024), wrote_header: false, limit: 256 * 1024 }
    }
    fn encode_block(&mut self) -> io::Result<()> {
        let v = self.buf.len() as u32 | 0x8000_0000; // :536
        self.w.write_all(&v.to_le_bytes())?;
        self.w.write_all(&self.buf)?;
        self.buf.clear();
        Ok(())
    }
    /// :550-561: writes the end mark (two zero u32) and returns the writer.
    pub fn finish(mut self) -> (W, io::Result<()>) {
        let mut res = self.flush();
        if res.is_ok() {
            res = self.w.write_all(&[0u8; 8]);
        }
        (self.w, res)
    }
}

impl<W: Write> Write for Encoder<W> {
    fn write(&mut self, mut buf: &[u8]) -> io::Result<usize> {
        if !self.wrote_header {
            self.w.write_all(&[0x04, 0x22, 0x4d, 0x18, 0x60, 0x50, 0x00])?; // :567-574
            self.wrote_header = true;
        }
        while !buf.is_empty() {
            let amt = (self.limit - self.buf.len()).min(buf.len());
            self.buf.extend_from_slice(&buf[..amt]);
            if self.buf.len() == self.limit {
                self.encode_block()?;
            }
            buf = &buf[amt..];
        }
        Ok(buf.len()) // the reference returns the length of the EMPTIED slice (:588), i.e. 0
    }
    fn flush(&mut self) -> io::Result<()> {
        if !self.buf.is_empty() {
            self.encode_block()?;
        }
        self.w.flush()
    }
}
