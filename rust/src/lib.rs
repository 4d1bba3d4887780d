//! `compress` 0.2.1 (rusty-shell/rust-compress) with its block kernels on an MI355X: the crate's public stream types keep
//! their names and signatures and become thin shims over the C-ABI of include/rcx.h -- buffer the stream, parse the framing
//! on the host, make ONE FFI call per batch of blocks, serve `read()` / `write()` from the result.
//!
//! Transcribed from the tested C++ twin (rust_compress_amd/host/compress.hpp, run against the reference's own tests by
//! tests/test_gpu_cpp_host.py); this crate itself cannot be compiled in the build image (no Rust toolchain).
//! There is no CPU fallback: without a HIP device `ctx()` panics.
pub mod rcx_sys;

// the reference's feature gates (src/lib.rs:19-50), module for module
#[cfg(feature = "bwt")]
pub mod bwt;
#[cfg(feature = "checksum")]
pub mod checksum;
#[cfg(feature = "entropy")]
pub mod entropy;
#[cfg(feature = "flate")]
pub mod flate;
#[cfg(feature = "lz4")]
pub mod lz4;
#[cfg(feature = "rle")]
pub mod rle;
#[cfg(feature = "zlib")]
pub mod zlib;

#[cfg(feature = "checksum")]
pub use checksum::adler::State32 as Adler32; // lib.rs:21

use rcx_sys::*;
use std::ffi::CStr;
use std::io::{self, Read};

/// One `rcx_ctx` per thread (the C library is thread-compatible, not thread-safe: include/rcx.h).
pub(crate) fn ctx() -> *mut rcx_ctx {
    thread_local! {
        static CTX: *mut rcx_ctx = {
            let mut h: *mut rcx_ctx = std::ptr::null_mut();
            let rc = unsafe { rcx_ctx_create(-1, &mut h) };
            assert!(rc == RCX_RC_OK, "rcx_ctx_create failed (no HIP device? there is no CPU fallback)");
            h
        };
    }
    CTX.with(|c| *c)
}

/// Per-block status -> the `io::Error` the reference returns (kinds and texts: flate.rs:56-65, zlib.rs:59-84,111-114,
/// lz4.rs:366,376, rle.rs:153, lib.rs:53-62,115-118).  Inputs on which the reference panics panic here too.
pub(crate) fn status_to_io(st: i32) -> io::Error {
    let msg = unsafe { CStr::from_ptr(rcx_status_string(st)) }.to_string_lossy().into_owned();
    match st {
        RCX_E_EOF => io::Error::new(io::ErrorKind::Other, "unexpected end of file"),
        RCX_E_MALFORMED | RCX_E_OUTPUT_TOO_SMALL => panic!("{}", msg),
        RCX_E_RLE_LONG_RUN => io::Error::new(io::ErrorKind::Other, msg),
        _ => io::Error::new(io::ErrorKind::InvalidInput, msg),
    }
}

/// Result of one batch call over host blobs.
pub(crate) struct BatchResult {
    pub out: Vec<Vec<u8>>,
    pub in_used: Vec<u64>,
    pub status: Vec<i32>,
    pub aux: Vec<u32>,
}

impl BatchResult {
    pub fn check(self) -> io::Result<BatchResult> {
        for &st in &self.status {
            if st != RCX_OK {
                return Err(status_to_io(st));
            }
        }
        Ok(self)
    }
}

/// Pack `blobs` into one host buffer (16-byte aligned blocks), call a batch entry point, unpack (compress.hpp run_batch).
pub(crate) fn run_batch<F>(blobs: &[&[u8]], caps: &[u64], call: F) -> BatchResult
where
    F: FnOnce(*mut rcx_ctx, *const rcx_batch, *mut u32) -> i32,
{
    let n = blobs.len();
    let (mut in_off, mut in_len, mut out_off) = (vec![0u64; n], vec![0u64; n], vec![0u64; n]);
    let (mut it, mut ot) = (0u64, 0u64);
    for i in 0..n {
        in_off[i] = it;
        in_len[i] = blobs[i].len() as u64;
        it += (blobs[i].len() as u64 + 15) & !15;
        out_off[i] = ot;
        ot += (caps[i] + 15) & !15;
    }
    let mut inp = vec![0u8; it as usize + 16];
    let mut out = vec![0u8; ot as usize + 16];
    for i in 0..n {
        inp[in_off[i] as usize..in_off[i] as usize + blobs[i].len()].copy_from_slice(blobs[i]);
    }
    let (mut out_len, mut in_used, mut status, mut aux) = (vec![0u64; n], vec![0u64; n], vec![0i32; n], vec![0u32; n]);
    let b = rcx_batch {
        in_base: inp.as_ptr(),
        in_off: in_off.as_ptr(),
        in_len: in_len.as_ptr(),
        out_base: out.as_mut_ptr(),
        out_off: out_off.as_ptr(),
        out_cap: caps.as_ptr(),
        out_len: out_len.as_mut_ptr(),
        in_used: in_used.as_mut_ptr(),
        status: status.as_mut_ptr(),
        nblocks: n as u32,
        mem: RCX_MEM_HOST,
    };
    let rc = call(ctx(), &b, aux.as_mut_ptr());
    assert!(rc == RCX_RC_OK, "rcx batch call failed: {}", unsafe { CStr::from_ptr(rcx_last_error(ctx())) }.to_string_lossy());
    let outs = (0..n).map(|i| out[out_off[i] as usize..(out_off[i] + out_len[i]) as usize].to_vec()).collect();
    BatchResult { out: outs, in_used, status, aux }
}

/// Decode one blob with output slots that grow 8x until it fits.  A kernel stops at a full slot, so the failed attempts
/// together cost a seventh of the one that fits (compress.hpp detail::grow_decode).
pub(crate) fn grow_decode<F>(d: &[u8], first_cap: u64, call: F) -> io::Result<BatchResult>
where
    F: Fn(*mut rcx_ctx, *const rcx_batch, *mut u32) -> i32,
{
    // the last attempt is the largest slot a block may have (the kernels index a block with 32 bits; run_batch rejects a
    // larger slot with RCX_RC_BAD_ARG): past it the block's own RCX_E_OUTPUT_TOO_SMALL is the answer, not a batch-level failure
    const MAX_BLOCK: u64 = 0xFFFF_FFFF;
    let mut cap = first_cap.max(1 << 16).min(MAX_BLOCK);
    loop {
        let r = run_batch(&[d], &[cap], &call);
        if r.status[0] == RCX_E_OUTPUT_TOO_SMALL && cap < MAX_BLOCK {
            cap = (cap * 8).min(MAX_BLOCK);
            continue;
        }
        return r.check();
    }
}

/// Several blobs of one kind through ONE batch call; the slots that were too small once more, eight times larger, together
/// (compress.hpp detail::decode_many).  A single stream alone on the GPU takes longer than on one host thread, a batch of eight or
/// more does not: a caller with many `Decoder`s uses the codec's `decode_many` instead of reading them one by one.
/// -> per blob (decoded bytes, input bytes used, aux word); the first blob that failed returns its Decoder's error.
pub(crate) fn decode_many_with<F>(blobs: &[&[u8]], call: F) -> io::Result<Vec<(Vec<u8>, usize, u32)>>
where
    F: Fn(*mut rcx_ctx, *const rcx_batch, *mut u32) -> i32,
{
    const MAX_BLOCK: u64 = 0xFFFF_FFFF;
    let n = blobs.len();
    let mut res: Vec<(Vec<u8>, usize, u32)> = vec![(Vec::new(), 0, 0); n];
    let mut status = vec![RCX_OK; n];
    if n == 0 {
        return Ok(res);
    }
    let mut cap = blobs.iter().map(|b| 4 * b.len() as u64).max().unwrap_or(0).max(1 << 16).min(MAX_BLOCK);
    let mut idx: Vec<usize> = (0..n).collect();
    loop {
        let part: Vec<&[u8]> = idx.iter().map(|&i| blobs[i]).collect();
        let caps = vec![cap; idx.len()];
        let r = run_batch(&part, &caps, &call);
        let mut redo = Vec::new();
        for (j, &i) in idx.iter().enumerate() {
            status[i] = r.status[j];
            if r.status[j] == RCX_E_OUTPUT_TOO_SMALL && cap < MAX_BLOCK {
                redo.push(i);
            } else {
                res[i] = (r.out[j].clone(), r.in_used[j] as usize, r.aux[j]);
            }
        }
        if redo.is_empty() {
            break;
        }
        idx = redo;
        cap = (cap * 8).min(MAX_BLOCK);
    }
    for &st in &status {
        if st != RCX_OK {
            return Err(status_to_io(st));
        }
    }
    Ok(res)
}

/// A reader that takes bytes back.  The batch decoders have to read ahead (a stream's end is only known once it is decoded);
/// the reference's decoders stop reading exactly at the end of their stream (flate.rs:250-260 reads byte by byte,
/// ari/mod.rs:289-292 `finish`) and its tests rely on the reader being left there (ari/test.rs:52-89).  Every Decoder keeps
/// its reader as `TailReader<R>` and hands the bytes behind its stream back, so `decoder.r`, `finish()` and `unwrap()` are a
/// reader positioned exactly after the stream.
pub struct TailReader<R: Read> {
    pub inner: R,
    tail: Vec<u8>,
    tpos: usize,
}

/// `decoder.r` is a `TailReader<R>` where the reference has an `R` (flate.rs:166, lz4.rs:320, bwt/mod.rs:325): it derefs to
/// the `R`, so call sites that use the reader's own methods (`decoder.r.get_ref()`, `decoder.r.position()`) keep compiling;
/// code that READS from it must go through the `TailReader` (its `Read` impl serves the handed-back bytes first), and
/// `into_inner()` gives the `R` back once those are used up.
impl<R: Read> std::ops::Deref for TailReader<R> {
    type Target = R;
    fn deref(&self) -> &R {
        &self.inner
    }
}

impl<R: Read> std::ops::DerefMut for TailReader<R> {
    fn deref_mut(&mut self) -> &mut R {
        &mut self.inner
    }
}

impl<R: Read> TailReader<R> {
    pub fn new(r: R) -> TailReader<R> {
        TailReader { inner: r, tail: Vec::new(), tpos: 0 }
    }
    /// The wrapped reader and the bytes handed back but not yet read again (empty once the caller has drained them).
    pub fn into_inner(self) -> (R, Vec<u8>) {
        let rest = self.tail[self.tpos..].to_vec();
        (self.inner, rest)
    }
    pub fn unread(&mut self, data: &[u8]) {
        if data.is_empty() {
            return;
        }
        let mut t = data.to_vec();
        t.extend_from_slice(&self.tail[self.tpos..]);
        self.tail = t;
        self.tpos = 0;
    }
}

impl<R: Read> Read for TailReader<R> {
    fn read(&mut self, dst: &mut [u8]) -> io::Result<usize> {
        if self.tpos < self.tail.len() {
            let k = dst.len().min(self.tail.len() - self.tpos);
            dst[..k].copy_from_slice(&self.tail[self.tpos..self.tpos + k]);
            self.tpos += k;
            return Ok(k);
        }
        self.inner.read(dst)
    }
}

/// Common part of every buffered decoder: decode everything on first read, then serve chunks of whatever size the caller asks
/// for (the reference's `read()` is chunk-size independent: flate.rs:551-582, lz4.rs:675-706).
pub(crate) struct Buffered {
    pub out: Vec<u8>,
    pub pos: usize,
    pub done: bool,
}

impl Buffered {
    pub fn new() -> Buffered {
        Buffered { out: Vec::new(), pos: 0, done: false }
    }
    /// `decode(raw) -> (decoded, consumed)`; `consumed = None`: the format runs to the reader's end.
    pub fn ensure<R: Read, F>(&mut self, r: &mut TailReader<R>, decode: F) -> io::Result<()>
    where
        F: FnOnce(&[u8]) -> io::Result<(Vec<u8>, Option<usize>)>,
    {
        if self.done {
            return Ok(());
        }
        let mut raw = Vec::new();
        r.read_to_end(&mut raw)?;
        let (out, consumed) = decode(&raw)?;
        if let Some(c) = consumed {
            r.unread(&raw[c..]);
        }
        self.out = out;
        self.pos = 0;
        self.done = true;
        Ok(())
    }
    pub fn serve(&mut self, dst: &mut [u8]) -> usize {
        let k = dst.len().min(self.out.len() - self.pos);
        dst[..k].copy_from_slice(&self.out[self.pos..self.pos + k]);
        self.pos += k;
        k
    }
    pub fn eof(&self) -> bool {
        self.done && self.pos == self.out.len()
    }
    pub fn reset(&mut self) {
        self.out.clear();
        self.pos = 0;
        self.done = false;
    }
}

pub(crate) fn le32(p: &[u8]) -> u32 {
    (p[0] as u32) | ((p[1] as u32) << 8) | ((p[2] as u32) << 16) | ((p[3] as u32) << 24)
}

pub(crate) fn eof_error() -> io::Error {
    io::Error::new(io::ErrorKind::Other, "unexpected end of file") // lib.rs:115-118
}
