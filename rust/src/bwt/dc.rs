//! Distance coding (reference: src/bwt/dc.rs:110-252), the `_simple` forms over u32 words.
use crate::rcx_sys::*;
use crate::{le32, run_batch};

/// dc.rs:153-159: 256 x init (first position of each symbol, or n), then the k distances in position order.
pub fn encode_simple(input: &[u8]) -> Vec<u32> {
    let cap = 4 * (256 + input.len() as u64);
    let r = run_batch(&[input], &[cap], |c, b, _| unsafe { rcx_dc_encode_batch(c, b) }).check().unwrap();
    r.out[0].chunks(4).map(le32).collect()
}

/// dc.rs:236-252
pub fn decode_simple(n: usize, distances: &[u32]) -> Vec<u8> {
    let blob: Vec<u8> = distances.iter().flat_map(|d| d.to_le_bytes().to_vec()).collect();
    let nn = [n as u64];
    let r = run_batch(&[&blob[..]], &[n as u64], |c, b, _| unsafe { rcx_dc_decode_batch(c, b, nn.as_ptr()) }).check().unwrap();
    r.out[0].clone()
}

pub type Symbol = u8;
pub type Rank = u8;
pub const TOTAL_SYMBOLS: usize = 0x100;

/// dc.rs:40-58: what a coding model may condition a distance on.
#[derive(PartialEq, Eq, Debug, Clone, Copy)]
pub struct Context {
    pub symbol: Symbol,
    pub last_rank: Rank,
    pub distance_limit: usize,
}

impl Context {
    pub fn new(s: Symbol, r: Rank, dmax: usize) -> Context {
        Context { symbol: s, last_rank: r, distance_limit: dmax }
    }
    fn from_words(w: &[u8]) -> Context {
        Context { symbol: w[0], last_rank: w[1], distance_limit: le32(&w[4..8]) as usize }
    }
}

/// dc.rs:110-149 in batch-backed form: what `encode(input, distances, mtf)` leaves behind -- the initial positions
/// (`EncodeIterator::get_init`, :83-86) and the `(distance, Context)` pairs its iterator yields (:88-103) -- computed by ONE
/// call of the device kernel (rcx_dc_encode_ctx_batch); iterate the vector where the reference iterates the iterator.
pub fn encode(input: &[u8]) -> ([usize; TOTAL_SYMBOLS], Vec<(u32, Context)>) {
    let n = input.len();
    let cap = 4 * (256 + n as u64) + 8 * n as u64;
    let r = run_batch(&[input], &[cap], |c, b, _| unsafe { rcx_dc_encode_ctx_batch(c, b) }).check().unwrap();
    let out = &r.out[0];
    let k = (out.len() - 4 * (256 + n)) / 8;
    let mut init = [0usize; TOTAL_SYMBOLS];
    for s in 0..TOTAL_SYMBOLS {
        init[s] = le32(&out[4 * s..4 * s + 4]) as usize;
    }
    let cb = 4 * (256 + n);
    let pairs = (0..k).map(|j| (le32(&out[4 * (256 + j)..4 * (257 + j)]), Context::from_words(&out[cb + 8 * j..cb + 8 * j + 8]))).collect();
    (init, pairs)
}

/// dc.rs:162-233 in batch-backed form: the block decoded from its initial positions and distances, and the `Context` the
/// reference hands to its distance callback before each distance is read (:208), in call order (rcx_dc_decode_ctx_batch).
pub fn decode(init: &[usize; TOTAL_SYMBOLS], distances: &[u32], n: usize) -> (Vec<u8>, Vec<Context>) {
    let mut blob: Vec<u8> = Vec::with_capacity(4 * (256 + distances.len()));
    for s in 0..TOTAL_SYMBOLS {
        blob.extend_from_slice(&(init[s] as u32).to_le_bytes());
    }
    for d in distances {
        blob.extend_from_slice(&d.to_le_bytes());
    }
    let co = (n + 7) & !7;
    let nn = [n as u64];
    let r = run_batch(&[&blob[..]], &[(co + 8 * distances.len()) as u64], |c, b, _| unsafe { rcx_dc_decode_ctx_batch(c, b, nn.as_ptr()) }).check().unwrap();
    let out = &r.out[0];
    let ctxs = (co..out.len()).step_by(8).map(|o| Context::from_words(&out[o..o + 8])).collect();
    (out[..n].to_vec(), ctxs)
}
