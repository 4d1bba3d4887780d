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
