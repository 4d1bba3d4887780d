//! The frequency-table models (reference: src/entropy/ari/table.rs) in batch-backed form.
use crate::rcx_sys::*;
use crate::run_batch;

/// table.rs:20-122: the adaptive 257-symbol table `ByteEncoder` / `ByteDecoder` code with (the device keeps it in LDS, one per
/// stream, with exact integer block sums).  A marker type: the coding itself is `ari::ByteEncoder` / `ari::ByteDecoder`.
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct Model;

/// table.rs:127-180 `SumProxy` (two 16-entry tables mixed 2:1 >> 0, updates 10 / 5) for the high nibble of every byte, with
/// `bin::SumProxy` (bin.rs:112-167) for the four low bits: the pairing of src/entropy/ari/test.rs:91-148, a byte stream per call.
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct SumProxy;

impl SumProxy {
    pub fn encode_bytes(bytes: &[u8]) -> Vec<u8> {
        let cap = unsafe { rcx_ari_byte_encode_bound(bytes.len() as u64) };
        let r = run_batch(&[bytes], &[cap], |c, b, _| unsafe { rcx_ari_proxy_encode_batch(c, b) }).check().unwrap();
        r.out[0].clone()
    }
    /// No end marker: `n` bytes are produced.
    pub fn decode_bytes(coded: &[u8], n: usize) -> std::io::Result<Vec<u8>> {
        let r = run_batch(&[coded], &[n as u64], |c, b, _| unsafe { rcx_ari_proxy_decode_batch(c, b) }).check()?;
        Ok(r.out[0].clone())
    }
}
