//! The binary models (reference: src/entropy/ari/bin.rs) in batch-backed form.  The crate codes one decision at a time through
//! `ari::Encoder::encode(bit, &model)`; no stream codec of the crate uses these models (only src/entropy/ari/test.rs drives
//! them), so what the device offers is what those tests do, a whole byte stream per call.
use crate::rcx_sys::*;
use crate::run_batch;

/// bin.rs:17-103: a two-symbol frequency model with exponential update; `rate` = the shift of the update (1..31).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct Model {
    pub rate: u32,
}

impl Model {
    /// bin.rs:34-43 `new_flat(threshold, rate)`: the device kernel uses the threshold the reference's test uses
    /// (`RANGE_DEFAULT_THRESHOLD >> 3`, test.rs:24).
    pub fn new_flat(rate: u32) -> Model {
        assert!(rate >= 1 && rate <= 31, "bin::Model: rate must be 1..31");
        Model { rate }
    }
    /// test.rs:22-38: every byte as eight decisions, least significant bit first, one adaptive model.
    pub fn encode_bytes(&self, bytes: &[u8]) -> Vec<u8> {
        let cap = unsafe { rcx_ari_byte_encode_bound(bytes.len() as u64) };
        let rate = self.rate;
        let r = run_batch(&[bytes], &[cap], |c, b, _| unsafe { rcx_ari_binary_encode_batch(c, b, rate) }).check().unwrap();
        r.out[0].clone()
    }
    /// test.rs:39-50: the coding has no end marker; `n` is the number of bytes to produce.
    pub fn decode_bytes(&self, coded: &[u8], n: usize) -> std::io::Result<Vec<u8>> {
        let rate = self.rate;
        let r = run_batch(&[coded], &[n as u64], |c, b, _| unsafe { rcx_ari_binary_decode_batch(c, b, rate) }).check()?;
        Ok(r.out[0].clone())
    }
}

/// bin.rs:112-167: two binary models mixed 1:1 (>> 1), rates 3 and 5 -- the low-nibble half of `table::SumProxy::code_bytes`.
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct SumProxy;
