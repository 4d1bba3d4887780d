//! Adaptive probability maps (reference: src/entropy/ari/apm.rs) in batch-backed form: `apm::Bit` refined through `apm::Gate`,
//! driven as src/entropy/ari/test.rs:150-182 drives them, a byte stream per call.  `Bit::to_wide` (ln) and `Gate::new` (exp)
//! have 4096 + 17 possible arguments: the library evaluates them once on the host with libm's logf / expf -- what Rust's
//! f32::ln / exp call -- and the device code is integer only.
use crate::rcx_sys::*;
use crate::run_batch;

/// apm.rs:36-110: a 12-bit probability with a shift-based update.
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct Bit;

/// apm.rs:115-198: 17 interpolation bins over the stretched probability.  A bit history skewed enough to push the index out
/// of the bins makes the reference panic on a slice bound (:162-166); the device reports `RCX_E_MALFORMED` at that decision.
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct Gate;

impl Gate {
    pub fn encode_bytes(bytes: &[u8]) -> std::io::Result<Vec<u8>> {
        let cap = unsafe { rcx_ari_byte_encode_bound(bytes.len() as u64) };
        let r = run_batch(&[bytes], &[cap], |c, b, _| unsafe { rcx_ari_apm_encode_batch(c, b) }).check()?;
        Ok(r.out[0].clone())
    }
    pub fn decode_bytes(coded: &[u8], n: usize) -> std::io::Result<Vec<u8>> {
        let r = run_batch(&[coded], &[n as u64], |c, b, _| unsafe { rcx_ari_apm_decode_batch(c, b) }).check()?;
        Ok(r.out[0].clone())
    }
}
