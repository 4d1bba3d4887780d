//! The binary models (reference: src/entropy/ari/bin.rs): per decision on the host (`ari::Encoder::encode(bit, &model)`), and a
//! whole byte stream per device call (what src/entropy/ari/test.rs:22-50 does with them).
use super::{Border, Model as AriModel};
use crate::rcx_sys::*;
use crate::run_batch;

/// bin.rs:17-103: the frequency of a zero out of a constant total, moved towards the coded value by 1 / 2^rate of the gap.
pub struct Model {
    zero: Border,
    total: Border,
    pub rate: Border,
}

impl Model {
    pub fn new_flat(threshold: Border, rate: Border) -> Model {
        Model { zero: threshold >> 1, total: threshold, rate }
    }
    pub fn new_custom(zero_percent: u8, threshold: Border, rate: Border) -> Model {
        assert!(threshold >= 100);
        Model { zero: (zero_percent as Border) * threshold / 100, total: threshold, rate }
    }
    pub fn reset_flat(&mut self) {
        self.zero = self.total >> 1;
    }
    pub fn get_probability_zero(&self) -> Border {
        self.zero
    }
    pub fn get_probability_one(&self) -> Border {
        self.total - self.zero
    }
    pub fn update_zero(&mut self) {
        self.zero += (self.total - self.zero) >> (self.rate as usize);
    }
    pub fn update_one(&mut self) {
        self.zero -= self.zero >> (self.rate as usize);
    }
    pub fn update(&mut self, value: bool) {
        if value {
            self.update_one()
        } else {
            self.update_zero()
        }
    }
    /// test.rs:22-38 as ONE device call: every byte as eight decisions, least significant bit first, under
    /// `Model::new_flat(RANGE_DEFAULT_THRESHOLD >> 3, rate)` (the threshold the kernel is built for; rate 1..31).
    pub fn encode_bytes(rate: Border, bytes: &[u8]) -> Vec<u8> {
        assert!(rate >= 1 && rate <= 31, "bin::Model: rate must be 1..31");
        let cap = unsafe { rcx_ari_byte_encode_bound(bytes.len() as u64) };
        let r = run_batch(&[bytes], &[cap], |c, b, _| unsafe { rcx_ari_binary_encode_batch(c, b, rate) }).check().unwrap();
        r.out[0].clone()
    }
    /// test.rs:39-50: the coding has no end marker; `n` is the number of bytes to produce.
    pub fn decode_bytes(rate: Border, coded: &[u8], n: usize) -> std::io::Result<Vec<u8>> {
        let r = run_batch(&[coded], &[n as u64], |c, b, _| unsafe { rcx_ari_binary_decode_batch(c, b, rate) }).check()?;
        Ok(r.out[0].clone())
    }
}

fn split(zero: Border, total: Border, value: bool) -> (Border, Border) {
    if value {
        (zero, total)
    } else {
        (0, zero)
    }
}

impl AriModel<bool> for Model {
    fn get_range(&self, value: bool) -> (Border, Border) {
        split(self.zero, self.total, value)
    }
    fn find_value(&self, offset: Border) -> (bool, Border, Border) {
        assert!(offset < self.total, "Invalid frequency offset {} requested under total {}", offset, self.total);
        let value = offset >= self.zero;
        let (lo, hi) = split(self.zero, self.total, value);
        (value, lo, hi)
    }
    fn get_denominator(&self) -> Border {
        self.total
    }
}

/// bin.rs:112-167: (wa * A + wb * B) >> ws of two binary models
pub struct SumProxy<'a> {
    first: &'a Model,
    second: &'a Model,
    w_first: Border,
    w_second: Border,
    w_shift: Border,
}

impl<'a> SumProxy<'a> {
    pub fn new(wa: Border, first: &'a Model, wb: Border, second: &'a Model, shift: Border) -> SumProxy<'a> {
        SumProxy { first, second, w_first: wa, w_second: wb, w_shift: shift }
    }
    fn mix(&self, a: Border, b: Border) -> Border {
        (self.w_first * a + self.w_second * b) >> (self.w_shift as usize)
    }
    fn get_probability_zero(&self) -> Border {
        self.mix(self.first.get_probability_zero(), self.second.get_probability_zero())
    }
}

impl<'a> AriModel<bool> for SumProxy<'a> {
    fn get_range(&self, value: bool) -> (Border, Border) {
        split(self.get_probability_zero(), self.get_denominator(), value)
    }
    fn find_value(&self, offset: Border) -> (bool, Border, Border) {
        let (zero, total) = (self.get_probability_zero(), self.get_denominator());
        assert!(offset < total, "Invalid frequency offset {} requested under total {}", offset, total);
        let value = offset >= zero;
        let (lo, hi) = split(zero, total, value);
        (value, lo, hi)
    }
    fn get_denominator(&self) -> Border {
        self.mix(self.first.get_denominator(), self.second.get_denominator())
    }
}
