//! The frequency-table models (reference: src/entropy/ari/table.rs): `Model` and `SumProxy` per symbol on the host, and the
//! whole-stream forms on the device.
use super::{Border, Model as AriModel};
use crate::rcx_sys::*;
use crate::run_batch;

pub type Frequency = u16;

/// table.rs:20-122: value -> frequency, with the running sum; halved (rounding up, so nothing reaches zero) whenever the sum
/// reaches `cut_threshold`.  `ari::ByteEncoder` / `ByteDecoder` run this model with 257 values on the device (LDS table per
/// stream); here it is the per-symbol form.
pub struct Model {
    total: Border,
    table: Vec<Frequency>,
    cut_threshold: Border,
    cut_shift: usize,
}

impl Model {
    pub fn new_custom<F: FnMut(usize) -> Frequency>(num_values: usize, threshold: Border, fn_init: F) -> Model {
        let table: Vec<Frequency> = (0..num_values).map(fn_init).collect();
        let total = table.iter().map(|&f| f as Border).sum();
        let mut m = Model { total, table, cut_threshold: threshold, cut_shift: 1 };
        while m.total >= threshold {
            m.downscale();
        }
        m
    }
    pub fn new_flat(num_values: usize, threshold: Border) -> Model {
        Model::new_custom(num_values, threshold, |_| 1)
    }
    pub fn reset_flat(&mut self) {
        self.table.iter_mut().for_each(|f| *f = 1);
        self.total = self.table.len() as Border;
    }
    /// `value` gains (total >> add_log) + add_const
    pub fn update(&mut self, value: usize, add_log: usize, add_const: Border) {
        let add = (self.total >> add_log) + add_const;
        assert!(add < 2 * self.cut_threshold);
        self.table[value] = self.table[value].wrapping_add(add as Frequency);
        self.total += add;
        if self.total >= self.cut_threshold {
            self.downscale();
            assert!(self.total < self.cut_threshold);
        }
    }
    pub fn downscale(&mut self) {
        let roundup = ((1u32 << self.cut_shift) - 1) as Frequency;
        let shift = self.cut_shift;
        self.total = 0;
        for f in self.table.iter_mut() {
            *f = f.wrapping_add(roundup) >> shift;
            self.total += *f as Border;
        }
    }
    pub fn get_frequencies(&self) -> &[Frequency] {
        &self.table[..]
    }
}

impl AriModel<usize> for Model {
    fn get_range(&self, value: usize) -> (Border, Border) {
        let lo: Border = self.table[..value].iter().map(|&f| f as Border).sum();
        (lo, lo + self.table[value] as Border)
    }
    fn find_value(&self, offset: Border) -> (usize, Border, Border) {
        assert!(offset < self.total, "Invalid frequency offset {} requested under total {}", offset, self.total);
        let (mut value, mut lo) = (0usize, 0 as Border);
        loop {
            let hi = lo + self.table[value] as Border;
            if hi > offset {
                return (value, lo, hi);
            }
            lo = hi;
            value += 1;
        }
    }
    fn get_denominator(&self) -> Border {
        self.total
    }
}

/// table.rs:127-180: (wa * A + wb * B) >> ws over two tables of one size
pub struct SumProxy<'a> {
    first: &'a Model,
    second: &'a Model,
    w_first: Border,
    w_second: Border,
    w_shift: Border,
}

impl<'a> SumProxy<'a> {
    pub fn new(wa: Border, fa: &'a Model, wb: Border, fb: &'a Model, shift: Border) -> SumProxy<'a> {
        assert_eq!(fa.get_frequencies().len(), fb.get_frequencies().len());
        SumProxy { first: fa, second: fb, w_first: wa, w_second: wb, w_shift: shift }
    }
    fn mix(&self, a: Border, b: Border) -> Border {
        (self.w_first * a + self.w_second * b) >> (self.w_shift as usize)
    }
    /// The pairing of src/entropy/ari/test.rs:91-148 as ONE device call per stream: the high nibble of every byte through
    /// `SumProxy::new(2, t0, 1, t1, 0)` (updates 10 / 5), the four low bits through `bin::SumProxy::new(1, b0, 1, b1, 1)`
    /// (rates 3 / 5).
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

impl<'a> AriModel<usize> for SumProxy<'a> {
    fn get_range(&self, value: usize) -> (Border, Border) {
        let (lo0, hi0) = self.first.get_range(value);
        let (lo1, hi1) = self.second.get_range(value);
        (self.mix(lo0, lo1), self.mix(hi0, hi1))
    }
    fn find_value(&self, offset: Border) -> (usize, Border, Border) {
        let total = self.get_denominator();
        assert!(offset < total, "Invalid frequency offset {} requested under total {}", offset, total);
        let (fa, fb) = (self.first.get_frequencies(), self.second.get_frequencies());
        let (mut value, mut lo) = (0usize, 0 as Border);
        loop {
            let hi = lo + self.mix(fa[value] as Border, fb[value] as Border);
            if hi > offset {
                return (value, lo, hi);
            }
            lo = hi;
            value += 1;
        }
    }
    fn get_denominator(&self) -> Border {
        self.mix(self.first.get_denominator(), self.second.get_denominator())
    }
}
