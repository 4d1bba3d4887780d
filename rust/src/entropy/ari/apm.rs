//! Adaptive probability maps (reference: src/entropy/ari/apm.rs): `Bit` and `Gate` per decision on the host, and the pairing of
//! src/entropy/ari/test.rs:150-182 as one device call per stream.  `Bit::to_wide` (ln) and `Gate::new` (exp) are f32 libm
//! calls; the device path evaluates their 4096 + 17 possible arguments once on the host and is integer only.
use super::{Border, Model as AriModel};
use crate::rcx_sys::*;
use crate::run_batch;

pub type FlatProbability = u16;
pub type WideProbability = i16;

const BIN_WEIGHT_BITS: usize = 8;
const BIN_WEIGHT_TOTAL: usize = 1 << BIN_WEIGHT_BITS;
const FLAT_TOTAL: isize = 1 << 12;
const WIDE_OFFSET: WideProbability = 1 << 11;
const PORTAL_OFFSET: usize = 8;
const PORTAL_BINS: usize = 2 * PORTAL_OFFSET + 1;

/// num's `ToPrimitive` for f32 (the crate's `.to_i16().unwrap()` / `.to_u16().unwrap()`): None outside the target's range.
fn f32_to_i16(x: f32) -> Option<i16> {
    if x > -32769.0 && x < 32768.0 { Some(x as i16) } else { None }
}
fn f32_to_u16(x: f32) -> Option<u16> {
    if x > -1.0 && x < 65536.0 { Some(x as u16) } else { None }
}

/// apm.rs:36-110: a 12-bit probability of a zero
#[derive(Copy, Clone)]
pub struct Bit(FlatProbability);

impl Bit {
    pub fn new_equal() -> Bit {
        Bit((FLAT_TOTAL >> 1) as FlatProbability)
    }
    pub fn to_flat(&self) -> FlatProbability {
        self.0
    }
    /// the stretched form, ln(p / (1 - p)) in units of 1 / 2048
    pub fn to_wide(&self) -> WideProbability {
        let p = (self.0 as f32) / (FLAT_TOTAL as f32);
        let d = (p / (1.0 - p)).ln();
        f32_to_i16(d * WIDE_OFFSET as f32).unwrap()
    }
    pub fn from_flat(fp: FlatProbability) -> Bit {
        Bit(fp)
    }
    pub fn from_wide(wp: WideProbability) -> Bit {
        let d = (wp as f32) / (WIDE_OFFSET as f32);
        let p = 1.0 / (1.0 + (-d).exp());
        Bit(f32_to_u16(p * FLAT_TOTAL as f32).unwrap())
    }
    pub fn update_zero(&mut self, rate: isize, bias: isize) {
        let one = FLAT_TOTAL - bias - (self.0 as isize);
        self.0 = self.0.wrapping_add((one >> (rate as usize)) as FlatProbability);
    }
    pub fn update_one(&mut self, rate: isize, bias: isize) {
        let zero = (self.0 as isize) - bias;
        self.0 = self.0.wrapping_sub((zero >> (rate as usize)) as FlatProbability);
    }
    pub fn update(&mut self, value: bool, rate: isize, bias: isize) {
        if value {
            self.update_one(rate, bias)
        } else {
            self.update_zero(rate, bias)
        }
    }
}

impl AriModel<bool> for Bit {
    fn get_range(&self, value: bool) -> (Border, Border) {
        let fp = self.0 as Border;
        if value { (fp, FLAT_TOTAL as Border) } else { (0, fp) }
    }
    fn find_value(&self, offset: Border) -> (bool, Border, Border) {
        assert!(offset < FLAT_TOTAL as Border, "Invalid bit offset {} requested", offset);
        let value = offset >= self.0 as Border;
        let (lo, hi) = self.get_range(value);
        (value, lo, hi)
    }
    fn get_denominator(&self) -> Border {
        FLAT_TOTAL as Border
    }
}

pub type BinCoords = (usize, usize); // (index, weight)

/// apm.rs:115-198: 17 bins over the stretched probability, interpolated by the low 8 bits
pub struct Gate {
    map: [Bit; PORTAL_BINS],
}

impl Gate {
    pub fn new() -> Gate {
        let mut map = [Bit::new_equal(); PORTAL_BINS];
        for (i, bit) in map.iter_mut().enumerate() {
            let rp = (i as f32) / (PORTAL_OFFSET as f32) - 1.0;
            *bit = Bit::from_wide(f32_to_i16(rp * WIDE_OFFSET as f32).unwrap());
        }
        Gate { map }
    }
    pub fn pass(&self, bit: &Bit) -> (Bit, BinCoords) {
        let (fp, coords) = self.pass_wide(bit.to_wide());
        (Bit::from_flat(fp), coords)
    }
    pub fn pass_wide(&self, wp: WideProbability) -> (FlatProbability, BinCoords) {
        let index = ((wp + WIDE_OFFSET) >> BIN_WEIGHT_BITS) as usize;
        let weight = wp as usize & (BIN_WEIGHT_TOTAL - 1);
        let (z0, z1) = (self.map[index].to_flat() as usize, self.map[index + 1].to_flat() as usize);
        let sum = z0 * (BIN_WEIGHT_TOTAL - weight) + z1 * weight;
        ((sum >> BIN_WEIGHT_BITS) as FlatProbability, (index, weight))
    }
    pub fn update_zero(&mut self, bc: BinCoords, rate: isize, bias: isize) {
        self.map[bc.0].update_zero(rate, bias);
        self.map[bc.0 + 1].update_zero(rate, bias);
    }
    pub fn update_one(&mut self, bc: BinCoords, rate: isize, bias: isize) {
        self.map[bc.0].update_one(rate, bias);
        self.map[bc.0 + 1].update_one(rate, bias);
    }
    pub fn update(&mut self, value: bool, bc: BinCoords, rate: isize, bias: isize) {
        if value {
            self.update_one(bc, rate, bias)
        } else {
            self.update_zero(bc, rate, bias)
        }
    }
    /// test.rs:150-182 as ONE device call per stream (every bit of every byte through `gate.pass(&bit)`, both updated with
    /// rate 10, bias 0).  A bit history that pushes the index out of the bins panics in the reference (:162-166); the device
    /// reports `RCX_E_MALFORMED` at that decision.
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
