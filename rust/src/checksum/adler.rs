//! Adler-32 (reference: src/checksum/adler.rs:22-51); the sum is computed on the device when `result()` is asked for.
use crate::rcx_sys::*;
use crate::run_batch;

pub struct State32 {
    data: Vec<u8>,
}

impl State32 {
    pub fn new() -> State32 {
        State32 { data: Vec::new() }
    }
    pub fn feed(&mut self, buf: &[u8]) {
        self.data.extend_from_slice(buf)
    }
    pub fn result(&self) -> u32 {
        run_batch(&[&self.data[..]], &[0], |c, b, a| unsafe { rcx_adler32_batch(c, b, a) }).aux[0]
    }
    pub fn reset(&mut self) {
        self.data.clear()
    }
}
