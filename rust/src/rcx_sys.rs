//! Raw FFI over include/rcx.h -- one declaration per export, same order as the header.
//! tests/test_rust_shim.py parses both files and fails if a symbol, an argument count or a struct field differs.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

/// Opaque `struct rcx_ctx`.
#[repr(C)]
pub struct rcx_ctx {
    _private: [u8; 0],
}
/// `struct rcx_multi` (opaque): one context per listed device.
#[repr(C)]
pub struct rcx_multi {
    _private: [u8; 0],
}

/// `struct rcx_batch`: struct-of-arrays batch descriptor, host arrays (include/rcx.h).
#[repr(C)]
pub struct rcx_batch {
    pub in_base: *const u8,
    pub in_off: *const u64,
    pub in_len: *const u64,
    pub out_base: *mut u8,
    pub out_off: *const u64,
    pub out_cap: *const u64,
    pub out_len: *mut u64,
    pub in_used: *mut u64,
    pub status: *mut i32,
    pub nblocks: u32,
    pub mem: c_int,
}

/// `struct rcx_dev_batch`: every array already in HBM.
#[repr(C)]
pub struct rcx_dev_batch {
    pub in_base: *const u8,
    pub in_off: *const u64,
    pub in_len: *const u64,
    pub out_base: *mut u8,
    pub out_off: *const u64,
    pub out_cap: *const u64,
    pub out_len: *mut u64,
    pub in_used: *mut u64,
    pub status: *mut i32,
    pub aux: *mut u32,
    pub nblocks: u32,
}

// enum rcx_status
pub const RCX_OK: i32 = 0;
pub const RCX_E_EOF: i32 = 1;
pub const RCX_E_OUTPUT_TOO_SMALL: i32 = 2;
pub const RCX_E_MALFORMED: i32 = 3;
pub const RCX_E_HUFFMAN_TREE_TOO_LARGE: i32 = 10;
pub const RCX_E_INVALID_BLOCK_CODE: i32 = 11;
pub const RCX_E_INVALID_HUFFMAN_HEADER_SYMBOL: i32 = 12;
pub const RCX_E_INVALID_HUFFMAN_TREE: i32 = 13;
pub const RCX_E_INVALID_HUFFMAN_TREE_HEADER: i32 = 14;
pub const RCX_E_INVALID_HUFFMAN_CODE: i32 = 15;
pub const RCX_E_INVALID_STATIC_SIZE: i32 = 16;
pub const RCX_E_NOT_ENOUGH_BITS: i32 = 17;
pub const RCX_E_ZLIB_FORMAT: i32 = 20;
pub const RCX_E_ZLIB_WINDOW: i32 = 21;
pub const RCX_E_ZLIB_DICT: i32 = 22;
pub const RCX_E_ZLIB_HEADER_CHECKSUM: i32 = 23;
pub const RCX_E_ZLIB_CHECKSUM: i32 = 24;
pub const RCX_E_RLE_LONG_RUN: i32 = 30;
pub const RCX_E_LZ4_MAGIC: i32 = 40;
pub const RCX_E_LZ4_VERSION: i32 = 41;
pub const RCX_E_LZ4_INPUT_TOO_LARGE: i32 = 42;
pub const RCX_E_BWT_BLOCK_TOO_LARGE: i32 = 60; // a block of 2^28 bytes or more: this implementation's limit (bwt/mod.rs:451 takes any usize)
pub const RCX_E_GZIP_MAGIC: i32 = 50;
pub const RCX_E_GZIP_METHOD: i32 = 51;
pub const RCX_E_GZIP_FLAGS: i32 = 52;
pub const RCX_E_GZIP_CRC: i32 = 53;
pub const RCX_E_GZIP_ISIZE: i32 = 54;
// enum rcx_rc
pub const RCX_RC_OK: c_int = 0;
pub const RCX_RC_BAD_ARG: c_int = -1;
pub const RCX_RC_NO_DEVICE: c_int = -2;
pub const RCX_RC_HIP_ERROR: c_int = -3;
pub const RCX_RC_NO_MEMORY: c_int = -4;
// enum rcx_mem
pub const RCX_MEM_HOST: c_int = 0;
pub const RCX_MEM_DEVICE: c_int = 1;
pub const RCX_W_EMPTY_BLOCK_MIDSTREAM: u32 = 1;
// enum rcx_codec
pub const RCX_LZ4_DECODE: c_int = 0;
pub const RCX_LZ4_ENCODE: c_int = 1;
pub const RCX_INFLATE: c_int = 2;
pub const RCX_ZLIB_DECODE: c_int = 3;
pub const RCX_ADLER32: c_int = 4;
pub const RCX_BWT_FORWARD: c_int = 5;
pub const RCX_BWT_INVERSE: c_int = 6;
pub const RCX_MTF_ENCODE: c_int = 7;
pub const RCX_MTF_DECODE: c_int = 8;
pub const RCX_DC_ENCODE: c_int = 9;
pub const RCX_DC_DECODE: c_int = 10;
pub const RCX_ARI_BYTE_ENCODE: c_int = 11;
pub const RCX_ARI_BYTE_DECODE: c_int = 12;
pub const RCX_RLE_ENCODE: c_int = 13;
pub const RCX_RLE_DECODE: c_int = 14;
pub const RCX_CRC32: c_int = 15;
pub const RCX_GZIP_DECODE: c_int = 16;
pub const RCX_ARI_BINARY_ENCODE: c_int = 17;
pub const RCX_ARI_BINARY_DECODE: c_int = 18;
pub const RCX_ARI_PROXY_ENCODE: c_int = 19;
pub const RCX_ARI_PROXY_DECODE: c_int = 20;
pub const RCX_ARI_APM_ENCODE: c_int = 21;
pub const RCX_ARI_APM_DECODE: c_int = 22;
pub const RCX_BWT_INVERSE_MINIMAL: c_int = 23;
pub const RCX_BWT_SUFFIXES: c_int = 24;
pub const RCX_BWT_INVERSION_TABLE: c_int = 25;
pub const RCX_CODEC_COUNT: c_int = 26;

#[link(name = "rcx")]
extern "C" {
    // ---- context
    pub fn rcx_ctx_create(device_id: c_int, out: *mut *mut rcx_ctx) -> c_int;
    pub fn rcx_ctx_destroy(ctx: *mut rcx_ctx);
    pub fn rcx_ctx_set_stream(ctx: *mut rcx_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn rcx_last_error(ctx: *const rcx_ctx) -> *const c_char;
    pub fn rcx_status_string(status: c_int) -> *const c_char;
    pub fn rcx_version() -> c_int;
    // ---- LZ4 (src/lz4.rs:602-627, 175-181)
    pub fn rcx_lz4_decode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_lz4_encode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_lz4_compression_bound(in_len: u64) -> u64;
    // ---- DEFLATE / zlib / Adler-32 (src/flate.rs, src/zlib.rs, src/checksum/adler.rs) + the gzip extension
    pub fn rcx_inflate_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, flags: *mut u32) -> c_int;
    pub fn rcx_zlib_decode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, flags: *mut u32) -> c_int;
    pub fn rcx_adler32_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, adler: *mut u32) -> c_int;
    pub fn rcx_crc32_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, crc: *mut u32) -> c_int;
    pub fn rcx_gzip_decode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, flags: *mut u32) -> c_int;
    // ---- BWT / MTF / DC (src/bwt/mod.rs, mtf.rs, dc.rs)
    pub fn rcx_bwt_forward_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, origin: *mut u32) -> c_int;
    pub fn rcx_bwt_suffixes_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, origin: *mut u32) -> c_int;
    pub fn rcx_bwt_inversion_table_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, origin: *const u32) -> c_int;
    pub fn rcx_bwt_inverse_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, origin: *const u32) -> c_int;
    pub fn rcx_bwt_inverse_minimal_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, origin: *const u32) -> c_int;
    pub fn rcx_mtf_encode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_mtf_decode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_dc_encode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_dc_decode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, n_out: *const u64) -> c_int;
    pub fn rcx_dc_encode_ctx_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_dc_decode_ctx_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, n_out: *const u64) -> c_int;
    // ---- range coders (src/entropy/ari/*.rs)
    pub fn rcx_ari_byte_encode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_ari_byte_decode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_ari_byte_encode_bound(in_len: u64) -> u64;
    pub fn rcx_ari_binary_encode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, rate: u32) -> c_int;
    pub fn rcx_ari_binary_decode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch, rate: u32) -> c_int;
    pub fn rcx_ari_proxy_encode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_ari_proxy_decode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_ari_apm_encode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_ari_apm_decode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    // ---- RLE (src/rle.rs)
    pub fn rcx_rle_encode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_rle_decode_batch(ctx: *mut rcx_ctx, b: *const rcx_batch) -> c_int;
    pub fn rcx_rle_encode_bound(in_len: u64) -> u64;
    // ---- device-resident batches
    pub fn rcx_scratch_bytes(codec: c_int, nblocks: u32, max_block: u64) -> u64;
    pub fn rcx_launch_dev(ctx: *mut rcx_ctx, codec: c_int, b: *const rcx_dev_batch, scratch: *mut c_void, scratch_bytes: u64) -> c_int;
    pub fn rcx_ctx_set_variant(ctx: *mut rcx_ctx, codec: c_int, variant: c_int) -> c_int;
    pub fn rcx_ctx_set_param(ctx: *mut rcx_ctx, codec: c_int, value: u32) -> c_int;
    // ---- more than one device: contiguous block ranges, one context per device, no collective (include/rcx.h)
    pub fn rcx_multi_create(device_ids: *const c_int, n: c_int, out: *mut *mut rcx_multi) -> c_int;
    pub fn rcx_multi_destroy(m: *mut rcx_multi);
    pub fn rcx_multi_count(m: *const rcx_multi) -> c_int;
    pub fn rcx_multi_ctx(m: *mut rcx_multi, i: c_int) -> *mut rcx_ctx;
    pub fn rcx_partition(weights: *const u64, nblocks: u32, parts: u32, bounds: *mut u32);
    pub fn rcx_multi_batch(m: *mut rcx_multi, codec: c_int, b: *const rcx_batch, aux_in: *const u32, aux_out: *mut u32, n_out: *const u64) -> c_int;
    pub fn rcx_multi_launch_dev(m: *mut rcx_multi, codec: c_int, per_device: *const *const rcx_dev_batch, scratch: *const *mut c_void, scratch_bytes: *const u64) -> c_int;
    pub fn rcx_multi_sync(m: *mut rcx_multi) -> c_int;
    pub fn rcx_multi_last_error(m: *const rcx_multi) -> *const c_char;
    // the batch on ONE device of the set: ranges out to their devices and the results back, device to device (RCCL, or peer copies)
    pub fn rcx_multi_scatter_dev(m: *mut rcx_multi, root: c_int, root_buf: *const u8, range_off: *const u64, peer_buf: *const *mut u8) -> c_int;
    pub fn rcx_multi_gather_dev(m: *mut rcx_multi, root: c_int, root_buf: *mut u8, range_off: *const u64, peer_buf: *const *const u8) -> c_int;
    pub fn rcx_multi_transport(m: *const rcx_multi) -> *const c_char;
    // ---- page-locked host memory: what lets rcx_lz4_decode_batch write the caller's buffer itself (include/rcx.h)
    pub fn rcx_host_register(ptr: *mut c_void, bytes: u64) -> c_int;
    pub fn rcx_host_unregister(ptr: *mut c_void) -> c_int;
    // ---- measurement aid: the device's own copy rate (GB/s, read + written)
    pub fn rcx_hbm_copy_probe(ctx: *mut rcx_ctx, bytes: u64, reps: c_int, gb_per_s: *mut f64) -> c_int;
}
