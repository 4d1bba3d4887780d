/*
 * oracle.h -- CPU restatement of rust-compress's block codecs.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity CHECKER for the HIP path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (rust_compress_amd/csrc/librcx.so) never links, loads or calls it.
 *
 * Every function restates the algorithm of the Rust reference line by line
 * (no algorithmic upgrades: bit-serial Huffman walk, byte-loop copies,
 * linear-scan frequency table, comparison-sorted suffixes, pointer-chase
 * inverse BWT), citing the reference file:line it follows.  Where the
 * reference panics or reads uninitialised memory the oracle returns
 * RCX_E_MALFORMED; where it grows a Vec the oracle returns
 * RCX_E_OUTPUT_TOO_SMALL if the caller's buffer is too small.
 *
 * Pinning: the Rust reference cannot be built here (no cargo/rustc), so the
 * oracle is pinned by the reference's own fixtures and known-answer tests
 * (tests/golden/: test.txt <-> test.z.*, test.lz4.*, RLE KATs), by Python's
 * zlib as an independent RFC-1951/1950 implementation, and by the reference's
 * round-trip properties.  Paths with no reference-owned vector (lz4 encode
 * bytes, BWT/MTF/DC/Ari bytes) are "parity pinned by restatement + round trip
 * only" -- see DESIGN.md.
 */
#ifndef RCX_ORACLE_H
#define RCX_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/rcx.h"   /* enum rcx_status only */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- lz4.rs ---- */
int      o_lz4_decode_block(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);
int      o_lz4_encode_block(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);
uint64_t o_lz4_compression_bound(uint64_t n);
int      o_lz4_frame_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, size_t* in_used);
int      o_lz4_frame_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);

/* ---- flate.rs / zlib.rs / checksum/adler.rs ---- */
int      o_inflate(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, size_t* in_used, uint32_t* flags);
int      o_zlib_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, size_t* in_used, uint32_t* flags);
uint32_t o_adler32(const uint8_t* buf, size_t n);
uint32_t o_adler32_feed(uint32_t state /* (b<<16)|a, start 1 */, const uint8_t* buf, size_t n);

/* ---- bwt/mod.rs ---- */
int o_bwt_compute_suffixes(const uint8_t* in, size_t n, uint32_t* sa);
int o_bwt_encode(const uint8_t* in, size_t n, uint8_t* L, uint32_t* origin);   /* encode_simple */
int o_bwt_inversion_table(const uint8_t* L, size_t n, uint32_t origin, uint32_t* table);
int o_bwt_decode(const uint8_t* L, size_t n, uint32_t origin, uint8_t* out);   /* decode_simple */
int o_bwt_decode_minimal(const uint8_t* L, size_t n, uint32_t origin, uint8_t* out);
int o_bwt_stream_encode(const uint8_t* in, size_t n, uint32_t block_size, uint8_t* out, size_t cap, size_t* out_len);
int o_bwt_stream_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);

/* ---- bwt/mtf.rs ---- */
void o_mtf_encode(const uint8_t* in, size_t n, uint8_t* out);   /* mtf::Encoder (identity start) */
void o_mtf_decode(const uint8_t* in, size_t n, uint8_t* out);   /* mtf::Decoder */

/* ---- bwt/dc.rs ---- */
typedef struct { uint8_t symbol; uint8_t last_rank; uint32_t distance_limit; } o_dc_context;
/* encode_simple::<u32>: words[0..256) = init, then *k distances; ctx (optional) gets *k entries */
int o_dc_encode(const uint8_t* in, size_t n, uint32_t* words, size_t cap_words, size_t* nwords, o_dc_context* ctx);
/* decode_simple: consumed (optional) = number of distances read; ctx (optional, cap >= n+1) */
int o_dc_decode(const uint32_t* words, size_t nwords, size_t n, uint8_t* out, size_t* consumed, o_dc_context* ctx);

/* ---- entropy/ari ---- */
int      o_ari_byte_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);
int      o_ari_byte_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, size_t* in_used);
uint64_t o_ari_byte_encode_bound(uint64_t n);
/* test.rs:22-50 binary model coder (bits LSB-first per byte), threshold>>3, given rate */
int      o_ari_binary_encode(const uint8_t* in, size_t n, uint32_t rate, uint8_t* out, size_t cap, size_t* out_len);
int      o_ari_binary_decode(const uint8_t* in, size_t n, uint32_t rate, uint8_t* out, size_t nbytes);
/* test.rs:91-148 proxy coder (table SumProxy for the high nibble, binary SumProxy for the low bits) */
int      o_ari_proxy_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);
int      o_ari_proxy_decode(const uint8_t* in, size_t n, uint8_t* out, size_t nbytes);
/* apm::Bit + apm::Gate as src/entropy/ari/test.rs:150-182 drives them (f32 ln/exp through this host's libm) */
int      o_ari_apm_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);
int      o_ari_apm_decode(const uint8_t* in, size_t n, uint8_t* out, size_t nbytes);
void     o_apm_tables(int16_t* stretch, uint16_t* gate);

/* ---- rle.rs ---- */
int      o_rle_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);
int      o_rle_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);
uint64_t o_rle_encode_bound(uint64_t n);

/* ---- batch drivers for the timed CPU baseline (threads = std::thread-like pthreads) ---- */
/* codec: enum rcx_codec value; aux as in rcx_dev_batch. Returns wall seconds. */
double o_batch_run(int codec, const uint8_t* in_base, const uint64_t* in_off, const uint64_t* in_len,
                   uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap,
                   uint64_t* out_len, uint64_t* in_used, int32_t* status, uint32_t* aux,
                   const uint64_t* n_out, uint32_t nblocks, int threads);

#ifdef __cplusplus
}
#endif
#endif
