// rcx_tu.h -- host-side entry points of the per-codec translation units (tu_*.hip).  librcx.so is built from several
// TUs so that hipcc compiles them in parallel; each TU holds its kernels and the launch code next to them.
#pragma once
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <hip/hip_runtime.h>
#include <string>
#include "rcx_dev.h"

// tu_lz4.hip
int rcx_tu_lz4_decode(hipStream_t s, rcx_kargs& k, int variant, std::string& err);
int rcx_tu_lz4_encode(hipStream_t s, rcx_kargs& k, int variant, std::string& err);
void rcx_tu_lz4_decode_mirror_again(hipStream_t s, rcx_kargs& k);
uint64_t rcx_tu_lz4_encode_scratch(uint32_t nblocks);
// tu_inflate.hip
void rcx_tu_inflate(hipStream_t s, rcx_kargs& k, bool zlib, int variant);
void rcx_tu_adler32(hipStream_t s, rcx_kargs& k);
void rcx_tu_crc32(hipStream_t s, rcx_kargs& k);
void rcx_tu_gzip_decode(hipStream_t s, rcx_kargs& k, int variant);
uint64_t rcx_tu_inflate_scratch(uint32_t nblocks);
bool rcx_tu_inflate_mirrors(uint32_t nblocks, int variant);     // would the launch store into a page-locked output buffer itself?
uint64_t rcx_tu_inflate_marks_offset(uint32_t nblocks);      // a mirrored launch: [count | 60 bytes | a byte per stream the first pass handed back]
uint64_t rcx_tu_gzip_scratch(uint32_t nblocks);
uint64_t rcx_tu_gzip_marks_offset(uint32_t nblocks);
// tu_bwt.hip
int rcx_tu_bwt_forward(hipStream_t s, rcx_kargs& k, int variant, std::string& err, bool sa_words);
int rcx_tu_bwt_inversion_table(hipStream_t s, rcx_kargs& k);
int rcx_tu_bwt_inverse(hipStream_t s, rcx_kargs& k, int variant, std::string& err, bool minimal);
uint64_t rcx_tu_bwt_forward_scratch(uint32_t nblocks, uint64_t max_block);
uint64_t rcx_tu_bwt_inverse_scratch(uint32_t nblocks, uint64_t max_block);
// tu_serial.hip
void rcx_tu_serial(hipStream_t s, int codec, rcx_kargs& k, int variant, uint32_t param);
uint64_t rcx_tu_dc_encode_scratch(uint32_t nblocks, uint64_t max_block);
