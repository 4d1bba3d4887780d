// tu_inflate.hip -- DEFLATE / zlib / gzip decode, Adler-32 and CRC-32 kernels + their launch code (one translation unit).
#include "rcx_tu.h"
#include "k_lz4_decode_v4.hip"        // Lz4V4 / Lz4V5: the batched sequence executor k_inflate3 feeds
#include "k_lz4_decode_v5.hip"
#include "k_inflate.hip"
#include "k_inflate2.hip"
#include "k_inflate3.hip"
#include "k_crc32.hip"
#include "k_gzip.hip"

void rcx_tu_inflate(hipStream_t s, rcx_kargs& k, bool zlib, int variant) { launch_inflate(s, k, zlib, variant); }
void rcx_tu_adler32(hipStream_t s, rcx_kargs& k) { launch_adler32(s, k); }
void rcx_tu_crc32(hipStream_t s, rcx_kargs& k) { launch_crc32(s, k); }
void rcx_tu_gzip_decode(hipStream_t s, rcx_kargs& k, int variant) { launch_gzip_decode(s, k, variant); }
uint64_t rcx_tu_inflate_scratch(uint32_t nblocks) { return inflate_scratch_bytes(nblocks); }
bool rcx_tu_inflate_mirrors(uint32_t nblocks, int variant) { return inflate_mirrors(nblocks, variant); }
uint64_t rcx_tu_inflate_marks_offset(uint32_t nblocks) { return inflate_marks_offset(nblocks); }
uint64_t rcx_tu_gzip_scratch(uint32_t nblocks) { return gzip_scratch_bytes(nblocks); }
uint64_t rcx_tu_gzip_marks_offset(uint32_t nblocks) { return ((gzip_scratch_bytes(nblocks) + 255) & ~255ull) + inflate_marks_offset(nblocks); }   // (launch_gzip_decode: the inflate path's own scratch behind the gzip arrays)
