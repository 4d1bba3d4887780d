// k_crc32.hip -- CRC-32 (IEEE 802.3 / RFC 1952 section 8, reflected polynomial 0xEDB88320) over whole blocks.
//
// NOT a function of the reference crate (it has zlib/Adler-32 only, src/zlib.rs, src/checksum/adler.rs): this is the
// SURVEY.md 8(f) rank-3 extension that lets BASELINE config 3 ("gzip members") be taken literally -- the gzip member
// trailer is CRC-32 + ISIZE.  The checker is Python's zlib.crc32 (an independent implementation), see tests.
//
// One wave per block.  The block is cut into 64 equal slices of `per` bytes (the < 64 remaining bytes are appended by
// lane 0 at the end); every lane runs the table-driven byte recurrence over its slice (256-entry table in LDS, 16 input
// bytes per load), and the 64 slice CRCs are merged by a 6-level tree with the GF(2) identity
//     crc(A || B) = crc(A) * x^(8|B|) mod P  xor  crc(B)
// (the pre/post conditioning cancels, as in zlib's crc32_combine).  All right-hand parts of one level have the same
// length, so the level's multiplier is the square of the previous one: one x^(8 per) by square-and-multiply per block.
#include "rcx_dev.h"

#define RCX_CRC_POLY 0xEDB88320u

__device__ __forceinline__ uint32_t rcx_crc_mulmod(uint32_t a, uint32_t b)      // a * b mod P, reflected bit order
{
    uint32_t p = 0;
#pragma unroll 1
    for (int i = 0; i < 32; i++) {
        p ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? RCX_CRC_POLY : 0u);
    }
    return p;
}
__device__ __forceinline__ uint32_t rcx_crc_xpow(uint64_t nbits)                // x^nbits mod P
{
    uint32_t p = 0x80000000u, sq = 0x40000000u;                               // x^0, x^1
    while (nbits) {
        if (nbits & 1) p = rcx_crc_mulmod(sq, p);
        sq = rcx_crc_mulmod(sq, sq);
        nbits >>= 1;
    }
    return p;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_crc32(rcx_kargs a)
{
    __shared__ uint32_t s_tab[256];
    for (unsigned i = threadIdx.x; i < 256; i += 64 * WAVES) {                 // the standard byte table
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? RCX_CRC_POLY : 0u);
        s_tab[i] = c;
    }
    __syncthreads();
    const unsigned w = threadIdx.x >> 6, lane = rcx_lane();
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));
    if (b >= a.nblocks) return;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint64_t n = a.in_len[b];
    const uint64_t per = n / 64;
    const uint8_t* p = in + per * lane;
    uint32_t c = 0xffffffffu;
    uint64_t i = 0;
    for (; i + 16 <= per; i += 16) {
        const rcx_u32x4 v = *(const rcx_u32x4_u*)(p + i);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t x = v[k];
#pragma unroll
            for (int j = 0; j < 4; j++) { c = s_tab[(c ^ x) & 0xffu] ^ (c >> 8); x >>= 8; }
        }
    }
    for (; i < per; i++) c = s_tab[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    c = ~c;                                                                   // CRC-32 of this lane's slice
    // merge: after level d, lane (multiple of 2^(d+1)) holds the CRC of 2^(d+1) consecutive slices
    uint32_t K = RCX_UNI(rcx_crc_xpow(8 * per));
#pragma unroll 1
    for (int d = 0; d < 6; d++) {
        const uint32_t right = (uint32_t)__shfl_down((int)c, 1 << d);
        const uint32_t merged = rcx_crc_mulmod(K, c) ^ right;
        c = (lane & ((2u << d) - 1u)) == 0 ? merged : c;
        K = RCX_UNI(rcx_crc_mulmod(K, K));
    }
    if (lane == 0) {
        uint32_t r = per ? c : 0u;                                            // no slices: the CRC of nothing
        r = ~r;
        for (uint64_t q = 64 * per; q < n; q++) r = s_tab[(r ^ in[q]) & 0xffu] ^ (r >> 8);
        r = ~r;
        if (a.aux) a.aux[b] = r;
        if (a.status) a.status[b] = RCX_OK;
        if (a.out_len) a.out_len[b] = 0;
        if (a.in_used) a.in_used[b] = n;
    }
}

static void launch_crc32(hipStream_t s, rcx_kargs& k)
{
    hipLaunchKernelGGL((k_crc32<4>), dim3((k.nblocks + 3) / 4), dim3(256), 0, s, k);
}
