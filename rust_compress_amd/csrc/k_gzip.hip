// k_gzip.hip -- gzip (RFC 1952) member framing around the DEFLATE kernel: header parse, CRC-32 + ISIZE trailer check.
//
// NOT in the reference crate (it parses RFC 1950 only, src/zlib.rs:55-126, which is the model for the structure of
// this file): SURVEY.md 8(f) rank 3, so that BASELINE config 3 ("gzip members") can be run literally.  One member
// per block; the whole decode stays on the device:
//   k_gzip_head  one lane per member: checks ID1 ID2 CM FLG, skips MTIME XFL OS, FEXTRA, FNAME, FCOMMENT, FHCRC;
//                writes shifted (offset, length) descriptors of the raw DEFLATE payload into scratch
//   k_inflate2   the production inflate kernel over those descriptors (k_inflate2.hip)
//   k_crc32      CRC-32 of each decoded block (k_crc32.hip)
//   k_gzip_tail  compares CRC32 / ISIZE (little endian, after the DEFLATE stream), merges the statuses
// FHCRC (CRC-16 of the header) is skipped, not verified: zlib's inflate does the same unless asked, and the checker
// used by the tests (Python's gzip module) ignores it too.
#include "rcx_dev.h"

struct GzipScratch {                 // carved from rcx_kargs::scratch, n entries each
    uint64_t* off2; uint64_t* len2; uint64_t* used2; int32_t* st_head; int32_t* st_infl; uint32_t* hdr; uint32_t* crc; uint32_t* fl;
};
__host__ __device__ static inline GzipScratch gzip_carve(void* scratch, uint32_t n)
{
    GzipScratch g;
    uint8_t* p = (uint8_t*)scratch;
    g.off2 = (uint64_t*)p; p += 8ull * n;
    g.len2 = (uint64_t*)p; p += 8ull * n;
    g.used2 = (uint64_t*)p; p += 8ull * n;
    g.st_head = (int32_t*)p; p += 4ull * n;
    g.st_infl = (int32_t*)p; p += 4ull * n;
    g.hdr = (uint32_t*)p; p += 4ull * n;
    g.crc = (uint32_t*)p; p += 4ull * n;
    g.fl = (uint32_t*)p;
    return g;
}
static uint64_t gzip_scratch_bytes(uint32_t nblocks) { return 48ull * nblocks + 256; }

__global__ void k_gzip_head(rcx_kargs a, GzipScratch g)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks) return;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint64_t n = a.in_len[b];
    int st = RCX_OK;
    uint64_t p = 10;
    if (n < 10) { st = RCX_E_EOF; p = n; }
    else if (in[0] != 0x1f || in[1] != 0x8b) st = RCX_E_GZIP_MAGIC;
    else if (in[2] != 8) st = RCX_E_GZIP_METHOD;
    else if (in[3] & 0xe0) st = RCX_E_GZIP_FLAGS;                      // reserved FLG bits must be zero
    else {
        const uint32_t flg = in[3];
        if (flg & 4) {                                                 // FEXTRA: u16 LE length + that many bytes
            if (n - p < 2) { st = RCX_E_EOF; p = n; }
            else {
                const uint64_t xl = (uint64_t)in[p] | ((uint64_t)in[p + 1] << 8);
                p += 2;
                if (n - p < xl) { st = RCX_E_EOF; p = n; } else p += xl;
            }
        }
        for (int which = 0; which < 2 && !st; which++) {               // FNAME, FCOMMENT: zero-terminated
            if (!(flg & (which ? 16u : 8u))) continue;
            for (;;) {
                if (p >= n) { st = RCX_E_EOF; break; }
                if (in[p++] == 0) break;
            }
        }
        if (!st && (flg & 2)) { if (n - p < 2) { st = RCX_E_EOF; p = n; } else p += 2; }   // FHCRC: skipped
    }
    g.st_head[b] = st;
    g.hdr[b] = (uint32_t)p;
    g.off2[b] = a.in_off[b] + (st ? 0 : p);
    g.len2[b] = st ? 0 : n - p;
}

__global__ void k_gzip_tail(rcx_kargs a, GzipScratch g)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks) return;
    int st = g.st_head[b];
    uint64_t used = g.hdr[b];
    if (st) { a.out_len[b] = 0; }
    else {
        st = g.st_infl[b];
        used += g.used2[b];
        if (!st) {
            const uint8_t* in = a.in_base + a.in_off[b];
            const uint64_t n = a.in_len[b];
            if (n - used < 8) st = RCX_E_EOF;
            else {
                const uint8_t* t = in + used;
                const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                const uint32_t isz = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                used += 8;
                if (crc != g.crc[b]) st = RCX_E_GZIP_CRC;
                else if (isz != (uint32_t)a.out_len[b]) st = RCX_E_GZIP_ISIZE;
            }
        }
    }
    a.status[b] = st;
    if (a.in_used) a.in_used[b] = used;
    if (a.aux) a.aux[b] = st ? 0u : g.fl[b];
}

// the whole member decode: head -> inflate -> crc32 -> tail (k.scratch holds gzip_scratch_bytes(n))
static void launch_gzip_decode(hipStream_t s, rcx_kargs& k, int v)
{
    const uint32_t n = k.nblocks;
    const GzipScratch g = gzip_carve(k.scratch, n);
    hipLaunchKernelGGL(k_gzip_head, dim3((n + 255) / 256), dim3(256), 0, s, k, g);
    rcx_kargs ki = k;                                      // the raw DEFLATE payloads
    ki.in_off = g.off2; ki.in_len = g.len2; ki.in_used = g.used2; ki.status = g.st_infl; ki.aux = g.fl;
    ki.scratch = (uint8_t*)k.scratch + ((gzip_scratch_bytes(n) + 255) & ~255ull);                     // the inflate path's own scratch
    ki.scratch_bytes = k.scratch_bytes > gzip_scratch_bytes(n) + 256 ? k.scratch_bytes - gzip_scratch_bytes(n) - 256 : 0;
    launch_inflate(s, ki, false, v);
    rcx_kargs kc = k;                                      // CRC-32 of what was decoded
    kc.in_base = k.out_base; kc.in_off = k.out_off; kc.in_len = k.out_len; kc.out_len = nullptr; kc.in_used = nullptr;
    kc.status = nullptr; kc.aux = g.crc;
    launch_crc32(s, kc);
    hipLaunchKernelGGL(k_gzip_tail, dim3((n + 255) / 256), dim3(256), 0, s, k, g);
}
