/* o_batch.c -- threaded batch driver over the oracle codecs, used ONLY for the timed CPU
 * baseline in bench.py and for bulk parity checks in tests (TEST INFRASTRUCTURE). */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <time.h>
#include "oracle.h"

typedef struct {
    int codec;
    const uint8_t* in_base; const uint64_t* in_off; const uint64_t* in_len;
    uint8_t* out_base; const uint64_t* out_off; const uint64_t* out_cap;
    uint64_t* out_len; uint64_t* in_used; int32_t* status; uint32_t* aux; const uint64_t* n_out;
    uint32_t nblocks;
    volatile uint32_t next;
} job_t;

static void run_one(job_t* j, uint32_t i)
{
    const uint8_t* in = j->in_base + j->in_off[i];
    size_t n = (size_t)j->in_len[i];
    uint8_t* out = j->out_base ? j->out_base + j->out_off[i] : NULL;
    size_t cap = j->out_cap ? (size_t)j->out_cap[i] : 0;
    size_t olen = 0, used = n;
    uint32_t fl = 0;
    int st = RCX_OK;
    switch (j->codec) {
    case RCX_LZ4_DECODE: st = o_lz4_decode_block(in, n, out, cap, &olen); break;
    case RCX_LZ4_ENCODE: st = o_lz4_encode_block(in, n, out, cap, &olen); break;
    case RCX_INFLATE: st = o_inflate(in, n, out, cap, &olen, &used, &fl); if (j->aux) j->aux[i] = fl; break;
    case RCX_ZLIB_DECODE: st = o_zlib_decode(in, n, out, cap, &olen, &used, &fl); if (j->aux) j->aux[i] = fl; break;
    case RCX_ADLER32: if (j->aux) j->aux[i] = o_adler32(in, n); break;
    case RCX_BWT_FORWARD: {
        uint32_t origin = 0;
        if (cap < n) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
        st = o_bwt_encode(in, n, out, &origin); olen = n;
        if (j->aux) j->aux[i] = origin;
        break; }
    case RCX_BWT_SUFFIXES: {
        if (cap < 4 * n) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
        st = o_bwt_compute_suffixes(in, n, (uint32_t*)out); olen = 4 * n;
        if (j->aux) { j->aux[i] = 0; for (size_t q = 0; q < n; q++) if (((uint32_t*)out)[q] == 0) j->aux[i] = (uint32_t)q; }
        break; }
    case RCX_BWT_INVERSION_TABLE: {
        const uint32_t origin = j->aux ? j->aux[i] : 0;
        if (origin >= n) { st = RCX_E_MALFORMED; break; }
        if (cap < 4 * n) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
        st = o_bwt_inversion_table(in, n, origin, (uint32_t*)out); olen = st ? 0 : 4 * n; break; }
    case RCX_BWT_INVERSE:
        if (cap < n) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
        if (n == 0) { olen = 0; break; }
        st = o_bwt_decode(in, n, j->aux ? j->aux[i] : 0, out); olen = n; break;
    case RCX_BWT_INVERSE_MINIMAL: {
        const uint32_t origin = j->aux ? j->aux[i] : 0;
        if (cap < n && origin < n) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
        st = o_bwt_decode_minimal(in, n, origin, out); olen = st ? 0 : n; break; }
    case RCX_MTF_ENCODE: if (cap < n) { st = RCX_E_OUTPUT_TOO_SMALL; break; } o_mtf_encode(in, n, out); olen = n; break;
    case RCX_MTF_DECODE: if (cap < n) { st = RCX_E_OUTPUT_TOO_SMALL; break; } o_mtf_decode(in, n, out); olen = n; break;
    case RCX_DC_ENCODE: {
        size_t nw = 0;
        st = o_dc_encode(in, n, (uint32_t*)out, cap / 4, &nw, NULL); olen = nw * 4; break; }
    case RCX_DC_DECODE: {
        size_t nn = j->n_out ? (size_t)j->n_out[i] : 0, cons = 0;
        if (cap < nn) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
        st = o_dc_decode((const uint32_t*)in, n / 4, nn, out, &cons, NULL); olen = nn; used = 4 * (256 + cons); break; }
    case RCX_ARI_BYTE_ENCODE: st = o_ari_byte_encode(in, n, out, cap, &olen); break;
    case RCX_ARI_BYTE_DECODE: st = o_ari_byte_decode(in, n, out, cap, &olen, &used); break;
    case RCX_RLE_ENCODE: st = o_rle_encode(in, n, out, cap, &olen); break;
    case RCX_RLE_DECODE: st = o_rle_decode(in, n, out, cap, &olen); break;
    default: st = RCX_E_MALFORMED;
    }
    if (j->out_len) j->out_len[i] = olen;
    if (j->in_used) j->in_used[i] = used;
    if (j->status) j->status[i] = st;
}

static void* worker(void* v)
{
    job_t* j = (job_t*)v;
    for (;;) {
        uint32_t i = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (i >= j->nblocks) break;
        run_one(j, i);
    }
    return NULL;
}

double o_batch_run(int codec, const uint8_t* in_base, const uint64_t* in_off, const uint64_t* in_len,
                   uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap,
                   uint64_t* out_len, uint64_t* in_used, int32_t* status, uint32_t* aux,
                   const uint64_t* n_out, uint32_t nblocks, int threads)
{
    job_t j = {codec, in_base, in_off, in_len, out_base, out_off, out_cap, out_len, in_used, status, aux, n_out, nblocks, 0};
    struct timespec t0, t1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (threads == 1) {
        worker(&j);
    } else {
        pthread_t th[256];
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, &j);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
