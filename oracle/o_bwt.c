/* o_bwt.c -- CPU restatement of src/bwt/mod.rs, src/bwt/mtf.rs, src/bwt/dc.rs
 * (TEST INFRASTRUCTURE, see oracle.h). */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define ALPHABET_SIZE 256

/* ---------- Radix, bwt/mod.rs:73-130 ---------- */
typedef struct { size_t freq[ALPHABET_SIZE + 1]; } radix_t;
static void radix_gather(radix_t* r, const uint8_t* in, size_t n)      /* :95-99 */
{
    memset(r, 0, sizeof(*r));
    for (size_t i = 0; i < n; i++) r->freq[in[i]]++;
}
static void radix_accumulate(radix_t* r)                                /* :102-109 */
{
    size_t n = 0;
    for (int i = 0; i <= ALPHABET_SIZE; i++) { size_t f = r->freq[i]; r->freq[i] = n; n += f; }
}
static void radix_shift(radix_t* r)                                     /* :123-129 */
{
    for (int i = ALPHABET_SIZE - 1; i >= 0; i--) r->freq[i + 1] = r->freq[i];
    r->freq[0] = 0;
}

/* the comparator of :160-162: input[a..].cmp(&input[b..]) -- lexicographic on byte
 * slices, a proper prefix sorts first (implicit sentinel smaller than every byte) */
typedef struct { const uint8_t* in; size_t n; } cmp_ctx;
static int suf_cmp(const void* pa, const void* pb, void* vc)
{
    const cmp_ctx* c = (const cmp_ctx*)vc;
    uint32_t a = *(const uint32_t*)pa, b = *(const uint32_t*)pb;
    size_t la = c->n - a, lb = c->n - b;
    size_t m = la < lb ? la : lb;
    int r = memcmp(c->in + a, c->in + b, m);
    if (r) return r;
    return (la > lb) - (la < lb);
}

/* compute_suffixes, bwt/mod.rs:136-166 */
int o_bwt_compute_suffixes(const uint8_t* in, size_t n, uint32_t* sa)
{
    radix_t radix;
    radix_gather(&radix, in, n);                                        /* :138-139 */
    radix_accumulate(&radix);
    for (size_t i = 0; i < n; i++) sa[radix.freq[in[i]]++] = (uint32_t)i;   /* place(), :144-147 */
    radix_shift(&radix);                                                /* :150 */
    cmp_ctx c = {in, n};
    for (int i = 0; i < ALPHABET_SIZE; i++) {                           /* :152-163 */
        size_t lo = radix.freq[i], hi = radix.freq[i + 1];
        if (lo == hi) continue;
        qsort_r(sa + lo, hi - lo, sizeof(uint32_t), suf_cmp, &c);
    }
    return RCX_OK;
}

/* encode_simple, bwt/mod.rs:214-219 via TransformIterator::next :193-203 */
int o_bwt_encode(const uint8_t* in, size_t n, uint8_t* L, uint32_t* origin)
{
    *origin = 0;
    if (n == 0) return RCX_OK;            /* get_origin() would unwrap None; nothing is emitted */
    uint32_t* sa = (uint32_t*)malloc(n * sizeof(uint32_t));
    if (!sa) return RCX_E_OUTPUT_TOO_SMALL;
    o_bwt_compute_suffixes(in, n, sa);
    for (size_t i = 0; i < n; i++) {
        if (sa[i] == 0) { *origin = (uint32_t)i; L[i] = in[n - 1]; }    /* :195-198 */
        else L[i] = in[sa[i] - 1];                                      /* :200 */
    }
    free(sa);
    return RCX_OK;
}

/* compute_inversion_table, bwt/mod.rs:223-239 */
int o_bwt_inversion_table(const uint8_t* L, size_t n, uint32_t origin, uint32_t* table)
{
    if (origin >= n) return RCX_E_MALFORMED;                            /* :230 index panic */
    radix_t radix;
    radix_gather(&radix, L, n);
    radix_accumulate(&radix);
    table[radix.freq[L[origin]]++] = 0;                                 /* :230 */
    for (size_t i = 0; i < origin; i++) table[radix.freq[L[i]]++] = (uint32_t)(i + 1);        /* :231-233 */
    for (size_t i = origin + 1; i < n; i++) table[radix.freq[L[i]]++] = (uint32_t)(i + 1);    /* :234-236 */
    return RCX_OK;
}

/* decode_simple, bwt/mod.rs:291-294 via InverseIterator::next :266-281 */
int o_bwt_decode(const uint8_t* L, size_t n, uint32_t origin, uint8_t* out)
{
    if (n == 0) return RCX_OK;
    uint32_t* table = (uint32_t*)malloc(n * sizeof(uint32_t));
    if (!table) return RCX_E_OUTPUT_TOO_SMALL;
    int st = o_bwt_inversion_table(L, n, origin, table);
    if (st) { free(table); return st; }
    size_t current = origin;
    for (size_t k = 0; k < n; k++) {                                    /* .take(n) */
        if (current == (size_t)-1) { free(table); return RCX_E_MALFORMED; }   /* iterator ended early */
        current = (size_t)table[current] - 1;                           /* wrapping_sub :270 */
        size_t p = current != (size_t)-1 ? current : origin;            /* :273-277 */
        out[k] = L[p];
    }
    free(table);
    return RCX_OK;
}

/* decode_minimal, bwt/mod.rs:298-315 -- reproduced faithfully, including the fact that it
 * is wrong whenever T[n-1] also occurs in L[..origin].  CPU only. */
int o_bwt_decode_minimal(const uint8_t* L, size_t n, uint32_t origin, uint8_t* out)
{
    if (n == 0) return origin == 0 ? RCX_OK : RCX_E_MALFORMED;          /* :300-302 */
    radix_t radix;
    radix_gather(&radix, L, n);
    radix_accumulate(&radix);
    size_t i = origin;
    for (size_t j = 0; j < n; j++) {                                    /* fold :309-314 */
        if (i >= n) return RCX_E_MALFORMED;
        uint8_t ch = L[i];
        out[n - j - 1] = ch;
        size_t offset = 0;
        for (size_t k = 0; k < i; k++) offset += (L[k] == ch);
        i = radix.freq[ch] + offset;
    }
    return RCX_OK;
}

static void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* bwt::Encoder one-shot write()+finish(): :492-508, encode_block :461-480, flush :510-517 */
int o_bwt_stream_encode(const uint8_t* in, size_t n, uint32_t block_size, uint8_t* out, size_t cap, size_t* out_len)
{
    *out_len = 0;
    if (block_size == 0) return RCX_E_MALFORMED;        /* :499 would loop forever */
    size_t nblk = (n + block_size - 1) / block_size;
    if (cap < 4 + n + 8 * nblk) return RCX_E_OUTPUT_TOO_SMALL;
    size_t o = 0;
    wr32(out + o, block_size); o += 4;                                  /* :494 */
    for (size_t p = 0; p < n; p += block_size) {
        size_t bn = n - p < block_size ? n - p : block_size;
        uint32_t origin;
        wr32(out + o, (uint32_t)bn); o += 4;                            /* :463 */
        int st = o_bwt_encode(in + p, bn, out + o, &origin);
        if (st) return st;
        o += bn;
        wr32(out + o, origin); o += 4;                                  /* :475 */
    }
    *out_len = o;
    return RCX_OK;
}

/* bwt::Decoder (extra_mem = true): read_header :362-371, decode_block :373-401, read :405-431 */
int o_bwt_stream_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len)
{
    size_t p = 0, end = 0;
    *out_len = 0;
    if (n - p < 4) return RCX_E_EOF;                                    /* :369 "unexpected end of file" */
    p += 4;                                                             /* max_block_size: informational */
    for (;;) {
        if (n - p < 4) break;                                           /* :376 EOF at block start ends */
        size_t bn = rd32(in + p); p += 4;
        if (n - p < bn) return RCX_E_EOF;                               /* push_exactly :382 */
        const uint8_t* L = in + p; p += bn;
        if (n - p < 4) return RCX_E_EOF;                                /* :384 */
        uint32_t origin = rd32(in + p); p += 4;
        if (bn == 0) return RCX_E_MALFORMED;                            /* :230 input[origin] panics */
        if (cap - end < bn) return RCX_E_OUTPUT_TOO_SMALL;
        int st = o_bwt_decode(L, bn, origin, out + end);
        if (st) return st;
        end += bn;
    }
    *out_len = end;
    return RCX_OK;
}

/* ---------- MTF, bwt/mtf.rs:44-91 ---------- */
typedef struct { uint8_t symbols[256]; } mtf_t;
static void mtf_new(mtf_t* m) { memset(m->symbols, 0, 256); }                         /* :51-53 */
static void mtf_reset_alphabetical(mtf_t* m) { for (int i = 0; i < 256; i++) m->symbols[i] = (uint8_t)i; }   /* :56-60 */
static int mtf_encode(mtf_t* m, uint8_t sym)                                          /* :63-79 */
{
    uint8_t next = m->symbols[0];
    if (next == sym) return 0;
    unsigned rank = 1;
    for (;;) {
        uint8_t t = m->symbols[rank]; m->symbols[rank] = next; next = t;              /* mem::swap */
        if (next == sym) break;
        rank++;
        if (rank >= 256) return -1;                                                   /* :76 assert */
    }
    m->symbols[0] = sym;
    return (int)rank;
}
static uint8_t mtf_decode(mtf_t* m, uint8_t rank)                                     /* :82-90 */
{
    uint8_t sym = m->symbols[rank];
    for (int i = (int)rank - 1; i >= 0; i--) m->symbols[i + 1] = m->symbols[i];
    m->symbols[0] = sym;
    return sym;
}
void o_mtf_encode(const uint8_t* in, size_t n, uint8_t* out)                          /* Encoder :95-129 */
{
    mtf_t m; mtf_reset_alphabetical(&m);
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)mtf_encode(&m, in[i]);
}
void o_mtf_decode(const uint8_t* in, size_t n, uint8_t* out)                          /* Decoder :133-169 */
{
    mtf_t m; mtf_reset_alphabetical(&m);
    for (size_t i = 0; i < n; i++) out[i] = mtf_decode(&m, in[i]);
}

/* ---------- DC, bwt/dc.rs ---------- */
/* encode :110-149 + EncodeIterator::next :88-104, in encode_simple::<u32> order :153-159 */
int o_dc_encode(const uint8_t* in, size_t n, uint32_t* words, size_t cap_words, size_t* nwords, o_dc_context* ctx)
{
    *nwords = 0;
    if (cap_words < 256) return RCX_E_OUTPUT_TOO_SMALL;
    uint32_t* dist = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!dist) return RCX_E_OUTPUT_TOO_SMALL;
    mtf_t mtf; mtf_new(&mtf);                                           /* &mut MTF::new() :156 */
    size_t num_unique = 0;
    size_t last[256], init[256];
    for (int i = 0; i < 256; i++) { last[i] = n; init[i] = n; }         /* :114-115 */
    for (size_t i = 0; i < n; i++) {                                    /* :117-138 */
        uint8_t sym = in[i];
        dist[i] = (uint32_t)n;                                          /* filler */
        size_t base = last[sym];
        last[sym] = i;
        if (base == n) {
            size_t rank = num_unique;
            mtf.symbols[rank] = sym;                                    /* :123 */
            mtf_encode(&mtf, sym);
            init[sym] = i;
            num_unique++;
        } else {
            int rank = mtf_encode(&mtf, sym);
            if (rank > 0) dist[base] = (uint32_t)(i - base - (size_t)rank - 1);   /* :134 */
        }
    }
    for (size_t rank = 0; rank < num_unique; rank++) {                  /* :139-144 */
        uint8_t sym = mtf.symbols[rank];
        size_t base = last[sym];
        dist[base] = (uint32_t)(n - base - rank - 1);
    }
    for (int i = 0; i < 256; i++) words[i] = (uint32_t)init[i];         /* get_init :157 */
    size_t k = 0, last_active = 0;
    size_t pos[256];
    memcpy(pos, init, sizeof(pos));
    for (size_t i = 0; i < n; i++) {                                    /* EncodeIterator :88-104 */
        if (dist[i] == (uint32_t)n) continue;                           /* filler */
        if (256 + k >= cap_words) { free(dist); return RCX_E_OUTPUT_TOO_SMALL; }
        uint8_t sym = in[i];
        size_t rank = last_active - pos[sym];                           /* :93 */
        last_active = i + 1;
        pos[sym] = i + 1 + dist[i];
        if (ctx) { ctx[k].symbol = sym; ctx[k].last_rank = (uint8_t)rank; ctx[k].distance_limit = (uint32_t)(n - i); }
        words[256 + k] = dist[i];
        k++;
    }
    free(dist);
    *nwords = 256 + k;
    return RCX_OK;
}

/* decode :162-233 driven as decode_simple :236-252 does */
int o_dc_decode(const uint32_t* words, size_t nwords, size_t n, uint8_t* out, size_t* consumed, o_dc_context* ctx)
{
    if (consumed) *consumed = 0;
    if (nwords < 256) return RCX_E_MALFORMED;                           /* :239-241 index panic */
    size_t next[256];
    for (int i = 0; i < 256; i++) next[i] = words[i];
    mtf_t mtf; mtf_new(&mtf);
    size_t i = 0;
    for (int sym = 0; sym < 256; sym++) {                               /* :168-179 */
        size_t d = next[sym];
        if (d < n) {
            size_t j = i;
            while (j > 0 && next[mtf.symbols[j - 1]] > d) { mtf.symbols[j] = mtf.symbols[j - 1]; j--; }
            mtf.symbols[j] = (uint8_t)sym;
            i++;
        }
    }
    if (i <= 1) {                                                       /* :180-187 */
        memset(out, mtf.symbols[0], n);
        return RCX_OK;
    }
    size_t alphabet_size = i;
    uint8_t ranks[256];
    memset(ranks, 0, sizeof(ranks));                                    /* :190-196 */
    size_t di = 256;
    i = 0;
    while (i < n) {                                                     /* :199-229 */
        uint8_t sym = mtf.symbols[0];
        size_t stop = next[mtf.symbols[1]];
        if (stop > n) return RCX_E_MALFORMED;                           /* output[i] index panic */
        while (i < stop) out[i++] = sym;
        if (ctx) { o_dc_context* c = &ctx[di - 256]; c->symbol = sym; c->last_rank = ranks[sym]; c->distance_limit = (uint32_t)(n + 1 - i); }
        di++;                                                           /* decode_simple closure :243-249 */
        if (di > nwords) return RCX_E_EOF;                              /* "Unexpected end of file" (unwrap) */
        size_t future = stop + words[di - 1];
        if (future > n) return RCX_E_MALFORMED;                         /* :213 assert */
        size_t rank = 1;
        while (rank < alphabet_size && future + rank > next[mtf.symbols[rank]]) {   /* :215-218 */
            mtf.symbols[rank - 1] = mtf.symbols[rank];
            rank++;
        }
        mtf.symbols[rank - 1] = sym;                                    /* :225 */
        next[sym] = future + rank - 1;                                  /* :227 */
        ranks[sym] = (uint8_t)(rank - 1);
    }
    for (int s = 0; s < 256; s++)                                       /* :230 assert */
        if (next[s] < n || next[s] >= n + alphabet_size) {
            /* symbols absent from the block keep next == init >= n; the reference asserts
             * on all 256 entries, so an absent symbol with init >= n+alphabet_size panics */
            return RCX_E_MALFORMED;
        }
    if (consumed) *consumed = di - 256;
    return RCX_OK;
}
