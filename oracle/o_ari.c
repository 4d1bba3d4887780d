/* o_ari.c -- CPU restatement of src/entropy/ari/{mod,table,bin}.rs
 * (TEST INFRASTRUCTURE, see oracle.h).  u32 arithmetic wraps exactly as the
 * reference's release build does; the linear-scan frequency table is kept. */
#include <string.h>
#include "oracle.h"

/* ari/mod.rs:51-61 */
#define SYMBOL_TOTAL 256u
#define BORDER_EXCESS 24u
#define BORDER_SYMBOL_MASK 0xff000000u
#define RANGE_DEFAULT_THRESHOLD (1u << 14)

typedef struct { uint32_t low, hai, threshold; } range_t;                /* mod.rs:67-78 */
static void range_new(range_t* r, uint32_t max_range) { r->low = 0; r->hai = ~0u; r->threshold = max_range; }   /* :82-91 */

/* RangeEncoder::process, mod.rs:117-150 */
static unsigned range_process(range_t* r, uint32_t total, uint32_t from, uint32_t to, uint8_t* output)
{
    uint32_t range = (r->hai - r->low) / total;
    uint32_t lo = r->low + range * from;
    uint32_t hi = r->low + range * to;
    unsigned num_shift = 0;
    for (;;) {
        if (((lo ^ hi) & BORDER_SYMBOL_MASK) != 0) {
            if (hi - lo > r->threshold) break;
            uint32_t lim = hi & BORDER_SYMBOL_MASK;
            if (hi - lim >= lim - lo) lo = lim;
            else hi = lim - 1;
        }
        output[num_shift++] = (uint8_t)(lo >> BORDER_EXCESS);
        lo <<= 8; hi <<= 8;
    }
    r->low = lo; r->hai = hi;
    return num_shift;
}
/* RangeEncoder::query, mod.rs:153-159 */
static uint32_t range_query(const range_t* r, uint32_t total, uint32_t code)
{
    uint32_t range = (r->hai - r->low) / total;
    return (code - r->low) / range;
}

/* ---- ari::Encoder / Decoder over a byte buffer, mod.rs:208-293 ---- */
typedef struct { uint8_t* out; size_t cap, o; range_t range; int overflow; } aenc_t;
typedef struct { const uint8_t* in; size_t n, p; range_t range; uint32_t code; unsigned pending; } adec_t;
static void aenc_new(aenc_t* e, uint8_t* out, size_t cap) { e->out = out; e->cap = cap; e->o = 0; e->overflow = 0; range_new(&e->range, RANGE_DEFAULT_THRESHOLD); }
static void aenc_put(aenc_t* e, const uint8_t* b, unsigned k)
{
    for (unsigned i = 0; i < k; i++) { if (e->o < e->cap) e->out[e->o++] = b[i]; else e->overflow = 1; }
}
static void aenc_encode(aenc_t* e, uint32_t total, uint32_t lo, uint32_t hi)   /* Model::encode :184-189 + Encoder::encode :223-227 */
{
    uint8_t buf[4];
    unsigned k = range_process(&e->range, total, lo, hi, buf);
    aenc_put(e, buf, k);
}
static void aenc_finish(aenc_t* e)                                       /* :230-237 */
{
    uint32_t code = e->range.low;                                        /* get_code_tail :163-168 */
    uint8_t b[4] = {(uint8_t)(code >> 24), (uint8_t)(code >> 16), (uint8_t)(code >> 8), (uint8_t)code};
    aenc_put(e, b, 4);
}
static void adec_new(adec_t* d, const uint8_t* in, size_t n) { d->in = in; d->n = n; d->p = 0; d->code = 0; d->pending = 4; range_new(&d->range, RANGE_DEFAULT_THRESHOLD); }
static int adec_feed(adec_t* d)                                          /* :271-278 */
{
    while (d->pending != 0) {
        if (d->p >= d->n) return 0;
        d->code = (d->code << 8) + d->in[d->p++];
        d->pending--;
    }
    return 1;
}

/* ---- table::Model, table.rs:20-122 ---- */
#define TAB_MAX 257
typedef struct { uint32_t total; uint16_t table[TAB_MAX]; unsigned n; uint32_t cut_threshold; unsigned cut_shift; } tab_t;
static void tab_downscale(tab_t* t)                                      /* :82-91 */
{
    uint16_t roundup = (uint16_t)((1u << t->cut_shift) - 1);
    t->total = 0;
    for (unsigned i = 0; i < t->n; i++) { t->table[i] = (uint16_t)((t->table[i] + roundup) >> t->cut_shift); t->total += t->table[i]; }
}
static void tab_new_flat(tab_t* t, unsigned num_values, uint32_t threshold)   /* :34-57 */
{
    t->n = num_values; t->cut_threshold = threshold; t->cut_shift = 1; t->total = 0;
    for (unsigned i = 0; i < num_values; i++) { t->table[i] = 1; t->total += 1; }
    while (t->total >= threshold) tab_downscale(t);
}
static void tab_reset_flat(tab_t* t) { for (unsigned i = 0; i < t->n; i++) t->table[i] = 1; t->total = t->n; }   /* :60-65 */
static void tab_update(tab_t* t, unsigned value, unsigned add_log, uint32_t add_const)   /* :69-79 */
{
    uint32_t add = (t->total >> add_log) + add_const;
    t->table[value] = (uint16_t)(t->table[value] + (uint16_t)add);
    t->total += add;
    if (t->total >= t->cut_threshold) tab_downscale(t);
}
static void tab_get_range(const tab_t* t, unsigned value, uint32_t* lo, uint32_t* hi)   /* :100-103 */
{
    uint32_t l = 0;
    for (unsigned i = 0; i < value; i++) l += t->table[i];
    *lo = l; *hi = l + t->table[value];
}
static int tab_find_value(const tab_t* t, uint32_t offset, unsigned* value, uint32_t* lo, uint32_t* hi)   /* :105-117 */
{
    if (offset >= t->total) return 0;                                    /* :106 assert */
    unsigned v = 0; uint32_t l = 0, h;
    while ((h = l + t->table[v]) <= offset) { l = h; v++; }
    *value = v; *lo = l; *hi = h;
    return 1;
}

/* ByteEncoder::write + finish, table.rs:185-224 */
int o_ari_byte_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len)
{
    aenc_t e; aenc_new(&e, out, cap);
    tab_t freq; tab_new_flat(&freq, SYMBOL_TOTAL + 1, RANGE_DEFAULT_THRESHOLD >> 2);   /* :192-199 */
    uint32_t lo, hi;
    for (size_t i = 0; i < n; i++) {                                     /* :211-219 */
        tab_get_range(&freq, in[i], &lo, &hi);
        aenc_encode(&e, freq.total, lo, hi);
        tab_update(&freq, in[i], 10, 1);
    }
    tab_get_range(&freq, SYMBOL_TOTAL, &lo, &hi);                        /* finish :203-207 */
    aenc_encode(&e, freq.total, lo, hi);
    aenc_finish(&e);
    *out_len = e.o;
    return e.overflow ? RCX_E_OUTPUT_TOO_SMALL : RCX_OK;
}
uint64_t o_ari_byte_encode_bound(uint64_t n) { return 2 * n + 16; }      /* <=4 bytes/symbol worst case is never near */

/* ByteDecoder::read to EOF symbol + finish, table.rs:229-273 */
int o_ari_byte_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, size_t* in_used)
{
    adec_t d; adec_new(&d, in, n);
    tab_t freq; tab_new_flat(&freq, SYMBOL_TOTAL + 1, RANGE_DEFAULT_THRESHOLD >> 2);
    size_t o = 0;
    int st = RCX_OK;
    *out_len = 0;
    if (in_used) *in_used = 0;
    for (;;) {
        if (!adec_feed(&d)) { st = RCX_E_MALFORMED; break; }             /* mod.rs:282 feed().unwrap() panics */
        uint32_t total = freq.total;                                     /* Model::decode mod.rs:193-203 */
        uint32_t offset = range_query(&d.range, total, d.code);
        unsigned value; uint32_t lo, hi; uint8_t tmp[4];
        if (!tab_find_value(&freq, offset, &value, &lo, &hi)) { st = RCX_E_MALFORMED; break; }
        d.pending = range_process(&d.range, total, lo, hi, tmp);
        if (value == SYMBOL_TOTAL) break;                                /* table.rs:263-266 */
        if (o >= cap) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
        tab_update(&freq, value, 10, 1);
        out[o++] = (uint8_t)value;
    }
    if (st == RCX_OK && !adec_feed(&d)) st = RCX_E_EOF;                  /* finish(): feed() error is returned, mod.rs:289-292 */
    *out_len = o;
    if (in_used) *in_used = d.p;
    return st;
}

/* ---- bin::Model, bin.rs:17-82 ---- */
typedef struct { uint32_t zero, total, rate; } bin_t;
static void bin_new_flat(bin_t* b, uint32_t threshold, uint32_t rate) { b->zero = threshold >> 1; b->total = threshold; b->rate = rate; }
static void bin_reset_flat(bin_t* b) { b->zero = b->total >> 1; }
static void bin_update(bin_t* b, int value)                              /* :60-82 */
{
    if (value) b->zero -= b->zero >> b->rate;
    else b->zero += (b->total - b->zero) >> b->rate;
}

/* test.rs:22-50: encode_binary / roundtrip_binary */
int o_ari_binary_encode(const uint8_t* in, size_t n, uint32_t rate, uint8_t* out, size_t cap, size_t* out_len)
{
    aenc_t e; aenc_new(&e, out, cap);
    bin_t bm; bin_new_flat(&bm, RANGE_DEFAULT_THRESHOLD >> 3, rate);
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 8; k++) {
            int bit = (in[i] >> k) & 1;
            if (bit) aenc_encode(&e, bm.total, bm.zero, bm.total);      /* get_range bin.rs:86-92 */
            else aenc_encode(&e, bm.total, 0, bm.zero);
            bin_update(&bm, bit);
        }
    aenc_finish(&e);
    *out_len = e.o;
    return e.overflow ? RCX_E_OUTPUT_TOO_SMALL : RCX_OK;
}
int o_ari_binary_decode(const uint8_t* in, size_t n, uint32_t rate, uint8_t* out, size_t nbytes)
{
    adec_t d; adec_new(&d, in, n);
    bin_t bm; bin_new_flat(&bm, RANGE_DEFAULT_THRESHOLD >> 3, rate);
    uint8_t tmp[4];
    for (size_t i = 0; i < nbytes; i++) {
        uint8_t value = 0;
        for (int k = 0; k < 8; k++) {
            if (!adec_feed(&d)) return RCX_E_MALFORMED;
            uint32_t offset = range_query(&d.range, bm.total, d.code);
            if (offset >= bm.total) return RCX_E_MALFORMED;              /* bin.rs:95 assert */
            int bit = !(offset < bm.zero);                               /* find_value bin.rs:94-103 */
            if (bit) d.pending = range_process(&d.range, bm.total, bm.zero, bm.total, tmp);
            else d.pending = range_process(&d.range, bm.total, 0, bm.zero, tmp);
            bin_update(&bm, bit);
            value = (uint8_t)(value + (bit << k));
        }
        out[i] = value;
    }
    return RCX_OK;
}

/* test.rs:91-148 roundtrip_proxy: table::SumProxy (table.rs:127-180) + bin::SumProxy (bin.rs:112-167) */
static uint32_t tsp_den(const tab_t* a, const tab_t* b, uint32_t wa, uint32_t wb, uint32_t ws) { return (wa * a->total + wb * b->total) >> ws; }
static uint32_t bsp_zero(const bin_t* a, const bin_t* b, uint32_t wa, uint32_t wb, uint32_t ws) { return (wa * a->zero + wb * b->zero) >> ws; }
static uint32_t bsp_den(const bin_t* a, const bin_t* b, uint32_t wa, uint32_t wb, uint32_t ws) { return (wa * a->total + wb * b->total) >> ws; }

int o_ari_proxy_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len)
{
    const uint32_t threshold = RANGE_DEFAULT_THRESHOLD >> 3;
    tab_t t0, t1; tab_new_flat(&t0, 16, threshold); tab_new_flat(&t1, 16, threshold);
    bin_t b0, b1; bin_new_flat(&b0, threshold, 3); bin_new_flat(&b1, threshold, 5);
    aenc_t e; aenc_new(&e, out, cap);
    for (size_t i = 0; i < n; i++) {
        unsigned high = in[i] >> 4;
        uint32_t lo0, hi0, lo1, hi1;
        tab_get_range(&t0, high, &lo0, &hi0); tab_get_range(&t1, high, &lo1, &hi1);   /* table.rs:150-155 */
        aenc_encode(&e, tsp_den(&t0, &t1, 2, 1, 0), (2 * lo0 + 1 * lo1) >> 0, (2 * hi0 + 1 * hi1) >> 0);
        tab_update(&t0, high, 10, 1); tab_update(&t1, high, 5, 1);
        for (int k = 0; k < 4; k++) {
            int bit = (in[i] >> k) & 1;
            uint32_t zero = bsp_zero(&b0, &b1, 1, 1, 1), tot = bsp_den(&b0, &b1, 1, 1, 1);
            if (bit) aenc_encode(&e, tot, zero, tot); else aenc_encode(&e, tot, 0, zero);
            bin_update(&b0, bit); bin_update(&b1, bit);
        }
    }
    aenc_finish(&e);
    *out_len = e.o;
    return e.overflow ? RCX_E_OUTPUT_TOO_SMALL : RCX_OK;
}
int o_ari_proxy_decode(const uint8_t* in, size_t n, uint8_t* out, size_t nbytes)
{
    const uint32_t threshold = RANGE_DEFAULT_THRESHOLD >> 3;
    tab_t t0, t1; tab_new_flat(&t0, 16, threshold); tab_new_flat(&t1, 16, threshold);
    bin_t b0, b1; bin_new_flat(&b0, threshold, 3); bin_new_flat(&b1, threshold, 5);
    tab_reset_flat(&t0); tab_reset_flat(&t1); bin_reset_flat(&b0); bin_reset_flat(&b1);
    adec_t d; adec_new(&d, in, n);
    uint8_t tmp[4];
    for (size_t i = 0; i < nbytes; i++) {
        if (!adec_feed(&d)) return RCX_E_MALFORMED;
        uint32_t tot = tsp_den(&t0, &t1, 2, 1, 0);
        uint32_t offset = range_query(&d.range, tot, d.code);
        if (offset >= tot) return RCX_E_MALFORMED;                       /* table.rs:158 assert */
        unsigned value = 0; uint32_t lo = 0, hi;                         /* SumProxy::find_value table.rs:157-173 */
        while ((hi = lo + ((2 * (uint32_t)t0.table[value] + 1 * (uint32_t)t1.table[value]) >> 0)) <= offset) {
            lo = hi; value++;
            if (value >= 16) return RCX_E_MALFORMED;
        }
        d.pending = range_process(&d.range, tot, lo, hi, tmp);
        unsigned high = value;
        tab_update(&t0, high, 10, 1); tab_update(&t1, high, 5, 1);
        uint8_t v = (uint8_t)(high << 4);
        for (int k = 0; k < 4; k++) {
            if (!adec_feed(&d)) return RCX_E_MALFORMED;
            uint32_t zero = bsp_zero(&b0, &b1, 1, 1, 1), t = bsp_den(&b0, &b1, 1, 1, 1);
            uint32_t off = range_query(&d.range, t, d.code);
            if (off >= t) return RCX_E_MALFORMED;
            int bit = !(off < zero);
            if (bit) d.pending = range_process(&d.range, t, zero, t, tmp);
            else d.pending = range_process(&d.range, t, 0, zero, tmp);
            v = (uint8_t)(v + (bit << k));
            bin_update(&b0, bit); bin_update(&b1, bit);
        }
        out[i] = v;
    }
    return RCX_OK;
}

/* ---- apm::Bit / apm::Gate, apm.rs:36-198, driven as test.rs:150-182 (roundtrip_apm) ----
 * f32 ln/exp: Rust's f32::ln / f32::exp are the platform libm's logf / expf, the functions called here (apm.rs:56-58,
 * 72-74); the reference cannot be run in this image, so these bits are pinned to glibc's libm, not to a Rust run. */
#include <math.h>
#define APM_FLAT_TOTAL 4096
#define APM_WIDE_OFFSET 2048
#define APM_BINS 17
static int apm_to_wide(uint32_t fp, int* wp)                             /* Bit::to_wide :53-59 */
{
    float p = (float)fp / (float)APM_FLAT_TOTAL;
    float d = logf(p / (1.0f - p));
    float w = d * (float)APM_WIDE_OFFSET;
    if (!(w > -32769.0f && w < 32768.0f)) return 0;                      /* to_i16().unwrap() */
    *wp = (int)(int16_t)w;
    return 1;
}
static uint32_t apm_from_wide(int wp)                                    /* Bit::from_wide :69-75 */
{
    float d = (float)wp / (float)APM_WIDE_OFFSET;
    float p = 1.0f / (1.0f + expf(-d));
    return (uint32_t)(uint16_t)(p * (float)APM_FLAT_TOTAL);
}
void o_apm_tables(int16_t* stretch, uint16_t* gate)                      /* stretch[fp] (0x8000 = to_i16 fails), Gate::new :144-154 */
{
    for (uint32_t fp = 0; fp < APM_FLAT_TOTAL; fp++) { int w; stretch[fp] = apm_to_wide(fp, &w) ? (int16_t)w : (int16_t)0x8000; }
    for (int i = 0; i < APM_BINS; i++) {
        float rp = (float)i / 8.0f - 1.0f;
        gate[i] = (uint16_t)apm_from_wide((int)(int16_t)(rp * (float)APM_WIDE_OFFSET));
    }
}
typedef struct { uint32_t bit; uint16_t map[APM_BINS]; int16_t stretch[APM_FLAT_TOTAL]; } apm_t;
static void apm_new(apm_t* a) { a->bit = APM_FLAT_TOTAL >> 1; o_apm_tables(a->stretch, a->map); }
static void apm_upd(uint16_t* fp, int value)                             /* Bit::update(value, 10, 0) :78-101 */
{
    if (!value) *fp = (uint16_t)(*fp + (uint16_t)((APM_FLAT_TOTAL - (int)*fp) >> 10));
    else *fp = (uint16_t)(*fp - (uint16_t)(((int)*fp) >> 10));
}
/* gate.pass(&bit) :157-173: 0 if the reference would panic (to_i16 fails, or the bin index leaves the map) */
static int apm_pass(const apm_t* a, uint32_t* fp_new, int* index)
{
    int16_t wp = a->stretch[a->bit & 4095];
    if (wp == (int16_t)0x8000) return 0;
    int idx = ((int)wp + APM_WIDE_OFFSET) >> 8;
    if (idx < 0 || idx + 1 >= APM_BINS) return 0;
    uint32_t weight = (uint32_t)(uint16_t)wp & 255u;
    uint32_t sum = (uint32_t)a->map[idx] * (256u - weight) + (uint32_t)a->map[idx + 1] * weight;
    *fp_new = (uint32_t)(uint16_t)(sum >> 8);
    *index = idx;
    return 1;
}
static void apm_update(apm_t* a, int value, int idx)
{
    uint16_t b = (uint16_t)a->bit; apm_upd(&b, value); a->bit = b;
    apm_upd(&a->map[idx], value); apm_upd(&a->map[idx + 1], value);
}
int o_ari_apm_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len)
{
    aenc_t e; aenc_new(&e, out, cap);
    apm_t a; apm_new(&a);
    *out_len = 0;
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 8; k++) {
            int bit = (in[i] >> k) & 1, idx; uint32_t fp;
            if (!apm_pass(&a, &fp, &idx)) return RCX_E_MALFORMED;        /* the reference panics here */
            if (bit) aenc_encode(&e, APM_FLAT_TOTAL, fp, APM_FLAT_TOTAL); else aenc_encode(&e, APM_FLAT_TOTAL, 0, fp);
            apm_update(&a, bit, idx);
        }
    aenc_finish(&e);
    *out_len = e.o;
    return e.overflow ? RCX_E_OUTPUT_TOO_SMALL : RCX_OK;
}
int o_ari_apm_decode(const uint8_t* in, size_t n, uint8_t* out, size_t nbytes)
{
    adec_t d; adec_new(&d, in, n);
    apm_t a; apm_new(&a);
    uint8_t tmp[4];
    for (size_t i = 0; i < nbytes; i++) {
        uint8_t v = 0;
        for (int k = 0; k < 8; k++) {
            int idx; uint32_t fp;
            if (!apm_pass(&a, &fp, &idx)) return RCX_E_MALFORMED;
            if (!adec_feed(&d)) return RCX_E_MALFORMED;
            uint32_t offset = range_query(&d.range, APM_FLAT_TOTAL, d.code);
            if (offset >= APM_FLAT_TOTAL) return RCX_E_MALFORMED;       /* apm.rs:116 assert */
            int bit = !(offset < fp);
            if (bit) d.pending = range_process(&d.range, APM_FLAT_TOTAL, fp, APM_FLAT_TOTAL, tmp);
            else d.pending = range_process(&d.range, APM_FLAT_TOTAL, 0, fp, tmp);
            apm_update(&a, bit, idx);
            v = (uint8_t)(v + (bit << k));
        }
        out[i] = v;
    }
    return RCX_OK;
}
