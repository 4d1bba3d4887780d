/* o_lz4.c -- CPU restatement of src/lz4.rs (TEST INFRASTRUCTURE, see oracle.h). */
#include <string.h>
#include <stdlib.h>
#include "oracle.h"

/* lz4.rs:40-53 */
#define MAGIC 0x184d2204u
#define ML_BITS 4u
#define ML_MASK ((1u << ML_BITS) - 1u)
#define RUN_BITS (8u - ML_BITS)
#define RUN_MASK ((1u << RUN_BITS) - 1u)
#define MIN_MATCH 4u
#define HASH_LOG 17u
#define HASH_TABLE_SIZE (1u << HASH_LOG)
#define HASH_SHIFT ((MIN_MATCH * 8u) - HASH_LOG)
#define INCOMPRESSIBLE 128u
#define UNINITHASH 0x88888888u
#define MAX_INPUT_SIZE 0x7e000000u

/* BlockDecoder::decode, lz4.rs:67-110 (+ length :112-122, bump :124-128, cp :131-140).
 * The DECR special case for offsets 1..3 (:100-106) is byte-equivalent to a plain
 * forward byte copy of 4+len bytes, which is what is written here. */
int o_lz4_decode_block(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len)
{
    size_t cur = 0, end = 0;
    *out_len = 0;
    while (cur < n) {                                   /* :68 */
        uint8_t code = in[cur++];                       /* :69 */
        size_t len = code >> 4;                         /* :73 */
        if (len == 0xf) {                               /* length(), :112-122 */
            for (;;) {
                if (cur >= n) return RCX_E_MALFORMED;   /* bump() index panic */
                uint8_t tmp = in[cur++];
                len += tmp;
                if (tmp != 0xff) break;
            }
        }
        if (len > 0) {                                  /* :75-85 */
            if (len > n - cur) return RCX_E_MALFORMED;  /* reads past the input slice */
            if (len > cap - end) return RCX_E_OUTPUT_TOO_SMALL;
            memcpy(out + end, in + cur, len);
            end += len;
            cur += len;
        }
        if (cur == n) break;                            /* :87 */
        if (n - cur < 2) return RCX_E_MALFORMED;        /* bump() panic */
        size_t back = (size_t)in[cur] | ((size_t)in[cur + 1] << 8);   /* :91 */
        cur += 2;
        if (back > end) return RCX_E_MALFORMED;         /* :93 usize underflow / OOB */
        if (back == 0) return RCX_E_MALFORMED;          /* copies not-yet-written bytes */
        size_t mlen = code & 0xf;                       /* :98 */
        if (mlen == 0xf) {
            for (;;) {
                if (cur >= n) return RCX_E_MALFORMED;
                uint8_t tmp = in[cur++];
                mlen += tmp;
                if (tmp != 0xff) break;
            }
        }
        mlen += 4;                                      /* :100-106 */
        if (mlen > cap - end) return RCX_E_OUTPUT_TOO_SMALL;
        size_t start = end - back;
        for (size_t i = 0; i < mlen; i++)               /* cp(), :134-136 */
            out[end + i] = out[start + i];
        end += mlen;
    }
    *out_len = end;                                     /* :109 */
    return RCX_OK;
}

/* compression_bound, lz4.rs:175-181; 0 stands for None */
uint64_t o_lz4_compression_bound(uint64_t n)
{
    if (n > MAX_INPUT_SIZE) return 0;
    return n + (n / 255) + 16 + 4;
}

static inline uint32_t seq_at(const uint8_t* in, uint32_t pos)   /* :185-190 */
{
    return ((uint32_t)in[pos + 3] << 24) | ((uint32_t)in[pos + 2] << 16) |
           ((uint32_t)in[pos + 1] << 8) | (uint32_t)in[pos];
}

/* write_literals, lz4.rs:192-224 */
static uint32_t write_literals(const uint8_t* in, uint8_t* out, uint32_t dest_pos,
                               uint32_t len, uint32_t ml_len, uint32_t pos)
{
    uint32_t ln = len;
    uint8_t code = (ln > RUN_MASK - 1) ? (uint8_t)RUN_MASK : (uint8_t)ln;
    if (ml_len > ML_MASK - 1) out[dest_pos] = (uint8_t)((code << ML_BITS) + ML_MASK);
    else                      out[dest_pos] = (uint8_t)((code << ML_BITS) + ml_len);
    dest_pos += 1;
    if (code == RUN_MASK) {
        ln -= RUN_MASK;
        while (ln > 254) { out[dest_pos++] = 255; ln -= 255; }
        out[dest_pos++] = (uint8_t)ln;
    }
    memcpy(out + dest_pos, in + pos, len);
    return dest_pos + len;
}

/* BlockEncoder::encode, lz4.rs:226-310, via encode_block :616-627.
 * u32 wrapping arithmetic throughout (release-profile semantics for :265). */
int o_lz4_encode_block(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len)
{
    *out_len = 0;
    uint64_t bound = o_lz4_compression_bound(n);
    if (bound == 0) return RCX_E_LZ4_INPUT_TOO_LARGE;   /* :229-230 -> 0 */
    if (cap < bound) return RCX_E_OUTPUT_TOO_SMALL;      /* :233-237 grows the Vec */
    uint32_t* table = (uint32_t*)calloc(HASH_TABLE_SIZE, sizeof(uint32_t));   /* :620 */
    if (!table) return RCX_E_OUTPUT_TOO_SMALL;
    uint32_t input_len = (uint32_t)n;
    uint32_t pos = 0, anchor = 0, dest_pos = 0;
    uint32_t step = 1, limit = INCOMPRESSIBLE;           /* :239-240 */
    for (;;) {
        if (pos + 12 > input_len) {                      /* :243-248 */
            dest_pos = write_literals(in, out, dest_pos, input_len - anchor, 0, anchor);
            break;
        }
        uint32_t seq = seq_at(in, pos);                  /* :250 */
        uint32_t hash = (seq * 2654435761u) >> HASH_SHIFT;   /* :251 */
        uint32_t r = table[hash] + UNINITHASH;           /* :252 */
        table[hash] = pos - UNINITHASH;                  /* :253 */
        if (((pos - r) >> 16) != 0 || seq != seq_at(in, r)) {   /* :255 */
            if (pos - anchor > limit) {                  /* :256-259 */
                limit <<= 1;
                step += 1 + (step >> 2);
            }
            pos += step;                                 /* :260 */
            continue;
        }
        if (step > 1) {                                  /* :264-269 */
            table[hash] = r - UNINITHASH;
            pos -= step - 1;
            step = 1;
            continue;
        }
        limit = INCOMPRESSIBLE;                          /* :271 */
        uint32_t ln = pos - anchor;                      /* :273 */
        uint32_t back = pos - r;                         /* :274 */
        uint32_t old_anchor = anchor;
        pos += MIN_MATCH;                                /* :277-279 */
        r += MIN_MATCH;
        anchor = pos;
        while (pos < input_len - 5 && in[pos] == in[r]) { pos++; r++; }   /* :281-284 */
        uint32_t ml_len = pos - anchor;                  /* :286 */
        dest_pos = write_literals(in, out, dest_pos, ln, ml_len, old_anchor);   /* :288 */
        out[dest_pos] = (uint8_t)back;                   /* :289-291 */
        out[dest_pos + 1] = (uint8_t)(back >> 8);
        dest_pos += 2;
        if (ml_len > ML_MASK - 1) {                      /* :293-304 */
            ml_len -= ML_MASK;
            while (ml_len > 254) { ml_len -= 255; out[dest_pos++] = 255; }
            out[dest_pos++] = (uint8_t)ml_len;
        }
        anchor = pos;                                    /* :306 */
    }
    free(table);
    *out_len = dest_pos;
    return RCX_OK;
}

/* --- frame reader: Decoder::read_header :363-420, decode_block :422-464, read :471-499 --- */
static int rd_u32le(const uint8_t* in, size_t n, size_t* p, uint32_t* v)
{
    if (n - *p < 4) return 0;
    *v = (uint32_t)in[*p] | ((uint32_t)in[*p + 1] << 8) | ((uint32_t)in[*p + 2] << 16) | ((uint32_t)in[*p + 3] << 24);
    *p += 4;
    return 1;
}

int o_lz4_frame_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, size_t* in_used)
{
    size_t p = 0, end = 0;
    uint32_t v;
    *out_len = 0;
    if (in_used) *in_used = 0;
    if (!rd_u32le(in, n, &p, &v)) return RCX_E_EOF;      /* :365 byteorder UnexpectedEof */
    if (v != MAGIC) return RCX_E_LZ4_MAGIC;              /* :366 */
    uint8_t bits[2] = {0, 0};                            /* :369-372: read() may be short */
    for (int i = 0; i < 2 && p < n; i++) bits[i] = in[p++];
    uint8_t flg = bits[0], bd = bits[1];
    if ((flg >> 6) != 1) return RCX_E_LZ4_VERSION;       /* :375-377 */
    int blk_checksum = (flg & 0x10) != 0;                /* :380 */
    int stream_size = (flg & 0x08) != 0;                 /* :382 */
    int preset_dictionary = (flg & 0x01) != 0;           /* :387 */
    static const size_t MAX_SIZES[8] = {0, 0, 0, 0, 64u << 10, 256u << 10, 1u << 20, 4u << 20};   /* :389-394 */
    size_t max_block_size = MAX_SIZES[(bd >> 4) & 7];
    (void)max_block_size;                                /* only sizes a reserve() :444 */
    if (stream_size) {                                   /* :402-406 */
        if (n - p < 8) return RCX_E_EOF;
        p += 8;
    }
    if (preset_dictionary) return RCX_E_MALFORMED;       /* :407 assert! */
    if (p >= n) return RCX_E_EOF;                        /* :417 header checksum byte, ignored */
    p += 1;
    for (;;) {                                           /* read() loop :480-496 */
        if (!rd_u32le(in, n, &p, &v)) return RCX_E_EOF;  /* :423 */
        if (v == 0) break;                               /* :425 */
        if (v & 0x80000000u) {                           /* :428-435 stored */
            size_t amt = v & 0x7fffffffu;
            if (n - p < amt) return RCX_E_EOF;           /* push_exactly */
            if (cap - end < amt) return RCX_E_OUTPUT_TOO_SMALL;
            memcpy(out + end, in + p, amt);
            p += amt; end += amt;
        } else {                                         /* :438-456 compressed */
            size_t bn = v, got = 0;
            if (n - p < bn) return RCX_E_EOF;
            int st = o_lz4_decode_block(in + p, bn, out + end, cap - end, &got);
            if (st != RCX_OK) return st;
            p += bn; end += got;
        }
        if (blk_checksum) {                              /* :459-462 skipped */
            if (n - p < 4) return RCX_E_EOF;
            p += 4;
        }
    }
    *out_len = end;
    if (in_used) *in_used = p;                           /* content checksum is never read */
    return RCX_OK;
}

/* frame writer, one-shot write()+finish(): Encoder::write :565-589, encode_block :530-541
 * (compress() is false :543-545 -> stored blocks), finish :550-561 */
int o_lz4_frame_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len)
{
    const size_t limit = 256 * 1024;                     /* :526 */
    size_t nblk = (n + limit - 1) / limit;
    size_t need = 7 + n + 4 * nblk + 8;
    *out_len = 0;
    if (cap < need) return RCX_E_OUTPUT_TOO_SMALL;
    size_t o = 0;
    out[o++] = 0x04; out[o++] = 0x22; out[o++] = 0x4d; out[o++] = 0x18;   /* :567 */
    out[o++] = 0x60;                                     /* :570 */
    out[o++] = 0x50;                                     /* :572 */
    out[o++] = 0x00;                                     /* :574 */
    for (size_t p = 0; p < n; p += limit) {
        size_t amt = n - p < limit ? n - p : limit;
        uint32_t hdr = (uint32_t)amt | 0x80000000u;      /* :536 */
        out[o++] = (uint8_t)hdr; out[o++] = (uint8_t)(hdr >> 8);
        out[o++] = (uint8_t)(hdr >> 16); out[o++] = (uint8_t)(hdr >> 24);
        memcpy(out + o, in + p, amt);
        o += amt;
    }
    memset(out + o, 0, 8);                               /* :553-558 */
    o += 8;
    *out_len = o;
    return RCX_OK;
}
