/* o_flate.c -- CPU restatement of src/flate.rs, src/zlib.rs, src/checksum/adler.rs
 * (TEST INFRASTRUCTURE, see oracle.h).  Bit-serial canonical Huffman walk and
 * byte-loop copies exactly as the reference does them. */
#include <string.h>
#include <pthread.h>
#include "oracle.h"

/* flate.rs:36-40 */
#define MAXBITS 15
#define MAXLCODES 286
#define MAXDCODES 30
#define MAXCODES (MAXLCODES + MAXDCODES)
#define HISTORY (32 * 1024)

typedef struct {                 /* flate.rs:69-77 */
    uint16_t count[MAXBITS + 1];
    uint16_t symbol[MAXCODES];
} huff_t;

typedef struct {                 /* the parts of flate::Decoder :164-177 that matter */
    const uint8_t* in; size_t n, p;          /* r */
    uint8_t* out; size_t cap, end;           /* block + output history, flattened */
    size_t bitbuf; unsigned bitcnt;
    int eof;
} fl_t;

/* HuffmanTree::construct, flate.rs:83-120 */
static int construct(huff_t* t, const uint16_t* lens, size_t nlens)
{
    memset(t, 0, sizeof(*t));
    for (size_t i = 0; i < nlens; i++) t->count[lens[i]]++;          /* :89-91 */
    if (t->count[0] == nlens) return RCX_OK;                          /* :93 */
    long left = 1;                                                    /* :98-103 */
    for (int i = 1; i <= MAXBITS; i++) {
        left *= 2;
        left -= t->count[i];
        if (left < 0) return RCX_E_INVALID_HUFFMAN_TREE;
    }
    uint16_t offs[MAXBITS + 1] = {0};                                 /* :106-109 */
    for (int i = 1; i < MAXBITS; i++) offs[i + 1] = offs[i] + t->count[i];
    for (size_t sym = 0; sym < nlens; sym++)                          /* :113-118 */
        if (lens[sym] != 0) t->symbol[offs[lens[sym]]++] = (uint16_t)sym;
    return RCX_OK;
}

/* Decoder::bits, flate.rs:250-260 */
static int bits(fl_t* s, unsigned cnt, uint16_t* ret)
{
    while (s->bitcnt < cnt) {
        if (s->p >= s->n) return RCX_E_EOF;                           /* :252 read_u8 */
        s->bitbuf |= (size_t)s->in[s->p++] << s->bitcnt;
        s->bitcnt += 8;
    }
    *ret = (uint16_t)(s->bitbuf & (((size_t)1 << cnt) - 1));
    s->bitbuf >>= cnt;
    s->bitcnt -= cnt;
    return RCX_OK;
}

/* HuffmanTree::decode, flate.rs:129-146 */
static int hdecode(const huff_t* t, fl_t* s, uint16_t* sym)
{
    uint16_t code = 0, first = 0, index = 0, b;
    for (int len = 1; len <= MAXBITS; len++) {
        int st = bits(s, 1, &b);
        if (st) return st;
        code |= b;
        uint16_t count = t->count[len];
        if (code < (uint16_t)(first + count)) { *sym = t->symbol[(uint16_t)(index + (code - first))]; return RCX_OK; }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return RCX_E_NOT_ENOUGH_BITS;
}

/* Decoder::codes, flate.rs:262-341 (tables :265-284) */
static const uint16_t EXTRALENS[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51,
                                       59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint16_t EXTRABITS[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                       4, 5, 5, 5, 5, 0};
static const uint16_t EXTRADIST[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385,
                                       513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint16_t EXTRADBITS[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9,
                                        10, 10, 11, 11, 12, 12, 13, 13};

static int codes(fl_t* s, const huff_t* lens, const huff_t* dist)
{
    for (;;) {
        uint16_t sym, x;
        int st = hdecode(lens, s, &sym);                              /* :287 */
        if (st) return st;
        if (sym < 256) {                                              /* :289 */
            if (s->end >= s->cap) return RCX_E_OUTPUT_TOO_SMALL;
            s->out[s->end++] = (uint8_t)sym;
        } else if (sym == 256) {                                      /* :290 */
            return RCX_OK;
        } else if (sym < 290) {                                       /* :291 */
            unsigned nn = sym - 257;
            if (nn > 29) return RCX_E_INVALID_HUFFMAN_CODE;           /* :294 (off by one) */
            if (nn == 29) return RCX_E_MALFORMED;                     /* :297 index panic */
            st = bits(s, EXTRABITS[nn], &x);                          /* :297-298 */
            if (st) return st;
            size_t len = (size_t)EXTRALENS[nn] + x;
            uint16_t d;
            st = hdecode(dist, s, &d);                                /* :302 */
            if (st) return st;
            if (d >= 30) return RCX_E_MALFORMED;                      /* :303 index panic */
            st = bits(s, EXTRADBITS[d], &x);                          /* :303-304 */
            if (st) return st;
            size_t dd = (size_t)EXTRADIST[d] + x;
            size_t hist = s->end < HISTORY ? s->end : HISTORY;        /* output.len() :314 */
            if (dd > hist) return RCX_E_INVALID_HUFFMAN_CODE;
            if (len > s->cap - s->end) return RCX_E_OUTPUT_TOO_SMALL;
            for (size_t i = 0; i < len; i++)                          /* :320-334 */
                s->out[s->end + i] = s->out[s->end + i - dd];
            s->end += len;
        } else {
            return RCX_E_INVALID_HUFFMAN_CODE;                        /* :336 */
        }
    }
}

/* Decoder::statik, flate.rs:237-246 */
static int statik(fl_t* s)
{
    if (s->n - s->p < 2) return RCX_E_EOF;
    uint16_t len = (uint16_t)(s->in[s->p] | (s->in[s->p + 1] << 8)); s->p += 2;
    if (s->n - s->p < 2) return RCX_E_EOF;
    uint16_t nlen = (uint16_t)(s->in[s->p] | (s->in[s->p + 1] << 8)); s->p += 2;
    if ((uint16_t)~nlen != len) return RCX_E_INVALID_STATIC_SIZE;     /* :240 */
    if (s->n - s->p < len) return RCX_E_EOF;                          /* push_exactly */
    if (s->cap - s->end < len) return RCX_E_OUTPUT_TOO_SMALL;
    memcpy(s->out + s->end, s->in + s->p, len);
    s->p += len; s->end += len;
    s->bitcnt = 0; s->bitbuf = 0;                                     /* :243-244 */
    return RCX_OK;
}

/* Decoder::fixed, flate.rs:343-395: the static tables are construct() of the RFC lengths
 * (that is how the reference generated them, :149-160) */
static huff_t FIX_LEN, FIX_DIST;
static void fixed_init(void)
{
    uint16_t arr[288];
    for (int i = 0; i < 144; i++) arr[i] = 8;
    for (int i = 144; i < 256; i++) arr[i] = 9;
    for (int i = 256; i < 280; i++) arr[i] = 7;
    for (int i = 280; i < 288; i++) arr[i] = 8;
    construct(&FIX_LEN, arr, 288);
    FIX_LEN.count[0] = 100;                                           /* :346 (388-288 zero lengths) */
    for (int i = 0; i < MAXDCODES; i++) arr[i] = 5;
    construct(&FIX_DIST, arr, MAXDCODES);
}
static int fixed(fl_t* s)
{
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    pthread_once(&once, fixed_init);
    return codes(s, &FIX_LEN, &FIX_DIST);
}

/* Decoder::dynamic, flate.rs:397-450 */
static int dynamic(fl_t* s)
{
    uint16_t x;
    int st;
    if ((st = bits(s, 5, &x))) return st;
    unsigned hlit = x + 257;                                          /* :398 */
    if ((st = bits(s, 5, &x))) return st;
    unsigned hdist = x + 1;                                           /* :399 */
    if ((st = bits(s, 4, &x))) return st;
    unsigned hclen = x + 4;                                           /* :400 */
    if (hlit > MAXLCODES || hdist > MAXDCODES) return RCX_E_HUFFMAN_TREE_TOO_LARGE;   /* :401 */
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint16_t lengths[MAXCODES];
    memset(lengths, 0, sizeof(lengths));
    for (unsigned i = 0; i < hclen; i++) {                            /* :412-414 */
        if ((st = bits(s, 3, &x))) return st;
        lengths[ORDER[i]] = x;
    }
    huff_t tree;
    if ((st = construct(&tree, lengths, 19))) return st;              /* :415 */
    memset(lengths, 0, sizeof(lengths));                              /* :419 */
    unsigned i = 0;
    while (i < hlit + hdist) {                                        /* :421-441 */
        uint16_t symbol;
        if ((st = hdecode(&tree, s, &symbol))) return st;
        if (symbol < 16) {
            lengths[i++] = symbol;
        } else if (symbol == 16) {
            if (i == 0) return RCX_E_INVALID_HUFFMAN_HEADER_SYMBOL;   /* :428 */
            uint16_t prev = lengths[i - 1];
            if ((st = bits(s, 2, &x))) return st;
            unsigned rep = x + 3;
            for (unsigned k = 0; k < rep; k++) {
                if (i >= MAXCODES) return RCX_E_MALFORMED;            /* :432 index panic */
                lengths[i++] = prev;
            }
        } else if (symbol == 17) {
            if ((st = bits(s, 3, &x))) return st;
            i += x + 3;
        } else if (symbol == 18) {
            if ((st = bits(s, 7, &x))) return st;
            i += x + 11;
        } else {
            return RCX_E_INVALID_HUFFMAN_HEADER_SYMBOL;               /* :439 */
        }
    }
    if (i > hlit + hdist) return RCX_E_INVALID_HUFFMAN_TREE_HEADER;   /* :442 */
    huff_t lencode, distcode;
    if ((st = construct(&lencode, lengths, hlit))) return st;         /* :445-446 */
    if ((st = construct(&distcode, lengths + hlit, hdist))) return st;/* :447-448 */
    return codes(s, &lencode, &distcode);
}

/* Decoder::block :195-206 looped to BFINAL (batch semantics; the reference's
 * Read::read serves one deflate block per refill, :469-473) */
int o_inflate(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, size_t* in_used, uint32_t* flags)
{
    fl_t s = {in, n, 0, out, cap, 0, 0, 0, 0};
    int st = RCX_OK;
    uint32_t fl = 0;
    while (!s.eof) {
        uint16_t x;
        size_t before = s.end;
        if ((st = bits(&s, 1, &x))) break;
        if (x == 1) s.eof = 1;                                        /* :198 */
        if ((st = bits(&s, 2, &x))) break;                            /* :199 */
        if (x == 0) st = statik(&s);
        else if (x == 1) st = fixed(&s);
        else if (x == 2) st = dynamic(&s);
        else st = RCX_E_INVALID_BLOCK_CODE;                           /* :203 */
        if (st) break;
        if (s.end == before && !s.eof) fl |= RCX_W_EMPTY_BLOCK_MIDSTREAM;   /* :474-476 quirk */
    }
    *out_len = s.end;
    if (in_used) *in_used = s.p;
    if (flags) *flags = fl;
    return st;
}

/* State32::feed / result, adler.rs:34-44 */
uint32_t o_adler32_feed(uint32_t state, const uint8_t* buf, size_t n)
{
    uint32_t a = state & 0xffff, b = state >> 16;
    for (size_t i = 0; i < n; i++) {
        a = (a + buf[i]) % 65521u;
        b = (a + b) % 65521u;
    }
    return (b << 16) | a;
}
uint32_t o_adler32(const uint8_t* buf, size_t n) { return o_adler32_feed(1, buf, n); }

/* zlib::Decoder::validate_header :55-86 and read :100-126 */
int o_zlib_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, size_t* in_used, uint32_t* flags)
{
    *out_len = 0;
    if (in_used) *in_used = 0;
    if (flags) *flags = 0;
    if (n < 1) return RCX_E_EOF;
    uint8_t cmf = in[0];
    if (in_used) *in_used = 1;
    if (n < 2) return RCX_E_EOF;
    uint8_t flg = in[1];
    if (in_used) *in_used = 2;                                        /* both bytes are read before the checks */
    if ((cmf & 0xf) != 0x8) return RCX_E_ZLIB_FORMAT;                 /* :58 */
    if ((cmf & 0xf0) != 0x70) return RCX_E_ZLIB_WINDOW;               /* :65 */
    if (flg & 0x20) return RCX_E_ZLIB_DICT;                           /* :72 */
    if ((((unsigned)cmf) * 256 + flg) % 31 != 0) return RCX_E_ZLIB_HEADER_CHECKSUM;   /* :79 */
    size_t used = 0;
    int st = o_inflate(in + 2, n - 2, out, cap, out_len, &used, flags);
    used += 2;
    if (in_used) *in_used = used;
    if (st) return st;
    if (n - used < 4) return RCX_E_EOF;                               /* :109 read_u32::<BigEndian> */
    uint32_t ck = ((uint32_t)in[used] << 24) | ((uint32_t)in[used + 1] << 16) | ((uint32_t)in[used + 2] << 8) | in[used + 3];
    used += 4;
    if (in_used) *in_used = used;
    if (ck != o_adler32(out, *out_len)) return RCX_E_ZLIB_CHECKSUM;   /* :110-114 */
    return RCX_OK;
}
