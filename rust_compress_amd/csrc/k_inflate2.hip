// k_inflate2.hip -- second inflate kernel (the production one; k_inflate.hip keeps the first, simple
// version for A/B).  Same mapping -- one LANE per stream, 64 streams per wave, because BASELINE config 3
// is 65 536 independent members and Huffman decoding is a serial chain per stream -- but every inner step
// was rebuilt around what made v1 slow on MI355X (13 GiB/s): an LDS round trip per code BIT, a global
// byte load per input byte, 64 scattered global byte stores per output byte, 3 waves per CU.
//   * canonical decode without memory: per tree, 15 left-justified limits and 15 index bases live in
//     registers; a symbol is `brev` + 15 compare/adds (its length) + 15 selects (its base) + ONE LDS read;
//     bit-exact with the reference's bit-serial walk incl. incomplete codes (code_len < first+count  <=>
//     rev15 < (first+count) << (15-len)); the exact serial walk is kept for the last bytes of a stream so
//     EOF / "not enough bits" statuses and in_used stay identical to src/flate.rs:129-146, :250-260;
//   * 64-bit bit buffer refilled 4 bytes at a time;
//   * output goes to a per-lane 128-byte LDS ring (word-interleaved across lanes: conflict-free) and is
//     drained to HBM 16 bytes at a time; near matches copy inside the ring, far matches use one 16-byte
//     gather per 16 bytes (the source is always drained: distance >= 112);
//   * symbols are u8 + a 9th-bit bitmap: 31 KB of LDS per wave -> 5 waves per CU.
// Reference citations as in k_inflate.hip (src/flate.rs:83-146, :195-450; src/zlib.rs:55-126).
#include "rcx_dev.h"

#define F2_W 128u                 /* ring bytes per lane */
#define F2_NEAR (F2_W - 16u)      /* distances below this are served from the ring */

typedef uint32_t __attribute__((aligned(1))) rcx_u32_un;

struct F2Huff {                   // one canonical code, register resident
    uint32_t lim[16];             // lim[k] = (first[k] + count[k]) << (15-k), k = 1..15
    int32_t base[16];             // base[k] = offs[k] - first[k]
};

struct F2 {
    // LDS views (lane-interleaved)
    uint8_t* lsym; uint32_t* lbit; uint8_t* dsym; uint32_t* ring; unsigned t;
    uint32_t lg;                      // log2(streams per wave): the interleave stride of the LDS views
    // stream state
    const uint8_t* in; uint64_t n, p;
    uint8_t* out; uint64_t cap, end, flushed;
    uint64_t bb; uint32_t bc;
    uint32_t nx; bool nxv;            // prefetched input word at p
    uint32_t a, b, pend, omis;

    __device__ __forceinline__ uint8_t& LS(uint32_t i) { return lsym[(i << lg) + t]; }
    __device__ __forceinline__ uint32_t& LB(uint32_t i) { return lbit[(i << lg) + t]; }
    __device__ __forceinline__ uint8_t& DS(uint32_t i) { return dsym[(i << lg) + t]; }
    __device__ __forceinline__ uint8_t* RB(uint64_t pos)          // ring byte of absolute output position
    {
        const uint32_t q = (uint32_t)(pos + omis) & (F2_W - 1u);
        return (uint8_t*)ring + ((((q >> 2) << lg) + t) << 2) + (q & 3u);
    }

    // ---- input bits ------------------------------------------------------------------------------
    __device__ __forceinline__ void refill()
    {
        while (bc <= 32 && p < n) {
            if (n - p >= 4) {
                if (!nxv) nx = *(const rcx_u32_un*)(in + p);
                bb |= (uint64_t)nx << bc; bc += 32; p += 4;
                nxv = n - p >= 4;                                  // prefetch the following word: by the time the
                if (nxv) nx = *(const rcx_u32_un*)(in + p);       // next refill needs it the ~1 us load has landed
            }
            else { bb |= (uint64_t)in[p] << bc; bc += 8; p += 1; }
        }
    }
    __device__ __forceinline__ int bits(uint32_t cnt, uint32_t& ret)      // flate.rs:250-260
    {
        if (bc < cnt) { refill(); if (bc < cnt) { bc = 0; return RCX_E_EOF; } }   // every byte was consumed
        ret = (uint32_t)bb & ((1u << cnt) - 1u);
        bb >>= cnt; bc -= cnt;
        return RCX_OK;
    }
    __device__ __forceinline__ uint64_t used() const { return p - (bc >> 3); }

    // ---- output ------------------------------------------------------------------------------------
    __device__ __forceinline__ void drain(uint64_t upto)               // ring -> HBM for [flushed, upto)
    {
        while (flushed < upto) {
            const uint64_t pos = flushed;
            if ((((uintptr_t)(out + pos)) & 15u) == 0 && upto - pos >= 16) {
                const uint32_t q = (uint32_t)(pos + omis) & (F2_W - 1u);
                const uint32_t w0 = ((q >> 2) << lg) + t, sw = 1u << lg;
                rcx_u32x4 v = {ring[w0], ring[w0 + sw], ring[w0 + 2 * sw], ring[w0 + 3 * sw]};
                *(rcx_u32x4*)(out + pos) = v;
                flushed += 16;
            } else { out[pos] = *RB(pos); flushed += 1; }
        }
    }
    __device__ __forceinline__ void emit(uint8_t x)
    {
        *RB(end) = x;
        end++;
        a += x; b += a;
        if (++pend == 5552) { a %= 65521u; b %= 65521u; pend = 0; }
        if ((((uintptr_t)(out + end)) & 15u) == 0) drain(end);
    }

    // ---- tables --------------------------------------------------------------------------------------
    // HuffmanTree::construct, flate.rs:83-120, producing the register form + symbols in LDS.
    template <int WHICH>    // 0: lit/len -> lsym+lbit, 1: dist / code-length -> dsym
    __device__ int construct(F2Huff& H, const uint8_t* lens, uint32_t nlens, bool& empty)
    {
        uint32_t cnt[16];
#pragma unroll
        for (int k = 0; k < 16; k++) cnt[k] = 0;
        for (uint32_t i = 0; i < nlens; i++) {
            const uint32_t l = lens[i];
#pragma unroll
            for (int k = 0; k < 16; k++) cnt[k] += (l == (uint32_t)k) ? 1u : 0u;
        }
        empty = cnt[0] == nlens;                                           // :93 no codes at all
#pragma unroll
        for (int k = 0; k < 16; k++) { H.lim[k] = 0; H.base[k] = 0; }
        if (empty) return RCX_OK;
        int left = 1;                                                      // :98-103
#pragma unroll
        for (int k = 1; k <= 15; k++) { left = left * 2 - (int)cnt[k]; if (left < 0) return RCX_E_INVALID_HUFFMAN_TREE; }
        uint32_t offs[16];
        uint32_t first = 0, o = 0;
#pragma unroll
        for (int k = 1; k <= 15; k++) {
            offs[k] = o;
            H.lim[k] = (first + cnt[k]) << (15 - k);
            H.base[k] = (int32_t)o - (int32_t)first;
            o += cnt[k];
            first = (first + cnt[k]) << 1;
        }
        if (WHICH == 0) for (uint32_t i = 0; i < 9; i++) LB(i) = 0;
        for (uint32_t sym = 0; sym < nlens; sym++) {                       // :113-118
            const uint32_t l = lens[sym];
            if (l != 0) {
                uint32_t at = 0;
#pragma unroll
                for (int k = 1; k <= 15; k++) { if (l == (uint32_t)k) { at = offs[k]; offs[k] = at + 1; } }
                if (WHICH == 0) { LS(at) = (uint8_t)sym; if (sym & 256u) LB(at >> 5) |= 1u << (at & 31u); }
                else DS(at) = (uint8_t)sym;
            }
        }
        return RCX_OK;
    }

    // HuffmanTree::decode, flate.rs:129-146
    template <int WHICH>
    __device__ __forceinline__ int decode(const F2Huff& H, uint32_t& sym)
    {
        if (bc < 15) refill();
        if (bc >= 15) {                                                    // register-only canonical decode
            const uint32_t rev = __brev((uint32_t)bb) >> 17;
            uint32_t len = 1;
#pragma unroll
            for (int k = 1; k <= 15; k++) len += (rev >= H.lim[k]) ? 1u : 0u;
            if (len > 15) { bb >>= 15; bc -= 15; return RCX_E_NOT_ENOUGH_BITS; }
            int32_t base = 0;
#pragma unroll
            for (int k = 1; k <= 15; k++) base = (len == (uint32_t)k) ? H.base[k] : base;
            const uint32_t idx = (rev >> (15u - len)) + (uint32_t)base;
            if (WHICH == 0) sym = (uint32_t)LS(idx) | (((LB(idx >> 5) >> (idx & 31u)) & 1u) << 8);
            else sym = DS(idx);
            bb >>= len; bc -= len;
            return RCX_OK;
        }
        // the last < 15 bits of the input: the reference's bit-serial walk, bit for bit
        uint32_t code = 0;
#pragma unroll 1
        for (int k = 1; k <= 15; k++) {
            uint32_t bit;
            const int st = bits(1, bit);
            if (st) return st;
            code = (code << 1) | bit;
            if ((code << (15 - k)) < H.lim[k] ) {
                int32_t base = 0;
#pragma unroll
                for (int j = 1; j <= 15; j++) base = (k == j) ? H.base[j] : base;
                const uint32_t idx = code + (uint32_t)base;
                if (WHICH == 0) sym = (uint32_t)LS(idx) | (((LB(idx >> 5) >> (idx & 31u)) & 1u) << 8);
                else sym = DS(idx);
                return RCX_OK;
            }
        }
        return RCX_E_NOT_ENOUGH_BITS;
    }
};

__device__ const uint16_t F2_EXTRALENS[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51,
                                              59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t F2_EXTRABITS[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                             4, 5, 5, 5, 5, 0};
__device__ const uint16_t F2_EXTRADIST[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385,
                                              513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t F2_EXTRADBITS[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9,
                                              10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t F2_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Decoder::codes, flate.rs:262-341, as a per-lane state machine.  The 64 lanes of a wave run 64 different
// streams, so the loop body is written to be executed by everybody: ONE "emit up to 4 pending bytes" path
// (literal, ring source with period handling for distances 1-3, or 4 bytes of a 16-byte far gather) and ONE
// "decode the next symbol" path.  (rocprof on the straightforward per-symbol loop: ~1250 wave instructions
// per decoded symbol, because every lane's literal waited for every other lane's inlined copy/drain code.)
__device__ int f2_codes(F2& s, const F2Huff& HL, const F2Huff& HD)
{
    uint32_t pend = 0, dd = 0, w = 0;          // pending bytes of the current symbol; dd == 0: the literal in w
    rcx_u32x4 g = {0, 0, 0, 0};                // far-match gather buffer
    uint32_t gpos = 16;
    for (;;) {
        if (pend) {                                                        // :289 / :320-334
            const uint32_t k = pend < 4 ? pend : 4;
            uint32_t w4;
            if (dd == 0) w4 = w;
            else if (dd < F2_NEAR) {                                       // ring source; bytes repeat with period dd < 4
                const uint32_t i1 = dd > 1 ? 1u : 0u;
                const uint32_t i2 = dd > 2 ? 2u : 0u;
                const uint32_t i3 = dd > 3 ? 3u : (dd == 2 ? 1u : 0u);
                const uint64_t sp = s.end - dd;
                const uint32_t b0 = *s.RB(sp), b1 = *s.RB(sp + i1), b2 = *s.RB(sp + i2), b3 = *s.RB(sp + i3);
                w4 = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            } else {                                                       // drained source: 16-byte gather, 4 bytes a step
                if (gpos >= 16) {
                    const uint64_t src = s.end - dd;
                    if (src + 16 <= s.cap) g = *(const rcx_u32x4_u*)(s.out + src);
                    else { uint32_t t4[4] = {0, 0, 0, 0}; for (uint32_t i = 0; i < 16 && src + i < s.end; i++) t4[i >> 2] |= (uint32_t)s.out[src + i] << (8 * (i & 3)); g = rcx_u32x4{t4[0], t4[1], t4[2], t4[3]}; }
                    gpos = 0;
                }
                w4 = gpos == 0 ? g[0] : gpos == 4 ? g[1] : gpos == 8 ? g[2] : g[3];
                gpos += 4;
            }
            const uint64_t e0 = s.end;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if ((uint32_t)u < k) { const uint32_t x = (w4 >> (8 * u)) & 0xffu; *s.RB(e0 + u) = (uint8_t)x; s.a += x; s.b += s.a; }
            }
            s.end = e0 + k; pend -= k;
            s.pend += k;
            if (s.pend >= 5548) { s.a %= 65521u; s.b %= 65521u; s.pend = 0; }
            const uint64_t al = s.end - (((uintptr_t)(s.out + s.end)) & 15u);   // last 16-byte boundary at or below end
            if (al > s.flushed && al <= s.end) s.drain(al);
        }
        if (pend) continue;
        uint32_t sym, x;
        int st = s.decode<0>(HL, sym);                                     // :287
        if (st) return st;
        if (sym < 256) {                                                   // :289
            if (s.end >= s.cap) return RCX_E_OUTPUT_TOO_SMALL;
            w = sym; dd = 0; pend = 1;
        } else if (sym == 256) {
            return RCX_OK;                                                 // :290
        } else if (sym < 290) {
            const uint32_t nn = sym - 257;
            if (nn > 29) return RCX_E_INVALID_HUFFMAN_CODE;                // :294 (off by one)
            if (nn == 29) return RCX_E_MALFORMED;                          // :297 index panic
            // EXTRALENS/EXTRABITS (:265-273) and EXTRADIST/EXTRADBITS (:275-284) in closed form: a per-lane table
            // lookup would be a dependent global load (~1 us) per match symbol
            const uint32_t lb = nn < 8 ? 0u : (nn == 28 ? 0u : (nn - 4u) >> 2);
            const uint32_t lbase = nn < 8 ? 3u + nn : (nn == 28 ? 258u : 3u + ((4u + (nn & 3u)) << lb));
            st = s.bits(lb, x);
            if (st) return st;
            const uint32_t len = lbase + x;
            uint32_t d;
            st = s.decode<1>(HD, d);                                       // :302
            if (st) return st;
            if (d >= 30) return RCX_E_MALFORMED;
            const uint32_t db = d < 4 ? 0u : (d - 2u) >> 1;
            const uint32_t dbase = d < 4 ? 1u + d : 1u + ((2u + (d & 1u)) << db);
            st = s.bits(db, x);
            if (st) return st;
            const uint32_t dist = dbase + x;
            const uint64_t hist = s.end < 32768u ? s.end : 32768u;         // output.len(), :314
            if (dist > hist) return RCX_E_INVALID_HUFFMAN_CODE;
            if (len > s.cap - s.end) return RCX_E_OUTPUT_TOO_SMALL;
            pend = len; dd = dist; gpos = 16;
            if (dist >= F2_NEAR && s.end - dist + 16 <= s.cap) {           // start the first gather now: its latency
                g = *(const rcx_u32x4_u*)(s.out + (s.end - dist));        // overlaps the other lanes' work
                gpos = 0;
            }
        } else {
            return RCX_E_INVALID_HUFFMAN_CODE;                             // :336
        }
    }
}

// Decoder::statik, flate.rs:237-246
__device__ int f2_stored(F2& s)
{
    s.p = s.used(); s.bb = 0; s.bc = 0; s.nxv = false;                     // the buffered bits are discarded
    if (s.n - s.p < 2) return RCX_E_EOF;
    const uint32_t len = (uint32_t)s.in[s.p] | ((uint32_t)s.in[s.p + 1] << 8); s.p += 2;
    if (s.n - s.p < 2) return RCX_E_EOF;
    const uint32_t nlen = (uint32_t)s.in[s.p] | ((uint32_t)s.in[s.p + 1] << 8); s.p += 2;
    if (((~nlen) & 0xffffu) != len) return RCX_E_INVALID_STATIC_SIZE;      // :240
    if (s.n - s.p < len) return RCX_E_EOF;
    if (s.cap - s.end < len) return RCX_E_OUTPUT_TOO_SMALL;
    for (uint32_t i = 0; i < len; i++) s.emit(s.in[s.p + i]);
    s.p += len;
    return RCX_OK;
}

__device__ int f2_fixed(F2& s, F2Huff& HL, F2Huff& HD, uint8_t* lens)
{
    for (unsigned i = 0; i < 144; i++) lens[i] = 8;
    for (unsigned i = 144; i < 256; i++) lens[i] = 9;
    for (unsigned i = 256; i < 280; i++) lens[i] = 7;
    for (unsigned i = 280; i < 288; i++) lens[i] = 8;
    bool e;
    s.construct<0>(HL, lens, 288, e);
    for (unsigned i = 0; i < 30; i++) lens[i] = 5;
    s.construct<1>(HD, lens, 30, e);
    return f2_codes(s, HL, HD);
}

// Decoder::dynamic, flate.rs:397-450
__device__ int f2_dynamic(F2& s, F2Huff& HL, F2Huff& HD, uint8_t* lens)
{
    uint32_t x;
    int st;
    if ((st = s.bits(5, x))) return st;
    const uint32_t hlit = x + 257;
    if ((st = s.bits(5, x))) return st;
    const uint32_t hdist = x + 1;
    if ((st = s.bits(4, x))) return st;
    const uint32_t hclen = x + 4;
    if (hlit > 286 || hdist > 30) return RCX_E_HUFFMAN_TREE_TOO_LARGE;     // :401
    for (unsigned i = 0; i < 19; i++) lens[i] = 0;
    for (unsigned i = 0; i < hclen; i++) {                                 // :412-414
        if ((st = s.bits(3, x))) return st;
        lens[F2_ORDER[i]] = (uint8_t)x;
    }
    bool e;
    if ((st = s.construct<1>(HD, lens, 19, e))) return st;                 // code-length code in the dist slots, :415
    for (unsigned i = 0; i < 320; i++) lens[i] = 0;                        // :419
    uint32_t i = 0;
    while (i < hlit + hdist) {                                             // :421-441
        uint32_t symbol;
        if ((st = s.decode<1>(HD, symbol))) return st;
        if (symbol < 16) {
            lens[i++] = (uint8_t)symbol;
        } else if (symbol == 16) {
            if (i == 0) return RCX_E_INVALID_HUFFMAN_HEADER_SYMBOL;        // :428
            const uint8_t prev = lens[i - 1];
            if ((st = s.bits(2, x))) return st;
            const uint32_t rep = x + 3;
            for (uint32_t k = 0; k < rep; k++) {
                if (i >= 316) return RCX_E_MALFORMED;                      // :432 index panic
                lens[i++] = prev;
            }
        } else if (symbol == 17) {
            if ((st = s.bits(3, x))) return st;
            i += x + 3;
        } else if (symbol == 18) {
            if ((st = s.bits(7, x))) return st;
            i += x + 11;
        } else {
            return RCX_E_INVALID_HUFFMAN_HEADER_SYMBOL;                    // :439
        }
    }
    if (i > hlit + hdist) return RCX_E_INVALID_HUFFMAN_TREE_HEADER;        // :442
    if ((st = s.construct<0>(HL, lens, hlit, e))) return st;               // :445-446
    if ((st = s.construct<1>(HD, lens + hlit, hdist, e))) return st;       // :447-448
    return f2_codes(s, HL, HD);
}

// LDS per stream: lsym 288 B + 9th-bit bitmap 9 words + dsym 32 B + ring 128 B = 484 B.
// SPW = streams (active lanes) per wave.  A lane's instruction stream is the union of what its wave's lanes do and a
// wave issues one instruction per >= 4 cycles, so with 65 536 members a full wave (SPW 64) means 4 waves per CU,
// one per SIMD, nothing to hide latency behind; SPW 16 gives 16 waves per CU with a quarter of the divergence each.
#define F2_LDS_PER_STREAM (288 + 9 * 4 + 32 + 128)

template <int SPW, int LG, int MINW>
__global__ __launch_bounds__(64, MINW) void k_inflate2(rcx_kargs a, int zlib)
{
    static_assert((1 << LG) == SPW && SPW <= 64, "streams per wave");
    __shared__ __align__(16) uint8_t s_mem[F2_LDS_PER_STREAM * SPW];
    const unsigned t = threadIdx.x;
    const uint32_t b = blockIdx.x * SPW + t;
    if (b >= a.nblocks) return;
    if ((zlib & 2) && a.status[b] != 0x7ff00001) return;                   // second pass: only what k_inflate3 handed back
    zlib &= 1;
    F2 s;
    s.lg = LG;
    s.lsym = s_mem; s.lbit = (uint32_t*)(s_mem + 288 * SPW); s.dsym = s_mem + 288 * SPW + 9 * 4 * SPW;
    s.ring = (uint32_t*)(s_mem + 288 * SPW + 9 * 4 * SPW + 32 * SPW); s.t = t;
    s.in = a.in_base + a.in_off[b]; s.n = a.in_len[b]; s.p = 0;
    s.out = a.out_base + a.out_off[b]; s.cap = a.out_cap[b]; s.end = 0; s.flushed = 0;
    s.bb = 0; s.bc = 0; s.nx = 0; s.nxv = false; s.a = 1; s.b = 0; s.pend = 0;
    s.omis = (uint32_t)((uintptr_t)s.out & 15u);
    uint8_t lens[320];
    F2Huff HL, HD;
    int st = RCX_OK;
    uint32_t flags = 0;
    if (zlib) {                                                            // validate_header, zlib.rs:55-86
        if (s.n < 2) { st = RCX_E_EOF; s.p = s.n; }
        else {
            const uint32_t cmf = s.in[0], flg = s.in[1];
            s.p = 2;
            if ((cmf & 0xf) != 0x8) st = RCX_E_ZLIB_FORMAT;
            else if ((cmf & 0xf0) != 0x70) st = RCX_E_ZLIB_WINDOW;
            else if (flg & 0x20) st = RCX_E_ZLIB_DICT;
            else if ((cmf * 256 + flg) % 31 != 0) st = RCX_E_ZLIB_HEADER_CHECKSUM;
        }
    }
    bool eof = false;
    while (!st && !eof) {                                                  // Decoder::block :195-206, to BFINAL
        uint32_t x;
        const uint64_t before = s.end;
        if ((st = s.bits(1, x))) break;
        if (x == 1) eof = true;                                            // :198
        if ((st = s.bits(2, x))) break;                                    // :199
        if (x == 0) st = f2_stored(s);
        else if (x == 1) st = f2_fixed(s, HL, HD, lens);
        else if (x == 2) st = f2_dynamic(s, HL, HD, lens);
        else st = RCX_E_INVALID_BLOCK_CODE;                                // :203
        if (!st && s.end == before && !eof) flags |= RCX_W_EMPTY_BLOCK_MIDSTREAM;   // :474-476 quirk
    }
    uint64_t used = s.used();                                              // a dry bit reader leaves p == n, bc == 0
    if (zlib && !st) {                                                     // zlib.rs:108-118
        uint64_t q = s.used();
        if (s.n - q < 4) st = RCX_E_EOF;
        else {
            const uint32_t ck = ((uint32_t)s.in[q] << 24) | ((uint32_t)s.in[q + 1] << 16) |
                                ((uint32_t)s.in[q + 2] << 8) | (uint32_t)s.in[q + 3];
            used = q + 4;
            const uint32_t mine = ((s.b % 65521u) << 16) | (s.a % 65521u);
            if (ck != mine) st = RCX_E_ZLIB_CHECKSUM;
        }
    }
    s.drain(s.end);                                                        // what was produced is delivered, error or not
    a.status[b] = st;
    a.out_len[b] = s.end;
    if (a.in_used) a.in_used[b] = used;
    if (a.aux) a.aux[b] = flags;
}
