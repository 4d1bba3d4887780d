// k_inflate.hip -- batched RFC-1951 inflate, RFC-1950 (zlib) unwrap and Adler-32 for gfx950.
//
// Replaces flate::Decoder::{block,statik,fixed,dynamic,codes} + HuffmanTree::{construct,decode}
// (src/flate.rs:83-146, :195-450), zlib::Decoder::{validate_header,read} (src/zlib.rs:55-126) and
// adler::State32 (src/checksum/adler.rs:22-51).  One LANE per stream, 64 streams per wave: Huffman
// decoding is a bit-serial chain per stream, so the width of the machine is spent on independent streams
// (BASELINE config 3 has 65 536 of them).  The canonical count/symbol tables of each stream live in LDS,
// lane-interleaved (entry e of lane t at [e*64+t]) so the 64 lanes of a wave hit 64 different banks.
// Acceptance and error statuses mirror the reference exactly (incomplete codes accepted, the n>29
// off-by-one, distance > history -> "invalid huffman code", ...); see oracle/o_flate.c.
// The Adler-32 of the zlib path is accumulated while bytes are produced (deferred modulo, same value as
// the reference's per-byte `%`).
#include "rcx_dev.h"

#define FL_MAXBITS 15
#define FL_MAXL 288          /* lit/len symbols incl. the fixed tree's 286/287 */
#define FL_MAXD 30
#define FL_HISTORY 32768u

#ifdef RCX_AB_VARIANTS            /* the first inflate kernel (one lane per stream, per-lane tables in LDS): A/B builds only */
struct FlTabs {              // per-lane views into the interleaved LDS tables
    uint16_t* lcount; uint16_t* lsym; uint16_t* dcount; uint16_t* dsym; uint8_t* lens; unsigned t;   // lens: per-lane private array (header parsing only)
    __device__ __forceinline__ uint16_t& LC(unsigned i) { return lcount[i * 64 + t]; }
    __device__ __forceinline__ uint16_t& LS(unsigned i) { return lsym[i * 64 + t]; }
    __device__ __forceinline__ uint16_t& DC(unsigned i) { return dcount[i * 64 + t]; }
    __device__ __forceinline__ uint16_t& DS(unsigned i) { return dsym[i * 64 + t]; }
    __device__ __forceinline__ uint8_t& LN(unsigned i) { return lens[i]; }
};

struct FlState {
    const uint8_t* in; uint64_t n, p;
    uint8_t* out; uint64_t cap, end;
    uint32_t bitbuf; uint32_t bitcnt;
    uint32_t a, b, pend;      // Adler-32 running sums (deferred modulo)
    __device__ __forceinline__ int bits(uint32_t cnt, uint32_t& ret)       // flate.rs:250-260
    {
        while (bitcnt < cnt) {
            if (p >= n) return RCX_E_EOF;
            bitbuf |= (uint32_t)in[p++] << bitcnt;
            bitcnt += 8;
        }
        ret = bitbuf & ((1u << cnt) - 1u);
        bitbuf >>= cnt;
        bitcnt -= cnt;
        return RCX_OK;
    }
    __device__ __forceinline__ void emit(uint8_t x)
    {
        out[end++] = x;
        a += x; b += a;
        if (++pend == 5552) { a %= 65521u; b %= 65521u; pend = 0; }
    }
};

// HuffmanTree::construct, flate.rs:83-120.  which: 0 = lit/len table, 1 = distance table.
// lens are read from T.LN(base + i).
__device__ int fl_construct(FlTabs& T, int which, unsigned base, unsigned nlens)
{
    uint16_t cnt[FL_MAXBITS + 1];
#pragma unroll
    for (int i = 0; i <= FL_MAXBITS; i++) cnt[i] = 0;
    for (unsigned i = 0; i < nlens; i++) {
        const unsigned l = T.LN(base + i);
#pragma unroll
        for (int k = 0; k <= FL_MAXBITS; k++) cnt[k] += (l == (unsigned)k) ? 1 : 0;
    }
#pragma unroll
    for (int i = 0; i <= FL_MAXBITS; i++) { if (which) T.DC(i) = cnt[i]; else T.LC(i) = cnt[i]; }
    if (cnt[0] == nlens) return RCX_OK;                                   // :93
    int left = 1;                                                         // :98-103
#pragma unroll
    for (int i = 1; i <= FL_MAXBITS; i++) {
        left = left * 2 - (int)cnt[i];
        if (left < 0) return RCX_E_INVALID_HUFFMAN_TREE;
    }
    uint16_t offs[FL_MAXBITS + 1];
    offs[0] = 0; offs[1] = 0;
#pragma unroll
    for (int i = 1; i < FL_MAXBITS; i++) offs[i + 1] = offs[i] + cnt[i];  // :106-109
    for (unsigned sym = 0; sym < nlens; sym++) {                          // :113-118
        const unsigned l = T.LN(base + sym);
        if (l != 0) {
            uint16_t o = 0;
#pragma unroll
            for (int k = 1; k <= FL_MAXBITS; k++) { if (l == (unsigned)k) { o = offs[k]; offs[k] = o + 1; } }
            if (which) T.DS(o) = (uint16_t)sym; else T.LS(o) = (uint16_t)sym;
        }
    }
    return RCX_OK;
}

// HuffmanTree::decode, flate.rs:129-146 (bit-serial canonical walk)
__device__ __forceinline__ int fl_decode(FlTabs& T, int which, FlState& s, uint32_t& sym)
{
    uint32_t code = 0, first = 0, index = 0;
    for (int len = 1; len <= FL_MAXBITS; len++) {
        uint32_t bit;
        const int st = s.bits(1, bit);
        if (st) return st;
        code |= bit;
        const uint32_t count = which ? T.DC(len) : T.LC(len);
        if (code < ((first + count) & 0xffffu)) {
            const uint32_t idx = (index + (code - first)) & 0xffffu;
            sym = which ? T.DS(idx) : T.LS(idx);
            return RCX_OK;
        }
        index += count;
        first += count;
        first = (first << 1) & 0xffffu;
        code = (code << 1) & 0xffffu;
    }
    return RCX_E_NOT_ENOUGH_BITS;
}

__device__ const uint16_t FL_EXTRALENS[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51,
                                              59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t FL_EXTRABITS[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                             4, 5, 5, 5, 5, 0};
__device__ const uint16_t FL_EXTRADIST[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385,
                                              513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t FL_EXTRADBITS[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9,
                                              10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t FL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Decoder::codes, flate.rs:262-341
__device__ int fl_codes(FlTabs& T, FlState& s)
{
    for (;;) {
        uint32_t sym, x;
        int st = fl_decode(T, 0, s, sym);                                 // :287
        if (st) return st;
        if (sym < 256) {                                                  // :289
            if (s.end >= s.cap) return RCX_E_OUTPUT_TOO_SMALL;
            s.emit((uint8_t)sym);
        } else if (sym == 256) {
            return RCX_OK;                                                // :290
        } else if (sym < 290) {
            const uint32_t nn = sym - 257;
            if (nn > 29) return RCX_E_INVALID_HUFFMAN_CODE;               // :294 (off by one)
            if (nn == 29) return RCX_E_MALFORMED;                         // :297 index panic
            st = s.bits(FL_EXTRABITS[nn], x);
            if (st) return st;
            const uint32_t len = (uint32_t)FL_EXTRALENS[nn] + x;
            uint32_t d;
            st = fl_decode(T, 1, s, d);                                   // :302
            if (st) return st;
            if (d >= 30) return RCX_E_MALFORMED;
            st = s.bits(FL_EXTRADBITS[d], x);
            if (st) return st;
            const uint32_t dd = (uint32_t)FL_EXTRADIST[d] + x;
            const uint64_t hist = s.end < FL_HISTORY ? s.end : FL_HISTORY;   // output.len(), :314
            if (dd > hist) return RCX_E_INVALID_HUFFMAN_CODE;
            if (len > s.cap - s.end) return RCX_E_OUTPUT_TOO_SMALL;
            for (uint32_t i = 0; i < len; i++) s.emit(s.out[s.end - dd]);  // :320-334
        } else {
            return RCX_E_INVALID_HUFFMAN_CODE;                            // :336
        }
    }
}

// Decoder::statik, flate.rs:237-246
__device__ int fl_stored(FlState& s)
{
    if (s.n - s.p < 2) return RCX_E_EOF;
    const uint32_t len = (uint32_t)s.in[s.p] | ((uint32_t)s.in[s.p + 1] << 8); s.p += 2;
    if (s.n - s.p < 2) return RCX_E_EOF;
    const uint32_t nlen = (uint32_t)s.in[s.p] | ((uint32_t)s.in[s.p + 1] << 8); s.p += 2;
    if (((~nlen) & 0xffffu) != len) return RCX_E_INVALID_STATIC_SIZE;     // :240
    if (s.n - s.p < len) return RCX_E_EOF;
    if (s.cap - s.end < len) return RCX_E_OUTPUT_TOO_SMALL;
    for (uint32_t i = 0; i < len; i++) s.emit(s.in[s.p + i]);
    s.p += len;
    s.bitcnt = 0; s.bitbuf = 0;                                           // :243-244
    return RCX_OK;
}

// Decoder::fixed, flate.rs:343-395 (tables = construct() of the RFC lengths, as :149-160 generated them)
__device__ int fl_fixed(FlTabs& T, FlState& s)
{
    for (unsigned i = 0; i < 144; i++) T.LN(i) = 8;
    for (unsigned i = 144; i < 256; i++) T.LN(i) = 9;
    for (unsigned i = 256; i < 280; i++) T.LN(i) = 7;
    for (unsigned i = 280; i < 288; i++) T.LN(i) = 8;
    fl_construct(T, 0, 0, 288);
    for (unsigned i = 0; i < FL_MAXD; i++) T.LN(i) = 5;
    fl_construct(T, 1, 0, FL_MAXD);
    return fl_codes(T, s);
}

// Decoder::dynamic, flate.rs:397-450
__device__ int fl_dynamic(FlTabs& T, FlState& s)
{
    uint32_t x;
    int st;
    if ((st = s.bits(5, x))) return st;
    const uint32_t hlit = x + 257;
    if ((st = s.bits(5, x))) return st;
    const uint32_t hdist = x + 1;
    if ((st = s.bits(4, x))) return st;
    const uint32_t hclen = x + 4;
    if (hlit > 286 || hdist > 30) return RCX_E_HUFFMAN_TREE_TOO_LARGE;    // :401
    for (unsigned i = 0; i < 19; i++) T.LN(i) = 0;
    for (unsigned i = 0; i < hclen; i++) {                                // :412-414
        if ((st = s.bits(3, x))) return st;
        T.LN(FL_ORDER[i]) = (uint8_t)x;
    }
    // the code-length tree is built in the DISTANCE table slots (19 <= 30 symbols), then replaced
    if ((st = fl_construct(T, 1, 0, 19))) return st;                      // :415
    for (unsigned i = 0; i < 316; i++) T.LN(i) = 0;                       // :419
    uint32_t i = 0;
    while (i < hlit + hdist) {                                            // :421-441
        uint32_t symbol;
        if ((st = fl_decode(T, 1, s, symbol))) return st;
        if (symbol < 16) {
            T.LN(i++) = (uint8_t)symbol;
        } else if (symbol == 16) {
            if (i == 0) return RCX_E_INVALID_HUFFMAN_HEADER_SYMBOL;       // :428
            const uint8_t prev = T.LN(i - 1);
            if ((st = s.bits(2, x))) return st;
            const uint32_t rep = x + 3;
            for (uint32_t k = 0; k < rep; k++) {
                if (i >= 316) return RCX_E_MALFORMED;                     // :432 index panic
                T.LN(i++) = prev;
            }
        } else if (symbol == 17) {
            if ((st = s.bits(3, x))) return st;
            i += x + 3;
        } else if (symbol == 18) {
            if ((st = s.bits(7, x))) return st;
            i += x + 11;
        } else {
            return RCX_E_INVALID_HUFFMAN_HEADER_SYMBOL;                   // :439
        }
    }
    if (i > hlit + hdist) return RCX_E_INVALID_HUFFMAN_TREE_HEADER;       // :442
    if ((st = fl_construct(T, 0, 0, hlit))) return st;                    // :445-446
    if ((st = fl_construct(T, 1, hlit, hdist))) return st;                // :447-448
    return fl_codes(T, s);
}

// LDS per 64-stream wave: lcount 16 + lsym 288 + dcount 16 + dsym 30 = 350 u16 per lane (44 800 B)
#define FL_WORDS (16 + FL_MAXL + 16 + FL_MAXD)

__global__ __launch_bounds__(64) void k_inflate(rcx_kargs a, int zlib)
{
    __shared__ uint16_t s_tab[FL_WORDS * 64];
    const unsigned t = threadIdx.x;
    const uint32_t b = blockIdx.x * 64 + t;
    if (b >= a.nblocks) return;
    FlTabs T;
    T.lcount = s_tab; T.lsym = s_tab + 16 * 64; T.dcount = s_tab + (16 + FL_MAXL) * 64;
    T.dsym = s_tab + (32 + FL_MAXL) * 64; T.t = t;
    uint8_t lens_priv[320];
    T.lens = lens_priv;
    FlState s;
    s.in = a.in_base + a.in_off[b]; s.n = a.in_len[b]; s.p = 0;
    s.out = a.out_base + a.out_off[b]; s.cap = a.out_cap[b]; s.end = 0;
    s.bitbuf = 0; s.bitcnt = 0; s.a = 1; s.b = 0; s.pend = 0;
    int st = RCX_OK;
    uint32_t flags = 0;
    if (zlib) {                                                           // validate_header, zlib.rs:55-86
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
    while (!st && !eof) {                                                 // Decoder::block :195-206, to BFINAL
        uint32_t x;
        const uint64_t before = s.end;
        if ((st = s.bits(1, x))) break;
        if (x == 1) eof = true;                                           // :198
        if ((st = s.bits(2, x))) break;                                   // :199
        if (x == 0) st = fl_stored(s);
        else if (x == 1) st = fl_fixed(T, s);
        else if (x == 2) st = fl_dynamic(T, s);
        else st = RCX_E_INVALID_BLOCK_CODE;                               // :203
        if (!st && s.end == before && !eof) flags |= RCX_W_EMPTY_BLOCK_MIDSTREAM;   // :474-476 quirk
    }
    if (zlib && !st) {                                                    // zlib.rs:108-118
        if (s.n - s.p < 4) st = RCX_E_EOF;
        else {
            const uint32_t ck = ((uint32_t)s.in[s.p] << 24) | ((uint32_t)s.in[s.p + 1] << 16) |
                                ((uint32_t)s.in[s.p + 2] << 8) | (uint32_t)s.in[s.p + 3];
            s.p += 4;
            const uint32_t mine = ((s.b % 65521u) << 16) | (s.a % 65521u);
            if (ck != mine) st = RCX_E_ZLIB_CHECKSUM;
        }
    }
    a.status[b] = st;
    a.out_len[b] = s.end;
    if (a.in_used) a.in_used[b] = s.p;
    if (a.aux) a.aux[b] = flags;
}

// adler::State32 over whole blocks, one wave per block: lane l sums a contiguous slice, slices are
// combined with the closed form  a = 1 + sum(x_i),  b = n + sum((n - i) * x_i)   (mod 65521).
#endif  // RCX_AB_VARIANTS

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_adler32(rcx_kargs a)
{
    const unsigned w = threadIdx.x >> 6, lane = rcx_lane();
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));
    if (b >= a.nblocks) return;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint64_t n = a.in_len[b];
    // With a starting at 0:  a = sum x_i,  b = sum (n - i) x_i.  The wave reads 1 KiB per step, lane l the 16-byte chunk at
    // p = 1024 * step + 16 * l (coalesced: a slice per lane read 16 of every 256 bytes per request and ran at a quarter of
    // the bandwidth); a chunk adds  A_c = sum x_j  to a and  (n - p - 16) * A_c + sum (16 - j) x_j  to b.
    const uint64_t nch = n >> 4;
    uint64_t SA = 0, SB = 0;
    uint32_t r = (uint32_t)((n - 16ull * lane - 16ull) % 65521ull);     // (n - p - 16) mod 65521 for the lane's first chunk (wraps harmlessly when n < 16)
    for (uint64_t c = lane; c < nch; c += 64) {
        const rcx_u32x4 v = *(const rcx_u32x4_u*)(in + 16 * c);
        uint32_t A = 0, W = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t x = v[k];
#pragma unroll
            for (int j = 0; j < 4; j++) { A += x & 0xffu; W += A; x >>= 8; }          // W = sum (16 - j) x_j
        }
        SA += A; SB += (uint64_t)W + (uint64_t)r * A;                    // < 2^29 per chunk: no overflow in 2^35 chunks
        r = r >= 1024u ? r - 1024u : r + 65521u - 1024u;
    }
    uint32_t sa = (uint32_t)(SA % 65521ull), sb = (uint32_t)(SB % 65521ull);
    if (lane == 0) {                                                     // the last n % 16 bytes
        uint32_t A = 0, B = 0;
        for (uint64_t i = nch << 4; i < n; i++) { A += in[i]; B += (uint32_t)(n - i) * in[i]; }
        sa = (sa + A) % 65521u; sb = (sb + B) % 65521u;
    }
    sa = rcx_wave_sum(sa); sb = rcx_wave_sum(sb);                        // 64 values < 65521: no overflow in u32
    if (lane == 0) {
        const uint32_t ra = (1u + sa) % 65521u;
        const uint32_t rb = (uint32_t)((n % 65521u + sb) % 65521u);
        if (a.aux) a.aux[b] = (rb << 16) | ra;
        if (a.status) a.status[b] = RCX_OK;
        if (a.out_len) a.out_len[b] = 0;
        if (a.in_used) a.in_used[b] = n;
    }
}

// (the launch bounds must stand on these first declarations: put on the definitions only, they were silently dropped and the
// kernels were compiled for 1024-thread workgroups -- no occupancy target reached the register allocator)
#ifndef INF3_OCC
#define INF3_OCC 6
#endif
template <int SPW, int LG, int MINW> __global__ __launch_bounds__(64, MINW) void k_inflate2(rcx_kargs a, int zlib);
template <int CB, bool SPEC = false, bool ADLER = false, bool MIRROR = false> __global__ __launch_bounds__(64, INF3_OCC) void k_inflate3(rcx_kargs a, int zlib);
__global__ void k_zlib_tail3(rcx_kargs a, const uint32_t* adler);
template <int WAVES> __global__ void k_adler32(rcx_kargs a);

// one lane per stream (k_inflate2); `flags`: bit 0 zlib framing, bit 1 only the blocks k_inflate3 handed back
static void launch_inflate2(hipStream_t s, rcx_kargs& k, int flags, int v)
{
    // streams per wave: the largest of 32/16/8 that still gives 2048 waves (measured: benchmarks/inflate_spw_sweep.py;
    // full waves are never the fastest); variants 2..5 pin it (A/B)
    const uint32_t n = k.nblocks;
    int spw = v == 2 ? 64 : v == 3 ? 32 : v == 4 ? 16 : v == 5 ? 8 : (n >= 32u * 2048u ? 32 : n >= 16u * 2048u ? 16 : 8);
    const int z = flags;
#ifdef RCX_AB_VARIANTS
    if (v == 6) { hipLaunchKernelGGL((k_inflate2<16, 4, 4>), dim3((n + 15) / 16), dim3(16), 0, s, k, z); return; }
    if (v == 7) { hipLaunchKernelGGL((k_inflate2<32, 5, 3>), dim3((n + 31) / 32), dim3(32), 0, s, k, z); return; }
    if (v == 8) { hipLaunchKernelGGL((k_inflate2<16, 4, 3>), dim3((n + 15) / 16), dim3(16), 0, s, k, z); return; }
    if (spw == 64) { hipLaunchKernelGGL((k_inflate2<64, 6, 1>), dim3((n + 63) / 64), dim3(64), 0, s, k, z); return; }
#endif
    if (spw >= 32) hipLaunchKernelGGL((k_inflate2<32, 5, 1>), dim3((n + 31) / 32), dim3(32), 0, s, k, z);
    else if (spw == 16) hipLaunchKernelGGL((k_inflate2<16, 4, 1>), dim3((n + 15) / 16), dim3(16), 0, s, k, z);
    else hipLaunchKernelGGL((k_inflate2<8, 3, 1>), dim3((n + 7) / 8), dim3(8), 0, s, k, z);
}

static constexpr uint32_t INF3_MAX_STREAMS = 0xffffffffu;   // every batch size measured (1024 .. 65536 streams) is faster wave-per-stream
// scratch the default path wants: Adler-32 values (zlib) and a stand-in for a null in_used
static uint64_t inflate_scratch_bytes(uint32_t nblocks) { return 13ull * nblocks + 512; }
static uint64_t inflate_scratch_min(uint32_t nblocks) { return 12ull * nblocks + 256; }      // (without the marks of a mirrored launch: what rcx_scratch_bytes asked for before them)
// ... and, for a mirrored launch, which streams the first pass handed back (their bytes reach the caller's buffer by a copy): a count
// in the first word, a byte per stream from byte 64 on
static uint64_t inflate_marks_offset(uint32_t nblocks) { return (12ull * nblocks + 256 + 63) & ~63ull; }
__global__ void k_inflate_mark(const int32_t* status, uint32_t n, uint8_t* marks)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const bool fb = status[b] == (int32_t)0x7ff00001;          // RCX_ST_FALLBACK (k_inflate3.hip)
    marks[64 + b] = fb ? 1 : 0;
    if (fb) atomicAdd((uint32_t*)marks, 1u);
}

// Variant 0: one wave per stream (k_inflate3), then k_inflate2 over the blocks it handed back (every error status and
// every unusual stream comes from the kernel that reproduces the reference case by case).  Variants 1..8: k_inflate /
// k_inflate2 only; 9: k_inflate2 with its automatic geometry.  Without the scratch (older callers) variant 0 is 9.
static void launch_inflate(hipStream_t s, rcx_kargs& k, bool zlib, int v)
{
    const uint32_t n = k.nblocks;
#ifdef RCX_AB_VARIANTS
    if (v == 1) { hipLaunchKernelGGL(k_inflate, dim3((n + 63) / 64), dim3(64), 0, s, k, zlib ? 1 : 0); return; }   // first version (A/B)
#endif
    // one wave per stream (16 waves per CU) against one lane per stream: 65536 streams of 16 KiB 28 vs 34 ms, 4096 streams
    // 2.2 vs 20.7 ms (benchmarks/inflate_spw_sweep.py); variant 10 forces it, 9 forces k_inflate2
    // The wave-per-stream kernel's symbol pass is the speculative one (Inf3::pass4: a lane per 128-bit segment, 11.7 against
    // 12.4 ms for config 3); 11 / 10: the window pass (Inf3::pass) with / without the second pass, 12: pass4 without it
    const bool spec = v == 0 || v == 12;
    const bool wave_per_stream = v == 10 || v == 11 || v == 12 || (v == 0 && n < INF3_MAX_STREAMS);
    // (k.out_mirror tells the caller what happened: cleared wherever the launch did NOT store into the caller's buffer, so that the plain
    //  copy follows -- rcx_api.hip asks inflate_mirrors() before it cuts the input into gated ranges, which only the mirroring kernel honours)
    if (!wave_per_stream || k.scratch == nullptr || k.scratch_bytes < inflate_scratch_min(n)) { k.out_mirror = nullptr; launch_inflate2(s, k, zlib ? 1 : 0, (v >= 9 && v <= 12) ? 0 : v); return; }
    rcx_kargs k3 = k;
    uint32_t* adler = (uint32_t*)k.scratch;
    if (!k3.in_used) k3.in_used = (uint64_t*)((uint8_t*)k.scratch + ((4ull * n + 63) & ~63ull));
    // zlib: the decoder sums the Adler-32 of what it writes itself (k_inflate3<.., true>; the separate k_adler32 pass over the
    // output was 1.08 of the launch's 4.4 GB of HBM traffic for config 3) and k_zlib_tail3 compares it with the trailer
    const bool mirror = k.out_mirror != nullptr && spec && k.scratch_bytes >= inflate_scratch_bytes(n);      // (rcx_api.hip sizes the scratch for it)
    if (!mirror) { k3.out_mirror = nullptr; k.out_mirror = nullptr; }
    if (zlib) {
        k3.scratch = adler;                                    // (the kernel's slot array: 4 bytes a stream)
        if (mirror) hipLaunchKernelGGL((k_inflate3<1024, true, true, true>), dim3(n), dim3(64), 0, s, k3, 1);
        else if (spec) hipLaunchKernelGGL((k_inflate3<1024, true, true>), dim3(n), dim3(64), 0, s, k3, 1);
        else hipLaunchKernelGGL((k_inflate3<1024, false, true>), dim3(n), dim3(64), 0, s, k3, 1);
        hipLaunchKernelGGL(k_zlib_tail3, dim3((n + 255) / 256), dim3(256), 0, s, k3, adler);
    } else {
        if (mirror) hipLaunchKernelGGL((k_inflate3<1024, true, false, true>), dim3(n), dim3(64), 0, s, k3, 0);
        else if (spec) hipLaunchKernelGGL((k_inflate3<1024, true, false>), dim3(n), dim3(64), 0, s, k3, 0);
        else hipLaunchKernelGGL((k_inflate3<1024, false, false>), dim3(n), dim3(64), 0, s, k3, 0);
    }
    if (mirror) {
        uint8_t* marks = (uint8_t*)k.scratch + inflate_marks_offset(n);
        (void)hipMemsetAsync(marks, 0, 64, s);
        hipLaunchKernelGGL(k_inflate_mark, dim3((n + 255) / 256), dim3(256), 0, s, k.status, n, marks);
    }
    if (v != 10 && v != 12) launch_inflate2(s, k, (zlib ? 1 : 0) | 2, 0);                      // 10: A/B, shows what the first pass handed back
}
// will launch_inflate(variant v) over n streams store into a page-locked output buffer (given the scratch rcx_scratch_bytes asks for)?
static bool inflate_mirrors(uint32_t n, int v) { return (v == 0 && n < INF3_MAX_STREAMS) || v == 12; }
static void launch_adler32(hipStream_t s, rcx_kargs& k)
{
    hipLaunchKernelGGL((k_adler32<4>), dim3((k.nblocks + 3) / 4), dim3(256), 0, s, k);
}
