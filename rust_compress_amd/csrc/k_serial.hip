// k_serial.hip -- the stream codecs whose inner loop is a serial chain per stream: MTF, DC, RLE and the
// adaptive byte range coder.  Parallelism comes from (a) many independent streams per launch and
// (b) whatever is wide inside one step: run detection/fill (64 bytes per step + ballot), the 256-entry
// MTF list search (4 entries per lane + ballot) and shift, RLE run emission (ballot + wave prefix sum).
//   MTF / DC / RLE : one wave per stream (state in LDS), runs of equal symbols skipped 64 bytes at a time
//   Ari            : one LANE per stream (64 streams per wave): the coder is a chain of u32 divides with a
//                    257-entry adaptive table per stream (LDS, lane-interleaved, 16-entry block sums)
#include "rcx_dev.h"
#include <type_traits>

// =================================================================================================
// MTF -- src/bwt/mtf.rs:63-90 with the stream codecs' identity start (:103-104, :141-142)
// =================================================================================================
// list[] (256 bytes, LDS).  find: lane l compares entries 4l..4l+3.
__device__ __forceinline__ uint32_t mtf_find(const uint8_t* lst, uint32_t sym, unsigned lane)
{
    const uint32_t w = *(const uint32_t*)(lst + 4 * lane);
    uint32_t hit = 4;
    if (((w >> 24) & 0xff) == sym) hit = 3;
    if (((w >> 16) & 0xff) == sym) hit = 2;
    if (((w >> 8) & 0xff) == sym) hit = 1;
    if ((w & 0xff) == sym) hit = 0;
    const unsigned long long m = __ballot(hit < 4);
    const int first = __ffsll(m) - 1;                       // symbols are unique in a well-formed list
    return 4u * (uint32_t)first + (uint32_t)__builtin_amdgcn_readlane(hit, first);
}
// lst[1..rank] = lst[0..rank-1]; lst[0] = sym     (rotate right by one, mtf.rs:68-78 / :85-89)
__device__ __forceinline__ void mtf_front(uint8_t* lst, uint32_t rank, uint32_t sym, unsigned lane)
{
    uint8_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t i = lane + 64u * k; v[k] = (i >= 1 && i <= rank) ? lst[i - 1] : (uint8_t)0; }
    rcx_wave_sync();
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t i = lane + 64u * k; if (i >= 1 && i <= rank) lst[i] = v[k]; }
    if (lane == 0) lst[0] = (uint8_t)sym;
    rcx_wave_sync();
}
// The same list held in registers: lane l keeps entries 4l .. 4l+3 in the bytes of one dword.  find needs no memory access
// and move-to-front is one DPP wave shift plus a few selects -- the LDS version paid three dependent LDS round trips and
// two wave syncs per symbol, which is what a serial MTF / DC step is made of.
struct MtfRegs {
    uint32_t w; unsigned lane;
    __device__ __forceinline__ void identity(unsigned lane_) { lane = lane_; const uint32_t b = 4u * lane; w = b | ((b + 1u) << 8) | ((b + 2u) << 16) | ((b + 3u) << 24); }
    __device__ __forceinline__ void zero(unsigned lane_) { lane = lane_; w = 0; }
    __device__ __forceinline__ uint32_t at(uint32_t pos) const          // pos uniform
    {
        return ((uint32_t)__builtin_amdgcn_readlane((int)w, (int)(pos >> 2)) >> (8u * (pos & 3u))) & 0xffu;
    }
    __device__ __forceinline__ void set(uint32_t pos, uint32_t sym)     // pos uniform
    {
        const uint32_t sh = 8u * (pos & 3u);
        w = lane == (pos >> 2) ? (w & ~(0xffu << sh)) | (sym << sh) : w;
    }
    __device__ __forceinline__ uint32_t find(uint32_t sym) const        // symbols are unique in a well-formed list
    {
        uint32_t hit = 4;
        if (((w >> 24) & 0xff) == sym) hit = 3;
        if (((w >> 16) & 0xff) == sym) hit = 2;
        if (((w >> 8) & 0xff) == sym) hit = 1;
        if ((w & 0xff) == sym) hit = 0;
        const unsigned long long m = __ballot(hit < 4);
        const int first = __ffsll(m) - 1;
        return 4u * (uint32_t)first + (uint32_t)__builtin_amdgcn_readlane((int)hit, first);
    }
    // entries 1..rank = old entries 0..rank-1, entry 0 = sym   (rotate right by one, mtf.rs:68-78 / :85-89)
    __device__ __forceinline__ void front(uint32_t rank, uint32_t sym)
    {
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x138, 0xf, 0xf, false);   // wave_shr:1: lane l-1's dword
        const uint32_t sh = (w << 8) | (prev >> 24);
        const uint32_t p0 = 4u * lane;
        uint32_t nw = w;
        if (p0 + 3u <= rank) nw = sh;
        else if (p0 <= rank) { const uint32_t mask = (1u << (8u * (rank - p0 + 1u))) - 1u; nw = (sh & mask) | (w & ~mask); }
        w = lane == 0 ? (nw & ~0xffu) | sym : nw;
    }
    // entries 0..rank-2 = old entries 1..rank-1, entry rank-1 = sym   (the DC decoder's move, dc.rs:219-227; rank >= 1)
    __device__ __forceinline__ void back(uint32_t rank, uint32_t sym)
    {
        const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x130, 0xf, 0xf, false);    // wave_shl:1: lane l+1's dword
        const uint32_t sh = (w >> 8) | (nxt << 24);
        const uint32_t p0 = 4u * lane;
        uint32_t nw = w;
        if (rank >= 2u) {
            const uint32_t lastq = rank - 2u;                  // last position that takes its right neighbour
            if (p0 + 3u <= lastq) nw = sh;
            else if (p0 <= lastq) { const uint32_t mask = (1u << (8u * (lastq - p0 + 1u))) - 1u; nw = (sh & mask) | (w & ~mask); }
        }
        const uint32_t pos = rank - 1u, bsh = 8u * (pos & 3u);
        w = lane == (pos >> 2) ? (nw & ~(0xffu << bsh)) | (sym << bsh) : nw;
    }
};

// Sequential input of one stream held across the lanes (lane j: element base + j), 128 elements per refill: a serial loop that
// did `x = in[i]` paid a full global-memory round trip (~700 ns) per step (34 ms for one 256 KiB block of BWT output, whatever
// the batch size); readlane from the window costs a few cycles.  The refill WAITS for its two loads on the spot (once per 128
// elements, ~1 us): a load left in flight across iterations makes the compiler put `s_waitcnt vmcnt(0)` in front of every use
// of the window in the caller's loop, and vmcnt counts the caller's STORES too -- each step of a serial MTF / DC loop then waited
// for the stores of the step before to reach the L2 (~1000 cycles per step: that was the "25 ms per block" floor of round 1).
template <typename T>
struct SeqWin {
    const T* in; uint32_t n, base; uint32_t cur, nxt; unsigned lane; bool have;
    __device__ __forceinline__ void refill(uint32_t b)
    {
        base = b;
        cur = b + lane < n ? (uint32_t)in[b + lane] : 0u;
        nxt = b + 64u + lane < n ? (uint32_t)in[b + 64u + lane] : 0u;
        RCX_WAIT_VMEM();
        have = true;
    }
    __device__ __forceinline__ void start(const T* in_, uint32_t n_, unsigned lane_) { in = in_; n = n_; lane = lane_; refill(0); }
    __device__ __forceinline__ void seek(uint32_t i)           // make element i (uniform, >= base) addressable
    {
        if (i < base + 64u) return;
        if (have && i < base + 128u) { base += 64u; cur = nxt; have = false; return; }
        refill(i & ~63u);
    }
    __device__ __forceinline__ uint32_t get(uint32_t i)        // i uniform, i < n
    {
        seek(i);
        return (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(i - base));
    }
    // length of the run of elements equal to `sym` starting at i (element i is known to equal sym)
    __device__ __forceinline__ uint32_t run(uint32_t i, uint32_t sym)
    {
        seek(i);
        const uint32_t p = base + lane;
        const unsigned long long m = __ballot(p >= i && (p >= n || cur != sym));
        if (m) return base + (uint32_t)(__ffsll(m) - 1) - i;
        uint32_t j = base + 64u;                               // the run leaves the window: 64 elements per step from memory
        for (;;) {
            const uint32_t q = j + lane;
            const unsigned long long m2 = __ballot(q >= n || (uint32_t)in[q] != sym);
            if (m2) return j + (uint32_t)(__ffsll(m2) - 1) - i;
            j += 64u;
        }
    }
};

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_mtf(rcx_kargs a, int decode)
{
    const unsigned w = threadIdx.x >> 6, lane = rcx_lane();
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));   // wave-uniform: keeps descriptors in SGPRs
    if (b >= a.nblocks) return;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint32_t n = (uint32_t)a.in_len[b];
    uint8_t* out = a.out_base + a.out_off[b];
    if (a.out_cap[b] < n) {
        if (lane == 0) { a.status[b] = RCX_E_OUTPUT_TOO_SMALL; a.out_len[b] = 0; if (a.in_used) a.in_used[b] = 0; }
        return;
    }
    MtfRegs L; L.identity(lane);                              // mtf.rs:103-104, :141-142
    uint32_t i = 0;
    SeqWin<uint8_t> win; win.start(in, n, lane);
    while (i < n) {
        const uint32_t x = win.get(i);
        if (!decode) {
            const uint32_t head = L.at(0);
            if (x == head) {                                  // rank 0: the whole run encodes to zeros
                const uint32_t rl = win.run(i, x);
                for (uint32_t t = lane; t < rl; t += 64) out[i + t] = 0;
                i += rl;
            } else {
                const uint32_t rank = L.find(x);
                if (lane == 0) out[i] = (uint8_t)rank;
                L.front(rank, x);
                i += 1;
            }
        } else {
            if (x == 0) {                                     // rank 0 repeats the front symbol
                const uint32_t rl = win.run(i, 0);
                const uint8_t head = (uint8_t)L.at(0);
                for (uint32_t t = lane; t < rl; t += 64) out[i + t] = head;
                i += rl;
            } else {
                const uint32_t sym = L.at(x);
                if (lane == 0) out[i] = (uint8_t)sym;
                L.front(x, sym);
                i += 1;
            }
        }
    }
    if (lane == 0) { a.status[b] = RCX_OK; a.out_len[b] = n; if (a.in_used) a.in_used[b] = n; }
}

// =================================================================================================
// DC -- src/bwt/dc.rs.  encode :110-149 (+ EncodeIterator order :88-104, encode_simple :153-159),
// decode :162-233 driven as decode_simple :236-252.  Words are little-endian u32.
// =================================================================================================
// Both DC loops are serial chains of one step per run of the input (~75 K steps for a 256 KiB block of BWT output), one wave per
// block, so what counts is the LATENCY of a step.  The first version kept `last[]` / `next[]` in LDS: two dependent LDS round trips
// per step (~1100 cycles per step, 35-38 ms per batch whatever its size).  Here everything a step touches lives in registers, rank
// ordered like the list itself -- lane l holds entries 4l .. 4l+3: their symbols in the bytes of one dword (MtfRegs) and one
// position per entry in four more registers -- and a step is readlane / compare / ballot / DPP shift, no memory on the chain.
struct DcRegs : MtfRegs {
    uint32_t v[4];                                            // a position per entry (encode: last occurrence, decode: next occurrence)
    // entries 1..rank = old entries 0..rank-1, entry 0 = (sym, val)
    __device__ __forceinline__ void front_v(uint32_t rank, uint32_t sym, uint32_t val)
    {
        const uint32_t prev3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[3], 0x138, 0xf, 0xf, false);   // wave_shr:1: lane l-1's v[3]
        const uint32_t p0 = 4u * lane;
        const uint32_t o0 = v[0], o1 = v[1], o2 = v[2];
        v[3] = (p0 + 3u <= rank) ? o2 : v[3];
        v[2] = (p0 + 2u <= rank) ? o1 : v[2];
        v[1] = (p0 + 1u <= rank) ? o0 : v[1];
        v[0] = lane == 0 ? val : (p0 <= rank ? prev3 : v[0]);
        front(rank, sym);
    }
    // entries 0..rank-2 = old entries 1..rank-1, entry rank-1 = (sym, val); rank >= 1
    __device__ __forceinline__ void back_v(uint32_t rank, uint32_t sym, uint32_t val)
    {
        const uint32_t next0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[0], 0x130, 0xf, 0xf, false);   // wave_shl:1: lane l+1's v[0]
        const uint32_t p0 = 4u * lane, q = rank - 1u;         // positions below q take their right neighbour, position q the new entry
        const uint32_t o1 = v[1], o2 = v[2], o3 = v[3];
        v[0] = (p0 < q) ? o1 : (p0 == q ? val : v[0]);
        v[1] = (p0 + 1u < q) ? o2 : (p0 + 1u == q ? val : v[1]);
        v[2] = (p0 + 2u < q) ? o3 : (p0 + 2u == q ? val : v[2]);
        v[3] = (p0 + 3u < q) ? next0 : (p0 + 3u == q ? val : v[3]);
        back(rank, sym);
    }
    // first entry below `count` that holds sym: its rank and its position value; false if there is none
    __device__ __forceinline__ bool find_v(uint32_t sym, uint32_t count, uint32_t& rank, uint32_t& val) const
    {
        const uint32_t p0 = 4u * lane;
        uint32_t hit = 4, hv = 0;
        if (((w >> 24) & 0xff) == sym && p0 + 3u < count) { hit = 3; hv = v[3]; }
        if (((w >> 16) & 0xff) == sym && p0 + 2u < count) { hit = 2; hv = v[2]; }
        if (((w >> 8) & 0xff) == sym && p0 + 1u < count) { hit = 1; hv = v[1]; }
        if ((w & 0xff) == sym && p0 < count) { hit = 0; hv = v[0]; }
        const unsigned long long m = __ballot(hit < 4);
        if (!m) return false;
        const int first = __ffsll(m) - 1;
        rank = 4u * (uint32_t)first + (uint32_t)__builtin_amdgcn_readlane((int)hit, first);
        val = (uint32_t)__builtin_amdgcn_readlane((int)hv, first);
        return true;
    }
};

// The same list with ONE entry per lane, for alphabets of at most 64 symbols (any text): find is one compare + ballot, a move
// is one DPP shift + one select per register -- a quarter of the vector work of the general layout, and with ~4000 blocks in
// flight these serial loops are bound by the CUs' vector issue slots (each step is a wave64 instruction stream in which a few
// lanes do useful work), not by latency.
struct DcRegs1 {
    uint32_t sy, v; unsigned lane;                            // lane l: the symbol at rank l and its position
    __device__ __forceinline__ void zero(unsigned lane_) { lane = lane_; sy = 0; v = 0; }
    __device__ __forceinline__ uint32_t sym_at(uint32_t pos) const { return (uint32_t)__builtin_amdgcn_readlane((int)sy, (int)pos); }
    __device__ __forceinline__ uint32_t val_at(uint32_t pos) const { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)pos); }
    __device__ __forceinline__ void set_val0(uint32_t val) { v = lane == 0 ? val : v; }
    __device__ __forceinline__ void front_v(uint32_t rank, uint32_t sym, uint32_t val)
    {
        const uint32_t ps = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sy, 0x138, 0xf, 0xf, false);    // wave_shr:1
        const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
        const bool mv = lane <= rank;
        sy = lane == 0 ? sym : (mv ? ps : sy);
        v = lane == 0 ? val : (mv ? pv : v);
    }
    __device__ __forceinline__ void back_v(uint32_t rank, uint32_t sym, uint32_t val)      // rank >= 1
    {
        const uint32_t ns = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sy, 0x130, 0xf, 0xf, false);    // wave_shl:1
        const uint32_t nv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
        const uint32_t q = rank - 1u;
        sy = lane < q ? ns : (lane == q ? sym : sy);
        v = lane < q ? nv : (lane == q ? val : v);
    }
    __device__ __forceinline__ bool find_v(uint32_t sym, uint32_t count, uint32_t& rank, uint32_t& val) const
    {
        const unsigned long long m = __ballot(sy == sym && lane < count);
        if (!m) return false;
        rank = (uint32_t)__ffsll(m) - 1u;
        val = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)rank);
        return true;
    }
    // first rank r in [1, A) with !(future + r > v[r]), else A   (dc.rs:214-218)
    __device__ __forceinline__ uint32_t first_fit(uint32_t future, uint32_t A) const
    {
        const unsigned long long m = __ballot(lane >= 1u && lane < A && !((uint64_t)future + lane > (uint64_t)v));
        return m ? (uint32_t)__ffsll(m) - 1u : A;
    }
    template <class F> __device__ __forceinline__ void each(uint32_t count, F f) const { if (lane < count) f(lane, sy, v); }
};
// (the general layout's versions of the same interface)
struct DcRegs4 : DcRegs {
    __device__ __forceinline__ void zero(unsigned lane_) { MtfRegs::zero(lane_); v[0] = v[1] = v[2] = v[3] = 0; }
    __device__ __forceinline__ uint32_t sym_at(uint32_t pos) const { return at(pos); }
    __device__ __forceinline__ uint32_t val_at(uint32_t pos) const
    {
        const uint32_t k = pos & 3u;
        const uint32_t x = k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : v[3];          // pos is uniform: scalar selects
        return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)(pos >> 2));
    }
    __device__ __forceinline__ void set_val0(uint32_t val) { v[0] = lane == 0 ? val : v[0]; }
    __device__ __forceinline__ uint32_t first_fit(uint32_t future, uint32_t A) const
    {
        uint32_t cand = 0xffffffffu;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const uint32_t r = 4u * lane + (uint32_t)k;
            if (r >= 1u && r < A && !((uint64_t)future + r > (uint64_t)v[k])) cand = r;
        }
        const unsigned long long m = __ballot(cand != 0xffffffffu);
        return m ? (uint32_t)__builtin_amdgcn_readlane((int)cand, __ffsll(m) - 1) : A;
    }
    template <class F> __device__ __forceinline__ void each(uint32_t count, F f) const
    {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) { const uint32_t r = 4u * lane + k; if (r < count) f(r, (w >> (8u * k)) & 0xffu, v[k]); }
    }
    // take over a one-entry-per-lane list (its entries 0 .. 63)
    __device__ __forceinline__ void from1(const DcRegs1& o)
    {
        lane = o.lane; w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int srcl = (int)(((4u * lane + (uint32_t)k) & 63u) << 2);
            const uint32_t sk = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl, (int)o.sy), vk = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl, (int)o.v);
            const bool have = lane < 16u;
            w |= have ? (sk & 0xffu) << (8 * k) : 0u;
            v[k] = have ? vk : 0u;
        }
    }
};

// dc.rs:117-138 from position i on, until the input ends or a (limit+1)-th distinct symbol turns up; returns where it stopped
template <class LT>
__device__ __forceinline__ uint32_t dc_encode_steps(LT& L, SeqWin<uint8_t>& win, uint32_t i, uint32_t n, uint32_t& num_unique, uint32_t limit,
                                                    uint32_t* words, uint32_t* dist, unsigned lane)
{
    while (i < n) {
        const uint32_t sym = win.get(i);
        uint32_t rank = 0, base = 0;
        if (!L.find_v(sym, num_unique, rank, base)) {         // first occurrence, :121-128: insert behind the others, move to front
            if (num_unique == limit) break;
            if (lane == 0) { words[sym] = i; dist[i] = n; }
            L.front_v(num_unique, sym, i);
            num_unique++;
            i += 1;
        } else if (base == i - 1) {                           // inside a run: rank 0, nothing is emitted
            const uint32_t rl = win.run(i, sym);
            for (uint32_t t = lane; t < rl; t += 64) dist[i + t] = n;
            L.set_val0(i + rl - 1);
            i += rl;
        } else {                                              // :129-136 (rank >= 1: the front symbol is the one at i - 1)
            if (lane == 0) { dist[i] = n; dist[base] = i - base - rank - 1; }
            L.front_v(rank, sym, i);
            i += 1;
        }
    }
    return i;
}

// withctx: also writes the coding CONTEXT of every distance (dc.rs:40-58; yielded with it by EncodeIterator, :88-103) -- eight
// bytes each, {symbol | last_rank << 8, distance_limit}, from byte 4 * (256 + n) of the block's slot on; out_len is then
// 4 * (256 + n) + 8 * k (k distances; the words are the first 4 * (256 + k) bytes as always).
// (the lane-per-chunk encoder's scratch slot, k_dcx_* below: its first word says whether it has taken the block)
#define DCX_CHUNKS 64u
#define DCX_MIN 8192u
#define DCX_NONE 0xffffffffu
#define DCX_O_MAP 64u                                   /* head: handled, alpha, ch, nchunks, ... */
#define DCX_O_LAST 512u                                 /* u32 [64 symbols][64 chunks], laid out as DCX_AT says */
#define DCX_AT(sym, chunk) (((((sym) >> 2) * DCX_CHUNKS + (chunk)) << 2) + ((sym) & 3u))
#define DCX_O_LRUN (DCX_O_LAST + 16384u)
#define DCX_O_RC (DCX_O_LRUN + 16384u)                  /* u32 [64 chunks] */
#define DCX_O_RANK (DCX_O_RC + 256u)                    /* u8 [64 symbols][64 chunks]: the symbol's place in the move-to-front list (0xff: not seen), laid out as DCX_RK says */
#define DCX_RK(sym, chunk) (((((sym) >> 4) * DCX_CHUNKS + (chunk)) << 4) + ((sym) & 15u))
#define DCX_SLOT (DCX_O_RANK + 4096u)                   /* 37 632 bytes a block */
static uint64_t dc_encode_scratch_bytes(uint32_t nblocks) { return (uint64_t)nblocks * DCX_SLOT + 256; }

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_dc_encode(rcx_kargs a, int withctx, int skip_handled)
{
    __shared__ uint32_t s_pos[WAVES][256];                  // (withctx) where each symbol's next occurrence was predicted, :94
    const unsigned w = threadIdx.x >> 6, lane = rcx_lane();
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));   // wave-uniform: keeps descriptors in SGPRs
    if (b >= a.nblocks) return;
    if (skip_handled && *(const uint32_t*)((const uint8_t*)a.scratch + (size_t)b * DCX_SLOT) == 1u) return;       // k_dcx_main has encoded it
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint32_t n = (uint32_t)a.in_len[b];
    uint32_t* words = (uint32_t*)(a.out_base + a.out_off[b]);
    // the out slot doubles as the dist[] work array (dc.rs:112 `distances`), so it must hold 256+n words
    if (a.out_cap[b] < 4ull * (256ull + n) + (withctx ? 8ull * n : 0ull) || ((uintptr_t)words & 3u)) {
        if (lane == 0) { a.status[b] = RCX_E_OUTPUT_TOO_SMALL; a.out_len[b] = 0; if (a.in_used) a.in_used[b] = 0; }
        return;
    }
    uint32_t* dist = words + 256;
    for (int k = 0; k < 4; k++) words[lane + 64 * k] = n;     // :114-115 (init[]: first occurrences, n = absent)
    rcx_wave_sync();
    uint32_t num_unique = 0;
    SeqWin<uint8_t> win; win.start(in, n, lane);
    // MTF::new(); the position of an entry = the last occurrence of its symbol.  One entry per lane while <= 64 symbols have turned up.
    DcRegs1 L1; L1.zero(lane);
    uint32_t i = dc_encode_steps(L1, win, 0, n, num_unique, 64u, words, dist, lane);
    auto sweep = [&](uint32_t rank, uint32_t, uint32_t base) { dist[base] = n - base - rank - 1; };      // :139-144
    if (i < n) {
        DcRegs4 L4; L4.from1(L1);
        i = dc_encode_steps(L4, win, i, n, num_unique, 256u, words, dist, lane);
        rcx_wave_sync();                                       // every filler store precedes the sweep's stores (same wave: in order)
        L4.each(num_unique, sweep);
    } else {
        rcx_wave_sync();
        L1.each(num_unique, sweep);
    }
    __threadfence_block();
    rcx_wave_sync();
    // compact the non-filler distances in position order (EncodeIterator :88-104): ballot + prefix popcount
    uint32_t k = 0;
    uint32_t last_active = 0;                                // (withctx) :97, one past the position of the last distance yielded
    uint32_t* const posn = s_pos[w];
    uint32_t* const ctxo = dist + n;                         // byte 4 * (256 + n) of the slot
    if (withctx) { for (int q = 0; q < 4; q++) posn[lane + 64 * q] = words[lane + 64 * q]; rcx_wave_sync(); }
    for (uint32_t j = 0; j < n; j += 64) {
        const uint32_t p = j + lane;
        const uint32_t d = p < n ? dist[p] : n;
        const bool keep = d != n;
        const unsigned long long m = __ballot(keep);
        if (withctx && m) {                                  // Context::new(sym, last_active - pos[sym], size - i), :94-102
            const uint32_t sym = keep ? in[p] : 0u;
            unsigned long long same = m;                     // the kept lanes that hold my symbol
#pragma unroll
            for (int bit = 0; bit < 8; bit++) {
                const unsigned long long bbm = __ballot(keep && ((sym >> bit) & 1u));
                same &= ((sym >> bit) & 1u) ? bbm : ~bbm;
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            const unsigned long long pk = m & below, ps = same & below;
            const uint32_t la = pk ? j + (63u - (uint32_t)__clzll(pk)) + 1u : last_active;
            const uint32_t ql = ps ? 63u - (uint32_t)__clzll(ps) : lane;
            const uint32_t dq = (uint32_t)__shfl((int)d, (int)ql);
            const uint32_t tabv = posn[sym];
            const uint32_t posv = ps ? j + ql + 1u + dq : tabv;
            if (keep) {
                const uint32_t at = k + (uint32_t)__popcll(pk);
                ctxo[2 * at] = sym | (((la - posv) & 0xffu) << 8);
                ctxo[2 * at + 1] = n - p;
            }
            rcx_wave_sync();
            if (keep && (same >> lane) == 1ull) posn[sym] = p + 1u + d;          // the last of its symbol in this window
            last_active = RCX_UNI(j + (63u - (uint32_t)__clzll(m)) + 1u);
            rcx_wave_sync();
        }
        rcx_wave_sync();                                     // all reads of this 64-slot window precede the writes
        if (keep) dist[k + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = d;
        k += (uint32_t)__popcll(m);
    }
    if (lane == 0) { a.status[b] = RCX_OK; a.out_len[b] = withctx ? 4ull * (256ull + n) + 8ull * k : 4ull * (256ull + k); if (a.in_used) a.in_used[b] = n; }
}

// -------------------------------------------------------------------------------------------------
// DC encode, ONE LANE PER CHUNK (alphabets of at most 64 symbols: any text; blocks of 8 KiB and more; no contexts).
// k_dc_encode above is one wave per block with the list in its registers: a wave64 instruction stream in which a few lanes work,
// ~16 instructions per byte, and the kernel is bound by the CUs' issue slots (3815 blocks of 256 KiB: 26 ms).  The state the loop
// carries is small and can be had for ANY position without running the loop: the list order is the order of the symbols' last
// occurrences, a symbol's pending distance belongs to the end of its last run.  So a block is cut into 64 chunks:
//   k_dcx_prep  (a workgroup per block) finds the alphabet, per chunk every symbol's last position and the number of runs, and
//               from these the state at every chunk's start: list order, last positions, the run each of those ended in;
//   k_dcx_main  (a wave per block) runs the reference's loop (dc.rs:117-138) for 64 chunks at once, a lane each, its lists and
//               tables in LDS (entry-major: lane-contiguous, no bank conflicts).  A distance is stored straight at its place: the
//               k-th word belongs to the k-th run's end (EncodeIterator :88-104 yields them in position order), and the lane that
//               meets a symbol again knows the run its previous occurrence ended -- no dist[] array, no compaction pass.
// ~1.4 wave instructions per byte instead of 16.  Blocks this path does not take (flag in the scratch slot) are left to k_dc_encode.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dcx_prep(rcx_kargs a)
{
    __shared__ uint32_t s_present[8];
    __shared__ uint8_t s_flag[256];
    __shared__ uint8_t s_map[256];
    __shared__ uint32_t s_lp[DCX_CHUNKS][64];            // per chunk and symbol: its last position there,
    __shared__ uint32_t s_lr[DCX_CHUNKS][64];            // the run (counted from the chunk's first run start; -1: the run the chunk begins in) it lies in
    __shared__ uint32_t s_rs[DCX_CHUNKS];                // run starts in the chunk
    __shared__ uint32_t s_alpha;
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    const unsigned tid = threadIdx.x, w = RCX_UNI(tid >> 6), lane = tid & 63u;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint32_t n = (uint32_t)a.in_len[b];
    uint32_t* words = (uint32_t*)(a.out_base + a.out_off[b]);
    uint8_t* slot = (uint8_t*)a.scratch + (size_t)b * DCX_SLOT;
    uint32_t* head = (uint32_t*)slot;
    // (what k_dc_encode refuses it refuses itself: the block is left to it)
    if (n < DCX_MIN || a.in_len[b] >= 0x7fffffffull || a.out_cap[b] < 4ull * (256ull + n) || ((uintptr_t)words & 3u)) { if (tid == 0) head[0] = 0; return; }
    s_flag[tid] = 0;
    __syncthreads();
    // the alphabet: a flag byte per value (plain stores of 1: whoever wins writes the same)
    for (uint32_t p = tid * 16u; p < n; p += 256u * 16u) {
        if (p + 16u <= n) {
            const rcx_u32x4 v = *(const rcx_u32x4_u*)(in + p);
#pragma unroll
            for (int k = 0; k < 4; k++) { s_flag[v[k] & 0xffu] = 1; s_flag[(v[k] >> 8) & 0xffu] = 1; s_flag[(v[k] >> 16) & 0xffu] = 1; s_flag[v[k] >> 24] = 1; }
        }
        else for (uint32_t t = p; t < n; t++) s_flag[in[t]] = 1;
    }
    __syncthreads();
    {
        const unsigned long long pm = __ballot(s_flag[tid] != 0);
        if (lane == 0) s_present[w] = (uint32_t)__popcll(pm);
        __syncthreads();
        uint32_t below = (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
        for (uint32_t k = 0; k < w; k++) below += s_present[k];
        s_map[tid] = (uint8_t)(below & 63u);
        if (tid == 0) s_alpha = s_present[0] + s_present[1] + s_present[2] + s_present[3];
    }
    __syncthreads();
    const uint32_t alpha = s_alpha;
    if (alpha > 64u) { if (tid == 0) head[0] = 0; return; }
    const uint32_t ch = (((n + DCX_CHUNKS - 1u) / DCX_CHUNKS) + 63u) & ~63u;      // a chunk: a multiple of 64 positions
    const uint32_t nchunks = (n + ch - 1u) / ch;
    for (uint32_t c = w; c < nchunks; c += 4) {
        const uint32_t cs = c * ch, ce = cs + ch < n ? cs + ch : n;
        s_lp[c][lane] = 0; s_lr[c][lane] = 0;                    // (position + 1 and run + 1 of the symbol's last occurrence in the chunk; 0: none)
        rcx_wave_sync();
        uint32_t runs = 0;
        uint32_t carry = cs ? in[cs - 1] : 0x100u;               // the byte in front of the window
        constexpr int PF = 8;                                    // (a window is 64 bytes per wave: eight windows' loads are issued together)
        for (uint32_t pb = cs; pb < ce; pb += 64 * PF) {
            uint32_t cc[PF];
#pragma unroll
            for (int k = 0; k < PF; k++) { const uint32_t p = pb + 64u * (uint32_t)k + lane; cc[k] = p < ce ? in[p] : 0u; }
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const uint32_t p0 = pb + 64u * (uint32_t)k;
                if (p0 >= ce) break;
                const uint32_t p = p0 + lane;
                const bool valid = p < ce;
                const uint32_t c8 = cc[k];
                uint32_t pv = (uint32_t)__shfl_up((int)c8, 1);
                if (lane == 0) pv = carry;
                const bool rs = valid && c8 != pv;
                const unsigned long long rsm = __ballot(rs);
                const uint32_t id = s_map[c8];
                // a symbol's last occurrence in the chunk ends a run: the lanes in front of a run start (and the chunk's last byte) raise the
                // symbol's entry -- positions and run numbers only grow, so two LDS max operations replace a match-any ranking of the window
                const unsigned long long endm = (rsm >> 1) | (1ull << 63);          // lane l ends a run if lane l + 1 starts one (lane 63: decided by the next window; it may raise early, harmlessly)
                if (valid && (((endm >> lane) & 1ull) || p + 1u == ce)) {
                    atomicMax(&s_lp[c][id], p + 1u);
                    atomicMax(&s_lr[c][id], runs + (uint32_t)__popcll(rsm & ((2ull << lane) - 1ull)));    // (run number + 1: the run the chunk begins in is 0)
                }
                runs += (uint32_t)__popcll(rsm);
                carry = (uint32_t)__shfl((int)c8, 63);
                rcx_wave_sync();
            }
        }
        if (lane == 0) s_rs[c] = runs;
    }
    __syncthreads();
    for (uint32_t i = tid; i < 256; i += 256) words[i] = n;       // init[]: n = absent (dc.rs:114-115); the main kernel fills in the first occurrences
    if (w != 0) return;
    // the state at every chunk's start, chunk after chunk: lane = symbol.  last1: position + 1 of the symbol's last occurrence so far (0: none)
    uint32_t cur_last1 = 0, cur_lr = 0, basec = 0;
    uint32_t* o_last = (uint32_t*)(slot + DCX_O_LAST);
    uint32_t* o_lrun = (uint32_t*)(slot + DCX_O_LRUN);
    uint32_t* o_rc = (uint32_t*)(slot + DCX_O_RC);
    for (uint32_t c = 0; c < nchunks; c++) {
        o_last[DCX_AT(lane, c)] = cur_last1;
        o_lrun[DCX_AT(lane, c)] = cur_lr;
        {   // list order: the later the last occurrence, the nearer the front
            uint32_t rank = 0;
            for (int d = 0; d < 64; d++) rank += (uint32_t)__shfl((int)cur_last1, d) > cur_last1 ? 1u : 0u;
            slot[DCX_O_RANK + DCX_RK(lane, c)] = cur_last1 ? (uint8_t)(0x80u | rank) : (uint8_t)0xff;      // (bit 7 is always set: see k_dcx_main)
        }
        if (lane == 0) o_rc[c] = basec - 1u;                      // the run position cs - 1 lies in (-1 in front of the block)
        const uint32_t lp1 = s_lp[c][lane];
        if (lp1) { cur_last1 = lp1; cur_lr = basec + s_lr[c][lane] - 1u; }
        basec += s_rs[c];
    }
    for (int k = 0; k < 4; k++) slot[DCX_O_MAP + lane + 64 * k] = s_map[lane + 64 * k];
    if (lane == 0) { head[1] = alpha; head[2] = ch; head[3] = nchunks; head[0] = 1; }
}

__global__ __launch_bounds__(64) void k_dcx_main(rcx_kargs a)
{
    __shared__ __align__(16) uint32_t s_last[64 * DCX_CHUNKS];      // DCX_AT: [symbol / 4][chunk][symbol % 4]
    __shared__ uint32_t s_lrun[64 * DCX_CHUNKS];
    __shared__ __align__(16) uint32_t s_rank[16 * DCX_CHUNKS];      // DCX_RK (bytes): sixteen symbols of a lane in one 16-byte access
    __shared__ uint8_t s_map[256];
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    const uint8_t* slot = (const uint8_t*)a.scratch + (size_t)b * DCX_SLOT;
    const uint32_t* head = (const uint32_t*)slot;
    if (head[0] != 1u) return;
    const unsigned lane = rcx_lane();
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint32_t n = (uint32_t)a.in_len[b];
    uint32_t* words = (uint32_t*)(a.out_base + a.out_off[b]);
    const uint32_t alpha = head[1], ch = head[2], nchunks = head[3];
    for (uint32_t i = lane; i < 64 * DCX_CHUNKS; i += 64) { s_last[i] = ((const uint32_t*)(slot + DCX_O_LAST))[i]; s_lrun[i] = ((const uint32_t*)(slot + DCX_O_LRUN))[i]; }
    for (uint32_t i = lane; i < 16 * DCX_CHUNKS; i += 64) s_rank[i] = ((const uint32_t*)(slot + DCX_O_RANK))[i];
    for (int k = 0; k < 4; k++) s_map[lane + 64 * k] = slot[DCX_O_MAP + lane + 64 * k];
    rcx_wave_sync();
    uint8_t* const rankb = (uint8_t*)s_rank;
    const bool mine = lane < nchunks;
    const uint32_t cs = lane * ch, ce = mine ? (cs + ch < n ? cs + ch : n) : 0u;
    uint32_t rc = mine ? ((const uint32_t*)(slot + DCX_O_RC))[lane] : 0u;
    uint32_t front = (mine && cs) ? (uint32_t)s_map[in[cs - 1u]] : 0xffu;      // the symbol whose run is open
    const uint32_t quads = (alpha + 15u) >> 4;
    for (uint32_t t0 = 0; t0 < ch; t0 += 16) {
        uint32_t q[4] = {0, 0, 0, 0};                            // sixteen bytes of every chunk
        {
            const uint32_t p = cs + t0;
            if (mine && p < ce) {
                if (p + 16u <= n) { const rcx_u32x4 v = *(const rcx_u32x4_u*)(in + p); q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3]; }
                else for (uint32_t t = 0; p + t < n; t++) q[t >> 2] |= (uint32_t)in[p + t] << (8u * (t & 3u));
            }
        }
        // the sixteen symbols first (sixteen independent table reads), then a step per byte.  A step is straight-line code -- a lane that
        // stays in its run reads and rewrites its own entries unchanged -- so that its LDS reads (the symbol's entry, its place, the
        // places of all symbols) are ONE round trip: with branches around them they were four, and a step took ~1500 cycles
        uint32_t ids[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t t = 0; t < 16; t++) ids[t >> 2] |= (uint32_t)s_map[(q[t >> 2] >> (8u * (t & 3u))) & 0xffu] << (8u * (t & 3u));
#pragma unroll
        for (uint32_t t = 0; t < 16; t++) {
            const uint32_t i = cs + t0 + t;
            const bool act = mine && i < ce;
            const uint32_t c8 = (q[t >> 2] >> (8u * (t & 3u))) & 0xffu;
            const uint32_t id = (ids[t >> 2] >> (8u * (t & 3u))) & 0xffu;
            const bool sw = act && id != front;                   // a run ends at i - 1, the run of `id` starts at i
            if (!__ballot(sw)) continue;
            const uint32_t ide = sw ? id : 0u, fre = (sw && front != 0xffu) ? front : ide;
            // (the open run's entry is closed below; `id` is another symbol, so its entry can be read first)
            const uint32_t base1r = s_last[DCX_AT(ide, lane)], lrr = s_lrun[DCX_AT(ide, lane)], rb = rankb[DCX_RK(ide, lane)];
            rcx_u32x4 v[4];
#pragma unroll
            for (int g = 0; g < 4; g++) v[g] = ((uint32_t)g < quads) ? *(const rcx_u32x4*)&s_rank[((uint32_t)g * DCX_CHUNKS + lane) * 4u] : rcx_u32x4{0, 0, 0, 0};
            const uint32_t base1 = sw ? base1r : 0u, lr = lrr;
            const uint32_t r = !sw ? 0u : (base1 ? (rb & 0x7fu) : 64u);     // its place in the list (mtf.rs:63-79); a new symbol pushes the whole list (dc.rs:123-124)
            // move to front without a list: every symbol in front of `id` moves one place down -- sixteen places per 16-byte access, four
            // per instruction (a byte is 0x80 | place, place < 64, or 0xff: x - r keeps its top bit exactly where place >= r, and no byte borrows)
            const uint32_t rr = r * 0x01010101u;                  // (r = 0 for a lane that stays in its run: nothing moves)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if ((uint32_t)g < quads) {
#pragma unroll
                    for (int k = 0; k < 4; k++) v[g][k] += (~(v[g][k] - rr) >> 7) & 0x01010101u;
                    *(rcx_u32x4*)&s_rank[((uint32_t)g * DCX_CHUNKS + lane) * 4u] = v[g];
                }
            }
            if (sw) {
                rankb[DCX_RK(id, lane)] = 0x80;
                if (front != 0xffu) { s_last[DCX_AT(fre, lane)] = i; s_lrun[DCX_AT(fre, lane)] = rc; }     // the open run ends at i - 1 (+ 1)
                rc++;
                front = id;
                if (base1) words[256u + lr] = i - (base1 - 1u) - r - 1u;       // dc.rs:134, stored where EncodeIterator will yield it
                else words[c8] = i;                                           // first occurrence: init[], dc.rs:126
            }
            rcx_wave_sync();
        }
    }
    // dc.rs:139-144: the distances still open at the end, one per symbol; lane = symbol, over the last chunk's table
    const uint32_t lc = nchunks - 1u;
    const uint32_t rcl = (uint32_t)__shfl((int)rc, (int)lc);
    rcx_wave_sync();
    if (lane == lc && front != 0xffu) { s_last[DCX_AT(front, lane)] = n; s_lrun[DCX_AT(front, lane)] = rc; }
    rcx_wave_sync();
    {
        const uint32_t my1 = s_last[DCX_AT(lane, lc)], mylr = s_lrun[DCX_AT(lane, lc)], rank = rankb[DCX_RK(lane, lc)] & 0x7fu;
        if (lane < alpha && my1) words[256u + mylr] = n - (my1 - 1u) - rank - 1u;
    }
    if (lane == 0) { a.status[b] = RCX_OK; a.out_len[b] = 4ull * (256ull + rcl + 1u); if (a.in_used) a.in_used[b] = n; }
}

// The decoder's step (dc.rs:199-229) for the one-entry-per-lane list, as ISA (gfx950).  A step is a chain -- the next occurrence of
// the second symbol (v_readlane) -> the run's end -> + the distance -> which rank fits (v_cmp, ballot, s_ff1) -> the list moves up
// (DPP shift + selects) -> the next step's v_readlane -- and a block is ~50 000 of them, one after the other: the kernel's time IS
// that chain (13.4 ms for 256 KiB blocks however many there are; both issue ports half idle, profiles/r04_ari_dc_sq_counters.txt).
// hipcc's loop spends ~70 instructions and 14 branches on a step, with 64-bit compares and the window's refill logic in line:
// ~590 cycles.  Here a step is 38 instructions and one taken branch; everything that is not the plain case -- the block's end, a
// run of 64 bytes or more, a distance outside the 64 words held in `cur`, any error -- LEAVES with the state as it stood at the
// top of that step, and the portable step below (which knows all the cases) takes it from there.
//   sy, v: the list (lane r: the symbol at rank r and its next position); i: output position; di: index of the next distance
//   cur: the distance words wbase .. wbase + 63 (a word a lane); maskA: lanes 1 .. A - 1
// Hazards (gfx940+): two wait states between a VALU write of an SGPR pair and a VALU read of it, two between a VALU write of a
// VGPR and a DPP read, one before a v_readlane of it: the instruction order below provides them.
#ifndef RCX_NO_DC_STEPS_ASM
// (second form: the CU's ONE scalar port -- SALU, v_readlane, v_cmp -- is what sixteen such chains share; the first form's 28 + 6
//  scalar-port instructions a step measured 10.6 ms.  `lim` = min(nwords, wbase + 64) folds two exits into one, the distance's
//  bound implies the stop's, a sentinel bit at rank A replaces the empty-ballot select, and the new entry is dropped into the
//  shifted registers with v_writelane so that ONE compare moves the list.)
//  third form: the place in the word window is the loop variable, and the ballot shifted right by one gives rank - 1 at once --
//  the lanes from A on hold 0xffffffff as their position, so every rank "fits" there and rank A needs no sentinel: 21 + 5.)
__device__ __forceinline__ void rcx_dc_fast_steps(uint32_t& sy, uint32_t& v, uint32_t& i, uint32_t& w, uint32_t n, uint32_t wlim, uint32_t cur,
                                                   uint64_t vmask, const uint8_t* out, uint32_t lane)
{
    // w: the next distance's place among the 64 words in `cur` (the caller keeps di = wbase + w); wlim: the first place that is not
    // there (or behind the stream's last word); vmask: lanes 0 .. A - 1 (above them v is 0xffffffff: every rank "fits" there, so
    // the ballot's lowest bit above rank 0 is the rank that fits or A itself)
    uint32_t stop, sym, d, t, fut, q, val, tmp, vsym, ns, nv;
    uint64_t m, cge;
    asm volatile(
        "L_top_%=:\n\t"
        "s_cmp_ge_u32 %[i], %[n]\n\t"
        "s_cbranch_scc1 L_out_%=\n\t"
        "s_cmp_ge_u32 %[w], %[wlim]\n\t"                         // the next distance is not among the 64 words held (or there is none)
        "s_cbranch_scc1 L_out_%=\n\t"
        "v_readlane_b32 %[stop], %[v], 1\n\t"
        "v_readlane_b32 %[sym], %[sy], 0\n\t"
        "v_readlane_b32 %[d], %[cur], %[w]\n\t"
        "s_sub_u32 %[t], %[stop], %[i]\n\t"                      // the run: 0 .. 63 bytes here (a stop below i wraps: leaves too)
        "s_cmp_gt_u32 %[t], 63\n\t"
        "s_cbranch_scc1 L_out_%=\n\t"
        "s_add_u32 %[fut], %[stop], %[d]\n\t"
        "s_cbranch_scc1 L_out_%=\n\t"
        "s_cmp_gt_u32 %[fut], %[n]\n\t"                          // (future <= n implies stop <= n)
        "s_cbranch_scc1 L_out_%=\n\t"
        // ---- the plain case: nothing below can fail
        "v_add_u32_e32 %[tmp], %[fut], %[lane]\n\t"
        "v_cmp_le_u32_e32 vcc, %[tmp], %[v]\n\t"                 // !(future + rank > next[rank]); lanes >= A: always
        "v_mov_b32_dpp %[ns], %[sy] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[nv], %[v] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_lshr_b64 vcc, vcc, 1\n\t"                             // rank 0 does not count, and the lowest bit left is rank - 1
        "s_ff1_i32_b64 %[q], vcc\n\t"
        "s_add_u32 %[val], %[fut], %[q]\n\t"
        "s_mov_b32 m0, %[q]\n\t"                                 // (a lane select in m0 is not a second scalar operand)
        "v_writelane_b32 %[ns], %[sym], m0\n\t"                  // entry q of the shifted list = (sym, future + rank - 1)
        "v_writelane_b32 %[nv], %[val], m0\n\t"
        "v_cmp_ge_u32_e64 %[cge], %[q], %[lane]\n\t"             // ranks 0 .. q change
        "v_mov_b32_e32 %[vsym], %[sym]\n\t"
        "s_bfm_b64 %[m], %[t], 0\n\t"
        "v_add_u32_e32 %[tmp], %[i], %[lane]\n\t"
        "v_cndmask_b32_e64 %[sy], %[sy], %[ns], %[cge]\n\t"
        "v_cndmask_b32_e64 %[v], %[v], %[nv], %[cge]\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "global_store_byte %[tmp], %[vsym], %[out]\n\t"          // the run's bytes
        "s_mov_b64 exec, -1\n\t"
        "s_mov_b32 %[i], %[stop]\n\t"
        "s_add_u32 %[w], %[w], 1\n\t"
        "s_branch L_top_%=\n\t"
        "L_out_%=:\n\t"
        : [sy] "+v"(sy), [v] "+v"(v), [i] "+s"(i), [w] "+s"(w), [stop] "=&s"(stop), [sym] "=&s"(sym), [d] "=&s"(d), [t] "=&s"(t),
          [fut] "=&s"(fut), [q] "=&s"(q), [val] "=&s"(val), [m] "=&s"(m), [cge] "=&s"(cge),
          [tmp] "=&v"(tmp), [vsym] "=&v"(vsym), [ns] "=&v"(ns), [nv] "=&v"(nv)
        : [n] "s"(n), [wlim] "s"(wlim), [cur] "v"(cur), [vmask] "s"(vmask), [out] "s"(out), [lane] "v"(lane)
        : "vcc", "scc", "m0", "memory");
}
#endif

// dc.rs:199-229 driven as decode_simple :236-252; returns the status
template <bool CTX, class LT>
__device__ __forceinline__ int dc_decode_steps(LT& L, SeqWin<uint32_t>& wwin, uint32_t& i, uint32_t n, uint32_t A, uint32_t& di, uint32_t nwords, uint8_t* out, unsigned lane,
                                               uint32_t* ctxo = nullptr, uint8_t* ranks = nullptr)
{
    while (i < n) {
#ifndef RCX_NO_DC_STEPS_ASM
        if constexpr (!CTX && std::is_same<LT, DcRegs1>::value) {
            // as many plain steps as there are, hand-written; what it leaves at is taken by the portable step below (the distance
            // window is moved there: wwin.get), and the loop comes back here
            if (di >= 256u && di < nwords && n < 0xffffff00u && A < 64u) {        // (future + rank is computed in 32 bits there; rank A is a bit of the ballot)
                wwin.seek(di);
                uint32_t ui = RCX_UNI(i), udi = RCX_UNI(di);
                const uint32_t wb = RCX_UNI(wwin.base), wlim = RCX_UNI(nwords - wb < 64u ? nwords - wb : 64u);
                uint32_t uw = RCX_UNI(udi - wb);
                rcx_dc_fast_steps(L.sy, L.v, ui, uw, n, wlim, wwin.cur, (1ull << A) - 1ull, out, lane);
                i = ui; di = wb + uw;
                if (i >= n) break;
            }
        }
#endif
        const uint32_t sym = L.sym_at(0);
        const uint32_t stop = L.val_at(1);
        if (stop > n) return RCX_E_MALFORMED;                  // output[i] index panic
        {   // the run: one predicated store (runs are ~5 bytes); a loop only where it is longer than the wave
            const uint32_t t0 = i + lane;
            if (t0 < stop) out[t0] = (uint8_t)sym;
            if (stop > i + 64u) {                                // (uniform)
#pragma unroll 1
                for (uint32_t t = t0 + 64u; t < stop; t += 64) out[t] = (uint8_t)sym;
            }
        }
        if (stop > i) i = stop;
        if (CTX && lane == 0 && di < nwords) {                 // Context::new(sym, ranks[sym], n + 1 - i), :208: what the distance callback is handed
            // (di == nwords: the callback finds no distance and the step ends in RCX_E_EOF below -- slot (nwords - 256) lies past
            //  the coff + 8 * (nwords - 256) bytes the caller provides)
            ctxo[2 * (di - 256u)] = sym | ((uint32_t)ranks[sym] << 8);
            ctxo[2 * (di - 256u) + 1] = n + 1u - i;
        }
        di++;                                                  // decode_simple closure :243-249
        if (di > nwords) return RCX_E_EOF;
        const uint32_t d = wwin.get(di - 1);
        const uint64_t future64 = (uint64_t)stop + d;
        if (future64 > n) return RCX_E_MALFORMED;              /* :213 assert */
        const uint32_t future = (uint32_t)future64;
        const uint32_t rank = L.first_fit(future, A);          // :214-218
        L.back_v(rank, sym, future + rank - 1);                // lst[0..rank-2] = lst[1..rank-1]; lst[rank-1] = sym, :225-227
        if (CTX && lane == 0) ranks[sym] = (uint8_t)(rank - 1u);                   // :228
    }
    return RCX_OK;
}

// withctx: also writes the Context handed to the distance callback at every step (dc.rs:199-229), eight bytes each as in
// k_dc_encode, from byte (n + 7) & ~7 of the block's slot on; out_len is then that offset + 8 * (distances consumed).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_dc_decode(rcx_kargs a, int withctx)
{
    __shared__ __align__(16) uint8_t s_lst[WAVES][256];    // only to order the alphabet once; the loop keeps the list in registers
    __shared__ uint32_t s_next[WAVES][256];
    __shared__ uint8_t s_ranks[WAVES][256];                // (withctx) :190-196, 228
    const unsigned w = threadIdx.x >> 6, lane = rcx_lane();
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));   // wave-uniform: keeps descriptors in SGPRs
    if (b >= a.nblocks) return;
    uint8_t* lst = s_lst[w];
    uint32_t* next = s_next[w];
    const uint32_t* words = (const uint32_t*)(a.in_base + a.in_off[b]);
    const uint32_t nwords = (uint32_t)(a.in_len[b] / 4);
    const uint32_t n = (uint32_t)a.n_out[b];
    uint8_t* out = a.out_base + a.out_off[b];
    int st = RCX_OK;
    uint32_t di = 256, i = 0, A = 0;
    const uint64_t coff = ((uint64_t)n + 7ull) & ~7ull;
    uint32_t* const ctxo = withctx ? (uint32_t*)(out + coff) : nullptr;
    uint8_t* const ranks = s_ranks[w];
    if (nwords < 256 || ((uintptr_t)words & 3u)) st = RCX_E_MALFORMED;          // :239-241
    else if (a.out_cap[b] < (withctx ? coff + 8ull * (nwords - 256u) : (uint64_t)n) || (withctx && ((uintptr_t)out & 7u))) st = RCX_E_OUTPUT_TOO_SMALL;
    if (!st) {
        if (withctx) { for (int k = 0; k < 4; k++) ranks[lane + 64 * k] = 0; }
        for (int k = 0; k < 4; k++) { next[lane + 64 * k] = words[lane + 64 * k]; lst[lane + 64 * k] = 0; }
        rcx_wave_sync();
        // :168-179 order the present symbols by first position: rank(sym) = #present symbols with a smaller
        // init (stable on ties by symbol value, as the insertion sort is)
        for (int k = 0; k < 4; k++) {
            const uint32_t sym = lane + 64 * k;
            const uint32_t d = next[sym];
            if (d < n) {
                uint32_t r = 0;
                for (uint32_t s2 = 0; s2 < 256; s2++) {
                    const uint32_t d2 = next[s2];
                    if (d2 < n && (d2 < d || (d2 == d && s2 < sym))) r++;
                }
                lst[r] = (uint8_t)sym;
            }
        }
        uint32_t cnt = 0;
        bool absent_bad = false;                               // :230 for the symbols the loop never touches
        for (int k = 0; k < 4; k++) cnt += (next[lane + 64 * k] < n) ? 1u : 0u;
        A = RCX_UNI(rcx_wave_sum(cnt));                       // (uniform, and SAID so: `i = n` below depends on it, and a loop counter the compiler takes
                                                               // for divergent turned the whole decode loop into exec-masked code with its counters in VGPRs)
        for (int k = 0; k < 4; k++) { const uint32_t x = next[lane + 64 * k]; absent_bad = absent_bad || (x >= n && x >= n + A); }
        rcx_wave_sync();
        if (A <= 1) {                                          // :180-187 redundant alphabet: no distance is read
            const uint8_t sym = lst[0];
            for (uint32_t t = lane; t < n; t += 64) out[t] = sym;
            i = n;
        }
        SeqWin<uint32_t> wwin; wwin.start(words, nwords, lane);
        bool bad = absent_bad;
        auto check = [&](uint32_t, uint32_t, uint32_t x) { bad = bad || x < n || x >= n + A; };           // :230 assert, listed symbols
        if (A <= 64) {                                         // one entry per lane (v = the symbol's next occurrence)
            DcRegs1 L; L.lane = lane; L.sy = lst[lane]; L.v = lane < A ? next[lst[lane]] : 0xffffffffu;
            st = withctx ? dc_decode_steps<true>(L, wwin, i, n, A, di, nwords, out, lane, ctxo, ranks) : dc_decode_steps<false>(L, wwin, i, n, A, di, nwords, out, lane);
            L.each(A, check);
        } else {
            DcRegs4 L; L.lane = lane; L.w = *(const uint32_t*)(lst + 4 * lane);
            for (int k = 0; k < 4; k++) L.v[k] = (4u * lane + (uint32_t)k < A) ? next[lst[4 * lane + k]] : 0xffffffffu;
            st = withctx ? dc_decode_steps<true>(L, wwin, i, n, A, di, nwords, out, lane, ctxo, ranks) : dc_decode_steps<false>(L, wwin, i, n, A, di, nwords, out, lane);
            L.each(A, check);
        }
        if (!st && A > 1 && __ballot(bad)) st = RCX_E_MALFORMED;
    }
    if (lane == 0) {
        a.status[b] = st; a.out_len[b] = st ? 0 : withctx ? coff + 8ull * (di - 256u) : (uint64_t)n;
        if (a.in_used) a.in_used[b] = st ? 0 : 4ull * di;
    }
}

// =================================================================================================
// RLE -- src/rle.rs.  encode :82-122 (one-shot write+finish), decode :194-259
// =================================================================================================
__device__ __forceinline__ uint32_t rle_size(uint32_t reps)            // bytes flush() writes, :96-122
{
    if (reps == 1) return 1;
    uint32_t v = reps - 2, k = 1;
    while (v >>= 7) k++;
    return 2 + k;
}
__device__ __forceinline__ void rle_put(uint8_t* out, uint32_t o, uint8_t byte, uint32_t reps)
{
    out[o] = byte;
    if (reps == 1) return;
    out[o + 1] = byte;
    uint32_t v = reps - 2, idx = o + 2;
    for (;;) {
        uint8_t x = (uint8_t)(v & 0x7f);
        v >>= 7;
        if (v == 0) { out[idx] = x | 0x80; break; }
        out[idx++] = x;
    }
}

// One wave per stream, 64 input bytes per step: run starts by neighbour compare, run lengths from the
// ballot of starts, output offsets from a wave prefix sum of the emitted sizes.  The last (open) run of a
// step is carried into the next step.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_rle_encode(rcx_kargs a)
{
    const unsigned w = threadIdx.x >> 6, lane = rcx_lane();
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));   // wave-uniform: keeps descriptors in SGPRs
    if (b >= a.nblocks) return;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint32_t n = (uint32_t)a.in_len[b];
    uint8_t* out = a.out_base + a.out_off[b];
    const uint64_t cap = a.out_cap[b];
    uint32_t o = 0;
    int st = RCX_OK;
    bool have = false; uint32_t cbyte = 0, clen = 0;         // carried open run
    for (uint32_t j = 0; j < n && !st; j += 64) {
        const uint32_t p = j + lane;
        const bool valid = p < n;
        const uint32_t x = valid ? in[p] : 0u;
        const uint32_t prev = __shfl_up(x, 1);
        const bool start = valid && (lane == 0 ? (!have || x != cbyte) : (x != prev));
        const unsigned long long S = __ballot(start);
        const uint32_t nvalid = n - j < 64 ? n - j : 64;
        if (!S) { clen += nvalid; continue; }                // the carried run covers the whole step
        const int first = __ffsll(S) - 1;
        // run starting at this lane ends at the next start (closed) or at the end of the step (open)
        const unsigned long long above = (lane == 63) ? 0ull : (S >> (lane + 1));
        const bool closed = start && above != 0;
        const uint32_t rl = closed ? (uint32_t)__ffsll(above) : 0u;
        uint32_t sz = closed ? rle_size(rl) : 0u;
        const uint32_t csz = have ? rle_size(clen + (uint32_t)first) : 0u;   // the carried run closes at `first`
        const uint32_t incl = rcx_wave_incl_scan(sz);
        const uint32_t total = csz + (uint32_t)__builtin_amdgcn_readlane(incl, 63);
        if ((uint64_t)o + total > cap) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
        if (have && lane == 0) rle_put(out, o, (uint8_t)cbyte, clen + (uint32_t)first);
        if (closed) rle_put(out, o + csz + incl - sz, (uint8_t)x, rl);
        o += total;
        const int lastl = 63 - __clzll(S);
        cbyte = __builtin_amdgcn_readlane(x, lastl);
        clen = nvalid - (uint32_t)lastl;
        have = true;
    }
    if (!st && have) {                                        // finish(): flush the open run, :62-66
        const uint32_t csz = rle_size(clen);
        if ((uint64_t)o + csz > cap) st = RCX_E_OUTPUT_TOO_SMALL;
        else { if (lane == 0) rle_put(out, o, (uint8_t)cbyte, clen); o += csz; }
    }
    if (lane == 0) { a.status[b] = st; a.out_len[b] = st ? 0 : o; if (a.in_used) a.in_used[b] = n; }
}

// decode: the Clean/Single/Run state machine (:194-259) is a serial parse (a length byte is arbitrary data),
// walked wave-uniformly; run bodies are filled 64 bytes per step.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_rle_decode(rcx_kargs a)
{
    const unsigned w = threadIdx.x >> 6, lane = rcx_lane();
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));   // wave-uniform: keeps descriptors in SGPRs
    if (b >= a.nblocks) return;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint32_t n = (uint32_t)a.in_len[b];
    uint8_t* out = a.out_base + a.out_off[b];
    const uint64_t cap = a.out_cap[b];
    uint64_t o = 0;
    int st = RCX_OK;
    uint32_t i = 0;
    while (i < n) {
        const uint32_t cur = __builtin_amdgcn_readfirstlane((uint32_t)in[i]);           // Clean -> Single(cur)
        if (i + 1 >= n) { if (o + 1 > cap) { st = RCX_E_OUTPUT_TOO_SMALL; break; } if (lane == 0) out[o] = (uint8_t)cur; o += 1; i += 1; break; }
        const uint32_t nx = __builtin_amdgcn_readfirstlane((uint32_t)in[i + 1]);
        if (nx != cur) {                                       // Single(cur) followed by a different byte
            // a stretch of singles: emit bytes while in[k] != in[k+1], 64 per step
            uint32_t k = i;
            for (;;) {
                const uint32_t p = k + lane;
                const bool stopper = (p + 1 >= n) || (in[p] == in[p + 1]);
                const unsigned long long m = __ballot(stopper);
                const uint32_t cnt = m ? (uint32_t)(__ffsll(m) - 1) : 64u;
                if (o + cnt > cap) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
                if (lane < cnt) out[o + lane] = in[p];
                o += cnt; k += cnt;
                if (m) break;
            }
            if (st) break;
            i = k;                                             // in[i] starts a pair or is the last byte
            continue;
        }
        // Run(RunBuilder): length bytes until one has bit 7, at most 9 (:151-158, :214-222)
        uint32_t q = i + 2, cntb = 0;
        uint64_t reps = 0;
        bool fin = false;
        while (q < n) {
            if (cntb >= 9) { st = RCX_E_RLE_LONG_RUN; break; }
            const uint32_t x = __builtin_amdgcn_readfirstlane((uint32_t)in[q]);
            q++;
            reps |= (uint64_t)(x & 0x7f) << (7 * cntb);
            cntb++;
            if (x & 0x80) { fin = true; break; }
        }
        if (st) break;
        (void)fin;                                             // EOF inside the header decodes what was read (:247-256)
        reps += 2;
        if (reps > cap - o) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
        for (uint64_t t = lane; t < reps; t += 64) out[o + t] = (uint8_t)cur;
        o += reps;
        i = q;
    }
    if (lane == 0) { a.status[b] = st; a.out_len[b] = st ? 0 : o; if (a.in_used) a.in_used[b] = n; }
}

// =================================================================================================
// Adaptive byte range coder -- src/entropy/ari/mod.rs:117-159 (RangeEncoder::process/query),
// table.rs:69-117 (Model::update/downscale/get_range/find_value), table.rs:185-273 (ByteEncoder/Decoder).
// One lane per stream.  Table in LDS, lane-interleaved: entry e of lane t at tab[e*64+t] (u16), plus
// 17 block sums of 16 entries (u16) so a cumulative frequency costs <= 16+15 reads, not 256.  The sums are
// exact integers, so every (lo, hi, total) equals the reference's linear scan.
// =================================================================================================
#define ARI_N 257
#define ARI_NB 17
struct AriTab {
    uint16_t* tab; uint16_t* bs; unsigned t; uint32_t total;
    __device__ __forceinline__ uint16_t& f(uint32_t e) { return tab[e * 64 + t]; }
    __device__ __forceinline__ uint16_t& s(uint32_t blk) { return bs[blk * 64 + t]; }
    __device__ void init()
    {
        for (uint32_t e = 0; e < ARI_N; e++) f(e) = 1;
        for (uint32_t k = 0; k < 16; k++) s(k) = 16;
        s(16) = 1;
        total = ARI_N;
    }
    __device__ void update(uint32_t v)                        // update(value, 10, 1), table.rs:69-79
    {
        const uint32_t add = (total >> 10) + 1;
        f(v) = (uint16_t)(f(v) + add);
        s(v >> 4) = (uint16_t)(s(v >> 4) + add);
        total += add;
        if (total >= 4096) {                                  // downscale, table.rs:82-91 (cut_shift = 1)
            total = 0;
            for (uint32_t k = 0; k < 16; k++) {
                uint32_t x[16], bsum = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) x[j] = f(16 * k + (uint32_t)j);
#pragma unroll
                for (int j = 0; j < 16; j++) { x[j] = (x[j] + 1) >> 1; f(16 * k + (uint32_t)j) = (uint16_t)x[j]; bsum += x[j]; }
                s(k) = (uint16_t)bsum;
                total += bsum;
            }
            { const uint32_t x = ((uint32_t)f(256) + 1) >> 1; f(256) = (uint16_t)x; s(16) = (uint16_t)x; total += x; }
        }
    }
    // Both lookups read the 16 block sums and the 16 entries of one block UNCONDITIONALLY (two batches of
    // independent LDS reads = two round trips per symbol) and select with predicates; a data-dependent loop of
    // dependent reads cost ~30 round trips per symbol.
    __device__ __forceinline__ void range_of(uint32_t v, uint32_t& lo, uint32_t& hi)      // get_range, table.rs:100-103
    {
        const uint32_t kb = v >> 4, jb = v & 15u, e0 = v & ~15u;
        uint32_t sk[16], fe[16];
#pragma unroll
        for (int k = 0; k < 16; k++) sk[k] = s((uint32_t)k);
#pragma unroll
        for (int j = 0; j < 16; j++) { const uint32_t e = e0 + (uint32_t)j; fe[j] = f(e < ARI_N ? e : ARI_N - 1); }
        uint32_t l = 0, fv = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) l += (uint32_t)k < kb ? sk[k] : 0u;
#pragma unroll
        for (int j = 0; j < 16; j++) { l += (uint32_t)j < jb ? fe[j] : 0u; fv = (uint32_t)j == jb ? fe[j] : fv; }
        lo = l; hi = l + fv;
    }
    // find_value for offset = x / range WITHOUT the division: c <= x / range  <=>  c * range <= x  (integers, range > 0), and every
    // product is a 24 x 24-bit one (c <= total < 2^13; range <= (2^32 - 1) / 257 < 2^24; c * range <= hai - low < 2^32): v_mul_u32_u24.
    __device__ __forceinline__ uint32_t find_x(uint32_t x, uint32_t range, uint32_t& lo, uint32_t& hi)
    {
        uint32_t sk[16];
#pragma unroll
        for (int k = 0; k < 16; k++) sk[k] = s((uint32_t)k);
        uint32_t c = 0, l = 0, kb = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) { c += sk[k]; const bool below = __umul24(c, range) <= x; kb += below ? 1u : 0u; l = below ? c : l; }
        const uint32_t e0 = 16u * kb;
        uint32_t fe[16];
#pragma unroll
        for (int j = 0; j < 16; j++) { const uint32_t e = e0 + (uint32_t)j; fe[j] = f(e < ARI_N ? e : ARI_N - 1); }
        uint32_t jb = 0, h = 0;
        c = l;
        bool found = false;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t nc = c + fe[j];
            const bool hit = !found && __umul24(nc, range) > x;
            if (hit) { lo = c; h = nc; jb = (uint32_t)j; }
            found = found || hit;
            c = nc;
        }
        hi = h;
        return e0 + jb;
    }
    __device__ __forceinline__ uint32_t find(uint32_t offset, uint32_t& lo, uint32_t& hi)  // find_value, table.rs:105-117
    {
        uint32_t sk[16];
#pragma unroll
        for (int k = 0; k < 16; k++) sk[k] = s((uint32_t)k);
        uint32_t c = 0, l = 0, kb = 0;                        // kb = first block whose cumulative end exceeds offset (16: the EOF entry)
#pragma unroll
        for (int k = 0; k < 16; k++) { c += sk[k]; const bool below = c <= offset; kb += below ? 1u : 0u; l = below ? c : l; }
        const uint32_t e0 = 16u * kb;
        uint32_t fe[16];
#pragma unroll
        for (int j = 0; j < 16; j++) { const uint32_t e = e0 + (uint32_t)j; fe[j] = f(e < ARI_N ? e : ARI_N - 1); }
        uint32_t jb = 0, h = 0;
        c = l;
        bool found = false;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t nc = c + fe[j];
            const bool hit = !found && nc > offset;
            if (hit) { lo = c; h = nc; jb = (uint32_t)j; }
            found = found || hit;
            c = nc;
        }
        hi = h;
        return e0 + jb;
    }
};
struct AriRange {                                              // RangeEncoder, mod.rs:67-91
    uint32_t low, hai;
    __device__ __forceinline__ unsigned process(uint32_t total, uint32_t from, uint32_t to, uint8_t* o4)   // :117-150
    {
        const uint32_t range = (hai - low) / total;
        uint32_t lo = low + range * from, hi = low + range * to;
        unsigned k = 0;
        for (;;) {
            if (((lo ^ hi) & 0xff000000u) != 0) {
                if (hi - lo > (1u << 14)) break;
                const uint32_t lim = hi & 0xff000000u;
                if (hi - lim >= lim - lo) lo = lim; else hi = lim - 1;
            }
            o4[k++] = (uint8_t)(lo >> 24);
            lo <<= 8; hi <<= 8;
        }
        low = lo; hai = hi;
        return k;
    }
};

// Sequential byte source for one lane: 8 bytes per global load, the next word requested 8 bytes ahead of its use
// (a dependent byte load per symbol is ~1 us of latency per symbol).
struct AriBytes {
    const uint8_t* in; uint64_t n, p; uint64_t cw, nw;
    __device__ __forceinline__ uint64_t load8(uint64_t q) const
    {
        if (q + 8 <= n) return *(const rcx_u64_u*)(in + q);
        uint64_t w = 0;
        for (uint64_t i = q; i < n; i++) w |= (uint64_t)in[i] << (8 * (i - q));
        return w;
    }
    __device__ __forceinline__ void start(const uint8_t* in_, uint64_t n_) { in = in_; n = n_; p = 0; cw = 0; nw = load8(0); }
    __device__ __forceinline__ uint32_t next()                 // caller checks p < n
    {
        if ((p & 7u) == 0) { cw = nw; nw = load8(p + 8); }
        const uint32_t v = (uint32_t)(cw >> (8 * (p & 7u))) & 0xffu;
        p++;
        return v;
    }
};

__global__ __launch_bounds__(64) void k_ari_byte(rcx_kargs a, int decode)
{
    __shared__ uint16_t s_tab[ARI_N * 64];
    __shared__ uint16_t s_bs[ARI_NB * 64];
    const unsigned t = threadIdx.x;
    const uint32_t b = blockIdx.x * 64 + t;
    if (b >= a.nblocks) return;
    AriTab T; T.tab = s_tab; T.bs = s_bs; T.t = t; T.init();
    AriRange R; R.low = 0; R.hai = 0xffffffffu;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint64_t n = a.in_len[b];
    uint8_t* out = a.out_base + a.out_off[b];
    const uint64_t cap = a.out_cap[b];
    uint64_t o = 0, used = 0;
    int st = RCX_OK;
    uint8_t tmp[4];
    AriBytes src; src.start(in, n);
    if (!decode) {                                             // ByteEncoder::write + finish, table.rs:203-219
        for (uint64_t i = 0; i <= n; i++) {
            const uint32_t v = i < n ? src.next() : 256u;      // EOF symbol on finish()
            uint32_t lo, hi;
            T.range_of(v, lo, hi);
            const unsigned k = R.process(T.total, lo, hi, tmp);
            if (o + k > cap) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
            for (unsigned j = 0; j < k; j++) out[o + j] = tmp[j];
            o += k;
            if (i < n) T.update(v);
        }
        if (!st) {                                             // Encoder::finish: 4-byte BE tail of `low`, mod.rs:230-237
            if (o + 4 > cap) st = RCX_E_OUTPUT_TOO_SMALL;
            else { out[o] = (uint8_t)(R.low >> 24); out[o + 1] = (uint8_t)(R.low >> 16); out[o + 2] = (uint8_t)(R.low >> 8); out[o + 3] = (uint8_t)R.low; o += 4; }
        }
        used = n;
    } else {                                                   // ByteDecoder::read to EOF + finish, table.rs:256-272
        uint32_t code = 0; unsigned pending = 4;
        for (;;) {
            while (pending) {                                  // feed(), mod.rs:271-278
                if (src.p >= n) { st = RCX_E_MALFORMED; break; }   // mod.rs:282 feed().unwrap() panics
                code = (code << 8) + src.next(); pending--;
            }
            if (st) break;
            const uint32_t total = T.total;
            const uint32_t range = (R.hai - R.low) / total;    // query(), mod.rs:153-159
#ifdef RCX_ARI_DIV2
            const uint32_t offset = (code - R.low) / range;
            if (offset >= total) { st = RCX_E_MALFORMED; break; }   // table.rs:106 assert
            uint32_t lo, hi;
            const uint32_t v = T.find(offset, lo, hi);
#else
            const uint32_t x = code - R.low;                   // offset = x / range is never formed (see find_x)
            if (x >= __umul24(total, range)) { st = RCX_E_MALFORMED; break; }   // offset >= total: table.rs:106 assert (hai - low > 2^14 > total, so range >= 4)
            uint32_t lo, hi;
            const uint32_t v = T.find_x(x, range, lo, hi);
#endif
            pending = R.process(total, lo, hi, tmp);
            if (v == 256) break;
            if (o >= cap) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
            T.update(v);
            out[o++] = (uint8_t)v;
        }
        uint64_t p = src.p;
        if (!st) { while (pending) { if (p >= n) { st = RCX_E_EOF; break; } p++; pending--; } }   // finish(), mod.rs:289-292
        used = p;
    }
    a.status[b] = st; a.out_len[b] = o; if (a.in_used) a.in_used[b] = used;
}

// -------------------------------------------------------------------------------------------------
// The other two models of src/entropy/ari, driven the way the reference's own tests drive them (they have no stream
// codec of their own): MODE 0 = bin::Model (bin.rs:17-103), 8 binary decisions per byte, LSB first (test.rs:22-50);
// MODE 1 = table::SumProxy over two 16-entry tables for the high nibble + bin::SumProxy over two binary models for
// the 4 low bits (table.rs:127-180, bin.rs:112-167, test.rs:91-148); MODE 2 = apm::Bit passed through an apm::Gate, 8
// decisions per byte (apm.rs:36-198, test.rs:150-182): `scratch` holds the stretch table (Bit::to_wide for every flat
// probability, 4096 x i16, 0x8000 where the reference's to_i16().unwrap() panics) and the 17 initial gate bins, both
// computed on the host with libm's logf / expf exactly as the reference computes them; everything on the device is
// integer.  One lane per stream; the 2 x 16 frequencies of MODE 1 / the 17 gate bins of MODE 2 sit in LDS, lane-strided.
// No coding here has an end marker: the decoder produces exactly out_cap[b] bytes.
struct AriBin {                                                // bin::Model
    uint32_t zero, total, rate;
    __device__ __forceinline__ void init(uint32_t threshold, uint32_t r) { zero = threshold >> 1; total = threshold; rate = r; }   // new_flat :30-37
    __device__ __forceinline__ void update(uint32_t bit)       // :60-82
    {
        if (bit) zero -= zero >> rate; else zero += (total - zero) >> rate;
    }
};
struct AriTab16 {                                              // table::Model with 16 values, cut_shift 1
    uint16_t* tab; unsigned t; uint32_t total;
    __device__ __forceinline__ uint16_t& f(uint32_t e) { return tab[e * 64u + t]; }
    __device__ __forceinline__ void init() { for (uint32_t e = 0; e < 16; e++) f(e) = 1; total = 16; }
    __device__ __forceinline__ void load(uint32_t* x)
    {
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = f((uint32_t)j);
    }
    __device__ __forceinline__ void update(uint32_t v, uint32_t add_log, uint32_t threshold)   // table.rs:69-91
    {
        const uint32_t add = (total >> add_log) + 1u;
        f(v) = (uint16_t)(f(v) + (uint16_t)add);
        total += add;
        if (total >= threshold) {
            total = 0;
            for (uint32_t e = 0; e < 16; e++) { const uint32_t x = ((uint32_t)f(e) + 1u) >> 1; f(e) = (uint16_t)x; total += x; }
        }
    }
};

template <int MODE>
__global__ __launch_bounds__(64) void k_ari_model(rcx_kargs a, int decode, uint32_t rate)
{
    __shared__ uint16_t s_t0[MODE == 1 ? 16 * 64 : MODE == 2 ? 17 * 64 : 1];
    __shared__ uint16_t s_t1[MODE == 1 ? 16 * 64 : 1];
    const unsigned t = threadIdx.x;
    const uint32_t b = blockIdx.x * 64 + t;
    if (b >= a.nblocks) return;
    const uint32_t threshold = (1u << 14) >> 3;               // RANGE_DEFAULT_THRESHOLD >> 3, test.rs:27,96
    AriBin B0, B1;
    AriTab16 T0, T1;
    B0.init(threshold, MODE ? 3u : rate); B1.init(threshold, 5u);
    T0.tab = s_t0; T1.tab = s_t1; T0.t = T1.t = t; T0.total = T1.total = 16;
    if (MODE == 1) { T0.init(); T1.init(); }
    // MODE 2 state: the Bit's flat probability, the gate bins (LDS), this decision's bin index
    const int16_t* stretch = (const int16_t*)a.scratch;
    uint32_t apm_fp = 2048u; int apm_idx = 0;
    if (MODE == 2) for (uint32_t e = 0; e < 17; e++) s_t0[e * 64u + t] = ((const uint16_t*)a.scratch)[4096 + e];
    AriRange R; R.low = 0; R.hai = 0xffffffffu;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint64_t n = a.in_len[b];
    uint8_t* out = a.out_base + a.out_off[b];
    const uint64_t cap = a.out_cap[b];
    uint64_t o = 0, used = 0;
    int st = RCX_OK;
    uint8_t tmp[4];
    AriBytes src; src.start(in, n);
    // the binary decision under the current model(s): (zero, total) of bin::Model or of the 1:1 >>1 SumProxy
    auto apm_upd = [&](uint32_t fp, uint32_t bit) -> uint32_t {          // Bit::update(value, 10, 0), apm.rs:78-101 (u16 state)
        return (bit ? fp - (fp >> 10) : fp + ((4096u - fp) >> 10)) & 0xffffu;
    };
    auto bin_zero = [&]() -> uint32_t {
        if (MODE == 2) {                                       // gate.pass(&bit), apm.rs:157-173
            const int wp = (int)stretch[apm_fp & 4095u];
            const int idx = (wp + 2048) >> 8;
            if (wp == -32768 || idx < 0 || idx > 15) { st = RCX_E_MALFORMED; return 0u; }   // the reference panics (unwrap / bounds)
            apm_idx = idx;
            const uint32_t w = (uint32_t)wp & 255u;
            return (((uint32_t)s_t0[(uint32_t)idx * 64u + t] * (256u - w) + (uint32_t)s_t0[((uint32_t)idx + 1u) * 64u + t] * w) >> 8) & 0xffffu;
        }
        return MODE ? (B0.zero + B1.zero) >> 1 : B0.zero;
    };
    auto bin_total = [&]() -> uint32_t { return MODE == 2 ? 4096u : MODE ? (B0.total + B1.total) >> 1 : B0.total; };
    auto bin_update = [&](uint32_t bit) {
        if (MODE == 2) {
            apm_fp = apm_upd(apm_fp, bit);
            uint16_t& g0 = s_t0[(uint32_t)apm_idx * 64u + t]; uint16_t& g1 = s_t0[((uint32_t)apm_idx + 1u) * 64u + t];
            g0 = (uint16_t)apm_upd(g0, bit); g1 = (uint16_t)apm_upd(g1, bit);
            return;
        }
        B0.update(bit); if (MODE) B1.update(bit);
    };
    if (!decode) {
        auto put = [&](uint32_t total, uint32_t lo, uint32_t hi) {
            const unsigned k = R.process(total, lo, hi, tmp);
            if (o + k > cap) { st = RCX_E_OUTPUT_TOO_SMALL; return; }
            for (unsigned j = 0; j < k; j++) out[o + j] = tmp[j];
            o += k;
        };
        for (uint64_t i = 0; i < n && !st; i++) {
            const uint32_t v = src.next();
            if (MODE == 1) {                                   // high nibble under 2*t0 + 1*t1, table.rs:150-155
                const uint32_t high = v >> 4;
                uint32_t x0[16], x1[16];
                T0.load(x0); T1.load(x1);
                uint32_t lo = 0, fv = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const uint32_t w = 2u * x0[j] + x1[j];
                    lo += (uint32_t)j < high ? w : 0u; fv = (uint32_t)j == high ? w : fv;
                }
                put(2u * T0.total + T1.total, lo, lo + fv);
                T0.update(high, 10, threshold); T1.update(high, 5, threshold);
            }
            for (int k = 0; k < (MODE == 1 ? 4 : 8) && !st; k++) {
                const uint32_t bit = (v >> k) & 1u;
                const uint32_t z = bin_zero(), tot = bin_total();
                if (st) break;
                if (bit) put(tot, z, tot); else put(tot, 0, z);     // get_range, bin.rs:86-92 / apm.rs:104-111
                bin_update(bit);
            }
        }
        if (!st) {                                             // Encoder::finish, mod.rs:230-237
            if (o + 4 > cap) st = RCX_E_OUTPUT_TOO_SMALL;
            else { out[o] = (uint8_t)(R.low >> 24); out[o + 1] = (uint8_t)(R.low >> 16); out[o + 2] = (uint8_t)(R.low >> 8); out[o + 3] = (uint8_t)R.low; o += 4; }
        }
        used = n;
    } else {
        uint32_t code = 0; unsigned pending = 4;
        auto feed = [&]() {                                    // mod.rs:271-278; the tests' decode().unwrap() panics at the end of input
            while (pending) {
                if (src.p >= n) { st = RCX_E_MALFORMED; return; }
                code = (code << 8) + src.next(); pending--;
            }
        };
        for (; o < cap && !st; ) {
            uint32_t v = 0;
            if (MODE == 1) {
                feed(); if (st) break;
                const uint32_t tot = 2u * T0.total + T1.total;
                const uint32_t offset = (code - R.low) / ((R.hai - R.low) / tot);
                if (offset >= tot) { st = RCX_E_MALFORMED; break; }     // table.rs:158 assert
                uint32_t x0[16], x1[16];
                T0.load(x0); T1.load(x1);
                uint32_t c = 0, lo = 0, hi = 0, high = 0; bool found = false;
#pragma unroll
                for (int j = 0; j < 16; j++) {                 // SumProxy::find_value, table.rs:157-173
                    const uint32_t nc = c + 2u * x0[j] + x1[j];
                    const bool hit = !found && nc > offset;
                    if (hit) { lo = c; hi = nc; high = (uint32_t)j; }
                    found = found || hit; c = nc;
                }
                if (!found) { st = RCX_E_MALFORMED; break; }
                pending = R.process(tot, lo, hi, tmp);
                T0.update(high, 10, threshold); T1.update(high, 5, threshold);
                v = high << 4;
            }
            for (int k = 0; k < (MODE == 1 ? 4 : 8); k++) {
                const uint32_t z = bin_zero(), tot = bin_total();
                if (st) break;
                feed(); if (st) break;
                const uint32_t offset = (code - R.low) / ((R.hai - R.low) / tot);
                if (offset >= tot) { st = RCX_E_MALFORMED; break; }     // bin.rs:95 assert
                const uint32_t bit = offset < z ? 0u : 1u;              // find_value, bin.rs:94-103
                pending = bit ? R.process(tot, z, tot, tmp) : R.process(tot, 0, z, tmp);
                bin_update(bit);
                v += bit << k;
            }
            if (st) break;
            out[o++] = (uint8_t)v;
        }
        used = src.p;
    }
    a.status[b] = st; a.out_len[b] = o; if (a.in_used) a.in_used[b] = used;
}

// -------------------------------------------------------------------------------------------------
// Same coder, one WAVE per stream (the pipeline has a few thousand long streams, not 64 K short ones: with one lane
// per stream that is 60 waves on the whole chip, each paying ~250 dependent instructions per symbol).  The 256 byte
// frequencies live in registers, 4 per lane (entry e in lane e>>2), the EOF entry and every coder variable are
// wave-uniform; each lane keeps the cumulative frequency in front of its entries, patched by `add` after an update
// (lanes above the updated one) and rebuilt by one DPP scan after a downscale.  (lo, hi) of a symbol are computed by
// every lane and read from the owner with v_readlane; the decoder finds the owner with a ballot.  Input is held 64
// bytes at a time across the lanes (next chunk prefetched), output is staged 64 bytes across the lanes and stored
// coalesced.  All integers equal the reference's (table.rs:69-117, mod.rs:117-159).
struct AriWave {
    uint32_t f0, f1, f2, f3, base;        // per lane: entries 4*lane .. 4*lane+3 and the cumulative frequency before them
    uint32_t f256, total;                 // uniform
    unsigned lane;
    __device__ void init() { f0 = f1 = f2 = f3 = 1; base = 4 * lane; f256 = 1; total = ARI_N; }
    __device__ __forceinline__ void rescan()
    {
        const uint32_t sl = f0 + f1 + f2 + f3;
        const uint32_t inc = rcx_wave_incl_scan(sl);
        base = inc - sl;
        total = RCX_UNI(__builtin_amdgcn_readlane(inc, 63)) + f256;
    }
    __device__ __forceinline__ void range_of(uint32_t v, uint32_t& lo, uint32_t& hi)      // v uniform
    {
        if (v == 256u) { lo = total - f256; hi = total; return; }
        const uint32_t idx = v & 3u, lv = v >> 2;
        const uint32_t l = base + (idx > 0 ? f0 : 0u) + (idx > 1 ? f1 : 0u) + (idx > 2 ? f2 : 0u);
        const uint32_t fv = idx == 0 ? f0 : idx == 1 ? f1 : idx == 2 ? f2 : f3;
        lo = RCX_UNI(__builtin_amdgcn_readlane(l, lv));
        hi = lo + RCX_UNI(__builtin_amdgcn_readlane(fv, lv));
    }
    __device__ __forceinline__ uint32_t find(uint32_t offset, uint32_t& lo, uint32_t& hi)  // offset uniform, < total
    {
        const uint32_t c1 = base + f0, c2 = c1 + f1, c3 = c2 + f2, c4 = c3 + f3;
        const unsigned long long own = __ballot(base <= offset && offset < c4);
        if (!own) { lo = total - f256; hi = total; return 256u; }
        const uint32_t lv = (uint32_t)__ffsll(own) - 1u;
        const uint32_t idx = (c1 <= offset ? 1u : 0u) + (c2 <= offset ? 1u : 0u) + (c3 <= offset ? 1u : 0u);
        const uint32_t l = idx == 0 ? base : idx == 1 ? c1 : idx == 2 ? c2 : c3;
        const uint32_t h = idx == 0 ? c1 : idx == 1 ? c2 : idx == 2 ? c3 : c4;
        lo = RCX_UNI(__builtin_amdgcn_readlane(l, lv));
        hi = RCX_UNI(__builtin_amdgcn_readlane(h, lv));
        return 4u * lv + RCX_UNI(__builtin_amdgcn_readlane(idx, lv));
    }
    __device__ __forceinline__ void update(uint32_t v)                                     // update(value, 10, 1), v < 256 uniform
    {
        const uint32_t add = (total >> 10) + 1, idx = v & 3u, lv = v >> 2;
        const uint32_t mine = lane == lv ? add : 0u;
        if (idx == 0) f0 += mine; else if (idx == 1) f1 += mine; else if (idx == 2) f2 += mine; else f3 += mine;
        base += lane > lv ? add : 0u;
        total += add;
        if (total >= 4096) {                                  // downscale, table.rs:82-91 (cut_shift = 1)
            f0 = (f0 + 1) >> 1; f1 = (f1 + 1) >> 1; f2 = (f2 + 1) >> 1; f3 = (f3 + 1) >> 1; f256 = (f256 + 1) >> 1;
            rescan();
        }
    }
};

template <int WAVES, bool DEC>
__global__ __launch_bounds__(64 * WAVES) void k_ari_byte_wave(rcx_kargs a)
{
    const int decode = DEC ? 1 : 0;
    // ENCODER: low / hai are wave-uniform but kept in VGPRs, so the coder's arithmetic (a 32-bit division per symbol) runs on
    // the vector ALU instead of the CU's single scalar unit (3815 streams: 148 -> 120 ms).  The DECODER has two dependent
    // divisions per symbol and is latency bound: the scalar form is faster there (175 vs 194 ms).
#define ARIW_V(x) (DEC ? (uint32_t)(x) : RCX_VGPR(x))
#define ARIW_ANY(c) (DEC ? (bool)(c) : (__ballot(c) != 0))
    const unsigned w = threadIdx.x >> 6, lane = rcx_lane();
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));
    if (b >= a.nblocks) return;
    AriWave T; T.lane = lane; T.init();
    uint32_t low = ARIW_V(0), hai = ARIW_V(0xffffffffu);
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint64_t n = a.in_len[b];
    uint8_t* out = a.out_base + a.out_off[b];
    const uint64_t cap = a.out_cap[b];
    uint64_t o = 0, used = 0, p = 0;
    int st = RCX_OK;
    uint32_t obuf = 0;                                         // lane j: output byte (o & ~63) + j
    // input window: lane j holds in[(p & ~63) + j]; `nxt` is the following 64 bytes, requested one window early
    uint32_t cur = lane < n ? in[lane] : 0u, nxt = 64 + lane < n ? in[64 + lane] : 0u;
#define ARIW_NEXT_BYTE(dst)                                                                             \
    do {                                                                                                \
        dst = RCX_UNI(__builtin_amdgcn_readlane(cur, (uint32_t)p & 63u));                               \
        p++;                                                                                            \
        if ((p & 63u) == 0) { cur = nxt; const uint64_t q = p + 64 + lane; nxt = q < n ? in[q] : 0u; }  \
    } while (0)
#define ARIW_PUT_BYTE(x)                                                                                \
    do {                                                                                                \
        obuf = lane == ((uint32_t)o & 63u) ? (x) : obuf;                                                \
        o++;                                                                                            \
        if ((o & 63u) == 0) out[o - 64 + lane] = (uint8_t)obuf;                                         \
    } while (0)
    // RangeEncoder::process, mod.rs:117-150, on wave-uniform values: emits 0..4 bytes
#define ARIW_PROCESS(total_, from_, to_, EMIT)                                                          \
    do {                                                                                                \
        const uint32_t range_ = DEC ? RCX_UNI((hai - low) / (total_)) : (hai - low) / ARIW_V(total_);   \
        uint32_t lo_ = low + range_ * (from_), hi_ = low + range_ * (to_);                              \
        for (;;) {                                                                                      \
            if (ARIW_ANY(((lo_ ^ hi_) & 0xff000000u) != 0)) {                                           \
                if (ARIW_ANY(hi_ - lo_ > (1u << 14))) break;                                            \
                const uint32_t lim_ = hi_ & 0xff000000u;                                                \
                const bool up_ = hi_ - lim_ >= lim_ - lo_;                                              \
                lo_ = up_ ? lim_ : lo_; hi_ = up_ ? hi_ : lim_ - 1;                                     \
            }                                                                                           \
            EMIT(lo_ >> 24);                                                                            \
            lo_ <<= 8; hi_ <<= 8;                                                                       \
        }                                                                                               \
        low = lo_; hai = hi_;                                                                           \
    } while (0)
    if (!decode) {                                             // ByteEncoder::write + finish, table.rs:203-219
        for (uint64_t i = 0; i <= n && !st; i++) {
            uint32_t v = 256u;                                 // EOF symbol on finish()
            if (i < n) ARIW_NEXT_BYTE(v);
            uint32_t lo, hi;
            T.range_of(v, lo, hi);
#define ARIW_EMIT_ENC(x) do { if (o >= cap) { st = RCX_E_OUTPUT_TOO_SMALL; } else { ARIW_PUT_BYTE(x); } } while (0)
            ARIW_PROCESS(T.total, lo, hi, ARIW_EMIT_ENC);
            if (st) break;
            if (i < n) T.update(v);
        }
        if (!st) {                                             // Encoder::finish: 4-byte BE tail of `low`, mod.rs:230-237
            if (o + 4 > cap) st = RCX_E_OUTPUT_TOO_SMALL;
            else { ARIW_PUT_BYTE(low >> 24); ARIW_PUT_BYTE((low >> 16) & 0xffu); ARIW_PUT_BYTE((low >> 8) & 0xffu); ARIW_PUT_BYTE(low & 0xffu); }
        }
        used = n;
    } else {                                                   // ByteDecoder::read to EOF + finish, table.rs:256-272
        uint32_t code = 0; unsigned pending = 4;
        for (;;) {
            while (pending) {                                  // feed(), mod.rs:271-278
                if (p >= n) { st = RCX_E_MALFORMED; break; }   // mod.rs:282 feed().unwrap() panics
                uint32_t x; ARIW_NEXT_BYTE(x);
                code = (code << 8) + x; pending--;
            }
            if (st) break;
            const uint32_t total = T.total;
            const uint32_t range = RCX_UNI((hai - low) / total);   // query(), mod.rs:153-159
            const uint32_t offset = RCX_UNI((code - low) / range);
            if (offset >= total) { st = RCX_E_MALFORMED; break; }  // table.rs:106 assert
            uint32_t lo, hi;
            const uint32_t v = T.find(offset, lo, hi);
#define ARIW_EMIT_DEC(x) do { pending++; } while (0)
            ARIW_PROCESS(total, lo, hi, ARIW_EMIT_DEC);
            if (v == 256) break;
            if (o >= cap) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
            T.update(v);
            ARIW_PUT_BYTE(v);
        }
        if (!st) { while (pending) { if (p >= n) { st = RCX_E_EOF; break; } p++; pending--; } }   // finish(), mod.rs:289-292
        used = p;
    }
    if ((o & 63u) && lane < (o & 63u)) out[(o & ~(uint64_t)63) + lane] = (uint8_t)obuf;     // the staged tail
    if (lane == 0) { a.status[b] = st; a.out_len[b] = o; if (a.in_used) a.in_used[b] = used; }
#undef ARIW_NEXT_BYTE
#undef ARIW_PUT_BYTE
#undef ARIW_PROCESS
#undef ARIW_EMIT_ENC
#undef ARIW_EMIT_DEC
#undef ARIW_V
#undef ARIW_ANY
}

// -------------------------------------------------------------------------------------------------
// The same coder, a QUAD of lanes per stream (k_ari_byte_quad): 16 streams a wave.
// Why: one lane per stream is the cheapest form in instructions (64 streams per wave instruction) but a symbol is a chain of
// ~300 dependent instructions for its lane -- sixteen block sums and sixteen entries read, multiplied, compared and selected one
// after the other -- and BASELINE config 5 has 15 260 streams: 239 waves on 1024 SIMDs, each alone on its SIMD and bound by that
// chain (18 ms, the largest kernel of the pipeline's decode).  Four lanes share a stream here: each reads FOUR sums / entries with
// one 8-byte LDS load, the quad's prefix and the position of the hit come from quad-permute DPP moves (VALU, no LDS round trip),
// the scalar state (low, hai, code, total) is kept identically in all four lanes, and the one division of a symbol is a
// reciprocal + two corrections (exact for the divisors a table total can take, 257..8191: the compiler's 32-bit division is two
// v_mul_hi and ~30 instructions).  A symbol is ~150 instructions; 15 260 streams are 954 waves, one per SIMD.
// Table in LDS, stream-major: 272 u16 entries (257 + zero padding: a padded entry never wins a search and stays zero when the
// table is halved), then 20 u16 block sums (17 + padding): 584 bytes a stream, every quad load 8-byte aligned.
// -------------------------------------------------------------------------------------------------
#define ARIQ_TAB 272
#define ARIQ_BS 20
#define RCX_QPERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int CTRL> __device__ __forceinline__ uint32_t rcx_qp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t rcx_quad_sum(uint32_t v) { v += rcx_qp<RCX_QPERM(1, 0, 3, 2)>(v); v += rcx_qp<RCX_QPERM(2, 3, 0, 1)>(v); return v; }
__device__ __forceinline__ uint32_t rcx_quad_max(uint32_t v)
{
    uint32_t t = rcx_qp<RCX_QPERM(1, 0, 3, 2)>(v); v = t > v ? t : v;
    t = rcx_qp<RCX_QPERM(2, 3, 0, 1)>(v); return t > v ? t : v;
}
__device__ __forceinline__ uint32_t rcx_quad_min(uint32_t v)
{
    uint32_t t = rcx_qp<RCX_QPERM(1, 0, 3, 2)>(v); v = t < v ? t : v;
    t = rcx_qp<RCX_QPERM(2, 3, 0, 1)>(v); return t < v ? t : v;
}
// the sum of v over the quad's lanes below q
__device__ __forceinline__ uint32_t rcx_quad_excl(uint32_t v, unsigned q)
{
    const uint32_t a = rcx_qp<RCX_QPERM(0, 0, 0, 0)>(v), b = rcx_qp<RCX_QPERM(1, 1, 1, 1)>(v), c = rcx_qp<RCX_QPERM(2, 2, 2, 2)>(v);
    return (q > 0 ? a : 0u) + (q > 1 ? b : 0u) + (q > 2 ? c : 0u);
}
#ifndef RCX_RCPF
#define RCX_RCPF(x) __builtin_amdgcn_rcpf(x)
#endif
// n / d for 257 <= d < 8192, exact whatever the last bits of the reciprocal are (checked over 10^9 (n, d) with the reciprocal
// off by up to two ulps either way): the first estimate is within 5 of the quotient, the remainder then fits a float exactly.
__device__ __forceinline__ uint32_t rcx_div_u13(uint32_t n, uint32_t d)
{
    const float r = RCX_RCPF((float)d);
    uint32_t q = (uint32_t)((float)n * r);
    int32_t rem = (int32_t)(n - __umul24(q, d));               // q <= (2^32 - 1) / 257 < 2^24
    const int32_t q2 = (int32_t)__builtin_floorf((float)rem * r);
    q += (uint32_t)q2; rem -= q2 * (int32_t)d;
    if (rem < 0) q--; else if (rem >= (int32_t)d) q++;
    return q;
}
struct AriQuad {
    uint16_t* tab; uint16_t* bs; unsigned q; uint32_t total;
    __device__ void init()
    {
        for (uint32_t e = q; e < ARIQ_TAB; e += 4) tab[e] = e < ARI_N ? 1 : 0;
        for (uint32_t k = q; k < ARIQ_BS; k += 4) bs[k] = k < 16 ? 16 : (k == 16 ? 1 : 0);
        total = ARI_N;
        rcx_wave_sync();
    }
    struct Hit { uint32_t cnt, lo, hi; };
    // four consecutive cumulative ends past c0 per lane (w4: their four u16 steps): how many ends e of the quad's sixteen have
    // e * range <= x (cnt), the largest such end or c0 (lo), the smallest end beyond (hi).  The ends only grow, so the flags are
    // ones then zeros.
    __device__ __forceinline__ Hit find4(uint64_t w4, uint32_t c0, uint32_t range, uint32_t x) const
    {
        const uint32_t lo32 = (uint32_t)w4, hi32 = (uint32_t)(w4 >> 32);
        const uint32_t l1 = lo32 & 0xffffu, l2 = l1 + (lo32 >> 16), l3 = l2 + (hi32 & 0xffffu), l4 = l3 + (hi32 >> 16);
        const uint32_t base = c0 + rcx_quad_excl(l4, q);
        const uint32_t e0 = base + l1, e1 = base + l2, e2 = base + l3, e3 = base + l4;
        const bool b0 = __umul24(e0, range) <= x, b1 = __umul24(e1, range) <= x, b2 = __umul24(e2, range) <= x, b3 = __umul24(e3, range) <= x;
        const uint32_t mylo = b3 ? e3 : b2 ? e2 : b1 ? e1 : b0 ? e0 : 0u;
        const uint32_t myhi = !b0 ? e0 : !b1 ? e1 : !b2 ? e2 : !b3 ? e3 : 0xffffffffu;
        Hit h;
        h.cnt = rcx_quad_sum((uint32_t)b0 + (uint32_t)b1 + (uint32_t)b2 + (uint32_t)b3);
        const uint32_t m = rcx_quad_max(mylo);
        h.lo = m > c0 ? m : c0;
        h.hi = rcx_quad_min(myhi);
        return h;
    }
    // find_value for offset = x / range without forming it (AriTab::find_x): -> the value, [lo, hi)
    __device__ __forceinline__ uint32_t find_x(uint32_t x, uint32_t range, uint32_t& lo, uint32_t& hi) const
    {
        const Hit hb = find4(*(const uint64_t*)(bs + 4 * q), 0u, range, x);          // sixteen block sums (the 17th block is what is left)
        const Hit he = find4(*(const uint64_t*)(tab + 16 * hb.cnt + 4 * q), hb.lo, range, x);
        lo = he.lo; hi = he.hi;
        return 16u * hb.cnt + he.cnt;
    }
    // get_range, table.rs:100-103
    __device__ __forceinline__ void range_of(uint32_t v, uint32_t& lo, uint32_t& hi) const
    {
        const uint32_t kb = v >> 4, jb = v & 15u;
        const uint64_t sw = *(const uint64_t*)(bs + 4 * q), ew = *(const uint64_t*)(tab + (v & ~15u) + 4 * q);
        auto part = [&](uint64_t w4, uint32_t upto, uint32_t& at) -> uint32_t {       // the lane's entries with index < upto, and the one at upto
            const uint32_t f0 = (uint32_t)w4 & 0xffffu, f1 = (uint32_t)w4 >> 16, f2 = (uint32_t)(w4 >> 32) & 0xffffu, f3 = (uint32_t)(w4 >> 48);
            const uint32_t i0 = 4 * q;
            at = i0 == upto ? f0 : i0 + 1 == upto ? f1 : i0 + 2 == upto ? f2 : i0 + 3 == upto ? f3 : 0u;
            return (i0 < upto ? f0 : 0u) + (i0 + 1 < upto ? f1 : 0u) + (i0 + 2 < upto ? f2 : 0u) + (i0 + 3 < upto ? f3 : 0u);
        };
        uint32_t dummy, fv;
        const uint32_t sb = part(sw, kb, dummy);
        const uint32_t se = part(ew, jb, fv);
        lo = rcx_quad_sum(sb + se);
        hi = lo + rcx_quad_max(fv);
    }
    // update(value, 10, 1), table.rs:69-91.  Frequencies stay below 2^13, so a 32-bit add on the dword that holds the u16 is exact.
    __device__ __forceinline__ void update(uint32_t v)
    {
        const uint32_t add = (total >> 10) + 1;
        if (q == 0) {
            atomicAdd((uint32_t*)tab + (v >> 1), add << (16u * (v & 1u)));
            atomicAdd((uint32_t*)bs + (v >> 5), add << (16u * ((v >> 4) & 1u)));
        }
        total += add;
        rcx_wave_sync();
        if (total >= 4096) {                                  // downscale (cut_shift = 1): four entries a lane and step, halved in place
            total = 0;
            for (uint32_t k = 0; k < 17; k++) {
                uint64_t* pw = (uint64_t*)(tab + 16 * k + 4 * q);
                uint64_t w = *pw;
                w = ((w + 0x0001000100010001ull) >> 1) & 0x7fff7fff7fff7fffull;
                *pw = w;
                const uint32_t a = (uint32_t)w, b = (uint32_t)(w >> 32);
                const uint32_t bsum = rcx_quad_sum((a & 0xffffu) + (a >> 16) + (b & 0xffffu) + (b >> 16));
                if (q == 0) bs[k] = (uint16_t)bsum;
                total += bsum;
            }
            rcx_wave_sync();
        }
    }
};

// The decoder's byte source without a loop: `need` <= 4 bytes enter the code word at once.  cw / nw are the 8-byte words at
// P = p & ~7 and P + 8 (nw requested a word ahead of its use); the four bytes at p are cut out of the pair, reversed and
// shifted in -- sixteen streams a wave are in sixteen different phases, and with a loop per byte every symbol paid the longest
// refill of the sixteen (two passes of ~35 instructions on average; this is ~15).
struct AriWin {
    const uint8_t* in; uint32_t n, p; uint64_t cw, nw;
    __device__ __forceinline__ uint64_t load8(uint32_t q) const
    {
        if (q + 8u <= n && q + 8u >= q) return *(const rcx_u64_u*)(in + q);
        uint64_t w = 0;
        for (uint32_t i = q; i < n; i++) w |= (uint64_t)in[i] << (8u * (i - q));
        return w;
    }
    __device__ __forceinline__ void start(const uint8_t* in_, uint32_t n_) { in = in_; n = n_; p = 0; cw = load8(0); nw = n > 8u ? load8(8) : 0ull; }
    // (code << 8 * need) + the next `need` bytes, most significant first (feed(), mod.rs:271-278); the caller made sure p + need <= n
    __device__ __forceinline__ uint32_t take(uint32_t code, uint32_t need)
    {
        const uint32_t sh = 8u * (p & 7u);
        const uint64_t w = (cw >> sh) | ((nw << 1) << (63u - sh));
        const uint32_t be = __builtin_bswap32((uint32_t)w);
        code = (uint32_t)(((((uint64_t)code) << 32) | (uint64_t)be) << (8u * need) >> 32);
        const uint32_t P = p & ~7u;
        p += need;
        if ((p & ~7u) != P) { cw = nw; nw = (P + 16u < n && P + 16u >= P) ? load8(P + 16u) : 0ull; }
        return code;
    }
};

template <bool DEC>
__global__ __launch_bounds__(256) void k_ari_byte_quad(rcx_kargs a)
{
    __shared__ __align__(16) uint16_t s_q[64 * (ARIQ_TAB + ARIQ_BS)];
    const unsigned tid = threadIdx.x, q = tid & 3u, sl = tid >> 2;
    const uint32_t b = blockIdx.x * 64 + sl;
    if (b >= a.nblocks) return;                                // (whole quads leave)
    AriQuad T; T.tab = s_q + sl * (ARIQ_TAB + ARIQ_BS); T.bs = T.tab + ARIQ_TAB; T.q = q; T.init();
    // The four waves of a SIMD (one of each of a CU's four workgroups) would finish one after the other: among waves of one priority
    // a SIMD issues the oldest first, the kernel is bound by the vector ALU, and the last wave alone keeps it a quarter busy.  So the
    // priority order rotates: every RCX_ARI_ROTATE symbols a wave takes the level (its slot + the period's number) mod 4.
#ifndef RCX_ARI_ROTATE
#define RCX_ARI_ROTATE 64
#endif
    const uint32_t wslot = RCX_ARI_ROTATE ? (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) : 0u;   // HW_ID[3:0]: the wave's slot on its SIMD
    uint32_t rot = 0;
    auto rotate = [&]() {
        if (RCX_ARI_ROTATE && (rot++ & (uint32_t)(RCX_ARI_ROTATE - 1)) == 0u) {
            switch ((wslot + rot / (uint32_t)(RCX_ARI_ROTATE ? RCX_ARI_ROTATE : 1)) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
    };
    uint32_t low = 0, hai = 0xffffffffu;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint64_t n = a.in_len[b];
    uint8_t* out = a.out_base + a.out_off[b];
    const uint64_t cap = a.out_cap[b];
    uint64_t o = 0, used = 0;
    int st = RCX_OK;
    AriBytes src; src.start(in, n);
    // RangeEncoder::process, mod.rs:117-150, with the symbol's range already divided: the bytes that leave, most significant
    // first in the low bytes of `ob`; returns their number
    auto process = [&](uint32_t range, uint32_t from, uint32_t to, uint32_t& ob) -> unsigned {
        uint32_t lo_ = low + __umul24(range, from), hi_ = low + __umul24(range, to);
        // The loop below leaves through its `break` only; the bytes on which lo_ and hi_ agree leave first, whatever the range is, so
        // they are counted with one v_ffbh and shifted out at once (at most three: a fourth, or the underflow case, is the loop's).
        // Nearly every call ends there -- and a wave's loop ran as long as the longest of its sixteen streams'.
        const uint32_t x_ = lo_ ^ hi_;
        unsigned k = x_ ? (unsigned)__builtin_clz(x_) >> 3 : 3u;
        k = k < 3u ? k : 3u;
        ob = (uint32_t)(((uint64_t)lo_ << (8u * k)) >> 32);
        lo_ <<= 8u * k; hi_ <<= 8u * k;
        for (;;) {
            if (((lo_ ^ hi_) & 0xff000000u) != 0) {
                if (hi_ - lo_ > (1u << 14)) break;
                const uint32_t lim = hi_ & 0xff000000u;
                if (hi_ - lim >= lim - lo_) lo_ = lim; else hi_ = lim - 1;
            }
            ob = (ob << 8) | (lo_ >> 24);
            k++;
            lo_ <<= 8; hi_ <<= 8;
        }
        low = lo_; hai = hi_;
        return k;
    };
    if (!DEC) {                                                // ByteEncoder::write + finish, table.rs:203-219
        for (uint64_t i = 0; i <= n; i++) {
            rotate();
            const uint32_t v = i < n ? src.next() : 256u;
            uint32_t lo, hi, ob;
            T.range_of(v, lo, hi);
            const unsigned k = process(rcx_div_u13(hai - low, T.total), lo, hi, ob);
            if (o + k > cap) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
            if (q == 0) for (unsigned j = 0; j < k; j++) out[o + j] = (uint8_t)(ob >> (8 * (k - 1 - j)));
            o += k;
            if (i < n) T.update(v);
        }
        if (!st) {                                             // Encoder::finish: 4-byte BE tail of `low`, mod.rs:230-237
            if (o + 4 > cap) st = RCX_E_OUTPUT_TOO_SMALL;
            else { if (q == 0) { out[o] = (uint8_t)(low >> 24); out[o + 1] = (uint8_t)(low >> 16); out[o + 2] = (uint8_t)(low >> 8); out[o + 3] = (uint8_t)low; } o += 4; }
        }
        used = n;
    } else {                                                   // ByteDecoder::read to EOF + finish, table.rs:256-272
        uint32_t code = 0; unsigned pending = 4;
        uint64_t acc0 = 0, acc1 = 0;
        AriWin win; win.start(in, (uint32_t)n);
        for (;;) {
            rotate();
            if (pending > win.n - win.p) { win.p = win.n; st = RCX_E_MALFORMED; break; }     // feed(), mod.rs:271-278: the stream ends inside it
            code = win.take(code, pending);
            const uint32_t total = T.total;
            const uint32_t range = rcx_div_u13(hai - low, total);  // query(), mod.rs:153-159
            const uint32_t x = code - low;
            if (x >= __umul24(total, range)) { st = RCX_E_MALFORMED; break; }   // offset >= total: table.rs:106 assert
            uint32_t lo, hi, ob;
            const uint32_t v = T.find_x(x, range, lo, hi);
            pending = process(range, lo, hi, ob);
            if (v == 256) break;
            if (o >= cap) { st = RCX_E_OUTPUT_TOO_SMALL; break; }
            T.update(v);
            // Sixteen decoded bytes leave with two 8-byte stores (every stream of the wave is at the same o: a symbol a step).  A byte
            // store per symbol was what the kernel's time consisted of: the next refill of the byte window -- some stream's, nearly
            // every step -- is a wait for vmcnt(0), and on gfx9 that is a wait for the store of the step before to reach memory.
            const uint32_t ph = (uint32_t)o & 15u;
            const uint64_t vb = (uint64_t)v << (8u * (ph & 7u));
            if (ph & 8u) acc1 |= vb; else acc0 |= vb;
            o++;
            if (ph == 15u) {
                if (q == 0) { *(rcx_u64_u*)(out + o - 16) = acc0; *(rcx_u64_u*)(out + o - 8) = acc1; }
                acc0 = 0; acc1 = 0;
            }
        }
        if (q == 0) {                                          // the bytes still in the registers (the stream's end, or an error: what was decoded is delivered)
            const uint32_t rem = (uint32_t)o & 15u;
            for (uint32_t j = 0; j < rem; j++) out[o - rem + j] = (uint8_t)((j < 8u ? acc0 : acc1) >> (8u * (j & 7u)));
        }
        uint64_t p = win.p;
        if (!st) { while (pending) { if (p >= n) { st = RCX_E_EOF; break; } p++; pending--; } }   // finish(), mod.rs:289-292
        used = p;
    }
    if (q == 0) { a.status[b] = st; a.out_len[b] = o; if (a.in_used) a.in_used[b] = used; }
}

// -------------------------------------------------------------------------------------------------
static void launch_serial(hipStream_t s, int codec, rcx_kargs& k, int v, uint32_t param)
{
    const uint32_t n = k.nblocks;
    switch (codec) {
    case RCX_MTF_ENCODE: hipLaunchKernelGGL((k_mtf<4>), dim3((n + 3) / 4), dim3(256), 0, s, k, 0); break;
    case RCX_MTF_DECODE: hipLaunchKernelGGL((k_mtf<4>), dim3((n + 3) / 4), dim3(256), 0, s, k, 1); break;
    case RCX_DC_ENCODE: {                                       // param 1: with contexts; variant 1: the wave-per-block kernel only
        const bool chunks = !(param & 1u) && v != 1 && k.scratch && k.scratch_bytes >= dc_encode_scratch_bytes(n);
        if (chunks) {
            hipLaunchKernelGGL(k_dcx_prep, dim3(n), dim3(256), 0, s, k);
            hipLaunchKernelGGL(k_dcx_main, dim3(n), dim3(64), 0, s, k);
        }
        hipLaunchKernelGGL((k_dc_encode<4>), dim3((n + 3) / 4), dim3(256), 0, s, k, (int)(param & 1u), chunks ? 1 : 0);
        break; }
    case RCX_DC_DECODE: hipLaunchKernelGGL((k_dc_decode<4>), dim3((n + 3) / 4), dim3(256), 0, s, k, (int)(param & 1u)); break;
    case RCX_RLE_ENCODE: hipLaunchKernelGGL((k_rle_encode<4>), dim3((n + 3) / 4), dim3(256), 0, s, k); break;
    case RCX_RLE_DECODE: hipLaunchKernelGGL((k_rle_decode<4>), dim3((n + 3) / 4), dim3(256), 0, s, k); break;
    case RCX_ARI_BYTE_ENCODE: case RCX_ARI_BYTE_DECODE: {
        // Three kernels (benchmarks/ari_variant_sweep.py, 16 K symbols a stream, encode / decode ms): a WAVE per stream has the
        // shortest chain per symbol (7.8 / 9.0 up to 1024 streams) but holds a wave slot per stream (4096 streams: 10.7 / 15.7,
        // 16 384: 30 / 49); a QUAD of lanes per stream (16 streams a wave) takes 12.2 / 15.0 up to 16 K streams and stays ahead of
        // a LANE per stream (17.7 / 24.4) at every size measured (262 144 streams: 69 / 105 against 74 / 127).  So: waves below
        // 4096 streams, quads from there on; the lane-per-stream kernel is variant 1 (2 / 3 pin the other two).
        const int dec = codec == RCX_ARI_BYTE_DECODE ? 1 : 0;
        const int kind = v == 1 ? 1 : v == 2 ? 2 : v == 3 ? 3 : (n < 4096u ? 2 : 3);
        if (kind == 3 && dec) hipLaunchKernelGGL((k_ari_byte_quad<true>), dim3((n + 63) / 64), dim3(256), 0, s, k);
        else if (kind == 3) hipLaunchKernelGGL((k_ari_byte_quad<false>), dim3((n + 63) / 64), dim3(256), 0, s, k);
        else if (kind == 2 && dec) hipLaunchKernelGGL((k_ari_byte_wave<4, true>), dim3((n + 3) / 4), dim3(256), 0, s, k);
        else if (kind == 2) hipLaunchKernelGGL((k_ari_byte_wave<4, false>), dim3((n + 3) / 4), dim3(256), 0, s, k);
        else hipLaunchKernelGGL(k_ari_byte, dim3((n + 63) / 64), dim3(64), 0, s, k, dec);
        break;
    }
    case RCX_ARI_BINARY_ENCODE: case RCX_ARI_BINARY_DECODE:
        hipLaunchKernelGGL((k_ari_model<0>), dim3((n + 63) / 64), dim3(64), 0, s, k, codec == RCX_ARI_BINARY_DECODE ? 1 : 0, param);
        break;
    case RCX_ARI_PROXY_ENCODE: case RCX_ARI_PROXY_DECODE:
        hipLaunchKernelGGL((k_ari_model<1>), dim3((n + 63) / 64), dim3(64), 0, s, k, codec == RCX_ARI_PROXY_DECODE ? 1 : 0, 0u);
        break;
    case RCX_ARI_APM_ENCODE: case RCX_ARI_APM_DECODE:          // k.scratch = stretch table + gate bins (rcx_api.hip)
        hipLaunchKernelGGL((k_ari_model<2>), dim3((n + 63) / 64), dim3(64), 0, s, k, codec == RCX_ARI_APM_DECODE ? 1 : 0, 0u);
        break;
    default: break;
    }
}
