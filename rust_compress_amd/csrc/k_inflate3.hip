// k_inflate3.hip -- DEFLATE decode, one WAVE per stream: a serial Huffman front end feeding the LZ4 decoder's batched
// executor (reference: Decoder::block/statik/fixed/dynamic/codes, src/flate.rs:195-450; HuffmanTree, :83-146).
//
// Why a third kernel: k_inflate2 spends one LANE per stream, so a wave executes the union of 32-64 lanes' paths through
// one big loop body (rocprof: ~1000 instructions per loop iteration, 29 GiB/s).  Here the wave follows ONE stream:
//   * symbols are decoded by wave-uniform (scalar) code from a 9-bit lookup table in LDS (code <= 9 bits: one LDS read;
//     longer codes: canonical limits, as in k_inflate2), literals go to an LDS literal buffer, matches become
//     {literal run, match length, distance} sequences -- exactly what an LZ4 block is made of;
//   * up to 64 sequences are then emitted by Lz4V5::emit5 (prefix sum of output positions, 16-byte HBM gathers for
//     matches older than the LDS window, redirection of chained matches, exec-narrowing byte stores, coalesced drain):
//     the code that decodes LZ4 at 300 GiB/s; matches longer than 64 bytes and stored blocks take its wave-wide paths.
// The tables of a dynamic block are built by the whole wave (LDS histogram, ballot ranks, per-symbol table fill).
//
// EXACTNESS.  The reference's error statuses and its in_used / flags conventions are intricate (k_inflate2 reproduces
// them one by one).  This kernel only has to be right on streams it ACCEPTS: anything unusual -- every error, an
// over-subscribed or empty code, a symbol without a code, a distance beyond the output, input overrun, a short output
// slot -- ends the block with the internal status RCX_ST_FALLBACK, and launch_inflate re-runs exactly those blocks with
// k_inflate2.  Accepted blocks produce the same bytes, in_used and flags as k_inflate2 (tests compare with the oracle).
#include "rcx_dev.h"

#define RCX_ST_FALLBACK 0x7ff00001           /* internal, never leaves the library */

// One pass of the symbol decoder on the vector ALU (hand-written: hipcc's version of the same loop costs ~17 scalar-unit
// instructions per literal and ~120 per match, and the CU's single scalar unit is what bounds this kernel).  All
// operands are wave-uniform values held in VGPRs.  The pass decodes symbols with short codes and BOOKS them itself:
// literals go to `litv` (literal j of the pass in lane j; the caller stores them at litbuf[litn0 + j]), a match of <= 64
// bytes closes the open literal run as a sequence descriptor {run source, L | M << 8 | dist << 16} at desc[ns], 32
// literals in a row close a run without a match.  It returns when something needs the caller:
//   status 0  limits: staging (a refill with off > lim), literal register / buffer (cnt >= room) or descriptors (ns >= 64) ran out
//          1  a match longer than 64 bytes: len, dist decoded, NOT booked (the wave-wide copy path takes it)
//          2  end of block (consumed)        3  the next bits are no lit/len code (or symbol 286/287): nothing consumed
//          4  a distance beyond the output or 32 KiB (the caller falls back)
//          5  length decoded into len, the next bits are no distance code (or symbol 30/31): nothing of it consumed
// Codes longer than the lookup tables cover are decoded from the canonical limits (INF_CANON).
// Bits: the 64-bit buffer is refilled 8 bytes at a time (only 32 counted; the rest are the same bits the next refill ORs
// in again).  ltab/dtab: base | extra_bits << 16 per length / distance symbol.  Fixed registers v80-v99, s[90:91].
#ifndef RCX_INF_RUN_CALL
__device__ __forceinline__ void rcx_inf_run(uint32_t& lo, uint32_t& hi, uint32_t& bc, uint32_t& off, uint32_t& cnt, uint32_t& litv,
                                            uint32_t& len, uint32_t& dist, uint32_t& status, uint32_t& ns, uint32_t& runL, uint32_t& runsrc,
                                            uint32_t& otot, uint32_t room, uint32_t lim, uint32_t lane, uint32_t litn0,
                                            uint32_t cb, uint32_t lutL, uint32_t lutD, uint32_t ltab, uint32_t dtab, uint32_t descb,
                                            uint32_t tabb, uint32_t symLb, uint32_t symDb)
{
    // registers: v80:81 bit buffer, v82 bits, v83 staged offset, v84 cnt, v85 litv, v86-v89 v97 temporaries, v90 len, v91 dist,
    // v92 status, v93 ns, v94 cnt at the start of the open run (minus what the run held before the pass), v95 run source,
    // v96 output bytes before this pass + matches booked in it (+ cnt = output so far), v79 cnt at which something happens
    // (the run reaches 32 literals or the pass is out of room), v98:99 descriptor
#define INF_REFILL(L, DRY)                                     \
        "v_cmp_gt_u32_e32 vcc, 33, v82\n\t"                    \
        "s_cbranch_vccz " L "\n\t"                             \
        "v_cmp_lt_u32_e32 vcc, %[lim], v83\n\t"                \
        "s_cbranch_vccnz " DRY "\n\t"                          \
        "v_add_u32_e32 v86, %[cb], v83\n\t"                    \
        "ds_read2_b32 v[86:87], v86 offset1:1\n\t"             \
        "v_add_u32_e32 v83, 4, v83\n\t"                        \
        "s_waitcnt lgkmcnt(0)\n\t"                             \
        "v_lshlrev_b64 v[86:87], v82, v[86:87]\n\t"            \
        "v_or_b32_e32 v80, v80, v86\n\t"                       \
        "v_or_b32_e32 v81, v81, v87\n\t"                       \
        "v_add_u32_e32 v82, 32, v82\n\t"                       \
        L ":\n\t"
#define INF_CONSUME                                            \
        "v_lshrrev_b64 v[80:81], v86, v[80:81]\n\t"            \
        "v_sub_u32_e32 v82, v82, v86\n\t"
    /* desc[ns] = {runsrc, runL | len << 8 | dist << 16} by lane 0; ns++, the next run starts at litn0 + cnt; out when ns = 64 */
#define INF_POST(OUT)                                          \
        "v_sub_u32_e32 v97, v84, v94\n\t"                      \
        "v_mov_b32_e32 v98, v95\n\t"                           \
        "v_lshl_or_b32 v99, v90, 8, v97\n\t"                   \
        "v_lshl_or_b32 v99, v91, 16, v99\n\t"                  \
        "v_lshl_add_u32 v97, v93, 3, %[descb]\n\t"             \
        "v_cmp_eq_u32_e32 vcc, 0, %[lane]\n\t"                 \
        "s_and_saveexec_b64 s[90:91], vcc\n\t"                 \
        "ds_write_b64 v97, v[98:99]\n\t"                       \
        "s_mov_b64 exec, s[90:91]\n\t"                         \
        "v_add_u32_e32 v93, 1, v93\n\t"                        \
        "v_mov_b32_e32 v94, v84\n\t"                           \
        "v_add_u32_e32 v95, %[litn0], v84\n\t"                 \
        "v_add_u32_e32 v79, 32, v94\n\t"                       \
        "v_min_u32_e32 v79, v79, %[room]\n\t"                  \
        "v_cmp_lt_u32_e32 vcc, 63, v93\n\t"                    \
        "s_cbranch_vccnz " OUT "\n\t"
    /* canonical decode of a code longer than the lookup table covers (HuffmanTree::decode, flate.rs:129-146): lim[l] / base[l] at
       TAB / TAB+64, symbols in canonical order at SYM; in: v80; out: v87 symbol, v86 code length; FAIL: these bits are no code */
#define INF_CANON(TAB, SYM, L0, TAG, FAIL)                     \
        "v_bfrev_b32_e32 v88, v80\n\t"                         \
        "v_lshrrev_b32_e32 v88, 17, v88\n\t"                   \
        "v_mov_b32_e32 v86, " L0 "\n\t"                        \
        "L_c" TAG "_%=:\n\t"                                   \
        "v_lshl_add_u32 v97, v86, 2, " TAB "\n\t"              \
        "ds_read_b32 v89, v97\n\t"                             \
        "ds_read_b32 v97, v97 offset:64\n\t"                   \
        "s_waitcnt lgkmcnt(0)\n\t"                             \
        "v_cmp_lt_u32_e32 vcc, v88, v89\n\t"                   \
        "s_cbranch_vccnz L_f" TAG "_%=\n\t"                    \
        "v_add_u32_e32 v86, 1, v86\n\t"                        \
        "v_cmp_gt_u32_e32 vcc, 16, v86\n\t"                    \
        "s_cbranch_vccnz L_c" TAG "_%=\n\t"                    \
        "s_branch " FAIL "\n\t"                                \
        "L_f" TAG "_%=:\n\t"                                   \
        "v_sub_u32_e32 v89, 15, v86\n\t"                       \
        "v_lshrrev_b32_e32 v89, v89, v88\n\t"                  \
        "v_add_u32_e32 v89, v89, v97\n\t"                      \
        "v_lshl_add_u32 v89, v89, 1, " SYM "\n\t"              \
        "ds_read_u16 v87, v89\n\t"                             \
        "s_waitcnt lgkmcnt(0)\n\t"
    asm volatile(
        "v_mov_b32_e32 v80, %[lo]\n\t" "v_mov_b32_e32 v81, %[hi]\n\t" "v_mov_b32_e32 v82, %[bc]\n\t" "v_mov_b32_e32 v83, %[off]\n\t"
        "v_mov_b32_e32 v84, %[cnt]\n\t" "v_mov_b32_e32 v85, %[litv]\n\t" "v_mov_b32_e32 v90, 0\n\t" "v_mov_b32_e32 v91, 0\n\t"
        "v_mov_b32_e32 v92, 0\n\t" "v_mov_b32_e32 v93, %[ns]\n\t" "v_sub_u32_e32 v94, %[cnt], %[runL]\n\t" "v_mov_b32_e32 v95, %[runsrc]\n\t"
        "v_mov_b32_e32 v96, %[otot]\n\t"
        "v_add_u32_e32 v79, 32, v94\n\t"
        "v_min_u32_e32 v79, v79, %[room]\n\t"
        "v_cmp_ge_u32_e32 vcc, v84, %[room]\n\t"               /* no room at all */
        "s_cbranch_vccnz L_out_%=\n\t"
        "v_cmp_lt_u32_e32 vcc, 63, v93\n\t"
        "s_cbranch_vccnz L_out_%=\n\t"
        "L_top_%=:\n\t"
        INF_REFILL("L_h1_%=", "L_out_%=")                      /* staged bytes ran out between symbols: status 0 */
        "v_and_b32_e32 v86, 0x1ff, v80\n\t"
        "v_lshl_add_u32 v86, v86, 1, %[lutL]\n\t"
        "ds_read_u16 v88, v86\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_and_b32_e32 v86, 15, v88\n\t"
        "v_cmp_lt_u32_e32 vcc, 0x7fff, v88\n\t"
        "s_cbranch_vccnz L_nonlit_%=\n\t"
        INF_CONSUME
        "v_lshrrev_b32_e32 v87, 4, v88\n\t"
        "L_lit_%=:\n\t"                                        /* a literal: v87, its bits are consumed */
        "v_cmp_eq_u32_e32 vcc, %[lane], v84\n\t"
        "v_cndmask_b32_e32 v85, v85, v87, vcc\n\t"
        "v_add_u32_e32 v84, 1, v84\n\t"
        "v_cmp_ne_u32_e32 vcc, v84, v79\n\t"
        "s_cbranch_vccnz L_top_%=\n\t"
        "v_cmp_ge_u32_e32 vcc, v84, %[room]\n\t"               /* out of room (checked before the 32-literal rule: the caller books it) */
        "s_cbranch_vccnz L_out_%=\n\t"
        "v_mov_b32_e32 v90, 0\n\t"                             /* 32 literals in a row: a run without a match */
        "v_mov_b32_e32 v91, 0\n\t"
        INF_POST("L_out_%=")
        "s_branch L_top_%=\n\t"
        "L_nonlit_%=:\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, v86\n\t"
        "s_cbranch_vccnz L_llong_%=\n\t"
        "v_bfe_u32 v87, v88, 4, 11\n\t"
        "L_nl2_%=:\n\t"                                        /* a symbol >= 256 in v87, code length v86, nothing consumed yet */
        "v_cmp_eq_u32_e32 vcc, 0x100, v87\n\t"
        "s_cbranch_vccnz L_eob_%=\n\t"
        "v_subrev_u32_e32 v87, 0x101, v87\n\t"
        "v_cmp_lt_u32_e32 vcc, 28, v87\n\t"
        "s_cbranch_vccnz L_slow_%=\n\t"
        INF_CONSUME
        "v_lshl_add_u32 v87, v87, 2, %[ltab]\n\t"
        "ds_read_b32 v89, v87\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshrrev_b32_e32 v86, 16, v89\n\t"
        "v_and_b32_e32 v89, 0xffff, v89\n\t"
        "v_bfm_b32 v87, v86, 0\n\t"
        "v_and_b32_e32 v87, v87, v80\n\t"
        "v_add_u32_e32 v90, v89, v87\n\t"
        INF_CONSUME
        INF_REFILL("L_h2_%=", "L_dslow_%=")                    /* ... behind a decoded length: the caller restages and decodes the distance */
        "v_and_b32_e32 v86, 0xff, v80\n\t"
        "v_lshl_add_u32 v86, v86, 1, %[lutD]\n\t"
        "ds_read_u16 v88, v86\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_and_b32_e32 v86, 15, v88\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, v86\n\t"
        "s_cbranch_vccnz L_dlong_%=\n\t"
        "v_bfe_u32 v87, v88, 4, 11\n\t"
        "L_d2_%=:\n\t"                                         /* a distance symbol in v87, code length v86, not consumed yet */
        "v_cmp_lt_u32_e32 vcc, 29, v87\n\t"
        "s_cbranch_vccnz L_dslow_%=\n\t"
        INF_CONSUME
        "v_lshl_add_u32 v87, v87, 2, %[dtab]\n\t"
        "ds_read_b32 v89, v87\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshrrev_b32_e32 v86, 16, v89\n\t"
        "v_and_b32_e32 v89, 0xffff, v89\n\t"
        "v_bfm_b32 v87, v86, 0\n\t"
        "v_and_b32_e32 v87, v87, v80\n\t"
        "v_add_u32_e32 v91, v89, v87\n\t"
        INF_CONSUME
        "v_add_u32_e32 v97, v96, v84\n\t"                      /* output so far */
        "v_cmp_gt_u32_e32 vcc, v91, v97\n\t"                   /* distance beyond it */
        "s_cbranch_vccnz L_bad_%=\n\t"
        "v_cmp_lt_u32_e32 vcc, 0x8000, v91\n\t"
        "s_cbranch_vccnz L_bad_%=\n\t"
        "v_cmp_lt_u32_e32 vcc, 64, v90\n\t"
        "s_cbranch_vccnz L_long_%=\n\t"
        "v_add_u32_e32 v96, v96, v90\n\t"
        INF_POST("L_out_%=")
        "s_branch L_top_%=\n\t"
        "L_llong_%=:\n\t"
        INF_CANON("%[tabb]", "%[symLb]", "10", "l", "L_slow_%=")
        "v_cmp_lt_u32_e32 vcc, 0xff, v87\n\t"
        "s_cbranch_vccnz L_nl2_%=\n\t"
        INF_CONSUME
        "s_branch L_lit_%=\n\t"
        "L_dlong_%=:\n\t"
        INF_CANON("%[tabd]", "%[symDb]", "9", "d", "L_dslow_%=")
        "s_branch L_d2_%=\n\t"
        "L_long_%=:\n\t"
        "v_mov_b32_e32 v92, 1\n\t"
        "s_branch L_out_%=\n\t"
        "L_bad_%=:\n\t"
        "v_mov_b32_e32 v92, 4\n\t"
        "s_branch L_out_%=\n\t"
        "L_eob_%=:\n\t"
        INF_CONSUME
        "v_mov_b32_e32 v92, 2\n\t"
        "s_branch L_out_%=\n\t"
        "L_slow_%=:\n\t"
        "v_mov_b32_e32 v92, 3\n\t"
        "s_branch L_out_%=\n\t"
        "L_dslow_%=:\n\t"
        "v_mov_b32_e32 v92, 5\n\t"
        "L_out_%=:\n\t"
        "v_mov_b32_e32 %[lo], v80\n\t" "v_mov_b32_e32 %[hi], v81\n\t" "v_mov_b32_e32 %[bc], v82\n\t" "v_mov_b32_e32 %[off], v83\n\t"
        "v_sub_u32_e32 %[runL], v84, v94\n\t" "v_add_u32_e32 %[otot], v96, v84\n\t"
        "v_mov_b32_e32 %[cnt], v84\n\t" "v_mov_b32_e32 %[litv], v85\n\t" "v_mov_b32_e32 %[len], v90\n\t" "v_mov_b32_e32 %[dist], v91\n\t"
        "v_mov_b32_e32 %[status], v92\n\t" "v_mov_b32_e32 %[ns], v93\n\t" "v_mov_b32_e32 %[runsrc], v95\n\t"
        : [lo] "+v"(lo), [hi] "+v"(hi), [bc] "+v"(bc), [off] "+v"(off), [cnt] "+v"(cnt), [litv] "+v"(litv),
          [ns] "+v"(ns), [runL] "+v"(runL), [runsrc] "+v"(runsrc), [otot] "+v"(otot),
          [len] "=&v"(len), [dist] "=&v"(dist), [status] "=&v"(status)
        : [room] "v"(room), [lim] "v"(lim), [lane] "v"(lane), [litn0] "v"(litn0), [cb] "v"(cb), [lutL] "v"(lutL), [lutD] "v"(lutD),
          [ltab] "v"(ltab), [dtab] "v"(dtab), [descb] "v"(descb), [tabb] "v"(tabb), [tabd] "v"(tabb + 128u), [symLb] "v"(symLb), [symDb] "v"(symDb)
        : "vcc", "memory", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94",
          "v95", "v96", "v97", "v98", "v99", "s90", "s91");
#undef INF_REFILL
#undef INF_CONSUME
#undef INF_POST
#undef INF_CANON
}
#define RCX_LDSADDR(p) rcx_vgpr((uint32_t)(uintptr_t)(p))     // low half of a generic LDS pointer = the LDS byte address
#define RCX_INF_RUN_CALL(lo, hi, bc, off, cnt, litv, len, dist, st, ns, runL, runsrc, otot, room, lim, lane, litn0, cb, lutL, lutD, ltab, dtab, desc, tab, symL, symD) \
    rcx_inf_run(lo, hi, bc, off, cnt, litv, len, dist, st, ns, runL, runsrc, otot, room, lim, lane, litn0, RCX_LDSADDR(cb), RCX_LDSADDR(lutL),    \
                RCX_LDSADDR(lutD), RCX_LDSADDR(ltab), RCX_LDSADDR(dtab), RCX_LDSADDR(desc), RCX_LDSADDR(tab), RCX_LDSADDR(symL), RCX_LDSADDR(symD))
#endif

template <int CB>
struct Inf3 : Lz4V5<CB, 1536, 1024> {
    // LDS per wave must stay below 10 KB for 16 waves per CU: 1536-byte batch output cap, 1 KiB of history in the window,
    // 512 literal bytes per batch, an 8-bit table for the distance code
    typedef Lz4V4<CB, false, 1536, 1024> B;
    static constexpr int LITCAP = 512;       // literal bytes per batch
    static constexpr int LUTBITS = 9, LUTN = 1 << LUTBITS;     // lit/len table
    static constexpr int DBITS = 8, DLUTN = 1 << DBITS;        // distance (and code-length) table
    // LDS views
    uint16_t* lutL; uint16_t* lutD; uint16_t* symL; uint16_t* symD;
    uint8_t* lens;                            // [0, 320): lit/len + distance code lengths, [320, 352): code-length code
    uint32_t* tab;                            // [0..16) lim L, [16..32) base L, [32..48) lim D, [48..64) base D, [64..80) histogram
    uint8_t* litbuf; uint32_t* desc;
    uint32_t* ltab;                           // [0..29) length symbols, [32..62) distance symbols: base | extra_bits << 16
    // bit reader and batch state (wave uniform)
    uint64_t bb; uint32_t bc, p;
    uint32_t otot; int ns; uint32_t litn, runL, runsrc;

    // Everything below is force-inlined into ONE engine loop (run) in which the big pieces -- staging, table build,
    // batch emit, the wave-wide copy paths -- have a single call site each: a real call would put this object in
    // scratch memory (first version: 947 scratch instructions, 4x slower than k_inflate2).
    enum { P_BLOCK = 0, P_STORED, P_DYNHDR, P_BUILD, P_CLENS, P_SYMBOLS, P_DONE };

    // ---- bits ---------------------------------------------------------------------------------------------------
    __device__ __forceinline__ bool staged(uint32_t bytes) const { return (int32_t)p - this->cbase + (int32_t)bytes <= CB; }
    __device__ __forceinline__ void refill()                         // the caller made sure the dword at p is staged
    {
        if (bc <= 32) {
            const uint32_t w = RCX_U(*(const uint32_t*)(this->cbuf + ((int32_t)p - this->cbase)));
            bb |= (uint64_t)w << bc; bc = RCX_U(bc + 32); p = RCX_U(p + 4);
        }
    }
    __device__ __forceinline__ uint32_t bits(uint32_t k)             // k <= 16; the caller keeps bc >= k
    {
        const uint32_t v = (uint32_t)bb & ((1u << k) - 1u);
        bb >>= k; bc -= k;
        return v;
    }
    __device__ __forceinline__ uint32_t used() const { return p - (bc >> 3); }

    // ---- canonical tables ---------------------------------------------------------------------------------------
    // HuffmanTree::construct (flate.rs:83-120) for nsym code lengths at L: 9-bit lookup table (entry = sym << 4 | len,
    // bit 15 set for everything that is not a literal; 0x8000 = no code of <= 9 bits starts like this), symbols in canonical order, limits/bases for the longer codes.
    // Returns 0, 1 (over-subscribed) or 2 (no code at all).
    __device__ __forceinline__ int build(const uint8_t* L, uint32_t nsym, uint16_t* lut, uint32_t lutbits, uint16_t* symtab, uint32_t* lim, uint32_t* base)
    {
        const uint32_t lutn = 1u << lutbits;
        const unsigned lane = this->lane;
        uint32_t* hist = tab + 64;
        if (lane < 16) hist[lane] = 0;
        for (uint32_t j = lane; j < lutn / 2; j += 64) ((uint32_t*)lut)[j] = 0x80008000u;   // "no short code" (bit 15, length 0)
        rcx_wave_sync();
        for (uint32_t c0 = 0; c0 < nsym; c0 += 64) {
            const uint32_t s = c0 + lane;
            if (s < nsym) atomicAdd(&hist[L[s]], 1u);
        }
        rcx_wave_sync();
        uint32_t code = 0, o = 0, run[16];
        int left = 1;
        bool over = false;
        const bool none = RCX_U(hist[0]) == nsym;
#pragma unroll
        for (int l = 1; l <= 15; l++) {
            const uint32_t c = RCX_U(hist[l]);
            left = left * 2 - (int)c;
            over = over || left < 0;
            run[l] = o;
            if (lane == 0) { lim[l] = (code + c) << (15 - l); base[l] = o - code; }
            code = (code + c) << 1; o += c;
        }
        rcx_wave_sync();
        if (none) return 2;
        if (over) return 1;
        for (uint32_t c0 = 0; c0 < nsym; c0 += 64) {
            const uint32_t s = c0 + lane;
            const uint32_t l = s < nsym ? L[s] : 0u;
            uint32_t pos = 0;
#pragma unroll
            for (int k = 1; k <= 15; k++) {
                const unsigned long long m = __ballot(l == (uint32_t)k);
                pos = l == (uint32_t)k ? run[k] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) : pos;
                run[k] += (uint32_t)__popcll(m);
            }
            if (l) {
                symtab[pos] = (uint16_t)s;
                if (l <= lutbits) {
                    const uint32_t cd = pos - base[l];                          // first[l] + rank
                    const uint32_t r = __brev(cd) >> (32u - l);
                    const uint16_t e = (uint16_t)((s << 4) | l | (s >= 256u ? 0x8000u : 0u));
                    for (uint32_t k = r; k < lutn; k += 1u << l) lut[k] = e;
                }
            }
        }
        rcx_wave_sync();
        return 0;
    }
    // HuffmanTree::decode (flate.rs:129-146); false: these bits are no code (the caller falls back)
    __device__ __forceinline__ bool decode(const uint16_t* lut, uint32_t lutbits, const uint16_t* symtab, const uint32_t* lim, const uint32_t* base, uint32_t& sym)
    {
        const uint32_t e = RCX_U(lut[(uint32_t)bb & ((1u << lutbits) - 1u)]);
        const uint32_t len = e & 15u;
        if (__builtin_expect(len != 0, 1)) { sym = (e >> 4) & 0x7ffu; bb >>= len; bc -= len; return true; }
        const uint32_t rev = __brev((uint32_t)bb) >> 17;
        bool ok = false;
#pragma unroll 1
        for (uint32_t l = lutbits + 1; l <= 15 && !ok; l++) {
            if (rev < RCX_U(lim[l])) {
                sym = RCX_U(symtab[(rev >> (15u - l)) + RCX_U(base[l])]);
                bb >>= l; bc -= l;
                ok = true;
            }
        }
        return ok;
    }

    // ---- sequences ----------------------------------------------------------------------------------------------
    __device__ __forceinline__ void post(uint32_t L, uint32_t M, uint32_t dist)      // close the open literal run with a match (or none)
    {
        if (this->lane == 0) { desc[2 * ns] = runsrc; desc[2 * ns + 1] = L | (M << 8) | (dist << 16); }
        ns = (int)RCX_U(ns + 1);
        runL = 0; runsrc = litn;
    }

    // Decoder::block to BFINAL (flate.rs:195-206) after an optional zlib header (zlib.rs:55-86): the engine loop
    __device__ void run(int zlib, int32_t* st_out, uint32_t* len_out, uint32_t* used_out, uint32_t* flags_out)
    {
        this->lane = rcx_lane();
        const unsigned lane = this->lane;
        this->init_window();
        otot = 0; ns = 0; litn = 0; runL = 0; runsrc = 0; bb = 0; bc = 0; p = 0;
        if (lane < 29) {                                               // EXTRALENS/EXTRABITS (:265-273), closed form
            const uint32_t nn = lane, lb = nn < 8 ? 0u : (nn == 28 ? 0u : (nn - 4u) >> 2);
            ltab[lane] = (nn < 8 ? 3u + nn : (nn == 28 ? 258u : 3u + ((4u + (nn & 3u)) << lb))) | (lb << 16);
        }
        if (lane < 30) {                                               // EXTRADIST/EXTRADBITS (:275-284)
            const uint32_t d = lane, db = d < 4 ? 0u : (d - 2u) >> 1;
            ltab[32 + lane] = (d < 4 ? 1u + d : 1u + ((2u + (d & 1u)) << db)) | (db << 16);
        }
        rcx_wave_sync();
        uint32_t* const limL = tab; uint32_t* const baseL = tab + 16; uint32_t* const limD = tab + 32; uint32_t* const baseD = tab + 48;
        int st = 0;
        uint32_t flags = 0;
        int phase = P_BLOCK;
        // pending work for the single-site handlers at the top of the loop
        bool want_flush = false, want_stage = true, realign = true;
        uint32_t stage_at = 0;
        uint32_t pend_why = 0, pend_L = 0, pend_M = 0, pend_off = 0, pend_src = 0, after_pos = 0;
        bool eof = false, stored_hdr = false;
        uint32_t before = 0, hlit = 0, hdist = 0, ci = 0, bjob = 0, carry_len = 0;
        if (zlib) {
            if (this->n < 2) st = RCX_ST_FALLBACK;
            else {
                const uint32_t cmf = RCX_U(this->in[0]), flg = RCX_U(this->in[1]);
                if ((cmf & 0xf) != 0x8 || (cmf & 0xf0) != 0x70 || (flg & 0x20) || (cmf * 256 + flg) % 31 != 0) st = RCX_ST_FALLBACK;
                stage_at = 2;
            }
        }
        while (!st && (phase != P_DONE || want_flush || pend_why)) {
            // ---- 1. emit the open batch (the one emit5 site)
            if (want_flush) {
                want_flush = false;
                if (runL) post(runL, 0, 0);
                if (ns) {
                    rcx_wave_sync();
                    const uint32_t w0 = (int)lane < ns ? desc[2 * lane] : 0u, w1 = (int)lane < ns ? desc[2 * lane + 1] : 0u;
                    int lo = 0, e = 0;
                    while (lo < ns && !e) e = this->template emit5<true>(ns, lo, w0, w1, litbuf);
                    if (e) { st = RCX_ST_FALLBACK; break; }
                }
                ns = 0; litn = 0; runL = 0; runsrc = 0;
            }
            // ---- 2. a long match or a stored block: the wave-wide paths of the LZ4 decoder (the one after_batch site)
            if (pend_why) {
                typename B::Batch bt; bt.ns = 0; bt.why = (int)pend_why; bt.perr = 0; bt.gL = pend_L; bt.gM = pend_M; bt.goff = pend_off; bt.gsrc = pend_src; bt.gnext = 0;
                int e = 0;
                if (this->after_batch(bt, e)) { st = RCX_ST_FALLBACK; break; }
                if (pend_why == (uint32_t)B::WIDE_) { want_stage = true; realign = true; stage_at = after_pos; }   // stored block: bits resume behind it
                pend_why = 0;
            }
            if (phase == P_DONE) {                                     // the final block was a stored one: only the position counts
                if (want_stage && realign) { p = stage_at; bb = 0; bc = 0; }
                continue;
            }
            // ---- 3. (re)stage compressed bytes (the one stage site); realign: the next bit is bit 0 of byte stage_at
            if (want_stage) {
                want_stage = false;
                if (realign) {
                    this->stage(stage_at);
                    const uint32_t mis = (uint32_t)((int32_t)stage_at - this->cbase) & 3u;
                    p = stage_at - mis; bb = 0; bc = 0;
                    refill();
                    bb >>= 8 * mis; bc -= 8 * mis;
                    realign = false;
                } else this->stage(p);
            }
            if (!staged(16)) { want_stage = true; continue; }          // every step below reads at most 12 bytes
            refill();

            if (phase == P_BLOCK) {
                before = otot;
                eof = bits(1) == 1;                                    // :198
                const uint32_t type = bits(2);                         // :199
                if (type == 0) { phase = P_STORED; stored_hdr = false; want_flush = true; }
                else if (type == 1) {
                    for (uint32_t i = lane; i < 288; i += 64) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                    if (lane < 30) lens[288 + lane] = 5;
                    rcx_wave_sync();
                    hlit = 288; hdist = 30; bjob = 1; phase = P_BUILD;
                } else if (type == 2) phase = P_DYNHDR;
                else st = RCX_ST_FALLBACK;                             // :203
            } else if (phase == P_STORED) {                            // Decoder::statik, flate.rs:237-246 (the batch is flushed)
                const uint32_t drop = bc & 7u;                         // the rest of the current byte
                bb >>= drop; bc -= drop;
                refill();
                const uint32_t len = bits(16);
                refill();
                const uint32_t nlen = bits(16);
                const uint32_t pos = used();
                if (((~nlen) & 0xffffu) != len || pos > this->n || this->n - pos < len) st = RCX_ST_FALLBACK;   // :240
                else {
                    after_pos = pos + len;
                    if (len) { pend_why = B::WIDE_; pend_L = len; pend_M = 0; pend_off = 0; pend_src = pos; otot = RCX_U(otot + len); }
                    else { want_stage = true; realign = true; stage_at = after_pos; }
                    if (otot == before && !eof) flags |= RCX_W_EMPTY_BLOCK_MIDSTREAM;
                    phase = eof ? P_DONE : P_BLOCK;
                }
            } else if (phase == P_DYNHDR) {                            // Decoder::dynamic, flate.rs:397-414
                hlit = bits(5) + 257; hdist = bits(5) + 1;
                const uint32_t hclen = bits(4) + 4;
                if (hlit > 286 || hdist > 30) st = RCX_ST_FALLBACK;    // :401
                else {
                    for (uint32_t j = lane; j < 88; j += 64) ((uint32_t*)lens)[j] = 0;      // 352 bytes
                    rcx_wave_sync();
                    const uint64_t ORD0 = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) |
                                          (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
                    const uint64_t ORD1 = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
                    for (uint32_t i = 0; i < hclen; i++) {             // :412-414, 57 bits at most: staged(16) covers it
                        refill();
                        const uint32_t x = bits(3);
                        const uint32_t ord = (uint32_t)((i < 12 ? ORD0 >> (5 * i) : ORD1 >> (5 * (i - 12))) & 31u);
                        if (lane == 0) lens[320 + ord] = (uint8_t)x;
                    }
                    rcx_wave_sync();
                    bjob = 0; phase = P_BUILD;
                }
            } else if (phase == P_BUILD) {                             // the one build site: job 0 = code-length code, 1 = lit/len + distance
                const uint32_t njobs = bjob == 0 ? 1u : 2u;
#pragma unroll 1
                for (uint32_t j = 0; j < njobs && !st; j++) {
                    const bool isD = bjob == 0 || j == 1;
                    const uint8_t* Lp = bjob == 0 ? lens + 320 : (j == 0 ? lens : lens + hlit);
                    const uint32_t nsym = bjob == 0 ? 19u : (j == 0 ? hlit : hdist);
                    const int r = build(Lp, nsym, isD ? lutD : lutL, isD ? (uint32_t)DBITS : (uint32_t)LUTBITS, isD ? symD : symL, isD ? limD : limL, isD ? baseD : baseL);
                    // no distance code at all is fine (:447-448, a block of literals only: its table stays empty);
                    // over-subscribed codes and an empty lit/len or code-length code go to the exact kernel
                    if (r == 1 || (r == 2 && !(bjob == 1 && j == 1))) st = RCX_ST_FALLBACK;
                }
                ci = 0;
                phase = bjob == 0 ? P_CLENS : P_SYMBOLS;
            } else if (phase == P_CLENS) {                             // :421-442
                const uint32_t ntot = hlit + hdist;
                while (ci < ntot && !st) {
                    if (!staged(16)) { want_stage = true; break; }
                    refill();
                    uint32_t symbol = 0;
                    if (!decode(lutD, DBITS, symD, limD, baseD, symbol)) { st = RCX_ST_FALLBACK; break; }
                    if (symbol < 16) {
                        if (lane == 0) lens[ci] = (uint8_t)symbol;
                        ci++;
                    } else if (symbol == 16) {
                        if (ci == 0) { st = RCX_ST_FALLBACK; break; }  // :428
                        rcx_wave_sync();
                        const uint32_t prev = RCX_U(lens[ci - 1]);
                        const uint32_t rep = bits(2) + 3;
                        if (ci + rep > ntot) { st = RCX_ST_FALLBACK; break; }
                        if (lane < rep) lens[ci + lane] = (uint8_t)prev;
                        ci += rep;
                    } else if (symbol == 17) ci += bits(3) + 3;
                    else if (symbol == 18) ci += bits(7) + 11;
                    else { st = RCX_ST_FALLBACK; break; }              // :439
                    ci = RCX_U(ci);
                }
                if (!st && !want_stage) {
                    if (ci > ntot) st = RCX_ST_FALLBACK;               // :442
                    rcx_wave_sync();
                    bjob = 1; phase = P_BUILD;
                }
            } else {                                                   // P_SYMBOLS: Decoder::codes, flate.rs:262-341
                for (;;) {
                    // The fast path: symbols with short codes are decoded AND booked on the vector ALU by rcx_inf_run
                    // (literal j of a pass in lane j, sequence descriptors written by the pass); the rest comes back as a status.
                    uint32_t fs = 5u, flen = carry_len, fdist = 0;
                    if (carry_len) carry_len = 0;                      // resuming behind a flush / restage with a decoded length
                    else {
                        const uint32_t a1 = (uint32_t)LITCAP - litn;
                        const uint32_t room = RCX_VGPR(a1 < 64u ? a1 : 64u);
                        uint32_t vlo = RCX_VGPR((uint32_t)bb), vhi = RCX_VGPR((uint32_t)(bb >> 32)), vbc = RCX_VGPR(bc);
                        uint32_t voff = RCX_VGPR((uint32_t)((int32_t)p - this->cbase)), vcnt = RCX_VGPR(0), litv = 0, vst = 0, vlen = 0, vdist = 0;
                        uint32_t vns = RCX_VGPR((uint32_t)ns), vrunL = RCX_VGPR(runL), vrunsrc = RCX_VGPR(runsrc), votot = RCX_VGPR(otot);
                        RCX_INF_RUN_CALL(vlo, vhi, vbc, voff, vcnt, litv, vlen, vdist, vst, vns, vrunL, vrunsrc, votot, room,
                                         RCX_VGPR((uint32_t)(CB - 8)), RCX_VGPR(lane), RCX_VGPR(litn), this->cbuf, lutL, lutD, ltab, ltab + 32, desc,
                                         tab, symL, symD);
                        const uint32_t cnt = RCX_U(vcnt);
                        fs = RCX_U(vst); flen = RCX_U(vlen); fdist = RCX_U(vdist);
                        if (cnt) if (lane < cnt) litbuf[litn + lane] = (uint8_t)litv;
                        litn = RCX_U(litn + cnt); ns = (int)RCX_U(vns); runL = RCX_U(vrunL); runsrc = RCX_U(vrunsrc); otot = RCX_U(votot);
                        bb = ((uint64_t)RCX_U(vhi) << 32) | RCX_U(vlo); bc = RCX_U(vbc); p = (uint32_t)(this->cbase + (int32_t)RCX_U(voff));
                    }
                    if (fs == 4u) { st = RCX_ST_FALLBACK; break; }      // :314 distance beyond the output
                    if (fs == 1u) {                                    // a match longer than 64 bytes: flush, then the wave-wide copy
                        otot = RCX_U(otot + flen);
                        want_flush = true;
                        pend_why = B::SOLO_; pend_L = 0; pend_M = flen; pend_off = fdist; pend_src = 0;
                        break;
                    }
                    if (fs == 2u) {                                    // :290 end of block (its bits are consumed: act on it first)
                        if (otot == before && !eof) flags |= RCX_W_EMPTY_BLOCK_MIDSTREAM;       // :474-476 quirk
                        phase = eof ? P_DONE : P_BLOCK;
                        break;
                    }
                    if (runL == (uint32_t)B::LCAP) post(runL, 0, 0);
                    if (fs == 5u && ns >= 64) { carry_len = flen; want_flush = true; break; }   // the decoded length survives the flush
                    if (fs != 5u && (litn >= (uint32_t)LITCAP || ns >= 64)) { want_flush = true; break; }
                    if (fs == 0u) { if (!staged(16)) { want_stage = true; break; } continue; }   // room or staging ran out
                    // fs == 3: a long (or no) lit/len code next; fs == 5: flen is decoded, the distance code is long (or none)
                    if (!staged(16)) { if (fs == 5u) carry_len = flen; want_stage = true; break; }   // the decoded length survives the restage
                    refill();                                          // >= 33 bits: a code (15) + extra (5) and more
                    uint32_t sym = 257;
                    if (fs != 5u && !decode(lutL, LUTBITS, symL, limL, baseL, sym)) { st = RCX_ST_FALLBACK; break; }
                    if (sym < 256) {                                   // :289
                        if (lane == 0) litbuf[litn] = (uint8_t)sym;
                        litn = RCX_U(litn + 1); runL = RCX_U(runL + 1); otot = RCX_U(otot + 1);
                        if (runL == (uint32_t)B::LCAP) post(runL, 0, 0);
                        if (litn >= (uint32_t)LITCAP || ns >= 64) { want_flush = true; break; }
                        continue;
                    }
                    if (sym == 256) {                                  // :290
                        if (otot == before && !eof) flags |= RCX_W_EMPTY_BLOCK_MIDSTREAM;       // :474-476 quirk
                        phase = eof ? P_DONE : P_BLOCK;
                        break;
                    }
                    const uint32_t nn = sym - 257;
                    if (nn >= 29) { st = RCX_ST_FALLBACK; break; }     // :294-297 (errors and the off-by-one)
                    const uint32_t lb = nn < 8 ? 0u : (nn == 28 ? 0u : (nn - 4u) >> 2);         // EXTRALENS/EXTRABITS, :265-273
                    const uint32_t lbase = nn < 8 ? 3u + nn : (nn == 28 ? 258u : 3u + ((4u + (nn & 3u)) << lb));
                    const uint32_t len = fs == 5u ? flen : lbase + bits(lb);
                    refill();                                          // >= 33 bits: a code (15) + extra (13)
                    uint32_t d = 0;
                    if (!decode(lutD, DBITS, symD, limD, baseD, d) || d >= 30) { st = RCX_ST_FALLBACK; break; }
                    const uint32_t db = d < 4 ? 0u : (d - 2u) >> 1;                             // EXTRADIST/EXTRADBITS, :275-284
                    const uint32_t dbase = d < 4 ? 1u + d : 1u + ((2u + (d & 1u)) << db);
                    const uint32_t dist = dbase + bits(db);
                    if (dist > otot || dist > 32768u) { st = RCX_ST_FALLBACK; break; }          // :314
                    otot = RCX_U(otot + len);
                    if (len <= (uint32_t)B::MCAP) {
                        post(runL, len, dist);
                        if (ns >= 64) { want_flush = true; break; }
                    } else {                                           // long match: flush, then the wave-wide in-window copy
                        want_flush = true;
                        pend_why = B::SOLO_; pend_L = 0; pend_M = len; pend_off = dist; pend_src = 0;
                        break;
                    }
                }
            }
        }
        if (!st) {                                                     // the tail of the last batch
            if (runL) post(runL, 0, 0);
            if (ns) {
                rcx_wave_sync();
                const uint32_t w0 = (int)lane < ns ? desc[2 * lane] : 0u, w1 = (int)lane < ns ? desc[2 * lane + 1] : 0u;
                int lo = 0, e = 0;
                while (lo < ns && !e) e = this->template emit5<true>(ns, lo, w0, w1, litbuf);
                if (e) st = RCX_ST_FALLBACK;
            }
        }
        if (!st && used() > this->n) st = RCX_ST_FALLBACK;             // ran into the zero padding: truncated input
        if (!st) this->flush(this->oend, true);
        *st_out = st; *len_out = st ? 0u : this->oend; *used_out = st ? 0u : used(); *flags_out = flags;
    }
};

#define INF3_LDS_EXTRA (2 * 1024 + 2 * 288 + 2 * 32 + 352 + 4 * 80 + (1024 + 64) + 4 * 128)

template <int CB>
__global__ __launch_bounds__(64) void k_inflate3(rcx_kargs a, int zlib)
{
    typedef Inf3<CB> S;
    __shared__ __align__(16) uint8_t s_cbuf[CB + 96];
    __shared__ __align__(16) uint8_t s_wbuf[S::WBUF5 + 16];     // + 16: lds_load16u reads one dword past the last staging slot
    __shared__ __align__(16) uint16_t s_lutL[512];
    __shared__ __align__(16) uint16_t s_lutD[S::DLUTN];
    __shared__ uint16_t s_symL[288];
    __shared__ uint16_t s_symD[32];
    __shared__ __align__(16) uint8_t s_lens[352];
    __shared__ uint32_t s_tab[80];
    __shared__ __align__(16) uint8_t s_lit[S::LITCAP + 64];
    __shared__ __align__(16) uint32_t s_desc[128];
    __shared__ uint32_t s_ltab[64];
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    S s;
    s.in = a.in_base + a.in_off[b];
    const uint64_t n64 = a.in_len[b], cap64 = a.out_cap[b];
    s.n = n64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)n64;
    s.out = a.out_base + a.out_off[b];
    s.cap = cap64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)cap64;
    s.cbuf = s_cbuf; s.wb_ = s_wbuf; s.epos = nullptr; s.ring = nullptr;
    s.lutL = s_lutL; s.lutD = s_lutD; s.symL = s_symL; s.symD = s_symD; s.lens = s_lens; s.tab = s_tab; s.litbuf = s_lit; s.desc = s_desc; s.ltab = s_ltab;
    int32_t st; uint32_t olen, used, flags;
    s.run(zlib, &st, &olen, &used, &flags);
    if ((threadIdx.x & 63u) == 0) {
        a.status[b] = st;
        a.out_len[b] = olen;
        if (a.in_used) a.in_used[b] = used;
        if (a.aux) a.aux[b] = flags;
    }
}

// zlib trailer after the wave-per-stream decode: Adler-32 (computed by k_adler32 into `adler`) against the 4 big-endian
// bytes after the DEFLATE stream (zlib.rs:108-118); a mismatch or a missing trailer goes to the exact kernel.
__global__ void k_zlib_tail3(rcx_kargs a, const uint32_t* adler)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks || a.status[b] != RCX_OK) return;
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint64_t n = a.in_len[b], q = a.in_used[b];
    if (n - q < 4) { a.status[b] = RCX_ST_FALLBACK; return; }
    const uint32_t ck = ((uint32_t)in[q] << 24) | ((uint32_t)in[q + 1] << 16) | ((uint32_t)in[q + 2] << 8) | (uint32_t)in[q + 3];
    if (ck != adler[b]) { a.status[b] = RCX_ST_FALLBACK; return; }
    a.in_used[b] = q + 4;
}
