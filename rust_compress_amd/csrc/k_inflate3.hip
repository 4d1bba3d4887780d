// k_inflate3.hip -- DEFLATE decode, one WAVE per stream: a serial Huffman front end feeding the LZ4 decoder's batched
// executor (reference: Decoder::block/statik/fixed/dynamic/codes, src/flate.rs:195-450; HuffmanTree, :83-146).
//
// Why a third kernel: k_inflate2 spends one LANE per stream, so a wave executes the union of 32-64 lanes' paths through
// one big loop body (rocprof: ~1000 instructions per loop iteration, 29 GiB/s).  Here the wave follows ONE stream:
//   * symbols are decoded a WINDOW at a time (Inf3::pass): lane j of the wave decodes the symbol that WOULD start at bit
//     position j of the next 64 -- one bit fetch and two table reads (9-bit lit/len table whose entries carry a length's
//     base and extra-bit count, 8-bit distance table) for the whole window -- and a scalar walk then hops from symbol to
//     symbol with v_readlane, booking literals and {literal run, match length, distance} sequences in registers (lane-index compare + select):
//     no LDS round trip and ~10-20 scalar instructions per symbol (the first version paid two to four dependent LDS reads
//     and ~45 vector instructions per symbol in a hand-written serial loop).  Codes longer than the tables, distances
//     beyond the output and the like drop to a general one-symbol path (canonical limits, as in k_inflate2);
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

// The walk over one pre-decoded window (see Inf3::pass).  Lane j of pk1 / pk2 describes the symbol that would start at bit j:
//   pk1: bits it takes (a length: including its distance symbol) | isLength << 7 | decodable << 8 | kind << 9 | literal << 16
//   pk2: a length: length << 8 | distance << 16 (the descriptor word less the literal run)
// `decodable` already says that the whole symbol lies inside the window (so the walk cannot run off it), that its codes were in
// the tables and that a match is at most 64 bytes long; `kind` says what else a lane is: 0 the window ends here, 1 end of block,
// 2 the general path must decode this symbol.
// Starting at window position `pos`, books symbols -- literal j of the pass into lane j of litv, descriptor j into lane j of
// (dw0, dw1) -- and returns why it stopped:
//   0 the window is used up (pos = where the next one starts)   2 end of block (consumed)   3 the symbol at pos needs the general path
//   4 a distance beyond the output   5 the literal register / buffer (cnt reached room) or the 64 descriptors ran out
// Hand-written: everything here is wave-uniform, i.e. work for the CU's ONE scalar unit, which all 16 waves share at about one
// instruction per cycle.  hipcc's version of this loop spends ~80 scalar instructions per symbol (copies between the loop's many
// exits) and the kernel ran at 34 ms for BASELINE config 3; this one spends 6 + 3 branches per literal and 17 + 3 per match.
// The wave simulator supplies a portable version through this hook.
#ifndef RCX_INF_WALK
__device__ __forceinline__ uint32_t rcx_inf_walk(uint32_t pk1, uint32_t pk2, uint32_t& pos, uint32_t& cnt, uint32_t& ns, uint32_t& runL,
                                                 uint32_t& otot, uint32_t& runsrc, uint32_t litn, uint32_t room, uint32_t& litv, uint32_t& dw0,
                                                 uint32_t& dw1)
{
    uint32_t code, e, d, a, b, lim, rb;          // rb: cnt at the start of the open literal run; lim: cnt at which something happens
    asm volatile(
        "s_mov_b32 %[code], 0\n\t"
        "s_sub_u32 %[rb], %[cnt], %[runL]\n\t"
        "s_sub_u32 %[otot], %[otot], %[cnt]\n\t"               /* otot - cnt only changes at matches */
        "s_add_u32 %[lim], %[rb], 32\n\t"
        "s_min_u32 %[lim], %[lim], %[room]\n\t"
        "L_top_%=:\n\t"
        "v_readlane_b32 %[e], %[pk1], %[pos]\n\t"
        "s_bitcmp1_b32 %[e], 8\n\t"
        "s_cbranch_scc0 L_stop_%=\n\t"
        "s_bitcmp1_b32 %[e], 7\n\t"
        "s_cbranch_scc1 L_len_%=\n\t"
        /* a literal */
        "s_bfe_u32 %[b], %[e], 0x80010\n\t"
        "s_and_b32 %[a], %[e], 0x7f\n\t"
        "s_mov_b32 m0, %[cnt]\n\t"
        "v_writelane_b32 %[litv], %[b], m0\n\t"
        "s_add_u32 %[cnt], %[cnt], 1\n\t"
        "s_add_u32 %[pos], %[pos], %[a]\n\t"
        "s_cmp_lt_u32 %[cnt], %[lim]\n\t"
        "s_cbranch_scc1 L_top_%=\n\t"
        "s_cmp_lt_u32 %[cnt], %[room]\n\t"                     /* the literal register / buffer is full: leave */
        "s_cbranch_scc0 L_lim_%=\n\t"
        "s_sub_u32 %[b], %[cnt], %[rb]\n\t"                    /* 32 literals in a row: a run without a match */
        "s_mov_b32 m0, %[ns]\n\t"
        "v_writelane_b32 %[dw0], %[runsrc], m0\n\t"
        "v_writelane_b32 %[dw1], %[b], m0\n\t"
        "s_add_u32 %[ns], %[ns], 1\n\t"
        "s_mov_b32 %[rb], %[cnt]\n\t"
        "s_add_u32 %[runsrc], %[litn], %[cnt]\n\t"
        "s_add_u32 %[lim], %[rb], 32\n\t"
        "s_min_u32 %[lim], %[lim], %[room]\n\t"
        "s_cmp_lt_u32 %[ns], 64\n\t"
        "s_cbranch_scc1 L_top_%=\n\t"
        "L_lim_%=:\n\t"
        "s_mov_b32 %[code], 5\n\t"
        "s_branch L_out_%=\n\t"
        /* a match of <= 64 bytes */
        "L_len_%=:\n\t"
        "v_readlane_b32 %[d], %[pk2], %[pos]\n\t"
        "s_lshr_b32 %[b], %[d], 16\n\t"
        "s_add_u32 %[a], %[otot], %[cnt]\n\t"                  /* output so far */
        "s_cmp_gt_u32 %[b], %[a]\n\t"
        "s_cbranch_scc1 L_bad_%=\n\t"
        "s_and_b32 %[a], %[e], 0x7f\n\t"
        "s_add_u32 %[pos], %[pos], %[a]\n\t"
        "s_bfe_u32 %[b], %[d], 0x80008\n\t"
        "s_add_u32 %[otot], %[otot], %[b]\n\t"
        "s_sub_u32 %[b], %[cnt], %[rb]\n\t"                    /* the open run's length */
        "s_or_b32 %[d], %[d], %[b]\n\t"
        "s_mov_b32 m0, %[ns]\n\t"
        "v_writelane_b32 %[dw0], %[runsrc], m0\n\t"
        "v_writelane_b32 %[dw1], %[d], m0\n\t"
        "s_add_u32 %[ns], %[ns], 1\n\t"
        "s_mov_b32 %[rb], %[cnt]\n\t"
        "s_add_u32 %[runsrc], %[litn], %[cnt]\n\t"
        "s_add_u32 %[lim], %[rb], 32\n\t"
        "s_min_u32 %[lim], %[lim], %[room]\n\t"
        "s_cmp_lt_u32 %[ns], 64\n\t"
        "s_cbranch_scc1 L_top_%=\n\t"
        "s_mov_b32 %[code], 5\n\t"
        "s_branch L_out_%=\n\t"
        "L_bad_%=:\n\t"
        "s_mov_b32 %[code], 4\n\t"
        "s_branch L_out_%=\n\t"
        /* not a plain symbol: the window's end, the end of the block, the general path */
        "L_stop_%=:\n\t"
        "s_bfe_u32 %[a], %[e], 0x30009\n\t"
        "s_cmp_eq_u32 %[a], 0\n\t"
        "s_cbranch_scc1 L_out_%=\n\t"
        "s_mov_b32 %[code], 3\n\t"
        "s_cmp_eq_u32 %[a], 1\n\t"
        "s_cbranch_scc0 L_out_%=\n\t"
        "s_and_b32 %[b], %[e], 0x7f\n\t"                       /* end of block */
        "s_add_u32 %[pos], %[pos], %[b]\n\t"
        "s_mov_b32 %[code], 2\n\t"
        "L_out_%=:\n\t"
        "s_sub_u32 %[runL], %[cnt], %[rb]\n\t"
        "s_add_u32 %[otot], %[otot], %[cnt]\n\t"
        : [code] "=&s"(code), [e] "=&s"(e), [d] "=&s"(d), [a] "=&s"(a), [b] "=&s"(b), [lim] "=&s"(lim), [rb] "=&s"(rb),
          [pos] "+s"(pos), [cnt] "+s"(cnt), [ns] "+s"(ns), [runL] "+s"(runL), [otot] "+s"(otot), [runsrc] "+s"(runsrc),
          [litv] "+v"(litv), [dw0] "+v"(dw0), [dw1] "+v"(dw1)
        : [pk1] "v"(pk1), [pk2] "v"(pk2), [litn] "s"(litn), [room] "s"(room)
        : "scc", "m0");
    return code;
}
#define RCX_INF_WALK rcx_inf_walk
#endif

// The segment walks of the speculative pass (Inf3::tile4) as ISA.  A step of a walk -- "how many bits does the symbol at bit q take"
// -- is, for a literal or a length whose codes are in the lookup tables, three aligned dword reads + two v_alignbit (the bits),
// one table read (lit/len: the entry's low four bits are the code's length, a length entry's its code + extra bits), for a length
// a second one at the bits behind it (distance code + its extra bits), a mark in the lane's 128-bit map when q lies inside its own
// segment, and an add.  hipcc compiles Inf3::hop4 and the loop around it into ~100 vector + ~40 scalar instructions a step: three
// nested divergent branches (end of block / long code / length), each with its exec bookkeeping, and the search for codes longer
// than the tables executed on nearly every step because one lane in 64 needs it.  Here every lane computes both the literal's and
// the length's answer and selects (45 vector + 12 scalar instructions a step); a lane that meets anything else -- end of block, a
// code longer than the tables, symbols 286 / 287 / 30 / 31 -- STALLS with q on that symbol, and the caller takes that one step
// with the portable code (which knows all the cases) before it calls again.  Phase timers had the tile builds at half of a
// member's time (-DINF3_PROF: 801 K of 1.6 M cycles).
//   q, e, s: the lane's position, the end of its walk, the start of its own segment (bits, relative to the staged bytes)
//   m0..m3: its map; act: the lanes still walking; cb / ll / ld: LDS byte addresses of the staged bytes and the two tables
// Returns the lanes that stalled.  The wave runs with all lanes on.
#ifndef RCX_NO_INF_WALK_ASM
__device__ __forceinline__ uint64_t rcx_inf_hops(uint32_t& q, uint32_t e, uint32_t s, uint32_t& m0, uint32_t& m1, uint32_t& m2, uint32_t& m3,
                                                 uint64_t act, uint32_t cb, uint32_t ll, uint32_t ld)
{
    uint32_t a, w0, w1, w2, lo, hi, eL, nb, eD, t, u, bit;
    uint64_t stall, tmp;
    asm volatile(
        "s_mov_b64 %[stall], 0\n\t"
        "L_top_%=:\n\t"
        "v_cmp_lt_u32_e32 vcc, %[q], %[e]\n\t"
        "s_and_b64 vcc, vcc, %[act]\n\t"
        "s_cbranch_vccz L_out_%=\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "v_lshrrev_b32_e32 %[a], 5, %[q]\n\t"
        "v_lshl_add_u32 %[a], %[a], 2, %[cb]\n\t"                 // the dword that holds bit q
        "ds_read_b32 %[w0], %[a]\n\t"
        "ds_read_b32 %[w1], %[a] offset:4\n\t"
        "ds_read_b32 %[w2], %[a] offset:8\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_alignbit_b32 %[lo], %[w1], %[w0], %[q]\n\t"            // 32 bits from bit q (the shift is q's low five bits)
        "v_and_b32_e32 %[t], 0x1ff, %[lo]\n\t"
        "v_lshl_add_u32 %[t], %[t], 1, %[ll]\n\t"
        "ds_read_u16 %[eL], %[t]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_alignbit_b32 %[hi], %[w2], %[w1], %[q]\n\t"            // and the 32 behind them
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_and_b32_e32 %[nb], 15, %[eL]\n\t"                      // a literal's code bits / a length's code + extra bits
        "v_alignbit_b32 %[t], %[hi], %[lo], %[nb]\n\t"            // the bits behind a length: its distance code
        "v_and_b32_e32 %[t], 0xff, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 1, %[ld]\n\t"
        "ds_read_u16 %[eD], %[t]\n\t"
        "v_sub_u32_e32 %[bit], %[q], %[s]\n\t"                    // (the mark, while the table read is under way)
        "v_lshrrev_b32_e32 %[a], 5, %[bit]\n\t"                   // map word 0..3, or beyond when q is in front of the lane's segment
        "v_lshlrev_b32_e64 %[bit], %[bit], 1\n\t"
        "v_cmp_gt_u32_e32 vcc, 0x1000, %[eL]\n\t"                 // a literal
        "v_cndmask_b32_e64 %[u], -1, 0, vcc\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_and_b32_e32 %[t], 15, %[eD]\n\t"                       // distance code bits (0: not in the table)
        "v_bfe_u32 %[eD], %[eD], 4, 5\n\t"                        // distance symbol
        "v_add_u32_e32 %[w0], -1, %[t]\n\t"
        "v_sub_u32_e32 %[w1], 29, %[eD]\n\t"
        "v_or_b32_e32 %[w0], %[w0], %[w1]\n\t"                    // negative: no code, or symbol 30 / 31
        "v_add_u32_e32 %[eD], -2, %[eD]\n\t"
        "v_ashrrev_i32_e32 %[eD], 1, %[eD]\n\t"
        "v_max_i32_e32 %[eD], 0, %[eD]\n\t"                       // its extra bits
        "v_add3_u32 %[t], %[nb], %[t], %[eD]\n\t"
        "v_cmp_lt_u32_e32 vcc, 0x7fff, %[eL]\n\t"                 // a length
        "v_cndmask_b32_e32 %[u], %[u], %[w0], vcc\n\t"
        "v_cndmask_b32_e32 %[nb], %[nb], %[t], vcc\n\t"
        "v_cmp_le_i32_e32 vcc, 0, %[u]\n\t"                       // the lanes this path takes
        "s_andn2_b64 %[tmp], exec, vcc\n\t"
        "s_or_b64 %[stall], %[stall], %[tmp]\n\t"
        "s_andn2_b64 %[act], %[act], %[tmp]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[a]\n\t"
        "v_cndmask_b32_e32 %[t], 0, %[bit], vcc\n\t"
        "v_or_b32_e32 %[m0], %[m0], %[t]\n\t"
        "v_cmp_eq_u32_e32 vcc, 1, %[a]\n\t"
        "v_cndmask_b32_e32 %[t], 0, %[bit], vcc\n\t"
        "v_or_b32_e32 %[m1], %[m1], %[t]\n\t"
        "v_cmp_eq_u32_e32 vcc, 2, %[a]\n\t"
        "v_cndmask_b32_e32 %[t], 0, %[bit], vcc\n\t"
        "v_or_b32_e32 %[m2], %[m2], %[t]\n\t"
        "v_cmp_eq_u32_e32 vcc, 3, %[a]\n\t"
        "v_cndmask_b32_e32 %[t], 0, %[bit], vcc\n\t"
        "v_or_b32_e32 %[m3], %[m3], %[t]\n\t"
        "v_add_u32_e32 %[q], %[q], %[nb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch L_top_%=\n\t"
        "L_out_%=:\n\t"
        "s_mov_b64 exec, -1\n\t"
        : [q] "+v"(q), [m0] "+v"(m0), [m1] "+v"(m1), [m2] "+v"(m2), [m3] "+v"(m3), [act] "+s"(act), [stall] "=&s"(stall), [tmp] "=&s"(tmp),
          [a] "=&v"(a), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [lo] "=&v"(lo), [hi] "=&v"(hi), [eL] "=&v"(eL), [nb] "=&v"(nb),
          [eD] "=&v"(eD), [t] "=&v"(t), [u] "=&v"(u), [bit] "=&v"(bit)
        : [e] "v"(e), [s] "v"(s), [cb] "s"(cb), [ll] "s"(ll), [ld] "s"(ld)
        : "vcc", "scc", "memory");
    return stall;
}
#endif

// The same loop for the link's repairs (round 6): the lanes in `act` walk from q as above, but every step first looks the position up in
// the lane's OLD map (o0..o3) -- a marked bit means the walk has met the one the lane made from its guessed start, and the lane leaves
// through `merged` -- and the marks go to a NEW map (t0..t3).  The serial link calls it with one lane in `act`; it took the portable hop4
// (~140 instructions) for every step of a repair: 256 steps a member, a sixth of its time (benchmarks/r6_inflate_tile.sh).
#ifndef RCX_NO_INF_WALK_ASM
__device__ __forceinline__ uint64_t rcx_inf_rewalk(uint32_t& q, uint32_t e, uint32_t s, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3,
                                                   uint32_t& t0, uint32_t& t1, uint32_t& t2, uint32_t& t3, uint64_t act, uint64_t& merged, uint32_t cb, uint32_t ll, uint32_t ld)
{
    uint32_t a, w0, w1, w2, lo, hi, eL, nb, eD, t, u, bit, ow;
    uint64_t stall, tmp;
    asm volatile(
        "s_mov_b64 %[stall], 0\n\t"
        "L_top_%=:\n\t"
        "v_cmp_lt_u32_e32 vcc, %[q], %[e]\n\t"
        "s_and_b64 vcc, vcc, %[act]\n\t"
        "s_cbranch_vccz L_out_%=\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "v_lshrrev_b32_e32 %[a], 5, %[q]\n\t"
        "v_lshl_add_u32 %[a], %[a], 2, %[cb]\n\t"                 // the dword that holds bit q
        "ds_read_b32 %[w0], %[a]\n\t"
        "ds_read_b32 %[w1], %[a] offset:4\n\t"
        "ds_read_b32 %[w2], %[a] offset:8\n\t"
        "v_sub_u32_e32 %[bit], %[q], %[s]\n\t"                    // (the look-up in the old map, while the bits are on their way)
        "v_lshrrev_b32_e32 %[a], 5, %[bit]\n\t"                   // map word 0..3
        "v_lshlrev_b32_e64 %[bit], %[bit], 1\n\t"
        "v_cmp_eq_u32_e32 vcc, 1, %[a]\n\t"
        "v_cndmask_b32_e32 %[ow], %[o0], %[o1], vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, 2, %[a]\n\t"
        "v_cndmask_b32_e32 %[ow], %[ow], %[o2], vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, 3, %[a]\n\t"
        "v_cndmask_b32_e32 %[ow], %[ow], %[o3], vcc\n\t"
        "v_and_b32_e32 %[ow], %[ow], %[bit]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[ow]\n\t"                      // met the old walk
        "s_or_b64 %[merged], %[merged], vcc\n\t"
        "s_andn2_b64 %[act], %[act], vcc\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz L_wait_%=\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_alignbit_b32 %[lo], %[w1], %[w0], %[q]\n\t"            // 32 bits from bit q (the shift is q's low five bits)
        "v_and_b32_e32 %[t], 0x1ff, %[lo]\n\t"
        "v_lshl_add_u32 %[t], %[t], 1, %[ll]\n\t"
        "ds_read_u16 %[eL], %[t]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_alignbit_b32 %[hi], %[w2], %[w1], %[q]\n\t"            // and the 32 behind them
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_and_b32_e32 %[nb], 15, %[eL]\n\t"                      // a literal's code bits / a length's code + extra bits
        "v_alignbit_b32 %[t], %[hi], %[lo], %[nb]\n\t"            // the bits behind a length: its distance code
        "v_and_b32_e32 %[t], 0xff, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 1, %[ld]\n\t"
        "ds_read_u16 %[eD], %[t]\n\t"
        "v_cmp_gt_u32_e32 vcc, 0x1000, %[eL]\n\t"                 // a literal
        "v_cndmask_b32_e64 %[u], -1, 0, vcc\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_and_b32_e32 %[t], 15, %[eD]\n\t"                       // distance code bits (0: not in the table)
        "v_bfe_u32 %[eD], %[eD], 4, 5\n\t"                        // distance symbol
        "v_add_u32_e32 %[w0], -1, %[t]\n\t"
        "v_sub_u32_e32 %[w1], 29, %[eD]\n\t"
        "v_or_b32_e32 %[w0], %[w0], %[w1]\n\t"                    // negative: no code, or symbol 30 / 31
        "v_add_u32_e32 %[eD], -2, %[eD]\n\t"
        "v_ashrrev_i32_e32 %[eD], 1, %[eD]\n\t"
        "v_max_i32_e32 %[eD], 0, %[eD]\n\t"                       // its extra bits
        "v_add3_u32 %[t], %[nb], %[t], %[eD]\n\t"
        "v_cmp_lt_u32_e32 vcc, 0x7fff, %[eL]\n\t"                 // a length
        "v_cndmask_b32_e32 %[u], %[u], %[w0], vcc\n\t"
        "v_cndmask_b32_e32 %[nb], %[nb], %[t], vcc\n\t"
        "v_cmp_le_i32_e32 vcc, 0, %[u]\n\t"                       // the lanes this path takes
        "s_andn2_b64 %[tmp], exec, vcc\n\t"
        "s_or_b64 %[stall], %[stall], %[tmp]\n\t"
        "s_andn2_b64 %[act], %[act], %[tmp]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[a]\n\t"
        "v_cndmask_b32_e32 %[t], 0, %[bit], vcc\n\t"
        "v_or_b32_e32 %[t0], %[t0], %[t]\n\t"
        "v_cmp_eq_u32_e32 vcc, 1, %[a]\n\t"
        "v_cndmask_b32_e32 %[t], 0, %[bit], vcc\n\t"
        "v_or_b32_e32 %[t1], %[t1], %[t]\n\t"
        "v_cmp_eq_u32_e32 vcc, 2, %[a]\n\t"
        "v_cndmask_b32_e32 %[t], 0, %[bit], vcc\n\t"
        "v_or_b32_e32 %[t2], %[t2], %[t]\n\t"
        "v_cmp_eq_u32_e32 vcc, 3, %[a]\n\t"
        "v_cndmask_b32_e32 %[t], 0, %[bit], vcc\n\t"
        "v_or_b32_e32 %[t3], %[t3], %[t]\n\t"
        "v_add_u32_e32 %[q], %[q], %[nb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch L_top_%=\n\t"
        "L_wait_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"                                // (the three reads of a step that met the old walk at once)
        "s_mov_b64 exec, -1\n\t"
        "s_branch L_top_%=\n\t"
        "L_out_%=:\n\t"
        "s_mov_b64 exec, -1\n\t"
        : [q] "+v"(q), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [act] "+s"(act), [merged] "+s"(merged), [stall] "=&s"(stall), [tmp] "=&s"(tmp),
          [a] "=&v"(a), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [lo] "=&v"(lo), [hi] "=&v"(hi), [eL] "=&v"(eL), [nb] "=&v"(nb),
          [eD] "=&v"(eD), [t] "=&v"(t), [u] "=&v"(u), [bit] "=&v"(bit), [ow] "=&v"(ow)
        : [e] "v"(e), [s] "v"(s), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [cb] "s"(cb), [ll] "s"(ll), [ld] "s"(ld)
        : "vcc", "scc", "memory");
    return stall;
}
#endif

#ifndef INF3_LINK_ISA
#define INF3_LINK_ISA 0                    /* 1: the link's repairs through rcx_inf_rewalk.  Measured (benchmarks/r6_inflate_modes.sh, one box): the link's cycles a member
                                              275 K -> 224 K, no step left to the portable hop4 -- and config 3 9.61 -> 9.68 ms: the kernel sits at its 80-VGPR cap with
                                              20-40 bytes of scratch a lane, and the loop's eighteen registers beside the tile's state cost more in spills than the
                                              steps save.  Off. */
#endif
#ifndef INF3_TCAP
#define INF3_TCAP 1024
#endif
#ifndef INF3_H
#define INF3_H 768
#endif
#ifndef INF3_SB
#define INF3_SB 0
#endif
#ifndef INF3_LITCAP
#define INF3_LITCAP 320
#endif
#ifndef INF3_OCC
#define INF3_OCC 6
#endif
#ifndef INF3_OWNER_SCAN
#define INF3_OWNER_SCAN 1                  /* 0: the owner of a symbol rank by binary search over the inclusive counts (A/B) */
#endif
template <int CB, bool SPEC = false, bool ADLER = false, bool MIRROR = false>
struct Inf3 : Lz4V5<CB, INF3_TCAP, INF3_H, false, INF3_SB, ADLER, MIRROR> {
    // The kernel is bound by the latency of its dependent phases, so what it needs is waves: 12 / 16 / 18 / 20 / 24 waves per CU take
    // 18.5 / 14.3 / 13.1 / 12.4 / 11.6 ms for config 3.  24 waves = 6400 bytes of LDS each (handed out in 1280-byte granules) and
    // 80 VGPRs: 1024-byte batch output cap, 768 bytes of history in the window, NO staging of gathered matches (every byte goes
    // straight to its place), 320 literal bytes per batch, an 8-bit table for the distance code, and the code lengths of a block
    // header share the literal buffer (the batch is emitted before a header is read).  History / batch cap splits of the same
    // 1808 bytes (512 + 1280, 640 + 1152, 768 + 1024, 896 + 896) measure within 1 %.
    typedef Lz4V4<CB, false, INF3_TCAP, INF3_H, ADLER, MIRROR> B;
    static constexpr int LITCAP = INF3_LITCAP;       // literal bytes per batch
    static constexpr int LUTBITS = 9, LUTN = 1 << LUTBITS;     // lit/len table
    static constexpr int DBITS = 8, DLUTN = 1 << DBITS;        // distance (and code-length) table
    // LDS views
    uint16_t* lutL; uint16_t* lutD; uint16_t* symL; uint16_t* symD;
    uint8_t* lens;                            // [0, 320): lit/len + distance code lengths, [320, 352): code-length code
    uint32_t* tab;                            // [0..16) lim L, [16..32) base L, [32..48) lim D, [48..64) base D, [64..80) histogram
    uint8_t* litbuf; uint32_t* desc;
    // bit reader and batch state (wave uniform)
    uint64_t bb; uint32_t bc, p;
    uint32_t otot; int ns; uint32_t litn, runL, runsrc;

    // Everything below is force-inlined into ONE engine loop (run) in which the big pieces -- staging, table build,
    // batch emit, the wave-wide copy paths -- have a single call site each: a real call would put this object in
    // scratch memory (first version: 947 scratch instructions, 4x slower than k_inflate2).
    enum { P_BLOCK = 0, P_STORED, P_DYNHDR, P_BUILD, P_CLENS, P_SYMBOLS, P_DONE };

    // The canonical limits and bases of the codes LONGER than the lookup tables (lit/len 10..15 bits, distance 9..15), packed
    // `limit | base << 16` (a limit is <= 2^15, a base lies in (-2^15, 288]) and kept in registers -- they are wave-uniform, i.e.
    // SGPRs.  The segment pass decodes 64 candidate symbols at once, and among 64 there is nearly always one with a long code:
    // the search over LDS-resident limits (two dependent LDS reads per code length, thirteen lengths) ran on every step of every
    // walk and was most of a tile's cost (phase timers, -DINF3_PROF: tile builds 862 K of a member's 1.66 M cycles).
    uint32_t lcL[6] = {0, 0, 0, 0, 0, 0}, lcD[7] = {0, 0, 0, 0, 0, 0, 0};
    // the length of the code `rev` (its 15 bits, left-justified) starts with, K0 <= length < K0 + N, and its place in the symbol
    // table; 0: none (flate.rs:129-146, the lengths beyond the lookup table)
    template <int N, int K0>
    static __device__ __forceinline__ uint32_t long_code(const uint32_t (&lc)[N], uint32_t rev, uint32_t& idx)
    {
        uint32_t l = 0;
        idx = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t lim = lc[i] & 0xffffu, base = (uint32_t)((int32_t)lc[i] >> 16);
            const bool hit = l == 0 && rev < lim;
            l = hit ? (uint32_t)(K0 + i) : l;
            idx = hit ? (rev >> (15 - (K0 + i))) + base : idx;
        }
        return l;
    }

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
    // `fused` (the lit/len table): entries are what Inf3::pass wants in one read --
    //   literal       0x0000 | value << 4 | code bits
    //   length        0x8000 | (base - 3) << 7 | extra bit count << 4 | code bits + extra bits      (EXTRALENS / EXTRABITS, flate.rs:265-273)
    //   end of block  0x1000 | code bits;   no code of <= 9 bits starts like this: 0x2000;   symbols 286 / 287: 0x3000 | code bits
    __device__ __forceinline__ int build(const uint8_t* L, uint32_t nsym, uint16_t* lut, uint32_t lutbits, uint16_t* symtab, uint32_t* lim, uint32_t* base, bool fused, bool nodist30 = false)
    {
        const uint32_t lutn = 1u << lutbits;
        const unsigned lane = this->lane;
        uint32_t* hist = tab + 64;
        if (lane < 16) hist[lane] = 0;
        for (uint32_t j = lane; j < lutn / 2; j += 64) ((uint32_t*)lut)[j] = fused ? 0x20002000u : 0x80008000u;   // "no short code"
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
            if (fused && l > LUTBITS) lcL[l - LUTBITS - 1] = ((code + c) << (15 - l)) | ((o - code) << 16);
            if (nodist30 && l > DBITS) lcD[l - DBITS - 1] = ((code + c) << (15 - l)) | ((o - code) << 16);
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
                if (l <= lutbits && !(nodist30 && s >= 30u)) {
                    const uint32_t cd = pos - base[l];                          // first[l] + rank
                    const uint32_t r = __brev(cd) >> (32u - l);
                    uint32_t e32 = (s << 4) | l | (s >= 256u ? 0x8000u : 0u);
                    if (fused && s >= 256u) {
                        const uint32_t nn = s - 257u, lb = nn < 8u ? 0u : (nn == 28u ? 0u : (nn - 4u) >> 2);
                        const uint32_t lbase = nn < 8u ? 3u + nn : (nn == 28u ? 258u : 3u + ((4u + (nn & 3u)) << lb));
                        e32 = s == 256u ? (0x1000u | l) : s > 285u ? (0x3000u | l) : (0x8000u | ((lbase - 3u) << 7) | (lb << 4) | (l + lb));
                    }
                    const uint16_t e = (uint16_t)e32;
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

    // One lit/len symbol on the general path (any code length).  kind 0: literal `val`; 1: end of block; 2: length `val`
    // (its extra bits consumed).  false: no code, or symbol 286 / 287 (the caller falls back).  The caller keeps >= 33 bits.
    __device__ __forceinline__ bool decode_ll(uint32_t& kind, uint32_t& val)
    {
        const uint32_t e = RCX_U(lutL[(uint32_t)bb & (uint32_t)(LUTN - 1)]);
        if (e & 0x8000u) {
            const uint32_t tot = e & 15u, xb = (e >> 4) & 7u;
            kind = 2; val = ((e >> 7) & 0xffu) + 3u + (((uint32_t)bb >> (tot - xb)) & ((1u << xb) - 1u));
            bb >>= tot; bc -= tot;
            return true;
        }
        const uint32_t sp = e >> 12;
        if (sp == 0u) { kind = 0; val = e >> 4; bb >>= (e & 15u); bc -= (e & 15u); return true; }
        if (sp == 1u) { kind = 1; val = 0; bb >>= (e & 15u); bc -= (e & 15u); return true; }
        if (sp == 3u) return false;
        const uint32_t rev = __brev((uint32_t)bb) >> 17;
        uint32_t sym = 0;
        bool ok = false;
#pragma unroll 1
        for (uint32_t l = LUTBITS + 1; l <= 15 && !ok; l++) {
            if (rev < RCX_U(tab[l])) {
                sym = RCX_U(symL[(rev >> (15u - l)) + RCX_U(tab[16 + l])]);
                bb >>= l; bc -= l;
                ok = true;
            }
        }
        if (!ok) return false;
        if (sym < 256u) { kind = 0; val = sym; return true; }
        if (sym == 256u) { kind = 1; val = 0; return true; }
        const uint32_t nn = sym - 257u;
        if (nn >= 29u) return false;                                   // :294-297 (errors and the off-by-one)
        const uint32_t lb = nn < 8 ? 0u : (nn == 28 ? 0u : (nn - 4u) >> 2);         // EXTRALENS/EXTRABITS, :265-273
        const uint32_t lbase = nn < 8 ? 3u + nn : (nn == 28 ? 258u : 3u + ((4u + (nn & 3u)) << lb));
        kind = 2; val = lbase + bits(lb);
        return true;
    }

    // register-resident twin of post(): descriptor `ns` goes to lane ns of (dw0, dw1)
    __device__ __forceinline__ void post_reg(uint32_t& dw0, uint32_t& dw1, uint32_t L, uint32_t M, uint32_t dist, uint32_t cnt)
    {
        if ((int)this->lane == ns) { dw0 = runsrc; dw1 = L | (M << 8) | (dist << 16); }      // v_cmp + 2 v_cndmask
        ns = (int)RCX_U(ns + 1);
        runL = 0; runsrc = RCX_U(litn + cnt);
    }

    // The symbol pass (Decoder::codes, flate.rs:262-341, for symbols with table-length codes).  Decodes and BOOKS symbols
    // until something needs the caller:
    //   0  limits: the staged bytes, the literal buffer (or 64 literals in this pass) or the 64 descriptors ran out
    //   1  a match longer than 64 bytes: flen, fdist decoded and consumed, NOT booked (the wave-wide copy path takes it)
    //   2  end of block (consumed)
    //   3  the next symbol needs the general path (a code longer than the tables, symbols 286/287/30/31): nothing of it consumed
    //   4  a distance beyond the output or 32 KiB (the caller falls back)
    __device__ __forceinline__ uint32_t pass(uint32_t& flen, uint32_t& fdist)
    {
        const unsigned lane = this->lane;
        uint32_t bp = RCX_U(8u * (uint32_t)((int32_t)p - this->cbase) - bc);       // the next unread bit, relative to cbuf[0]
        const uint32_t a1 = (uint32_t)LITCAP - litn;
        const uint32_t room = RCX_U(a1 < 64u ? a1 : 64u);
        const uint32_t ns0 = RCX_U((uint32_t)ns), litn0 = RCX_U(litn);
        uint32_t ucnt = 0, uns = ns0, urunL = RCX_U(runL), uotot = RCX_U(otot), ursrc = RCX_U(runsrc);   // wave uniform: SGPRs across the windows
        uint32_t litv = 0, dw0 = 0, dw1 = 0, status = 0;
        flen = 0; fdist = 0;
        const uint32_t l63 = 63u - lane;
        if (room && uns < 64u)
            for (;;) {
                if ((bp >> 3) + 16u > (uint32_t)CB) break;                             // restage
                // ---- every lane: the symbol that would start at bit bp + lane.  Flags are computed arithmetically: a compare
                // writes a lane mask to SGPRs through the same port the walk below lives on.
                const uint32_t bl = bp + lane, ba = (bl >> 3) & ~3u;
                const uint32_t wlo = *(const uint32_t*)(this->cbuf + ba), whi = *(const uint32_t*)(this->cbuf + ba + 4);
                const uint32_t sh = (uint32_t)((((uint64_t)whi << 32) | wlo) >> (bl & 31u));
                const uint32_t eL = lutL[sh & (uint32_t)(LUTN - 1)], eD = lutD[sh & (uint32_t)(DLUTN - 1)];
                const uint32_t isLen = eL >> 15;                                       // 0 / 1
                const uint32_t spm = ((eL >> 12) & 7u) & (isLen - 1u);                 // 0 literal, 1 end of block, 2 long code, 3 symbol 286/287
                const uint32_t tot = eL & 15u, xb = (eL >> 4) & 7u;
                const uint32_t lenv = ((eL >> 7) & 0xffu) + 3u + ((sh >> ((tot - xb) & 15u)) & ((1u << xb) - 1u));
                const uint32_t nbD = eD & 15u, dsy = (eD >> 4) & 31u;                  // the table holds no short code for symbols 30 / 31
                const int32_t xs = (int32_t)(dsy >> 1) - 1;
                const uint32_t xbD = (uint32_t)(xs < 0 ? 0 : xs);                      // EXTRADIST / EXTRADBITS (flate.rs:275-284), closed form
                const uint32_t dbase = ((2u + (dsy & 1u)) << xbD) + 1u - (((dsy - 2u) >> 31) << 1);
                const uint32_t distv = dbase + ((sh >> nbD) & ((1u << xbD) - 1u));
                const uint32_t okD = nbD < 1u ? nbD : 1u;
                // a length's distance symbol was decoded by the lane where it starts
                const uint32_t dpk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane + tot) << 2), (int)(distv | ((nbD + xbD) << 16) | (okD << 24)));
                const uint32_t hop = tot + (((dpk >> 16) & 0xffu) & (0u - isLen));
                const uint32_t iw = ((l63 - hop) >> 31) ^ 1u;                          // the whole symbol lies inside the window
                const uint32_t okLit = (((spm + 7u) >> 3) ^ 1u);
                const uint32_t okS = okLit ^ ((okLit ^ ((dpk >> 24) & 1u)) & (0u - isLen));          // its codes were in the tables
                const uint32_t lg = ((64u - lenv) >> 31) & isLen;                      // a match longer than 64 bytes: general path
                const uint32_t dec = iw & okS & (lg ^ 1u);
                const uint32_t kbn = (0x2210u >> (spm << 2)) & 3u;                     // what else a lane is: 1 end of block, 2 general path
                const uint32_t kind = (kbn ^ ((kbn ^ 2u) & (0u - isLen))) & (0u - iw); // 0: the window ends here
                const uint32_t pk1 = hop | (isLen << 7) | (dec << 8) | (kind << 9) | (((eL >> 4) & 0xffu) << 16);
                const uint32_t pk2 = ((lenv & 0xffu) << 8) | (dpk << 16);
                // ---- the walk: wave-uniform, from symbol to symbol
                uint32_t pos = 0;
                const uint32_t why = RCX_INF_WALK(pk1, pk2, pos, ucnt, uns, urunL, uotot, ursrc, litn0, room, litv, dw0, dw1);
                bp += pos;
                if (why) { status = why == 5u ? 0u : why; break; }
            }
        if (ucnt) if (lane < ucnt) litbuf[litn0 + lane] = (uint8_t)litv;
        if (uns > ns0) if (lane >= ns0 && lane < uns) { desc[2 * lane] = dw0; desc[2 * lane + 1] = dw1; }
        litn = litn0 + ucnt; ns = (int)uns; runL = urunL; otot = uotot; runsrc = ursrc;
        // the bit reader resumes at bit bp
        const uint32_t pa = (bp >> 3) & ~3u, drop = bp - 8u * pa;
        p = (uint32_t)(this->cbase + (int32_t)pa); bb = 0; bc = 0;
        refill();
        bb >>= drop; bc -= drop;
        return status;
    }

    // ---- the SPECULATIVE symbol pass (SPEC) ---------------------------------------------------------------------------
    // pass() above decodes a candidate symbol at EVERY bit of a 64-bit window (75 vector instructions) to find the 6-7 real ones
    // and walks them with scalar code: ~44 instructions per symbol.  A Huffman stream resynchronises by itself -- a decoder
    // started at an arbitrary bit is on the true symbol boundaries after ~90 bits (p90 207, measured on zlib level 1/6/9
    // members of text) -- so here the next 4096 staged bits are cut into 64 SEGMENTS of 64 bits and decoded all at once, a LANE
    // per segment, each lane starting PRE4 = 256 bits before its segment and marking the symbol starts it sees inside it in a
    // 64-bit register; lane 0 starts at the true position.  The segments are linked as in k_lz4_decode_v8 (the walk of the
    // segment before mine left it at a bit my map has marked: the walks have met; else the true walk is followed by hand), and
    // what is left is the exact set of symbol starts of the tile.  Then a lane per SYMBOL: 64 symbols at a time are decoded in
    // full, literals go to the literal buffer by prefix sum, matches and full 32-literal runs become descriptors, the limits
    // (literal buffer, 64 descriptors, distance beyond the output) cut the batch where they bite, and the first symbol the fast
    // path does not take (end of block, a code without table entry, a match longer than 64) stops it.  The tile survives the
    // flushes in between (two registers).
    // a segment's map: symbol starts inside my 128 bits of the tile (two registers a lane)
    struct Map { uint64_t lo, hi; };
    static __device__ __forceinline__ void mset(Map& m, uint32_t r) { if (r < 64u) m.lo |= 1ull << r; else m.hi |= 1ull << (r - 64u); }
    static __device__ __forceinline__ bool mtest(const Map& m, uint32_t r) { return ((r < 64u ? m.lo >> r : m.hi >> (r - 64u)) & 1ull) != 0; }
    static __device__ __forceinline__ void mbelow(Map& m, uint32_t r)            // drop the marks below bit r (r <= 128)
    {
        if (r >= 128u) { m.lo = 0; m.hi = 0; }
        else if (r >= 64u) { m.lo = 0; m.hi &= ~((1ull << (r - 64u)) - 1ull); }
        else m.lo &= ~((1ull << r) - 1ull);
    }
    Map tmap = {0, 0};
#ifdef INF3_PROF
    uint64_t pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // cycles: [0] emit [1] wide copies [2] staging [3] block headers + tables [4] symbol passes in all [5] of them tile builds [6] chunks consumed [7] symbols booked
#if defined(INF3_PROF_TILE) || defined(INF3_PROF_LINK)
#define INF3_T(slot, code) do { code; } while (0)
#else
#define INF3_T(slot, code) do { const uint64_t t0__ = __builtin_readcyclecounter(); code; pf[slot] += __builtin_readcyclecounter() - t0__; } while (0)
#endif
#else
#define INF3_T(slot, code) do { code; } while (0)
#endif
    uint32_t tb_ = 0, tns_ = 0; int32_t tcb_ = 0;    // tile: first bit (relative to cbuf[0]), segments (0: none), the staging it was built on
#ifndef INF3_PRE4
#define INF3_PRE4 448
#endif
    static constexpr uint32_t SEGB4 = 128, PRE4 = INF3_PRE4, XEOB = 0xfffffffeu, XGEN = 0xffffffffu;

    __device__ __forceinline__ uint64_t bits64(uint32_t b) const                 // 64 bits from bit b of the staged bytes
    {
        const uint32_t* q = (const uint32_t*)(this->cbuf + ((b >> 3) & ~3u));
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], sh = b & 31u;
        const uint32_t lo = (uint32_t)((((uint64_t)w1 << 32) | w0) >> sh), hi = (uint32_t)((((uint64_t)w2 << 32) | w1) >> sh);
        return (uint64_t)lo | ((uint64_t)hi << 32);
    }
    // One symbol from the bits w, lane by lane.  kind 0: literal `val`; 1: end of block; 2: match of val <= 64 bytes at `dist`;
    // 3: a longer match; 4: no code / symbols 286, 287, 30, 31 (the general path decides).  nb: the bits it takes (a match:
    // with its extra bits and its distance symbol).  Codes longer than the tables are decoded canonically (flate.rs:129-146).
    __device__ __forceinline__ void sym4(uint64_t w, uint32_t& kind, uint32_t& nb, uint32_t& val, uint32_t& dist) const
    {
        const uint32_t w32 = (uint32_t)w;
        const uint32_t eL = lutL[w32 & (uint32_t)(LUTN - 1)];
        kind = 4; nb = 0; val = 0; dist = 0;
        uint32_t tot = 0; bool isLen = false;
        if (eL & 0x8000u) {
            tot = eL & 15u;
            const uint32_t xb = (eL >> 4) & 7u;
            val = ((eL >> 7) & 0xffu) + 3u + ((w32 >> (tot - xb)) & ((1u << xb) - 1u));
            isLen = true;
        } else {
            const uint32_t sp = eL >> 12;
            if (sp == 0u) { kind = 0; val = (eL >> 4) & 0xffu; nb = eL & 15u; }
            else if (sp == 1u) { kind = 1; nb = eL & 15u; }
            else if (sp == 2u) {
                uint32_t ix;
                const uint32_t l = long_code<6, LUTBITS + 1>(lcL, __brev(w32) >> 17, ix);
                const uint32_t sym = l ? symL[ix] : 0x7fffu;
                if (l) {
                    if (sym < 256u) { kind = 0; val = sym; nb = l; }
                    else if (sym == 256u) { kind = 1; nb = l; }
                    else {
                        const uint32_t nn = sym - 257u;
                        if (nn < 29u) {
                            const uint32_t lb = nn < 8 ? 0u : (nn == 28 ? 0u : (nn - 4u) >> 2);
                            const uint32_t lbase = nn < 8 ? 3u + nn : (nn == 28 ? 258u : 3u + ((4u + (nn & 3u)) << lb));
                            val = lbase + ((w32 >> l) & ((1u << lb) - 1u));
                            tot = l + lb; isLen = true;
                        }
                    }
                }
            }
        }
        if (isLen) {
            const uint32_t wd = (uint32_t)(w >> tot);
            const uint32_t eD = lutD[wd & (uint32_t)(DLUTN - 1)];
            uint32_t nbD = eD & 15u, dsy = (eD >> 4) & 31u;
            if (nbD == 0u) {
                uint32_t ix;
                nbD = long_code<7, DBITS + 1>(lcD, __brev(wd) >> 17, ix);
                if (nbD) dsy = symD[ix];
            }
            if (nbD != 0u && dsy < 30u) {
                const uint32_t xbD = dsy < 4u ? 0u : (dsy - 2u) >> 1;                        // EXTRADIST / EXTRADBITS (flate.rs:275-284)
                const uint32_t dbase = dsy < 4u ? 1u + dsy : 1u + ((2u + (dsy & 1u)) << xbD);
                dist = dbase + ((wd >> nbD) & ((1u << xbD) - 1u));
                nb = tot + nbD + xbD;
                kind = val <= (uint32_t)B::MCAP ? 2u : 3u;
            }
        }
    }
    // sym4 for the walk: only what kind of symbol it is (1 end of block, 4 the general path, 0 anything else) and the bits it takes
    __device__ __forceinline__ void hop4(uint64_t w, uint32_t& kind, uint32_t& nb) const
    {
        const uint32_t w32 = (uint32_t)w;
        const uint32_t eL = lutL[w32 & (uint32_t)(LUTN - 1)];
        kind = 0; nb = eL & 15u;
        bool isLen = (eL & 0x8000u) != 0;
        if (!isLen && (eL >> 12)) {
            const uint32_t sp = eL >> 12;
            kind = sp == 1u ? 1u : 4u;
            if (sp == 2u) {                                          // a code longer than the table
                uint32_t ix;
                const uint32_t l = long_code<6, LUTBITS + 1>(lcL, __brev(w32) >> 17, ix);
                const uint32_t sym = l ? symL[ix] : 0x7fffu;
                if (l && sym <= 256u) { kind = sym == 256u ? 1u : 0u; nb = l; }
                else if (l && sym - 257u < 29u) {
                    const uint32_t nn = sym - 257u;
                    nb = l + (nn < 8 ? 0u : (nn == 28 ? 0u : (nn - 4u) >> 2));
                    isLen = true; kind = 0;
                }
            }
        }
        if (isLen) {
            const uint32_t wd = (uint32_t)(w >> nb);
            const uint32_t eD = lutD[wd & (uint32_t)(DLUTN - 1)];
            uint32_t nbD = eD & 15u, dsy = (eD >> 4) & 31u;
            if (nbD == 0u) {
                uint32_t ix;
                nbD = long_code<7, DBITS + 1>(lcD, __brev(wd) >> 17, ix);
                if (nbD) dsy = symD[ix];
            }
            if (nbD != 0u && dsy < 30u) nb += nbD + (dsy < 4u ? 0u : (dsy - 2u) >> 1);
            else kind = 4;
        }
    }
    __device__ __forceinline__ Map lanemap(const Map& v, int k) const
    {
        Map r;
        r.lo = (uint64_t)RCX_U(__builtin_amdgcn_readlane((uint32_t)v.lo, k)) | ((uint64_t)RCX_U(__builtin_amdgcn_readlane((uint32_t)(v.lo >> 32), k)) << 32);
        r.hi = (uint64_t)RCX_U(__builtin_amdgcn_readlane((uint32_t)v.hi, k)) | ((uint64_t)RCX_U(__builtin_amdgcn_readlane((uint32_t)(v.hi >> 32), k)) << 32);
        return r;
    }
    // Build the tile at bit bp: every segment decoded at once, then linked.
    __device__ __forceinline__ void tile4(uint32_t bp, uint32_t nseg)
    {
        const unsigned lane = this->lane;
        tb_ = bp; tcb_ = this->cbase; tns_ = nseg;
        const uint32_t s = bp + SEGB4 * lane, e = s + SEGB4;
        const bool mine = lane < nseg;
        uint32_t q = lane == 0 ? bp : (s >= PRE4 ? s - PRE4 : 0u);
        Map map = {0, 0};
        uint32_t ex = 0;
        bool live = mine;
        for (;;) {
#ifdef INF3_PROF_TILE                      /* (attribution: slots 0 / 1 / 2 = rounds of [ISA walk, portable step], cycles in the ISA walk, cycles in the portable step; 3 = tiles) */
            const uint64_t tt0_ = __builtin_readcyclecounter();
            pf[0] += 1;
#endif
#ifndef RCX_NO_INF_WALK_ASM
            {   // the hand-written loop walks until every lane is through or has stalled on a symbol it leaves to hop4 below
                uint32_t m0 = (uint32_t)map.lo, m1 = (uint32_t)(map.lo >> 32), m2 = (uint32_t)map.hi, m3 = (uint32_t)(map.hi >> 32);
                (void)rcx_inf_hops(q, e, s, m0, m1, m2, m3, __ballot(live), RCX_U((uint32_t)(uintptr_t)this->cbuf), RCX_U((uint32_t)(uintptr_t)lutL),
                                   RCX_U((uint32_t)(uintptr_t)lutD));
                map.lo = (uint64_t)m0 | ((uint64_t)m1 << 32); map.hi = (uint64_t)m2 | ((uint64_t)m3 << 32);
            }
#endif
#ifdef INF3_PROF_TILE
            const uint64_t tt1_ = __builtin_readcyclecounter();
            pf[1] += tt1_ - tt0_;
#endif
            const bool go = live && q < e;
            if (!__ballot(go)) break;
            if (go) {
                uint32_t kind, nb;
                hop4(bits64(q), kind, nb);
                if (q >= s) mset(map, q - s);                        // (an end of block or a symbol for the general path is marked too: the consumer stops on it)
                if (kind == 1u) { ex = XEOB; live = false; }
                else if (kind == 4u) { ex = XGEN; live = false; }
                else q += nb;
            }
#ifdef INF3_PROF_TILE
            pf[2] += __builtin_readcyclecounter() - tt1_;
#endif
        }
#ifdef INF3_PROF_TILE
        pf[3] += 1;
#endif
        if (live) ex = q;
#ifdef INF3_PROF_LINK                      /* (attribution: slots 0 / 1 / 2 / 3 = segments the link had to repair, cycles in the link, hops of the repairs, tiles) */
        const uint64_t tl0_ = __builtin_readcyclecounter();
#endif
        // link (see k_lz4_decode_v8.hip): the usual case lane by lane, the rest in order by scalar code over the lanes' registers.
        // (Round 6 measured it -- benchmarks/r6_inflate_tile.sh "" -DINF3_PROF_LINK=1: 79 of a member's 320 segments are repaired, 3.2 hops
        // each, 290 K of a member's 1.57 M cycles -- and tried the repairs as a VECTOR pass, every failed segment walked again at once from
        // the exit in front of it, repeated until the exits stand: failures come in runs, a run's later segments are walked again from
        // entries that are themselves wrong, and the passes -- 14 a tile, 172 steps -- took three times the serial repairs: 27.6 ms.  Removed.)
        uint32_t lowv = 0; bool clr = false;
        const uint32_t eprev = (uint32_t)__shfl_up((int)ex, 1);
        const bool chk = mine && lane > 0;
        bool ok = false;
        if (chk && eprev >= s && eprev < e) { lowv = eprev - s; ok = mtest(map, lowv); }
        if (!ok) lowv = 0;
        unsigned long long bad = __ballot(chk && !ok);
        int k = 0; uint32_t cin = 0; bool forced = false;
        for (;;) {
            if (!forced) {
                if (!bad) break;
                k = __ffsll(bad) - 1;
                cin = RCX_U(__builtin_amdgcn_readlane(ex, k - 1));
            }
            bad &= ~(1ull << k);
#ifdef INF3_PROF_LINK
            pf[0] += 1;
#endif
            // (the stream has ended, or left for the general path: every segment from here on is beyond it -- all of them at once, where each
            //  took a turn of this loop: the last tile of every block has up to 63 of them)
            if (cin >= XEOB) { if ((int)lane >= k) { clr = true; lowv = 0; } break; }
            const uint32_t sk = bp + SEGB4 * (uint32_t)k, ek = sk + SEGB4;
            const uint32_t exk = RCX_U(__builtin_amdgcn_readlane(ex, k));
            uint32_t X;
            if (cin >= ek) { if ((int)lane == k) { clr = true; lowv = 0; } X = cin; }            // the stream ended (or left for the general path) before this segment
            else {
                const Map mk = lanemap(map, k);
                if (mtest(mk, cin - sk)) { if ((int)lane == k) { lowv = cin - sk; clr = false; } X = exk; }
#if !defined(RCX_NO_INF_WALK_ASM) && INF3_LINK_ISA
                else {                                               // follow the true walk until it meets k's map, leaves k or stops: lane k, hand-written loop
                    Map tm = {0, 0};                                 // (the marks of the steps the portable code takes: every lane holds them)
                    uint32_t q2 = cin, stop = 0, qv = cin, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
                    bool merged = false;
                    for (;;) {
                        uint64_t mg = 0;
                        const uint64_t st = rcx_inf_rewalk(qv, e, s, (uint32_t)map.lo, (uint32_t)(map.lo >> 32), (uint32_t)map.hi, (uint32_t)(map.hi >> 32), t0, t1, t2, t3,
                                                           1ull << k, mg, RCX_U((uint32_t)(uintptr_t)this->cbuf), RCX_U((uint32_t)(uintptr_t)lutL), RCX_U((uint32_t)(uintptr_t)lutD));
                        q2 = RCX_U(__builtin_amdgcn_readlane(qv, k));
                        if (mg) { merged = true; break; }
                        if (!st) break;                              // left the segment
                        uint32_t kind, nb;                           // a symbol the loop leaves to hop4 (end of block, a long code ...)
#ifdef INF3_PROF_LINK
                        pf[2] += 1;
#endif
                        hop4(bits64(q2), kind, nb);
                        kind = RCX_U(kind); nb = RCX_U(nb);
                        mset(tm, q2 - sk);
                        if (kind == 1u) { stop = XEOB; break; }
                        if (kind == 4u) { stop = XGEN; break; }
                        q2 += nb; qv = q2;
                        if (q2 >= ek) break;
                    }
                    if ((int)lane == k) {
                        Map nm = map;
                        if (merged) mbelow(nm, q2 - sk); else { nm.lo = 0; nm.hi = 0; }
                        nm.lo |= tm.lo | (uint64_t)t0 | ((uint64_t)t1 << 32); nm.hi |= tm.hi | (uint64_t)t2 | ((uint64_t)t3 << 32);
                        map = nm; lowv = 0; clr = false;
                    }
                    X = stop ? stop : merged ? exk : q2;
                }
#else
                else {                                               // follow the true walk until it meets k's map, leaves k or stops
                    Map tm = {0, 0};
                    uint32_t q2 = cin, stop = 0;
                    while (q2 < ek && !mtest(mk, q2 - sk)) {
                        uint32_t kind, nb;
#ifdef INF3_PROF_LINK
                        pf[2] += 1;
#endif
                        hop4(bits64(q2), kind, nb);
                        kind = RCX_U(kind); nb = RCX_U(nb);
                        mset(tm, q2 - sk);
                        if (kind == 1u) { stop = XEOB; break; }
                        if (kind == 4u) { stop = XGEN; break; }
                        q2 += nb;
                    }
                    const bool merged = !stop && q2 < ek;
                    Map nm = mk;
                    if (merged) mbelow(nm, q2 - sk); else { nm.lo = 0; nm.hi = 0; }
                    nm.lo |= tm.lo; nm.hi |= tm.hi;
                    if ((int)lane == k) { map = nm; lowv = 0; clr = false; }
                    X = stop ? stop : merged ? exk : q2;
                }
#endif
            }
            forced = X != exk && k + 1 < (int)nseg;
            if (forced) { k++; cin = X; }
        }
        if (clr || !mine) { map.lo = 0; map.hi = 0; }
        mbelow(map, lowv);
        tmap = map;
#ifdef INF3_PROF_LINK
        pf[1] += __builtin_readcyclecounter() - tl0_; pf[3] += 1;
#endif
    }

    __device__ __forceinline__ uint32_t pass4(uint32_t& flen, uint32_t& fdist)
    {
        const unsigned lane = this->lane;
        const uint32_t LIMB = 8u * (uint32_t)CB - 64u;               // a symbol that starts below this bit is staged in full (<= 48 bits)
        flen = 0; fdist = 0;
        for (;;) {
            uint32_t bp = RCX_U(8u * (uint32_t)((int32_t)p - this->cbase) - bc);   // the next unread bit, relative to cbuf[0]
            // ---- the tile: the one left from the last call if the stream is still on it, else a new one
            bool have = tns_ != 0 && tcb_ == this->cbase && bp >= tb_ && bp < tb_ + SEGB4 * tns_;
            if (have) have = mtest(lanemap(tmap, (int)((bp - tb_) / SEGB4)), (bp - tb_) % SEGB4);
            if (!have) {
                tns_ = 0;
                if (bp + 2u * SEGB4 > LIMB) return pass(flen, fdist);               // too few staged bits for a tile: the window pass (it knows when to restage)
                const uint32_t nseg = (LIMB - bp) / SEGB4 < 64u ? (LIMB - bp) / SEGB4 : 64u;
                INF3_T(5, tile4(bp, nseg));
            }
            {                                                        // marks below the stream's position are history
                const int32_t rel = (int32_t)bp - (int32_t)(tb_ + SEGB4 * lane);
                if (rel > 0) mbelow(tmap, (uint32_t)rel);
            }
            // ---- the next 64 symbols, a lane each
            const uint32_t cnt = (uint32_t)__popcll(tmap.lo) + (uint32_t)__popcll(tmap.hi);
            const uint32_t incl = rcx_wave_incl_scan(cnt);
            const uint32_t total = RCX_U(__builtin_amdgcn_readlane(incl, 63));
            if (total == 0u) { tns_ = 0; continue; }                // (cannot happen right after a build: bp itself is marked)
            const uint32_t nv = total < 64u ? total : 64u;
#ifdef INF3_PROF
            pf[6] += 1;
#endif
#if defined(INF3_DUMMY_SALU) || defined(INF3_DUMMY_VALU) || defined(INF3_DUMMY_LDS)
            {   // port experiment (benchmarks/r5_inflate_ports.sh): N extra instructions a chunk of 64 symbols on one port, results untouched
                uint32_t ds_ = 0, dv_ = lane;
#ifdef INF3_DUMMY_SALU
#pragma unroll
                for (int k_ = 0; k_ < INF3_DUMMY_SALU; k_++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ds_) : : "scc");
#endif
#ifdef INF3_DUMMY_VALU
#pragma unroll
                for (int k_ = 0; k_ < INF3_DUMMY_VALU; k_++) asm volatile("v_add_u32_e32 %0, 1, %0" : "+v"(dv_));
#endif
#ifdef INF3_DUMMY_LDS
#pragma unroll
                for (int k_ = 0; k_ < INF3_DUMMY_LDS; k_++) asm volatile("ds_read_b32 %0, %1" : "=v"(dv_) : "v"(0u) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                asm volatile("" : : "s"(ds_), "v"(dv_));
            }
#endif
            uint32_t j = 0;                                          // the lane whose map holds symbol `lane`: the number of lanes with incl <= lane
#if INF3_OWNER_SCAN
            {   // ... = the last lane with symbols whose first one has a rank <= lane: every such lane leaves its number at its first rank
                // (64 bytes: the histogram words of the table builds, free between them), a running maximum fills the ranks between --
                // one LDS round trip and six DPP steps where the binary search over `incl` was six dependent ds_bpermute round trips
                uint8_t* const own = (uint8_t*)(tab + 64);
                if (lane < 16u) ((uint32_t*)own)[lane] = 0u;
                rcx_wave_sync();
                const uint32_t excl = incl - cnt;
                if (cnt && excl < 64u) own[excl] = (uint8_t)lane;
                rcx_wave_sync();
                uint32_t v = own[lane], t;
                t = RCX_DPP0(v, 0x111, 0xf); v = t > v ? t : v;
                t = RCX_DPP0(v, 0x112, 0xf); v = t > v ? t : v;
                t = RCX_DPP0(v, 0x114, 0xf); v = t > v ? t : v;
                t = RCX_DPP0(v, 0x118, 0xf); v = t > v ? t : v;
                t = RCX_DPP0(v, 0x142, 0xa); v = t > v ? t : v;
                t = RCX_DPP0(v, 0x143, 0xc); v = t > v ? t : v;
                j = v;
                rcx_wave_sync();
            }
#else
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((j + (uint32_t)step - 1u) << 2), (int)incl);
                j = v <= lane ? j + (uint32_t)step : j;
            }
#endif
            const uint32_t jj = j < 63u ? j : 63u;
            uint32_t r = lane - (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)(incl - cnt));
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)(uint32_t)tmap.lo);
            const uint32_t m1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)(uint32_t)(tmap.lo >> 32));
            const uint32_t m2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)(uint32_t)tmap.hi);
            const uint32_t m3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)(uint32_t)(tmap.hi >> 32));
            const bool valid = lane < nv;
            uint32_t w = m0, base = 0;                               // the r-th set bit of the 128-bit map
            { const uint32_t c = (uint32_t)__popc(m0); if (r >= c) { r -= c; w = m1; base = 32; const uint32_t c1 = (uint32_t)__popc(m1);
                if (r >= c1) { r -= c1; w = m2; base = 64; const uint32_t c2 = (uint32_t)__popc(m2); if (r >= c2) { r -= c2; w = m3; base = 96; } } } }
            { const uint32_t c = (uint32_t)__popc(w & 0xffffu); if (r >= c) { r -= c; w >>= 16; base += 16; } }
            { const uint32_t c = (uint32_t)__popc(w & 0xffu); if (r >= c) { r -= c; w >>= 8; base += 8; } }
            { const uint32_t c = (uint32_t)__popc(w & 0xfu); if (r >= c) { r -= c; w >>= 4; base += 4; } }
            { const uint32_t c = (uint32_t)__popc(w & 0x3u); if (r >= c) { r -= c; w >>= 2; base += 2; } }
            { const uint32_t c = w & 1u; if (r >= c) { base += 1; } }
            const uint32_t b = valid ? tb_ + SEGB4 * jj + base : tb_;
            uint32_t kind, nb, val, dist;
            sym4(bits64(b), kind, nb, val, dist);
            // ---- what gets booked: symbols before the first one the fast path does not take, within the limits
            const unsigned long long stopm = __ballot(valid && (kind == 1u || kind >= 3u));
            const uint32_t g = stopm ? (uint32_t)__ffsll(stopm) - 1u : nv;
            const bool isL = lane < g && kind == 0u, isM = lane < g && kind == 2u;
            const uint32_t litn0 = RCX_U(litn), ns0 = RCX_U((uint32_t)ns), runL0 = RCX_U(runL), otot0 = RCX_U(otot);
            const uint32_t ob = isL ? 1u : isM ? val : 0u;
            const uint32_t obefore = otot0 + rcx_wave_incl_scan(ob) - ob;
            const unsigned long long below = (1ull << lane) - 1ull;
            const unsigned long long litm = __ballot(isL), matm = __ballot(isM);
            const uint32_t lbefore = (uint32_t)__popcll(litm & below);             // literals in front of this lane
            const unsigned long long mb = matm & below;
            const uint32_t lastM = mb ? 64u - (uint32_t)__clzll(mb) : 0u;          // (index + 1) of the last match in front of this lane
            const uint32_t lbM = lastM ? (uint32_t)__popcll(litm & ((1ull << (lastM - 1u)) - 1ull)) : 0u;
            const uint32_t R = lastM ? lbefore - lbM : lbefore + runL0;          // literals since the last match (a run closes every 32)
            const bool closes = isL && ((R + 1u) & 31u) == 0u;
            const bool emit = isM || closes;
            const uint32_t eidx = ns0 + (uint32_t)__popcll(__ballot(emit) & below);
            const bool over = (isL && litn0 + lbefore >= (uint32_t)LITCAP) || eidx >= 64u;       // (nothing is booked behind the 64th descriptor: an open run would be the 65th)
            const bool badd = isM && dist > obefore;
            const unsigned long long cutm = __ballot(over || badd);
            const uint32_t c = cutm ? (uint32_t)__ffsll(cutm) - 1u : g;           // symbols [0, c) are booked
            if (isL && lane < c) litbuf[litn0 + lbefore] = (uint8_t)val;
            if (emit && lane < c) {
                desc[2 * eidx] = closes ? litn0 + lbefore + 1u - 32u : litn0 + lbefore - (R & 31u);
                desc[2 * eidx + 1] = closes ? 32u : ((R & 31u) | (val << 8) | (dist << 16));
            }
#ifdef INF3_PROF
            pf[7] += c;
#endif
            // the state behind symbol c - 1
            if (c) {
                const int cl = (int)c - 1;
                const uint32_t e1 = emit ? 1u : 0u, l1 = isL ? 1u : 0u;
                litn = RCX_U(__builtin_amdgcn_readlane(litn0 + lbefore + l1, cl));
                ns = (int)RCX_U(__builtin_amdgcn_readlane(eidx + e1, cl));
                otot = RCX_U(__builtin_amdgcn_readlane(obefore + ob, cl));
                runL = RCX_U(__builtin_amdgcn_readlane(emit ? 0u : (R + 1u) & 31u, cl));
                runsrc = litn - runL;
            }
            rcx_wave_sync();
            // ---- where the stream stands now, and why the pass stopped (if it did)
            uint32_t status = 0xffu, nbp;
            if (c < nv) {
                const uint32_t kc = RCX_U(__builtin_amdgcn_readlane(kind, (int)c));
                nbp = RCX_U(__builtin_amdgcn_readlane(b, (int)c));
                if ((cutm >> c) & 1ull) status = RCX_U(__builtin_amdgcn_readlane(badd ? 1u : 0u, (int)c)) ? 4u : 0u;
                else if (kc == 1u) { nbp += RCX_U(__builtin_amdgcn_readlane(nb, (int)c)); status = 2u; }
                else status = 3u;
            } else nbp = RCX_U(__builtin_amdgcn_readlane(b + nb, (int)nv - 1));
            // the bit reader resumes at bit nbp
            const uint32_t pa = (nbp >> 3) & ~3u, drop = nbp - 8u * pa;
            p = (uint32_t)(this->cbase + (int32_t)pa); bb = 0; bc = 0;
            refill();
            bb >>= drop; bc -= drop;
            if (status != 0xffu) return status;
        }
    }

    // Decoder::block to BFINAL (flate.rs:195-206) after an optional zlib header (zlib.rs:55-86): the engine loop
    __device__ void run(int zlib, int32_t* st_out, uint32_t* len_out, uint32_t* used_out, uint32_t* flags_out)
    {
        this->lane = rcx_lane();
        const unsigned lane = this->lane;
        this->init_window();
        otot = 0; ns = 0; litn = 0; runL = 0; runsrc = 0; bb = 0; bc = 0; p = 0;
        uint32_t* const limL = tab; uint32_t* const baseL = tab + 16; uint32_t* const limD = tab + 32; uint32_t* const baseD = tab + 48;
        int st = 0;
        uint32_t flags = 0;
        int phase = P_BLOCK;
        // pending work for the single-site handlers at the top of the loop
        bool want_flush = false, want_stage = true, realign = true;
        uint32_t stage_at = 0;
        uint32_t pend_why = 0, pend_L = 0, pend_M = 0, pend_off = 0, pend_src = 0, after_pos = 0;
        bool eof = false, stored_hdr = false;
        uint32_t before = 0, hlit = 0, hdist = 0, ci = 0, bjob = 0;
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
#ifdef INF3_PROF
                const uint64_t tfl0 = __builtin_readcyclecounter();
#endif
                want_flush = false;
                if (runL) post(runL, 0, 0);
                if (ns) {
                    rcx_wave_sync();
                    const uint32_t w0 = (int)lane < ns ? desc[2 * lane] : 0u, w1 = (int)lane < ns ? desc[2 * lane + 1] : 0u;
                    int lo = 0, e = 0;
#ifdef INF3_CUT_EMIT                                       /* attribution build (wrong output on purpose): the front end alone */
                    lo = ns;
                    this->oend = RCX_U(otot - (pend_why ? pend_L + pend_M : 0u));
                    this->gflush = this->oend & ~15u;
                    this->lbase = (int32_t)RCX_U(this->lbase_for(this->oend));
#endif
                    while (lo < ns && !e) e = this->template emit5<true>(ns, lo, w0, w1, litbuf);
#ifdef RCX_SIM_TRACE
                    if (e && this->lane == 0) { fprintf(stderr, "SPEC%d emit5 -> %d ns %d litn %u otot %u oend %u\n", (int)SPEC, e, ns, litn, otot, this->oend);
                        for (int i = 0; i < ns; i++) fprintf(stderr, "  d%d src %u L %u M %u off %u\n", i, desc[2*i], desc[2*i+1] & 255, (desc[2*i+1] >> 8) & 255, desc[2*i+1] >> 16); }
#endif
                    if (e) { st = RCX_ST_FALLBACK; break; }
                }
                ns = 0; litn = 0; runL = 0; runsrc = 0;
#ifdef INF3_PROF
#if !defined(INF3_PROF_TILE) && !defined(INF3_PROF_LINK)
                pf[0] += __builtin_readcyclecounter() - tfl0;
#endif
#endif
            }
            // ---- 2. a long match or a stored block: the wave-wide paths of the LZ4 decoder (the one after_batch site)
            if (pend_why) {
                typename B::Batch bt; bt.ns = 0; bt.why = (int)pend_why; bt.perr = 0; bt.gL = pend_L; bt.gM = pend_M; bt.goff = pend_off; bt.gsrc = pend_src; bt.gnext = 0;
                int e = 0;
                bool ab_; INF3_T(1, ab_ = this->after_batch(bt, e));
                if (ab_) { st = RCX_ST_FALLBACK; break; }
                if (pend_why == (uint32_t)B::WIDE_) { want_stage = true; realign = true; stage_at = after_pos; }   // stored block: bits resume behind it
                pend_why = 0;
            }
            if (phase == P_DONE) {                                     // the final block was a stored one: only the position counts
                if (want_stage && realign) { p = stage_at; bb = 0; bc = 0; }
                continue;
            }
            // ---- 3. (re)stage compressed bytes (the one stage site); realign: the next bit is bit 0 of byte stage_at
            if (want_stage) {
#ifdef INF3_PROF
                const uint64_t tst0 = __builtin_readcyclecounter();
#endif
                want_stage = false;
                if (realign) {
                    this->stage(stage_at);
                    const uint32_t mis = (uint32_t)((int32_t)stage_at - this->cbase) & 3u;
                    p = stage_at - mis; bb = 0; bc = 0;
                    refill();
                    bb >>= 8 * mis; bc -= 8 * mis;
                    realign = false;
                } else this->stage(p - ((bc + 7u) >> 3));              // from the byte of the next unread bit: pass() reads the bits from the buffer
#ifdef INF3_PROF
#if !defined(INF3_PROF_TILE) && !defined(INF3_PROF_LINK)
                pf[2] += __builtin_readcyclecounter() - tst0;
#endif
#endif
            }
            if (!staged(16)) { want_stage = true; continue; }          // every step below reads at most 12 bytes
            refill();
#ifdef INF3_PROF
            const uint64_t tph0 = __builtin_readcyclecounter();
            const int ph0 = phase;
#endif

            if (phase == P_BLOCK) {
                if (ns || litn || runL) { want_flush = true; continue; }   // `lens` lives in the literal buffer: no open batch across a header
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
                    const int r = build(Lp, nsym, isD ? lutD : lutL, isD ? (uint32_t)DBITS : (uint32_t)LUTBITS, isD ? symD : symL, isD ? limD : limL, isD ? baseD : baseL, !isD, isD && bjob == 1);
                    // no distance code at all is fine (:447-448, a block of literals only: its table stays empty);
                    // over-subscribed codes and an empty lit/len or code-length code go to the exact kernel
                    if (r == 1 || (r == 2 && !(bjob == 1 && j == 1))) st = RCX_ST_FALLBACK;
                }
                ci = 0;
                phase = bjob == 0 ? P_CLENS : P_SYMBOLS;
            } else if (phase == P_CLENS) {                             // :421-442
                const uint32_t ntot = hlit + hdist;
#ifndef INF3_SERIAL_CLENS
                // The code-length symbols of a header, a WINDOW of 64 bits at a time: lane l decodes the symbol that would start at bit l of
                // the window (a code of <= 7 bits and its <= 7 extra bits: one table read), a scalar walk over the lanes' "next" offsets
                // picks the real ones (~13 of 64), a wave scan gives each its place in `lens`, a repeat (16) takes its value from the last
                // symbol before it that set one (17 / 18 write zeros, which `lens` already holds).  One symbol at a time this was ~50
                // instructions a symbol and 7 % of a member's time.  Anything the reference rejects (:428, :439, :442) goes to the exact kernel.
                while (ci < ntot && !st) {
                    const uint32_t bp = RCX_U(8u * (uint32_t)((int32_t)p - this->cbase) - bc);   // the next unread bit, relative to cbuf[0]
                    if (bp + 192u > 8u * (uint32_t)CB) { want_stage = true; break; }            // (lane 63 reads 12 bytes from bit bp + 63 on)
                    const uint64_t w = bits64(bp + lane);
                    const uint32_t e = lutD[(uint32_t)w & (uint32_t)(DLUTN - 1)];
                    const uint32_t cl = e & 15u, sy = (e >> 4) & 0x7ffu;
                    const uint32_t xb = sy == 16u ? 2u : sy == 17u ? 3u : sy == 18u ? 7u : 0u;
                    const uint32_t xv = (uint32_t)(w >> cl) & ((1u << xb) - 1u);
                    const uint32_t nxt = lane + cl + xb;                                        // where the symbol behind this one starts
                    const uint32_t cnt = sy < 16u ? 1u : sy == 18u ? 11u + xv : 3u + xv;       // entries of `lens` it stands for
                    const unsigned long long badm = __ballot(cl == 0u || sy > 18u);            // no code starts like this
                    unsigned long long real = 0;
                    for (uint32_t q = 0; q < 64u;) {                                           // (wave-uniform)
                        real |= 1ull << q;
                        if ((badm >> q) & 1ull) break;
                        q = RCX_U(__builtin_amdgcn_readlane(nxt, (int)q));
                    }
                    const bool isreal = ((real >> lane) & 1ull) != 0;
                    const uint32_t c = isreal ? cnt : 0u;
                    const uint32_t incl = rcx_wave_incl_scan(c);
                    const uint32_t cik = ci + incl - c;                                        // the entries in front of this symbol
                    const bool act = isreal && cik < ntot;                                     // (the reference's loop ends at ntot)
                    const unsigned long long actm = __ballot(act);
                    if ((actm & badm) || __ballot(act && (cik + cnt > ntot || (sy == 16u && cik == 0u)))) { st = RCX_ST_FALLBACK; break; }   // :439, :442 / a repeat past the end, :428
                    const unsigned long long setm = __ballot(act && sy != 16u) & ((1ull << lane) - 1ull);   // the symbols in front that set the running value
                    const uint32_t carry = RCX_U(lens[ci ? ci - 1u : 0u]);
                    const uint32_t sv = (uint32_t)__shfl((int)sy, setm ? 63 - (int)__clzll(setm) : 0);
                    const uint32_t val = setm ? (sv < 16u ? sv : 0u) : carry;
                    if (act && sy < 16u) lens[cik] = (uint8_t)sy;
                    if (act && sy == 16u) {
#pragma unroll
                        for (uint32_t j = 0; j < 6u; j++) if (j < cnt) lens[cik + j] = (uint8_t)val;
                    }
                    rcx_wave_sync();
                    const int lastl = 63 - (int)__clzll(actm);                                 // (lane 0 is real and ci < ntot: actm != 0)
                    ci = RCX_U(ci + (uint32_t)__builtin_amdgcn_readlane(incl, lastl));
                    const uint32_t nbp = bp + RCX_U(__builtin_amdgcn_readlane(nxt, lastl));
                    const uint32_t pa = (nbp >> 3) & ~3u, drop = nbp - 8u * pa;                // the bit reader resumes at bit nbp
                    p = (uint32_t)(this->cbase + (int32_t)pa); bb = 0; bc = 0;
                    refill();
                    bb >>= drop; bc -= drop;
                }
#else
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
#endif
                if (!st && !want_stage) {
                    if (ci > ntot) st = RCX_ST_FALLBACK;               // :442
                    rcx_wave_sync();
                    bjob = 1; phase = P_BUILD;
                }
            } else {                                                   // P_SYMBOLS: Decoder::codes, flate.rs:262-341
                for (;;) {
                    // The fast path: symbols with table-length codes are decoded AND booked by pass(); the rest comes back as a status.
                    uint32_t flen = 0, fdist = 0;
                    const uint32_t fs = SPEC ? pass4(flen, fdist) : pass(flen, fdist);
#ifdef RCX_SIM_TRACE
                    if (fs >= 3u && this->lane == 0) fprintf(stderr, "SPEC%d pass -> %u at bit %u otot %u ns %d litn %u runL %u\n", (int)SPEC, fs, 8u * (uint32_t)((int32_t)p - this->cbase) - bc, otot, ns, litn, runL);
#endif
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
                    if (runL == (uint32_t)B::LCAP) post(runL, 0, 0);     // the pass left (literal register full) on the literal that filled the run
                    if (litn >= (uint32_t)LITCAP || ns >= 64) { want_flush = true; break; }
                    if (fs == 0u) { if (!staged(16)) { want_stage = true; break; } continue; }   // room or staging ran out
                    // fs == 3: one symbol on the general path
                    if (!staged(16)) { want_stage = true; break; }
                    refill();                                          // >= 33 bits: a code (15) + extra (5) and more
                    uint32_t kind = 0, val = 0;
                    if (!decode_ll(kind, val)) { st = RCX_ST_FALLBACK; break; }
                    if (kind == 0u) {                                  // :289
                        if (lane == 0) litbuf[litn] = (uint8_t)val;
                        litn = RCX_U(litn + 1); runL = RCX_U(runL + 1); otot = RCX_U(otot + 1);
                        if (runL == (uint32_t)B::LCAP) post(runL, 0, 0);
                        if (litn >= (uint32_t)LITCAP || ns >= 64) { want_flush = true; break; }
                        continue;
                    }
                    if (kind == 1u) {                                  // :290
                        if (otot == before && !eof) flags |= RCX_W_EMPTY_BLOCK_MIDSTREAM;       // :474-476 quirk
                        phase = eof ? P_DONE : P_BLOCK;
                        break;
                    }
                    const uint32_t len = val;
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
#ifdef INF3_PROF
#ifdef INF3_PROF_CLENS                                         /* (attribution: the code-length decode counted apart, in the wide copies' slot) */
            pf[ph0 == P_SYMBOLS ? 4 : ph0 == P_CLENS ? 1 : 3] += __builtin_readcyclecounter() - tph0;
#else
#if defined(INF3_PROF_TILE) || defined(INF3_PROF_LINK)
            if (ph0 == P_SYMBOLS) pf[4] += __builtin_readcyclecounter() - tph0;
#else
            pf[ph0 == P_SYMBOLS ? 4 : 3] += __builtin_readcyclecounter() - tph0;
#endif
#endif
#endif
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

// ADLER (zlib streams): the Adler-32 of the decoded bytes is summed while they leave the window (Lz4V4::flush and the wave-wide
// copies) and lands in the first 4 * nblocks bytes of the scratch, where k_zlib_tail3 compares it with the stream's trailer.
// MIRROR: the host-memory entry points with a page-locked output buffer (rcx_api.hip): what leaves the window is stored a second time
// in the caller's buffer (Lz4V4::mirror_to, the wave-wide copies); streams the first pass hands back are copied out behind the second.
template <int CB, bool SPEC, bool ADLER, bool MIRROR>
#ifndef INF3_VGPR
#define INF3_VGPR 96
#endif
__global__ __launch_bounds__(64, INF3_OCC) void k_inflate3(rcx_kargs a, int zlib)
{
    typedef Inf3<CB, SPEC, ADLER, MIRROR> S;
    __shared__ __align__(16) uint8_t s_cbuf[CB + 96];
    __shared__ __align__(16) uint8_t s_wbuf[S::WBUF5 + 16];     // + 16: lds_load16u reads one dword past the last staging slot
    __shared__ __align__(16) uint16_t s_lutL[512];
    __shared__ __align__(16) uint16_t s_lutD[S::DLUTN];
    __shared__ uint16_t s_symL[288];
    __shared__ uint16_t s_symD[32];
    __shared__ uint32_t s_tab[80];
    __shared__ __align__(16) uint8_t s_lit[S::LITCAP + 64];   // (also the 352 bytes of code lengths while a block header is read)
    static_assert(S::LITCAP + 64 >= 352, "code lengths share the literal buffer");
    __shared__ __align__(16) uint32_t s_desc[128];
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    if (MIRROR && a.gate) {
        // the stream's compressed bytes may still be on their way in (rcx_api.hip: one launch, the input in ranges on a copy stream; the
        // same gate as k_lz4_decode_v8's).  A stream that gives up leaves RCX_ST_GATE, which neither the trailer check nor the second
        // pass looks at: the host decodes the batch again behind one copy.
        uint32_t r = 0;
#pragma unroll
        for (int i = 0; i < 15; i++) r += b >= a.gate_bnd[i] ? 1u : 0u;
        if (r || a.gate_all) {
            if (threadIdx.x == 0) {
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                uint32_t ok = 1;
                if (b == (r ? a.gate_bnd[r - 1] : 0u)) {
                    while (__hip_atomic_load(a.gate_host + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.gate_seq) {
                        __builtin_amdgcn_s_sleep(60);
                        if (__builtin_amdgcn_s_memrealtime() - t0 > (uint64_t)a.gate_ticks) { ok = 0; break; }
                    }
                    if (ok) __hip_atomic_store(a.gate + r, a.gate_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                while (ok && __hip_atomic_load(a.gate + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.gate_seq) {
                    __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > (uint64_t)a.gate_ticks) ok = 0;
                }
                s_desc[0] = ok;
            }
            __syncthreads();
            const uint32_t ok = s_desc[0];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            __syncthreads();
            if (!ok) { if (threadIdx.x == 0) { a.status[b] = (int32_t)RCX_ST_GATE; a.out_len[b] = 0; } return; }
        }
    }
    S s;
    s.in = a.in_base + a.in_off[b];
    const uint64_t n64 = a.in_len[b], cap64 = a.out_cap[b];
    s.n = n64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)n64;
    s.out = a.out_base + a.out_off[b];
    if (MIRROR) s.out2 = a.out_mirror + a.out_off[b];
    s.cap = cap64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)cap64;
    s.cbuf = s_cbuf; s.wb_ = s_wbuf; s.epos = nullptr; s.ring = nullptr;
    s.lutL = s_lutL; s.lutD = s_lutD; s.symL = s_symL; s.symD = s_symD; s.lens = s_lit; s.tab = s_tab; s.litbuf = s_lit; s.desc = s_desc;
    s.lmap = s_desc;                                             // (the descriptors are in registers while emit5 runs: its scratch)
    if (ADLER) {
        // the sums' two words: the last 8 bytes of the literal buffer's 64 bytes of slack (a literal load reads at most 36 bytes
        // past the LITCAP literals, the code lengths end at byte 352)
        static_assert(S::LITCAP + 36 <= S::LITCAP + 56 && 352 <= S::LITCAP + 56, "the literal buffer's last 8 bytes are free");
        s.adp = (uint32_t*)(s_lit + S::LITCAP + 56);
        if ((threadIdx.x & 63u) == 0) { s.adp[0] = 0; s.adp[1] = 0; }
    }
    int32_t st; uint32_t olen, used, flags;
    s.run(zlib, &st, &olen, &used, &flags);
#ifdef INF3_PROF
    if ((threadIdx.x & 63u) == 0 && a.scratch && a.scratch_bytes >= 128) {       // phase totals -> the last 128 bytes of the scratch
        unsigned long long* q = (unsigned long long*)((uint8_t*)a.scratch + ((a.scratch_bytes - 128) & ~7ull));
        for (int i = 0; i < 8; i++) atomicAdd(&q[i], (unsigned long long)s.pf[i]);
    }
#endif
    if ((threadIdx.x & 63u) == 0) {
        a.status[b] = st;
        a.out_len[b] = olen;
        if (a.in_used) a.in_used[b] = used;
        if (a.aux) a.aux[b] = flags;
        if (ADLER && a.scratch) ((uint32_t*)a.scratch)[b] = st ? 0u : s.ad_result(olen);
    }
}

// zlib trailer after the wave-per-stream decode: Adler-32 (summed by k_inflate3<.., true> on the way out, into `adler`) against the 4 big-endian
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
