// k_lz4_decode_v8.hip -- LZ4 block decode, two waves per block as in v5, with a SEGMENT-PARALLEL parser wave
// (reference: BlockDecoder::decode, src/lz4.rs:67-140; the executor wave is Lz4V5's, unchanged).
//
// Why: with the executor switched off the v5 kernel still takes 0.44 of its 0.67 ms (4096 x 64 KiB of text, A/B variant 21):
// the token walk -- v_readlane + s_bitset + s_add per token, one window of 64 input bytes at a time, ~95 scalar instructions
// per window -- keeps the CU's one scalar port 80 % busy on its own.  A walk is serial, but WHERE a walk starts hardly matters:
// two walks over the same bytes that ever meet stay together, and on real streams they meet after ~33 bytes (p90 85, measured
// on every synthetic distribution).
//
// So the parser wave stages 4 KiB of input in LDS (coalesced 16-byte loads, as v4/v5 do), cuts it into 64 SEGMENTS of 64 bytes
// and walks them all at once, a LANE per segment: each lane starts PRE = 128 bytes BEFORE its segment -- a guess -- and marks
// the token starts it visits inside its segment in a 64-bit register; the lane that holds the true cursor starts there.  Then
// the segments are LINKED in order (scalar code over registers, no memory): the true walk enters segment k where the last
// true segment's walk left; if that byte is marked in k's map the two walks have met, the marks below it are dropped and k's
// exit is the true one; if not (a few per cent) the true walk is followed by hand until it meets the map or leaves the
// segment.  A segment the true walk jumps over (a long literal run) is cleared.  What is left is the exact set of token
// starts, and from there on everything is a lane per TOKEN: the maps are unpacked into a list of positions, 64 list entries
// make a batch, and each lane reads its token's fields from 16 bytes of the staged input (token, <= 12 literals, offset and
// the first length extension fit; the few others, and the last 20 bytes of a block, take a bounds-checked byte path through
// global memory that reproduces the reference's error cases, as collect()'s general path does).
//
// (Two earlier forms of the same idea, measured and dropped: a lane per 256..384-byte segment reading global memory -- 64 lanes,
// 64 cache lines per load instruction: a step of the walk took 3600 cycles with sixteen blocks on a CU, 0.25 ms before the first
// batch -- and linking by "is the entry byte marked" without the head start: 68 of 94 segments needed the hand walk.)
#include "rcx_dev.h"
#include "k_lz4_emit6.hip"

// how long the waves sleep between looks at the ring (x 64 cycles): the parser waits for one of eight slots that free up
// every ~14 K cycles, the executor for a batch it needs at once
#ifndef RCX_V8_PSLEEP
#define RCX_V8_PSLEEP 16
#endif
#ifndef RCX_V8_ESLEEP
#define RCX_V8_ESLEEP 2                     /* (round 6, with emit6: 4 -> 2 and RCX_V8_LOW 4 -> 6: 0.508 -> 0.497 ms, benchmarks/r6_modes.sh) */
#endif
#ifndef RCX_FIELDS_ISA
#define RCX_FIELDS_ISA 1                   /* fields(): the fast path as ISA (round 6) */
#endif
#ifndef RCX_WALK_FORM
#define RCX_WALK_FORM 5                  /* 5 (round 6, the default): walk_steps2 -- the loop as ISA again, cheaper than hipcc's on BOTH ports (22 + 12 a step against
                                            ~33 + ~45): parser alone 0.375 -> 0.326 ms, the launch 0.4755 -> 0.4634; the portable loop takes the block's last 20 bytes.
                                            0: eight bytes + a second read where needed (portable), 1: sixteen bytes, the step as ISA, 2: form 0's whole loop as ISA,
                                            3: form 2 for a block's FIRST chunk, form 0 after it, 4: form 2 whenever the executor has fewer than RCX_WALK_ISA_LOW batches in front of it.
                                            Form 2 shortens the walk's latency (fewer scalar instructions, no divergent branches) and costs 30 more vector instructions a
                                            batch: a loss while the executors fill the vector ALU (0.534 against 0.528 ms), a gain before the block's first batch, when
                                            nothing else runs on the SIMD (3: 0.5224; 4 with LOW 2 / 3 / 4 / 6: 0.5233 / 0.5220 / 0.5229 / 0.5240 -- no better than 3) */
#endif
#ifndef RCX_V8_PRE
#define RCX_V8_PRE 128                     /* the guessing lanes' head start, bytes (the kernel template's default) */
#endif
#ifndef RCX_WALK_STEPS
#define RCX_WALK_STEPS 16                  /* steps of the ISA walk between two looks at the ring (ring_prio): 4 / 8 / 16 / 64: 0.4604 / 0.4587 / 0.4565 / 0.479 ms */
#endif
#ifndef RCX_WALK_ISA_LOW
#define RCX_WALK_ISA_LOW 3
#endif
#ifndef RCX_WALK_PRIO
#define RCX_WALK_PRIO RCX_PARSER_PRIO    /* issue priority while a chunk after the first is staged, walked and linked */
#endif
// RCX_V8_ADAPT: the parser's issue priority follows the ring.  The executors are the launch's critical waves and the parser idles
// 40 % of its life, so the parser runs COLD while the executor has RCX_V8_LOW batches or more in front of it and HOT when the ring
// runs low (the first chunk of a block, a chunk boundary the ring does not cover).
// Measured on the headline (benchmarks/r5_lz4_flagcount.sh, one box, ms): fixed priority 2 (round 4) 0.5486 / 0.5451 | LOW 3 HOT 3 COLD 1
// 0.556 | COLD 0: LOW 3 0.539-0.541, LOW 4 0.532, LOW 6 0.541, LOW 8 0.559 | LOW 4 HOT 2 COLD 0 0.545.
#ifndef RCX_V8_PREFETCH
#define RCX_V8_PREFETCH 0
#endif
#ifndef RCX_V8_ADAPT
#define RCX_V8_ADAPT 1
#endif
#ifndef RCX_V8_LOW
#define RCX_V8_LOW 6                        /* round 6: the executor's batch is shorter with emit6 -- 3 / 4 / 6 / 8: 0.513 / 0.508 / 0.498 / 0.501 ms */
#endif
#ifndef RCX_V8_LOW_WALK
#define RCX_V8_LOW_WALK RCX_V8_LOW       /* the same threshold while a chunk is staged, walked and linked (a long stretch without a batch) */
#endif
#ifndef RCX_V8_HOT
#define RCX_V8_HOT 3
#endif
#ifndef RCX_V8_COLD
#define RCX_V8_COLD 0
#endif
// RCX_AGE_DYN: which executors of a SIMD give way (RCX_AGE_PRIO, k_lz4_decode_v5.hip) follows their PROGRESS instead of their age.  A SIMD
// issues the oldest wave first, the static rule (the older half one level down in rounds and drain) takes 2 % off that, and the blocks of
// a launch still finish 20 % apart (benchmarks/r4_lz4_tail.py: 1.10 M .. 1.34 M ticks, an average block's slot busy 90 % of the launch)
// while the launch ends with the last.  Every RCX_AGE_DYN-th batch an executor publishes the share of its input it has consumed in a
// word per (SIMD, wave slot) of g_lz4_prog, looks at the eight words of its SIMD (read one period ago: nobody waits for the load) and
// gives way when more running executors are behind it than ahead.  Parsers mark their slots, finished executors theirs.
#ifndef RCX_AGE_DYN
#define RCX_AGE_DYN 0
#endif
#ifndef RCX_AGE_LOW
#define RCX_AGE_LOW 0                    /* a nibble per age rank like RCX_AGE_DUTY: of every four batches, how many run their plain stretches (header, scan, loads) one level down */
#endif
#ifndef RCX_AGE_DUTY
#define RCX_AGE_DUTY 0x4200              /* 0: off (RCX_AGE_SPLIT alone).  Else a nibble per age rank (rank 0 = the oldest pair of waves on the SIMD, in bits 3:0): of every
                                            four batches, how many the executor handles at the YOUNG half's priority levels -- 0x4400 is what RCX_AGE_SPLIT 2 does.
                                            Round 6 (benchmarks/r6_age.sh: when each block ends, by rank): with 0x4400 the ranks end at 458 / 486 / 455 / 485 us -- the
                                            boost of rank 2 is too strong, rank 1 has none -- and a CU's last block at 498; 0x4200: 437 / 467 / 472 / 480, the last at
                                            489; 0.4562 -> 0.4505 ms, G-runs 0.714 -> 0.690.  0x4210 / 0x4310 / 0x3210 / 0x4211 within 0.5 % of it, 0x4320 best for
                                            G-runs (0.669) and no gain for a text; lowering the old ranks' executor one more level (RCX_AGE_PRIO 42): 0.459 */
#endif
#ifndef RCX_RUNSPLIT
#define RCX_RUNSPLIT 1                   /* a run longer than SPLIT bytes as pieces that copy side by side (post()); 0: one lane fills it (A/B) */
#endif
#define RCX_PROG_FIN 0xfffffffeu
#define RCX_PROG_PARSER 0xffffffffu
#define RCX_PROG_LIVE 0xffffff00u               /* running executors are below this */
#if RCX_AGE_DYN
__device__ uint32_t g_lz4_prog[65536 * 8];      // [XCC | SE | SH | CU | pipe | SIMD][wave slot]
#else
__device__ uint32_t g_lz4_prog[8];
#endif
// Measured (benchmarks/r5_lz4_flagcount.sh, one box): off 0.5189 / 0.5193 ms | every 8th batch 0.5231 | 4th 0.5254 | 2nd 0.556 | every batch 0.775
// -- 13 vector and 16 scalar instructions more a batch even at a period of 8, and nothing back: one level of issue priority in the copy
// rounds and the drain is too weak a lever to move a block's finishing time (the static rule bought 2 %), and a block that finishes early
// does not leave its slot idle -- the others of its CU speed up (a block alone takes 0.49 ms).  Kept as a switch, off.

#ifndef RCX_X6_MAXRUNS
#define RCX_X6_MAXRUNS 2
#endif
#ifndef RCX_V8_STAT
#define RCX_V8_STAT(slot, v) ((void)0)                       /* the wave simulator counts batches through emit6 / emit5 here */
#endif


#define V8P_T0() uint64_t t0_ = PROF8 ? (uint64_t)__builtin_readcyclecounter() : 0
#define V8P_ADD(slot) do { if (PROF8) { const uint64_t t1_ = (uint64_t)__builtin_readcyclecounter(); pp[slot] += t1_ - t0_; t0_ = t1_; } } while (0)

template <int TC = 1024, int HH = 1024, int SB = 16, bool PROF8 = false, int PRE_ = 128, int SPLIT_ = 32, bool PRED = false, int PRR = 2, int CUT = 0, bool MIRROR = false, bool X6 = false>
struct Lz4V8 : Lz4X6<Lz4V5<1024, TC, HH, PROF8, SB, false, MIRROR>, PROF8> {
    // A match longer than SPLIT bytes is handed over as TWO entries -- (L, SPLIT, offset) and (0, M - SPLIT, offset): the same
    // bytes -- so that the executor's copy rounds, 16 bytes a lane and round, are paced by SPLIT and not by the 64-byte cap
    // (3.8 of its 4.6 rounds per batch of text were the long matches').  0: off.
    static constexpr int SPLIT = SPLIT_;
    static_assert(SPLIT_ == 0 || 2 * SPLIT_ >= 64, "two entries cover the 64-byte cap");
    uint32_t* prog = nullptr; uint32_t wslot = 0;  // RCX_AGE_DYN: this SIMD's eight words, my wave slot
    uint32_t agerank = 0;                         // RCX_AGE_DUTY: the executor's age rank among its SIMD's four
    bool agey = true;                             // RCX_AGE_PRIO (k_lz4_decode_v5.hip): this wave is in the younger half of its SIMD's (true: the plain levels)
    typedef Lz4V5<1024, TC, HH, PROF8, SB, false, MIRROR> P5;
    typedef typename P5::B B;
    static constexpr int SEGB = 64, NSEG = 64, CH = SEGB * NSEG;      // a chunk: 64 segments of 64 bytes, a lane each
    static constexpr int PRE = PRE_;                                  // head start of a guessing lane (staged in front of the chunk)
    static constexpr int CSLACK = 32;                                 // staged beyond the chunk: a token's fields are read 16 bytes at a time
    static constexpr int CBUF8 = PRE + CH + CSLACK;
    static constexpr int LWIN = 8;                                    // segments unpacked into the list at a time (512 bytes of input)
    static constexpr int LISTN = LWIN * 22 + 64;                      // a token is >= 3 bytes (the last one apart): <= 22 per segment
    static_assert((PRE % 16) == 0 && (CBUF8 % 16) == 0, "staging is 16 bytes a lane");
    int16_t* list;                                                    // LDS: LISTN token positions, relative to the chunk
    uint32_t po = 0, prlo = 0;                                        // parser's view of the output position / window floor (PRED)
    // The ring between the two waves: EIGHT batches deep -- between the last batch of a chunk and the first of the next the parser
    // stages, walks and links (~55K cycles alone; the executor takes ~10K a batch) -- and a batch entry is ONE word,
    // L | M << 8 | offset << 16: where the literals lie follows from the token's position, and that from the position of the
    // batch's first token (hdr[7]) and the sizes of the tokens in front (3 + L bytes, + 1 from L = 15 on, + 1 from M = 19 on:
    // up to the per-lane caps each extension is a single byte), a wave scan on the executor's side.
    static constexpr int NSLOT8 = 8;
    struct Slot8 { uint32_t hdr[8]; uint32_t d[64]; };
    struct Ring8 { Slot8 slot[NSLOT8]; volatile uint32_t tail, abort_, head, pad; };      // (tail | abort_: one aligned 64-bit read for the parser's look at the ring)
    RCX_LDS_AS Ring8* ring8;
    uint64_t pp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};          // PROF8 (A/B builds): cycles per parser phase, counts

    // byte q of the input (q < n): from the staged bytes when it is among them (two loads kept apart: a select between the
    // two POINTERS would be a flat load)
    __device__ __forceinline__ uint32_t gb(uint32_t q) const
    {
        const int32_t i = (int32_t)q - this->cbase;
        uint32_t v;
        if (i >= 0 && i < CBUF8) v = ((const RCX_LDS_AS uint8_t*)this->cbuf)[i]; else v = ((const RCX_GLOBAL_AS uint8_t*)this->in)[q];
        return v;
    }
    // four bytes at q (q + 4 <= n), same sources
    __device__ __forceinline__ uint32_t gd(uint32_t q) const
    {
        const int32_t i = (int32_t)q - this->cbase;
        uint32_t v;
        if (i >= 0 && i + 8 <= CBUF8) v = B::lds_load4u(this->cbuf, i); else v = *(const rcx_u32_u*)(this->in + q);
        return v;
    }
    // A run of 255s in a length extension, four bytes a step (a 64 KiB literal run has 257 extension bytes): advances q over
    // whole words of 0xff and returns how many bytes that were.
    __device__ __forceinline__ uint32_t skip_ff(uint32_t& q) const
    {
        const uint32_t n = this->n;
        uint32_t k = 0;
        while (q + 4u <= n && q + 4u > q && gd(q) == 0xffffffffu) { q += 4; k += 4; }
        return k;
    }
    // Where the token after the one at p starts (n: there is none).  Bounds-checked; never faults on garbage.  The slow path of
    // next_tok_c (a guessing lane walks garbage: on incompressible data one token in sixteen has a length extension).
    __device__ uint32_t next_tok(uint32_t p) const
    {
        const uint32_t n = this->n;
        uint32_t q = p + 1;
        const uint32_t t = gb(p);
        uint32_t L = t >> 4;
        if (L == 15) {
            L += 255u * skip_ff(q);
            for (;;) { if (q >= n) return n; const uint32_t x = gb(q++); L += x; if (x != 255) break; }
        }
        if (L >= n - q) return n;                                    // the literals reach the end: the last token
        q += L;
        if (n - q < 2) return n;
        q += 2;
        if ((t & 15u) == 15u) {
            skip_ff(q);
            for (;;) { if (q >= n) return n; const uint32_t x = gb(q++); if (x != 255) break; }
        }
        return q;
    }
#if !defined(RCX_NO_WALK_ASM)
    // The walk's LOOP as ISA (RCX_WALK_FORM 2): up to `steps` steps of every walking lane (mine, p < e) -- mark the token start if it lies
    // in the lane's own segment, read eight bytes at p in one LDS round trip, form the next token's position as next_tok_c does, and
    // read the match-length extension byte a second time only where it lies beyond the eight (a branch the wave takes when a lane
    // needs it).  hipcc's loop for the portable form executes ~45 scalar, ~14 compare, ~10 branch and ~35 vector instructions a
    // step (every `if` of the step an s_and_saveexec / s_cbranch pair); this is 8 + 10 + 4 and 28 -- at 324 steps a block the walk
    // was a third of the kernel's scalar-port instructions.  Leaves with `slow` = the lanes whose token it does not decide (a run of
    // 255s, bytes that are not staged, the block's last 20 bytes: marked, NOT advanced -- the caller takes next_tok for them), or
    // with steps = ~0 when no lane walks any more.
    __device__ __forceinline__ uint64_t walk_steps(uint32_t& p, uint32_t e, uint32_t sg, uint32_t& mlo, uint32_t& mhi, uint64_t mine, uint32_t& steps) const
    {
        const uint32_t adj = RCX_U((uint32_t)(uintptr_t)this->cbuf - (uint32_t)this->cbase);     // LDS byte address of input byte q = q + adj
        const uint32_t cend = RCX_U((uint32_t)(uintptr_t)this->cbuf + (uint32_t)CBUF8);
        const uint32_t n = this->n, n20 = n - 20u;
        uint32_t ad, a4, sh, d0, d1, d2, w0, w1, L, t1, hop, tm, q, x, rel, bit, t2;
        uint64_t slow = 0, sA, sB, sS;
        asm volatile(
            "L_top_%=:\n\t"
            "v_cmp_lt_u32_e32 vcc, %[p], %[e]\n\t"
            "s_and_b64 vcc, vcc, %[mine]\n\t"
            "s_cbranch_vccz L_none_%=\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "v_add_u32_e32 %[ad], %[adj], %[p]\n\t"
            "v_and_b32_e32 %[a4], -4, %[ad]\n\t"
            "ds_read_b32 %[d0], %[a4]\n\t"
            "ds_read_b32 %[d1], %[a4] offset:4\n\t"
            "ds_read_b32 %[d2], %[a4] offset:8\n\t"
            "v_sub_u32_e32 %[rel], %[p], %[sg]\n\t"                     // the mark, while the bytes are on their way
            "v_and_b32_e32 %[sh], 3, %[ad]\n\t"
            "v_cmp_gt_u32_e32 vcc, 64, %[rel]\n\t"                     // (two steps in three are head start: no lane in its own segment yet)
            "s_cbranch_vccz L_nomark_%=\n\t"
            "v_bfm_b32 %[bit], 1, %[rel]\n\t"
            "v_cmp_gt_u32_e32 vcc, 32, %[rel]\n\t"
            "v_add_u32_e32 %[t2], -32, %[rel]\n\t"
            "v_cmp_gt_u32_e64 %[sA], 32, %[t2]\n\t"
            "v_cndmask_b32_e32 %[t1], 0, %[bit], vcc\n\t"
            "v_or_b32_e32 %[mlo], %[mlo], %[t1]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32_e64 %[t1], 0, %[bit], %[sA]\n\t"
            "v_or_b32_e32 %[mhi], %[mhi], %[t1]\n\t"
            "L_nomark_%=:\n\t"
            "v_cmp_lt_u32_e64 %[sS], %[n20], %[p]\n\t"                  // the block's last 20 bytes
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_alignbyte_b32 %[w0], %[d1], %[d0], %[sh]\n\t"
            "v_alignbyte_b32 %[w1], %[d2], %[d1], %[sh]\n\t"
            "v_bfe_u32 %[L], %[w0], 4, 4\n\t"
            "v_bfe_u32 %[t1], %[w0], 8, 8\n\t"
            "v_cmp_eq_u32_e32 vcc, 15, %[L]\n\t"
            "v_and_b32_e32 %[tm], 15, %[w0]\n\t"
            "v_add_u32_e32 %[t1], 16, %[t1]\n\t"
            "v_cndmask_b32_e32 %[hop], %[L], %[t1], vcc\n\t"            // 15 + the extension byte + one more byte in front of the offset
            "v_add_u32_e32 %[hop], 3, %[hop]\n\t"
            "v_cmp_eq_u32_e64 %[sA], 15, %[tm]\n\t"
            "v_cmp_lt_u32_e32 vcc, 7, %[hop]\n\t"
            "v_and_b32_e32 %[t2], 7, %[hop]\n\t"
            "v_add_u32_e32 %[q], %[p], %[hop]\n\t"
            "v_perm_b32 %[x], %[w1], %[w0], %[t2]\n\t"
            "s_and_b64 vcc, vcc, %[sA]\n\t"                              // a match-length extension beyond the eight bytes in hand
            "s_cbranch_vccz L_no2_%=\n\t"
            "s_and_saveexec_b64 %[sB], vcc\n\t"
            "v_add_u32_e32 %[t2], %[ad], %[hop]\n\t"
            "v_mov_b32_e32 %[x], 0xff\n\t"                              // not staged: \"the extension goes on\" = the slow path
            "v_cmp_gt_u32_e32 vcc, %[cend], %[t2]\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "ds_read_u8 %[x], %[t2]\n\t"
            "s_mov_b64 exec, %[sB]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "L_no2_%=:\n\t"
            "v_and_b32_e32 %[x], 0xff, %[x]\n\t"
            "v_add_u32_e32 %[t1], 1, %[tm]\n\t"
            "v_lshl_or_b32 %[x], %[tm], 8, %[x]\n\t"
            "v_cmp_ge_u32_e64 %[sA], %[q], %[n]\n\t"                     // (only a long literal run gets there: p <= n - 20)
            "v_cmp_le_u32_e64 %[sB], %[k274], %[hop]\n\t"                // literal length 15 and its extension 255
            "v_cmp_eq_u32_e32 vcc, 0xfff, %[x]\n\t"                     // match length 15 and its extension 255 (or out of sight)
            "v_lshrrev_b32_e32 %[t1], 4, %[t1]\n\t"
            "s_or_b64 %[sA], %[sA], %[sB]\n\t"
            "s_or_b64 vcc, vcc, %[sS]\n\t"
            "v_add_u32_e32 %[q], %[q], %[t1]\n\t"
            "s_or_b64 vcc, vcc, %[sA]\n\t"
            "s_cbranch_vccnz L_slow_%=\n\t"
            "v_mov_b32_e32 %[p], %[q]\n\t"
            "s_mov_b64 exec, -1\n\t"
            "s_sub_u32 %[steps], %[steps], 1\n\t"
            "s_cmp_lg_u32 %[steps], 0\n\t"
            "s_cbranch_scc1 L_top_%=\n\t"
            "s_branch L_out_%=\n\t"
            "L_slow_%=:\n\t"
            "s_mov_b64 %[slow], vcc\n\t"
            "s_andn2_b64 exec, exec, vcc\n\t"
            "v_mov_b32_e32 %[p], %[q]\n\t"
            "s_sub_u32 %[steps], %[steps], 1\n\t"
            "s_branch L_out_%=\n\t"
            "L_none_%=:\n\t"
            "s_mov_b32 %[steps], -1\n\t"
            "L_out_%=:\n\t"
            "s_mov_b64 exec, -1\n\t"
            : [p] "+v"(p), [mlo] "+v"(mlo), [mhi] "+v"(mhi), [steps] "+s"(steps), [slow] "+s"(slow), [sA] "=&s"(sA), [sB] "=&s"(sB), [sS] "=&s"(sS),
              [ad] "=&v"(ad), [a4] "=&v"(a4), [sh] "=&v"(sh), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [w0] "=&v"(w0), [w1] "=&v"(w1), [L] "=&v"(L),
              [t1] "=&v"(t1), [hop] "=&v"(hop), [tm] "=&v"(tm), [q] "=&v"(q), [x] "=&v"(x), [rel] "=&v"(rel), [bit] "=&v"(bit), [t2] "=&v"(t2)
            : [e] "v"(e), [sg] "v"(sg), [mine] "s"(mine), [adj] "s"(adj), [cend] "s"(cend), [n] "s"(n), [n20] "s"(n20), [k274] "s"(274u)
            : "vcc", "scc", "memory");
        return slow;
    }

    // The loop again, cheaper on BOTH ports (RCX_WALK_FORM 5, round 6): the walking variable is the token's LDS ADDRESS (`ad`; the bound
    // `ead` is the segment's end or the block's last 20 bytes, whichever comes first, 0 for a lane that does not walk -- the caller's
    // portable loop takes what is left of the block's tail), the mark is one 64-bit shift and one 64-bit add under `exec`, the token's low nibble rides
    // along in the v_perm that fetches the match-length extension byte (selector byte 1 = the token, bytes 2-3 = zero), the step to
    // the next token is one v_addc (carry in = "match length 15"), and a token whose literals reach the block's end is not asked
    // for -- such a lane leaves the loop past its bound, the caller clamps.  22 vector instructions a step + 2 where a mark is set,
    // 12 on the scalar port with branches and waits (hipcc's loop: ~33 + ~45; walk_steps: 30 + 9 and ~22).  `exec` stays narrowed from
    // step to step (a lane past its bound stays there).  Leaves with `slow` = the lanes whose token it does not decide (a run of
    // 255s, an extension byte that is not staged), NOT advanced, or with steps = ~0 when no lane walks any more.
    __device__ __forceinline__ uint64_t walk_steps2(uint32_t& ad, uint32_t ead, uint32_t sgad, uint64_t& m, uint32_t& steps) const
    {
        const uint32_t cend = RCX_U((uint32_t)(uintptr_t)this->cbuf + (uint32_t)CBUF8);
        uint32_t a4, sh, d0, d1, d2, w0, w1, L, t1, hop, tm, x, rel, t2;
        uint64_t slow = 0, sA, sB, sW, bit;
        asm volatile(
            "L_top_%=:\n\t"
            "v_cmp_lt_u32_e32 vcc, %[ad], %[ead]\n\t"
            "s_cbranch_vccz L_none_%=\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "v_and_b32_e32 %[a4], -4, %[ad]\n\t"
            "ds_read_b32 %[d0], %[a4]\n\t"
            "ds_read_b32 %[d1], %[a4] offset:4\n\t"
            "ds_read_b32 %[d2], %[a4] offset:8\n\t"
            "v_sub_u32_e32 %[rel], %[ad], %[sgad]\n\t"                  // the mark, while the bytes are on their way
            "v_and_b32_e32 %[sh], 3, %[ad]\n\t"
            "v_cmp_gt_u32_e32 vcc, 64, %[rel]\n\t"                     // (two steps in three are head start: no lane in its own segment yet)
            "s_cbranch_vccz L_nomark_%=\n\t"
            "s_and_saveexec_b64 %[sW], vcc\n\t"
            "v_lshlrev_b64 %[bit], %[rel], 1\n\t"
            "v_lshl_add_u64 %[m], %[bit], 0, %[m]\n\t"                  // (a 64-bit OR the vector unit has not; a lane never marks a byte twice)
            "s_mov_b64 exec, %[sW]\n\t"
            "L_nomark_%=:\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_alignbyte_b32 %[w0], %[d1], %[d0], %[sh]\n\t"
            "v_alignbyte_b32 %[w1], %[d2], %[d1], %[sh]\n\t"
            "v_bfe_u32 %[L], %[w0], 4, 4\n\t"
            "v_bfe_u32 %[t1], %[w0], 8, 8\n\t"
            "v_cmp_eq_u32_e32 vcc, 15, %[L]\n\t"
            "v_and_b32_e32 %[tm], 15, %[w0]\n\t"
            "v_add_u32_e32 %[t1], 19, %[t1]\n\t"
            "v_add_u32_e32 %[L], 3, %[L]\n\t"
            "v_cndmask_b32_e32 %[hop], %[L], %[t1], vcc\n\t"            // token, (15 + the extension byte + that byte), literals, offset: where the match-length extension would be
            "v_cmp_eq_u32_e64 %[sA], 15, %[tm]\n\t"
            "v_cmp_lt_u32_e32 vcc, 7, %[hop]\n\t"
            "v_and_or_b32 %[t2], %[hop], 7, %[ksel]\n\t"
            "v_perm_b32 %[x], %[w1], %[w0], %[t2]\n\t"                  // token << 8 | the byte at `hop` of the eight in hand
            "s_and_b64 vcc, vcc, %[sA]\n\t"                              // a match-length extension beyond the eight bytes
            "s_cbranch_vccz L_no2_%=\n\t"
            "s_and_saveexec_b64 %[sB], vcc\n\t"
            "v_add_u32_e32 %[t2], %[ad], %[hop]\n\t"
            "v_mov_b32_e32 %[x], 0xfff\n\t"                             // not staged: \"the extension goes on\" = the slow path
            "v_cmp_gt_u32_e32 vcc, %[cend], %[t2]\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "ds_read_u8 %[x], %[t2]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_or_b32_e32 %[x], 0xf00, %[x]\n\t"
            "s_mov_b64 exec, %[sB]\n\t"
            "L_no2_%=:\n\t"
            "v_and_b32_e32 %[x], 0xfff, %[x]\n\t"
            "v_cmp_le_u32_e64 %[sB], %[k274], %[hop]\n\t"                // literal length 15 and its extension 255
            "v_cmp_eq_u32_e32 vcc, 0xfff, %[x]\n\t"                     // match length 15 and its extension 255 (or out of sight)
            "v_addc_co_u32_e64 %[ad], %[sW], %[ad], %[hop], %[sA]\n\t"   // + 1 for a match-length extension byte
            "s_or_b64 vcc, vcc, %[sB]\n\t"
            "s_cbranch_vccnz L_slow_%=\n\t"
            "s_sub_u32 %[steps], %[steps], 1\n\t"
            "s_cmp_lg_u32 %[steps], 0\n\t"
            "s_cbranch_scc1 L_top_%=\n\t"
            "s_branch L_out_%=\n\t"
            "L_slow_%=:\n\t"
            "s_mov_b64 %[slow], vcc\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "v_subb_co_u32_e64 %[ad], %[sW], %[ad], %[hop], %[sA]\n\t"   // the lanes the caller takes stay where they were
            "s_sub_u32 %[steps], %[steps], 1\n\t"
            "s_branch L_out_%=\n\t"
            "L_none_%=:\n\t"
            "s_mov_b32 %[steps], -1\n\t"
            "L_out_%=:\n\t"
            "s_mov_b64 exec, -1\n\t"
            : [ad] "+v"(ad), [m] "+v"(m), [steps] "+s"(steps), [slow] "+s"(slow), [sA] "=&s"(sA), [sB] "=&s"(sB), [sW] "=&s"(sW),
              [a4] "=&v"(a4), [sh] "=&v"(sh), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [w0] "=&v"(w0), [w1] "=&v"(w1), [L] "=&v"(L),
              [t1] "=&v"(t1), [hop] "=&v"(hop), [tm] "=&v"(tm), [x] "=&v"(x), [rel] "=&v"(rel), [bit] "=&v"(bit), [t2] "=&v"(t2)
            : [ead] "v"(ead), [sgad] "v"(sgad), [cend] "s"(cend), [k274] "s"(274u), [ksel] "s"(0x0c0c0000u)
            : "vcc", "scc", "memory");
        return slow;
    }
#endif

    // The same from the staged bytes: p lies in [cbase, chunk end) (the walk never leaves them).  ONE LDS round trip for nearly every
    // token: three aligned dwords give the eight bytes from p on -- the token, the literal-length extension, and (79 % of a text's
    // tokens have no literals) the match-length extension at p + 3; an extension byte further on is a second read.  A run of 255s
    // (a literal run of 270 bytes or more, a match of 274), bytes that are not staged and the last 20 bytes of the block go
    // through next_tok.  Round 5: the walk's step was the token byte, then the extension byte, and a call of next_tok whenever ANY of
    // the 64 lanes met a literal run of 15 or more -- most steps; 970 cycles a step with the CU to itself, and the executor waited
    // for its first batch and at chunk boundaries for 11 % of the kernel's time.
    __device__ __forceinline__ uint32_t next_tok_c(uint32_t p, bool on = true) const
    {
        const uint32_t n = this->n;
        bool slow = n < 20u || p > n - 20u;
        const int32_t i = (on && !slow) ? (int32_t)p - this->cbase : 0;      // (a lane that is not walking reads the buffer's first bytes)
        uint32_t q;
#if RCX_WALK_FORM == 0 || RCX_WALK_FORM >= 3
        // eight bytes in one round trip (three aligned dwords), a second read where the match-length extension lies further on
        q = 0;
        if (on && !slow) {
            const uint32_t* a = (const uint32_t*)(this->cbuf + (i & ~3));
            const uint32_t sh = (uint32_t)i & 3u;
            const uint32_t d0 = a[0], d1 = a[1], d2 = a[2];
            const uint32_t w0 = RCX_ALIGNBYTE(d1, d0, sh), w1 = RCX_ALIGNBYTE(d2, d1, sh);      // bytes p .. p + 7
            const uint32_t t = w0 & 0xffu, b1 = (w0 >> 8) & 0xffu;
            uint32_t L = t >> 4, hop = 3u;                            // hop: token, literals, offset -- where the match-length extension would be
            if (L == 15u) { L += b1; hop = 4u; if (b1 == 255u) slow = true; }
            hop += L;
            q = p + hop;
            if (q >= n) slow = true;                                  // (only a long literal run gets there: p <= n - 20)
            if ((t & 15u) == 15u && !slow) {
                uint32_t x;
                if (hop < 8u) x = ((hop < 4u ? w0 : w1) >> (8u * (hop & 3u))) & 0xffu;
                else if (i + (int32_t)hop < CBUF8) x = ((const RCX_LDS_AS uint8_t*)this->cbuf)[i + (int32_t)hop];
                else x = 255u;
                if (x == 255u) slow = true; else q++;
            }
        }
#elif !defined(RCX_NO_WALK_ASM)
        // The step as ISA: hipcc turns every select and every `||` of the portable form below into an s_and_saveexec / branch pair
        // (~100 instructions, 35 of them on the scalar port, a second LDS round trip wherever a lane needs byte 8 or later).  Here:
        // five aligned dwords in one round trip, 29 vector instructions, 5 compares.  The byte at `hop` of sixteen is two v_perm
        // (selector 13 = 0xff: a byte not among the sixteen reads as "the extension goes on" and the lane takes the slow path).
        {
            const uint32_t ad = (uint32_t)(uintptr_t)this->cbuf + (uint32_t)i;     // (low half of a generic LDS pointer = the LDS byte address)
            const uint32_t a4 = ad & ~3u, sh = ad & 3u;
            uint64_t d01, d23; uint32_t d4;
            asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %3 offset0:2 offset1:3\n\tds_read_b32 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(d01), "=&v"(d23), "=&v"(d4) : "v"(a4) : "memory");
            const uint32_t d0 = (uint32_t)d01, d1 = (uint32_t)(d01 >> 32), d2 = (uint32_t)d23, d3 = (uint32_t)(d23 >> 32);
            uint32_t w0, w1, w2, w3, L, t1, hop, tm, hop2, sel, xlo, xhi, inc;
            uint64_t sA, sB, sm;
            asm volatile(
                "v_alignbyte_b32 %[w0], %[d1], %[d0], %[sh]\n\t"
                "v_alignbyte_b32 %[w1], %[d2], %[d1], %[sh]\n\t"
                "v_bfe_u32 %[L], %[w0], 4, 4\n\t"
                "v_bfe_u32 %[t1], %[w0], 8, 8\n\t"
                "v_cmp_eq_u32_e32 vcc, 15, %[L]\n\t"
                "v_alignbyte_b32 %[w2], %[d3], %[d2], %[sh]\n\t"
                "v_alignbyte_b32 %[w3], %[d4], %[d3], %[sh]\n\t"
                "v_add_u32_e32 %[t1], 16, %[t1]\n\t"
                "v_cndmask_b32_e32 %[hop], %[L], %[t1], vcc\n\t"          // 15 + the extension byte + one more byte in front of the offset
                "v_add_u32_e32 %[hop], 3, %[hop]\n\t"
                "v_and_b32_e32 %[tm], 15, %[w0]\n\t"
                "v_add_u32_e32 %[q], %[p], %[hop]\n\t"
                "v_cmp_gt_u32_e32 vcc, 8, %[hop]\n\t"
                "v_add_u32_e32 %[hop2], -8, %[hop]\n\t"
                "v_and_b32_e32 %[sel], 7, %[hop]\n\t"
                "v_cmp_gt_u32_e64 %[sA], 8, %[hop2]\n\t"
                "v_perm_b32 %[xlo], %[w1], %[w0], %[sel]\n\t"
                "v_add_u32_e32 %[inc], 1, %[tm]\n\t"
                "v_cndmask_b32_e64 %[sel], 13, %[hop2], %[sA]\n\t"
                "v_perm_b32 %[xhi], %[w3], %[w2], %[sel]\n\t"
                "v_cndmask_b32_e32 %[xlo], %[xhi], %[xlo], vcc\n\t"
                "v_lshrrev_b32_e32 %[inc], 4, %[inc]\n\t"
                "v_and_b32_e32 %[xlo], 0xff, %[xlo]\n\t"
                "v_lshl_or_b32 %[xlo], %[tm], 8, %[xlo]\n\t"
                "v_cmp_eq_u32_e32 vcc, 0xfff, %[xlo]\n\t"                // match length 15 and its extension 255 (or out of sight)
                "v_cmp_ge_u32_e64 %[sA], %[q], %[n]\n\t"                 // (only a long literal run gets there: p <= n - 20)
                "v_cmp_le_u32_e64 %[sB], %[k274], %[hop]\n\t"            // literal length 15 and its extension 255
                "v_add_u32_e32 %[q], %[q], %[inc]\n\t"
                "s_or_b64 vcc, vcc, %[sA]\n\t"
                "s_or_b64 %[sm], vcc, %[sB]\n\t"
                : [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [L] "=&v"(L), [t1] "=&v"(t1), [hop] "=&v"(hop), [tm] "=&v"(tm),
                  [hop2] "=&v"(hop2), [sel] "=&v"(sel), [xlo] "=&v"(xlo), [xhi] "=&v"(xhi), [inc] "=&v"(inc), [q] "=&v"(q), [sA] "=&s"(sA), [sB] "=&s"(sB), [sm] "=&s"(sm)
                : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [d4] "v"(d4), [sh] "v"(sh), [p] "v"(p), [n] "s"(n), [k274] "s"(274u)
                : "vcc", "scc");
            slow = slow || RCX_INV_BALLOT(sm);
        }
#else
        {
            const uint32_t* a = (const uint32_t*)(this->cbuf + (i & ~3));
            const uint32_t sh = (uint32_t)i & 3u;
            const uint32_t d0 = a[0], d1 = a[1], d2 = a[2], d3 = a[3], d4 = a[4];
            const uint32_t w0 = RCX_ALIGNBYTE(d1, d0, sh), w1 = RCX_ALIGNBYTE(d2, d1, sh), w2 = RCX_ALIGNBYTE(d3, d2, sh), w3 = RCX_ALIGNBYTE(d4, d3, sh);   // bytes p .. p + 15
            const uint32_t t = w0 & 0xffu, b1 = (w0 >> 8) & 0xffu;
            const bool l15 = (t >> 4) == 15u;
            const uint32_t L = l15 ? 15u + b1 : t >> 4;
            const uint32_t hop = L + (l15 ? 4u : 3u);                // token, extension, literals, offset: where the match-length extension would be
            q = p + hop;
            const bool m15 = (t & 15u) == 15u;
            const uint32_t ws = hop < 8u ? (hop < 4u ? w0 : w1) : (hop < 12u ? w2 : w3);
            const uint32_t x = (ws >> (8u * (hop & 3u))) & 0xffu;
            slow = slow || (l15 && b1 == 255u) || q >= n || (m15 && (hop >= 16u || x == 255u));      // (q >= n: only a long run gets there, p <= n - 20)
            q += m15 ? 1u : 0u;
        }
#endif
#ifdef RCX_WALK_NOSLOW                    /* (instruction attribution of the parser-only build: results wrong on purpose) */
        if (on && slow && q <= p) q = p + 3u;
#else
        if (on && slow) q = next_tok(p);
#endif
        return q;
    }

    // Stage [cs - PRE, cs + CH + CSLACK) of the input (what exists of it); (cs + the input's misalignment) is a multiple of 16.
    __device__ void stage8(int32_t cs)
    {
        const uint8_t* in = this->in; const uint32_t n = this->n; uint8_t* cbuf = this->cbuf;
        const unsigned lane = this->lane;
        this->cbase = cs - PRE;
        constexpr int NR = (CBUF8 / 16 + 63) / 64;
        rcx_u32x4 v[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = r * 64 + (int)lane;
            const int32_t pos = this->cbase + 16 * j;
            v[r] = rcx_u32x4{0, 0, 0, 0};
            if (j < CBUF8 / 16 && pos >= 0 && (uint32_t)pos + 16 <= n) v[r] = *(const rcx_u32x4*)(in + pos);
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = r * 64 + (int)lane;
            const int32_t pos = this->cbase + 16 * j;
            if (j < CBUF8 / 16) {
                if (pos >= 0 && (uint32_t)pos + 16 <= n) *(rcx_u32x4*)(cbuf + 16 * j) = v[r];
                else {
                    for (int t = 0; t < 16; t++) {
                        const int32_t q = pos + t;
                        cbuf[16 * j + t] = (q >= 0 && (uint32_t)q < n) ? in[q] : (uint8_t)0;
                    }
                }
            }
        }
        rcx_wave_sync();
    }

    // Fields of the token at p, lane by lane (collect()'s general path, restated per lane): perr 0, RCX_E_MALFORMED, or -1 with
    // gL = L ("the literals are copied, then the offset read fails", lz4.rs:96-111).
    __device__ __forceinline__ void fields(uint32_t p, bool on, uint32_t& L, uint32_t& M, uint32_t& off, uint32_t& src, int& perr)
    {
        const uint8_t* in = this->in; const uint32_t n = this->n;
        L = 0; M = 0; off = 0; src = p + 1; perr = 0;
        bool slow = on;
        const int32_t ci = (int32_t)p - this->cbase;
        // Round 6: TWO dependent LDS reads, a lane per token, no loop -- the token and its literal-length extension, then (once the literal
        // run's length is known) the offset and the first match-length extension behind it.  The first version read 16 bytes at once and
        // took the byte-by-byte path below for every token with more than 12 literals: one token in forty, so four batches in five had a
        // lane in that path, and it was a fifth of the parser's time (the parser, not the executor, is what bounds this kernel: 0.41 of the
        // 0.50 ms with the executor switched off, benchmarks/pmc_insts.sh variant 45).  Left to the slow path: a run of 255s in either
        // length, bytes that are not staged, the block's last 20 bytes.
        const bool nok = n >= 20u && n <= 0xfffffe00u;                // (uniform: 32-bit position arithmetic below cannot wrap)
#if RCX_FIELDS_ISA && !defined(RCX_NO_WALK_ASM)
        // The fast path as ISA (round 6): hipcc's code for the C++ below is ~75 vector and ~45 scalar instructions a batch -- every `&&`
        // an s_and_saveexec / branch pair, every default a v_mov on each side of it; this is 30 + 14, straight down, `exec` narrowed
        // twice (the lanes whose literal run is in sight, then the lanes whose match length is) and `done` = the lanes it decided.
        if (nok) {
            const uint32_t cb0 = RCX_U((uint32_t)(uintptr_t)this->cbuf);
            uint32_t cix, ad, a4, sh, d0, d1, w0, Ln, b1, Mn, t, Lf, i, pe;
            uint64_t sB, sL, sX, done = 0;
            asm volatile(
                "v_subrev_u32_e32 %[ci], %[cbase], %[p]\n\t"
                "v_cmp_gt_u32_e32 vcc, %[k1], %[ci]\n\t"                 // staged: 0 <= ci, ci + 8 <= CBUF8
                "v_cmp_ge_u32_e64 %[sB], %[n20], %[p]\n\t"               // not in the block's last 20 bytes
                "s_and_b64 vcc, vcc, %[on]\n\t"
                "s_and_b64 vcc, vcc, %[sB]\n\t"
                "s_cbranch_vccz L_end_%=\n\t"
                "s_mov_b64 exec, vcc\n\t"
                "v_add_u32_e32 %[ad], %[cb0], %[ci]\n\t"
                "v_and_b32_e32 %[a4], -4, %[ad]\n\t"
                "ds_read_b32 %[d0], %[a4]\n\t"
                "ds_read_b32 %[d1], %[a4] offset:4\n\t"
                "v_and_b32_e32 %[sh], 3, %[ad]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_alignbyte_b32 %[w0], %[d1], %[d0], %[sh]\n\t"         // token, literal-length extension, ...
                "v_bfe_u32 %[Ln], %[w0], 4, 4\n\t"
                "v_bfe_u32 %[b1], %[w0], 8, 8\n\t"
                "v_and_b32_e32 %[Mn], 15, %[w0]\n\t"
                "v_cmp_eq_u32_e32 vcc, 15, %[Ln]\n\t"
                "v_cndmask_b32_e32 %[t], 0, %[b1], vcc\n\t"
                "s_mov_b64 %[sL], vcc\n\t"
                "v_add_u32_e32 %[Lf], %[Ln], %[t]\n\t"
                "v_addc_co_u32_e64 %[i], %[sX], %[Lf], 1, vcc\n\t"       // where the offset lies, from the token
                "v_cmp_ne_u32_e64 %[sB], %[k270], %[Lf]\n\t"             // (15 + 255: the extension goes on)
                "v_add3_u32 %[pe], %[p], %[i], 3\n\t"
                "v_add_u32_e32 %[ad], %[ad], %[i]\n\t"
                "v_cmp_ge_u32_e32 vcc, %[n], %[pe]\n\t"                  // offset and one extension byte lie in the block ...
                "s_and_b64 %[sB], %[sB], vcc\n\t"
                "v_cmp_gt_u32_e32 vcc, %[cend7], %[ad]\n\t"              // ... and are staged
                "s_and_b64 %[sB], %[sB], vcc\n\t"
                "s_and_b64 exec, exec, %[sB]\n\t"
                "s_cbranch_execz L_end_%=\n\t"
                "v_and_b32_e32 %[a4], -4, %[ad]\n\t"
                "ds_read_b32 %[d0], %[a4]\n\t"
                "ds_read_b32 %[d1], %[a4] offset:4\n\t"
                "v_and_b32_e32 %[sh], 3, %[ad]\n\t"
                "v_cmp_eq_u32_e32 vcc, 15, %[Mn]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_alignbyte_b32 %[w0], %[d1], %[d0], %[sh]\n\t"         // offset lo, hi, first match-length extension byte
                "v_bfe_u32 %[t], %[w0], 16, 8\n\t"
                "v_cndmask_b32_e32 %[t], 0, %[t], vcc\n\t"
                "v_cmp_ne_u32_e64 %[sB], %[k255], %[t]\n\t"              // (255: the extension goes on)
                "s_and_b64 exec, exec, %[sB]\n\t"
                "v_mov_b32_e32 %[L], %[Lf]\n\t"
                "v_and_b32_e32 %[off], 0xffff, %[w0]\n\t"
                "v_add3_u32 %[M], %[Mn], %[t], 4\n\t"
                "v_addc_co_u32_e64 %[src], %[sX], %[p], 1, %[sL]\n\t"
                "s_mov_b64 %[done], exec\n\t"
                "L_end_%=:\n\t"
                "s_mov_b64 exec, -1\n\t"
                : [L] "+v"(L), [M] "+v"(M), [off] "+v"(off), [src] "+v"(src), [done] "+s"(done), [sB] "=&s"(sB), [sL] "=&s"(sL), [sX] "=&s"(sX),
                  [ci] "=&v"(cix), [ad] "=&v"(ad), [a4] "=&v"(a4), [sh] "=&v"(sh), [d0] "=&v"(d0), [d1] "=&v"(d1), [w0] "=&v"(w0), [Ln] "=&v"(Ln), [b1] "=&v"(b1),
                  [Mn] "=&v"(Mn), [t] "=&v"(t), [Lf] "=&v"(Lf), [i] "=&v"(i), [pe] "=&v"(pe)
                : [p] "v"(p), [on] "s"(__ballot(on)), [cbase] "s"((uint32_t)this->cbase), [k1] "s"((uint32_t)(CBUF8 - 7)), [n20] "s"(n - 20u), [cb0] "s"(cb0),
                  [k270] "s"(270u), [k255] "s"(255u), [n] "s"(n), [cend7] "s"(cb0 + (uint32_t)(CBUF8 - 7))
                : "vcc", "scc", "memory");
            slow = on && !RCX_INV_BALLOT(done);
        }
        const bool fast = false;
        if (false) {
#else
        const bool fast = on && nok && p <= n - 20u && ci >= 0 && ci + 8 <= CBUF8;
        if (__ballot(fast)) {
#endif
            const uint32_t v0 = B::lds_load4u(this->cbuf, fast ? ci : 0);
            const uint32_t t = v0 & 0xffu, Ln = t >> 4, Mn = t & 15u, b1 = (v0 >> 8) & 0xffu;
            const uint32_t lx = Ln == 15u ? 1u : 0u;
            const uint32_t Lf = Ln + (lx ? b1 : 0u);
            const uint32_t i = 1u + lx + Lf;                            // where the offset lies, from the token
            const bool in2 = fast && !(lx && b1 == 255u) && p + i + 3u <= n && ci + (int32_t)i + 8 <= CBUF8;
            const uint32_t w = B::lds_load4u(this->cbuf, in2 ? ci + (int32_t)i : 0);      // offset lo, hi, first extension byte
            const uint32_t x = (w >> 16) & 0xffu;
            if (in2 && !(Mn == 15u && x == 255u)) {
                L = Lf; off = w & 0xffffu; M = Mn + 4u + (Mn == 15u ? x : 0u); src = p + 1u + lx;
                slow = false;
            }
        }
        if (__ballot(slow)) {
            if (slow) {                                              // (bytes through gb(): staged ones come from LDS)
                uint32_t q = p + 1;
                const uint32_t t = gb(p);
                L = t >> 4;
                if (L == 15) {
                    L += 255u * skip_ff(q);
                    for (;;) { if (q >= n) { perr = RCX_E_MALFORMED; break; } const uint32_t x = gb(q++); L += x; if (x != 255) break; }
                }
                src = q;
                if (!perr && L > n - q) perr = RCX_E_MALFORMED;
                if (!perr) {
                    q += L;
                    if (q != n) {
                        if (n - q < 2) perr = -1;
                        else {
                            off = gb(q) | (gb(q + 1) << 8);
                            q += 2;
                            M = t & 15u;
                            if (M == 15) {
                                M += 255u * skip_ff(q);
                                for (;;) { if (q >= n) { perr = -1; break; } const uint32_t x = gb(q++); M += x; if (x != 255) break; }
                            }
                            M += 4;
                        }
                    }
                }
            }
        }
    }

    __device__ __forceinline__ void ring_prio(uint32_t head, uint32_t low = RCX_V8_LOW)
    {
        if (head - RCX_U(this->ring8->tail) < low) __builtin_amdgcn_s_setprio(RCX_V8_HOT); else __builtin_amdgcn_s_setprio(RCX_V8_COLD);
    }
    // post one batch (p0: where its first token starts); returns false when the executor has given up
    // lane's token becomes entry `idx` (and idx + 1 when w1b != 0) of the batch, if `put`
    __device__ __forceinline__ bool post(uint32_t& head, int ns, int why, int perr, uint32_t gL, uint32_t gM, uint32_t goff, uint32_t gsrc, uint32_t p0, bool put, uint32_t idx, uint32_t w1, uint32_t w1b,
                                         uint32_t runM = 0, uint32_t runoff = 1, bool plain = false)
    {
        const unsigned lane = this->lane;
        auto ring = this->ring8;
        uint32_t t;
        for (;;) {
            // (tail and abort in ONE read and one scalar move each way: the two words are neighbours, and the priority below follows the tail just read)
            const uint64_t ta = *(const volatile RCX_LDS_AS uint64_t*)&ring->tail;
            t = RCX_U((uint32_t)ta);
            if (RCX_U((uint32_t)(ta >> 32))) return false;
            if (head - t < (uint32_t)NSLOT8) break;
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_sleep(RCX_V8_PSLEEP);
        }
        if (RCX_V8_ADAPT) { if (head - t < (uint32_t)RCX_V8_LOW) __builtin_amdgcn_s_setprio(RCX_V8_HOT); else __builtin_amdgcn_s_setprio(RCX_V8_COLD); }
        else if ((RCX_AGE_PRIO & 16) && this->agey) __builtin_amdgcn_s_setprio(RCX_PARSER_PRIO + 1); else __builtin_amdgcn_s_setprio(RCX_PARSER_PRIO);
        rcx_wave_sync();
        RCX_LDS_AS Slot8* sl = &ring->slot[head % NSLOT8];
        if (put) { sl->d[idx] = w1; if (w1b) sl->d[idx + 1] = w1b; }
        if (RCX_RUNSPLIT && SPLIT && __ballot(put && runM != 0u)) {
            // a RUN (a match that overlaps itself: offset < 16 and < its length) longer than SPLIT bytes: pieces of SPLIT bytes, and every piece
            // after the first copies from a whole number of periods back that lands in what the FIRST piece (and the period in front of
            // it) wrote -- byte x of a run equals byte x - m * offset for every m that stays behind the run's start -- so the pieces wait
            // for the first one only and copy side by side, where one lane filled the run 16 bytes a round (G-runs: 7.6 rounds an emit call)
            if (put && runM != 0u) {
                const uint32_t cnt = (runM + (uint32_t)SPLIT - 1u) / (uint32_t)SPLIT;
                for (uint32_t k = 1; k < cnt; k++) {
                    const uint32_t back = (((uint32_t)SPLIT * k + runoff - 1u) / runoff) * runoff;       // in [SPLIT k, SPLIT k + offset)
                    const uint32_t lk = runM - (uint32_t)SPLIT * k < (uint32_t)SPLIT ? runM - (uint32_t)SPLIT * k : (uint32_t)SPLIT;
                    sl->d[idx + k] = 0x80u | (lk << 8) | (back << 16);
                }
            }
        }
        if (PRED) {
            // Match chains are shortened HERE, by the wave that has the time: an entry whose whole source lies in the match
            // bytes of ONE earlier entry of the batch copies from that entry's source instead (Lz4V4::emit's redirection),
            // so that the executor need not wait for it.  The parser follows the output position itself (po; prlo: the
            // oldest byte a window rebuilt after a wide sequence can hold) and uses a window bound that is never below the
            // executor's: max(prlo, entry's first byte - H) -- the window base only moves in make_room(), to at most H
            // bytes behind the output position of the emit call the entry belongs to.  The shift replaces the offset in
            // the entry; the executor cannot tell (a redirected source is in the window, at least M bytes back).
            uint32_t T = 0;
            if (ns > 0) {
                rcx_wave_sync();
                const uint32_t e = (int)lane < ns ? sl->d[lane] : 0u;
                const uint32_t eL = (e & 0x80u) ? 0u : e & 0x7fu, eM = (e >> 8) & 0xffu, eoff = e >> 16;
                const uint32_t len = eL + eM;
                const uint32_t incl = rcx_wave_incl_scan(len);
                T = RCX_U(__builtin_amdgcn_readlane(incl, 63));
                const uint32_t o0 = po;
                const uint32_t ostart = o0 + incl - len, mdst = ostart + eL;
                const uint32_t slo = mdst - eoff;
                const uint32_t shi = (slo + eM < mdst) ? slo + eM : mdst;
                const uint32_t lb = ostart > (uint32_t)B::H ? ostart - (uint32_t)B::H : 0u;
                const uint32_t re = prlo > lb ? prlo : lb;
                const bool ok = eM && eoff != 0 && eoff <= mdst && slo >= re;
                const bool inb = ok && shi > o0;
                if (__ballot(inb)) {
                    const uint32_t ka = this->lane_of(ostart, slo > o0 ? slo : o0);
                    const uint32_t kb = this->lane_of(ostart, shi > o0 ? shi - 1 : o0);
                    const uint32_t pmd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ka << 2), (int)((!ok || eoff < eM) ? 0xffffffffu : mdst));
                    uint32_t prod = (inb && ka == kb && slo >= pmd && eoff >= eM) ? ka : 64u;
                    uint32_t S = eoff;
#pragma unroll
                    for (int rr = 0; rr < PRR; rr++) {
                        if (!__ballot(prod < 64u)) break;
                        const uint32_t j = prod < 64u ? prod : lane;
                        const uint32_t Sj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)S);
                        const uint32_t pj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)prod);
                        if (prod < 64u) {
                            if (mdst - S - Sj >= re && S + Sj <= mdst && S + Sj < 0x10000u) { S += Sj; prod = pj; } else prod = 64u;
                        }
                    }
                    if (S != eoff) sl->d[lane] = (e & 0xffffu) | (S << 16);
                }
            }
            po = RCX_U(po + T + ((why == B::SOLO_ || why == B::WIDE_) ? gL + gM : 0u));
            if (why == B::SOLO_ || why == B::WIDE_) { const uint32_t r = po > (uint32_t)B::RH ? po - (uint32_t)B::RH : 0u; prlo = r > prlo ? r : prlo; }
        }
        if (lane == 0) {
            sl->hdr[0] = (uint32_t)ns; sl->hdr[1] = (uint32_t)why | (plain ? 0x100u : 0u); sl->hdr[7] = p0;
            if (why != B::GO) {                                      // (the executor reads these behind a batch with a reason only)
                sl->hdr[2] = (uint32_t)perr; sl->hdr[3] = gL; sl->hdr[4] = gM; sl->hdr[5] = goff; sl->hdr[6] = gsrc;
            }
        }
        rcx_wave_sync();
        head++;
        if (lane == 0) ring->head = head;
        return true;
    }

    // Up to 64 list entries from `pos` on -> one batch (cut at the first entry the executor takes alone: a long sequence, or
    // an error).  Returns the entries used up, or -1 when the parser is done (error posted / executor gone).
    __device__ __forceinline__ int batch8(uint32_t& head, int32_t cs, uint32_t pos, uint32_t m)
    {
        const unsigned lane = this->lane;
        const bool on = lane < m;
        const uint32_t tp = (uint32_t)(cs + (on ? (int32_t)list[pos + lane] : 0));
        uint32_t L, M, off, src; int perr;
        V8P_T0();
        RCX_MARK("p8_fields");
        fields(tp, on, L, M, off, src, perr);
        V8P_ADD(3);
        RCX_MARK("p8_fields_end");
        // (a long match with a short period -- a run -- stays in the batch: its lane fills it 16 bytes a round from inside the window, where
        // a stop per run cut a batch of G-runs to ~14 sequences: 7 % of its runs are longer than 64 bytes)
        const bool longrun = M > (uint32_t)B::MCAP && M <= 255u && off != 0u && off < 16u;
        const unsigned long long stop = __ballot(on && (perr != 0 || L > (uint32_t)B::LCAP || (M > (uint32_t)B::MCAP && !longrun)));
        int nt = (int)m, why = B::GO, gerr = 0;                          // nt: tokens that become entries
        uint32_t gL = 0, gM = 0, goff = 0, gsrc = 0;
        int g = -1;
        if (stop) { g = __ffsll(stop) - 1; nt = g; }
        // entries: one per token, two for a match longer than SPLIT; at most 64 a batch
        // (a short period copies itself: halves with the same offset would only wait for each other, so a RUN's pieces copy from whole periods
        // back, post(); a token in the batch with M > MCAP is a run of <= 255 bytes, see `stop`, and SPLIT < M <= 2 SPLIT makes two pieces either way)
        const bool big = SPLIT && (int)lane < nt && M > (uint32_t)SPLIT;
        const bool two = big && off >= 16u;
        const bool rs = RCX_RUNSPLIT && big && off < 16u;
        const uint32_t ecnt = (int)lane < nt ? (RCX_RUNSPLIT ? (big ? (M + (uint32_t)SPLIT - 1u) / (uint32_t)SPLIT : 1u) : (two ? 2u : 1u)) : 0u;
        const uint32_t eincl = SPLIT ? rcx_wave_incl_scan(ecnt) : (uint32_t)lane + ecnt;
        uint32_t ne = (uint32_t)nt;
        if (SPLIT) {
            ne = nt ? RCX_U(__builtin_amdgcn_readlane(eincl, nt - 1)) : 0u;
            if (ne > 64u) {                                          // the tokens whose entries fit; the rest (and a stop) come next time
                nt = (int)__popcll(__ballot((int)lane < nt && eincl <= 64u));
                ne = RCX_U(__builtin_amdgcn_readlane(eincl, nt - 1));
                g = -1;
            }
        }
        if (g >= 0) {
            gerr = __builtin_amdgcn_readlane(perr, g);
            gL = RCX_U(__builtin_amdgcn_readlane(L, g)); gM = RCX_U(__builtin_amdgcn_readlane(M, g));
            goff = RCX_U(__builtin_amdgcn_readlane(off, g)); gsrc = RCX_U(__builtin_amdgcn_readlane(src, g));
            if (gerr) { why = B::ERR_; if (gerr > 0) gL = 0; }
            else why = (gL + gM <= (uint32_t)B::SOLO) ? B::SOLO_ : B::WIDE_;
        }
        const bool put = (int)lane < nt;
        const uint32_t M1 = (RCX_RUNSPLIT ? big : two) ? (uint32_t)SPLIT : M;
        // a PLAIN batch (the executor's emit6): every token with an offset, something
        // to emit, >= 40 bytes in front of the block's end -- and the first one >= 3 bytes behind its start
        bool plain = false;
        if (X6 && SPLIT) {
            const uint32_t n40 = this->n >= 40u ? this->n - 40u : 0u;
            const bool odd = (M != 0u && off == 0u) || L + M == 0u || tp > n40;      // (runs stay: emit6 fills them by period doubling)
            // (a batch with more than RCX_X6_MAXRUNS runs goes through emit5, whose round loop doubles all its runs side by side: emit6 fills its
            //  runs one after the other as they become ready -- right for a text's one run in seven batches, slower for G-runs' several a batch)
            plain = nt > 0 && RCX_U(tp) >= 3u && !__ballot(put && odd) && __popcll(__ballot(put && M != 0u && off < 16u && off < M)) <= RCX_X6_MAXRUNS;
#ifdef RCX_V8_WHY_STATS                  /* (simulator only: why a batch is not plain) */
            { const bool b0 = __ballot(put && M != 0u && off == 0u) != 0, b1 = __ballot(put && M != 0u && off < 16u && off < M && off > L) != 0, b2 = __ballot(put && L + M == 0u) != 0,
                         b3 = __ballot(put && (uint64_t)tp + 40u > (uint64_t)this->n) != 0, b6 = __ballot(put && M != 0u && off < 16u && off < M && off <= L) != 0, b4 = RCX_U(tp) < 3u;
              RCX_V8_STAT(0, lane == 0 && nt > 0 && b0); RCX_V8_STAT(1, lane == 0 && nt > 0 && b1); RCX_V8_STAT(2, lane == 0 && nt > 0 && b2); RCX_V8_STAT(3, lane == 0 && nt > 0 && b3);
              RCX_V8_STAT(4, lane == 0 && nt > 0 && b4); RCX_V8_STAT(5, lane == 0 && nt > 0); RCX_V8_STAT(6, lane == 0 && nt > 0 && b6); }
#endif
        }
        if (!post(head, (int)ne, why, gerr, gL, gM, goff, gsrc, RCX_U(tp), put, eincl - ecnt, L | (M1 << 8) | (off << 16),
                  (put && two) ? (0x80u | ((M - (uint32_t)SPLIT) << 8) | (off << 16)) : 0u, (put && rs) ? M : 0u, off ? off : 1u, plain)) return -1;
        V8P_ADD(4);
        RCX_MARK("p8_post_end");
        if (PROF8) pp[9] += 1;
        if (why == B::ERR_) return -1;
        return nt + (g >= 0 ? 1 : 0);
    }


    // (emit6 -- the straight-line executor of plain batches -- and its frame store live in k_lz4_emit6.hip: Lz4X6, this struct's base)
    __device__ void run_executor8(int32_t* st_out, uint32_t* len_out)
    {
        this->lane = rcx_lane();
        const unsigned lane = this->lane;
        this->init_window();
        int st = RCX_OK;
        uint32_t tail = 0;
        auto ring = this->ring8;
        uint32_t dyn_pv = RCX_PROG_PARSER, dyn_mine = 0;
        const uint32_t dyn_inv = RCX_AGE_DYN ? RCX_U(0xffffffffu / (this->n ? this->n : 1u)) : 0u;
        if (RCX_AGE_DYN && lane == 0) __hip_atomic_store(prog + wslot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            uint64_t te0 = PROF8 ? (uint64_t)__builtin_readcyclecounter() : 0;
            while (RCX_U(ring->head) == tail) __builtin_amdgcn_s_sleep(RCX_V8_ESLEEP);
            if (PROF8) { const uint64_t w_ = (uint64_t)__builtin_readcyclecounter() - te0; this->pw[0] += w_; this->pw[2] += 1; if (tail == 0) this->pw[9] = w_; }
            uint64_t te1 = PROF8 ? (uint64_t)__builtin_readcyclecounter() : 0;
            RCX_MARK("x8_polled");
            rcx_wave_sync();
            const RCX_LDS_AS Slot8* sl = &ring->slot[tail % NSLOT8];
            typename B::Batch bt;
            const uint32_t h0_ = sl->hdr[0], h1_ = sl->hdr[1], h7_ = sl->hdr[7];      // (all of the slot's usual reads in one LDS round trip)
            uint32_t w1 = sl->d[lane];
            bt.ns = (int)RCX_U(h0_); bt.why = (int)(RCX_U(h1_) & 0xffu);
            const bool plain6 = X6 && RCX_X6_MODE != 2 && (RCX_U(h1_) & 0x100u) != 0u;
            {   // (looked at only behind a batch with a reason: left undefined -- five s_mov a batch on the executor's chain)
                uint32_t u_;
                RCX_NOINIT_S(u_);
                bt.perr = (int)u_; bt.gL = u_; bt.gM = u_; bt.goff = u_; bt.gsrc = u_; bt.gnext = 0;
            }
            if (bt.why != B::GO) {                        // the one long sequence / the error behind the batch: only then (five v_readfirstlane: scalar-port work)
                bt.perr = (int)RCX_U(sl->hdr[2]);
                bt.gL = RCX_U(sl->hdr[3]); bt.gM = RCX_U(sl->hdr[4]); bt.goff = RCX_U(sl->hdr[5]); bt.gsrc = RCX_U(sl->hdr[6]);
            }
            const uint32_t p0 = RCX_U(h7_);
            rcx_wave_sync();
            if (RCX_AGE_DYN && (tail % (uint32_t)(RCX_AGE_DYN ? RCX_AGE_DYN : 1)) == 0u && bt.ns > 0) {
                // decide on what was read a period ago, against where I stood then; publish where I stand now; ask again
                const unsigned long long behind = __ballot(lane < 8u && dyn_pv < dyn_mine);
                const unsigned long long ahead = __ballot(lane < 8u && dyn_pv > dyn_mine && dyn_pv < RCX_PROG_LIVE);
                if (tail) agey = __popcll(behind) <= __popcll(ahead);
                uint32_t mine = RCX_U(p0 * dyn_inv);
                mine = mine < RCX_PROG_LIVE ? mine : RCX_PROG_LIVE - 1u;
                if (lane == 0) __hip_atomic_store(prog + wslot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dyn_pv = __hip_atomic_load(prog + (lane & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dyn_mine = mine;
            }
            if (RCX_AGE_DUTY) agey = (tail & 3u) < ((uint32_t)(RCX_AGE_DUTY >> (4u * agerank)) & 15u);
            const int agev = (agey ? 1 : 0) | ((RCX_AGE_LOW && (tail & 3u) < ((uint32_t)(RCX_AGE_LOW >> (4u * agerank)) & 15u)) ? 2 : 0);
            tail++;
            if (lane == 0) ring->tail = tail;             // the slot is in registers: hand it back
            // where each entry's literals lie: behind its token, and the tokens follow one another (the second half of a split
            // match, flagged 0x80, is no token)
            if (X6 && plain6) { RCX_V8_STAT(10, lane == 0 && bt.why == B::SOLO_); RCX_V8_STAT(11, lane == 0 && bt.why == B::WIDE_); RCX_V8_STAT(12, lane == 0 ? (uint32_t)bt.ns : 0u); RCX_V8_STAT(13, lane == 0); }
            if (X6 && plain6 && this->emit6(bt.ns, w1, p0, agev)) {
                if (this->after_batch(bt, st)) break;
                continue;
            }
            RCX_V8_STAT(9, lane == 0 && bt.ns > 0);
            RCX_V8_STAT(10, lane == 0 && bt.why == B::SOLO_); RCX_V8_STAT(11, lane == 0 && bt.why == B::WIDE_); RCX_V8_STAT(12, lane == 0 ? (uint32_t)bt.ns : 0u); RCX_V8_STAT(13, lane == 0);
            const bool cont = (w1 & 0x80u) != 0;
            if (cont) w1 &= ~0xffu;
            const uint32_t L = w1 & 0xffu, M = (w1 >> 8) & 0xffu;
            const uint32_t hop = ((int)lane < bt.ns && !cont) ? 3u + L + (L >= 15u ? 1u : 0u) + (M >= 19u ? 1u : 0u) : 0u;
            const uint32_t w0 = p0 + rcx_wave_incl_scan(hop) - hop + 1u + (L >= 15u ? 1u : 0u);
            int lo = 0, e = 0;
            RCX_MARK("x8_pre_emit");
            if (PROF8) { const uint64_t t_ = (uint64_t)__builtin_readcyclecounter(); this->pw[1] += t_ - te1; te1 = t_; }
#if defined(RCX_DUMMY_SALU) || defined(RCX_DUMMY_VALU) || defined(RCX_DUMMY_VCMP) || defined(RCX_DUMMY_VCMPX) || defined(RCX_DUMMY_RL) || defined(RCX_DUMMY_SNOP) || defined(RCX_DUMMY_BR) || defined(RCX_DUMMY_LDS)
            {   // port experiment (benchmarks/r5_lz4_ports.sh): N extra instructions a batch on one port, results untouched
                uint32_t ds_ = 0, dv_ = this->lane;
#ifdef RCX_DUMMY_SALU
#pragma unroll
                for (int k_ = 0; k_ < RCX_DUMMY_SALU; k_++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ds_) : : "scc");
#endif
#ifdef RCX_DUMMY_VALU
#pragma unroll
                for (int k_ = 0; k_ < RCX_DUMMY_VALU; k_++) asm volatile("v_add_u32_e32 %0, 1, %0" : "+v"(dv_));
#endif
#ifdef RCX_DUMMY_VCMP
#pragma unroll
                for (int k_ = 0; k_ < RCX_DUMMY_VCMP; k_++) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %0" : : "v"(dv_) : "vcc");
#endif
#ifdef RCX_DUMMY_VCMPX
#pragma unroll
                for (int k_ = 0; k_ < RCX_DUMMY_VCMPX; k_++) asm volatile("v_cmpx_le_u32_e32 vcc, %0, %0" : : "v"(dv_) : "vcc");   // (always true: exec stays)
#endif
#ifdef RCX_DUMMY_RL
#pragma unroll
                for (int k_ = 0; k_ < RCX_DUMMY_RL; k_++) asm volatile("v_readlane_b32 %0, %1, 0" : "=s"(ds_) : "v"(dv_));
#endif
#ifdef RCX_DUMMY_SNOP
#pragma unroll
                for (int k_ = 0; k_ < RCX_DUMMY_SNOP; k_++) asm volatile("s_nop 0");
#endif
#ifdef RCX_DUMMY_BR
#pragma unroll
                for (int k_ = 0; k_ < RCX_DUMMY_BR; k_++) asm volatile("s_cbranch_execz 1f\n1:");                                   // (exec is never zero here: not taken)
#endif
#ifdef RCX_DUMMY_LDS
#pragma unroll
                for (int k_ = 0; k_ < RCX_DUMMY_LDS; k_++) asm volatile("ds_read_b32 %0, %1" : "=v"(dv_) : "v"(0u) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                asm volatile("" : : "s"(ds_), "v"(dv_));
            }
#endif
            // CUT 8 (A/B): the executor only drains the ring -- what is left is the parser wave's instructions; 32: nor its own scan
            if (CUT & 8) { if (!(CUT & 32)) this->oend += RCX_U(__builtin_amdgcn_readlane(w0, 63)) & 1u; lo = bt.ns; if (bt.why == B::END_ || bt.why == B::ERR_) break; continue; }   // (nor the long sequence behind the batch: with no output its checks would end the block at once -- round 5's "parser share" measured a few batches per block)
            while (lo < bt.ns && !e) e = this->template emit5<false, PRED, CUT, (SPLIT ? SPLIT : 64), typename std::conditional<X6, Lz4V8, void>::type>(bt.ns, lo, w0, w1, nullptr, agev);
            RCX_MARK("x8_post_emit");
            if (e) { st = e; break; }
            if (PROF8) te1 = (uint64_t)__builtin_readcyclecounter();
            if (this->after_batch(bt, st)) break;
            if (PROF8) this->pw[3] += (uint64_t)__builtin_readcyclecounter() - te1;
            RCX_MARK("x8_loop_end");
        }
        if (st && lane == 0) ring->abort_ = 1;
        if (RCX_AGE_DYN && lane == 0) __hip_atomic_store(prog + wslot, RCX_PROG_FIN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!st) this->flush(this->oend, true);
        *st_out = st;
        *len_out = st ? 0u : this->oend;
    }

    __device__ void run_parser8()
    {
        this->lane = rcx_lane();
        const unsigned lane = this->lane;
        const uint32_t n = this->n;
        const uint32_t inmis = (uint32_t)((uintptr_t)this->in & 15u);
        uint32_t head = 0;
        uint32_t c = 0;                                              // the true walk's cursor
        int32_t cs = -(int32_t)inmis;                                // chunk start (cs + inmis is a multiple of 16)
        uint32_t lcnt = 0;                                           // list entries carried over from the last chunk
        while (c < n) {
            V8P_T0();
            if (RCX_V8_ADAPT) ring_prio(head, RCX_V8_LOW_WALK);
            else if (c != 0 && RCX_WALK_PRIO != RCX_PARSER_PRIO) __builtin_amdgcn_s_setprio(RCX_WALK_PRIO);
            else if ((RCX_AGE_PRIO & 16) && this->agey) __builtin_amdgcn_s_setprio(RCX_PARSER_PRIO + 1); else __builtin_amdgcn_s_setprio(RCX_PARSER_PRIO);   // (the ring drains while a chunk is staged, walked and linked)
            stage8(cs);
#if RCX_V8_PREFETCH
            {   // the next chunk's cache lines on their way to the L2 while this one is parsed (gfx950 has no prefetch instruction: a load
                // nobody waits for -- the parser's next vector-memory wait is the next chunk's staging)
                const int64_t q = (int64_t)cs + CH + CSLACK + 64 * (int64_t)lane;
                if (q >= 0 && q + 4 <= (int64_t)n) { const uint32_t x = *(const rcx_u32_u*)(this->in + q); asm volatile("" : : "v"(x)); }
            }
#endif
            const int nseg = ((int64_t)n - cs >= CH) ? NSEG : (int)(((int64_t)n - cs + SEGB - 1) / SEGB);
            const int k0 = (int)(((int32_t)c - cs) / SEGB);
            // ---- 1. every segment walked at once (a lane each)
            const int32_t s = cs + SEGB * (int32_t)lane;
            const uint32_t e = ((int64_t)s + SEGB < (int64_t)n) ? (uint32_t)(s + SEGB) : n;
            // (a token whose literal run reaches past the chunk -- incompressible data is ONE such token -- is the chunk's only
            // token: nothing to guess at, the 63 other lanes would walk garbage)
            const uint32_t nx = RCX_U(next_tok_c(c));
            const bool giant = nx >= n || (int64_t)nx >= (int64_t)cs + CH;
            const bool mine = !giant && (int)lane >= k0 && (int)lane < nseg;
            int32_t p0 = s - PRE; p0 = p0 < 0 ? 0 : p0;
            uint32_t p = ((int)lane == k0) ? c : (uint32_t)p0;
            uint64_t map = (giant && (int)lane == k0) ? 1ull << (c - (uint32_t)s) : 0ull;
            RCX_MARK("p8_walk");
#if (RCX_WALK_FORM == 5) && !defined(RCX_NO_WALK_ASM)
            if (n >= 20u) {                                          // (the portable loop below takes what this one leaves: the block's last 20 bytes)
                const uint32_t adj = RCX_U((uint32_t)(uintptr_t)this->cbuf - (uint32_t)this->cbase);     // LDS byte address of input byte q = q + adj
                const uint32_t e20 = e < n - 19u ? e : n - 19u;
                const uint32_t ead = mine ? e20 + adj : 0u, sgad = (uint32_t)s + adj;
                uint32_t ad = p + adj;
                for (;;) {
                    uint32_t steps = RCX_WALK_STEPS;
                    const uint64_t slow = walk_steps2(ad, ead, sgad, map, steps);
                    if (PROF8) pp[7] += steps == 0xffffffffu ? 0 : RCX_WALK_STEPS - steps;
                    if (slow) { if (RCX_INV_BALLOT(slow)) ad = next_tok(ad - adj) + adj; }
                    else if (steps == 0xffffffffu) break;
                    if (RCX_V8_ADAPT) ring_prio(head, RCX_V8_LOW_WALK);
                }
                p = ad - adj; p = p < n ? p : n;
            }
#elif (RCX_WALK_FORM >= 2) && !defined(RCX_NO_WALK_ASM)
            if (n >= 20u && (RCX_WALK_FORM == 2 || head == 0 || (RCX_WALK_FORM == 4 && head - RCX_U(this->ring8->tail) < (uint32_t)RCX_WALK_ISA_LOW))) {
                uint32_t mlo = (uint32_t)map, mhi = (uint32_t)(map >> 32);
                const uint64_t minem = __ballot(mine);
                for (;;) {
                    uint32_t steps = 4;
                    const uint64_t slow = walk_steps(p, e, (uint32_t)s, mlo, mhi, minem, steps);
                    if (PROF8) pp[7] += steps == 0xffffffffu ? 0 : 4 - steps;
                    if (slow) { if (RCX_INV_BALLOT(slow)) p = next_tok(p); }
                    else if (steps == 0xffffffffu) break;
                    if (RCX_V8_ADAPT) ring_prio(head, RCX_V8_LOW_WALK);
                }
                map = (uint64_t)mlo | ((uint64_t)mhi << 32);
            } else
#endif
            for (uint32_t wstep = 0;; wstep++) {
                const bool go = mine && p < e;
                if (!__ballot(go)) break;
                if (PROF8) pp[7] += 1;
                if (RCX_V8_ADAPT && (wstep & 3u) == 3u) ring_prio(head, RCX_V8_LOW_WALK);
#if RCX_WALK_FORM == 0 || RCX_WALK_FORM >= 3
                if (go) {
                    if ((int32_t)p >= s) map |= 1ull << (p - (uint32_t)s);
                    p = next_tok_c(p);
                }
#else
                const uint32_t q = next_tok_c(p, go);                // (no branch around the step: every lane computes, the walking ones keep the result)
                const uint64_t bit = (go && (int32_t)p >= s) ? 1ull << ((p - (uint32_t)s) & 63u) : 0ull;
                map |= bit;
                p = go ? q : p;
#endif
            }
            RCX_MARK("p8_walk_end");
            const uint32_t ex = p;                                   // where the segment's walk left it
            V8P_ADD(0);
            if (RCX_V8_ADAPT) ring_prio(head, RCX_V8_LOW_WALK);
            // ---- 2. link the segments.  Every lane first checks the usual case by itself: the walk of the segment before mine left
            // it at a byte my map has marked.  What is left -- a jump over a segment, walks that have not met -- is settled in
            // order by scalar code over the lanes' registers (and whatever that changes is checked again downstream).
            uint32_t lowv = 0; bool clr = false;
            if (giant) c = nx;
            else {
                const uint32_t eprev = (uint32_t)__shfl_up((int)ex, 1);
                const bool chk = mine && (int)lane > k0;
                bool ok = false;
                if (chk && eprev < e) { lowv = eprev - (uint32_t)s; ok = (map >> lowv) & 1ull; }
                if (!ok) lowv = 0;
                unsigned long long bad = __ballot(chk && !ok);
#ifdef RCX_LINK_NOREPAIR                  /* (instruction attribution of the parser-only build: results wrong on purpose) */
                bad = 0;
#endif
                c = RCX_U(__builtin_amdgcn_readlane(ex, nseg - 1));
                int k = 0; uint32_t cin = 0; bool forced = false;
                for (;;) {
                    if (!forced) {
                        if (!bad) break;
                        k = __ffsll(bad) - 1;
                        cin = RCX_U(__builtin_amdgcn_readlane(ex, k - 1));
                    }
                    bad &= ~(1ull << k);
                    const uint32_t sk = (uint32_t)(cs + SEGB * k);
                    const uint32_t ek = (sk + SEGB < n) ? sk + SEGB : n;
                    const uint32_t exk = RCX_U(__builtin_amdgcn_readlane(ex, k));
                    uint32_t X;
                    if (cin >= ek) { if ((int)lane == k) { clr = true; lowv = 0; } X = cin; }          // jumped over
                    else {
                        const uint64_t mk = (uint64_t)RCX_U(__builtin_amdgcn_readlane((uint32_t)map, k)) | ((uint64_t)RCX_U(__builtin_amdgcn_readlane((uint32_t)(map >> 32), k)) << 32);
                        if ((mk >> (cin - sk)) & 1ull) { if ((int)lane == k) { lowv = cin - sk; clr = false; } X = exk; }
                        else {                                       // follow the true walk until it meets k's map or leaves k
                            if (PROF8) pp[8] += 1;
                            uint64_t tm = 0;
                            uint32_t q = cin;
                            while (q < ek && !((mk >> (q - sk)) & 1ull)) { tm |= 1ull << (q - sk); q = RCX_U(next_tok_c(q)); }
                            const bool merged = q < ek;
                            const uint64_t nm = (merged ? mk & ~((1ull << (q - sk)) - 1ull) : 0ull) | tm;
                            if ((int)lane == k) { map = nm; lowv = 0; clr = false; }
                            X = merged ? exk : q;
                        }
                    }
                    if (k == nseg - 1) c = X;
                    forced = X != exk && k + 1 < nseg;
                    if (forced) { k++; cin = X; }
                }
            }
            if (clr) map = 0;
            map &= ~((1ull << lowv) - 1ull);
            if (!mine && !giant) map = 0;
            V8P_ADD(1);
            RCX_MARK("p8_link_end");
            if (PROF8) pp[6] += 1;
            // ---- 3. a lane per token: the maps unpacked into a position list (eight input bytes a lane), 64 entries a batch
            const int nwin = (nseg + LWIN - 1) / LWIN;
            for (int w = k0 / LWIN; w < nwin; w++) {
                if (RCX_V8_ADAPT) ring_prio(head);
                const int sl = LWIN * w + (int)(lane >> 3);                   // the segment whose map byte (lane & 7) this lane unpacks
                const uint32_t mlo = (uint32_t)__builtin_amdgcn_ds_bpermute(sl << 2, (int)(uint32_t)map);
                const uint32_t mhi = (uint32_t)__builtin_amdgcn_ds_bpermute(sl << 2, (int)(uint32_t)(map >> 32));
                uint32_t bits = (((lane & 4u) ? mhi : mlo) >> (8u * (lane & 3u))) & 0xffu;
                if (sl >= nseg) bits = 0;
                const uint32_t cnt = (uint32_t)__popc(bits);
                const uint32_t incl = rcx_wave_incl_scan(cnt);
                uint32_t at = lcnt + incl - cnt;
                const int32_t bpos = SEGB * sl + 8 * (int32_t)(lane & 7u);
                for (;;) {
                    if (!__ballot(bits != 0)) break;
                    if (bits) { const uint32_t b = (uint32_t)__ffs(bits) - 1u; bits &= bits - 1u; list[at++] = (int16_t)(bpos + (int32_t)b); }
                }
                lcnt += RCX_U(__builtin_amdgcn_readlane(incl, 63));
                rcx_wave_sync();
                V8P_ADD(2);
                RCX_MARK("p8_list_end");
                const bool last = (w + 1 == nwin) && c >= n;                  // the end of the block: nothing is carried over
                uint32_t pos = 0;
                while (lcnt - pos >= 64u || (last && lcnt > pos)) {
                    const int used = batch8(head, cs, pos, lcnt - pos < 64u ? lcnt - pos : 64u);
                    if (used < 0) return;
                    pos += (uint32_t)used;
                    if (PROF8) t0_ = (uint64_t)__builtin_readcyclecounter();
                }
                if (pos) {                                           // the entries left over move to the front
                    const uint32_t left = lcnt - pos;
                    const int16_t v = lane < left ? list[pos + lane] : (int16_t)0;
                    rcx_wave_sync();
                    if (lane < left) list[lane] = v;
                    lcnt = left;
                    rcx_wave_sync();
                }
            }
            // ---- the next chunk: right behind this one, or where the cursor is if a long literal run jumped further
            int32_t ncs = cs + CH;
            if (c < n && (int64_t)c >= (int64_t)ncs + CH) ncs = (int32_t)((c + inmis) & ~15u) - (int32_t)inmis;
            if (lcnt) {                                              // carried entries are relative to the chunk
                if ((int32_t)(int16_t)RCX_U((uint32_t)(uint16_t)list[0]) - (ncs - cs) < -30000) {     // (too far back for 16 bits: hand them over as a short batch)
                    uint32_t pos = 0;
                    while (lcnt > pos) { const int used = batch8(head, cs, pos, lcnt - pos); if (used < 0) return; pos += (uint32_t)used; }
                    lcnt = 0;
                } else {
                    if (lane < lcnt) list[lane] = (int16_t)(list[lane] - (int16_t)(ncs - cs));
                    rcx_wave_sync();
                }
            }
            cs = ncs;
        }
        post(head, 0, B::END_, 0, 0, 0, 0, 0, 0, false, 0, 0, 0);
    }
};

// X6: plain batches through emit6 (-1: wherever nothing else is being measured or cut out).  The mask table takes the place of the
// gathered matches' staging slots (SB = 0: emit5 stores what it gathers straight to its place).
template <int TC = 1024, int HH = 768, bool PROF8 = false, int PRE_ = RCX_V8_PRE, int SPLIT_ = 32, bool PRED = false, int PRR = 2, int CUT = 0, bool MIRROR = false, int X6_ = -1>
__global__ __launch_bounds__(128, 8) void k_lz4_decode_v8(rcx_kargs a, int only_status = 0)
{
    constexpr bool X6 = RCX_X6_MODE != 0 && (X6_ < 0 ? (CUT == 0 && !PRED && SPLIT_ != 0 && !MIRROR) : X6_ != 0);      // (MIRROR: the launch is bound by the PCIe link, and with emit6 inlined as well the kernel spills)
    typedef Lz4V8<TC, HH, X6 ? 0 : 16, PROF8, PRE_, SPLIT_, PRED, PRR, CUT, MIRROR, X6> S;
    __shared__ __align__(16) uint32_t s_mtab[X6 ? 20 * RCX_X6_MROW : 4];
    const uint64_t tk0 = PROF8 ? (uint64_t)__builtin_readcyclecounter() : 0;
    const uint64_t rt0 = PROF8 ? (uint64_t)__builtin_amdgcn_s_memrealtime() : 0;        // (100 MHz, one clock for the whole GPU: when the block started and ended)
    __shared__ __align__(16) uint8_t s_wbuf[S::WBUF5 + 16 + (X6 ? 16 : 0)];    // (X6: 16 bytes in front -- a source frame starts up to 3 bytes below the window)
    __shared__ __align__(16) typename S::Ring8 s_ring;
    __shared__ __align__(16) uint8_t s_cbuf[S::CBUF8 + 16];
    __shared__ int16_t s_list[S::LISTN];
    __shared__ uint32_t s_lmap[64];
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    if (only_status && a.status[b] != only_status) return;
    if (MIRROR && a.gate) {
        // the block's compressed bytes may still be on their way in (rcx_api.hip: one launch, the input in ranges on a copy stream).
        // Nothing of this block has been read yet -- no cache holds a line of it -- and the acquire behind the flag keeps it that way.
        uint32_t r = 0;
#pragma unroll
        for (int i = 0; i < 15; i++) r += b >= a.gate_bnd[i] ? 1u : 0u;
        if (r || a.gate_all) {
            if (threadIdx.x == 0) {
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                uint32_t ok = 1;
                if (b == (r ? a.gate_bnd[r - 1] : 0u)) {                       // the range's first block: the host's word, then everybody's
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
                s_lmap[0] = ok;
            }
            __syncthreads();
            const uint32_t ok = s_lmap[0];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            __syncthreads();
            if (!ok) { if (threadIdx.x == 0) a.status[b] = (int32_t)RCX_ST_GATE; return; }
        }
    }
    if (threadIdx.x == 0) { RCX_LDS_AS typename S::Ring8* r0 = (RCX_LDS_AS typename S::Ring8*)&s_ring; r0->head = 0; r0->tail = 0; r0->abort_ = 0; }
    __syncthreads();
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    S s;
    s.in = a.in_base + a.in_off[b];
    s.n = (uint32_t)a.in_len[b];
    s.out = a.out_base + a.out_off[b];
    if (MIRROR) s.out2 = a.out_mirror + a.out_off[b];
    const uint64_t cap64 = a.out_cap[b];
    s.cap = cap64 > 0xffffffffull ? 0xffffffffu : (uint32_t)cap64;
    s.cbuf = s_cbuf;
    s.wb_ = s_wbuf + (X6 ? 16 : 0);
    s.epos = nullptr;
    s.ring = nullptr;
    s.ring8 = (RCX_LDS_AS typename S::Ring8*)&s_ring;
    s.list = s_list;
    s.lmap = s_lmap;
    s.mtab = (RCX_LDS_AS uint32_t*)s_mtab;
    if (RCX_AGE_DYN) {
        const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);            // HW_ID[15:0]: wave, SIMD, pipe, CU, SH, SE
        const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;     // XCC_ID
        s.prog = g_lz4_prog + 8u * (((hw >> 4) & 0xfffu) | (xcc << 12));
        s.wslot = hw & 7u;
    }
    if (role == 0) {
        if (RCX_AGE_DYN && (threadIdx.x & 63u) == 0) __hip_atomic_store(s.prog + s.wslot, RCX_PROG_PARSER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (RCX_AGE_PRIO & 16) s.agey = (((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) >> 1) & 7u) >= (uint32_t)RCX_AGE_SPLIT;
        s.run_parser8();
        if (PROF8 && a.scratch && (threadIdx.x & 63u) == 0) {          // [0..9] parser phases, [10] parser total
            uint64_t* q = (uint64_t*)a.scratch + (size_t)b * 32;
            for (int i = 0; i < 10; i++) q[i] = s.pp[i];
            q[10] = (uint64_t)__builtin_readcyclecounter() - tk0;
        }
        return;
    }
    int32_t st; uint32_t olen;
    // HW_ID[3:0] = the wave's slot on its SIMD; on a GPU that starts the launch empty the slots fill in dispatch order (a workgroup's
    // two waves take a pair), so slot >> 1 is the executor's age rank among the four of its SIMD
    const uint32_t hwid = (PROF8 || RCX_AGE_PRIO) ? (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) : 0u;
    s.agey = RCX_AGE_PRIO ? ((hwid >> 1) & 7u) >= (uint32_t)RCX_AGE_SPLIT : true;
    s.agerank = (hwid >> 1) & 3u;
    RCX_SETPRIO_EXEC(s.agey);
    if (X6) { s.lane = rcx_lane(); s.mtab_init(); }
    s.run_executor8(&st, &olen);
    if (PROF8 && a.scratch && (threadIdx.x & 63u) == 0) {              // [11] executor total, [12] its wave slot, [16..] executor phases (Lz4V5::pw)
        uint64_t* q = (uint64_t*)a.scratch + (size_t)b * 32;
        q[11] = (uint64_t)__builtin_readcyclecounter() - tk0;
        q[12] = hwid;
        q[13] = rt0; q[14] = (uint64_t)__builtin_amdgcn_s_memrealtime();
        q[15] = (uint64_t)(uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) | ((uint64_t)(uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32);   // HW_ID, XCC_ID
        for (int i = 0; i < 12; i++) q[16 + i] = s.pw[i];
    }
    if ((threadIdx.x & 63u) == 0) {
        a.status[b] = st;
        a.out_len[b] = olen;
        if (a.in_used) a.in_used[b] = s.n;
    }
}
