// k_lz4_decode_v5.hip -- LZ4 block decode, two waves per block: a PARSER wave and an EXECUTOR wave in one
// workgroup, coupled by a ring of batch descriptors in LDS (reference: BlockDecoder::decode, src/lz4.rs:67-140).
//
// Why: v4 (one wave per block) is latency bound, not throughput bound.  Measured on MI355X with 4096 blocks
// (16 waves per CU): a wave waits on s_waitcnt 45 % of its life while VALU, SALU and LDS are each 50-60 % busy,
// and the token walk (a serial SALU chain) and the copy rounds (LDS round trips) never overlap inside one wave.
// BASELINE configs[1] offers 4096 blocks = 16 waves per CU, so the only way to more waves in flight is inside
// a block.  The walk needs nothing from the output side, so it runs ahead in its own wave:
//
//   parser   (wave 0): stages the compressed bytes in LDS, walks the tokens (Lz4V4::collect, unchanged),
//                      extracts each entry's fields and posts {literal source, L, M, offset} descriptors;
//   executor (wave 1): scan, validation, literals (fetched from HBM/L2, the parser's staging buffer is private),
//                      old-match gathers, dependency masks, redirection, copy rounds, flush (Lz4V4's window code).
//
// Both waves are always co-resident (same workgroup), so the LDS hand-off cannot deadlock: the parser only blocks
// on a full ring, the executor only on an empty one, and an executor-side error raises `abort` for the parser.
// LDS operations of one wave are performed in issue order, so "write data, then write head" / "read data, then
// write tail" need no more than compiler ordering.  64 VGPRs (launch bound) keep 8 waves per SIMD = 16 blocks per CU.
#include "rcx_dev.h"
#ifndef RCX_LDS_AS
#define RCX_LDS_AS __attribute__((address_space(3)))
#endif
#ifndef RCX_EXEC_PRIO
#define RCX_EXEC_PRIO 1
#endif
#ifndef RCX_PARSER_PRIO
#define RCX_PARSER_PRIO 2
#endif
#ifndef RCX_INF_ROUNDS_PRIO
#define RCX_INF_ROUNDS_PRIO 1
#endif
#ifndef RCX_FLUSH_PRIO
#define RCX_FLUSH_PRIO 2
#endif
#ifndef RCX_ROUND_PRIO
#define RCX_ROUND_PRIO 3
#endif
// RCX_AGE_PRIO: among waves of one priority a SIMD issues the OLDEST first, so of the sixteen blocks of a CU the four dispatched
// first run 17 % faster than the four dispatched last (1.23 M against 1.46 M timer ticks, in four steps that follow the executor
// wave's slot on its SIMD: benchmarks/r4_lz4_tail.py) and the launch ends with the last.  An executor knows its age rank among the
// four of its SIMD from its wave slot (HW_ID: on a GPU that starts the launch empty the slots fill in dispatch order), and the
// older half gives way where it hurts least.  Measured on the headline, same box, two runs each (benchmarks/r4_lz4_age.sh):
//   0 (no age term) 0.612-0.617 ms | bit 1, the older half's copy rounds one level down: 0.605-0.612 | bits 1 + 3, its drain too
//   (10, the default): 0.598-0.605 | bit 0, the younger half's default phase one level up: 0.639-0.642 | bit 4, the younger half's parser one up: 0.625 | bit 5, the older half's default phase one down:
//   no change | the split after one or three ranks instead of two: 0.602-0.616.
#ifndef RCX_AGE_PRIO
#define RCX_AGE_PRIO 10
#endif
#ifndef RCX_DEGUARD
#define RCX_DEGUARD 1                    /* 0: every section behind its ballot, as before round 5 (A/B) */
#endif
#ifndef RCX_AGE_SPLIT
#define RCX_AGE_SPLIT 2                  /* age ranks below this are "old" */
#endif
// (young: bit 0 = this batch at the young half's levels; bit 1 = RCX_AGE_LOW: this batch's plain stretches one level down)
#define RCX_SETPRIO_EXEC(young) do { if ((int)(young) & 2) __builtin_amdgcn_s_setprio(RCX_EXEC_PRIO - 1); else if ((RCX_AGE_PRIO & 1) && ((int)(young) & 1)) __builtin_amdgcn_s_setprio(RCX_EXEC_PRIO + 1); else if ((RCX_AGE_PRIO & 32) && !((int)(young) & 1)) __builtin_amdgcn_s_setprio(RCX_EXEC_PRIO - 1); else __builtin_amdgcn_s_setprio(RCX_EXEC_PRIO); } while (0)
#define RCX_SETPRIO_ROUND(young) do { if ((RCX_AGE_PRIO & 2) && !((int)(young) & 1)) __builtin_amdgcn_s_setprio(RCX_ROUND_PRIO - 1); else __builtin_amdgcn_s_setprio(RCX_ROUND_PRIO); } while (0)
// bit 3: the older half drains one level down
#define RCX_SETPRIO_FLUSH(young) do { if ((RCX_AGE_PRIO & 8) && !((int)(young) & 1)) __builtin_amdgcn_s_setprio(RCX_FLUSH_PRIO - 1); else __builtin_amdgcn_s_setprio(RCX_FLUSH_PRIO); } while (0)
#include <type_traits>
#ifdef RCX_MARKS
#define RCX_MARK(name) asm volatile("; MARK " name)
#else
#define RCX_MARK(name) do { } while (0)
#endif

// The copy rounds of emit5 as ISA (gfx950): every round, the lanes whose producers are done (no pending lane among `dep`) and
// whose match does not overlap itself copy up to 16 bytes -- five aligned dword reads + v_alignbyte, the exec-narrowing byte
// stores of rcx_lds_store16 -- and the loop ends when no such lane is left (the caller keeps self-overlapping matches out of `pend`: emit5's
// portable loop takes those).  hipcc's loop for the same source carries the lane sets as booleans through v_cndmask / v_cmp_ne
// pairs and re-derives exec at every `if`: ~35 vector + ~25 scalar instructions a round around the stores; this is 17 + 9.
// The pending set is ONE compiler-allocated SGPR pair (its halves are read through vcc): a first version named six SGPRs of its
// own (s84..s89), which raised the kernel's SGPR count from 78 to 96, cost the eighth wave per SIMD, and ran 0.89 instead of
// 0.63 ms.  The wave runs with all lanes on (exec is restored to -1).
//   src_a / dst_a: LDS byte addresses of the lane's source and destination; mc: bytes to copy; prog: bytes done (in / out)
// Returns the lanes still pending.
#ifndef RCX_NO_ROUNDS_ASM
__device__ __forceinline__ uint64_t rcx_lz4_rounds(uint32_t src_a, uint32_t dst_a, uint32_t mc, uint32_t dep_lo, uint32_t dep_hi, uint64_t pend, uint32_t& prog)
{
    // (round 5: the pending set lives in vcc for the whole loop -- every compare writes a scratch pair in its e64 form -- so a round
    // has three scalar instructions where it had seven: no copy to vcc for the halves, exec straight from the readiness test, the
    // lanes that finish collected in one pair and taken out once, which also sets the loop's condition.  A scalar instruction on the
    // executor's chain costs it ~11 cycles)
    uint32_t t0, t1, a, a4, nv, da, d0, d1, d2, d3, d4;
    uint64_t sT, sF;
#define RCX_RB4(V, O0, O1, O2, O3)                                         \
        "v_cmpx_lt_u32_e64 %[sT], " #O0 ", %[nv]\n\t"                      \
        "s_cbranch_execz L_sdone_%=\n\t"                                   \
        "ds_write_b8 %[da], %[" V "] offset:" #O0 "\n\t"                    \
        "v_cmpx_lt_u32_e64 %[sT], " #O1 ", %[nv]\n\t"                      \
        "v_lshrrev_b32_e32 %[t0], 8, %[" V "]\n\t"                          \
        "ds_write_b8 %[da], %[t0] offset:" #O1 "\n\t"                       \
        "v_cmpx_lt_u32_e64 %[sT], " #O2 ", %[nv]\n\t"                      \
        "ds_write_b8_d16_hi %[da], %[" V "] offset:" #O2 "\n\t"             \
        "v_cmpx_lt_u32_e64 %[sT], " #O3 ", %[nv]\n\t"                      \
        "ds_write_b8_d16_hi %[da], %[t0] offset:" #O3 "\n\t"
    asm volatile(
        "s_mov_b64 vcc, %[pend]\n\t"
        "L_top_%=:\n\t"
        "v_and_b32_e32 %[t0], vcc_lo, %[dlo]\n\t"
        "v_and_b32_e32 %[t1], vcc_hi, %[dhi]\n\t"
        "v_or_b32_e32 %[t0], %[t0], %[t1]\n\t"
        "v_cmp_eq_u32_e64 %[sT], 0, %[t0]\n\t"
        "s_and_b64 exec, %[sT], vcc\n\t"                        // the ready lanes; scc: any
        "s_cbranch_scc0 L_out_%=\n\t"
        "v_add_u32_e32 %[a], %[srca], %[prog]\n\t"
        "v_sub_u32_e32 %[nv], %[mc], %[prog]\n\t"
        "v_and_b32_e32 %[a4], -4, %[a]\n\t"
        "ds_read_b32 %[d0], %[a4]\n\t"
        "ds_read_b32 %[d1], %[a4] offset:4\n\t"
        "ds_read_b32 %[d2], %[a4] offset:8\n\t"
        "ds_read_b32 %[d3], %[a4] offset:12\n\t"
        "ds_read_b32 %[d4], %[a4] offset:16\n\t"
        "v_and_b32_e32 %[a], 3, %[a]\n\t"
        "v_min_u32_e32 %[nv], 16, %[nv]\n\t"
        "v_add_u32_e32 %[da], %[dsta], %[prog]\n\t"
        "v_add_u32_e32 %[prog], %[prog], %[nv]\n\t"
        "v_cmp_ge_u32_e64 %[sF], %[prog], %[mc]\n\t"             // of this round's lanes, those that are done with it
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_alignbyte_b32 %[d0], %[d1], %[d0], %[a]\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_alignbyte_b32 %[d1], %[d2], %[d1], %[a]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_alignbyte_b32 %[d2], %[d3], %[d2], %[a]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_alignbyte_b32 %[d3], %[d4], %[d3], %[a]\n\t"
        RCX_RB4("d0", 0, 1, 2, 3) RCX_RB4("d1", 4, 5, 6, 7) RCX_RB4("d2", 8, 9, 10, 11) RCX_RB4("d3", 12, 13, 14, 15)
        "L_sdone_%=:\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_andn2_b64 vcc, vcc, %[sF]\n\t"                        // scc: a lane still pending
        "s_cbranch_scc1 L_top_%=\n\t"
        "L_out_%=:\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_mov_b64 %[pend], vcc\n\t"
        : [pend] "+s"(pend), [prog] "+v"(prog), [t0] "=&v"(t0), [t1] "=&v"(t1), [a] "=&v"(a), [a4] "=&v"(a4), [nv] "=&v"(nv), [da] "=&v"(da),
          [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [d4] "=&v"(d4), [sT] "=&s"(sT), [sF] "=&s"(sF)
        : [srca] "v"(src_a), [dsta] "v"(dst_a), [mc] "v"(mc), [dlo] "v"(dep_lo), [dhi] "v"(dep_hi)
        : "vcc", "scc", "memory");
#undef RCX_RB4
    return pend;
}

// The same loop for batches that hold SELF-OVERLAPPING matches (offset < 16 and < length: runs).  Such a lane copies by period
// doubling: its source stays at match - offset while `offset + done` bytes of the pattern stand -- round k may copy that many,
// a whole number of periods, without reading a byte it has yet to write -- and once 16 bytes could be copied at a time the lane
// switches to a source D = offset * ceil(16 / offset) bytes back (same phase, no overlap) and is an ordinary lane from then on.
// Per lane: `sa` the source address, `pm` the mask on `done` in the source offset (0 while doubling, -1 after), `cb` the cap base
// (bytes of a round <= done + cb: the offset while doubling, 16 after); `sconv` = sa after the switch (an ordinary lane's own
// sa: the switch is idempotent for it, so no lane mask is needed).  Eight more vector instructions a round than the loop above,
// which batches without such matches keep.  The portable loop of emit5 reads such a source byte by byte, eight bytes a pass
// (~100 instructions), twice, before its lanes go on as ordinary ones.
// Every pending lane is eligible here, and the lowest pending lane never waits: the loop ends with nothing pending.
__device__ __forceinline__ uint64_t rcx_lz4_rounds_ovl(uint32_t sa, uint32_t sconv, uint32_t pm, uint32_t cb, uint32_t dst_a, uint32_t mc,
                                                       uint32_t dep_lo, uint32_t dep_hi, uint64_t pend, uint32_t& prog)
{
    uint32_t t0, t1, a, a4, nv, da, d0, d1, d2, d3, d4;
#define RCX_RB4(V, O0, O1, O2, O3)                                         \
        "v_cmpx_lt_u32_e32 vcc, " #O0 ", %[nv]\n\t"                        \
        "s_cbranch_execz L_sdone_%=\n\t"                                   \
        "ds_write_b8 %[da], %[" V "] offset:" #O0 "\n\t"                    \
        "v_cmpx_lt_u32_e32 vcc, " #O1 ", %[nv]\n\t"                        \
        "v_lshrrev_b32_e32 %[t0], 8, %[" V "]\n\t"                          \
        "ds_write_b8 %[da], %[t0] offset:" #O1 "\n\t"                       \
        "v_cmpx_lt_u32_e32 vcc, " #O2 ", %[nv]\n\t"                        \
        "ds_write_b8_d16_hi %[da], %[" V "] offset:" #O2 "\n\t"             \
        "v_cmpx_lt_u32_e32 vcc, " #O3 ", %[nv]\n\t"                        \
        "ds_write_b8_d16_hi %[da], %[t0] offset:" #O3 "\n\t"
    asm volatile(
        "L_top_%=:\n\t"
        "s_mov_b64 vcc, %[pend]\n\t"
        "v_and_b32_e32 %[t0], vcc_lo, %[dlo]\n\t"
        "v_and_b32_e32 %[t1], vcc_hi, %[dhi]\n\t"
        "v_or_b32_e32 %[t0], %[t0], %[t1]\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[t0]\n\t"
        "s_and_b64 vcc, vcc, %[pend]\n\t"
        "s_cbranch_vccz L_out_%=\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "v_and_b32_e32 %[a], %[prog], %[pm]\n\t"
        "v_add_u32_e32 %[a], %[sa], %[a]\n\t"
        "v_sub_u32_e32 %[nv], %[mc], %[prog]\n\t"
        "v_and_b32_e32 %[a4], -4, %[a]\n\t"
        "ds_read_b32 %[d0], %[a4]\n\t"
        "ds_read_b32 %[d1], %[a4] offset:4\n\t"
        "ds_read_b32 %[d2], %[a4] offset:8\n\t"
        "ds_read_b32 %[d3], %[a4] offset:12\n\t"
        "ds_read_b32 %[d4], %[a4] offset:16\n\t"
        "v_and_b32_e32 %[a], 3, %[a]\n\t"
        "v_add_u32_e32 %[t1], %[prog], %[cb]\n\t"              // what stands of the pattern (an ordinary lane: >= 16)
        "v_min_u32_e32 %[nv], 16, %[nv]\n\t"
        "v_min_u32_e32 %[nv], %[nv], %[t1]\n\t"
        "v_add_u32_e32 %[da], %[dsta], %[prog]\n\t"
        "v_add_u32_e32 %[prog], %[prog], %[nv]\n\t"
        "v_cmp_lt_u32_e32 vcc, %[prog], %[mc]\n\t"
        "s_andn2_b64 %[pend], %[pend], exec\n\t"
        "s_or_b64 %[pend], %[pend], vcc\n\t"
        "v_add_u32_e32 %[t1], %[prog], %[cb]\n\t"              // 16 bytes a round possible from now on: the lane turns ordinary
        "v_cmp_le_u32_e32 vcc, 16, %[t1]\n\t"
        "v_cndmask_b32_e32 %[sa], %[sa], %[sconv], vcc\n\t"
        "v_cndmask_b32_e64 %[pm], %[pm], -1, vcc\n\t"
        "v_cndmask_b32_e64 %[cb], %[cb], 16, vcc\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_alignbyte_b32 %[d0], %[d1], %[d0], %[a]\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_alignbyte_b32 %[d1], %[d2], %[d1], %[a]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_alignbyte_b32 %[d2], %[d3], %[d2], %[a]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_alignbyte_b32 %[d3], %[d4], %[d3], %[a]\n\t"
        RCX_RB4("d0", 0, 1, 2, 3) RCX_RB4("d1", 4, 5, 6, 7) RCX_RB4("d2", 8, 9, 10, 11) RCX_RB4("d3", 12, 13, 14, 15)
        "L_sdone_%=:\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_cmp_lg_u64 %[pend], 0\n\t"
        "s_cbranch_scc1 L_top_%=\n\t"
        "L_out_%=:\n\t"
        "s_mov_b64 exec, -1\n\t"
        : [pend] "+s"(pend), [prog] "+v"(prog), [sa] "+v"(sa), [pm] "+v"(pm), [cb] "+v"(cb), [t0] "=&v"(t0), [t1] "=&v"(t1), [a] "=&v"(a), [a4] "=&v"(a4),
          [nv] "=&v"(nv), [da] "=&v"(da), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [d4] "=&v"(d4)
        : [sconv] "v"(sconv), [dsta] "v"(dst_a), [mc] "v"(mc), [dlo] "v"(dep_lo), [dhi] "v"(dep_hi)
        : "vcc", "scc", "memory");
#undef RCX_RB4
    return pend;
}
#endif

// SB: bytes of a gathered (old) match that are staged per lane and ride the copy rounds; the rest is stored straight to its place
template <int CB, int TC = 2560, int HH = 2048, bool PROF5 = false, int SB = 32, bool ADLER = false, bool MIRROR = false>
struct Lz4V5 : Lz4V4<CB, false, TC, HH, ADLER, MIRROR> {
    static_assert(SB == 0 || SB == 16 || SB == 32, "staging slot");
    typedef Lz4V4<CB, false, TC, HH, ADLER, MIRROR> B;
    static constexpr int NSLOT = 3;
    struct Slot { uint32_t hdr[16]; uint32_t desc[64][2]; };
    struct Ring { Slot slot[NSLOT]; volatile uint32_t head, tail, abort_, pad; };
    static constexpr int STAGE5 = B::LIN + 64;         // 64 lanes x 32 bytes of old-match staging (bytes 32.. of a longer
    static constexpr int WBUF5 = STAGE5 + 64 * SB;     // gathered match go straight to their place): 16 blocks per CU fit
    RCX_LDS_AS Ring* ring;                         // address space 3: the volatile head / tail accesses must be ds_read / ds_write, not flat
    // 256 bytes of LDS scratch (or null): "which entry produces output byte x" as a bitmap of the entries' first bytes + a
    // running count per word, two LDS reads and a popcount, instead of a binary search over the lanes (Lz4V4::lane_of: six
    // DEPENDENT ds_bpermute round trips, twice per batch: ~1500 cycles of the executor's ~11 000 per batch)
    uint32_t* lmap = nullptr;
    static constexpr int LMW = (B::TCAP + 31) / 32 + 1;              // bitmap words; the counts (a byte each) follow them
    static constexpr bool LMOK = LMW <= 50;                          // 4 * LMW + LMW <= 256
    uint64_t pw[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // PROF5: [0] cycles waiting on the ring, [2] batches; executor phases [4] scan+validate [5] loads+chains [6] literal/gather stores [7] copy rounds [8] flush [9] rounds [10] emit calls

    // ------------------------------------------------------------------------------------------ parser wave
    __device__ void run_parser()
    {
        this->lane = rcx_lane();
        const unsigned lane = this->lane;
        uint32_t cur = 0;
        if (this->n) this->stage(0); else { this->cbase = 0; this->cend = 0; }
        uint32_t s_L = 0, s_M = 0, s_off = 0, s_src = 0;
        uint32_t head = 0;
        for (;;) {
            const typename B::Batch bt = this->collect(cur, s_L, s_M, s_off, s_src);
            rcx_wave_sync();
            // entry fields (the executor never sees the staging buffer)
            uint32_t w0 = 0, w1 = 0;
            if ((int)lane < bt.ns) {
                const uint32_t e = this->epos[lane];
                uint32_t L, M, off, src;
                if (e & B::FLAG) { L = s_L; M = s_M; off = s_off; src = s_src; }
                else {
                    const uint32_t t = this->cbuf[(int32_t)e - this->cbase];
                    L = t >> 4; M = (t & 15u) + 4u; src = e + 1;
                    const uint32_t w = B::lds_load4u(this->cbuf, (int32_t)(src + L) - this->cbase);
                    off = w & 0xffffu;
                    if (M == 19u) M += (w >> 16) & 0xffu;
                }
                w0 = src; w1 = L | (M << 8) | (off << 16);
            }
            // a free slot
            uint64_t tp0 = PROF5 ? (uint64_t)__builtin_readcyclecounter() : 0;
            for (;;) {
                const uint32_t t = RCX_U(ring->tail);
                if (RCX_U(ring->abort_)) return;
                if (head - t < (uint32_t)NSLOT) break;
                __builtin_amdgcn_s_sleep(16);                        // ~1K cycles: an executor batch takes ~20K
            }
            if (PROF5) { pw[0] += (uint64_t)__builtin_readcyclecounter() - tp0; pw[2] += 1; }
            rcx_wave_sync();
            RCX_LDS_AS Slot* sl = &ring->slot[head % NSLOT];
            *(RCX_LDS_AS uint64_t*)sl->desc[lane] = (uint64_t)w0 | ((uint64_t)w1 << 32);
            if (lane == 0) {
                sl->hdr[0] = (uint32_t)bt.ns; sl->hdr[1] = (uint32_t)bt.why; sl->hdr[2] = (uint32_t)bt.perr;
                sl->hdr[3] = bt.gL; sl->hdr[4] = bt.gM; sl->hdr[5] = bt.goff; sl->hdr[6] = bt.gsrc;
            }
            rcx_wave_sync();
            head++;
            if (lane == 0) ring->head = head;
            // Issue priorities (s_setprio), found by measurement: an executor in its dependency analysis and copy rounds (chains
            // of LDS round trips) runs at 3, in its drain at 2, elsewhere at 1; the parser runs at 2 while the ring has a free
            // slot -- a batch delivered sooner is a batch the executor never waits for -- and at 0 once the ring is full (it
            // would only collect a batch it cannot post).  No priorities: 300 GiB/s on G-text; these: 364 (with the LDS fixes).
            if (head - RCX_U(ring->tail) < (uint32_t)NSLOT) __builtin_amdgcn_s_setprio(RCX_PARSER_PRIO); else __builtin_amdgcn_s_setprio(0);
            if (bt.why == B::END_ || bt.why == B::ERR_) return;
            if (bt.why == B::STAGE_) { this->stage(cur); continue; }
            if (bt.why == B::SOLO_ || bt.why == B::WIDE_) cur = bt.gnext;
        }
    }

    // ------------------------------------------------------------------------------------------ executor wave
    // One batch: lane i holds entry i's descriptor (w0 = literal source position, w1 = L | M << 8 | offset << 16).
    // Entries [lo, hi) as in Lz4V4::emit (hi < ns only when the batch's output exceeds TCAP bytes); advances lo.
    // LITLDS: the literal bytes sit in an LDS buffer (`litbuf`, with 32 bytes of slack) instead of the input in HBM --
    // that is how the inflate front end (k_inflate3.hip) feeds this executor.
    // NORED: the parser has already shortened the chains (k_lz4_decode_v8.hip: the offsets ARE the shifts).
    // CUT (A/B builds, results wrong on purpose): phases left out so that instruction counters can be attributed -- 1: copy
    // rounds, 2: chain analysis, 4: literal / gather stores, 16: window drain
    // FARCAP: the longest match that is gathered with 16-byte loads when its source has left the window (a longer one takes the
    // byte path): k_lz4_decode_v8 hands over matches of at most 32 bytes, so two of the four gathers and their stores are not compiled
    // FS (round 6): the derived executor that has Lz4X6's frame store (k_lz4_emit6.hip), or void -- with it (and no staging slots, SB = 0) a
    // gathered match goes to its place as one or two masked frames loaded from `source - a` instead of two exec-narrowing byte stores
    template <bool LITLDS = false, bool NORED = false, int CUT = 0, int FARCAP = 64, class FS = void>
    __device__ int emit5(int ns, int& lo, uint32_t w0, uint32_t w1, const uint8_t* litbuf = nullptr, int young = 1)   // young: RCX_AGE_PRIO (true: the plain levels)
    {
        const unsigned lane = this->lane;
        const uint8_t* in = this->in; uint8_t* out = this->out; uint8_t* wb_ = this->wb_;
        const uint32_t cap = this->cap, n = this->n;
        uint64_t tq_ = PROF5 ? (uint64_t)__builtin_readcyclecounter() : 0;
#define V5P_ADD(slot) do { RCX_MARK("emit5_" #slot); if (PROF5) { const uint64_t t1_ = (uint64_t)__builtin_readcyclecounter(); pw[slot] += t1_ - tq_; tq_ = t1_; } } while (0)
        RCX_MARK("emit5_in");
        this->make_room(B::TCAP);
        RCX_MARK("emit5_room");
        V5P_ADD(11);
        const int lo0 = lo;
        bool act = (int)lane >= lo && (int)lane < ns;
        uint32_t L = act ? w1 & 0xffu : 0u, M = act ? (w1 >> 8) & 0xffu : 0u, off = act ? w1 >> 16 : 0u;
        const uint32_t src = w0;
        const uint32_t len = L + M;
        const uint32_t incl = rcx_wave_incl_scan(len);
        uint32_t T = RCX_U(__builtin_amdgcn_readlane(incl, 63));
        int hi = ns;
        if (T > (uint32_t)B::TCAP) {                                  // rare: take the prefix that fits (an entry is <= 96 bytes)
            hi = lo + (int)__popcll(__ballot(act && incl <= (uint32_t)B::TCAP));
            act = act && (int)lane < hi;
            if (!act) { L = 0; M = 0; off = 0; }
            T = RCX_U(__builtin_amdgcn_readlane(incl, hi - 1));
        }
        lo = hi;
        const uint32_t oend0 = this->oend;
        const uint32_t ostart = oend0 + incl - len;
        const uint32_t mdst = ostart + L;
        int err = 0;
        if ((uint64_t)oend0 + T <= (uint64_t)cap) {                   // (uniform) the whole batch fits: only the offsets can be wrong
            if (act && M && (off == 0 || off > mdst)) err = RCX_E_MALFORMED;
        } else if (act) {
            if (L > cap - ostart || ostart > cap) err = RCX_E_OUTPUT_TOO_SMALL;
            else if (M && (off == 0 || off > mdst)) err = RCX_E_MALFORMED;
            else if (M && M > cap - mdst) err = RCX_E_OUTPUT_TOO_SMALL;
        }
        const unsigned long long bad = __ballot(err != 0);
        if (bad) return __builtin_amdgcn_readlane(err, __ffsll(bad) - 1);
        V5P_ADD(4);

        const int32_t lbase = this->lbase;
        const uint32_t re = this->rlo_eff();
        const int32_t li_o = (int32_t)ostart - lbase;
        const int32_t li_m = li_o + (int32_t)L;
        const uint32_t slo = mdst - off;
        const uint32_t shi = (slo + M < mdst) ? slo + M : mdst;
        const bool isfar = M && slo < re;                            // source drained and slid out of the window

        // ---- loads first: literals (parser staged them a moment ago: L2 hits) and old matches, 16 bytes each
        rcx_u32x4 g0, g1;                                            // (RCX_NOINIT: not zeroed -- a register that was not loaded is stored with length 0; eight v_mov a batch each)
        RCX_NOINIT4(g0); RCX_NOINIT4(g1);
        const bool lit16 = L && (LITLDS || (uint64_t)src + 32u <= (uint64_t)n);
        const bool litb = L && !lit16;                               // within 32 bytes of the block's end: byte loads
        if (LITLDS) {
            if (__ballot(lit16)) {
                const int32_t qi = lit16 ? (int32_t)src : 0;
                uint32_t x0, x1, x2, x3;
                B::lds_load16u(litbuf, qi, x0, x1, x2, x3);
                g0 = rcx_u32x4{x0, x1, x2, x3};
                if (__ballot(lit16 && L > 16)) {
                    B::lds_load16u(litbuf, qi + 16, x0, x1, x2, x3);
                    g1 = rcx_u32x4{x0, x1, x2, x3};
                }
            }
        } else if (!(CUT & 0x400) && (RCX_DEGUARD ? n >= 32u : __ballot(lit16) != 0)) {      // (RCX_DEGUARD: a ballot guard is a v_cmp, SALU and a branch -- four vector instructions each, DESIGN 3.1 -- around a load that nearly every batch issues anyway)
            // EVERY lane loads (a lane without such literals reads the block's first bytes -- there are >= 32 of them when any lane
            // has lit16 -- and its store below has length 0): no exec juggling, no zeroed registers for the lanes left out
            const uint32_t q = lit16 ? src : 0u;
            g0 = *(const rcx_u32x4_u*)(in + q);
            if (__ballot(lit16 && L > 16)) g1 = *(const rcx_u32x4_u*)(in + q + 16);
        }
        rcx_u32x4 f0, f1, f2, f3;
        RCX_NOINIT4(f0); RCX_NOINIT4(f1); RCX_NOINIT4(f2); RCX_NOINIT4(f3);
        constexpr uint32_t FC = FARCAP < B::MCAP ? (uint32_t)FARCAP : (uint32_t)B::MCAP;
        constexpr bool FR = !std::is_void<FS>::value && SB == 0 && FC <= 32;
        const uint32_t af = FR ? (uint32_t)li_m & 3u : 0u;                                  // (FR: the frame's shift -- the window is 16-byte aligned)
        const bool far16 = isfar && M <= FC && (uint64_t)slo + (uint32_t)B::MCAP <= (uint64_t)cap && (!FR || slo >= 4u);      // (k_lz4_decode_v8 batches runs of up to 255 bytes: if one is ever outside the window, byte loads)
        const bool farb = isfar && !far16;
        uint32_t f2w = 0;
        if (!(CUT & 0x200) && (RCX_DEGUARD ? cap >= 64u : __ballot(far16) != 0)) {   // the same: all lanes load, from the output's first 64 bytes where there is no far match
            const uint32_t q = far16 ? slo - af : 0u;
            f0 = *(const rcx_u32x4_u*)(out + q);
            if (FC > 16 && __ballot(far16 && af + M > 16)) f1 = *(const rcx_u32x4_u*)(out + q + 16);
            if (FR && __ballot(far16 && af + M > 32)) f2w = *(const rcx_u32_u*)(out + q + 32);
            if (FC > 32 && __ballot(far16 && M > 32)) f2 = *(const rcx_u32x4_u*)(out + q + 32);
            if (FC > 48 && __ballot(far16 && M > 48)) f3 = *(const rcx_u32x4_u*)(out + q + 48);
        }

        if (!LITLDS) RCX_SETPRIO_ROUND(young); else if (RCX_INF_ROUNDS_PRIO) __builtin_amdgcn_s_setprio(RCX_ROUND_PRIO);   // (the inflate executor: the plain level)
#if defined(RCX_DUMMY3_MOV) || defined(RCX_DUMMY3_ADD1) || defined(RCX_DUMMY3_ADD2) || defined(RCX_DUMMY3_ALIGN) || defined(RCX_DUMMY3_SALU) || defined(RCX_DUMMY3_ADD1I)
        {   // port experiment, second site: N extra instructions a batch at the executor's HIGH priority (benchmarks/r5_lz4_ports2.sh)
            uint32_t dv_ = this->lane, dw_ = this->lane ^ 5u, dx_ = this->lane + 9u, ds_ = 0;
#ifdef RCX_DUMMY3_MOV
#pragma unroll
            for (int k_ = 0; k_ < RCX_DUMMY3_MOV; k_++) asm volatile("v_mov_b32_e32 %0, 0" : "=v"(dv_));
#endif
#ifdef RCX_DUMMY3_ADD1
#pragma unroll
            for (int k_ = 0; k_ < RCX_DUMMY3_ADD1; k_++) asm volatile("v_add_u32_e32 %0, 1, %0" : "+v"(dv_));
#endif
#ifdef RCX_DUMMY3_ADD1I
#pragma unroll
            for (int k_ = 0; k_ < RCX_DUMMY3_ADD1I; k_++) { if (k_ & 1) asm volatile("v_add_u32_e32 %0, 1, %0" : "+v"(dv_)); else asm volatile("v_add_u32_e32 %0, 1, %0" : "+v"(dw_)); }   // two independent chains
#endif
#ifdef RCX_DUMMY3_ADD2
#pragma unroll
            for (int k_ = 0; k_ < RCX_DUMMY3_ADD2; k_++) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(dv_) : "v"(dw_));
#endif
#ifdef RCX_DUMMY3_ALIGN
#pragma unroll
            for (int k_ = 0; k_ < RCX_DUMMY3_ALIGN; k_++) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(dv_) : "v"(dw_), "v"(dx_));
#endif
#ifdef RCX_DUMMY3_SALU
#pragma unroll
            for (int k_ = 0; k_ < RCX_DUMMY3_SALU; k_++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ds_) : : "scc");
#endif
            asm volatile("" : : "v"(dv_), "v"(dw_), "v"(dx_), "s"(ds_));
        }
#endif
        // ---- producer lanes of [slo, shi) inside this batch, chains redirected (see Lz4V4::emit) while the loads fly
        unsigned long long dep = 0;
        bool inb = M && !isfar && shi > oend0;
        uint32_t S = off;
        if (!(CUT & 2) && (RCX_DEGUARD || __ballot(inb))) {
            uint32_t ka, kb;
            if (LMOK && lmap != nullptr && !__ballot(act && len == 0u)) {       // (an empty entry shares its first byte with the next one: the search handles that)
                uint8_t* const lcnt = (uint8_t*)(lmap + LMW);
                if (lane < (unsigned)LMW) lmap[lane] = 0;
                rcx_wave_sync();
                const uint32_t rel = ostart - oend0;
                if (act) atomicOr(&lmap[rel >> 5], 1u << (rel & 31u));
                rcx_wave_sync();
                const uint32_t pc = (uint32_t)__popc(lane < (unsigned)LMW ? lmap[lane] : 0u);
                const uint32_t ex = rcx_wave_incl_scan(pc) - pc;
                if (lane < (unsigned)LMW) lcnt[lane] = (uint8_t)ex;
                rcx_wave_sync();
                // entry_of(x): the active entry whose first byte is the last one <= x = lo0 + lcnt[w] + popc(lmap[w] up to x's bit) - 1
                // (both lookups' four reads in one LDS round trip: hipcc waits for the first pair before it asks for the second)
                uint32_t ra = (slo > oend0 ? slo : oend0) - oend0, rb = (shi > oend0 ? shi - 1 : oend0) - oend0;
                ra = ra < (uint32_t)B::TCAP ? ra : (uint32_t)B::TCAP; rb = rb < (uint32_t)B::TCAP ? rb : (uint32_t)B::TCAP;
                uint32_t ca = lcnt[ra >> 5], ma = lmap[ra >> 5], cb2 = lcnt[rb >> 5], mb = lmap[rb >> 5];
                RCX_SETTLE4(ca, ma, cb2, mb);
                ka = ((uint32_t)lo0 + ca + (uint32_t)__popc(ma & (0xffffffffu >> (31u - (ra & 31u)))) - 1u) & 63u;
                kb = ((uint32_t)lo0 + cb2 + (uint32_t)__popc(mb & (0xffffffffu >> (31u - (rb & 31u)))) - 1u) & 63u;
            } else {
                ka = this->lane_of(ostart, slo > oend0 ? slo : oend0);
                kb = this->lane_of(ostart, shi > oend0 ? shi - 1 : oend0);
            }
            uint32_t prod = 64u;
            if (!NORED) {
                const uint32_t pmd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ka << 2), (int)((isfar || off < M) ? 0xffffffffu : mdst));
                prod = (inb && ka == kb && slo >= pmd && off >= M) ? ka : 64u;
            }
            if (!NORED) {
                // the lane's chain state as ONE word (producer | first << 7 | last << 13 | in-batch << 19) through the redirection rounds:
                // two selects a round instead of an s_and_saveexec nest around five assignments (the executor's scalar instructions
                // are its dearest)
                uint32_t st = prod | (ka << 7) | (kb << 13) | ((uint32_t)inb << 19);
#pragma unroll
                for (int rr = 0; rr < B::RR; rr++) {
                    const bool has = (st & 64u) == 0u;
                    if (!__ballot(has)) break;
                    const uint32_t j = has ? (st & 63u) : lane;
                    const uint32_t Sj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)S);
                    const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)st);
                    const uint32_t Sn = S + Sj;
                    const bool ok = has && mdst - Sn >= re && Sn <= mdst;
                    S = ok ? Sn : S;
                    st = ok ? pk : (st | 64u);                        // (not shortened: no producer any more; the range stays)
                }
                ka = (st >> 7) & 63u; kb = (st >> 13) & 63u; inb = (st >> 19) & 1u;
            }
            if (inb) {
                const unsigned long long upto = (kb >= 63) ? ~0ull : ((2ull << kb) - 1ull);
                dep = upto & ~((1ull << ka) - 1ull) & ((1ull << lane) - 1ull);
            }
        }

        V5P_ADD(5);
        // ---- literals, then gathered matches: registers -> their place in the window
        if (!(CUT & 4) && (RCX_DEGUARD || __ballot(L != 0))) {
            RCX_LDS_STORE16(wb_ + li_o, g0[0], g0[1], g0[2], g0[3], lit16 ? (L < 16u ? L : 16u) : 0u);
            if (__ballot(lit16 && L > 16)) RCX_LDS_STORE16(wb_ + li_o + 16, g1[0], g1[1], g1[2], g1[3], (lit16 && L > 16u) ? L - 16u : 0u);
        }
        if (!(CUT & 4) && (RCX_DEGUARD || __ballot(isfar))) {
            uint8_t* d = wb_ + li_m;
            const uint32_t mf = far16 ? M : 0u;
            if (SB != 0 && far16) { uint8_t* sl = wb_ + STAGE5 + SB * (int32_t)lane; *(rcx_u32x4*)sl = f0; if (SB == 32) *(rcx_u32x4*)(sl + 16) = f1; }
            if constexpr (FR) {
                const uint32_t n1 = mf < 16u ? mf : 16u;
                uint8_t* fp = d - (int32_t)af;
                if (n1) static_cast<FS*>(this)->template store_frame<5>(fp, f0[0], f0[1], f0[2], f0[3], f1[0], af, n1);
                if (mf > 16u) static_cast<FS*>(this)->template store_frame<5>(fp + 16, f1[0], f1[1], f1[2], f1[3], f2w, af, mf - n1);
            }
            if (!FR && SB == 0) RCX_LDS_STORE16(d, f0[0], f0[1], f0[2], f0[3], mf < 16u ? mf : 16u);     // no staging: every gathered byte goes straight to its place
            if (!FR && SB <= 16 && __ballot(mf > 16)) RCX_LDS_STORE16(d + 16, f1[0], f1[1], f1[2], f1[3], mf > 16u ? (mf < 32u ? mf - 16u : 16u) : 0u);
            if (FC > 32 && __ballot(mf > 32)) RCX_LDS_STORE16(d + 32, f2[0], f2[1], f2[2], f2[3], mf > 32u ? (mf < 48u ? mf - 32u : 16u) : 0u);
            if (FC > 48 && __ballot(mf > 48)) RCX_LDS_STORE16(d + 48, f3[0], f3[1], f3[2], f3[3], mf > 48u ? mf - 48u : 0u);
        }
        // the byte paths -- literals within 32 bytes of the block's end, a gathered match too long or too close to the end of the output
        // for 16-byte loads -- behind ONE test (they write other bytes than the stores above: the order does not matter)
        if (!(CUT & 4) && __ballot(litb || farb)) {
            for (uint32_t i = 0; __ballot(litb && i < L); i++)
                if (litb && i < L) wb_[li_o + (int32_t)i] = in[src + i];
            uint8_t* d = wb_ + li_m;
            for (uint32_t i = 0; __ballot(farb && i < M); i++)
                if (farb && i < M) d[i] = out[slo + i];
        }
        rcx_wave_sync();
        V5P_ADD(6);

        // ---- window matches: copy rounds (16 bytes per ready lane), see Lz4V4::emit.  A short-period match (off < 16, off < M)
        // reads its periodic source 8 bytes a pass; once 16 bytes stand it copies from off * ceil(16 / off) >= 16 bytes
        // behind (same period, no overlap) on the plain path.  Batches without such a match (nearly all of a text) run the
        // loop instantiated without that switch: its `sbase` / `ovl` stay loop invariant.
        {
            const int32_t sbase0 = far16 ? STAGE5 + SB * (int32_t)lane : (int32_t)(mdst - S) - lbase;
            const bool ovl0 = M && !isfar && off < 16u && off < M;
            const uint32_t Mc = far16 ? (M < (uint32_t)SB ? M : (uint32_t)SB) : M;     // staged gathers ride the rounds for their first SB bytes
            bool pending0 = M != 0 && !farb && !(SB == 0 && far16);
            uint32_t prog0 = 0;
#ifndef RCX_NO_ROUNDS_ASM
            if (!(CUT & 1) && !(CUT & 64)) {    // the hand-written loop takes every round the plain (non-overlapping) lanes can make
                if (CUT & 128) __builtin_amdgcn_s_setprio(1);
                const uint32_t wa = (uint32_t)(uintptr_t)wb_;      // (low half of a generic LDS pointer = the LDS byte address)
                uint64_t left;
                if (!(CUT & 0x800) && __ballot(ovl0)) {            // runs in the batch: the loop that also copies them, by period doubling
                    const uint32_t D = off * ((((uint32_t)(0x11111111223357F0ull >> (4u * (off & 15u)))) & 15u) + 1u);   // off * ceil(16 / off)
                    const uint32_t sa = wa + (uint32_t)sbase0;
                    left = rcx_lz4_rounds_ovl(sa, ovl0 ? wa + (uint32_t)li_m - D : sa, ovl0 ? 0u : 0xffffffffu, ovl0 ? off : 16u, wa + (uint32_t)li_m, Mc,
                                              (uint32_t)dep, (uint32_t)(dep >> 32), __ballot(pending0), prog0);
                } else
                {   // (no self-overlapping lane in this batch -- the other loop takes those; only the A/B build without it has any: kept out here, the portable loop below copies them)
                    const uint64_t hold = (CUT & 0x800) ? (__ballot(pending0) & __ballot(ovl0)) : 0ull;
                    left = rcx_lz4_rounds(wa + (uint32_t)sbase0, wa + (uint32_t)li_m, Mc, (uint32_t)dep, (uint32_t)(dep >> 32),
                                          __ballot(pending0) & ~hold, prog0) | hold;
                }
                pending0 = RCX_INV_BALLOT(left);
            }
#endif
            auto rounds = [&](auto conv) __attribute__((always_inline)) {
                constexpr bool CONV = decltype(conv)::value;
                int32_t sbase = sbase0;
                bool ovl = ovl0;
                bool pending = pending0;
                uint32_t prog = prog0, r = 0;
                for (;;) {
                    const unsigned long long pm = __ballot(pending);
                    if (!pm) break;
                    if (PROF5) pw[9] += 1;
                    const bool ready = pending && (pm & dep) == 0;
                    const bool rn = ready && !ovl;
                    uint32_t v0, v1, v2 = 0, v3 = 0, nv;
                    if (__ballot(rn)) {
                        const int32_t rb = rn ? sbase + (int32_t)prog : 0;
                        // five ALIGNED dwords + v_alignbyte: an unaligned ds_read_b64 holds the CU's LDS pipe ~24 cycles
                        // (SQ_LDS_UNALIGNED_STALL was 19 % of the kernel's cycles), an aligned pair ~4
                        B::lds_load16u(wb_, rb, v0, v1, v2, v3);
                        nv = rn ? (Mc - prog < 16u ? Mc - prog : 16u) : 0u;
                    } else {
                        const bool ro = ready && ovl;
                        uint32_t b[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) { b[u] = wb_[ro ? sbase + (int32_t)r : 0]; r = !ro ? r : (r + 1 == off) ? 0u : r + 1; }
                        v0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                        v1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
                        nv = ro ? (M - prog < 8u ? M - prog : 8u) : 0u;
                    }
                    rcx_wave_sync();
                    RCX_LDS_STORE16(wb_ + li_m + (int32_t)prog, v0, v1, v2, v3, nv);
                    rcx_wave_sync();
                    prog += nv;
                    pending = pending && prog < Mc;
                    if (CONV && ovl && prog >= 16u) {
                        ovl = false;
                        sbase = li_m - (int32_t)(off * (((uint32_t)(0x11111111223357F0ull >> (4u * (off & 15u))) & 15u) + 1u));
                    }
                }
            };
            if (CUT & 1) {} else if (RCX_DEGUARD && !(CUT & 64) && !__ballot(pending0)) {}      // (the hand-written loop left nothing: no second look)
            else if (__ballot(ovl0 && M > 16u)) rounds(std::true_type{}); else rounds(std::false_type{});
            if (!LITLDS) RCX_SETPRIO_FLUSH(young); else if (RCX_INF_ROUNDS_PRIO) __builtin_amdgcn_s_setprio(0);
        }
        V5P_ADD(7);
        this->oend = RCX_U(oend0 + T);
        if (!(CUT & 16)) this->flush(this->oend, false); else this->gflush = this->oend & ~15u;
        if (!LITLDS && RCX_FLUSH_PRIO != RCX_EXEC_PRIO) RCX_SETPRIO_EXEC(young);
        V5P_ADD(8);
        if (PROF5) pw[10] += 1;
#undef V5P_ADD
        return 0;
    }

    __device__ void run_executor(int32_t* st_out, uint32_t* len_out)
    {
        this->lane = rcx_lane();
        const unsigned lane = this->lane;
        this->init_window();
        int st = RCX_OK;
        uint32_t tail = 0;
        for (;;) {
            uint64_t te0 = PROF5 ? (uint64_t)__builtin_readcyclecounter() : 0;
            while (RCX_U(ring->head) == tail) __builtin_amdgcn_s_sleep(4);
            if (PROF5) { pw[0] += (uint64_t)__builtin_readcyclecounter() - te0; pw[2] += 1; }
            rcx_wave_sync();
            const RCX_LDS_AS Slot* sl = &ring->slot[tail % NSLOT];
            typename B::Batch bt;
            bt.ns = (int)RCX_U(sl->hdr[0]); bt.why = (int)RCX_U(sl->hdr[1]); bt.perr = (int)RCX_U(sl->hdr[2]);
            bt.gL = RCX_U(sl->hdr[3]); bt.gM = RCX_U(sl->hdr[4]); bt.goff = RCX_U(sl->hdr[5]); bt.gsrc = RCX_U(sl->hdr[6]);
            bt.gnext = 0;
            const uint64_t d = *(const RCX_LDS_AS uint64_t*)sl->desc[lane];
            const uint32_t w0 = (uint32_t)d, w1 = (uint32_t)(d >> 32);
            rcx_wave_sync();
            tail++;
            if (lane == 0) ring->tail = tail;             // the slot is in registers: hand it back
            int lo = 0, e = 0;
            while (lo < bt.ns && !e) e = emit5(bt.ns, lo, w0, w1);
            if (e) { st = e; break; }
            if (bt.why == B::STAGE_) continue;
            if (this->after_batch(bt, st)) break;
        }
        if (st && lane == 0) ring->abort_ = 1;
        if (!st) this->flush(this->oend, true);
        *st_out = st;
        *len_out = st ? 0u : this->oend;
    }
};

template <int CB, int TC = 2560, int HH = 2048, bool PROF5 = false, int SB = 32>
__global__ __launch_bounds__(128, 8) void k_lz4_decode_v5(rcx_kargs a, int only_status = 0)
{
    typedef Lz4V5<CB, TC, HH, PROF5, SB> S;
    const uint64_t tk0 = PROF5 ? (uint64_t)__builtin_readcyclecounter() : 0;
    __shared__ __align__(16) uint8_t s_cbuf[CB + 96];
    __shared__ __align__(16) uint8_t s_wbuf[S::WBUF5 + 16];     // + 16: the last staging slot is read one dword past its end
    __shared__ uint32_t s_epos[64];
    __shared__ __align__(16) typename S::Ring s_ring;
    __shared__ uint32_t s_lmap[64];
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    if (only_status && a.status[b] != only_status) return;       // second pass over the blocks another kernel handed back
    if (threadIdx.x == 0) { RCX_LDS_AS typename S::Ring* r0 = (RCX_LDS_AS typename S::Ring*)&s_ring; r0->head = 0; r0->tail = 0; r0->abort_ = 0; }   // (through the LDS pointer: volatile stores through the generic one are flat)
    __syncthreads();
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    S s;
    s.in = a.in_base + a.in_off[b];
    s.n = (uint32_t)a.in_len[b];
    s.out = a.out_base + a.out_off[b];
    const uint64_t cap64 = a.out_cap[b];
    s.cap = cap64 > 0xffffffffull ? 0xffffffffu : (uint32_t)cap64;
    s.cbuf = s_cbuf;
    s.wb_ = s_wbuf;
    s.epos = s_epos;
    s.ring = (RCX_LDS_AS typename S::Ring*)&s_ring;
    s.lmap = s_lmap;
    if (role == 0) {
        s.run_parser();
        if (PROF5 && a.scratch && (threadIdx.x & 63u) == 0) {          // [0..3] parser: ring-full wait, total, posts
            uint64_t* q = (uint64_t*)a.scratch + (size_t)b * 8;
            q[0] = s.pw[0]; q[1] = (uint64_t)__builtin_readcyclecounter() - tk0; q[2] = s.pw[2];
        }
        return;
    }
    int32_t st; uint32_t olen;
    __builtin_amdgcn_s_setprio(RCX_EXEC_PRIO);
    s.run_executor(&st, &olen);
    if (PROF5 && a.scratch && (threadIdx.x & 63u) == 0) {              // [4..7] executor: ring-empty wait, total, batches
        uint64_t* q = (uint64_t*)a.scratch + (size_t)b * 8 + 4;
        q[0] = s.pw[0]; q[1] = (uint64_t)__builtin_readcyclecounter() - tk0; q[2] = s.pw[2];
    }
    if ((threadIdx.x & 63u) == 0) {
        a.status[b] = st;
        a.out_len[b] = olen;
        if (a.in_used) a.in_used[b] = s.n;
    }
}
