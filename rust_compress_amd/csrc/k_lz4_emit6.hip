// k_lz4_emit6.hip -- the EXECUTOR of the headline path: LZ4 block decode, one plain batch of up to 64 sequences as one straight
// instruction stream (reference: BlockDecoder::decode, src/lz4.rs:67-140).
//
// Where the headline kernel lives (k_lz4_decode_v8<...>, BASELINE configs[1]):
//   k_lz4_decode_v8.hip  the kernel, the PARSER wave (segment-parallel token walk, batches of 64 entries, the ring) and the executor wave's
//                        loop (run_executor8: take a batch from the ring, emit6 or emit5, drain);
//   k_lz4_emit6.hip      THIS FILE: Lz4X6::emit6, what the executor wave runs for a plain batch -- nearly all of a text's -- and its store;
//   k_lz4_decode_v5.hip  Lz4V5::emit5: every other batch (literal runs above 32 bytes never get here; matches that overlap themselves,
//                        the block's first and last tokens, anything malformed), status for status as the reference;
//   k_lz4_decode_v4.hip  Lz4V4: the output window in LDS (make_room, flush / drain, the wave-wide copies of long sequences).
#pragma once
#include "rcx_dev.h"
// (Lz4V5 / Lz4V4 -- k_lz4_decode_v5.hip, k_lz4_decode_v4.hip -- come first in the translation unit: tu_lz4.hip)

#ifndef RCX_X6_MROW
#define RCX_X6_MROW 9                        /* dwords a row of the mask table (8: rows 32 bytes apart, read as b128 + b32 -- the k-th dwords of rows e, e + 4, e + 8 ... share a bank) */
#endif
#ifndef RCX_X6_RR
#define RCX_X6_RR 3                         /* emit6's redirection rounds (pointer doubling) before the copy rounds: 0 / 1 / 2 / 3 / 4 / 5 / 6: 0.587 / 0.485 / 0.4516 / 0.4468 /
                                               0.4516 / 0.4585 / 0.4668 ms (emit5 keeps B::RR = 2; tests/wavesim: 2.95 copy rounds a batch at 2) */
#endif
#ifndef RCX_X6_MODE
#define RCX_X6_MODE 1                    /* A/B: 0 = no emit6 at all (round 5's kernel), 2 = its LDS layout and the parser's flag, but every batch through emit5 */
#endif
#ifndef RCX_X6_ROUNDS_ASM
#define RCX_X6_ROUNDS_ASM 1              /* 0: emit6's copy rounds as hipcc compiles them (A/B) */
#endif
#ifndef RCX_V8_STAT
#define RCX_V8_STAT(slot, v) ((void)0)                       /* the wave simulator counts batches through emit6 / emit5 here */
#endif

// The copy rounds of Lz4V8::emit6 as ISA (the loop of rcx_lz4_rounds, k_lz4_decode_v5.hip, with emit6's store): every round the lanes whose
// producers are done -- no pending lane among `dep` -- read six aligned dwords from `source - a` (a: the destination's misalignment), shift them
// by the source's own and store up to 16 bytes as five ds_mskor_b32 under the masks of table entry a + n.  exec holds the ready lanes only: an LDS
// atomic under an empty mask would still take its turn at the dwords it shares with its neighbours.  24 vector + 4 scalar instructions a round
// (hipcc's loop for the same source: 44 + 6, every lane at every LDS instruction).
//   s0: LDS address of source - a; d4: LDS address of the destination's frame (4-byte aligned); mc: bytes to copy (0: none); tb: LDS address of
//   the mask table + 32 a; fix: ~0 << 8 a.  Returns the lanes still pending when no lane is ready: none, unless a lane was handed a dependency on
//   itself (emit6 does that to a run, which it fills itself).
#if !defined(RCX_NO_ROUNDS_ASM) && !defined(RCX_NO_MSKOR_ASM)
__device__ __forceinline__ uint64_t rcx_lz4_rounds6(uint32_t s0, uint32_t d4, uint32_t mc, uint32_t dep_lo, uint32_t dep_hi, uint64_t pend, uint32_t tb, uint32_t fix)
{
    uint32_t t0, t1, nv, sa, sh, da, r0, r1, r2, r3, r4, r5, m0, m1, m2, m3, m4, prog = 0;
    uint64_t sT, sF;
    asm volatile(
        "s_mov_b64 vcc, %[pend]\n\t"
        "L_top_%=:\n\t"
        "v_and_b32_e32 %[t0], vcc_lo, %[dlo]\n\t"
        "v_and_b32_e32 %[t1], vcc_hi, %[dhi]\n\t"
        "v_or_b32_e32 %[t0], %[t0], %[t1]\n\t"
        "v_cmp_eq_u32_e64 %[sT], 0, %[t0]\n\t"
        "s_and_b64 exec, %[sT], vcc\n\t"                          // the ready lanes; scc: any
        "s_cbranch_scc0 L_out_%=\n\t"
        "v_add_u32_e32 %[sa], %[s0], %[prog]\n\t"
        "v_sub_u32_e32 %[nv], %[mc], %[prog]\n\t"
        "v_and_b32_e32 %[t0], -4, %[sa]\n\t"
        "ds_read_b32 %[r0], %[t0]\n\t"
        "ds_read_b32 %[r1], %[t0] offset:4\n\t"
        "ds_read_b32 %[r2], %[t0] offset:8\n\t"
        "ds_read_b32 %[r3], %[t0] offset:12\n\t"
        "ds_read_b32 %[r4], %[t0] offset:16\n\t"
        "ds_read_b32 %[r5], %[t0] offset:20\n\t"
        "v_min_u32_e32 %[nv], 16, %[nv]\n\t"
        "v_mad_u32_u24 %[t1], %[nv], %[mrow], %[tb]\n\t"
        "ds_read_b32 %[m0], %[t1]\n\t"
        "ds_read_b32 %[m1], %[t1] offset:4\n\t"
        "ds_read_b32 %[m2], %[t1] offset:8\n\t"
        "ds_read_b32 %[m3], %[t1] offset:12\n\t"
        "ds_read_b32 %[m4], %[t1] offset:16\n\t"
        "v_and_b32_e32 %[sh], 3, %[sa]\n\t"
        "v_add_u32_e32 %[da], %[d4], %[prog]\n\t"
        "v_add_u32_e32 %[prog], %[prog], %[nv]\n\t"
        "v_cmp_ge_u32_e64 %[sF], %[prog], %[mc]\n\t"              // of this round's lanes, those that are done with it
        "s_waitcnt lgkmcnt(9)\n\t"
        "v_alignbyte_b32 %[r0], %[r1], %[r0], %[sh]\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        "v_alignbyte_b32 %[r1], %[r2], %[r1], %[sh]\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_alignbyte_b32 %[r2], %[r3], %[r2], %[sh]\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        "v_alignbyte_b32 %[r3], %[r4], %[r3], %[sh]\n\t"
        "s_waitcnt lgkmcnt(5)\n\t"
        "v_alignbyte_b32 %[r4], %[r5], %[r4], %[sh]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_and_b32_e32 %[m0], %[m0], %[fix]\n\t"
        "v_and_b32_e32 %[r0], %[r0], %[m0]\n\t"
        "ds_mskor_b32 %[da], %[m0], %[r0]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_and_b32_e32 %[r1], %[r1], %[m1]\n\t"
        "ds_mskor_b32 %[da], %[m1], %[r1] offset:4\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_and_b32_e32 %[r2], %[r2], %[m2]\n\t"
        "ds_mskor_b32 %[da], %[m2], %[r2] offset:8\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_and_b32_e32 %[r3], %[r3], %[m3]\n\t"
        "ds_mskor_b32 %[da], %[m3], %[r3] offset:12\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_and_b32_e32 %[r4], %[r4], %[m4]\n\t"
        "ds_mskor_b32 %[da], %[m4], %[r4] offset:16\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_andn2_b64 vcc, vcc, %[sF]\n\t"                          // scc: a lane still pending
        "s_cbranch_scc1 L_top_%=\n\t"
        "L_out_%=:\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_mov_b64 %[pend], vcc\n\t"
        : [pend] "+s"(pend), [prog] "+v"(prog), [t0] "=&v"(t0), [t1] "=&v"(t1), [nv] "=&v"(nv), [sa] "=&v"(sa), [sh] "=&v"(sh), [da] "=&v"(da),
          [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5),
          [m0] "=&v"(m0), [m1] "=&v"(m1), [m2] "=&v"(m2), [m3] "=&v"(m3), [m4] "=&v"(m4), [sT] "=&s"(sT), [sF] "=&s"(sF)
        : [s0] "v"(s0), [d4] "v"(d4), [mc] "v"(mc), [dlo] "v"(dep_lo), [dhi] "v"(dep_hi), [tb] "v"(tb), [fix] "v"(fix), [mrow] "s"(4u * (uint32_t)RCX_X6_MROW)
        : "vcc", "scc", "memory");
    return pend;
}
#define RCX_HAVE_ROUNDS6 1
#endif

// Base: the Lz4V5 instance (window, ring-less executor state, lmap); PROF: phase timers into Base::pw (A/B builds)
template <class Base, bool PROF8>
struct Lz4X6 : Base {
    typedef Base P5;
    typedef typename Base::B B;
    // ------------------------------------------------------------------------------------------ round 6: the PLAIN batch, straight line
    // Nearly every batch of a text is plain: no match that overlaps itself, no token within 40 bytes of the
    // block's end, nothing malformed at first sight (the parser decides that much per token and says so in the slot header, batch8()).
    // For such a batch the executor runs emit6(): the same steps as Lz4V5::emit5 -- scan, validation, loads, producer bitmap, two
    // redirection levels, literal / gathered-match stores, copy rounds, drain -- but as ONE straight instruction stream in which every
    // lane executes every instruction: a lane that has nothing to store stores under an empty mask.  What makes that possible is the
    // store: the bytes [a, a + n) of a 4-byte ALIGNED frame of five dwords go out as five ds_mskor_b32 (LDS atomic: mem = (mem & ~mask) |
    // data; lanes that share a boundary dword are served one after the other), the masks read from a 20-entry table in LDS (entry e =
    // a + n: the bytes below e; the low a bytes of the first dword are cleared with one shift), and the data arrives already IN PLACE:
    // literals and gathered matches are loaded from `address - a` (global loads take any alignment), window matches are read as six
    // aligned dwords from `source - a` and shifted by its misalignment.  No exec mask is touched, no v_cmpx, no branch but the round loop's.
    // benchmarks/micro/lds_mskor_store.hip: a store of <= 16 bytes takes 204 cycles alone (the exec-narrowing byte stores: 612) and
    // 26 cycles of a CU at sixteen waves (43), exact across lanes that share dwords.
    // Anything else -- and whatever emit6 finds at second sight: output that does not fit, an offset beyond the output, a gathered match
    // that starts in the block's first four bytes -- goes through emit5 as before.
    RCX_LDS_AS uint32_t* mtab = nullptr;                              // LDS: 20 entries of 8 dwords (5 used)

    __device__ __forceinline__ void mtab_init()
    {
        for (uint32_t i = this->lane; i < 20u * (uint32_t)RCX_X6_MROW; i += 64u) {
            const int32_t e = (int32_t)(i / (uint32_t)RCX_X6_MROW), k = (int32_t)(i % (uint32_t)RCX_X6_MROW);
            int32_t t = e - 4 * k; t = t < 0 ? 0 : t > 4 ? 4 : t;
            mtab[i] = (k < 5 && t) ? (0xffffffffu >> (8 * (4 - t))) : 0u;
        }
        rcx_wave_sync();
    }
    // bytes [a, a + n) of the frame {w0..w4} to the 4-byte aligned LDS address fp (a < 4, n <= 16); NW: dwords of the frame that can hold any
    template <int NW>
    __device__ __forceinline__ void store_frame(uint8_t* fp, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t a, uint32_t n)
    {
#ifdef RCX_NO_MSKOR_ASM
        const uint32_t w[5] = {w0, w1, w2, w3, w4};
        for (uint32_t j = a; j < a + n; j++) fp[j] = (uint8_t)(w[j >> 2] >> (8u * (j & 3u)));
#else
        const RCX_LDS_AS uint32_t* te = mtab + (uint32_t)RCX_X6_MROW * (a + n);
#if RCX_X6_MROW == 8
        const rcx_u32x4 mv = *(const RCX_LDS_AS rcx_u32x4*)te;
#else
        rcx_u32x4 mv; mv[0] = te[0]; mv[1] = te[1]; mv[2] = te[2]; mv[3] = te[3];       // (rows of nine dwords: no two rows' k-th dwords in one bank)
#endif
        const uint32_t m0 = mv[0] & (0xffffffffu << (8u * a));
        const uint32_t fa = (uint32_t)(uintptr_t)fp;                 // (low half of a generic LDS pointer = the LDS byte address)
        if (NW >= 5) {
            const uint32_t m4 = te[4];
            asm volatile("ds_mskor_b32 %0, %1, %2\n\tds_mskor_b32 %0, %3, %4 offset:4\n\tds_mskor_b32 %0, %5, %6 offset:8\n\tds_mskor_b32 %0, %7, %8 offset:12\n\tds_mskor_b32 %0, %9, %10 offset:16"
                         :: "v"(fa), "v"(m0), "v"(w0 & m0), "v"(mv[1]), "v"(w1 & mv[1]), "v"(mv[2]), "v"(w2 & mv[2]), "v"(mv[3]), "v"(w3 & mv[3]), "v"(m4), "v"(w4 & m4) : "memory");
        } else {
            asm volatile("ds_mskor_b32 %0, %1, %2\n\tds_mskor_b32 %0, %3, %4 offset:4\n\tds_mskor_b32 %0, %5, %6 offset:8\n\tds_mskor_b32 %0, %7, %8 offset:12"
                         :: "v"(fa), "v"(m0), "v"(w0 & m0), "v"(mv[1]), "v"(w1 & mv[1]), "v"(mv[2]), "v"(w2 & mv[2]), "v"(mv[3]), "v"(w3 & mv[3]) : "memory");
        }
#endif
    }

    // One plain batch (ns entries, lane i holds entry i's word; p0: where the batch's first token starts).  Returns false -- nothing
    // touched but the window's position (make_room) -- when the batch has to go through emit5 after all.
    __device__ __forceinline__ bool emit6(int ns, uint32_t w1raw, uint32_t p0, int young)
    {
        const unsigned lane = this->lane;
        const uint8_t* in = this->in; uint8_t* out = this->out; uint8_t* wb_ = this->wb_;
        const uint32_t cap = this->cap, n = this->n;
        uint64_t tq_ = PROF8 ? (uint64_t)__builtin_readcyclecounter() : 0;
#define X6P_ADD(slot) do { RCX_MARK("emit6_" #slot); if (PROF8) { const uint64_t t1_ = (uint64_t)__builtin_readcyclecounter(); this->pw[slot] += t1_ - tq_; tq_ = t1_; } } while (0)
        this->make_room(B::TCAP);
        X6P_ADD(11);
        const uint32_t oend0 = this->oend;
        const uint32_t w1 = (int)lane < ns ? w1raw : 0u;
        const uint32_t L = w1 & 0x7fu, M = (w1 >> 8) & 0xffu, off = w1 >> 16;
        const uint32_t lx = L >= 15u ? 1u : 0u;                         // (a literal run of 15..32 bytes has one extension byte)
        const uint32_t hop = ((int)lane < ns && !(w1 & 0x80u)) ? 3u + L + lx + (M >= 19u ? 1u : 0u) : 0u;     // (the second half of a split match is no token)
        const uint32_t len = L + M;
        const uint32_t sc = rcx_wave_incl_scan(len | (hop << 16));   // output positions and literal sources in one scan (64 x 64 and 64 x 37: no carry between the halves)
        const uint32_t T = RCX_U(__builtin_amdgcn_readlane(sc, 63)) & 0xffffu;
        if (T > (uint32_t)B::TCAP || cap - oend0 < T + 64u) return false;          // (the gathers below read up to 36 bytes from where a match starts)
        const uint32_t lincl = sc & 0xffffu;
        const uint32_t ostart = oend0 + lincl - len;
        const uint32_t mdst = ostart + L;
        const uint32_t src = p0 + (sc >> 16) - hop + 1u + lx;
        const uint32_t slo = mdst - off;
        const uint32_t re = this->rlo_eff();
        const bool isfar = slo < re;                                  // source drained and slid out of the window (no match: slo = mdst >= re)
        if (__ballot(off > mdst || (isfar && slo < 4u))) return false;
        X6P_ADD(4);

        const int32_t lbase = this->lbase;
        const int32_t li_o = (int32_t)ostart - lbase, li_m = li_o + (int32_t)L;
        const uint32_t a1 = (uint32_t)li_o & 3u, a2 = (uint32_t)li_m & 3u;          // (the window is 16-byte aligned)
        // ---- loads first, every lane: the literals' frames (a token of a plain batch lies >= 40 bytes in front of the block's end and
        // >= 3 bytes behind its start), the gathered match's two frames
        uint32_t ql = src - a1; { const uint32_t qm = n - 36u; ql = ql < qm ? ql : qm; }
        const bool lit2 = __ballot(L > 16u) != 0;
        const rcx_u32x4 g0 = *(const rcx_u32x4_u*)(in + ql);
        const rcx_u32x4 g1 = *(const rcx_u32x4_u*)(in + ql + 16);
        uint32_t g2 = 0;
        if (lit2) g2 = *(const rcx_u32_u*)(in + ql + 32);
        rcx_u32x4 f0, f1; uint32_t f2;
        RCX_NOINIT4(f0); RCX_NOINIT4(f1); f2 = 0;
#ifndef RCX_X6_NOFAR                     /* (attribution: no gathers -- wrong results on purpose) */
        if (re) {
#else
        if (false) {
#endif
            const uint32_t qf = isfar ? slo - a2 : 0u;
            f0 = *(const rcx_u32x4_u*)(out + qf);
            f1 = *(const rcx_u32x4_u*)(out + qf + 16);
            f2 = *(const rcx_u32_u*)(out + qf + 32);
        }
        RCX_SETPRIO_ROUND(young);

        // ---- which entries produce [slo, shi): a bitmap of the entries' first bytes + a running count per word (Lz4V5::emit5)
        const uint32_t shi = (slo + M < mdst) ? slo + M : mdst;
        bool inb = M != 0u && !isfar && shi > oend0;
        uint32_t S = off;
        uint32_t ka, kb;
        {
            uint32_t* const lmap = this->lmap;
            uint8_t* const lcnt = (uint8_t*)(lmap + P5::LMW);
            lmap[lane] = 0;                                          // (all 64 words: the counts' bytes among them)
            rcx_wave_sync();
            const uint32_t rel = lincl - len;                        // (a lane beyond the batch marks the byte behind it: nobody looks there)
            atomicOr(&lmap[rel >> 5], 1u << (rel & 31u));
            rcx_wave_sync();
            const uint32_t pc = (uint32_t)__popc(lmap[lane]);
            const uint32_t ex = rcx_wave_incl_scan(pc) - pc;
            lcnt[lane] = (uint8_t)ex;
            rcx_wave_sync();
            int32_t ra = (int32_t)(slo - oend0), rb = (int32_t)(shi - oend0) - 1;
            ra = ra > 0 ? ra : 0; rb = rb > 0 ? rb : 0;
            uint32_t ca = lcnt[ra >> 5], ma = lmap[ra >> 5], cb2 = lcnt[rb >> 5], mb = lmap[rb >> 5];
            RCX_SETTLE4(ca, ma, cb2, mb);
            ka = (ca + (uint32_t)__popc(ma & (0xffffffffu >> (31u - ((uint32_t)ra & 31u)))) - 1u) & 63u;
            kb = (cb2 + (uint32_t)__popc(mb & (0xffffffffu >> (31u - ((uint32_t)rb & 31u)))) - 1u) & 63u;
        }
        unsigned long long dep = 0;
        {
            const uint32_t pmd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ka << 2), (int)((isfar || off < M) ? 0xffffffffu : mdst));
            const uint32_t prod = (inb && ka == kb && slo >= pmd && off >= M) ? ka : 64u;
            RCX_V8_STAT(28, inb && ka != kb); RCX_V8_STAT(29, inb && ka == kb && slo < pmd); RCX_V8_STAT(30, inb && ka == kb && slo >= pmd && off < M);   // (simulator: why a source is not redirected: spans entries / starts in literals / overlaps itself)
            RCX_V8_STAT(31, inb && ka == kb && shi <= pmd && pmd != 0xffffffffu);
            uint32_t st = prod | (ka << 7) | (kb << 13) | ((uint32_t)inb << 19);
#pragma unroll
            for (int rr = 0; rr < RCX_X6_RR; rr++) {
                const bool has = (st & 64u) == 0u;
                const uint32_t j = has ? (st & 63u) : lane;
                const uint32_t Sj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)S);
                const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)st);
                const uint32_t Sn = S + Sj;
                const bool ok = has && mdst - Sn >= re && Sn <= mdst;
                S = ok ? Sn : S;
                st = ok ? pk : (st | 64u);
            }
            ka = (st >> 7) & 63u; kb = (st >> 13) & 63u; inb = (st >> 19) & 1u;
            const unsigned long long upto = (kb >= 63) ? ~0ull : ((2ull << kb) - 1ull);
            dep = inb ? (upto & ~((1ull << ka) - 1ull) & ((1ull << lane) - 1ull)) : 0ull;
        }

        X6P_ADD(5);
        // ---- literals and gathered matches: registers -> their place in the window
        {
            const uint32_t n1 = L < 16u ? L : 16u;
            uint8_t* fp = wb_ + (li_o - (int32_t)a1);
            if (n1) store_frame<5>(fp, g0[0], g0[1], g0[2], g0[3], g1[0], a1, n1);
            if (lit2) { if (L > 16u) store_frame<5>(fp + 16, g1[0], g1[1], g1[2], g1[3], g2, a1, L - n1); }
        }
        if (re) {
            const uint32_t mf = isfar ? M : 0u, n1 = mf < 16u ? mf : 16u;
            uint8_t* fp = wb_ + (li_m - (int32_t)a2);
            if (n1) store_frame<5>(fp, f0[0], f0[1], f0[2], f0[3], f1[0], a2, n1);
            if (mf > 16u) store_frame<5>(fp + 16, f1[0], f1[1], f1[2], f1[3], f2, a2, mf - n1);
        }
        rcx_wave_sync();
        X6P_ADD(6);

        // ---- window matches: copy rounds, 16 bytes per ready lane.  A RUN (a match that overlaps itself: offset < 16 and < its length; one
        // batch in seven of a text holds one, and all of those went through emit5 at first) waits like any lane for the producers of its
        // period -- the bytes [match - offset, match) -- and then fills itself in ONE go by period doubling: off, 2 off, 4 off ... (whole periods) up to 16
        // bytes a step, a frame each, read from what the lane has just written (a lane's LDS operations are performed in order).  The
        // hand-written loop never finds such a lane ready (it is handed a dependency on itself): it returns with the runs and whatever waits
        // for them still pending, the runs whose periods stand are filled, and the loop goes on.
        {
            const int32_t dfr = li_m - (int32_t)a2;
            const bool ovl = M != 0u && off < 16u && off < M;          // (never a gathered match: its source ends where it begins)
            const int32_t sfr = isfar ? dfr : (int32_t)(mdst - S) - lbase - (int32_t)a2;     // the source, as far back from a dword boundary as the destination (a gathered match reads itself: every lane reads)
            const uint32_t Mc = isfar ? 0u : M;
            auto fill_run = [&]() __attribute__((always_inline)) {      // (called by the lanes of ready runs only: no collective inside)
                uint32_t done = 0, back = off;                          // back: a whole number of periods, all of them standing behind the write position
                while (done < M) {
                    uint32_t c = back < M - done ? back : M - done;
                    c = c < 16u ? c : 16u;
                    const int32_t d = li_m + (int32_t)done;
                    const uint32_t ar = (uint32_t)d & 3u;
                    const int32_t sb = d - (int32_t)back - (int32_t)ar;
                    const uint32_t* q = (const uint32_t*)(wb_ + (sb & ~3));
                    const uint32_t sh = (uint32_t)sb & 3u;
                    const uint32_t r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4], r5 = q[5];
                    store_frame<5>(wb_ + (d - (int32_t)ar), RCX_ALIGNBYTE(r1, r0, sh), RCX_ALIGNBYTE(r2, r1, sh), RCX_ALIGNBYTE(r3, r2, sh),
                                   RCX_ALIGNBYTE(r4, r3, sh), RCX_ALIGNBYTE(r5, r4, sh), ar, c);
                    done += c;
                    if (2u * back <= off + done) back *= 2u;
                }
            };
            unsigned long long pm = __ballot(Mc != 0u);
#ifdef RCX_HAVE_ROUNDS6
            if (RCX_X6_ROUNDS_ASM) {
                const uint32_t wa = (uint32_t)(uintptr_t)wb_;      // (low half of a generic LDS pointer = the LDS byte address)
                const unsigned long long depx = dep | (ovl ? 1ull << lane : 0ull);
                for (;;) {
                    pm = rcx_lz4_rounds6(wa + (uint32_t)sfr, wa + (uint32_t)dfr, Mc, (uint32_t)depx, (uint32_t)(depx >> 32), pm,
                                         (uint32_t)(uintptr_t)mtab + a2 * (4u * (uint32_t)RCX_X6_MROW), 0xffffffffu << (8u * a2));
                    if (!pm) break;
                    const bool fill = ovl && ((pm >> lane) & 1ull) && (pm & dep) == 0ull;
                    if (fill) fill_run();
                    rcx_wave_sync();
                    pm &= ~__ballot(fill);
                }
            } else
#endif
            {
            uint32_t prog = 0;
            RCX_V8_STAT(16, Mc != 0u); RCX_V8_STAT(17, Mc != 0u && dep != 0ull); RCX_V8_STAT(18, Mc > 16u);       // (simulator: window matches, those that wait, the long ones)
            uint32_t nr_ = 0;
            while (pm) {
                nr_++; RCX_V8_STAT(14, lane == 0);
                const bool ready = ((pm >> lane) & 1ull) && (pm & dep) == 0ull;
                const bool fill = ready && ovl;
                const uint32_t left = Mc - prog;
                const uint32_t nv = (ready && !ovl) ? (left < 16u ? left : 16u) : 0u;
                const int32_t sb = sfr + (int32_t)prog;
                const uint32_t* q = (const uint32_t*)(wb_ + (sb & ~3));
                const uint32_t sh = (uint32_t)sb & 3u;
                uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0;
                if (nv) { r0 = q[0]; r1 = q[1]; r2 = q[2]; r3 = q[3]; r4 = q[4]; r5 = q[5]; }
                rcx_wave_sync();
                // (only the lanes that copy: an LDS atomic under an empty mask still takes its turn at the dwords it shares with its neighbours)
                if (nv) store_frame<5>(wb_ + (dfr + (int32_t)prog), RCX_ALIGNBYTE(r1, r0, sh), RCX_ALIGNBYTE(r2, r1, sh), RCX_ALIGNBYTE(r3, r2, sh),
                               RCX_ALIGNBYTE(r4, r3, sh), RCX_ALIGNBYTE(r5, r4, sh), a2, nv);
                if (fill) fill_run();                                  // (a ready run's period stands: nobody writes it any more, nobody reads the run yet)
                rcx_wave_sync();
                prog += fill ? Mc : nv;
                pm = __ballot(prog < Mc);
            }
            RCX_V8_STAT(20 + (nr_ < 7u ? nr_ : 7u), lane == 0);
            }
        }
        X6P_ADD(7);
        RCX_SETPRIO_FLUSH(young);
        this->oend = RCX_U(oend0 + T);
        this->flush(this->oend, false);
        if (RCX_FLUSH_PRIO != RCX_EXEC_PRIO) RCX_SETPRIO_EXEC(young);
        X6P_ADD(8);
        if (PROF8) this->pw[10] += 1;
#undef X6P_ADD
        RCX_V8_STAT(8, lane == 0);
        return true;
    }
};
