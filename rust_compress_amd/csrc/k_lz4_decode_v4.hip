// k_lz4_decode_v4.hip -- LZ4 block decode, fourth iteration of the batched design (see k_lz4_decode.hip
// for v1-v3 and the reference citations: BlockDecoder::decode, src/lz4.rs:67-140).
//
// What changed against v3, each item from the rocprof PMC profile of v3 on MI355X (40.6 instructions and
// 458 cycles per sequence, 40 % of the wave's time in memory waits, 9.2 K static instructions):
//   * one emit() call site (state machine) instead of nine inlined copies: the hot loop fits the I-cache;
//   * the token walk is 9 scalar instructions per hop: v_readlane of the hop distance, a visited-position
//     bit mask (s_lshl/s_or), and ONE vector compaction per 64-byte window (mbcnt + ds_write of the token
//     positions into an LDS list) instead of a v_cndmask per hop;
//   * matches older than the LDS window are fetched with one 16-byte HBM load per lane into a per-lane
//     32-byte LDS staging slot and then ride the same byte-copy rounds as window matches (v3 spent
//     ~130 instructions per batch extracting those bytes);
//   * a copy round reads up to 32 bytes per lane into registers, then writes them: one LDS round trip
//     per dependency level instead of one per 8-byte chunk;
//   * sequences longer than the per-lane caps but <= 1 KiB are copied by the whole wave INSIDE the LDS
//     window (periodic source for overlapping matches): run-heavy data no longer drains/refills through HBM;
//   * the window slides lazily (only when the next batch would not fit), which also lengthens the history.
#include "rcx_dev.h"

#ifndef RCX_INV_BALLOT
#define RCX_INV_BALLOT(m) __builtin_amdgcn_inverse_ballot_w64(m)      // uniform 64-bit mask -> per-lane predicate, no VALU
#endif
#ifndef RCX_ALIGNBYTE
#define RCX_ALIGNBYTE(hi, lo, sh) __builtin_amdgcn_alignbyte((hi), (lo), (sh))   // ({hi,lo} >> 8*sh) & 0xffffffff
#endif
#ifndef RCX_LDS_AS
#define RCX_LDS_AS __attribute__((address_space(3)))
#endif
#ifndef RCX_GLOBAL_AS
#define RCX_GLOBAL_AS __attribute__((address_space(1)))
#endif
#ifndef RCX_U
#define RCX_U(x) ((uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x)))
#endif

// PROF: phase timers (s_memtime via __builtin_readcyclecounter) accumulated per block into scratch; A/B builds only
#define LZ4P_T0() uint64_t t0_ = PROF ? (uint64_t)__builtin_readcyclecounter() : 0
#define LZ4P_ADD(slot) do { if (PROF) { const uint64_t t1_ = (uint64_t)__builtin_readcyclecounter(); prof[slot] += t1_ - t0_; t0_ = t1_; } } while (0)

// Sums of a 16-byte chunk for Adler-32 by position (zlib over the inflate front end, ADLER below): A = sum x_j, J = sum j * x_j
#ifndef RCX_SAD_U8
#define RCX_SAD_U8(a, c) __builtin_amdgcn_sad_u8((a), 0u, (c))                  // c + the four bytes of a
#define RCX_UDOT4(a, w, c) __builtin_amdgcn_udot4((a), (w), (c), false)         // c + sum of a's bytes times w's bytes
#endif
__device__ __forceinline__ void rcx_adler_chunk(const rcx_u32x4 v, uint32_t rel, uint32_t& tA, uint32_t& tR)
{
    uint32_t A = RCX_SAD_U8(v[0], 0u); A = RCX_SAD_U8(v[1], A); A = RCX_SAD_U8(v[2], A); A = RCX_SAD_U8(v[3], A);
    uint32_t J = RCX_UDOT4(v[0], 0x03020100u, 0u); J = RCX_UDOT4(v[1], 0x07060504u, J); J = RCX_UDOT4(v[2], 0x0b0a0908u, J); J = RCX_UDOT4(v[3], 0x0f0e0d0cu, J);
    tA += A; tR += rel * A + J;
}

// ADLER (the inflate front end's zlib streams): every byte that leaves for global memory -- the window's drain, the wave-wide
// copies -- is also summed for adler::State32 (src/checksum/adler.rs:29-44) on its way out, by POSITION:
//   S0 = sum x_i,  S1 = sum i * x_i  (mod 65521)   =>   a = 1 + S0,  b = N + N * S0 - S1   for a stream of N bytes,
// so that the chunks may arrive in any order and k_adler32's second pass over the output (a quarter of the launch's HBM traffic)
// is not needed.  A lane sums its own chunks relative to the call's first byte (u32 is enough: a call moves a few KiB, or is
// cut into such pieces), the call ends with two wave sums and scalar arithmetic on the wave-uniform S0 / S1.
// MIRROR (the host-memory entry point of the LZ4 decoder, rcx_api.hip): every byte that leaves for global memory is stored a second
// time at the same offset of `out2` -- the caller's page-locked host buffer, written over PCIe while the block is still being
// decoded -- so that no device-to-host copy follows the launch.  The copy in HBM stays: old matches are gathered from it.
#ifndef RCX_V4_RR
#define RCX_V4_RR 2
#endif
template <int CB, bool PROF = false, int TC = 2560, int HH = 2048, bool ADLER = false, bool MIRROR = false>
struct Lz4V4 {
    uint64_t prof[16];
    // ADLER: S0, S1 mod 65521 live in two words of LDS (`adp`, zeroed by the kernel), not in registers: the inflate kernel has
    // neither a VGPR nor an SGPR to spare (80 + 106 at six waves per SIMD; two more wave-uniform values alive across the whole
    // kernel spilled 44 dwords to scratch and config 3 went from 9.9 to 11.1 ms).  A wave's LDS operations are performed in order.
    uint32_t* adp = nullptr;
    __device__ __forceinline__ void ad_commit(uint32_t from0, uint32_t tA, uint32_t tR)
    {
        const uint32_t sA = RCX_U(__builtin_amdgcn_readlane(rcx_wave_incl_scan(tA), 63));
        const uint32_t sR = RCX_U(__builtin_amdgcn_readlane(rcx_wave_incl_scan(tR), 63));
        const uint32_t a = sA % 65521u;
        const uint32_t s0 = adp[0], s1 = adp[1];
        rcx_wave_sync();
        if (lane == 0) {
            adp[0] = (s0 + a) % 65521u;
            adp[1] = (s1 + (from0 % 65521u) * a % 65521u + sR % 65521u) % 65521u;
        }
        rcx_wave_sync();
    }
    // Adler-32 of the `total` bytes that have left
    __device__ __forceinline__ uint32_t ad_result(uint32_t total) const
    {
        const uint32_t N = total % 65521u, s0 = adp[0], s1 = adp[1];
        const uint32_t a = (1u + s0) % 65521u;
        const uint32_t b = (uint32_t)(((uint64_t)N + (uint64_t)N * s0 + 65521ull - s1) % 65521ull);
        return (b << 16) | a;
    }
    static constexpr int H = HH;                   // history kept when the window slides (2048 for LZ4)
    static constexpr int LCAP = 32, MCAP = 64;     // per-lane caps of a batched sequence
    static constexpr int WINMAX = 22 * (14 + MCAP);// most output one 64-byte token window can add (22 tokens)
    static constexpr int TCAP = TC;                // output bytes per batch (2560 for LZ4; the inflate front end uses less LDS)
    static constexpr int SOLO = TC < 1024 ? TC : 1024;   // wave-cooperative in-window copy up to this many bytes
    static constexpr int LIN = H + 16 + TCAP;
    static constexpr int STAGE = LIN + 64;         // 64 bytes of read slack, then 64 lanes x MCAP bytes of old-match staging
    static constexpr int WBUF = STAGE + 64 * MCAP;
    static constexpr int RH = 128;
    static constexpr int RR = RCX_V4_RR;           // redirection rounds (pointer doubling) before the copy rounds
    static constexpr int MARGIN = LCAP + 16;
    static constexpr uint32_t FLAG = 0x80000000u;
    static_assert(SOLO <= TCAP && (CB % 1024) == 0 && RH >= 2 * MCAP && (STAGE % 16) == 0, "geometry");

    const uint8_t* in; uint8_t* out; uint32_t n, cap;
    uint8_t* out2 = nullptr;               // MIRROR: same misalignment (mod 256) as `out`
    uint32_t mflush = 0;                   // MIRROR: what has left for out2
    uint8_t* cbuf; uint8_t* wb_;          // wb_: [lin | slack | staging]
    uint32_t* epos;                        // token positions of the batch (LDS, 64 entries)
    int32_t cbase; uint32_t cend;
    int32_t lbase;
    uint32_t oend, gflush, rlo, omis;
    unsigned lane;

    // 4 / 16 bytes at any offset of a 16-byte aligned LDS buffer as aligned dword reads + v_alignbyte (an unaligned ds_read
    // holds the LDS pipe ~24 cycles).  The address is formed as base + (idx & ~3), never through an integer, so that the
    // loads stay ds_read (a pointer rebuilt from an integer is a flat pointer: flat_load).  The buffer must be readable one
    // dword past the bytes asked for.
    static __device__ __forceinline__ uint32_t lds_load4u(const uint8_t* base, int32_t idx)
    {
        const uint32_t* q = (const uint32_t*)(base + (idx & ~3));
        return RCX_ALIGNBYTE(q[1], q[0], (uint32_t)idx & 3u);
    }
    static __device__ __forceinline__ void lds_load16u(const uint8_t* base, int32_t idx, uint32_t& v0, uint32_t& v1, uint32_t& v2, uint32_t& v3)
    {
        const uint32_t* q = (const uint32_t*)(base + (idx & ~3));
        const uint32_t sh = (uint32_t)idx & 3u;
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
        v0 = RCX_ALIGNBYTE(d1, d0, sh); v1 = RCX_ALIGNBYTE(d2, d1, sh); v2 = RCX_ALIGNBYTE(d3, d2, sh); v3 = RCX_ALIGNBYTE(d4, d3, sh);
    }
    __device__ __forceinline__ int32_t lbase_for(uint32_t pos) const
    {
        return (((int32_t)pos - H + (int32_t)omis) & ~15) - (int32_t)omis;
    }
    __device__ __forceinline__ uint32_t rlo_eff() const
    {
        const uint32_t lb0 = lbase > 0 ? (uint32_t)lbase : 0u;
        return rlo > lb0 ? rlo : lb0;
    }

    __device__ void stage(uint32_t cur)
    {
        const uint32_t inmis = (uint32_t)((uintptr_t)in & 15u);
        cbase = (int32_t)RCX_U((int32_t)((cur + inmis) & ~15u) - (int32_t)inmis);
#pragma unroll
        for (int r = 0; r < CB / 1024; r++) {
            const int j = r * 64 + (int)lane;
            const int32_t pos = cbase + 16 * j;
            if (pos >= 0 && (uint32_t)pos + 16 <= n) {
                *(rcx_u32x4*)(cbuf + 16 * j) = *(const rcx_u32x4*)(in + pos);
            } else {
                for (int t = 0; t < 16; t++) {
                    const int32_t q = pos + t;
                    cbuf[16 * j + t] = (q >= 0 && (uint32_t)q < n) ? in[q] : (uint8_t)0;
                }
            }
        }
        const uint32_t lim = (uint32_t)(cbase + CB);
        cend = RCX_U(n < lim ? n : lim);
        rcx_wave_sync();
    }

    __device__ __forceinline__ uint32_t peek(uint32_t q) const
    {
        const int32_t idx = (int32_t)q - cbase;
        uint32_t v;                                             // two loads kept apart: a select between the two POINTERS is a flat load
        if (idx >= 0 && q < cend) v = ((const RCX_LDS_AS uint8_t*)cbuf)[idx]; else v = ((const RCX_GLOBAL_AS uint8_t*)in)[q];
        return RCX_U(v);
    }

    // make room for `need` more output bytes in the window; slides by a multiple of 16, keeps H bytes
    __device__ void make_room(uint32_t need)
    {
        if ((int32_t)oend - lbase + (int32_t)need + 16 <= LIN) return;
        const int32_t nb = (int32_t)RCX_U(lbase_for(oend));
        const int32_t delta = nb - lbase;
        if (delta <= 0) return;
        const int32_t keep = ((int32_t)oend - nb + 15) & ~15;
        if (delta < LIN) {
            constexpr int NK = (H + 32 + 1023) / 1024;         // keep <= H + 31: all reads, then all writes
            rcx_u32x4 v[NK];
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int32_t j = 1024 * k + 16 * (int32_t)lane;
                v[k] = rcx_u32x4{0, 0, 0, 0};
                if (j < keep && j + delta + 16 <= LIN + 64) v[k] = *(const rcx_u32x4*)(wb_ + j + delta);
            }
            rcx_wave_sync();
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int32_t j = 1024 * k + 16 * (int32_t)lane;
                if (j < keep && j + delta + 16 <= LIN + 64) *(rcx_u32x4*)(wb_ + j) = v[k];
            }
            rcx_wave_sync();
        }
        lbase = nb;
    }

    __device__ void flush(uint32_t to, bool final)
    {
        uint32_t from = gflush;
        if (to <= from) { if (MIRROR && final) mirror_to(to, true); return; }
        if (!ADLER && !final && ((omis + from) & 15u) == 0u) {     // the usual drain: gflush is 16-byte aligned after the block's first one
            const uint32_t nch = (to - from) >> 4;
            #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (uint32_t c = RCX_VGPR(lane); c < nch; c += 64) {
                const uint32_t p = from + 16 * c;
                *(rcx_u32x4*)(out + p) = *(const rcx_u32x4*)(wb_ + ((int32_t)p - lbase));
            }
            gflush = RCX_U(from + nch * 16);
            if (MIRROR) mirror_to(gflush, false);
            return;
        }
        const uint32_t mis = (uint32_t)((uintptr_t)(out + from) & 15u);
        uint32_t head = mis ? 16u - mis : 0u;
        if (head > to - from) head = final ? to - from : 0u;
        if (mis && head == 0 && !final) return;
        const uint32_t from0 = from;
        uint32_t tA = 0, tR = 0;                                   // ADLER: this lane's sums, positions relative to from0 (a drain is < 4 KiB)
        if (head) {
            if (lane < head) { const uint8_t x = wb_[(int32_t)(from + lane) - lbase]; out[from + lane] = x; if (ADLER) { tA += x; tR += lane * (uint32_t)x; } }
            from += head;
        }
        const uint32_t nch = (to - from) >> 4;
        #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (uint32_t c = RCX_VGPR(lane); c < nch; c += 64) {      // (laundered: `out + 16 * lane` hoisted to the kernel's top is a spill)
            const uint32_t p = from + 16 * c;
            const rcx_u32x4 v = *(const rcx_u32x4*)(wb_ + ((int32_t)p - lbase));
            *(rcx_u32x4*)(out + p) = v;
            if (ADLER) rcx_adler_chunk(v, p - from0, tA, tR);
        }
        from += nch * 16;
        if (final) {
            const uint32_t tail = to - from;
            if (lane < tail) { const uint8_t x = wb_[(int32_t)(from + lane) - lbase]; out[from + lane] = x; if (ADLER) { tA += x; tR += (from + lane - from0) * (uint32_t)x; } }
            from = to;
        }
        if (ADLER) ad_commit(from0, tA, tR);
        gflush = RCX_U(from);
        if (MIRROR) mirror_to(final ? to : gflush, final);
    }

    // MIRROR: the drained bytes [mflush, to) once more, from the window to the caller's page-locked buffer -- in whole 256-byte lines of
    // that buffer (a drain moves ~700 bytes from wherever the last one stopped: written as they come they cross PCIe as partial lines,
    // 43 GB/s against the link's 55), the rest with the next drain: fewer than 256 bytes stay behind, and the window keeps H >= 768
    // bytes of history.  final: everything, byte-exact (the block's end, or a wave-wide copy follows that writes both places itself).
    __device__ void mirror_to(uint32_t to, bool final)
    {
        uint32_t from = mflush;
        const uint32_t A = (uint32_t)((uintptr_t)out2 & 255u);
        uint32_t lim = to;
        if (!final) { const uint32_t al = (A + to) & ~255u; if (al <= A + from) return; lim = al - A; }
        if (lim <= from) return;
        uint32_t head = (0u - (A + from)) & 15u;
        if (head > lim - from) head = lim - from;
        if (head) {
            if (lane < head) out2[from + lane] = wb_[(int32_t)(from + lane) - lbase];
            from += head;
        }
        const uint32_t nch = (lim - from) >> 4;
        #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (uint32_t c = RCX_VGPR(lane); c < nch; c += 64) {
            const uint32_t p = from + 16 * c;
            *(rcx_u32x4*)(out2 + p) = *(const rcx_u32x4*)(wb_ + ((int32_t)p - lbase));
        }
        from += nch * 16;
        if (lim > from) {                                          // (final only: a line is a multiple of 16)
            if (lane < lim - from) out2[from + lane] = wb_[(int32_t)(from + lane) - lbase];
            from = lim;
        }
        mflush = RCX_U(from);
    }

    __device__ void repair()
    {
        lbase = (int32_t)RCX_U(lbase_for(oend));
        rlo = oend > (uint32_t)RH ? oend - RH : 0u;
        rcx_wave_sync();
        #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (uint32_t p = rlo + lane; p < oend; p += 64) wb_[(int32_t)p - lbase] = out[p];
        rcx_wave_sync();
    }

    __device__ void wide_literals(uint32_t src, uint32_t len)
    {
        uint8_t* d = out + oend;
        uint8_t* d2 = MIRROR ? out2 + oend : nullptr;
        const uint8_t* s = in + src;
        const uint32_t mis = (uint32_t)((uintptr_t)d & 15u);
        uint32_t head = mis ? 16u - mis : 0u;
        if (head > len) head = len;
        uint32_t tA = 0, tR = 0;                                   // ADLER: relative to oend; the chunk loop commits every 2 KiB (u32 sums)
        if (lane < head) { const uint8_t x = s[lane]; d[lane] = x; if (MIRROR) d2[lane] = x; if (ADLER) { tA += x; tR += lane * (uint32_t)x; } }
        if (ADLER) ad_commit(oend, tA, tR);
        const uint32_t nb = (len - head) >> 4;
        if (!ADLER) {
            #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (uint32_t c = RCX_VGPR(lane); c < nb; c += 64) {
                if (MIRROR) { const rcx_u32x4 v = *(const rcx_u32x4_u*)(s + head + 16 * c); *(rcx_u32x4*)(d + head + 16 * c) = v; *(rcx_u32x4*)(d2 + head + 16 * c) = v; }
                else *(rcx_u32x4*)(d + head + 16 * c) = *(const rcx_u32x4_u*)(s + head + 16 * c);
            }
        } else {
            #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (uint32_t c0 = 0; c0 < nb; c0 += 128) {           // 128 chunks a step: two per lane, positions below 2 KiB
                tA = 0; tR = 0;
                for (uint32_t c = c0 + lane; c < nb && c < c0 + 128; c += 64) {
                    const rcx_u32x4 v = *(const rcx_u32x4_u*)(s + head + 16 * c);
                    *(rcx_u32x4*)(d + head + 16 * c) = v;
                    if (MIRROR) *(rcx_u32x4*)(d2 + head + 16 * c) = v;
                    rcx_adler_chunk(v, 16 * (c - c0), tA, tR);
                }
                ad_commit(oend + head + 16 * c0, tA, tR);
            }
        }
        const uint32_t done = head + nb * 16;
        tA = 0; tR = 0;
        if (lane < len - done) { const uint8_t x = s[done + lane]; d[done + lane] = x; if (MIRROR) d2[done + lane] = x; if (ADLER) { tA += x; tR += lane * (uint32_t)x; } }
        if (ADLER) ad_commit(oend + done, tA, tR);
    }

    __device__ void wide_match(uint32_t off, uint32_t len)
    {
        uint32_t e = off, d = oend, rem = len;
        while (rem) {
            uint32_t C = rem < e ? rem : e;
            if (C > 1024) C = 1024;
            const uint32_t i0 = 16 * RCX_VGPR(lane);
            uint32_t tA = 0, tR = 0;                               // ADLER: relative to d, C <= 1024
            if (i0 + 16 <= C) {
                const rcx_u32x4 v = *(const rcx_u32x4_u*)(out + d - e + i0);
                *(rcx_u32x4_u*)(out + d + i0) = v;
                if (MIRROR) *(rcx_u32x4_u*)(out2 + d + i0) = v;
                if (ADLER) rcx_adler_chunk(v, i0, tA, tR);
            } else if (i0 < C) {
                for (uint32_t t = i0; t < C; t++) { const uint8_t x = out[d - e + t]; out[d + t] = x; if (MIRROR) out2[d + t] = x; if (ADLER) { tA += x; tR += t * (uint32_t)x; } }
            }
            if (ADLER) ad_commit(d, tA, tR);
            rcx_wave_sync();
            d += C; rem -= C;
            if (C == e && e < 1024) e *= 2;
        }
    }

    // one long sequence, copied by the whole wave inside the LDS window.  Returns 0 or an rcx_status.
    __device__ int solo(uint32_t lit_src, uint32_t L, uint32_t off, uint32_t M)
    {
        if (L > cap - oend) return RCX_E_OUTPUT_TOO_SMALL;
        const uint32_t mdst = oend + L;
        if (M) {
            if (off == 0 || off > mdst) return RCX_E_MALFORMED;
            if (M > cap - mdst) return RCX_E_OUTPUT_TOO_SMALL;
            // a source that starts before the window must lie entirely in drained HBM; else take the wide path
            const uint32_t slo0 = mdst - off;
            const uint32_t need = off < M ? off : M;
            if (slo0 < rlo_eff() && slo0 + need > gflush) return -2;
        }
        make_room(L + M);
        const int32_t li = (int32_t)oend - lbase;
        #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (uint32_t i = lane; i < L; i += 64) wb_[li + (int32_t)i] = in[lit_src + i];
        rcx_wave_sync();
        if (M) {
            const uint32_t slo = mdst - off;
            const int32_t lm = li + (int32_t)L;
            if (slo >= rlo_eff()) {                                   // source in the window (may overlap itself)
                const int32_t ls = (int32_t)slo - lbase;
                if (off >= M) {
                    #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                    for (uint32_t i = lane; i < M; i += 64) wb_[lm + (int32_t)i] = wb_[ls + (int32_t)i];
                } else {                                              // periodic: only finished bytes are read
                    #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                    for (uint32_t i = lane; i < M; i += 64) wb_[lm + (int32_t)i] = wb_[ls + (int32_t)(i % off)];
                }
            } else if (off >= M) {                                    // drained long ago: HBM -> window
                #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (uint32_t i = lane; i < M; i += 64) wb_[lm + (int32_t)i] = out[slo + i];
            } else {
                #pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (uint32_t i = lane; i < M; i += 64) wb_[lm + (int32_t)i] = out[slo + (i % off)];
            }
            rcx_wave_sync();
        }
        oend = RCX_U(oend + L + M);
        flush(oend, false);
        return 0;
    }

    // Largest lane k with ostart[k] <= x (ostart is non-decreasing over the lanes): binary search with ds_bpermute.
    // The probe lo + step never exceeds 63, and the byte address (lane * 4) is what is carried.
    __device__ __forceinline__ uint32_t lane_of(uint32_t ostart, uint32_t x) const
    {
        uint32_t lo4 = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const uint32_t c4 = lo4 + 4u * (uint32_t)step;
            const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c4, (int)ostart);
            lo4 = v <= x ? c4 : lo4;
        }
        return lo4 >> 2;
    }

    // Emits entries [lo, hi) of the batch, hi = ns unless their output exceeds TCAP bytes (then the prefix that
    // fits; the caller comes back for the rest), and advances lo to hi.  Returns 0 or an rcx_status.
    __device__ int emit(int ns, int& lo, uint32_t s_L, uint32_t s_M, uint32_t s_off, uint32_t s_src)
    {
        LZ4P_T0();
        make_room(TCAP);
        LZ4P_ADD(1);
        bool act = (int)lane >= lo && (int)lane < ns;
        uint32_t L = 0, M = 0, off = 0, src = (uint32_t)cbase;
        if (act) {
            const uint32_t e = epos[lane];
            if (e & FLAG) { L = s_L; M = s_M; off = s_off; src = s_src; }
            else {
                const uint32_t t = cbuf[(int32_t)e - cbase];
                L = t >> 4; M = (t & 15u) + 4u; src = e + 1;
                const uint32_t w = lds_load4u(cbuf, (int32_t)(src + L) - cbase);   // offset lo, hi, extension byte
                off = w & 0xffffu;
                if (M == 19u) M += (w >> 16) & 0xffu;                    // one match-length extension byte (< 255, lz4.rs:112-122)
            }
        }
        const uint32_t len = L + M;
        const uint32_t incl = rcx_wave_incl_scan(len);
        uint32_t T = RCX_U(__builtin_amdgcn_readlane(incl, 63));
        int hi = ns;
        if (T > (uint32_t)TCAP) {                                     // rare: take the prefix that fits (an entry is <= 96 bytes)
            hi = lo + (int)__popcll(__ballot(act && incl <= (uint32_t)TCAP));
            act = act && (int)lane < hi;
            if (!act) { L = 0; M = 0; off = 0; }
            T = RCX_U(__builtin_amdgcn_readlane(incl, hi - 1));
        }
        lo = hi;
        const uint32_t oend0 = oend;
        const uint32_t ostart = oend0 + incl - len;
        const uint32_t mdst = ostart + L;
        int err = 0;
        if (act) {
            if (L > cap - ostart || ostart > cap) err = RCX_E_OUTPUT_TOO_SMALL;
            else if (M && (off == 0 || off > mdst)) err = RCX_E_MALFORMED;
            else if (M && M > cap - mdst) err = RCX_E_OUTPUT_TOO_SMALL;
        }
        const unsigned long long bad = __ballot(err != 0);
        if (bad) return __builtin_amdgcn_readlane(err, __ffsll(bad) - 1);
        LZ4P_ADD(2);

        const uint32_t re = rlo_eff();
        const int32_t li_o = (int32_t)ostart - lbase;
        const int32_t li_m = li_o + (int32_t)L;
        const uint32_t slo = mdst - off;
        const uint32_t shi = (slo + M < mdst) ? slo + M : mdst;
        const bool isfar = M && slo < re;                            // source drained and slid out of the window

        // ---- old matches: up to four 16-byte HBM gathers per lane into the lane's staging slot
        rcx_u32x4 f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0}, f2 = {0, 0, 0, 0}, f3 = {0, 0, 0, 0};
        const unsigned long long anyfar = __ballot(isfar);
        const bool far16 = isfar && (uint64_t)slo + (uint32_t)MCAP <= (uint64_t)cap;
        const bool farb = isfar && !far16;                           // within MCAP bytes of the slot end: byte loads
        if (anyfar) {
            if (far16) {
                f0 = *(const rcx_u32x4_u*)(out + slo);
                if (M > 16) f1 = *(const rcx_u32x4_u*)(out + slo + 16);
                if (M > 32) f2 = *(const rcx_u32x4_u*)(out + slo + 32);
                if (M > 48) f3 = *(const rcx_u32x4_u*)(out + slo + 48);
            }
        }

        // ---- literals: compressed window -> output window, 16 bytes per step (exec-narrowing byte stores)
        {
            const int32_t sb = (int32_t)src - cbase;
            for (uint32_t i0 = 0; __ballot(i0 < L); i0 += 16) {
                const bool on = i0 < L;
                const int32_t rb = on ? sb + (int32_t)i0 : 0;
                uint32_t x0, x1, x2, x3;
                lds_load16u(cbuf, rb, x0, x1, x2, x3);
                const uint32_t nv = on ? (L - i0 < 16u ? L - i0 : 16u) : 0u;
                RCX_LDS_STORE16(wb_ + li_o + (int32_t)i0, x0, x1, x2, x3, nv);
            }
        }
        if (anyfar) {                                                // gathered bytes -> the lane's staging slot
            uint8_t* sl = wb_ + STAGE + MCAP * (int32_t)lane;
            if (far16) {
                *(rcx_u32x4*)(sl) = f0; *(rcx_u32x4*)(sl + 16) = f1;
                if (M > 32) { *(rcx_u32x4*)(sl + 32) = f2; *(rcx_u32x4*)(sl + 48) = f3; }
            }
            for (uint32_t i = 0; __ballot(farb && i < M); i++)
                if (farb && i < M) sl[i] = out[slo + i];
        }
        rcx_wave_sync();
        LZ4P_ADD(3);

        // ---- matches.  Producer lanes of [slo, shi) inside this batch = lanes ka..kb; copy when none is pending.
        if (__ballot(M != 0)) {
            // Chains are the rule in text (the reference's compressor always points at the MOST RECENT occurrence,
            // so the 5th " the " of a batch copies from the 4th, which copies from the 3rd ...): a match whose
            // source lies wholly inside ONE earlier match of the batch is redirected to that match's own source
            // (bytes x of match j equal bytes x - S_j), with pointer doubling: RR rounds cut 2^RR levels.
            unsigned long long dep = 0;
            bool inb = M && !isfar && shi > oend0;
            uint32_t S = off;                                         // current shift: the source is [mdst - S, +M)
            if (__ballot(inb)) {
                uint32_t ka = lane_of(ostart, slo > oend0 ? slo : oend0);
                uint32_t kb = lane_of(ostart, shi > oend0 ? shi - 1 : oend0);
                // producer usable for redirection: its match is in the window and does not overlap itself
                const uint32_t pmd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ka << 2), (int)((isfar || off < M) ? 0xffffffffu : mdst));
                uint32_t prod = (inb && ka == kb && slo >= pmd && off >= M) ? ka : 64u;
#pragma unroll
                for (int rr = 0; rr < RR; rr++) {
                    if (!__ballot(prod < 64u)) break;
                    const uint32_t j = prod < 64u ? prod : lane;
                    const uint32_t Sj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)S);
                    const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)(prod | (ka << 7) | (kb << 13) | ((uint32_t)inb << 19)));
                    if (prod < 64u) {
                        if (mdst - S - Sj >= re && S + Sj <= mdst) {
                            S += Sj; prod = pk & 127u; ka = (pk >> 7) & 63u; kb = (pk >> 13) & 63u; inb = (pk >> 19) & 1u;
                        } else prod = 64u;                            // would leave the window: wait for the producer instead
                    }
                }
                if (inb) {
                    const unsigned long long upto = (kb >= 63) ? ~0ull : ((2ull << kb) - 1ull);
                    dep = upto & ~((1ull << ka) - 1ull) & ((1ull << lane) - 1ull);
                }
            }
            LZ4P_ADD(4);
            int32_t sbase = isfar ? STAGE + MCAP * (int32_t)lane : (int32_t)(mdst - S) - lbase;
            // A chunk of 16 bytes only needs its source to be 16 bytes behind: matches with off >= 16 ride the
            // plain path even when they overlap themselves; off < 16 reads a periodic source 8 bytes at a time
            // until 16 bytes stand, then copies from off * ceil(16 / off) >= 16 bytes behind (same period) on the plain path.
            bool ovl = M && !isfar && off < 16u && off < M;
            const bool anyovl = __ballot(ovl) != 0;
            bool pending = M != 0;
            uint32_t prog = 0, r = 0;
            for (;;) {
                const unsigned long long pm = __ballot(pending);
                if (!pm) break;
                if (PROF) prof[12] += 1;
                const bool ready = pending && (pm & dep) == 0;
                const bool rn = ready && !ovl;
                uint32_t v0, v1, v2 = 0, v3 = 0, nv;
                if (__ballot(rn)) {
                    const int32_t rb = rn ? sbase + (int32_t)prog : 0;
                    lds_load16u(wb_, rb, v0, v1, v2, v3);
                    nv = rn ? (M - prog < 16u ? M - prog : 16u) : 0u;
                } else {                                   // self-overlapping short-period matches, batched up
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
                pending = pending && prog < M;
                if (anyovl && ovl && prog >= 16u) {                      // off * ceil(16 / off) bytes behind: same period, no overlap
                    ovl = false;
                    sbase = li_m - (int32_t)(off * (((uint32_t)(0x11111111223357F0ull >> (4u * (off & 15u))) & 15u) + 1u));
                }
            }
        }
        LZ4P_ADD(5);
        oend = RCX_U(oend0 + T);
        flush(oend, false);
        LZ4P_ADD(6);
        if (PROF) prof[10] += 1;
        return 0;
    }

    // What stopped a batch, and the one long sequence that goes with SOLO_/WIDE_ (or the error with ERR_).
    enum { GO = 0, STAGE_ = 1, SOLO_ = 2, WIDE_ = 3, END_ = 4, ERR_ = 5 };
    struct Batch { int ns, why, perr; uint32_t gL, gM, goff, gsrc, gnext; };

    // Parse side: walk tokens from `cur`, filling epos[0..ns) (and the s_* fields of general-path entries).
    __device__ __forceinline__ Batch collect(uint32_t& cur, uint32_t& s_L, uint32_t& s_M, uint32_t& s_off, uint32_t& s_src)
    {
        int ns = 0;
        int why = GO;
        int perr = 0;
        uint32_t gL = 0, gM = 0, goff = 0, gsrc = 0, gnext = 0;
        // positions from which a token may not fit the staged bytes any more (re-stage first); none once the tail is staged
        const uint32_t stage_lim = RCX_U(cend < n ? cend - (uint32_t)MARGIN : 0xffffffffu);
        const uint32_t fast_lim = RCX_U(cend >= 20 ? cend - 20 : 0);
        for (;;) {
            cur = RCX_U(cur); ns = (int)RCX_U(ns);
            if (cur >= n) { why = END_; break; }
            if (cur > stage_lim) { why = STAGE_; break; }
            if (ns >= 64) break;
            // register window: hop distance of the candidate token at cur+lane (128 = general path) and the
            // output bytes it produces.  A match-length nibble of 15 followed by ONE extension byte that keeps
            // the match within MCAP stays on the vector path (second, dependent LDS read).
            uint64_t tw0_ = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
            const uint32_t q = cur + lane;
            uint32_t dv;
            {                                                     // every lane reads (the staging buffer has slack); selects, no EXEC games
                const int32_t qi = (int32_t)q - cbase;
                const uint32_t t = cbuf[qi];
                const uint32_t L = t >> 4, M = t & 15u;
                const uint32_t x = cbuf[qi + 3 + (int32_t)L];
                const bool ext = M == 15u;
                const bool ok = q < fast_lim && L != 15u && (!ext || x <= (uint32_t)(MCAP - 19));
                dv = ok ? (ext ? 4u : 3u) + L : 128u;
            }
            uint32_t rel = 0;
            uint64_t vis = 0;
            if (PROF) { dv = RCX_U(dv) * 0u + dv; const uint64_t t = (uint64_t)__builtin_readcyclecounter(); prof[14] += t - tw0_; tw0_ = t; }
            RCX_HOP_WALK(dv, rel, vis);                       // the serial token chain
            if (PROF) { const uint64_t t = (uint64_t)__builtin_readcyclecounter(); prof[15] += t - tw0_; }
            if (PROF) prof[13] += 1;
            bool general = false;                             // stopped at a token that needs the general path?
            if (__builtin_expect(rel >= 128, 0)) { general = true; rel -= 128; vis &= ~(1ull << rel); }
            // compaction: a visited position p is a token start -> epos[ns + rank]; the batch's output is capped
            bool mark = RCX_INV_BALLOT(vis);
            bool full = false;
            if (vis) {
                uint32_t rank = (uint32_t)__popcll(vis & ((1ull << lane) - 1ull));
                if (ns + (int)__popcll(vis) > 64) {               // keep the first 64 - ns tokens, emit, resume at the first rejected one
                    const unsigned long long rej = __ballot(mark && ns + (int)rank >= 64);
                    rel = (uint32_t)__ffsll(rej) - 1u;
                    vis &= (1ull << rel) - 1ull;
                    mark = mark && lane < rel;
                    general = false; full = true;
                }
                if (mark) epos[ns + (int)rank] = q;
                ns += (int)__popcll(vis);
            }
            if (full) { cur += rel; break; }
            cur += rel;
            if (!general) continue;                           // window ran out: next window
            if (cur >= n) { why = END_; break; }
            if (cur > stage_lim) { why = STAGE_; break; }

            // ---- general path for the token at `cur`
            const uint32_t t = peek(cur);
            uint32_t p = cur + 1;
            uint32_t L = t >> 4;
            if (L == 15) {
                for (;;) {
                    if (p >= n) { perr = RCX_E_MALFORMED; break; }
                    const uint32_t x = peek(p); p++;
                    L += x;
                    if (x != 255) break;
                }
                if (perr) { why = ERR_; break; }
            }
            const uint32_t lit_src = p;
            if (L > n - p) { perr = RCX_E_MALFORMED; why = ERR_; break; }
            p += L;
            uint32_t M = 0, off = 0;
            if (p != n) {
                if (n - p < 2) { perr = -1; gL = L; why = ERR_; break; }     // literals copy, then the offset read panics
                off = peek(p) | (peek(p + 1) << 8);
                p += 2;
                M = t & 15u;
                if (M == 15) {
                    for (;;) {
                        if (p >= n) { perr = -1; gL = L; break; }            // same order: literal overflow first
                        const uint32_t x = peek(p); p++;
                        M += x;
                        if (x != 255) break;
                    }
                    if (perr) { why = ERR_; break; }
                }
                M += 4;
            }
            const bool eligible = L <= (uint32_t)LCAP && M <= (uint32_t)MCAP && lit_src + L <= cend && (int32_t)lit_src >= cbase;
            if (eligible && ns < 64) {
                if (lane == 0) epos[ns] = FLAG;
                s_L = ((int)lane == ns) ? L : s_L;
                s_M = ((int)lane == ns) ? M : s_M;
                s_off = ((int)lane == ns) ? off : s_off;
                s_src = ((int)lane == ns) ? lit_src : s_src;
                ns++;
                cur = p;
            } else if (eligible) {
                break;                                        // batch full: emit, then this token is parsed again
            } else {
                gL = L; gM = M; goff = off; gsrc = lit_src; gnext = p;
                why = (L + M <= (uint32_t)SOLO) ? SOLO_ : WIDE_;
                break;
            }
        }
        Batch bt; bt.ns = ns; bt.why = why; bt.perr = perr; bt.gL = gL; bt.gM = gM; bt.goff = goff; bt.gsrc = gsrc; bt.gnext = gnext;
        return bt;
    }

    // Execute side: what follows a batch.  Returns 0 to go on, 1 when the block is finished (st holds the status).
    __device__ __forceinline__ int after_batch(const Batch& bt, int& st)
    {
        int why = bt.why;
        if (why == GO || why == STAGE_) return 0;              // (the usual batch: one test, not four)
        if (why == END_) return 1;
        if (why == ERR_) { st = bt.perr > 0 ? bt.perr : ((bt.gL > cap - oend) ? RCX_E_OUTPUT_TOO_SMALL : RCX_E_MALFORMED); return 1; }
        if (why == SOLO_) {
            const int e = solo(bt.gsrc, bt.gL, bt.goff, bt.gM);
            if (e == -2) why = WIDE_;
            else if (e) { st = e; return 1; }
        }
        if (why == WIDE_) {
            flush(oend, true);
            if (bt.gL > cap - oend) { st = RCX_E_OUTPUT_TOO_SMALL; return 1; }
            if (bt.gL) { wide_literals(bt.gsrc, bt.gL); oend += bt.gL; }
            if (bt.gM) {
                if (bt.goff == 0 || bt.goff > oend) { st = RCX_E_MALFORMED; return 1; }
                if (bt.gM > cap - oend) { st = RCX_E_OUTPUT_TOO_SMALL; return 1; }
                rcx_wave_sync();
                wide_match(bt.goff, bt.gM);
                oend += bt.gM;
            }
            oend = RCX_U(oend);
            gflush = oend;
            if (MIRROR) mflush = oend;
            repair();
        }
        return 0;
    }

    __device__ void init_window()
    {
        omis = (uint32_t)((uintptr_t)out & 15u);
        oend = 0; gflush = 0; rlo = 0; mflush = 0;
        lbase = (int32_t)RCX_U(lbase_for(0));
    }

    __device__ void run(int32_t* st_out, uint32_t* len_out)
    {
        lane = rcx_lane();
        init_window();
        int st = RCX_OK;
        uint32_t cur = 0;
        if (n) stage(0); else { cbase = 0; cend = 0; }
        uint32_t s_L = 0, s_M = 0, s_off = 0, s_src = 0;      // fields of general-path entries (per lane)
        if (PROF) for (int i = 0; i < 16; i++) prof[i] = 0;
        for (;;) {
            LZ4P_T0();
            const Batch bt = collect(cur, s_L, s_M, s_off, s_src);
            rcx_wave_sync();
            LZ4P_ADD(0);
            if (PROF) prof[11] += (uint64_t)bt.ns;
            int lo = 0, e = 0;
            while (lo < bt.ns && !e) e = emit(bt.ns, lo, s_L, s_M, s_off, s_src);
            if (e) { st = e; break; }
            if (PROF) t0_ = (uint64_t)__builtin_readcyclecounter();
            if (bt.why == STAGE_) { stage(cur); LZ4P_ADD(7); continue; }
            if (after_batch(bt, st)) break;
            if (bt.why == SOLO_ || bt.why == WIDE_) cur = bt.gnext;
            LZ4P_ADD(8);
        }
        if (!st) flush(oend, true);
        *st_out = st;
        *len_out = st ? 0u : oend;
    }
};

template <int CB, int WAVES, bool PROF = false>
__global__ __launch_bounds__(64 * WAVES) void k_lz4_decode_v4(rcx_kargs a)
{
    typedef Lz4V4<CB, PROF> S;
    __shared__ __align__(16) uint8_t s_cbuf[WAVES][CB + 96];
    __shared__ __align__(16) uint8_t s_wbuf[WAVES][S::WBUF + 16];   // + 16: lds_load16u reads one dword past the last staging slot
    __shared__ uint32_t s_epos[WAVES][64];
    const unsigned w = threadIdx.x >> 6;
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));
    if (b >= a.nblocks) return;
    S s;
    s.in = a.in_base + a.in_off[b];
    s.n = (uint32_t)a.in_len[b];
    s.out = a.out_base + a.out_off[b];
    const uint64_t cap64 = a.out_cap[b];
    s.cap = cap64 > 0xffffffffull ? 0xffffffffu : (uint32_t)cap64;
    s.cbuf = s_cbuf[w];
    s.wb_ = s_wbuf[w];
    s.epos = s_epos[w];
    int32_t st; uint32_t olen;
    s.run(&st, &olen);
    if ((threadIdx.x & 63u) == 0) {
        a.status[b] = st;
        a.out_len[b] = olen;
        if (a.in_used) a.in_used[b] = s.n;
        if (PROF && a.scratch) for (int i = 0; i < 16; i++) ((uint64_t*)a.scratch)[(size_t)b * 16 + i] = s.prof[i];
    }
}
