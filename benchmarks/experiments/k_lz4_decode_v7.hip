// k_lz4_decode_v7.hip -- LZ4 block decode, two waves per block as in v5 (parser || executor through an LDS ring), with a
// CHUNK-CENTRIC executor (reference: BlockDecoder::decode, src/lz4.rs:67-140).
//
// Why: v5's executor is a lane per SEQUENCE, and a sequence's bytes land at an arbitrary byte address: gfx950 serialises
// unaligned LDS accesses, so every literal run and every 16 bytes of a match went out through exec-narrowing byte stores
// (~44 instructions per 16 bytes per call site, ~7 call sites per batch) and the copy rounds were paced by the longest match of
// the 64 lanes.  The PMC model of DESIGN 3.0 says the kernel is bound by instructions issued, so the stores had to go, not move.
//
// Here the batch is turned round once the 64 sequences are placed: a lane owns one 16-BYTE ALIGNED CHUNK of the batch's
// output (<= 64 chunks: the batch cap is 1008 bytes) and assembles it in four registers from the pieces that overlap it --
// literal runs and matches, in output order -- each piece ONE unaligned 16-byte read of its source (aligned dwords +
// v_alignbyte, shifted so that source byte and chunk byte line up) merged with v_bfi under a "keep the bytes below" mask from a
// 16-entry LDS table.  A piece overshoots to the end of the chunk and the next piece overwrites the overshoot (the CPU
// decoders' wild copy, in registers), so there is no byte-granular store anywhere: the chunk goes to the window as one
// ds_write_b128.  Every source is an LDS address: the window itself (matches inside the history), a 16-byte-aligned POOL slot
// per sequence for literals (one or two unaligned 16-byte global loads of the input, which the parser wave read a moment ago)
// and for matches older than the window (up to four 16-byte gathers from the block's own output in HBM).
//
// Dependencies inside the batch: a piece may only read bytes that stand.  Every chunk lane publishes `pos`, the end of the
// bytes it has produced; a source range is ready when the chunk(s) it lies in have passed it (two ds_bpermute).  Chains are
// the rule in text (the compressor points at the most recent occurrence), so the sequence-level redirection of v4/v5 stays:
// a match lying wholly inside one earlier match of the batch is re-pointed at that match's own source (pointer doubling).
// Simulated on G-text (64 chunks of 16 bytes): 16 rounds per KiB without redirection, 7.8 with four levels cut.
// A self-overlapping match with a period below 16 builds its chunk from the period's bytes (byte gathers: the run-heavy path).
#include "../../rust_compress_amd/csrc/rcx_dev.h"   // (experiment: lives outside the product tree, built only into librcx_ab.so)
#ifndef RCX_V7_STAT
#define RCX_V7_STAT(slot, v) ((void)0)                       // the wave simulator counts rounds / pieces here
#endif
#ifndef RCX_V7_ROUND_PRIO
#define RCX_V7_ROUND_PRIO 3
#endif

template <int CB, int TC = 1008, int HH = 2048, bool PROF5 = false, int POOLB = 2048, int RR7 = 2>
struct Lz4V7 : Lz4V5<CB, TC, HH, PROF5, 0> {
    typedef Lz4V5<CB, TC, HH, PROF5, 0> P5;
    typedef typename P5::B B;
    static_assert(TC <= 1008 && (POOLB % 16) == 0 && POOLB >= 96 + 96, "a batch is at most 64 chunks; one entry always fits the pool");
    static constexpr int POOL7 = B::LIN + 64;           // 64 bytes of read slack after the window, then the pool
    static constexpr int WBUF7 = POOL7 + POOLB + 32;    // (+ 32: a piece reads up to 20 bytes from its source's last dword)
    rcx_u32x4* tab;                                     // LDS, 64 entries: {window position | L << 16 | M << 24, aL, aM, period}
    const rcx_u32x4* lut;                               // LDS, 16 entries: mask of the low x bytes of 16

    // One batch: lane i holds entry i's descriptor (w0 = literal source position, w1 = L | M << 8 | offset << 16), as emit5.
    template <bool LITLDS = false>
    __device__ int emit7(int ns, int& lo, uint32_t w0, uint32_t w1, const uint8_t* litbuf = nullptr)
    {
        const unsigned lane = this->lane;
        const uint8_t* in = this->in; uint8_t* out = this->out; uint8_t* wb_ = this->wb_;
        const uint32_t cap = this->cap, n = this->n;
        this->make_room(B::TCAP);
        const int lo0 = lo;
        bool act = (int)lane >= lo && (int)lane < ns;
        uint32_t L = act ? w1 & 0xffu : 0u, M = act ? (w1 >> 8) & 0xffu : 0u, off = act ? w1 >> 16 : 0u;
        const uint32_t src = w0;
        const uint32_t len = L + M;
        const uint32_t incl = rcx_wave_incl_scan(len);
        uint32_t T = RCX_U(__builtin_amdgcn_readlane(incl, 63));
        int hi = ns;
        if (T > (uint32_t)B::TCAP) {                                  // rare: take the prefix that fits (an entry is <= 96 bytes)
            hi = lo0 + (int)__popcll(__ballot(act && incl <= (uint32_t)B::TCAP));
            act = act && (int)lane < hi;
            if (!act) { L = 0; M = 0; off = 0; }
        }
        const uint32_t oend0 = this->oend;
        const uint32_t ostart = oend0 + incl - len;
        const uint32_t mdst = ostart + L;
        const int32_t lbase = this->lbase;
        const uint32_t re = this->rlo_eff();
        const uint32_t slo = mdst - off;
        bool isfar = M && slo < re;                                   // source drained and slid out of the window

        // ---- pool slots (16-byte granules): literals, then the gathered match
        uint32_t lsz = (!LITLDS && L) ? (L + 15u) & ~15u : 0u;
        uint32_t fsz = isfar ? (M + 15u) & ~15u : 0u;
        const uint32_t pin = rcx_wave_incl_scan(lsz + fsz);
        if (RCX_U(__builtin_amdgcn_readlane(pin, 63)) > (uint32_t)POOLB) {     // rare: the prefix whose slots fit
            hi = lo0 + (int)__popcll(__ballot(act && pin <= (uint32_t)POOLB));
            act = act && (int)lane < hi;
            if (!act) { L = 0; M = 0; off = 0; lsz = 0; fsz = 0; isfar = false; }
        }
        if (hi != ns) T = RCX_U(__builtin_amdgcn_readlane(incl, hi - 1));
        lo = hi;
        int err = 0;
        if (act) {
            if (L > cap - ostart || ostart > cap) err = RCX_E_OUTPUT_TOO_SMALL;
            else if (M && (off == 0 || off > mdst)) err = RCX_E_MALFORMED;
            else if (M && M > cap - mdst) err = RCX_E_OUTPUT_TOO_SMALL;
        }
        const unsigned long long bad = __ballot(err != 0);
        if (bad) return __builtin_amdgcn_readlane(err, __ffsll(bad) - 1);

        const int32_t li_o = (int32_t)ostart - lbase;
        const int32_t li_m = li_o + (int32_t)L;
        const uint32_t shi = (slo + M < mdst) ? slo + M : mdst;
        const int32_t pl = POOL7 + (int32_t)(pin - lsz - fsz);        // the lane's literal slot; the match slot follows it
        const int32_t pf = pl + (int32_t)lsz;

        // ---- loads first: literals (the parser staged them a moment ago: L2 hits) and old matches, 16 bytes each
        rcx_u32x4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
        const bool lit16 = !LITLDS && L && (uint64_t)src + 32u <= (uint64_t)n;
        const bool litb = !LITLDS && L && !lit16;                    // within 32 bytes of the block's end: byte loads
        if (lit16) { g0 = *(const rcx_u32x4_u*)(in + src); if (L > 16) g1 = *(const rcx_u32x4_u*)(in + src + 16); }
        rcx_u32x4 f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0}, f2 = {0, 0, 0, 0}, f3 = {0, 0, 0, 0};
        const bool far16 = isfar && (uint64_t)slo + (uint32_t)B::MCAP <= (uint64_t)cap;
        const bool farb = isfar && !far16;
        if (far16) {
            f0 = *(const rcx_u32x4_u*)(out + slo);
            if (M > 16) f1 = *(const rcx_u32x4_u*)(out + slo + 16);
            if (M > 32) f2 = *(const rcx_u32x4_u*)(out + slo + 32);
            if (M > 48) f3 = *(const rcx_u32x4_u*)(out + slo + 48);
        }

        __builtin_amdgcn_s_setprio(RCX_V7_ROUND_PRIO);
        // ---- chains: a match lying wholly inside ONE earlier match of the batch takes that match's source (Lz4V4::emit)
        uint32_t S = off;
        {
            bool inb = M && !isfar && shi > oend0;
            if (__ballot(inb)) {
                uint32_t ka = this->lane_of(ostart, slo > oend0 ? slo : oend0);
                uint32_t kb = this->lane_of(ostart, shi > oend0 ? shi - 1 : oend0);
                const uint32_t pmd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ka << 2), (int)((isfar || off < M) ? 0xffffffffu : mdst));
                uint32_t prod = (inb && ka == kb && slo >= pmd && off >= M) ? ka : 64u;
#pragma unroll
                for (int rr = 0; rr < RR7; rr++) {
                    if (!__ballot(prod < 64u)) break;
                    const uint32_t j = prod < 64u ? prod : lane;
                    const uint32_t Sj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)S);
                    const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)prod);
                    if (prod < 64u) {
                        if (mdst - S - Sj >= re && S + Sj <= mdst) { S += Sj; prod = pk; }
                        else prod = 64u;
                    }
                }
            }
        }

        // ---- the batch's table, and the gathered bytes into their slots
        {
            const int32_t aL = LITLDS ? (int32_t)(litbuf - wb_) + (int32_t)src - li_o : pl - li_o;      // source of window byte p: wb_[aL + p]
            const int32_t aM = isfar ? pf - li_m : -(int32_t)S;
            const uint32_t per = (M && !isfar && S < 16u && S < M) ? S : 0u;                             // self-overlap, period below a chunk
            tab[lane] = rcx_u32x4{(uint32_t)li_o | (L << 16) | (M << 24), (uint32_t)aL, (uint32_t)aM, per};
            if (lit16) { *(rcx_u32x4*)(wb_ + pl) = g0; if (L > 16) *(rcx_u32x4*)(wb_ + pl + 16) = g1; }
            for (uint32_t i = 0; __ballot(litb && i < L); i++)
                if (litb && i < L) wb_[pl + (int32_t)i] = in[src + i];
            if (far16) {
                *(rcx_u32x4*)(wb_ + pf) = f0;
                if (M > 16) *(rcx_u32x4*)(wb_ + pf + 16) = f1;
                if (M > 32) *(rcx_u32x4*)(wb_ + pf + 32) = f2;
                if (M > 48) *(rcx_u32x4*)(wb_ + pf + 48) = f3;
            }
            for (uint32_t i = 0; __ballot(farb && i < M); i++)
                if (farb && i < M) wb_[pf + (int32_t)i] = out[slo + i];
        }

        // ---- chunks: lane k owns window bytes [cw, cw + 16)
        const int32_t oe0 = (int32_t)oend0 - lbase;
        const int32_t cb = oe0 & ~15;
        const int32_t tot = oe0 + (int32_t)T;
        const int32_t cw = cb + 16 * (int32_t)lane;
        int32_t pos = cw > oe0 ? cw : oe0;
        int32_t cend = cw + 16 < tot ? cw + 16 : tot;
        if (cw >= tot) { pos = cw + 16; cend = cw + 16; }            // beyond the batch: counts as done
        uint32_t si = this->lane_of((uint32_t)li_o, (uint32_t)(pos < tot ? pos : oe0));
        rcx_wave_sync();
        rcx_u32x4 acc = *(const rcx_u32x4*)(wb_ + cw);               // (the first chunk keeps the bytes the last batch left in it)
        RCX_V7_STAT(0, lane == 0); RCX_V7_STAT(4, lane == 0 ? T : 0); RCX_V7_STAT(5, lane == 0 ? (uint32_t)(hi - lo0) : 0);
        for (;;) {
            const bool busy = pos < cend;
            if (!__ballot(busy)) break;
            RCX_V7_STAT(1, lane == 0);
            const rcx_u32x4 d = tab[si & 63u];
            const int32_t dso = (int32_t)(d[0] & 0xffffu);
            const int32_t mds = dso + (int32_t)((d[0] >> 16) & 0xffu);
            const int32_t mend = mds + (int32_t)(d[0] >> 24);
            const bool isl = pos < mds;
            const int32_t segend = isl ? mds : mend;
            const int32_t e = segend < cend ? segend : cend;
            const int32_t nb = e - pos;                               // bytes of this piece (0: an empty entry, stepped over)
            const int32_t sa = (int32_t)(isl ? d[1] : d[2]) + pos;    // where the piece's first byte comes from
            const uint32_t per = isl ? 0u : d[3];
            // what has to stand: the piece's source, or the period in front of a short-period match
            const int32_t ca = per ? mds - (int32_t)per : sa;
            const int32_t cn = per ? (int32_t)per : nb;
            int32_t j0 = (ca - cb) >> 4; j0 = j0 < 0 ? 0 : j0;
            const int32_t p0 = __builtin_amdgcn_ds_bpermute(j0 << 2, pos);
            const int32_t p1 = __builtin_amdgcn_ds_bpermute((j0 < 63 ? j0 + 1 : 63) << 2, pos);
            const int32_t lim = (p0 >= cb + 16 * j0 + 16) ? p1 : p0;
            const bool chk = !isl && ca < POOL7 && ca + cn > oe0;     // a window source that reaches into this batch
            int32_t cnt = nb;
            if (chk) {
                const int32_t av = lim - ca;
                if (per) cnt = av >= cn ? nb : 0;
                else cnt = av < nb ? av : nb;
                cnt = cnt < 0 ? 0 : cnt;
            }
            const bool go = busy && cnt > 0;
            RCX_V7_STAT(2, go); RCX_V7_STAT(3, busy && !go); RCX_V7_STAT(6, go && per);
            uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            if (__ballot(go && !per)) {
                const int32_t A = go ? sa - (pos - cw) : 0;           // source byte t lines up with chunk byte t
                B::lds_load16u(wb_, A, v0, v1, v2, v3);
            }
            if (__ballot(go && per)) {                                // byte t of the chunk = period[(cw + t - mds) mod per]
                if (go && per) {
                    const int32_t ps = mds - (int32_t)per;
                    uint32_t r = (uint32_t)(cw - mds + 16 * (int32_t)per) % per;
                    uint32_t b[16];
#pragma unroll
                    for (int t = 0; t < 16; t++) { b[t] = wb_[ps + (int32_t)r]; r = (r + 1 == per) ? 0u : r + 1; }
                    v0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                    v1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
                    v2 = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
                    v3 = b[12] | (b[13] << 8) | (b[14] << 16) | (b[15] << 24);
                }
            }
            rcx_wave_sync();
            if (go) {
                const rcx_u32x4 m = lut[(pos - cw) & 15];             // keep the bytes below pos, take the rest (overshoot: overwritten later)
                acc[0] = (acc[0] & m[0]) | (v0 & ~m[0]);
                acc[1] = (acc[1] & m[1]) | (v1 & ~m[1]);
                acc[2] = (acc[2] & m[2]) | (v2 & ~m[2]);
                acc[3] = (acc[3] & m[3]) | (v3 & ~m[3]);
                *(rcx_u32x4*)(wb_ + cw) = acc;
                pos += cnt;
            }
            if (busy && pos >= mend) si++;
            rcx_wave_sync();
        }
        __builtin_amdgcn_s_setprio(RCX_FLUSH_PRIO);
        this->oend = RCX_U(oend0 + T);
        this->flush(this->oend, false);
        if (!LITLDS && RCX_FLUSH_PRIO != RCX_EXEC_PRIO) __builtin_amdgcn_s_setprio(RCX_EXEC_PRIO);
        return 0;
    }

    template <bool NOEXEC = false>      // NOEXEC (A/B builds): batches are taken off the ring and dropped -- what the parser wave costs alone
    __device__ void run_executor7(int32_t* st_out, uint32_t* len_out)
    {
        this->lane = rcx_lane();
        const unsigned lane = this->lane;
        this->init_window();
        int st = RCX_OK;
        uint32_t tail = 0;
        auto ring = this->ring;
        for (;;) {
            while (RCX_U(ring->head) == tail) __builtin_amdgcn_s_sleep(4);
            rcx_wave_sync();
            const RCX_LDS_AS typename P5::Slot* sl = &ring->slot[tail % P5::NSLOT];
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
            if (NOEXEC) { if (bt.why == B::END_ || bt.why == B::ERR_) break; continue; }
            while (lo < bt.ns && !e) e = emit7(bt.ns, lo, w0, w1);
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

// mask of the low x bytes of a 16-byte chunk, x = 0..15 (one table per workgroup)
__device__ __forceinline__ void rcx_v7_lut_init(rcx_u32x4* lut, unsigned t)
{
    if (t < 16) {
        rcx_u32x4 m;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = (int)t - 4 * k;
            m[k] = x <= 0 ? 0u : x >= 4 ? 0xffffffffu : ((1u << (8 * x)) - 1u);
        }
        lut[t] = m;
    }
}

template <int CB, int TC = 1008, int HH = 2048, int POOLB = 2048, int RR7 = 2, bool NOEXEC = false>
__global__ __launch_bounds__(128, 8) void k_lz4_decode_v7(rcx_kargs a, int only_status = 0)
{
    typedef Lz4V7<CB, TC, HH, false, POOLB, RR7> S;
    __shared__ __align__(16) uint8_t s_cbuf[CB + 96];
    __shared__ __align__(16) uint8_t s_wbuf[16 + S::WBUF7];      // 16 bytes in front: a piece's shifted read starts up to 15 bytes before its source
    __shared__ uint32_t s_epos[64];
    __shared__ __align__(16) typename S::P5::Ring s_ring;
    __shared__ __align__(16) rcx_u32x4 s_tab[64];
    __shared__ __align__(16) rcx_u32x4 s_lut[16];
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    if (only_status && a.status[b] != only_status) return;       // second pass over the blocks another kernel handed back
    if (threadIdx.x == 0) { RCX_LDS_AS typename S::P5::Ring* r0 = (RCX_LDS_AS typename S::P5::Ring*)&s_ring; r0->head = 0; r0->tail = 0; r0->abort_ = 0; }
    rcx_v7_lut_init(s_lut, threadIdx.x);
    __syncthreads();
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    S s;
    s.in = a.in_base + a.in_off[b];
    s.n = (uint32_t)a.in_len[b];
    s.out = a.out_base + a.out_off[b];
    const uint64_t cap64 = a.out_cap[b];
    s.cap = cap64 > 0xffffffffull ? 0xffffffffu : (uint32_t)cap64;
    s.cbuf = s_cbuf;
    s.wb_ = s_wbuf + 16;
    s.epos = s_epos;
    s.ring = (RCX_LDS_AS typename S::P5::Ring*)&s_ring;
    s.tab = s_tab;
    s.lut = s_lut;
    if (role == 0) { s.run_parser(); return; }
    int32_t st; uint32_t olen;
    __builtin_amdgcn_s_setprio(RCX_EXEC_PRIO);
    s.template run_executor7<NOEXEC>(&st, &olen);
    if ((threadIdx.x & 63u) == 0) {
        a.status[b] = st;
        a.out_len[b] = olen;
        if (a.in_used) a.in_used[b] = s.n;
    }
}
