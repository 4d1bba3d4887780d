// k_lz4_decode_v6.hip -- LZ4 block decode, one WORKGROUP per block, the block's whole 64 KiB history resident in LDS
// (reference: BlockDecoder::decode, src/lz4.rs:67-140).
//
// Why (measured on v5, profiles/pmc_lz4_decode.json): with one or two waves per block, sixteen blocks share a CU and
// 512 blocks share an XCD's 4 MiB L2, so only a ~2 KiB window of each block's output can live in LDS and every older
// match source is a 16-64 byte gather that misses the L2: 3.95x the algorithmic HBM traffic.  An LZ4 offset is a u16,
// so 64 KiB of LDS holds EVERY possible match source of a block; gfx950 has 160 KiB per CU, i.e. two such blocks per
// CU.  Two blocks per CU means the parallelism has to come from inside a block, so nothing here is serial per token:
//
//   parse   (per 256*W compressed bytes, all waves): every byte position is treated as a token start and gets its hop
//           target; pointer doubling inside 64-position register windows (ds_bpermute) gives each position the place
//           where its chain leaves the window, a second composition the place where it leaves the wave's 4 windows;
//           the true chain is then W dependent LDS reads (every wave walks it redundantly), and lane t of a window
//           fetches the t-th token of the chain by binary lifting over the stored doubling levels.  Tokens with more
//           than one length-extension byte are left to the fallback kernel (below).
//   execute (per 256*W output bytes, all waves): one LANE per output BYTE.  A bitmap of sequence starts + popcount
//           finds the byte's sequence, the byte is a literal (staged input), an old match byte (ring) or a match byte
//           whose source is produced by this same batch; those carry a root pointer, and pointer jumping through an
//           LDS root array (FINAL once the byte stands) resolves chains in log(depth) rounds without barriers.
//   drain   16-byte coalesced stores of the ring to HBM: input read once, output written once.
//
// Exactness by fallback: this kernel only has to be right on blocks it accepts.  Anything unusual -- malformed input,
// a short output slot, more than 64 KiB of output, long extension chains (incompressible or run-only data) -- ends the
// block with the internal status RCX_ST_BAIL6 and the exact kernel (k_lz4_decode_v5) re-runs those blocks.
#include "../../rust_compress_amd/csrc/rcx_dev.h"   // (experiment: lives outside the product tree, built only into librcx_ab.so)

#define RCX_ST_BAIL6 0x7ff00002           /* internal, never leaves the library */
#ifdef RCX_SIM_TRACE
#define V6_TRACE(...) do { if (threadIdx.x == 0) fprintf(stderr, __VA_ARGS__); } while (0)
#else
#define V6_TRACE(...) ((void)0)
#endif

template <int W>
struct Lz4V6Cfg {
    static constexpr int NT = 64 * W;
    static constexpr int SC = 256 * W;                      // compressed positions parsed per round
    static constexpr int SM = 320;                          // staged beyond them (a fast-path token reads <= 275 bytes ahead)
    static constexpr int STG = SC + SM;
    static constexpr int BB = 256 * W;                      // output bytes per batch
    static constexpr int NBMAX = 4;                         // batches per round (more output than that: fallback)
    static constexpr int DN = SC / 3 + 8;                   // a token is >= 3 bytes
    static constexpr int TOKW = 96;                         // tokens per wave and round (4 windows x <= 22)
    static constexpr int RING = 65536;
    static constexpr int O_RING = 0;
    static constexpr int O_STAGE = RING + 16;
    static constexpr int O_EXIT = O_STAGE + STG;            // u16[SC]; later the root array u16[BB]
    static constexpr int O_DESC = O_EXIT + 2 * SC;          // u64[DN]
    static constexpr int O_TOK = O_DESC + 8 * DN;           // u16[W][TOKW]
    static constexpr int O_BMAP = O_TOK + 2 * W * TOKW;     // u32[NBMAX][BB / 32]
    static constexpr int O_MISC = O_BMAP + 4 * NBMAX * (BB / 32);
    static constexpr int TOTAL = O_MISC + 128;
    static_assert(SC == BB, "the root array reuses the exit map");
    static_assert(O_STAGE % 16 == 0 && O_EXIT % 16 == 0 && O_DESC % 16 == 0 && O_BMAP % 4 == 0 && O_MISC % 4 == 0, "alignment");
    static_assert(TOTAL <= 81920, "two workgroups per CU");
};

template <int W, bool PROF = false>
__global__ __launch_bounds__(64 * W, (2 * 64 * W) / 256) void k_lz4_decode_v6(rcx_kargs a)
{
    typedef Lz4V6Cfg<W> C;
    // PROF: wave 0 accumulates cycles per phase -> scratch[b][16] (u64): stage, hops, chain, tokens, descs, classify, resolve, drain, rounds, batches
    uint64_t pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tp = PROF ? (uint64_t)__builtin_readcyclecounter() : 0;
#define V6_LAP(i) do { if (PROF) { const uint64_t t_ = (uint64_t)__builtin_readcyclecounter(); pt[i] += t_ - tp; tp = t_; } } while (0)
    __shared__ __align__(16) uint8_t lds[C::TOTAL];
#define V6_U16(o) (*(uint16_t*)(lds + (o)))
#define V6_U32(o) (*(uint32_t*)(lds + (o)))
#define V6_U64(o) (*(uint64_t*)(lds + (o)))
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = RCX_UNI(tid >> 6);
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint64_t n64 = a.in_len[b];
    uint8_t* out = a.out_base + a.out_off[b];
    const uint64_t cap64 = a.out_cap[b];
    const uint32_t cap = cap64 > (uint64_t)C::RING ? (uint32_t)C::RING : (uint32_t)cap64;
    const uint32_t mis = (uint32_t)((uintptr_t)out & 15u);
    const uint32_t n = (uint32_t)n64;
    uint32_t* const misc = (uint32_t*)(lds + C::O_MISC);        // [0..W) tokens per wave, [W..2W) output bytes per wave, [31] bail
    if (tid == 0) misc[31] = (n64 > 0x40000000ull) ? 1u : 0u;
    __syncthreads();

    uint32_t cpos = 0, opos = 0, drained = 0;                   // compressed / output position, ring bytes already in HBM
    bool bail = false;
    rcx_u32x4 pf = {0, 0, 0, 0};
    bool pf_ok = false;
    static_assert(C::STG <= C::NT * 16, "one 16-byte line per thread stages a round");
    while (cpos < n) {
        // ------------------------------------------------------------------------------------------------ stage
        const uint32_t e0 = (uint32_t)(((uintptr_t)in + cpos) & 15u);
        const uint8_t* g0 = in + ((int64_t)cpos - (int64_t)e0);         // 16-byte aligned; up to 15 bytes before `in` in the first round
        const uint32_t nrel = e0 + (n - cpos);                  // end of the compressed bytes, stage relative
        {
            const uint32_t idx = tid * 16u;
            const uint8_t* g = g0 + idx;
            if (pf_ok) *(rcx_u32x4*)(lds + C::O_STAGE + idx) = pf;           // fetched during the previous round
            else if (idx < (uint32_t)C::STG && idx < nrel) {
                if (g >= in && idx + 16u <= nrel) *(rcx_u32x4*)(lds + C::O_STAGE + idx) = *(const rcx_u32x4*)g;
                else
                    for (uint32_t j = 0; j < 16u; j++)
                        if (g + j >= in && idx + j < nrel) lds[C::O_STAGE + idx + j] = g[j];
            }
        }
        // zero this round's bitmaps
        for (uint32_t i = tid; i < (uint32_t)(C::NBMAX * (C::BB / 32)); i += (uint32_t)C::NT) V6_U32(C::O_BMAP + 4 * i) = 0;
        __syncthreads();
        V6_LAP(0);
        if (RCX_UNI(misc[31])) { V6_TRACE("v6 b%u: bail at stage\n", b); bail = true; break; }

        // ------------------------------------------------------------------------------------------------ hops + doubling
        // Everything below is a chain of dependent LDS round trips (~100+ cycles each), so the wave's 4 windows (later its
        // 4 chunks) go through every step TOGETHER: 4 independent accesses in flight per step.
        const uint32_t gb = 256u * w;                           // this wave's group of 4 windows
        uint32_t lv[4][5];                                      // doubling levels: position after 1, 2, 4, 8, 16 hops (or where the chain left)
        uint32_t ex[4];                                         // where the chain from a position leaves its window
        {
            uint32_t t[4], e1[4], e2[4], op[4];
            bool slow[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t i = gb + 64u * k + lane; t[k] = lds[C::O_STAGE + i]; e1[k] = lds[C::O_STAGE + i + 1]; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = gb + 64u * k + lane;
                uint32_t L = t[k] >> 4, ls = i + 1;
                slow[k] = false;
                if (L == 15u) { slow[k] = e1[k] == 255u; L += e1[k]; ls = i + 2; }
                op[k] = ls + L;                                 // where the offset would be
                e2[k] = lds[C::O_STAGE + op[k] + 2];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = gb + 64u * k + lane;
                uint32_t nx = op[k] + 2;
                bool sl = slow[k];
                if (op[k] == nrel) nx = nrel;                   // the final, literal-only sequence
                else if (op[k] + 2 > nrel) sl = true;           // truncated
                else if ((t[k] & 15u) == 15u) { sl = sl || e2[k] == 255u || op[k] + 2 >= nrel; nx = op[k] + 3; }
                ex[k] = i >= nrel ? nrel : sl ? (0x8000u | i) : nx;      // past the end: absorbing
                lv[k][0] = ex[k];
            }
#pragma unroll
            for (int r = 1; r <= 5; r++) {
                uint32_t peer[4];
#pragma unroll
                for (int k = 0; k < 4; k++) peer[k] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((ex[k] - (gb + 64u * k)) & 63u) << 2), (int)ex[k]);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    ex[k] = ex[k] - (gb + 64u * k) < 64u ? peer[k] : ex[k];
                    if (r < 5) lv[k][r] = ex[k];
                }
            }
        }
        // where the chain leaves the wave's group: follow the window exits (own group only: own writes, in order)
#pragma unroll
        for (int k = 0; k < 4; k++) V6_U16(C::O_EXIT + 2 * (gb + 64u * k + lane)) = (uint16_t)ex[k];
        rcx_wave_sync();
        {
            uint32_t gx[4] = {ex[0], ex[1], ex[2], ex[3]};
#pragma unroll
            for (int st = 0; st < 3; st++) {
                uint32_t t2[3];
#pragma unroll
                for (int k = 0; k < 3 - st; k++) t2[k] = V6_U16(C::O_EXIT + 2 * (gx[k] - gb < 256u ? gx[k] : gb));
#pragma unroll
                for (int k = 0; k < 3 - st; k++) gx[k] = gx[k] - gb < 256u ? t2[k] : gx[k];
            }
            rcx_wave_sync();
#pragma unroll
            for (int k = 0; k < 3; k++) V6_U16(C::O_EXIT + 2 * (gb + 64u * k + lane)) = (uint16_t)gx[k];
        }
        __syncthreads();
        V6_LAP(1);

        // ------------------------------------------------------------------------------------------------ the chain (every wave)
        const uint32_t lim = nrel < (uint32_t)C::SC ? nrel : (uint32_t)C::SC;
        uint32_t e = e0, my_entry = 0xffffffffu;
        bool flagged = false;
        for (uint32_t g = 0; g < (uint32_t)W; g++) {
            if (e < lim && e - 256u * g < 256u) {
                if (g == w) my_entry = e;
                e = RCX_UNI(V6_U16(C::O_EXIT + 2 * e));
                if (e & 0x8000u) { flagged = true; break; }
            }
        }
        if (flagged) { V6_TRACE("v6 b%u: flagged token at %u (cpos %u e0 %u nrel %u)\n", b, e & 0x7fffu, cpos, e0, nrel); bail = true; break; }                    // uniform over the workgroup: every wave walks the same chain
        const uint32_t e_next = e;                              // >= lim: the next round's entry, or nrel = end of the block
        // the next round's input is known now: fetch it into registers while this round runs (HBM latency off the path)
        {
            const uint32_t cpn = cpos + (e_next - e0);
            pf_ok = false;
            if (cpn < n) {
                const uint32_t e0n = (uint32_t)(((uintptr_t)in + cpn) & 15u);
                const uint8_t* gn = in + ((int64_t)cpn - (int64_t)e0n) + tid * 16u;
                const uint32_t nreln = e0n + (n - cpn);
                if (tid * 16u < (uint32_t)C::STG && gn >= in && tid * 16u + 16u <= nreln) { pf = *(const rcx_u32x4*)gn; pf_ok = true; }
            }
        }
        V6_LAP(2);

        // ------------------------------------------------------------------------------------------------ tokens of my windows
        uint32_t ntok = 0;
        {
            uint32_t node[4];
            bool have[4];
            uint32_t cur = my_entry;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t wb = gb + 64u * k;
                have[k] = cur < lim && cur - wb < 64u;                // uniform
                node[k] = have[k] ? cur : 0xffffu;
                if (have[k]) cur = RCX_UNI(__builtin_amdgcn_readlane((int)ex[k], (int)RCX_UNI((cur - wb) & 63u)));
            }
#pragma unroll
            for (int bit = 0; bit < 5; bit++) {
                uint32_t peer[4];
#pragma unroll
                for (int k = 0; k < 4; k++) peer[k] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((node[k] - (gb + 64u * k)) & 63u) << 2), (int)lv[k][bit]);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if ((lane >> bit) & 1u) node[k] = node[k] - (gb + 64u * k) < 64u ? peer[k] : 0xffffu;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool valid = have[k] && lane < 32u && node[k] - (gb + 64u * k) < 64u && node[k] < lim;
                const unsigned long long vm = __ballot(valid);
                if (valid) V6_U16(C::O_TOK + 2 * (w * C::TOKW + ntok + lane)) = (uint16_t)node[k];
                ntok += (uint32_t)__popcll(vm);
            }
        }
        rcx_wave_sync();
        // token fields; two vectors of 64 per wave at most
        uint32_t f_src[2], f_L[2], f_M[2], f_off[2], f_incl[2];
        uint32_t wtot = 0;
#pragma unroll
        for (int v = 0; v < 2; v++) {
            const uint32_t j = 64u * v + lane;
            if (v == 1 && ntok <= 64u) { f_src[1] = f_L[1] = f_M[1] = f_off[1] = f_incl[1] = 0; continue; }     // uniform
            const bool act = j < ntok;
            const uint32_t q = act ? V6_U16(C::O_TOK + 2 * (w * C::TOKW + j)) : 0u;
            const uint32_t t = lds[C::O_STAGE + q], e1 = lds[C::O_STAGE + q + 1];
            uint32_t L = t >> 4, ls = q + 1;
            if (L == 15u) { L += e1; ls = q + 2; }
            const uint32_t op = ls + L;
            const bool last = op >= nrel;
            const uint32_t o0 = lds[C::O_STAGE + op], o1 = lds[C::O_STAGE + op + 1], e2 = lds[C::O_STAGE + op + 2];
            uint32_t M = (t & 15u) + 4u;
            if ((t & 15u) == 15u) M += e2;
            if (last || !act) M = 0;
            if (!act) L = 0;
            f_src[v] = ls; f_L[v] = L; f_M[v] = M; f_off[v] = last ? 0u : (o0 | (o1 << 8));
            f_incl[v] = rcx_wave_incl_scan(L + M) + wtot;
            wtot = RCX_UNI(__builtin_amdgcn_readlane((int)f_incl[v], 63));
        }
        if (lane == 0) { misc[w] = ntok; misc[W + w] = wtot; }
        __syncthreads();
        V6_LAP(3);
        uint32_t cbase = 0, obase = opos, nd = 0, chunk_out = 0;
        for (uint32_t g = 0; g < (uint32_t)W; g++) {
            const uint32_t c = RCX_UNI(misc[g]), o = RCX_UNI(misc[W + g]);
            if (g < w) { cbase += c; obase += o; }
            nd += c; chunk_out += o;
        }
        bool bad = chunk_out > (uint32_t)(C::NBMAX * C::BB) || opos + chunk_out > cap || nd > (uint32_t)C::DN;
#pragma unroll
        for (int v = 0; v < 2; v++) {
            const uint32_t j = 64u * v + lane;
            if (j < ntok) {
                const uint32_t len = f_L[v] + f_M[v];
                const uint32_t os = obase + f_incl[v] - len;
                if (f_M[v] && (f_off[v] == 0u || f_off[v] > os + f_L[v])) bad = true;
                if (!bad) {
                    V6_U64(C::O_DESC + 8 * (cbase + j)) = (uint64_t)((os & 0xffffu) | (f_off[v] << 16)) | ((uint64_t)(f_src[v] | (f_L[v] << 12)) << 32);
                    if (len) {
                        const uint32_t r = os - opos;
                        atomicOr((uint32_t*)(lds + C::O_BMAP) + (r >> 5), 1u << (r & 31u));
                    }
                }
            }
        }
        if (bad) misc[31] = 1;
        __syncthreads();
        V6_LAP(4);
        pt[8] += 1;
        if (RCX_UNI(misc[31])) { V6_TRACE("v6 b%u: bad round: nd %u chunk_out %u opos %u cap %u\n", b, nd, chunk_out, opos, cap); bail = true; break; }

        // ------------------------------------------------------------------------------------------------ execute
        // One lane per output byte.  The root array (the dead exit map) holds one u16 per byte of the batch: 0x8000 | byte once
        // the byte stands, else the batch-relative position this byte copies (always an earlier one).
        uint32_t sbase = 0;
        for (uint32_t bb = opos; bb < opos + chunk_out; bb += (uint32_t)C::BB) {
            const uint32_t bsize = (opos + chunk_out - bb < (uint32_t)C::BB) ? opos + chunk_out - bb : (uint32_t)C::BB;
            const uint32_t bm = C::O_BMAP + 4 * (C::BB / 32) * ((bb - opos) / (uint32_t)C::BB);
            const uint32_t wv = V6_U32(bm + 4 * lane);
            const uint32_t pc = (uint32_t)__popc(wv);
            const uint32_t incl = rcx_wave_incl_scan(pc);
            const uint32_t pre = incl - pc + sbase;
            const uint32_t btot = RCX_UNI(__builtin_amdgcn_readlane((int)incl, 63));
            uint32_t root[4];
            bool pend[4];
            const uint32_t nch = RCX_UNI((bsize + 63u) >> 6);              // 64-byte chunks in this batch; chunk w + W*c is mine
            {
                uint32_t word[4], pr[4], ad[4], rt[4];
                uint64_t d[4];
                bool act[4], near[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (w + (uint32_t)W * c >= nch) continue;                   // uniform
                    const uint32_t r = 64u * (w + (uint32_t)W * c) + lane;
                    word[c] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((r >> 5) << 2), (int)wv);
                    pr[c] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((r >> 5) << 2), (int)pre);
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (w + (uint32_t)W * c >= nch) continue;
                    const uint32_t r = 64u * (w + (uint32_t)W * c) + lane;
                    act[c] = r < bsize;
                    const uint32_t idx = pr[c] + (uint32_t)__popc(word[c] & ((2u << (r & 31u)) - 1u)) - 1u;
                    d[c] = V6_U64(C::O_DESC + 8 * (act[c] ? idx : 0u));
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    pend[c] = false; root[c] = 0;
                    if (w + (uint32_t)W * c >= nch) continue;
                    const uint32_t r = 64u * (w + (uint32_t)W * c) + lane;
                    const uint32_t p = bb + r;
                    const uint32_t d0 = (uint32_t)d[c], d1 = (uint32_t)(d[c] >> 32);
                    const uint32_t os = d0 & 0xffffu, off = d0 >> 16, lsrc = d1 & 0xfffu, L = (d1 >> 12) & 0x1ffu;
                    const uint32_t rel = (p - os) & 0xffffu;
                    const bool lit = rel < L;
                    const uint32_t k = rel - L;
                    uint32_t sp = p - off;                                      // source position of a match byte
                    if (!lit && off && off <= k) sp = (os + L - off) + k % off; // self-overlapping match: periodic source
                    near[c] = act[c] && !lit && sp >= bb;
                    rt[c] = sp - bb;
                    ad[c] = lit ? (uint32_t)C::O_STAGE + lsrc + rel : (uint32_t)C::O_RING + mis + (sp & 0xffffu);
                    ad[c] = (act[c] && !near[c]) ? ad[c] : 0u;
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (w + (uint32_t)W * c >= nch) continue;
                    ad[c] = lds[ad[c]];                                         // the byte
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (w + (uint32_t)W * c >= nch) continue;
                    const uint32_t r = 64u * (w + (uint32_t)W * c) + lane;
                    if (act[c] && !near[c]) lds[C::O_RING + mis + bb + r] = (uint8_t)ad[c];
                    if (act[c]) V6_U16(C::O_EXIT + 2 * r) = (uint16_t)(near[c] ? rt[c] : (0x8000u | ad[c]));
                    pend[c] = near[c]; root[c] = rt[c];
                }
            }
            __syncthreads();
            V6_LAP(5);
            // pointer jumping: a byte whose root stands copies it and stands itself; otherwise it moves one root closer
            for (;;) {
                bool on[4];
                bool any = false;
#pragma unroll
                for (int c = 0; c < 4; c++) { on[c] = __ballot(pend[c]) != 0; any = any || on[c]; }
                if (!any) break;
                uint32_t rr[4];
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (on[c]) rr[c] = V6_U16(C::O_EXIT + 2 * (pend[c] ? root[c] : 0u));
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (!on[c]) continue;
                    const uint32_t r = 64u * (w + (uint32_t)W * c) + lane;
                    if (pend[c]) {
                        V6_U16(C::O_EXIT + 2 * r) = (uint16_t)rr[c];
                        if (rr[c] & 0x8000u) { lds[C::O_RING + mis + bb + r] = (uint8_t)rr[c]; pend[c] = false; }
                        root[c] = rr[c];
                    }
                }
            }
            __syncthreads();
            V6_LAP(6);
            pt[9] += 1;
            sbase += btot;
        }
        opos += chunk_out;
        // ------------------------------------------------------------------------------------------------ drain whole 16-byte lines
        {
            const uint32_t upto = (opos + mis) & ~15u;              // ring index
            for (uint32_t i = drained + tid * 16u; i < upto; i += (uint32_t)C::NT * 16u) {
                const rcx_u32x4 v = *(const rcx_u32x4*)(lds + C::O_RING + i);
                if (i >= mis) *(rcx_u32x4*)(out - mis + i) = v;
                else
                    for (uint32_t j = mis; j < 16u; j++) out[j - mis] = lds[C::O_RING + j];
            }
            if (upto > drained) drained = upto;
        }
        V6_LAP(7);
        cpos += e_next - e0;
    }
    if (!bail) {
        // tail: the bytes of the last, partial line
        const uint32_t endi = opos + mis;
        for (uint32_t i = drained + tid; i < endi; i += (uint32_t)C::NT)
            if (i >= mis) out[i - mis] = lds[C::O_RING + i];
    }
    if (PROF && a.scratch && tid == 0) {
        uint64_t* q = (uint64_t*)a.scratch + (size_t)b * 16;
        for (int i = 0; i < 10; i++) q[i] = pt[i];
    }
    if (tid == 0) {
        a.status[b] = bail ? (int32_t)RCX_ST_BAIL6 : RCX_OK;
        a.out_len[b] = bail ? 0u : opos;
        if (a.in_used) a.in_used[b] = n64;
    }
#undef V6_LAP
#undef V6_U16
#undef V6_U32
#undef V6_U64
}
