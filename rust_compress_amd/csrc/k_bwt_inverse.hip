// k_bwt_inverse.hip -- batched inverse BWT by list ranking (see k_bwt.hip for the overview).
// Replaces compute_inversion_table + InverseIterator, src/bwt/mod.rs:223-282.
#include <string>
#include <vector>
#include "rcx_dev.h"
#ifndef RCX_ALIGNBYTE
#define RCX_ALIGNBYTE(hi, lo, sh) __builtin_amdgcn_alignbyte((hi), (lo), (sh))   // ({hi,lo} >> 8*sh) & 0xffffffff
#endif

// ---------------------------------------------------------------------------------------------------
// inverse
// ---------------------------------------------------------------------------------------------------
#define BWTI_THREADS 1024
#define BWTI_WAVES (BWTI_THREADS / 64)
#define BWTI_MAXMARK 16384       /* marked nodes per block */
#define BWTI_SLOTS 8             /* chains a thread chases at once */
#define BWTI_CHUNK 1024u         /* blocks per launch: bounds the jump-table scratch */

// per in-flight block: the 4n-byte jump table + BWTI_CAPX n bytes where the walkers park what they emit on the first chase
#define BWTI_CAPX 16u
static uint64_t bwti_table_bytes(uint64_t max_block) { return (max_block * 4 + 255) & ~255ull; }
static __host__ __device__ inline uint64_t bwti_cap(uint64_t stride) { const uint64_t c = BWTI_CAPX * (stride ? stride : 1); return c < 0xfff0u ? c : 0xfff0u; }   // a parked length fits 16 bits
// (the six-launch pipeline's scratch layout: see k_bwti3_* below)
#define BWTI3_NODES 65536u
#define BWTI3_SUPERS 16384u
#define BWTI3_TILE 65536u
#define BWTI3_NONE 0xffffffffu
struct bwti_node { uint32_t a, b; };

struct bwti3_layout { uint64_t park, park2, nodes, sup, rk, lng, lcnt, slot; };
static inline uint32_t bwti3_pitch_for(uint64_t n)
{
    const uint64_t stride = n > BWTI3_NODES ? (n + BWTI3_NODES - 1) / BWTI3_NODES : 1;
    const uint64_t p = (8 * stride + 15) & ~15ull;         // P(a chain is longer) = e^-8
    return (uint32_t)(p < 32 ? 32 : (p > 0xfff0u ? 0xfff0u : p));
}
static bwti3_layout bwti3_layout_for(uint64_t maxn)
{
    const uint64_t mmax = (maxn < BWTI3_NODES ? maxn : BWTI3_NODES) + 17;
    auto al = [](uint64_t v) { return (v + 255) & ~255ull; };
    bwti3_layout l;
    l.park = bwti_table_bytes(maxn);                       // a chain's first 16 bytes: node after node, dense
    l.park2 = l.park + al(mmax * 16);                      // its bytes 16 .. pitch - 1 (1 % of the chains get that far)
    l.nodes = l.park2 + al(mmax * (uint64_t)(bwti3_pitch_for(maxn) - 16));
    l.sup = l.nodes + al(mmax * 8);
    l.rk = l.sup + al((BWTI3_SUPERS + 16) * 8ull);
    l.lng = l.rk + al((BWTI3_SUPERS + 16) * 4ull);
    l.lcnt = l.lng + al(mmax * 8);
    l.slot = l.lcnt + 256;
    return l;
}

static uint64_t bwti_slot_bytes(uint64_t max_block)
{
    const uint64_t stride = (max_block + BWTI_MAXMARK - 1) / BWTI_MAXMARK;
    const uint64_t one = bwti_table_bytes(max_block) + ((bwti_cap(stride) * (BWTI_MAXMARK + 1) + 255) & ~255ull);     // k_bwt_inverse: table | park
    const uint64_t six = bwti3_layout_for(max_block).slot;
    return one > six ? one : six;
}
static uint64_t bwt_inverse_scratch_bytes(uint32_t nblocks, uint64_t max_block)
{
    const uint64_t nb = nblocks < BWTI_CHUNK ? nblocks : BWTI_CHUNK;
    return nb * bwti_slot_bytes(max_block) + 256;
}

// One workgroup (16 waves) per block.  List ranking: every `stride`-th slot of the jump table (and origin) is a marked node; a
// walker chases from its marked node to the next one (parking the bytes it passes), the marked nodes are ranked among themselves
// by pointer jumping in LDS, and the parked chains are copied to their places.  The chase is a chain of dependent random loads,
// so what counts is the LONGEST chain of a block (gaps between marked nodes are geometric: mean `stride`, maximum about
// stride x ln(marked nodes)): 16384 marked nodes instead of 4096 cut it from ~530 to ~150 steps, and a thread keeps 8 chains in
// flight and starts its next marked node the moment one of them ends.
//
// MODE 1 (MIN) is the reference's OTHER inverse, decode_minimal (src/bwt/mod.rs:298-315): i = origin; n times { ch = L[i]; the text is
// written BACKWARDS from its end; i = C[ch] + #{k < i : L[k] == ch} }.  That is a walk of n steps along the LF permutation
// (table[i] = LF(i), a coalesced store here), with no special slot for origin -- which is why the reference's function returns a
// wrong text whenever T[n-1] also occurs in L[..origin], and why the walk can close a cycle shorter than n (the output is then
// periodic).  Both are reproduced: the marked-node list is cut where it returns to origin's node, ranked the same way, the parked
// chains are copied in reverse, and a short cycle is replicated down the block.  The reference has no "not a BWT" check on this
// path: every (L, origin < n) has an answer.
//
// MODE 2 is the table inverse again, computed along the same backward walk: the jump table is the inverse of the permutation
// place(i) = slot of L[i] in the reference's placement order (origin first in its symbol), so the chain origin -> table[origin]-1 -> ...
// read backwards is y_0 = origin, y_{j+1} = place(y_j), with out[n-1-j] = L[y_j]; origin has no predecessor in the table, so the
// reference's chain always ends at the wrap slot, and "it covers the block" is "the cycle of place through origin has length n".
// place(i) is stored at i -- coalesced, where MODE 0 scatters 4-byte entries -- which is what makes it the default (1024 x 256 KiB: 8.2 -> 7.6 ms; MODE 0 stays as variant 2).
template <int MODE>
__global__ __launch_bounds__(BWTI_THREADS) void k_bwt_inverse(rcx_kargs a, uint32_t block0, uint64_t table_stride, uint64_t table_bytes, uint32_t capx)
{
    constexpr bool MIN = MODE == 1, LF = MODE != 0;
    __shared__ uint32_t s_rk[BWTI_MAXMARK + 16];      // walker's emissions, then (pointer jumping) emissions from this node to the chain's end
    __shared__ uint16_t s_next[BWTI_MAXMARK + 16];    // marked node -> next marked node id (or NONE16)
    __shared__ uint16_t s_len[BWTI_MAXMARK + 16];     // min(emissions, 0xffff): parked chains are shorter than that
    __shared__ uint32_t s_tot[256];
    __shared__ uint32_t s_ok;
    uint32_t (*s_cnt)[256] = (uint32_t (*)[256])s_rk;  // phases 1-2: per-wave symbol counters -> running slots (16 KiB of s_rk)
    static_assert(BWTI_WAVES * 256 <= BWTI_MAXMARK, "the counters live in s_rk");
    const uint32_t slot = blockIdx.x;
    const uint32_t b = block0 + slot;
    if (b >= a.nblocks) return;
    const unsigned tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const uint8_t* L = a.in_base + a.in_off[b];
    const uint32_t n = (uint32_t)a.in_len[b];
    uint8_t* out = a.out_base + a.out_off[b];
    const uint32_t origin = a.aux ? a.aux[b] : 0u;
    uint32_t* table = (uint32_t*)((uint8_t*)a.scratch + (size_t)slot * table_stride);
    uint8_t* park = (uint8_t*)table + table_bytes;            // walker m parks its first-chase bytes at park[m * cap ..]
    const bool packed = n < 0xffffffu;                        // index + 1 fits 24 bits: the entry also holds the byte
    if (n == 0 || a.out_cap[b] < n || origin >= n) {
        if (tid == 0) {
            // mod.rs:230 index panic; decode_minimal: n == 0 is Ok only with origin == 0 (:300-302), `i >= n` is an error (:310)
            a.status[b] = n == 0 ? ((MIN && origin != 0) ? RCX_E_MALFORMED : RCX_OK) : (origin >= n ? RCX_E_MALFORMED : RCX_E_OUTPUT_TOO_SMALL);
            a.out_len[b] = 0; if (a.in_used) a.in_used[b] = n;
        }
        return;
    }
    for (unsigned i = tid; i < BWTI_WAVES * 256; i += BWTI_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t per = ((n + BWTI_WAVES - 1) / BWTI_WAVES + 63u) & ~63u;
    const uint32_t w0 = w * per < n ? w * per : n, w1 = w0 + per < n ? w0 + per : n;
    if constexpr (LF) {
        // ---- 1+2 (backward walk).  Two passes over the wave's slice of L, 64 positions per step, equal bytes found by the sorter's
        // hand-written match-any ballots: the first only counts (one LDS update per distinct byte of a step: the LDS-atomic
        // histogram it replaces serialised on the text's frequent bytes and cost as much as the ranking itself), the second, after
        // the prefix sums, stores place(i) = base + rank AT i.  Origin is left out of the counts: its place is the first slot of its
        // symbol (mod.rs:230).
        for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool valid = i < w1 && (MIN || i != origin);
            const uint32_t c = i < w1 ? L[i] : 0u;
            const unsigned long long peers = BWS_PEERS(valid, c);
            if (valid && (peers & ((1ull << lane) - 1ull)) == 0) s_cnt[w][c] += (uint32_t)__popcll(peers);   // one lane per distinct byte
        }
        __syncthreads();
        const uint32_t osym = L[origin];
        if (tid < 256) {
            uint32_t tot = (!MIN && tid == osym) ? 1u : 0u;
            for (int ww = 0; ww < BWTI_WAVES; ww++) tot += s_cnt[ww][tid];
            s_tot[tid] = tot;
        }
        __syncthreads();
        if (tid == 0) { uint32_t acc = 0; for (int c = 0; c < 256; c++) { const uint32_t t = s_tot[c]; s_tot[c] = acc; acc += t; } }
        __syncthreads();
        if (tid < 256) {
            uint32_t acc = s_tot[tid] + ((!MIN && tid == osym) ? 1u : 0u);
            for (int ww = 0; ww < BWTI_WAVES; ww++) { const uint32_t t = s_cnt[ww][tid]; s_cnt[ww][tid] = acc; acc += t; }
        }
        __syncthreads();
        // the entry carries the byte the walker emits there (L[i]) in its top 8 bits when the block is shorter than 2^24: the
        // chase then costs ONE random load per step instead of two
        for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool valid = i < w1 && (MIN || i != origin);
            const uint32_t c = i < w1 ? L[i] : 0u;
            const unsigned long long peers = BWS_PEERS(valid, c);
            const uint32_t before = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
            uint32_t basec = 0;
            if (valid) basec = s_cnt[w][c];
            rcx_wave_sync();
            if (valid) {
                table[i] = packed ? (basec + before) | (c << 24) : basec + before;
                if (before == 0) s_cnt[w][c] = basec + (uint32_t)__popcll(peers);   // group leader advances the counter
            }
            else if (i < w1 && i == origin) table[i] = packed ? s_tot[osym] | (c << 24) : s_tot[osym];
            rcx_wave_sync();
        }
    }
    else {
    // ---- 1. histogram per wave slice (each wave owns a contiguous slice of L), mod.rs:226-228
    for (uint32_t i = w0 + lane; i < w1; i += 64) atomicAdd(&s_cnt[w][L[i]], 1u);
    __syncthreads();
    // exclusive prefix over (symbol major, wave minor); the `origin` element goes first in its symbol (mod.rs:230)
    const uint32_t osym = L[origin];
    if (tid < 256) {
        uint32_t tot = 0;
        for (int ww = 0; ww < BWTI_WAVES; ww++) tot += s_cnt[ww][tid];
        s_tot[tid] = tot;
    }
    __syncthreads();
    if (tid == 0) { uint32_t acc = 0; for (int c = 0; c < 256; c++) { const uint32_t t = s_tot[c]; s_tot[c] = acc; acc += t; } }
    __syncthreads();
    if (tid < 256) {
        uint32_t acc = s_tot[tid] + ((!MIN && tid == osym) ? 1u : 0u);      // slot 0 of osym is reserved for origin
        for (int ww = 0; ww < BWTI_WAVES; ww++) { const uint32_t t = s_cnt[ww][tid]; s_cnt[ww][tid] = acc; acc += t; }
    }
    __syncthreads();
    // the origin element itself was counted in its wave's slice: take it out of that slice's budget
    if (!MIN && tid == 0) {
        if (LF) table[origin] = packed ? s_tot[osym] | (osym << 24) : s_tot[osym];   // place(origin): the first slot of its symbol
        else table[s_tot[osym]] = packed ? (osym << 24) : 0u;     // table[place(L[origin])] = 0 (+ the byte there, see below)
        const uint32_t ow = origin / per;
        for (int ww = (int)ow + 1; ww < BWTI_WAVES; ww++) s_cnt[ww][osym] -= 1u;
    }
    __syncthreads();
    // ---- 2. stable scatter: 64 positions per step, rank among equal bytes by 8 ballots (mod.rs:231-236)
    for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool valid = i < w1 && (MIN || i != origin);
        const uint32_t c = i < w1 ? L[i] : 0u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const unsigned long long m = __ballot((c >> bit) & 1u);
            peers &= ((c >> bit) & 1u) ? m : ~m;
        }
        const uint32_t before = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        uint32_t basec = 0;
        if (valid) basec = s_cnt[w][c];
        rcx_wave_sync();
        if (valid) {
            // the entry carries the byte the walker will emit from there (L[i]) in its top 8 bits when the block is shorter
            // than 2^24: the chase then costs ONE random load per step instead of two (the kernel is bound by random accesses)
            if (LF) table[i] = packed ? (basec + before) | (c << 24) : basec + before;    // LF(i) / place(i), and the byte emitted AT i
            else table[basec + before] = packed ? (i + 1u) | (c << 24) : i + 1u;
            if (before == 0) s_cnt[w][c] = basec + (uint32_t)__popcll(peers);   // group leader advances the counter
        }
        rcx_wave_sync();
    }
    }
    __threadfence_block();
    __syncthreads();
    // ---- 3. list ranking.  marked nodes: every `stride`-th slot, plus origin (id M0 if not already marked)
    uint32_t stride = (n + BWTI_MAXMARK - 1) / BWTI_MAXMARK;
    if (stride < 1) stride = 1;
    const uint32_t M0 = (n + stride - 1) / stride;
    const bool origin_marked = (origin % stride) == 0;
    const uint32_t M = M0 + (origin_marked ? 0u : 1u);
    const uint32_t NONE = 0xffffffffu, NONE16 = 0xffffu;
    // First chase: a walker also parks the bytes it emits (up to `cap`, 16 times the mean chain length); once the marked
    // nodes are ranked, a parked chain is COPIED to its place -- only a chain longer than `cap` is chased a second time.
    // (`capx` < BWTI_CAPX is the test knob, variant 1: it parks less, so that second chases happen on ordinary inputs)
    const uint32_t pitch = (uint32_t)bwti_cap(stride);
    const uint32_t cap = capx >= BWTI_CAPX ? pitch : (((capx * stride + 15u) & ~15u) < pitch ? ((capx * stride + 15u) & ~15u) : pitch);
    // a thread owns the marked nodes tid, tid + 1024, ...; it chases BWTI_SLOTS of them at once and refills a slot when its chain ends
    for (int pass = 0; pass < 2; pass++) {
        uint32_t cur[BWTI_SLOTS], cnt[BWTI_SLOTS], wr[BWTI_SLOTS], mid[BWTI_SLOTS]; bool live[BWTI_SLOTS];
        uint64_t pk[BWTI_SLOTS], pk2[BWTI_SLOTS];                         // pass 0: the chain's bytes, parked sixteen at a time
        uint32_t nextm = tid;
        auto start = [&](int q) {                                         // the thread's next marked node -> slot q
            live[q] = false; wr[q] = NONE; cnt[q] = 0; cur[q] = 0; mid[q] = 0; pk[q] = 0; pk2[q] = 0;
            while (nextm < M) {
                const uint32_t m = nextm; nextm += BWTI_THREADS;
                if (LF && pass == 1 && s_next[m] != NONE16) continue;     // not on origin's cycle: the walk never comes here
                if (LF && pass == 1 && s_len[m] <= cap) {                // step j of the walk writes out[n - 1 - j] (mod.rs:312)
                    const uint8_t* src = park + (size_t)m * pitch;
                    const uint32_t len = s_len[m];
                    uint8_t* dst = out + (n - 1u - (s_tot[0] - s_rk[m]));
                    uint32_t t = 0;
                    for (; t + 8 <= len; t += 8) *(rcx_u64_u*)(dst - t - 7) = __builtin_bswap64(*(const uint64_t*)(src + t));
                    for (; t < len; t++) *(dst - t) = src[t];
                    continue;
                }
                if (pass == 1 && s_len[m] <= cap) {                       // parked on the first chase: copy, no second chase
                    const uint8_t* src = park + (size_t)m * pitch;
                    const uint32_t len = s_len[m], dstp = n - s_rk[m];
                    uint32_t t = 0;
                    for (; t + 16 <= len && dstp + t + 16 <= n; t += 16)          // park slots are 16-byte aligned, `out` need not be
                        *(rcx_u32x4_u*)(out + dstp + t) = *(const rcx_u32x4*)(src + t);
                    for (; t < len; t++) if (dstp + t < n) out[dstp + t] = src[t];
                    continue;
                }
                live[q] = true; mid[q] = m; cur[q] = m < M0 ? m * stride : origin;
                if (pass == 1) wr[q] = LF ? s_tot[0] - s_rk[m] : n - s_rk[m];
                break;
            }
        };
#pragma unroll
        for (int q = 0; q < BWTI_SLOTS; q++) start(q);
        for (;;) {
            bool any = false;
#pragma unroll
            for (int q = 0; q < BWTI_SLOTS; q++) any = any || live[q];
            if (!any) break;
            uint32_t v[BWTI_SLOTS], c2[BWTI_SLOTS]; uint8_t ch[BWTI_SLOTS];
#pragma unroll
            for (int q = 0; q < BWTI_SLOTS; q++) v[q] = live[q] ? table[cur[q]] : 0u;    // the jump-table loads, all in flight (non-temporal loads: 7.4 -> 10 ms)
#pragma unroll
            for (int q = 0; q < BWTI_SLOTS; q++) {
                if (LF) {                                                 // the entry AT cur: LF(cur) and (packed) L[cur]
                    if (packed) { ch[q] = (uint8_t)(v[q] >> 24); c2[q] = v[q] & 0xffffffu; }
                    else { c2[q] = v[q]; ch[q] = live[q] ? L[cur[q]] : (uint8_t)0; }
                    v[q] = 1u;                                            // no wrap slot on this path
                }
                else if (packed) { ch[q] = (uint8_t)(v[q] >> 24); v[q] &= 0xffffffu; c2[q] = v[q] ? v[q] - 1u : origin; }
                else { c2[q] = v[q] ? v[q] - 1u : origin; ch[q] = live[q] ? L[c2[q]] : (uint8_t)0; }
            }
#pragma unroll
            for (int q = 0; q < BWTI_SLOTS; q++) {
                if (!live[q]) continue;
                uint32_t nxt = NONE16;
                bool stop;
                if (v[q] == 0) stop = true;                               // wrapped: L[origin] was emitted, chain ends
                else {
                    const bool mk = (c2[q] % stride) == 0 || c2[q] == origin;
                    stop = mk || cnt[q] + 1 >= n;
                    if (mk) nxt = (c2[q] == origin && !origin_marked) ? M0 : c2[q] / stride;
                    cur[q] = c2[q];
                }
                if (pass == 1) { if (wr[q] + cnt[q] < n) out[LF ? n - 1u - (wr[q] + cnt[q]) : wr[q] + cnt[q]] = ch[q]; }
                else {
                    // a byte store per step kept ~260 K partial-line writes per block on their way to HBM (the lines leave the L2
                    // long before a walker comes back to them): 8 bytes per store took 14.2 -> 8.5 ms, 16 per store another 3 %
                    if (cnt[q] & 8u) pk2[q] |= (uint64_t)ch[q] << (8u * (cnt[q] & 7u)); else pk[q] |= (uint64_t)ch[q] << (8u * (cnt[q] & 7u));
                    if (((cnt[q] & 15u) == 15u || stop) && (cnt[q] & ~15u) < cap) {
                        *(rcx_u32x4*)(park + (size_t)mid[q] * pitch + (cnt[q] & ~15u)) = rcx_u32x4{(uint32_t)pk[q], (uint32_t)(pk[q] >> 32), (uint32_t)pk2[q], (uint32_t)(pk2[q] >> 32)};
                        pk[q] = 0; pk2[q] = 0;
                    }
                }
                cnt[q]++;
                if (stop) {
                    if (pass == 0) { const uint32_t m = mid[q]; s_next[m] = (uint16_t)nxt; s_rk[m] = cnt[q]; s_len[m] = (uint16_t)(cnt[q] < 0xffffu ? cnt[q] : 0xffffu); }
                    start(q);
                }
            }
        }
        __syncthreads();
        if (pass == 0) {
            // rank the marked nodes: pointer jumping, s_rk[m] becomes the number of bytes emitted from m to the end of its chain
            constexpr int PER = (BWTI_MAXMARK + 1 + BWTI_THREADS - 1) / BWTI_THREADS;
            const uint32_t mstart = origin_marked ? origin / stride : M0;
            if (LF) {                                                     // a permutation: the list from origin's node comes back to it; cut it there
                for (uint32_t m = tid; m < M; m += BWTI_THREADS) if (s_next[m] == mstart) s_next[m] = (uint16_t)NONE16;
                __syncthreads();
            }
            for (uint32_t span = 1; span < M; span <<= 1) {
                uint32_t add[PER]; uint16_t nn[PER];
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const uint32_t m = tid + (uint32_t)j * BWTI_THREADS;
                    add[j] = 0; nn[j] = (uint16_t)NONE16;
                    if (m < M) { const uint32_t nx = s_next[m]; if (nx != NONE16) { add[j] = s_rk[nx]; nn[j] = s_next[nx]; } }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const uint32_t m = tid + (uint32_t)j * BWTI_THREADS;
                    if (m < M && s_next[m] != NONE16) { s_rk[m] += add[j]; s_next[m] = nn[j]; }
                }
                __syncthreads();
            }
            if (tid == 0) {                                               // origin's chain must end (no loop) and cover the block: else not a BWT
                const uint32_t m = mstart;
                s_ok = (MIN || (s_next[m] == NONE16 && s_rk[m] == n)) ? 1u : 0u;    // (MODE 2: the cycle through origin is the whole block)
                if (LF) s_tot[0] = s_rk[m];                               // steps until the walk is back at origin
            }
            __syncthreads();
            if (!s_ok) break;
        }
    }
    if (MIN) {
        // a cycle of c < n steps: step j >= c repeats step j - c (mod.rs:309-314 just keeps walking).  Doubling copies down the block.
        __threadfence_block();
        __syncthreads();
        for (uint32_t have = s_tot[0]; have < n;) {
            const uint32_t cp = have < n - have ? have : n - have;
            for (uint32_t t = tid; t < cp; t += BWTI_THREADS) out[n - 1u - (have + t)] = out[n - 1u - t];
            have += cp;
            __threadfence_block();
            __syncthreads();
        }
    }
    if (tid == 0) {
        const bool ok = s_ok != 0;
        a.status[b] = ok ? RCX_OK : RCX_E_MALFORMED;
        a.out_len[b] = ok ? n : 0;
        if (a.in_used) a.in_used[b] = n;
    }
}


// ---------------------------------------------------------------------------------------------------
// The default: the same inverse (MODE 2: the backward walk along place()) as SIX launches.
// What bounds the one-workgroup-per-block kernel above is the parallelism a block offers: its 16 384 chains have geometric
// lengths (mean 16 steps, the longest ~150), so on average only ~1700 of them are alive, and the kernel makes up for it with 256
// blocks in flight -- 32 MiB of tables per XCD against 4 MiB of L2, every step a 128-byte line from HBM for the 4 bytes it wants
// (31 GB per batch of 268 MB).  Here a block has up to 65 536 marked nodes (every 4th slot of a 256 KiB block: mean 4 steps, the
// longest ~45), the chase is cut loose from "one workgroup per block", and the order of the workgroups keeps an XCD on few blocks:
//   k_bwti_table     one workgroup per block: place(i) (+ L[i] in the top byte) stored at i -- phases 1+2 above;
//   k_bwti3_chase    one workgroup per (block, slice of its marked nodes); workgroup i -> (block, slice) is chosen so that the
//                    workgroups resident on ONE XCD (the dispatcher deals them round-robin: i mod 8) belong to few blocks: XCD x
//                    takes blocks x, x + 8, ... one after the other, SL slices each.  A walker keeps its chain's bytes in a
//                    16-byte shift register and parks them with ONE store (slots of 16 bytes, node after node: dense), and
//                    writes its node record {emissions, next marked node};
//   k_bwti3_contract 65 536 nodes do not fit the LDS ranking: every SUP-th node (and origin's) is a SUPER node; a walker per super
//                    node follows the node records to the next super node, leaves {emissions before it in this super chain,
//                    super id, length} in every node it passes and writes the super record {emissions, next super node};
//   k_bwti3_rank     one workgroup per block: <= 16 385 super records -> LDS, pointer jumping, status;
//   k_bwti3_emit     one workgroup per (block, 64 KiB of its output): every node's place is rank(super) + offset; the chains
//                    that fall into the tile are copied (reversed) into LDS and the tile is streamed out -- no scattered store
//                    reaches HBM.  A chain longer than its park slot (1 %) is put on a list;
//   k_bwti3_long     the listed chains are chased again and write their bytes beyond the slot directly.
// ---------------------------------------------------------------------------------------------------
struct bwti3_geom {
    uint32_t stride, M0, M, mstart, pitch, cap, SUP, S0, S, sstart; bool origin_marked;
    __device__ bwti3_geom(uint32_t n, uint32_t origin, uint32_t shortpark)
    {
        stride = n > BWTI3_NODES ? (uint32_t)(((uint64_t)n + BWTI3_NODES - 1) / BWTI3_NODES) : 1u;
        M0 = (uint32_t)(((uint64_t)n + stride - 1) / stride);
        origin_marked = (origin % stride) == 0;
        M = M0 + (origin_marked ? 0u : 1u);
        mstart = origin_marked ? origin / stride : M0;
        const uint64_t p = (8ull * stride + 15) & ~15ull;
        pitch = (uint32_t)(p < 32 ? 32 : (p > 0xfff0u ? 0xfff0u : p));
        cap = shortpark ? 1u : pitch;                            // (the tests' knob: everything beyond a chain's first byte takes k_bwti3_long)
        SUP = (M0 + BWTI3_SUPERS - 1) / BWTI3_SUPERS;
        if (SUP < 1) SUP = 1;
        S0 = (M0 + SUP - 1) / SUP;
        const bool reg = mstart < M0 && (mstart % SUP) == 0;
        S = S0 + (reg ? 0u : 1u);
        sstart = reg ? mstart / SUP : S0;
    }
    __device__ bool is_super(uint32_t m) const { return (m < M0 && (m % SUP) == 0) || m == mstart; }
    __device__ uint32_t super_of(uint32_t m) const { return (m < M0 && (m % SUP) == 0) ? m / SUP : S0; }
    __device__ uint32_t node_of(uint32_t s) const { return s < S0 ? s * SUP : mstart; }
};

// what every kernel of the pipeline needs of its block; false: nothing to do (k_bwti_table has written the status)
struct bwti3_block {
    const uint8_t* L; uint8_t* out; uint32_t n, origin, b; uint8_t* base;
    __device__ bool init(const rcx_kargs& a, uint32_t block0, uint32_t slot, uint64_t slot_stride)
    {
        b = block0 + slot;
        if (b >= a.nblocks) return false;
        L = a.in_base + a.in_off[b];
        n = (uint32_t)a.in_len[b];
        out = a.out_base + a.out_off[b];
        origin = a.aux ? a.aux[b] : 0u;
        base = (uint8_t*)a.scratch + (size_t)slot * slot_stride;
        return !(n == 0 || a.out_cap[b] < n || origin >= n);
    }
};

__global__ __launch_bounds__(BWTI_THREADS) void k_bwti_table(rcx_kargs a, uint32_t block0, uint64_t table_stride, uint64_t lcnt_off)
{
    __shared__ uint32_t s_cnt[BWTI_WAVES][256];
    __shared__ uint32_t s_tot[256];
    const uint32_t slot = blockIdx.x;
    const uint32_t b = block0 + slot;
    if (b >= a.nblocks) return;
    const unsigned tid = threadIdx.x, w = RCX_UNI(tid >> 6), lane = tid & 63u;
    const uint8_t* L = a.in_base + a.in_off[b];
    const uint32_t n = (uint32_t)a.in_len[b];
    const uint32_t origin = a.aux ? a.aux[b] : 0u;
    uint32_t* table = (uint32_t*)((uint8_t*)a.scratch + (size_t)slot * table_stride);
    const bool packed = n < 0xffffffu;
    if (n == 0 || a.out_cap[b] < n || origin >= n) {              // (the other kernels skip such a block)
        if (tid == 0) {
            a.status[b] = n == 0 ? RCX_OK : (origin >= n ? RCX_E_MALFORMED : RCX_E_OUTPUT_TOO_SMALL);
            a.out_len[b] = 0; if (a.in_used) a.in_used[b] = n;
        }
        return;
    }
    if (tid == 0) *(uint32_t*)((uint8_t*)table + lcnt_off) = 0;   // the block's list of long chains is empty
    for (unsigned i = tid; i < BWTI_WAVES * 256; i += BWTI_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t per = ((n + BWTI_WAVES - 1) / BWTI_WAVES + 63u) & ~63u;
    const uint32_t w0 = w * per < n ? w * per : n, w1 = w0 + per < n ? w0 + per : n;
    // (a step is 64 bytes of L per wave: eight steps' loads are issued together, or the loop is one memory latency per 64 bytes)
    constexpr int PF = 8;
    for (uint32_t i0 = w0; i0 < w1; i0 += 64 * PF) {
        uint32_t cc[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) { const uint32_t i = i0 + 64u * (uint32_t)k + lane; cc[k] = i < w1 ? L[i] : 0u; }
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const uint32_t i = i0 + 64u * (uint32_t)k + lane;
            if (i0 + 64u * (uint32_t)k >= w1) break;
            const bool valid = i < w1 && i != origin;
            const uint32_t c = cc[k];
            const unsigned long long peers = BWS_PEERS(valid, c);
            if (valid && (peers & ((1ull << lane) - 1ull)) == 0) s_cnt[w][c] += (uint32_t)__popcll(peers);
            rcx_wave_sync();
        }
    }
    __syncthreads();
    const uint32_t osym = L[origin];
    if (tid < 256) {
        uint32_t tot = (tid == osym) ? 1u : 0u;
        for (int ww = 0; ww < BWTI_WAVES; ww++) tot += s_cnt[ww][tid];
        s_tot[tid] = tot;
    }
    __syncthreads();
    if (tid == 0) { uint32_t acc = 0; for (int c = 0; c < 256; c++) { const uint32_t t = s_tot[c]; s_tot[c] = acc; acc += t; } }
    __syncthreads();
    if (tid < 256) {
        uint32_t acc = s_tot[tid] + ((tid == osym) ? 1u : 0u);
        for (int ww = 0; ww < BWTI_WAVES; ww++) { const uint32_t t = s_cnt[ww][tid]; s_cnt[ww][tid] = acc; acc += t; }
    }
    __syncthreads();
    for (uint32_t i0 = w0; i0 < w1; i0 += 64 * PF) {
        uint32_t cc[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) { const uint32_t i = i0 + 64u * (uint32_t)k + lane; cc[k] = i < w1 ? L[i] : 0u; }
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const uint32_t i = i0 + 64u * (uint32_t)k + lane;
            if (i0 + 64u * (uint32_t)k >= w1) break;
            const bool valid = i < w1 && i != origin;
            const uint32_t c = cc[k];
            const unsigned long long peers = BWS_PEERS(valid, c);
            const uint32_t before = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
            uint32_t basec = 0;
            if (valid) basec = s_cnt[w][c];
            rcx_wave_sync();
            if (valid) {
                table[i] = packed ? (basec + before) | (c << 24) : basec + before;
                if (before == 0) s_cnt[w][c] = basec + (uint32_t)__popcll(peers);
            }
            else if (i < w1 && i == origin) table[i] = packed ? s_tot[osym] | (c << 24) : s_tot[osym];
            rcx_wave_sync();
        }
    }
}

// workgroup index -> (slot of the launch, slice): XCD x = index mod 8 works through slots x, x + 8, ..., `SL` workgroups each
__device__ __forceinline__ bool bwti3_map(uint32_t nslots, uint32_t SL, uint32_t& slot, uint32_t& slice)
{
    const uint32_t wg = blockIdx.x, x = wg & 7u, j = wg >> 3;
    slot = (j / SL) * 8u + x; slice = j % SL;
    return slot < nslots;
}

// OCC bytes of LDS nobody uses: they bound the workgroups resident on a CU, and with them the blocks whose tables an XCD chases at once
template <int THREADS, int SLOTS, int OCC>
__global__ __launch_bounds__(THREADS) RCX_SGPR_CAP void k_bwti3_chase(rcx_kargs a, uint32_t block0, uint32_t nslots, bwti3_layout lay, uint32_t shortpark, uint32_t SL)
{
    __shared__ uint8_t s_occ[OCC];
    uint32_t slot, slice;
    if (!bwti3_map(nslots, SL, slot, slice)) return;
    bwti3_block B;
    if (!B.init(a, block0, slot, lay.slot)) return;
    const unsigned tid = threadIdx.x;
    if (shortpark == 0xdeadbeefu) s_occ[tid % OCC] = 1;          // (keeps the array)
    const uint32_t n = B.n, origin = B.origin;
    const uint32_t* table = (const uint32_t*)B.base;
    uint8_t* park = B.base + lay.park;
    uint8_t* park2 = B.base + lay.park2;
    bwti_node* nodes = (bwti_node*)(B.base + lay.nodes);
    const bool packed = n < 0xffffffu;
    const bwti3_geom g(n, origin, shortpark);
    const uint32_t stride = g.stride, M0 = g.M0, pitch = g.pitch, cap = g.cap;
    // c2 / stride without a division: with magic = ceil(2^32 / stride) the high half of c2 * magic IS the quotient as long as
    // c2 * (magic * stride - 2^32) < 2^32, i.e. for c2 < 2^32 / stride -- every packed block (c2 < 2^24, stride <= 2^8); larger blocks divide
    const uint32_t magic = stride > 1 ? (uint32_t)((0x100000000ull + stride - 1) / stride) : 0u;
    const uint32_t per = (g.M + SL - 1) / SL;
    const uint32_t m_lo = slice * per, m_hi = m_lo + per < g.M ? m_lo + per : g.M;
    uint32_t cur[SLOTS], cnt[SLOTS], mid[SLOTS]; bool live[SLOTS];
    // the chain's last 16 bytes, a shift register with the NEWEST byte at the bottom: the walk writes the text backwards, so a group of
    // k <= 16 bytes parked like this is k consecutive bytes of the output, in order, from byte 0 of the slot
    uint32_t p0[SLOTS], p1[SLOTS], p2[SLOTS], p3[SLOTS];
    uint32_t nextm = m_lo + tid;
    auto start = [&](int q) {
        live[q] = false; cnt[q] = 0; cur[q] = 0; mid[q] = 0; p0[q] = 0; p1[q] = 0; p2[q] = 0; p3[q] = 0;
        if (nextm < m_hi) {
            const uint32_t m = nextm; nextm += THREADS;
            live[q] = true; mid[q] = m; cur[q] = m < M0 ? m * stride : origin;
        }
    };
#pragma unroll
    for (int q = 0; q < SLOTS; q++) start(q);
    for (;;) {
        bool any = false;
#pragma unroll
        for (int q = 0; q < SLOTS; q++) any = any || live[q];
        if (!any) break;
        uint32_t v[SLOTS], c2[SLOTS], ch[SLOTS];
#pragma unroll
        for (int q = 0; q < SLOTS; q++) v[q] = live[q] ? table[cur[q]] : 0u;
#pragma unroll
        for (int q = 0; q < SLOTS; q++) {
            if (packed) { ch[q] = v[q] >> 24; c2[q] = v[q] & 0xffffffu; }
            else { c2[q] = v[q]; ch[q] = live[q] ? B.L[cur[q]] : 0u; }
        }
#pragma unroll
        for (int q = 0; q < SLOTS; q++) {
            if (!live[q]) continue;
            uint32_t quo;
            if (stride == 1) quo = c2[q];
            else if (packed) quo = __umulhi(c2[q], magic);
            else quo = c2[q] / stride;
            const bool mk = quo * stride == c2[q] || c2[q] == origin;
            const bool stop = mk || cnt[q] + 1 >= n;
            cur[q] = c2[q];
            p3[q] = RCX_ALIGNBYTE(p3[q], p2[q], 3u); p2[q] = RCX_ALIGNBYTE(p2[q], p1[q], 3u);
            p1[q] = RCX_ALIGNBYTE(p1[q], p0[q], 3u); p0[q] = (p0[q] << 8) | ch[q];
            if (((cnt[q] & 15u) == 15u || stop) && (cnt[q] & ~15u) < pitch) {
                const uint32_t gi = cnt[q] >> 4;
                uint8_t* dst = gi == 0 ? park + (size_t)mid[q] * 16u : park2 + (size_t)mid[q] * (pitch - 16u) + (size_t)(gi - 1u) * 16u;
                *(rcx_u32x4*)dst = rcx_u32x4{p0[q], p1[q], p2[q], p3[q]};
            }
            cnt[q]++;
            if (stop) {
                bwti_node nd; nd.a = cnt[q];
                nd.b = !mk ? BWTI3_NONE : (c2[q] == origin && !g.origin_marked) ? M0 : quo;
                nodes[mid[q]] = nd;
                start(q);
            }
        }
    }
    (void)cap;
}

// node records {emissions, next node} -> {emissions of the super chain before this node, super id | min(emissions, 2^17 - 1) << 15};
// super records {emissions of the super chain, next super node or NONE}.  The node graph is a permutation (place() is one), so a node
// has ONE predecessor: it is read and rewritten by one walker only.  A chain that meets no super node within M steps is a cycle
// without one (not a BWT: the rank kernel will find that origin's cycle does not cover the block).
template <int THREADS, int SLOTS, int OCC>
__global__ __launch_bounds__(THREADS) void k_bwti3_contract(rcx_kargs a, uint32_t block0, uint32_t nslots, bwti3_layout lay, uint32_t shortpark, uint32_t SL)
{
    __shared__ uint8_t s_occ[OCC];
    uint32_t slot, slice;
    if (!bwti3_map(nslots, SL, slot, slice)) return;
    bwti3_block B;
    if (!B.init(a, block0, slot, lay.slot)) return;
    const unsigned tid = threadIdx.x;
    if (shortpark == 0xdeadbeefu) s_occ[tid % OCC] = 1;
    bwti_node* nodes = (bwti_node*)(B.base + lay.nodes);
    bwti_node* sup = (bwti_node*)(B.base + lay.sup);
    const bwti3_geom g(B.n, B.origin, shortpark);
    const uint32_t per = (g.S + SL - 1) / SL;
    const uint32_t s_lo = slice * per, s_hi = s_lo + per < g.S ? s_lo + per : g.S;
    uint32_t m[SLOTS], acc[SLOTS], sid[SLOTS], steps[SLOTS]; bool live[SLOTS];
    uint32_t nexts = s_lo + tid;
    auto start = [&](int q) {
        live[q] = false; m[q] = 0; acc[q] = 0; sid[q] = 0; steps[q] = 0;
        if (nexts < s_hi) { sid[q] = nexts; nexts += THREADS; live[q] = true; m[q] = g.node_of(sid[q]); }
    };
#pragma unroll
    for (int q = 0; q < SLOTS; q++) start(q);
    for (;;) {
        bool any = false;
#pragma unroll
        for (int q = 0; q < SLOTS; q++) any = any || live[q];
        if (!any) break;
        bwti_node r[SLOTS];
#pragma unroll
        for (int q = 0; q < SLOTS; q++) { r[q].a = 0; r[q].b = 0; if (live[q]) r[q] = nodes[m[q]]; }
#pragma unroll
        for (int q = 0; q < SLOTS; q++) {
            if (!live[q]) continue;
            bwti_node w; w.a = acc[q]; w.b = sid[q] | ((r[q].a < 0x1ffffu ? r[q].a : 0x1ffffu) << 15);
            nodes[m[q]] = w;
            acc[q] += r[q].a;
            steps[q]++;
            const uint32_t nx = r[q].b;
            uint32_t ns = BWTI3_NONE; bool end = true;
            if (nx == BWTI3_NONE || nx == g.mstart || steps[q] > g.M) ns = BWTI3_NONE;      // (the list from origin's node comes back to it: cut there)
            else if (g.is_super(nx)) ns = g.super_of(nx);
            else { end = false; m[q] = nx; }
            if (end) { bwti_node sr; sr.a = acc[q]; sr.b = ns; sup[sid[q]] = sr; start(q); }
        }
    }
}

__global__ __launch_bounds__(BWTI_THREADS) void k_bwti3_rank(rcx_kargs a, uint32_t block0, bwti3_layout lay, uint32_t shortpark)
{
    __shared__ uint32_t s_rk[BWTI3_SUPERS + 16];
    __shared__ uint16_t s_next[BWTI3_SUPERS + 16];
    __shared__ uint32_t s_ok;
    bwti3_block B;
    if (!B.init(a, block0, blockIdx.x, lay.slot)) return;
    const unsigned tid = threadIdx.x;
    const bwti_node* sup = (const bwti_node*)(B.base + lay.sup);
    uint32_t* rk = (uint32_t*)(B.base + lay.rk);
    const bwti3_geom g(B.n, B.origin, shortpark);
    const uint32_t S = g.S, NONE16 = 0xffffu;
    for (uint32_t s = tid; s < S; s += BWTI_THREADS) {
        const bwti_node r = sup[s];
        s_rk[s] = r.a;
        s_next[s] = (uint16_t)(r.b == BWTI3_NONE ? NONE16 : r.b);
    }
    __syncthreads();
    constexpr int PER = (BWTI3_SUPERS + 1 + BWTI_THREADS - 1) / BWTI_THREADS;
    for (uint32_t span = 1; span < S; span <<= 1) {
        uint32_t add[PER]; uint16_t nn[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t s = tid + (uint32_t)j * BWTI_THREADS;
            add[j] = 0; nn[j] = (uint16_t)NONE16;
            if (s < S) { const uint32_t nx = s_next[s]; if (nx != NONE16) { add[j] = s_rk[nx]; nn[j] = s_next[nx]; } }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t s = tid + (uint32_t)j * BWTI_THREADS;
            if (s < S && s_next[s] != NONE16) { s_rk[s] += add[j]; s_next[s] = nn[j]; }
        }
        __syncthreads();
    }
    if (tid == 0) s_ok = (s_next[g.sstart] == NONE16 && s_rk[g.sstart] == B.n) ? 1u : 0u;    // the cycle of place() through origin is the whole block
    __syncthreads();
    const bool ok = s_ok != 0;
    if (ok) for (uint32_t s = tid; s < S; s += BWTI_THREADS) rk[s] = s_rk[s];
    if (tid == 0) {
        a.status[B.b] = ok ? RCX_OK : RCX_E_MALFORMED;
        a.out_len[B.b] = ok ? B.n : 0;
        if (a.in_used) a.in_used[B.b] = B.n;
    }
}

// a 16-byte group shifted down by `sh` bytes (the tests' short parking keeps the OLDEST bytes of a group, which sit on top)
__device__ __forceinline__ rcx_u32x4 bwti_shr128(rcx_u32x4 v, uint32_t sh)
{
    uint32_t w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3];
    if (sh & 8u) { w0 = w2; w1 = w3; w2 = 0; w3 = 0; }
    if (sh & 4u) { w0 = w1; w1 = w2; w2 = w3; w3 = 0; }
    const uint32_t bs = sh & 3u;
    return rcx_u32x4{RCX_ALIGNBYTE(w1, w0, bs), RCX_ALIGNBYTE(w2, w1, bs), RCX_ALIGNBYTE(w3, w2, bs), RCX_ALIGNBYTE(0u, w3, bs)};
}

// Step j of the walk writes out[n - 1 - j] (mod.rs:312 read backwards); a node's first step is n - rank(super) + offset.
__global__ __launch_bounds__(BWTI_THREADS) void k_bwti3_emit(rcx_kargs a, uint32_t block0, uint32_t nslots, bwti3_layout lay, uint32_t shortpark, uint32_t parts)
{
    __shared__ uint32_t s_rk[BWTI3_SUPERS + 16];
    __shared__ __align__(16) uint8_t s_tile[BWTI3_TILE + 32];       // 16 bytes of slack on either side: a group that straddles the tile's edge is stored whole
    uint32_t slot, part;
    if (!bwti3_map(nslots, parts, slot, part)) return;
    bwti3_block B;
    if (!B.init(a, block0, slot, lay.slot)) return;
    if (a.status[B.b] != RCX_OK) return;
    const uint32_t n = B.n;
    const uint64_t tlo64 = (uint64_t)part * BWTI3_TILE;
    if (tlo64 >= n) return;
    const uint32_t tlo = (uint32_t)tlo64, thi = n - tlo < BWTI3_TILE ? n : tlo + BWTI3_TILE;       // out[tlo, thi)
    const unsigned tid = threadIdx.x;
    const bwti_node* nodes = (const bwti_node*)(B.base + lay.nodes);
    const uint8_t* park = B.base + lay.park;
    const uint8_t* park2 = B.base + lay.park2;
    const uint32_t* rk = (const uint32_t*)(B.base + lay.rk);
    uint64_t* lng = (uint64_t*)(B.base + lay.lng);
    uint32_t* lcnt = (uint32_t*)(B.base + lay.lcnt);
    const bwti3_geom g(n, B.origin, shortpark);
    for (uint32_t s = tid; s < g.S; s += BWTI_THREADS) s_rk[s] = rk[s];
    __syncthreads();
    // eight nodes a thread and step: the records, then the park slots of those that fall into the tile, are loaded together (one
    // node at a time this loop was two dependent loads per iteration and the kernel 1.7 ms)
    constexpr int U = 8;
    bwti_node rn[U];                                                    // the NEXT step's records: in flight while this step's chains are stored
#pragma unroll
    for (int u = 0; u < U; u++) { const uint32_t m = tid + (uint32_t)u * BWTI_THREADS; rn[u].a = 0; rn[u].b = 0; if (m < g.M) rn[u] = nodes[m]; }
#pragma unroll
    for (int u = 0; u < U; u++) { rn[u].a = RCX_VGPR(rn[u].a); rn[u].b = RCX_VGPR(rn[u].b); }       // (settled here and at the end of every step: see there)
    for (uint32_t mb = 0; mb < g.M; mb += BWTI_THREADS * U) {          // (every thread makes every step: the stores below are wave-wide)
        const uint32_t m0 = mb + tid;
        bwti_node r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            r[u] = rn[u];
            const uint32_t m = m0 + BWTI_THREADS * U + (uint32_t)u * BWTI_THREADS;
            rn[u].a = 0; rn[u].b = 0;
            if (m < g.M) rn[u] = nodes[m];
        }
        uint32_t hi[U], len[U]; bool in[U]; rcx_u32x4 pv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t m = m0 + (uint32_t)u * BWTI_THREADS;
            const uint32_t c = r[u].b >> 15, sid = r[u].b & 0x7fffu;
            len[u] = c < g.cap ? c : g.cap;
            const uint32_t j0 = n - s_rk[sid] + r[u].a;           // the chain's bytes go to out[hi], out[hi - 1], ...
            hi[u] = n - 1u - j0;
            const uint32_t lo = hi[u] + 1u >= len[u] ? hi[u] + 1u - len[u] : 0u;
            in[u] = m < g.M && len[u] != 0 && hi[u] >= tlo && lo < thi;
            if (m < g.M && c > g.cap && hi[u] >= tlo && hi[u] < thi) {      // longer than its slot: the tile that holds its first byte lists it
                const uint32_t k = atomicAdd(lcnt, 1u);
                lng[k] = (uint64_t)m | ((uint64_t)j0 << 32);
            }
            pv[u] = rcx_u32x4{0, 0, 0, 0};
            if (in[u]) pv[u] = *(const rcx_u32x4*)(park + (size_t)m * 16u);
        }
        if (shortpark) {                                         // (a loop of its own: used where it is loaded, each slot was waited for before the next was requested)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t c = r[u].b >> 15;
                if (in[u]) pv[u] = bwti_shr128(pv[u], (c < 16u ? c : 16u) - len[u]);             // (cap = 1: the chain's first byte is the group's last)
            }
        }
        // group k of a chain (its bytes 16k .. 16k + e - 1, parked in output order) is out[hi - 16k - e + 1 ..]: one unaligned store of
        // e <= 16 bytes into the tile (RCX_LDS_STORE16: byte stores under a narrowing EXEC)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t e = len[u] < 16u ? len[u] : 16u;
            const int32_t at = (int32_t)(hi[u] + 1u - e) - (int32_t)tlo + 16;
            const bool fits = in[u] && at >= 0 && at < (int32_t)(BWTI3_TILE + 16u);
            RCX_LDS_STORE16(s_tile + (fits ? at : 0), pv[u][0], pv[u][1], pv[u][2], pv[u][3], fits ? e : 0u);
        }
        bool more = false;
#pragma unroll
        for (int u = 0; u < U; u++) more = more || (in[u] && len[u] > 16u);
        if (__ballot(more)) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t m = m0 + (uint32_t)u * BWTI_THREADS;
                for (uint32_t t0 = 16; __ballot(in[u] && t0 < len[u]); t0 += 16) {
                    const bool on = in[u] && t0 < len[u];
                    rcx_u32x4 v = rcx_u32x4{0, 0, 0, 0};
                    if (on) v = *(const rcx_u32x4*)(park2 + (size_t)m * (g.pitch - 16u) + (t0 - 16u));
                    const uint32_t e = on ? (len[u] - t0 < 16u ? len[u] - t0 : 16u) : 0u;
                    const int32_t at = (int32_t)(hi[u] + 1u - t0 - e) - (int32_t)tlo + 16;
                    const bool fits = on && at >= 0 && at < (int32_t)(BWTI3_TILE + 16u);
                    RCX_LDS_STORE16(s_tile + (fits ? at : 0), v[0], v[1], v[2], v[3], fits ? e : 0u);
                }
            }
        }
        // the next step's records have arrived by now, and the compiler is told so HERE: left to itself it waits for them at the top
        // of the next step with vmcnt(0), right behind the request for the step after -- which is then never in flight during a step
#pragma unroll
        for (int u = 0; u < U; u++) { rn[u].a = RCX_VGPR(rn[u].a); rn[u].b = RCX_VGPR(rn[u].b); }
    }
    __syncthreads();
    const uint32_t tn = thi - tlo;
    uint8_t* dst = B.out + tlo;
    for (uint32_t o = tid * 16u; o < tn; o += BWTI_THREADS * 16u) {
        if (o + 16u <= tn) *(rcx_u32x4_u*)(dst + o) = *(const rcx_u32x4*)(s_tile + 16u + o);
        else for (uint32_t t = o; t < tn; t++) dst[t] = s_tile[16u + t];
    }
}

// the chains on the block's list again, from their node: the bytes beyond the park slot go straight to their places
__global__ __launch_bounds__(256) void k_bwti3_long(rcx_kargs a, uint32_t block0, bwti3_layout lay, uint32_t shortpark)
{
    bwti3_block B;
    if (!B.init(a, block0, blockIdx.x, lay.slot)) return;
    if (a.status[B.b] != RCX_OK) return;
    const uint32_t n = B.n, origin = B.origin;
    const uint32_t count = *(const uint32_t*)(B.base + lay.lcnt);
    if (count == 0) return;
    const uint32_t* table = (const uint32_t*)B.base;
    const uint64_t* lng = (const uint64_t*)(B.base + lay.lng);
    const bool packed = n < 0xffffffu;
    const bwti3_geom g(n, origin, shortpark);
    for (uint32_t k = threadIdx.x; k < count; k += 256) {
        const uint64_t e = lng[k];
        const uint32_t m = (uint32_t)e, j0 = (uint32_t)(e >> 32);
        uint32_t cur = m < g.M0 ? m * g.stride : origin;
        for (uint32_t t = 0; t < n; t++) {
            const uint32_t v = table[cur];
            uint32_t c2; uint8_t ch;
            if (packed) { ch = (uint8_t)(v >> 24); c2 = v & 0xffffffu; }
            else { c2 = v; ch = B.L[cur]; }
            if (t >= g.cap && j0 + t < n) B.out[n - 1u - (j0 + t)] = ch;
            if ((c2 % g.stride) == 0 || c2 == origin) break;
            cur = c2;
        }
    }
}

// compute_inversion_table itself, mod.rs:223-239: table[place(L[origin])] = 0, then table[place(L[i])] = i + 1 for the other i in
// order -- place() hands out a symbol's slots first come, first served, so origin takes the first slot of its symbol.  One
// workgroup per block, a wave per slice of L: count per (wave, symbol), prefix over (symbol major, wave minor), stable scatter
// with the rank among equal bytes of a step from ballots.  Output: a little-endian u32 per entry (4n bytes, slot 4-byte aligned).
__global__ __launch_bounds__(BWTI_THREADS) void k_bwt_inversion_table(rcx_kargs a)
{
    __shared__ uint32_t s_cnt[BWTI_WAVES][256];
    __shared__ uint32_t s_tot[256];
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    const unsigned tid = threadIdx.x, w = RCX_UNI(tid >> 6), lane = tid & 63u;
    const uint8_t* L = a.in_base + a.in_off[b];
    const uint32_t n = (uint32_t)a.in_len[b];
    const uint32_t origin = a.aux ? a.aux[b] : 0u;
    uint32_t* table = (uint32_t*)(a.out_base + a.out_off[b]);
    if (origin >= n || a.out_cap[b] < 4ull * n || ((uintptr_t)table & 3u)) {
        if (tid == 0) {          // input[origin] is an index panic (:230), also for the empty block
            a.status[b] = origin >= n ? RCX_E_MALFORMED : RCX_E_OUTPUT_TOO_SMALL;
            a.out_len[b] = 0; if (a.in_used) a.in_used[b] = n;
        }
        return;
    }
    for (unsigned i = tid; i < BWTI_WAVES * 256; i += BWTI_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t per = ((n + BWTI_WAVES - 1) / BWTI_WAVES + 63u) & ~63u;
    const uint32_t w0 = w * per < n ? w * per : n, w1 = w0 + per < n ? w0 + per : n;
    for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool valid = i < w1 && i != origin;
        const uint32_t c = i < w1 ? L[i] : 0u;
        const unsigned long long peers = BWS_PEERS(valid, c);
        if (valid && (peers & ((1ull << lane) - 1ull)) == 0) s_cnt[w][c] += (uint32_t)__popcll(peers);
        rcx_wave_sync();
    }
    __syncthreads();
    const uint32_t osym = L[origin];
    if (tid < 256) {
        uint32_t tot = tid == osym ? 1u : 0u;
        for (int ww = 0; ww < BWTI_WAVES; ww++) tot += s_cnt[ww][tid];
        s_tot[tid] = tot;
    }
    __syncthreads();
    if (tid == 0) { uint32_t acc = 0; for (int c = 0; c < 256; c++) { const uint32_t t = s_tot[c]; s_tot[c] = acc; acc += t; } }
    __syncthreads();
    if (tid < 256) {
        uint32_t acc = s_tot[tid] + (tid == osym ? 1u : 0u);             // slot 0 of origin's symbol is origin's
        for (int ww = 0; ww < BWTI_WAVES; ww++) { const uint32_t t = s_cnt[ww][tid]; s_cnt[ww][tid] = acc; acc += t; }
    }
    __syncthreads();
    if (tid == 0) table[s_tot[osym]] = 0u;
    for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool valid = i < w1 && i != origin;
        const uint32_t c = i < w1 ? L[i] : 0u;
        const unsigned long long peers = BWS_PEERS(valid, c);
        const uint32_t before = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        uint32_t basec = 0;
        if (valid) basec = s_cnt[w][c];
        rcx_wave_sync();
        if (valid) {
            table[basec + before] = i + 1u;
            if (before == 0) s_cnt[w][c] = basec + (uint32_t)__popcll(peers);
        }
        rcx_wave_sync();
    }
    if (tid == 0) { a.status[b] = RCX_OK; a.out_len[b] = 4ull * n; if (a.in_used) a.in_used[b] = n; }
}
static int launch_bwt_inversion_table(hipStream_t s, rcx_kargs& k)
{
    if (k.nblocks) hipLaunchKernelGGL(k_bwt_inversion_table, dim3(k.nblocks), dim3(BWTI_THREADS), 0, s, k);
    return RCX_RC_OK;
}

static int launch_bwt_inverse(hipStream_t s, rcx_kargs& k, int variant, std::string& err, bool minimal = false)
{
    const uint32_t capx = (variant & 1) ? 1u : BWTI_CAPX;        // variant bit 0: park 16 bytes per walker at most (tests: second chases)
    const bool scatter = (variant & 2) != 0;                     // variant bit 1: the forward chase over the scattered jump table (A/B)
                                                                 // variant bit 2: one workgroup per block from start to end (k_bwt_inverse<2>); bits 4..7: chase geometry (A/B)
    const uint32_t nb = k.nblocks;
    std::vector<uint64_t> h_len(nb);
    if (hipMemcpyAsync(h_len.data(), k.in_len, nb * 8ull, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) { err = "bwt inverse: cannot read in_len"; return RCX_RC_HIP_ERROR; }
    uint64_t maxn = 0;
    for (uint32_t b = 0; b < nb; b++) if (h_len[b] > maxn) maxn = h_len[b];
    if (maxn >= 0xfffffff0ull) { err = "bwt inverse: block too large"; return RCX_RC_BAD_ARG; }
    const uint64_t stride = bwti_slot_bytes(maxn), tbytes = bwti_table_bytes(maxn);
    const uint32_t chunk = nb < BWTI_CHUNK ? nb : BWTI_CHUNK;
    if ((uint64_t)chunk * stride > k.scratch_bytes) { err = "bwt inverse: scratch too small"; return RCX_RC_BAD_ARG; }
    for (uint32_t b0 = 0; b0 < nb; b0 += BWTI_CHUNK) {
        const uint32_t cnt = nb - b0 < BWTI_CHUNK ? nb - b0 : BWTI_CHUNK;
        if (!minimal && !scatter && !(variant & 4)) {
            // grids of (block, slice): XCD x (workgroup index mod 8) works through blocks x, x + 8, ... with `slices` workgroups each
            const uint32_t per_xcd = (cnt + 7u) / 8u;
            const bwti3_layout lay = bwti3_layout_for(maxn);
            const uint32_t parts = (uint32_t)((maxn + BWTI3_TILE - 1) / BWTI3_TILE);
            hipLaunchKernelGGL(k_bwti_table, dim3(cnt), dim3(BWTI_THREADS), 0, s, k, b0, lay.slot, lay.lcnt);
#define BWTI_CHASE(T, Q, O, SLICES) hipLaunchKernelGGL((k_bwti3_chase<T, Q, O>), dim3(8u * per_xcd * (SLICES)), dim3(T), 0, s, k, b0, cnt, lay, capx < BWTI_CAPX ? 1u : 0u, (uint32_t)(SLICES))
            switch ((variant >> 4) & 15) {
            default: BWTI_CHASE(256, 4, 16384, 64); break;             // eight workgroups a CU, 64 slices a block: an XCD chases 4 blocks (4 MiB of tables) at a time
            case 1: BWTI_CHASE(256, 4, 65536, 16); break;
            case 2: BWTI_CHASE(256, 4, 65536, 32); break;
            case 3: BWTI_CHASE(256, 4, 40960, 32); break;
            case 4: BWTI_CHASE(256, 8, 65536, 16); break;
            case 5: BWTI_CHASE(256, 4, 16384, 16); break;
            case 6: BWTI_CHASE(256, 4, 16384, 32); break;
            case 7: BWTI_CHASE(512, 4, 65536, 16); break;
            case 8: BWTI_CHASE(256, 4, 40960, 16); break;
            case 9: BWTI_CHASE(256, 4, 40960, 64); break;
            }
#undef BWTI_CHASE
#define BWTI_CONTRACT(T, Q, O, SLICES) hipLaunchKernelGGL((k_bwti3_contract<T, Q, O>), dim3(8u * per_xcd * (SLICES)), dim3(T), 0, s, k, b0, cnt, lay, capx < BWTI_CAPX ? 1u : 0u, (uint32_t)(SLICES))
            switch ((variant >> 8) & 15) {
            default: BWTI_CONTRACT(256, 1, 16384, 64); break;
            case 1: BWTI_CONTRACT(256, 4, 40960, 8); break;
            case 2: BWTI_CONTRACT(256, 2, 16384, 32); break;
            case 3: BWTI_CONTRACT(256, 1, 16384, 32); break;
            case 4: BWTI_CONTRACT(256, 4, 16384, 16); break;
            case 5: BWTI_CONTRACT(128, 1, 16384, 64); break;
            case 6: BWTI_CONTRACT(256, 1, 40960, 64); break;
            }
#undef BWTI_CONTRACT
            hipLaunchKernelGGL(k_bwti3_rank, dim3(cnt), dim3(BWTI_THREADS), 0, s, k, b0, lay, capx < BWTI_CAPX ? 1u : 0u);
            hipLaunchKernelGGL(k_bwti3_emit, dim3(8u * per_xcd * parts), dim3(BWTI_THREADS), 0, s, k, b0, cnt, lay, capx < BWTI_CAPX ? 1u : 0u, parts);
            hipLaunchKernelGGL(k_bwti3_long, dim3(cnt), dim3(256), 0, s, k, b0, lay, capx < BWTI_CAPX ? 1u : 0u);
        }
        else if (minimal) hipLaunchKernelGGL(k_bwt_inverse<1>, dim3(cnt), dim3(BWTI_THREADS), 0, s, k, b0, stride, tbytes, capx);
        else if (scatter) hipLaunchKernelGGL(k_bwt_inverse<0>, dim3(cnt), dim3(BWTI_THREADS), 0, s, k, b0, stride, tbytes, capx);
        else hipLaunchKernelGGL(k_bwt_inverse<2>, dim3(cnt), dim3(BWTI_THREADS), 0, s, k, b0, stride, tbytes, capx);
    }
    return RCX_RC_OK;
}
