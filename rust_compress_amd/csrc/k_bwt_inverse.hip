// k_bwt_inverse.hip -- batched inverse BWT by list ranking (see k_bwt.hip for the overview).
// Replaces compute_inversion_table + InverseIterator, src/bwt/mod.rs:223-282.
#include <string>
#include <vector>
#include "rcx_dev.h"

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
static uint64_t bwti_slot_bytes(uint64_t max_block)
{
    const uint64_t stride = (max_block + BWTI_MAXMARK - 1) / BWTI_MAXMARK;
    return bwti_table_bytes(max_block) + ((bwti_cap(stride) * (BWTI_MAXMARK + 1) + 255) & ~255ull);
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

static int launch_bwt_inverse(hipStream_t s, rcx_kargs& k, int variant, std::string& err, bool minimal = false)
{
    const uint32_t capx = (variant & 1) ? 1u : BWTI_CAPX;        // variant bit 0: park 16 bytes per walker at most (tests: second chases)
    const bool scatter = (variant & 2) != 0;                     // variant bit 1: the forward chase over the scattered jump table (A/B)
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
        if (minimal) hipLaunchKernelGGL(k_bwt_inverse<1>, dim3(cnt), dim3(BWTI_THREADS), 0, s, k, b0, stride, tbytes, capx);
        else if (scatter) hipLaunchKernelGGL(k_bwt_inverse<0>, dim3(cnt), dim3(BWTI_THREADS), 0, s, k, b0, stride, tbytes, capx);
        else hipLaunchKernelGGL(k_bwt_inverse<2>, dim3(cnt), dim3(BWTI_THREADS), 0, s, k, b0, stride, tbytes, capx);
    }
    return RCX_RC_OK;
}
