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
#define BWTI_MAXMARK 4096        /* marked nodes per block (<= 4 per thread) */
#define BWTI_CHUNK 1024u         /* blocks per launch: bounds the jump-table scratch */

// per in-flight block: the 4n-byte jump table + BWTI_CAPX n bytes where the walkers park what they emit on the first chase
#define BWTI_CAPX 16u
static uint64_t bwti_table_bytes(uint64_t max_block) { return (max_block * 4 + 255) & ~255ull; }
static uint64_t bwti_slot_bytes(uint64_t max_block)
{
    const uint64_t stride = (max_block + BWTI_MAXMARK - 1) / BWTI_MAXMARK;
    return bwti_table_bytes(max_block) + (((uint64_t)BWTI_CAPX * (stride ? stride : 1) * (BWTI_MAXMARK + 1) + 255) & ~255ull);
}
static uint64_t bwt_inverse_scratch_bytes(uint32_t nblocks, uint64_t max_block)
{
    const uint64_t nb = nblocks < BWTI_CHUNK ? nblocks : BWTI_CHUNK;
    return nb * bwti_slot_bytes(max_block) + 256;
}

// One workgroup (16 waves) per block.
__global__ __launch_bounds__(BWTI_THREADS) void k_bwt_inverse(rcx_kargs a, uint32_t block0, uint64_t table_stride, uint64_t table_bytes)
{
    __shared__ uint32_t s_cnt[BWTI_WAVES][256];       // per-wave symbol counters -> running slots
    __shared__ uint32_t s_next[BWTI_MAXMARK + 1];     // marked node -> next marked node id (or NONE)
    __shared__ uint32_t s_len[BWTI_MAXMARK + 1];      // emissions of that walker
    __shared__ uint32_t s_base[BWTI_MAXMARK + 1];     // chain position of its first emission
    __shared__ uint32_t s_total;
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
            a.status[b] = n == 0 ? RCX_OK : (origin >= n ? RCX_E_MALFORMED : RCX_E_OUTPUT_TOO_SMALL);   // mod.rs:230 index panic
            a.out_len[b] = 0; if (a.in_used) a.in_used[b] = n;
        }
        return;
    }
    // ---- 1. histogram per wave slice (each wave owns a contiguous slice of L), mod.rs:226-228
    for (unsigned i = tid; i < BWTI_WAVES * 256; i += BWTI_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t per = ((n + BWTI_WAVES - 1) / BWTI_WAVES + 63u) & ~63u;
    const uint32_t w0 = w * per < n ? w * per : n, w1 = w0 + per < n ? w0 + per : n;
    for (uint32_t i = w0 + lane; i < w1; i += 64) atomicAdd(&s_cnt[w][L[i]], 1u);
    __syncthreads();
    // exclusive prefix over (symbol major, wave minor); the `origin` element goes first in its symbol (mod.rs:230)
    const uint32_t osym = L[origin];
    if (tid < 256) {
        uint32_t tot = 0;
        for (int ww = 0; ww < BWTI_WAVES; ww++) tot += s_cnt[ww][tid];
        s_base[tid] = tot;                                   // reuse s_base as the 256-bin totals
    }
    __syncthreads();
    if (tid == 0) { uint32_t acc = 0; for (int c = 0; c < 256; c++) { const uint32_t t = s_base[c]; s_base[c] = acc; acc += t; } }
    __syncthreads();
    if (tid < 256) {
        uint32_t acc = s_base[tid] + (tid == osym ? 1u : 0u);     // slot 0 of osym is reserved for origin
        for (int ww = 0; ww < BWTI_WAVES; ww++) { const uint32_t t = s_cnt[ww][tid]; s_cnt[ww][tid] = acc; acc += t; }
    }
    __syncthreads();
    // the origin element itself was counted in its wave's slice: take it out of that slice's budget
    if (tid == 0) {
        table[s_base[osym]] = packed ? (osym << 24) : 0u;         // table[place(L[origin])] = 0 (+ the byte there, see below)
        const uint32_t ow = origin / per;
        for (int ww = (int)ow + 1; ww < BWTI_WAVES; ww++) s_cnt[ww][osym] -= 1u;
    }
    __syncthreads();
    // ---- 2. stable scatter: 64 positions per step, rank among equal bytes by 8 ballots (mod.rs:231-236)
    for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool valid = i < w1 && i != origin;
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
            table[basec + before] = packed ? (i + 1u) | (c << 24) : i + 1u;
            if (before == 0) s_cnt[w][c] = basec + (uint32_t)__popcll(peers);   // group leader advances the counter
        }
        rcx_wave_sync();
    }
    __threadfence_block();
    __syncthreads();
    // ---- 3. list ranking.  marked nodes: every `stride`-th slot, plus origin (id M0 if not already marked)
    uint32_t stride = (n + BWTI_MAXMARK - 1) / BWTI_MAXMARK;
    if (stride < 1) stride = 1;
    const uint32_t M0 = (n + stride - 1) / stride;
    const bool origin_marked = (origin % stride) == 0;
    const uint32_t M = M0 + (origin_marked ? 0u : 1u);
    const uint32_t NONE = 0xffffffffu;
    // First chase: a walker also parks the bytes it emits (up to `cap`, 16 times the mean chain length); once the marked
    // nodes are ranked, a parked chain is COPIED to its place -- only a chain longer than `cap` is chased a second time.
    const uint32_t cap = BWTI_CAPX * stride;
    // each thread owns marked nodes tid, tid+1024, ... (<= 4 + 1), chased 4 at a time
    for (int pass = 0; pass < 2; pass++) {
        for (uint32_t m0 = tid; m0 < M; m0 += 4 * BWTI_THREADS) {
            uint32_t cur[4], cnt[4], wr[4]; bool live[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t m = m0 + q * BWTI_THREADS;
                live[q] = m < M;
                cur[q] = live[q] ? (m < M0 ? m * stride : origin) : 0u;
                cnt[q] = 0;
                wr[q] = (pass == 1 && live[q]) ? s_base[m] : NONE;
                if (pass == 1 && wr[q] == NONE) live[q] = false;          // not reachable from origin
                if (pass == 1 && live[q] && s_len[m] <= cap) {             // parked on the first chase: copy, no second chase
                    const uint8_t* src = park + (size_t)m * cap;
                    const uint32_t len = s_len[m];
                    uint32_t t = 0;
                    for (; t + 16 <= len && wr[q] + t + 16 <= n; t += 16)          // park slots are 16-byte aligned, `out` need not be
                        *(rcx_u32x4_u*)(out + wr[q] + t) = *(const rcx_u32x4*)(src + t);
                    for (; t < len; t++) if (wr[q] + t < n) out[wr[q] + t] = src[t];
                    live[q] = false;
                }
            }
            for (;;) {
                if (!(live[0] || live[1] || live[2] || live[3])) break;
                uint32_t v[4], c2[4]; uint8_t ch[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = live[q] ? table[cur[q]] : 0u;          // 4 jump-table loads in flight
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (packed) { ch[q] = (uint8_t)(v[q] >> 24); v[q] &= 0xffffffu; c2[q] = v[q] ? v[q] - 1u : origin; }
                    else { c2[q] = v[q] ? v[q] - 1u : origin; ch[q] = live[q] ? L[c2[q]] : (uint8_t)0; }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (!live[q]) continue;
                    uint32_t nxt = NONE;
                    bool stop;
                    if (v[q] == 0) stop = true;                           // wrapped: L[origin] was emitted, chain ends
                    else {
                        const bool mk = (c2[q] % stride) == 0 || c2[q] == origin;
                        stop = mk || cnt[q] + 1 >= n;
                        if (mk) nxt = (c2[q] == origin && !origin_marked) ? M0 : c2[q] / stride;
                        cur[q] = c2[q];
                    }
                    if (pass == 1) { if (wr[q] + cnt[q] < n) out[wr[q] + cnt[q]] = ch[q]; }
                    else if (cnt[q] < cap) park[(size_t)(m0 + q * BWTI_THREADS) * cap + cnt[q]] = ch[q];
                    cnt[q]++;
                    if (stop) {
                        live[q] = false;
                        if (pass == 0) { const uint32_t m = m0 + q * BWTI_THREADS; s_next[m] = nxt; s_len[m] = cnt[q]; }
                    }
                }
            }
        }
        __syncthreads();
        if (pass == 0) {
            for (uint32_t m = tid; m < M; m += BWTI_THREADS) s_base[m] = NONE;
            __syncthreads();
            if (tid == 0) {                                               // rank the marked nodes along the chain
                uint32_t m = origin_marked ? origin / stride : M0, pos = 0, steps = 0;
                while (m != NONE && steps <= M && pos < n) { s_base[m] = pos; pos += s_len[m]; m = s_next[m]; steps++; }
                s_total = pos;
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        const bool ok = s_total == n;                                     // else the chain ended early / looped: not a BWT
        a.status[b] = ok ? RCX_OK : RCX_E_MALFORMED;
        a.out_len[b] = ok ? n : 0;
        if (a.in_used) a.in_used[b] = n;
    }
}

static int launch_bwt_inverse(hipStream_t s, rcx_kargs& k, int variant, std::string& err)
{
    (void)variant;
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
        hipLaunchKernelGGL(k_bwt_inverse, dim3(cnt), dim3(BWTI_THREADS), 0, s, k, b0, stride, tbytes);
    }
    return RCX_RC_OK;
}
