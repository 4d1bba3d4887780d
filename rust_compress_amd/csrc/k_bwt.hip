// k_bwt.hip -- batched Burrows-Wheeler transform, forward and inverse, for gfx950.
//
// FORWARD replaces compute_suffixes + TransformIterator (src/bwt/mod.rs:136-204).  The reference sorts
// suffixes by plain byte-slice order (a proper prefix sorts first == an implicit sentinel below every
// byte), so the suffix array is unique and any correct sorter gives bit-identical (L, origin).  Here:
// prefix doubling with the hand-written segmented sorter of k_bwt_sort.hip: round 0 orders the suffixes of a block by a
// 64-bit key of their first <= 10 symbols (alphabet compacted), round r by rank[suffix + h]; groups are sorted where they
// lie (radix partition / LDS bitonic sort / one lane per suffix), a suffix alone in its group is final and never moves again.
//
// INVERSE replaces compute_inversion_table + InverseIterator (src/bwt/mod.rs:223-282).  The reference's
// n-step pointer chase is replaced by list ranking: the jump table is built with a stable wave-parallel
// counting scatter (origin first, exactly the reference's placement order), every `stride`-th slot (and
// origin) is a marked node, up to 4096 walkers per block chase from one marked node to the next
// (4 independent chains per lane in flight), one lane ranks the marked nodes, and the walkers chase again
// writing the text at their final offsets.
#include <string>
#include <vector>
#include "rcx_dev.h"
#include "k_bwt_sort.hip"

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
struct BwtfArgs {
    const uint8_t* in_base; const uint64_t* in_off; const uint64_t* in_len;
    const uint32_t* bstart;          // [nblocks+1] exclusive prefix of lengths (global suffix index base)
    uint32_t nblocks;
};

// Alphabet of the whole batch: hist[0..255] = symbol counts of a 1/64 SAMPLE (only the order-0 entropy estimate uses them),
// hist[256..263] = EXACT 256-bit presence set of every byte of every block (the symbol map must cover all of them).
// Presence is kept in 8 registers per lane (LDS atomics on a skewed text serialise: the first version took 13 ms).
__global__ __launch_bounds__(256) void k_bwtf_hist(BwtfArgs a, uint32_t* hist)
{
    __shared__ uint32_t s_h[256];
    __shared__ uint32_t s_p[8];
    s_h[threadIdx.x] = 0;
    if (threadIdx.x < 8) s_p[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t b = blockIdx.y;
    const uint32_t n = (uint32_t)a.in_len[b];
    const uint8_t* T = a.in_base + a.in_off[b];
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto mark = [&](uint32_t v) {
        const uint32_t bit = 1u << (v & 31u), w = v >> 5;
#pragma unroll
        for (int k = 0; k < 8; k++) m[k] |= (w == (uint32_t)k) ? bit : 0u;
    };
    const uint32_t nch = n >> 4;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nch; c += gridDim.x * blockDim.x) {
        const rcx_u32x4 x = *(const rcx_u32x4_u*)(T + 16ull * c);
#pragma unroll
        for (int d = 0; d < 4; d++)
#pragma unroll
            for (int e = 0; e < 4; e++) mark((x[d] >> (8 * e)) & 0xffu);
        if ((c & 63u) == 0)
#pragma unroll
            for (int d = 0; d < 4; d++)
#pragma unroll
                for (int e = 0; e < 4; e++) atomicAdd(&s_h[(x[d] >> (8 * e)) & 0xffu], 1u);
    }
    if (blockIdx.x == 0) for (uint32_t i = (nch << 4) + threadIdx.x; i < n; i += blockDim.x) { mark(T[i]); atomicAdd(&s_h[T[i]], 1u); }
#pragma unroll
    for (int k = 0; k < 8; k++) if (m[k]) atomicOr(&s_p[k], m[k]);
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
    if (threadIdx.x < 8 && s_p[threadIdx.x]) atomicOr(&hist[256 + threadIdx.x], s_p[threadIdx.x]);
}
// first key of suffix i: block | `nsym` symbols of `bits` bits each, a symbol = 1 + rank of the byte among the bytes that
// occur in the batch (order preserving), 0 = past the end (the reference's implicit sentinel order)
__global__ __launch_bounds__(256) void k_bwtf_init(BwtfArgs a, uint64_t* keys, uint32_t* vals, const uint8_t* map, uint32_t nsym, uint32_t bits, uint32_t plus1)
{
    __shared__ uint32_t s_map[256];
    __shared__ uint16_t s_sym[256 + 16];                          // the symbols of a tile of 256 suffixes and the 16 that follow
    s_map[threadIdx.x] = (uint32_t)map[threadIdx.x] + plus1;      // plus1: the map is the identity and a symbol is byte + 1
    __syncthreads();
    const uint32_t b = blockIdx.y;
    const uint32_t n = (uint32_t)a.in_len[b];
    const uint8_t* T = a.in_base + a.in_off[b];
    const uint32_t g0 = a.bstart[b];
    for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += gridDim.x * 256u) {
        for (uint32_t t = threadIdx.x; t < 256u + 16u; t += 256u) s_sym[t] = (i0 + t < n) ? (uint16_t)s_map[T[i0 + t]] : (uint16_t)0;
        __syncthreads();
        const uint32_t i = i0 + threadIdx.x;
        if (i < n) {
            uint64_t k = b;
            for (uint32_t c = 0; c < nsym; c++) k = (k << bits) | (uint64_t)s_sym[threadIdx.x + c];
            keys[g0 + i] = k;
            vals[g0 + i] = g0 + i;
        }
        __syncthreads();
    }
}
// Round 0, first radix level, fused with the key construction: one workgroup per block reads the TEXT (an LDS tile of symbols
// at a time), builds each suffix's key on the fly, counts and scatters by the key's top 8 bits.  Against k_bwtf_init + k_bws_seed +
// the generic level (which reads the 8-byte keys twice and the 4-byte suffixes once, writes 12 bytes and reads them again to mark
// the small bins) a suffix costs two reads of its text byte and ONE 12-byte write: a bin of one is final, a bin of <= BWS_WAVE is
// written to saA / keyA marked for the dense passes, a larger bin goes to the other buffer and on the list its size asks for.
#define BWS_FTHREADS 1024u      /* threads of k_bws_first: one workgroup per block, so its waves are all the latency hiding a CU gets */
#define BWS_FT 8192u            /* suffixes per LDS tile of k_bws_first */
__global__ __launch_bounds__(BWS_FTHREADS, 8) RCX_SGPR_CAP void k_bws_first(BwsState s, BwtfArgs a, const uint8_t* map, uint32_t nsym, uint32_t bits, uint32_t plus1,
                                                   uint32_t top_shift, uint32_t topn)
{
    __shared__ uint32_t s_map[256];
    __shared__ uint16_t s_sym[BWS_FT + 16];
    __shared__ uint32_t s_h12[4096], s_tot[256], s_beg[256], s_ws[BWS_FTHREADS / 64];
    __shared__ uint8_t s_lut[4096];
    __shared__ uint32_t s_vlo[256], s_vhi[256];                           // smallest and largest top-12 value of a bin
    __shared__ uint32_t s_th[256], s_ts[256], s_tc[256], s_gcur[256];      // the tile's digit counts, their scan, its cursors; the block's cursors
    __shared__ uint16_t s_perm[BWS_FT];
    __shared__ uint8_t s_dig[BWS_FT];
    __shared__ uint32_t s_one;
    const uint32_t b = blockIdx.x, tid = threadIdx.x, wave = RCX_UNI(tid >> 6), lane = tid & 63u;      // (the wave's number is uniform: said so, what is computed from it stays on the scalar unit)
    const uint32_t n = (uint32_t)a.in_len[b];
    if (n == 0) return;
    const uint8_t* T = a.in_base + a.in_off[b];
    const uint32_t g0 = a.bstart[b];
    if (tid < 256) s_map[tid] = (uint32_t)map[tid] + plus1;
    for (uint32_t i = tid; i < 4096u; i += BWS_FTHREADS) s_h12[i] = 0;
    if (tid < 256) { s_tot[tid] = 0; s_vlo[tid] = 0xffffffffu; s_vhi[tid] = 0; }
    if (tid == 0) s_one = 0;
    __syncthreads();
    // A tile: the symbols of suffixes i0 .. i0+BWS_FT-1 and the 16 that follow (0 = past the end).  Its text is REQUESTED (fetch) as
    // one aligned 8-byte word per thread (+ three threads' second word) while the tile before it is worked on, and turned into symbols
    // (tile) when that one is done: read a byte per thread and trip, a tile was nine dependent round trips to memory with every wave
    // of the block waiting at the barrier behind them -- most of this kernel's time.  (An aligned word that holds one byte of the
    // text lies in the text's pages: the bytes of it before or behind the text are read and not used.)
    static_assert(BWS_FT == 8u * BWS_FTHREADS, "a word per thread and tile");
    const uint32_t mis = (uint32_t)((uintptr_t)T & 7u);
    const uint64_t* Tw = (const uint64_t*)(T - mis);
    uint64_t tw0 = 0, tw1 = 0;
    auto fetch = [&](uint32_t i0) {
        tw0 = 0; tw1 = 0;
        if (i0 < n) {
            const uint64_t* q = Tw + (i0 >> 3);
            if (i0 + 8u * tid < n + mis) tw0 = q[tid];
            if (tid < 3u && i0 + BWS_FT + 8u * tid < n + mis) tw1 = q[BWS_FTHREADS + tid];
        }
    };
    auto settle = [&]() {                                      // (before a tile's stores go out: see BwsLocal::settle)
        tw0 = (uint64_t)RCX_VGPR((uint32_t)tw0) | ((uint64_t)RCX_VGPR((uint32_t)(tw0 >> 32)) << 32);
        tw1 = (uint64_t)RCX_VGPR((uint32_t)tw1) | ((uint64_t)RCX_VGPR((uint32_t)(tw1 >> 32)) << 32);
    };
    auto tile = [&](uint32_t i0) {                             // from what fetch(i0) requested
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) {
            const uint32_t t = 8u * tid + j - mis;                                              // (wraps below 0: not a tile position)
            if (t < BWS_FT + 16u) s_sym[t] = (i0 + t < n) ? (uint16_t)s_map[(uint32_t)(tw0 >> (8u * j)) & 0xffu] : (uint16_t)0;
        }
        if (tid < 3u) {
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) {
                const uint32_t t = BWS_FT + 8u * tid + j - mis;
                if (t < BWS_FT + 16u) s_sym[t] = (i0 + t < n) ? (uint16_t)s_map[(uint32_t)(tw1 >> (8u * j)) & 0xffu] : (uint16_t)0;
            }
        }
        __syncthreads();
    };
    // (the key is put together from 32-bit pieces of as many symbols as fit one, a shift-and-or per symbol: built in 64 bits symbol by
    // symbol it was three to four times the vector instructions, and this kernel builds 2^27 of them a pass)
    const uint32_t nper = 32u / bits;
    const uint64_t kblock = nsym * bits < 64u ? (uint64_t)b << (nsym * bits) : 0ull;
    auto key_at = [&](uint32_t t) {
        uint64_t k = 0; uint32_t acc = 0, m = 0;
        for (uint32_t c = 0; c < nsym; c++) {
            acc = (acc << bits) | (uint32_t)s_sym[t + c];
            if (++m == nper) { k = (k << (nper * bits)) | acc; acc = 0; m = 0; }      // (uniform)
        }
        if (m) k = (k << (m * bits)) | acc;
        return k | kblock;
    };
    const uint32_t nshift = top_shift > 8u ? top_shift - 8u : 0u;
    // The first level does not split by the key's top 8 bits (a text's first symbol and a quarter: ~160 bins of very unequal size, most
    // of them above BWS_LMAX and through one or two more radix levels), but into 256 RANGES of its top 12 bits that hold about n / 256
    // suffixes each: bin(v) = 256 * (suffixes with a smaller top-12 value) / n, from the block's own histogram.  Any monotone map
    // of a key prefix keeps the bins in key order; what changes is that the levels below may take no key bit for sorted (the bins
    // are listed with the first level's own shift), and that nearly every bin of a text goes straight to the LDS sorts.
    const uint32_t kbits = top_shift + 8u, b12 = kbits < 12u ? kbits : 12u, sh12 = kbits - b12, m12 = (1u << b12) - 1u;
    // (the key's top b12 bits lie in its first n12 symbols -- two of a text's ten, four of DNA's sixteen: the two passes that only bin
    // a suffix read those, not the whole key)
    const uint32_t n12 = (b12 + bits - 1u) / bits, drop12 = n12 * bits - b12;
    auto top12_at = [&](uint32_t t) {
        uint32_t v = 0;
        for (uint32_t c = 0; c < n12; c++) v = (v << bits) | (uint32_t)s_sym[t + c];
        return (v >> drop12) & m12;
    };
    if (n <= BWS_LMAX) {                                       // a small block: keys and identity order, listed as one group (what k_bws_seed did)
        for (uint32_t i0 = 0; i0 < n; i0 += BWS_FT) {
            fetch(i0); tile(i0);
            for (uint32_t t = tid; t < BWS_FT; t += BWS_FTHREADS) {
                const uint32_t i = i0 + t;
                if (i < n) { s.keyA[g0 + i] = key_at(t); s.saA[g0 + i] = (g0 + i) | (i == 0 ? (BWS_HEAD | (s.par ^ BWS_PAR)) : 0u); }
            }
        }
        if (tid == 0) {
            if (n == 1) { s.saA[g0] = g0 | BWS_HEAD | BWS_FINAL; s.rank[g0] = g0; }
            else if (n > BWS_LWAVE) { const uint32_t q = atomicAdd(&s.cnt[9], 1u); s.localw[q] = BwsSeg{g0, n, top_shift}; }
            else if (n > BWS_WAVE) { const uint32_t q = atomicAdd(&s.cnt[6], 1u); s.local[q] = BwsSeg{g0, n, top_shift}; }
            else if (!bws_dense_ok(g0, n)) { const uint32_t q = atomicAdd(&s.cnt[2], 1u); s.small[q] = BwsSeg{g0, n, 0u}; }
            else bws_flag_dense(s, s.rs, g0, n);
        }
        return;
    }
    // ---- count
    fetch(0);
    for (uint32_t i0 = 0; i0 < n; i0 += BWS_FT) {
        tile(i0);
        fetch(i0 + BWS_FT);
        for (uint32_t t = tid; t < BWS_FT; t += BWS_FTHREADS)
            if (i0 + t < n) atomicAdd(&s_h12[top12_at(t)], 1u);
    }
    fetch(0);                                                  // (the placing pass's first tile, or the one-digit pass's: requested across the scan)
    __syncthreads();
    static_assert(BWS_FTHREADS * 4u == 4096u, "four top-12 values per thread");
    {   // exclusive scan of the 4096 counts (4 per thread, wave scans, the waves' totals), then the bin of every top-12 value
        const uint32_t c0 = s_h12[4 * tid], c1 = s_h12[4 * tid + 1], c2 = s_h12[4 * tid + 2], c3 = s_h12[4 * tid + 3], c = c0 + c1 + c2 + c3;
        const uint32_t inc = rcx_wave_incl_scan(c);
        if (lane == 63) s_ws[wave] = inc;
        __syncthreads();
        uint32_t before = inc - c;
        for (uint32_t w = 0; w < wave; w++) before += s_ws[w];
        const uint32_t e[4] = {before, before + c0, before + c0 + c1, before + c0 + c1 + c2}, cc[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t d = (uint32_t)(((uint64_t)e[q] << 8) / n);
            s_lut[4 * tid + q] = (uint8_t)(d < 255u ? d : 255u);
            if (cc[q]) { atomicAdd(&s_tot[d < 255u ? d : 255u], cc[q]); atomicMin(&s_vlo[d < 255u ? d : 255u], 4u * tid + (uint32_t)q); atomicMax(&s_vhi[d < 255u ? d : 255u], 4u * tid + (uint32_t)q); if (cc[q] == n) s_one = 1; }
        }
    }
    __syncthreads();
    if (s_one) {                                               // every suffix starts with the same digit: the generic levels go on from the next one
        for (uint32_t i0 = 0; i0 < n; i0 += BWS_FT) {
            tile(i0);
            fetch(i0 + BWS_FT); settle();
            for (uint32_t t = tid; t < BWS_FT; t += BWS_FTHREADS) {
                const uint32_t i = i0 + t;
                if (i < n) { s.keyA[g0 + i] = key_at(t); s.saA[g0 + i] = (g0 + i) | (i == 0 ? (BWS_HEAD | (s.par ^ BWS_PAR)) : 0u); }
            }
        }
        if (tid == 0) { const uint32_t q = atomicAdd(&s.cnt[BWS_CLEVEL + 1u], 1u); s.large[1][q] = BwsSeg{g0, n, nshift}; }
        return;
    }
    if (tid < 64) {
        const uint32_t t0 = s_tot[4 * tid], t1 = s_tot[4 * tid + 1], t2 = s_tot[4 * tid + 2], t3 = s_tot[4 * tid + 3];
        const uint32_t ex = rcx_wave_incl_scan(t0 + t1 + t2 + t3) - (t0 + t1 + t2 + t3);
        s_beg[4 * tid] = ex; s_beg[4 * tid + 1] = ex + t0; s_beg[4 * tid + 2] = ex + t0 + t1; s_beg[4 * tid + 3] = ex + t0 + t1 + t2;
    }
    __syncthreads();
    // ---- place.  The suffixes of a tile are first ordered by digit INSIDE the tile (a permutation of tile positions in LDS: the keys
    // are rebuilt from the symbols), then written out in that order: a wave's stores are runs of consecutive addresses, one run per
    // digit present in the tile, instead of 64 stores scattered over the block's bins.  Scattered 8- and 4-byte stores reached HBM
    // as partial-line writes: 8.7 GB written per 1024 blocks for 3.2 GB of keys and suffixes.  (The order inside a bin is whatever the
    // LDS atomics make it: the levels below sort every bin to the end of its key anyway.)
    if (tid < 256) { s_gcur[tid] = s_beg[tid]; s_th[tid] = 0; }
    for (uint32_t i0 = 0; i0 < n; i0 += BWS_FT) {
        tile(i0);
        fetch(i0 + BWS_FT);
        const uint32_t tn = n - i0 < BWS_FT ? n - i0 : BWS_FT;
        for (uint32_t t = tid; t < BWS_FT; t += BWS_FTHREADS) {         // digits of the tile + their counts
            const bool ok = t < tn;
            const uint32_t d = ok ? (uint32_t)s_lut[top12_at(t)] : 0x100u;
            if (ok) s_dig[t] = (uint8_t)d;
            const unsigned long long peers = BWS_PEERS(ok, d);
            if (ok && (uint32_t)__ffsll(peers) - 1u == lane) atomicAdd(&s_th[d], (uint32_t)__popcll(peers));
        }
        __syncthreads();
        if (tid < 64) {                                        // exclusive scan of the tile's counts
            const uint32_t t0 = s_th[4 * tid], t1 = s_th[4 * tid + 1], t2 = s_th[4 * tid + 2], t3 = s_th[4 * tid + 3];
            const uint32_t ex = rcx_wave_incl_scan(t0 + t1 + t2 + t3) - (t0 + t1 + t2 + t3);
            s_ts[4 * tid] = ex; s_ts[4 * tid + 1] = ex + t0; s_ts[4 * tid + 2] = ex + t0 + t1; s_ts[4 * tid + 3] = ex + t0 + t1 + t2;
            s_tc[4 * tid] = ex; s_tc[4 * tid + 1] = ex + t0; s_tc[4 * tid + 2] = ex + t0 + t1; s_tc[4 * tid + 3] = ex + t0 + t1 + t2;
        }
        __syncthreads();
        for (uint32_t t = tid; t < BWS_FT; t += BWS_FTHREADS) {         // tile position -> place in the tile's digit order
            const bool ok = t < tn;
            const uint32_t d = ok ? (uint32_t)s_dig[t] : 0x100u;
            const unsigned long long peers = BWS_PEERS(ok, d);
            const uint32_t leader = (uint32_t)__ffsll(peers) - 1u;
            uint32_t bse = 0;
            if (ok && leader == lane) bse = atomicAdd(&s_tc[d], (uint32_t)__popcll(peers));
            bse = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((leader & 63u) << 2), (int)bse);
            if (ok) s_perm[bse + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull))] = (uint16_t)t;
        }
        __syncthreads();
        settle();
        for (uint32_t q = tid; q < tn; q += BWS_FTHREADS) {
            const uint32_t t = s_perm[q], d = s_dig[t];
            const uint64_t k = key_at(t);
            const uint32_t p = s_gcur[d] + (q - s_ts[d]), g = g0 + i0 + t;
            const uint32_t c = s_tot[d], b0 = s_beg[d], at = g0 + p;
            if (c == 1u) { s.saA[at] = g | BWS_HEAD | BWS_FINAL; s.rank[g] = at; }
            else if (c <= BWS_WAVE) {
                s.saA[at] = g | (p == b0 ? (BWS_HEAD | (s.par ^ BWS_PAR)) : 0u); s.keyA[at] = k;
                if (p == b0 && bws_dense_ok(at, c)) bws_flag_dense(s, s.rs, at, c);
            } else { s.keyB[at] = k; s.saB[at] = g; }
        }
        __syncthreads();
        if (tid < 256) { s_gcur[tid] += s_th[tid]; s_th[tid] = 0; }
    }
    __syncthreads();
    if (tid < 256) {                                           // where each bin goes (waves 0-3 whole: wave-uniform calls)
        const uint32_t c = s_tot[tid], at = g0 + s_beg[tid];
        // the key bits a bin's values share are sorted inside it (all twelve when it holds ONE value, a frequent pair of symbols): the
        // levels below start right under that common prefix
        const uint32_t dv = c ? s_vlo[tid] ^ s_vhi[tid] : 0u;
        const uint32_t unsorted = sh12 + (dv ? 32u - (uint32_t)__clz((int)dv) : 0u);      // bits [0, unsorted) may differ inside the bin
        const uint32_t shift = unsorted > 8u ? unsorted - 8u : 0u;                        // the first digit: the eight highest bits that may differ
        const BwsSeg nx{at, c, shift | (1u << 8)};
        bws_append(s.large[1], &s.cnt[BWS_CLEVEL + 1u], c > BWS_LMAX, nx);
        bws_append(s.local, &s.cnt[6], c > BWS_WAVE && c <= BWS_LWAVE, nx);
        bws_append(s.localw, &s.cnt[9], c > BWS_LWAVE && c <= BWS_LMAX, nx);
        bws_append(s.small, &s.cnt[2], c >= 2u && c <= BWS_WAVE && !bws_dense_ok(at, c), BwsSeg{at, c, 0u});
    }
}
// L[j] = T[SA[j]-1], or T[n-1] where SA[j] == 0 (that j is `origin`), mod.rs:193-203
__global__ void k_bwtf_emit(BwtfArgs a, const uint32_t* sa, uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap, uint32_t* origin)
{
    uint32_t bx, b;
    bws_xcd_grid(bx, b);                                    // T[] is read at random: a block's text stays in one XCD's L2
    const uint32_t n = (uint32_t)a.in_len[b];
    if (out_cap[b] < n) return;
    const uint32_t g0 = a.bstart[b];
    const uint8_t* T = a.in_base + a.in_off[b];
    uint8_t* out = out_base + out_off[b];
    for (uint32_t base = bx * BWS_GCHUNK; base < n; base += gridDim.x * BWS_GCHUNK) {      // eight per thread: see k_bws_gather
        uint32_t i[BWS_GEPT]; uint8_t c[BWS_GEPT];
#pragma unroll
        for (int q = 0; q < BWS_GEPT; q++) { const uint32_t jl = base + (uint32_t)q * 256u + threadIdx.x; i[q] = jl < n ? (sa[g0 + jl] & BWS_IDX) - g0 : 1u; }
#pragma unroll
        for (int q = 0; q < BWS_GEPT; q++) c[q] = T[i[q] == 0 ? n - 1 : i[q] - 1];
#pragma unroll
        for (int q = 0; q < BWS_GEPT; q++) {
            const uint32_t jl = base + (uint32_t)q * 256u + threadIdx.x;
            if (jl >= n) continue;
            out[jl] = c[q];
            if (i[q] == 0 && origin) origin[b] = jl;
        }
    }
}
// compute_suffixes itself, mod.rs:136-166: the sorted suffix array of the block, a little-endian u32 per suffix (4n bytes, the
// slot 4-byte aligned); origin (the j with SA[j] == 0) rides in aux as for the transform.  The sorter's SA words index the pass's
// concatenated text and carry flags: what leaves is the index inside the block.
__global__ void k_bwtf_emit_sa(BwtfArgs a, const uint32_t* sa, uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap, uint32_t* origin)
{
    const uint32_t b = blockIdx.y;
    const uint32_t n = (uint32_t)a.in_len[b];
    if (out_cap[b] < 4ull * n || ((uintptr_t)(out_base + out_off[b]) & 3u)) return;
    const uint32_t g0 = a.bstart[b];
    uint32_t* out = (uint32_t*)(out_base + out_off[b]);
    for (uint32_t jl = blockIdx.x * blockDim.x + threadIdx.x; jl < n; jl += gridDim.x * blockDim.x) {
        const uint32_t i = (sa[g0 + jl] & BWS_IDX) - g0;
        out[jl] = i;
        if (i == 0 && origin) origin[b] = jl;
    }
}
__global__ void k_bwtf_finish(rcx_kargs a, uint32_t sa_words)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks) return;
    const uint64_t n = a.in_len[b];
    if (sa_words) {
        const bool ok = a.out_cap[b] >= 4 * n && (((uintptr_t)(a.out_base + a.out_off[b]) & 3u) == 0 || n == 0);
        a.status[b] = ok ? RCX_OK : RCX_E_OUTPUT_TOO_SMALL;
        a.out_len[b] = ok ? 4 * n : 0;
        if (a.in_used) a.in_used[b] = n;
        if (n == 0 && a.aux) a.aux[b] = 0;
        return;
    }
    const bool ok = a.out_cap[b] >= n;
    a.status[b] = ok ? RCX_OK : RCX_E_OUTPUT_TOO_SMALL;
    a.out_len[b] = ok ? n : 0;
    if (a.in_used) a.in_used[b] = n;
    if (n == 0 && a.aux) a.aux[b] = 0;
}

static inline uint32_t bits_for(uint64_t v) { uint32_t b = 1; while ((1ull << b) <= v && b < 63) b++; return b; }

// Suffixes per sorting pass: a suffix index shares its SA word with three flags, so a pass takes < 2^28 suffixes (1024 blocks of
// 256 KiB); larger batches are sorted pass after pass.  A pass of 2^27 suffixes is the default: the random accesses of a round
// then range over half the address space, which is worth more than the second pass's fixed cost (~1.7 ms of launches and counter
// reads): 1024 x 256 KiB take 38.5 instead of 39.1 ms (text), 18.4 instead of 20.6 ms (DNA); 2^26 is slower again
// (benchmarks/bwt_pass_size.py).  A single block may still use the whole range.
#define BWTF_MAXN 0x0fffffffu
#define BWTF_PASSN 0x08000000u

static uint64_t bwt_forward_scratch_bytes(uint32_t nblocks, uint64_t max_block)
{
    uint64_t N = (uint64_t)nblocks * max_block;
    const uint64_t pass = max_block > (uint64_t)BWTF_PASSN ? max_block : (uint64_t)BWTF_PASSN;     // a pass: at most BWTF_PASSN suffixes, or one longer block
    if (N > pass) N = pass;
    if (N > (uint64_t)BWTF_MAXN) N = BWTF_MAXN;
    // keys 2 x 8N, SA 2 x 4N, rank 4N, group lists, bstart, counters, histogram
    return 28 * N + 5 * (N / BWS_WAVE + nblocks + 1024) * sizeof(BwsSeg) + 2 * (N / 16 + 4096) * sizeof(BwsSeg) + 2 * (N / BWS_LWAVE + nblocks + 1024) * sizeof(BwsSeg) + (uint64_t)(nblocks + 2) * 4 + 4 * (N / 64 + 512) + (N / 256 + nblocks + 1024) + (1ull << 20);
}

// Page-locked words for the per-round read-backs (the counters of a round, the histogram): a copy into pageable memory goes
// through the runtime's staging buffer and costs the host tens of microseconds more per round, with the GPU idle behind it.
// (a pool, not one per host thread: a context may be driven from pool threads that come and go, and page-locked memory held per
//  thread grew with every thread that ever sorted; the pool grows to the number of CONCURRENT callers and is kept for the life of
//  the process -- nothing is handed back at exit, when the runtime may already be gone.)  When page-locked memory is not to be had
// the read-backs land in ordinary memory: slower by the runtime's staging copy, not an error.
#include <mutex>
struct BwtfWords {
    uint32_t* p = nullptr;
    static std::mutex& mu() { static std::mutex m; return m; }
    static std::vector<uint32_t*>& pool() { static std::vector<uint32_t*> v; return v; }
    BwtfWords()
    {
        {
            std::lock_guard<std::mutex> g(mu());
            if (!pool().empty()) { p = pool().back(); pool().pop_back(); }
        }
        if (!p) {
            const size_t bytes = 4 * (64 + BWS_NFLAG) + 4 * 264;
            void* q = nullptr;
            if (hipHostMalloc(&q, bytes, hipHostMallocDefault) == hipSuccess) p = (uint32_t*)q;
            else { (void)hipGetLastError(); p = (uint32_t*)malloc(bytes); }
        }
    }
    ~BwtfWords() { if (p) { std::lock_guard<std::mutex> g(mu()); pool().push_back(p); } }
    BwtfWords(const BwtfWords&) = delete;
    BwtfWords& operator=(const BwtfWords&) = delete;
};

__global__ void k_bwtf_too_large(int32_t* status, uint64_t* out_len, uint64_t* in_used, uint32_t* aux)
{
    if (threadIdx.x == 0) { *status = RCX_E_BWT_BLOCK_TOO_LARGE; *out_len = 0; if (in_used) *in_used = 0; if (aux) *aux = 0; }
}

static int launch_bwt_forward(hipStream_t s, rcx_kargs& k, int variant, std::string& err, bool sa_words = false)
{
    const uint32_t pass_blocks = variant > 0 ? (uint32_t)variant : 0xffffffffu;      // A/B knob: at most `variant` blocks per sorting pass
    const uint32_t nb_all = k.nblocks;
    BwtfWords words_;                                        // (back to the pool when the call returns)
    uint32_t* const pinned = words_.p;
    if (!pinned) { err = "bwt forward: cannot allocate the read-back words"; return RCX_RC_NO_MEMORY; }
    std::vector<uint64_t> h_len(nb_all);
    if (nb_all && (hipMemcpyAsync(h_len.data(), k.in_len, nb_all * 8ull, hipMemcpyDeviceToHost, s) != hipSuccess ||
                   hipStreamSynchronize(s) != hipSuccess)) { err = "bwt forward: cannot read in_len"; return RCX_RC_HIP_ERROR; }
    for (uint32_t lo = 0; lo < nb_all;) {
        // the next pass: as many blocks as stay below BWTF_MAXN suffixes
        uint32_t nb = 0; uint64_t N64 = 0, maxn = 0;
        while (lo + nb < nb_all && nb < pass_blocks && N64 + h_len[lo + nb] <= (uint64_t)(nb ? BWTF_PASSN : BWTF_MAXN)) { N64 += h_len[lo + nb]; if (h_len[lo + nb] > maxn) maxn = h_len[lo + nb]; nb++; }
        if (nb == 0) {
            // a block of 2^28 bytes or more: a limit of this sorter (an SA word keeps four flag bits beside the suffix index), not of the
            // format -- the block gets RCX_E_BWT_BLOCK_TOO_LARGE (include/rcx.h), the rest of the batch is transformed
            hipLaunchKernelGGL(k_bwtf_too_large, dim3(1), dim3(64), 0, s, k.status + lo, k.out_len + lo, k.in_used ? k.in_used + lo : nullptr, k.aux ? k.aux + lo : nullptr);
            lo += 1;
            continue;
        }
        rcx_kargs kk = k;
        kk.in_off += lo; kk.in_len += lo; kk.out_off += lo; kk.out_cap += lo; kk.out_len += lo; kk.status += lo;
        if (kk.in_used) kk.in_used += lo;
        if (kk.aux) kk.aux += lo;
        kk.nblocks = nb;
        std::vector<uint32_t> h_bstart(nb + 1);
        { uint32_t a = 0; for (uint32_t b = 0; b < nb; b++) { h_bstart[b] = a; a += (uint32_t)h_len[lo + b]; } h_bstart[nb] = a; }
        const uint32_t N = (uint32_t)N64;
        if (N) {
            uint8_t* p = (uint8_t*)k.scratch;
            auto carve = [&](size_t bytes) { uint8_t* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
            BwsState st;
            st.keyA = (uint64_t*)carve(8ull * N); st.keyB = (uint64_t*)carve(8ull * N);
            st.saA = (uint32_t*)carve(4ull * N); st.saB = (uint32_t*)carve(4ull * N);
            st.rank = (uint32_t*)carve(4ull * N);
            const size_t nlarge = (size_t)N / BWS_WAVE + nb + 1024, nmid = (size_t)N / 16 + 4096;
            st.large[0] = (BwsSeg*)carve(nlarge * sizeof(BwsSeg)); st.large[1] = (BwsSeg*)carve(nlarge * sizeof(BwsSeg)); st.nlarge = (BwsSeg*)carve(nlarge * sizeof(BwsSeg));
            st.local = (BwsSeg*)carve(nlarge * sizeof(BwsSeg)); st.nlocal = (BwsSeg*)carve(nlarge * sizeof(BwsSeg));
            const size_t nlw = (size_t)N / BWS_LWAVE + nb + 1024;
            st.localw = (BwsSeg*)carve(nlw * sizeof(BwsSeg)); st.nlocalw = (BwsSeg*)carve(nlw * sizeof(BwsSeg));
            st.small = (BwsSeg*)carve(nmid * sizeof(BwsSeg)); st.nsmall = (BwsSeg*)carve(nmid * sizeof(BwsSeg));
            uint32_t* bstart = (uint32_t*)carve(4ull * (nb + 1));
            st.cnt = (uint32_t*)carve(4 * (64 + BWS_NFLAG));
            uint32_t* hist = (uint32_t*)carve(1056); uint8_t* symmap = (uint8_t*)carve(256);
            const size_t nact = ((size_t)N / 64 + 64 + 7) & ~(size_t)7;
            const size_t ndone = (size_t)N / 256 + nb + 64;                    // k_bws_gather's chunk flags (blockDim 256), cleared with act[]
            uint8_t* act0 = (uint8_t*)carve(4 * nact + ndone);
            st.act0 = act0; st.nact = (uint32_t)nact;
            st.gdone = act0 + 4 * nact;
            st.n = N; st.par = 0; st.rs = 0;
            if ((uint64_t)(p - (uint8_t*)k.scratch) > k.scratch_bytes) { err = "bwt forward: scratch too small"; return RCX_RC_BAD_ARG; }
            if (hipMemcpyAsync(bstart, h_bstart.data(), 4ull * (nb + 1), hipMemcpyHostToDevice, s) != hipSuccess) { err = "bwt forward: H2D"; return RCX_RC_HIP_ERROR; }
            BwtfArgs fa{kk.in_base, kk.in_off, kk.in_len, bstart, nb};
            const uint32_t gx = (uint32_t)((maxn + 255) / 256 < 1024 ? (maxn + 255) / 256 : 1024);
            const uint32_t gxg = (uint32_t)((maxn + BWS_GCHUNK - 1) / BWS_GCHUNK < 1024 ? (maxn + BWS_GCHUNK - 1) / BWS_GCHUNK : 1024) + (maxn ? 0u : 1u);   // gather / emit: 2048 suffixes per workgroup
            // Alphabet compaction: the bytes that occur get dense, order-preserving codes, so that more symbols fit the first
            // key when the alphabet is small and skewed (text: 10 symbols of 6 bits instead of 7 of 9; DNA: 16).
            uint32_t nsym = 7, sbits = 9; bool plain_bytes = true;
            uint8_t h_map[256];                                   // (lives as long as the pass: its upload is not waited for on its own)
            {
                uint32_t* const h_hist = pinned + (64 + BWS_NFLAG);
                if (hipMemsetAsync(hist, 0, 1056, s) != hipSuccess) { err = "bwt forward: memset"; return RCX_RC_HIP_ERROR; }
                // (eight 16-byte chunks per thread: a workgroup per 4 KiB ended in 1 M workgroups' worth of atomics on the same 264 words)
                const uint32_t gxh = (uint32_t)((maxn + 32767) / 32768 < 1024 ? (maxn + 32767) / 32768 : 1024);
                hipLaunchKernelGGL(k_bwtf_hist, dim3(gxh ? gxh : 1, nb), dim3(256), 0, s, fa, hist);
                if (hipMemcpyAsync(h_hist, hist, 1056, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                    err = "bwt forward: histogram"; return RCX_RC_HIP_ERROR; }
                uint32_t sigma = 0;
                auto present = [&](int v) { return (h_hist[256 + (v >> 5)] >> (v & 31)) & 1u; };
                for (int v = 0; v < 256; v++) h_map[v] = present(v) ? (uint8_t)(++sigma > 255 ? 255 : sigma) : 0;
                if (sigma < 256) { sbits = bits_for(sigma); nsym = 64 / sbits; if (nsym > 16) nsym = 16; plain_bytes = false; }
                else for (int v = 0; v < 256; v++) h_map[v] = (uint8_t)v;                  // codes 1..256 do not fit a byte: symbol = byte + 1 (9 bits)
                // Not every symbol that fits is worth a place in the first key: behind the first level's 12 bits the key is sorted in
                // 8-bit LSD passes over ALL suffixes, and symbols that fill only part of a last digit cost a whole pass -- the first
                // doubling round sorts them for the suffixes still tied then, which is cheaper (DNA: 12 symbols = 12 + 24 bits, three
                // passes, 8.9 ms; 16 symbols = 12 + 36, five passes, 9.7 ms.  Text's 10 x 6 = 12 + 48 bits fill six digits exactly).
                // Among the top quarter of the symbol counts that fit: a smaller count where its bits fill their digits 5 % better.
                if (nsym * sbits > 20) {
                    uint32_t best = nsym; double beff = 0;
                    for (uint32_t c = nsym; c >= nsym - nsym / 4 && c * sbits > 20; c--) {
                        const uint32_t r = c * sbits - 12, passes = (r + 7) / 8;
                        const double eff = (double)r / (8.0 * passes);
                        if (c == nsym || eff > beff * 1.05) { beff = eff; best = c; }
                    }
                    nsym = best;
                }
                if (const char* e = getenv("RCX_BWT_NSYM")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 2 && v <= nsym) nsym = v; }   // (experiments)
                if (hipMemcpyAsync(symmap, h_map, 256, hipMemcpyHostToDevice, s) != hipSuccess) { err = "bwt forward: symbol map"; return RCX_RC_HIP_ERROR; }
            }
            if (hipMemsetAsync(st.cnt, 0, 4 * (64 + BWS_NFLAG), s) != hipSuccess || hipMemsetAsync(act0, 0, 4 * nact + ndone, s) != hipSuccess) { err = "bwt forward: memset"; return RCX_RC_HIP_ERROR; }
            const uint32_t kbits0 = nsym * sbits, top0 = kbits0 > 8 ? kbits0 - 8 : 0;
            const uint32_t kbits1 = bits_for(maxn), top1 = kbits1 > 8 ? kbits1 - 8 : 0;           // later keys: local rank + 1 <= maxn
            const bool fused_first = top0 > 0;                    // keys of more than 8 bits: the first level is built straight from the text
            if (fused_first) hipLaunchKernelGGL(k_bws_first, dim3(nb), dim3(BWS_FTHREADS), 0, s, st, fa, symmap, nsym, sbits, plain_bytes ? 1u : 0u, top0, top1);
            else {
                hipLaunchKernelGGL(k_bwtf_init, dim3(gx ? gx : 1, nb), dim3(256), 0, s, fa, st.keyA, st.saA, symmap, nsym, sbits, plain_bytes ? 1u : 0u);
                hipLaunchKernelGGL(k_bws_seed, dim3((nb + 255) / 256), dim3(256), 0, s, st, bstart, nb, top0);
            }
            uint32_t h = nsym;
            const uint32_t gdense = (((N / 64 + 2 + 4 * BWS_DW - 1) / (4 * BWS_DW)) + 7u) & ~7u;      // a multiple of 8: see k_bws_dense
            // grid-stride kernels over the group lists: no more workgroups than there can be groups
            auto grid_for = [&](uint32_t min_group, uint32_t per_wg, uint32_t cap) { const uint32_t g = N / min_group / per_wg + 8u; return g < cap ? g : cap; };
            const uint32_t gpart = grid_for(BWS_LMAX, 1, 2048), glw = grid_for(BWS_WAVE, 4, 8192), glg = grid_for(BWS_LWAVE, 1, 4096), gsm = grid_for(BWS_WAVE, 4, 2048);
            bool converged = false;
            // what the round before listed for this one (the host has the counts anyway): a kernel whose lists are empty is not
            // launched -- an empty launch is ~5 us of kernel and ~5 us of dependency gap, nine of them in a late round
            uint32_t n_large = 1, n_local = 1, n_localw = 1, n_small = 1;
            for (int round = 0; round < 64; round++) {
                st.par = (round & 1) ? BWS_PAR : 0u; st.rs = (uint32_t)(round & 1);
                const uint32_t top = round == 0 ? top0 : top1, topn = top1;
                // (the 32-bit key path packs the group id above a 24-bit rank in the wave sorts: blocks of 2^24 bytes and more take the 64-bit one)
                const bool wide = round == 0 || kbits1 > 24;
                if (round) { if (wide) hipLaunchKernelGGL(k_bws_gather<uint64_t>, dim3(gxg, nb), dim3(256), 0, s, st, bstart, nb, h); else hipLaunchKernelGGL(k_bws_gather<uint32_t>, dim3(gxg, nb), dim3(256), 0, s, st, bstart, nb, h); }
                const int levels = (int)((top + 7) / 8) + 1 + ((round == 0 && fused_first) ? 1 : 0);   // (k_bws_first's bins start from the top digit again)
                // (a partition step lists for the next level and for the three local sorts of THIS round; the local sorts and the dense
                //  passes list for the next round)
                const bool do_part = round == 0 || n_large, do_lw = do_part || n_local, do_lg = do_part || n_localw, do_sm = do_part || n_small;
                for (int lv = (round == 0 && fused_first) ? 1 : 0; lv < levels && do_part; lv++) {
                    if (wide) hipLaunchKernelGGL(k_bws_partition<uint64_t>, dim3(gpart), dim3(512), 0, s, st, lv, topn);
                    else hipLaunchKernelGGL(k_bws_partition<uint32_t>, dim3(gpart), dim3(512), 0, s, st, lv, topn);
                }
                if (wide) {
                    if (do_lw) hipLaunchKernelGGL(k_bws_local_wave<uint64_t>, dim3(glw), dim3(256), 0, s, st, topn);
                    if (do_lg) hipLaunchKernelGGL(k_bws_local_wg<uint64_t>, dim3(glg), dim3(256), 0, s, st, topn);
                    if (do_sm) hipLaunchKernelGGL(k_bws_small<uint64_t>, dim3(gsm), dim3(256), 0, s, st, topn);
                    hipLaunchKernelGGL(k_bws_dense<uint64_t>, dim3(gdense), dim3(256), 0, s, st, 0u); hipLaunchKernelGGL(k_bws_dense<uint64_t>, dim3(gdense), dim3(256), 0, s, st, 32u);
                } else {
                    if (do_lw) hipLaunchKernelGGL(k_bws_local_wave<uint32_t>, dim3(glw), dim3(256), 0, s, st, topn);
                    if (do_lg) hipLaunchKernelGGL(k_bws_local_wg<uint32_t>, dim3(glg), dim3(256), 0, s, st, topn);
                    if (do_sm) hipLaunchKernelGGL(k_bws_small<uint32_t>, dim3(gsm), dim3(256), 0, s, st, topn);
                    hipLaunchKernelGGL(k_bws_dense<uint32_t>, dim3(gdense), dim3(256), 0, s, st, 0u); hipLaunchKernelGGL(k_bws_dense<uint32_t>, dim3(gdense), dim3(256), 0, s, st, 32u);
                }
                uint32_t* const hc = pinned;
                // (one synchronisation a round: the counters are copied out, THEN k_bws_round_end turns them into the next round's)
                if (hipMemcpyAsync(hc, st.cnt, 4 * (64 + BWS_NFLAG), hipMemcpyDeviceToHost, s) != hipSuccess) { err = "bwt forward: sync failed"; return RCX_RC_HIP_ERROR; }
                hipLaunchKernelGGL(k_bws_round_end, dim3(1), dim3(256), 0, s, st);
                if (hipStreamSynchronize(s) != hipSuccess) { err = "bwt forward: sync failed"; return RCX_RC_HIP_ERROR; }
                hc[5] = 0;
                for (uint32_t f = 0; f < BWS_NFLAG; f++) hc[5] |= hc[64 + f];
                if (getenv("RCX_BWT_TRACE")) fprintf(stderr, "bwt forward round %d: h=%u unresolved %u (listed: %u large, %u + %u local, %u small)\n", round, h, hc[5], hc[3], hc[7], hc[10], hc[4]);
#ifdef BWS_PROF
                if (getenv("RCX_BWT_TRACE")) {
                    fprintf(stderr, "  k_bws_local_wg phases (ticks >> 8, all waves, cumulative): fill %u count %u scan %u scatter %u heads %u wordscan %u store %u last-barrier %u; passes x waves %u skipped %u groups x waves %u suffixes x waves %u\n",
                            hc[32], hc[33], hc[34], hc[35], hc[36], hc[37], hc[38], hc[39], hc[40], hc[41], hc[42], hc[43]);
                }
#endif
                if (hc[5] == 0) { converged = true; break; }
                if (hc[3] > nlarge || hc[7] > nlarge || hc[6] > nlarge || hc[4] > nmid || hc[9] > nlw || hc[10] > nlw) { err = "bwt forward: group list overflow"; return RCX_RC_HIP_ERROR; }
                n_large = hc[3]; n_small = hc[4]; n_local = hc[7]; n_localw = hc[10];
                std::swap(st.large[0], st.nlarge); std::swap(st.small, st.nsmall); std::swap(st.local, st.nlocal); std::swap(st.localw, st.nlocalw);
                h = round == 0 ? nsym : 2 * h;
            }
            if (!converged) { err = "bwt forward: did not converge"; return RCX_RC_HIP_ERROR; }
            if (sa_words) hipLaunchKernelGGL(k_bwtf_emit_sa, dim3(gx ? gx : 1, nb), dim3(256), 0, s, fa, st.saA, kk.out_base, kk.out_off, kk.out_cap, kk.aux);
            else hipLaunchKernelGGL(k_bwtf_emit, dim3(gxg, nb), dim3(256), 0, s, fa, st.saA, kk.out_base, kk.out_off, kk.out_cap, kk.aux);
        }
        hipLaunchKernelGGL(k_bwtf_finish, dim3((nb + 255) / 256), dim3(256), 0, s, kk, sa_words ? 1u : 0u);
        lo += nb;
    }
    return RCX_RC_OK;
}

