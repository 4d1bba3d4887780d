// k_bwt.hip -- batched Burrows-Wheeler transform, forward and inverse, for gfx950.
//
// FORWARD replaces compute_suffixes + TransformIterator (src/bwt/mod.rs:136-204).  The reference sorts
// suffixes by plain byte-slice order (a proper prefix sorts first == an implicit sentinel below every
// byte), so the suffix array is unique and any correct sorter gives bit-identical (L, origin).  Here:
// prefix doubling over the WHOLE batch at once -- one 64-bit key per suffix (block | rank[i] | rank[i+h],
// rank 0 = "past the end"), a device-wide LSD radix sort per round (rocPRIM primitive), re-ranking by
// group flags + max-scan, h = 4, 8, 16, ...; a suffix that is alone in its group is written to the suffix
// array and dropped, so round k sorts only what h = 2^(k+1) bytes could not separate.
//
// INVERSE replaces compute_inversion_table + InverseIterator (src/bwt/mod.rs:223-282).  The reference's
// n-step pointer chase is replaced by list ranking: the jump table is built with a stable wave-parallel
// counting scatter (origin first, exactly the reference's placement order), every `stride`-th slot (and
// origin) is a marked node, up to 4096 walkers per block chase from one marked node to the next
// (4 independent chains per lane in flight), one lane ranks the marked nodes, and the walkers chase again
// writing the text at their final offsets.
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <string>
#include "rcx_dev.h"

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
__global__ void k_bwtf_init(BwtfArgs a, uint64_t* keys, uint32_t* vals, const uint8_t* map, uint32_t nsym, uint32_t bits, uint32_t plus1)
{
    __shared__ uint32_t s_map[256];
    s_map[threadIdx.x] = (uint32_t)map[threadIdx.x] + plus1;      // plus1: the map is the identity and a symbol is byte + 1
    __syncthreads();
    const uint32_t b = blockIdx.y;
    const uint32_t n = (uint32_t)a.in_len[b];
    const uint8_t* T = a.in_base + a.in_off[b];
    const uint32_t g0 = a.bstart[b];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t k = b;
        for (uint32_t c = 0; c < nsym; c++) k = (k << bits) | (i + c < n ? (uint64_t)s_map[T[i + c]] : 0u);
        keys[g0 + i] = k;
        vals[g0 + i] = g0 + i;
    }
}
// ---- one refinement round over the still-unresolved suffixes U (sorted by key = block | rank | next rank) ----
// pair[j] = (index of the first element of j's OLD group, index of the first element of j's NEW group), as
// "j if a group starts here else 0" for a component-wise max-scan.  `so`: key >> so identifies the old group.
struct BwtfPair { uint32_t s, g; };
struct BwtfPairMax {
    __device__ __host__ BwtfPair operator()(const BwtfPair& a, const BwtfPair& b) const
    {
        BwtfPair r; r.s = a.s > b.s ? a.s : b.s; r.g = a.g > b.g ? a.g : b.g; return r;
    }
};
struct BwtfFlagOf {                                           // the same pair, computed on the fly as the scan's input
    const uint64_t* keys; uint32_t so;
    __device__ BwtfPair operator()(uint32_t j) const
    {
        const uint64_t k = keys[j], kp = j ? keys[j - 1] : ~k;
        BwtfPair p; p.s = (j && (k >> so) == (kp >> so)) ? 0u : j; p.g = (j && k == kp) ? 0u : j;
        return p;
    }
};
__global__ void k_bwtf_flags(const uint64_t* keys, BwtfPair* pair, uint32_t n, uint32_t so)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t k = keys[j], kp = j ? keys[j - 1] : ~k;
    BwtfPair p; p.s = (j && (k >> so) == (kp >> so)) ? 0u : j; p.g = (j && k == kp) ? 0u : j;
    pair[j] = p;
}
// new rank of element j = rank of its old group + (start of its new group - start of its old group); a new group
// of one element is final: its suffix goes to SA[new rank] and leaves U.
__global__ void k_bwtf_rank(const uint64_t* keys, const uint32_t* vals, const BwtfPair* pair, const uint32_t* bstart,
                            uint32_t* rank, uint32_t* sa, uint32_t* keep, uint32_t n, uint32_t sb, uint32_t br, uint32_t br1, int round0)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t k = keys[j];
    const BwtfPair p = pair[j];
    const uint32_t b = (uint32_t)(k >> sb);
    const uint32_t base = round0 ? bstart[b] : bstart[b] + (uint32_t)((k >> br) & ((1ull << br1) - 1ull));
    const uint32_t nr = base + (p.g - p.s);
    const uint32_t g = vals[j];
    rank[g] = nr;
    const bool single = p.g == j && (j + 1 == n || keys[j + 1] != k);
    if (single) sa[nr] = g;
    keep[j] = single ? 0u : 1u;
}
// compact the survivors and give them their next key: block | new local rank | local rank + 1 of suffix + h (0 = past the end)
__global__ void k_bwtf_next(const uint64_t* keys, const uint32_t* vals, const uint32_t* keep, const uint32_t* pos,
                            const uint32_t* rank, const uint32_t* bstart, uint64_t* keys_out, uint32_t* vals_out,
                            uint32_t n, uint32_t sb, uint32_t br, uint32_t br1, uint32_t h)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || !keep[j]) return;
    const uint32_t b = (uint32_t)(keys[j] >> sb);
    const uint32_t g = vals[j];
    const uint32_t g0 = bstart[b], e = bstart[b + 1];
    const uint64_t r1 = rank[g] - g0;                             // local rank, br1 bits (no sentinel needed here)
    const uint64_t r2 = (g + h < e) ? rank[g + h] - g0 + 1u : 0u;
    const uint32_t o = pos[j];
    keys_out[o] = ((uint64_t)b << (br1 + br)) | (r1 << br) | r2;
    vals_out[o] = g;
}
__global__ void k_bwtf_count(const uint32_t* keep, const uint32_t* pos, uint32_t n, uint32_t* out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = n ? pos[n - 1] + keep[n - 1] : 0u;
}
// L[j] = T[SA[j]-1], or T[n-1] where SA[j] == 0 (that j is `origin`), mod.rs:193-203
__global__ void k_bwtf_emit(BwtfArgs a, const uint32_t* sa, uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap, uint32_t* origin)
{
    const uint32_t b = blockIdx.y;
    const uint32_t n = (uint32_t)a.in_len[b];
    if (out_cap[b] < n) return;
    const uint32_t g0 = a.bstart[b];
    const uint8_t* T = a.in_base + a.in_off[b];
    uint8_t* out = out_base + out_off[b];
    for (uint32_t jl = blockIdx.x * blockDim.x + threadIdx.x; jl < n; jl += gridDim.x * blockDim.x) {
        const uint32_t i = sa[g0 + jl] - g0;
        if (i == 0) { out[jl] = T[n - 1]; if (origin) origin[b] = jl; }
        else out[jl] = T[i - 1];
    }
}
__global__ void k_bwtf_finish(rcx_kargs a)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks) return;
    const uint64_t n = a.in_len[b];
    const bool ok = a.out_cap[b] >= n;
    a.status[b] = ok ? RCX_OK : RCX_E_OUTPUT_TOO_SMALL;
    a.out_len[b] = ok ? n : 0;
    if (a.in_used) a.in_used[b] = n;
    if (n == 0 && a.aux) a.aux[b] = 0;
}

static inline uint32_t bits_for(uint64_t v) { uint32_t b = 1; while ((1ull << b) <= v && b < 63) b++; return b; }

// Blocks per suffix-sort pass: with <= 1024 the block index takes 10 key bits, which keeps the refinement rounds of 256 KiB
// blocks at 6 radix passes (47 key bits) and the scratch at 13 GB; larger batches are sorted 1024 blocks at a time.
#define BWTF_CHUNK 1024u

static uint64_t bwt_forward_scratch_bytes(uint32_t nblocks, uint64_t max_block)
{
    const uint64_t N = (uint64_t)(nblocks < BWTF_CHUNK ? nblocks : BWTF_CHUNK) * max_block;
    // keys 2x8N, vals 2x4N, rank 4N, SA 4N, group pairs 8N, keep 4N, positions 4N, bstart, counters, sort/scan temp
    return 48 * N + N / 16 + (uint64_t)(nblocks + 2) * 4 + (64ull << 20);
}

static int launch_bwt_forward(hipStream_t s, rcx_kargs& k, int variant, std::string& err)
{
    (void)variant;
    if (k.nblocks > BWTF_CHUNK) {
        for (uint32_t lo = 0; lo < k.nblocks; lo += BWTF_CHUNK) {
            rcx_kargs kk = k;
            kk.in_off += lo; kk.in_len += lo; kk.out_off += lo; kk.out_cap += lo; kk.out_len += lo; kk.status += lo;
            if (kk.in_used) kk.in_used += lo;
            if (kk.aux) kk.aux += lo;
            if (kk.n_out) kk.n_out += lo;
            kk.nblocks = k.nblocks - lo < BWTF_CHUNK ? k.nblocks - lo : BWTF_CHUNK;
            const int rc = launch_bwt_forward(s, kk, variant, err);
            if (rc) return rc;
        }
        return RCX_RC_OK;
    }
    const uint32_t nb = k.nblocks;
    std::vector<uint64_t> h_len(nb);
    if (hipMemcpyAsync(h_len.data(), k.in_len, nb * 8ull, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) { err = "bwt forward: cannot read in_len"; return RCX_RC_HIP_ERROR; }
    std::vector<uint32_t> h_bstart(nb + 1);
    uint64_t N64 = 0, maxn = 0;
    for (uint32_t b = 0; b < nb; b++) { h_bstart[b] = (uint32_t)N64; N64 += h_len[b]; if (h_len[b] > maxn) maxn = h_len[b]; }
    h_bstart[nb] = (uint32_t)N64;
    if (N64 >= 0xffffffffull) { err = "bwt forward: batch larger than 4 Gi suffixes"; return RCX_RC_BAD_ARG; }
    const uint32_t N = (uint32_t)N64;
    // key fields: block index (0 .. nb-1), local rank (0 .. maxn-1), local rank + 1 with 0 = past the end (0 .. maxn)
    const uint32_t bblk = bits_for(nb ? nb - 1 : 0), br = bits_for(maxn), br1 = bits_for(maxn ? maxn - 1 : 0);
    if (bblk + br1 + br > 64 || bblk + 36 > 64) { err = "bwt forward: block too large for 64-bit keys"; return RCX_RC_BAD_ARG; }
    uint32_t nsym = 4, sbits = 9; bool plain_bytes = true;
    if (N) {
        // carve scratch
        uint8_t* p = (uint8_t*)k.scratch;
        auto carve = [&](size_t bytes) { uint8_t* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
        uint64_t* keysA = (uint64_t*)carve(8ull * N); uint64_t* keysB = (uint64_t*)carve(8ull * N);
        uint32_t* valsA = (uint32_t*)carve(4ull * N); uint32_t* valsB = (uint32_t*)carve(4ull * N);
        uint32_t* rank = (uint32_t*)carve(4ull * N);  uint32_t* sa = (uint32_t*)carve(4ull * N);
        BwtfPair* pair = (BwtfPair*)carve(8ull * N);
        uint32_t* keep = (uint32_t*)carve(4ull * N);  uint32_t* pos = (uint32_t*)carve(4ull * N);
        uint32_t* bstart = (uint32_t*)carve(4ull * (nb + 1)); uint32_t* counter = (uint32_t*)carve(256);
        uint32_t* hist = (uint32_t*)carve(1056); uint8_t* symmap = (uint8_t*)carve(256);
        size_t sort_tmp = 0, scan_tmp = 0, scan2_tmp = 0;
        {
            rocprim::double_buffer<uint64_t> dk(keysA, keysB); rocprim::double_buffer<uint32_t> dv(valsA, valsB);
            (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, dk, dv, N, 0, 64, s);
            (void)rocprim::inclusive_scan(nullptr, scan_tmp, rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), BwtfFlagOf{keysA, 36u}),
                                          pair, N, BwtfPairMax(), s);
            (void)rocprim::exclusive_scan(nullptr, scan2_tmp, keep, pos, 0u, N, rocprim::plus<uint32_t>(), s);
        }
        size_t tmp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
        if (scan2_tmp > tmp_bytes) tmp_bytes = scan2_tmp;
        void* tmp = carve(tmp_bytes);
        if ((uint64_t)(p - (uint8_t*)k.scratch) > k.scratch_bytes) { err = "bwt forward: scratch too small"; return RCX_RC_BAD_ARG; }
        if (hipMemcpyAsync(bstart, h_bstart.data(), 4ull * (nb + 1), hipMemcpyHostToDevice, s) != hipSuccess) { err = "bwt forward: H2D"; return RCX_RC_HIP_ERROR; }
        BwtfArgs fa{k.in_base, k.in_off, k.in_len, bstart, nb};
        const uint32_t gx = (uint32_t)((maxn + 255) / 256 < 1024 ? (maxn + 255) / 256 : 1024);
        // Alphabet compaction: the bytes that occur get dense, order-preserving codes, so that more symbols fit the first
        // key when the alphabet is small and skewed (text: 7 of 7 bits instead of 4 of 9; DNA: 16).  High-entropy data is
        // resolved by 4 bytes anyway and keeps the shorter key (fewer radix passes).
        {
            uint32_t h_hist[264];
            if (hipMemsetAsync(hist, 0, 1056, s) != hipSuccess) { err = "bwt forward: memset"; return RCX_RC_HIP_ERROR; }
            hipLaunchKernelGGL(k_bwtf_hist, dim3(gx ? gx : 1, nb), dim3(256), 0, s, fa, hist);
            if (hipMemcpyAsync(h_hist, hist, 1056, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                err = "bwt forward: histogram"; return RCX_RC_HIP_ERROR; }
            uint8_t h_map[256]; uint32_t sigma = 0; double H0 = 0;
            auto present = [&](int v) { return (h_hist[256 + (v >> 5)] >> (v & 31)) & 1u; };
            uint64_t sampled = 0;
            for (int v = 0; v < 256; v++) { h_map[v] = present(v) ? (uint8_t)(++sigma > 255 ? 255 : sigma) : 0; sampled += h_hist[v]; }
            if (sigma == 256) for (int v = 0; v < 256; v++) h_map[v] = (uint8_t)v;          // codes 1..256 do not fit a byte: keep byte + 1 below
            for (int v = 0; v < 256; v++) if (h_hist[v]) { const double pr = (double)h_hist[v] / (double)sampled; H0 -= pr * log2(pr); }
            if (!sampled) H0 = 8.0;
            if (sigma < 256 && H0 < 6.0) {
                sbits = bits_for(sigma);
                nsym = (64 - bblk - 1) / sbits; if (nsym > 16) nsym = 16; if (nsym < 4) nsym = 4;
                while (nsym > 4 && nsym * sbits + bblk > 60) nsym--;                      // at most 8 radix passes in round 0
            } else {
                for (int v = 0; v < 256; v++) h_map[v] = (uint8_t)v;                         // plain bytes: symbol = byte + 1 (9 bits) via the +1 below
            }
            plain_bytes = !(sigma < 256 && H0 < 6.0);
            if (hipMemcpyAsync(symmap, h_map, 256, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                err = "bwt forward: symbol map"; return RCX_RC_HIP_ERROR; }
        }
        hipLaunchKernelGGL(k_bwtf_init, dim3(gx ? gx : 1, nb), dim3(256), 0, s, fa, keysA, valsA, symmap, nsym, sbits, plain_bytes ? 1u : 0u);
        rocprim::double_buffer<uint64_t> dk(keysA, keysB); rocprim::double_buffer<uint32_t> dv(valsA, valsB);
        // Round 0 sorts every suffix by its first 4 bytes; each later round sorts only the suffixes whose group
        // is still larger than one (Larsson-Sadakane style discarding) by (group rank, rank of suffix + h).
        uint32_t sb = nsym * sbits, so = nsym * sbits, end_bit = nsym * sbits + bblk, h = nsym, n = N;
        for (int round = 0; round < 40 && n; round++) {
            const uint32_t gn = (n + 255) / 256;
            size_t tb = tmp_bytes;
            if (rocprim::radix_sort_pairs(tmp, tb, dk, dv, n, 0, end_bit, s) != hipSuccess) { err = "bwt forward: radix sort failed"; return RCX_RC_HIP_ERROR; }
            tb = tmp_bytes;                                      // group flags computed inside the scan's loads (no flag array round trip)
            if (rocprim::inclusive_scan(tmp, tb, rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), BwtfFlagOf{dk.current(), so}),
                                        pair, n, BwtfPairMax(), s) != hipSuccess) { err = "bwt forward: scan failed"; return RCX_RC_HIP_ERROR; }
            hipLaunchKernelGGL(k_bwtf_rank, dim3(gn), dim3(256), 0, s, dk.current(), dv.current(), pair, bstart, rank, sa, keep, n, sb, br, br1, round == 0 ? 1 : 0);
            tb = tmp_bytes;
            if (rocprim::exclusive_scan(tmp, tb, keep, pos, 0u, n, rocprim::plus<uint32_t>(), s) != hipSuccess) { err = "bwt forward: scan failed"; return RCX_RC_HIP_ERROR; }
            hipLaunchKernelGGL(k_bwtf_count, dim3(1), dim3(64), 0, s, keep, pos, n, counter);
            hipLaunchKernelGGL(k_bwtf_next, dim3(gn), dim3(256), 0, s, dk.current(), dv.current(), keep, pos, rank, bstart,
                               dk.alternate(), dv.alternate(), n, sb, br, br1, h);
            uint32_t left = 0;
            (void)hipMemcpyAsync(&left, counter, 4, hipMemcpyDeviceToHost, s);
            if (hipStreamSynchronize(s) != hipSuccess) { err = "bwt forward: sync failed"; return RCX_RC_HIP_ERROR; }
            if (getenv("RCX_BWT_TRACE")) fprintf(stderr, "bwt forward round %d: h=%u elements %u -> survivors %u\n", round, h, n, left);
            dk.swap(); dv.swap();
            n = left; sb = br1 + br; so = br; end_bit = br1 + br + bblk; h *= 2;
        }
        if (n) { err = "bwt forward: did not converge"; return RCX_RC_HIP_ERROR; }
        hipLaunchKernelGGL(k_bwtf_emit, dim3(gx ? gx : 1, nb), dim3(256), 0, s, fa, sa, k.out_base, k.out_off, k.out_cap, k.aux);
    }
    hipLaunchKernelGGL(k_bwtf_finish, dim3((nb + 255) / 256), dim3(256), 0, s, k);
    return RCX_RC_OK;
}

