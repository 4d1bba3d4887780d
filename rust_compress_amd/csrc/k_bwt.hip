// k_bwt.hip -- batched Burrows-Wheeler transform, forward and inverse, for gfx950.
//
// FORWARD replaces compute_suffixes + TransformIterator (src/bwt/mod.rs:136-204).  The reference sorts
// suffixes by plain byte-slice order (a proper prefix sorts first == an implicit sentinel below every
// byte), so the suffix array is unique and any correct sorter gives bit-identical (L, origin).  Here:
// prefix doubling over the WHOLE batch at once -- one 64-bit key per suffix (block | rank[i] | rank[i+h],
// rank 0 = "past the end"), a device-wide LSD radix sort per round (rocPRIM primitive), re-ranking by
// head flags + max-scan, h = 4, 8, 16, ... until every suffix is alone in its group.
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

__global__ void k_bwtf_init(BwtfArgs a, uint64_t* keys, uint32_t* vals, uint32_t maxn)
{
    const uint32_t b = blockIdx.y;
    const uint32_t n = (uint32_t)a.in_len[b];
    const uint8_t* T = a.in_base + a.in_off[b];
    const uint32_t g0 = a.bstart[b];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t k = b;
#pragma unroll
        for (int c = 0; c < 4; c++) k = (k << 9) | (i + c < n ? (uint64_t)T[i + c] + 1u : 0u);   // 0 = past the end
        keys[g0 + i] = k;
        vals[g0 + i] = g0 + i;
    }
}
// head[j] = j if a new (block, key) group starts at sorted position j, else 0
__global__ void k_bwtf_heads(const uint64_t* keys, uint32_t* head, uint32_t N, uint32_t* ngroups)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    bool h = false;
    if (j < N) { h = (j == 0) || keys[j] != keys[j - 1]; head[j] = h ? j : 0u; }
    const unsigned long long m = __ballot(h);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(ngroups, (uint32_t)__popcll(m));
}
// rank[suffix] = (group start - block start) + 1
__global__ void k_bwtf_rank(const uint64_t* keys, const uint32_t* vals, const uint32_t* gs, const uint32_t* bstart,
                            uint32_t* rank, uint32_t N, uint32_t blk_shift)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const uint32_t b = (uint32_t)(keys[j] >> blk_shift);
    rank[vals[j]] = gs[j] - bstart[b] + 1u;
}
__global__ void k_bwtf_next(const uint64_t* keys, const uint32_t* vals, const uint32_t* rank, const uint32_t* bstart,
                            uint64_t* keys_out, uint32_t N, uint32_t blk_shift_old, uint32_t br, uint32_t h)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const uint32_t b = (uint32_t)(keys[j] >> blk_shift_old);
    const uint32_t g = vals[j];
    const uint32_t e = bstart[b + 1];
    const uint64_t r1 = rank[g];
    const uint64_t r2 = (g + h < e) ? rank[g + h] : 0u;
    keys_out[j] = ((uint64_t)b << (2 * br)) | (r1 << br) | r2;
}
// L[j] = T[SA[j]-1], or T[n-1] where SA[j] == 0 (that j is `origin`), mod.rs:193-203
__global__ void k_bwtf_emit(BwtfArgs a, const uint64_t* keys, const uint32_t* vals, uint32_t N, uint32_t blk_shift,
                            uint8_t* out_base, const uint64_t* out_off, const uint64_t* out_cap, uint32_t* origin)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const uint32_t b = (uint32_t)(keys[j] >> blk_shift);
    const uint32_t g0 = a.bstart[b];
    const uint32_t n = (uint32_t)a.in_len[b];
    if (out_cap[b] < n) return;
    const uint8_t* T = a.in_base + a.in_off[b];
    const uint32_t i = vals[j] - g0, jl = j - g0;
    uint8_t* out = out_base + out_off[b];
    if (i == 0) { out[jl] = T[n - 1]; if (origin) origin[b] = jl; }
    else out[jl] = T[i - 1];
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

static uint64_t bwt_forward_scratch_bytes(uint32_t nblocks, uint64_t max_block)
{
    const uint64_t N = (uint64_t)nblocks * max_block;
    // keys 2x8N, vals 2x4N, rank 4N, heads 4N, bstart, counters, sort/scan temp
    return 32 * N + N / 16 + (uint64_t)(nblocks + 2) * 4 + (64ull << 20);
}

static int launch_bwt_forward(hipStream_t s, rcx_kargs& k, int variant, std::string& err)
{
    (void)variant;
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
    const uint32_t bblk = bits_for(nb), br = bits_for(maxn + 1);
    if (bblk + 2 * br > 64 || bblk + 36 > 64) { err = "bwt forward: block too large for 64-bit keys"; return RCX_RC_BAD_ARG; }
    if (N) {
        // carve scratch
        uint8_t* p = (uint8_t*)k.scratch;
        auto carve = [&](size_t bytes) { uint8_t* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
        uint64_t* keysA = (uint64_t*)carve(8ull * N); uint64_t* keysB = (uint64_t*)carve(8ull * N);
        uint32_t* valsA = (uint32_t*)carve(4ull * N); uint32_t* valsB = (uint32_t*)carve(4ull * N);
        uint32_t* rank = (uint32_t*)carve(4ull * N);  uint32_t* head = (uint32_t*)carve(4ull * N);
        uint32_t* bstart = (uint32_t*)carve(4ull * (nb + 1)); uint32_t* counter = (uint32_t*)carve(256);
        size_t sort_tmp = 0, scan_tmp = 0;
        {
            rocprim::double_buffer<uint64_t> dk(keysA, keysB); rocprim::double_buffer<uint32_t> dv(valsA, valsB);
            rocprim::radix_sort_pairs(nullptr, sort_tmp, dk, dv, N, 0, 64, s);
            rocprim::inclusive_scan(nullptr, scan_tmp, head, head, N, rocprim::maximum<uint32_t>(), s);
        }
        const size_t tmp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
        void* tmp = carve(tmp_bytes);
        if ((uint64_t)(p - (uint8_t*)k.scratch) > k.scratch_bytes) { err = "bwt forward: scratch too small"; return RCX_RC_BAD_ARG; }
        if (hipMemcpyAsync(bstart, h_bstart.data(), 4ull * (nb + 1), hipMemcpyHostToDevice, s) != hipSuccess) { err = "bwt forward: H2D"; return RCX_RC_HIP_ERROR; }
        BwtfArgs fa{k.in_base, k.in_off, k.in_len, bstart, nb};
        const uint32_t gx = (uint32_t)((maxn + 255) / 256 < 1024 ? (maxn + 255) / 256 : 1024);
        hipLaunchKernelGGL(k_bwtf_init, dim3(gx ? gx : 1, nb), dim3(256), 0, s, fa, keysA, valsA, (uint32_t)maxn);
        rocprim::double_buffer<uint64_t> dk(keysA, keysB); rocprim::double_buffer<uint32_t> dv(valsA, valsB);
        uint32_t blk_shift = 36, end_bit = 36 + bblk, h = 4;
        const uint32_t gN = (N + 255) / 256;
        for (int round = 0; round < 40; round++) {
            size_t tb = tmp_bytes;
            if (rocprim::radix_sort_pairs(tmp, tb, dk, dv, N, 0, end_bit, s) != hipSuccess) { err = "bwt forward: radix sort failed"; return RCX_RC_HIP_ERROR; }
            (void)hipMemsetAsync(counter, 0, 4, s);
            hipLaunchKernelGGL(k_bwtf_heads, dim3(gN), dim3(256), 0, s, dk.current(), head, N, counter);
            uint32_t groups = 0;
            (void)hipMemcpyAsync(&groups, counter, 4, hipMemcpyDeviceToHost, s);
            if (hipStreamSynchronize(s) != hipSuccess) { err = "bwt forward: sync failed"; return RCX_RC_HIP_ERROR; }
            if (groups == N) break;                                      // every suffix is alone: SA is final
            tb = tmp_bytes;
            rocprim::inclusive_scan(tmp, tb, head, head, N, rocprim::maximum<uint32_t>(), s);
            hipLaunchKernelGGL(k_bwtf_rank, dim3(gN), dim3(256), 0, s, dk.current(), dv.current(), head, bstart, rank, N, blk_shift);
            hipLaunchKernelGGL(k_bwtf_next, dim3(gN), dim3(256), 0, s, dk.current(), dv.current(), rank, bstart, dk.alternate(), N, blk_shift, br, h);
            dk.swap();                                                   // new keys, same value order
            blk_shift = 2 * br; end_bit = 2 * br + bblk; h *= 2;
        }
        hipLaunchKernelGGL(k_bwtf_emit, dim3(gN), dim3(256), 0, s, fa, dk.current(), dv.current(), N, blk_shift,
                           k.out_base, k.out_off, k.out_cap, k.aux);
    }
    hipLaunchKernelGGL(k_bwtf_finish, dim3((nb + 255) / 256), dim3(256), 0, s, k);
    return RCX_RC_OK;
}

