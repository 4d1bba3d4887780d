// k_bwt_sort.hip -- the suffix sorter of the forward BWT (reference: bwt::compute_suffixes, src/bwt/mod.rs:136-166), hand-written.
//
// Prefix doubling, but the groups of a round are sorted WHERE THEY LIE instead of by device-wide sorts of (block | rank | rank)
// keys (the first version: rocPRIM radix sorts + scans, ~250 bytes of HBM traffic per suffix and round).  SA[j] holds a suffix
// (global index) plus three flags: HEAD (a group starts at j), FINAL (the group is this one suffix: its place is final) and the
// parity of the round that made the group.  Round 0 sorts by a 64-bit key of the first <= 10 symbols, round r > 0 by the 32-bit
// key `rank[suffix + h]` (0 = past the end: the reference's "a proper prefix sorts first"), gathered for every non-final
// position before anything is moved.  A group is then sorted by the path that fits its size:
//   * more than 64 suffixes: k_bws_partition, one workgroup per group: MSD radix step on 8 key bits (per-wave histograms in
//     LDS, scatter to the other buffer); bins of more than 64 are groups of the next level, singletons are final at once,
//     everything in between is marked in place for the wave-level sorts;
//   * at most 64, inside a 64-suffix window (aligned, or shifted by 32) -- nearly all groups of a text: k_bws_dense, ONE LANE
//     PER SUFFIX over the whole array: the HEAD bits of a window give every lane its group, a wave-wide bitonic sort over
//     ds_bpermute by (group, key) orders every group of the window at once.  No descriptors, no launch per group;
//   * the few groups of 33..64 that straddle both window grids: k_bws_small, one wave per group, the same wave sort.
// Every path ends the same way: runs of equal keys are the new groups (rank = position of the run's first suffix, written in
// place -- the keys of the round were gathered before), runs of one are FINAL.  Groups that the dense pass cannot take are
// appended to the next round's lists.  Per suffix and round: one gather of a rank (random), one scatter of a rank (random) and
// ~30 bytes of streaming traffic.
#include "rcx_dev.h"

#define BWS_FINAL 0x80000000u
#define BWS_HEAD  0x40000000u
#define BWS_PAR   0x20000000u
#define BWS_RV    0x10000000u            /* on a HEAD word: every member's rank[] is this group's first position (a finished sort wrote them) */
#define BWS_IDX   0x0fffffffu            /* suffix index: batches of < 2^28 suffixes per pass */
#define BWS_WAVE  32u                    /* the largest group left to the wave-level sorts: it always lies inside a 64-suffix window, aligned or shifted by 32 */
#define BWS_LMAX  2048u                  /* the largest group sorted to the end of its key inside LDS (k_bws_local), by a workgroup ... */
#ifndef BWS_LWAVE
#define BWS_LWAVE 256u                   /* ... or, up to this size, by one wave */
#endif

struct BwsSeg { uint32_t start, len, info; };                 // info: key shift of the next radix step | buffer << 8
// counters: [16 + l] the large list of radix level l (large[l & 1]), [2] small list, [3] next round's large list, [4] next round's small list,
// [6] local list (groups of BWS_WAVE+1 .. BWS_LWAVE: sorted in LDS by one wave), [7] next round's; [9] / [10] the same for the
// groups of BWS_LWAVE+1 .. BWS_LMAX (a workgroup each)
struct BwsState {
    uint64_t* keyA; uint64_t* keyB;          // u64 keys (round 0); the u32 keys of later rounds use the first half of each
    uint32_t* saA; uint32_t* saB;            // saA is the suffix array proper, saB the partition steps' other buffer
    uint32_t* rank;
    BwsSeg* large[2]; BwsSeg* small; BwsSeg* nlarge; BwsSeg* nsmall; BwsSeg* local; BwsSeg* nlocal; BwsSeg* localw; BwsSeg* nlocalw;
    uint32_t* cnt;
    uint32_t n;                               // suffixes in this pass
    uint32_t par;                             // parity bit groups made in THIS round carry (BWS_PAR or 0)
    uint8_t* act0; uint32_t nact;             // "a group for the dense passes lies in this window": four arrays of nact bytes, [round parity * 2
    uint32_t rs;                              // + grid], one byte per 64-suffix window (grid 0: aligned, grid 1: shifted by 32); rs = round & 1
    uint8_t* gdone;                           // k_bws_gather: "every suffix of this workgroup's chunk is FINAL" (it stays so)
};

template <class K> __device__ __forceinline__ K* bws_keys(const BwsState& s, int buf) { return (K*)(buf ? s.keyB : s.keyA); }
__device__ __forceinline__ uint32_t* bws_sa(const BwsState& s, int buf) { return buf ? s.saB : s.saA; }
// a group the dense passes sort: it lies inside a 64-suffix window, either aligned or shifted by 32
__device__ __forceinline__ bool bws_dense_ok(uint32_t a, uint32_t len)
{
    const uint32_t z = a + len - 1u;
    return len <= BWS_WAVE && ((a >> 6) == (z >> 6) || ((a + 32u) >> 6) == ((z + 32u) >> 6));
}

#define BWS_CLEVEL 16u                   /* cnt[16 + l]: groups on the list of radix level l (large[l & 1]) */
// Tell the dense passes of round parity `set` about the group [a, a + len) (which bws_dense_ok): the aligned window if it lies
// inside one (the aligned pass runs first and takes it), else the shifted one.
__device__ __forceinline__ void bws_flag_dense(const BwsState& s, uint32_t set, uint32_t a, uint32_t len)
{
    // (the four arrays are one allocation and the address is COMPUTED: four pointers indexed by a lane's value are a vector load from
    // the kernel arguments, and the wait for it -- vmcnt(0), one counter for loads and stores on gfx9 -- sat out every scattered
    // saA / rank store the wave had just issued)
    const uint32_t z = a + len - 1u;
    const bool aligned = (a >> 6) == (z >> 6);
    s.act0[(size_t)(set * 2u + (aligned ? 0u : 1u)) * s.nact + (aligned ? (a >> 6) : ((a + 32u) >> 6))] = 1;
}
// Append `seg` to a list for the lanes with `want`: ONE atomic per wave (a single word takes ~88 atomics per microsecond on
// this chip: one atomic per group made the first version's passes take 40 ms whatever else they did).  Wave-uniform call.
__device__ __forceinline__ void bws_append(BwsSeg* list, uint32_t* counter, bool want, const BwsSeg& seg)
{
    const unsigned long long m = __ballot(want);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll(m) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(leader << 2), (int)base);
    if (want) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = seg;
}
// "Some group is still unresolved": 1024 flag words (cnt[64 ..]), each wave stores to its own -- a single word that every wave
// reads or writes is a hot spot of the memory system just like an atomic.  The host ORs them after the round.
#define BWS_NFLAG 1024u
__device__ __forceinline__ void bws_flag_unresolved(const BwsState& s)
{
    s.cnt[64u + ((blockIdx.x * 8u + (threadIdx.x >> 6)) & (BWS_NFLAG - 1u))] = 1u;
}
// A finished run of equal keys [a, a + len) in saA (for the lanes with `is`): hand it to the next round unless the dense passes
// find it by themselves.  Wave-uniform call.
__device__ __forceinline__ void bws_new_group(const BwsState& s, bool is, uint32_t a, uint32_t len, uint32_t top_shift)
{
    if (__ballot(is)) { if ((threadIdx.x & 63u) == 0) bws_flag_unresolved(s); }
    const bool listed = is && !bws_dense_ok(a, len);
    if (is && !listed) bws_flag_dense(s, s.rs ^ 1u, a, len);
    bws_append(s.nlarge, &s.cnt[3], listed && len > BWS_LMAX, BwsSeg{a, len, top_shift});
    bws_append(s.nlocal, &s.cnt[7], listed && len > BWS_WAVE && len <= BWS_LWAVE, BwsSeg{a, len, top_shift});
    bws_append(s.nlocalw, &s.cnt[10], listed && len > BWS_LWAVE && len <= BWS_LMAX, BwsSeg{a, len, top_shift});
    bws_append(s.nsmall, &s.cnt[4], listed && len <= BWS_WAVE, BwsSeg{a, len, 0u});
}

// x of lane ^ J2.  Partners at distance 1, 2, 4, 8 come over DPP (quad permutes; row shifts + a select): VALU work -- a
// ds_bpermute holds the CU's LDS crossbar for ~16 cycles, and a wave sort would issue 84 of them.  Distances 16 and 32 keep it.
template <int J2>
__device__ __forceinline__ uint32_t bws_xchg(uint32_t x, uint32_t lane)
{
    if (J2 == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    if (J2 == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false);        // quad_perm [2,3,0,1]
    if (J2 == 4 || J2 == 8) {
        const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x100 + J2, 0xf, 0xf, false);   // row_shl: lane i <- i + J2
        const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + J2, 0xf, 0xf, false);   // row_shr: lane i <- i - J2
        return (lane & (uint32_t)J2) ? dn : up;
    }
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ (uint32_t)J2) << 2), (int)x);
}
template <class K, int K2, int J2>
__device__ __forceinline__ void bws_cmpx(uint32_t lane, uint32_t& c0, uint32_t& klo, uint32_t& khi, uint32_t& val, uint32_t& was)
{
    const uint32_t oc = sizeof(K) == 8 ? bws_xchg<J2>(c0, lane) : 0u;          // u32 keys carry the group in their top bits
    const uint32_t ol = bws_xchg<J2>(klo, lane);
    const uint32_t oh = sizeof(K) == 8 ? bws_xchg<J2>(khi, lane) : 0u;
    const uint32_t ov = bws_xchg<J2>(val | (was << 31), lane);
    // is the partner's element smaller than mine? (ties by suffix index: deterministic)
    const bool ol_lt = oc < c0 || (oc == c0 && (oh < khi || (oh == khi && (ol < klo || (ol == klo && (ov & BWS_IDX) < val)))));
    const bool lower = (lane & (uint32_t)J2) == 0, up = (lane & (uint32_t)K2) == 0;
    const bool take = (lower == up) ? ol_lt : !ol_lt;          // the lower lane of an ascending pair keeps the smaller element
    if (take) { c0 = oc; klo = ol; khi = oh; val = ov & BWS_IDX; was = ov >> 31; }
}
// Wave-wide bitonic sort of one element per lane by (c0, key, suffix): 21 compare-exchange steps.  `was` rides along.
template <class K>
__device__ __forceinline__ void bws_wave_sort(uint32_t lane, uint32_t& c0, uint32_t& klo, uint32_t& khi, uint32_t& val, uint32_t& was)
{
#define BWS_S(K2, J2) bws_cmpx<K, K2, J2>(lane, c0, klo, khi, val, was);
    BWS_S(2, 1)
    BWS_S(4, 2) BWS_S(4, 1)
    BWS_S(8, 4) BWS_S(8, 2) BWS_S(8, 1)
    BWS_S(16, 8) BWS_S(16, 4) BWS_S(16, 2) BWS_S(16, 1)
    BWS_S(32, 16) BWS_S(32, 8) BWS_S(32, 4) BWS_S(32, 2) BWS_S(32, 1)
    BWS_S(64, 32) BWS_S(64, 16) BWS_S(64, 8) BWS_S(64, 4) BWS_S(64, 2) BWS_S(64, 1)
#undef BWS_S
}
// After the sort: the run of equal (c0, key) a lane belongs to, as [rs, re) in lanes; rhead: the lane opens its run.
template <class K>
__device__ __forceinline__ void bws_wave_runs(uint32_t lane, uint32_t c0, uint32_t klo, uint32_t khi, bool& rhead, uint32_t& rs, uint32_t& re)
{
    const int pa = (int)(((lane + 63u) & 63u) << 2);
    const uint32_t pc = sizeof(K) == 8 ? (uint32_t)__builtin_amdgcn_ds_bpermute(pa, (int)c0) : c0;
    const uint32_t pl = (uint32_t)__builtin_amdgcn_ds_bpermute(pa, (int)klo);
    const uint32_t ph = sizeof(K) == 8 ? (uint32_t)__builtin_amdgcn_ds_bpermute(pa, (int)khi) : 0u;
    rhead = lane == 0 || pc != c0 || pl != klo || ph != khi;
    const unsigned long long rh = __ballot(rhead);
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const unsigned long long rb = rh & le, ra = rh & ~le;
    rs = 63u - (uint32_t)__clzll(rb);
    re = ra ? (uint32_t)__ffsll(ra) - 1u : 64u;
}

// A (chunk, block) grid whose workgroups read or write at random inside their block's arrays: workgroups go to the 8 XCDs round-robin in
// dispatch order (x fastest), so dispatched as they are every XCD pulls every block's lines into its own L2 -- up to eight fetches of
// each line.  XCD x takes the x-th eighth of the grid instead: the workgroups resident on it sit in one or two blocks.
#ifndef BWS_XCD_GRID
#define BWS_XCD_GRID 1
#endif
#define BWS_GEPT 8                       /* suffixes per thread and step of k_bws_gather / k_bwtf_emit */
#define BWS_GCHUNK (256u * BWS_GEPT)
__device__ __forceinline__ void bws_xcd_grid(uint32_t& bx, uint32_t& by)
{
    const uint32_t gx = gridDim.x, total = gx * gridDim.y;
    uint32_t f = blockIdx.y * gx + blockIdx.x;
    if (BWS_XCD_GRID && (total & 7u) == 0) f = (f & 7u) * (total >> 3) + (f >> 3);
    bx = f % gx; by = f / gx;
}

// ---- keys of a doubling round: key[j] = local rank + 1 of (suffix at j) + h, 0 = past the end of its block ------------------
template <class K>
__global__ __launch_bounds__(256) void k_bws_gather(BwsState s, const uint32_t* bstart, uint32_t nblocks, uint32_t h)
{
    __shared__ uint32_t s_open;
    uint32_t bx, b;
    bws_xcd_grid(bx, b);
    const uint32_t g0 = bstart[b], e = bstart[b + 1];
    const uint32_t j0 = g0 + bx * BWS_GCHUNK;
    if (j0 >= e) return;
    // A text is mostly sorted after two or three rounds, and what is left lies in runs of the suffix array: a chunk all of whose
    // suffixes are FINAL says so once, and later rounds read one byte of it instead of its SA words.  The chunks of a block, and of
    // consecutive blocks, get consecutive distinct slots.
    uint8_t* done = s.gdone + j0 / BWS_GCHUNK + b;
    if (*done) return;
    if (threadIdx.x == 0) s_open = 0;
    __syncthreads();
    K* key = (K*)s.keyA;
    bool open = false;
    // eight suffixes per thread, their loads in flight together: with one per thread a launch was a million workgroups of one dependent
    // load -> load -> store chain each, and the 2048 workgroup slots of the GPU turned over no faster than that chain's latency
    for (uint32_t base = j0; base < e; base += gridDim.x * BWS_GCHUNK) {
        uint32_t v[BWS_GEPT], r[BWS_GEPT];
#pragma unroll
        for (int q = 0; q < BWS_GEPT; q++) { const uint32_t j = base + (uint32_t)q * 256u + threadIdx.x; v[q] = j < e ? s.saA[j] : BWS_FINAL; }
#pragma unroll
        for (int q = 0; q < BWS_GEPT; q++) {
            const uint32_t g = (v[q] & BWS_IDX) + h;
            const bool in = !(v[q] & BWS_FINAL) && g < e;
            r[q] = in ? s.rank[g] - g0 + 1u : 0u;
        }
#pragma unroll
        for (int q = 0; q < BWS_GEPT; q++) {
            if (v[q] & BWS_FINAL) continue;
            open = true;
            key[base + (uint32_t)q * 256u + threadIdx.x] = r[q];
        }
    }
    if (open) s_open = 1u;
    __syncthreads();
    if (threadIdx.x == 0 && !s_open) *done = 1;
}

// ---- round 0 seed: every block is one group, identity order -----------------------------------------------------------------
__global__ void k_bws_seed(BwsState s, const uint32_t* bstart, uint32_t nblocks, uint32_t top_shift)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t a = bstart[b], len = bstart[b + 1] - a;
    if (len == 0) return;
    if (len == 1) { s.saA[a] = a | BWS_HEAD | BWS_FINAL; s.rank[a] = a; return; }
    // the seed groups carry the parity of "the round before round 0" so that the dense passes take the small ones
    s.saA[a] |= BWS_HEAD | (s.par ^ BWS_PAR);
    if (len > BWS_LMAX) { const uint32_t i = atomicAdd(&s.cnt[BWS_CLEVEL], 1u); s.large[0][i] = BwsSeg{a, len, top_shift}; }
    else if (len > BWS_LWAVE) { const uint32_t i = atomicAdd(&s.cnt[9], 1u); s.localw[i] = BwsSeg{a, len, top_shift}; }
    else if (len > BWS_WAVE) { const uint32_t i = atomicAdd(&s.cnt[6], 1u); s.local[i] = BwsSeg{a, len, top_shift}; }
    else if (!bws_dense_ok(a, len)) { const uint32_t i = atomicAdd(&s.cnt[2], 1u); s.small[i] = BwsSeg{a, len, 0u}; }
    else bws_flag_dense(s, s.rs, a, len);
}

// ---- groups of more than BWS_LMAX suffixes: one MSD radix step (8 key bits) per launch, one workgroup per group ---------------
// A bin of at most BWS_LMAX suffixes leaves the level structure: k_bws_local sorts it to the end of its key inside LDS.

// what became of bin `d` of group sg (count c, first place a): the list it goes to.  Wave-uniform call (64 bins at a time).
__device__ __forceinline__ void bws_route_bin(const BwsState& s, BwsSeg* lnext, uint32_t* clnext, bool done, uint32_t c, uint32_t a,
                                              uint32_t shift, int dst, uint32_t top_shift)
{
    bws_new_group(s, done && c >= 2u, a, c, top_shift);
    const BwsSeg nx{a, c, (shift > 8u ? shift - 8u : 0u) | ((uint32_t)dst << 8)};
    bws_append(lnext, clnext, !done && c > BWS_LMAX, nx);
    bws_append(s.local, &s.cnt[6], !done && c > BWS_WAVE && c <= BWS_LWAVE, nx);
    bws_append(s.localw, &s.cnt[9], !done && c > BWS_LWAVE && c <= BWS_LMAX, nx);
    bws_append(s.small, &s.cnt[2], !done && c >= 2u && c <= BWS_WAVE && !bws_dense_ok(a, c), BwsSeg{a, c, 0u});
}
// the element pass after the scatter: a singleton is final; a bin of equal keys (every bit used) is a finished group; a bin of
// <= BWS_WAVE goes to saA / keyA marked for the wave-level sorts; a larger one stays where it is for the next level.
template <class K>
__device__ __forceinline__ void bws_mark(const BwsState& s, const BwsSeg& sg, uint32_t p, K k, uint32_t g, uint32_t c, uint32_t beg, bool done, int dst)
{
    if (c > BWS_WAVE && !done) return;
    const uint32_t a = sg.start + p;
    if (c == 1u) { s.saA[a] = g | BWS_HEAD | BWS_FINAL; s.rank[g] = a; }
    else if (done) { s.saA[a] = g | (p == beg ? (BWS_HEAD | BWS_RV | s.par) : 0u); s.rank[g] = sg.start + beg; }
    else {
        s.saA[a] = g | (p == beg ? (BWS_HEAD | (s.par ^ BWS_PAR)) : 0u);
        bws_keys<K>(s, 0)[a] = k;
        if (p == beg && bws_dense_ok(a, c)) bws_flag_dense(s, s.rs, a, c);          // this round's dense passes take it
    }
}
// digit peers of a wave: the lanes (among `ok`) that hold the same digit as mine.  Per digit bit: nm = -bit (v_bfe_i32), the
// ballot of the bit (v_cmp into vcc), and peers &= ~(ballot ^ nm) on both halves (one v_bitop3_b32 each, truth table 0x90 =
// a & ~(b ^ c)): 5 issue slots per bit.  Hand-written: the compiler's version of the same C took 9-11 instructions per bit, and
// the sorter's LDS passes are made of this.
#ifndef BWS_PEERS
__device__ __forceinline__ unsigned long long bws_peers(bool ok, uint32_t d)
{
    const unsigned long long okm = __ballot(ok);
    uint32_t plo = (uint32_t)okm, phi = (uint32_t)(okm >> 32), n0, n1;
#define BWS_PB(B, NA, NB)                                           \
        "v_cmp_ne_u32_e32 vcc, 0, %[" NA "]\n\t"                     \
        "v_bfe_i32 %[" NB "], %[d], " #B " + 1, 1\n\t"               \
        "s_nop 0\n\t"                                               \
        "v_bitop3_b32 %[plo], %[plo], vcc_lo, %[" NA "] bitop3:0x90\n\t" \
        "v_bitop3_b32 %[phi], %[phi], vcc_hi, %[" NA "] bitop3:0x90\n\t"
    asm volatile(
        "v_bfe_i32 %[n0], %[d], 0, 1\n\t"
        BWS_PB(0, "n0", "n1") BWS_PB(1, "n1", "n0") BWS_PB(2, "n0", "n1") BWS_PB(3, "n1", "n0")
        BWS_PB(4, "n0", "n1") BWS_PB(5, "n1", "n0") BWS_PB(6, "n0", "n1")
        "v_cmp_ne_u32_e32 vcc, 0, %[n1]\n\t"
        "s_nop 1\n\t"
        "v_bitop3_b32 %[plo], %[plo], vcc_lo, %[n1] bitop3:0x90\n\t"
        "v_bitop3_b32 %[phi], %[phi], vcc_hi, %[n1] bitop3:0x90\n\t"
        : [plo] "+v"(plo), [phi] "+v"(phi), [n0] "=&v"(n0), [n1] "=&v"(n1)
        : [d] "v"(d)
        : "vcc");
#undef BWS_PB
    return ((unsigned long long)phi << 32) | plo;
}
#define BWS_PEERS bws_peers
#endif

#ifndef BWS_PU
#define BWS_PU 4u                       /* chunks of 512 suffixes a partition step has in flight */
#endif
template <class K>
__global__ __launch_bounds__(512) RCX_SGPR_CAP void k_bws_partition(BwsState s, int level, uint32_t top_shift)
{
    __shared__ uint32_t s_hist[8][256];           // per wave; after the scan: the wave's next free place in each bin
    __shared__ uint32_t s_tot[256], s_beg[256];
    __shared__ uint32_t s_one;
    const BwsSeg* list = s.large[level & 1];
    BwsSeg* lnext = s.large[(level + 1) & 1];
    uint32_t* clnext = &s.cnt[BWS_CLEVEL + level + 1];      // (a counter per level: two that alternate wanted a memset between the launches)
    const uint32_t nseg = s.cnt[BWS_CLEVEL + level];
    const uint32_t tid = threadIdx.x, wave = RCX_UNI(tid >> 6), lane = tid & 63u;
    {
        for (uint32_t e = blockIdx.x; e < nseg; e += gridDim.x) {
            const BwsSeg sg = list[e];
            uint32_t shift = sg.info & 0xffu;
            const int src = (int)((sg.info >> 8) & 1u), dst = src ^ 1;
            const K* ks = bws_keys<K>(s, src) + sg.start; K* kd = bws_keys<K>(s, dst) + sg.start;
            const uint32_t* ss = bws_sa(s, src) + sg.start; uint32_t* sd = bws_sa(s, dst) + sg.start;
            for (;;) {
                for (uint32_t i = tid; i < 8 * 256; i += 512) ((uint32_t*)s_hist)[i] = 0;
                if (tid == 0) s_one = 0;
                __syncthreads();
                // (BWS_PU chunks of 512 keys requested together: a chunk per trip was one round trip to memory per 512 suffixes, with
                // eight waves a workgroup and few workgroups -- these groups are the large ones -- to hide it)
                for (uint32_t i0 = 0; i0 < sg.len; i0 += 512u * BWS_PU) {
                    K kk[BWS_PU];
#pragma unroll
                    for (uint32_t u = 0; u < BWS_PU; u++) { const uint32_t i = i0 + 512u * u + tid; kk[u] = i < sg.len ? ks[i] : (K)0; }
#pragma unroll
                    for (uint32_t u = 0; u < BWS_PU; u++) {
                        if (i0 + 512u * u >= sg.len) break;                // (uniform)
                        const bool ok = i0 + 512u * u + tid < sg.len;
                        const uint32_t d = ok ? (uint32_t)(kk[u] >> shift) & 0xffu : 0x100u;      // text digits are skewed: lanes with the same digit count once
                        const unsigned long long peers = BWS_PEERS(ok, d);
                        if (ok && (uint32_t)__ffsll(peers) - 1u == lane) atomicAdd(&s_hist[wave][d], (uint32_t)__popcll(peers));
                    }
                }
                __syncthreads();
                if (tid < 256) {
                    uint32_t t = 0;
#pragma unroll
                    for (int w = 0; w < 8; w++) t += s_hist[w][tid];
                    s_tot[tid] = t;
                    if (t == sg.len) s_one = 1;
                }
                __syncthreads();
                if (!s_one || shift == 0u) break;
                shift = shift > 8u ? shift - 8u : 0u;
                __syncthreads();
            }
            const bool done = shift == 0u;                     // every key bit used after this step: a bin is a group of equal keys
            if (tid < 64) {                            // exclusive scan of the 256 totals: 4 per lane + a wave scan
                const uint32_t t0 = s_tot[4 * tid], t1 = s_tot[4 * tid + 1], t2 = s_tot[4 * tid + 2], t3 = s_tot[4 * tid + 3];
                const uint32_t ex = rcx_wave_incl_scan(t0 + t1 + t2 + t3) - (t0 + t1 + t2 + t3);
                s_beg[4 * tid] = ex; s_beg[4 * tid + 1] = ex + t0; s_beg[4 * tid + 2] = ex + t0 + t1; s_beg[4 * tid + 3] = ex + t0 + t1 + t2;
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t o = s_beg[tid];
#pragma unroll
                for (int w = 0; w < 8; w++) { const uint32_t c = s_hist[w][tid]; s_hist[w][tid] = o; o += c; }
            }
            __syncthreads();
            // the scatter: BWS_PU chunks at a time, the next batch's keys and suffixes requested before this batch's stores go out and
            // waited for (settle) before them too
            K ka[BWS_PU], kb[BWS_PU]; uint32_t ga[BWS_PU], gb[BWS_PU];
            auto request = [&](uint32_t i0, K* kk, uint32_t* gg) {
#pragma unroll
                for (uint32_t u = 0; u < BWS_PU; u++) {
                    const uint32_t i = i0 + 512u * u + tid;
                    kk[u] = 0; gg[u] = 0;
                    if (i < sg.len) { kk[u] = ks[i]; gg[u] = ss[i]; }
                }
            };
            auto settle = [&](K* kk, uint32_t* gg) {
#pragma unroll
                for (uint32_t u = 0; u < BWS_PU; u++) {
                    if (sizeof(K) == 8) kk[u] = (K)((uint64_t)RCX_VGPR((uint32_t)kk[u]) | ((uint64_t)RCX_VGPR((uint32_t)((uint64_t)kk[u] >> 32)) << 32));
                    else kk[u] = (K)RCX_VGPR((uint32_t)kk[u]);
                    gg[u] = RCX_VGPR(gg[u]);
                }
            };
            request(0, ka, ga);
            for (uint32_t i0 = 0; i0 < sg.len; i0 += 512u * BWS_PU) {
                request(i0 + 512u * BWS_PU, kb, gb);
                uint32_t pp[BWS_PU];
#pragma unroll
                for (uint32_t u = 0; u < BWS_PU; u++) {
                    pp[u] = 0;
                    if (i0 + 512u * u >= sg.len) break;                    // (uniform)
                    const bool ok = i0 + 512u * u + tid < sg.len;
                    const uint32_t d = ok ? (uint32_t)(ka[u] >> shift) & 0xffu : 0x100u;
                    const unsigned long long peers = BWS_PEERS(ok, d);
                    const uint32_t leader = (uint32_t)__ffsll(peers) - 1u;
                    uint32_t bse = 0;
                    if (ok && leader == lane) bse = atomicAdd(&s_hist[wave][d], (uint32_t)__popcll(peers));
                    bse = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((leader & 63u) << 2), (int)bse);
                    pp[u] = bse + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
                }
                settle(kb, gb);
#pragma unroll
                for (uint32_t u = 0; u < BWS_PU; u++) {
                    if (i0 + 512u * u + tid < sg.len) {
                        const K k = ka[u];
                        const uint32_t d = (uint32_t)(k >> shift) & 0xffu, g = ga[u] & BWS_IDX, p = pp[u];
                        // the group lies in buffer B: a bin that is finished or goes to the dense passes is written to saA / keyA at once
                        // (from buffer A that would overwrite suffixes other threads have yet to read: the pass below does it then)
                        if (src == 1 && (s_tot[d] <= BWS_WAVE || done)) bws_mark<K>(s, sg, p, k, g, s_tot[d], s_beg[d], done, dst);
                        else { kd[p] = k; sd[p] = g; }
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < BWS_PU; u++) { ka[u] = kb[u]; ga[u] = gb[u]; }
            }
            __syncthreads();
            if (src == 0) {
                for (uint32_t p0 = 0; p0 < sg.len; p0 += 512u * BWS_PU) {
                    K kk[BWS_PU]; uint32_t gg[BWS_PU];
#pragma unroll
                    for (uint32_t u = 0; u < BWS_PU; u++) { const uint32_t p = p0 + 512u * u + tid; kk[u] = 0; gg[u] = 0; if (p < sg.len) { kk[u] = kd[p]; gg[u] = sd[p]; } }
                    settle(kk, gg);                                            // (one wait for the batch, before the first chunk's stores)
#pragma unroll
                    for (uint32_t u = 0; u < BWS_PU; u++) {
                        const uint32_t p = p0 + 512u * u + tid;
                        if (p < sg.len) { const uint32_t d = (uint32_t)(kk[u] >> shift) & 0xffu; bws_mark<K>(s, sg, p, kk[u], gg[u], s_tot[d], s_beg[d], done, dst); }
                    }
                }
            }
            if (tid < 256) bws_route_bin(s, lnext, clnext, done, s_tot[tid], sg.start + s_beg[tid], shift, dst, top_shift);
            __syncthreads();
        }
    }
}

// ---- groups of BWS_WAVE+1 .. BWS_LMAX suffixes: sorted to the END of their key inside LDS, one pass over HBM -----------------
// A team (one wave for <= BWS_LWAVE suffixes, else the workgroup's four waves) loads the group's keys and suffixes into LDS and
// sorts a 16-bit permutation by the key bits the levels above have not used: stable LSD radix passes of 8 bits (a wave owns a
// contiguous quarter of the group, so wave-major order is group order; lanes with the same digit are ranked by ballots, as in the
// partition step), passes whose digit is the same for the whole group are skipped.  Then the runs of equal keys are read off a
// bitmap of run heads and saA / rank are written once: this round is over for the group.
// The new groups a wave finds are queued in LDS and appended to the next round's lists 64+ at a time.  Appending them as they
// turn up (one atomic with return per 64-suffix chunk that holds a listed group) made the finishing phase 70 % of these
// kernels: the wait for the atomic is an s_waitcnt vmcnt(0), and that also waits for the chunk's scattered saA / rank stores.
#define BWS_QCAP 120u                     /* (with the rest of k_bws_local_wg's LDS: 32 granules of 1280 bytes, four workgroups per CU) */
struct BwsQueue {
    uint32_t* q; uint32_t n; uint32_t ts;                     // q: LDS, BWS_QCAP x 2 words (start, len | list << 28); n: entries waiting (wave-uniform); ts: top_shift
    // wave-uniform call: the lanes with `is` hand in the group [a, a + len) made this round
    __device__ __forceinline__ void push(const BwsState& s, bool is, uint32_t a, uint32_t len, uint32_t top_shift)
    {
        const uint32_t lane = threadIdx.x & 63u;
        if (__ballot(is)) { if (lane == 0) bws_flag_unresolved(s); }
        const bool listed = is && !bws_dense_ok(a, len);
        if (is && !listed) bws_flag_dense(s, s.rs ^ 1u, a, len);
        const unsigned long long m = __ballot(listed);
        if (!m) return;
        ts = top_shift;
        if (n + 64u > BWS_QCAP) flush(s);
        const uint32_t tag = len > BWS_LMAX ? 0u : len > BWS_LWAVE ? 2u : len > BWS_WAVE ? 1u : 3u;
        if (listed) {
            const uint32_t i = n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            q[2 * i] = a; q[2 * i + 1] = len | (tag << 28);                  // (a group is shorter than 2^28: BWS_IDX)
        }
        n += (uint32_t)__popcll(m);
        rcx_wave_sync();
    }
    __device__ __forceinline__ void flush(const BwsState& s)
    {
        const uint32_t lane = threadIdx.x & 63u;
        rcx_wave_sync();
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool have = i < n;
            const uint32_t a = have ? q[2 * i] : 0u, lt = have ? q[2 * i + 1] : 0u;
            const uint32_t tag = lt >> 28, len = lt & 0x0fffffffu;
            const BwsSeg seg{a, len, tag == 3u ? 0u : ts};
            bws_append(s.nlarge, &s.cnt[3], have && tag == 0u, seg);
            bws_append(s.nlocal, &s.cnt[7], have && tag == 1u, seg);
            bws_append(s.nlocalw, &s.cnt[10], have && tag == 2u, seg);
            bws_append(s.nsmall, &s.cnt[4], have && tag == 3u, seg);
        }
        n = 0;
        rcx_wave_sync();
    }
};

// -DBWS_PROF: where a team's time goes (s_memtime ticks per phase, summed over the waves into cnt[32 ..]; RCX_BWT_TRACE prints them)
#ifdef BWS_PROF
#define BWS_LAP0() do { tl = __builtin_readcyclecounter(); } while (0)
#define BWS_LAP(slot) do { const uint64_t n__ = __builtin_readcyclecounter(); pf[slot] += n__ - tl; tl = n__; } while (0)
#define BWS_CNT(slot, v) do { pf[slot] += (v); } while (0)
#else
#define BWS_LAP0() do { } while (0)
#define BWS_LAP(slot) do { } while (0)
#define BWS_CNT(slot, v) do { } while (0)
#endif
template <class K, int NW>
struct BwsLocal {
#ifdef BWS_PROF
    uint64_t pf[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl = 0;
    __device__ void report(const BwsState& s, uint32_t base) const
    {
        if (lane == 0) for (int q = 0; q < 12; q++) atomicAdd(&s.cnt[base + q], (uint32_t)(pf[q] >> (q < 8 ? 8 : 0)));
    }
#endif
    static constexpr uint32_t CAP = NW == 1 ? BWS_LWAVE : BWS_LMAX;
    static constexpr uint32_t MAXSTEP = CAP / (64u * NW);
    K* key; uint32_t* val; uint16_t* pa; uint16_t* pb; uint32_t* hist;      // hist: [NW][256]
    uint32_t* bits; uint32_t* misc;                                          // bits[CAP / 32 + 2]: hist + 128, free once the passes are over (the
                                                                             // run lookup's two scan arrays take hist[0 .. 128)); misc[8] (NW > 1)
    uint32_t t, w, lane;                                                     // thread, wave and lane inside the team
    BwsQueue* Q;                                                             // the calling wave's queue of new groups

    __device__ __forceinline__ void sync() const { if (NW == 1) rcx_wave_sync(); else __syncthreads(); }

    // The NEXT group's keys and suffixes are requested while this one is sorted, and before this one's results are stored: on gfx9 a
    // wait for a load also sits out every OLDER store (one counter), so a group that loaded its input after the previous group's
    // scattered stores waited for those to drain at its first barrier.
    // A wave-uniform load (a group's descriptor, its first SA word) is a vector load all the same, and the compiler moves its result
    // to scalar registers AT ONCE -- behind an s_waitcnt vmcnt(0) that sits out everything in flight, the prefetch itself and the
    // previous group's scattered stores.  Such values are loaded through a lane's index (RCX_VGPR) and stay in vector registers
    // until the next fill has waited for the prefetch anyway: pfirst, and the descriptor of the group after the next (pend()).
    K pk[MAXSTEP]; uint32_t pv[MAXSTEP]; uint32_t pfirst;
    uint32_t qs, ql, qi;
    __device__ __forceinline__ void pend(const BwsSeg* list, uint32_t idx, bool have)
    {
        qs = 0; ql = 0; qi = 0;
        if (have) { const uint32_t* p = (const uint32_t*)(list + idx) + RCX_VGPR(0u); qs = p[0]; ql = p[1]; qi = p[2]; }
    }
    // What prefetch() and pend() requested has arrived (it was requested a whole sort ago): said HERE, before this group's stores go
    // out, because the compiler cannot count the stores of a loop and would wait for vmcnt(0) -- the loads and every store after
    // them -- where the next group first touches these registers.
    __device__ __forceinline__ void settle()
    {
#pragma unroll
        for (uint32_t k = 0; k < MAXSTEP; k++) {
            if (sizeof(K) == 8) pk[k] = (K)((uint64_t)RCX_VGPR((uint32_t)pk[k]) | ((uint64_t)RCX_VGPR((uint32_t)((uint64_t)pk[k] >> 32)) << 32));
            else pk[k] = (K)RCX_VGPR((uint32_t)pk[k]);
            pv[k] = RCX_VGPR(pv[k]);
        }
        pfirst = RCX_VGPR(pfirst); qs = RCX_VGPR(qs); ql = RCX_VGPR(ql); qi = RCX_VGPR(qi);
    }
    __device__ __forceinline__ void prefetch(const BwsState& s, const BwsSeg sg)
    {
        const uint32_t T = 64u * NW;
        const int src = (int)((sg.info >> 8) & 1u);
        const K* ks = bws_keys<K>(s, src) + sg.start;
        const uint32_t* ss = bws_sa(s, src) + sg.start;
#pragma unroll
        for (uint32_t k = 0; k < MAXSTEP; k++) {
            const uint32_t i = t + k * T;
            pk[k] = 0; pv[k] = 0;
            if (i < sg.len) { pk[k] = ks[i]; pv[k] = ss[i]; }
        }
        pfirst = ss[RCX_VGPR(0u)];
    }

    // sorts the group whose input prefetch() has requested and returns the group after it (len 0: none), whose input it requests;
    // list[idx2] (if `have2`) is the group after that one
    __device__ BwsSeg run(const BwsState& s, const BwsSeg sg, uint32_t top_shift, const BwsSeg* list, uint32_t idx2, bool have2)
    {
        const uint32_t len = sg.len, T = 64u * NW;
        const uint32_t shift = sg.info & 0xffu;                              // bits [0, shift + 8) of the key are still unsorted
        const bool whole = (pfirst & BWS_RV) != 0;                           // a group an earlier round's sort made (not a bin of this round's radix levels)
        BWS_LAP0(); BWS_CNT(10, 1); BWS_CNT(11, len);
#pragma unroll
        for (uint32_t k = 0; k < MAXSTEP; k++) {
            const uint32_t i = t + k * T;
            if (i < len) { key[i] = pk[k]; val[i] = pv[k] & BWS_IDX; pa[i] = (uint16_t)i; }
        }
        const BwsSeg nx{RCX_UNI(qs), RCX_UNI(ql), RCX_UNI(qi)};              // (it came with this group's input)
        if (nx.len) prefetch(s, nx);
        pend(list, idx2, have2);
        const uint32_t cs = (((len + NW - 1u) / NW) + 63u) & ~63u;           // a wave's contiguous share
        const uint32_t w0 = w * cs, w1 = (w0 + cs < len) ? w0 + cs : len;
        sync();
        BWS_LAP(0);
        for (uint32_t sh = 0; sh < shift + 8u; sh += 8) {
            uint32_t* myh = hist + 256u * w;
#pragma unroll
            for (int q = 0; q < 4; q++) myh[lane + 64 * q] = 0;
            if (NW > 1 && t == 0) misc[0] = 0;
            rcx_wave_sync();
            // ---- count: digit, peers, rank among the peers; the wave's histogram.  A step's LDS round trips (permutation entry -> that
            // key's digit byte -> the histogram) are not waited for one by one: the entries of all steps are read, then the digits, and
            // the histogram takes each step's count with a returning add -- what comes back is the number of this wave's EARLIER elements
            // with the digit, so that the scatter below is one independent read per element (cursors advanced step by step were
            // MAXSTEP dependent round trips more).
            uint32_t inf[MAXSTEP];                                           // digit (0x100: not an element) | place among the wave's elements with the digit << 9 (12 bits) | entry << 21
            {
                uint32_t ev[MAXSTEP], dv[MAXSTEP], bf[MAXSTEP], ld[MAXSTEP];
                const uint8_t* kb = (const uint8_t*)key + (sh >> 3);         // (sh is a multiple of 8: the digit is a byte of the little-endian key)
#pragma unroll
                for (uint32_t st = 0; st < MAXSTEP; st++) { const uint32_t i = w0 + 64u * st + lane; ev[st] = pa[i]; ev[st] = i < w1 ? ev[st] : 0u; }   // (i < CAP whatever the step)
#pragma unroll
                for (uint32_t st = 0; st < MAXSTEP; st++) dv[st] = kb[ev[st] * (uint32_t)sizeof(K)];
#pragma unroll
                for (uint32_t st = 0; st < MAXSTEP; st++) {
                    inf[st] = 0x100u; bf[st] = 0; ld[st] = 0;
                    if (w0 + 64u * st < w1) {                                // wave-uniform
                        const bool ok = w0 + 64u * st + lane < w1;
                        const uint32_t d = ok ? dv[st] : 0x100u;
                        const unsigned long long peers = BWS_PEERS(ok, d);
                        const uint32_t plo = (uint32_t)peers, phi = (uint32_t)(peers >> 32);
                        // (the first peer: v_ffbl gives -1 for 0, so the smaller of the two halves' answers; my place among them: v_mbcnt)
                        const uint32_t lo1 = (uint32_t)__ffs((int)plo) - 1u, hi1 = (uint32_t)__ffs((int)phi) - 1u + 32u;
                        ld[st] = (lo1 < hi1 ? lo1 : hi1) << 2;
                        const uint32_t rk = (uint32_t)__builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
                        if (ok && rk == 0u) bf[st] = atomicAdd(&myh[d], (uint32_t)__popc(plo) + (uint32_t)__popc(phi));
                        inf[st] = (ev[st] << 21) | (rk << 9) | d;
                        rcx_wave_sync();
                    }
                }
#pragma unroll
                for (uint32_t st = 0; st < MAXSTEP; st++) {
                    if (w0 + 64u * st < w1) inf[st] += (uint32_t)__builtin_amdgcn_ds_bpermute((int)ld[st], (int)bf[st]) << 9;
                }
            }
            sync();
            BWS_LAP(1);
            // ---- totals, "one digit only", exclusive scan, per-wave cursors
            bool one;
            if (NW == 1) {
                const uint32_t t0 = myh[4 * lane], t1 = myh[4 * lane + 1], t2 = myh[4 * lane + 2], t3 = myh[4 * lane + 3];
                one = __ballot(t0 == len || t1 == len || t2 == len || t3 == len) != 0;
                const uint32_t ex = rcx_wave_incl_scan(t0 + t1 + t2 + t3) - (t0 + t1 + t2 + t3);
                rcx_wave_sync();
                if (!one) { myh[4 * lane] = ex; myh[4 * lane + 1] = ex + t0; myh[4 * lane + 2] = ex + t0 + t1; myh[4 * lane + 3] = ex + t0 + t1 + t2; }
                rcx_wave_sync();
            } else {
                uint32_t c[NW], sum = 0;
#pragma unroll
                for (int k = 0; k < NW; k++) { c[k] = hist[256 * k + t]; sum += c[k]; }
                if (sum == len) misc[0] = 1;
                const uint32_t inc = rcx_wave_incl_scan(sum);
                if (lane == 63) misc[1 + w] = inc;
                __syncthreads();
                one = misc[0] != 0;
                uint32_t o = inc - sum;
#pragma unroll
                for (int k = 0; k < NW; k++) o += ((uint32_t)k < w) ? misc[1 + k] : 0u;
                if (!one) {
#pragma unroll
                    for (int k = 0; k < NW; k++) { hist[256 * k + t] = o; o += c[k]; }
                }
                __syncthreads();
            }
            BWS_LAP(2);
            if (one) { BWS_CNT(9, 1); continue; }                            // the whole group shares this digit: nothing moves
            BWS_CNT(8, 1);
            // ---- scatter (stable): where the wave's elements with the digit start + the element's place among them
#pragma unroll
            for (uint32_t st = 0; st < MAXSTEP; st++) {
                if (w0 + 64u * st < w1) {
                    const uint32_t f = inf[st];
                    if (!(f & 0x100u)) pb[myh[f & 0xffu] + ((f >> 9) & 0xfffu)] = (uint16_t)(f >> 21);
                }
            }
            sync();
            BWS_LAP(3);
            { uint16_t* x = pa; pa = pb; pb = x; }
        }
        settle();
        // ---- runs of equal keys: a bitmap of run heads, then every position looks up its run
        const uint32_t nwords = (len + 31u) >> 5;
        for (uint32_t c0 = 64u * w; c0 < len; c0 += T) {
            const uint32_t p = c0 + lane;
            bool head = false;
            if (p < len) head = p == 0 || key[pa[p]] != key[pa[p - 1]];
            const unsigned long long hm = __ballot(head);
            if (lane == 0) { bits[c0 >> 5] = (uint32_t)hm; bits[(c0 >> 5) + 1] = (uint32_t)(hm >> 32); }
        }
        sync();
        BWS_LAP(4);
        // per bitmap word: the last head at or before its end and the first head at or after its start (a wave scan over the <= 64
        // words; a position that walked the bitmap to its run's ends took ~100 K cycles in a group of 2048 equal keys)
        if (w == 0) {
            const uint32_t h = lane < nwords ? bits[lane] : 0u;
            uint32_t lastp = h ? (lane << 5) + 32u - (uint32_t)__clz((int)h) : 0u;           // position + 1; 0 = none
            uint32_t firstp = h ? (lane << 5) + (uint32_t)__ffs((int)h) - 1u : 0xffffffffu;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t lo = (uint32_t)__shfl_up((int)lastp, d), fo = (uint32_t)__shfl_down((int)firstp, d);
                if ((int)lane >= d && lo > lastp) lastp = lo;
                if ((int)lane + d < 64 && fo < firstp) firstp = fo;
            }
            hist[lane] = lastp; hist[64u + lane] = firstp;
        }
        sync();
        BWS_LAP(5);
        for (uint32_t c0 = 64u * w; c0 < len; c0 += T) {
            const uint32_t p = c0 + lane;
            const bool in = p < len;
            uint32_t rs = 0, re = len, g = 0;
            if (in) {
                const uint32_t wi = p >> 5, bb = p & 31u;
                const uint32_t m = bits[wi] & (0xffffffffu >> (31u - bb));                       // heads at or before p (position 0 is one)
                rs = m ? (wi << 5) + 31u - (uint32_t)__clz((int)m) : hist[wi - 1u] - 1u;
                const uint32_t m2 = bb == 31u ? 0u : bits[wi] & (0xffffffffu << (bb + 1u));      // heads after p
                if (m2) re = (wi << 5) + (uint32_t)__ffs((int)m2) - 1u;
                else { const uint32_t nx = wi + 1u < nwords ? hist[64u + wi + 1u] : 0xffffffffu; re = nx < len ? nx : len; }
                g = val[pa[p]];
                const bool single = re - rs == 1u;
                s.saA[sg.start + p] = g | (rs == p ? (BWS_HEAD | BWS_RV | s.par) : 0u) | (single ? BWS_FINAL : 0u);
#if defined(BWS_CUT_RANK) && BWS_CUT_RANK == 2
                s.rank[sg.start + p] = sg.start + rs;
#elif defined(BWS_CUT_RANK) && BWS_CUT_RANK == 3
                s.rank[(sg.start & ~0x3ffffu) + ((g * 2654435761u) & 0x3ffffu)] = sg.start + rs;
#elif !defined(BWS_CUT_RANK)
                if (!(whole && rs == 0u)) s.rank[g] = sg.start + rs;             // the run at the start of a group whose ranks stand keeps its rank
#endif
            }
            Q->push(s, in && rs == p && re - rs >= 2u, sg.start + rs, re - rs, top_shift);
        }
        BWS_LAP(6);
        sync();
        BWS_LAP(7);
        return nx;
    }
};

// The wave-sized groups and the workgroup-sized ones get a kernel each: the first keeps 20 KiB of LDS per workgroup (four
// waves, a group each) so that seven workgroups share a CU -- these passes are chains of dependent LDS round trips, what
// they need is waves in flight (with one kernel for both, 38 KiB per workgroup: four per CU, the vector ALU 21 % busy).
#ifndef BWS_LW_OCC
#define BWS_LW_OCC 6
#endif
template <class K>
__global__ __launch_bounds__(256, BWS_LW_OCC) RCX_SGPR_CAP void k_bws_local_wave(BwsState s, uint32_t top_shift)
{
    __shared__ __align__(16) K s_key[4 * BWS_LWAVE];
    __shared__ uint32_t s_val[4 * BWS_LWAVE];
    __shared__ uint16_t s_pa[4 * BWS_LWAVE], s_pb[4 * BWS_LWAVE];
    __shared__ uint32_t s_hist[4][256];
    __shared__ uint32_t s_q[4][2 * BWS_QCAP];
    const uint32_t nseg = s.cnt[6];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    BwsQueue Q; Q.q = s_q[wave]; Q.n = 0; Q.ts = top_shift;
    BwsLocal<K, 1> L; L.Q = &Q;
    L.key = s_key + BWS_LWAVE * wave; L.val = s_val + BWS_LWAVE * wave; L.pa = s_pa + BWS_LWAVE * wave; L.pb = s_pb + BWS_LWAVE * wave;
    L.hist = s_hist[wave]; L.bits = s_hist[wave] + 128; L.misc = nullptr;
    L.t = lane; L.w = 0; L.lane = lane;
    {
        uint32_t e = blockIdx.x * 4u + wave;
        const uint32_t step = gridDim.x * 4u;
        BwsSeg sg{0, 0, 0};
        if (e < nseg) { sg = s.local[e]; L.prefetch(s, sg); L.pend(s.local, e + step, e + step < nseg); L.settle(); }      // (settled here too: the loop below then never waits where a group starts)
        while (e < nseg) {
            sg = L.run(s, sg, top_shift, s.local, e + 2u * step, e + 2u * step < nseg);
            e += step;
        }
    }
    Q.flush(s);
}
template <class K>
__global__ __launch_bounds__(256) void k_bws_local_wg(BwsState s, uint32_t top_shift)
{
    __shared__ __align__(16) K s_key[BWS_LMAX];
    __shared__ uint32_t s_val[BWS_LMAX];
    __shared__ uint16_t s_pa[BWS_LMAX], s_pb[BWS_LMAX];
    __shared__ uint32_t s_hist[4][256];
    __shared__ uint32_t s_misc[8];
    __shared__ uint32_t s_q[4][2 * BWS_QCAP];
    const uint32_t nsegw = s.cnt[9];
    BwsQueue Q; Q.q = s_q[threadIdx.x >> 6]; Q.n = 0; Q.ts = top_shift;
    BwsLocal<K, 4> L; L.Q = &Q;
    L.key = s_key; L.val = s_val; L.pa = s_pa; L.pb = s_pb; L.hist = &s_hist[0][0]; L.bits = &s_hist[0][0] + 128; L.misc = s_misc;
    L.t = threadIdx.x; L.w = threadIdx.x >> 6; L.lane = threadIdx.x & 63u;
    {
        uint32_t e = blockIdx.x;
        BwsSeg sg{0, 0, 0};
        if (e < nsegw) { sg = s.localw[e]; L.prefetch(s, sg); L.pend(s.localw, e + gridDim.x, e + gridDim.x < nsegw); L.settle(); }
        while (e < nsegw) {
            sg = L.run(s, sg, top_shift, s.localw, e + 2u * gridDim.x, e + 2u * gridDim.x < nsegw);
            e += gridDim.x;
        }
    }
    Q.flush(s);
#ifdef BWS_PROF
    L.report(s, 32u);
#endif
}

// ---- between two rounds: the lists the round filled for the next one become its own (the host swaps the pointers), the flag words and
// the level counters are cleared.  On the device, behind the host's copy of the counters in stream order: done from the host (a
// memset, an upload of the counts, a second synchronisation a round) the GPU stood idle ~40 us more per round.
__global__ void k_bws_round_end(BwsState s)
{
    const uint32_t t = threadIdx.x;
    __shared__ uint32_t s_n[4];
    if (t == 0) { s_n[0] = s.cnt[3]; s_n[1] = s.cnt[4]; s_n[2] = s.cnt[7]; s_n[3] = s.cnt[10]; }
    __syncthreads();
    if (t < 32u) s.cnt[t] = t == BWS_CLEVEL ? s_n[0] : t == 2u ? s_n[1] : t == 6u ? s_n[2] : t == 9u ? s_n[3] : 0u;
    for (uint32_t f = t; f < BWS_NFLAG; f += blockDim.x) s.cnt[64u + f] = 0u;
}

// ---- the few groups of <= 64 the dense passes cannot take (33..64 suffixes across both window grids): one wave per group ----
template <class K>
__global__ __launch_bounds__(256) RCX_SGPR_CAP void k_bws_small(BwsState s, uint32_t top_shift)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nseg = s.cnt[2];
    for (uint32_t si = blockIdx.x * 4u + (threadIdx.x >> 6); si < nseg; si += gridDim.x * 4u) {
        const BwsSeg sg = s.small[si];
        const bool in = lane < sg.len;
        const uint32_t v = in ? s.saA[sg.start + lane] : 0u;
        const K key = in ? bws_keys<K>(s, 0)[sg.start + lane] : (K)0;
        uint32_t c0 = in ? 0u : 1u;                         // the padding lanes sort behind the group
        uint32_t klo = (uint32_t)key, khi = sizeof(K) == 8 ? (uint32_t)((uint64_t)key >> 32) : 0u;
        uint32_t val = in ? (v & BWS_IDX) : (BWS_IDX - lane), was = in ? 1u : 0u;
        if (sizeof(K) == 4) { klo = (in ? klo : 0u) | (c0 << 24); c0 = 0; }
        bws_wave_sort<K>(lane, c0, klo, khi, val, was);
        bool rhead; uint32_t rs, re;
        bws_wave_runs<K>(lane, c0, klo, khi, rhead, rs, re);
        if (was) {
            const bool single = re - rs == 1u;
            s.saA[sg.start + lane] = val | (rhead ? (BWS_HEAD | BWS_RV | s.par) : 0u) | (single ? BWS_FINAL : 0u);
            s.rank[val] = sg.start + rs;
        }
        bws_new_group(s, was && rhead && re - rs >= 2u, sg.start + rs, re - rs, top_shift);
    }
}

// ---- groups of <= 64 inside a window: one lane per suffix over the whole array ----------------------------------------------
// Windows of 64 suffixes starting at `off` (0 or 32: a group of <= 32 suffixes that straddles an aligned window lies inside a
// shifted one).  The lanes of a window sort by (start of my group, key, lane): that orders every group the window contains
// and moves nothing else -- a wave-wide bitonic sort over ds_bpermute (21 compare-exchange steps whatever the group sizes).
// A window's input: its SA words, the word after it (is the next window's first suffix a head?) and its keys.  Requested one window
// ahead -- BEFORE the window in hand stores its results -- and waited for (settle) before those stores go out too: loads and stores
// share one counter, and a wait for loads requested after a window's scattered rank[] stores sat those out, window after window.
template <class K> struct BwsWin {
    uint32_t v, nv; K key;
    __device__ __forceinline__ void load(const BwsState& s, uint32_t j0, uint32_t lane)
    {
        const uint32_t j = j0 + lane;
        v = j < s.n ? s.saA[j] : (BWS_HEAD | BWS_FINAL);
        nv = (j0 + 64u >= s.n) ? BWS_HEAD : s.saA[j0 + 64u + RCX_VGPR(0u)];      // (through a lane's index: a uniform load is waited for at once)
        key = j < s.n ? bws_keys<K>(s, 0)[j] : (K)0;
    }
    __device__ __forceinline__ void settle()
    {
        v = RCX_VGPR(v); nv = RCX_VGPR(nv);
        if (sizeof(K) == 8) key = (K)((uint64_t)RCX_VGPR((uint32_t)key) | ((uint64_t)RCX_VGPR((uint32_t)((uint64_t)key >> 32)) << 32));
        else key = (K)RCX_VGPR((uint32_t)key);
    }
};
template <class K>
__device__ __forceinline__ void bws_dense_window(const BwsState& s, uint32_t j0, uint32_t lane, const BwsWin<K>& W, BwsWin<K>& next)
{
    const uint32_t j = j0 + lane;
    const bool in = j < s.n;
    const uint32_t v = W.v;
    const bool nexthead = (W.nv & BWS_HEAD) != 0;
    const unsigned long long heads = __ballot((v & BWS_HEAD) != 0);
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const unsigned long long hb = heads & le, ha = heads & ~le;
    const bool has_head = hb != 0;
    const uint32_t gs = has_head ? 63u - (uint32_t)__clzll(hb) : 0u;
    const uint32_t ge = ha ? (uint32_t)__ffsll(ha) - 1u : 64u;
    // my group lies inside this window, is not final, and was made BEFORE this round (this round's groups are sorted already)
    const uint32_t hv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(gs << 2), (int)v);
    const bool mine = in && has_head && (ha != 0 || nexthead) && ge - gs >= 2u && (hv & BWS_PAR) != s.par && !(hv & BWS_FINAL);
    // (one way through, whatever the window holds: sort, settle the next window's input, store)
    const bool any = __ballot(mine) != 0;
    const K key = W.key;
    uint32_t val = v & BWS_IDX, was = mine ? 1u : 0u;
    uint32_t sp = 0, sw = 0, rv = 0, gpos = 0, glen = 0;         // saA[sp] = sw (st), rank[val] = rv (rk); a new group [gpos, gpos + glen) (newg)
    bool st = false, rk = false, newg = false;
    if (any) {
        const uint32_t maxlen = rcx_wave_max(mine ? ge - gs : 0u);
        if (maxlen <= 12u) {
            // small groups only (the usual case after the first rounds): a rank sort, 2-3 ds_bpermute per member of the largest group
            const uint32_t klo = (uint32_t)key, khi = sizeof(K) == 8 ? (uint32_t)((uint64_t)key >> 32) : 0u;
            uint32_t less = 0, lt = 0, eq = 0;
            // (no branch in the body -- a lane outside its group adds zeros -- and four steps per trip: their ds_bpermute round trips overlap;
            // one step per trip waited ~120 cycles for each)
#pragma unroll 4
            for (uint32_t t = 0; t < maxlen; t++) {
                const uint32_t srcl = gs + t;
                const int pa = (int)((srcl & 63u) << 2);
                const uint32_t ol = (uint32_t)__builtin_amdgcn_ds_bpermute(pa, (int)klo);
                const uint32_t oh = sizeof(K) == 8 ? (uint32_t)__builtin_amdgcn_ds_bpermute(pa, (int)khi) : 0u;
                const bool on = mine && srcl < ge;
                const bool l = on && (oh < khi || (oh == khi && ol < klo)), e = on && oh == khi && ol == klo;
                lt += l ? 1u : 0u; eq += e ? 1u : 0u;
                less += (l || (e && srcl < lane)) ? 1u : 0u;           // equal keys keep the order they stand in (one ds_bpermute less per step)
            }
            st = mine; sp = j0 + gs + less;
            sw = val | (less == lt ? (BWS_HEAD | BWS_RV | s.par) : 0u) | (eq == 1u ? BWS_FINAL : 0u);
            rk = mine && !((hv & BWS_RV) && lt == 0u); rv = j0 + gs + lt;       // the run at the start of a group whose ranks stand keeps its rank
            newg = mine && less == lt && eq >= 2u; gpos = j0 + gs + lt; glen = eq;
        } else {
            uint32_t c0 = mine ? gs : lane;                          // composite sort key: (group start or my own lane, key)
            uint32_t klo = (uint32_t)key, khi = sizeof(K) == 8 ? (uint32_t)((uint64_t)key >> 32) : 0u;
            if (sizeof(K) == 4) { klo = (mine ? klo : 0u) | (c0 << 24); c0 = 0; }   // 32-bit keys are local ranks (< 2^24): the group rides in the key's top bits (a bystander's stale key must not)
            bool rhead; uint32_t rs, re;
            bws_wave_sort<K>(lane, c0, klo, khi, val, was);
            bws_wave_runs<K>(lane, c0, klo, khi, rhead, rs, re);
            st = was != 0; sp = j;
            sw = val | (rhead ? (BWS_HEAD | BWS_RV | s.par) : 0u) | (re - rs == 1u ? BWS_FINAL : 0u);
            rk = was && !((hv & BWS_RV) && rs == gs); rv = j0 + rs;
            newg = was && rhead && re - rs >= 2u; gpos = j0 + rs; glen = re - rs;
        }
    }
    next.settle();
    if (st) s.saA[sp] = sw;
    if (rk) s.rank[val] = rv;
    if (newg) bws_flag_dense(s, s.rs ^ 1u, gpos, glen);                                 // next round's dense passes
    const unsigned long long newgroups = __ballot(newg);
    if (lane == 0 && newgroups) bws_flag_unresolved(s);
}

// One wave per BWS_DW consecutive windows of the grid `off` (0: aligned, 32: shifted); only windows somebody flagged are looked at.
#ifndef BWS_DW
#define BWS_DW 8u
#endif
template <class K>
__global__ __launch_bounds__(256) RCX_SGPR_CAP void k_bws_dense(BwsState s, uint32_t off)
{
    const uint32_t lane = threadIdx.x & 63u;
    // Workgroups go round-robin to the 8 XCDs, each with its own 4 MiB L2: XCD x takes the x-th EIGHTH of the windows, so that the
    // workgroups resident on it at any time sit in two or three blocks and their scattered rank[] stores (one per suffix moved,
    // anywhere in the block's 1 MiB) meet in that L2 instead of leaving it as partial lines.  (gridDim.x is a multiple of 8.)
    const uint32_t vwg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t w0 = (vwg * 4u + RCX_UNI(threadIdx.x >> 6)) * BWS_DW;               // first window (in the grid's own numbering); uniform, and said so:
    if ((uint64_t)w0 * 64u >= (uint64_t)s.n + 64u) return;                           // the window loop below is scalar code then
    uint8_t* act = s.act0 + (size_t)(s.rs * 2u + (off ? 1u : 0u)) * s.nact;
    const unsigned long long fv = *(const unsigned long long*)(act + w0);
    const unsigned long long f = (unsigned long long)RCX_UNI((uint32_t)fv) | ((unsigned long long)RCX_UNI((uint32_t)(fv >> 32)) << 32);
    if (!f) return;
    // the flagged windows, each one's input requested while the one before it is sorted
    auto start_of = [&](uint32_t k) { const uint32_t win = w0 + k; return off ? win * 64u - 32u : win * 64u; };
    auto next_flagged = [&](uint32_t k) {
        for (; k < BWS_DW; k++) {
            if (!((f >> (8u * k)) & 0xffu)) continue;
            if (off && w0 + k == 0) continue;                    // the shifted window [-32, 32) holds nothing the aligned pass cannot take
            if (start_of(k) < s.n) break;
        }
        return k;
    };
    uint32_t k = next_flagged(0);
    BwsWin<K> W, Wn;
    W.v = 0; W.nv = 0; W.key = 0;
    if (k < BWS_DW) { W.load(s, start_of(k), lane); W.settle(); }
    while (k < BWS_DW) {
        const uint32_t k2 = next_flagged(k + 1u);
        Wn.v = 0; Wn.nv = 0; Wn.key = 0;
        if (k2 < BWS_DW) Wn.load(s, start_of(k2), lane);
        bws_dense_window<K>(s, start_of(k), lane, W, Wn);
        W = Wn; k = k2;
    }
    if (lane == 0) *(unsigned long long*)(act + w0) = 0ull;
}
