// wavesim.h -- a tiny wave64 SIMT simulator.  TEST INFRASTRUCTURE ONLY.
//
// There is no GPU in the build container, so kernel LOGIC is debugged here: the .hip kernel
// sources under rust_compress_amd/csrc/ are compiled unmodified with g++ behind this header
// (force-included), every lane of a workgroup runs as a fiber, and each wave-level collective
// (__ballot, __shfl, ds_bpermute, readfirstlane, wave_barrier, __syncthreads, ...) is a
// rendezvous of the wave's (or block's) live lanes.  It is NOT a product path: nothing in
// rust_compress_amd/ includes, links or loads it, kernels contain no #ifdef for it, and the
// `-m gpu` tests / smoke() / bench.py never touch it.
//
// Fidelity rules the kernels follow so that "passes here" means something on hardware:
//   * cross-lane traffic through LDS/global inside a wave is always separated by
//     __builtin_amdgcn_wave_barrier() (free on hardware, a rendezvous here): load phase,
//     barrier, store phase, barrier;
//   * collectives are only called under wave-uniform control flow;
//   * blocks never communicate inside a launch (blocks run one after another here).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>
#include <type_traits>

namespace ws {

struct dim3_ { unsigned x = 1, y = 1, z = 1; dim3_() {} dim3_(unsigned a, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

struct Wave {
    int live = 0;                       // lanes not yet exited
    int arrived[2] = {0, 0};
    long epoch[2] = {-1, -1};
    uint64_t slot[2][64];
    bool valid[2][64];
};
struct Lane {
    void* sp = nullptr;
    bool done = false;
    unsigned tid = 0;
    Wave* wave = nullptr;
    long phase = 0;                     // wave-collective phase counter
    long bphase = 0;                    // block-barrier phase counter
};
struct Block {
    std::vector<Lane> lanes;
    std::vector<Wave> waves;
    int live = 0;
    int barrier_arrived[2] = {0, 0};
    long barrier_epoch[2] = {-1, -1};
    unsigned long progress = 0;
};

extern Lane* cur;
extern Block* blk;
extern void* sched_sp;
extern dim3_ g_blockIdx, g_blockDim, g_gridDim;

extern "C" void ws_switch(void** from_sp, void* to_sp);
void launch(dim3_ grid, dim3_ block, const std::function<void()>& body);

inline void yield_() { ws_switch(&cur->sp, sched_sp); }

// Rendezvous of all live lanes of the caller's wave.  Every lane contributes one 64-bit value and gets
// back everybody's contribution (valid[] false for exited lanes).
inline void wave_exchange(uint64_t v, uint64_t out[64], bool val[64])
{
    Lane& me = *cur; Wave& w = *me.wave;
    const int b = (int)(me.phase & 1);
    if (w.epoch[b] != me.phase) {       // first arrival of this phase recycles the buffer
        w.epoch[b] = me.phase; w.arrived[b] = 0;
        for (int i = 0; i < 64; i++) w.valid[b][i] = false;
    }
    w.slot[b][me.tid & 63] = v;
    w.valid[b][me.tid & 63] = true;
    w.arrived[b]++;
    blk->progress++;
    while (w.arrived[b] < w.live) yield_();
    for (int i = 0; i < 64; i++) { out[i] = w.slot[b][i]; val[i] = w.valid[b][i]; }
    me.phase++;
}
inline void wave_barrier() { uint64_t o[64]; bool v[64]; wave_exchange(0, o, v); }
inline void block_barrier()
{
    Lane& me = *cur; Block& B = *blk;
    const int b = (int)(me.bphase & 1);
    if (B.barrier_epoch[b] != me.bphase) { B.barrier_epoch[b] = me.bphase; B.barrier_arrived[b] = 0; }
    B.barrier_arrived[b]++;
    B.progress++;
    while (B.barrier_arrived[b] < B.live) yield_();
    me.bphase++;
}

template <class T> inline uint64_t to_u64(T x) { uint64_t r = 0; static_assert(sizeof(T) <= 8, ""); std::memcpy(&r, &x, sizeof(T)); return r; }
template <class T> inline T from_u64(uint64_t r) { T x; std::memcpy(&x, &r, sizeof(T)); return x; }

inline unsigned long long ballot(int pred)
{
    uint64_t o[64]; bool v[64];
    wave_exchange(pred ? 1 : 0, o, v);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) if (v[i] && o[i]) m |= 1ull << i;
    return m;
}
template <class T> inline T shfl(T x, int src)
{
    uint64_t o[64]; bool v[64];
    wave_exchange(to_u64(x), o, v);
    src &= 63;
    return v[src] ? from_u64<T>(o[src]) : T(0);       // reading an inactive lane: undefined on HW, 0 here
}
template <class T> inline T shfl_up(T x, unsigned d)
{
    uint64_t o[64]; bool v[64];
    wave_exchange(to_u64(x), o, v);
    int l = (int)(cur->tid & 63) - (int)d;
    return l >= 0 && v[l] ? from_u64<T>(o[l]) : x;
}
template <class T> inline T shfl_down(T x, unsigned d)
{
    uint64_t o[64]; bool v[64];
    wave_exchange(to_u64(x), o, v);
    int l = (int)(cur->tid & 63) + (int)d;
    return l < 64 && v[l] ? from_u64<T>(o[l]) : x;
}
template <class T> inline T shfl_xor(T x, int m)
{
    uint64_t o[64]; bool v[64];
    wave_exchange(to_u64(x), o, v);
    int l = (int)(cur->tid & 63) ^ m;
    return l < 64 && v[l] ? from_u64<T>(o[l]) : x;
}
template <class T> inline T readfirstlane(T x)
{
    uint64_t o[64]; bool v[64];
    wave_exchange(to_u64(x), o, v);
    for (int i = 0; i < 64; i++) if (v[i]) return from_u64<T>(o[i]);
    return x;
}
// __builtin_amdgcn_update_dpp for the controls the kernels use (row_shr/row_shl/row_bcast15/31/wave_shr1/wave_shl1)
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    uint64_t o[64]; bool v[64];
    wave_exchange(to_u64(src), o, v);
    const int l = (int)(cur->tid & 63), row = l >> 4, pos = l & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (pos >> 2)) & 1)) return old;
    int s = -1;
    if (ctrl >= 0 && ctrl <= 0xff) s = (l & ~3) + ((ctrl >> (2 * (l & 3))) & 3);                       // quad_perm
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { int n = ctrl - 0x110; if (pos >= n) s = l - n; }
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { int n = ctrl - 0x100; if (pos + n < 16) s = l + n; }
    else if (ctrl == 0x142) { if (row >= 1) s = row * 16 - 1; }
    else if (ctrl == 0x143) { if (row >= 2) s = 31; }
    else if (ctrl == 0x138) { if (l >= 1) s = l - 1; }
    else if (ctrl == 0x130) { if (l < 63) s = l + 1; }
    else { fprintf(stderr, "wavesim: unsupported dpp ctrl 0x%x\n", ctrl); abort(); }
    if (s < 0 || !v[s]) return bound_ctrl ? 0 : old;
    return from_u64<int>(o[s]);
}
inline int bpermute(int byte_addr, int x) { return shfl<int>(x, (byte_addr >> 2) & 63); }

struct tid_proxy { unsigned x, y, z; };
inline tid_proxy get_tid() { return tid_proxy{cur->tid, 0, 0}; }

template <class T> inline T atomic_add(T* p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <class T> inline T atomic_max(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomic_min(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomic_or(T* p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <class T> inline T atomic_exch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomic_cas(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

}  // namespace ws

// ---------------- HIP surface used by the kernels ----------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __align__(n) alignas(n)
#ifndef __restrict__
#define __restrict__ __restrict
#endif

#define threadIdx (ws::get_tid())
#define blockIdx (ws::g_blockIdx)
#define blockDim (ws::g_blockDim)
#define gridDim (ws::g_gridDim)
typedef ws::dim3_ dim3;
static const int warpSize = 64;

struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

#define __syncthreads() ws::block_barrier()
#define __ballot(p) ws::ballot((p))
#define __any(p) (ws::ballot((p)) != 0)
#define __all(p) (ws::ballot(!(p)) == 0)
#define __shfl(x, l, ...) ws::shfl((x), (int)(l))
#define __shfl_up(x, d, ...) ws::shfl_up((x), (unsigned)(d))
#define __shfl_down(x, d, ...) ws::shfl_down((x), (unsigned)(d))
#define __shfl_xor(x, m, ...) ws::shfl_xor((x), (int)(m))
static inline unsigned __umul24(unsigned a, unsigned b) { return (unsigned)((unsigned long long)(a & 0xffffffu) * (b & 0xffffffu)); }
#define __popcll(x) __builtin_popcountll((unsigned long long)(x))
#define __popc(x) __builtin_popcount((unsigned)(x))
static inline int __ffsll(unsigned long long x) { return x ? __builtin_ctzll(x) + 1 : 0; }
static inline int __ffs(unsigned x) { return x ? __builtin_ctz(x) + 1 : 0; }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; i++) if (x & (1u << i)) r |= 1u << (31 - i); return r; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned __byte_perm_(unsigned a, unsigned b, unsigned s) { (void)a; (void)b; (void)s; return 0; }

#define __builtin_amdgcn_readfirstlane(x) ws::readfirstlane((x))
#define __builtin_amdgcn_readlane(x, l) ws::shfl((x), (int)(l))
#define __builtin_amdgcn_writelane(v, l, o) (((int)(ws::cur->tid & 63) == (int)(l)) ? (int)(v) : (int)(o))
#define __builtin_amdgcn_update_dpp(o, s, c, rm, bm, bc) ws::update_dpp((o), (s), (c), (rm), (bm), (bc))
#define __builtin_amdgcn_ds_bpermute(a, x) ws::bpermute((a), (x))
#define __builtin_amdgcn_wave_barrier() ws::wave_barrier()
#define __builtin_amdgcn_s_barrier() ws::block_barrier()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ws::yield_()        // spin-wait loops let the other waves of the block run
#define __builtin_amdgcn_s_setprio(n) ((void)0)
// (the gate of the host-memory LZ4 decoder: a clock that advances with every look at it, plain loads and stores for the system-scope atomics)
static inline unsigned long long ws_memrealtime_() { static unsigned long long t = 0; return t += 1000; }
#define __builtin_amdgcn_s_memrealtime() ws_memrealtime_()
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(volatile const uint32_t*)(p))
#define __hip_atomic_store(p, v, order, scope) (*(volatile uint32_t*)(p) = (v))
#define __builtin_amdgcn_s_getreg(n) (0u)              // (HW_ID: every wave in slot 0 here; only issue priorities depend on it)
#define RCX_LDS_AS
#define RCX_GLOBAL_AS
#define __builtin_amdgcn_sched_barrier(n) ((void)0)
#define __builtin_amdgcn_mbcnt_lo(m, b) (__builtin_popcount((uint32_t)(m) & ((ws::cur->tid & 63) < 32 ? (1u << (ws::cur->tid & 31)) - 1u : 0xffffffffu)) + (b))
#define __builtin_amdgcn_mbcnt_hi(m, b) (__builtin_popcount((uint32_t)(m) & ((ws::cur->tid & 63) < 32 ? 0u : (1u << (ws::cur->tid & 31)) - 1u)) + (b))
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
namespace ws { extern unsigned long long g_stat[32]; }
#define RCX_V7_STAT(slot, v) (ws::g_stat[slot] += (unsigned long long)(v))
#define RCX_V8_STAT(slot, v) (ws::g_stat[slot] += (unsigned long long)(v))
// portable version of rcx_dev.h's hand-scheduled LZ4 token walk (the product uses inline gfx950 asm)
static inline void ws_hop_walk(uint32_t dv, uint32_t& rel, uint64_t& vis)
{
    for (;;) {
        const uint32_t d = (uint32_t)ws::shfl((int)dv, (int)rel);
        vis |= 1ull << rel;
        rel += d;
        if (rel > 63) break;
    }
}
// portable version of rcx_dev.h's exec-narrowing LDS byte store
static inline void ws_lds_store16(uint8_t* p, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t nv)
{
    const uint32_t v[4] = {v0, v1, v2, v3};
    for (uint32_t i = 0; i < nv && i < 16; i++) p[i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
}

// portable version of k_bwt_sort.hip's hand-written digit-peers routine: the lanes among `ok` that hold the same digit as mine
static inline unsigned long long ws_bws_peers(bool ok, uint32_t d)
{
    unsigned long long peers = ws::ballot(ok);
    for (int bit = 0; bit < 8; bit++) {
        const unsigned long long m = ws::ballot(((d >> bit) & 1u) != 0);
        peers &= ((d >> bit) & 1u) ? m : ~m;
    }
    return peers;
}

// portable version of k_inflate3.hip's hand-written window walk (same contract: see rcx_inf_walk there)
static inline uint32_t ws_inf_walk(uint32_t pk1, uint32_t pk2, uint32_t& pos, uint32_t& cnt, uint32_t& ns, uint32_t& runL, uint32_t& otot,
                                   uint32_t& runsrc, uint32_t litn, uint32_t room, uint32_t& litv, uint32_t& dw0, uint32_t& dw1)
{
    const uint32_t lane = ws::cur->tid & 63;
    for (;;) {
        const uint32_t e = (uint32_t)ws::shfl((int)pk1, (int)pos);
        if (!(e & 0x100u)) {
            const uint32_t kind = (e >> 9) & 7u;
            if (kind == 0) return 0;
            if (kind != 1) return 3;
            pos += e & 0x7fu;
            return 2;
        }
        if (!(e & 0x80u)) {
            if (lane == cnt) litv = (e >> 16) & 0xffu;
            cnt++; runL++; otot++; pos += e & 0x7fu;
            if (cnt >= room) return 5;
            if (runL == 32) {
                if (lane == ns) { dw0 = runsrc; dw1 = 32; }
                ns++; runL = 0; runsrc = litn + cnt;
                if (ns >= 64) return 5;
            }
        } else {
            const uint32_t d = (uint32_t)ws::shfl((int)pk2, (int)pos);
            if ((d >> 16) > otot) return 4;
            pos += e & 0x7fu;
            otot += (d >> 8) & 0xffu;
            if (lane == ns) { dw0 = runsrc; dw1 = d | runL; }
            ns++; runL = 0; runsrc = litn + cnt;
            if (ns >= 64) return 5;
        }
    }
}
// the kernels' scan is an asm block of v_add_u32_dpp; here the same six steps through the simulator's update_dpp
static inline uint32_t ws_wave_incl_scan(uint32_t v)
{
    v += (uint32_t)ws::update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)ws::update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)ws::update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)ws::update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)ws::update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)ws::update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
#define RCX_WAVE_INCL_SCAN ws_wave_incl_scan
#define RCX_RCPF(x) (1.0f / (x))       // v_rcp_f32: rcx_div_u13 is exact whatever the reciprocal's last bits are
#define RCX_NO_INF_WALK_ASM 1          // Inf3::tile4's hand-written hop loop: the simulator walks with the portable hop4 alone
#define RCX_NO_DC_STEPS_ASM 1          // k_dc_decode's hand-written step loop: the simulator runs the portable step alone
#define RCX_NO_WALK_ASM 1              // Lz4V8::next_tok_c's hand-written step: the simulator runs the portable form
#define RCX_NO_MSKOR_ASM 1             // Lz4V8::store_frame's ds_mskor_b32 stores: byte stores here
#define RCX_NO_ROUNDS_ASM 1            // emit5's hand-written copy-round loop: the simulator runs the portable loop alone
#define RCX_LDS_STORE16 ws_lds_store16
static inline uint32_t ws_sad_u8(uint32_t a, uint32_t c) { return c + (a & 255u) + ((a >> 8) & 255u) + ((a >> 16) & 255u) + (a >> 24); }          // v_sad_u8 against 0
static inline uint32_t ws_udot4(uint32_t a, uint32_t w, uint32_t c) { for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 255u) * ((w >> (8 * i)) & 255u); return c; }   // v_dot4_u32_u8
#define RCX_SAD_U8(a, c) ws_sad_u8((a), (c))
#define RCX_UDOT4(a, w, c) ws_udot4((a), (w), (c))
#define RCX_INF_WALK ws_inf_walk
#define BWS_PEERS ws_bws_peers
#define RCX_WAIT_VMEM() ((void)0)
#define RCX_VGPR(x) ((uint32_t)(x))
#define RCX_SETTLE4(a, b, c, d) do { } while (0)
#define RCX_NOINIT4(v) do { (v) = rcx_u32x4{0, 0, 0, 0}; } while (0)
#define RCX_NOINIT_S(x) do { (x) = 0; } while (0)
#define RCX_ALIGNBYTE(hi, lo, sh) ((uint32_t)(((((uint64_t)(hi)) << 32) | (uint32_t)(lo)) >> (8 * ((sh) & 3u))))
#define RCX_INV_BALLOT(m) ((((m) >> (threadIdx.x & 63u)) & 1ull) != 0)
#define RCX_HOP_WALK ws_hop_walk
#define __builtin_readcyclecounter() 0ull
#define __threadfence() ((void)0)
#define __threadfence_block() ((void)0)

#define atomicAdd(p, v) ws::atomic_add((p), (decltype(*(p) + 0))(v))
#define atomicMax(p, v) ws::atomic_max((p), (decltype(*(p) + 0))(v))
#define atomicMin(p, v) ws::atomic_min((p), (decltype(*(p) + 0))(v))
#define atomicOr(p, v) ws::atomic_or((p), (decltype(*(p) + 0))(v))
#define atomicExch(p, v) ws::atomic_exch((p), (decltype(*(p) + 0))(v))
#define atomicCAS(p, c, v) ws::atomic_cas((p), (decltype(*(p) + 0))(c), (decltype(*(p) + 0))(v))
