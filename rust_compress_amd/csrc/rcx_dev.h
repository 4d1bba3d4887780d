// rcx_dev.h -- device-side common definitions for the gfx950 kernels.
#pragma once
#include <stdint.h>
#include "../../include/rcx.h"

// Kernel argument block: the device-resident batch descriptor (mirrors rcx_dev_batch).
struct rcx_kargs {
    const uint8_t* in_base;
    const uint64_t* in_off;
    const uint64_t* in_len;
    uint8_t* out_base;
    const uint64_t* out_off;
    const uint64_t* out_cap;
    uint64_t* out_len;
    uint64_t* in_used;   // may be null
    int32_t* status;
    uint32_t* aux;       // codec extra, may be null
    const uint64_t* n_out;   // dc decode: decoded length, may be null
    void* scratch;
    uint64_t scratch_bytes;
    uint32_t nblocks;
};

#define RCX_WAVE 64

// 16-byte vector; the _u flavour may sit at any byte address (global memory only: unaligned DS is slow)
typedef unsigned int rcx_u32x4 __attribute__((vector_size(16)));
typedef rcx_u32x4 __attribute__((aligned(1))) rcx_u32x4_u;

__device__ __forceinline__ unsigned rcx_lane() { return threadIdx.x & 63u; }

// DPP move: lanes whose source is outside the row / masked off get 0.  ctrl: row_shr:n = 0x110+n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143 (gfx9 encodings).
#define RCX_DPP0(v, ctrl, row_mask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (row_mask), 0xf, false))

// wave64 inclusive prefix sum on the VALU (6 DPP adds, no LDS crossbar round trips)
__device__ __forceinline__ uint32_t rcx_wave_incl_scan(uint32_t v)
{
    v += RCX_DPP0(v, 0x111, 0xf);      // row_shr:1
    v += RCX_DPP0(v, 0x112, 0xf);      // row_shr:2
    v += RCX_DPP0(v, 0x114, 0xf);      // row_shr:4
    v += RCX_DPP0(v, 0x118, 0xf);      // row_shr:8   -> inclusive scan inside each row of 16
    v += RCX_DPP0(v, 0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    v += RCX_DPP0(v, 0x143, 0xc);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t rcx_wave_max(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t t = __shfl_xor(v, d);
        v = t > v ? t : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t rcx_wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// LZ4 token walk over one 64-position register window.  dv: lane p holds the hop distance of the candidate
// token at window position p, or 128 if that token needs the general path.  Starting at `rel`, follows the
// chain while it stays inside the window and sets mark = 1 in the lane of every token start visited.
// On return rel >= 128 means: stopped at a general-path token at window position rel-128 (its lane is
// marked too); otherwise 64 <= rel < 128 is where the chain left the window.
// Hand-scheduled because the CU's single scalar unit is the bottleneck of the parse (rocprof + phase timers:
// hipcc's loop cost 11 SALU per hop, ~168 cycles with 16 waves per CU): 3 SALU + 3 VALU per hop, no taken
// branch for 4 hops.  SALU reads of a VALU-written SGPR interlock in hardware; the lane select is SALU-written.
// The wave simulator supplies a portable version through this hook.
#ifndef RCX_HOP_WALK
__device__ __forceinline__ void rcx_hop_walk(uint32_t dv, uint32_t lanev, uint32_t& rel, uint32_t& mark)
{
    uint32_t d;
#define RCX_HOP1                                              \
        "v_readlane_b32 %[d], %[dv], %[rel]\n\t"              \
        "v_cmp_eq_u32_e32 vcc, %[rel], %[lanev]\n\t"          \
        "v_cndmask_b32_e64 %[mark], %[mark], 1, vcc\n\t"      \
        "s_add_u32 %[rel], %[rel], %[d]\n\t"                  \
        "s_cmp_gt_u32 %[rel], 63\n\t"
    asm volatile(
        "L_hop_%=:\n\t"
        RCX_HOP1 "s_cbranch_scc1 L_done_%=\n\t"
        RCX_HOP1 "s_cbranch_scc1 L_done_%=\n\t"
        RCX_HOP1 "s_cbranch_scc1 L_done_%=\n\t"
        RCX_HOP1 "s_cbranch_scc0 L_hop_%=\n\t"
        "L_done_%=:\n\t"
        : [d] "=&s"(d), [rel] "+s"(rel), [mark] "+v"(mark)
        : [dv] "v"(dv), [lanev] "v"(lanev)
        : "scc", "vcc");
#undef RCX_HOP1
}
#define RCX_HOP_WALK rcx_hop_walk
#endif

// Cross-lane ordering inside one wave for traffic through LDS/global: hardware executes a wave's
// memory instructions in order, so this only has to stop the COMPILER from reordering (no ISA emitted).
__device__ __forceinline__ void rcx_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
