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

// Cross-lane ordering inside one wave for traffic through LDS/global: hardware executes a wave's
// memory instructions in order, so this only has to stop the COMPILER from reordering (no ISA emitted).
__device__ __forceinline__ void rcx_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
