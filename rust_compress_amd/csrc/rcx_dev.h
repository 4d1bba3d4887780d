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
    uint8_t* out_mirror; // LZ4 decode from host memory: the caller's page-locked output buffer as the device sees it (k_lz4_decode_v4.hip, MIRROR), else null
    // ... and its input arrives WHILE the launch runs: blocks from gate_bnd[i - 1] on (range i = 1..15) start when gate[i] == gate_seq
    // (rcx_api.hip), or give up after gate_ticks (100 MHz) with RCX_ST_GATE
    uint32_t* gate;
    const uint32_t* gate_host;   // the same words in page-locked host memory, set by the calling thread when a range's copy has completed: the
                                 // range's FIRST block watches its word there (across PCIe) and passes it on to gate[i] for the others
    uint32_t gate_seq, gate_ticks;
    uint32_t gate_all;           // range 0 waits too (at gate[0]; its first block is block 0): the launch is enqueued before any input has arrived
    uint32_t gate_bnd[15];
};
#define RCX_ST_GATE 0x7ff00003               /* internal, never leaves the library: the block's input did not arrive in time, run it again */

#define RCX_WAVE 64

// A wave's SGPRs come out of 800 per SIMD in granules of 16, and the trap handler takes 16 more per wave: eight waves per SIMD
// need <= 80.  hipcc counts without the trap handler's share and reports "Occupancy: 8" up to 96 -- a kernel with 81..96 SGPRs
// runs seven waves per SIMD (found the hard way: six SGPRs named in an asm block took k_lz4_decode_v8 from 78 to 96 and from
// 0.63 to 0.89 ms, the sixteenth workgroup of a CU waiting for a second residency round).  Kernels whose registers would allow
// eight waves carry this cap; what does not fit is spilled to VGPR lanes, of which those kernels have plenty.
#ifndef RCX_SGPR_CAP
#define RCX_SGPR_CAP __attribute__((amdgpu_num_sgpr(80)))
#endif

// 16-byte vector; the _u flavour may sit at any byte address (global memory only: unaligned DS is slow)
typedef unsigned int rcx_u32x4 __attribute__((vector_size(16)));
typedef rcx_u32x4 __attribute__((aligned(1))) rcx_u32x4_u;
typedef uint64_t __attribute__((aligned(1))) rcx_u64_u;
typedef uint32_t __attribute__((aligned(1))) rcx_u32_u;

__device__ __forceinline__ unsigned rcx_lane() { return threadIdx.x & 63u; }

// DPP move: lanes whose source is outside the row / masked off get 0.  ctrl: row_shr:n = 0x110+n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143 (gfx9 encodings).
#define RCX_DPP0(v, ctrl, row_mask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (row_mask), 0xf, false))

// wave64 inclusive prefix sum on the VALU: six DPP adds, no LDS crossbar round trips.  Written as ISA: from the builtin form
// (v += update_dpp(0, v, ...)) hipcc makes v_mov 0 + v_mov_dpp + v_add per step, 18 VALU a scan -- three scans a batch were
// 5 % of the LZ4 decoder's vector instructions.  `v_add_u32_dpp v, v, v`: a lane whose DPP source is outside its row (or whose
// row is masked off) is disabled and keeps v, which is what a scan wants; gfx9 needs two wait states between a VALU write of a
// VGPR and a DPP read of it, and nothing inserts them inside an asm block.  The wave simulator keeps the builtin form
// (RCX_WAVE_INCL_SCAN hook).
#ifndef RCX_WAVE_INCL_SCAN
__device__ __forceinline__ uint32_t rcx_wave_incl_scan(uint32_t v)
{
    asm("s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        : "+v"(v));
    return v;
}
#define RCX_WAVE_INCL_SCAN rcx_wave_incl_scan
#else
__device__ __forceinline__ uint32_t rcx_wave_incl_scan(uint32_t v) { return RCX_WAVE_INCL_SCAN(v); }
#endif
// wave64 maximum, uniform (an SGPR): the same six DPP steps as the scan and one v_readlane -- the butterfly over ds_bpermute it replaces
// was six dependent LDS-crossbar round trips (~700 cycles on the critical path of every k_bws_dense window)
__device__ __forceinline__ uint32_t rcx_wave_max(uint32_t v)
{
    uint32_t t;
    t = RCX_DPP0(v, 0x111, 0xf); v = t > v ? t : v;
    t = RCX_DPP0(v, 0x112, 0xf); v = t > v ? t : v;
    t = RCX_DPP0(v, 0x114, 0xf); v = t > v ? t : v;
    t = RCX_DPP0(v, 0x118, 0xf); v = t > v ? t : v;
    t = RCX_DPP0(v, 0x142, 0xa); v = t > v ? t : v;
    t = RCX_DPP0(v, 0x143, 0xc); v = t > v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t rcx_wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// LZ4 token walk over one 64-position register window.  dv: lane p holds the hop distance of the candidate
// token at window position p, or 128 if that token needs the general path.  Starting at `rel`, follows the
// chain while it stays inside the window and sets bit p of `vis` for every token start p visited.
// On return rel >= 128 means: stopped at a general-path token at window position rel-128 (its bit is set
// too); otherwise 64 <= rel < 128 is where the chain left the window.
// Hand-scheduled (hipcc's loop cost 11 SALU per hop): a hop is v_readlane (distance) + s_bitset1_b64 (visited
// mark, kept in an SGPR pair) + ONE s_add whose carry is the window-exit test (position biased by -64); no taken
// branch for 4 hops.  A v_writelane mark (plain VALU rate, one SALU less) measured the same: per CU the scalar
// unit retires ~1 instruction per cycle and VALU is the busier pipe (benchmarks/micro/issue_rate.hip, rocprof).
// SALU reads of a VALU-written SGPR interlock in hardware; the lane select is SALU-written.
// The wave simulator supplies a portable version through this hook.
#ifndef RCX_HOP_WALK
__device__ __forceinline__ void rcx_hop_walk(uint32_t dv, uint32_t& rel, uint64_t& vis)
{
    uint32_t d;
    uint64_t mk = 0;
    rel -= 64u;                      // biased so that leaving the window is the carry of the s_add (no s_cmp)
#define RCX_HOP1                                              \
        "v_readlane_b32 %[d], %[dv], %[rel]\n\t"              \
        "s_bitset1_b64 %[mk], %[rel]\n\t"                     \
        "s_add_u32 %[rel], %[rel], %[d]\n\t"
    asm volatile(
        "L_hop_%=:\n\t"
        RCX_HOP1 "s_cbranch_scc1 L_done_%=\n\t"
        RCX_HOP1 "s_cbranch_scc1 L_done_%=\n\t"
        RCX_HOP1 "s_cbranch_scc1 L_done_%=\n\t"
        RCX_HOP1 "s_cbranch_scc0 L_hop_%=\n\t"
        "L_done_%=:\n\t"
        : [d] "=&s"(d), [rel] "+s"(rel), [mk] "+s"(mk)
        : [dv] "v"(dv)
        : "scc");
    rel += 64u;
    vis = mk;
#undef RCX_HOP1
}
#define RCX_HOP_WALK rcx_hop_walk
#endif

// Store the low `nv` (0..16, per lane) bytes of the 128-bit value {v0,v1,v2,v3} at LDS pointer p.
// EXEC is narrowed byte by byte (v_cmpx) and the byte index rides in the DS offset field, so a byte costs one
// VALU compare + one ds_write_b8 (plus a shift for odd bytes) and the sequence ends at the first multiple of
// four that no lane reaches.  Byte stores because gfx950 serialises unaligned wider LDS accesses (measured:
// ~24 cycles per unaligned ds_write_b16/b32/b64 against 2-3 per ds_write_b8, benchmarks/micro/lds_cost.hip).
#ifndef RCX_LDS_STORE16
__device__ __forceinline__ void rcx_lds_store16(uint8_t* p, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t nv)
{
    const uint32_t a = (uint32_t)(uintptr_t)p;       // low half of a generic LDS pointer = the LDS byte address
    uint32_t t; uint64_t sv;
#define RCX_ST4(V, O0, O1, O2, O3)                                        \
        "v_cmpx_lt_u32_e32 vcc, " #O0 ", %[nv]\n\t"                       \
        "s_cbranch_execz L_end_%=\n\t"                                    \
        "ds_write_b8 %[a], %[" V "] offset:" #O0 "\n\t"                    \
        "v_cmpx_lt_u32_e32 vcc, " #O1 ", %[nv]\n\t"                       \
        "v_lshrrev_b32_e32 %[t], 8, %[" V "]\n\t"                         \
        "ds_write_b8 %[a], %[t] offset:" #O1 "\n\t"                       \
        "v_cmpx_lt_u32_e32 vcc, " #O2 ", %[nv]\n\t"                       \
        "ds_write_b8_d16_hi %[a], %[" V "] offset:" #O2 "\n\t"             \
        "v_cmpx_lt_u32_e32 vcc, " #O3 ", %[nv]\n\t"                       \
        "ds_write_b8_d16_hi %[a], %[t] offset:" #O3 "\n\t"     /* t = V >> 8 from byte 1 (its lanes include these): bits 23:16 = byte 3 */
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        RCX_ST4("v0", 0, 1, 2, 3) RCX_ST4("v1", 4, 5, 6, 7) RCX_ST4("v2", 8, 9, 10, 11) RCX_ST4("v3", 12, 13, 14, 15)
        "L_end_%=:\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [t] "=&v"(t), [sv] "=&s"(sv)
        : [a] "v"(a), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [nv] "v"(nv)
        : "vcc", "memory");
#undef RCX_ST4
}
#define RCX_LDS_STORE16 rcx_lds_store16
#endif

// s_waitcnt vmcnt(0): every vector memory operation of the wave so far has completed (the simulator defines it away)
#ifndef RCX_WAIT_VMEM
#define RCX_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
#endif

#ifndef RCX_UNI
#define RCX_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x)))   // wave-uniform value -> SGPR
#endif

// Launder a wave-uniform value into a VGPR so that what is computed from it runs on the (idle) vector ALU instead of
// the CU's single scalar unit; decisions come back through __ballot.  The wave simulator defines it as the identity.
#ifndef RCX_VGPR
__device__ __forceinline__ uint32_t rcx_vgpr(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
#define RCX_VGPR(x) rcx_vgpr(x)
// "all of these have been requested before any is used": keeps hipcc from waiting for one load before it issues the next
// a 16-byte register set that is deliberately not initialised: "defined" for the compiler by an empty asm, no instruction emitted
#define RCX_NOINIT4(v) asm volatile("" : "=v"(v))
#define RCX_NOINIT_S(x) asm volatile("" : "=s"(x))      /* a wave-uniform value, likewise */
#define RCX_SETTLE4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#endif

// Cross-lane ordering inside one wave for traffic through LDS/global: hardware executes a wave's
// memory instructions in order, so this only has to stop the COMPILER from reordering (no ISA emitted).
__device__ __forceinline__ void rcx_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
