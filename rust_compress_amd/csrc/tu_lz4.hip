// tu_lz4.hip -- LZ4 block decode / encode kernels + their launch code (one translation unit).
// The shipped library holds the default decoder and one exact fallback: 0 = two waves per block with the segment-parallel parser
// (k_lz4_decode_v8), 15 = with the serial-walk parser (k_lz4_decode_v5; k_lz4_decode_v4 is the base class both build on).
// Everything else -- the one-wave kernel as a variant of its own (11), the earlier generations (v1-v3), the workgroup-per-block
// and chunk-centric experiments (v6, v7; benchmarks/experiments/), profiling and attribution instantiations -- is compiled only
// with -DRCX_AB_VARIANTS into librcx_ab.so, for benchmarks/ and the A/B history in DESIGN.md.
#include "rcx_tu.h"
#ifndef RCX_V45_PRE
#define RCX_V45_PRE 128                  /* (attribution: the parser-only variant with another head start) */
#endif
#ifdef RCX_AB_VARIANTS
#include "../../benchmarks/experiments/k_lz4_decode_v1_v3.hip"
#endif
#include "k_lz4_decode_v4.hip"
#include "k_lz4_decode_v5.hip"
#include "k_lz4_decode_v8.hip"
#ifdef RCX_AB_VARIANTS
#include "../../benchmarks/experiments/k_lz4_decode_v7.hip"
#include "../../benchmarks/experiments/k_lz4_decode_v6.hip"
#endif
#include "k_lz4_encode.hip"

static const uint32_t LZ4E_CHUNK = 8192;        // LZ4 blocks encoded per launch (512 KiB table each): 32 waves per CU

uint64_t rcx_tu_lz4_encode_scratch(uint32_t nblocks) { return (uint64_t)(nblocks < LZ4E_CHUNK ? nblocks : LZ4E_CHUNK) * LZ4E_TABLE * 4ull; }

// Default (0): two waves per block with the SEGMENT-PARALLEL parser wave (k_lz4_decode_v8: the token walk no longer owns the
// CU's scalar port; 4096 x 64 KiB: text 0.636 against 0.662 ms, words 0.80 / 0.84, rand 0.151 / 0.155, runs 1.19 / 1.17).
// 15: the serial-walk parser wave (k_lz4_decode_v5, the default of rounds 1-2).  A/B library: 11 = one wave per block (k_lz4_decode_v4), 20 = v7.
int rcx_tu_lz4_decode(hipStream_t s, rcx_kargs& k, int v, std::string& err)
{
    const uint32_t n = k.nblocks;
    if ((v == 0 || v == 23) && k.out_mirror) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 0, true>), dim3(n), dim3(128), 0, s, k, 0);   // + every byte to the caller's page-locked buffer
    else if (v == 0 || v == 23) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768>), dim3(n), dim3(128), 0, s, k, 0);
    else if (v == 15) hipLaunchKernelGGL((k_lz4_decode_v5<2048, 1536, 2048>), dim3(n), dim3(128), 0, s, k, 0);
#ifdef RCX_AB_VARIANTS
    else if (v == 11) hipLaunchKernelGGL((k_lz4_decode_v4<1024, 1>), dim3(n), dim3(64), 0, s, k);
    else if (v == 20) hipLaunchKernelGGL((k_lz4_decode_v7<1024, 1008, 2048, 2048>), dim3(n), dim3(128), 0, s, k, 0);
    else if (v == 1) hipLaunchKernelGGL(k_lz4_decode_v1, dim3(n), dim3(64), 0, s, k);
    else if (v == 2) hipLaunchKernelGGL((k_lz4_decode_v3<2048, 2048, 64, 64, 4>), dim3((n + 3) / 4), dim3(256), 0, s, k);
    else if (v == 3) hipLaunchKernelGGL((k_lz4_decode_v3<4096, 2048, 64, 64, 1>), dim3(n), dim3(64), 0, s, k);
    else if (v == 4) hipLaunchKernelGGL((k_lz4_decode_v3<2048, 2048, 32, 32, 1>), dim3(n), dim3(64), 0, s, k);
    else if (v == 5) hipLaunchKernelGGL((k_lz4_decode_v2<4096, 2048, 64, 64, 1>), dim3(n), dim3(64), 0, s, k);
    else if (v == 6) hipLaunchKernelGGL((k_lz4_decode_v3<2048, 2048, 64, 64, 1>), dim3(n), dim3(64), 0, s, k);
    else if (v == 7) hipLaunchKernelGGL((k_lz4_decode_v4<2048, 1>), dim3(n), dim3(64), 0, s, k);
    else if (v == 9) hipLaunchKernelGGL((k_lz4_decode_v4<1024, 1, true>), dim3(n), dim3(64), 0, s, k);   // phase timers -> scratch
    else if (v == 8) hipLaunchKernelGGL((k_lz4_decode_v4<1024, 4>), dim3((n + 3) / 4), dim3(256), 0, s, k);
    else if (v == 12) hipLaunchKernelGGL((k_lz4_decode_v5<1024, 1536, 3072>), dim3(n), dim3(128), 0, s, k, 0);   // longer history, smaller batch cap
    else if (v == 13) hipLaunchKernelGGL((k_lz4_decode_v5<1024, 2048, 2560>), dim3(n), dim3(128), 0, s, k, 0);
    else if (v == 16) hipLaunchKernelGGL((k_lz4_decode_v5<2048, 2048, 1536>), dim3(n), dim3(128), 0, s, k, 0);
    else if (v == 14) hipLaunchKernelGGL((k_lz4_decode_v5<1024, 2560, 2048, true>), dim3(n), dim3(128), 0, s, k, 0);   // ring wait timers -> scratch
    else if (v == 10) hipLaunchKernelGGL((k_lz4_decode_v5<1024>), dim3(n), dim3(128), 0, s, k, 0);   // 1 KiB of staged input, batch cap 2560
    else if (v == 17) {
        // workgroup per block, history in LDS; the blocks it hands back (RCX_ST_BAIL6) are re-run by the exact two-wave kernel
        hipLaunchKernelGGL((k_lz4_decode_v6<8>), dim3(n), dim3(512), 0, s, k);
        hipLaunchKernelGGL((k_lz4_decode_v5<2048, 1536, 2048>), dim3(n), dim3(128), 0, s, k, (int)RCX_ST_BAIL6);
    }
    else if (v == 21) hipLaunchKernelGGL((k_lz4_decode_v7<1024, 1008, 2048, 2048, 2, true>), dim3(n), dim3(128), 0, s, k, 0);     // parser wave alone (wrong output on purpose)
    else if (v == 22) hipLaunchKernelGGL((k_lz4_decode_v7<2048, 1008, 1536, 2048, 2, true>), dim3(n), dim3(128), 0, s, k, 0);
    else if (v == 24) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, true>), dim3(n), dim3(128), 0, s, k, 0);     // parser phase timers -> scratch
    else if (v == 29) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 0>), dim3(n), dim3(128), 0, s, k, 0);    // no split of long matches
    else if (v == 32) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, true, 2>), dim3(n), dim3(128), 0, s, k, 0);   // chains shortened by the parser (more instructions in total: slower)
    else if (v == 33) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, true, 3>), dim3(n), dim3(128), 0, s, k, 0);  // three redirection rounds
    else if (v == 31) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 96, 32>), dim3(n), dim3(128), 0, s, k, 0);    // shorter head start
    else if (v == 25) hipLaunchKernelGGL((k_lz4_decode_v5<2048, 1024, 1024, false, 32>), dim3(n), dim3(128), 0, s, k, 0);   // LDS trade-offs of the executor
    else if (v == 26) hipLaunchKernelGGL((k_lz4_decode_v5<2048, 1024, 1024, false, 16>), dim3(n), dim3(128), 0, s, k, 0);
    else if (v == 27) hipLaunchKernelGGL((k_lz4_decode_v5<2048, 1280, 1024, false, 0>), dim3(n), dim3(128), 0, s, k, 0);
    else if (v == 28) hipLaunchKernelGGL((k_lz4_decode_v5<2048, 1536, 2048, false, 16>), dim3(n), dim3(128), 0, s, k, 0);
    // instruction attribution (results wrong on purpose; bench.py --variant 4x with RCX_BENCH_EXPERIMENT_NOCHECK=1): phases cut out
    else if (v == 41) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 1>), dim3(n), dim3(128), 0, s, k, 0);     // no copy rounds
    else if (v == 42) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 3>), dim3(n), dim3(128), 0, s, k, 0);     // nor chain analysis
    else if (v == 43) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 7>), dim3(n), dim3(128), 0, s, k, 0);     // nor literal / gather stores
    else if (v == 44) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 23>), dim3(n), dim3(128), 0, s, k, 0);    // nor the drain
    else if (v == 45) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, RCX_V45_PRE, 32, false, 2, 8>), dim3(n), dim3(128), 0, s, k, 0);     // executor only empties the ring: the parser wave's share
    else if (v == 46) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 64>), dim3(n), dim3(128), 0, s, k, 0);    // the compiled copy-round loop only (exact)
    else if (v == 47) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 128>), dim3(n), dim3(128), 0, s, k, 0);   // hand-written rounds at priority 1 (exact)
    else if (v == 48) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 0x200>), dim3(n), dim3(128), 0, s, k, 0);   // no gathers of old matches (their latency)
    else if (v == 49) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 0x600>), dim3(n), dim3(128), 0, s, k, 0);   // nor literal loads
    else if (v == 50) hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 0x800>), dim3(n), dim3(128), 0, s, k, 0);   // runs copied by the portable loop (exact): what period doubling in the hand-written loop buys
    else if (v == 19) hipLaunchKernelGGL((k_lz4_decode_v6<8, true>), dim3(n), dim3(512), 0, s, k);      // phase timers -> scratch, no second pass
#endif
    else { err = "lz4 decode: unknown kernel variant (A/B variants need a -DRCX_AB_VARIANTS build)"; return RCX_RC_BAD_ARG; }
    return RCX_RC_OK;
}

// the blocks a gated launch gave up on (RCX_ST_GATE: their input was late), once more -- the input is all there now
void rcx_tu_lz4_decode_mirror_again(hipStream_t s, rcx_kargs& k)
{
    hipLaunchKernelGGL((k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 0, true>), dim3(k.nblocks), dim3(128), 0, s, k, (int)RCX_ST_GATE);
}

int rcx_tu_lz4_encode(hipStream_t s, rcx_kargs& k, int v, std::string& err)
{
    const uint32_t n = k.nblocks;
    if (k.scratch_bytes < rcx_tu_lz4_encode_scratch(n)) { err = "lz4 encode: scratch too small"; return RCX_RC_BAD_ARG; }
    for (uint32_t b0 = 0; b0 < n; b0 += LZ4E_CHUNK) {
        const uint32_t cnt = n - b0 < LZ4E_CHUNK ? n - b0 : LZ4E_CHUNK;
        if (hipMemsetAsync(k.scratch, 0, (size_t)cnt * LZ4E_TABLE * 4ull, s) != hipSuccess) { err = "lz4 encode: hipMemsetAsync failed"; return RCX_RC_HIP_ERROR; }
        if (v == 1) hipLaunchKernelGGL(k_lz4_encode, dim3(cnt), dim3(64), 0, s, k, b0);          // serial probe chain
        else if (v == 2) hipLaunchKernelGGL(k_lz4_encode_w<64>, dim3(cnt), dim3(64), 0, s, k, b0);
        else hipLaunchKernelGGL(k_lz4_encode_w<8>, dim3(cnt), dim3(64), 0, s, k, b0);
    }
    return RCX_RC_OK;
}
