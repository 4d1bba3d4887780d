// Issue-rate probe for gfx950: instructions per cycle per CU for SALU, VALU, v_readlane and mixed streams,
// as a function of waves per CU.  build: hipcc --offload-arch=gfx950 -O3 -o issue_rate.bin issue_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* o, int iters, uint64_t* cyc)
{
    uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3, v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0)      // independent SALU
            asm volatile(REP64("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
        if (MODE == 1)      // independent VALU
            asm volatile(REP64("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 2)      // dependent SALU chain
            asm volatile(REP64("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n")
                         : "+s"(s0) :: "scc");
        if (MODE == 3)      // dependent VALU chain
            asm volatile(REP64("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n")
                         : "+v"(v0));
        if (MODE == 4)      // the hop: readlane -> s_add -> (bitset) chain
            asm volatile(REP64("v_readlane_b32 %1, %2, %0\n s_bitset1_b64 %3, %0\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, 63\n")
                         : "+s"(s0), "+s"(s1) : "v"(v1), "s"((uint64_t)0) : "scc");
        if (MODE == 5)      // VALU + SALU interleaved, independent
            asm volatile(REP64("v_add_u32 %0, %0, 1\n s_add_u32 %2, %2, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %3, %3, 1\n")
                         : "+v"(v0), "+v"(v1), "+s"(s0), "+s"(s1) :: "scc");
        if (MODE == 6)      // v_cmp -> s_and -> v_cndmask ping-pong (dependent through SGPR pair)
            asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %1, %1, 1\n")
                         : "+v"(v0), "+v"(v1) :: "vcc", "scc");
        if (MODE == 7)      // SALU + never-taken branch pairs
            asm volatile(REP64("s_add_u32 %0, %0, 1\n s_cbranch_execz 1f\n s_add_u32 %1, %1, 1\n s_cbranch_execz 1f\n") "1:\n"
                         : "+s"(s0), "+s"(s1) :: "scc");
        if (MODE == 8)      // SALU + s_nop pairs
            asm volatile(REP64("s_add_u32 %0, %0, 1\n s_nop 0\n s_add_u32 %1, %1, 1\n s_nop 0\n")
                         : "+s"(s0), "+s"(s1) :: "scc");
        if (MODE == 9)      // SALU + s_waitcnt pairs
            asm volatile(REP64("s_add_u32 %0, %0, 1\n s_waitcnt lgkmcnt(0)\n s_add_u32 %1, %1, 1\n s_waitcnt vmcnt(0)\n")
                         : "+s"(s0), "+s"(s1) :: "scc");
        if (MODE == 10)     // hop with VALU marking: readlane, writelane, s_add, s_and
            asm volatile(REP64("v_readlane_b32 %1, %2, %0\n v_writelane_b32 %3, 1, %0\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, 63\n")
                         : "+s"(s0), "+s"(s1), "+v"(v1), "+v"(v2) :: "scc");
        if (MODE == 11)     // v_cmp (VOPC writes vcc) only
            asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0\n")
                         :: "v"(v0), "v"(v1) : "vcc");
        if (MODE == 12)     // s_and_saveexec / s_or exec pairs (exec-mask control flow)
            asm volatile(REP64("s_and_saveexec_b64 %0, exec\n s_or_b64 exec, exec, %0\n s_and_saveexec_b64 %0, exec\n s_or_b64 exec, exec, %0\n")
                         : "=s"(*(uint64_t*)&s0) :: "scc");
        if (MODE == 13)     // v_cmp + s_add interleaved: do VOPC SGPR writes share the scalar unit's budget?
            asm volatile(REP64("v_cmp_lt_u32 vcc, %2, %3\n s_add_u32 %0, %0, 1\n v_cmp_lt_u32 vcc, %3, %2\n s_add_u32 %1, %1, 1\n")
                         : "+s"(s0), "+s"(s1) : "v"(v0), "v"(v1) : "vcc", "scc");
        if (MODE == 14)     // v_cmpx (writes EXEC and vcc)
            asm volatile(REP64("v_cmpx_le_u32 vcc, 0, %0\n v_cmpx_le_u32 vcc, 0, %0\n v_cmpx_le_u32 vcc, 0, %0\n v_cmpx_le_u32 vcc, 0, %0\n")
                         :: "v"(v0) : "vcc");
        if (MODE == 15)     // v_cndmask reading vcc
            asm volatile(REP64("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n")
                         : "+v"(v0), "+v"(v1) :: "vcc");
        if (MODE == 16)     // v_readlane only (independent destinations)
            asm volatile(REP64("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n")
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(v0));
        if (MODE == 17)     // v_writelane only
            asm volatile(REP64("v_writelane_b32 %0, 1, 3\n v_writelane_b32 %1, 1, 5\n v_writelane_b32 %0, 1, 7\n v_writelane_b32 %1, 1, 9\n")
                         : "+v"(v0), "+v"(v1));
        if (MODE == 18)     // v_cmp_e64 into an SGPR pair + s_and on it (ballot-style)
            asm volatile(REP64("v_cmp_lt_u32_e64 %0, %1, %2\n v_cmp_lt_u32_e64 %0, %2, %1\n v_cmp_lt_u32_e64 %0, %1, %2\n v_cmp_lt_u32_e64 %0, %2, %1\n")
                         : "=s"(*(uint64_t*)&s0) : "v"(v0), "v"(v1));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    o[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, uint32_t* o, uint64_t* cyc)
{
    const int iters = 2000;
    const double ninstr = (double)iters * 64 * 4;
    printf("%-34s", name);
    for (int wpc : {1, 4, 8, 16, 32}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<256 * wpc, 64>>>(o, iters, cyc); hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MODE><<<256 * wpc, 64>>>(o, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint64_t h[64]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 64; i++) s += (double)h[i];
        const double cpw = s / 64 / ninstr;              // counter ticks per instruction per wave (first 64 blocks)
        // wall clock: wave-instructions per ns per CU (x clock period = per cycle)
        printf("  %2dw: %5.2f t/i/wave, %5.2f instr/ns/CU", wpc, cpw, ninstr * 256.0 * wpc / (ms * 1e6) / 256.0);
    }
    printf("\n");
}

int main()
{
    uint32_t* o; uint64_t* cyc;
    hipMalloc(&o, 256 * 32 * 256); hipMalloc(&cyc, 256 * 32 * 8);
    run<0>("SALU independent", o, cyc);
    run<2>("SALU dependent chain", o, cyc);
    run<1>("VALU independent", o, cyc);
    run<3>("VALU dependent chain", o, cyc);
    run<4>("hop (readlane,bitset,add,and)", o, cyc);
    run<5>("VALU+SALU interleaved", o, cyc);
    run<6>("v_cmp/s_and/v_cndmask/v_add", o, cyc);
    run<7>("SALU + untaken branch", o, cyc);
    run<8>("SALU + s_nop", o, cyc);
    run<9>("SALU + s_waitcnt", o, cyc);
    run<10>("hop (readlane,writelane,add,and)", o, cyc);
    run<11>("v_cmp -> vcc", o, cyc);
    run<12>("s_and_saveexec / s_or exec", o, cyc);
    run<13>("v_cmp + s_add interleaved", o, cyc);
    run<14>("v_cmpx", o, cyc);
    run<15>("v_cndmask (vcc)", o, cyc);
    run<16>("v_readlane", o, cyc);
    run<17>("v_writelane", o, cyc);
    run<18>("v_cmp_e64 -> sgpr pair", o, cyc);
    return 0;
}
