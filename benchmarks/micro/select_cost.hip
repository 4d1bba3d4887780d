// What does a per-lane select cost on gfx950?  v_cndmask with an SGPR-pair mask against arithmetic selects
// (sign-smear + v_bfi / v_and) and other VALU forms that read scalar registers.  Wave-instructions per ns per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o select_cost.bin select_cost.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* o, int iters, uint64_t* cyc)
{
    uint32_t v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 5, v5 = 7;
    uint32_t s0 = blockIdx.x;
    uint64_t m = 0x5555555555555555ull;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0)      // v_cndmask, implicit vcc, independent destinations
            asm volatile(REP64("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %5, %4, vcc\n")
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4), "v"(v5) : "vcc");
        if (MODE == 1)      // v_cndmask_e64, mask in an SGPR pair
            asm volatile(REP64("v_cndmask_b32_e64 %0, %4, %5, %6\n v_cndmask_b32_e64 %1, %5, %4, %6\n v_cndmask_b32_e64 %2, %4, %5, %6\n v_cndmask_b32_e64 %3, %5, %4, %6\n")
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4), "v"(v5), "s"(m));
        if (MODE == 2)      // the usual pair: v_cmp -> vcc, v_cndmask
            asm volatile(REP64("v_cmp_lt_u32 vcc, %4, %0\n v_cndmask_b32 %0, %4, %5, vcc\n v_cmp_lt_u32 vcc, %5, %1\n v_cndmask_b32 %1, %5, %4, vcc\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5) : "vcc");
        if (MODE == 3)      // arithmetic select: sub, sign smear, bfi  (a < b ? x : y for values < 2^31)
            asm volatile(REP64("v_sub_u32 %2, %4, %0\n v_ashrrev_i32 %2, 31, %2\n v_bfi_b32 %0, %2, %4, %5\n v_sub_u32 %3, %5, %1\n v_ashrrev_i32 %3, 31, %3\n v_bfi_b32 %1, %3, %5, %4\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5));
        if (MODE == 4)      // VALU with one SGPR source
            asm volatile(REP64("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(s0));
        if (MODE == 5)      // v_mov from SGPR
            asm volatile(REP64("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n")
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "s"(s0));
        if (MODE == 6)      // carry chain through vcc
            asm volatile(REP64("v_add_co_u32 %0, vcc, %4, %0\n v_addc_co_u32 %1, vcc, %5, %1, vcc\n v_add_co_u32 %2, vcc, %4, %2\n v_addc_co_u32 %3, vcc, %5, %3, vcc\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5) : "vcc");
        if (MODE == 7)      // v_cndmask dependent chain (each result feeds the next)
            asm volatile(REP64("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n")
                         : "+v"(v0), "+v"(v1) :: "vcc");
        if (MODE == 8)      // exec-masked move instead of a select: s_mov exec, mask; v_mov; s_mov exec, -1
            asm volatile(REP64("s_mov_b64 exec, %4\n v_mov_b32 %0, %2\n s_mov_b64 exec, -1\n v_mov_b32 %1, %3\n")
                         : "+v"(v0), "+v"(v1) : "v"(v4), "v"(v5), "s"(m));
        if (MODE == 9)      // v_min / v_max (selects that are really clamps)
            asm volatile(REP64("v_min_u32 %0, %4, %0\n v_max_u32 %1, %5, %1\n v_min_u32 %2, %4, %2\n v_max_u32 %3, %5, %3\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    o[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, uint32_t* o, uint64_t* cyc, double per_iter)
{
    const int iters = 2000;
    const double ninstr = (double)iters * 64 * per_iter;
    printf("%-40s", name);
    for (int wpc : {1, 4, 8, 16, 32}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<MODE><<<256 * wpc, 64>>>(o, iters, cyc); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<MODE><<<256 * wpc, 64>>>(o, iters, cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %2dw: %5.2f i/ns/CU", wpc, ninstr * wpc / (ms * 1e6));
    }
    printf("\n");
}

int main()
{
    uint32_t* o; uint64_t* cyc;
    (void)hipMalloc(&o, 256 * 32 * 256); (void)hipMalloc(&cyc, 256 * 32 * 8);
    run<0>("v_cndmask vcc, independent", o, cyc, 4);
    run<1>("v_cndmask_e64 sgpr pair", o, cyc, 4);
    run<7>("v_cndmask vcc, dependent chain", o, cyc, 4);
    run<2>("v_cmp + v_cndmask pairs", o, cyc, 4);
    run<3>("sub + ashr + bfi (arithmetic select)", o, cyc, 6);
    run<4>("v_add with an SGPR source", o, cyc, 4);
    run<5>("v_mov from SGPR", o, cyc, 4);
    run<6>("v_add_co / v_addc_co (vcc carry)", o, cyc, 4);
    run<8>("s_mov exec + v_mov (exec-masked move)", o, cyc, 4);
    run<9>("v_min / v_max", o, cyc, 4);
    return 0;
}
