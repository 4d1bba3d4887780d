// LDS cost table for gfx950: cycles per wave64 LDS instruction at saturation (16 waves/CU all issuing), by access
// width, alignment and address pattern.  READ and WRITE variants.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_cost.bin lds_cost.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// W: bytes per lane (1,2,4,8,16); WR: write instead of read
template <int W, bool WR>
__global__ __launch_bounds__(64) void k(uint32_t* o, const uint32_t* addr, int iters, uint64_t* cyc)
{
    __shared__ __align__(16) uint8_t s[8192 + 64];
    for (int i = threadIdx.x; i < 8192 + 64; i += 64) s[i] = (uint8_t)(i * 7 + (i >> 8));
    __syncthreads();
    const uint32_t a0 = addr[threadIdx.x];
    uint32_t acc = 0;
    uint32_t base = (uint32_t)(uintptr_t)s;   // LDS byte address (low 32 bits of the generic pointer are the LDS offset)
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t a = base + ((a0 + 64 * u * (W > 8 ? 2 : 1)) & 8191u & ~(W == 16 ? 15u : W == 18 ? 3u : 0u));
            if (!WR) {
                uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
                if (W == 1) asm volatile("ds_read_u8 %0, %1" : "=v"(v0) : "v"(a));
                if (W == 2) asm volatile("ds_read_u16 %0, %1" : "=v"(v0) : "v"(a));
                if (W == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(a));
                if (W == 8) { uint64_t v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); v0 = (uint32_t)v; v1 = (uint32_t)(v >> 32); }
                if (W == 16) { __attribute__((ext_vector_type(4))) uint32_t v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); v0 = v.x; v1 = v.y; v2 = v.z; v3 = v.w; }
                if (W == 17) { __attribute__((ext_vector_type(4))) uint32_t v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); v0 = v.x; v1 = v.y; v2 = v.z; v3 = v.w; }
                if (W == 18) { uint64_t v; asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(v) : "v"(a)); v0 = (uint32_t)v; v1 = (uint32_t)(v >> 32); }
                if (W == 19) { asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v0) : "v"(a & 255u), "v"(acc)); }
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc ^= v0 ^ v1 ^ v2 ^ v3;
            } else {
                const uint32_t v0 = acc + u;
                if (W == 1) asm volatile("ds_write_b8 %0, %1" :: "v"(a), "v"(v0) : "memory");
                if (W == 2) asm volatile("ds_write_b16 %0, %1" :: "v"(a), "v"(v0) : "memory");
                if (W == 4) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v0) : "memory");
                if (W == 8) { uint64_t v = ((uint64_t)v0 << 32) | v0; asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v) : "memory"); }
                if (W == 16) { __attribute__((ext_vector_type(4))) uint32_t v = {v0, v0, v0, v0}; asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v) : "memory"); }
                if (W == 20) asm volatile("ds_or_b32 %0, %1" :: "v"(a & ~3u), "v"(v0) : "memory");          // LDS atomic without return, aligned dword
                if (W == 21) asm volatile("ds_write_b32 %0, %1" :: "v"(a & ~3u), "v"(v0) : "memory");       // the same addresses, plain store
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    o[blockIdx.x * 64 + threadIdx.x] = acc + s[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W, bool WR>
static void run(const char* pat, const uint32_t* haddr, uint32_t* daddr, uint32_t* o, uint64_t* cyc)
{
    const int iters = 500;
    hipMemcpy(daddr, haddr, 256, hipMemcpyHostToDevice);
    double c[2];
    int bl[2] = {256, 4096};
    for (int j = 0; j < 2; j++) {
        for (int rep = 0; rep < 2; rep++) { k<W, WR><<<bl[j], 64>>>(o, daddr, iters, cyc); hipDeviceSynchronize(); }
        uint64_t h[64]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 64; i++) s += (double)h[i];
        c[j] = s / 64 / (iters * 8);
    }
    printf("%-5s W=%2d %-22s  1 wave/CU: %6.1f cyc/instr   16 waves/CU: %6.1f cyc/instr/wave = %5.1f cyc/instr/CU\n",
           WR ? "write" : "read", W, pat, c[0], c[1], c[1] / 16);
}

int main()
{
    uint32_t *daddr, *o; uint64_t* cyc;
    hipMalloc(&daddr, 256); hipMalloc(&o, 4096 * 256); hipMalloc(&cyc, 4096 * 8);
    uint32_t lin1[64], rnd[64], rnda[64], str[64];
    srand(3);
    auto pats = [&](int W) {
        for (int i = 0; i < 64; i++) {
            lin1[i] = W * i;                                   // consecutive, aligned
            rnda[i] = (rand() % (8192 / W)) * W;               // random, naturally aligned
            rnd[i] = rand() % 8000;                            // random, any byte address
            str[i] = 12 * i + 1;                               // LZ-like: lanes ~12 bytes apart, odd addresses
        }
    };
#define ALL(W)                                                                              \
    pats(W);                                                                                \
    run<W, false>("linear aligned", lin1, daddr, o, cyc); run<W, false>("random aligned", rnda, daddr, o, cyc); \
    if (W < 16) { run<W, false>("random unaligned", rnd, daddr, o, cyc); run<W, false>("stride12+1 (LZ-like)", str, daddr, o, cyc); } \
    run<W, true>("linear aligned", lin1, daddr, o, cyc); run<W, true>("random aligned", rnda, daddr, o, cyc);   \
    if (W < 16) { run<W, true>("random unaligned", rnd, daddr, o, cyc); run<W, true>("stride12+1 (LZ-like)", str, daddr, o, cyc); }
    ALL(1) ALL(2) ALL(4) ALL(8) ALL(16)
    pats(1);
    run<17, false>("b128 random unaligned", rnd, daddr, o, cyc); run<17, false>("b128 stride12+1", str, daddr, o, cyc);
    run<18, false>("read2_b32 random 4-al", rnd, daddr, o, cyc); run<18, false>("read2_b32 stride12+1", str, daddr, o, cyc);
    run<19, false>("bpermute random", rnd, daddr, o, cyc);
    run<20, true>("ds_or_b32 random", rnd, daddr, o, cyc); run<20, true>("ds_or_b32 stride12+1", str, daddr, o, cyc);
    run<21, true>("ds_write_b32 random", rnd, daddr, o, cyc); run<21, true>("ds_write_b32 stride12+1", str, daddr, o, cyc);
    return 0;
}
