// Microbenchmark + correctness probe: unaligned 8/4/2-byte LDS accesses on gfx950 (does the hardware honour
// them at any byte address, and what do they cost next to byte accesses?).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint16_t __attribute__((aligned(1))) u16u;

template <int MODE>
__global__ __launch_bounds__(64) void k(uint8_t* o, const uint32_t* src, const uint32_t* dst, int iters, uint64_t* cyc)
{
    __shared__ __align__(16) uint8_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = (uint8_t)(i * 7 + (i >> 8));
    __syncthreads();
    const uint32_t a = src[threadIdx.x], d = dst[threadIdx.x];
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        const uint32_t aa = (a + 13 * it) & 4095, dd = 4096 + 64 * threadIdx.x + ((d + 13 * it) % 56);
        if (MODE == 0) {
            uint8_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = ((volatile uint8_t*)s)[aa + u];
#pragma unroll
            for (int u = 0; u < 8; u++) ((volatile uint8_t*)s)[dd + u] = v[u];
        } else if (MODE == 1) {
            const uint64_t v = *(const u64u*)(s + aa);
            *(u64u*)(s + dd) = v;
        } else {
            const uint64_t v = *(const u64u*)(s + aa);
            *(u32u*)(s + dd) = (uint32_t)v;
            *(u16u*)(s + dd + 4) = (uint16_t)(v >> 32);
            s[dd + 6] = (uint8_t)(v >> 48);
            s[dd + 7] = (uint8_t)(v >> 56);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    __syncthreads();
    for (int i = threadIdx.x; i < 8192; i += 64) o[(size_t)blockIdx.x * 8192 + i] = s[i];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int nb = 4096, iters = 2000;
    uint32_t hs[64], hd[64];
    srand(5);
    for (int i = 0; i < 64; i++) { hs[i] = rand() & 4095; hd[i] = rand() % 56; }   // each lane writes inside its own 64-byte cell
    uint32_t *ds_, *dd_; uint8_t* o[3]; uint64_t* cyc;
    hipMalloc(&ds_, 256); hipMalloc(&dd_, 256); hipMalloc(&cyc, nb * 8);
    hipMemcpy(ds_, hs, 256, hipMemcpyHostToDevice); hipMemcpy(dd_, hd, 256, hipMemcpyHostToDevice);
    uint8_t* h[3];
    for (int m = 0; m < 3; m++) { hipMalloc(&o[m], (size_t)nb * 8192); h[m] = (uint8_t*)malloc(8192); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int m = 0; m < 3; m++) {
        for (int blocks : {256, 4096}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (m == 0) k<0><<<blocks, 64>>>(o[m], ds_, dd_, iters, cyc);
                if (m == 1) k<1><<<blocks, 64>>>(o[m], ds_, dd_, iters, cyc);
                if (m == 2) k<2><<<blocks, 64>>>(o[m], ds_, dd_, iters, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            uint64_t c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
            printf("mode %d blocks %4d: %.3f ms, %.1f cycles/iteration (wave 0)\n", m, blocks, ms, (double)c0 / iters);
        }
        hipMemcpy(h[m], o[m], 8192, hipMemcpyDeviceToHost);
    }
    printf("b64 == bytes: %s;  split == bytes: %s\n", memcmp(h[0], h[1], 8192) ? "MISMATCH" : "ok", memcmp(h[0], h[2], 8192) ? "MISMATCH" : "ok");
    return 0;
}
