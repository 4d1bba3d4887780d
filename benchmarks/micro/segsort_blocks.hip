// Round 0 of the BWT suffix sort, two ways: the device-wide 46-bit radix sort of (block | 4 symbols) used now against
// rocprim::segmented_radix_sort_pairs over the 36 symbol bits with one segment per block (1024 x 256 Ki suffixes).
// build: hipcc --offload-arch=gfx950 -O3 -o segsort_blocks.bin segsort_blocks.hip
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void fill(uint64_t* k, uint32_t* v, uint32_t n, uint32_t seg)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    k[i] = ((uint64_t)(i / seg) << 36) | (x & ((1ull << 36) - 1));     // text-like keys are far less uniform; timing only
    v[i] = i;
}
int main()
{
    const uint32_t SEG = 262144, NS = 1024, N = SEG * NS;
    uint64_t *K0, *K1; uint32_t *v0, *v1, *off;
    (void)hipMalloc(&K0, 8ull * N); (void)hipMalloc(&K1, 8ull * N); (void)hipMalloc(&v0, 4ull * N); (void)hipMalloc(&v1, 4ull * N);
    (void)hipMalloc(&off, 4ull * (NS + 1));
    std::vector<uint32_t> h(NS + 1); for (uint32_t i = 0; i <= NS; i++) h[i] = i * SEG;
    (void)hipMemcpy(off, h.data(), 4ull * (NS + 1), hipMemcpyHostToDevice);
    size_t t1 = 0, t2 = 0; void* tmp = nullptr;
    rocprim::double_buffer<uint64_t> dk(K0, K1); rocprim::double_buffer<uint32_t> dv(v0, v1);
    (void)rocprim::radix_sort_pairs(nullptr, t1, dk, dv, N, 0, 46);
    (void)rocprim::segmented_radix_sort_pairs(nullptr, t2, dk, dv, N, NS, off, off + 1, 0, 36);
    (void)hipMalloc(&tmp, t1 > t2 ? t1 : t2);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++)
        for (int rep = 0; rep < 2; rep++) {
            fill<<<(N + 255) / 256, 256>>>(dk.current(), dv.current(), N, SEG);
            (void)hipEventRecord(e0);
            if (mode == 0) (void)rocprim::radix_sort_pairs(tmp, t1, dk, dv, N, 0, 46);
            if (mode == 1) (void)rocprim::segmented_radix_sort_pairs(tmp, t2, dk, dv, N, NS, off, off + 1, 0, 36);
            if (mode == 2) (void)rocprim::radix_sort_pairs(tmp, t1, dk, dv, N, 0, 40);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.2f ms\n", mode == 0 ? "device-wide, 46 bits" : mode == 1 ? "segmented (1024 x 256 Ki), 36 bits" : "device-wide, 40 bits", ms);
        }
    return 0;
}
