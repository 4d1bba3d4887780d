// Feasibility probe: rocprim::segmented_radix_sort_pairs on many tiny segments (what a group-wise BWT refinement round
// would need) against the device-wide 48-bit radix_sort_pairs the suffix sorter uses now.
// build: hipcc --offload-arch=gfx950 -O3 -o segsort.bin segsort.hip
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
int main()
{
    const uint32_t N = 200u << 20;
    for (int avg : {3, 8, 64, 2000}) {
        std::vector<uint32_t> hoff; hoff.reserve(N / 2 + 2);
        uint32_t p = 0; srand(7);
        while (p < N) { hoff.push_back(p); p += 2 + rand() % (2 * avg - 3); }
        const uint32_t S = (uint32_t)hoff.size(); hoff.push_back(N);
        uint32_t *k0, *k1, *v0, *v1, *off; uint64_t *K0, *K1;
        hipMalloc(&k0, 4ull * N); hipMalloc(&k1, 4ull * N); hipMalloc(&v0, 4ull * N); hipMalloc(&v1, 4ull * N); hipMalloc(&off, 4ull * (S + 1));
        hipMalloc(&K0, 8ull * N); hipMalloc(&K1, 8ull * N);
        hipMemcpy(off, hoff.data(), 4ull * (S + 1), hipMemcpyHostToDevice);
        hipMemset(k0, 0x5a, 4ull * N); hipMemset(K0, 0x5a, 8ull * N);
        size_t tb = 0, tb2 = 0; void* tmp = nullptr;
        rocprim::segmented_radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, N, S, off, off + 1, 0, 19);
        rocprim::double_buffer<uint64_t> dk(K0, K1); rocprim::double_buffer<uint32_t> dv(v0, v1);
        rocprim::radix_sort_pairs(nullptr, tb2, dk, dv, N, 0, 48);
        hipMalloc(&tmp, tb > tb2 ? tb : tb2);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms1 = 0, ms2 = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            rocprim::segmented_radix_sort_pairs(tmp, tb, k0, k1, v0, v1, N, S, off, off + 1, 0, 19);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
            hipEventRecord(e0);
            rocprim::radix_sort_pairs(tmp, tb2, dk, dv, N, 0, 48);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms2, e0, e1);
        }
        printf("N=%u segments=%u (avg %d): segmented 19-bit %.2f ms, device-wide 48-bit %.2f ms\n", N, S, avg, ms1, ms2);
        hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(off); hipFree(K0); hipFree(K1); hipFree(tmp);
    }
    return 0;
}
