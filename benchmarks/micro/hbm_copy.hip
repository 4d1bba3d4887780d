// What a plain device-to-device copy reaches on gfx950, by kernel shape (for rcx_hbm_copy_probe): GB/s = 2 * bytes / time.
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_copy.bin hbm_copy.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((vector_size(16)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_stride(const u32x4* __restrict__ s, u32x4* __restrict__ d, uint64_t n)
{
    const uint64_t st = (uint64_t)gridDim.x * 256u;
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    for (; i + (U - 1) * st < n; i += U * st) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(s + i + u * st) : s[i + u * st];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * st); else d[i + u * st] = v[u]; }
    }
    for (; i < n; i += st) d[i] = s[i];
}
// one shot: a workgroup copies a contiguous tile of 256 * U chunks
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_tile(const u32x4* __restrict__ s, u32x4* __restrict__ d, uint64_t n)
{
    const uint64_t base = (uint64_t)blockIdx.x * 256u * U + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + u * 256u < n) v[u] = NT ? __builtin_nontemporal_load(s + base + u * 256u) : s[base + u * 256u];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + u * 256u < n) { if (NT) __builtin_nontemporal_store(v[u], d + base + u * 256u); else d[base + u * 256u] = v[u]; }
}
template <class F> static double run(F launch, uint64_t bytes)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 10; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * bytes * 10 / (ms * 1e-3) / 1e9;
}
int main()
{
    const uint64_t bytes = 1ull << 30, n = bytes / 16;
    u32x4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
    for (int wgcu : {4, 8, 16, 32}) {
        const int g = 256 * wgcu;
        printf("stride grid %5d: U1 %.0f  U4 %.0f  U8 %.0f  U4nt %.0f GB/s\n", g,
               run([&] { k_stride<1, false><<<g, 256>>>(a, b, n); }, bytes), run([&] { k_stride<4, false><<<g, 256>>>(a, b, n); }, bytes),
               run([&] { k_stride<8, false><<<g, 256>>>(a, b, n); }, bytes), run([&] { k_stride<4, true><<<g, 256>>>(a, b, n); }, bytes));
    }
    printf("tile: U1 %.0f  U2 %.0f  U4 %.0f  U8 %.0f  U4nt %.0f GB/s\n",
           run([&] { k_tile<1, false><<<(unsigned)(n / 256), 256>>>(a, b, n); }, bytes), run([&] { k_tile<2, false><<<(unsigned)(n / 512), 256>>>(a, b, n); }, bytes),
           run([&] { k_tile<4, false><<<(unsigned)(n / 1024), 256>>>(a, b, n); }, bytes), run([&] { k_tile<8, false><<<(unsigned)(n / 2048), 256>>>(a, b, n); }, bytes),
           run([&] { k_tile<4, true><<<(unsigned)(n / 1024), 256>>>(a, b, n); }, bytes));
    double g = 0; hipMemcpy(b, a, bytes, hipMemcpyDeviceToDevice); hipDeviceSynchronize();
    printf("hipMemcpy D2D: %.0f GB/s\n", run([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes));
    return 0;
}
