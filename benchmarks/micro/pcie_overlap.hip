// pcie_overlap.hip -- can the two directions of the host link run at once on this box, and by what means?
// benchmarks/micro/pcie_duplex.py (round 2) found that an H2D and a D2H hipMemcpyAsync on two streams take as long as one after
// the other (57 GB/s each way).  Copy ENGINES may be the shared resource, not the link: this measures the same pair with the
// copies done by kernels (loads / stores of page-locked host memory mapped into the device's address space) and mixed.
//   hipcc --offload-arch=gfx950 -O3 -o pcie_overlap pcie_overlap.hip && ./pcie_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((vector_size(16)));
__global__ void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v = __builtin_nontemporal_load(src + i);
        __builtin_nontemporal_store(v, dst + i);
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t IN = 98u << 20, OUT = 268u << 20;
    void *hin, *hout, *din, *dout;
    CHK(hipHostMalloc(&hin, IN, hipHostMallocDefault)); CHK(hipHostMalloc(&hout, OUT, hipHostMallocDefault));
    CHK(hipMalloc(&din, IN)); CHK(hipMalloc(&dout, OUT));
    memset(hin, 1, IN); memset(hout, 0, OUT);
    CHK(hipMemset(dout, 2, OUT));
    hipStream_t s1, s2;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    void *dhin, *dhout;                       // device-side addresses of the page-locked buffers
    CHK(hipHostGetDevicePointer(&dhin, hin, 0)); CHK(hipHostGetDevicePointer(&dhout, hout, 0));
    auto run = [&](const char* name, int h2d_kind, int d2h_kind, int grid_in, int grid_out) {   // kind: 0 none, 1 hipMemcpyAsync, 2 kernel
        double best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            CHK(hipDeviceSynchronize());
            const double t0 = now();
            if (h2d_kind == 1) CHK(hipMemcpyAsync(din, hin, IN, hipMemcpyHostToDevice, s1));
            if (h2d_kind == 2) hipLaunchKernelGGL(k_copy, dim3(grid_in), dim3(256), 0, s1, (const u32x4*)dhin, (u32x4*)din, IN / 16);
            if (d2h_kind == 1) CHK(hipMemcpyAsync(hout, dout, OUT, hipMemcpyDeviceToHost, s2));
            if (d2h_kind == 2) hipLaunchKernelGGL(k_copy, dim3(grid_out), dim3(256), 0, s2, (const u32x4*)dout, (u32x4*)dhout, OUT / 16);
            CHK(hipStreamSynchronize(s1)); CHK(hipStreamSynchronize(s2));
            const double t = now() - t0;
            if (rep && t < best) best = t;
        }
        const double bytes = (h2d_kind ? IN : 0) + (d2h_kind ? OUT : 0);
        printf("%-58s %7.2f ms  %6.1f GB/s total\n", name, best * 1e3, bytes / best / 1e9);
    };
    run("H2D 98 MB, copy engine", 1, 0, 0, 0);
    run("D2H 268 MB, copy engine", 0, 1, 0, 0);
    run("both, copy engines, two streams", 1, 1, 0, 0);
    for (int g : {64, 256, 1024}) {
        char nm[96];
        snprintf(nm, sizeof nm, "H2D by kernel (%d workgroups)", g); run(nm, 2, 0, g, g);
        snprintf(nm, sizeof nm, "D2H by kernel (%d workgroups)", g); run(nm, 0, 2, g, g);
        snprintf(nm, sizeof nm, "both by kernels (%d workgroups each)", g); run(nm, 2, 2, g, g);
        snprintf(nm, sizeof nm, "H2D copy engine + D2H kernel (%d)", g); run(nm, 1, 2, g, g);
        snprintf(nm, sizeof nm, "H2D kernel (%d) + D2H copy engine", g); run(nm, 2, 1, g, g);
    }
    // pageable buffers through the runtime, for the record
    void* pin = malloc(IN); void* pout = malloc(OUT); memset(pin, 1, IN); memset(pout, 0, OUT);
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipDeviceSynchronize());
        double t0 = now(); CHK(hipMemcpy(din, pin, IN, hipMemcpyHostToDevice)); double t1 = now(); CHK(hipMemcpy(pout, dout, OUT, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pageable: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, IN / (t1 - t0) / 1e9, (t2 - t1) * 1e3, OUT / (t2 - t1) / 1e9);
    }
    // hipHostRegister cost on pageable memory
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); CHK(hipHostRegister(pout, OUT, hipHostRegisterDefault)); double t1 = now(); CHK(hipHostUnregister(pout)); double t2 = now();
        printf("hipHostRegister 268 MB: %.2f ms, unregister %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
    }
    return 0;
}
