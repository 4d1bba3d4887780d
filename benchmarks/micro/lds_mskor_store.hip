// Byte-exact LDS store of n (0..16) bytes at any address, two ways, on gfx950: correctness across lanes that share a dword, and cost at
// sixteen waves per CU.
//   A: the exec-narrowing byte stores of rcx_lds_store16 (v_cmpx + ds_write_b8 per byte)
//   B: five ds_mskor_b32 on the aligned dwords of the 20-byte frame that holds [p, p + n), masks from a 20-entry table in LDS
//      (entry e = a + n: bytes below e are 0xff; the first dword's low a bytes are cleared with one shift)
// build: hipcc --offload-arch=gfx950 -O3 -o lds_mskor_store.bin lds_mskor_store.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

__device__ __forceinline__ void store_a(uint32_t a, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t nv)
{
    uint32_t t; uint64_t sv;
#define ST4(V, O0, O1, O2, O3)                                        \
        "v_cmpx_lt_u32_e32 vcc, " #O0 ", %[nv]\n\t"                       \
        "s_cbranch_execz L_end_%=\n\t"                                    \
        "ds_write_b8 %[a], %[" V "] offset:" #O0 "\n\t"                    \
        "v_cmpx_lt_u32_e32 vcc, " #O1 ", %[nv]\n\t"                       \
        "v_lshrrev_b32_e32 %[t], 8, %[" V "]\n\t"                         \
        "ds_write_b8 %[a], %[t] offset:" #O1 "\n\t"                       \
        "v_cmpx_lt_u32_e32 vcc, " #O2 ", %[nv]\n\t"                       \
        "ds_write_b8_d16_hi %[a], %[" V "] offset:" #O2 "\n\t"             \
        "v_cmpx_lt_u32_e32 vcc, " #O3 ", %[nv]\n\t"                       \
        "ds_write_b8_d16_hi %[a], %[t] offset:" #O3 "\n\t"
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        ST4("v0", 0, 1, 2, 3) ST4("v1", 4, 5, 6, 7) ST4("v2", 8, 9, 10, 11) ST4("v3", 12, 13, 14, 15)
        "L_end_%=:\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [t] "=&v"(t), [sv] "=&s"(sv)
        : [a] "v"(a), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [nv] "v"(nv)
        : "vcc", "memory");
#undef ST4
}

// w0..w4: the frame's dwords, already in place (dword k holds what belongs at (p & ~3) + 4k); tbl: LDS address of the mask table
__device__ __forceinline__ void store_b(uint32_t p, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t nv, uint32_t tbl)
{
    const uint32_t a = p & 3u, e = a + nv;
    const uint32_t ta = tbl + (e << 5);
    uint32_t m0, m1, m2, m3, m4;
    __attribute__((ext_vector_type(4))) uint32_t mv;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=v"(mv), "=v"(m4) : "v"(ta) : "memory");
    m0 = mv.x & (0xffffffffu << (8u * a)); m1 = mv.y; m2 = mv.z; m3 = mv.w;
    const uint32_t pa = p & ~3u;
    asm volatile("ds_mskor_b32 %0, %1, %2\n\t"
                 "ds_mskor_b32 %0, %3, %4 offset:4\n\t"
                 "ds_mskor_b32 %0, %5, %6 offset:8\n\t"
                 "ds_mskor_b32 %0, %7, %8 offset:12\n\t"
                 "ds_mskor_b32 %0, %9, %10 offset:16\n\t"
                 :: "v"(pa), "v"(m0), "v"(w0 & m0), "v"(m1), "v"(w1 & m1), "v"(m2), "v"(w2 & m2), "v"(m3), "v"(w3 & m3), "v"(m4), "v"(w4 & m4) : "memory");
}

// mode 0: A, 1: B.  Lane l stores n[l] bytes of value (l + 1 + it) at offset off[l] of a 2 KiB buffer; checked on the host for it = 0
template <int MODE>
__global__ __launch_bounds__(64) void k(uint8_t* o, const uint32_t* off, const uint32_t* nn, int iters, uint64_t* cyc, int needand)
{
    __shared__ __align__(16) uint8_t s[2048 + 64];
    __shared__ __align__(16) uint32_t tb[20 * 8];
    for (int i = threadIdx.x; i < 2048 + 64; i += 64) s[i] = 0xee;
    for (int e = threadIdx.x; e < 20; e += 64) for (int k2 = 0; k2 < 8; k2++) {
        uint32_t m = 0;
        for (int b = 0; b < 4; b++) if (k2 < 5 && 4 * k2 + b < e) m |= 0xffu << (8 * b);
        tb[e * 8 + k2] = m;
    }
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)s, tbl = (uint32_t)(uintptr_t)tb;
    const uint32_t p = base + off[threadIdx.x], n = nn[threadIdx.x];
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        const uint32_t val = ((threadIdx.x + 1 + it) & 0xffu) * 0x01010101u;
        if (MODE == 0) store_a(p, val, val, val, val, n);
        else store_b(p, val, val, val, val, val, n, tbl);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < 2048; i += 64) o[i] = s[i];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    uint8_t* o; uint32_t *doff, *dn; uint64_t* cyc;
    hipMalloc(&o, 2048); hipMalloc(&doff, 256); hipMalloc(&dn, 256); hipMalloc(&cyc, 4096 * 8);
    srand(5);
    int bad = 0;
    for (int trial = 0; trial < 200; trial++) {
        // contiguous entries like a batch: lane l's bytes follow lane l - 1's
        uint32_t off[64], n[64]; uint32_t pos = rand() % 16;
        for (int l = 0; l < 64; l++) { n[l] = (trial < 100) ? rand() % 17 : 4 + rand() % 10; off[l] = pos; pos += n[l]; }
        hipMemcpy(doff, off, 256, hipMemcpyHostToDevice); hipMemcpy(dn, n, 256, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; mode++) {
            if (mode == 0) k<0><<<1, 64>>>(o, doff, dn, 1, cyc, 0); else k<1><<<1, 64>>>(o, doff, dn, 1, cyc, 0);
            hipDeviceSynchronize();
            std::vector<uint8_t> h(2048), ref(2048, 0xee);
            hipMemcpy(h.data(), o, 2048, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; l++) for (uint32_t b = 0; b < n[l]; b++) ref[off[l] + b] = (uint8_t)(l + 1);
            if (memcmp(h.data(), ref.data(), 2048)) { bad++; if (bad < 4) printf("MISMATCH trial %d mode %d\n", trial, mode); }
        }
    }
    printf("correctness: %d mismatches in 400 runs\n", bad);
    for (int dist = 0; dist < 2; dist++) {
        uint32_t off[64], n[64]; uint32_t pos = 5;
        for (int l = 0; l < 64; l++) { n[l] = dist == 0 ? 4 + rand() % 13 : 16; off[l] = pos; pos += n[l]; }
        hipMemcpy(doff, off, 256, hipMemcpyHostToDevice); hipMemcpy(dn, n, 256, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; mode++) {
            double c[2]; int bl[2] = {256, 4096};
            for (int j = 0; j < 2; j++) {
                for (int rep = 0; rep < 2; rep++) { if (mode == 0) k<0><<<bl[j], 64>>>(o, doff, dn, 2000, cyc, 0); else k<1><<<bl[j], 64>>>(o, doff, dn, 2000, cyc, 0); hipDeviceSynchronize(); }
                uint64_t h[64]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
                double s = 0; for (int i = 0; i < 64; i++) s += (double)h[i];
                c[j] = s / 64 / 2000;
            }
            printf("%s lens %-6s: 1 wave/CU %7.1f cyc/store   16 waves/CU %7.1f cyc/store/wave = %6.1f cyc/store/CU\n", mode ? "B mskor+table" : "A byte stores ", dist ? "16" : "4..16", c[0], c[1], c[1] / 16);
        }
    }
    return bad != 0;
}
