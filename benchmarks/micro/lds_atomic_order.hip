// Do the lanes of one ds_add_rtn_u32 that hit the SAME LDS word get their pre-op values in lane order?  (gfx950 probe.)
// If they do, a wave can rank equal radix digits with one returning add per element instead of a match-any (33 vector
// instructions).  Every trial: 64 lanes pick words by a pattern, add 1 with return, and the value each lane got is compared with the
// number of LOWER lanes that picked the same word.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic_order.hip -o /tmp/lds_atomic_order && /tmp/lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint32_t* bad, uint32_t* trials, uint32_t rounds, uint32_t seed)
{
    __shared__ uint32_t w[4][1024];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = seed * 2654435761u + blockIdx.x * 40503u + wave * 977u + 1u;
    uint32_t nbad = 0, ntr = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t mode = (x >> 28) & 7u, k = 1u + ((x >> 8) % 64u), stride = 1u + ((x >> 16) & 63u);
        uint32_t h = (lane * 2246822519u + x) ; h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
        uint32_t a;
        switch (mode) {
        case 0: a = 0; break;                                      // all the same word
        case 1: a = h % k; break;                                  // random over k words
        case 2: a = (h % k) * stride; break;                       // ... spread over the banks
        case 3: a = (lane / k) * stride; break;                    // runs of k lanes
        case 4: a = (lane % k) * 32u; break;                       // same bank, different words
        case 5: a = (h & 1u) ? 5u : (lane * 3u) % 997u; break;    // half on one word, half scattered
        case 6: a = ((lane ^ (x & 63u)) % k) * stride; break;
        default: a = (h >> 7) % 256u; break;                       // a radix digit
        }
        a %= 1024u;
        for (uint32_t i = lane; i < 1024u; i += 64u) w[wave][i] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t got = atomicAdd(&w[wave][a], 1u);
        uint32_t want = 0;
        for (uint32_t l = 0; l < 64u; l++) {
            const uint32_t al = (uint32_t)__builtin_amdgcn_readlane((int)a, (int)l);
            if (al == a && l < lane) want++;
        }
        if (got != want) nbad++;
        ntr++;
        __builtin_amdgcn_wave_barrier();
    }
    atomicAdd(bad, nbad); atomicAdd(trials, ntr);
}
int main()
{
    uint32_t *d, h[2] = {0, 0};
    hipMalloc(&d, 8); hipMemset(d, 0, 8);
    for (uint32_t s = 0; s < 8; s++) hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, d, d + 1, 2000u, s);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("lane-trials %u, out of lane order %u\n", h[1], h[0]);
    return h[0] != 0;
}
