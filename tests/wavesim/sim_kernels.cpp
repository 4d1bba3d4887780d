// sim_kernels.cpp -- runs the UNMODIFIED .hip kernel sources on the wave64 simulator (TEST INFRASTRUCTURE).
// Built by tests/wavesim/build.py with:  g++ -include wavesim.h sim_kernels.cpp wavesim.cpp
#define RCX_AB_VARIANTS 1              /* the simulator runs every kernel generation */
#include "../../benchmarks/experiments/k_lz4_decode_v1_v3.hip"
#include "../../rust_compress_amd/csrc/k_lz4_decode_v4.hip"
#include "../../rust_compress_amd/csrc/k_lz4_decode_v5.hip"
#include "../../benchmarks/experiments/k_lz4_decode_v7.hip"
#include "../../rust_compress_amd/csrc/k_lz4_decode_v8.hip"
#include "../../benchmarks/experiments/k_lz4_decode_v6.hip"
#include "../../rust_compress_amd/csrc/k_lz4_encode.hip"
#define hipStream_t int
static inline int hipMemsetAsync(void* d, int v, size_t n, int) { memset(d, v, n); return 0; }
#define hipLaunchKernelGGL(kern, grid, block, shm, stream, ...) ws::launch(grid, block, [&] { kern(__VA_ARGS__); })
#include "../../rust_compress_amd/csrc/k_serial.hip"
#include "../../rust_compress_amd/csrc/k_inflate.hip"
#include "../../rust_compress_amd/csrc/k_inflate2.hip"
#include "../../rust_compress_amd/csrc/k_inflate3.hip"
#include "../../rust_compress_amd/csrc/k_crc32.hip"
#include "../../rust_compress_amd/csrc/k_gzip.hip"
#define hipSuccess 0
#define hipMemcpyDeviceToHost 0
static inline int hipMemcpyAsync(void* d, const void* s, size_t n, int, int) { memcpy(d, s, n); return 0; }
static inline int hipStreamSynchronize(int) { return 0; }
#define hipMemcpyHostToDevice 1
#define hipHostMallocDefault 0
static inline int hipHostMalloc(void** p, size_t n, int) { *p = malloc(n); return *p ? 0 : 1; }
static inline int hipGetLastError() { return 0; }
#include "../../rust_compress_amd/csrc/k_bwt_inverse.hip"
#include "../../rust_compress_amd/csrc/k_bwt.hip"

extern "C" int sim_launch(int codec, int variant, const rcx_kargs* a)
{
    rcx_kargs k = *a;
    switch (codec) {
    case RCX_LZ4_DECODE:
        if (variant == 1) ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_decode_v1(k); });
        else if (variant == 2) ws::launch(dim3((k.nblocks + 3) / 4), dim3(256), [&] { k_lz4_decode_v3<2048, 2048, 64, 64, 4>(k); });
        else if (variant == 3) ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_decode_v3<4096, 2048, 64, 64, 1>(k); });
        else if (variant == 4) ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_decode_v3<2048, 2048, 32, 32, 1>(k); });
        else if (variant == 5) ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_decode_v2<4096, 2048, 64, 64, 1>(k); });
        else if (variant == 6) ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_decode_v3<2048, 2048, 64, 64, 1>(k); });
        else if (variant == 7) ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_decode_v4<2048, 1>(k); });
        else if (variant == 8) ws::launch(dim3((k.nblocks + 3) / 4), dim3(256), [&] { k_lz4_decode_v4<1024, 4>(k); });
        else if (variant == 10) ws::launch(dim3(k.nblocks), dim3(128), [&] { k_lz4_decode_v5<1024>(k); });
        else if (variant == 17) {
            ws::launch(dim3(k.nblocks), dim3(512), [&] { k_lz4_decode_v6<8>(k); });
            ws::launch(dim3(k.nblocks), dim3(128), [&] { k_lz4_decode_v5<2048, 1536, 2048>(k, (int)RCX_ST_BAIL6); });
        }
        else if (variant == 23) ws::launch(dim3(k.nblocks), dim3(128), [&] { k_lz4_decode_v8<1024, 768>(k); });
        else if (variant == 20) ws::launch(dim3(k.nblocks), dim3(128), [&] { k_lz4_decode_v7<1024, 1008, 2048, 2048>(k); });
        else if (variant == 18) ws::launch(dim3(k.nblocks), dim3(512), [&] { k_lz4_decode_v6<8>(k); });      // no second pass: which blocks bail
        else if (variant == 60) ws::launch(dim3(k.nblocks), dim3(128), [&] { k_lz4_decode_v8<1024, 768, false, 128, 32, false, 2, 0, true>(k); });   // + the mirror (and the gates)
        else if (variant == 0) ws::launch(dim3(k.nblocks), dim3(128), [&] { k_lz4_decode_v8<1024, 768>(k); });
        else if (variant == 15) ws::launch(dim3(k.nblocks), dim3(128), [&] { k_lz4_decode_v5<2048, 1536, 2048>(k); });
        else ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_decode_v4<1024, 1>(k); });
        return 0;
    case RCX_LZ4_ENCODE:
        if (variant == 1) ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_encode(k, 0); });
        else if (variant == 2) ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_encode_w<64>(k, 0); });
        else ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_encode_w<8>(k, 0); });
        return 0;
    case RCX_MTF_ENCODE: case RCX_MTF_DECODE: case RCX_DC_ENCODE: case RCX_DC_DECODE:
    case RCX_ARI_BYTE_ENCODE: case RCX_ARI_BYTE_DECODE: case RCX_RLE_ENCODE: case RCX_RLE_DECODE:
    case RCX_ARI_BINARY_ENCODE: case RCX_ARI_BINARY_DECODE: case RCX_ARI_PROXY_ENCODE: case RCX_ARI_PROXY_DECODE:
    case RCX_ARI_APM_ENCODE: case RCX_ARI_APM_DECODE:            // scratch = stretch table + gate bins, from the caller
        launch_serial(0, codec, k, variant, (uint32_t)variant);      // the binary model's rate rides in `variant` here
        return 0;
    case RCX_INFLATE: launch_inflate(0, k, false, variant); return 0;
    case RCX_ZLIB_DECODE: launch_inflate(0, k, true, variant); return 0;
    case RCX_ADLER32: launch_adler32(0, k); return 0;
    case RCX_CRC32: launch_crc32(0, k); return 0;
    case RCX_GZIP_DECODE: launch_gzip_decode(0, k, variant); return 0;
    case RCX_BWT_INVERSE: case RCX_BWT_INVERSE_MINIMAL: { std::string err; int st = 0; return launch_bwt_inverse(st, k, variant, err, codec == RCX_BWT_INVERSE_MINIMAL); }
    case RCX_BWT_FORWARD: case RCX_BWT_SUFFIXES: { std::string err; int st = 0; const int rc = launch_bwt_forward(st, k, variant, err, codec == RCX_BWT_SUFFIXES); if (rc) fprintf(stderr, "wavesim: %s\n", err.c_str()); return rc; }
    case RCX_BWT_INVERSION_TABLE: { int st = 0; return launch_bwt_inversion_table(st, k); }
    default:
        return -1;
    }
}

extern "C" unsigned long long* sim_stats() { return ws::g_stat; }
