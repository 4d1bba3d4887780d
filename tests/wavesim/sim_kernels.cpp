// sim_kernels.cpp -- runs the UNMODIFIED .hip kernel sources on the wave64 simulator (TEST INFRASTRUCTURE).
// Built by tests/wavesim/build.py with:  g++ -include wavesim.h sim_kernels.cpp wavesim.cpp
#include "../../rust_compress_amd/csrc/k_lz4_decode.hip"

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
        else ws::launch(dim3(k.nblocks), dim3(64), [&] { k_lz4_decode_v3<2048, 2048, 64, 64, 1>(k); });
        return 0;
    default:
        return -1;
    }
}
