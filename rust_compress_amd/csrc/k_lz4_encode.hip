// k_lz4_encode.hip -- batched LZ4 block compression, bit-exact with the reference's greedy compressor.
//
// Replaces BlockEncoder::encode (src/lz4.rs:226-310, via encode_block :616-627).  The compressed bytes
// depend on the exact hash-table history, the skip acceleration (`step`/`limit`) and the back-track branch
// (:264-269), so the probe chain is inherently serial per block: one wave per block walks it with
// wave-uniform state (every lane holds the same pos/anchor/step; the 2^17-entry table lives in HBM
// scratch, 512 KiB per in-flight block), while the wide parts -- match extension (64 byte compares per
// step + ballot) and literal copies -- use all 64 lanes.  u32 wrapping arithmetic as in a release build.
#include "rcx_dev.h"

#define LZ4E_HASH_LOG 17u
#define LZ4E_TABLE (1u << LZ4E_HASH_LOG)
#define LZ4E_UNINIT 0x88888888u

typedef uint32_t __attribute__((aligned(1))) rcx_u32_u;

// write_literals, lz4.rs:192-224 (token, literal-length extension, literal bytes)
__device__ __forceinline__ uint32_t lz4e_write_literals(const uint8_t* in, uint8_t* out, uint32_t dest_pos,
                                                        uint32_t len, uint32_t ml_len, uint32_t pos, unsigned lane)
{
    uint32_t ln = len;
    const uint32_t code = ln > 14u ? 15u : ln;
    const uint32_t tok = (code << 4) + (ml_len > 14u ? 15u : ml_len);
    if (lane == 0) out[dest_pos] = (uint8_t)tok;
    dest_pos += 1;
    if (code == 15u) {
        ln -= 15u;
        const uint32_t n255 = ln / 255u;            // `while ln > 254 { 255; ln -= 255 }` then the remainder
        for (uint32_t i = lane; i < n255; i += 64) out[dest_pos + i] = 255;
        dest_pos += n255;
        if (lane == 0) out[dest_pos] = (uint8_t)(ln - n255 * 255u);
        dest_pos += 1;
    }
    for (uint32_t i = lane; i < len; i += 64) out[dest_pos + i] = in[pos + i];
    return dest_pos + len;
}

__global__ __launch_bounds__(64) void k_lz4_encode(rcx_kargs a, uint32_t block0)
{
    const uint32_t slot = blockIdx.x;
    const uint32_t b = block0 + slot;
    if (b >= a.nblocks) return;
    const unsigned lane = rcx_lane();
    const uint8_t* in = a.in_base + a.in_off[b];
    const uint64_t n64 = a.in_len[b];
    uint8_t* out = a.out_base + a.out_off[b];
    const uint64_t cap = a.out_cap[b];
    if (n64 > 0x7e000000ull) {                                   // compression_bound() == None, :229-230
        if (lane == 0) { a.status[b] = RCX_E_LZ4_INPUT_TOO_LARGE; a.out_len[b] = 0; if (a.in_used) a.in_used[b] = 0; }
        return;
    }
    const uint32_t input_len = (uint32_t)n64;
    if (cap < (uint64_t)input_len + input_len / 255u + 20u) {    // :233-237 grows the Vec instead
        if (lane == 0) { a.status[b] = RCX_E_OUTPUT_TOO_SMALL; a.out_len[b] = 0; if (a.in_used) a.in_used[b] = 0; }
        return;
    }
    uint32_t* table = (uint32_t*)a.scratch + (size_t)slot * LZ4E_TABLE;   // zero-filled by the host, :620
    uint32_t pos = 0, anchor = 0, dest_pos = 0, step = 1, limit = 128;
    for (;;) {
        if (pos + 12 > input_len) {                              // :243-248
            dest_pos = lz4e_write_literals(in, out, dest_pos, input_len - anchor, 0, anchor, lane);
            break;
        }
        const uint32_t seq = __builtin_amdgcn_readfirstlane(*(const rcx_u32_u*)(in + pos));   // :250
        const uint32_t hash = (seq * 2654435761u) >> 15;         // :251
        uint32_t r = __builtin_amdgcn_readfirstlane(table[hash]) + LZ4E_UNINIT;   // :252
        rcx_wave_sync();
        if (lane == 0) table[hash] = pos - LZ4E_UNINIT;          // :253
        rcx_wave_sync();
        bool miss = ((pos - r) >> 16) != 0;                      // :255
        if (!miss) miss = seq != (uint32_t)__builtin_amdgcn_readfirstlane(*(const rcx_u32_u*)(in + r));
        if (miss) {
            if (pos - anchor > limit) { limit <<= 1; step += 1 + (step >> 2); }   // :256-259
            pos += step;
            continue;
        }
        if (step > 1) {                                          // :264-269
            if (lane == 0) table[hash] = r - LZ4E_UNINIT;
            rcx_wave_sync();
            pos -= step - 1;
            step = 1;
            continue;
        }
        limit = 128;                                             // :271
        const uint32_t ln = pos - anchor, back = pos - r, old_anchor = anchor;
        pos += 4; r += 4; anchor = pos;                          // :277-279
        // :281-284 match extension, 64 bytes per step
        const uint32_t stop = input_len - 5;
        for (;;) {
            const uint32_t p = pos + lane;
            const bool differ = (p >= stop) || (in[p] != in[r + lane]);
            const unsigned long long m = __ballot(differ);
            if (m) { const uint32_t k = (uint32_t)__ffsll(m) - 1u; pos += k; r += k; break; }
            pos += 64; r += 64;
        }
        uint32_t ml_len = pos - anchor;                          // :286
        dest_pos = lz4e_write_literals(in, out, dest_pos, ln, ml_len, old_anchor, lane);   // :288
        if (lane == 0) { out[dest_pos] = (uint8_t)back; out[dest_pos + 1] = (uint8_t)(back >> 8); }   // :289-291
        dest_pos += 2;
        if (ml_len > 14u) {                                      // :293-304
            ml_len -= 15u;
            const uint32_t n255 = ml_len / 255u;
            for (uint32_t i = lane; i < n255; i += 64) out[dest_pos + i] = 255;
            dest_pos += n255;
            if (lane == 0) out[dest_pos] = (uint8_t)(ml_len - n255 * 255u);
            dest_pos += 1;
        }
        anchor = pos;                                            // :306
    }
    if (lane == 0) {
        a.status[b] = RCX_OK;
        a.out_len[b] = dest_pos;
        if (a.in_used) a.in_used[b] = input_len;
    }
}
