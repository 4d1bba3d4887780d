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

// -------------------------------------------------------------------------------------------------
// Windowed probe (the default).  The chain above pays three dependent global round trips per probed position
// (input word, table entry, candidate word).  Between two events the positions the reference visits are known in
// advance -- pos, pos+step, pos+2*step, ... -- so 64 of them are probed at once, one per lane: every lane loads its
// word, its table entry and its candidate's word, and the first lane at which ANYTHING but a plain miss happens
// (a hit, the end of the input, the `limit` test firing, or a lane whose table entry an earlier lane of the same
// window rewrites) ends the window.  The lanes before it are plain misses and commit their table writes together;
// the event position is then handled exactly as the serial loop does, with its three values taken from the lane
// (or reloaded, in the rewritten-entry case).  Same-window rewrites are found through a 1024-slot LDS board: each lane
// posts min(lane) under its hash's low bits and is `certain` only if it reads its own number back.  A window starts W0 lanes wide -- a probe is a random
// HBM access, and on compressible data a hit ends it within a few positions -- widens (16, 64) while nothing happens, and
// shrinks to one lane (the serial probe, no board) after an event at its very first position.
// Every table entry, every emitted byte equals the serial kernel's (and the reference's).
template <uint32_t W0>
__global__ __launch_bounds__(64) void k_lz4_encode_w(rcx_kargs a, uint32_t block0)
{
    __shared__ uint32_t s_slot[1024];
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
    for (uint32_t i = lane; i < 1024u; i += 64) s_slot[i] = 0xffffffffu;
    rcx_wave_sync();
    uint32_t pos = 0, anchor = 0, dest_pos = 0, step = 1, limit = 128;
    uint32_t W = W0;                                             // lanes probing in this window: every probe is a random table access
    for (;;) {
        // ---- one window: lane i < W probes pos + i*step
        const uint32_t pi = pos + lane * step;
        const bool inw = lane < W;
        const bool live = inw && (uint64_t)pi + 12u <= (uint64_t)input_len;      // :243
        uint32_t seq = 0, hash = 0, r = 0;
        bool hit = false, unc = false;
        if (live) {
            seq = *(const rcx_u32_u*)(in + pi);                  // :250
            hash = (seq * 2654435761u) >> 15;                    // :251
            r = table[hash] + LZ4E_UNINIT;                       // :252
        }
        if (W > 1u) {                                            // the LDS board (a one-lane window has nobody before it)
            if (live) atomicMin(&s_slot[hash & 1023u], (uint32_t)lane);
            rcx_wave_sync();
            if (live) unc = s_slot[hash & 1023u] != lane;
            rcx_wave_sync();
            if (live) s_slot[hash & 1023u] = 0xffffffffu;
        }
        if (live && ((pi - r) >> 16) == 0) hit = seq == *(const rcx_u32_u*)(in + r);   // :255
        const bool cond = (pi - anchor) > limit;                 // :256 (only looked at on a miss; ending the window on it is enough)
        const unsigned long long evm = __ballot(inw && (!live || hit || unc || cond));
        const uint32_t e = evm ? (uint32_t)__ffsll(evm) - 1u : 64u;
        if (inw && lane < e) table[hash] = pi - LZ4E_UNINIT;     // :253 for the plain misses
        rcx_wave_sync();
        if (e == 64u) { pos += W * step; W = W < 4u ? 4u : W < 16u ? 16u : 64u; continue; }   // nothing happened: widen
        pos += e * step;
        if (pos + 12 > input_len) {                              // :243-248
            dest_pos = lz4e_write_literals(in, out, dest_pos, input_len - anchor, 0, anchor, lane);
            break;
        }
        // ---- the event position, as the serial loop handles it
        uint32_t seq_e, hash_e, r_e; bool hit_e;
        const bool unc_e = (__ballot(unc) >> e) & 1ull;
        W = (e == 0u || unc_e) ? 1u : e == 1u ? 2u : e == 2u ? 4u : W0;   // about twice the distance the last event was found at
        if (unc_e) {                                             // an earlier lane of this window may have rewritten the entry
            seq_e = __builtin_amdgcn_readfirstlane(*(const rcx_u32_u*)(in + pos));
            hash_e = (seq_e * 2654435761u) >> 15;
            r_e = __builtin_amdgcn_readfirstlane(table[hash_e]) + LZ4E_UNINIT;
            hit_e = ((pos - r_e) >> 16) == 0;
            if (hit_e) hit_e = seq_e == (uint32_t)__builtin_amdgcn_readfirstlane(*(const rcx_u32_u*)(in + r_e));
        } else {
            seq_e = __builtin_amdgcn_readlane(seq, e); hash_e = __builtin_amdgcn_readlane(hash, e);
            r_e = __builtin_amdgcn_readlane(r, e); hit_e = (__ballot(hit) >> e) & 1ull;
        }
        rcx_wave_sync();
        if (lane == 0) table[hash_e] = pos - LZ4E_UNINIT;        // :253
        rcx_wave_sync();
        if (!hit_e) {
            if (pos - anchor > limit) { limit <<= 1; step += 1 + (step >> 2); }   // :256-259
            pos += step;
            continue;
        }
        if (step > 1) {                                          // :264-269
            if (lane == 0) table[hash_e] = r_e - LZ4E_UNINIT;
            rcx_wave_sync();
            pos -= step - 1;
            step = 1;
            continue;
        }
        limit = 128;                                             // :271
        uint32_t r2 = r_e;
        const uint32_t ln = pos - anchor, back = pos - r2, old_anchor = anchor;
        pos += 4; r2 += 4; anchor = pos;                         // :277-279
        const uint32_t stop = input_len - 5;
        for (;;) {                                               // :281-284 match extension, 64 bytes per step
            const uint32_t p = pos + lane;
            const bool differ = (p >= stop) || (in[p] != in[r2 + lane]);
            const unsigned long long m = __ballot(differ);
            if (m) { const uint32_t k = (uint32_t)__ffsll(m) - 1u; pos += k; r2 += k; break; }
            pos += 64; r2 += 64;
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
