/* o_rle.c -- CPU restatement of src/rle.rs (TEST INFRASTRUCTURE, see oracle.h).
 * One-shot semantics (a single write() followed by finish()), which is what the
 * reference's known-answer tests pin (rle.rs:320-352). */
#include <string.h>
#include "oracle.h"

uint64_t o_rle_encode_bound(uint64_t n) { return n + n / 2 + 16; }       /* worst case: pairs "bb" -> 3 bytes */

/* Encoder::flush, rle.rs:96-122 */
static int rle_flush(uint8_t byte, uint64_t reps, uint8_t* out, size_t cap, size_t* o)
{
    if (reps == 1) {
        if (*o >= cap) return 0;
        out[(*o)++] = byte;
    } else if (reps > 1) {
        uint8_t buf[12];
        uint64_t reps_encode = reps - 2;
        int index = 2;
        buf[0] = byte; buf[1] = byte;
        for (;;) {
            buf[index] = (uint8_t)(reps_encode & 0x7f);
            reps_encode >>= 7;
            if (reps_encode == 0) { buf[index] |= 0x80; break; }
            index++;
        }
        if (cap - *o < (size_t)(index + 1)) return 0;
        memcpy(out + *o, buf, (size_t)index + 1);
        *o += (size_t)index + 1;
    }
    return 1;
}

/* Encoder::write :82-94 + process_byte :69-79 + finish :62-66 */
int o_rle_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len)
{
    size_t o = 0;
    *out_len = 0;
    if (n == 0) return RCX_OK;                    /* write_all(b"") never calls write(); flush with reps==0 emits nothing */
    uint8_t byte = in[0];
    uint64_t reps = 1;
    for (size_t i = 1; i < n; i++) {
        if (in[i] == byte) reps++;
        else {
            if (!rle_flush(byte, reps, out, cap, &o)) return RCX_E_OUTPUT_TOO_SMALL;
            reps = 1; byte = in[i];
        }
    }
    if (!rle_flush(byte, reps, out, cap, &o)) return RCX_E_OUTPUT_TOO_SMALL;
    *out_len = o;
    return RCX_OK;
}

/* Decoder::read_run :194-259 + read_byte :176-192, flattened to a whole-buffer decode */
int o_rle_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len)
{
    enum { CLEAN, SINGLE, RUN } state = CLEAN;
    uint8_t cur = 0, slice[9];
    unsigned byte_count = 0;
    size_t o = 0;
    *out_len = 0;
#define EMIT(b, r) do { uint64_t rr_ = (r); if (rr_ > cap - o) { *out_len = o; return RCX_E_OUTPUT_TOO_SMALL; } \
                        memset(out + o, (b), (size_t)rr_); o += (size_t)rr_; } while (0)
#define TO_RUN(reps_) do { uint64_t acc_ = 0; for (unsigned k_ = 0; k_ < 9; k_++) acc_ |= ((uint64_t)(slice[k_] & 0x7f)) << (k_ * 7); (reps_) = 2 + acc_; } while (0)
    for (size_t i = 0; i < n; i++) {
        uint8_t byte = in[i];
        if (state == CLEAN) { cur = byte; state = SINGLE; }                       /* :203-205 */
        else if (state == SINGLE) {
            if (byte == cur) { state = RUN; memset(slice, 0, 9); byte_count = 0; }   /* :207-208 RunBuilder::new */
            else { EMIT(cur, 1); cur = byte; }                                    /* :210-212 */
        } else {
            if (byte_count >= 9) { *out_len = o; return RCX_E_RLE_LONG_RUN; }     /* add_byte :151-158 */
            slice[byte_count++] = byte;
            if (byte & 0x80) {                                                    /* :218-222 */
                uint64_t reps; TO_RUN(reps);
                EMIT(cur, reps);
                state = CLEAN;
            }
        }
    }
    if (state == SINGLE) EMIT(cur, 1);                                            /* :247-256 EOF flush */
    else if (state == RUN) { uint64_t reps; TO_RUN(reps); EMIT(cur, reps); }
    *out_len = o;
    return RCX_OK;
}
