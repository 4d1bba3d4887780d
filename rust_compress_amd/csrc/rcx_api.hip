// rcx_api.hip -- the C-ABI of include/rcx.h over the gfx950 kernels.  Host side: descriptor staging,
// HBM staging for host-memory batches, kernel dispatch on the ctx stream.  There is no CPU code path:
// every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

#include "rcx_tu.h"

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = n + n / 8 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct rcx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::string err;
    int variant[RCX_CODEC_COUNT] = {0};
    uint32_t param[RCX_CODEC_COUNT] = {0};
    DevBuf d_in, d_out, d_desc, d_scratch;
    DevBuf d_apm;                        // apm stretch table + gate bins (filled on first use)
    uint8_t* h_desc = nullptr; size_t h_desc_cap = 0;      // page-locked: the descriptors' way in and the results' way out are small copies the call waits for
    hipStream_t copy_stream = nullptr;   // the host-memory LZ4 decode: compressed ranges on their way in under the launch that decodes them
    std::vector<hipEvent_t> piece_ev;
    bool gate_bad = false;               // a gated launch ran into its time limit (the copies did not run beside it): one copy in front of the launch for the next GATE_RETRY calls, then ranges are tried again
    uint32_t gate_bad_calls = 0;         // calls left before the next try (rcx_ctx_set_param(ctx, codec, 0) of either decoder clears it at once)
    DevBuf d_gate; uint32_t* h_gate = nullptr; uint32_t gate_seq = 0;      // "range r has arrived" words (device; their page-locked source)
};

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return e_ == hipErrorOutOfMemory ? RCX_RC_NO_MEMORY : RCX_RC_HIP_ERROR;           \
        }                                                                                     \
    } while (0)

extern "C" int rcx_version(void) { return 1; }

extern "C" int rcx_ctx_create(int device_id, rcx_ctx** out)
{
    if (!out) return RCX_RC_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RCX_RC_NO_DEVICE;
    rcx_ctx* c = new rcx_ctx();
    if (device_id < 0) { if (hipGetDevice(&c->device) != hipSuccess) { delete c; return RCX_RC_NO_DEVICE; } }
    else { if (device_id >= ndev || hipSetDevice(device_id) != hipSuccess) { delete c; return RCX_RC_NO_DEVICE; } c->device = device_id; }
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return RCX_RC_HIP_ERROR; }
    c->stream = c->own_stream;
    *out = c;
    return RCX_RC_OK;
}

extern "C" void rcx_ctx_destroy(rcx_ctx* c)
{
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    c->d_in.release(); c->d_out.release(); c->d_desc.release(); c->d_scratch.release(); c->d_apm.release();
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (hipEvent_t e : c->piece_ev) (void)hipEventDestroy(e);
    c->d_gate.release();
    if (c->h_gate) (void)hipHostFree(c->h_gate);
    if (c->h_desc) (void)hipHostFree(c->h_desc);
    delete c;
}

extern "C" int rcx_ctx_set_stream(rcx_ctx* c, void* s)
{
    if (!c) return RCX_RC_BAD_ARG;
    c->stream = (hipStream_t)s;          // NULL = the HIP null (legacy default) stream
    return RCX_RC_OK;
}

extern "C" int rcx_ctx_set_variant(rcx_ctx* c, int codec, int variant)
{
    if (!c || codec < 0 || codec >= RCX_CODEC_COUNT) return RCX_RC_BAD_ARG;
    c->variant[codec] = variant;
    return RCX_RC_OK;
}

extern "C" int rcx_ctx_set_param(rcx_ctx* c, int codec, uint32_t value)
{
    if (!c || codec < 0 || codec >= RCX_CODEC_COUNT) return RCX_RC_BAD_ARG;
    c->param[codec] = value;
    if (codec == RCX_LZ4_DECODE || codec == RCX_INFLATE || codec == RCX_ZLIB_DECODE) { c->gate_bad = false; c->gate_bad_calls = 0; }   // (setting a decoder's host-path knobs also ends the back-off a late gate started)
    return RCX_RC_OK;
}

extern "C" const char* rcx_last_error(const rcx_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

extern "C" const char* rcx_status_string(int s)
{
    switch (s) {
    case RCX_OK: return "ok";
    case RCX_E_EOF: return "unexpected end of file";
    case RCX_E_OUTPUT_TOO_SMALL: return "output buffer too small";
    case RCX_E_MALFORMED: return "malformed input (the reference panics here)";
    case RCX_E_HUFFMAN_TREE_TOO_LARGE: return "huffman tree too large";
    case RCX_E_INVALID_BLOCK_CODE: return "invalid block code";
    case RCX_E_INVALID_HUFFMAN_HEADER_SYMBOL: return "invalid huffman header symbol";
    case RCX_E_INVALID_HUFFMAN_TREE: return "invalid huffman tree";
    case RCX_E_INVALID_HUFFMAN_TREE_HEADER: return "invalid huffman tree header";
    case RCX_E_INVALID_HUFFMAN_CODE: return "invalid huffman code";
    case RCX_E_INVALID_STATIC_SIZE: return "invalid static size";
    case RCX_E_NOT_ENOUGH_BITS: return "not enough bits";
    case RCX_E_ZLIB_FORMAT: return "unsupported zlib stream format";
    case RCX_E_ZLIB_WINDOW: return "unsupported zlib window size";
    case RCX_E_ZLIB_DICT: return "unsupported initial dictionary in the output stream";
    case RCX_E_ZLIB_HEADER_CHECKSUM: return "invalid zlib header checksum";
    case RCX_E_ZLIB_CHECKSUM: return "invalid checksum on zlib stream";
    case RCX_E_RLE_LONG_RUN: return "Overly long run";
    case RCX_E_LZ4_MAGIC: return "";
    case RCX_E_LZ4_VERSION: return "";
    case RCX_E_LZ4_INPUT_TOO_LARGE: return "input too large";
    case RCX_E_BWT_BLOCK_TOO_LARGE: return "bwt block of 2^28 bytes or more";
    case RCX_E_GZIP_MAGIC: return "not a gzip member";
    case RCX_E_GZIP_METHOD: return "unsupported gzip compression method";
    case RCX_E_GZIP_FLAGS: return "reserved gzip flags set";
    case RCX_E_GZIP_CRC: return "invalid CRC-32 on gzip member";
    case RCX_E_GZIP_ISIZE: return "invalid length on gzip member";
    default: return "unknown status";
    }
}

extern "C" uint64_t rcx_lz4_compression_bound(uint64_t n) { return n > 0x7e000000ull ? 0 : n + n / 255 + 16 + 4; }
extern "C" uint64_t rcx_ari_byte_encode_bound(uint64_t n) { return 2 * n + 16; }
extern "C" uint64_t rcx_rle_encode_bound(uint64_t n) { return n + n / 2 + 16; }

// ---- scratch requirements ---------------------------------------------------------------------
extern "C" uint64_t rcx_scratch_bytes(int codec, uint32_t nblocks, uint64_t max_block)
{
    switch (codec) {
    case RCX_LZ4_ENCODE: return rcx_tu_lz4_encode_scratch(nblocks);
    case RCX_BWT_FORWARD: return rcx_tu_bwt_forward_scratch(nblocks, max_block);
    case RCX_BWT_SUFFIXES: return rcx_tu_bwt_forward_scratch(nblocks, max_block);          // max_block: the longest INPUT block (the slot is 4n)
    case RCX_DC_ENCODE: return rcx_tu_dc_encode_scratch(nblocks, max_block);   // (optional: without it the wave-per-block kernel encodes every block; 0 when no block is long enough for the chunk path)
    case RCX_BWT_INVERSE: case RCX_BWT_INVERSE_MINIMAL: return rcx_tu_bwt_inverse_scratch(nblocks, max_block);
    case RCX_INFLATE: case RCX_ZLIB_DECODE: return rcx_tu_inflate_scratch(nblocks);
    case RCX_GZIP_DECODE: return rcx_tu_gzip_scratch(nblocks) + rcx_tu_inflate_scratch(nblocks) + 512;   // + the carve's alignment slack
    default: return 0;
    }
}

// ---- kernel dispatch (the kernels and their launch code live in the tu_*.hip translation units) ------------------
// param_over >= 0 replaces the context's codec parameter for this one launch (the *_ctx_batch entry points' `withctx`)
static int launch_codec(rcx_ctx* c, int codec, rcx_kargs& k, int param_over = -1)
{
    hipStream_t s = c->stream;
    const uint32_t n = k.nblocks;
    if (n == 0) return RCX_RC_OK;
    const int v = c->variant[codec];
    switch (codec) {
    case RCX_LZ4_DECODE: {
        int rc = rcx_tu_lz4_decode(s, k, v, c->err);
        if (rc) return rc;
        break; }
    case RCX_LZ4_ENCODE: {
        int rc = rcx_tu_lz4_encode(s, k, v, c->err);
        if (rc) return rc;
        break; }
    case RCX_INFLATE:
    case RCX_ZLIB_DECODE:
        rcx_tu_inflate(s, k, codec == RCX_ZLIB_DECODE, v);
        break;
    case RCX_ADLER32:
        rcx_tu_adler32(s, k);
        break;
    case RCX_CRC32:
        rcx_tu_crc32(s, k);
        break;
    case RCX_GZIP_DECODE:
        if (k.scratch_bytes < rcx_tu_gzip_scratch(n)) { c->err = "gzip decode: scratch too small"; return RCX_RC_BAD_ARG; }
        rcx_tu_gzip_decode(s, k, v);
        break;
    case RCX_BWT_FORWARD: case RCX_BWT_SUFFIXES: {
        int rc = rcx_tu_bwt_forward(s, k, v, c->err, codec == RCX_BWT_SUFFIXES);
        if (rc) return rc;
        break; }
    case RCX_BWT_INVERSION_TABLE: {
        int rc = rcx_tu_bwt_inversion_table(s, k);
        if (rc) return rc;
        break; }
    case RCX_BWT_INVERSE: case RCX_BWT_INVERSE_MINIMAL: {
        int rc = rcx_tu_bwt_inverse(s, k, v, c->err, codec == RCX_BWT_INVERSE_MINIMAL);
        if (rc) return rc;
        break; }
    case RCX_ARI_APM_ENCODE: case RCX_ARI_APM_DECODE: {
        if (!c->d_apm.p) {                                       // apm.rs:53-59, 69-75, 144-154 through this host's libm
            std::vector<uint16_t> h(4096 + 32, 0);
            for (uint32_t fp = 0; fp < 4096; fp++) {
                const float p = (float)fp / 4096.0f;
                const float w = logf(p / (1.0f - p)) * 2048.0f;
                h[fp] = (w > -32769.0f && w < 32768.0f) ? (uint16_t)(int16_t)w : (uint16_t)0x8000;
            }
            for (int i = 0; i < 17; i++) {
                const float rp = (float)i / 8.0f - 1.0f;
                const int16_t wp = (int16_t)(rp * 2048.0f);
                const float pr = 1.0f / (1.0f + expf(-((float)wp / 2048.0f)));
                h[4096 + i] = (uint16_t)(pr * 4096.0f);
            }
            HIPCHK(c, c->d_apm.reserve(h.size() * 2));
            HIPCHK(c, hipMemcpy(c->d_apm.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        k.scratch = c->d_apm.p; k.scratch_bytes = (4096 + 32) * 2;
        rcx_tu_serial(s, codec, k, v, 0);
        break; }
    case RCX_ARI_BINARY_ENCODE: case RCX_ARI_BINARY_DECODE:
        if (c->param[codec] < 1 || c->param[codec] > 31) { c->err = "ari binary: rate must be 1..31"; return RCX_RC_BAD_ARG; }
        [[fallthrough]];
    case RCX_MTF_ENCODE: case RCX_MTF_DECODE: case RCX_DC_ENCODE: case RCX_DC_DECODE:
    case RCX_ARI_PROXY_ENCODE: case RCX_ARI_PROXY_DECODE:
    case RCX_ARI_BYTE_ENCODE: case RCX_ARI_BYTE_DECODE: case RCX_RLE_ENCODE: case RCX_RLE_DECODE:
        rcx_tu_serial(s, codec, k, v, param_over >= 0 ? (uint32_t)param_over : c->param[codec]);
        break;
    default:
        c->err = "unknown codec";
        return RCX_RC_BAD_ARG;
    }
    HIPCHK(c, hipGetLastError());
    return RCX_RC_OK;
}

extern "C" int rcx_launch_dev(rcx_ctx* c, int codec, const rcx_dev_batch* b, void* scratch, uint64_t scratch_bytes)
{
    if (!c || !b || codec < 0 || codec >= RCX_CODEC_COUNT) return RCX_RC_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    rcx_kargs k;
    k.in_base = b->in_base; k.in_off = b->in_off; k.in_len = b->in_len;
    k.out_base = b->out_base; k.out_off = b->out_off; k.out_cap = b->out_cap;
    k.out_len = b->out_len; k.in_used = b->in_used; k.status = b->status; k.aux = b->aux;
    k.n_out = nullptr; k.scratch = scratch; k.scratch_bytes = scratch_bytes; k.nblocks = b->nblocks; k.out_mirror = nullptr; k.gate = nullptr; k.gate_all = 0;
    if (codec == RCX_DC_DECODE) { c->err = "dc decode needs n_out: use rcx_dc_decode_batch"; return RCX_RC_BAD_ARG; }
    return launch_codec(c, codec, k);
}

// ---- host-descriptor batch path -------------------------------------------------------------------
// Descriptor block layout in HBM (all 8-byte aligned):
//   in_off[n] in_len[n] out_off[n] out_cap[n] n_out[n] | out_len[n] in_used[n] | status[n] aux[n]
static const uint32_t GATE_RETRY = 64;      // calls that go back to one copy in front of the launch after a gate ran into its limit (one descheduling of the calling thread is enough for that)
static int run_batch(rcx_ctx* c, int codec, const rcx_batch* b, const uint32_t* aux_in, uint32_t* aux_out,
                     const uint64_t* n_out, bool needs_out, int param_over = -1)
{
    if (!c) return RCX_RC_BAD_ARG;
    if (!b || (b->nblocks && (!b->in_off || !b->in_len || !b->status))) { c->err = "null descriptor array"; return RCX_RC_BAD_ARG; }
    if (needs_out && b->nblocks && (!b->out_off || !b->out_cap || !b->out_len)) { c->err = "null output descriptor"; return RCX_RC_BAD_ARG; }
    const uint32_t n = b->nblocks;
    if (n == 0) return RCX_RC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    uint64_t in_span = 0, out_span = 0, max_block = 0, max_in = 0;
    for (uint32_t i = 0; i < n; i++) {
        // the kernels index a block with 32-bit offsets: a block of 4 GiB or more (or a range that wraps) is a caller error,
        // not something to decode a prefix of
        if (b->in_len[i] >> 32 || b->in_off[i] + b->in_len[i] < b->in_off[i] ||
            (needs_out && (b->out_cap[i] >> 32 || b->out_off[i] + b->out_cap[i] < b->out_off[i]))) {
            c->err = "block " + std::to_string(i) + ": lengths of 4 GiB or more are not supported (per-block limit 2^32 - 1 bytes)";
            return RCX_RC_BAD_ARG;
        }
        const uint64_t e = b->in_off[i] + b->in_len[i];
        if (e > in_span) in_span = e;
        if (b->in_len[i] > max_block) max_block = b->in_len[i];
        if (b->in_len[i] > max_in) max_in = b->in_len[i];
        if (needs_out) {
            const uint64_t o = b->out_off[i] + b->out_cap[i];
            if (o > out_span) out_span = o;
            if (b->out_cap[i] > max_block) max_block = b->out_cap[i];
        }
    }
    if ((in_span && !b->in_base) || (out_span && !b->out_base)) { c->err = "null data pointer"; return RCX_RC_BAD_ARG; }
    const uint8_t* d_in = b->in_base;
    uint8_t* d_out = b->out_base;
    // LZ4 decode from host memory into a PAGE-LOCKED output buffer (hipHostMalloc / hipHostRegister: the device can address it):
    // the decoder stores every byte that leaves its window a second time straight into that buffer (k_lz4_decode_v4.hip, MIRROR),
    // so the decoded bytes cross PCIe while the launch runs and no device-to-host copy follows it; and the compressed bytes travel in
    // as block ranges on a copy stream while the launch already decodes the ranges before them (below).  One copy
    // each way (what every other codec and a pageable buffer get) costs in + kernel + out = 1.8 + 0.5 + 4.9 ms on the headline
    // workload; this is the outbound 4.9 ms and little else.  rcx_ctx_set_param(ctx, RCX_LZ4_DECODE, 1) keeps the plain copies.
    uint8_t* mirror = nullptr;
    uint32_t pieces = 1;
    const bool inf_mirror = (codec == RCX_INFLATE || codec == RCX_ZLIB_DECODE || codec == RCX_GZIP_DECODE) && !(c->param[codec] & 1u);    // (the inflate front end drains through the same window; the
                                                                                                              //  streams its first pass hands back are copied out behind the second)
    if (b->mem == RCX_MEM_HOST && (codec == RCX_LZ4_DECODE || (inf_mirror && rcx_tu_inflate_mirrors(n, 0))) && out_span && c->variant[codec] == 0 && !(c->param[codec] & 1u)) {
        hipPointerAttribute_t at;
        // (the WHOLE span must be page-locked and mapped as one range: a buffer registered only in part would take the kernel's stores
        //  into unmapped addresses -- the last byte's attributes must continue the first's)
        auto covered = [](const void* base, uint64_t span, hipPointerAttribute_t& first) {
            hipPointerAttribute_t last;
            if (hipPointerGetAttributes(&first, base) != hipSuccess || first.type != hipMemoryTypeHost || !first.devicePointer) return false;
            if (span <= 1) return true;
            if (hipPointerGetAttributes(&last, (const uint8_t*)base + span - 1) != hipSuccess || last.type != hipMemoryTypeHost || !last.devicePointer) return false;
            return (const uint8_t*)last.devicePointer == (const uint8_t*)first.devicePointer + (span - 1);
        };
        if (c->gate_bad && c->gate_bad_calls && --c->gate_bad_calls == 0) c->gate_bad = false;
        if (covered(b->out_base, out_span, at)) {
            mirror = (uint8_t*)at.devicePointer;
            // (ranges only from page-locked INPUT: a pageable buffer is staged piece by piece, by copies that may need the compute
            // units the waiting blocks would hold)
            hipPointerAttribute_t ai;
            // (gzip members: their headers are parsed by a kernel of its own in front of the decoder, which wants every member there)
            if (codec != RCX_GZIP_DECODE && in_span && !c->gate_bad && covered(b->in_base, in_span, ai)) {
                pieces = ((c->param[codec] >> 8) & 255u) ? ((c->param[codec] >> 8) & 255u) : 16u;
                if (pieces > 16u) pieces = 16u;             // (the most; see below)
                if (pieces > n / 128u) pieces = n / 128u ? n / 128u : 1u;
            } else (void)hipGetLastError();
        } else (void)hipGetLastError();
    }
    size_t out_shift = 0;
    if (b->mem == RCX_MEM_HOST) {
        out_shift = mirror ? (size_t)((uintptr_t)mirror & 255u) : 0;           // the copy in HBM and the host buffer: the same alignment
        HIPCHK(c, c->d_in.reserve(in_span + 64));
        HIPCHK(c, c->d_out.reserve(out_span + out_shift + 64));
        if (in_span && pieces <= 1) HIPCHK(c, hipMemcpyAsync(c->d_in.p, b->in_base, in_span, hipMemcpyHostToDevice, s));
        d_in = (const uint8_t*)c->d_in.p;
        d_out = (uint8_t*)c->d_out.p + out_shift;
    } else if (b->mem != RCX_MEM_DEVICE) { c->err = "bad mem kind"; return RCX_RC_BAD_ARG; }

    const size_t N = n;
    const size_t in_words = 5 * N;                 // u64
    const size_t res_words = 2 * N;                // u64
    const size_t desc_bytes = (in_words + res_words) * 8 + 2 * N * 4 + 64;
    HIPCHK(c, c->d_desc.reserve(desc_bytes));
    if (desc_bytes > c->h_desc_cap) {
        if (c->h_desc) (void)hipHostFree(c->h_desc);
        c->h_desc = nullptr; c->h_desc_cap = 0;
        HIPCHK(c, hipHostMalloc((void**)&c->h_desc, desc_bytes + desc_bytes / 4, hipHostMallocDefault));
        c->h_desc_cap = desc_bytes + desc_bytes / 4;
    }
    uint64_t* h64 = (uint64_t*)c->h_desc;
    memcpy(h64 + 0 * N, b->in_off, N * 8);
    memcpy(h64 + 1 * N, b->in_len, N * 8);
    if (needs_out) { memcpy(h64 + 2 * N, b->out_off, N * 8); memcpy(h64 + 3 * N, b->out_cap, N * 8); }
    else memset(h64 + 2 * N, 0, 2 * N * 8);
    if (n_out) memcpy(h64 + 4 * N, n_out, N * 8); else memset(h64 + 4 * N, 0, N * 8);
    int32_t* h_status = (int32_t*)(h64 + 7 * N);
    uint32_t* h_aux = (uint32_t*)(h_status + N);
    for (size_t i = 0; i < N; i++) h_status[i] = RCX_E_MALFORMED;
    if (aux_in) memcpy(h_aux, aux_in, N * 4); else memset(h_aux, 0, N * 4);
    memset(h64 + 5 * N, 0, 2 * N * 8);
    HIPCHK(c, hipMemcpyAsync(c->d_desc.p, c->h_desc, (7 * N) * 8 + 2 * N * 4, hipMemcpyHostToDevice, s));
    uint64_t* d64 = (uint64_t*)c->d_desc.p;

    rcx_kargs k;
    k.in_base = d_in; k.in_off = d64; k.in_len = d64 + N;
    k.out_base = d_out; k.out_off = d64 + 2 * N; k.out_cap = d64 + 3 * N;
    k.n_out = d64 + 4 * N;
    k.out_len = d64 + 5 * N; k.in_used = d64 + 6 * N;
    k.status = (int32_t*)(d64 + 7 * N); k.aux = (uint32_t*)(k.status + N);
    k.nblocks = n;
    uint64_t sb = rcx_scratch_bytes(codec, n, codec == RCX_BWT_SUFFIXES ? max_in : max_block);
    if (codec == RCX_DC_ENCODE && param_over > 0) sb = 0;          // withctx: the wave-per-block kernel encodes, no chunk states
    if (codec == RCX_DC_ENCODE && sb && c->d_scratch.reserve(sb + 64) != hipSuccess) {
        // the chunk states are optional (37 KiB a block): a batch too large for them falls back to the wave-per-block kernel
        (void)hipGetLastError();
        k.scratch = nullptr; k.scratch_bytes = 0;
    } else {
        HIPCHK(c, c->d_scratch.reserve(sb + 64));
        k.scratch = c->d_scratch.p; k.scratch_bytes = c->d_scratch.cap;
        if (codec == RCX_DC_ENCODE && !sb) { k.scratch = nullptr; k.scratch_bytes = 0; }
    }
    k.out_mirror = mirror; k.gate = nullptr; k.gate_host = nullptr; k.gate_seq = 0; k.gate_ticks = 0; k.gate_all = 0;
    for (int i = 0; i < 15; i++) k.gate_bnd[i] = 0xffffffffu;
    bool gated = false;
    if (pieces > 1) {
        // ONE launch, the input in ranges: the blocks of a range (the first one small: a sixty-fourth of the blocks) start when this
        // thread has seen the range's copy complete and said so in a page-locked word (k_lz4_decode_v8, `gate`).  A launch per range
        // was built first and measured: each one ends with the link drained and begins with nothing to send, 5.9 ms for three
        // growing ranges, 6.6 for eight equal ones, against 7.0 for one launch behind one copy and 5.6 for this.
        // A range's compressed bytes are the span from its lowest to its highest input byte, widened to whole 256-byte lines of the
        // staging buffer (a line two ranges share is complete the first time anybody reads it; what the widening copies early are the
        // caller's own bytes).  Blocks that do not lie in index order make the spans overlap: more than a quarter of the input twice and
        // the call goes back to one copy in front of the launch.  A block whose input does not arrive in gate_ticks gives up with
        // RCX_ST_GATE and is decoded by a second launch below (the copies cannot be held up by the waiting blocks as long as a copy
        // engine moves them; a copy done by a kernel could be, and then this is what ends the wait).
        std::vector<uint32_t> bnd(1, 0);
        const uint32_t fdiv = ((c->param[codec] >> 16) & 255u) ? ((c->param[codec] >> 16) & 255u) : 64u;       // (tuning)
        const uint32_t first = n / fdiv > 64u ? n / fdiv : 64u;
        for (uint32_t pc = 1; pc < pieces; pc++) {
            const uint32_t at = first + (uint32_t)((uint64_t)(n - first) * (pc - 1) / (pieces - 1));
            if (at > bnd.back() && at < n) bnd.push_back(at);
        }
        bnd.push_back(n);
        pieces = (uint32_t)bnd.size() - 1;
        std::vector<uint64_t> lo(pieces, ~0ull), hi(pieces, 0);
        uint64_t moved = 0;
        for (uint32_t pc = 0; pc < pieces; pc++) {
            for (uint32_t i = bnd[pc]; i < bnd[pc + 1]; i++) {
                if (!b->in_len[i]) continue;
                if (b->in_off[i] < lo[pc]) lo[pc] = b->in_off[i];
                if (b->in_off[i] + b->in_len[i] > hi[pc]) hi[pc] = b->in_off[i] + b->in_len[i];
            }
            if (hi[pc] > lo[pc]) {
                lo[pc] &= ~255ull;
                hi[pc] = (hi[pc] + 255ull) & ~255ull; if (hi[pc] > in_span) hi[pc] = in_span;
                moved += hi[pc] - lo[pc];
            }
        }
        if (pieces <= 1 || moved > in_span + in_span / 4) {
            pieces = 1;
            if (in_span) HIPCHK(c, hipMemcpyAsync(c->d_in.p, b->in_base, in_span, hipMemcpyHostToDevice, s));
        } else {
            if (!c->copy_stream) {
                // a stream of ANOTHER priority than the launch's: HIP hands its few hardware queues to the streams of one priority in
                // turn, and a copy stream that shares the launch's queue stands behind the launch it is meant to feed (every gate ran into
                // its limit for one context in two: 66 ms a call).  The LOWEST priority: it carries copy-engine work only, and the
                // high-priority queues stay with whoever uses them for kernels (pipeline.PipelineLanes keeps its two lanes apart that way)
                int least = 0, greatest = 0;
                HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
                HIPCHK(c, hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, least));
            }
            while (c->piece_ev.size() < pieces) { hipEvent_t e; HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->piece_ev.push_back(e); }
            const uint32_t seq = ++c->gate_seq ? c->gate_seq : ++c->gate_seq;       // (never 0)
            if (!c->h_gate) HIPCHK(c, hipHostMalloc((void**)&c->h_gate, 64, hipHostMallocDefault));
            // the gate words hold anything BUT this call's number before the launch: recycled page-locked or device memory is not zero, and a
            // stale word that happened to equal `seq` would let a range's blocks read input that has not arrived
            for (int w = 0; w < 16; w++) __atomic_store_n((volatile uint32_t*)(c->h_gate + w), seq - 1u, __ATOMIC_RELAXED);
            HIPCHK(c, c->d_gate.reserve(64));
            HIPCHK(c, hipMemsetAsync(c->d_gate.p, 0, 64, s));                        // (0 is never a call's number; in front of the launch on its stream)
            {
                hipPointerAttribute_t ga;
                HIPCHK(c, hipPointerGetAttributes(&ga, c->h_gate));
                k.gate_host = (const uint32_t*)ga.devicePointer;
            }
            k.gate = (uint32_t*)c->d_gate.p; k.gate_seq = seq;
            for (uint32_t pc = 1; pc < pieces; pc++) k.gate_bnd[pc - 1] = bnd[pc];
            const uint64_t ticks = 1000000ull + in_span / 50ull;                   // 10 ms + the input at 5 GB/s (100 MHz ticks)
            k.gate_ticks = ticks > 0xffffffffull ? 0xffffffffu : (uint32_t)ticks;
            // (range 0 waits at a gate like the others: the launch is on its way to the device while the first bytes are)
            k.gate_all = 1;
            // a failure behind the launch must not leave it spinning at its gates under the next call's copies: open every gate (the blocks
            // decode whatever has arrived; the call fails anyway), drain both streams, then return
            bool launched = false;
            auto bail = [&](hipError_t e, const char* what) {
                c->err = std::string(what) + ": " + hipGetErrorString(e);
                (void)hipGetLastError();
                if (launched) for (int w = 0; w < 16; w++) __atomic_store_n((volatile uint32_t*)(c->h_gate + w), seq, __ATOMIC_RELEASE);
                (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamSynchronize(s);
                return RCX_RC_HIP_ERROR;
            };
            for (uint32_t pc = 0; pc < pieces; pc++) {
                hipError_t e = hipSuccess;
                if (hi[pc] > lo[pc]) e = hipMemcpyAsync((uint8_t*)c->d_in.p + lo[pc], b->in_base + lo[pc], hi[pc] - lo[pc], hipMemcpyHostToDevice, c->copy_stream);
                if (e != hipSuccess) return bail(e, "host path: range copy");
                if ((e = hipEventRecord(c->piece_ev[pc], c->copy_stream)) != hipSuccess) return bail(e, "host path: event record");
                if (pc == 0) {
                    const int rcg = launch_codec(c, codec, k, param_over);
                    if (rcg) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamSynchronize(s); return rcg; }
                    launched = true;
                }
            }
            // this thread tells the launch what has arrived (a word copied in behind each range would be the natural signal; such a small
            // copy is done by a kernel, and a kernel does not run while every slot of the device holds a waiting block: built, measured --
            // every gate ran into its time limit)
            for (uint32_t pc = 0; pc < pieces; pc++) {
                const hipError_t e = hipEventSynchronize(c->piece_ev[pc]);
                if (e != hipSuccess) return bail(e, "host path: event wait");
                __atomic_store_n((volatile uint32_t*)(c->h_gate + pc), seq, __ATOMIC_RELEASE);
            }
            gated = true;
        }
    }
    if (!gated) {
        int rc = launch_codec(c, codec, k, param_over);
        if (rc) return rc;
    }
    if (mirror && !k.out_mirror) mirror = nullptr;                      // the launch says it did not store into the caller's buffer after all: the plain copy below
    HIPCHK(c, hipMemcpyAsync(h64 + 5 * N, d64 + 5 * N, 2 * N * 8 + 2 * N * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (gated) {
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        bool again = false;
        for (size_t i = 0; i < N && !again; i++) again = h_status[i] == (int32_t)RCX_ST_GATE;
        if (again && codec != RCX_LZ4_DECODE) {                 // (the inflate path's second pass and trailer check passed these streams by: the whole batch again, behind one copy)
            c->gate_bad = true; c->gate_bad_calls = GATE_RETRY;
            return run_batch(c, codec, b, aux_in, aux_out, n_out, needs_out, param_over);
        }
        if (again) {
            c->gate_bad = true; c->gate_bad_calls = GATE_RETRY;
            k.gate = nullptr;
            rcx_tu_lz4_decode_mirror_again(s, k);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(h64 + 5 * N, d64 + 5 * N, 2 * N * 8 + 2 * N * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
        }
    }
    if (mirror && inf_mirror && k.scratch) {
        // the streams the first pass handed back were decoded into HBM alone: a few, one copy each; many (a batch of corrupted streams), the span
        const uint8_t* marks = (const uint8_t*)k.scratch + (codec == RCX_GZIP_DECODE ? rcx_tu_gzip_marks_offset(n) : rcx_tu_inflate_marks_offset(n));
        uint32_t nfb = 0;
        HIPCHK(c, hipMemcpy(&nfb, marks, 4, hipMemcpyDeviceToHost));
        if (nfb) {
            const uint64_t* ol = h64 + 5 * N;
            if (nfb > n / 8u + 16u) {
                uint64_t used_span = 0;
                for (uint32_t i = 0; i < n; i++) {
                    const uint64_t l = ol[i] < b->out_cap[i] ? ol[i] : b->out_cap[i];
                    if (l && b->out_off[i] + l > used_span) used_span = b->out_off[i] + l;
                }
                if (used_span) HIPCHK(c, hipMemcpy(b->out_base, d_out, used_span, hipMemcpyDeviceToHost));
            } else {
                std::vector<uint8_t> fb(n);
                HIPCHK(c, hipMemcpy(fb.data(), marks + 64, n, hipMemcpyDeviceToHost));
                for (uint32_t i = 0; i < n; i++) {
                    const uint64_t l = ol[i] < b->out_cap[i] ? ol[i] : b->out_cap[i];
                    if (fb[i] && l) HIPCHK(c, hipMemcpyAsync(b->out_base + b->out_off[i], d_out + b->out_off[i], l, hipMemcpyDeviceToHost, s));
                }
                HIPCHK(c, hipStreamSynchronize(s));
            }
        }
    }
    if (b->mem == RCX_MEM_HOST && out_span && !mirror) {
        // only what was produced travels back: the span up to the last byte any block wrote, not the slots' capacity
        uint64_t used_span = 0;
        const uint64_t* ol = h64 + 5 * N;
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t l = ol[i] < b->out_cap[i] ? ol[i] : b->out_cap[i];
            if (l && b->out_off[i] + l > used_span) used_span = b->out_off[i] + l;
        }
        if (used_span) HIPCHK(c, hipMemcpy(b->out_base, d_out, used_span, hipMemcpyDeviceToHost));
    }
    if (b->out_len) memcpy(b->out_len, h64 + 5 * N, N * 8);
    if (b->in_used) memcpy(b->in_used, h64 + 6 * N, N * 8);
    memcpy(b->status, h_status, N * 4);
    if (aux_out) memcpy(aux_out, h_aux, N * 4);
    return RCX_RC_OK;
}

extern "C" int rcx_lz4_decode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_LZ4_DECODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_lz4_encode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_LZ4_ENCODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_inflate_batch(rcx_ctx* c, const rcx_batch* b, uint32_t* flags) { return run_batch(c, RCX_INFLATE, b, nullptr, flags, nullptr, true); }
extern "C" int rcx_zlib_decode_batch(rcx_ctx* c, const rcx_batch* b, uint32_t* flags) { return run_batch(c, RCX_ZLIB_DECODE, b, nullptr, flags, nullptr, true); }
extern "C" int rcx_adler32_batch(rcx_ctx* c, const rcx_batch* b, uint32_t* adler) { return run_batch(c, RCX_ADLER32, b, nullptr, adler, nullptr, false); }
extern "C" int rcx_crc32_batch(rcx_ctx* c, const rcx_batch* b, uint32_t* crc) { return run_batch(c, RCX_CRC32, b, nullptr, crc, nullptr, false); }
extern "C" int rcx_gzip_decode_batch(rcx_ctx* c, const rcx_batch* b, uint32_t* flags) { return run_batch(c, RCX_GZIP_DECODE, b, nullptr, flags, nullptr, true); }
extern "C" int rcx_bwt_forward_batch(rcx_ctx* c, const rcx_batch* b, uint32_t* origin) { return run_batch(c, RCX_BWT_FORWARD, b, nullptr, origin, nullptr, true); }
extern "C" int rcx_bwt_suffixes_batch(rcx_ctx* c, const rcx_batch* b, uint32_t* origin) { return run_batch(c, RCX_BWT_SUFFIXES, b, nullptr, origin, nullptr, true); }
extern "C" int rcx_bwt_inversion_table_batch(rcx_ctx* c, const rcx_batch* b, const uint32_t* origin) { return run_batch(c, RCX_BWT_INVERSION_TABLE, b, origin, nullptr, nullptr, true); }
extern "C" int rcx_bwt_inverse_batch(rcx_ctx* c, const rcx_batch* b, const uint32_t* origin) { return run_batch(c, RCX_BWT_INVERSE, b, origin, nullptr, nullptr, true); }
extern "C" int rcx_bwt_inverse_minimal_batch(rcx_ctx* c, const rcx_batch* b, const uint32_t* origin) { return run_batch(c, RCX_BWT_INVERSE_MINIMAL, b, origin, nullptr, nullptr, true); }
extern "C" int rcx_mtf_encode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_MTF_ENCODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_mtf_decode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_MTF_DECODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_dc_encode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_DC_ENCODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_dc_decode_batch(rcx_ctx* c, const rcx_batch* b, const uint64_t* n_out) { return run_batch(c, RCX_DC_DECODE, b, nullptr, nullptr, n_out, true); }
// the same with the coding contexts behind the payload (include/rcx.h): the kernels' `withctx` rides in the codec's parameter
extern "C" int rcx_dc_encode_ctx_batch(rcx_ctx* c, const rcx_batch* b)
{
    if (!c) return RCX_RC_BAD_ARG;
    return run_batch(c, RCX_DC_ENCODE, b, nullptr, nullptr, nullptr, true, 1);
}
extern "C" int rcx_dc_decode_ctx_batch(rcx_ctx* c, const rcx_batch* b, const uint64_t* n_out)
{
    if (!c) return RCX_RC_BAD_ARG;
    return run_batch(c, RCX_DC_DECODE, b, nullptr, nullptr, n_out, true, 1);
}
extern "C" int rcx_ari_byte_encode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_ARI_BYTE_ENCODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_ari_byte_decode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_ARI_BYTE_DECODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_ari_binary_encode_batch(rcx_ctx* c, const rcx_batch* b, uint32_t rate)
{
    if (!c) return RCX_RC_BAD_ARG;
    c->param[RCX_ARI_BINARY_ENCODE] = rate;
    return run_batch(c, RCX_ARI_BINARY_ENCODE, b, nullptr, nullptr, nullptr, true);
}
extern "C" int rcx_ari_binary_decode_batch(rcx_ctx* c, const rcx_batch* b, uint32_t rate)
{
    if (!c) return RCX_RC_BAD_ARG;
    c->param[RCX_ARI_BINARY_DECODE] = rate;
    return run_batch(c, RCX_ARI_BINARY_DECODE, b, nullptr, nullptr, nullptr, true);
}
extern "C" int rcx_ari_proxy_encode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_ARI_PROXY_ENCODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_ari_proxy_decode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_ARI_PROXY_DECODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_ari_apm_encode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_ARI_APM_ENCODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_ari_apm_decode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_ARI_APM_DECODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_rle_encode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_RLE_ENCODE, b, nullptr, nullptr, nullptr, true); }
extern "C" int rcx_rle_decode_batch(rcx_ctx* c, const rcx_batch* b) { return run_batch(c, RCX_RLE_DECODE, b, nullptr, nullptr, nullptr, true); }

// ---- more than one device -------------------------------------------------------------------------------------------------
// RCCL, loaded on first use (librcx.so itself does not link it: the library loads wherever HIP does, and a process that has torch's RCCL
// mapped already gets that one -- same SONAME).  Only what a grouped point-to-point exchange needs.
#include <dlfcn.h>
struct RcclApi {
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    static RcclApi& get()
    {
        static RcclApi a = [] {
            RcclApi r;
            void* h = nullptr;
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
            if (!h) return r;
            r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
            r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
            r.Send = (decltype(r.Send))dlsym(h, "ncclSend");
            r.Recv = (decltype(r.Recv))dlsym(h, "ncclRecv");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
            r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
            return r;
        }();
        return a;
    }
};
static const int RCCL_UINT8 = 1;                 // ncclUint8 (nccl.h: ncclInt8 = 0, ncclUint8 = 1)

struct rcx_multi {
    std::vector<rcx_ctx*> ctx;
    std::string err;
    // the device-to-device shard path (rcx_multi_scatter_dev / rcx_multi_gather_dev)
    int transport = 0;                           // 0: not chosen yet, 1: RCCL, 2: peer copies
    std::vector<RcclApi::comm_t> comm;
    std::vector<hipEvent_t> ev;                  // peer copies: one event per context
};

extern "C" int rcx_multi_create(const int* device_ids, int n, rcx_multi** out)
{
    if (!out) return RCX_RC_BAD_ARG;
    *out = nullptr;
    if (!device_ids || n <= 0) return RCX_RC_BAD_ARG;
    rcx_multi* m = new rcx_multi();
    for (int i = 0; i < n; i++) {
        rcx_ctx* c = nullptr;
        const int rc = rcx_ctx_create(device_ids[i], &c);
        if (rc != RCX_RC_OK) { for (rcx_ctx* d : m->ctx) rcx_ctx_destroy(d); delete m; return rc; }
        m->ctx.push_back(c);
    }
    *out = m;
    return RCX_RC_OK;
}
extern "C" void rcx_multi_destroy(rcx_multi* m)
{
    if (!m) return;
    for (RcclApi::comm_t q : m->comm) if (q) (void)RcclApi::get().CommDestroy(q);
    for (size_t g = 0; g < m->ev.size(); g++) if (m->ev[g]) { (void)hipSetDevice(m->ctx[g]->device); (void)hipEventDestroy(m->ev[g]); }
    for (rcx_ctx* c : m->ctx) rcx_ctx_destroy(c);
    delete m;
}
extern "C" const char* rcx_multi_transport(const rcx_multi* m) { return !m ? "" : m->transport == 1 ? "rccl" : m->transport == 2 ? "peer" : ""; }

// choose and set up the transport once per set
static int multi_transport_init(rcx_multi* m)
{
    if (m->transport) return RCX_RC_OK;
    const size_t G = m->ctx.size();
    bool distinct = true;
    for (size_t a = 0; a < G; a++) for (size_t b = a + 1; b < G; b++) if (m->ctx[a]->device == m->ctx[b]->device) distinct = false;
    const char* want = getenv("RCX_MULTI_TRANSPORT");
    const bool force_peer = want && !strcmp(want, "peer"), force_rccl = want && !strcmp(want, "rccl");
    if (!force_peer && distinct && RcclApi::get().ok) {
        std::vector<int> devs(G);
        for (size_t g = 0; g < G; g++) devs[g] = m->ctx[g]->device;
        m->comm.assign(G, nullptr);
        const int e = RcclApi::get().CommInitAll(m->comm.data(), (int)G, devs.data());
        if (e == 0) { m->transport = 1; return RCX_RC_OK; }
        m->comm.clear();
        if (force_rccl) { m->err = std::string("ncclCommInitAll: ") + (RcclApi::get().GetErrorString ? RcclApi::get().GetErrorString(e) : "failed"); return RCX_RC_HIP_ERROR; }
    } else if (force_rccl) { m->err = distinct ? "RCX_MULTI_TRANSPORT=rccl: librccl could not be loaded" : "RCX_MULTI_TRANSPORT=rccl: the set lists a device more than once (RCCL wants one rank per device)"; return RCX_RC_BAD_ARG; }
    // peer copies: every device reads / writes every other's memory where the hardware allows (xGMI within a node); where it does not,
    // hipMemcpyPeerAsync stages through the host by itself
    for (size_t a = 0; a < G; a++) {
        if (hipSetDevice(m->ctx[a]->device) != hipSuccess) { (void)hipGetLastError(); m->err = "hipSetDevice failed"; return RCX_RC_HIP_ERROR; }
        for (size_t b = 0; b < G; b++) {
            if (m->ctx[a]->device == m->ctx[b]->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, m->ctx[a]->device, m->ctx[b]->device) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(m->ctx[b]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                else (void)hipGetLastError();
            } else (void)hipGetLastError();
        }
    }
    m->ev.assign(G, nullptr);
    for (size_t g = 0; g < G; g++) {
        if (hipSetDevice(m->ctx[g]->device) != hipSuccess || hipEventCreateWithFlags(&m->ev[g], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); m->err = "cannot create an event"; return RCX_RC_HIP_ERROR; }
    }
    m->transport = 2;
    return RCX_RC_OK;
}

// ranges root -> peers (scatter) or peers -> root (gather); see include/rcx.h
static int multi_move(rcx_multi* m, int root, uint8_t* root_buf, const uint64_t* range_off, uint8_t* const* peer_buf, bool scatter)
{
    if (!m || m->ctx.empty()) return RCX_RC_BAD_ARG;
    m->err.clear();
    const int G = (int)m->ctx.size();
    if (root < 0 || root >= G || !range_off || !peer_buf) { m->err = "bad root / null range table"; return RCX_RC_BAD_ARG; }
    for (int g = 0; g < G; g++) {
        if (range_off[g + 1] < range_off[g]) { m->err = "ranges must not decrease"; return RCX_RC_BAD_ARG; }
        if (range_off[g + 1] > range_off[g] && (!root_buf || (!peer_buf[g] && g != root))) { m->err = "null buffer for a non-empty range"; return RCX_RC_BAD_ARG; }
    }
    const int rc = multi_transport_init(m);
    if (rc != RCX_RC_OK) return rc;
    rcx_ctx* R = m->ctx[(size_t)root];
    if (m->transport == 1) {
        RcclApi& N = RcclApi::get();
        int e = N.GroupStart();
        for (int g = 0; g < G && e == 0; g++) {
            const uint64_t bytes = range_off[g + 1] - range_off[g];
            if (!bytes || (g == root && !peer_buf[g])) continue;
            uint8_t* rp = root_buf + range_off[g];
            rcx_ctx* P = m->ctx[(size_t)g];
            if (scatter) { e = N.Send(rp, bytes, RCCL_UINT8, g, m->comm[(size_t)root], R->stream); if (!e) e = N.Recv(peer_buf[g], bytes, RCCL_UINT8, root, m->comm[(size_t)g], P->stream); }
            else { e = N.Send(peer_buf[g], bytes, RCCL_UINT8, root, m->comm[(size_t)g], P->stream); if (!e) e = N.Recv(rp, bytes, RCCL_UINT8, g, m->comm[(size_t)root], R->stream); }
        }
        const int e2 = N.GroupEnd();
        if (e || e2) { m->err = std::string(scatter ? "scatter" : "gather") + ": " + (N.GetErrorString ? N.GetErrorString(e ? e : e2) : "RCCL error"); return RCX_RC_HIP_ERROR; }
        return RCX_RC_OK;
    }
    // peer copies.  scatter: every receiving stream waits for what the root's stream has produced so far, then copies its range in;
    // gather: every sending stream copies its range out behind its own launches, and the root's stream waits for all of them.
#define MCHK(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) { (void)hipGetLastError(); m->err = std::string(#call) + ": " + hipGetErrorString(e_); return RCX_RC_HIP_ERROR; } } while (0)
    if (scatter) { MCHK(hipSetDevice(R->device)); MCHK(hipEventRecord(m->ev[(size_t)root], R->stream)); }
    for (int g = 0; g < G; g++) {
        const uint64_t bytes = range_off[g + 1] - range_off[g];
        if (!bytes || (g == root && !peer_buf[g])) continue;
        uint8_t* rp = root_buf + range_off[g];
        rcx_ctx* P = m->ctx[(size_t)g];
        MCHK(hipSetDevice(P->device));
        if (scatter) {
            if (g != root) MCHK(hipStreamWaitEvent(P->stream, m->ev[(size_t)root], 0));
            MCHK(hipMemcpyPeerAsync(peer_buf[g], P->device, rp, R->device, bytes, P->stream));
        } else {
            MCHK(hipMemcpyPeerAsync(rp, R->device, peer_buf[g], P->device, bytes, P->stream));
            if (g != root) { MCHK(hipEventRecord(m->ev[(size_t)g], P->stream)); MCHK(hipSetDevice(R->device)); MCHK(hipStreamWaitEvent(R->stream, m->ev[(size_t)g], 0)); }
        }
    }
#undef MCHK
    return RCX_RC_OK;
}
extern "C" int rcx_multi_scatter_dev(rcx_multi* m, int root, const uint8_t* root_buf, const uint64_t* range_off, uint8_t* const* peer_buf)
{
    return multi_move(m, root, const_cast<uint8_t*>(root_buf), range_off, peer_buf, true);
}
extern "C" int rcx_multi_gather_dev(rcx_multi* m, int root, uint8_t* root_buf, const uint64_t* range_off, const uint8_t* const* peer_buf)
{
    return multi_move(m, root, root_buf, range_off, const_cast<uint8_t* const*>(reinterpret_cast<const uint8_t* const*>(peer_buf)), false);
}
extern "C" int rcx_multi_count(const rcx_multi* m) { return m ? (int)m->ctx.size() : 0; }
extern "C" rcx_ctx* rcx_multi_ctx(rcx_multi* m, int i) { return (m && i >= 0 && (size_t)i < m->ctx.size()) ? m->ctx[(size_t)i] : nullptr; }
extern "C" const char* rcx_multi_last_error(const rcx_multi* m) { return m ? m->err.c_str() : "null multi"; }

// contiguous ranges balanced by weight: range g ends where the running sum first reaches g / parts of the total (what
// rust_compress_amd/dist.py `partition` and host/compress.hpp `partition` compute)
extern "C" void rcx_partition(const uint64_t* weights, uint32_t nblocks, uint32_t parts, uint32_t* bounds)
{
    if (!bounds || parts == 0) return;
    for (uint32_t g = 0; g <= parts; g++) bounds[g] = nblocks;
    bounds[0] = 0;
    if (!weights || nblocks == 0) return;
    long double total = 0, run = 0;
    for (uint32_t i = 0; i < nblocks; i++) total += (long double)weights[i];
    uint32_t g = 1;
    for (uint32_t i = 0; i < nblocks && g < parts; i++) {
        while (g < parts && run >= total * g / parts) bounds[g++] = i;
        run += (long double)weights[i];
    }
}

// one codec's host-descriptor entry point by its number (what rcx_multi_batch runs on a range)
static int run_codec(rcx_ctx* c, int codec, const rcx_batch* b, const uint32_t* aux_in, uint32_t* aux_out, const uint64_t* n_out)
{
    switch (codec) {
    case RCX_ADLER32: case RCX_CRC32: return run_batch(c, codec, b, nullptr, aux_out, nullptr, false);
    case RCX_DC_DECODE: if (!n_out) { c->err = "dc decode: n_out missing"; return RCX_RC_BAD_ARG; } return run_batch(c, codec, b, nullptr, nullptr, n_out, true);
    case RCX_BWT_INVERSE: case RCX_BWT_INVERSE_MINIMAL: case RCX_BWT_INVERSION_TABLE:
        if (!aux_in) { c->err = "bwt inverse: origins missing"; return RCX_RC_BAD_ARG; }
        return run_batch(c, codec, b, aux_in, nullptr, nullptr, true);
    default:
        if (codec < 0 || codec >= RCX_CODEC_COUNT) { c->err = "unknown codec"; return RCX_RC_BAD_ARG; }
        return run_batch(c, codec, b, nullptr, aux_out, nullptr, true);
    }
}

extern "C" int rcx_multi_batch(rcx_multi* m, int codec, const rcx_batch* b, const uint32_t* aux_in, uint32_t* aux_out, const uint64_t* n_out)
{
    if (!m || m->ctx.empty()) return RCX_RC_BAD_ARG;
    if (!b || (b->nblocks && (!b->in_off || !b->in_len || !b->status))) { m->err = "null descriptor array"; return RCX_RC_BAD_ARG; }
    if (b->mem != RCX_MEM_HOST) { m->err = "rcx_multi_batch takes host-memory batches (device-resident ranges: rcx_multi_launch_dev)"; return RCX_RC_BAD_ARG; }
    const uint32_t n = b->nblocks, G = (uint32_t)m->ctx.size();
    if (n == 0) return RCX_RC_OK;
    const bool has_out = b->out_off && b->out_cap;
    m->err.clear();
    std::vector<uint32_t> bounds; std::vector<int> rcs; std::vector<std::thread> th;
    try { bounds.resize(G + 1); rcs.assign(G, RCX_RC_OK); th.reserve(G); } catch (...) { m->err = "out of memory"; return RCX_RC_NO_MEMORY; }
    rcx_partition(has_out ? b->out_cap : b->in_len, n, G, bounds.data());
    int spawn_rc = RCX_RC_OK;
    for (uint32_t g = 0; g < G && spawn_rc == RCX_RC_OK; g++) {
        const uint32_t a0 = bounds[g], a1 = bounds[g + 1];
        if (a1 <= a0) continue;
        try {
        th.emplace_back([&, g, a0, a1] {
          try {
            // the range's own view of the host buffers: offsets rebased to the range's first byte, so that only its span travels
            const uint32_t k = a1 - a0;
            uint64_t lo_in = ~0ull, lo_out = ~0ull;
            for (uint32_t i = a0; i < a1; i++) { if (b->in_off[i] < lo_in) lo_in = b->in_off[i]; if (has_out && b->out_off[i] < lo_out) lo_out = b->out_off[i]; }
            std::vector<uint64_t> io(k), oo(has_out ? k : 0);
            for (uint32_t i = 0; i < k; i++) { io[i] = b->in_off[a0 + i] - lo_in; if (has_out) oo[i] = b->out_off[a0 + i] - lo_out; }
            rcx_batch sb = *b;
            sb.in_base = b->in_base ? b->in_base + lo_in : nullptr; sb.in_off = io.data(); sb.in_len = b->in_len + a0;
            if (has_out) { sb.out_base = b->out_base ? b->out_base + lo_out : nullptr; sb.out_off = oo.data(); sb.out_cap = b->out_cap + a0; }
            if (b->out_len) sb.out_len = b->out_len + a0;
            if (b->in_used) sb.in_used = b->in_used + a0;
            sb.status = b->status + a0;
            sb.nblocks = k;
            rcs[g] = run_codec(m->ctx[g], codec, &sb, aux_in ? aux_in + a0 : nullptr, aux_out ? aux_out + a0 : nullptr, n_out ? n_out + a0 : nullptr);
          } catch (const std::bad_alloc&) { m->ctx[g]->err = "out of memory"; rcs[g] = RCX_RC_NO_MEMORY; }
            catch (...) { m->ctx[g]->err = "unexpected exception"; rcs[g] = RCX_RC_HIP_ERROR; }
        });
        } catch (...) { spawn_rc = RCX_RC_NO_MEMORY; }                       // (std::system_error: no thread to be had; the ones started are joined below)
    }
    for (std::thread& t : th) t.join();
    if (spawn_rc != RCX_RC_OK) { m->err = "cannot start a worker thread"; return spawn_rc; }
    for (uint32_t g = 0; g < G; g++)
        if (rcs[g] != RCX_RC_OK) { m->err = "device range " + std::to_string(g) + ": " + m->ctx[g]->err; return rcs[g]; }
    return RCX_RC_OK;
}

extern "C" int rcx_multi_launch_dev(rcx_multi* m, int codec, const rcx_dev_batch* const* per_device, void* const* scratch, const uint64_t* scratch_bytes)
{
    if (!m || !per_device) return RCX_RC_BAD_ARG;
    m->err.clear();
    for (size_t g = 0; g < m->ctx.size(); g++) {
        if (!per_device[g] || per_device[g]->nblocks == 0) continue;
        const int rc = rcx_launch_dev(m->ctx[g], codec, per_device[g], scratch ? scratch[g] : nullptr, scratch_bytes ? scratch_bytes[g] : 0);
        if (rc != RCX_RC_OK) { m->err = "device range " + std::to_string(g) + ": " + m->ctx[g]->err; return rc; }
    }
    return RCX_RC_OK;
}

extern "C" int rcx_multi_sync(rcx_multi* m)
{
    if (!m) return RCX_RC_BAD_ARG;
    for (size_t g = 0; g < m->ctx.size(); g++) {
        rcx_ctx* c = m->ctx[g];
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { m->err = "device range " + std::to_string(g) + ": synchronize failed"; return RCX_RC_HIP_ERROR; }
    }
    return RCX_RC_OK;
}

// ---- measurement aid: the device's own copy rate (include/rcx.h) ------------------------------------------------------------------
// ONE 16-byte chunk a thread, a workgroup per 4 KiB, the whole buffer in one launch: 6.23 TB/s on the MI355X box, what
// /opt/skills/guides/MI355X_MICROARCH.md measures with a float4 copy (6.29).  benchmarks/micro/hbm_copy.hip tried the shapes: grid-stride
// loops (4 .. 32 workgroups a CU, 1 / 4 / 8 chunks in flight a thread, nontemporal or not) reach 4.3 - 5.5 TB/s, hipMemcpy 4.7, torch's
// copy_ 5.4 -- more loads in flight per thread do not help a copy, more workgroups in flight do.
__global__ __launch_bounds__(256) void k_hbm_copy(const rcx_u32x4* __restrict__ src, rcx_u32x4* __restrict__ dst, uint64_t n16)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
extern "C" int rcx_hbm_copy_probe(rcx_ctx* c, uint64_t bytes, int reps, double* gb_per_s)
{
    if (!c || !gb_per_s || bytes < 4096 || reps < 1) return RCX_RC_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t n16 = bytes / 16;
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = RCX_RC_OK;
    if ((n16 + 255) / 256 > 0x7fffffffull) { c->err = "copy probe: buffer too large"; return RCX_RC_BAD_ARG; }
    const uint32_t grid = (uint32_t)((n16 + 255) / 256);
    float ms = 0;
    if (hipMalloc(&a, n16 * 16) != hipSuccess || hipMalloc(&b, n16 * 16) != hipSuccess) { (void)hipGetLastError(); c->err = "copy probe: out of device memory"; rc = RCX_RC_NO_MEMORY; }
    else if (hipMemsetAsync(a, 0x5a, n16 * 16, c->stream) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); c->err = "copy probe: setup failed"; rc = RCX_RC_HIP_ERROR; }
    else {
        hipLaunchKernelGGL(k_hbm_copy, dim3(grid), dim3(256), 0, c->stream, (const rcx_u32x4*)a, (rcx_u32x4*)b, n16);      // warm-up
        (void)hipEventRecord(e0, c->stream);
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_hbm_copy, dim3(grid), dim3(256), 0, c->stream, (const rcx_u32x4*)a, (rcx_u32x4*)b, n16);
        (void)hipEventRecord(e1, c->stream);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0) { (void)hipGetLastError(); c->err = "copy probe: timing failed"; rc = RCX_RC_HIP_ERROR; }
        else *gb_per_s = 2.0 * (double)(n16 * 16) * reps / ((double)ms * 1e-3) / 1e9;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return rc;
}

// ---- page-locking a caller's buffers (include/rcx.h) -----------------------------------------------------------------------------
extern "C" int rcx_host_register(void* ptr, uint64_t bytes)
{
    if (!ptr || !bytes) return RCX_RC_BAD_ARG;
    const hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e == hipSuccess) return RCX_RC_OK;
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? RCX_RC_NO_MEMORY : e == hipErrorNoDevice ? RCX_RC_NO_DEVICE : RCX_RC_HIP_ERROR;
}
extern "C" int rcx_host_unregister(void* ptr)
{
    if (!ptr) return RCX_RC_BAD_ARG;
    const hipError_t e = hipHostUnregister(ptr);
    if (e == hipSuccess) return RCX_RC_OK;
    (void)hipGetLastError();
    return RCX_RC_HIP_ERROR;
}
