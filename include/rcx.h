/*
 * rcx.h -- C ABI of the MI355X-native block-codec engine ("rcx" = rust-compress on CDNA).
 *
 * This is the drop-in boundary for the hot path of the Rust crate `compress`
 * (rusty-shell/rust-compress): every entry point below replaces one per-block
 * kernel that the crate's `Decoder<R: Read>` / `Encoder<W: Write>` types call
 * once per block; here it is called once per BATCH of independent blocks.
 * The reference interface each entry point replaces is cited as
 * `reference: <file>:<lines>` (paths relative to the crate root).
 *
 * Conventions (all entry points):
 *   - plain C99 types only; no C++/torch/HIP types cross the boundary
 *     (a HIP stream is passed as `void*`).
 *   - struct-of-arrays batch descriptors: block i reads
 *     in_base[in_off[i] .. in_off[i]+in_len[i]) and writes at most out_cap[i]
 *     bytes at out_base[out_off[i]..]; the callee allocates nothing the caller
 *     keeps, and retains no pointer after return.
 *   - `mem` says where the DATA buffers (in_base/out_base) live:
 *       RCX_MEM_DEVICE: HBM pointers (hipMalloc / torch.cuda tensors);
 *       RCX_MEM_HOST:   host pointers; the library stages them through HBM.
 *     The small descriptor arrays (offsets, lengths, status, ...) are ALWAYS
 *     host arrays; the library copies them to/from the device.
 *   - return value = batch-level status (bad arguments, HIP failure);
 *     per-block results go to status[i] (enum rcx_status). A failing block
 *     never affects another block. Nothing aborts or throws across the ABI.
 *   - the rcx_*_batch calls are synchronous from the caller's view (results are
 *     complete on return); rcx_launch_dev() only enqueues on the ctx stream.
 *   - there is NO CPU fallback: without a usable HIP device every call fails
 *     with RCX_E_NO_DEVICE.
 */
#ifndef RCX_H
#define RCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-block status: mirrors the reference's io::Error strings 1:1 ------ */
enum rcx_status {
    RCX_OK = 0,
    /* lib.rs:53-62,115-118 "unexpected end of file" / byteorder UnexpectedEof */
    RCX_E_EOF = 1,
    /* new: caller's out_cap[i] too small (the reference grows a Vec instead) */
    RCX_E_OUTPUT_TOO_SMALL = 2,
    /* input on which the reference panics (index OOB / assert!) or reads
     * uninitialised memory: lz4.rs:93,135,407; bwt/mod.rs:114,230; dc.rs:213;
     * flate.rs:297,432; ari/mod.rs:282 */
    RCX_E_MALFORMED = 3,
    /* flate.rs:56-65 */
    RCX_E_HUFFMAN_TREE_TOO_LARGE = 10,     /* "huffman tree too large" */
    RCX_E_INVALID_BLOCK_CODE = 11,         /* "invalid block code" */
    RCX_E_INVALID_HUFFMAN_HEADER_SYMBOL = 12, /* "invalid huffman header symbol" */
    RCX_E_INVALID_HUFFMAN_TREE = 13,       /* "invalid huffman tree" */
    RCX_E_INVALID_HUFFMAN_TREE_HEADER = 14,/* "invalid huffman tree header" */
    RCX_E_INVALID_HUFFMAN_CODE = 15,       /* "invalid huffman code" */
    RCX_E_INVALID_STATIC_SIZE = 16,        /* "invalid static size" */
    RCX_E_NOT_ENOUGH_BITS = 17,            /* "not enough bits" */
    /* zlib.rs:59-84,111-114 */
    RCX_E_ZLIB_FORMAT = 20,     /* "unsupported zlib stream format" */
    RCX_E_ZLIB_WINDOW = 21,     /* "unsupported zlib window size" */
    RCX_E_ZLIB_DICT = 22,       /* "unsupported initial dictionary in the output stream" */
    RCX_E_ZLIB_HEADER_CHECKSUM = 23, /* "invalid zlib header checksum" */
    RCX_E_ZLIB_CHECKSUM = 24,   /* "invalid checksum on zlib stream" */
    /* rle.rs:153 */
    RCX_E_RLE_LONG_RUN = 30,    /* "Overly long run" */
    /* lz4.rs:366,376: InvalidInput with empty message */
    RCX_E_LZ4_MAGIC = 40,
    RCX_E_LZ4_VERSION = 41,
    /* lz4.rs:229-230: compression_bound() == None -> encode returns 0 */
    RCX_E_LZ4_INPUT_TOO_LARGE = 42,
    /* gzip member framing (RFC 1952; extension, see rcx_gzip_decode_batch) */
    RCX_E_GZIP_MAGIC = 50,          /* ID1 ID2 != 1f 8b */
    RCX_E_GZIP_METHOD = 51,         /* CM != 8 */
    RCX_E_GZIP_FLAGS = 52,          /* reserved FLG bits set */
    RCX_E_GZIP_CRC = 53,            /* CRC32 trailer mismatch */
    RCX_E_GZIP_ISIZE = 54,          /* ISIZE trailer mismatch */
    /* a limit of this implementation, not of the format: bwt::Encoder::new(w, block_size) takes any usize (bwt/mod.rs:451), the suffix
     * sorter keeps four flag bits beside a suffix index -- a block of 2^28 bytes or more gets this status (out_len 0) and the rest of
     * the batch is transformed.  The host mirrors turn it into io::ErrorKind::InvalidInput. */
    RCX_E_BWT_BLOCK_TOO_LARGE = 60  /* "bwt block of 2^28 bytes or more" */
};

/* ---- batch-level return codes --------------------------------------------- */
enum rcx_rc {
    RCX_RC_OK = 0,
    RCX_RC_BAD_ARG = -1,
    RCX_RC_NO_DEVICE = -2,   /* no HIP device / runtime: there is no CPU path */
    RCX_RC_HIP_ERROR = -3,
    RCX_RC_NO_MEMORY = -4
};

enum rcx_mem { RCX_MEM_HOST = 0, RCX_MEM_DEVICE = 1 };

/* inflate per-stream flag bits (out: flags[i]) */
#define RCX_W_EMPTY_BLOCK_MIDSTREAM 1u /* flate.rs:474-476 quirk: the reference's
                                          read() returns Ok(0) here; we decode on */

typedef struct rcx_ctx rcx_ctx;

/* ---- context ---------------------------------------------------------------- */
/* One ctx per caller thread and device (thread-compatible, not thread-safe).
 * device_id < 0 selects the current HIP device. */
int  rcx_ctx_create(int device_id, rcx_ctx** out);
void rcx_ctx_destroy(rcx_ctx* ctx);
/* Launch on the caller's HIP stream (hipStream_t passed as void*; NULL = the HIP
 * null stream).  A new ctx launches on its own non-blocking stream. */
int  rcx_ctx_set_stream(rcx_ctx* ctx, void* hip_stream);
const char* rcx_last_error(const rcx_ctx* ctx);
const char* rcx_status_string(int status);   /* the reference's error text */
int  rcx_version(void);

/* A batch descriptor shared by every codec (struct-of-arrays, host arrays). */
typedef struct rcx_batch {
    const uint8_t*  in_base;   /* mem */
    const uint64_t* in_off;    /* host [n] */
    const uint64_t* in_len;    /* host [n] */
    uint8_t*        out_base;  /* mem */
    const uint64_t* out_off;   /* host [n] */
    const uint64_t* out_cap;   /* host [n] */
    uint64_t*       out_len;   /* host [n], written */
    uint64_t*       in_used;   /* host [n] or NULL, written: bytes consumed */
    int32_t*        status;    /* host [n], written: enum rcx_status */
    uint32_t        nblocks;
    int             mem;       /* enum rcx_mem for in_base/out_base */
} rcx_batch;

/* ---- LZ4 -------------------------------------------------------------------- */
/* reference: src/lz4.rs:602-611 decode_block() -> BlockDecoder::decode :67-110
 * RCX_MEM_HOST with a PAGE-LOCKED out_base (hipHostMalloc / hipHostRegister): the decoder stores the decoded bytes straight into
 * the caller's buffer while it runs (nothing beyond out_len[i] of a block's slot is written for a block that decodes; a block that
 * fails, or one whose late input made the library decode it a second time, may leave bytes of its first attempt anywhere in its
 * slot -- never outside it) and, when in_base is page-locked
 * too, the compressed bytes arrive in block ranges under the launch; with pageable buffers one copy each way around the launch.
 * Same results either way.  rcx_ctx_set_param(ctx, RCX_LZ4_DECODE, 1) keeps the plain copies (A/B). */
int rcx_lz4_decode_batch(rcx_ctx*, const rcx_batch*);
/* reference: src/lz4.rs:616-627 encode_block() -> BlockEncoder::encode :226-310
 * (bit-exact: hash/skip/backtrack heuristics reproduced) */
int rcx_lz4_encode_batch(rcx_ctx*, const rcx_batch*);
/* reference: src/lz4.rs:175-181 compression_bound(); 0 == None */
uint64_t rcx_lz4_compression_bound(uint64_t in_len);

/* ---- DEFLATE / zlib / Adler-32 ---------------------------------------------- */
/* reference: src/flate.rs:195-206,237-246,262-341,343-450 (one RFC-1951 stream
 * per block, decoded to BFINAL). flags may be NULL.
 * RCX_MEM_HOST with a PAGE-LOCKED out_base: as rcx_lz4_decode_batch -- rcx_inflate_batch, rcx_zlib_decode_batch and rcx_gzip_decode_batch
 * store the decoded bytes straight into the caller's buffer while the launch runs (streams the fast kernel hands to the exact one
 * arrive by a copy behind it); same results as with pageable buffers.  rcx_ctx_set_param(ctx, <codec>, 1) keeps the plain copies. */
int rcx_inflate_batch(rcx_ctx*, const rcx_batch*, uint32_t* flags);
/* reference: src/zlib.rs:55-126 (header checks, inflate, Adler-32 BE trailer) */
int rcx_zlib_decode_batch(rcx_ctx*, const rcx_batch*, uint32_t* flags);
/* reference: src/checksum/adler.rs:22-51; out_len/out_base unused,
 * adler[i] = State32::result() of block i */
int rcx_adler32_batch(rcx_ctx*, const rcx_batch*, uint32_t* adler);
/* ---- extension beyond the reference ("gzip members" of BASELINE config 3) ----
 * The reference has RFC 1950 (zlib) framing only, src/zlib.rs:55-126; these two follow its structure for RFC 1952.
 * crc[i] = CRC-32 (IEEE 802.3, as in the gzip trailer) of block i; out_len/out_base unused. */
int rcx_crc32_batch(rcx_ctx*, const rcx_batch*, uint32_t* crc);
/* one gzip member per block: header (FEXTRA/FNAME/FCOMMENT/FHCRC skipped), DEFLATE stream decoded to BFINAL,
 * CRC32 + ISIZE trailer verified; in_used[i] = bytes of the member.  flags as rcx_inflate_batch. */
int rcx_gzip_decode_batch(rcx_ctx*, const rcx_batch*, uint32_t* flags);

/* ---- BWT / MTF / DC --------------------------------------------------------- */
/* reference: src/bwt/mod.rs:136-219 compute_suffixes + TransformIterator.
 * out block i receives L (n bytes); origin[i] = get_origin(). */
int rcx_bwt_forward_batch(rcx_ctx*, const rcx_batch*, uint32_t* origin);
/* reference: src/bwt/mod.rs:136-166 compute_suffixes (pub): the sorted suffix array itself.  out block i receives n
 * little-endian u32 suffix indices (out_cap >= 4n, the slot 4-byte aligned; out_len = 4n); origin[i] (may be NULL) =
 * the position of suffix 0, what TransformIterator::get_origin reports for the same array. */
int rcx_bwt_suffixes_batch(rcx_ctx*, const rcx_batch*, uint32_t* origin);
/* reference: src/bwt/mod.rs:223-239 compute_inversion_table (pub): in block i = L (n bytes), origin[i]; out block i receives
 * the n little-endian u32 table entries (out_cap >= 4n, 4-byte aligned; out_len = 4n): table[place(L[origin])] = 0, then
 * table[place(L[j])] = j + 1 for every other j in order.  status: origin >= n is RCX_E_MALFORMED (the index panic of :230,
 * also for an empty block). */
int rcx_bwt_inversion_table_batch(rcx_ctx*, const rcx_batch*, const uint32_t* origin);
/* reference: src/bwt/mod.rs:223-294 compute_inversion_table + InverseIterator */
int rcx_bwt_inverse_batch(rcx_ctx*, const rcx_batch*, const uint32_t* origin);
/* reference: src/bwt/mod.rs:298-315 decode_minimal, what bwt::Decoder runs with extra_mem = false (:397-399): n steps of
 * i <- C[L[i]] + #{k < i : L[k] == L[i]} from i = origin, the text written backwards.  Reproduced as the reference computes
 * it, which is NOT the inverse of rcx_bwt_forward_batch in general (wrong whenever T[n-1] also occurs in L[..origin]; right
 * for e.g. "abracadabra", the reference's only test of it, :549-551).  status: origin >= n is RCX_E_MALFORMED (:310),
 * n == 0 is RCX_OK only with origin == 0 (:300-302); there is no other failure. */
int rcx_bwt_inverse_minimal_batch(rcx_ctx*, const rcx_batch*, const uint32_t* origin);
/* reference: src/bwt/mtf.rs:63-90 with the stream codecs' identity start :103,141 */
int rcx_mtf_encode_batch(rcx_ctx*, const rcx_batch*);
int rcx_mtf_decode_batch(rcx_ctx*, const rcx_batch*);
/* reference: src/bwt/dc.rs:110-159 encode_simple::<u32> order: out block i =
 * little-endian u32 words: 256 x init, then k distances (out_len = 4*(256+k)) */
int rcx_dc_encode_batch(rcx_ctx*, const rcx_batch*);
/* reference: src/bwt/dc.rs:162-252 decode_simple; in = the words above,
 * n_out[i] = decoded length n (dc carries no length itself) */
int rcx_dc_decode_batch(rcx_ctx*, const rcx_batch*, const uint64_t* n_out);
/* The same two with the coding CONTEXT of every distance (reference: src/bwt/dc.rs:40-58 `Context`; yielded next to each
 * distance by EncodeIterator :88-103, handed to decode's distance callback :199-229; the reference's test :268-289 checks
 * that the two sides see the same contexts).  A context is 8 bytes: u32 LE symbol | last_rank << 8, u32 LE distance_limit.
 *   encode: block i's slot must hold 4*(256+n) + 8*n bytes (n = in_len[i]); words as above in its first 4*(256+k) bytes,
 *           the k contexts from byte 4*(256+n) on; out_len[i] = 4*(256+n) + 8*k.
 *   decode: the slot must be 8-byte aligned and hold ((n+7)&~7) + 8*(in_len[i]/4 - 256) bytes; the n decoded bytes first,
 *           one context per distance consumed from byte (n+7)&~7 on; out_len[i] = that offset + 8 * (distances consumed). */
int rcx_dc_encode_ctx_batch(rcx_ctx*, const rcx_batch*);
int rcx_dc_decode_ctx_batch(rcx_ctx*, const rcx_batch*, const uint64_t* n_out);

/* ---- adaptive byte range coder ---------------------------------------------- */
/* reference: src/entropy/ari/table.rs:185-224 ByteEncoder::write + finish
 * (RangeEncoder::process mod.rs:117-150, table::Model :69-117) */
int rcx_ari_byte_encode_batch(rcx_ctx*, const rcx_batch*);
/* reference: src/entropy/ari/table.rs:229-273 ByteDecoder::read (+ finish);
 * in_used[i] = bytes consumed so the next stream stays addressable */
int rcx_ari_byte_decode_batch(rcx_ctx*, const rcx_batch*);
uint64_t rcx_ari_byte_encode_bound(uint64_t in_len);
/* The crate's other two models have no stream codec; these entry points drive them exactly as the reference's
 * tests do.  reference: src/entropy/ari/bin.rs:17-103 bin::Model::new_flat(RANGE_DEFAULT_THRESHOLD >> 3, rate),
 * 8 decisions per byte LSB first (src/entropy/ari/test.rs:22-50).  rate must be 1..31.  The coding has no end
 * marker: the decoder produces exactly out_cap[i] bytes.  Encoded size <= rcx_ari_byte_encode_bound(n). */
int rcx_ari_binary_encode_batch(rcx_ctx*, const rcx_batch*, uint32_t rate);
int rcx_ari_binary_decode_batch(rcx_ctx*, const rcx_batch*, uint32_t rate);
/* reference: table::SumProxy (table.rs:127-180, weights 2:1 >> 0, update 10/5) for the high nibble + bin::SumProxy
 * (bin.rs:112-167, weights 1:1 >> 1, rates 3 and 5) for the low 4 bits, as src/entropy/ari/test.rs:91-148 */
int rcx_ari_proxy_encode_batch(rcx_ctx*, const rcx_batch*);
int rcx_ari_proxy_decode_batch(rcx_ctx*, const rcx_batch*);
/* reference: apm::Bit passed through an apm::Gate, update(rate 10, bias 0), 8 decisions per byte LSB first, as
 * src/entropy/ari/test.rs:150-182 (apm.rs:36-198).  The f32 ln / exp of Bit::to_wide / from_wide are evaluated on the
 * host with libm's logf / expf (what Rust's f32::ln / exp call) into a 4096-entry stretch table and the 17 initial
 * gate bins; the device code is integer only.  A status of RCX_E_MALFORMED on ENCODE means the reference panics on
 * that input (a skewed enough bit history drives the gate index out of its 17 bins, apm.rs:162-166).  The decoder
 * produces exactly out_cap[i] bytes. */
int rcx_ari_apm_encode_batch(rcx_ctx*, const rcx_batch*);
int rcx_ari_apm_decode_batch(rcx_ctx*, const rcx_batch*);

/* ---- RLE -------------------------------------------------------------------- */
/* reference: src/rle.rs:82-122 (one-shot write + finish) */
int rcx_rle_encode_batch(rcx_ctx*, const rcx_batch*);
/* reference: src/rle.rs:194-259 */
int rcx_rle_decode_batch(rcx_ctx*, const rcx_batch*);
uint64_t rcx_rle_encode_bound(uint64_t in_len);

/* ---- device-resident descriptors (benchmark / pipeline use) ------------------ */
/* Same kernels, but every array (offsets, lengths, status, ...) already lives
 * in HBM, nothing is copied and nothing is synchronised: the call enqueues on
 * the ctx stream and returns. This is what bench.py times.
 * ONE exception: RCX_BWT_FORWARD / RCX_BWT_SUFFIXES read in_len back and wait once per prefix-doubling round for a few counter words
 * (the host decides which of the sorter's kernels the next round needs, csrc/k_bwt.hip) -- the call returns when the last round has been
 * enqueued, with the transform's final kernels still running.  A caller that overlaps BWT batches gives each its own context and
 * thread (pipeline.PipelineLanes does). */
typedef struct rcx_dev_batch {
    const uint8_t*  in_base;
    const uint64_t* in_off;
    const uint64_t* in_len;
    uint8_t*        out_base;
    const uint64_t* out_off;
    const uint64_t* out_cap;
    uint64_t*       out_len;
    uint64_t*       in_used;   /* may be NULL */
    int32_t*        status;
    uint32_t*       aux;       /* codec extra (origin / adler / flags), may be NULL */
    uint32_t        nblocks;
} rcx_dev_batch;

enum rcx_codec {
    RCX_LZ4_DECODE = 0, RCX_LZ4_ENCODE, RCX_INFLATE, RCX_ZLIB_DECODE, RCX_ADLER32,
    RCX_BWT_FORWARD, RCX_BWT_INVERSE, RCX_MTF_ENCODE, RCX_MTF_DECODE,
    RCX_DC_ENCODE, RCX_DC_DECODE, RCX_ARI_BYTE_ENCODE, RCX_ARI_BYTE_DECODE,
    RCX_RLE_ENCODE, RCX_RLE_DECODE, RCX_CRC32, RCX_GZIP_DECODE,
    RCX_ARI_BINARY_ENCODE, RCX_ARI_BINARY_DECODE, RCX_ARI_PROXY_ENCODE, RCX_ARI_PROXY_DECODE,
    RCX_ARI_APM_ENCODE, RCX_ARI_APM_DECODE, RCX_BWT_INVERSE_MINIMAL,
    RCX_BWT_SUFFIXES, RCX_BWT_INVERSION_TABLE, RCX_CODEC_COUNT
};
/* scratch bytes (HBM) the codec needs for nblocks blocks of <= max_block bytes.  Required for LZ4 encode, BWT and gzip
 * decode; for RCX_INFLATE / RCX_ZLIB_DECODE it is what the default (wave-per-stream) decoder needs -- without it
 * rcx_launch_dev falls back to the lane-per-stream kernel (same results, slower on small batches). */
uint64_t rcx_scratch_bytes(int codec, uint32_t nblocks, uint64_t max_block);
int rcx_launch_dev(rcx_ctx*, int codec, const rcx_dev_batch*, void* scratch, uint64_t scratch_bytes);
/* Measurement aid (no counterpart in the reference): what a plain copy reaches on this device -- `bytes` read and `bytes` written per
 * pass by a kernel that moves 16 bytes a thread (a workgroup per 4 KiB), `reps` passes timed with events on the ctx stream; *gb_per_s =
 * 2 * bytes / time.  The "achievable" line next to the 8 TB/s spec peak in bench.py's roofline (hipMemcpy and torch's copy_ read 15-25 %
 * lower: benchmarks/micro/hbm_copy.hip). */
int rcx_hbm_copy_probe(rcx_ctx*, uint64_t bytes, int reps, double* gb_per_s);
/* kernel variant knob for A/B measurements (0 = default/best). */
int rcx_ctx_set_variant(rcx_ctx*, int codec, int variant);
/* codec parameter for rcx_launch_dev (the *_batch entry points take it as an argument): the rate of RCX_ARI_BINARY_*;
 * RCX_LZ4_DECODE: bit 0 = host-memory batches by plain copies (see rcx_lz4_decode_batch), bits 8-15 / 16-23 tuning of the ranges */
int rcx_ctx_set_param(rcx_ctx*, int codec, uint32_t value);

/* ---- more than one device ---------------------------------
 * Blocks are independent in every codec (reference: src/lz4.rs:445-456 one block per frame block, src/bwt/mod.rs:373-401 one
 * record per block, src/entropy/ari/test.rs:52-89 self-terminating streams), so a batch shards by contiguous block ranges and
 * needs no collective: range g runs on device g.  An rcx_multi owns one rcx_ctx per listed device (a device may be listed more
 * than once: two ranges on one GPU).  Three layers, all without Python:
 *   rcx_partition        the ranges: contiguous, balanced by `weights` (decoded bytes), bounds[parts + 1], bounds[0] = 0
 *   rcx_multi_batch      a HOST-memory batch: every range is staged to its device, run and copied back by a host thread of its
 *                        own (one context each: contexts are thread-compatible, not thread-safe); per-block results land in the
 *                        caller's arrays exactly as the one-device entry point of that codec leaves them.  aux_in / aux_out /
 *                        n_out: what that entry point takes beside the batch (origins in, origins / flags / checksums out,
 *                        DC decode's lengths), or NULL.
 *   rcx_multi_launch_dev DEVICE-resident ranges: per_device[g] (arrays in device g's HBM, or NULL for none) is enqueued on device
 *                        g's context stream like rcx_launch_dev; rcx_multi_sync waits for every device.
 *   rcx_multi_scatter_dev / rcx_multi_gather_dev   the batch sits in the HBM of ONE device of the set (`root`): contiguous byte
 *                        ranges travel to the devices that work on them and the results come back DEVICE TO DEVICE, no host staging
 *                        ("RCCL only to scatter inputs and gather outputs").  range_off[g] .. range_off[g + 1] are device
 *                        g's bytes in root_buf (count + 1 host words; an empty range is skipped); peer_buf[g] is device g's buffer,
 *                        range-relative (its byte 0 is root_buf[range_off[g]]); peer_buf[root] may be NULL: that range stays where it
 *                        is.  Both calls only enqueue, each transfer ordered on the streams of the two contexts it connects:
 *                        scatter -> rcx_multi_launch_dev -> gather -> rcx_multi_sync needs no wait in between.
 *                        Transport: RCCL -- grouped ncclSend / ncclRecv, one communicator per device under ncclCommInitAll, librccl
 *                        loaded on first use -- when the set's devices are distinct; a set that lists a device twice (what a one-GPU
 *                        box can test), a host without librccl, or RCX_MULTI_TRANSPORT=peer in the environment use
 *                        hipMemcpyPeerAsync between the contexts' streams (events order it).  rcx_multi_transport() names the one
 *                        in use ("rccl" / "peer"; "" before the first transfer).  Measured on one device only: see DESIGN.md 4. */
typedef struct rcx_multi rcx_multi;
int  rcx_multi_create(const int* device_ids, int n, rcx_multi** out);
void rcx_multi_destroy(rcx_multi*);
int  rcx_multi_count(const rcx_multi*);
rcx_ctx* rcx_multi_ctx(rcx_multi*, int i);      /* device i's context: its stream, variant and parameter knobs, last error */
void rcx_partition(const uint64_t* weights, uint32_t nblocks, uint32_t parts, uint32_t* bounds);
int  rcx_multi_batch(rcx_multi*, int codec, const rcx_batch*, const uint32_t* aux_in, uint32_t* aux_out, const uint64_t* n_out);
int  rcx_multi_scatter_dev(rcx_multi*, int root, const uint8_t* root_buf, const uint64_t* range_off, uint8_t* const* peer_buf);
int  rcx_multi_gather_dev(rcx_multi*, int root, uint8_t* root_buf, const uint64_t* range_off, const uint8_t* const* peer_buf);
const char* rcx_multi_transport(const rcx_multi*);
int  rcx_multi_launch_dev(rcx_multi*, int codec, const rcx_dev_batch* const* per_device, void* const* scratch, const uint64_t* scratch_bytes);
int  rcx_multi_sync(rcx_multi*);
const char* rcx_multi_last_error(const rcx_multi*);

/* ---- page-locked host memory (no counterpart in the reference: its Vec<u8> buffers are pageable) ----------------------------------
 * rcx_lz4_decode_batch writes a page-locked output buffer directly and takes page-locked input in ranges under the launch (above).
 * A host language without the HIP headers pins its own allocation with these: rcx_host_register(ptr, bytes) page-locks
 * [ptr, ptr + bytes) for every device (hipHostRegister, portable + mapped; ~0.1 ms per MiB the first time), rcx_host_unregister
 * undoes it before the memory is freed.  Return enum rcx_rc. */
int rcx_host_register(void* ptr, uint64_t bytes);
int rcx_host_unregister(void* ptr);

#ifdef __cplusplus
}
#endif
#endif /* RCX_H */
