"""ctypes binding of csrc/librcx.so (the C-ABI declared in include/rcx.h).

The library is the product: if it is missing this module raises -- there is no CPU fallback and no
import of anything under oracle/.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# RCX_AB=1 loads the A/B build (csrc/librcx_ab.so, -DRCX_AB_VARIANTS: earlier kernel generations, profiling
# instantiations, experiments) that benchmarks/ compares against; the product is librcx.so.
AB = bool(os.environ.get("RCX_AB"))
LIB_PATH = os.path.join(_HERE, "csrc", os.environ.get("RCX_LIB_FILE") or ("librcx_ab.so" if AB else "librcx.so"))    # RCX_LIB_FILE: a build made with RCX_EXTRA_FLAGS (experiments)
# kernel variants each build accepts (rcx_ctx_set_variant): default + one fallback per codec in the shipped library
LZ4_DECODE_VARIANTS = (0, 15, 11, 20, 1, 2, 3, 4, 5, 6, 7, 8, 10, 17) if AB else (0, 15)
INFLATE_VARIANTS = (0, 9, 10, 11, 1) if AB else (0, 9, 10, 11)

# enum rcx_codec
(LZ4_DECODE, LZ4_ENCODE, INFLATE, ZLIB_DECODE, ADLER32, BWT_FORWARD, BWT_INVERSE, MTF_ENCODE, MTF_DECODE,
 DC_ENCODE, DC_DECODE, ARI_BYTE_ENCODE, ARI_BYTE_DECODE, RLE_ENCODE, RLE_DECODE, CRC32, GZIP_DECODE,
 ARI_BINARY_ENCODE, ARI_BINARY_DECODE, ARI_PROXY_ENCODE, ARI_PROXY_DECODE, ARI_APM_ENCODE, ARI_APM_DECODE,
 BWT_INVERSE_MINIMAL, BWT_SUFFIXES, BWT_INVERSION_TABLE, CODEC_COUNT) = range(27)
MEM_HOST, MEM_DEVICE = 0, 1
# enum rcx_status (the ones Python code names; include/rcx.h has them all)
E_EOF, E_OUTPUT_TOO_SMALL, E_MALFORMED = 1, 2, 3
E_GZIP_MAGIC, E_GZIP_METHOD, E_GZIP_FLAGS, E_GZIP_CRC, E_GZIP_ISIZE = 50, 51, 52, 53, 54
E_BWT_BLOCK_TOO_LARGE = 60
RC_OK, RC_BAD_ARG, RC_NO_DEVICE, RC_HIP_ERROR, RC_NO_MEMORY = 0, -1, -2, -3, -4

EXPORTS = [
    "rcx_version", "rcx_ctx_create", "rcx_ctx_destroy", "rcx_ctx_set_stream", "rcx_ctx_set_variant", "rcx_last_error",
    "rcx_status_string", "rcx_lz4_decode_batch", "rcx_lz4_encode_batch", "rcx_lz4_compression_bound",
    "rcx_inflate_batch", "rcx_zlib_decode_batch", "rcx_adler32_batch", "rcx_crc32_batch", "rcx_gzip_decode_batch",
    "rcx_bwt_forward_batch", "rcx_bwt_inverse_batch", "rcx_mtf_encode_batch", "rcx_mtf_decode_batch", "rcx_dc_encode_batch",
    "rcx_dc_decode_batch", "rcx_dc_encode_ctx_batch", "rcx_dc_decode_ctx_batch", "rcx_ari_byte_encode_batch", "rcx_ari_byte_decode_batch", "rcx_ari_byte_encode_bound",
    "rcx_rle_encode_batch", "rcx_rle_decode_batch", "rcx_rle_encode_bound", "rcx_scratch_bytes", "rcx_launch_dev",
    "rcx_ari_binary_encode_batch", "rcx_ari_binary_decode_batch", "rcx_ari_proxy_encode_batch", "rcx_ari_proxy_decode_batch",
    "rcx_ctx_set_param", "rcx_ari_apm_encode_batch", "rcx_ari_apm_decode_batch", "rcx_bwt_inverse_minimal_batch",
    "rcx_bwt_suffixes_batch", "rcx_bwt_inversion_table_batch",
    "rcx_multi_create", "rcx_multi_destroy", "rcx_multi_count", "rcx_multi_ctx", "rcx_partition", "rcx_multi_batch",
    "rcx_multi_launch_dev", "rcx_multi_sync", "rcx_multi_last_error", "rcx_host_register", "rcx_host_unregister",
    "rcx_hbm_copy_probe", "rcx_multi_scatter_dev", "rcx_multi_gather_dev", "rcx_multi_transport",
]


class Batch(C.Structure):          # struct rcx_batch
    _fields_ = [("in_base", C.c_void_p), ("in_off", C.c_void_p), ("in_len", C.c_void_p),
                ("out_base", C.c_void_p), ("out_off", C.c_void_p), ("out_cap", C.c_void_p),
                ("out_len", C.c_void_p), ("in_used", C.c_void_p), ("status", C.c_void_p),
                ("nblocks", C.c_uint32), ("mem", C.c_int)]


class DevBatch(C.Structure):       # struct rcx_dev_batch
    _fields_ = [("in_base", C.c_void_p), ("in_off", C.c_void_p), ("in_len", C.c_void_p),
                ("out_base", C.c_void_p), ("out_off", C.c_void_p), ("out_cap", C.c_void_p),
                ("out_len", C.c_void_p), ("in_used", C.c_void_p), ("status", C.c_void_p),
                ("aux", C.c_void_p), ("nblocks", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("rust_compress_amd: %s is missing -- build it with `python __graft_entry__.py build` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm ships its own libamdhip64 and this package hands torch's device pointers and streams to librcx.so: both must
        # live on ONE HIP runtime, which they do when torch is in the process first (the loader then resolves librcx.so's dependency
        # to the copy already mapped).  Loaded the other way round, rcx_ctx_create found "no HIP device" on a box with one.
        # RCX_NO_TORCH_PRELOAD=1 skips this for pure-ctypes users who never hand over torch pointers (no multi-second import).
        if "torch" not in sys.modules and not os.environ.get("RCX_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.rcx_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.rcx_ctx_destroy.argtypes = [C.c_void_p]
        L.rcx_ctx_destroy.restype = None
        L.rcx_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.rcx_ctx_set_variant.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rcx_last_error.argtypes = [C.c_void_p]
        L.rcx_last_error.restype = C.c_char_p
        L.rcx_status_string.argtypes = [C.c_int]
        L.rcx_status_string.restype = C.c_char_p
        for name in ("rcx_lz4_compression_bound", "rcx_ari_byte_encode_bound", "rcx_rle_encode_bound"):
            f = getattr(L, name)
            f.argtypes = [C.c_uint64]
            f.restype = C.c_uint64
        L.rcx_scratch_bytes.argtypes = [C.c_int, C.c_uint32, C.c_uint64]
        L.rcx_scratch_bytes.restype = C.c_uint64
        L.rcx_launch_dev.argtypes = [C.c_void_p, C.c_int, C.POINTER(DevBatch), C.c_void_p, C.c_uint64]
        for name in ("rcx_lz4_decode_batch", "rcx_lz4_encode_batch", "rcx_mtf_encode_batch", "rcx_mtf_decode_batch",
                     "rcx_dc_encode_batch", "rcx_dc_encode_ctx_batch", "rcx_ari_byte_encode_batch", "rcx_ari_byte_decode_batch",
                     "rcx_rle_encode_batch", "rcx_rle_decode_batch", "rcx_ari_proxy_encode_batch", "rcx_ari_proxy_decode_batch",
                     "rcx_ari_apm_encode_batch", "rcx_ari_apm_decode_batch"):
            getattr(L, name).argtypes = [C.c_void_p, C.POINTER(Batch)]
        for name in ("rcx_ari_binary_encode_batch", "rcx_ari_binary_decode_batch"):
            getattr(L, name).argtypes = [C.c_void_p, C.POINTER(Batch), C.c_uint32]
        L.rcx_ctx_set_param.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        for name in ("rcx_inflate_batch", "rcx_zlib_decode_batch", "rcx_adler32_batch", "rcx_crc32_batch",
                     "rcx_gzip_decode_batch", "rcx_bwt_forward_batch",
                     "rcx_bwt_inverse_batch", "rcx_bwt_inverse_minimal_batch", "rcx_dc_decode_batch", "rcx_dc_decode_ctx_batch",
                     "rcx_bwt_suffixes_batch", "rcx_bwt_inversion_table_batch"):
            getattr(L, name).argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p]
        # more than one device (include/rcx.h: rcx_multi_*)
        L.rcx_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        L.rcx_multi_destroy.argtypes = [C.c_void_p]
        L.rcx_multi_destroy.restype = None
        L.rcx_multi_count.argtypes = [C.c_void_p]
        L.rcx_multi_ctx.argtypes = [C.c_void_p, C.c_int]
        L.rcx_multi_ctx.restype = C.c_void_p
        L.rcx_partition.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.rcx_partition.restype = None
        L.rcx_multi_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p]
        L.rcx_multi_launch_dev.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(DevBatch)), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.rcx_multi_sync.argtypes = [C.c_void_p]
        L.rcx_multi_last_error.argtypes = [C.c_void_p]
        L.rcx_multi_last_error.restype = C.c_char_p
        L.rcx_host_register.argtypes = [C.c_void_p, C.c_uint64]
        L.rcx_host_unregister.argtypes = [C.c_void_p]
        L.rcx_hbm_copy_probe.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
        L.rcx_multi_scatter_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
        L.rcx_multi_gather_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
        L.rcx_multi_transport.argtypes = [C.c_void_p]
        L.rcx_multi_transport.restype = C.c_char_p
        _lib = L
    return _lib
