"""rust_compress_amd -- MI355X-native (gfx950) block-codec engine behind the `compress::*` hot path.

Only what the path needs lives here: csrc/ (HIP kernels + the C-ABI of include/rcx.h), the ctypes
binding, the batch API, the stream-codec mirror of the reference interface (compress.py) and the
synthetic data generators.  There is no CPU fallback: the HIP library must be built and a GPU present.
"""
from . import _native  # noqa: F401
from .api import BlockError, Context, DeviceBatch, RcxError, Result  # noqa: F401

__all__ = ["Context", "DeviceBatch", "Result", "BlockError", "RcxError"]
