"""`_dry_codec` with a FAULT INJECTED (TEST INFRASTRUCTURE): on the rank named by RCX_DRY_FAULT_RANK the grouped point-to-point
exchange inside `dist.scatter_blocks` raises -- a rank that leaves the end-to-end leg's scatter half way, with its peers
waiting for it.  tests/test_bench_dry.py uses it to check that bench.py still prints its line with the headline value."""
import inspect
import os

from _dry_codec import *  # noqa: F401,F403
import rust_compress_amd.dist as _D

_group0 = _D._group


def _faulty_group(ops):
    if os.environ.get("RANK") == os.environ.get("RCX_DRY_FAULT_RANK", "1") and any(f.function == "scatter_blocks" for f in inspect.stack()[1:4]):
        raise RuntimeError("injected fault: rank %s leaves the scatter" % os.environ.get("RANK"))
    return _group0(ops)


_D._group = _faulty_group
