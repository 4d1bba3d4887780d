"""Bulk randomized decoder parity (the same code as benchmarks/fuzz_gpu.py, 100 000 streams): mutated LZ4 / zlib / raw
DEFLATE / RLE / Ari streams through every decoder kernel and the oracle -- statuses everywhere, bytes and consumed counts
wherever the oracle succeeds."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))


@pytest.mark.gpu
def test_bulk_decoder_fuzz(ctx):
    import fuzz_gpu
    assert fuzz_gpu.main(100000, 7, ctx) == 0
