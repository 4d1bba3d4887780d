"""Bulk randomized decoder parity (the same code as benchmarks/fuzz_gpu*.py): 100 000 mutated LZ4 / zlib / raw DEFLATE / RLE /
Ari streams, 6000 mutated DC streams and 2000 arbitrary (L, origin) pairs for the inverse BWT, through the decoder kernels and the
oracle -- statuses everywhere, bytes and consumed counts wherever the oracle succeeds."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))


@pytest.mark.gpu
def test_bulk_decoder_fuzz(ctx):
    import fuzz_gpu
    assert fuzz_gpu.main(100000, 7, ctx) == 0


@pytest.mark.gpu
def test_dc_decode_fuzz(ctx):
    import fuzz_gpu_dc
    assert fuzz_gpu_dc.main(6000, 5, ctx) == 0


@pytest.mark.gpu
def test_bwt_inverse_fuzz(ctx):
    import fuzz_gpu_bwti
    assert fuzz_gpu_bwti.main(2000, 6, ctx) == 0
    assert fuzz_gpu_bwti.main(1000, 8, ctx, variant=1) == 0             # walkers park 16 bytes at most: second chases
    assert fuzz_gpu_bwti.main(1000, 9, ctx, variant=2) == 0             # the forward chase over the scattered jump table (A/B kernel)
    assert fuzz_gpu_bwti.main(1200, 7, ctx, minimal=True) == 0          # decode_minimal, src/bwt/mod.rs:298-315


@pytest.mark.gpu
def test_boundary_sizes_every_codec(ctx):
    """Sizes at the powers of two and at the kernels' wave / tile / window sizes (-1, 0, +1), four distributions, every codec
    against the oracle (benchmarks/edge_sizes.py)."""
    import edge_sizes
    assert edge_sizes.main(ctx) == 0
