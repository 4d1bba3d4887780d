"""world_size-N gloo worker: scatter compressed blocks from rank 0, decode the local range with the ORACLE
(stand-in codec: the product has no CPU path), gather on rank 0, compare with the source."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402
from rust_compress_amd import batch as B, dist as D, synth  # noqa: E402
from rust_compress_amd import _native as N  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nblocks = 37
    raws = [synth.gen(("text", "runs", "rand")[i % 3], 1000 + 577 * i, i).tobytes() for i in range(nblocks)]
    raws[5] = b""
    base = off = lens = bounds = None
    if rank == 0:
        blobs = [O.lz4_encode_block(r) for r in raws]
        bnp, off, lens = B.pack(blobs)
        base = torch.from_numpy(bnp)
        bounds = D.partition([len(r) for r in raws], world)
        assert bounds[0] == 0 and bounds[-1] == nblocks and all(np.diff(bounds) >= 0)
    lbase, loff, llen, bounds = D.scatter_blocks(base, off, lens, bounds, root=0, device="cpu")
    a, b = int(bounds[rank]), int(bounds[rank + 1])
    assert len(loff) == b - a and len(llen) == b - a
    caps = [len(r) for r in raws[a:b]]
    total, ooff, ocap = B.layout(caps)
    out = np.zeros(total, dtype=np.uint8)
    if b > a:
        _, out_len, _, status = O.batch_run(N.LZ4_DECODE, lbase.numpy(), loff, llen, out, ooff, ocap, threads=2)
        assert not status.any() and list(out_len) == caps
    else:
        out_len = np.zeros(0, dtype=np.uint64)
    packed, lens_all = D.gather_blocks(torch.from_numpy(out), ooff, out_len, bounds, root=0)
    got = None
    if rank == 0:
        ends = np.cumsum(lens_all)
        got = [packed[int(e - l): int(e)] for e, l in zip(ends, lens_all)]
    else:
        assert packed is None
    if rank == 0:
        assert len(got) == nblocks
        for g, r in zip(got, raws):
            assert bytes(g.numpy().tobytes()) == r
        print("DIST_OK world=%d bounds=%s" % (world, list(map(int, bounds))))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
