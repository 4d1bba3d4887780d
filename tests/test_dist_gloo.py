"""CPU suite: the N>1 sharding path (partition + scatter/gather of variable-length blocks) on gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

from rust_compress_amd import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_properties():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 64, 1000):
            w = rng.integers(0, 70000, n)
            b = D.partition(w, world)
            assert len(b) == world + 1 and b[0] == 0 and b[-1] == n and (np.diff(b) >= 0).all()
            if n >= 8 * world and w.sum() > 0:
                shares = [w[b[g]:b[g + 1]].sum() for g in range(world)]
                assert max(shares) <= w.sum() / world + w.max() + 1
    assert list(D.partition([1] * 8, 8)) == list(range(9))


@pytest.mark.parametrize("world", [2, 3])
def test_scatter_decode_gather_gloo(world, oracle):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world), os.path.join(ROOT, "tests", "_dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "DIST_OK world=%d" % world in p.stdout
