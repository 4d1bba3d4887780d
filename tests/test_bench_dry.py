"""CPU suite: `bench.py --gpus N` really starts N ranks and its end-to-end leg (root scatter -> decode -> root gather)
moves the right bytes.  Runs the benchmark's own code on gloo with the oracle as the block codec (--dry-gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gpus", [2, 3])
def test_bench_gpus_n_spawns_n_ranks(gpus, oracle):
    env = dict(os.environ, RCX_BENCH_DRY_CODEC="_dry_codec", PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-gloo", "--nblocks", "5", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == gpus and res["steps"] == 2 and res["scaling"] == "weak"
    e = res["end_to_end"]
    assert e["verified"] is True and e["bytes_gathered"] == (gpus - 1) * 5 * 65536 and e["bytes_scattered"] > 0
    assert e["scatter_ms"] > 0 and e["gather_ms"] > 0 and "gloo" in e["transport"]


def test_bench_single_rank_dry(oracle):
    env = dict(os.environ, RCX_BENCH_DRY_CODEC="_dry_codec", PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-gloo", "--nblocks", "4", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["end_to_end"]["verified"] is True and res["end_to_end"]["bytes_gathered"] == 0
