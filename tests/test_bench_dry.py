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
    # BASELINE config 5's sharding: ONE stream's RCXQ container split by block ranges, scattered, decoded per rank, gathered ==
    # the source; and raw ranges scattered, encoded per rank, the containers gathered and joined == what one device writes
    c5 = [o for o in res["other_configs"] if o.get("config") == 5][0]
    assert c5["n_gpus"] == gpus and c5["sharded_container_verified"] is True and c5["joined_equals_single_device"] is True
    assert sum(c5["block_ranges"]) == c5["blocks"] == 11 and len(c5["block_ranges"]) == gpus and min(c5["block_ranges"]) >= 1


@pytest.mark.parametrize("gpus", [2, 3])
def test_bench_line_survives_a_rank_failing_inside_the_scatter(gpus, oracle):
    """A side leg that dies on ONE rank in the middle of a collective exchange (here: rank 1 raises inside the end-to-end leg's
    scatter, its peers are left waiting) must not cost the headline: rank 0 still prints exactly one line, with the value,
    and the failed leg reported as {"error": ...}."""
    env = dict(os.environ, RCX_BENCH_DRY_CODEC="_dry_codec_faulty", RCX_DRY_FAULT_RANK="1", RCX_BENCH_PG_TIMEOUT="6", RCX_BENCH_SIDE_TIMEOUT="120",
               PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-gloo", "--nblocks", "5", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-3000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == gpus and res["value"] > 0 and res["ms_per_step"] > 0 and res["roofline"]["frac"] > 0
    assert "error" in res["end_to_end"] and "end_to_end" in res["side_legs_failed"]
    # what came after the failed leg and needs the process group is skipped, not attempted on a group in an unknown state
    assert all("error" in o for o in res["other_configs"])


def test_bench_watchdog_prints_the_line_when_a_leg_hangs(oracle):
    """The same with the process group's own timeout far away (what RCCL looks like: a stuck collective never raises): the
    watchdog ends the legs and rank 0's line is still there."""
    env = dict(os.environ, RCX_BENCH_DRY_CODEC="_dry_codec_faulty", RCX_DRY_FAULT_RANK="1", RCX_BENCH_PG_TIMEOUT="600", RCX_BENCH_SIDE_TIMEOUT="12",
               PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-gloo", "--nblocks", "5", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-3000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and "side_legs_note" in res


def test_bench_single_rank_dry(oracle):
    env = dict(os.environ, RCX_BENCH_DRY_CODEC="_dry_codec", PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-gloo", "--nblocks", "4", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["end_to_end"]["verified"] is True and res["end_to_end"]["bytes_gathered"] == 0
    assert res["other_configs"][0]["sharded_container_verified"] is True


def test_container_split_join(oracle):
    """pipeline.split_container / join_containers: every shard is a valid container of its block range, and the shards joined
    in rank order are the stream again, byte for byte (the container one device writes)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import _dry_codec as D
    from rust_compress_amd import pipeline as P, dist, synth
    data = synth.gen("text", 9 * 2048 + 77, 3).tobytes()
    whole = D.pipe_encode(data, 2048)
    bs, parts, lens, praw, clen, p = P.parse_container(whole)
    assert bs == 2048 and parts == 16 and lens.tolist() == [2048] * 9 + [77]
    for world in (1, 2, 3, 7, 12):
        bounds = dist.partition(lens, world)
        shards = P.split_container(whole, bounds)
        assert len(shards) == world and P.join_containers(shards) == whole
        assert b"".join(D.pipe_decode(x) for x in shards) == data
    with pytest.raises(P.ContainerError):
        P.parse_container(whole[:40])
    with pytest.raises(P.ContainerError):
        P.join_containers([whole, D.pipe_encode(data[:100], 1024)])
