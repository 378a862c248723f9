"""bench.py's multi-rank flows rehearsed on one GPU: two gloo ranks sharing device 0 (BENCH_DIST_BACKEND=gloo
BENCH_DEVICE=0), launched the way the driver launches N > 1."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--nb", "200000", "--nlist", "256", "--test", "2000", "--blobs", "500", "--steps", "2", "--warmup", "1", "--no-cpu"]


def _run(cmd, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_shards_mode_two_ranks_equals_one():
    """--mode shards (BASELINE config 4): the merged distances of two list-id shards equal those of the single index"""
    one = _run([sys.executable, "bench.py", "--mode", "shards", "--gpus", "1"] + SMALL, {})
    port = 29600 + os.getpid() % 300
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--mode", "shards", "--gpus", "2"] + SMALL,
               {"BENCH_DIST_BACKEND": "gloo", "BENCH_DEVICE": "0"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["distances_sha256"] == two["config"]["distances_sha256"]
    assert one["config"]["recall_at_k_mean"] == two["config"]["recall_at_k_mean"] > 0.5
    assert two["config"]["shard_bytes_max_over_min"] < 1.2
    for j in (one, two):
        assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] <= 1.0
