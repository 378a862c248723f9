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

    def two_ranks(attempt):
        port = 29600 + (os.getpid() + 37 * attempt) % 300
        return _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                     "--master-port", str(port), "bench.py", "--mode", "shards", "--gpus", "2"] + SMALL,
                    {"BENCH_DIST_BACKEND": "gloo", "BENCH_DEVICE": "0"})

    two = two_ranks(0)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["distances_sha256"] == two["config"]["distances_sha256"]
    assert one["config"]["recall_at_k_mean"] == two["config"]["recall_at_k_mean"] > 0.5
    assert two["config"]["shard_bytes_max_over_min"] < 1.2
    for j in (one, two):
        assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] <= 1.0
        rows = j["config"]["per_rank_ms_per_step"]["rows"]
        assert len(rows) == j["n_gpus"] and all(len(r) == 4 for r in rows)
    # the coarse ranking is computed once: each of the two ranks ranks its half of the batch (counted in queries; how long that
    # takes is bench.py's business, not a pass / fail matter)
    assert one["config"]["coarse_queries_per_rank"] == [2000] and two["config"]["coarse_queries_per_rank"] == [1000, 1000]


@pytest.mark.gpu
def test_gpus_n_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (before touching the GPU), records how many
    ranks the process group saw, and gives the result of the launcher-started run: the merged distances of two list-id shards
    equal the single index's.  (Rehearsal on one device: gloo, both ranks on GPU 0.)"""
    one = _run([sys.executable, "bench.py", "--mode", "shards", "--gpus", "1"] + SMALL, {})
    two = _run([sys.executable, "bench.py", "--mode", "shards", "--gpus", "2"] + SMALL, {"BENCH_DIST_BACKEND": "gloo", "BENCH_DEVICE": "0"})
    assert two["n_gpus"] == 2 and two["config"]["ranks_seen"] == 2 and two["config"]["collective_backend"] == "gloo"
    assert one["config"]["ranks_seen"] == 1 and one["config"]["single_index_distances_sha256"] is None
    # rank 0 searched the undivided index as well: the shards' merged distances are that result (else the run exits non-zero)
    assert two["config"]["shards_equal_single_index"] is True and two["config"]["single_index_distances_sha256"] == two["config"]["distances_sha256"]
    assert one["config"]["distances_sha256"] == two["config"]["distances_sha256"]


def test_gpus_n_refuses_to_degrade():
    """more GPUs asked for than the node shows: an error before anything is measured, never a one-GPU run that says n_gpus 1"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_DEVICE")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "64"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and '{"metric"' not in r.stdout
    # a launcher whose world size differs from --gpus is refused as well
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1"], cwd=ROOT, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29433", BENCH_DIST_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0


@pytest.mark.gpu
def test_default_mode_two_ranks_records_both_splits(tmp_path):
    """the command the driver issues for N > 1 (no --mode): replicas of the adaptive search, and behind them the list-id shards of
    north_star / BASELINE config 4 as a `shards` block of the same line; query slices rotate; real-data files are taken when given"""
    small = ["--nb", "200000", "--nlist", "1024", "--train", "500", "--test", "500", "--blobs", "500", "--steps", "4", "--warmup", "1",
             "--no-cpu", "--no-legs", "--kmeans", "torch", "--in-flight", "2"]
    port = 29900 + os.getpid() % 90
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2"] + small, {"BENCH_DIST_BACKEND": "gloo", "BENCH_DEVICE": "0"})
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["data"] == "synthetic"
    assert two["config"]["query_slices"] == 4 and "round_hint" in two["config"]
    assert len(two["config"]["per_rank_value"]) == 2 and two["config"]["ranks_seen"] == 2 and two["config"]["collective_backend"] == "gloo"
    sh = two["shards"]
    assert sh["n_gpus"] == 2 and sh["scaling"] == "strong" and sh["value"] > 0 and len(sh["distances_sha256"]) == 64
    assert sh["shards_equal_single_index"] is True
    assert len(sh["per_rank_ms_per_step"]["rows"]) == 2 and sh["roofline"]["bound"] == "hbm"


@pytest.mark.gpu
def test_bench_takes_real_data_files(tmp_path):
    """--data DIR: base / query / ground-truth files in the harness's layouts go through the same pipeline"""
    import numpy as np
    rs = np.random.RandomState(11)
    cen = rs.randint(0, 160, size=(200, 32))
    xb = np.clip(cen[rs.randint(0, 200, 60000)] + rs.randn(60000, 32) * 20, 0, 255).astype(np.uint8)
    xq = np.clip(cen[rs.randint(0, 200, 1200)] + rs.randn(1200, 32) * 20, 0, 255).astype(np.uint8)
    for name, x in (("sift_base.u8bin", xb), ("sift_query.u8bin", xq)):
        with open(tmp_path / name, "wb") as f:
            np.array(x.shape, dtype=np.int32).tofile(f)
            x.tofile(f)
    # exact ground truth ids (.ivecs) for the first 300 queries
    xbf, xqf = xb.astype(np.float32), xq[:300].astype(np.float32)
    dist = (xqf ** 2).sum(1)[:, None] + (xbf ** 2).sum(1)[None, :] - 2 * xqf @ xbf.T
    gt = np.argsort(dist, axis=1, kind="stable")[:, :100].astype(np.int32)
    rows = np.empty((300, 101), dtype=np.int32)
    rows[:, 0] = 100
    rows[:, 1:] = gt
    rows.tofile(tmp_path / "sift_groundtruth.ivecs")
    j = _run([sys.executable, "bench.py", "--data", str(tmp_path), "--nlist", "1024", "--train", "500", "--test", "300", "--steps", "3",
              "--warmup", "1", "--no-cpu", "--no-legs", "--kmeans", "torch", "--in-flight", "1"], {})
    assert j["data"].startswith("real: sift_base.u8bin") and j["dtype"] == "u8"
    assert j["config"]["nb"] == 60000 and j["config"]["query_slices"] == 2
    assert j["config"]["ground_truth_file"]["agrees_with_brute_force"] is True
