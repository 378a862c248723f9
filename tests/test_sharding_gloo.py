"""N > 1 plumbing on CPU: two gloo ranks each search their list-id shard (the CPU oracle stands in
for the per-GPU engine here -- the engine itself is covered by test_gpu_parity.test_shards_by_list_id),
rank 0 gathers and merges with the product's host merge and must reproduce the reference's
IndexShards output."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import torch.distributed as dist
    from util import load_case
    from auncel_amd import capi, sharding
    from oracle import pyoracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case, gold = load_case(name)
        owner = np.arange(case["nlist"]) % world  # the golden shards are owner(l) = l % nshard
        # the coarse ranking is computed once: rank r holds the key rows of queries [r n / N, (r + 1) n / N) only and
        # the ranks all-gather them (ragged shares: 7 queries over 2 ranks would be 3 + 4)
        nq = case["xq"].shape[0]
        counts = [(r + 1) * nq // world - r * nq // world for r in range(world)]
        q0 = rank * nq // world
        keys = sharding.allgather_rows(gold["coarse_keys_sse"][q0:q0 + counts[rank]], counts, dist)
        cdis = sharding.allgather_rows(gold["coarse_dis_sse"][q0:q0 + counts[rank]], counts, dist)
        ok = np.array_equal(keys, gold["coarse_keys_sse"]) and np.array_equal(cdis.view(np.uint32), gold["coarse_dis_sse"].view(np.uint32))
        for k in case["ks"]:
            la = sharding.local_assignment(gold["assign"], owner, rank)
            sub = pyoracle.Lists(case["metric"], case["centroids"], case["xb"], la)
            D, I, _ = pyoracle.search_preassigned(sub, case["xq"], int(k), keys, cdis)
            out = sharding.gather_and_merge(D, I, case["metric"], capi.merge_tables, dist)
            if rank == 0:
                ok &= np.array_equal(out[1], gold[f"I_shards_k{k}"]) and np.array_equal(out[0].view(np.uint32), gold[f"D_shards_k{k}"].view(np.uint32))
            else:
                ok &= out is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


class _OracleHandle:
    """stands in for capi.Handle in sharding.run_pipelined: tickets run on a small thread pool, the CPU oracle does the work"""

    def __init__(self, pyoracle, case, gold, sub, k):
        from concurrent.futures import ThreadPoolExecutor
        self.orc, self.case, self.gold, self.sub, self.k = pyoracle, case, gold, sub, k
        self.pool, self.jobs, self.next = ThreadPoolExecutor(3), {}, 1

    def _submit(self, fn):
        t, self.next = self.next, self.next + 1
        self.jobs[t] = self.pool.submit(fn)
        return t

    def submit_coarse_resident(self, start, n, nprobe, mode=0, want_dis=False):
        assert nprobe == self.case["nprobe"]
        return self._submit(lambda: (None, self.gold["coarse_keys_sse"][start:start + n].copy()))

    def submit_search_resident_preassigned(self, start, n, k, keys, out=None):
        def run():
            # (IVF-Flat ignores the coarse distances: IndexIVFFlat.cpp:106)
            D, I, _ = self.orc.search_preassigned(self.sub, self.case["xq"][start:start + n], int(k), keys, self.gold["coarse_dis_sse"][start:start + n])
            out[0][:], out[1][:] = D, I
            return out
        return self._submit(run)

    def wait(self, t):
        a, b = self.jobs.pop(t).result()
        tm = {"coarse_ms": 0.0, "scan_ms": 0.0, "select_ms": 0.0, "scan_launches": 1.0, "scan_min_bytes": 0.0}
        return a, b, tm, {}


def _worker_pipelined(rank, world, port, name, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import torch.distributed as dist
    from util import load_case
    from auncel_amd import capi, sharding
    from oracle import pyoracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case, gold = load_case(name)
        owner = np.arange(case["nlist"]) % world
        nq = case["xq"].shape[0]
        counts = [(r + 1) * nq // world - r * nq // world for r in range(world)]
        k = int(case["ks"][0])
        sub = pyoracle.Lists(case["metric"], case["centroids"], case["xb"], sharding.local_assignment(gold["assign"], owner, rank))
        h = _OracleHandle(pyoracle, case, gold, sub, k)
        ok = True
        for lag in (1, 3):
            out, acc = sharding.run_pipelined(h, case["metric"], capi.merge_tables, nq, k, case["nprobe"], counts, rank, dist, 5, lag=lag)
            if rank == 0:
                ok &= acc["steps_merged"] == 5
                ok &= np.array_equal(out[1], gold[f"I_shards_k{k}"]) and np.array_equal(out[0].view(np.uint32), gold[f"D_shards_k{k}"].view(np.uint32))
            else:
                ok &= out is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["fixed_gauss_l2_d96", "fixed_dups"])
def test_two_rank_pipelined_steps(name):
    """sharding.run_pipelined (what bench.py --mode shards times): several steps in flight, the two collectives of a step issued by one
    thread in a fixed order, the merge on its own thread -- two gloo ranks, the CPU oracle in the engine's place; every step's merged
    table is the reference's IndexShards output"""
    from auncel_amd import build
    build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 250)
    procs = [ctx.Process(target=_worker_pipelined, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


@pytest.mark.parametrize("name", ["fixed_gauss_l2_d96", "fixed_deep_ip_d96", "fixed_dups"])
def test_two_rank_shards(name):
    from auncel_amd import build
    build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_owner_balance():
    from auncel_amd import sharding
    rs = np.random.RandomState(0)
    sizes = rs.randint(0, 5000, size=4096)
    owner = sharding.assign_owners(sizes, 8)
    load = np.bincount(owner, weights=sizes, minlength=8)
    assert load.max() - load.min() <= sizes.max()
    assert np.array_equal(owner, sharding.assign_owners(sizes, 8))
    la = sharding.local_assignment(np.array([0, 5, -1, 7]), np.array([1, 0, 0, 0, 0, 1, 0, 1]), 1)
    assert list(la) == [0, 5, -1, 7]
    la = sharding.local_assignment(np.array([0, 5, -1, 7]), np.array([1, 0, 0, 0, 0, 1, 0, 1]), 0)
    assert list(la) == [-1, -1, -1, -1]
