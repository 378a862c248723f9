#!/usr/bin/env python3
"""BASELINE configs[3]: SIFT-10M-like, IVF4096,Flat, k=10, fixed nprobe, inverted lists sharded by list id over the GPUs
of one node (the reference's IndexShards over sub-indexes sharing one coarse quantizer, Auncel/IndexShards.cpp:261-311),
per-GPU partial top-k gathered and merged on rank 0 with merge_tables semantics.  Strong scaling: the batch and the
database are fixed, every rank scans only the probed lists it owns.

    python scripts/bench_shards.py                                            # 1 GPU (one shard holding every list)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_shards.py

Rank 0 prints one JSON line; the merged result of the last step is checked against the exact ground truth (recall)
and, at N > 1, the single-index result of rank 0's own full search is not available, so parity at scale rests on
tests/test_sharding_gloo.py + tests/test_gpu_parity.py::test_shards (goldens of the reference's IndexShards)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=10_000_000)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from auncel_amd import capi, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    d, nlist = 128, args.nlist
    xb_t, _, draw = bench.gen_data(torch, dev, args.nb, 0, d, 20000, bench.SIGMA, 1235)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    xq_t = draw(args.nq, g)
    cen_np, _ = capi.kmeans(capi.METRIC_L2, xb_t.cpu().numpy(), nlist, niter=25, device=local)  # the reference's IVF training
    cen_t = torch.from_numpy(cen_np).to(dev)
    gtD, _ = bench.ground_truth(torch, xb_t, xq_t[:1000], args.k)
    # every rank derives the same list assignment (exact on this integer data), then keeps the lists it owns
    cn = (cen_t * cen_t).sum(1)
    assign = torch.empty(args.nb, dtype=torch.long, device=dev)
    for i0 in range(0, args.nb, 1 << 18):
        x = xb_t[i0:i0 + (1 << 18)]
        assign[i0:i0 + x.shape[0]] = ((x * x).sum(1)[:, None] + cn[None, :] - 2 * x @ cen_t.T).argmin(1)
    assign = assign.cpu().numpy()
    sizes = np.bincount(assign, minlength=nlist)
    owner = sharding.assign_owners(sizes, world)
    mine = sharding.local_assignment(assign, owner, rank)
    xb, xq, cen = xb_t.cpu().numpy(), xq_t.cpu().numpy(), cen_t.cpu().numpy()
    del xb_t, xq_t, cen_t
    torch.cuda.empty_cache()
    h = capi.Handle(d, nlist, capi.METRIC_L2, local)
    h.set_centroids(cen)
    keep = mine >= 0
    h.add(xb[keep], xids=np.nonzero(keep)[0].astype(np.int64), precomputed_idx=mine[keep])
    del xb
    h.set_queries(xq)

    def step():
        D, I = h.search_resident(0, args.nq, args.k, args.nprobe)
        return sharding.gather_and_merge(D, I, capi.METRIC_L2, capi.merge_tables, dist if world > 1 else None)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        D, I = out
        rec = bench.recall_dist(D[:1000], gtD, args.k)
        print(json.dumps({"config": "4", "metric": "queries/s, fixed nprobe, lists sharded by list id, host merge", "value": args.nq * args.steps / el,
                          "n_gpus": world, "scaling": "strong", "nb": args.nb, "nq": args.nq, "nlist": nlist, "nprobe": args.nprobe,
                          "k": args.k, "ms_per_step": 1e3 * el / args.steps, "recall_at_k": float(rec.mean()),
                          "shard_bytes_max_over_min": float(np.bincount(owner, weights=sizes, minlength=world).max() /
                                                            max(np.bincount(owner, weights=sizes, minlength=world).min(), 1))}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
