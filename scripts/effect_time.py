#!/usr/bin/env python3
"""eval/effect_time.cpp's loop on the cfg-2 workload: one time-bounded search per query (Error_sys::time_search ->
amd_ivf_search_timed), budgets cycling over a list, latency / probe depth / recall@10 per budget.

usage (GPU box): python scripts/effect_time.py [--nq 400] [--budgets 0.5,1,2,5,10,20]"""
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
    ap.add_argument("--nq", type=int, default=400)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--budgets", default="0.5,1,2,5,10,20")
    args = ap.parse_args()
    import torch
    from auncel_amd import capi
    dev = torch.device("cuda", 0)
    d = 128
    xb_t, _, draw = bench.gen_data(torch, dev, args.nb, 0, d, 20000, bench.SIGMA, 1235)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    xq_t = draw(args.nq, g)
    gtD, _ = bench.ground_truth(torch, xb_t, xq_t, args.k)
    xb, xq = xb_t.cpu().numpy(), xq_t.cpu().numpy()
    del xb_t, xq_t
    torch.cuda.empty_cache()
    cen, _ = capi.kmeans(capi.METRIC_L2, xb, args.nlist, niter=25)
    h = capi.Handle(d, args.nlist, capi.METRIC_L2, 0)
    h.set_centroids(cen)
    h.add(xb)
    del xb
    h.set_queries(xq)
    bl = [float(b) for b in args.budgets.split(",")]
    budgets = np.array([bl[i % len(bl)] for i in range(args.nq)], np.float32)
    for i in range(8):
        h.search_timed(i, 1, args.k, args.nlist, budgets)  # warm-up
    lat, used, rec = np.zeros(args.nq), np.zeros(args.nq), np.zeros(args.nq)
    for i in range(args.nq):
        t0 = time.perf_counter()
        D, I, u = h.search_timed(i, 1, args.k, args.nlist, budgets)
        lat[i] = (time.perf_counter() - t0) * 1e3
        used[i] = u[0]
        rec[i] = bench.recall_dist(D, gtD[i:i + 1], 10)[0]
    rows = []
    for b in bl:
        m = budgets == np.float32(b)
        rows.append({"budget_ms": b, "queries": int(m.sum()), "latency_ms_mean": float(lat[m].mean()), "latency_ms_p99": float(np.percentile(lat[m], 99)),
                     "within_budget": float((lat[m] <= b).mean()), "nprobe_mean": float(used[m].mean()), "recall_at_10": float(rec[m].mean())})
        print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
