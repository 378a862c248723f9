"""one fixed-nprobe search of the GIST-like config under rocprofv3: what the kernels of a search take"""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import numpy as np, torch
import bench_configs as bc
from auncel_amd import capi
dev = torch.device("cuda", 0)
kind = sys.argv[1] if len(sys.argv) > 1 else "gist"
nb, nq, nlist, k, nprobe, metric = (1_000_000, 10000, 4096, 10, 32, capi.METRIC_L2) if kind == "gist" else (10_000_000, 10000, 4096, 100, 16, capi.METRIC_IP)
xb_t, xq_t = bc.gen(torch, dev, kind, nb, nq)
xb, xq = xb_t.cpu().numpy(), xq_t.cpu().numpy()
del xb_t, xq_t
cen, _ = capi.kmeans(metric, xb, nlist, niter=10)
h = capi.Handle(xb.shape[1], nlist, metric, 0)
h.set_centroids(cen); h.add(xb); h.set_queries(xq)
sz = np.array([h.list_size(l) for l in range(nlist)])
print("MARK lists: mean %.0f min %d max %d; chunks of 256: %d (ideal %.0f); chunks with <= 128 vectors: %d" % (sz.mean(), sz.min(), sz.max(), np.ceil(sz / 256).sum(), sz.sum() / 256, ((sz % 256 > 0) & (sz % 256 <= 128)).sum()))
for _ in range(3):
    t0 = time.perf_counter(); h.search_resident(0, nq, k, nprobe); dt = time.perf_counter() - t0
print("MARK search ms", dt * 1e3, h.last_timing())
