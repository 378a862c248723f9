"""kernel / copy timeline of one single-query adaptive call (run under rocprofv3 --kernel-trace --memory-copy-trace)"""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from auncel_amd import capi
dev = torch.device('cuda', 0)
nb, d, nlist, K, topk, ts, ses = 2_000_000, 128, 4096, 100, 10, 1000, 200
xb_t, _, draw = bench.gen_data(torch, dev, nb, 0, d, 20000, bench.SIGMA, 1235)
g = torch.Generator(device=dev); g.manual_seed(777)
xq_t = draw(ts + ses, g)
gtD, gtI = bench.ground_truth(torch, xb_t, xq_t, K)
xb, xq = xb_t.cpu().numpy(), xq_t.cpu().numpy()
del xb_t, xq_t; torch.cuda.empty_cache()
cen, _ = capi.kmeans(capi.METRIC_L2, xb, nlist, niter=10, coarse_mode=0, device=0)
h = capi.Handle(d, nlist, capi.METRIC_L2, 0); h.set_centroids(cen); h.add(xb); del xb
h.set_interdis(None); h.set_queries(xq)
ntr = 0
while (1 << ntr) <= nlist // 8: ntr += 1
raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
h.train_samples(0, ts, K, gtD, ts, raw)
h.set_tuner(K, [capi.trace_sb(r) for r in raw], capi.arcos_table())
req = np.full(ts + ses, 0.95, dtype=np.float32)
import time
for i in range(30):
    np_ = np.zeros(ts + ses, dtype=np.uint64); tr_ = np.zeros(ts + ses, dtype=np.float32)
    t0 = time.perf_counter(); h.search_adaptive(ts + i, 1, topk, 1.0, 0.5, req, np_, tr_); dt = time.perf_counter() - t0
print("last call ms", dt * 1e3, h.last_timing())
