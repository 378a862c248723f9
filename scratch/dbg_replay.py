"""replay (ordered selection) cost probe: per-wave cycles / heap updates for fixed nprobe, k=100"""
import sys, time, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from auncel_amd import capi
dev = torch.device('cuda', 0)
nb, d, nlist, nq = int(os.environ.get('NB', 10_000_000)), 128, 4096, 5000
xb_t, _, draw = bench.gen_data(torch, dev, nb, 0, d, 20000, bench.SIGMA, 1235)
g = torch.Generator(device=dev); g.manual_seed(5)
xq_t = draw(nq, g)
xb, xq = xb_t.cpu().numpy(), xq_t.cpu().numpy()
del xb_t, xq_t; torch.cuda.empty_cache()
cen, _ = capi.kmeans(capi.METRIC_L2, xb, nlist, niter=25)  # the reference's IVF training
h = capi.Handle(d, nlist, capi.METRIC_L2, 0); h.set_centroids(cen); h.add(xb); del xb
h.set_queries(xq)
for k, nprobe in ((100, 12), (100, 32), (10, 32)):
    for _ in range(2):
        t0 = time.perf_counter(); D, I = h.search_resident(0, nq, k, nprobe); dt = time.perf_counter() - t0
    tm = h.last_timing()
    print(f"k {k} nprobe {nprobe}: wall {dt*1e3:.2f}ms scan {tm['scan_ms']:.2f} select {tm['select_ms']:.2f}", flush=True)
