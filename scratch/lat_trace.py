"""a few single-query searches on a 1M index, for a rocprofv3 --kernel-trace timeline (profiles/timeline.py)"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from auncel_amd import capi, synth
nb, d, nlist = 1_000_000, 128, 1024
xb, xq = synth.sift_like(nb, 64, d=d, nblobs=2000, sigma=35.0, seed=1234)
cen, _ = capi.kmeans(capi.METRIC_L2, xb, nlist, niter=5)
h = capi.Handle(d, nlist, capi.METRIC_L2, 0); h.set_centroids(cen); h.add(xb); h.set_queries(xq)
for i in range(30):
    t0 = time.perf_counter(); h.search_resident(i, 1, 10, 16); dt = time.perf_counter() - t0
print("last call ms", dt * 1e3, h.last_timing())
