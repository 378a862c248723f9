"""perf probe: 10M x 128 IVF4096, fixed nprobe (k=10 and k=100) + timing breakdown"""
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
for k in (10, 100):
    for nprobe in (8, 32, 64):
        h.search_resident(0, nq, k, nprobe)
        best = None
        for _ in range(3):
            t0 = time.perf_counter(); D, I = h.search_resident(0, nq, k, nprobe); dt = time.perf_counter() - t0
            tm = h.last_timing()
            if best is None or dt < best[0]: best = (dt, tm)
        dt, tm = best
        print(f"k {k} nprobe {nprobe}: qps {nq/dt:.0f} wall {dt*1e3:.2f}ms scan {tm['scan_ms']:.2f} select {tm['select_ms']:.2f} coarse {tm['coarse_ms']:.2f} "
              f"slot_eff {tm['slot_efficiency']:.3f} scan GB/s {tm['scan_bytes']/1e6/max(tm['scan_ms'],1e-9):.0f} Gdist/s {tm['scan_bytes']/512/1e6/max(tm['scan_ms'],1e-9):.1f}", flush=True)
