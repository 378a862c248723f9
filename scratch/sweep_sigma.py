import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from auncel_amd import capi
dev = torch.device('cuda', 0)
nb, d, nlist, nq = 10_000_000, 128, 4096, 2000
for sigma, blobs in [(float(s), 20000) for s in os.environ.get("SIGMAS", "38,40,42").split(",")]:
    xb_t, _, draw = bench.gen_data(torch, dev, nb, 0, d, blobs, sigma, 1235)
    g = torch.Generator(device=dev); g.manual_seed(5)
    xq_t = draw(nq, g)
    gtD, gtI = bench.ground_truth(torch, xb_t, xq_t, 100)
    xb, xq = xb_t.cpu().numpy(), xq_t.cpu().numpy()
    del xb_t, xq_t; torch.cuda.empty_cache()
    cen, _ = capi.kmeans(capi.METRIC_L2, xb, nlist, niter=25)  # the reference's IVF training
    h = capi.Handle(d, nlist, capi.METRIC_L2, 0); h.set_centroids(cen); h.add(xb); del xb
    h.set_queries(xq)
    for nprobe in (4, 8, 16, 32, 64, 128):
        h.search_resident(0, nq, 10, nprobe)
        t0 = time.perf_counter(); D, I = h.search_resident(0, nq, 10, nprobe); dt = time.perf_counter() - t0
        rec = bench.recall_dist(D, gtD, 10).mean()
        tm = h.last_timing()
        print(f"sigma {sigma} blobs {blobs} nprobe {nprobe}: recall@10 {rec:.4f} qps {nq/dt:.0f} scan {tm['scan_ms']:.2f}ms select {tm['select_ms']:.2f}ms coarse {tm['coarse_ms']:.2f}ms total {tm['total_ms']:.2f} GB/s {tm['scan_bytes']/1e6/max(tm['scan_ms'],1e-9):.0f}", flush=True)
    h.close()
