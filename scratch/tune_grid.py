import sys, time, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from auncel_amd import capi
dev = torch.device('cuda', 0)
nb, d, nlist, K, topk, ts, ses = 10_000_000, 128, 4096, 100, 10, 5000, 5000
xb_t, _, draw = bench.gen_data(torch, dev, nb, 0, d, 20000, bench.SIGMA, 1235)
g = torch.Generator(device=dev); g.manual_seed(777)
xq_t = draw(ts + ses, g)
cen_t = bench.kmeans_centroids(torch, xb_t, nlist, 4, 99)
gtD, gtI = bench.ground_truth(torch, xb_t, xq_t, K)
xb, xq, cen = xb_t.cpu().numpy(), xq_t.cpu().numpy(), cen_t.cpu().numpy()
del xb_t, xq_t, cen_t; torch.cuda.empty_cache()
h = capi.Handle(d, nlist, capi.METRIC_L2, 0); h.set_centroids(cen); h.add(xb); del xb
h.set_interdis(None); h.set_queries(xq)
ntr = 10
raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
h.train_samples(0, ts, K, gtD, ts, raw)
traces = [capi.trace_sb(r) for r in raw]
h.set_tuner(K, traces, capi.arcos_table())
req = np.full(ts + ses, 0.95, dtype=np.float32)
for std_m in (-2.0, -1.0, 0.0, 1.0, 3.0):
    for mult in (1.0, 1.25, 1.5, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0, 16.0):
        np_ = np.zeros(ts + ses, dtype=np.uint64); tr_ = np.zeros(ts + ses, dtype=np.float32)
        h.search_adaptive(ts, ses, topk, mult, std_m, req, np_, tr_)
        np_[:] = 0
        t0 = time.perf_counter(); D, I = h.search_adaptive(ts, ses, topk, mult, std_m, req, np_, tr_); dt = time.perf_counter() - t0
        rec = bench.recall_dist(D, gtD[ts:], topk).mean()
        tm = h.last_timing()
        print(f"std_m {std_m} mult {mult}: recall {rec:.4f} nprobe {np_[ts:].mean():.1f} qps {ses/dt:.0f} ms {dt*1e3:.1f} scan {tm['scan_ms']:.1f} sel {tm['select_ms']:.1f} rounds {tm['rounds']:.0f} eff {tm['slot_efficiency']:.2f}", flush=True)
        if rec > 0.975: break
