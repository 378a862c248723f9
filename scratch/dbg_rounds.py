"""round-by-round view of the adaptive search on the bench workload (AUNCEL_AMD_DEBUG_TIMING=1)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from auncel_amd import capi
dev = torch.device('cuda', 0)
nb, d, nlist, K, topk, ts, ses = int(os.environ.get('NB', 10_000_000)), 128, 4096, 100, 10, 5000, 5000
xb_t, _, draw = bench.gen_data(torch, dev, nb, 0, d, 20000, bench.SIGMA, 1235)
g = torch.Generator(device=dev); g.manual_seed(777)
xq_t = draw(ts + ses, g)
cen_t = bench.kmeans_centroids(torch, xb_t, nlist, 4, 99)
gtD, gtI = bench.ground_truth(torch, xb_t, xq_t, K)
xb, xq, cen = xb_t.cpu().numpy(), xq_t.cpu().numpy(), cen_t.cpu().numpy()
del xb_t, xq_t; torch.cuda.empty_cache()
h = capi.Handle(d, nlist, capi.METRIC_L2, 0); h.set_centroids(cen); h.add(xb); del xb
h.set_interdis(None); h.set_queries(xq)
ntr = 0
while (1 << ntr) <= nlist // 8: ntr += 1
raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
h.train_samples(0, ts, K, gtD, ts, raw)
traces = [capi.trace_sb(r) for r in raw]
h.set_tuner(K, traces, capi.arcos_table())
req = np.full(ts + ses, 0.95, dtype=np.float32)
def run(tag):
    for it in range(3):
        np_ = np.zeros(ts + ses, dtype=np.uint64); tr_ = np.zeros(ts + ses, dtype=np.float32)
        t0 = time.perf_counter()
        D, I = h.search_adaptive(ts, ses, topk, 1.0, 0.5, req, np_, tr_)
        dt = time.perf_counter() - t0
        print(tag, f"wall {dt*1e3:.2f} ms", h.last_timing(), "nprobe mean/max", np_[ts:].mean(), np_[ts:].max(), flush=True)
    return D, I, np_
run("default")
