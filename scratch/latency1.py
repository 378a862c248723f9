"""batch-1 latency of the three search entries on the cfg-2 index (eval/bound.cpp and effect_time.cpp call one search per query)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from auncel_amd import capi
dev = torch.device('cuda', 0)
nb, d, nlist, K, topk = int(os.environ.get('NB', 10_000_000)), 128, 4096, 100, 10
ts = ses = 2000
xb_t, _, draw = bench.gen_data(torch, dev, nb, 0, d, 20000, bench.SIGMA, 1235)
g = torch.Generator(device=dev); g.manual_seed(5)
xq_t = draw(ts + ses, g)
gtD, _ = bench.ground_truth(torch, xb_t, xq_t, K)
xb, xq = xb_t.cpu().numpy(), xq_t.cpu().numpy()
del xb_t, xq_t; torch.cuda.empty_cache()
cen, _ = capi.kmeans(capi.METRIC_L2, xb, nlist, niter=25)
h = capi.Handle(d, nlist, capi.METRIC_L2, 0); h.set_centroids(cen); h.add(xb); del xb
h.set_interdis(None); h.set_queries(xq)
ntr = 0
while (1 << ntr) <= nlist // 8: ntr += 1
raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
gt_all = np.zeros((ts + ses, K), dtype=np.float32); gt_all[:ts] = gtD[:ts]
h.train_samples(0, ts, K, gt_all, ts, raw)
traces = [capi.trace_sb(r) for r in raw]
h.set_tuner(K, traces, capi.arcos_table())
req = np.full(ts + ses, 0.95, dtype=np.float32)

def lat(fn, n=int(os.environ.get('CALLS', 200))):
    for i in range(10): fn(ts + i)
    t, e = np.zeros(n), np.zeros(n)
    for i in range(n):
        t0 = time.perf_counter(); fn(ts + i); t[i] = (time.perf_counter() - t0) * 1e3
        e[i] = h.last_timing()['total_ms']  # inside the C entry point (what a C / C++ caller such as eval/bound.cpp sees)
    lat.engine = e
    return t

np_all, tr_all = np.zeros(ts + ses, dtype=np.uint64), np.zeros(ts + ses, dtype=np.float32)  # (indexed by query id: a harness keeps them)

def adaptive(i):
    np_all[i] = 0; tr_all[i] = 0
    return h.search_adaptive(i, 1, topk, 1.0, 0.5, req, np_all, tr_all)

bud = np.full(ts + ses, 2.0, np.float32)
for name, fn in (("search_resident k=10 nprobe=16", lambda i: h.search_resident(i, 1, 10, 16)),
                 ("search_resident k=100 nprobe=16", lambda i: h.search_resident(i, 1, 100, 16)),
                 ("search_adaptive (bound 0.95)", adaptive),
                 ("search_timed budget 2 ms", lambda i: h.search_timed(i, 1, K, nlist, bud))):
    if os.environ.get('ONLY') and os.environ['ONLY'] not in name:
        continue
    rows0 = h.coarse_tie_rows()
    t = lat(fn)
    tm = h.last_timing()
    if "adaptive" in name:
        print(f"  ({h.coarse_tie_rows() - rows0} of the {len(t)} adaptive calls were repeated with the reference's heap order: a run of equal coarse distances below 2 my_nprobe + 14)")
    print(f"{name}: median {np.median(t):.3f} ms p90 {np.percentile(t, 90):.3f} min {t.min():.3f} through ctypes; inside the C entry point median {np.median(lat.engine):.3f} p90 {np.percentile(lat.engine, 90):.3f} | last call kernels: coarse {tm['coarse_ms']:.3f} scan {tm['scan_ms']:.3f} select {tm['select_ms']:.3f} total {tm['total_ms']:.3f} rounds {tm['rounds']:.0f}", flush=True)
if os.environ.get('AUNCEL_AMD_DEBUG_TIMING'):
    adaptive(ts + 300)
