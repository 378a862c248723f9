"""does the recall estimate a query has after round 0 (12 probes) predict how many probes it needs in the end?
The estimate at stage 12 is >= r exactly when the query fires by stage 12 under the bound r (up to non-monotonicity), so a few
adaptive searches with lower bounds bucket the queries without touching a kernel."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from auncel_amd import capi
dev = torch.device('cuda', 0)
nb, d, nlist, K, topk = 10_000_000, 128, 4096, 100, 10
ts = ses = 5000
xb_t, _, draw = bench.gen_data(torch, dev, nb, 0, d, 20000, bench.SIGMA, 1235)
g = torch.Generator(device=dev); g.manual_seed(1235 + 17)
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
h.set_tuner(K, [capi.trace_sb(r) for r in raw], capi.arcos_table())
res = {}
for r in (0.3, 0.5, 0.7, 0.8, 0.9, 0.95):
    req = np.full(ts + ses, r, dtype=np.float32)
    np_ = np.zeros(ts + ses, dtype=np.uint64); tr_ = np.zeros(ts + ses, dtype=np.float32)
    h.search_adaptive(ts, ses, topk, 1.0, 0.5, req, np_, tr_)
    res[r] = np_[ts:].astype(np.int64)
final = res[0.95]
late = final > 12
print("queries past round 0:", int(late.sum()), "of", ses)
prev = np.zeros(ses, bool)
for r in (0.9, 0.8, 0.7, 0.5, 0.3, 0.0):
    fired = (res[r] <= 12) if r > 0 else np.ones(ses, bool)
    sel = late & fired & ~prev
    prev |= fired
    if sel.sum():
        f = final[sel]
        print(f"estimate@12 in [{r}, next): {int(sel.sum()):5d} queries, final nprobe percentiles 25/50/75/90/99 = {np.percentile(f, [25, 50, 75, 90, 99]).tolist()}, share <= 24: {(f <= 24).mean():.2f}, <= 42: {(f <= 42).mean():.2f}")
