"""find the query whose my_nprobe differs between the engine and the CPU restatement on the bench workload"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from auncel_amd import capi
from oracle import pyoracle
dev = torch.device('cuda', 0)
nb, d, nlist, K, topk, ts, ses = 10_000_000, 128, 4096, 100, 10, 5000, 5000
xb_t, _, draw = bench.gen_data(torch, dev, nb, 0, d, 20000, bench.SIGMA, 1235)
g = torch.Generator(device=dev); g.manual_seed(777)
xq_t = draw(ts + ses, g)
gtD, gtI = bench.ground_truth(torch, xb_t, xq_t, K)
xb, xq = xb_t.cpu().numpy(), xq_t.cpu().numpy()
del xb_t, xq_t; torch.cuda.empty_cache()
cen, _ = capi.kmeans(capi.METRIC_L2, xb, nlist, niter=25, coarse_mode=0, device=0)
h = capi.Handle(d, nlist, capi.METRIC_L2, 0); h.set_centroids(cen); h.add(xb); del xb
h.set_interdis(None); h.set_queries(xq)
ntr = 0
while (1 << ntr) <= nlist // 8: ntr += 1
tfit = 4000
raw = [np.full((tfit * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
h.train_samples(0, tfit, K, gtD, tfit, raw)
traces = [capi.trace_sb(r) for r in raw]
h.set_tuner(K, traces, capi.arcos_table())
req = np.full(ts + ses, 0.95, dtype=np.float32)
mult, sm = 1.0, 0.5
def gpu(tag):
    np_ = np.zeros(ts + ses, dtype=np.uint64); tr_ = np.zeros(ts + ses, dtype=np.float32)
    D, I = h.search_adaptive(ts, ses, topk, mult, sm, req, np_, tr_)
    return D, I, np_
D, I, np_gpu = gpu("replay")
os.environ["AUNCEL_AMD_LANES"] = "1"
D2, I2, np_lanes = gpu("lanes")
os.environ["AUNCEL_AMD_LANES"] = "0"
codes, ids, off = [], [], np.zeros(nlist + 1, dtype=np.uintp)
for l in range(nlist):
    c, i_ = h.get_list(l); codes.append(c); ids.append(i_); off[l + 1] = off[l] + len(i_)
lists = pyoracle.Lists.__new__(pyoracle.Lists)
lists.metric, lists.centroids, lists.nlist, lists.d = pyoracle.METRIC_L2, cen, nlist, d
lists.off, lists.codes, lists.ids = off, np.concatenate(codes), np.concatenate(ids)
lists.struct = pyoracle.OrcIndex(lists.metric, d, nlist, pyoracle._s(lists.off), pyoracle._f(lists.codes), pyoracle._i(lists.ids))
xs = xq[ts:]
tun = pyoracle.Tuner(h.get_interdis(), traces, K, ts + ses, arcos=capi.arcos_table())
stt = tun.struct(topk, req, mult, sm)
cd, ck = pyoracle.knn(pyoracle.METRIC_L2, xs, cen, nlist, nthreads=16)
oD, oI, _ = pyoracle.search_preassigned(lists, xs, K, ck, cd, tuner=stt, offset=ts, nthreads=16)
cpu = tun.my_nprobe[ts:].astype(np.uint64)
bad = np.nonzero(cpu != np_gpu[ts:])[0]
print("replay vs cpu: differing", bad, "gpu", np_gpu[ts:][bad], "cpu", cpu[bad], "lanes", np_lanes[ts:][bad])
print("lanes vs cpu differing:", np.nonzero(cpu != np_lanes[ts:])[0])
for q in bad:
    print("query", q, "D equal", np.array_equal(D[q], oD[q]), "I equal", np.array_equal(I[q], oI[q]))
    # single-query runs: batch of one goes through the same kernels with one active wave
    for lanes in ("0", "1"):
        os.environ["AUNCEL_AMD_LANES"] = lanes
        np1 = np.zeros(ts + ses, dtype=np.uint64); tr1 = np.zeros(ts + ses, dtype=np.float32)
        h.search_adaptive(ts + int(q), 1, topk, mult, sm, req, np1, tr1)
        print("  alone, lanes", lanes, "my_nprobe", np1[ts + int(q)])
    for m2 in (1.0,):
        for rg in ("12", "3.5", "2"):
            os.environ["AUNCEL_AMD_LANES"] = "0"
            # different round schedules change where the rule's cache is reset
            print("  (schedule env is read once per process; skipped)") if False else None
