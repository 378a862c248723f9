"""perf probe of the fp32 tile kernel: float data (ARITH 0), cfg 3-like (d 96, IP, k 100) and cfg 5-like (d 960, L2, k 10) shapes"""
import sys, time, os, numpy as np
sys.path.insert(0, '/root/repo')
from auncel_amd import capi
rs = np.random.RandomState(1)
shapes = [("cfg3-like", 2_000_000, 96, 1024, 0, 100, 64), ("cfg5-like", 250_000, 960, 1024, 1, 10, 64), ("d128 L2", 2_000_000, 128, 1024, 1, 10, 64),
          ("cfg5-true np32", 1_000_000, 960, 4096, 1, 10, 32), ("cfg5-true np64", 1_000_000, 960, 4096, 1, 10, 64), ("cfg5-true np8 (one dense round)", 1_000_000, 960, 4096, 1, 10, 8)]
only = os.environ.get("SHAPE")
for name, nb, d, nlist, metric, k, nprobe in shapes:
    if only and only not in name: continue
    nq = 10000
    cen0 = rs.randn(nlist, d).astype(np.float32)
    xb = np.empty((nb, d), np.float32)
    for i0 in range(0, nb, 100000):
        m = min(100000, nb - i0)
        xb[i0:i0 + m] = cen0[rs.randint(0, nlist, m)] + 0.5 * rs.randn(m, d).astype(np.float32)
    xq = (cen0[rs.randint(0, nlist, nq)] + 0.5 * rs.randn(nq, d).astype(np.float32)).astype(np.float32)
    if metric == 0:
        xb /= np.linalg.norm(xb, axis=1, keepdims=True); xq /= np.linalg.norm(xq, axis=1, keepdims=True)
    h = capi.Handle(d, nlist, metric, 0)
    h.set_centroids(xb[rs.choice(nb, nlist, replace=False)].copy())
    h.add(xb); del xb
    h.set_queries(xq)
    h.search_resident(0, nq, k, nprobe)
    best = None
    for _ in range(3):
        h.stats(reset=True)
        t0 = time.perf_counter(); D, I = h.search_resident(0, nq, k, nprobe); dt = time.perf_counter() - t0
        tm = h.last_timing(); st = h.stats()
        if best is None or dt < best[0]: best = (dt, tm, st)
    dt, tm, st = best
    nd = st["ndis"]
    print(f"{name}: arith {h.scan_arith()} qps {nq/dt:.0f} wall {dt*1e3:.2f}ms scan {tm['scan_ms']:.2f} ({tm['scan_launches']:.0f} launches) select {tm['select_ms']:.2f} coarse {tm['coarse_ms']:.2f} "
          f"slot_eff {tm['slot_efficiency']:.3f} ndis {nd:.3e} Gdist/s {nd/1e6/max(tm['scan_ms'],1e-9):.1f} Telem/s {nd*d/1e9/max(tm['scan_ms'],1e-9):.2f}", flush=True)
    del h
