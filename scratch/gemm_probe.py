"""coarse GEMM mode on the bench shape (5000 x 4096 x 128) and on d = 960: kernel time comes from rocprofv3 --stats"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from auncel_amd import capi
rs = np.random.RandomState(0)
for d, nlist, nq in ((128, 4096, 5000), (960, 4096, 5000), (96, 4096, 5000)):
    cen = rs.randn(nlist, d).astype(np.float32)
    xq = rs.randn(nq, d).astype(np.float32)
    h = capi.Handle(d, nlist, 1, 0)
    h.set_centroids(cen)
    for rep in range(5):
        D, I = h.coarse(xq, 16, mode=1)
    D0, I0 = h.coarse(xq, 16, mode=0)
    print(d, "ids equal to the exact mode on", float((I == I0).mean()), "max rel dist diff", float(np.abs(D - D0).max() / np.abs(D0).max()), "coarse_ms", h.last_timing()["coarse_ms"])
