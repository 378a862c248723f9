import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from auncel_amd import capi
rs = np.random.RandomState(1)
d = 8
for nlist, nq in ((4096, 1), (4096, 64), (4096, 1024), (4000, 1), (1024, 1), (1024, 64)):
    cen = rs.randint(0, 6, size=(nlist, d)).astype(np.float32)
    xq = rs.randint(0, 6, size=(nq, d)).astype(np.float32)
    for mode in ("id", "heap"):
        os.environ["AUNCEL_AMD_COARSE_TIES"] = mode
        h = capi.Handle(d, nlist, 1, 0)
        h.set_centroids(cen)
        for rep in range(3):
            h.coarse(xq, nlist, mode=0)
            t = h.last_timing()
        print(nlist, nq, mode, "coarse_ms", round(t["coarse_ms"], 3), "total", round(t["total_ms"], 3), "rows", h.coarse_tie_rows())
