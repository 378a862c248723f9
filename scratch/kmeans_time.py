"""time amd_ivf_kmeans on an IVF4096-sized problem: 1M x 128 training points (256 per centroid), 10 iterations"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from auncel_amd import capi
rs = np.random.RandomState(3)
cen = rs.rand(2000, 128).astype(np.float32) * 160
x = np.floor(np.clip(cen[rs.randint(0, 2000, 1048576)] + rs.randn(1048576, 128).astype(np.float32) * 35, 0, 255)).astype(np.float32)
for mode in (0, 1):
    t0 = time.perf_counter(); c, obj = capi.kmeans(capi.METRIC_L2, x, 4096, niter=10, coarse_mode=mode); dt = time.perf_counter() - t0
    print(f"coarse_mode {mode}: {dt:.2f}s for 10 iterations, objective {obj[0]:.4g} -> {obj[-1]:.4g}", flush=True)
