"""does torch still find the device after the library has used it in this process?"""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from auncel_amd import capi
step = sys.argv[1] if len(sys.argv) > 1 else "search"
capi.lib()
if step in ("create", "search"):
    h = capi.Handle(32, 16, capi.METRIC_L2, 0)
if step == "search":
    rs = np.random.RandomState(0)
    xb = rs.randn(2000, 32).astype(np.float32); cen = xb[:16].copy()
    h.set_centroids(cen); h.add(xb)
    D, I = h.search(xb[:300], 10, 4)
import torch
try:
    print(step, "torch:", torch.cuda.is_available(), torch.cuda.device_count(), torch.zeros(4, device="cuda").sum().item())
except Exception as e:  # noqa: BLE001
    print(step, "torch FAILED:", str(e)[:100])
