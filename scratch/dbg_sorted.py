import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/tests/golden']
import numpy as np
os.environ["AUNCEL_AMD_FIXED_ROUNDS"] = sys.argv[2] if len(sys.argv) > 2 else "1"
from auncel_amd import capi
from oracle import pyoracle as oracle
import test_gpu_random as T
seed = int(sys.argv[1])
c = T.make_case(seed)
print({k: v for k, v in c.items() if k in ("d", "nlist", "metric", "k", "nprobe", "kind")}, c["xb"].shape, c["xq"].shape)
lists = oracle.Lists(c["metric"], c["cen"], c["xb"], c["assign"])
npq = min(c["nprobe"], c["nlist"])
cd, ck = oracle.knn(c["metric"], c["xq"], c["cen"], npq)
keys = np.full((c["xq"].shape[0], c["nprobe"]), -1, dtype=np.int64)
keys[:, :npq] = ck
h = capi.Handle(c["d"], c["nlist"], c["metric"], 0)
h.set_centroids(c["cen"])
h.set_lists_from_assign(c["xb"], c["assign"])
eD, eI, est = oracle.search_preassigned(lists, c["xq"], c["k"], keys, np.zeros(keys.shape, np.float32))
for mode in ("0", "1"):
    os.environ["AUNCEL_AMD_SORTED"] = mode
    h.stats(reset=True)
    D, I = h.search_preassigned(c["xq"], c["k"], keys)
    st = h.stats()
    bad = np.nonzero((I != eI).any(1) | (D.view(np.uint32) != eD.view(np.uint32)).any(1))[0]
    print("sorted", mode, "bad queries", len(bad), "of", len(I), "stats", [st["nlist"], st["ndis"], st["nheap_updates"]], list(est))
    for q in bad[:3]:
        j = np.nonzero((I[q] != eI[q]) | (D[q].view(np.uint32) != eD[q].view(np.uint32)))[0]
        print(" q", q, "first diff at", j[:5], "got", I[q][j[:5]], D[q][j[:5]], "want", eI[q][j[:5]], eD[q][j[:5]])
        print("   dup values in expected row:", len(eD[q]) - len(np.unique(eD[q])))
