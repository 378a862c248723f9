import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tests/golden']
from util import load_case, traces_from_gold
from auncel_amd import capi
case, gold = load_case('auncel_sift_d32')
K, ts = case["max_topk"], case["train_num"]
h = capi.Handle(case["d"], case["nlist"], case["metric"], 0)
h.set_centroids(gold["centroids"]); h.set_lists_from_assign(case["xb"], gold["assign"])
h.set_interdis(None); h.set_queries(case["xq"])
ntr=8
raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
D, I = h.train_samples(0, ts, K, gold["gtD"], ts, raw)
for i in range(ntr):
    g = gold[f"raw_trace{i}"]
    bad = np.nonzero(raw[i].view(np.uint32) != g.view(np.uint32))
    print(i, len(bad[0]), 'col counts', np.bincount(bad[1], minlength=2))
    for r,c in list(zip(*bad))[:5]:
        print('   row',r,'col',c,'got',raw[i][r], 'want', g[r])
