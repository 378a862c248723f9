"""Randomised differential tests of the adaptive (Auncel) search against the pinned CPU oracle: small indexes, many
shapes -- register / LDS heaps, query_topk on both sides of the parallel cur_num, byte and float data, both metrics,
random traces, bounds and multipliers, profile on and off.  The oracle restates IndexIVF::search_preassigned's tune
branch and is itself held to the compiled reference by test_oracle_golden.py."""
import os

import numpy as np
import pytest

# AUNCEL_TEST_SEED_OFFSET=<n>: the same 150 shapes drawn from other seeds (one-off fuzzing after kernel changes)
SEED_OFFSET = int(os.environ.get("AUNCEL_TEST_SEED_OFFSET", "0"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from auncel_amd import capi
    capi.lib()
    return capi


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make_case(seed):
    rs = np.random.RandomState(5000 + SEED_OFFSET + seed)
    nlist = int(rs.choice([64, 128, 256]))
    d = int(rs.choice([16, 32, 64]))
    metric = 1 if rs.rand() < 0.75 else 0
    K = int(rs.choice([10, 20, 100, 130])) if metric == 1 else int(rs.choice([5, 10]))
    nb = int(rs.choice([6000, 20000])) if metric == 1 else 40000
    nq = int(rs.choice([7, 40, 150]))
    kind = rs.choice(["bytes", "float"]) if metric == 1 else "unit"
    nblobs = nlist // 2
    centres = rs.rand(nblobs, d) * 160.0
    if kind == "bytes":
        def draw(n):
            return np.floor(np.clip(centres[rs.randint(0, nblobs, n)] + rs.randn(n, d) * 30.0, 0, 255)).astype(np.float32)
    elif kind == "float":
        def draw(n):
            return (centres[rs.randint(0, nblobs, n)] / 40.0 + rs.randn(n, d) * 0.8).astype(np.float32)
    else:
        def draw(n):
            x = centres[rs.randint(0, nblobs, n)] / 160.0 - 0.5 + rs.randn(n, d) * 0.15
            return (x / np.linalg.norm(x, axis=1, keepdims=True) * 0.9).astype(np.float32)
    xb, xq = draw(nb), draw(nq)
    cen = (xb[rs.choice(nb, nlist, replace=False)] + rs.randn(nlist, d) * 1e-3).astype(np.float32)  # no exact coarse ties
    ntr = 1
    while (1 << ntr) <= nlist // 8:
        ntr += 1
    traces = []
    for _ in range(ntr):
        n = int(rs.randint(3, 60))
        x = np.sort(rs.rand(n) * 25.0).astype(np.float32)
        x += np.arange(n, dtype=np.float32) * 1e-3  # strictly ascending
        traces.append((x, (0.5 + rs.rand(n) * 2.5).astype(np.float32), (rs.rand(n) * 0.5).astype(np.float32)))
    qk = int(rs.choice([1, 3, 10, min(K, 40)]))
    qk = min(qk, K)
    return dict(nlist=nlist, d=d, metric=metric, K=K, xb=xb, xq=xq, cen=cen, traces=traces, query_topk=qk, kind=str(kind),
                req=rs.choice([0.5, 0.8, 0.9, 0.95, 0.99], size=nq).astype(np.float32),
                multipler=float(rs.choice([1.0, 1.3, 2.0, 3.7])), std_m=float(rs.choice([0.0, 1.0, 2.0])),
                profile=bool(rs.rand() < 0.5))


@pytest.mark.parametrize("seed", range(150))
def test_adaptive_against_oracle(capi, oracle, monkeypatch, seed):
    c = make_case(seed)
    # both stream widths of the selection kernel (the narrow one is what 5000-query batches get)
    monkeypatch.setenv("AUNCEL_AMD_REPLAY_NLD", "16" if seed % 2 else "32")
    # every third shape with the reference's heap as the selection for every query (the default: sorted array + tie_fix)
    monkeypatch.setenv("AUNCEL_AMD_SELECT", "heap" if seed % 3 == 0 else "sorted")
    nq, K = c["xq"].shape[0], c["K"]
    _, a = oracle.knn(c["metric"], c["xb"], c["cen"], 1, nthreads=8)
    assign = a[:, 0]
    lists = oracle.Lists(c["metric"], c["cen"], c["xb"], assign)
    if c["metric"] == 0 and lists.sizes.min() < K:
        pytest.skip("the reference's inner-product rule needs lists of at least max_topk vectors")
    cd, ck = oracle.knn(c["metric"], c["xq"], c["cen"], c["nlist"], nthreads=8)
    gtD, _ = oracle.knn(c["metric"], c["xq"], c["xb"], K, nthreads=8)
    arcos = capi.arcos_table()
    tun = oracle.Tuner(oracle.interdis(c["metric"], c["cen"]), c["traces"], K, nq, arcos=arcos)
    stt = tun.struct(c["query_topk"], c["req"], c["multipler"], c["std_m"], gt_D=gtD, profile=c["profile"])
    tag = {k: v for k, v in c.items() if k in ("nlist", "d", "metric", "K", "query_topk", "kind", "multipler", "std_m", "profile")}
    try:
        eD, eI, est = oracle.search_preassigned(lists, c["xq"], K, ck, cd, tuner=stt, offset=0, nthreads=1)
        expect_error = None
    except RuntimeError as e:
        expect_error = str(e)

    h = capi.Handle(c["d"], c["nlist"], c["metric"], 0)
    h.set_centroids(c["cen"])
    h.set_lists_from_assign(c["xb"], assign)
    h.set_interdis(None)
    h.set_tuner(K, c["traces"], arcos)
    h.set_queries(c["xq"])
    my_np = np.zeros(nq, dtype=np.uint64)
    t_rec = np.zeros(nq, dtype=np.float32)
    h.stats(reset=True)
    if expect_error is not None:
        with pytest.raises(capi.EngineError):
            h.search_adaptive(0, nq, c["query_topk"], c["multipler"], c["std_m"], c["req"], my_np, t_rec, gt_D=gtD, profile=c["profile"])
        return
    D, I = h.search_adaptive(0, nq, c["query_topk"], c["multipler"], c["std_m"], c["req"], my_np, t_rec, gt_D=gtD, profile=c["profile"])
    assert np.array_equal(my_np.astype(np.int64), tun.my_nprobe.astype(np.int64)), tag
    assert np.array_equal(I, eI), tag
    assert np.array_equal(bits(D), bits(eD)), tag
    assert np.array_equal(bits(t_rec), bits(tun.t_recalls)), tag
    st = h.stats()
    assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(est), tag
