"""Coarse rankings longer than 128 entries when two centroids sit at exactly the same fp32 distance from a query.  The
reference ranks with a binary heap and heap-sorts it (knn_L2sqr_sse / knn_inner_product_sse, Auncel/utils.cpp:417-490,
Heap.h:88-142,295-322): the order inside such a run is what the heap's history leaves.  The engine sorts, then re-runs
that heap for the rows concerned (heap_tie_order_kernel); the oracle restates the reference's heap.  Integer data on a
coarse grid makes such runs the rule instead of the one-in-thousands exception of real descriptors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from auncel_amd import capi
    capi.lib()
    return capi


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def grid_points(rs, n, d, span):
    return rs.randint(0, span, size=(n, d)).astype(np.float32)


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("nlist,nprobe", [(300, 8), (1024, 32), (4096, 100), (4096, 128), (300, 129), (300, 300), (300, 307), (1024, 200), (1024, 1024),
                                          (4096, 4096), (4096, 549), (8192, 8192)])
@pytest.mark.parametrize("nq,mode", [(1, None), (19, None), (40, "heap")])
def test_ranking_equals_reference_heap(capi, oracle, monkeypatch, metric, nlist, nprobe, nq, mode):
    rs = np.random.RandomState(nlist * 7 + nprobe + metric)
    d = 8
    cen = grid_points(rs, nlist, d, 6)
    xq = grid_points(rs, nq, d, 6)
    if mode:
        monkeypatch.setenv("AUNCEL_AMD_COARSE_TIES", mode)
    else:
        monkeypatch.delenv("AUNCEL_AMD_COARSE_TIES", raising=False)
    h = capi.Handle(d, nlist, metric, 0)
    h.set_centroids(cen)
    before = h.coarse_tie_rows()
    D, I = h.coarse(xq, nprobe, mode=0)
    eD, eI = oracle.knn(metric, xq, cen, nprobe)  # a heap of nprobe entries, also when that is more than there are centroids
    assert np.array_equal(bits(D), bits(eD))
    assert np.array_equal(I, eI)
    assert (I[:, min(nprobe, nlist):] == -1).all()
    if nprobe > 128:  # (shorter rankings come from the reference's heap replayed for every row: nothing to re-run)
        assert h.coarse_tie_rows() - before == nq  # every row of this data holds runs of equal distances


def test_id_order_without_the_heap(capi, oracle, monkeypatch):
    """AUNCEL_AMD_COARSE_TIES=id: same distances, runs of equal distances in centroid-number order"""
    monkeypatch.setenv("AUNCEL_AMD_COARSE_TIES", "id")
    rs = np.random.RandomState(5)
    cen, xq = grid_points(rs, 500, 8, 6), grid_points(rs, 5, 8, 6)
    h = capi.Handle(8, 500, 1, 0)
    h.set_centroids(cen)
    D, I = h.coarse(xq, 500, mode=0)
    eD, eI = oracle.knn(1, xq, cen, 500)
    assert np.array_equal(bits(D), bits(eD))
    assert not np.array_equal(I, eI)
    for q in range(5):
        order = np.lexsort((np.arange(500), eD[q]))  # by distance, then centroid number
        full = ((xq[q][None, :] - cen) ** 2).sum(1)
        assert np.array_equal(I[q], np.lexsort((np.arange(500), full)))
        assert np.array_equal(np.sort(I[q]), np.sort(eI[q])) and order.shape == (500,)
    assert h.coarse_tie_rows() == 0


@pytest.mark.parametrize("seed", range(12))
def test_adaptive_search_with_coarse_ties(capi, oracle, monkeypatch, seed):
    """the whole Auncel flow on data whose coarse rankings are full of equal distances: set_online's windows and the probe
    order both read the ranking, so my_nprobe and the stats only agree if the runs are ordered as the reference's"""
    rs = np.random.RandomState(8000 + seed)
    nlist, d, K = int(rs.choice([136, 256])), 16, int(rs.choice([10, 100]))
    nq = int(rs.choice([5, 60]))
    # calls of 20 queries and more: the heap's order for every ranking ("heap"), or the whole call in centroid-number order and
    # then, again and with the heap's order, just the queries whose first run of equal distances lies within what they read ("redo")
    ties = ("redo" if seed % 2 else "heap") if nq >= 20 else "auto"
    monkeypatch.setenv("AUNCEL_AMD_COARSE_TIES", ties)
    nb = 20000
    xb, xq = grid_points(rs, nb, d, 12), grid_points(rs, nq, d, 12)
    cen = xb[rs.choice(nb, nlist, replace=False)].copy()
    ntr = 1
    while (1 << ntr) <= nlist // 8:
        ntr += 1
    traces = []
    for _ in range(ntr):
        n = int(rs.randint(3, 60))
        x = np.sort(rs.rand(n) * 25.0).astype(np.float32)
        x += np.arange(n, dtype=np.float32) * 1e-3
        traces.append((x, (0.5 + rs.rand(n) * 2.5).astype(np.float32), (rs.rand(n) * 0.5).astype(np.float32)))
    qk = min(int(rs.choice([1, 10])), K)
    req = rs.choice([0.8, 0.9, 0.95, 0.99], size=nq).astype(np.float32)
    mult, sm = float(rs.choice([1.0, 2.0])), float(rs.choice([0.0, 1.0]))
    _, a = oracle.knn(1, xb, cen, 1, nthreads=8)
    assign = a[:, 0]
    lists = oracle.Lists(1, cen, xb, assign)
    cd, ck = oracle.knn(1, xq, cen, nlist, nthreads=8)
    assert any((cd[q, 1:] == cd[q, :-1]).any() for q in range(nq))
    gtD, _ = oracle.knn(1, xq, xb, K, nthreads=8)
    arcos = capi.arcos_table()
    tun = oracle.Tuner(oracle.interdis(1, cen), traces, K, nq, arcos=arcos)
    stt = tun.struct(qk, req, mult, sm, gt_D=gtD, profile=False)
    try:
        eD, eI, est = oracle.search_preassigned(lists, xq, K, ck, cd, tuner=stt, offset=0, nthreads=1)
    except RuntimeError:
        pytest.skip("the reference throws on this draw (cosine_theorem precondition)")
    h = capi.Handle(d, nlist, 1, 0)
    h.set_centroids(cen)
    h.set_lists_from_assign(xb, assign)
    h.set_interdis(None)
    h.set_tuner(K, traces, arcos)
    h.set_queries(xq)
    my_np = np.zeros(nq, dtype=np.uint64)
    t_rec = np.zeros(nq, dtype=np.float32)
    h.stats(reset=True)
    D, I = h.search_adaptive(0, nq, qk, mult, sm, req, my_np, t_rec, gt_D=gtD, profile=False)
    assert np.array_equal(my_np.astype(np.int64), tun.my_nprobe.astype(np.int64))
    assert np.array_equal(I, eI)
    assert np.array_equal(bits(D), bits(eD))
    st = h.stats()
    assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(est)
    assert st["nq"] == nq
    assert h.coarse_tie_rows() > 0
    if ties == "redo":
        # nlist a power of two: the heap's order is applied to the pass under way (rankings patched); else a second, short call
        assert 0 < h.last_tie_redone() + h.last_tie_patched() <= 2 * nq
        if nlist & (nlist - 1):
            assert h.last_tie_patched() == 0 and h.last_tie_redone() > 0


@pytest.mark.parametrize("seed", range(10))
def test_tie_order_applied_to_the_pass_under_way(capi, oracle, monkeypatch, seed):
    """AUNCEL_AMD_COARSE_TIES=redo at nlist = 2^m: the rankings with a run of equal distances in what a query can read go through
    the reference's heap while round 0 is planned and scanned, and the heap's order is written over the ranking and the rows
    scanned so far before the first selection (tie_patch_kernel) -- runs inside round 0's rows, across its end (the planner takes
    the whole run in), behind it.  Everything the reference returns must come out of that one pass."""
    rs = np.random.RandomState(9100 + seed)
    nlist, d, K = int(rs.choice([256, 1024])), 16, int(rs.choice([10, 100]))
    nq = int(rs.choice([24, 200, 700]))
    span = int(rs.choice([14, 40]))  # coarse grid: runs everywhere; finer: a run here and there
    monkeypatch.setenv("AUNCEL_AMD_COARSE_TIES", "redo")
    if seed % 3 == 2:
        monkeypatch.setenv("AUNCEL_AMD_SELECT", "heap")  # (rows padded to 64 instead of 1024: the patch moves rows by that)
    nb = 30000
    xb, xq = grid_points(rs, nb, d, span), grid_points(rs, nq, d, span)
    cen = xb[rs.choice(nb, nlist, replace=False)].copy()
    ntr = 1
    while (1 << ntr) <= nlist // 8:
        ntr += 1
    traces = []
    for _ in range(ntr):
        n = int(rs.randint(3, 60))
        x = np.sort(rs.rand(n) * 25.0).astype(np.float32)
        x += np.arange(n, dtype=np.float32) * 1e-3
        traces.append((x, (0.5 + rs.rand(n) * 2.5).astype(np.float32), (rs.rand(n) * 0.5).astype(np.float32)))
    qk = min(int(rs.choice([1, 10])), K)
    req = rs.choice([0.8, 0.9, 0.95, 0.99], size=nq).astype(np.float32)
    mult, sm = float(rs.choice([1.0, 2.0])), float(rs.choice([0.0, 1.0]))
    _, a = oracle.knn(1, xb, cen, 1, nthreads=8)
    assign = a[:, 0]
    lists = oracle.Lists(1, cen, xb, assign)
    cd, ck = oracle.knn(1, xq, cen, nlist, nthreads=8)
    gtD, _ = oracle.knn(1, xq, xb, K, nthreads=8)
    arcos = capi.arcos_table()
    tun = oracle.Tuner(oracle.interdis(1, cen), traces, K, nq, arcos=arcos)
    stt = tun.struct(qk, req, mult, sm, gt_D=gtD, profile=False)
    try:
        eD, eI, est = oracle.search_preassigned(lists, xq, K, ck, cd, tuner=stt, offset=0, nthreads=8)
    except RuntimeError:
        pytest.skip("the reference throws on this draw (cosine_theorem precondition)")
    h = capi.Handle(d, nlist, 1, 0)
    h.set_centroids(cen)
    h.set_lists_from_assign(xb, assign)
    h.set_interdis(None)
    h.set_tuner(K, traces, arcos)
    h.set_queries(xq)
    for rep in range(2):  # (the second call finds the workspaces, hints and slots of the first)
        my_np = np.zeros(nq, dtype=np.uint64)
        t_rec = np.zeros(nq, dtype=np.float32)
        h.stats(reset=True)
        D, I = h.search_adaptive(0, nq, qk, mult, sm, req, my_np, t_rec, gt_D=gtD, profile=False)
        assert np.array_equal(my_np.astype(np.int64), tun.my_nprobe.astype(np.int64))
        assert np.array_equal(I, eI)
        assert np.array_equal(bits(D), bits(eD))
        st = h.stats()
        assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(est)
        assert st["nq"] == nq
        assert h.last_tie_patched() > 0
        # the pass covers every query that got a slot (512 of them) and read no further than what was ranked (nreal entries): what is
        # searched again is the overflow and the deep readers
        nreal = min(max(nlist // 8 + 21, int((nlist // 8) * mult) + 2) + 16, nlist)
        deep = int((2 * tun.my_nprobe.astype(np.int64) + 14 + 1 >= nreal).sum())
        assert h.last_tie_redone() <= max(0, nq - 512) + deep
