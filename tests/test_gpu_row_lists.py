"""Threshold rounds whose selection reads the rows' marked candidates from the lists compact_rows_kernel leaves (option row_lists,
calls of >= 256 queries): rows with none, a few and more than a row's list holds (those go to the arena, or back to the mask walk
when that is full), byte codes and fp32
lists (filter + exact rescoring), both metrics -- against the pinned CPU oracle and against the mask walk (row_lists 0), bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make(rs, kind, nlist, nb, nq, d):
    if kind == "clustered":  # few candidates of a later row beat the k-th best of the first ones
        cen = rs.randint(0, 256, size=(nlist, d)).astype(np.float32)
        assign = rs.randint(0, nlist, size=nb)
        xb = np.clip(cen[assign] + rs.randint(-20, 21, size=(nb, d)), 0, 255).astype(np.float32)
        xq = np.clip(cen[rs.randint(0, nlist, size=nq)] + rs.randint(-20, 21, size=(nq, d)), 0, 255).astype(np.float32)
    elif kind == "uniform":  # no structure: every row holds dozens of candidates that pass the threshold
        cen = rs.randint(0, 256, size=(nlist, d)).astype(np.float32)
        assign = rs.randint(0, nlist, size=nb)
        xb = rs.randint(0, 256, size=(nb, d)).astype(np.float32)
        xq = rs.randint(0, 256, size=(nq, d)).astype(np.float32)
    else:  # "float": not byte-valued, the fp32 lists
        cen = rs.standard_normal((nlist, d)).astype(np.float32) * 4
        assign = rs.randint(0, nlist, size=nb)
        xb = (cen[assign] + rs.standard_normal((nb, d))).astype(np.float32)
        xq = (cen[rs.randint(0, nlist, size=nq)] + rs.standard_normal((nq, d))).astype(np.float32)
    return cen, assign, xb, xq


@pytest.mark.parametrize("arena", [None, 4096, 64])
@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("kind", ["clustered", "uniform", "float"])
def test_row_lists_equal_the_mask_walk(oracle, monkeypatch, metric, kind, arena):
    """arena: entries for the lists longer than CL_CAP (default 4 Mi); with a small one some (4096) or nearly all (64) of the long
    rows go back to the mask walk"""
    from auncel_amd import capi
    if arena is not None:
        monkeypatch.setenv("AUNCEL_AMD_CL_ARENA", str(arena))
    rs = np.random.RandomState(9300 + metric + len(kind))
    for nlist, nq, nb, d in ((32, 300, 20000, 32), (48, 700, 30000, 64)):
        cen, assign, xb, xq = make(rs, kind, nlist, nb, nq, d)
        lists = oracle.Lists(metric, cen, xb, assign)
        cd, ck = oracle.knn(metric, xq, cen, nlist)
        for k in (10, 100):
            eD, eI, est = oracle.search_preassigned(lists, xq, k, ck, cd)
            for rl in (1, 0):
                h = capi.Handle(d, nlist, metric, 0)
                h.set_centroids(cen)
                h.set_lists_from_assign(xb, assign)
                h.set_option("row_lists", rl)
                h.set_option("fixed_rounds", 2)  # a dense round over the first probes, then one threshold round over the rest
                h.stats(reset=True)
                D, I = h.search_preassigned(xq, k, ck, cd)
                assert np.array_equal(I, eI), (kind, nlist, nq, k, rl)
                assert np.array_equal(bits(D), bits(eD)), (kind, nlist, nq, k, rl)
                st = h.stats()
                assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(est), (kind, nlist, nq, k, rl)
                h.close()


def test_row_lists_with_ragged_probe_tables(oracle):
    """probe tables with holes (-1) and lists of every length from empty to several mask-word steps (> 4096 vectors)"""
    from auncel_amd import capi
    rs = np.random.RandomState(9391)
    nlist, nq, d, metric = 40, 400, 32, 1
    sizes = np.concatenate([[0, 1, 63, 64, 65, 4096, 4097, 9000], rs.randint(1, 900, size=nlist - 8)])
    assign = np.repeat(np.arange(nlist), sizes)
    nb = len(assign)
    cen = rs.randint(0, 256, size=(nlist, d)).astype(np.float32)
    xb = np.clip(cen[assign] + rs.randint(-40, 41, size=(nb, d)), 0, 255).astype(np.float32)
    xq = np.clip(cen[rs.randint(0, nlist, size=nq)] + rs.randint(-40, 41, size=(nq, d)), 0, 255).astype(np.float32)
    lists = oracle.Lists(metric, cen, xb, assign)
    cd, ck = oracle.knn(metric, xq, cen, nlist)
    ck = ck.copy()
    ck[rs.rand(*ck.shape) < 0.1] = -1
    for k in (10, 100):
        eD, eI, est = oracle.search_preassigned(lists, xq, k, ck, cd)
        h = capi.Handle(d, nlist, metric, 0)
        h.set_centroids(cen)
        h.set_lists_from_assign(xb, assign)
        h.set_option("row_lists", 1)
        h.set_option("fixed_rounds", 2)
        h.stats(reset=True)
        D, I = h.search_preassigned(xq, k, ck, cd)
        assert np.array_equal(I, eI), k
        assert np.array_equal(bits(D), bits(eD)), k
        st = h.stats()
        assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(est), k
        h.close()
