"""Threshold rounds of the byte-code scan with up to 64 queries per item (scan_mfma_pair_kernel): lists probed by 1 .. 200 queries
(items of one and of two query blocks, the second one full, ragged or a single query), ragged chunks, every K-step count, both
metrics -- against the pinned CPU oracle, and against the one-block kernels (option scan_pipelined 3 and 0), bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("d", [24, 64, 100, 128])
def test_two_query_blocks_per_item(oracle, metric, d):
    from auncel_amd import capi
    rs = np.random.RandomState(8100 + d + metric)
    for nlist, nq, nb in ((4, 70, 3000), (8, 200, 5000), (16, 513, 6000), (6, 33, 900)):
        cen = rs.randint(0, 256, size=(nlist, d)).astype(np.float32)
        assign = rs.randint(0, nlist, size=nb)
        assign[: nb // 3] = 0  # a long list (several chunks) that most queries probe
        xb = np.clip(cen[assign] + rs.randint(-30, 31, size=(nb, d)), 0, 255).astype(np.float32)
        xq = np.clip(cen[rs.randint(0, nlist, size=nq)] + rs.randint(-30, 31, size=(nq, d)), 0, 255).astype(np.float32)
        lists = oracle.Lists(metric, cen, xb, assign)
        nprobe = nlist  # >= 16 probes or every list: a dense round over the first probes, then one threshold round
        cd, ck = oracle.knn(metric, xq, cen, nprobe)
        for k in (10, 100):
            eD, eI, est = oracle.search_preassigned(lists, xq, k, ck, cd)
            got = []
            for pipe in (7, 3, 0):
                h = capi.Handle(d, nlist, metric, 0)
                h.set_centroids(cen)
                h.set_lists_from_assign(xb, assign)
                h.set_option("scan_pipelined", pipe)
                h.set_option("fixed_rounds", 2)  # (dense + threshold round whatever nprobe is)
                h.stats(reset=True)
                D, I = h.search_preassigned(xq, k, ck, cd)
                assert h.scan_arith() == 2
                assert np.array_equal(I, eI), (nlist, nq, k, pipe)
                assert np.array_equal(bits(D), bits(eD)), (nlist, nq, k, pipe)
                st = h.stats()
                assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(est), (nlist, nq, k, pipe)
                h.close()
