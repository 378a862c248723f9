"""Exact coarse rankings of large fixed-nprobe calls through the matrix cores (coarse_pick_kernel): approximate distances to every
centroid, exact recomputation of the candidates, the reference's heap for queries in which equal distances meet.  Ids and
distances must be the exact path's -- and the pinned oracle's -- bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("kind", ["float", "bytes", "float_d100", "ties"])
def test_coarse_pick_equals_the_exact_ranking(metric, kind):
    from auncel_amd import capi
    from oracle import pyoracle
    rs = np.random.RandomState(17)
    d = 100 if kind == "float_d100" else 64
    nlist, n = 1024, 1500
    if kind in ("bytes", "ties"):
        cen = rs.randint(0, 256, size=(nlist, d)).astype(np.float32)
        xq = np.clip(cen[rs.randint(0, nlist, n)] + rs.randint(-40, 41, size=(n, d)), 0, 255).astype(np.float32)
        if kind == "ties":  # duplicated centroids: every query meets runs of exactly equal distances among its best
            cen[1::2] = cen[0::2]
    else:
        cen = rs.randn(nlist, d).astype(np.float32)
        xq = (cen[rs.randint(0, nlist, n)] + 0.7 * rs.randn(n, d)).astype(np.float32)
        if metric == 0:
            cen /= np.linalg.norm(cen, axis=1, keepdims=True)
            xq /= np.linalg.norm(xq, axis=1, keepdims=True)
    h = capi.Handle(d, nlist, metric, 0)
    h.set_centroids(cen)
    for nprobe in (1, 8, 64, 128):
        h.set_option("coarse_pick", 0)
        D0, I0 = h.coarse(xq, nprobe, mode=0)
        assert h.last_coarse_pick() == 0
        for form in (2, 1):  # approximate distances from fp16 operands (the default) / fp32 operands
            h.set_option("coarse_pick", form)
            D1, I1 = h.coarse(xq, nprobe, mode=0)
            picked = h.last_coarse_pick()
            assert np.array_equal(I0, I1) and np.array_equal(bits(D0), bits(D1)), (kind, nprobe, form)
            if kind == "ties":
                assert picked < n // 2  # (almost every ranking goes through the reference's heap)
            else:
                assert picked > 0.9 * n, (kind, nprobe, form, picked)
        oD, oI = pyoracle.knn(metric, xq[:200], cen, nprobe)
        assert np.array_equal(I1[:200], oI) and np.array_equal(bits(D1[:200]), bits(oD)), (kind, nprobe)
    # a small call keeps the exact path (latency), and so does a ranking that is read almost whole
    h.coarse(xq[:100], 8, mode=0)
    assert h.last_coarse_pick() == 0


def test_coarse_pick_without_a_usable_fp16_scale():
    """queries of which one is far outside what one fp16 scale can hold (2^40 beside 1): every ranking comes back flagged and is
    recomputed exactly; magnitudes far from 1 on both sides are scaled"""
    from auncel_amd import capi
    rs = np.random.RandomState(23)
    d, nlist, n = 64, 1024, 600
    cen = rs.randn(nlist, d).astype(np.float32)
    xq = (cen[rs.randint(0, nlist, n)] + 0.7 * rs.randn(n, d)).astype(np.float32)
    for cs, qs, one in ((1.0, 1.0, 2.0 ** 40), (3e4, 1e-3, 1.0)):
        c2, x2 = (cen * np.float32(cs)).astype(np.float32), (xq * np.float32(qs)).astype(np.float32)
        x2[7] *= np.float32(one)
        h = capi.Handle(d, nlist, 1, 0)
        h.set_centroids(c2)
        h.set_option("coarse_pick", 0)
        D0, I0 = h.coarse(x2, 16, mode=0)
        h.set_option("coarse_pick", 2)
        D1, I1 = h.coarse(x2, 16, mode=0)
        assert np.array_equal(I0, I1) and np.array_equal(bits(D0), bits(D1)), (cs, qs, one)
        assert (h.last_coarse_pick() == 0) == (one != 1.0)
        h.close()
