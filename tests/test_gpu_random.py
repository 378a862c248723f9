"""Randomised differential tests of the HIP path against the pinned CPU oracle (which restates the reference and is held
to the compiled reference's goldens by test_oracle_golden.py): odd dimensions, ragged and empty lists, k below and above
the register-heap limit, nprobe beyond nlist, heavy ties, byte-valued and float data, both metrics, one and two rounds,
max_codes, store_pairs, range search.  Small cases, many shapes."""
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
    rs = np.random.RandomState(1000 + SEED_OFFSET + seed)
    d = int(rs.choice([4, 8, 12, 16, 30, 32, 48, 64, 100, 128]))
    nlist = int(rs.choice([1, 2, 7, 16, 33, 64]))
    nb = int(rs.choice([50, 300, 2000, 9000]))
    nq = int(rs.choice([1, 3, 19, 64, 130]))
    metric = int(rs.choice([0, 1]))
    kind = rs.choice(["bytes", "smallint", "float", "dups", "wideint"])
    if kind == "bytes":
        xb = rs.randint(0, 256, size=(nb, d)).astype(np.float32)
        xq = rs.randint(0, 256, size=(nq, d)).astype(np.float32)
    elif kind == "smallint":
        xb = rs.randint(-20, 21, size=(nb, d)).astype(np.float32)
        xq = rs.randint(-20, 21, size=(nq, d)).astype(np.float32)
    elif kind == "wideint":
        # signed integers whose differences pass 4096: (x - y)^2 is no longer exact in fp32, so the fused scan
        # (fma(t, t, acc)) would round differently from the reference's mul + add -- it must not be chosen here
        xb = rs.randint(-4000, 4001, size=(nb, d)).astype(np.float32)
        xq = rs.randint(-4000, 4001, size=(nq, d)).astype(np.float32)
    elif kind == "dups":
        base = rs.randint(0, 5, size=(max(nb // 20, 2), d)).astype(np.float32)
        xb = base[rs.randint(0, len(base), size=nb)]
        xq = base[rs.randint(0, len(base), size=nq)]
    else:
        xb = rs.randn(nb, d).astype(np.float32)
        xq = rs.randn(nq, d).astype(np.float32)
    cen = xb[rs.choice(nb, size=nlist, replace=nb < nlist)].copy()
    if kind == "float":
        cen += (rs.randn(*cen.shape) * 0.01).astype(np.float32)
    # ragged on purpose: some lists empty, one list large
    assign = rs.randint(0, nlist, size=nb)
    if nlist > 2:
        assign[assign == 1] = 0
    k = int(rs.choice([1, 5, 10, 64, 100, 127, 128, 200]))
    nprobe = int(rs.choice([1, 2, 5, 16, 24, nlist, nlist + 3]))
    return dict(d=d, nlist=nlist, metric=metric, xb=xb, xq=xq, cen=cen, assign=assign, k=k, nprobe=nprobe, kind=str(kind))


@pytest.mark.parametrize("seed", range(150))
def test_search_and_range_against_oracle(capi, oracle, monkeypatch, seed):
    c = make_case(seed)
    lists = oracle.Lists(c["metric"], c["cen"], c["xb"], c["assign"])
    # coarse ranking from the oracle's exact path (padded with -1 beyond nlist, as the reference's heap leaves it)
    npq = min(c["nprobe"], c["nlist"])
    cd, ck = oracle.knn(c["metric"], c["xq"], c["cen"], npq)
    keys = np.full((c["xq"].shape[0], c["nprobe"]), -1, dtype=np.int64)
    keys[:, :npq] = ck
    h = capi.Handle(c["d"], c["nlist"], c["metric"], 0)
    h.set_centroids(c["cen"])
    h.set_lists_from_assign(c["xb"], c["assign"])
    for pairs, max_codes in ((False, 0), (True, 0), (False, max(1, len(c["xb"]) // 7))):
        eD, eI, est = oracle.search_preassigned(lists, c["xq"], c["k"], keys, np.zeros(keys.shape, np.float32), store_pairs=pairs,
                                                max_codes=max_codes)
        for rounds in ("1", "2"):
            monkeypatch.setenv("AUNCEL_AMD_FIXED_ROUNDS", rounds)
            monkeypatch.setenv("AUNCEL_AMD_SELECT", "heap" if (seed + int(rounds)) % 3 == 0 else "sorted")  # both selection paths
            h.stats(reset=True)
            D, I = h.search_preassigned(c["xq"], c["k"], keys, store_pairs=pairs, max_codes=max_codes)
            tag = f"{c['kind']} d={c['d']} nlist={c['nlist']} k={c['k']} nprobe={c['nprobe']} pairs={pairs} mc={max_codes} rounds={rounds}"
            assert np.array_equal(I, eI), tag
            assert np.array_equal(bits(D), bits(eD)), tag
            st = h.stats()
            assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(est), tag
    monkeypatch.delenv("AUNCEL_AMD_FIXED_ROUNDS")
    monkeypatch.delenv("AUNCEL_AMD_SELECT")
    # range search around the median of the exact k-th distances
    eD, _, _ = oracle.search_preassigned(lists, c["xq"], 1, keys, np.zeros(keys.shape, np.float32))
    fin = eD[np.isfinite(eD) & (np.abs(eD) < 1e37)]
    radius = float(np.median(fin)) * (1.5 if c["metric"] == 1 else 0.7) if fin.size else 1.0
    elims, elab, edis, est = oracle.range_search_preassigned(lists, c["xq"], radius, keys)
    h.stats(reset=True)
    lims, lab, dis = h.range_search(c["xq"], radius, c["nprobe"], keys=keys)
    assert np.array_equal(lims, elims) and np.array_equal(lab, elab) and np.array_equal(bits(dis), bits(edis))
    st = h.stats()
    assert [st["nlist"], st["ndis"]] == list(est)


def test_oversized_k_fails_loudly(capi):
    """k beyond the selection kernel's LDS heap must come back as an error, not as untouched output buffers"""
    rs = np.random.RandomState(3)
    xb = rs.randn(4000, 16).astype(np.float32)
    h = capi.Handle(16, 4, 1, 0)
    h.set_centroids(xb[:4].copy())
    h.set_lists_from_assign(xb, rs.randint(0, 4, size=4000))
    keys = np.tile(np.arange(4, dtype=np.int64), (2, 1))
    with pytest.raises(RuntimeError, match="LDS heap"):
        h.search_preassigned(xb[:2], 3000, keys)
    D, I = h.search_preassigned(xb[:2], 2000, keys)  # the largest sizes still run
    assert (I[:, 0] == np.arange(2)).all()
