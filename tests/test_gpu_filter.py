"""fp32 lists: threshold rounds as matrix-core filter + exact recomputation (auncel_amd/csrc/ivf_filter.hip) against the
pinned CPU oracle.  The filter may keep too much, never too little: every (D, I) and every statistic must be the oracle's,
bit for bit, in every kernel shape (4 / 8 / 12 / 16 register-resident pieces up to 128 dimensions; the workgroup form and
the one-wave form beyond), with 1 to 4 query blocks per item, ragged lists, and candidates that sit on or within an ulp of
the thresholds.  Every case runs over both copies the filter can stream: scaled fp16 (option "filter" = 2, the default) and
fp32 (1)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[2, 1], ids=["fp16", "fp32"], autouse=True)
def form(request, monkeypatch):
    monkeypatch.setenv("AUNCEL_AMD_FILTER", str(request.param))
    return request.param


@pytest.fixture(scope="module")
def capi():
    from auncel_amd import capi
    capi.lib()
    return capi


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def clustered(rs, nb, nq, d, nlist, spread=0.35):
    cen = rs.randn(nlist, d).astype(np.float32)
    assign = rs.randint(0, nlist, size=nb)
    if nlist > 3:
        assign[assign == 1] = 0          # an empty list and a long one
    xb = (cen[assign] + spread * rs.randn(nb, d)).astype(np.float32)
    xq = (cen[rs.randint(0, nlist, size=nq)] + spread * rs.randn(nq, d)).astype(np.float32)
    return cen, assign, xb, xq


def run_and_compare(capi, oracle, metric, cen, assign, xb, xq, k, nprobe, expect_filter=True, store_pairs=False, max_codes=0, arith=0):
    nlist, d = cen.shape
    lists = oracle.Lists(metric, cen, xb, assign)
    npq = min(nprobe, nlist)
    cd, ck = oracle.knn(metric, xq, cen, npq)
    eD, eI, est = oracle.search_preassigned(lists, xq, k, ck, cd, store_pairs=store_pairs, max_codes=max_codes)
    h = capi.Handle(d, nlist, metric, 0)
    h.set_centroids(cen)
    h.set_lists_from_assign(xb, assign)
    h.stats(reset=True)
    D, I = h.search_preassigned(xq, k, ck, cd, store_pairs=store_pairs, max_codes=max_codes)
    launches, kept = h.last_filter()
    assert h.scan_arith() == arith
    if expect_filter:
        assert launches >= 1, "the search did not go through the filter"
    assert np.array_equal(I, eI)
    assert np.array_equal(bits(D), bits(eD))
    st = h.stats()
    assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(est)
    h.close()
    return launches, kept


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("d", [20, 64, 96, 128, 136, 200, 520, 960])
def test_filter_rounds_equal_the_oracle(capi, oracle, d, metric):
    rs = np.random.RandomState(7000 + d + metric)
    nb = 5000 if d <= 200 else 2500
    # queries per list (= per item, up to 128): ~19, ~37, ~75, ~130 -> 1, 2, 3, 4 (+ 1) query blocks in the workgroup form
    for nlist, nq in ((64, 150), (32, 150), (16, 150), (16, 260)):
        cen, assign, xb, xq = clustered(rs, nb, nq, d, nlist)
        run_and_compare(capi, oracle, metric, cen, assign, xb, xq, k=10, nprobe=8)


@pytest.mark.parametrize("d", [136, 960])
def test_one_wave_form_beyond_128_dimensions(capi, oracle, monkeypatch, d):
    monkeypatch.setenv("AUNCEL_AMD_FILTER_NARROW", "1")
    rs = np.random.RandomState(7100 + d)
    cen, assign, xb, xq = clustered(rs, 2500, 150, d, 16)
    run_and_compare(capi, oracle, 1, cen, assign, xb, xq, k=10, nprobe=8)


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("d", [32, 96, 200])
def test_candidates_on_and_next_to_the_threshold(capi, oracle, d, metric):
    """Every vector exists several times, spread over the lists, next to copies that differ by one ulp in one coordinate: the
    k-th best distance of every query is shared by, or within an ulp of, candidates in later probes.  A filter bound that is
    too tight, a rescoring that rounds differently, or a selection that orders equal values differently shows up here."""
    rs = np.random.RandomState(7200 + d + metric)
    nlist, nq, k = 16, 160, 12
    cen = rs.randn(nlist, d).astype(np.float32)
    seeds = (cen[rs.randint(0, nlist, size=300)] + 0.3 * rs.randn(300, d)).astype(np.float32)
    rows = []
    for v in seeds:
        for _ in range(3):
            rows.append(v)                                   # exact copies
        for _ in range(4):
            w = v.copy()
            c = rs.randint(0, d)
            w[c] = np.nextafter(w[c], np.float32(np.inf if rs.rand() < 0.5 else -np.inf), dtype=np.float32)
            rows.append(w)                                   # one ulp away in one coordinate
    xb = np.stack(rows).astype(np.float32)
    xb = xb[rs.permutation(len(xb))]
    assign = rs.randint(0, nlist, size=len(xb))              # copies land in different lists: ties across probes
    xq = (seeds[rs.randint(0, len(seeds), size=nq)] + 0.05 * rs.randn(nq, d)).astype(np.float32)
    launches, kept = run_and_compare(capi, oracle, metric, cen, assign, xb, xq, k=k, nprobe=nlist)
    assert kept > 0


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_beyond_128_dimensions(capi, oracle, seed):
    """the workgroup form on random shapes: dimensions that are no multiple of 8 (the last piece is zero-padded) or of the stage
    length, one to several chunks per list, k up to the register-array limit, both metrics"""
    rs = np.random.RandomState(7300 + seed)
    d = int(rs.choice([130, 132, 136, 190, 250, 333, 512, 700, 1000]))
    nlist = int(rs.choice([8, 16, 40]))
    nb = int(rs.choice([1500, 4000]))
    nq = int(rs.choice([140, 300]))
    k = int(rs.choice([1, 10, 50, 128]))
    nprobe = int(rs.choice([4, 8, nlist]))
    metric = int(rs.choice([0, 1]))
    cen, assign, xb, xq = clustered(rs, nb, nq, d, nlist, spread=float(rs.choice([0.2, 0.5])))
    run_and_compare(capi, oracle, metric, cen, assign, xb, xq, k=k, nprobe=nprobe, expect_filter=nq * nprobe >= 1024)
    # (list, position) pairs instead of ids; a cap on the codes a query may visit (the reference's loop breaks mid-ranking)
    run_and_compare(capi, oracle, metric, cen, assign, xb, xq, k=k, nprobe=nprobe, expect_filter=False, store_pairs=True)
    run_and_compare(capi, oracle, metric, cen, assign, xb, xq, k=k, nprobe=nprobe, expect_filter=False, max_codes=max(1, nb // 6))


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("scale", [1e-3, 1.0, 3e4])
def test_fp16_form_scales_the_operands(capi, oracle, form, scale, metric):
    """magnitudes far from 1 on the two sides (the fp16 copies are scaled by powers of two into [2^14, 2^15)), lists and queries
    scaled differently"""
    rs = np.random.RandomState(7400 + metric)
    cen, assign, xb, xq = clustered(rs, 5000, 150, 96, 32)
    s = np.float32(scale)
    run_and_compare(capi, oracle, metric, cen * s, assign, xb * s, xq * s * np.float32(0.37), k=10, nprobe=8)


@pytest.mark.parametrize("metric", [1, 0])
def test_fp16_form_on_integers_it_holds_exactly(capi, oracle, form, metric):
    """integers of magnitude <= 2048 that are no bytes (negative values): the fp32 path, fp16 copies exact, no extra margin"""
    rs = np.random.RandomState(7500 + metric)
    nlist, d = 32, 64
    cen = rs.randint(-200, 200, size=(nlist, d)).astype(np.float32)
    assign = rs.randint(0, nlist, size=5000)
    xb = (cen[assign] + rs.randint(-40, 40, size=(5000, d))).astype(np.float32)
    xq = (cen[rs.randint(0, nlist, size=150)] + rs.randint(-40, 40, size=(150, d))).astype(np.float32)
    run_and_compare(capi, oracle, metric, cen, assign, xb, xq, k=10, nprobe=8, arith=1)  # (small integers: fused arithmetic is exact)


def test_fp16_form_without_a_usable_scale(capi, oracle, form):
    """one scale per matrix is only safe while no row is tiny beside the largest (its elements would underflow in fp16) and the
    magnitudes are within 2^+-24: a call whose queries break that keeps every candidate and recomputes it exactly, lists that break
    it are filtered over their fp32 copy; rows of zeros are exact in any scale"""
    rs = np.random.RandomState(7600)
    cen, assign, xb, xq = clustered(rs, 4000, 150, 64, 16)
    for factor in (2.0 ** 50, 2.0 ** -20, 0.0):
        xq2 = xq.copy()
        xq2[17] *= np.float32(factor)
        run_and_compare(capi, oracle, 1, cen, assign, xb, xq2, k=10, nprobe=8)
        xb2 = xb.copy()
        xb2[123] *= np.float32(factor if factor < 1 else 2.0 ** 40)
        run_and_compare(capi, oracle, 1, cen, assign, xb2, xq, k=10, nprobe=8)


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("d", [20, 96, 128, 960])
def test_dense_rounds_from_rows_and_from_the_lane_ordered_copy(capi, oracle, d, metric):
    """option "lanes": scan_lanes_kernel (the lists in the order a wave consumes them) and scan_tiles_kernel (rows staged through
    LDS) are the same arithmetic in the same order -- ids, distances and statistics equal each other and the oracle's in every
    tile shape (1 .. 64 queries on a list, ragged lists, a list of a few vectors, an empty one)"""
    rs = np.random.RandomState(7300 + d + metric)
    for nlist, nq in ((64, 100), (16, 260), (4, 300)):
        cen, assign, xb, xq = clustered(rs, 3000, nq, d, nlist)
        assign[:5] = nlist - 1  # (some vectors in the last list whatever the draw)
        lists = oracle.Lists(metric, cen, xb, assign)
        cd, ck = oracle.knn(metric, xq, cen, min(8, nlist))
        eD, eI, est = oracle.search_preassigned(lists, xq, 10, ck, cd)
        got = []
        for lanes in (1, 0):
            h = capi.Handle(d, nlist, metric, 0)
            h.set_centroids(cen)
            h.set_lists_from_assign(xb, assign)
            h.set_option("lanes", lanes)
            h.stats(reset=True)
            D, I = h.search_preassigned(xq, 10, ck, cd)
            st = h.stats()
            assert np.array_equal(I, eI) and np.array_equal(bits(D), bits(eD)), (lanes, nlist)
            assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(est)
            got.append((D, I))
            h.close()
        assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(bits(got[0][0]), bits(got[1][0]))


def test_an_index_whose_lane_ordered_copy_does_not_fit_still_searches(capi, oracle, monkeypatch):
    """the lane-ordered copy is the fp32 lists once more: where it does not fit beside them (here: AUNCEL_AMD_LANES_NOFIT, the
    hipMemGetInfo check answered "no") the search must not fail -- the dense rounds read the rows (scan_tiles_kernel), the state is
    remembered (no second attempt), results are the oracle's"""
    monkeypatch.setenv("AUNCEL_AMD_LANES_NOFIT", "1")
    rs = np.random.RandomState(7350)
    cen, assign, xb, xq = clustered(rs, 3000, 260, 96, 16)
    lists = oracle.Lists(1, cen, xb, assign)
    cd, ck = oracle.knn(1, xq, cen, 8)
    eD, eI, est = oracle.search_preassigned(lists, xq, 10, ck, cd)
    h = capi.Handle(96, 16, 1, 0)
    h.set_centroids(cen)
    h.set_lists_from_assign(xb, assign)
    assert h.get_option("lanes") == 1  # (the option is on: it is the copy that is missing)
    for _ in range(2):
        D, I = h.search_preassigned(xq, 10, ck, cd)
        assert np.array_equal(I, eI) and np.array_equal(bits(D), bits(eD))
    h.close()


def test_large_fp32_searches_wait_for_each_other(capi, oracle):
    """option "fp32_in_flight": with room for one, three threads' searches of >= 256 queries run one after the other and every one
    returns the oracle's result"""
    import threading
    rs = np.random.RandomState(7400)
    cen, assign, xb, xq = clustered(rs, 4000, 300, 64, 16)
    lists = oracle.Lists(1, cen, xb, assign)
    cd, ck = oracle.knn(1, xq, cen, 8)
    eD, eI, _ = oracle.search_preassigned(lists, xq, 10, ck, cd)
    h = capi.Handle(64, 16, 1, 0)
    h.set_centroids(cen)
    h.set_lists_from_assign(xb, assign)
    h.set_option("fp32_in_flight", 1)
    ctxs = [h.clone() for _ in range(3)]
    out, errs = [None] * 3, []

    def work(i):
        try:
            for _ in range(3):
                out[i] = ctxs[i].search_preassigned(xq, 10, ck, cd)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errs and all(not t.is_alive() for t in ts)
    for D, I in out:
        assert np.array_equal(I, eI) and np.array_equal(bits(D), bits(eD))
    for c in ctxs:
        c.close()
    h.close()
