"""GPU parity tests proper: the HIP path, called through the C ABI, against the golden outputs of
the compiled reference (tests/golden) and against the pinned CPU oracle.  Bar: ids bit-exact,
distances bit-exact (stronger than the 1e-4 relative the north star asks for)."""
import numpy as np
import pytest

from util import KMEANS, AUNCEL, AUNCEL_BIG, FIXED, load_case, traces_from_gold

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from auncel_amd import capi
    assert capi.device_count() >= 1
    return capi


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make_index(capi, case, gold, centroids=None):
    cen = case["centroids"] if centroids is None else centroids
    h = capi.Handle(case["d"], case["nlist"], case["metric"], 0)
    h.set_centroids(cen)
    h.set_lists_from_assign(case["xb"], gold["assign"])
    return h


@pytest.mark.parametrize("name", FIXED)
def test_coarse_exact(capi, name):
    case, gold = load_case(name)
    if case["d"] % 4 != 0:
        pytest.skip("reference takes the BLAS path when d % 4 != 0")
    h = make_index(capi, case, gold)
    D, I = h.coarse(case["xq"], case["nprobe"], mode=0)
    assert np.array_equal(I, gold["coarse_keys_sse"])
    assert np.array_equal(bits(D), bits(gold["coarse_dis_sse"]))


@pytest.mark.parametrize("name", FIXED)
def test_search_preassigned(capi, name):
    case, gold = load_case(name)
    h = make_index(capi, case, gold)
    for l in range(case["nlist"]):
        assert h.list_size(l) == gold["list_sizes"][l]
    for k in case["ks"]:
        for pairs in (False, True):
            suf = f"_k{k}" + ("_pairs" if pairs else "")
            h.stats(reset=True)
            D, I = h.search_preassigned(case["xq"], int(k), gold["coarse_keys_sse"], gold["coarse_dis_sse"], store_pairs=pairs)
            assert np.array_equal(I, gold["I" + suf]), suf
            assert np.array_equal(bits(D), bits(gold["D" + suf])), suf
            st = h.stats()
            assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(gold["stats" + suf]), suf
            assert st["nq"] == case["xq"].shape[0]
        if case["max_codes"]:
            D, I = h.search_preassigned(case["xq"], int(k), gold["coarse_keys_sse"], gold["coarse_dis_sse"],
                                        max_codes=case["max_codes"])
            assert np.array_equal(I, gold[f"I_k{k}_maxcodes"])
            assert np.array_equal(bits(D), bits(gold[f"D_k{k}_maxcodes"]))


@pytest.mark.parametrize("name", [n for n in FIXED if n != "fixed_odd_d30"])
def test_search_end_to_end(capi, name):
    """IndexIVF::search = coarse + scan on the device (exact coarse kernel)"""
    case, gold = load_case(name)
    h = make_index(capi, case, gold)
    k = int(case["ks"][0])
    D, I = h.search(case["xq"], k, case["nprobe"], coarse_mode=0)
    assert np.array_equal(I, gold[f"I_k{k}"])
    assert np.array_equal(bits(D), bits(gold[f"D_k{k}"]))
    h.set_queries(case["xq"])
    n = case["xq"].shape[0]
    D, I = h.search_resident(3, n - 3, k, case["nprobe"])
    assert np.array_equal(I, gold[f"I_k{k}"][3:])
    assert np.array_equal(bits(D), bits(gold[f"D_k{k}"][3:]))


@pytest.mark.parametrize("name", FIXED)
def test_scanner_api(capi, name):
    """set_query / set_list / scan_codes / distance_to_code on a caller-owned raw heap
    (reference tests/test_lowlevel_ivf.cpp:82-220)"""
    case, gold = load_case(name)
    h = make_index(capi, case, gold)
    k = int(case["ks"][0])
    is_l2 = case["metric"] == 1
    fmax = np.finfo(np.float32).max
    for i in range(min(3, gold["scan_heap_D"].shape[0])):
        simi = np.full(k, fmax if is_l2 else -fmax, dtype=np.float32)
        idxi = np.full(k, -1, dtype=np.int64)
        for p in range(case["nprobe"]):
            key = int(gold["coarse_keys_sse"][i, p])
            if key < 0 or h.list_size(key) == 0:
                continue
            nup = h.scan_codes(case["xq"][i], key, simi, idxi)
            assert nup == gold["scan_nup"][i, p]
            assert bits(h.distance_to_code(case["xq"][i], key, 0)) == bits(gold["scan_dist_to_code"][i, p])
        assert np.array_equal(bits(simi), bits(gold["scan_heap_D"][i]))
        assert np.array_equal(idxi, gold["scan_heap_I"][i])


@pytest.mark.parametrize("name", ["fixed_sift_l2", "fixed_gauss_l2_d96", "fixed_deep_ip_d96", "fixed_dups", "fixed_odd_d30"])
def test_scanner_over_list_parts(capi, oracle, name):
    """amd_ivf_scan_codes_at / amd_ivf_scan_codes_range: the reference's scanner scans the n codes at whatever pointer it is handed
    (IndexIVFFlat.cpp:117-155).  A list handed over in parts of any length leaves the heap the whole list leaves (values, labels,
    update counts: the goldens of the whole-list scan), store_pairs labels count from the part's first code, and the range scan
    reports the entries inside the radius in position order with the reference's distance bits (oracle: its per-pair distance)"""
    case, gold = load_case(name)
    h = make_index(capi, case, gold)
    k = int(case["ks"][0])
    is_l2 = case["metric"] == 1
    fmax = np.finfo(np.float32).max
    rs = np.random.RandomState(5)
    for i in range(min(3, gold["scan_heap_D"].shape[0])):
        simi = np.full(k, fmax if is_l2 else -fmax, dtype=np.float32)
        idxi = np.full(k, -1, dtype=np.int64)
        simi_p, idxi_p = simi.copy(), idxi.copy()
        for p in range(case["nprobe"]):
            key = int(gold["coarse_keys_sse"][i, p])
            sz = h.list_size(key) if key >= 0 else 0
            if sz == 0:
                continue
            cuts = np.unique(np.concatenate([[0, sz], rs.randint(0, sz + 1, size=3)]))  # up to four parts, some may be one vector
            nup = 0
            _, ids = h.get_list(key)
            for a, b in zip(cuts[:-1], cuts[1:]):
                nup += h.scan_codes_at(case["xq"][i], key, int(a), int(b - a), simi, idxi)
                before = idxi_p.copy()
                h.scan_codes_at(case["xq"][i], key, int(a), int(b - a), simi_p, idxi_p, store_pairs=True)
                new = ~np.isin(idxi_p, before)  # (what this part admitted; entries the sifting only moved are older labels)
                assert np.array_equal(bits(simi_p), bits(simi))
                assert np.all((idxi_p[new] >> 32) == key) and np.all((idxi_p[new] & 0xffffffff) < b - a)
                assert np.array_equal(ids[(idxi_p[new] & 0xffffffff) + a], idxi[new])
            assert nup == gold["scan_nup"][i, p]
            # an empty run and a run past the end
            assert h.scan_codes_at(case["xq"][i], key, sz, 0, simi, idxi) == 0
            with pytest.raises(capi.EngineError, match="beyond the end"):
                h.scan_codes_at(case["xq"][i], key, 1, sz, simi, idxi)
        assert np.array_equal(bits(simi), bits(gold["scan_heap_D"][i]))
        assert np.array_equal(idxi, gold["scan_heap_I"][i])
    # range scan of list parts: the per-pair distances of the pinned oracle, filtered at a radius that keeps about a third
    for i in range(2):
        key = int(gold["coarse_keys_sse"][i, 0])
        sz = h.list_size(key)
        if sz < 4:
            continue
        codes, _ = h.get_list(key)
        # (fvec_L2sqr / fvec_inner_product are symmetric in their arguments, element by element: each code against the one query)
        od = oracle.knn(case["metric"], codes, case["xq"][i][None, :], 1)[0][:, 0]
        radius = float(np.sort(od)[sz // 3 if is_l2 else sz - sz // 3])
        a, b = sz // 4, sz - 1
        pos, dis = h.scan_codes_range(case["xq"][i], key, a, b - a, radius)
        want = np.nonzero(od[a:b] < radius if is_l2 else od[a:b] > radius)[0]
        assert np.array_equal(pos, want.astype(np.uint32))
        assert np.array_equal(bits(dis), bits(od[a:b][want]))
        pos0, _ = h.scan_codes_range(case["xq"][i], key, 0, 0, radius)
        assert len(pos0) == 0


@pytest.mark.parametrize("name", ["fixed_sift_l2", "fixed_ragged", "fixed_dups"])
def test_add_builds_reference_lists(capi, name):
    """IndexIVFFlat::add_core: nearest-centroid assignment + append in input order"""
    case, gold = load_case(name)
    h = capi.Handle(case["d"], case["nlist"], case["metric"], 0)
    h.set_centroids(case["centroids"])
    h.add(case["xb"][:1000])
    h.add(case["xb"][1000:])
    assert h.ntotal == case["xb"].shape[0]
    a = gold["assign"]
    for l in range(case["nlist"]):
        codes, ids = h.get_list(l)
        want = np.nonzero(a == l)[0]
        assert np.array_equal(ids, want)
        assert np.array_equal(codes, case["xb"][want])


@pytest.mark.parametrize("name", [n for n in FIXED if n not in ("fixed_gist_l2_d960", "fixed_odd_d30")])
def test_shards_by_list_id(capi, name):
    """config 4's contract: lists sharded by list id, per-shard top-k merged on the host"""
    case, gold = load_case(name)
    nshard, a = case["nshard"], gold["assign"]
    for k in case["ks"]:
        allD, allI = [], []
        for s in range(nshard):
            h = capi.Handle(case["d"], case["nlist"], case["metric"], 0)
            h.set_centroids(case["centroids"])
            h.set_lists_from_assign(case["xb"], np.where(a % nshard == s, a, -1))
            D, I = h.search(case["xq"], int(k), case["nprobe"])
            allD.append(D)
            allI.append(I)
        D, I = capi.merge_tables(case["metric"], np.stack(allD), np.stack(allI))
        assert np.array_equal(I, gold[f"I_shards_k{k}"])
        assert np.array_equal(bits(D), bits(gold[f"D_shards_k{k}"]))


@pytest.mark.parametrize("name", FIXED)
def test_range_search(capi, name):
    """IndexIVF::range_search_preassigned: lims, labels and distances in the reference's order, bit for bit"""
    case, gold = load_case(name)
    h = make_index(capi, case, gold)
    radius = float(case["radius"][0])
    h.stats(reset=True)
    lims, labels, dist = h.range_search(case["xq"], radius, case["nprobe"], keys=gold["coarse_keys_sse"])
    assert np.array_equal(lims, gold["range_lims"])
    assert np.array_equal(labels, gold["range_labels"][:lims[-1]])
    assert np.array_equal(bits(dist), bits(gold["range_distances"][:lims[-1]]))
    st = h.stats()
    assert [st["nlist"], st["ndis"]] == list(gold["range_stats"]) and st["nq"] == case["xq"].shape[0]
    if case["d"] % 4 == 0:  # own coarse ranking (exact mode) gives the same keys
        lims2, labels2, dist2 = h.range_search(case["xq"], radius, case["nprobe"])
        assert np.array_equal(lims2, lims) and np.array_equal(labels2, labels) and np.array_equal(bits(dist2), bits(dist))
    # nothing inside the radius / everything inside it
    none = -1.0 if case["metric"] == 1 else 3.0e38
    lims0, labels0, _ = h.range_search(case["xq"], none, case["nprobe"], keys=gold["coarse_keys_sse"])
    assert lims0[-1] == 0 and labels0.size == 0
    every = 3.0e38 if case["metric"] == 1 else -3.0e38
    limsa, _, _ = h.range_search(case["xq"][:4], every, case["nprobe"], keys=gold["coarse_keys_sse"][:4])
    sizes = np.array([h.list_size(l) for l in range(case["nlist"])])
    k4 = gold["coarse_keys_sse"][:4]
    assert np.array_equal(np.diff(limsa), np.where(k4 >= 0, sizes[np.clip(k4, 0, None)], 0).sum(1))


def test_invalid_key_is_an_engine_error(capi):
    case, gold = load_case("fixed_ragged")
    h = make_index(capi, case, gold)
    keys = gold["coarse_keys_sse"].copy()
    keys[0, 0] = case["nlist"] + 3
    with pytest.raises(capi.EngineError) as e:
        h.search_preassigned(case["xq"], 10, keys)
    assert e.value.code == -2 and "Invalid key" in str(e.value)


# ------------------------------------------------------------------------------ Auncel
@pytest.mark.parametrize("name", AUNCEL)
def test_interdis_table(capi, name):
    case, gold = load_case(name)
    h = capi.Handle(case["d"], case["nlist"], case["metric"], 0)
    h.set_centroids(gold["centroids"])
    h.set_interdis(None)
    assert np.array_equal(bits(h.get_interdis()), bits(gold["interdis_cem"]))


@pytest.mark.parametrize("select", ["sorted", "heap"])
@pytest.mark.parametrize("name", AUNCEL)
def test_adaptive_search(capi, monkeypatch, name, select):
    """Error_sys::search: per-query error-bounded nprobe (my_nprobe), results, recall log -- with the sorted-array selection
    (+ tie_fix_kernel for the queries in which equal distances meet) and with the reference's heap replayed for every query"""
    monkeypatch.setenv("AUNCEL_AMD_SELECT", select)
    case, gold = load_case(name)
    K, ts, ses = case["max_topk"], case["train_num"], case["test_num"]
    h = make_index(capi, case, gold, gold["centroids"])
    h.set_interdis(None)
    h.set_tuner(K, traces_from_gold(gold), gold["arcos_list"])
    h.set_queries(case["xq"])
    for r in range(len(case["topks"])):
        for prof in (False, True):
            req = np.full(ts + ses, case["require_acc"][r], dtype=np.float32)
            my_np = np.zeros(ts + ses, dtype=np.uint64)
            t_rec = np.zeros(ts + ses, dtype=np.float32)
            h.stats(reset=True)
            D, I = h.search_adaptive(ts, ses, int(case["topks"][r]), float(case["multipler"][r]), float(case["std_m"][r]),
                                     req, my_np, t_rec, gt_D=gold["gtD"], profile=prof)
            suf = f"_r{r}" + ("_prof" if prof else "")
            assert np.array_equal(my_np[ts:], gold["my_nprobe" + suf]), suf
            assert np.array_equal(I, gold["I" + suf]), suf
            assert np.array_equal(bits(D), bits(gold["D" + suf])), suf
            assert np.array_equal(bits(t_rec[ts:]), bits(gold["t_recalls" + suf])), suf
            st = h.stats()
            assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(gold["stats" + suf]), suf


@pytest.mark.parametrize("select", ["sorted", "heap"])
@pytest.mark.parametrize("name", AUNCEL_BIG)
def test_adaptive_search_nlist4096(capi, monkeypatch, name, select):
    """BASELINE config 2's shape against the compiled reference: IVF4096 (max_num 532, ten traces, the prefix coarse
    ranking, the round planner at 4096 lists), both selection paths"""
    import hashlib
    monkeypatch.setenv("AUNCEL_AMD_SELECT", select)
    case, gold = load_case(name)
    K, ts, ses = case["max_topk"], case["train_num"], case["test_num"]
    h = make_index(capi, case, gold, gold["centroids"])
    h.set_interdis(None)
    inter = h.get_interdis()
    assert hashlib.sha256(np.ascontiguousarray(inter).tobytes()).hexdigest() == str(gold["interdis_cem_sha"])
    h.set_tuner(K, traces_from_gold(gold), gold["arcos_list"])
    h.set_queries(case["xq"])
    for r in range(len(case["topks"])):
        for prof in (False, True):
            req = np.full(ts + ses, case["require_acc"][r], dtype=np.float32)
            my_np = np.zeros(ts + ses, dtype=np.uint64)
            t_rec = np.zeros(ts + ses, dtype=np.float32)
            h.stats(reset=True)
            D, I = h.search_adaptive(ts, ses, int(case["topks"][r]), float(case["multipler"][r]), float(case["std_m"][r]),
                                     req, my_np, t_rec, gt_D=gold["gtD"], profile=prof)
            suf = f"_r{r}" + ("_prof" if prof else "")
            assert np.array_equal(my_np[ts:], gold["my_nprobe" + suf]), suf
            assert np.array_equal(I, gold["I" + suf]), suf
            assert np.array_equal(bits(D), bits(gold["D" + suf])), suf
            assert np.array_equal(bits(t_rec[ts:]), bits(gold["t_recalls" + suf])), suf
            st = h.stats()
            assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(gold["stats" + suf]), suf


@pytest.mark.parametrize("name", AUNCEL + AUNCEL_BIG)
def test_overhead_profile(capi, name):
    """error_pro::overhead_profile (eval/overhead.cpp): rule on every probe, verdict ignored, every query to stage nlist / 8;
    the plain nlist / 8 search gives the same results (what the class mirror times as "without ELP")"""
    case, gold = load_case(name)
    K, ts, ses = case["max_topk"], case["train_num"], case["test_num"]
    h = make_index(capi, case, gold, gold["centroids"])
    h.set_interdis(None)
    h.set_tuner(K, traces_from_gold(gold), gold["arcos_list"])
    h.set_queries(case["xq"])
    req = np.full(ts + ses, case["require_acc"][0], dtype=np.float32)
    my_np = np.zeros(ts + ses, dtype=np.uint64)
    t_rec = np.zeros(ts + ses, dtype=np.float32)
    h.stats(reset=True)
    D, I = h.search_adaptive(ts, ses, int(case["topks"][0]), float(case["multipler"][0]), float(case["std_m"][0]), req, my_np, t_rec,
                             gt_D=gold["gtD"], profile=2)
    assert not my_np.any() and not gold["my_nprobe_overhead"].any()
    assert np.array_equal(I, gold["I_overhead"])
    assert np.array_equal(bits(D), bits(gold["D_overhead"]))
    st = h.stats()
    assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(gold["stats_overhead"])
    D2, I2 = h.search_resident(ts, ses, K, case["nlist"] // 8)
    assert np.array_equal(I2, I) and np.array_equal(bits(D2), bits(D))


@pytest.mark.parametrize("name", AUNCEL)
def test_trace_training_samples(capi, oracle, name):
    """Error_sys::sys_train's search pass: raw (sum_angle, kscaling) samples per power-of-two stage"""
    case, gold = load_case(name)
    K, ts = case["max_topk"], case["train_num"]
    h = make_index(capi, case, gold, gold["centroids"])
    h.set_interdis(None)
    h.set_queries(case["xq"])
    ntr = len(traces_from_gold(gold))
    raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
    D, I = h.train_samples(0, ts, K, gold["gtD"], ts, raw)
    # The reference trains in batches of >= 20 queries, i.e. through vendor-BLAS coarse distances
    # (unpinned rounding; the k-means centroids are not integers even on integer data).  The
    # expectation is therefore the pinned oracle fed with the exact coarse ranking, which is what
    # the engine's coarse kernel produces bit for bit (test_coarse_exact).
    lists = oracle.Lists(case["metric"], gold["centroids"], case["xb"], gold["assign"])
    exp_raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
    eD, eI = oracle.train_samples(lists, case["xq"][:ts], K, gold["coarse_keys_sse"][:ts], gold["coarse_dis_sse"][:ts],
                                  gold["interdis_cem"], gold["arcos_list"], gold["gtD"], 0, ts, exp_raw)
    assert np.array_equal(I, eI)
    assert np.array_equal(bits(D), bits(eD))
    for i in range(ntr):
        assert np.array_equal(bits(raw[i]), bits(exp_raw[i])), i


def test_fused_path_is_bit_identical(capi, monkeypatch):
    """integer-valued operands (|v| <= 4095): the engine switches the scan to fma(t, t, acc); every
    product is exactly representable there, so results must not change by a single bit (and they are
    compared with the reference's goldens by the tests above as well)"""
    case, gold = load_case("fixed_sift_l2")
    h1 = make_index(capi, case, gold)
    monkeypatch.setenv("AUNCEL_AMD_NO_FUSED", "1")
    h2 = make_index(capi, case, gold)
    monkeypatch.delenv("AUNCEL_AMD_NO_FUSED")
    for k in (10, 100):
        D1, I1 = h1.search(case["xq"], k, case["nprobe"])
        D2, I2 = h2.search(case["xq"], k, case["nprobe"])
        assert np.array_equal(I1, I2) and np.array_equal(bits(D1), bits(D2))
    # values outside the exact range switch it off again: same index, queries scaled by 1000
    xq = case["xq"] * np.float32(1000.0)
    D1, I1 = h1.search(xq, 10, case["nprobe"])
    D2, I2 = h2.search(xq, 10, case["nprobe"])
    assert np.array_equal(I1, I2) and np.array_equal(bits(D1), bits(D2))


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_byte_code_path_is_bit_identical(capi, monkeypatch, metric):
    """integers 0..255 on both sides and d * 255^2 <= 2^24: the lists are kept as bytes and the scan runs on
    integer dot products.  Every partial sum of the reference is then an exact integer, so nothing may change
    by a single bit against the fp32 kernels (which the tests above pin to the reference's goldens)."""
    case, gold = load_case("fixed_sift_l2")
    m = capi.METRIC_L2 if metric == "l2" else capi.METRIC_IP

    def build():
        h = capi.Handle(case["d"], case["nlist"], m, 0)
        h.set_centroids(case["centroids"])
        h.add(case["xb"])
        return h

    h1 = build()
    monkeypatch.setenv("AUNCEL_AMD_NO_BYTES", "1")
    h2 = build()
    monkeypatch.delenv("AUNCEL_AMD_NO_BYTES")
    for k in (10, 100):
        D1, I1 = h1.search(case["xq"], k, case["nprobe"])
        D2, I2 = h2.search(case["xq"], k, case["nprobe"])
        assert h1.scan_arith() == 2 and h2.scan_arith() == 1
        assert np.array_equal(I1, I2) and np.array_equal(bits(D1), bits(D2))
    # a query outside 0..255 sends the call back to fp32: same index, one value of 256
    xq = case["xq"].copy()
    xq[0, 0] = 256.0
    D1, I1 = h1.search(xq, 10, case["nprobe"])
    D2, I2 = h2.search(xq, 10, case["nprobe"])
    assert h1.scan_arith() == 1
    assert np.array_equal(I1, I2) and np.array_equal(bits(D1), bits(D2))
    # non-integer queries: reference order
    D1, I1 = h1.search(case["xq"] + np.float32(0.5), 10, case["nprobe"])
    D2, I2 = h2.search(case["xq"] + np.float32(0.5), 10, case["nprobe"])
    assert h1.scan_arith() == 0
    assert np.array_equal(I1, I2) and np.array_equal(bits(D1), bits(D2))


@pytest.mark.parametrize("name", FIXED)
def test_coarse_gemm_mode(capi, name):
    """mode 1 = |x|^2 + |y|^2 - 2 x.y on the fp32 matrix cores (the reference's BLAS branch).  Integer-valued
    data: every summation order is exact, so the reference's MKL-made goldens apply bit for bit; float data:
    vendor-BLAS rounding is unpinned, distances must agree with the exact kernel to 1e-4 relative."""
    case, gold = load_case(name)
    h = make_index(capi, case, gold)
    D, I = h.coarse(case["xq"], case["nprobe"], mode=1)
    if name in ("fixed_sift_l2", "fixed_odd_d30", "fixed_ragged", "fixed_dups"):
        assert np.array_equal(bits(D), bits(gold["coarse_dis_blas"]))
        same = I == gold["coarse_keys_blas"]
        for q, j in zip(*np.nonzero(~same)):  # ids may differ only inside groups of equal distances
            tie = D[q] == D[q, j]
            assert set(I[q][tie]) == set(gold["coarse_keys_blas"][q][tie])
    else:
        De, Ie = h.coarse(case["xq"], case["nprobe"], mode=0)
        assert np.allclose(D, De, rtol=1e-4, atol=1e-5)
        assert (I == Ie).mean() > 0.99
    # mode -1 follows the reference's switch: >= 20 queries -> GEMM branch
    Dm, Im = h.coarse(case["xq"], case["nprobe"], mode=-1)
    assert np.array_equal(bits(Dm), bits(D)) and np.array_equal(Im, I)
    D1, I1 = h.coarse(case["xq"][:5], case["nprobe"], mode=-1)
    if case["d"] % 4 == 0:
        De, Ie = h.coarse(case["xq"][:5], case["nprobe"], mode=0)
        assert np.array_equal(bits(D1), bits(De)) and np.array_equal(I1, Ie)


def test_clone_searches_concurrently(capi):
    """amd_ivf_clone: two contexts over one index, a batch in flight on each from two threads; both must reproduce the
    reference's adaptive-search goldens and the owner's fixed-nprobe result; mutators are refused on the clone"""
    import threading
    case, gold = load_case("auncel_sift_d32")
    K, ts, ses = case["max_topk"], case["train_num"], case["test_num"]
    h = make_index(capi, case, gold, gold["centroids"])
    h.set_interdis(None)
    h.set_tuner(K, traces_from_gold(gold), gold["arcos_list"])
    c = h.clone()
    D0, I0 = h.search(case["xq"], 10, 8)
    out, errs = {}, []

    def work(name, hh):
        try:
            hh.set_queries(case["xq"])
            for _ in range(3):
                req = np.full(ts + ses, case["require_acc"][0], dtype=np.float32)
                my_np = np.zeros(ts + ses, dtype=np.uint64)
                t_rec = np.zeros(ts + ses, dtype=np.float32)
                D, I = hh.search_adaptive(ts, ses, int(case["topks"][0]), float(case["multipler"][0]), float(case["std_m"][0]),
                                          req, my_np, t_rec, gt_D=gold["gtD"], profile=False)
                out[name] = (D, I, my_np[ts:].copy(), hh.search(case["xq"], 10, 8))
        except Exception as e:  # noqa: BLE001
            errs.append((name, e))

    th = [threading.Thread(target=work, args=(n, hh)) for n, hh in (("owner", h), ("clone", c))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for name in ("owner", "clone"):
        D, I, my_np, (Df, If) = out[name]
        assert np.array_equal(my_np, gold["my_nprobe_r0"]), name
        assert np.array_equal(I, gold["I_r0"]) and np.array_equal(bits(D), bits(gold["D_r0"])), name
        assert np.array_equal(If, I0) and np.array_equal(bits(Df), bits(D0)), name
    with pytest.raises(capi.EngineError):
        c.add(case["xb"][:4])
    assert c.ntotal == h.ntotal
    c.close()


@pytest.mark.parametrize("name", KMEANS)
def test_kmeans_on_the_device(capi, name):
    """amd_ivf_kmeans: training set resident, assignment and centroid update on the GPU, the reference's sequential choices
    on the host; centroids bit for bit against the compiled reference's Clustering::train"""
    case, gold = load_case(name)
    cen, obj = capi.kmeans(case["metric"], case["x"], case["k"], case["niter"], case["seed"], case["max_points_per_centroid"],
                           bool(case["spherical"]))
    assert np.array_equal(bits(cen), bits(gold["centroids"]))
    if case["x"].shape[0] < 20:
        assert np.array_equal(bits(obj), bits(gold["obj"]))
    else:
        assert np.allclose(obj, gold["obj"], rtol=1e-5)


def _check_timed(oracle, lists, xq, k, cd, ck, D, I, used, nprobe):
    """whatever the clock did, query i must hold search_preassigned's result over its first used[i] probes"""
    assert used.min() >= 1 and used.max() <= nprobe
    for u in np.unique(used):
        sel = np.nonzero(used == u)[0]
        u = int(u)
        eD, eI, _ = oracle.search_preassigned(lists, xq[sel], k, ck[sel, :u], cd[sel, :u])
        assert np.array_equal(I[sel], eI), u
        assert np.array_equal(bits(D[sel]), bits(eD)), u


@pytest.mark.parametrize("name", ["fixed_sift_l2", "fixed_deep_ip_d96", "fixed_ragged", "fixed_gist_l2_d960"])
def test_time_bounded_search(capi, oracle, name):
    """Error_sys::time_search: the probe loop left on a per-query time budget.  The stopping points depend
    on the clock, the results may not: they are pinned per query against the oracle at the probe count the engine reports."""
    case, gold = load_case(name)
    h = make_index(capi, case, gold)
    lists = oracle.Lists(case["metric"], case["centroids"], case["xb"], gold["assign"])
    xq, nlist = case["xq"], case["nlist"]
    n = xq.shape[0]
    k = int(case["ks"][-1])
    cd, ck = h.coarse(xq, nlist, mode=0)
    h.set_queries(xq)
    rs = np.random.RandomState(7)
    # no time at all: round 0 (4 probes) and the one probe every query is owed afterwards
    D, I, used = h.search_timed(0, n, k, nlist, np.zeros(n, np.float32))
    assert used.max() <= min(5, nlist)
    _check_timed(oracle, lists, xq, k, cd, ck, D, I, used, nlist)
    # all the time in the world: the full probe loop = plain search with nprobe = nlist
    D, I, used = h.search_timed(0, n, k, nlist, np.full(n, 1e9, np.float32))
    assert np.all(used == nlist)
    fD, fI = h.search(xq, k, nlist, coarse_mode=0)
    assert np.array_equal(I, fI) and np.array_equal(bits(D), bits(fD))
    _check_timed(oracle, lists, xq[:8], k, cd[:8], ck[:8], D[:8], I[:8], used[:8], nlist)
    # budgets around the cost of a round, a slice of the resident queries (budgets are indexed by absolute id),
    # a shorter probe loop, and the same through the host-pointer entry
    for rep in range(3):
        b = rs.choice([0.0, 0.05, 0.1, 0.2, 0.4, 0.8, 3.0], size=n).astype(np.float32)
        nprobe = [nlist, max(1, nlist // 2), nlist][rep]
        if rep < 2:
            D, I, used = h.search_timed(3, n - 3, k, nprobe, b)
        else:
            D, I, used = h.search_timed_x(xq[3:], 3, k, nprobe, b)
        _check_timed(oracle, lists, xq[3:], k, cd[3:], ck[3:], D, I, used, nprobe)


def test_selection_log_overflow_falls_back_to_the_heap(capi, oracle):
    """Every candidate of a list is better than the one before it (vectors ordered by decreasing distance to the query):
    every one of them is admitted, the sorted-array selection's admission log (32 k entries) runs over, and the search is
    repeated with the reference's heap -- same result as the oracle, nheap_updates included."""
    n, d, k = 3000, 4, 10
    xb = np.zeros((n, d), dtype=np.float32)
    xb[:, 0] = np.arange(n, 0, -1, dtype=np.float32)          # the query sits at the origin: distances n^2 .. 1
    xb[:, 1] = (np.arange(n) % 2).astype(np.float32)          # two lists, alternating
    cen = np.array([[n / 2, 0, 0, 0], [n / 2, 1, 0, 0]], dtype=np.float32)
    assign = (np.arange(n) % 2).astype(np.int64)
    xq = np.zeros((3, d), dtype=np.float32)
    xq[1, 1] = 1.0
    xq[2, 0] = 7.5
    keys = np.tile(np.array([0, 1], dtype=np.int64), (3, 1))
    lists = oracle.Lists(1, cen, xb, assign)
    eD, eI, est = oracle.search_preassigned(lists, xq, k, keys, np.zeros(keys.shape, np.float32))
    assert est[2] > 2 * 32 * k  # more admissions than a query's log holds
    h = capi.Handle(d, 2, capi.METRIC_L2, 0)
    h.set_centroids(cen)
    h.set_lists_from_assign(xb, assign)
    h.stats(reset=True)
    D, I = h.search_preassigned(xq, k, keys)
    assert np.array_equal(I, eI)
    assert np.array_equal(bits(D), bits(eD))
    st = h.stats()
    assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(est)


@pytest.mark.parametrize("name", AUNCEL)
def test_adaptive_search_over_the_callers_coarse_ranking(capi, name):
    """IndexIVF::search_preassigned in tune mode with the keys / coarse_dis the caller passes (Auncel/IndexIVF.cpp:382-386): the
    reference's own per-query rankings go in, its (D, I, my_nprobe, t_recalls) must come out"""
    case, gold = load_case(name)
    K, ts, ses = case["max_topk"], case["train_num"], case["test_num"]
    h = make_index(capi, case, gold, gold["centroids"])
    h.set_interdis(None)
    h.set_tuner(K, traces_from_gold(gold), gold["arcos_list"])
    keys, cd = gold["coarse_keys_sse"][ts:], gold["coarse_dis_sse"][ts:]
    for r in range(len(case["topks"])):
        for prof in (False, True):
            req = np.full(ts + ses, case["require_acc"][r], dtype=np.float32)
            my_np = np.zeros(ts + ses, dtype=np.uint64)
            t_rec = np.zeros(ts + ses, dtype=np.float32)
            D, I = h.search_adaptive_pre(case["xq"][ts:], ts, keys, cd, int(case["topks"][r]), float(case["multipler"][r]),
                                         float(case["std_m"][r]), req, my_np, t_rec, gt_D=gold["gtD"], profile=prof)
            suf = f"_r{r}" + ("_prof" if prof else "")
            assert np.array_equal(my_np[ts:], gold["my_nprobe" + suf]), suf
            assert np.array_equal(I, gold["I" + suf]), suf
            assert np.array_equal(bits(D), bits(gold["D" + suf])), suf
            assert np.array_equal(bits(t_rec[ts:]), bits(gold["t_recalls" + suf])), suf
    # a ranking too short for set_online (entries 0 .. nlist/8 + 20) is refused like the reference's precondition
    short = case["nlist"] // 8 + 20
    with pytest.raises(capi.EngineError) as e:
        h.search_adaptive_pre(case["xq"][ts:], ts, keys[:, :short], cd[:, :short], int(case["topks"][0]), 1.0, 1.0,
                              np.full(ts + ses, 0.9, np.float32), np.zeros(ts + ses, np.uint64), np.zeros(ts + ses, np.float32))
    assert e.value.code == -2


@pytest.mark.parametrize("name", AUNCEL)
def test_training_over_the_references_blas_ranking(capi, name):
    """Error_sys::sys_train ranks its training batch through vendor BLAS (unpinned rounding); handed that very ranking
    (search_preassigned's keys / coarse_dis arguments), the training branch must write the reference's raw traces bit for bit"""
    case, gold = load_case(name)
    if "coarse_keys_blas_train" not in gold:
        pytest.skip("golden without the training batch's ranking")
    K, ts = case["max_topk"], case["train_num"]
    h = make_index(capi, case, gold, gold["centroids"])
    h.set_interdis(None)
    ntr = len(traces_from_gold(gold))
    raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
    h.train_samples_pre(case["xq"][:ts], 0, gold["coarse_keys_blas_train"], gold["coarse_dis_blas_train"], K, gold["gtD"], ts, raw)
    for i in range(ntr):
        assert np.array_equal(bits(raw[i]), bits(gold[f"raw_trace{i}"])), i


@pytest.mark.parametrize("select", ["sorted", "heap"])
@pytest.mark.parametrize("name", AUNCEL[:2])
def test_results_written_straight_into_page_locked_buffers(capi, monkeypatch, name, select):
    """a caller that hands over page-locked (D, I) gets its rows written by the selection kernels themselves as queries finish
    (no copy at the end): same bits as the golden outputs, for the sorted-array selection, its tie replay and the heap kernels"""
    import ctypes
    monkeypatch.setenv("AUNCEL_AMD_SELECT", select)
    case, gold = load_case(name)
    K, ts, ses = case["max_topk"], case["train_num"], case["test_num"]
    h = make_index(capi, case, gold, gold["centroids"])
    h.set_interdis(None)
    h.set_tuner(K, traces_from_gold(gold), gold["arcos_list"])
    h.set_queries(case["xq"])
    hip = capi.hip_runtime()  # (the runtime the library itself runs on: a second one in the process would find no device)

    def pinned(shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = ctypes.c_void_p()
        assert hip.hipHostMalloc(ctypes.byref(ptr), ctypes.c_size_t(nbytes), ctypes.c_uint(0)) == 0
        buf = (ctypes.c_char * nbytes).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape), ptr

    Dp, dptr = pinned((ses, K), np.float32)
    Ip, iptr = pinned((ses, K), np.int64)
    for r in range(len(case["topks"])):
        req = np.full(ts + ses, case["require_acc"][r], dtype=np.float32)
        my_np = np.zeros(ts + ses, dtype=np.uint64)
        t_rec = np.zeros(ts + ses, dtype=np.float32)
        Dp.fill(np.nan)
        Ip.fill(-7)
        D, I = h.search_adaptive(ts, ses, int(case["topks"][r]), float(case["multipler"][r]), float(case["std_m"][r]),
                                 req, my_np, t_rec, gt_D=gold["gtD"], out=(Dp, Ip))
        assert D is Dp and I is Ip
        assert h.last_direct_out()
        assert np.array_equal(I, gold[f"I_r{r}"])
        assert np.array_equal(bits(D), bits(gold[f"D_r{r}"]))
        assert np.array_equal(my_np[ts:], gold[f"my_nprobe_r{r}"])
    del Dp, Ip, D, I
    hip.hipHostFree(dptr)
    hip.hipHostFree(iptr)


@pytest.mark.parametrize("name", AUNCEL[:2])
def test_asynchronous_searches_equal_the_synchronous_ones(capi, name):
    """amd_ivf_submit_adaptive / amd_ivf_wait: several searches in flight from one caller thread, each on one of the handle's
    internal contexts; results, my_nprobe and t_recalls as the golden outputs, whatever the order the tickets are waited in;
    an error of a search comes back from its wait()"""
    case, gold = load_case(name)
    K, ts, ses = case["max_topk"], case["train_num"], case["test_num"]
    h = make_index(capi, case, gold, gold["centroids"])
    h.set_interdis(None)
    h.set_tuner(K, traces_from_gold(gold), gold["arcos_list"])
    h.set_queries(case["xq"])
    h.set_async_depth(3)
    jobs = []
    for rep in range(2):
        for r in range(len(case["topks"])):
            req = np.full(ts + ses, case["require_acc"][r], dtype=np.float32)
            my_np = np.zeros(ts + ses, dtype=np.uint64)
            t_rec = np.zeros(ts + ses, dtype=np.float32)
            t = h.submit_adaptive(ts, ses, int(case["topks"][r]), float(case["multipler"][r]), float(case["std_m"][r]), req, my_np, t_rec,
                                  gt_D=gold["gtD"])
            jobs.append((t, r, my_np, t_rec))
    for t, r, my_np, t_rec in reversed(jobs):
        D, I, timing, diag = h.wait(t)
        assert np.array_equal(my_np[ts:], gold[f"my_nprobe_r{r}"])
        assert np.array_equal(I, gold[f"I_r{r}"])
        assert np.array_equal(bits(D), bits(gold[f"D_r{r}"]))
        assert np.array_equal(bits(t_rec[ts:]), bits(gold[f"t_recalls_r{r}"]))
        assert timing["total_ms"] > 0 and timing["rounds"] >= 1
    # a failing search: query_topk beyond max_topk is the engine's error, delivered by wait()
    bad = h.submit_adaptive(ts, ses, K + 1, 1.0, 1.0, np.full(ts + ses, 0.9, np.float32), np.zeros(ts + ses, np.uint64),
                            np.zeros(ts + ses, np.float32), gt_D=gold["gtD"])
    with pytest.raises(capi.EngineError, match="query_topk"):
        h.wait(bad)
    with pytest.raises(capi.EngineError, match="ticket"):
        h.wait(bad)
    # the synchronous entry point still works next to it
    req = np.full(ts + ses, case["require_acc"][0], dtype=np.float32)
    my_np = np.zeros(ts + ses, dtype=np.uint64)
    t_rec = np.zeros(ts + ses, dtype=np.float32)
    D, I = h.search_adaptive(ts, ses, int(case["topks"][0]), float(case["multipler"][0]), float(case["std_m"][0]), req, my_np, t_rec,
                             gt_D=gold["gtD"])
    assert np.array_equal(I, gold["I_r0"])
    # depth 0 releases the internal contexts (no ticket out); the next submit starts a pool of the new depth
    h.set_async_depth(0)
    h.set_async_depth(2)
    t = h.submit_adaptive(ts, ses, int(case["topks"][0]), float(case["multipler"][0]), float(case["std_m"][0]), req, my_np, t_rec,
                          gt_D=gold["gtD"])
    with pytest.raises(capi.EngineError, match="tickets"):
        h.set_async_depth(0)
    D2, I2, _, _ = h.wait(t)
    assert np.array_equal(I2, gold["I_r0"]) and np.array_equal(bits(D2), bits(gold["D_r0"]))
    h.close()


@pytest.mark.parametrize("name", ["fixed_sift_l2", "fixed_deep_ip_d96"])
def test_asynchronous_fixed_nprobe_searches(capi, name):
    """amd_ivf_submit_search_resident: fixed-nprobe searches in flight from one caller; same bits as the synchronous call"""
    case, gold = load_case(name)
    h = make_index(capi, case, gold)
    h.set_queries(case["xq"])
    nq, k = case["xq"].shape[0], int(case["ks"][0])
    eD, eI = h.search_resident(0, nq, k, case["nprobe"])
    h.set_async_depth(2)
    tickets = [h.submit_search_resident(0, nq, k, case["nprobe"]) for _ in range(5)]
    for t in tickets:
        D, I, timing, diag = h.wait(t)
        assert np.array_equal(I, eI) and np.array_equal(bits(D), bits(eD))
    h.close()


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("select", [1, 0])
def test_non_finite_distances_in_the_first_list(capi, metric, select):
    """Distances that are +inf / NaN / beyond FLT_MAX among the FIRST candidates of a search: the reference admits a candidate
    only if it is strictly better than the heap top, which starts at (+/-)FLT_MAX (IndexIVFFlat.cpp:129, Heap.h:317-320), so
    such a candidate never enters -- not even into the empty heap (the block-wise fill of the first k did not
    test).  k below and above the number of candidates; both selection forms; stats included."""
    from oracle import pyoracle
    rs = np.random.RandomState(5)
    d, nlist, per = 8, 4, 300
    cen = (rs.rand(nlist, d) * 10).astype(np.float32)
    xb = (cen[np.repeat(np.arange(nlist), per)] + rs.randn(nlist * per, d)).astype(np.float32)
    assign = np.repeat(np.arange(nlist), per).astype(np.int64)
    # the first vectors of every list: overflowing, infinite and NaN distances between finite ones
    for l in range(nlist):
        b = l * per
        xb[b + 0] = 3e19 if metric == 1 else -3e38      # L2: (x - y)^2 overflows to +inf | IP: sum overflows to -inf
        xb[b + 2, 0] = np.nan
        xb[b + 3, 1] = np.inf if metric == 1 else -np.inf
        xb[b + 7] = xb[b + 0]
        xb[b + 150, 3] = np.nan
    xq = (cen[rs.randint(0, nlist, 40)] + rs.randn(40, d)).astype(np.float32)
    if metric == 0:
        xq = np.abs(xq)
    h = capi.Handle(d, nlist, metric, 0)
    h.set_centroids(cen)
    h.set_lists_from_assign(xb, assign)
    h.set_option("select", select)
    lists = pyoracle.Lists(metric, cen, xb, assign)
    od, ok = pyoracle.knn(metric, xq, cen, 3)
    with np.errstate(all="ignore"):
        for k in (10, 100, 128):     # 128 > the 100-candidate log block; all below the 300 candidates of a list
            OD, OI, ost = pyoracle.search_preassigned(lists, xq, k, ok, od)
            h.stats(reset=True)
            D, I = h.search_preassigned(xq, k, ok, od)
            assert np.array_equal(I, OI), (k, np.nonzero((I != OI).any(1))[0][:5])
            assert np.array_equal(bits(D), bits(OD)), k
            st = h.stats()
            assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(ost), k
        # more neighbours asked for than one probe holds finite candidates for (k > n of the first list)
        OD, OI, ost = pyoracle.search_preassigned(lists, xq, 128, ok[:, :1], od[:, :1])
        h.stats(reset=True)
        D, I = h.search_preassigned(xq, 128, ok[:, :1], od[:, :1])
        assert np.array_equal(I, OI) and np.array_equal(bits(D), bits(OD))
        st = h.stats()
        assert [st["nlist"], st["ndis"], st["nheap_updates"]] == list(ost)


def test_options_through_the_abi(capi, monkeypatch):
    """amd_ivf_set_option / amd_ivf_get_option: what was set wins over the debugging environment variable of the same meaning,
    which wins over the default; an unknown key is an engine error (-2); a search context answers for its index"""
    case, gold = load_case(FIXED[0])
    h = make_index(capi, case, gold)
    assert h.get_option("select") == 1 and h.get_option("coarse_ties") == -1 and h.get_option("scan_pipelined") == 7
    monkeypatch.setenv("AUNCEL_AMD_SELECT", "heap")
    assert h.get_option("select") == 0
    h.set_option("select", 1)
    assert h.get_option("select") == 1
    c = h.clone()
    assert c.get_option("select") == 1
    c.set_option("round_first", 6)
    assert h.get_option("round_first") == 6
    h.set_option("select", None)
    assert h.get_option("select") == 0
    monkeypatch.delenv("AUNCEL_AMD_SELECT")
    assert h.get_option("select") == 1
    with pytest.raises(capi.EngineError) as e:
        h.set_option("no_such_option", 1)
    assert e.value.code == -2
    k = int(case["ks"][0])
    for sel in (0, 1):  # both selections give the golden result
        h.set_option("select", sel)
        D, I = h.search_preassigned(case["xq"], k, gold["coarse_keys_sse"], gold["coarse_dis_sse"])
        assert np.array_equal(I, gold[f"I_k{k}"]) and np.array_equal(bits(D), bits(gold[f"D_k{k}"]))


def test_per_xcd_counters_are_exact(capi):
    """the workgroup-scope adds on per-XCD rows that the round planning, the statistics and the end of a search rest on
    (ivf_dev.h: xcd_local_add): under contention every add is counted and every returned value is handed out exactly once"""
    r = capi.self_check(0)
    assert r["counted"] == r["adds"] == 1 << 20 and r["wrong"] == 0
    assert 0 < r["xcd_mask"] < 256  # XCD numbers 0..7 only: the tables have eight rows
