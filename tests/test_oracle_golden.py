"""Pins the CPU restatement (oracle/ivf_oracle.cpp) against outputs of the compiled reference
(tests/golden/*.npz, made by tests/golden/make_golden.py through oracle/_ref/ref_harness).
Bit-exact on ids AND distances for every pinned path."""
import numpy as np
import pytest

from util import AUNCEL, AUNCEL_BIG, FIXED, KMEANS, load_case, traces_from_gold


def _lists(oracle, case, gold, cen=None):
    cen = case["centroids"] if cen is None else cen
    return oracle.Lists(case["metric"], cen, case["xb"], gold["assign"])


@pytest.mark.parametrize("name", FIXED)
def test_coarse_exact_and_assign(oracle, name):
    case, gold = load_case(name)
    if case["d"] % 4 != 0:
        pytest.skip("reference takes the BLAS path when d % 4 != 0")
    D, I = oracle.knn(case["metric"], case["xq"], case["centroids"], case["nprobe"])
    assert np.array_equal(I, gold["coarse_keys_sse"])
    assert np.array_equal(D.view(np.uint32), gold["coarse_dis_sse"].view(np.uint32))
    # add(): every vector goes to its nearest centroid (k=1 through the same code)
    _, a = oracle.knn(case["metric"], case["xb"][:2000], case["centroids"], 1, gemm=False)
    exact = case["name"] if "name" in case else None
    # the reference assigns with nx >= 20 -> BLAS; on integer data that is exact
    if name in ("fixed_sift_l2", "fixed_ragged", "fixed_dups"):
        assert np.array_equal(a[:, 0], gold["assign"][:2000])


@pytest.mark.parametrize("name", FIXED)
def test_coarse_gemm_path_integer_data(oracle, name):
    case, gold = load_case(name)
    if name not in ("fixed_sift_l2", "fixed_odd_d30", "fixed_ragged"):
        pytest.skip("float data: BLAS summation order is unpinned")
    D, I = oracle.knn(case["metric"], case["xq"], case["centroids"], case["nprobe"], gemm=True)
    assert np.array_equal(D, gold["coarse_dis_blas"])
    # ids can differ only inside groups of exactly equal distances
    assert np.array_equal(np.sort(I, axis=1) if False else I, gold["coarse_keys_blas"]) or _ties_only(D, I, gold["coarse_keys_blas"])


def _ties_only(D, I, Iref):
    bad = np.nonzero(I != Iref)
    for q, j in zip(*bad):
        same = D[q] == D[q, j]
        if set(I[q][same]) != set(Iref[q][same]):
            return False
    return True


@pytest.mark.parametrize("name", FIXED)
def test_range_search_preassigned(oracle, name):
    """IndexIVF::range_search_preassigned: lims, labels and distances in the reference's order"""
    case, gold = load_case(name)
    lists = _lists(oracle, case, gold)
    lims, labels, dist, stats = oracle.range_search_preassigned(lists, case["xq"], float(case["radius"][0]), gold["coarse_keys_sse"])
    assert np.array_equal(lims, gold["range_lims"])
    assert np.array_equal(labels, gold["range_labels"][:lims[-1]])
    assert np.array_equal(dist.view(np.uint32), gold["range_distances"][:lims[-1]].view(np.uint32))
    assert list(stats) == list(gold["range_stats"])


@pytest.mark.parametrize("name", FIXED)
def test_search_preassigned(oracle, name):
    case, gold = load_case(name)
    lists = _lists(oracle, case, gold)
    assert np.array_equal(lists.sizes, gold["list_sizes"])
    for k in case["ks"]:
        for pairs in (False, True):
            suf = f"_k{k}" + ("_pairs" if pairs else "")
            D, I, st = oracle.search_preassigned(lists, case["xq"], int(k), gold["coarse_keys_sse"],
                                                 gold["coarse_dis_sse"], store_pairs=pairs)
            assert np.array_equal(I, gold["I" + suf]), suf
            assert np.array_equal(D.view(np.uint32), gold["D" + suf].view(np.uint32)), suf
            assert np.array_equal(st, gold["stats" + suf]), suf
        if case["max_codes"]:
            D, I, _ = oracle.search_preassigned(lists, case["xq"], int(k), gold["coarse_keys_sse"],
                                                gold["coarse_dis_sse"], max_codes=case["max_codes"])
            assert np.array_equal(I, gold[f"I_k{k}_maxcodes"])
            assert np.array_equal(D.view(np.uint32), gold[f"D_k{k}_maxcodes"].view(np.uint32))


@pytest.mark.parametrize("name", FIXED)
def test_scanner_raw_heap(oracle, name):
    case, gold = load_case(name)
    lists = _lists(oracle, case, gold)
    k = int(case["ks"][0])
    ns = gold["scan_heap_D"].shape[0]
    is_l2 = case["metric"] == 1
    for i in range(ns):
        simi = np.full(k, np.finfo(np.float32).max if is_l2 else -np.finfo(np.float32).max, dtype=np.float32)
        idxi = np.full(k, -1, dtype=np.int64)
        for p in range(case["nprobe"]):
            key = int(gold["coarse_keys_sse"][i, p])
            if key < 0:
                continue
            b, e = int(lists.off[key]), int(lists.off[key + 1])
            if e == b:
                continue
            nup = oracle.scan_codes(case["metric"], case["xq"][i], lists.codes[b:e], lists.ids[b:e], key, False, simi, idxi)
            assert nup == gold["scan_nup"][i, p]
            d0 = (oracle.lib().orc_fvec_L2sqr if is_l2 else oracle.lib().orc_fvec_inner_product)(
                oracle._f(oracle.f32(case["xq"][i])), oracle._f(lists.codes[b:b + 1]), oracle.C.c_size_t(case["d"]))
            assert np.float32(d0) == gold["scan_dist_to_code"][i, p]
        assert np.array_equal(simi.view(np.uint32), gold["scan_heap_D"][i].view(np.uint32))
        assert np.array_equal(idxi, gold["scan_heap_I"][i])


@pytest.mark.parametrize("name", [n for n in FIXED if n not in ("fixed_gist_l2_d960", "fixed_odd_d30")])
def test_shards_merge(oracle, name):
    case, gold = load_case(name)
    nshard = case["nshard"]
    a = gold["assign"]
    for k in case["ks"]:
        allD, allI = [], []
        for s in range(nshard):
            sub = oracle.Lists(case["metric"], case["centroids"], case["xb"], np.where(a % nshard == s, a, -1))
            D, I, _ = oracle.search_preassigned(sub, case["xq"], int(k), gold["coarse_keys_sse"], gold["coarse_dis_sse"])
            allD.append(D)
            allI.append(I)
        D, I = oracle.merge_tables(case["metric"], np.stack(allD), np.stack(allI))
        assert np.array_equal(I, gold[f"I_shards_k{k}"])
        assert np.array_equal(D.view(np.uint32), gold[f"D_shards_k{k}"].view(np.uint32))


@pytest.mark.parametrize("name", AUNCEL)
def test_auncel_offline(oracle, name):
    case, gold = load_case(name)
    K, ts = case["max_topk"], case["train_num"]
    cen = gold["centroids"]
    assert np.array_equal(oracle.interdis(case["metric"], cen).view(np.uint32), gold["interdis_cem"].view(np.uint32))
    assert np.array_equal(oracle.arcos_table().view(np.uint32), gold["arcos_list"].view(np.uint32))
    lists = _lists(oracle, case, gold, cen)
    ntr = len(traces_from_gold(gold))
    raw = [np.full((ts * (K // 4), 2), -1, dtype=np.float32) for _ in range(ntr)]
    # the reference trains in 10 batches through the BLAS coarse path: feed its coarse output
    D, I = oracle.train_samples(lists, case["xq"][:ts], K, gold["coarse_keys_blas_train"], gold["coarse_dis_blas_train"],
                                gold["interdis_cem"], gold["arcos_list"], gold["gtD"], 0, ts, raw)
    assert np.array_equal(I, gold["train_I"])
    assert np.array_equal(D.view(np.uint32), gold["train_D"].view(np.uint32))
    for i in range(ntr):
        assert np.array_equal(raw[i].view(np.uint32), gold[f"raw_trace{i}"].view(np.uint32)), i
        x, y, s = oracle.trace_sb(gold[f"raw_trace{i}"])
        assert np.array_equal(x.view(np.uint32), gold[f"sb_trace{i}"][:, 0].view(np.uint32)), i
        assert np.array_equal(y.view(np.uint32), gold[f"sb_trace{i}"][:, 1].view(np.uint32)), i
        assert np.array_equal(s.view(np.uint32), gold[f"sb_stds{i}"].view(np.uint32)), i


@pytest.mark.parametrize("name", AUNCEL)
def test_auncel_set_online(oracle, name):
    case, gold = load_case(name)
    ts, nlist = case["train_num"], case["nlist"]
    for i in range(0, case["test_num"], 7):
        dtb, c2c = oracle.set_online(case["metric"], nlist, gold["coarse_dis_sse"][ts + i], gold["coarse_keys_sse"][ts + i],
                                     gold["interdis_cem"], gold["arcos_list"])
        assert np.array_equal(dtb.view(np.uint32), gold["disToBoundary"][i].view(np.uint32))
        assert np.array_equal(c2c.view(np.uint32), gold["cenTocen"][i].view(np.uint32))


@pytest.mark.parametrize("name", AUNCEL)
def test_auncel_online(oracle, name):
    case, gold = load_case(name)
    K, ts, ses = case["max_topk"], case["train_num"], case["test_num"]
    lists = _lists(oracle, case, gold, gold["centroids"])
    traces = traces_from_gold(gold)
    for r in range(len(case["topks"])):
        for prof in (False, True):
            tun = oracle.Tuner(gold["interdis_cem"], traces, K, ts + ses, arcos=gold["arcos_list"])
            req = np.full(ts + ses, case["require_acc"][r], dtype=np.float32)
            st = tun.struct(int(case["topks"][r]), req, float(case["multipler"][r]), float(case["std_m"][r]),
                            gt_D=gold["gtD"], profile=prof)
            D, I, stats = oracle.search_preassigned(lists, case["xq"][ts:], K, gold["coarse_keys_sse"][ts:],
                                                    gold["coarse_dis_sse"][ts:], tuner=st, offset=ts)
            suf = f"_r{r}" + ("_prof" if prof else "")
            assert np.array_equal(tun.my_nprobe[ts:].astype(np.uint64), gold["my_nprobe" + suf]), suf
            assert np.array_equal(I, gold["I" + suf]), suf
            assert np.array_equal(D.view(np.uint32), gold["D" + suf].view(np.uint32)), suf
            assert np.array_equal(tun.t_recalls[ts:].view(np.uint32), gold["t_recalls" + suf].view(np.uint32)), suf
            assert np.array_equal(stats, gold["stats" + suf]), suf


@pytest.mark.parametrize("name", AUNCEL + AUNCEL_BIG)
def test_auncel_overhead_profile(oracle, name):
    """error_pro::overhead_profile (eval/overhead.cpp:284-290): the rule runs on every probe, its verdict is ignored, the
    probe loop ends at stage nlist / 8"""
    case, gold = load_case(name)
    K, ts, ses = case["max_topk"], case["train_num"], case["test_num"]
    cen = gold["centroids"]
    inter = gold["interdis_cem"] if "interdis_cem" in gold else oracle.interdis(case["metric"], cen)
    ck = gold["coarse_keys_sse"][ts:] if "coarse_keys_sse" in gold else gold["coarse_keys_sse_test"]
    cd = gold["coarse_dis_sse"][ts:] if "coarse_dis_sse" in gold else gold["coarse_dis_sse_test"]
    lists = _lists(oracle, case, gold, cen)
    tun = oracle.Tuner(inter, traces_from_gold(gold), K, ts + ses, arcos=gold["arcos_list"])
    req = np.full(ts + ses, case["require_acc"][0], dtype=np.float32)
    st = tun.struct(int(case["topks"][0]), req, float(case["multipler"][0]), float(case["std_m"][0]), gt_D=gold["gtD"], overhead_profile=True)
    D, I, stats = oracle.search_preassigned(lists, case["xq"][ts:], K, ck, cd, tuner=st, offset=ts)
    assert np.array_equal(tun.my_nprobe[ts:].astype(np.uint64), gold["my_nprobe_overhead"]) and not gold["my_nprobe_overhead"].any()
    assert np.array_equal(I, gold["I_overhead"])
    assert np.array_equal(D.view(np.uint32), gold["D_overhead"].view(np.uint32))
    assert np.array_equal(stats, gold["stats_overhead"])


@pytest.mark.parametrize("name", AUNCEL_BIG)
def test_auncel_online_nlist4096(oracle, name):
    """BASELINE config 2's shape (IVF4096: max_num 532, ten traces): the restatement against the compiled reference's run.
    The O(nlist^2) table is pinned through its digest, the coarse ranking through the columns the probe loop can reach."""
    import hashlib
    case, gold = load_case(name)
    K, ts, ses, nlist = case["max_topk"], case["train_num"], case["test_num"], case["nlist"]
    cen = gold["centroids"]
    inter = oracle.interdis(case["metric"], cen)
    assert hashlib.sha256(np.ascontiguousarray(inter).tobytes()).hexdigest() == str(gold["interdis_cem_sha"])
    assert np.array_equal(inter[::4099].view(np.uint32), gold["interdis_cem_sample"].view(np.uint32))
    lists = _lists(oracle, case, gold, cen)
    traces = traces_from_gold(gold)
    assert len(traces) == 10
    ck, cd = gold["coarse_keys_sse_test"], gold["coarse_dis_sse_test"]
    for i in range(0, ses, 13):
        dtb, _ = oracle.set_online(case["metric"], nlist, cd[i], ck[i], inter, gold["arcos_list"])
        assert np.array_equal(dtb.view(np.uint32), gold["disToBoundary"][i].view(np.uint32))
    for r in range(len(case["topks"])):
        for prof in (False, True):
            tun = oracle.Tuner(inter, traces, K, ts + ses, arcos=gold["arcos_list"])
            req = np.full(ts + ses, case["require_acc"][r], dtype=np.float32)
            st = tun.struct(int(case["topks"][r]), req, float(case["multipler"][r]), float(case["std_m"][r]), gt_D=gold["gtD"], profile=prof)
            D, I, stats = oracle.search_preassigned(lists, case["xq"][ts:], K, ck, cd, tuner=st, offset=ts)
            suf = f"_r{r}" + ("_prof" if prof else "")
            assert np.array_equal(tun.my_nprobe[ts:].astype(np.uint64), gold["my_nprobe" + suf]), suf
            assert np.array_equal(I, gold["I" + suf]), suf
            assert np.array_equal(D.view(np.uint32), gold["D" + suf].view(np.uint32)), suf
            assert np.array_equal(tun.t_recalls[ts:].view(np.uint32), gold["t_recalls" + suf].view(np.uint32)), suf
            assert np.array_equal(stats, gold["stats" + suf]), suf


@pytest.mark.parametrize("name", KMEANS)
def test_kmeans(oracle, name):
    """Clustering::train over an IndexFlat (what Level1Quantizer::train_q1 runs): sub-sampling and seeding permutations,
    fp32 centroid sums in point order, void-cluster splitting, spherical post-processing.  Centroids bit for bit on every
    case (on the integer-valued case the reference's BLAS assignment picks the same centroids as the exact kernel); the
    objective bit for bit where the reference assigns through its exact path (fewer than 20 points)."""
    case, gold = load_case(name)
    cen, obj = oracle.kmeans(case["metric"], case["x"], case["k"], case["niter"], case["seed"], case["max_points_per_centroid"],
                             bool(case["spherical"]))
    assert np.array_equal(cen.view(np.uint32), gold["centroids"].view(np.uint32))
    if case["x"].shape[0] < 20:
        assert np.array_equal(obj.view(np.uint32), gold["obj"].view(np.uint32))
    else:
        assert np.allclose(obj, gold["obj"], rtol=1e-5)


@pytest.mark.parametrize("name", ["auncel_sift_d32"])
def test_reference_bench_mode(oracle, name):
    """bench.py's cpu_baseline (kind "reference") runs the compiled reference through ref_harness's `bench` mode on an index
    assembled from lists and traces handed over by the caller: same (D, I, my_nprobe) as the reference's own end-to-end run"""
    from oracle import refbench
    if not refbench.available():
        pytest.skip("oracle/_ref/ref_harness not built (reference sources absent)")
    case, gold = load_case(name)
    ts, ses, K = case["train_num"], case["test_num"], case["max_topk"]
    lists = oracle.Lists(case["metric"], gold["centroids"], case["xb"], gold["assign"])
    r = 0
    out = refbench.run(gold["centroids"], lists.off, lists.codes, lists.ids, traces_from_gold(gold), case["xq"][ts:ts + ses], ts, K,
                       int(case["topks"][r]), float(case["require_acc"][r]), float(case["multipler"][r]), float(case["std_m"][r]),
                       single_thread_queries=8, threads=4)
    assert np.array_equal(out["I"], gold[f"I_r{r}"])
    assert np.array_equal(out["D"].view(np.uint32), gold[f"D_r{r}"].view(np.uint32))
    assert np.array_equal(out["my_nprobe"], gold[f"my_nprobe_r{r}"])
    assert out["seconds_one_thread"] > 0 and out["seconds_all_threads"] > 0 and out["threads"] == 4
