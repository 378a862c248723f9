// Test driver for the host-side C++ mirror (auncel_amd/csrc/host): runs the flows of the reference's
// own callers -- tests/test_lowlevel_ivf.cpp (scanner API vs search), IndexShards, and eval/bound.cpp
// (Error_sys train -> set_queries -> one search() per query) -- on a bundle prepared by
// tests/test_host_mirror.py and compares with the expected tensors stored in it.
// usage: host_mirror_driver <fixed|auncel> <bundle.tb>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <memory>
#include <thread>

#include "../../auncel_amd/csrc/host/AutoTune.h"
#include "../../auncel_amd/csrc/host/AuxIndexStructures.h"
#include "../../auncel_amd/csrc/host/Clustering.h"
#include "../../auncel_amd/csrc/host/FaissException.h"
#include "../../auncel_amd/csrc/host/Heap.h"
#include "../../auncel_amd/csrc/host/IndexFlat.h"
#include "../../auncel_amd/csrc/host/IndexIVFFlat.h"
#include "../../auncel_amd/csrc/host/IndexShards.h"
#include "../../auncel_amd/csrc/host/profile.h"
#include "../../oracle/tbundle.h"

using namespace faiss;
typedef Index::idx_t idx_t;

static int g_fail = 0;
static void expect(bool ok, const std::string& what) {
    if (!ok) {
        printf("MISMATCH: %s\n", what.c_str());
        g_fail++;
    }
}
static bool same_f(const float* a, const float* b, size_t n) { return memcmp(a, b, n * 4) == 0; }
static bool same_i(const idx_t* a, const int64_t* b, size_t n) { return memcmp(a, b, n * 8) == 0; }

static int run_fixed(const tb::Bundle& in) {
    size_t d = in.scalar<size_t>("d"), nlist = in.scalar<size_t>("nlist"), nprobe = in.scalar<size_t>("nprobe");
    MetricType mt = in.scalar<int>("metric") == 0 ? METRIC_INNER_PRODUCT : METRIC_L2;
    const tb::Tensor &cen = in.get("centroids"), &xb = in.get("xb"), &xq = in.get("xq"), &ks = in.get("ks");
    size_t nb = xb.dims[0], nq = xq.dims[0];

    std::unique_ptr<Index> index(index_factory((int)d, ("IVF" + std::to_string(nlist) + ",Flat").c_str(), mt));
    IndexIVF* ix = dynamic_cast<IndexIVF*>(index.get());
    expect(ix != nullptr && index->type == IVF, "index_factory type");
    ix->quantizer->add(nlist, cen.as<float>());
    ix->is_trained = true;
    dynamic_cast<IndexFlat*>(ix->quantizer)->coarse_mode = 0;  // expectations come from the exact coarse path
    ix->coarse_mode = 0;
    index->add(nb / 3, xb.as<float>());
    index->add(nb - nb / 3, xb.as<float>() + (nb / 3) * d);
    expect((size_t)index->ntotal == nb, "ntotal");
    const int64_t* ls = in.get("list_sizes").as<int64_t>();
    bool sizes_ok = true;
    for (size_t l = 0; l < nlist; l++) sizes_ok &= ix->invlists->list_size(l) == (size_t)ls[l];
    expect(sizes_ok, "list sizes after add");
    ix->nprobe = nprobe;

    for (size_t ki = 0; ki < ks.numel(); ki++) {
        size_t k = ks.as<int64_t>()[ki];
        std::string suf = "_k" + std::to_string(k);
        std::vector<float> D(nq * k);
        std::vector<idx_t> I(nq * k);
        indexIVF_stats.reset();
        index->search(nq, xq.as<float>(), k, D.data(), I.data());
        expect(same_i(I.data(), in.get("I" + suf).as<int64_t>(), nq * k), "search ids" + suf);
        expect(same_f(D.data(), in.get("D" + suf).as<float>(), nq * k), "search distances" + suf);
        const int64_t* st = in.get("stats" + suf).as<int64_t>();
        expect(indexIVF_stats.nq == nq && indexIVF_stats.nlist == (size_t)st[0] && indexIVF_stats.ndis == (size_t)st[1] &&
                   indexIVF_stats.nheap_updates == (size_t)st[2], "indexIVF_stats" + suf);
        // search_preassigned with the quantizer's own output, store_pairs
        std::vector<float> cd(nq * nprobe);
        std::vector<idx_t> ck(nq * nprobe);
        ix->quantizer->search(nq, xq.as<float>(), nprobe, cd.data(), ck.data());
        ix->search_preassigned(nq, xq.as<float>(), k, ck.data(), cd.data(), D.data(), I.data(), true);
        expect(same_i(I.data(), in.get("I" + suf + "_pairs").as<int64_t>(), nq * k), "store_pairs ids" + suf);

        // tests/test_lowlevel_ivf.cpp:82-220: manual quantizer->search + scanner == index->search
        std::unique_ptr<InvertedListScanner> sc(ix->get_InvertedListScanner());
        const int64_t* Iref = in.get("I" + suf).as<int64_t>();
        for (size_t i = 0; i < std::min<size_t>(nq, 4); i++) {
            std::vector<float> simi(k);
            std::vector<idx_t> idxi(k);
            if (mt == METRIC_L2) maxheap_heapify(k, simi.data(), idxi.data()); else minheap_heapify(k, simi.data(), idxi.data());
            sc->set_query(xq.as<float>() + i * d);
            for (size_t p = 0; p < nprobe; p++) {
                idx_t key = ck[i * nprobe + p];
                if (key < 0 || ix->invlists->list_size(key) == 0) continue;
                sc->set_list(key, cd[i * nprobe + p]);
                InvertedLists::ScopedCodes codes(ix->invlists, key);
                InvertedLists::ScopedIds ids(ix->invlists, key);
                sc->scan_codes(ix->invlists->list_size(key), codes.get(), ids.get(), simi.data(), idxi.data(), k);
            }
            if (mt == METRIC_L2) maxheap_reorder(k, simi.data(), idxi.data()); else minheap_reorder(k, simi.data(), idxi.data());
            expect(same_i(idxi.data(), Iref + i * k, k), "scanner ids == search ids, query " + std::to_string(i));
        }
    }

    {   // tests/test_lowlevel_ivf.cpp:426-564 (ThreadedSearch): the probes of a query dealt to three threads, one scanner and one
        // result heap per thread, heaps merged with heap_addn -- and, beyond the reference's test, every list handed over in two
        // HALVES (scan_codes scans whatever run of codes it is handed, IndexIVFFlat.cpp:117-137)
        const size_t k = ks.as<int64_t>()[0];
        const int64_t* Iref = in.get("I_k" + std::to_string(k)).as<int64_t>();
        const float* Dref = in.get("D_k" + std::to_string(k)).as<float>();
        std::vector<float> cd(nq * nprobe);
        std::vector<idx_t> ck(nq * nprobe);
        ix->quantizer->search(nq, xq.as<float>(), nprobe, cd.data(), ck.data());
        const int nproc = 3;
        const bool l2 = mt == METRIC_L2;
        for (size_t i = 0; i < std::min<size_t>(nq, 6); i++) {
            std::vector<idx_t> I(k * nproc, -1);
            std::vector<float> D(k * nproc, l2 ? HUGE_VALF : -HUGE_VALF);
            auto work = [&](int rank) {
                std::unique_ptr<InvertedListScanner> sc(ix->get_InvertedListScanner());
                sc->set_query(xq.as<float>() + i * d);
                for (size_t j = rank; j < nprobe; j += nproc) {
                    const idx_t key = ck[i * nprobe + j];
                    if (key < 0) continue;
                    sc->set_list(key, cd[i * nprobe + j]);
                    const size_t sz = ix->invlists->list_size(key), half = sz / 2;
                    InvertedLists::ScopedCodes codes(ix->invlists, key);
                    InvertedLists::ScopedIds ids(ix->invlists, key);
                    sc->scan_codes(half, codes.get(), ids.get(), D.data() + rank * k, I.data() + rank * k, k);
                    sc->scan_codes(sz - half, codes.get() + half * ix->code_size, ids.get() + half, D.data() + rank * k, I.data() + rank * k, k);
                }
            };
            std::vector<std::thread> th;
            for (int r = 0; r < nproc; r++) th.emplace_back(work, r);
            for (int r = 0; r < nproc; r++) {
                th[r].join();
                if (r == 0) continue;
                if (l2) maxheap_addn(k, D.data(), I.data(), D.data() + r * k, I.data() + r * k, k);
                else minheap_addn(k, D.data(), I.data(), D.data() + r * k, I.data() + r * k, k);
            }
            if (l2) maxheap_reorder(k, D.data(), I.data()); else minheap_reorder(k, D.data(), I.data());
            // (threads + merge order equal distances differently from the serial loop: the reference's test data has no ties either;
            // compare the distances, and the ids where a distance is unique)
            bool ok = same_f(D.data(), Dref + i * k, k);
            for (size_t j = 0; j < k && ok; j++) {
                const bool tie = (j > 0 && D[j] == D[j - 1]) || (j + 1 < k && D[j] == D[j + 1]);
                if (!tie) ok = I[j] == Iref[i * k + j];
            }
            expect(ok, "threaded scanners over list halves == search, query " + std::to_string(i));
        }
        // one scanner, serial, lists in three parts of uneven length: the same heap as the whole lists give (ids AND order), through
        // the list's own ids, through a caller-owned id array, and with store_pairs (j counts from the part's first code)
        for (size_t i = 0; i < std::min<size_t>(nq, 3); i++) {
            std::unique_ptr<InvertedListScanner> sc(ix->get_InvertedListScanner()), sp(ix->get_InvertedListScanner(true));
            std::vector<float> s0(k), s1(k), s2(k);
            std::vector<idx_t> i0(k), i1(k), i2(k);
            auto heapify = [&](std::vector<float>& v, std::vector<idx_t>& l) { if (l2) maxheap_heapify(k, v.data(), l.data()); else minheap_heapify(k, v.data(), l.data()); };
            heapify(s0, i0);
            heapify(s1, i1);
            heapify(s2, i2);
            sc->set_query(xq.as<float>() + i * d);
            sp->set_query(xq.as<float>() + i * d);
            bool pairs_ok = true;
            for (size_t p = 0; p < nprobe; p++) {
                const idx_t key = ck[i * nprobe + p];
                if (key < 0 || ix->invlists->list_size(key) == 0) continue;
                sc->set_list(key, cd[i * nprobe + p]);
                sp->set_list(key, cd[i * nprobe + p]);
                const size_t sz = ix->invlists->list_size(key);
                const size_t cut[4] = {0, sz / 5, sz / 5 + (sz - sz / 5) / 2, sz};
                const uint8_t* codes = ix->invlists->get_codes(key);
                const idx_t* ids = ix->invlists->get_ids(key);
                std::vector<idx_t> mine(ids, ids + sz);
                for (idx_t& v : mine) v += 1000000;
                for (int part = 0; part < 3; part++) {
                    const size_t a = cut[part], n = cut[part + 1] - cut[part];
                    sc->scan_codes(n, codes + a * ix->code_size, ids + a, s0.data(), i0.data(), k);
                    sc->scan_codes(n, codes + a * ix->code_size, mine.data() + a, s1.data(), i1.data(), k);
                    std::vector<idx_t> before(i2);
                    sp->scan_codes(n, codes + a * ix->code_size, nullptr, s2.data(), i2.data(), k);
                    // what this part admitted carries key << 32 | j with j counted from the part's first code (IndexIVFFlat.cpp:131)
                    for (size_t s = 0; s < k; s++) {
                        const size_t j = (size_t)(i2[s] & 0xffffffffll);
                        const bool here = s2[s] == s0[s] && i2[s] >= 0 && (i2[s] >> 32) == key && j < n && ids[j + a] == i0[s];
                        const bool empty = i2[s] == -1 && i0[s] == -1;
                        const bool earlier = std::find(before.begin(), before.end(), i2[s]) != before.end();  // (an earlier part or list)
                        if (!(here || empty || earlier)) pairs_ok = false;
                    }
                }
            }
            if (l2) { maxheap_reorder(k, s0.data(), i0.data()); maxheap_reorder(k, s1.data(), i1.data()); }
            else { minheap_reorder(k, s0.data(), i0.data()); minheap_reorder(k, s1.data(), i1.data()); }
            expect(same_i(i0.data(), Iref + i * k, k) && same_f(s0.data(), Dref + i * k, k), "scanner over list parts == search, query " + std::to_string(i));
            bool mine_ok = same_f(s1.data(), Dref + i * k, k);
            for (size_t j = 0; j < k; j++) mine_ok &= i1[j] == (Iref[i * k + j] < 0 ? -1 : Iref[i * k + j] + 1000000);
            expect(mine_ok, "scanner with the caller's own id array, query " + std::to_string(i));
            expect(pairs_ok, "store_pairs labels count from the part's first code, query " + std::to_string(i));
        }
        // a code pointer that is not in the current list is refused (the lists live in HBM)
        {
            std::unique_ptr<InvertedListScanner> sc(ix->get_InvertedListScanner());
            sc->set_query(xq.as<float>());
            size_t l0 = 0;
            while (l0 < nlist && ix->invlists->list_size(l0) == 0) l0++;
            sc->set_list((idx_t)l0, 0.f);
            std::vector<float> sv(k);
            std::vector<idx_t> iv(k);
            bool threw = false;
            try { sc->scan_codes(1, reinterpret_cast<const uint8_t*>(xq.as<float>()), nullptr, sv.data(), iv.data(), k); } catch (const FaissException&) { threw = true; }
            expect(threw, "foreign code pointer refused");
            threw = false;
            try { sc->scan_codes(ix->invlists->list_size(l0) + 1, ix->invlists->get_codes(l0), ix->invlists->get_ids(l0), sv.data(), iv.data(), k); } catch (const FaissException&) { threw = true; }
            expect(threw, "run past the end of the list refused");
        }
    }

    {   // search_and_reconstruct / reconstruct_n / make_direct_map + reconstruct (IndexIVF.cpp:305-328,869-938)
        const size_t k = ks.as<int64_t>()[0];
        std::vector<float> D(nq * k), R(nq * k * d);
        std::vector<idx_t> I(nq * k);
        index->search_and_reconstruct(nq, xq.as<float>(), k, D.data(), I.data(), R.data());
        expect(same_i(I.data(), in.get("I_k" + std::to_string(k)).as<int64_t>(), nq * k), "search_and_reconstruct ids");
        bool rec_ok = true;
        for (size_t i = 0; i < nq * k; i++) {
            if (I[i] < 0) {
                for (size_t j = 0; j < d; j++) rec_ok &= std::isnan(R[i * d + j]);
            } else {
                rec_ok &= memcmp(&R[i * d], xb.as<float>() + (size_t)I[i] * d, d * sizeof(float)) == 0;
            }
        }
        expect(rec_ok, "search_and_reconstruct vectors");
        const size_t n0 = nb / 3, nn = std::min<size_t>(50, nb - n0);
        std::vector<float> rn(nn * d);
        index->reconstruct_n(n0, nn, rn.data());
        expect(memcmp(rn.data(), xb.as<float>() + n0 * d, nn * d * sizeof(float)) == 0, "reconstruct_n");
        ix->make_direct_map(true);
        std::vector<float> r1(d);
        index->reconstruct((idx_t)(nb - 1), r1.data());
        expect(memcmp(r1.data(), xb.as<float>() + (nb - 1) * d, d * sizeof(float)) == 0, "reconstruct via direct map");
        ix->make_direct_map(false);
    }

    if (in.has("range_lims") && d % 4 == 0) {  // IndexIVF::range_search through the class mirror (RangeSearchResult as in the reference)
        const float radius = in.get("radius").as<float>()[0];
        RangeSearchResult res(nq);
        indexIVF_stats.reset();
        index->range_search(nq, xq.as<float>(), radius, &res);
        const int64_t* gl = in.get("range_lims").as<int64_t>();
        bool lims_ok = true;
        for (size_t i = 0; i <= nq; i++) lims_ok &= (int64_t)res.lims[i] == gl[i];
        expect(lims_ok, "range_search lims");
        if (lims_ok) {
            expect(same_i(res.labels, in.get("range_labels").as<int64_t>(), res.lims[nq]), "range_search labels");
            expect(same_f(res.distances, in.get("range_distances").as<float>(), res.lims[nq]), "range_search distances");
        }
        const int64_t* st = in.get("range_stats").as<int64_t>();
        expect(indexIVF_stats.nlist == (size_t)st[0] && indexIVF_stats.ndis == (size_t)st[1], "range_search stats");
    }

    if (in.has("range_lims") && d % 4 == 0) {
        // IndexIVF::range_search_preassigned as the reference writes it (IndexIVF.cpp:760-857): one RangeSearchPartialResult, a
        // RangeQueryResult per query, InvertedListScanner::scan_codes_range per probed list -- here in two halves per list
        const float radius = in.get("radius").as<float>()[0];
        std::vector<float> cd(nq * nprobe);
        std::vector<idx_t> ck(nq * nprobe);
        ix->quantizer->search(nq, xq.as<float>(), nprobe, cd.data(), ck.data());
        const size_t nsub = std::min<size_t>(nq, 8);
        const int64_t* gl = in.get("range_lims").as<int64_t>();
        for (int pairs = 0; pairs < 2; pairs++) {
            RangeSearchResult res(nsub);
            RangeSearchPartialResult pres(&res);
            std::unique_ptr<InvertedListScanner> sc(ix->get_InvertedListScanner(pairs != 0));
            std::vector<idx_t> expect_pairs;
            for (size_t i = 0; i < nsub; i++) {
                sc->set_query(xq.as<float>() + i * d);
                RangeQueryResult& qres = pres.new_result((idx_t)i);
                for (size_t p = 0; p < nprobe; p++) {
                    const idx_t key = ck[i * nprobe + p];
                    if (key < 0 || ix->invlists->list_size(key) == 0) continue;
                    sc->set_list(key, cd[i * nprobe + p]);
                    const size_t sz = ix->invlists->list_size(key), half = (sz + 1) / 2;
                    const uint8_t* codes = ix->invlists->get_codes(key);
                    const idx_t* ids = ix->invlists->get_ids(key);
                    sc->scan_codes_range(half, codes, ids, radius, qres);
                    sc->scan_codes_range(sz - half, codes + half * ix->code_size, ids + half, radius, qres);
                }
            }
            pres.finalize();
            bool ok = true;
            for (size_t i = 0; i <= nsub; i++) ok &= (int64_t)res.lims[i] == gl[i];
            expect(ok, std::string("scan_codes_range lims") + (pairs ? " (store_pairs)" : ""));
            if (!ok) continue;
            expect(same_f(res.distances, in.get("range_distances").as<float>(), res.lims[nsub]), std::string("scan_codes_range distances") + (pairs ? " (store_pairs)" : ""));
            if (!pairs) {
                expect(same_i(res.labels, in.get("range_labels").as<int64_t>(), res.lims[nsub]), "scan_codes_range labels");
            } else {
                // list << 32 | j, j from the half's first code: the stored id at that place is the golden label
                const int64_t* glab = in.get("range_labels").as<int64_t>();
                bool pok = true;
                for (size_t e = 0; e < res.lims[nsub] && pok; e++) {
                    const idx_t key = res.labels[e] >> 32, j = res.labels[e] & 0xffffffffll;
                    const size_t sz = ix->invlists->list_size(key), half = (sz + 1) / 2;
                    const idx_t* ids = ix->invlists->get_ids(key);
                    pok = ((size_t)j < half && ids[j] == glab[e]) || ((size_t)j + half < sz && ids[j + half] == glab[e]);
                }
                expect(pok, "scan_codes_range store_pairs labels");
            }
        }
        // RangeSearchPartialResult::merge over two partial results that split the queries (parallel_mode != 0's ending, IndexIVF.cpp:846-851)
        {
            RangeSearchResult res(nsub);
            std::vector<RangeSearchPartialResult*> parts{new RangeSearchPartialResult(&res), new RangeSearchPartialResult(&res)};
            std::unique_ptr<InvertedListScanner> sc(ix->get_InvertedListScanner());
            for (size_t i = 0; i < nsub; i++) {
                sc->set_query(xq.as<float>() + i * d);
                RangeQueryResult& qres = parts[i & 1]->new_result((idx_t)i);
                for (size_t p = 0; p < nprobe; p++) {
                    const idx_t key = ck[i * nprobe + p];
                    if (key < 0 || ix->invlists->list_size(key) == 0) continue;
                    sc->set_list(key, cd[i * nprobe + p]);
                    sc->scan_codes_range(ix->invlists->list_size(key), ix->invlists->get_codes(key), ix->invlists->get_ids(key), radius, qres);
                }
            }
            RangeSearchPartialResult::merge(parts);
            bool ok = true;
            for (size_t i = 0; i <= nsub; i++) ok &= (int64_t)res.lims[i] == gl[i];
            expect(ok && same_i(res.labels, in.get("range_labels").as<int64_t>(), res.lims[nsub]) &&
                       same_f(res.distances, in.get("range_distances").as<float>(), res.lims[nsub]), "RangeSearchPartialResult::merge");
        }
    }

    if (in.scalar_or<int>("dedup", 0) > 0) {  // IndexIVFFlatDedup against the compiled reference's (same two add() calls)
        IndexIVFFlatDedup dd(ix->quantizer, d, nlist, mt);
        dd.coarse_mode = 0;
        const size_t half = nb / 2;
        dd.add(half, xb.as<float>());
        dd.add(nb - half, xb.as<float>() + half * d);
        dd.nprobe = nprobe;
        const int64_t* dls = in.get("dedup_list_sizes").as<int64_t>();
        bool ok = true;
        for (size_t l = 0; l < nlist; l++) ok &= dd.invlists->list_size(l) == (size_t)dls[l];
        expect(ok, "dedup list sizes");
        const int64_t* tot = in.get("dedup_ntotal_ninst").as<int64_t>();
        expect(dd.ntotal == tot[0] && (int64_t)dd.instances.size() == tot[1], "dedup ntotal / instances");
        const tb::Tensor &ck = in.get("coarse_keys_sse"), &cd = in.get("coarse_dis_sse");
        std::vector<idx_t> keys(ck.as<int64_t>(), ck.as<int64_t>() + nq * nprobe);
        for (size_t ki = 0; ki < ks.numel(); ki++) {
            size_t k = ks.as<int64_t>()[ki];
            std::string suf = "_dedup_k" + std::to_string(k);
            std::vector<float> D(nq * k);
            std::vector<idx_t> I(nq * k);
            dd.search_preassigned(nq, xq.as<float>(), k, keys.data(), cd.as<float>(), D.data(), I.data(), false);
            expect(same_i(I.data(), in.get("I" + suf).as<int64_t>(), nq * k), "dedup search_preassigned ids" + suf);
            expect(same_f(D.data(), in.get("D" + suf).as<float>(), nq * k), "dedup search_preassigned distances" + suf);
            dd.search(nq, xq.as<float>(), k, D.data(), I.data());  // coarse ranking on the device (exact kernel)
            expect(same_i(I.data(), in.get("I" + suf).as<int64_t>(), nq * k), "dedup search ids" + suf);
            expect(same_f(D.data(), in.get("D" + suf).as<float>(), nq * k), "dedup search distances" + suf);
        }
        bool threw = false;
        try { dd.search_preassigned(1, xq.as<float>(), 1, keys.data(), cd.as<float>(), nullptr, nullptr, true); } catch (const FaissException&) { threw = true; }
        expect(threw, "dedup store_pairs throws");
        // IndexIVFFlatDedup::remove_ids (IndexIVFFlat.cpp:381-448): the first half of the ids goes; a stored vector whose id goes
        // lives on under a surviving copy's id.  Every result is a surviving id at its true distance, and the distances are those of
        // an index that only ever held the survivors.
        {
            IDSelectorRange sel(0, (idx_t)half);
            const idx_t before = dd.ntotal;
            const long gone = dd.remove_ids(sel);
            expect(dd.ntotal == before - gone && gone >= 0, "dedup remove_ids count");  // (an entry with a surviving copy stays)
            IndexIVFFlatDedup fresh(ix->quantizer, d, nlist, mt);
            fresh.coarse_mode = 0;
            fresh.nprobe = nprobe;
            std::vector<long> keep(nb - half);
            for (size_t i = half; i < nb; i++) keep[i - half] = (long)i;
            fresh.add_with_ids((idx_t)(nb - half), xb.as<float>() + half * d, keep.data());
            const size_t k = ks.as<int64_t>()[0];
            std::vector<float> D(nq * k), Df(nq * k);
            std::vector<idx_t> I(nq * k), If(nq * k);
            dd.search(nq, xq.as<float>(), k, D.data(), I.data());
            fresh.search(nq, xq.as<float>(), k, Df.data(), If.data());
            bool ok = same_f(D.data(), Df.data(), nq * k);
            for (size_t i = 0; i < nq * k && ok; i++) ok = I[i] < 0 ? If[i] < 0 : I[i] >= (idx_t)half && I[i] < (idx_t)nb;
            expect(ok, "dedup search after remove_ids == an index of the survivors");
        }
    }

    {   // IndexIVF::remove_ids (IndexIVF.cpp:955-987) + IDSelectorBatch: the lists in HBM follow the host lists
        IndexIVFFlat work(ix->quantizer, d, nlist, mt), fresh(ix->quantizer, d, nlist, mt);
        work.coarse_mode = fresh.coarse_mode = 0;
        work.nprobe = fresh.nprobe = nprobe;
        work.add(nb, xb.as<float>());
        const size_t k = ks.as<int64_t>()[0];
        std::vector<float> D(nq * k), Df(nq * k);
        std::vector<idx_t> I(nq * k), If(nq * k);
        work.search(nq, xq.as<float>(), k, D.data(), I.data());  // (the engine holds the full lists now)
        std::vector<idx_t> drop;
        std::vector<long> keep;
        for (size_t i = 0; i < nb; i++) (i % 3 == 1 ? drop : keep).push_back((idx_t)i);
        IDSelectorBatch sel((long)drop.size(), drop.data());
        const long gone = work.remove_ids(sel);
        expect(gone == (long)drop.size() && work.ntotal == (idx_t)keep.size(), "remove_ids count");
        std::vector<float> xk(keep.size() * d);
        for (size_t i = 0; i < keep.size(); i++) memcpy(&xk[i * d], xb.as<float>() + (size_t)keep[i] * d, d * sizeof(float));
        fresh.add_with_ids((idx_t)keep.size(), xk.data(), keep.data());
        work.search(nq, xq.as<float>(), k, D.data(), I.data());
        fresh.search(nq, xq.as<float>(), k, Df.data(), If.data());
        bool ok = same_f(D.data(), Df.data(), nq * k);
        for (size_t i = 0; i < nq * k && ok; i++) {
            // (equal distances -- also one just behind the k-th, which the row does not show -- may come out in another order)
            const bool tie = (i % k > 0 && D[i] == D[i - 1]) || (i % k + 1 < k && D[i] == D[i + 1]) || i % k + 1 == k;
            ok = I[i] < 0 ? If[i] < 0 : (I[i] % 3 != 1 && (tie || I[i] == If[i]));  // (removal moves entries inside a list: order among equals may differ)
        }
        expect(ok, "search after remove_ids == an index of the survivors");
    }

    size_t nshard = in.scalar_or<size_t>("nshard", 0);
    if (nshard) {
        std::vector<idx_t> a(nb), gid(nb);
        ix->quantizer->assign(nb, xb.as<float>(), a.data());
        for (size_t i = 0; i < nb; i++) gid[i] = i;
        std::vector<std::unique_ptr<IndexIVFFlat>> subs;
        IndexShards shards((idx_t)d, false, false);
        for (size_t s = 0; s < nshard; s++) {
            std::unique_ptr<IndexIVFFlat> sub(new IndexIVFFlat(ix->quantizer, d, nlist, mt));
            sub->coarse_mode = 0;
            std::vector<idx_t> pa(a);
            for (size_t i = 0; i < nb; i++) if ((size_t)pa[i] % nshard != s) pa[i] = -1;
            sub->add_core(nb, xb.as<float>(), gid.data(), pa.data());
            sub->nprobe = nprobe;
            shards.add_shard(sub.get());
            subs.push_back(std::move(sub));
        }
        // serial, then one host thread per shard (IndexShards(threaded = true): each thread drives its shard's engine,
        // on its own GPU where the node has several)
        for (int threaded = 0; threaded < 2; threaded++) {
            shards.threaded = threaded != 0;
            for (size_t ki = 0; ki < ks.numel(); ki++) {
                size_t k = ks.as<int64_t>()[ki];
                std::vector<float> D(nq * k);
                std::vector<idx_t> I(nq * k);
                shards.search(nq, xq.as<float>(), k, D.data(), I.data());
                const std::string tag = threaded ? " (threaded)" : "";
                expect(same_i(I.data(), in.get("I_shards_k" + std::to_string(k)).as<int64_t>(), nq * k), "shards ids" + tag);
                expect(same_f(D.data(), in.get("D_shards_k" + std::to_string(k)).as<float>(), nq * k), "shards distances" + tag);
            }
        }
        // the same shards made by the mirror itself: IndexIVF::copy_subset_to type 3 (lists with l % nshard == s: the goldens' owner
        // rule) under an IndexShards, and IndexShardsByList (type 4: byte-balanced owners) -- a C++ caller's way to the list-id shards
        {
            IndexIVFFlat* src = dynamic_cast<IndexIVFFlat*>(ix);
            std::vector<std::unique_ptr<IndexIVFFlat>> cut;
            IndexShards by_mod((idx_t)d, true, false);
            for (size_t s = 0; s < nshard; s++) {
                cut.emplace_back(new IndexIVFFlat(ix->quantizer, d, nlist, mt));
                cut.back()->coarse_mode = 0;
                cut.back()->nprobe = nprobe;
                src->copy_subset_to(*cut.back(), 3, (idx_t)nshard, (idx_t)s);
                by_mod.add_shard(cut.back().get());
            }
            expect(by_mod.ntotal == (idx_t)nb, "copy_subset_to type 3 keeps every vector once");
            IndexShardsByList by_bytes(*src, (int)nshard, true);
            std::vector<int> owner = ivf_list_owners(*src, (int)nshard);
            std::vector<size_t> load(nshard, 0);
            size_t longest = 0;
            for (size_t l = 0; l < nlist; l++) load[owner[l]] += ix->invlists->list_size(l), longest = std::max(longest, ix->invlists->list_size(l));
            expect(*std::max_element(load.begin(), load.end()) - *std::min_element(load.begin(), load.end()) <= longest, "list owners balanced by bytes");
            for (size_t ki = 0; ki < ks.numel(); ki++) {
                size_t k = ks.as<int64_t>()[ki];
                std::vector<float> D(nq * k);
                std::vector<idx_t> I(nq * k);
                by_mod.search(nq, xq.as<float>(), k, D.data(), I.data());
                expect(same_i(I.data(), in.get("I_shards_k" + std::to_string(k)).as<int64_t>(), nq * k) &&
                           same_f(D.data(), in.get("D_shards_k" + std::to_string(k)).as<float>(), nq * k), "copy_subset_to(3) shards == the reference's IndexShards");
                by_bytes.search(nq, xq.as<float>(), k, D.data(), I.data());
                // (another owner rule: equal distances may come out in another order; the distances are the single index's)
                expect(same_f(D.data(), in.get("D_k" + std::to_string(k)).as<float>(), nq * k), "IndexShardsByList distances == the single index's");
            }
            // the reference's own cuts: by id range, by id modulo, by position (IndexIVF.cpp:1055-1117)
            for (int type = 0; type < 3; type++) {
                IndexIVFFlat a(ix->quantizer, d, nlist, mt), b(ix->quantizer, d, nlist, mt);
                if (type == 0) { src->copy_subset_to(a, 0, 0, (idx_t)(nb / 2)); src->copy_subset_to(b, 0, (idx_t)(nb / 2), (idx_t)nb); }
                if (type == 1) { src->copy_subset_to(a, 1, 2, 0); src->copy_subset_to(b, 1, 2, 1); }
                if (type == 2) { src->copy_subset_to(a, 2, 0, (idx_t)(nb / 2)); src->copy_subset_to(b, 2, (idx_t)(nb / 2), (idx_t)nb); }
                bool ok = a.ntotal + b.ntotal == (idx_t)nb && (type != 2 || a.ntotal == (idx_t)(nb / 2));
                for (size_t l = 0; l < nlist && ok; l++) ok = a.invlists->list_size(l) + b.invlists->list_size(l) == ix->invlists->list_size(l);
                expect(ok, "copy_subset_to type " + std::to_string(type) + " splits the index in two");
            }
        }
        // IndexShards::add from one host thread per shard while the shards share one quantizer (IndexFlat::search is
        // re-entrant in the reference): the same lists as the serial add, shard by shard
        {
            std::vector<std::unique_ptr<IndexIVFFlat>> ser, par;
            IndexShards s_ser((idx_t)d, false, false), s_par((idx_t)d, true, false);
            for (size_t s = 0; s < nshard; s++) {
                ser.emplace_back(new IndexIVFFlat(ix->quantizer, d, nlist, mt));
                par.emplace_back(new IndexIVFFlat(ix->quantizer, d, nlist, mt));
                ser.back()->coarse_mode = par.back()->coarse_mode = 0;
                s_ser.add_shard(ser.back().get());
                s_par.add_shard(par.back().get());
            }
            s_ser.add((idx_t)nb, xb.as<float>());
            s_par.add((idx_t)nb, xb.as<float>());
            bool same = s_ser.ntotal == s_par.ntotal;
            for (size_t s = 0; s < nshard && same; s++)
                for (size_t l = 0; l < nlist && same; l++) {
                    const size_t n0 = ser[s]->invlists->list_size(l);
                    same = n0 == par[s]->invlists->list_size(l) &&
                           (n0 == 0 || (memcmp(ser[s]->invlists->get_ids(l), par[s]->invlists->get_ids(l), n0 * sizeof(idx_t)) == 0 &&
                                        memcmp(ser[s]->invlists->get_codes(l), par[s]->invlists->get_codes(l), n0 * d * sizeof(float)) == 0));
                }
            expect(same, "threaded IndexShards::add with a shared quantizer == serial add");
        }
    }
    return g_fail;
}

static int run_auncel(const tb::Bundle& in) {
    size_t d = in.scalar<size_t>("d"), nlist = in.scalar<size_t>("nlist");
    size_t K = in.scalar<size_t>("max_topk"), ts = in.scalar<size_t>("train_num"), ses = in.scalar<size_t>("test_num");
    const tb::Tensor &xb = in.get("xb"), &xq = in.get("xq"), &cen = in.get("centroids");
    size_t nb = xb.dims[0], nq = ts + ses;

    IndexFlatL2 quantizer(d);
    quantizer.add(nlist, cen.as<float>());
    quantizer.coarse_mode = 0;
    IndexIVFFlat index(&quantizer, d, nlist, METRIC_L2);
    index.coarse_mode = 0;  // goldens: exact coarse ranking (the reference's one-query-per-call path)
    // train_q1's table for these centroids (the k-means itself is out of scope: centroids come from the fixture)
    index.interdis_cem.assign(in.get("interdis_cem").as<float>(), in.get("interdis_cem").as<float>() + nlist * (nlist - 1) / 2);
    index.add(nb, xb.as<float>());

    // ---- eval/bound.cpp:356-396
    Error_sys err_sys(&index, nq, K);
    err_sys.set_gt(in.get("gtD").as<float>(), reinterpret_cast<const idx_t*>(in.get("gtI").as<int64_t>()));
    err_sys.sys_train(ts, xq.as<float>());
    size_t ntr = index.t->traces.size();
    for (size_t i = 0; i < ntr; i++) {
        const tb::Tensor& e = in.get("exp_sb_trace" + std::to_string(i));
        const Trace& tr = index.t->traces[i];
        bool ok = tr.trace.size() == e.dims[0] && memcmp(&tr.trace[0].first, e.as<float>(), e.dims[0] * 8) == 0 &&
                  memcmp(tr.stds.data(), in.get("exp_sb_stds" + std::to_string(i)).as<float>(), e.dims[0] * 4) == 0;
        expect(ok, "sys_train trace " + std::to_string(i));
    }
    // continue with the reference's own trained traces so that the online goldens apply
    for (size_t i = 0; i < ntr; i++) {
        const tb::Tensor& g = in.get("sb_trace" + std::to_string(i));
        Trace& tr = index.t->traces[i];
        tr.trace.resize(g.dims[0]);
        memcpy(&tr.trace[0].first, g.as<float>(), g.dims[0] * 8);
        tr.stds.assign(in.get("sb_stds" + std::to_string(i)).as<float>(), in.get("sb_stds" + std::to_string(i)).as<float>() + g.dims[0]);
    }
    index.t->traces_version++;

    const tb::Tensor &topks = in.get("topks"), &accs = in.get("require_acc"), &mults = in.get("multipler"), &stdms = in.get("std_m");
    for (size_t r = 0; r < topks.numel(); r++) {
        for (int batched = 0; batched < 2; batched++) {
            std::vector<float> acc(nq, accs.as<float>()[r]);
            err_sys.set_topk(topks.as<int64_t>()[r]);
            err_sys.set_queries(ses, xq.as<float>(), acc.data(), ts + ses);
            index.t->multipler = mults.as<float>()[r];
            index.t->std_m = stdms.as<float>()[r];
            index.t->profile = false;
            std::vector<float> D(ses * K);
            std::vector<int64_t> I(ses * K);
            if (!batched)
                for (size_t i = ts; i < ts + ses; i++) err_sys.search(D.data() + K * (i - ts), I.data() + K * (i - ts), i, 1);
            else
                err_sys.search(D.data(), I.data(), ts, ses);
            std::string suf = "_r" + std::to_string(r);
            std::string tag = suf + (batched ? " (one batch)" : " (one query per call)");
            expect(memcmp(I.data(), in.get("I" + suf).as<int64_t>(), ses * K * 8) == 0, "adaptive ids" + tag);
            expect(same_f(D.data(), in.get("D" + suf).as<float>(), ses * K), "adaptive distances" + tag);
            expect(memcmp(index.t->my_nprobe + ts, in.get("my_nprobe" + suf).as<uint64_t>(), ses * 8) == 0, "my_nprobe" + tag);
        }
    }
    // ---- eval/overhead.cpp:284-290: ix->t->overhead_profile = true, then Error_sys::search over the batch
    if (in.has("I_overhead")) {
        std::vector<float> acc(nq, accs.as<float>()[0]);
        err_sys.set_topk(topks.as<int64_t>()[0]);
        err_sys.set_queries(ses, xq.as<float>(), acc.data(), ts + ses);
        index.t->multipler = mults.as<float>()[0];
        index.t->std_m = stdms.as<float>()[0];
        index.t->profile = false;
        index.t->overhead_profile = true;
        std::vector<float> D(ses * K);
        std::vector<int64_t> I(ses * K);
        for (size_t i = ts; i < ts + ses; i++) err_sys.search(D.data() + K * (i - ts), I.data() + K * (i - ts), i, 1);
        index.t->overhead_profile = false;
        expect(memcmp(I.data(), in.get("I_overhead").as<int64_t>(), ses * K * 8) == 0, "overhead_profile ids");
        expect(same_f(D.data(), in.get("D_overhead").as<float>(), ses * K), "overhead_profile distances");
        bool zero = true;
        for (size_t i = ts; i < ts + ses; i++) zero = zero && index.t->my_nprobe[i] == 0;
        expect(zero, "overhead_profile leaves my_nprobe alone");
    }
    // ---- eval/effect_time.cpp:270-297: time-bounded search, budgets (ms) in the accuracy array.  With budgets nobody can
    //      use up the probe loop runs to its end: the plain search with nprobe = nlist; with no budget at all a query still
    //      gets its first probes and returns something sorted
    {
        const size_t n_t = std::min<size_t>(ses, 16);
        std::vector<float> budget(nq, 1e9f);
        err_sys.set_queries(ses, xq.as<float>(), budget.data(), ts + ses);
        std::vector<float> D(n_t * K), Df(n_t * K);
        std::vector<int64_t> I(n_t * K);
        std::vector<idx_t> If(n_t * K);
        for (size_t i = 0; i < n_t; i++) err_sys.time_search(D.data() + K * i, I.data() + K * i, ts + i, 1);
        expect(index.t->time_tune, "time_search leaves time_tune on (profile.cpp:242)");
        index.t->time_tune = false;
        index.nprobe = nlist;
        index.search(n_t, xq.as<float>() + ts * d, K, Df.data(), If.data());
        expect(memcmp(I.data(), If.data(), n_t * K * 8) == 0 && same_f(D.data(), Df.data(), n_t * K), "time_search with unlimited budgets = full probe loop");
        std::fill(budget.begin(), budget.end(), 0.f);
        err_sys.time_search(D.data(), I.data(), ts, (long)n_t);
        index.t->time_tune = false;
        bool sorted = true;
        for (size_t i = 0; i < n_t; i++)
            for (size_t j = 1; j < K; j++) sorted &= D[i * K + j - 1] <= D[i * K + j];
        expect(sorted && I[0] >= 0, "time_search without budget returns the first probes' results");
    }
    // tune mode without a tuner is an error, as in the reference (IndexIVF.cpp:514-515)
    {
        IndexIVFFlat bare(&quantizer, d, nlist, METRIC_L2);
        bare.set_tune_mode();
        bare.nprobe = nlist;
        bool threw = false;
        std::vector<float> D(K);
        std::vector<idx_t> I(K);
        try { bare.search(1, xq.as<float>(), K, D.data(), I.data(), 0); } catch (const FaissException&) { threw = true; }
        expect(threw, "tune without init_tune throws FaissException");
    }
    return g_fail;
}

// faiss::Clustering of the mirror (GPU assignment, reference update procedure) against the compiled reference's
// centroids / objective, and IndexIVF::train through index_factory
static int run_kmeans(const tb::Bundle& in) {
    size_t d = in.scalar<size_t>("d"), k = in.scalar<size_t>("k");
    const tb::Tensor& x = in.get("x");
    const size_t n = x.dims[0];
    MetricType mt = in.scalar<int>("metric") == 0 ? METRIC_INNER_PRODUCT : METRIC_L2;
    ClusteringParameters cp;
    cp.niter = in.scalar<int>("niter");
    cp.seed = in.scalar<int>("seed");
    cp.spherical = in.scalar<int>("spherical") != 0;
    cp.max_points_per_centroid = in.scalar<int>("max_points_per_centroid");
    Clustering clus((int)d, (int)k, cp);
    IndexFlat index(d, mt);
    index.coarse_mode = 0;  // exact assignment kernel (the reference's BLAS rounding is unpinned)
    clus.train(n, x.as<float>(), index);
    const float* gc = in.get("centroids").as<float>();
    expect(clus.centroids.size() == k * d && same_f(clus.centroids.data(), gc, k * d), "Clustering::train centroids");
    const tb::Tensor& go = in.get("obj");
    bool obj_ok = clus.obj.size() == go.numel();
    for (size_t i = 0; obj_ok && i < clus.obj.size(); i++) {
        const float a = clus.obj[i], b = go.as<float>()[i];
        obj_ok = n < 20 ? memcmp(&a, &b, 4) == 0 : std::fabs(a - b) <= 1e-5f * std::fabs(b);
    }
    expect(obj_ok, "Clustering::train objective");
    expect((size_t)index.ntotal == k, "index holds the final centroids");

    if (mt == METRIC_L2 && !cp.spherical) {  // the same through IndexIVF::train (Level1Quantizer::train_q1)
        std::unique_ptr<Index> ivf(index_factory((int)d, ("IVF" + std::to_string(k) + ",Flat").c_str(), mt));
        IndexIVF* ix = dynamic_cast<IndexIVF*>(ivf.get());
        ix->cp = cp;
        dynamic_cast<IndexFlat*>(ix->quantizer)->coarse_mode = 0;
        ivf->train(n, x.as<float>());
        expect(ivf->is_trained && (size_t)ix->quantizer->ntotal == k, "IndexIVF::train trains the quantizer");
        std::vector<float> rec(k * d);
        IndexFlat* q = dynamic_cast<IndexFlat*>(ix->quantizer);
        expect(q->xb.size() == k * d && same_f(q->xb.data(), gc, k * d), "IndexIVF::train centroids");
    }
    return g_fail;
}

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    try {
        tb::Bundle in = tb::Bundle::load(argv[2]);
        const std::string cmd = argv[1];
        int f = cmd == "fixed" ? run_fixed(in) : cmd == "kmeans" ? run_kmeans(in) : run_auncel(in);
        printf(f ? "FAILED %d checks\n" : "ALL OK\n", f);
        return f ? 1 : 0;
    } catch (const std::exception& e) {
        printf("EXCEPTION: %s\n", e.what());
        return 3;
    }
}
