// TEST INFRASTRUCTURE.  INTEGRATION.md route B, executed: the reference's own classes (compiled from /root/reference into
// oracle/_ref/libfaiss_ref.a) drive integration/AmdIndexIVFFlat.h, the subclass a maintainer adds to the reference tree, and
// every result is compared bit for bit with the same calls on the reference's CPU IndexIVFFlat:
//   * IndexIVF::search with fixed nprobe (batched and one query per call, k = 10 / 100, nprobe 1 / 8 / 200, store_pairs),
//   * the reference's IndexShards(threaded) over two AmdIndexIVFFlat shards (two engine handles driven from its WorkerThreads),
//   * Error_sys::sys_train (the training branch, traces after Trace::SB),
//   * Error_sys::search, one query per call as eval/bound.cpp:380-386 does (tune branch: D, I, my_nprobe, t_recalls).
// Built by oracle/Makefile (target subclass) in the container that has the reference; the binary travels to the GPU box.
#include <memory>
#include "Auncel/gpu_amd/AmdIndexIVFFlat.h"

#include <cstdio>
#include <cstring>
#include <new>
#include <random>

#include "Auncel/IVF_pro.h"
#include "Auncel/IndexShards.h"
#include "Auncel/profile.h"

using namespace faiss;
typedef Index::idx_t idx_t;

static int bad = 0;
static void expect(bool ok, const char* what) {
    std::printf("%-72s %s\n", what, ok ? "equal" : "DIFFERENT");
    if (!ok) bad++;
}
template <class T> static bool same(const std::vector<T>& a, const std::vector<T>& b) {
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0;
}

int main(int argc, char** argv) {
    const bool bytes = argc > 1 && !std::strcmp(argv[1], "bytes");  // uint8-valued data (byte-code scan) or float data
    const size_t d = 32, nb = 60000, nlist = 1024, ts = 100, ses = 60, nq = ts + ses, K = 100;
    std::mt19937 rng(7);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<float> centres(256 * d), xb(nb * d), xq(nq * d);
    for (float& v : centres) v = bytes ? 40.f + 30.f * g(rng) + 80.f : g(rng);
    auto draw = [&](std::vector<float>& out, size_t n) {
        for (size_t i = 0; i < n; i++) {
            const size_t c = rng() % 256;
            for (size_t j = 0; j < d; j++) {
                float v = centres[c * d + j] + (bytes ? 25.f : 0.6f) * g(rng);
                if (bytes) v = std::floor(std::min(255.f, std::max(0.f, v)));
                out[i * d + j] = v;
            }
        }
    };
    draw(xb, nb);
    draw(xq, nq);

    // ---- the reference on the CPU
    IndexFlat q1(d, METRIC_L2);
    IndexIVFFlat ref(&q1, d, nlist, METRIC_L2);
    ref.cp.niter = 6;
    ref.set_tune_mode();  // bound.cpp:261-263: train also fills interdis_cem
    ref.train(nb, xb.data());
    ref.set_tune_off();
    ref.add(nb, xb.data());
    // ---- the same index behind the subclass
    IndexFlat q2(d, METRIC_L2);
    q2.add(nlist, q1.xb.data());
    AmdIndexIVFFlat amd(&q2, d, nlist, METRIC_L2, 0);
    amd.is_trained = true;
    amd.interdis_cem = ref.interdis_cem;
    amd.add(nb, xb.data());

    // ---- fixed nprobe through IndexIVF::search (the quantizer runs on the host in both; search_preassigned is the override)
    ref.init_tune(0, 1, nullptr, nullptr, nullptr, nullptr, nullptr);  // plain searches dereference `t` (IndexIVF.cpp:529)
    amd.init_tune(0, 1, nullptr, nullptr, nullptr, nullptr, nullptr);
    for (size_t k : {(size_t)10, (size_t)100})
        for (size_t nprobe : {(size_t)1, (size_t)8, (size_t)200}) {
            ref.nprobe = amd.nprobe = nprobe;
            std::vector<float> D1(nq * k), D2(nq * k);
            std::vector<idx_t> I1(nq * k), I2(nq * k);
            ref.search(nq, xq.data(), k, D1.data(), I1.data());
            amd.search(nq, xq.data(), k, D2.data(), I2.data());
            char what[128];
            std::snprintf(what, sizeof what, "search, %zu queries in one call, k %zu nprobe %zu", nq, k, nprobe);
            expect(same(D1, D2) && same(I1, I2), what);
            for (size_t i = 0; i < 20; i++) {
                ref.search(1, xq.data() + i * d, k, D1.data() + i * k, I1.data() + i * k);
                amd.search(1, xq.data() + i * d, k, D2.data() + i * k, I2.data() + i * k);
            }
            std::snprintf(what, sizeof what, "search, one query per call, k %zu nprobe %zu", k, nprobe);
            expect(same(D1, D2) && same(I1, I2), what);
        }
    {
        ref.nprobe = amd.nprobe = 8;
        std::vector<float> cd(nq * 8), D1(nq * 10), D2(nq * 10);
        std::vector<idx_t> ck(nq * 8), I1(nq * 10), I2(nq * 10);
        q1.search(nq, xq.data(), 8, cd.data(), ck.data());
        ref.search_preassigned(nq, xq.data(), 10, ck.data(), cd.data(), D1.data(), I1.data(), true);
        amd.search_preassigned(nq, xq.data(), 10, ck.data(), cd.data(), D2.data(), I2.data(), true);
        expect(same(D1, D2) && same(I1, I2), "search_preassigned, store_pairs");
    }

    // ---- get_InvertedListScanner, the second override point: the subclass's scanner next to the reference's own IVFFlatScanner on
    // the same lists -- whole lists, lists in two halves, single codes, the range scan (tests/test_lowlevel_ivf.cpp:82-220,426-564)
    {
        const size_t k = 10, np = 8;
        std::vector<float> cd(nq * np);
        std::vector<idx_t> ck(nq * np);
        q1.search(nq, xq.data(), np, cd.data(), ck.data());
        bool heaps = true, pairs = true, single = true, ranges = true;
        for (int sp = 0; sp < 2; sp++) {
            std::unique_ptr<InvertedListScanner> a(ref.get_InvertedListScanner(sp != 0)), b(amd.get_InvertedListScanner(sp != 0));
            for (size_t i = 0; i < 6; i++) {
                std::vector<float> s1(k), s2(k);
                std::vector<idx_t> l1(k), l2(k);
                maxheap_heapify(k, s1.data(), l1.data());
                maxheap_heapify(k, s2.data(), l2.data());
                a->set_query(xq.data() + i * d);
                b->set_query(xq.data() + i * d);
                RangeSearchResult r1(1), r2(1);
                RangeSearchPartialResult p1(&r1), p2(&r2);
                RangeQueryResult& qr1 = p1.new_result(0);
                RangeQueryResult& qr2 = p2.new_result(0);
                float radius = 0;
                for (size_t p = 0; p < np; p++) {
                    const idx_t key = ck[i * np + p];
                    const size_t sz = key < 0 ? 0 : ref.invlists->list_size(key);
                    if (!sz) continue;
                    a->set_list(key, cd[i * np + p]);
                    b->set_list(key, cd[i * np + p]);
                    const size_t half = sz / 2;
                    size_t u1 = 0, u2 = 0;
                    // (the reference's lists and the subclass's hold the same codes in the same order: same adds)
                    u1 += a->scan_codes(half, ref.invlists->get_codes(key), ref.invlists->get_ids(key), s1.data(), l1.data(), k);
                    u1 += a->scan_codes(sz - half, ref.invlists->get_codes(key) + half * ref.code_size, ref.invlists->get_ids(key) + half, s1.data(), l1.data(), k);
                    u2 += b->scan_codes(half, amd.invlists->get_codes(key), amd.invlists->get_ids(key), s2.data(), l2.data(), k);
                    u2 += b->scan_codes(sz - half, amd.invlists->get_codes(key) + half * amd.code_size, amd.invlists->get_ids(key) + half, s2.data(), l2.data(), k);
                    (sp ? pairs : heaps) &= u1 == u2 && same(s1, s2) && same(l1, l2);
                    single &= a->distance_to_code(ref.invlists->get_codes(key) + (sz - 1) * ref.code_size) ==
                              b->distance_to_code(amd.invlists->get_codes(key) + (sz - 1) * amd.code_size);
                    if (p == 0) radius = s1[0];  // (the worst of the best ten so far: keeps a handful per list)
                    a->scan_codes_range(half, ref.invlists->get_codes(key), ref.invlists->get_ids(key), radius, qr1);
                    a->scan_codes_range(sz - half, ref.invlists->get_codes(key) + half * ref.code_size, ref.invlists->get_ids(key) + half, radius, qr1);
                    b->scan_codes_range(half, amd.invlists->get_codes(key), amd.invlists->get_ids(key), radius, qr2);
                    b->scan_codes_range(sz - half, amd.invlists->get_codes(key) + half * amd.code_size, amd.invlists->get_ids(key) + half, radius, qr2);
                }
                p1.finalize();
                p2.finalize();
                ranges &= r1.lims[1] == r2.lims[1] && std::memcmp(r1.labels, r2.labels, r1.lims[1] * sizeof(idx_t)) == 0 &&
                          std::memcmp(r1.distances, r2.distances, r1.lims[1] * sizeof(float)) == 0;
            }
        }
        expect(heaps, "scanner: scan_codes over list halves, heaps and update counts as the reference's scanner");
        expect(pairs, "scanner: the same with store_pairs (labels count from the half's first code)");
        expect(single, "scanner: distance_to_code");
        expect(ranges, "scanner: scan_codes_range over list halves, results in the reference's order");
    }

    // ---- the reference's IndexShards (threaded: one WorkerThread per shard, IndexShards.cpp:261-311) over two shards that split the
    // data by id range, once with the reference's CPU shards, once with two AmdIndexIVFFlat (two engine handles driven concurrently)
    {
        const size_t half = nb / 2;
        IndexFlat qa(d, METRIC_L2), qb(d, METRIC_L2), qc(d, METRIC_L2), qd(d, METRIC_L2);
        for (IndexFlat* q : {&qa, &qb, &qc, &qd}) q->add(nlist, q1.xb.data());
        IndexIVFFlat ca(&qa, d, nlist, METRIC_L2), cb(&qb, d, nlist, METRIC_L2);
        AmdIndexIVFFlat ga(&qc, d, nlist, METRIC_L2, 0), gb(&qd, d, nlist, METRIC_L2, 0);
        for (IndexIVF* ix : std::initializer_list<IndexIVF*>{&ca, &cb, &ga, &gb}) {
            ix->is_trained = true;
            ix->init_tune(0, 1, nullptr, nullptr, nullptr, nullptr, nullptr);
            ix->nprobe = 16;
        }
        ca.add(half, xb.data());
        cb.add(nb - half, xb.data() + half * d);
        ga.add(half, xb.data());
        gb.add(nb - half, xb.data() + half * d);
        IndexShards cpu_shards((int)d, true, true), amd_shards((int)d, true, true);
        cpu_shards.add_shard(&ca);
        cpu_shards.add_shard(&cb);
        amd_shards.add_shard(&ga);
        amd_shards.add_shard(&gb);
        const size_t k = 10;
        std::vector<float> D1(nq * k), D2(nq * k);
        std::vector<idx_t> I1(nq * k), I2(nq * k);
        cpu_shards.search(nq, xq.data(), k, D1.data(), I1.data());
        amd_shards.search(nq, xq.data(), k, D2.data(), I2.data());
        expect(same(D1, D2) && same(I1, I2), "IndexShards(threaded) over two shards, successive ids, k 10 nprobe 16");
    }

    // ---- Auncel: exact ground truth, training, adaptive search
    std::vector<float> gtD(nq * K);
    std::vector<idx_t> gtI(nq * K);
    {
        IndexFlat flat(d, METRIC_L2);
        flat.add(nb, xb.data());
        for (size_t i = 0; i < nq; i++) flat.search(1, xq.data() + i * d, K, gtD.data() + i * K, gtI.data() + i * K);
    }
    Error_sys es1(&ref, nq, K), es2(&amd, nq, K);  // bound.cpp:356 (each makes its index a fresh error_pro)
    es1.set_gt(gtD.data(), gtI.data());
    es2.set_gt(gtD.data(), gtI.data());
    es1.sys_train(ts, xq.data());
    es2.sys_train(ts, xq.data());
    bool traces_same = ref.t->traces.size() == amd.t->traces.size();
    for (size_t i = 0; traces_same && i < ref.t->traces.size(); i++)
        traces_same = ref.t->traces[i].trace == amd.t->traces[i].trace && ref.t->traces[i].stds == amd.t->traces[i].stds;
    expect(traces_same, "Error_sys::sys_train: traces after Trace::SB");

    for (int prof = 0; prof < 2; prof++)
        for (size_t topk : {(size_t)10, (size_t)100}) {
            std::vector<float> req(nq, topk == 10 ? 0.9f : 0.95f);
            std::vector<float> D1(ses * K), D2(ses * K);
            std::vector<idx_t> I1(ses * K), I2(ses * K);
            Error_sys* es[2] = {&es1, &es2};
            IndexIVF* ix[2] = {&ref, &amd};
            for (int w = 0; w < 2; w++) {
                es[w]->set_topk(topk);
                es[w]->set_queries(ses, xq.data(), req.data(), ts + ses);
                ix[w]->t->multipler = 1.5f;
                ix[w]->t->std_m = 1.0f;
                ix[w]->t->profile = prof == 1;
                float* D = w ? D2.data() : D1.data();
                idx_t* I = w ? I2.data() : I1.data();
                for (size_t i = ts; i < ts + ses; i++) es[w]->search(D + (i - ts) * K, I + (i - ts) * K, i, 1);
            }
            char what[128];
            std::snprintf(what, sizeof what, "Error_sys::search per query, query_topk %zu%s: D, I", topk, prof ? ", profile" : "");
            expect(same(D1, D2) && same(I1, I2), what);
            bool np = true, tr = true;
            for (size_t i = ts; i < ts + ses; i++) {
                np = np && ref.t->my_nprobe[i] == amd.t->my_nprobe[i];
                tr = tr && std::memcmp(&ref.t->t_recalls[i], &amd.t->t_recalls[i], 4) == 0;
            }
            std::snprintf(what, sizeof what, "Error_sys::search per query, query_topk %zu%s: my_nprobe", topk, prof ? ", profile" : "");
            expect(np, what);
            if (prof) expect(tr, "                                               t_recalls");
        }
    // ---- the whole batch in one call (the reference ranks it through sgemm): with device_coarse off the subclass searches over
    // the keys / coarse_dis its caller passes, i.e. the reference's own ranking, and must return the reference's result
    {
        amd.device_coarse = false;
        std::vector<float> req(nq, 0.9f);
        std::vector<float> D1(ses * K), D2(ses * K);
        std::vector<idx_t> I1(ses * K), I2(ses * K);
        Error_sys* es[2] = {&es1, &es2};
        IndexIVF* ix[2] = {&ref, &amd};
        for (int w = 0; w < 2; w++) {
            es[w]->set_topk(10);
            es[w]->set_queries(ses, xq.data(), req.data(), ts + ses);
            ix[w]->t->multipler = 1.5f;
            ix[w]->t->std_m = 1.0f;
            ix[w]->t->profile = false;
            es[w]->search(w ? D2.data() : D1.data(), w ? I2.data() : I1.data(), ts, ses);
        }
        bool np = true;
        for (size_t i = ts; i < ts + ses; i++) np = np && ref.t->my_nprobe[i] == amd.t->my_nprobe[i];
        expect(same(D1, D2) && same(I1, I2) && np, "Error_sys::search, the batch in one call, caller's coarse ranking: D, I, my_nprobe");
        amd.device_coarse = true;
    }

    // ---- traces retrained in place (same storage, same bin counts): the device copy has to follow
    {
        for (IndexIVF* ix : std::initializer_list<IndexIVF*>{&ref, &amd})
            for (Trace& tr : ix->t->traces)
                for (auto& pt : tr.trace) pt.second *= 1.25f;
        std::vector<float> req(nq, 0.9f);
        std::vector<float> D1(ses * K), D2(ses * K);
        std::vector<idx_t> I1(ses * K), I2(ses * K);
        Error_sys* es[2] = {&es1, &es2};
        for (int w = 0; w < 2; w++) {
            es[w]->set_topk(10);
            es[w]->set_queries(ses, xq.data(), req.data(), ts + ses);
            for (size_t i = ts; i < ts + ses; i++) es[w]->search((w ? D2.data() : D1.data()) + (i - ts) * K, (w ? I2.data() : I1.data()) + (i - ts) * K, i, 1);
        }
        bool np = true;
        for (size_t i = ts; i < ts + ses; i++) np = np && ref.t->my_nprobe[i] == amd.t->my_nprobe[i];
        expect(same(D1, D2) && same(I1, I2) && np, "Error_sys::search after the traces changed in place");
    }

    // ---- add after a search: the device lists follow
    {
        ref.set_tune_off();
        amd.set_tune_off();
        std::vector<float> extra(2000 * d);
        draw(extra, 2000);
        ref.add(2000, extra.data());
        amd.add(2000, extra.data());
        ref.nprobe = amd.nprobe = 8;
        std::vector<float> D1(nq * 10), D2(nq * 10);
        std::vector<idx_t> I1(nq * 10), I2(nq * 10);
        ref.search(nq, xq.data(), 10, D1.data(), I1.data());
        amd.search(nq, xq.data(), 10, D2.data(), I2.data());
        expect(same(D1, D2) && same(I1, I2), "search after an add that followed a search");
    }

    // ---- an index destroyed and another one constructed at the same address (and searched from the same thread): the second
    // must not see anything of the first (search contexts are owned by the index, not keyed by its address)
    {
        alignas(AmdIndexIVFFlat) static unsigned char spot[sizeof(AmdIndexIVFFlat)];
        const size_t part[2] = {nb / 3, nb / 2};
        for (int round = 0; round < 2; round++) {
            IndexFlat qr(d, METRIC_L2), qg(d, METRIC_L2);
            qr.add(nlist, q1.xb.data());
            qg.add(nlist, q1.xb.data());
            IndexIVFFlat cpu(&qr, d, nlist, METRIC_L2);
            AmdIndexIVFFlat* gpu = new (spot) AmdIndexIVFFlat(&qg, d, nlist, METRIC_L2, 0);
            for (IndexIVF* ix : std::initializer_list<IndexIVF*>{&cpu, gpu}) {
                ix->is_trained = true;
                ix->init_tune(0, 1, nullptr, nullptr, nullptr, nullptr, nullptr);
                ix->nprobe = 8;
                ix->add(part[round], xb.data() + (round ? 1000 * d : 0));
            }
            std::vector<float> D1(nq * 10), D2(nq * 10);
            std::vector<idx_t> I1(nq * 10), I2(nq * 10);
            cpu.search(nq, xq.data(), 10, D1.data(), I1.data());
            gpu->search(nq, xq.data(), 10, D2.data(), I2.data());
            expect(same(D1, D2) && same(I1, I2), round ? "a second index constructed where a destroyed one was" : "an index in static storage");
            gpu->~AmdIndexIVFFlat();
        }
    }
    std::printf(bad ? "SUBCLASS PARITY FAILED (%d)\n" : "SUBCLASS PARITY OK\n", bad);
    return bad ? 1 : 0;
}
