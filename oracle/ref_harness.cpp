// TEST INFRASTRUCTURE ONLY (oracle/): driver that links against the *compiled reference*
// (/root/reference/Auncel/*.cpp built into oracle/_ref/ by oracle/Makefile) and dumps what
// the reference computes for a given input bundle.  It is how tests/golden/*.npz are made
// (tests/golden/make_golden.py) and how the CPU restatement in oracle/ivf_oracle.cpp is
// pinned.  It is built only in the container that has /root/reference; the binary under
// oracle/_ref/ travels to the GPU box, where bench.py's cpu_baseline leg runs its `bench` mode
// (kind "reference", oracle/refbench.py).  Nothing in the product imports or links this.
//
// usage: ref_harness <fixed|auncel|io|kmeans|bench|fixedbench> <in.tb> <out.tb>
//
// Reference entry points exercised (file:line in /root/reference/Auncel):
//   IndexFlat::search             IndexFlat.cpp:42-56   (knn_L2sqr_sse / _blas, utils.cpp:454-655)
//   IndexIVFFlat::add_core        IndexIVFFlat.cpp:41-80
//   IndexIVF::search_preassigned  IndexIVF.cpp:382-736
//   IVFFlatScanner                IndexIVFFlat.cpp:94-158
//   IndexShards::search           IndexShards.cpp:261-311 (merge_tables :44-105)
//   Level1Quantizer::train_q1     IndexIVF.cpp:71-137   (interdis_cem :97-117)
//   IndexIVF::init_tune           IndexIVF.cpp:203-244
//   error_pro / Trace             IVF_pro.cpp
//   Error_sys                     profile.cpp:28-280
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <omp.h>
#include <cstdio>
#include <iostream>
#include <memory>

#include "Heap.h"
#include "AuxIndexStructures.h"
#include "Clustering.h"
#include "IVF_pro.h"
#include "IndexFlat.h"
#include "IndexIVF.h"
#include "IndexIVFFlat.h"
#include "IndexShards.h"
#include "InvertedLists.h"
#include "index_io.h"
#include "profile.h"
#include "tbundle.h"

using namespace faiss;
typedef Index::idx_t idx_t;

static std::vector<int64_t> to_i64(const std::vector<idx_t>& v) { return std::vector<int64_t>(v.begin(), v.end()); }

// coarse quantisation through the reference's exact (nx<20) path: one query per call
static void coarse_exact(const Index* q, size_t nq, const float* x, size_t d, size_t nprobe,
                         std::vector<float>& dis, std::vector<idx_t>& keys) {
    dis.resize(nq * nprobe);
    keys.resize(nq * nprobe);
    for (size_t i = 0; i < nq; i++) q->search(1, x + i * d, nprobe, dis.data() + i * nprobe, keys.data() + i * nprobe);
}

static int run_fixed(const tb::Bundle& in, tb::Bundle& out) {
    size_t d = in.scalar<size_t>("d"), nlist = in.scalar<size_t>("nlist");
    size_t nprobe = in.scalar<size_t>("nprobe");
    int metric = in.scalar<int>("metric");  // 0 = IP, 1 = L2 (reference MetricType)
    const tb::Tensor& cen = in.get("centroids");
    const tb::Tensor& xb = in.get("xb");
    const tb::Tensor& xq = in.get("xq");
    const tb::Tensor& ks = in.get("ks");
    size_t nb = xb.dims[0], nq = xq.dims[0];
    MetricType mt = metric == 0 ? METRIC_INNER_PRODUCT : METRIC_L2;

    IndexFlat quantizer(d, mt);
    quantizer.add(nlist, cen.as<float>());
    IndexIVFFlat index(&quantizer, d, nlist, mt);
    // reference needs a tuner object even for plain search (t dereferenced at IndexIVF.cpp:529)
    index.init_tune(0, 1, nullptr, nullptr, nullptr, nullptr, nullptr);
    index.add(nb, xb.as<float>());
    index.nprobe = nprobe;

    {  // list assignment of every database vector (pins add ordering)
        std::vector<idx_t> a(nb);
        quantizer.assign(nb, xb.as<float>(), a.data());
        out.put_i64("assign", {nb}, to_i64(a).data());
        std::vector<int64_t> sizes(nlist);
        for (size_t l = 0; l < nlist; l++) sizes[l] = index.invlists->list_size(l);
        out.put_i64("list_sizes", {nlist}, sizes.data());
    }

    std::vector<float> cd_sse, cd_blas(nq * nprobe);
    std::vector<idx_t> ck_sse, ck_blas(nq * nprobe);
    if (d % 4 == 0) {
        coarse_exact(&quantizer, nq, xq.as<float>(), d, nprobe, cd_sse, ck_sse);
    }
    // batched path (nx >= 20 -> BLAS); distances depend on the BLAS build: parity unpinned
    quantizer.search(nq, xq.as<float>(), nprobe, cd_blas.data(), ck_blas.data());
    if (d % 4 != 0) { cd_sse = cd_blas; ck_sse = ck_blas; }
    out.put_f32("coarse_dis_sse", {nq, nprobe}, cd_sse.data());
    out.put_i64("coarse_keys_sse", {nq, nprobe}, to_i64(ck_sse).data());
    out.put_f32("coarse_dis_blas", {nq, nprobe}, cd_blas.data());
    out.put_i64("coarse_keys_blas", {nq, nprobe}, to_i64(ck_blas).data());

    for (size_t ki = 0; ki < ks.numel(); ki++) {
        size_t k = ks.as<int64_t>()[ki];
        for (int sp = 0; sp < 2; sp++) {
            std::vector<float> D(nq * k);
            std::vector<idx_t> I(nq * k);
            indexIVF_stats.reset();
            index.search_preassigned(nq, xq.as<float>(), k, ck_sse.data(), cd_sse.data(), D.data(), I.data(), sp == 1);
            std::string suf = "_k" + std::to_string(k) + (sp ? "_pairs" : "");
            out.put_f32("D" + suf, {nq, k}, D.data());
            out.put_i64("I" + suf, {nq, k}, to_i64(I).data());
            int64_t st[3] = {(int64_t)indexIVF_stats.nlist, (int64_t)indexIVF_stats.ndis, (int64_t)indexIVF_stats.nheap_updates};
            out.put_i64("stats" + suf, {3}, st);
        }
        if (in.scalar_or<int>("max_codes", 0) > 0) {
            index.max_codes = in.scalar<int>("max_codes");
            std::vector<float> D(nq * k);
            std::vector<idx_t> I(nq * k);
            index.search_preassigned(nq, xq.as<float>(), k, ck_sse.data(), cd_sse.data(), D.data(), I.data(), false);
            out.put_f32("D_k" + std::to_string(k) + "_maxcodes", {nq, k}, D.data());
            out.put_i64("I_k" + std::to_string(k) + "_maxcodes", {nq, k}, to_i64(I).data());
            index.max_codes = 0;
        }
    }

    {  // scanner API (tests/test_lowlevel_ivf.cpp:82-220 pattern): raw heap, before reorder
        size_t k = ks.as<int64_t>()[0];
        size_t ns = std::min<size_t>(nq, 8);
        std::vector<float> H(ns * k), dtc(ns * nprobe, 0.f);
        std::vector<idx_t> HI(ns * k);
        std::vector<int64_t> nup(ns * nprobe, 0);
        std::unique_ptr<InvertedListScanner> sc(index.get_InvertedListScanner(false));
        for (size_t i = 0; i < ns; i++) {
            float* simi = H.data() + i * k;
            idx_t* idxi = HI.data() + i * k;
            if (mt == METRIC_L2) maxheap_heapify(k, simi, idxi); else minheap_heapify(k, simi, idxi);
            sc->set_query(xq.as<float>() + i * d);
            for (size_t p = 0; p < nprobe; p++) {
                idx_t key = ck_sse[i * nprobe + p];
                if (key < 0) continue;
                size_t ls = index.invlists->list_size(key);
                if (!ls) continue;
                sc->set_list(key, cd_sse[i * nprobe + p]);
                InvertedLists::ScopedCodes codes(index.invlists, key);
                InvertedLists::ScopedIds ids(index.invlists, key);
                dtc[i * nprobe + p] = sc->distance_to_code(codes.get());
                nup[i * nprobe + p] = sc->scan_codes(ls, codes.get(), ids.get(), simi, idxi, k);
            }
        }
        out.put_f32("scan_heap_D", {ns, k}, H.data());
        out.put_i64("scan_heap_I", {ns, k}, to_i64(HI).data());
        out.put_f32("scan_dist_to_code", {ns, nprobe}, dtc.data());
        out.put_i64("scan_nup", {ns, nprobe}, nup.data());
    }

    if (in.has("radius")) {  // IndexIVF::range_search_preassigned (IndexIVF.cpp:759-857) on the same coarse ranking
        const float radius = in.get("radius").as<float>()[0];
        RangeSearchResult res(nq);
        indexIVF_stats.reset();
        index.range_search_preassigned(nq, xq.as<float>(), radius, ck_sse.data(), cd_sse.data(), &res);
        std::vector<int64_t> lims(nq + 1);
        for (size_t i = 0; i <= nq; i++) lims[i] = (int64_t)res.lims[i];
        const size_t tot = res.lims[nq];
        out.put_i64("range_lims", {nq + 1}, lims.data());
        std::vector<idx_t> lab(res.labels, res.labels + tot);
        std::vector<int64_t> lab64 = to_i64(lab);
        int64_t dummy_l = 0;
        float dummy_d = 0.f;
        out.put_i64("range_labels", {tot}, tot ? lab64.data() : &dummy_l);
        out.put_f32("range_distances", {tot}, tot ? res.distances : &dummy_d);
        int64_t st[2] = {(int64_t)indexIVF_stats.nlist, (int64_t)indexIVF_stats.ndis};
        out.put_i64("range_stats", {2}, st);
    }

    if (in.scalar_or<int>("dedup", 0) > 0) {  // IndexIVFFlatDedup (IndexIVFFlat.cpp:233-380) over the same quantizer and data
        IndexIVFFlatDedup dd(&quantizer, d, nlist, mt);
        dd.init_tune(0, 1, nullptr, nullptr, nullptr, nullptr, nullptr);
        const size_t half = nb / 2;  // two calls: the second finds duplicates of the first in the lists
        dd.add(half, xb.as<float>());
        dd.add(nb - half, xb.as<float>() + half * d);
        dd.nprobe = nprobe;
        std::vector<int64_t> sizes(nlist);
        for (size_t l = 0; l < nlist; l++) sizes[l] = dd.invlists->list_size(l);
        out.put_i64("dedup_list_sizes", {nlist}, sizes.data());
        int64_t tot[2] = {(int64_t)dd.ntotal, (int64_t)dd.instances.size()};
        out.put_i64("dedup_ntotal_ninst", {2}, tot);
        for (size_t ki = 0; ki < ks.numel(); ki++) {
            size_t k = ks.as<int64_t>()[ki];
            std::vector<float> D(nq * k);
            std::vector<idx_t> I(nq * k);
            dd.search_preassigned(nq, xq.as<float>(), k, ck_sse.data(), cd_sse.data(), D.data(), I.data(), false);
            out.put_f32("D_dedup_k" + std::to_string(k), {nq, k}, D.data());
            out.put_i64("I_dedup_k" + std::to_string(k), {nq, k}, to_i64(I).data());
        }
    }

    size_t nshard = in.scalar_or<size_t>("nshard", 0);
    if (nshard > 0) {  // lists sharded by list id: owner(l) = l % nshard, global ids
        std::vector<idx_t> a(nb), gid(nb);
        quantizer.assign(nb, xb.as<float>(), a.data());
        for (size_t i = 0; i < nb; i++) gid[i] = i;
        std::vector<std::unique_ptr<IndexIVFFlat>> subs;
        IndexShards shards((idx_t)d, false, false);
        for (size_t s = 0; s < nshard; s++) {
            std::unique_ptr<IndexIVFFlat> sub(new IndexIVFFlat(&quantizer, d, nlist, mt));
            sub->init_tune(0, 1, nullptr, nullptr, nullptr, nullptr, nullptr);
            std::vector<idx_t> pa(a);
            for (size_t i = 0; i < nb; i++) if ((size_t)pa[i] % nshard != s) pa[i] = -1;
            sub->add_core(nb, xb.as<float>(), gid.data(), pa.data());
            sub->nprobe = nprobe;
            shards.add_shard(sub.get());
            subs.push_back(std::move(sub));
        }
        for (size_t ki = 0; ki < ks.numel(); ki++) {
            size_t k = ks.as<int64_t>()[ki];
            std::vector<float> D(nq * k);
            std::vector<idx_t> I(nq * k);
            // nq >= 20 would take the BLAS coarse path inside each shard; keep the exact one
            for (size_t i = 0; i < nq; i++)
                shards.search(1, xq.as<float>() + i * d, k, D.data() + i * k, I.data() + i * k);
            out.put_f32("D_shards_k" + std::to_string(k), {nq, k}, D.data());
            out.put_i64("I_shards_k" + std::to_string(k), {nq, k}, to_i64(I).data());
        }
    }
    return 0;
}

// Clustering::train over an IndexFlat, as Level1Quantizer::train_q1 runs it (IndexIVF.cpp:84-92)
static int run_kmeans(const tb::Bundle& in, tb::Bundle& out) {
    size_t d = in.scalar<size_t>("d"), k = in.scalar<size_t>("k");
    const tb::Tensor& x = in.get("x");
    size_t n = x.dims[0];
    MetricType mt = in.scalar<int>("metric") == 0 ? METRIC_INNER_PRODUCT : METRIC_L2;
    ClusteringParameters cp;
    cp.niter = in.scalar<int>("niter");
    cp.seed = in.scalar_or<int>("seed", 1234);
    cp.spherical = in.scalar_or<int>("spherical", 0) != 0;
    cp.max_points_per_centroid = in.scalar_or<int>("max_points_per_centroid", 256);
    Clustering clus(d, k, cp);
    IndexFlat index(d, mt);
    clus.train(n, x.as<float>(), index);
    out.put_f32("centroids", {k, d}, clus.centroids.data());
    out.put_f32("obj", {clus.obj.size()}, clus.obj.data());
    return 0;
}

static std::vector<uint8_t> slurp(const char* fn) {
    FILE* f = fopen(fn, "rb");
    std::vector<uint8_t> b;
    if (!f) return b;
    int c;
    while ((c = fgetc(f)) != EOF) b.push_back((uint8_t)c);
    fclose(f);
    return b;
}

// write_index of the reference on a small IVF-Flat index: before add ("sprs" list table) and after ("full")
static int run_io(const tb::Bundle& in, tb::Bundle& out) {
    size_t d = in.scalar<size_t>("d"), nlist = in.scalar<size_t>("nlist");
    MetricType mt = in.scalar<int>("metric") == 0 ? METRIC_INNER_PRODUCT : METRIC_L2;
    const tb::Tensor& cen = in.get("centroids");
    const tb::Tensor& xb = in.get("xb");
    size_t nb = xb.dims[0];
    IndexFlat quantizer(d, mt);
    quantizer.add(nlist, cen.as<float>());
    IndexIVFFlat index(&quantizer, d, nlist, mt);
    index.nprobe = in.scalar<size_t>("nprobe");
    write_index(&index, "empty.index");
    std::vector<uint8_t> b0 = slurp("empty.index");
    out.put("index_empty", tb::U8, {b0.size()}, b0.data());
    index.init_tune(0, 1, nullptr, nullptr, nullptr, nullptr, nullptr);
    index.add(nb, xb.as<float>());
    write_index(&index, "full.index");
    std::vector<uint8_t> b1 = slurp("full.index");
    out.put("index_full", tb::U8, {b1.size()}, b1.data());
    std::vector<idx_t> a(nb);
    quantizer.assign(nb, xb.as<float>(), a.data());
    out.put_i64("assign", {nb}, to_i64(a).data());
    return 0;
}

static void dump_traces(const error_pro* t, size_t ntr, const std::string& pre, tb::Bundle& out) {
    for (size_t i = 0; i < ntr; i++) {
        const Trace& tr = t->traces[i];
        std::vector<float> xy(tr.trace.size() * 2);
        for (size_t j = 0; j < tr.trace.size(); j++) { xy[2 * j] = tr.trace[j].first; xy[2 * j + 1] = tr.trace[j].second; }
        out.put_f32(pre + "trace" + std::to_string(i), {tr.trace.size(), 2}, xy.data());
        if (!tr.stds.empty()) out.put_f32(pre + "stds" + std::to_string(i), {tr.stds.size()}, tr.stds.data());
    }
}

static int run_auncel(const tb::Bundle& in, tb::Bundle& out) {
    size_t d = in.scalar<size_t>("d"), nlist = in.scalar<size_t>("nlist");
    int metric = in.scalar<int>("metric");
    size_t K = in.scalar<size_t>("max_topk"), ts = in.scalar<size_t>("train_num"), ses = in.scalar<size_t>("test_num");
    const tb::Tensor& xb = in.get("xb");
    const tb::Tensor& xq = in.get("xq");  // (ts + ses) x d : train queries first
    const tb::Tensor& topks = in.get("topks");
    const tb::Tensor& accs = in.get("require_acc");     // one scalar bound per run
    const tb::Tensor& mults = in.get("multipler");
    const tb::Tensor& stdms = in.get("std_m");
    size_t nb = xb.dims[0], nq = xq.dims[0];
    if (nq != ts + ses) throw std::runtime_error("xq rows != train_num + test_num");
    MetricType mt = metric == 0 ? METRIC_INNER_PRODUCT : METRIC_L2;

    IndexFlat quantizer(d, mt);
    IndexIVFFlat index(&quantizer, d, nlist, mt);
    index.cp.niter = in.scalar_or<int>("kmeans_niter", 10);
    if (in.has("centroids")) {
        // centroids supplied: fill the table exactly as train_q1 does (IndexIVF.cpp:97-100)
        quantizer.add(nlist, in.get("centroids").as<float>());
        index.is_trained = true;
        index.interdis_cem.resize(nlist * (nlist - 1) / 2);
        if (mt == METRIC_L2) fvec_inter_vecs(index.interdis_cem.data(), in.get("centroids").as<float>(), nlist, d);
        else throw std::runtime_error("supplied centroids only for L2");
    } else {
        index.set_tune_mode();  // bound.cpp:261-263
        index.train(nb, xb.as<float>());
        index.set_tune_off();
    }
    out.put_f32("centroids", {nlist, d}, quantizer.xb.data());
    out.put_f32("interdis_cem", {index.interdis_cem.size()}, index.interdis_cem.data());
    index.add(nb, xb.as<float>());
    {
        std::vector<idx_t> a(nb);
        quantizer.assign(nb, xb.as<float>(), a.data());
        out.put_i64("assign", {nb}, to_i64(a).data());
    }

    // exact ground truth, one query per call (exact kernel), K wide
    std::vector<float> gtD(nq * K);
    std::vector<idx_t> gtI(nq * K);
    {
        IndexFlat flat(d, mt);
        flat.add(nb, xb.as<float>());
        for (size_t i = 0; i < nq; i++) flat.search(1, xq.as<float>() + i * d, K, gtD.data() + i * K, gtI.data() + i * K);
    }
    out.put_f32("gtD", {nq, K}, gtD.data());
    out.put_i64("gtI", {nq, K}, to_i64(gtI).data());

    // full coarse ranking per query (what search() feeds search_preassigned in tune mode)
    {
        std::vector<float> cd;
        std::vector<idx_t> ck;
        coarse_exact(&quantizer, nq, xq.as<float>(), d, nlist, cd, ck);
        out.put_f32("coarse_dis_sse", {nq, nlist}, cd.data());
        out.put_i64("coarse_keys_sse", {nq, nlist}, to_i64(ck).data());
        std::vector<float> cdb(nq * nlist);
        std::vector<idx_t> ckb(nq * nlist);
        // the 10 training batches of sys_train (profile.cpp:109-136) go through the BLAS path
        size_t bs = ts / 10;
        for (size_t q0 = 0; q0 < ts; q0 += bs)
            quantizer.search(bs, xq.as<float>() + q0 * d, nlist, cdb.data() + q0 * nlist, ckb.data() + q0 * nlist);
        out.put_f32("coarse_dis_blas_train", {ts, nlist}, cdb.data());
        out.put_i64("coarse_keys_blas_train", {ts, nlist}, to_i64(ckb).data());
    }

    // ---- offline: manual replication of sys_train (profile.cpp:88-156) to get the raw samples
    Error_sys es(&index, nq, K);  // bound.cpp:356
    es.set_gt(gtD.data(), gtI.data());
    size_t ntr = 0;
    {
        index.init_tune(ts, K, xq.as<float>(), es.train_D.data(), es.train_I.data(), nullptr, nullptr);
        ntr = index.t->traces.size();
        index.set_train_mode();
        index.nprobe = nlist;
        std::vector<float> D(ts * K);
        std::vector<idx_t> I(ts * K);
        size_t bs = ts / 10;
        for (size_t q0 = 0; q0 < ts; q0 += bs)
            index.search(bs, xq.as<float>() + q0 * d, K, D.data() + q0 * K, I.data() + q0 * K, q0);
        index.set_train_off();
        out.put_f32("train_D", {ts, K}, D.data());
        out.put_i64("train_I", {ts, K}, to_i64(I).data());
        dump_traces(index.t, ntr, "raw_", out);
        index.t->train(METRIC_L2);
        dump_traces(index.t, ntr, "sb_", out);
        es.is_trained = true;
        out.put_f32("arcos_list", {index.t->arcos_list.size()}, index.t->arcos_list.data());
    }
    // cross-check against the real sys_train on a second index (same data => same traces)
    if (in.scalar_or<int>("check_sys_train", 1)) {
        IndexFlat q2(d, mt);
        q2.add(nlist, quantizer.xb.data());
        IndexIVFFlat ix2(&q2, d, nlist, mt);
        ix2.interdis_cem = index.interdis_cem;
        ix2.add(nb, xb.as<float>());
        Error_sys es2(&ix2, nq, K);
        es2.set_gt(gtD.data(), gtI.data());
        es2.sys_train(ts, xq.as<float>());
        bool same = ix2.t->traces.size() == ntr;
        for (size_t i = 0; same && i < ntr; i++)
            same = ix2.t->traces[i].trace == index.t->traces[i].trace && ix2.t->traces[i].stds == index.t->traces[i].stds;
        out.put_scalar_i64("sys_train_matches_manual", same ? 1 : 0);
        if (!same) fprintf(stderr, "WARNING: sys_train != manual replication\n");
    }

    // set_online outputs for every test query (IVF_pro.cpp:196-238)
    {
        size_t max_num = nlist / 8 + 20;
        std::vector<float> dtb(ses * max_num), c2c(ses * max_num);
        std::vector<float> req(nq, 0.9f);
        index.t->require_acc = req.data();
        const float* cd = out.get("coarse_dis_sse").as<float>();
        const int64_t* ck = out.get("coarse_keys_sse").as<int64_t>();
        for (size_t i = 0; i < ses; i++) {
            std::vector<float> a, b;
            std::vector<idx_t> keys(ck + (ts + i) * nlist, ck + (ts + i + 1) * nlist);
            index.t->set_online(ts + i, K, cd + (ts + i) * nlist, keys.data(), index.interdis_cem.data(), a, b);
            memcpy(dtb.data() + i * max_num, a.data(), max_num * 4);
            memcpy(c2c.data() + i * max_num, b.data(), max_num * 4);
        }
        out.put_f32("disToBoundary", {ses, max_num}, dtb.data());
        out.put_f32("cenTocen", {ses, max_num}, c2c.data());
    }

    // ---- online runs: bound.cpp:371-396
    for (size_t r = 0; r < topks.numel(); r++) {
        size_t topk = topks.as<int64_t>()[r];
        float acc = accs.as<float>()[r];
        std::vector<float> req(nq, acc);
        for (int prof = 0; prof < 2; prof++) {
            for (int batched = 0; batched < 2; batched++) {
                es.set_topk(topk);
                es.set_queries(ses, xq.as<float>(), req.data(), ts + ses);
                index.t->multipler = mults.as<float>()[r];
                index.t->std_m = stdms.as<float>()[r];
                index.t->profile = prof == 1;
                std::vector<float> D(ses * K);
                std::vector<int64_t> I(ses * K);
                indexIVF_stats.reset();
                if (!batched) {
                    for (size_t i = ts; i < ts + ses; i++) es.search(D.data() + (i - ts) * K, I.data() + (i - ts) * K, i, 1);
                } else {
                    es.search(D.data(), I.data(), ts, ses);
                }
                std::string suf = "_r" + std::to_string(r) + (prof ? "_prof" : "") + (batched ? "_batched" : "");
                out.put_f32("D" + suf, {ses, K}, D.data());
                out.put_i64("I" + suf, {ses, K}, I.data());
                std::vector<uint64_t> np(index.t->my_nprobe + ts, index.t->my_nprobe + ts + ses);
                out.put_u64("my_nprobe" + suf, {ses}, np.data());
                out.put_f32("t_recalls" + suf, {ses}, index.t->t_recalls + ts);
                int64_t st[3] = {(int64_t)indexIVF_stats.nlist, (int64_t)indexIVF_stats.ndis, (int64_t)indexIVF_stats.nheap_updates};
                out.put_i64("stats" + suf, {3}, st);
            }
        }
    }
    // ---- eval/overhead.cpp:284-290: the same search with ix->t->overhead_profile on (rule on every probe, no stop before
    // stage nlist / 8).  Results and counters are deterministic; the two times it prints are not recorded.
    {
        std::vector<float> req(nq, accs.as<float>()[0]);
        es.set_topk(topks.as<int64_t>()[0]);
        es.set_queries(ses, xq.as<float>(), req.data(), ts + ses);
        index.t->multipler = mults.as<float>()[0];
        index.t->std_m = stdms.as<float>()[0];
        index.t->profile = false;
        index.t->overhead_profile = true;
        std::vector<float> D(ses * K);
        std::vector<int64_t> I(ses * K);
        indexIVF_stats.reset();
        // one query per call: the exact coarse path (n < 20), the ranking the restatement and the engine are fed
        for (size_t i = ts; i < ts + ses; i++) es.search(D.data() + (i - ts) * K, I.data() + (i - ts) * K, i, 1);
        index.t->overhead_profile = false;
        out.put_f32("D_overhead", {ses, K}, D.data());
        out.put_i64("I_overhead", {ses, K}, I.data());
        std::vector<uint64_t> np(index.t->my_nprobe + ts, index.t->my_nprobe + ts + ses);
        out.put_u64("my_nprobe_overhead", {ses}, np.data());
        int64_t st[3] = {(int64_t)indexIVF_stats.nlist, (int64_t)indexIVF_stats.ndis, (int64_t)indexIVF_stats.nheap_updates};
        out.put_i64("stats_overhead", {3}, st);
    }
    return 0;
}

// bench.py's cpu_baseline, kind "reference": the compiled reference itself, timed on the host.  The index is assembled from
// the centroids and the inverted lists the engine built (same list order: ArrayInvertedLists::add_entries per list), the
// traces are the trained ones, and the search is bound.cpp's loop -- one Error_sys::search(D, I, i, 1) per query
// (eval/bound.cpp:380-386) -- first on one thread, as the reference runs (its IndexIVF.cpp has no OpenMP), then the same
// per-query calls spread over the host cores (every query writes only its own slots of my_nprobe / t_recalls).
static int run_bench(const tb::Bundle& in, tb::Bundle& out) {
    const size_t d = in.scalar<size_t>("d"), nlist = in.scalar<size_t>("nlist"), K = in.scalar<size_t>("max_topk");
    const size_t topk = in.scalar<size_t>("topk"), id0 = in.scalar<size_t>("id0");
    const size_t S1 = in.scalar<size_t>("single_thread_queries");
    const tb::Tensor &cen = in.get("centroids"), &off = in.get("list_off"), &codes = in.get("codes"), &ids = in.get("ids");
    const tb::Tensor& xq = in.get("xq");  // S x d: queries id0 .. id0 + S
    const size_t S = xq.dims[0], nall = ((id0 + S + 9) / 10) * 10;
    const float acc = in.get("require_acc").as<float>()[0];

    IndexFlat quantizer(d, METRIC_L2);
    quantizer.add(nlist, cen.as<float>());
    IndexIVFFlat index(&quantizer, d, nlist, METRIC_L2);
    index.is_trained = true;
    index.interdis_cem.resize(nlist * (nlist - 1) / 2);
    fvec_inter_vecs(index.interdis_cem.data(), cen.as<float>(), nlist, d);  // train_q1, IndexIVF.cpp:97-100
    const int64_t* lo = off.as<int64_t>();
    for (size_t l = 0; l < nlist; l++) {
        const size_t n = (size_t)(lo[l + 1] - lo[l]);
        if (n) index.invlists->add_entries(l, n, (const idx_t*)(ids.as<int64_t>() + lo[l]), (const uint8_t*)(codes.as<float>() + (size_t)lo[l] * d));
    }
    index.ntotal = lo[nlist];

    Error_sys es(&index, nall, K);  // resets index.t
    std::vector<float> zeroD(nall * K, 0.f);  // train_D is read for logging only (IndexIVF.cpp:509)
    index.init_tune(nall, K, nullptr, zeroD.data(), nullptr, nullptr, nullptr);
    const size_t ntr = index.t->traces.size();
    for (size_t i = 0; i < ntr; i++) {
        const tb::Tensor &tx = in.get("trace_x" + std::to_string(i)), &ty = in.get("trace_y" + std::to_string(i)), &tsd = in.get("trace_std" + std::to_string(i));
        Trace& tr = index.t->traces[i];
        tr.trace.resize(tx.numel());
        for (size_t j = 0; j < tx.numel(); j++) tr.trace[j] = std::make_pair(tx.as<float>()[j], ty.as<float>()[j]);
        tr.stds.assign(tsd.as<float>(), tsd.as<float>() + tsd.numel());
    }
    es.is_trained = true;
    std::vector<float> xall(nall * d, 0.f), req(nall, acc);
    memcpy(xall.data() + id0 * d, xq.as<float>(), S * d * sizeof(float));
    std::vector<float> D(S * K);
    std::vector<int64_t> I(S * K);
    auto arm = [&]() {
        es.set_topk(topk);
        es.set_queries(S, xall.data(), req.data(), nall);
        index.t->multipler = (float)in.scalar<double>("multipler");
        index.t->std_m = (float)in.scalar<double>("std_m");
        index.t->profile = false;
    };
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

    arm();
    const int max_threads = omp_get_max_threads();
    omp_set_num_threads(1);  // the quantizer's own `omp parallel for` (utils.cpp:454-490) would otherwise raise a full team per query
    double t0 = now();
    for (size_t i = 0; i < std::min(S1, S); i++) es.search(D.data() + i * K, I.data() + i * K, id0 + i, 1);
    const double t_single = now() - t0;
    omp_set_num_threads(max_threads);

    arm();
    index.set_tune_mode();
    index.nprobe = nlist;
    int nthreads = 1;
    t0 = now();
#pragma omp parallel
    {
#pragma omp single
        nthreads = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
        for (long i = 0; i < (long)S; i++)
            index.search(1, xall.data() + (id0 + i) * d, K, D.data() + i * K, (idx_t*)(I.data() + i * K), id0 + i);
    }
    const double t_all = now() - t0;
    index.set_tune_off();

    out.put_f32("D", {S, K}, D.data());
    out.put_i64("I", {S, K}, I.data());
    std::vector<uint64_t> np(index.t->my_nprobe + id0, index.t->my_nprobe + id0 + S);
    out.put_u64("my_nprobe", {S}, np.data());
    out.put_scalar_f64("seconds_one_thread", t_single);
    out.put_scalar_i64("queries_one_thread", (int64_t)std::min(S1, S));
    out.put_scalar_f64("seconds_all_threads", t_all);
    out.put_scalar_i64("threads", nthreads);
    return 0;
}

// scripts/bench_configs.py: the compiled reference's plain IndexIVF::search on an index assembled from the engine's lists,
// one query per call (the exact coarse path, utils.cpp:417-490) spread over the host threads; either metric.
static int run_fixedbench(const tb::Bundle& in, tb::Bundle& out) {
    const size_t d = in.scalar<size_t>("d"), nlist = in.scalar<size_t>("nlist"), k = in.scalar<size_t>("k"), nprobe = in.scalar<size_t>("nprobe");
    const MetricType mt = in.scalar<int>("metric") == 0 ? METRIC_INNER_PRODUCT : METRIC_L2;
    const tb::Tensor &cen = in.get("centroids"), &off = in.get("list_off"), &codes = in.get("codes"), &ids = in.get("ids"), &xq = in.get("xq");
    const size_t S = xq.dims[0];
    IndexFlat quantizer(d, mt);
    quantizer.add(nlist, cen.as<float>());
    IndexIVFFlat index(&quantizer, d, nlist, mt);
    index.is_trained = true;
    index.init_tune(0, 1, nullptr, nullptr, nullptr, nullptr, nullptr);  // `t` is dereferenced by plain searches too (IndexIVF.cpp:529)
    const int64_t* lo = off.as<int64_t>();
    for (size_t l = 0; l < nlist; l++) {
        const size_t n = (size_t)(lo[l + 1] - lo[l]);
        if (n) index.invlists->add_entries(l, n, (const idx_t*)(ids.as<int64_t>() + lo[l]), (const uint8_t*)(codes.as<float>() + (size_t)lo[l] * d));
    }
    index.ntotal = lo[nlist];
    index.nprobe = nprobe;
    std::vector<float> D(S * k);
    std::vector<int64_t> I(S * k);
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    int nthreads = 1;
    const double t0 = now();
#pragma omp parallel
    {
#pragma omp single
        nthreads = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
        for (long i = 0; i < (long)S; i++) index.search(1, xq.as<float>() + i * d, k, D.data() + i * k, (idx_t*)(I.data() + i * k));
    }
    const double t_all = now() - t0;
    out.put_f32("D", {S, k}, D.data());
    out.put_i64("I", {S, k}, I.data());
    out.put_scalar_f64("seconds_all_threads", t_all);
    out.put_scalar_i64("threads", nthreads);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s <fixed|auncel|io|kmeans|bench|fixedbench> <in.tb> <out.tb>\n", argv[0]);
        return 2;
    }
    try {
        tb::Bundle in = tb::Bundle::load(argv[2]);
        tb::Bundle out;
        std::string cmd = argv[1];
        // note: sys_train writes Validation_*.log into the CWD: run from a scratch dir
        int rc = cmd == "fixed" ? run_fixed(in, out) : cmd == "auncel" ? run_auncel(in, out) : cmd == "io" ? run_io(in, out) : cmd == "kmeans" ? run_kmeans(in, out) : cmd == "bench" ? run_bench(in, out) : cmd == "fixedbench" ? run_fixedbench(in, out) : 2;
        if (rc == 0) out.save(argv[3]);
        return rc;
    } catch (const std::exception& e) {
        fprintf(stderr, "ref_harness: %s\n", e.what());
        return 1;
    }
}
