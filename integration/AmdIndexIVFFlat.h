// AmdIndexIVFFlat -- the file a maintainer adds to the reference tree (Auncel/gpu_amd/AmdIndexIVFFlat.h) to put the
// MI355X engine behind the reference's own classes: an IndexIVFFlat whose search_preassigned (the reference's override
// point, Auncel/IndexIVF.h:189-195) calls the C ABI of include/auncel_amd.h.  Nothing of the reference is restated here:
// lists stay in its ArrayInvertedLists, traces in its error_pro, and this class mirrors them onto the device when they
// change.  tests/test_integration_subclass.py compiles this header against the reference's unmodified headers.
#ifndef AMD_INDEX_IVF_FLAT_H
#define AMD_INDEX_IVF_FLAT_H

#include <cstdint>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "../IndexFlat.h"
#include "../IndexIVFFlat.h"
#include "../FaissAssert.h"
#include <auncel_amd.h>

namespace faiss {

struct AmdIndexIVFFlat : IndexIVFFlat {
    int device = 0;                      // one index = one GPU; IndexShards gives every shard its own
    mutable amd_ivf_t* h = nullptr;
    mutable bool lists_stale = true;     // set by every add: the device copy is rebuilt before the next search
    mutable const void* traces_seen = nullptr;
    mutable size_t traces_bins = 0;

    AmdIndexIVFFlat(Index* quantizer, size_t d, size_t nlist, MetricType metric = METRIC_L2, int device_ = 0)
        : IndexIVFFlat(quantizer, d, nlist, metric), device(device_) {}
    ~AmdIndexIVFFlat() override {
        if (h) amd_ivf_destroy(h);
    }

    static void check(int rc) {
        if (rc == -2) FAISS_THROW_MSG(amd_ivf_last_error());           // what the reference throws as FaissException
        if (rc != 0) throw std::runtime_error(amd_ivf_last_error());   // no device / HIP failure: there is no CPU path
    }

    void add_with_ids(idx_t n, const float* x, const long* xids) override {
        IndexIVFFlat::add_with_ids(n, x, xids);
        lists_stale = true;
    }
    void reset() override {
        IndexIVFFlat::reset();
        lists_stale = true;
    }

    // centroids, lists and the centroid table -> device (once per change)
    void sync() const {
        if (!h) check(amd_ivf_create((int)d, nlist, metric_type == METRIC_L2 ? 1 : 0, device, &h));
        if (!lists_stale) return;
        const IndexFlat* q = dynamic_cast<const IndexFlat*>(quantizer);
        FAISS_THROW_IF_NOT_MSG(q, "AmdIndexIVFFlat needs an IndexFlat quantizer");
        check(amd_ivf_set_centroids(h, q->xb.data()));
        const ArrayInvertedLists* al = dynamic_cast<const ArrayInvertedLists*>(invlists);
        FAISS_THROW_IF_NOT_MSG(al, "AmdIndexIVFFlat needs ArrayInvertedLists");
        std::vector<size_t> sz(nlist);
        std::vector<const float*> codes(nlist);
        std::vector<const int64_t*> ids(nlist);
        for (size_t l = 0; l < nlist; l++) {
            sz[l] = al->ids[l].size();
            codes[l] = reinterpret_cast<const float*>(al->codes[l].data());
            ids[l] = reinterpret_cast<const int64_t*>(al->ids[l].data());
        }
        check(amd_ivf_set_lists(h, sz.data(), codes.data(), ids.data()));
        if (!interdis_cem.empty()) check(amd_ivf_set_interdis(h, interdis_cem.data()));
        lists_stale = false;
        traces_seen = nullptr;
    }

    // error_pro's trained traces (after Trace::SB) and its acos table -> device, when they have changed
    void upload_traces_if_changed() const {
        FAISS_THROW_IF_NOT_MSG(t, "tune mode without init_tune");
        size_t bins = 0;
        for (const Trace& tr : t->traces) bins += tr.trace.size();
        if (traces_seen == t->traces.data() && traces_bins == bins) return;
        const size_t nt = t->traces.size();
        std::vector<std::vector<float>> xs(nt), ys(nt);
        std::vector<size_t> len(nt);
        std::vector<const float*> px(nt), py(nt), ps(nt);
        for (size_t i = 0; i < nt; i++) {
            const Trace& tr = t->traces[i];
            len[i] = tr.trace.size();
            xs[i].resize(len[i]);
            ys[i].resize(len[i]);
            for (size_t j = 0; j < len[i]; j++) {
                xs[i][j] = tr.trace[j].first;
                ys[i][j] = tr.trace[j].second;
            }
            px[i] = xs[i].data();
            py[i] = ys[i].data();
            ps[i] = tr.stds.data();
        }
        if (t->arcos_list.empty()) t->construct_arcos();
        check(amd_ivf_set_tuner(h, t->max_topk, nt, len.data(), px.data(), py.data(), ps.data(), t->arcos_list.data()));
        traces_seen = t->traces.data();
        traces_bins = bins;
    }

    // one search context per calling thread: search() stays re-entrant like the reference's
    amd_ivf_t* context() const {
        thread_local std::unordered_map<const AmdIndexIVFFlat*, amd_ivf_t*> mine;
        amd_ivf_t*& c = mine[this];
        if (!c) check(amd_ivf_clone(h, &c));
        return c;
    }

    void search_preassigned(idx_t n, const float* x, idx_t k, const idx_t* keys, const float* coarse_dis, float* D, idx_t* I,
                            bool store_pairs, const IVFSearchParameters* params = nullptr) const override {
        sync();
        const size_t offset = ((size_t)k >> 32) & 0xffffffffu;  // Auncel packs the query offset into k (IndexIVF.cpp:371-373)
        k &= 0xffffffff;
        if (tune) {  // Error_sys::search: per-query error-bounded stop
            upload_traces_if_changed();
            check(amd_ivf_search_adaptive_x(context(), n, x, offset, t->query_topk, t->multipler, t->std_m, t->require_acc,
                                            t->train_D, (t->profile ? 1 : 0) | (t->overhead_profile ? 2 : 0), /*coarse as the reference*/ -1,
                                            reinterpret_cast<uint64_t*>(t->my_nprobe), t->t_recalls, D, reinterpret_cast<int64_t*>(I)));
        } else if (training) {  // Error_sys::sys_train: (sum_angle, kscaling) samples into the traces' raw storage
            std::vector<float*> raw;
            for (Trace& tr : t->traces) raw.push_back(&tr.trace[0].first);
            check(amd_ivf_train_samples_x(h, n, x, offset, k, t->train_D, t->train_num, -1, raw.data(), D, reinterpret_cast<int64_t*>(I)));
        } else if (t && t->time_tune) {  // Error_sys::time_search: budgets (ms) travel in require_acc
            check(amd_ivf_search_timed_x(context(), n, x, offset, k, nprobe, t->require_acc, -1, nullptr, D, reinterpret_cast<int64_t*>(I)));
        } else {
            check(amd_ivf_search_preassigned(context(), n, x, k, params ? params->nprobe : nprobe, reinterpret_cast<const int64_t*>(keys),
                                             coarse_dis, D, reinterpret_cast<int64_t*>(I), store_pairs ? 1 : 0,
                                             params ? params->max_codes : max_codes));
        }
        size_t st[4];
        amd_ivf_stats(context(), st, 1);
        indexIVF_stats.nq += st[0];
        indexIVF_stats.nlist += st[1];
        indexIVF_stats.ndis += st[2];
        indexIVF_stats.nheap_updates += st[3];
    }
};

}  // namespace faiss
#endif
