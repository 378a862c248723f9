// AmdIndexIVFFlat -- the file a maintainer adds to the reference tree (Auncel/gpu_amd/AmdIndexIVFFlat.h) to put the
// MI355X engine behind the reference's own classes: an IndexIVFFlat whose search_preassigned (the reference's override
// point, Auncel/IndexIVF.h:189-195) calls the C ABI of include/auncel_amd.h.  Nothing of the reference is restated here:
// lists stay in its ArrayInvertedLists, traces in its error_pro, and this class mirrors them onto the device when they
// change.  tests/test_integration_subclass.py compiles this header against the reference's unmodified headers.
#ifndef AMD_INDEX_IVF_FLAT_H
#define AMD_INDEX_IVF_FLAT_H

#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../IndexFlat.h"
#include "../IndexIVFFlat.h"
#include "../FaissAssert.h"
#include <auncel_amd.h>

namespace faiss {

struct AmdIndexIVFFlat : IndexIVFFlat {
    int device = 0;                      // one index = one GPU; IndexShards gives every shard its own
    // tune / train mode: true = the engine ranks the centroids itself (the fast path; the reference's own arithmetic for calls
    // of fewer than 20 queries, exact distances in centroid order inside runs of equal ones otherwise); false = the keys and
    // coarse_dis the caller passes to search_preassigned are uploaded and used as they are (the reference's quantizer, its BLAS)
    bool device_coarse = true;

    AmdIndexIVFFlat(Index* quantizer, size_t d, size_t nlist, MetricType metric = METRIC_L2, int device_ = 0)
        : IndexIVFFlat(quantizer, d, nlist, metric), device(device_) {}
    ~AmdIndexIVFFlat() override {
        // the per-thread search contexts are clones of h and must go first (include/auncel_amd.h: "h must outlive its clones")
        for (auto& c : contexts) amd_ivf_destroy(c.second);
        if (h) amd_ivf_destroy(h);
    }
    AmdIndexIVFFlat(const AmdIndexIVFFlat&) = delete;
    AmdIndexIVFFlat& operator=(const AmdIndexIVFFlat&) = delete;

    static void check(int rc) {
        if (rc == -2) FAISS_THROW_MSG(amd_ivf_last_error());           // what the reference throws as FaissException
        if (rc != 0) throw std::runtime_error(amd_ivf_last_error());   // no device / HIP failure: there is no CPU path
    }

    void add_with_ids(idx_t n, const float* x, const long* xids) override {
        IndexIVFFlat::add_with_ids(n, x, xids);
        std::lock_guard<std::mutex> lock(mu);
        lists_stale = true;
    }
    void reset() override {
        IndexIVFFlat::reset();
        std::lock_guard<std::mutex> lock(mu);
        lists_stale = true;
    }

    void search_preassigned(idx_t n, const float* x, idx_t k, const idx_t* keys, const float* coarse_dis, float* D, idx_t* I,
                            bool store_pairs, const IVFSearchParameters* params = nullptr) const override {
        amd_ivf_t* ctx = prepare(tune);
        const size_t offset = ((size_t)k >> 32) & 0xffffffffu;  // Auncel packs the query offset into k (IndexIVF.cpp:371-373)
        k &= 0xffffffff;
        const size_t np = params ? params->nprobe : nprobe;
        const int64_t* keys64 = reinterpret_cast<const int64_t*>(keys);
        if (tune) {  // Error_sys::search: per-query error-bounded stop
            const int prof = (t->profile ? 1 : 0) | (t->overhead_profile ? 2 : 0);
            if (device_coarse)
                check(amd_ivf_search_adaptive_x(ctx, n, x, offset, t->query_topk, t->multipler, t->std_m, t->require_acc, t->train_D, prof,
                                                /*coarse as the reference*/ -1, reinterpret_cast<uint64_t*>(t->my_nprobe), t->t_recalls, D,
                                                reinterpret_cast<int64_t*>(I)));
            else
                check(amd_ivf_search_adaptive_pre(ctx, n, x, offset, np, keys64, coarse_dis, t->query_topk, t->multipler, t->std_m,
                                                  t->require_acc, t->train_D, prof, reinterpret_cast<uint64_t*>(t->my_nprobe), t->t_recalls,
                                                  D, reinterpret_cast<int64_t*>(I)));
        } else if (training) {  // Error_sys::sys_train: (sum_angle, kscaling) samples into the traces' raw storage
            std::vector<float*> raw;
            for (Trace& tr : t->traces) raw.push_back(&tr.trace[0].first);
            std::lock_guard<std::mutex> lock(mu);  // (runs on the index's own handle: one training pass at a time)
            ctx = h;
            if (device_coarse)
                check(amd_ivf_train_samples_x(h, n, x, offset, k, t->train_D, t->train_num, -1, raw.data(), D, reinterpret_cast<int64_t*>(I)));
            else
                check(amd_ivf_train_samples_pre(h, n, x, offset, np, keys64, coarse_dis, k, t->train_D, t->train_num, raw.data(), D,
                                                reinterpret_cast<int64_t*>(I)));
        } else if (t && t->time_tune) {  // Error_sys::time_search: budgets (ms) travel in require_acc
            check(amd_ivf_search_timed_x(ctx, n, x, offset, k, np, t->require_acc, -1, nullptr, D, reinterpret_cast<int64_t*>(I)));
        } else {
            check(amd_ivf_search_preassigned(ctx, n, x, k, np, keys64, coarse_dis, D, reinterpret_cast<int64_t*>(I), store_pairs ? 1 : 0,
                                             params ? params->max_codes : max_codes));
        }
        size_t st[4];
        amd_ivf_stats(ctx, st, 1);
        indexIVF_stats.nq += st[0];
        indexIVF_stats.nlist += st[1];
        indexIVF_stats.ndis += st[2];
        indexIVF_stats.nheap_updates += st[3];
    }

   private:
    // Device state.  search_preassigned is const and re-entrant like the reference's: everything below is guarded by `mu`,
    // and every calling thread searches on its own context (a clone of h: own stream and work buffers, same lists).
    mutable std::mutex mu;
    mutable amd_ivf_t* h = nullptr;
    mutable std::map<std::thread::id, amd_ivf_t*> contexts;  // owned: destroyed before h
    mutable bool lists_stale = true;     // set by every add: the device copy is rebuilt before the next search
    mutable bool traces_known = false;
    mutable uint64_t traces_digest = 0;  // of what the device holds

    // everything the device copy of the traces is made from, byte for byte: retraining into the same storage is seen
    uint64_t digest_of_traces() const {
        uint64_t hsh = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t bytes) {  // FNV-1a over 4-byte words (everything hashed here is made of them)
            const unsigned char* c = static_cast<const unsigned char*>(p);
            for (size_t i = 0; i + 4 <= bytes; i += 4) {
                uint32_t w;
                std::memcpy(&w, c + i, 4);
                hsh = (hsh ^ w) * 1099511628211ull;
            }
        };
        const size_t nt = t->traces.size();
        mix(&nt, sizeof(nt));
        mix(&t->max_topk, sizeof(t->max_topk));
        for (const Trace& tr : t->traces) {
            const size_t len = tr.trace.size();
            mix(&len, sizeof(len));
            if (len) mix(tr.trace.data(), len * sizeof(tr.trace[0]));
            if (!tr.stds.empty()) mix(tr.stds.data(), tr.stds.size() * sizeof(tr.stds[0]));
        }
        return hsh;
    }

    // centroids, lists, centroid table and (tune mode) traces -> device, once per change; returns the caller's context
    amd_ivf_t* prepare(bool need_traces) const {
        std::lock_guard<std::mutex> lock(mu);
        if (!h) check(amd_ivf_create((int)d, nlist, metric_type == METRIC_L2 ? 1 : 0, device, &h));
        if (lists_stale) {
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
            traces_known = false;
        }
        if (need_traces) {
            FAISS_THROW_IF_NOT_MSG(t, "tune mode without init_tune");
            const uint64_t dg = digest_of_traces();
            if (!traces_known || dg != traces_digest) {
                // error_pro's trained traces (after Trace::SB) and its acos table
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
                traces_known = true;
                traces_digest = dg;
            }
        }
        amd_ivf_t*& c = contexts[std::this_thread::get_id()];
        if (!c) check(amd_ivf_clone(h, &c));
        return c;
    }
};

}  // namespace faiss
#endif
