// AmdIndexIVFFlat -- the file a maintainer adds to the reference tree (Auncel/gpu_amd/AmdIndexIVFFlat.h) to put the
// MI355X engine behind the reference's own classes: an IndexIVFFlat whose search_preassigned (the reference's override
// point, Auncel/IndexIVF.h:189-195) calls the C ABI of include/auncel_amd.h.  Nothing of the reference is restated here:
// lists stay in its ArrayInvertedLists, traces in its error_pro, and this class mirrors them onto the device when they
// change.  tests/test_integration_subclass.py compiles this header against the reference's unmodified headers.
#ifndef AMD_INDEX_IVF_FLAT_H
#define AMD_INDEX_IVF_FLAT_H

#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../AuxIndexStructures.h"
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
    // Engine policy as fields, the way the reference exposes nprobe / max_codes / parallel_mode (IndexIVF.h:97-143); they reach
    // the engine through amd_ivf_set_option before the next search.  -1 = the engine's default.
    int coarse_tie_order = -1;           // inside runs of bit-equal coarse distances: 0 centroid number, 1 the reference's heap, 2 redo
    int selection = -1;                  // 0 the reference's heap replayed for every query, 1 sorted arrays + tie replay
    /// any other option of include/auncel_amd.h by name ("filter", "round_first", "scan_pipelined", ...)
    void set_engine_option(const char* key, double value) {
        std::lock_guard<std::mutex> lock(mu);
        extra_options[key] = value;
        options_stale = true;
    }
    size_t max_contexts = 8;             // search contexts kept for concurrent callers (each owns a stream and work buffers)

    AmdIndexIVFFlat(Index* quantizer, size_t d, size_t nlist, MetricType metric = METRIC_L2, int device_ = 0)
        : IndexIVFFlat(quantizer, d, nlist, metric), device(device_) {}
    ~AmdIndexIVFFlat() override {
        // the search contexts are clones of h and must go first (include/auncel_amd.h: "h must outlive its clones")
        for (amd_ivf_t* c : all_contexts) amd_ivf_destroy(c);
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
        // a context for the duration of the call (returned to the pool by the lease); a trace-training pass runs on the index's
        // own handle and has it to itself
        Lease lease(this, tune, training && !tune);
        amd_ivf_t* ctx = lease.ctx;
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

    // The reference's second override point (IndexIVF.h:228-229): a scanner with all five virtuals of IndexIVF.h:316-358.  The
    // reference's own scans the codes at whatever pointer it is handed (IndexIVFFlat.cpp:117-155); the codes live in HBM here, so the
    // pointer names a run of the current list -- any run: tests/test_lowlevel_ivf.cpp:426-564 deals lists (and the mirror's driver
    // halves of lists) to threads.  Every call leases a search context for its duration, as search_preassigned does.
    struct Scanner : InvertedListScanner {
        const AmdIndexIVFFlat* ix;
        bool store_pairs;
        std::vector<float> q;
        idx_t list_no;
        Scanner(const AmdIndexIVFFlat* ix_, bool sp) : ix(ix_), store_pairs(sp), list_no(-1) {}
        void set_query(const float* query) override { q.assign(query, query + ix->d); }
        void set_list(idx_t l, float) override { list_no = l; }
        size_t offset_of(size_t n, const uint8_t* codes) const {
            FAISS_THROW_IF_NOT_MSG(list_no >= 0, "set_list first");
            const uint8_t* base = ix->invlists->get_codes(list_no);
            const size_t sz = ix->invlists->list_size(list_no);
            FAISS_THROW_IF_NOT_MSG(n == 0 || (codes >= base && codes < base + sz * ix->code_size && (size_t)(codes - base) % ix->code_size == 0),
                                   "codes must point at a code of the current list");
            const size_t offset = n ? (size_t)(codes - base) / ix->code_size : 0;
            FAISS_THROW_IF_NOT_MSG(offset + n <= sz, "codes run past the end of the current list");
            return offset;
        }
        float distance_to_code(const uint8_t* code) const override {
            const size_t offset = offset_of(1, code);
            Lease lease(ix, false, false);
            float dis = 0;
            check(amd_ivf_distance_to_code(lease.ctx, q.data(), (size_t)list_no, offset, &dis));
            return dis;
        }
        size_t scan_codes(size_t n, const uint8_t* codes, const idx_t* ids, float* simi, idx_t* idxi, size_t k) const override {
            const size_t offset = offset_of(n, codes);
            if (n == 0) return 0;
            size_t nup = 0;
            Lease lease(ix, false, false);
            if (store_pairs || ids == ix->invlists->get_ids(list_no) + offset) {
                check(amd_ivf_scan_codes_at(lease.ctx, q.data(), (size_t)list_no, offset, n, store_pairs ? 1 : 0, k, simi,
                                            reinterpret_cast<int64_t*>(idxi), &nup));
                return nup;
            }
            // a caller-owned id array: the heap's entries travel as tags, the admitted ones come back as positions (ids[j])
            std::vector<int64_t> lab(k);
            for (size_t i = 0; i < k; i++) lab[i] = -(int64_t)i - 2;
            check(amd_ivf_scan_codes_at(lease.ctx, q.data(), (size_t)list_no, offset, n, 1, k, simi, lab.data(), &nup));
            std::vector<idx_t> old(idxi, idxi + k);
            for (size_t i = 0; i < k; i++) idxi[i] = lab[i] < 0 ? old[(size_t)(-lab[i] - 2)] : ids[lab[i] & 0xffffffffll];
            return nup;
        }
        void scan_codes_range(size_t n, const uint8_t* codes, const idx_t* ids, float radius, RangeQueryResult& res) const override {
            const size_t offset = offset_of(n, codes);
            if (n == 0) return;
            size_t count = 0;
            Lease lease(ix, false, false);
            check(amd_ivf_scan_codes_range(lease.ctx, q.data(), (size_t)list_no, offset, n, radius, &count));
            if (!count) return;
            std::vector<uint32_t> pos(count);
            std::vector<float> dis(count);
            check(amd_ivf_scan_codes_range_results(lease.ctx, pos.data(), dis.data()));
            for (size_t i = 0; i < count; i++) res.add(dis[i], store_pairs ? (idx_t)((idx_t)list_no << 32 | (idx_t)pos[i]) : ids[pos[i]]);
        }
    };
    InvertedListScanner* get_InvertedListScanner(bool store_pairs = false) const override { return new Scanner(this, store_pairs); }

   private:
    // Device state.  search_preassigned is const and re-entrant like the reference's.  Everything below is guarded by `mu`.
    // A call checks a search context out of a bounded pool (clones of h: own stream and work buffers, same lists) and gives it
    // back when it ends -- callers that spawn a thread per search (the reference's IndexShards does) no longer leave a context
    // per thread id behind.  Searches hold the index "shared"; re-uploading lists / traces (after an add, a retraining) and a
    // trace-training pass on h itself wait for the searches in flight to drain and hold it "exclusive" (ADVICE round 3).
    mutable std::mutex mu;
    mutable std::condition_variable cv;
    mutable int searching = 0;           // calls holding the index shared
    mutable bool exclusive = false;
    mutable amd_ivf_t* h = nullptr;
    mutable std::vector<amd_ivf_t*> all_contexts, free_contexts;  // owned: destroyed before h
    mutable std::map<std::string, double> extra_options;
    mutable bool options_stale = true;
    mutable int applied_ties = -2, applied_select = -2;
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

    struct Lease {
        const AmdIndexIVFFlat* ix;
        amd_ivf_t* ctx = nullptr;
        bool excl;
        Lease(const AmdIndexIVFFlat* ix_, bool need_traces, bool exclusive_) : ix(ix_), excl(exclusive_) {
            std::unique_lock<std::mutex> lock(ix->mu);
            for (;;) {
                ix->cv.wait(lock, [&] { return !ix->exclusive; });
                if (!excl && !ix->needs_upload(need_traces)) break;
                // uploads and training passes wait for the searches in flight, then have the index to themselves
                ix->exclusive = true;
                ix->cv.wait(lock, [&] { return ix->searching == 0; });
                try {
                    ix->upload(need_traces);
                } catch (...) {
                    ix->exclusive = false;
                    ix->cv.notify_all();
                    throw;
                }
                if (excl) {  // (stays exclusive until the lease ends; the training pass runs on h)
                    ctx = ix->h;
                    return;
                }
                ix->exclusive = false;
                ix->cv.notify_all();
            }
            ix->cv.wait(lock, [&] { return !ix->free_contexts.empty() || ix->all_contexts.size() < ix->max_contexts; });
            if (ix->free_contexts.empty()) {
                amd_ivf_t* c = nullptr;
                check(amd_ivf_clone(ix->h, &c));
                ix->all_contexts.push_back(c);
                ix->free_contexts.push_back(c);
            }
            ctx = ix->free_contexts.back();
            ix->free_contexts.pop_back();
            ix->searching++;
        }
        ~Lease() {
            std::lock_guard<std::mutex> lock(ix->mu);
            if (excl) {
                if (ctx) ix->exclusive = false;
            } else if (ctx) {
                ix->free_contexts.push_back(ctx);
                ix->searching--;
            }
            ix->cv.notify_all();
        }
        Lease(const Lease&) = delete;
        Lease& operator=(const Lease&) = delete;
    };

    // (mu held) does the device copy lag behind the index?
    bool needs_upload(bool need_traces) const {
        if (!h || lists_stale || options_stale || applied_ties != coarse_tie_order || applied_select != selection) return true;
        if (need_traces) return !traces_known || digest_of_traces() != traces_digest;
        return false;
    }

    // (mu held, no search in flight) centroids, lists, centroid table, options and (tune mode) traces -> device, once per change
    void upload(bool need_traces) const {
        if (!h) check(amd_ivf_create((int)d, nlist, metric_type == METRIC_L2 ? 1 : 0, device, &h));
        if (options_stale || applied_ties != coarse_tie_order || applied_select != selection) {
            const double unset = std::nan("");
            check(amd_ivf_set_option(h, "coarse_ties", coarse_tie_order < 0 ? unset : (double)coarse_tie_order));
            check(amd_ivf_set_option(h, "select", selection < 0 ? unset : (double)selection));
            for (const auto& kv : extra_options) check(amd_ivf_set_option(h, kv.first.c_str(), kv.second));
            applied_ties = coarse_tie_order;
            applied_select = selection;
            options_stale = false;
        }
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
    }
};

}  // namespace faiss
#endif
