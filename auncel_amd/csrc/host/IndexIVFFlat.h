// faiss::IndexIVFFlat (Auncel/IndexIVFFlat.h:25-58)
#pragma once
#include <unordered_map>

#include "IndexIVF.h"

namespace faiss {

struct IndexIVFFlat : IndexIVF {
    IndexIVFFlat(Index* quantizer, size_t d, size_t nlist_, MetricType = METRIC_L2);
    IndexIVFFlat() {}

    /// precomputed_idx: list numbers from a previous assignment, entries < 0 skipped (IndexIVFFlat.cpp:41-80)
    virtual void add_core(idx_t n, const float* x, const long* xids, const long* precomputed_idx);
    void add_with_ids(idx_t n, const float* x, const long* xids) override;
    void reconstruct_from_offset(idx_t list_no, idx_t offset, float* recons) const override;  ///< IndexIVFFlat.cpp:226-230
};

/// faiss::IndexIVFFlatDedup (Auncel/IndexIVFFlat.h:62-107): equal vectors are stored once; `instances` maps the id that
/// is stored to the ids of its copies, and search results are expanded from it on the host.  update_vectors / range_search /
/// reconstruct_from_offset are "not implemented" in the reference too.
struct IndexIVFFlatDedup : IndexIVFFlat {
    std::unordered_multimap<idx_t, idx_t> instances;

    IndexIVFFlatDedup(Index* quantizer, size_t d, size_t nlist_, MetricType = METRIC_L2);
    IndexIVFFlatDedup() {}

    void train(idx_t n, const float* x) override;  ///< also dedups the training set
    void add_with_ids(idx_t n, const float* x, const long* xids) override;
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const override;
    void search_preassigned(idx_t n, const float* x, idx_t k, const idx_t* assign, const float* centroid_dis, float* distances,
                            idx_t* labels, bool store_pairs, const IVFSearchParameters* params = nullptr) const override;
    void range_search(idx_t n, const float* x, float radius, RangeSearchResult* result) const override;
    void reconstruct_from_offset(idx_t list_no, idx_t offset, float* recons) const override;
    long remove_ids(const IDSelector& sel) override;  ///< IndexIVFFlat.cpp:381-448

   private:
    void expand_instances(idx_t n, idx_t k, float* distances, idx_t* labels) const;
};

}  // namespace faiss
