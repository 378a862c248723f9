// faiss::IndexIVFFlat (Auncel/IndexIVFFlat.h:25-58)
#pragma once
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

}  // namespace faiss
