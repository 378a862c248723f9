// faiss::RangeSearchResult as the Auncel tree declares it (Auncel/AuxIndexStructures.h:31-50).
#pragma once
#include <cstddef>

#include "Index.h"

namespace faiss {

struct RangeSearchResult {
    size_t nq;     ///< nb of queries
    size_t* lims;  ///< size (nq + 1)

    typedef Index::idx_t idx_t;

    idx_t* labels;     ///< result for query i is labels[lims[i]:lims[i+1]]
    float* distances;  ///< corresponding distances (not sorted)

    size_t buffer_size;  ///< size of the result buffers used

    /// lims must be allocated on input to range_search.
    explicit RangeSearchResult(idx_t nq, bool alloc_lims = true);

    /// called when lims contains the nb of elements result entries for each query
    virtual void do_allocation();

    virtual ~RangeSearchResult();
};

}  // namespace faiss
