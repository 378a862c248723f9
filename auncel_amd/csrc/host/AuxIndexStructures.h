// faiss::RangeSearchResult as the Auncel tree declares it (Auncel/AuxIndexStructures.h:31-50).
#pragma once
#include <cstddef>
#include <unordered_set>
#include <vector>

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

/// which ids an operation applies to (Auncel/AuxIndexStructures.h:54-90): remove_ids takes one
struct IDSelector {
    typedef Index::idx_t idx_t;
    virtual bool is_member(idx_t id) const = 0;
    virtual ~IDSelector() {}
};
/// ids in [imin, imax)
struct IDSelectorRange : IDSelector {
    idx_t imin, imax;
    IDSelectorRange(idx_t imin, idx_t imax) : imin(imin), imax(imax) {}
    bool is_member(idx_t id) const override { return id >= imin && id < imax; }
};
/// ids of a set
struct IDSelectorBatch : IDSelector {
    std::unordered_set<idx_t> set;
    IDSelectorBatch(long n, const idx_t* indices) : set(indices, indices + n) {}
    bool is_member(idx_t id) const override { return set.count(id) != 0; }
};

/// Blocks of (id, distance) results that grow without moving (Auncel/AuxIndexStructures.h:104-132): what the scanners' range
/// results are collected in before they are copied into a RangeSearchResult.
struct BufferList {
    typedef Index::idx_t idx_t;
    size_t buffer_size;  ///< entries per block
    struct Buffer {
        idx_t* ids;
        float* dis;
    };
    std::vector<Buffer> buffers;
    size_t wp;  ///< write position in the last block

    explicit BufferList(size_t buffer_size);
    ~BufferList();
    void append_buffer();
    void add(idx_t id, float dis);
    /// entries [ofs, ofs + n) of the blocks seen as one array
    void copy_range(size_t ofs, size_t n, idx_t* dest_ids, float* dest_dis);
};

struct RangeSearchPartialResult;

/// the results of one query, reported one by one by InvertedListScanner::scan_codes_range (AuxIndexStructures.h:138-147)
struct RangeQueryResult {
    using idx_t = Index::idx_t;
    idx_t qno;
    size_t nres;
    RangeSearchPartialResult* pres;
    void add(float dis, idx_t id);
};

/// one thread's share of a range search (AuxIndexStructures.h:149-178)
struct RangeSearchPartialResult : BufferList {
    RangeSearchResult* res;
    explicit RangeSearchPartialResult(RangeSearchResult* res_in);
    std::vector<RangeQueryResult> queries;
    RangeQueryResult& new_result(idx_t qno);
    void finalize();                            ///< set_lims + do_allocation + copy_result (one thread)
    void set_lims();                            ///< counts into res->lims
    void copy_result(bool incremental = false); ///< entries to res->labels / distances at res->lims
    static void merge(std::vector<RangeSearchPartialResult*>& partial_results, bool do_delete = true);
};

}  // namespace faiss
