// faiss::IndexIVF with Auncel's additions (Auncel/IndexIVF.h:37-374), backed by the MI355X engine.
#pragma once
#include <memory>
#include <vector>

#include "IVF_pro.h"
#include "Index.h"
#include "Clustering.h"
#include "InvertedLists.h"

struct amd_ivf;

namespace faiss {

struct Level1Quantizer {
    Index* quantizer;
    size_t nlist;
    std::vector<float> interdis_cem;  ///< packed centroid-to-centroid table (IVF_pro.cpp:21-39 layout)
    char quantizer_trains_alone;
    bool own_fields;
    ClusteringParameters cp;
    Index* clustering_index;

    void train_q1(size_t n, const float* x, bool verbose, MetricType metric_type);
    Level1Quantizer(Index* quantizer, size_t nlist, bool t = false);
    Level1Quantizer();
    ~Level1Quantizer();
};

struct IVFSearchParameters {
    size_t nprobe;
    size_t max_codes;
    virtual ~IVFSearchParameters() {}
};

struct InvertedListScanner;
struct RangeQueryResult;

struct IndexIVF : Index, Level1Quantizer {
    InvertedLists* invlists;
    bool own_invlists;
    bool training = false;
    error_pro* t;
    size_t code_size;
    size_t nprobe;
    size_t max_codes;
    int parallel_mode;
    bool maintain_direct_map;
    std::vector<idx_t> direct_map;
    /// coarse quantisation inside search(): -1 = the reference's switch (exact kernel below 20 queries, GEMM
    /// formulation on the matrix cores from 20 on, utils.cpp:644-655), 0 = always exact, 1 = always GEMM
    int coarse_mode = -1;
    /// Engine policy as fields (the reference exposes nprobe / max_codes / parallel_mode the same way, IndexIVF.h:97-143); they
    /// reach the engine through amd_ivf_set_option before the next search.  -1 = the engine's default.
    int coarse_tie_order = -1;  ///< inside runs of bit-equal coarse distances: 0 centroid number, 1 the reference's heap, 2 redo
    int selection = -1;         ///< 0 the reference's heap replayed for every query, 1 sorted arrays + tie replay
    /// any other option of include/auncel_amd.h by name ("filter", "round_first", "scan_pipelined", ...)
    void set_engine_option(const char* key, double value);

    IndexIVF(Index* quantizer, size_t d, size_t nlist, size_t code_size, MetricType metric = METRIC_L2);
    IndexIVF();
    ~IndexIVF() override;

    void set_tune_mode() override;
    /// the engine of this index and of its flat quantizer move together (before their first use)
    void set_device(int device) override;
    bool device_bound() const override;
    void set_tune_off() override;
    void set_train_mode();
    void set_train_off();
    void init_tune(size_t train_num, size_t topk, const float* train_q, const float* train_D, const long* train_I,
                   float* train_cd, long* train_ci);

    void reset() override;
    void train(idx_t n, const float* x) override;
    void add(idx_t n, const float* x) override;
    void add_with_ids(idx_t n, const float* x, const long* xids) override = 0;

    virtual void search_preassigned(idx_t n, const float* x, idx_t k, const idx_t* assign, const float* centroid_dis,
                                    float* distances, idx_t* labels, bool store_pairs,
                                    const IVFSearchParameters* params = nullptr) const;
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const override;
    /// Auncel overload: `offset` = absolute id of query 0 (IndexIVF.cpp:355-378)
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, size_t offset) const;

    /// IndexIVF.cpp:305-328,869-945
    void make_direct_map(bool new_maintain_direct_map = true);
    void reconstruct(idx_t key, float* recons) const override;
    void reconstruct_n(idx_t i0, idx_t ni, float* recons) const override;
    void search_and_reconstruct(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, float* recons) const override;
    virtual void reconstruct_from_offset(idx_t list_no, idx_t offset, float* recons) const;

    /// IndexIVF.cpp:740-857
    void range_search(idx_t n, const float* x, float radius, RangeSearchResult* result) const override;
    void range_search_preassigned(idx_t nx, const float* x, float radius, const idx_t* keys, const float* coarse_dis,
                                  RangeSearchResult* result) const;

    virtual InvertedListScanner* get_InvertedListScanner(bool store_pairs = false) const;

    /// IndexIVF.cpp:955-987
    long remove_ids(const IDSelector& sel) override;
    /// A subset of the entries copied into `other` (same nlist and code size).  subset_type 0: ids in [a1, a2); 1: ids with
    /// id % a1 == a2; 2: the a1-th to a2-th entry of ntotal, cut list by list (IndexIVF.cpp:1055-1117, what the reference's
    /// GpuAutoTune shards an IVF index with: by VECTOR).  Two more cut by LIST, whole lists with their ids -- the shards north_star
    /// names, every probed list on exactly one GPU: 3: lists with list_no % a1 == a2; 4: lists whose owner among a1 shards is a2,
    /// owners balanced by list bytes (ivf_list_owners below).
    void copy_subset_to(IndexIVF& other, int subset_type, idx_t a1, idx_t a2) const;

    size_t get_list_size(size_t list_no) const { return invlists->list_size(list_no); }
    void replace_invlists(InvertedLists* il, bool own = false);

    /// the engine handle (created lazily; contents refreshed when centroids / lists / traces changed)
    amd_ivf* engine() const;
    /// register a query matrix as resident in HBM (what Error_sys::set_queries / sys_train pass in)
    void set_resident_queries(const float* x, size_t n) const;

   private:
    friend struct EngineScanner;
    mutable amd_ivf* gpu_ = nullptr;
    mutable size_t lists_version_ = (size_t)-1;
    mutable size_t centroid_count_ = (size_t)-1;
    mutable size_t centroid_version_ = (size_t)-1;
    mutable float interdis_print_[3] = {0, 0, 0};
    mutable size_t traces_version_ = (size_t)-1;
    mutable const float* interdis_uploaded_ = nullptr;
    mutable size_t interdis_size_ = 0;
    mutable const float* resident_ptr_ = nullptr;
    mutable size_t resident_n_ = 0;
    mutable int applied_ties_ = -2, applied_select_ = -2;
    void sync_engine(bool need_tuner) const;
    void fold_stats() const;
};

/// scans one list for one query on a caller-owned raw heap (IndexIVF.h:316-358)
struct InvertedListScanner {
    using idx_t = Index::idx_t;
    virtual void set_query(const float* query_vector) = 0;
    virtual void set_list(idx_t list_no, float coarse_dis) = 0;
    virtual float distance_to_code(const uint8_t* code) const = 0;
    /// `codes` may point at any code of the current list, `n` codes are scanned from there (IndexIVFFlat.cpp:117-137)
    virtual size_t scan_codes(size_t n, const uint8_t* codes, const idx_t* ids, float* distances, idx_t* labels,
                              size_t k) const = 0;
    /// results within `radius`, reported to `result` in the order of the codes (IndexIVF.h:349-354; the default fails, as the
    /// reference's: IndexIVF.cpp:949-955)
    virtual void scan_codes_range(size_t n, const uint8_t* codes, const idx_t* ids, float radius, RangeQueryResult& result) const;
    virtual ~InvertedListScanner() {}
};

/// owner[l] in [0, nshard) for every list: longest list first onto the least loaded shard (deterministic; the rule
/// auncel_amd/sharding.py:assign_owners and bench.py --mode shards use)
std::vector<int> ivf_list_owners(const IndexIVF& index, int nshard);

struct IndexIVFStats {
    size_t nq, nlist, ndis, nheap_updates;
    double quantization_time, search_time;
    IndexIVFStats() { reset(); }
    void reset();
};
extern IndexIVFStats indexIVF_stats;

}  // namespace faiss
