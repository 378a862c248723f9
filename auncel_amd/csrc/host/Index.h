// faiss::Index as the Auncel tree declares it (Auncel/Index.h:42-134): same field and method names,
// so that code written against the reference (eval/*.cpp) compiles against this mirror.  Every
// compute method of the concrete classes goes to the MI355X engine through include/auncel_amd.h.
#pragma once
#include <cstddef>
// (the system headers the reference's Index.h brings in: callers such as eval/bound.cpp rely on them, Auncel/Index.h:14-17)
#include <cstdio>
#include <sstream>
#include <typeinfo>
#include <cstdint>
#include <string>

namespace faiss {

enum MetricType { METRIC_INNER_PRODUCT = 0, METRIC_L2 = 1 };
typedef enum IndexType { IVF, HNSW, OTHER } IndexType;

struct RangeSearchResult;
struct IDSelector;

struct Index {
    using idx_t = long;
    using component_t = float;
    using distance_t = float;

    // Auncel additions (Index.h:71-77)
    bool tune = false;
    IndexType type = OTHER;
    virtual void set_tune_mode() { tune = true; }
    virtual void set_tune_off() { tune = false; }

    /// MI355X this index's engine lives on: -1 = AUNCEL_AMD_DEVICE or device 0.  Not in the reference (its indexes are host
    /// objects); IndexShards::add_shard deals unassigned shards round-robin over the node's GPUs, one host thread each.
    int amd_device = -1;
    virtual void set_device(int device) { amd_device = device; }
    /// true once the index holds device state on `amd_device` (it can no longer be given another device)
    virtual bool device_bound() const { return false; }

    int d;
    idx_t ntotal;
    bool verbose;
    bool is_trained;
    MetricType metric_type;

    explicit Index(idx_t d = 0, MetricType metric = METRIC_L2)
        : d((int)d), ntotal(0), verbose(false), is_trained(true), metric_type(metric) {}
    virtual ~Index() {}

    virtual void train(idx_t /*n*/, const float* /*x*/) {}
    virtual void add(idx_t n, const float* x) = 0;
    virtual void add_with_ids(idx_t n, const float* x, const long* xids);
    virtual void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const = 0;
    /// all vectors within `radius` of each query (Index.h:106-117); only the IVF classes implement it here
    virtual void range_search(idx_t n, const float* x, float radius, RangeSearchResult* result) const;
    virtual void reset() = 0;
    /// removes the ids the selector names, returns how many (Index.h:159-162); only the IVF classes implement it here
    virtual long remove_ids(const IDSelector& sel);
    /// stored vector of one id / a range of ids, search + stored vectors of the results (Index.h:119-157); only the IVF
    /// classes implement them here
    virtual void reconstruct(idx_t key, float* recons) const;
    virtual void reconstruct_n(idx_t i0, idx_t ni, float* recons) const;
    virtual void search_and_reconstruct(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, float* recons) const;
    /// nearest-neighbour labels only (Index.cpp:42-48)
    void assign(idx_t n, const float* x, idx_t* labels, idx_t k = 1);
};

}  // namespace faiss
