// faiss::IndexShards (Auncel/IndexShards.h:20-100): fan a query batch out to sub-indexes (one per GPU
// when lists are sharded by list id) and k-way merge their sorted rows on the host.
#pragma once
#include <vector>

#include "Index.h"

namespace faiss {

struct IndexShards : Index {
    bool threaded;
    bool successive_ids;
    std::vector<Index*> shards;

    explicit IndexShards(idx_t d, bool threaded = false, bool successive_ids = true);
    void add_shard(Index* index);
    int count() const { return (int)shards.size(); }
    Index* at(int i) { return shards[i]; }

    void add(idx_t n, const float* x) override;
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const override;
    void train(idx_t n, const float* x) override;
    void reset() override;
};

struct IndexIVFFlat;
/// `index` cut into `nshard` IndexIVFFlat sub-indexes BY LIST (IndexIVF::copy_subset_to type 4: byte-balanced owners) that share its
/// quantizer and keep its ids, under an IndexShards(successive_ids = false) that owns them: the list-id shards of north_star /
/// BASELINE config 4 for a C++ caller.  The reference's analogue cuts by vector (gpu/GpuAutoTune.cpp:201-221, copy_subset_to 1 / 2).
/// Shard i goes to GPU i % (GPUs of the node) unless `devices` says otherwise.
struct IndexShardsByList : IndexShards {
    std::vector<IndexIVFFlat*> owned;
    IndexShardsByList(const IndexIVFFlat& index, int nshard, bool threaded = true, const int* devices = nullptr);
    ~IndexShardsByList() override;
    IndexShardsByList(const IndexShardsByList&) = delete;
    IndexShardsByList& operator=(const IndexShardsByList&) = delete;
};

}  // namespace faiss
