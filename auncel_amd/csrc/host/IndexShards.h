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

}  // namespace faiss
