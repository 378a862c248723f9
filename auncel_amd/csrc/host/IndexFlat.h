// faiss::IndexFlat (Auncel/IndexFlat.h:21-87) -- the coarse quantiser.  `xb` is the host copy of the
// vectors (public in the reference, harnesses read it); search() runs on the GPU.
#pragma once
#include <mutex>
#include <vector>

#include "Index.h"

struct amd_ivf;

namespace faiss {

struct IndexFlat : Index {
    std::vector<float> xb;  ///< database vectors, size ntotal * d

    explicit IndexFlat(idx_t d, MetricType metric = METRIC_L2);
    IndexFlat() {}
    ~IndexFlat() override;

    void add(idx_t n, const float* x) override;
    void reset() override;
    /// IndexFlat::search -> knn_L2sqr / knn_inner_product (IndexFlat.cpp:42-56): exact kernel for
    /// n < 20 && d % 4 == 0, the BLAS-formulation otherwise, as the reference switches (utils.cpp:644-655)
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const override;

    /// which coarse mode search() asks the engine for: -1 reference switch (default), 0 exact, 1 GEMM
    int coarse_mode = -1;
    /// bumped by add() / reset(): device copies of the vectors (here and in an IndexIVF using this as its quantizer)
    /// are refreshed when it moves
    size_t version = 0;

   private:
    // search() is const and, as in the reference, may be called from several threads (shards sharing one quantizer): the device
    // copy is created lazily and the engine handle has one stream and one set of work buffers, so calls are serialised here
    mutable std::mutex gpu_mu_;
    mutable amd_ivf* gpu_ = nullptr;
    mutable idx_t gpu_ntotal_ = -1;
};

struct IndexFlatL2 : IndexFlat {
    explicit IndexFlatL2(idx_t d) : IndexFlat(d, METRIC_L2) {}
    IndexFlatL2() {}
};

struct IndexFlatIP : IndexFlat {
    explicit IndexFlatIP(idx_t d) : IndexFlat(d, METRIC_INNER_PRODUCT) {}
    IndexFlatIP() {}
};

}  // namespace faiss
