// faiss::Clustering as the Auncel tree declares it (Auncel/Clustering.h:20-101): k-means with the reference's
// procedure (Clustering.cpp:75-226, utils.cpp:229-239,1078-1159); the assignment of every iteration -- all of the
// arithmetic that matters -- runs on the MI355X through the index passed to train(), the centroid update follows the
// reference's fp32 summation order on the host.
#pragma once
#include <vector>

#include "Index.h"

namespace faiss {

struct ClusteringParameters {
    int niter = 25;
    int nredo = 1;
    bool verbose = false;
    bool spherical = false;
    bool int_centroids = false;
    bool update_index = false;
    bool frozen_centroids = false;
    int min_points_per_centroid = 39;
    int max_points_per_centroid = 256;
    int seed = 1234;
};

struct Clustering : ClusteringParameters {
    typedef Index::idx_t idx_t;
    size_t d;  ///< dimension of the vectors
    size_t k;  ///< nb of centroids

    /// centroids (k * d)
    std::vector<float> centroids;
    /// objective values (sum of distances reported by index) over iterations
    std::vector<float> obj;

    Clustering(int d, int k);
    Clustering(int d, int k, const ClusteringParameters& cp);

    /// Index is used during the assignment stage
    virtual void train(idx_t n, const float* x, faiss::Index& index);

    /// Post-process the centroids after each centroid update (spherical / int_centroids)
    void post_process_centroids();

    virtual ~Clustering() {}
};

/// simplified interface: returns the final quantization error (Clustering.cpp:244-256)
float kmeans_clustering(size_t d, size_t n, size_t k, const float* x, float* centroids);

}  // namespace faiss
