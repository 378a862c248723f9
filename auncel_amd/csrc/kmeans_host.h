// Host-side pieces of the reference's k-means that are sequential by definition (Clustering.cpp:75-226,
// utils.cpp:111-137,229-239,1126-1159): its random generator and permutation, the splitting of void clusters, the
// centroid post-processing.  Shared by the engine's GPU k-means (ivf_engine.hip) and by faiss::Clustering of the class
// mirror (host/faiss_amd.cpp), which must make the same choices bit for bit.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <random>
#include <utility>
#include <vector>

namespace amdivf_kmeans {

struct RefRng {  // faiss::RandomGenerator
    std::mt19937 mt;
    explicit RefRng(long seed) : mt((unsigned int)seed) {}
    int rand_int(int max) { return mt() % max; }
    float rand_float() { return mt() / float(mt.max()); }
};

inline void rand_perm(int* perm, size_t n, long seed) {
    for (size_t i = 0; i < n; i++) perm[i] = (int)i;
    RefRng rng(seed);
    for (size_t i = 0; i + 1 < n; i++) {
        int i2 = (int)i + rng.rand_int((int)(n - i));
        std::swap(perm[i], perm[i2]);
    }
}

// Empty clusters after an update step (the reference's rule, utils.cpp:1126-1159, as a specification: which cluster donates is
// decided by a fixed-seed generator, so the draws -- one per candidate donor, candidates cycling from cluster 0 -- and the
// arithmetic have to be these):
//   for every empty cluster e, in index order:
//     walk the clusters round-robin from 0; candidate c donates when a fresh draw u in [0, 1] falls below (size_c - 1) / (n - k);
//     e becomes a copy of the donor, then the two are pushed apart: even coordinates of e are scaled by 1 + 2^-10 and the donor's by
//     1 - 2^-10, odd coordinates the other way round; e takes half of the donor's points (rounded down).
// means: k x d, updated in place; population: points per cluster, updated.  Returns how many clusters were re-seeded.
inline int split_void_clusters(float* means, std::vector<size_t>& population, size_t d, size_t k, size_t n) {
    RefRng draws(1234);
    const double nudge = 1.0 / 1024.0;
    int reseeded = 0;
    for (size_t empty = 0; empty < k; empty++) {
        if (population[empty] != 0) continue;
        size_t donor = 0;
        for (;;) {
            const float share = (population[donor] - 1.0) / (float)(n - k);  // (double arithmetic, rounded to float: as specified)
            const float u = draws.rand_float();
            if (u < share) break;
            donor = donor + 1 == k ? 0 : donor + 1;
        }
        float* into = means + empty * d;
        float* from = means + donor * d;
        for (size_t j = 0; j < d; j++) {
            into[j] = from[j];
            const bool even = (j & 1) == 0;
            into[j] *= even ? 1 + nudge : 1 - nudge;  // (float *= double: one rounding)
            from[j] *= even ? 1 - nudge : 1 + nudge;
        }
        population[empty] = population[donor] / 2;
        population[donor] -= population[empty];
        reseeded++;
    }
    return reseeded;
}

// fvec_norm_L2sqr in the reference's SSE order (utils_simd.cpp:137-155)
inline float norm_L2sqr_sse(const float* x, size_t d) {
    float s[4] = {0, 0, 0, 0};
    size_t i = 0;
    for (; i + 4 <= d; i += 4)
        for (int l = 0; l < 4; l++) s[l] += x[i + l] * x[i + l];
    for (int l = 0; i + l < d; l++) s[l] += x[i + l] * x[i + l];
    return (s[0] + s[1]) + (s[2] + s[3]);
}

// Clustering::post_process_centroids (Clustering.cpp:63-73, fvec_renorm_L2 utils.cpp:377-392)
inline void post_process(float* c, size_t d, size_t k, bool spherical, bool int_centroids) {
    if (spherical) {
        for (size_t i = 0; i < k; i++) {
            float* xi = c + i * d;
            float nr = norm_L2sqr_sse(xi, d);
            if (nr > 0) {
                const float inv_nr = 1.0 / sqrtf(nr);
                for (size_t j = 0; j < d; j++) xi[j] *= inv_nr;
            }
        }
    }
    if (int_centroids)
        for (size_t i = 0; i < k * d; i++) c[i] = roundf(c[i]);
}

}  // namespace amdivf_kmeans
