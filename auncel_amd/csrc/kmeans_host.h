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

// the tail of km_update_centroids: every cluster without points is re-seeded from a bigger one
// (centroids: k x d means, hassign: points per cluster, both updated); returns the number of splits
inline int split_void_clusters(float* centroids, std::vector<size_t>& hassign, size_t d, size_t k, size_t n) {
    size_t nsplit = 0;
    RefRng rng(1234);
    const double EPS = 1 / 1024.;
    for (size_t ci = 0; ci < k; ci++) {
        if (hassign[ci] == 0) {
            size_t cj;
            for (cj = 0; 1; cj = (cj + 1) % k) {
                float p = (hassign[cj] - 1.0) / (float)(n - k);
                float r = rng.rand_float();
                if (r < p) break;
            }
            memcpy(centroids + ci * d, centroids + cj * d, sizeof(*centroids) * d);
            for (size_t j = 0; j < d; j++) {
                if (j % 2 == 0) {
                    centroids[ci * d + j] *= 1 + EPS;
                    centroids[cj * d + j] *= 1 - EPS;
                } else {
                    centroids[ci * d + j] *= 1 - EPS;
                    centroids[cj * d + j] *= 1 + EPS;
                }
            }
            hassign[ci] = hassign[cj] / 2;
            hassign[cj] -= hassign[ci];
            nsplit++;
        }
    }
    return (int)nsplit;
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
