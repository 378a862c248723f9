/* TEST INFRASTRUCTURE ONLY (oracle/): C ABI of the CPU restatement of the reference's
 * IVF-Flat search hot path.  See ivf_oracle.cpp for the reference file:line each function
 * follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (auncel_amd/, libauncel_amd.so) never does.
 *
 * Parity status: PINNED.  Every function below is checked against outputs of the compiled
 * reference (oracle/_ref/ref_harness, flags -O3 -msse4 -mpopcnt) committed as
 * tests/golden/<case>.npz by tests/test_oracle_golden.py; the only unpinned piece is the BLAS
 * coarse path (orc_knn with gemm=1), whose summation order belongs to the vendor BLAS.
 */
#ifndef IVF_ORACLE_H
#define IVF_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_METRIC_IP 0
#define ORC_METRIC_L2 1

/* inverted lists, CSR-packed: list l holds vectors [off[l], off[l+1]) of codes/ids */
typedef struct {
    int metric;
    size_t d, nlist;
    const size_t* list_off; /* nlist + 1 */
    const float* codes;     /* off[nlist] x d */
    const int64_t* ids;     /* off[nlist] */
} orc_index_t;

/* Auncel tuner state (error_pro, IVF_pro.h:79-180) */
typedef struct {
    size_t max_topk, query_topk, ntraces;
    float multipler, std_m;
    const float* interdis_cem; /* nlist(nlist-1)/2 */
    const float* arcos_list;   /* 500 */
    const size_t* trace_off;   /* ntraces + 1, offsets into trace_x/y/std */
    const float *trace_x, *trace_y, *trace_std;
    const float* require_acc; /* indexed by query id (i + offset) */
    const float* gt_D;        /* train_D: query-id x max_topk, may be NULL if !profile */
    size_t* my_nprobe;        /* in/out, indexed by query id */
    float* t_recalls;         /* out, indexed by query id */
    int profile;
    int overhead_profile;     /* IVF_pro.h:95: rule evaluated on every probe, no stop before stage nlist / 8 (eval/overhead.cpp) */
} orc_tuner_t;

float orc_fvec_L2sqr(const float* x, const float* y, size_t d);
float orc_fvec_inner_product(const float* x, const float* y, size_t d);

/* k-NN of nx queries in ny vectors, sorted best first (IndexFlat::search).  gemm=0: exact
 * per-pair path (knn_*_sse); gemm=1: norms + dot-product formulation (knn_*_blas, UNPINNED) */
void orc_knn(int metric, const float* x, const float* y, size_t d, size_t nx, size_t ny, size_t k, float* D,
             int64_t* I, int gemm, int nthreads);

/* packed upper-triangular centroid-to-centroid table (Level1Quantizer::train_q1) */
void orc_interdis(int metric, float* centroids_inout, size_t nlist, size_t d, float* out);

void orc_arcos_table(float* out500);

/* error_pro::set_online: returns 0, or -1 when the reference would throw */
int orc_set_online(int metric, size_t nlist, const float* cd, const int64_t* ci, const float* interdis_cem,
                   const float* arcos_list, float* disToBoundary, float* cenTocen);

/* IVFFlatScanner::scan_codes on a raw heap; returns the number of heap updates */
size_t orc_scan_codes(int metric, size_t d, const float* query, size_t list_size, const float* codes,
                      const int64_t* ids, int64_t list_no, int store_pairs, size_t k, float* simi, int64_t* idxi);

/* IndexIVF::search_preassigned, plain (tuner == NULL) or tune mode (tuner != NULL).
 * stats: {nlist, ndis, nheap_updates} accumulated.  Returns 0 or -1 (reference would throw;
 * message through orc_last_error). */
int orc_search_preassigned(const orc_index_t* ix, size_t n, const float* x, size_t k, size_t nprobe,
                           const int64_t* keys, const float* coarse_dis, float* D, int64_t* I, int store_pairs,
                           size_t max_codes, orc_tuner_t* tuner, size_t offset, size_t* stats, int nthreads);

/* Clustering::train (Clustering.cpp:75-226) with an IndexFlat of `metric`, nredo 1, no input centroids: sub-sampling by
 * rand_perm(seed) beyond k * max_points_per_centroid, centroids seeded from rand_perm(seed + 1), niter x { assignment to the
 * nearest / most similar centroid; objective = float sum of the assignment distances in point order; km_update_centroids
 * (utils.cpp:1078-1159): fp32 sums in point order, division by the count, void clusters split off bigger ones with
 * RandomGenerator(1234) and the 1/1024 perturbation }; spherical / int_centroids post-processing.  gemm: assignment
 * through the |x|^2+|y|^2-2xy restatement of the BLAS branch (the reference takes it from 20 points on; vendor rounding
 * unpinned) instead of the exact kernel.  obj receives niter values. */
void orc_kmeans(int metric, size_t d, size_t n, const float* x, size_t k, int niter, long seed, size_t max_points_per_centroid,
                int spherical, int int_centroids, int gemm, float* centroids, float* obj, int nthreads);

/* IndexIVF::range_search_preassigned (IndexIVF.cpp:759-857) with IVFFlatScanner::scan_codes_range
 * (IndexIVFFlat.cpp:139-155): per query, probes in order, list entries in order, every entry with
 * radius > dis (L2) / radius < dis (IP).  Two calls: with labels == NULL only lims (n + 1) is filled;
 * then with buffers of lims[n] entries.  stats: {nlist, ndis} accumulated.  Returns 0 or -1 (invalid key). */
int orc_range_search_preassigned(const orc_index_t* ix, size_t n, const float* x, float radius, size_t nprobe,
                                 const int64_t* keys, size_t* lims, int64_t* labels, float* distances, size_t* stats);

/* training branch of search_preassigned: raw (sum_angle, kscaling) samples.
 * raw_traces: ntraces arrays of train_num*(max_topk/4) (x,y) pairs, pre-filled with (-1,-1) */
int orc_train_samples(const orc_index_t* ix, size_t n, const float* x, size_t max_topk, size_t nprobe,
                      const int64_t* keys, const float* coarse_dis, const float* interdis_cem,
                      const float* arcos_list, const float* gt_D, size_t offset, size_t train_num, float** raw_traces,
                      float* D, int64_t* I);

/* Trace::SB: returns number of buckets written to out_x/out_y/out_std (capacity n/bs + 1) */
size_t orc_trace_sb(float* raw_xy, size_t n, size_t bs, float* out_x, float* out_y, float* out_std);

/* merge_tables of IndexShards::search */
void orc_merge_tables(int metric, size_t n, size_t k, size_t nshard, const float* all_D, const int64_t* all_I,
                      float* D, int64_t* I);

const char* orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
