// TEST INFRASTRUCTURE ONLY (oracle/): CPU restatement of the reference's IVF-Flat search
// hot path (Auncel = Faiss 1.5.2 + IVF_pro).  It is the checker for the HIP path and the
// "port" CPU baseline of bench.py; the product never links or loads it.
//
// Parity status: PINNED against the compiled reference (oracle/_ref/ref_harness, reference
// default flags -O3 -msse4 -mpopcnt) through tests/golden/*.npz, see
// tests/test_oracle_golden.py.  Unpinned: the gemm=1 branch of orc_knn (vendor BLAS order).
//
// Build: g++ -O3 -msse4 -ffp-contract=off (no FMA contraction: the reference's SSE build
// rounds every product and every sum separately).
//
// Everything below is written from the behaviour of these reference functions
// (paths relative to /root/reference/Auncel):
//   fvec_L2sqr / fvec_inner_product (SSE)    utils_simd.cpp:391-443
//   heap_pop/push/heapify/reorder            Heap.h:88-142,184-208,295-322
//   knn_L2sqr_sse / knn_inner_product_sse    utils.cpp:417-490
//   knn_L2sqr_blas / knn_inner_product_blas  utils.cpp:494-608
//   IVFFlatScanner::scan_codes               IndexIVFFlat.cpp:117-137
//   IndexIVF::search_preassigned             IndexIVF.cpp:382-736
//   fvec_inter_vecs / train_q1 table         IVF_pro.cpp:21-39, IndexIVF.cpp:97-111
//   cosine_theorem, kscaling                 IVF_pro.cpp:41-51, 72-82
//   Trace::search / Trace::SB                IVF_pro.cpp:84-149
//   error_pro::{construct_arcos,sum_angle,arcos,set_online,cur_num}  IVF_pro.cpp:151-291
//   merge_tables                             IndexShards.cpp:44-105
#include "ivf_oracle.h"

#include <omp.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <random>
#include <string>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_err;

struct OracleError {
    std::string msg;
};

// ---------------------------------------------------------------- distances (SSE order)
// Four running sums, lane l taking elements 4i+l; products and sums rounded separately;
// a zero-padded tail; final (s0+s1)+(s2+s3)  [utils_simd.cpp:391-416 + two hadd_ps]
inline float l2sqr(const float* x, const float* y, size_t d) {
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    size_t i = 0;
    for (; i + 4 <= d; i += 4) {
        float a0 = x[i] - y[i], a1 = x[i + 1] - y[i + 1], a2 = x[i + 2] - y[i + 2], a3 = x[i + 3] - y[i + 3];
        s0 += a0 * a0;
        s1 += a1 * a1;
        s2 += a2 * a2;
        s3 += a3 * a3;
    }
    if (i < d) {
        float a[4] = {0, 0, 0, 0};
        for (size_t j = 0; i + j < d; j++) a[j] = x[i + j] - y[i + j];
        s0 += a[0] * a[0];
        s1 += a[1] * a[1];
        s2 += a[2] * a[2];
        s3 += a[3] * a[3];
    }
    return (s0 + s1) + (s2 + s3);
}

inline float inner(const float* x, const float* y, size_t d) {
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    size_t i = 0;
    for (; i + 4 <= d; i += 4) {
        s0 += x[i] * y[i];
        s1 += x[i + 1] * y[i + 1];
        s2 += x[i + 2] * y[i + 2];
        s3 += x[i + 3] * y[i + 3];
    }
    {  // the reference always adds the (possibly all-zero) masked tail: utils_simd.cpp:432-437
        float p[4] = {0, 0, 0, 0};
        for (size_t j = 0; i + j < d; j++) p[j] = x[i + j] * y[i + j];
        s0 += p[0];
        s1 += p[1];
        s2 += p[2];
        s3 += p[3];
    }
    return (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------- binary heap (Heap.h)
// IsMax = true : max-heap, keeps the k smallest (L2).  "better(a,b)" == C::cmp(a,b).
template <bool IsMax> struct Ord {
    static inline bool cmp(float a, float b) { return IsMax ? a > b : a < b; }
    static inline float neutral() { return IsMax ? FLT_MAX : -FLT_MAX; }
};

template <bool IsMax, class TI> inline void hpop(size_t k, float* val, TI* ids) {
    val--;
    ids--;
    float v = val[k];
    size_t i = 1;
    for (;;) {
        size_t c1 = i << 1, c2 = c1 + 1;
        if (c1 > k) break;
        if (c2 == k + 1 || Ord<IsMax>::cmp(val[c1], val[c2])) {
            if (Ord<IsMax>::cmp(v, val[c1])) break;
            val[i] = val[c1];
            ids[i] = ids[c1];
            i = c1;
        } else {
            if (Ord<IsMax>::cmp(v, val[c2])) break;
            val[i] = val[c2];
            ids[i] = ids[c2];
            i = c2;
        }
    }
    val[i] = val[k];
    ids[i] = ids[k];
}

template <bool IsMax, class TI> inline void hpush(size_t k, float* val, TI* ids, float v, TI id) {
    val--;
    ids--;
    size_t i = k;
    while (i > 1) {
        size_t f = i >> 1;
        if (!Ord<IsMax>::cmp(v, val[f])) break;
        val[i] = val[f];
        ids[i] = ids[f];
        i = f;
    }
    val[i] = v;
    ids[i] = id;
}

template <bool IsMax> inline void hinit(size_t k, float* val, int64_t* ids) {
    for (size_t i = 0; i < k; i++) {
        val[i] = Ord<IsMax>::neutral();
        ids[i] = -1;
    }
}

template <bool IsMax> inline size_t hreorder(size_t k, float* val, int64_t* ids) {
    size_t ii = 0;
    for (size_t i = 0; i < k; i++) {
        float v = val[0];
        int64_t id = ids[0];
        hpop<IsMax>(k - i, val, ids);
        val[k - ii - 1] = v;
        ids[k - ii - 1] = id;
        if (id != -1) ii++;
    }
    size_t nel = ii;
    memmove(val, val + k - ii, ii * sizeof(*val));
    memmove(ids, ids + k - ii, ii * sizeof(*ids));
    for (; ii < k; ii++) {
        val[ii] = Ord<IsMax>::neutral();
        ids[ii] = -1;
    }
    return nel;
}

// ---------------------------------------------------------------- Auncel geometry
inline float arcos_lut(const float* lut, float x) {  // error_pro::arcos, arcos_size = 500
    if (!(x <= 1. && x >= -1.)) throw OracleError{"arcos's domain definition is [-1, 1]"};
    const size_t arcos_size = 500;
    int index = x * arcos_size / 2 + arcos_size / 2;
    return lut[index];
}

inline float cosine_theorem(float a, float b, float c) {
    if (!(a <= b)) throw OracleError{"cosine theorem's prerequisites"};
    float temp = std::pow(a, 2) + std::pow(c, 2) - std::pow(b, 2);  // pow(float,int) -> double
    temp = temp / (2 * c);
    return c / 2 - temp;
}

inline float sum_angle(const float* lut, float kdis, const float* dtb, size_t n, size_t start) {
    float sum = 0;
    for (size_t i = start; i < start + n; i++) {
        if (dtb[i] >= kdis) continue;
        sum += arcos_lut(lut, dtb[i] / kdis);
    }
    return sum;
}

struct TraceView {
    const float *x, *y, *sd;
    size_t n;
    float search(float k, float std_m) const {  // Trace::search
        float sc = std_m;
        if (k <= x[0]) return y[0] + sc * sd[0];
        if (k >= x[n - 1]) {
            float ampli = k / x[n - 1];
            return (y[n - 1] + sc * sd[n - 1]) * ampli;
        }
        size_t high = n - 1, low = 0, middle = 0;
        while (low <= high) {
            middle = (low + high) / 2;
            if (x[middle] < k) low = middle + 1;
            else high = middle - 1;
        }
        if (x[low] > k) low--;
        return y[low] + sc * sd[low];
    }
};

inline float kscaling(float kdis, size_t in, const float* gt, size_t max_topk) {
    size_t index = 0;
    for (; index < max_topk; index++) {
        if (std::fabs(gt[index] - kdis) / kdis < 1e-5 || std::fabs(gt[index] - kdis) < 1e-5) break;
    }
    if (index >= max_topk) return -1;
    return (index + 1) / float(in + 1);
}

void set_online(int metric, size_t nlist, const float* cd, const int64_t* ci, const float* interdis, const float* lut,
                float* dtb, float* c2c) {
    size_t max_num = nlist / 8 + 20;
    size_t cur = ci[0];
    std::vector<float> cend(max_num);
    if (metric == ORC_METRIC_IP)
        for (size_t i = 0; i < max_num; i++) cend[i] = arcos_lut(lut, cd[i]);
    for (size_t k = 1; k <= max_num; k++) {
        size_t dst = ci[k];
        size_t i = cur < dst ? cur : dst, j = cur < dst ? dst : cur;
        c2c[k - 1] = interdis[(2 * nlist - 1 - i) * i / 2 + j - 1 - i];
    }
    for (size_t k = 0; k < max_num - 1; k++)
        dtb[k] = metric == ORC_METRIC_L2 ? cosine_theorem(cd[0], cd[k + 1], c2c[k])
                                         : cosine_theorem(cend[0], cend[k + 1], c2c[k]);
    dtb[max_num - 1] = 0;
}

size_t cur_num(const orc_tuner_t* t, const float* lut, const float* Ds, const float* dtb, size_t index,
               size_t query_k) {
    TraceView tr{t->trace_x + t->trace_off[index], t->trace_y + t->trace_off[index],
                 t->trace_std + t->trace_off[index], t->trace_off[index + 1] - t->trace_off[index]};
    size_t nprobe = size_t(1) << index;
    size_t high = query_k - 1, low = 0, middle = 0;
    float std_m = t->std_m;
    if (query_k * tr.search(sum_angle(lut, Ds[high], dtb, 15, nprobe - 1), std_m) <= query_k * 1.005) return query_k;
    while (low <= high) {
        middle = (low + high) / 2;
        if (middle <= 0) return 0;
        if ((middle + 1) * tr.search(sum_angle(lut, Ds[middle], dtb, 15, nprobe - 1), std_m) <= query_k)
            low = middle + 1;
        else
            high = middle - 1;
    }
    return low + 1;
}

// ---------------------------------------------------------------- the search driver
struct TrainCtx {
    const float* interdis;
    const float* lut;
    const float* gt_D;
    size_t train_num;
    float** raw;
};

template <bool IsMax>
size_t scan_list(const orc_index_t* ix, const float* q, int64_t key, bool store_pairs, size_t k, float* simi,
                 int64_t* idxi, size_t* nheap) {
    size_t b = ix->list_off[key], e = ix->list_off[key + 1], d = ix->d;
    const float* codes = ix->codes + b * d;
    for (size_t j = 0; j < e - b; j++) {
        float dis = IsMax ? l2sqr(q, codes + j * d, d) : inner(q, codes + j * d, d);
        if (Ord<IsMax>::cmp(simi[0], dis)) {
            hpop<IsMax>(k, simi, idxi);
            int64_t id = store_pairs ? (key << 32 | (int64_t)j) : ix->ids[b + j];
            hpush<IsMax>(k, simi, idxi, dis, id);
            (*nheap)++;
        }
    }
    return e - b;
}

template <bool IsMax>
void search_one(const orc_index_t* ix, size_t i, const float* x, size_t k, size_t nprobe, const int64_t* keys,
                const float* coarse_dis, float* D, int64_t* I, bool store_pairs, size_t max_codes,
                orc_tuner_t* t, const TrainCtx* tc, size_t offset, size_t* st) {
    const size_t nlist = ix->nlist, d = ix->d;
    const bool tune = t != nullptr, training = tc != nullptr;
    const size_t id_q = i + offset;
    const float* q = x + i * d;
    float* simi = D + i * k;
    int64_t* idxi = I + i * k;
    hinit<IsMax>(k, simi, idxi);
    const int64_t* qk = keys + i * nprobe;
    const float* qcd = coarse_dis + i * nprobe;
    const float* lut = tune ? t->arcos_list : training ? tc->lut : nullptr;

    std::vector<float> dtb, c2c;
    size_t query_k = 0, stoped = 0;
    float true_KD_K = 0, pre_val = 0;
    if (tune) {
        if (k != t->max_topk) throw OracleError{"tune mode needs k == max_topk"};
        query_k = t->query_topk;
        if (t->gt_D) true_KD_K = t->gt_D[id_q * k + query_k - 1];
    }
    if (tune || training) {
        size_t max_num = nlist / 8 + 20;
        if (nprobe <= max_num) throw OracleError{"tune/train mode needs nprobe > nlist/8 + 20"};
        dtb.assign(max_num, 0.f);
        c2c.assign(max_num, 0.f);
        set_online(ix->metric, nlist, qcd, qk, tune ? t->interdis_cem : tc->interdis, lut, dtb.data(), c2c.data());
    }
    std::vector<float> tmp(k);
    size_t nscan = 0;
    for (size_t ik = 0; ik < nprobe; ik++) {
        int64_t key = qk[ik];
        if (key >= 0) {
            if (key >= (int64_t)nlist) throw OracleError{"Invalid key"};
            if (ix->list_off[key + 1] > ix->list_off[key]) {
                st[0]++;
                nscan += scan_list<IsMax>(ix, q, key, store_pairs, k, simi, idxi, &st[2]);
            }
        }
        if (max_codes && nscan >= max_codes) break;
        if (tune) {
            size_t stage = ik + 1, ind = 0;
            size_t tmp_stage = stage >= nlist / 8 ? nlist / 8 - 1 : stage;
            while (tmp_stage > (size_t(1) << ind)) ind++;
            memcpy(tmp.data(), simi, sizeof(float) * k);
            if (!IsMax)
                for (size_t j = 0; j < k; j++) tmp[j] = arcos_lut(lut, tmp[j]);
            std::sort(tmp.begin(), tmp.end());
            size_t pre_num = cur_num(t, lut, tmp.data(), dtb.data(), ind, query_k);
            float recall = pre_num / float(query_k);
            size_t cnt = 0;
            float max_val = -1;
            size_t stops = t->require_acc[id_q] * 12;
            if (IsMax) {
                for (size_t j = 0; j < k; j++) {
                    max_val = std::fmax(max_val, simi[j]);
                    if (simi[j] <= true_KD_K * 1.0005) cnt++;
                }
            } else {
                max_val = FLT_MAX;
                for (size_t j = 0; j < k; j++) {
                    max_val = std::fmin(max_val, simi[j]);
                    if (simi[j] >= true_KD_K * 0.9995) cnt++;
                }
            }
            if (stage > 1) {
                if (max_val == pre_val) stoped++;
                else stoped = 0;
                if (stoped >= stops) recall = 1;
            }
            pre_val = max_val;
            float true_recall = cnt / float(query_k);
            float require_recall = t->require_acc[id_q];
            if (t->overhead_profile) {  // IndexIVF.cpp:614,634-637: the rule has run, its verdict is ignored
                if (stage >= nlist / 8) break;
                continue;  // (tune and training are never on together: Error_sys sets one or the other)
            }
            size_t& np = t->my_nprobe[id_q];
            if (recall >= require_recall && np == 0) {
                np = stage * t->multipler;
                if (np >= nlist) t->t_recalls[id_q] = 1.;
            }
            if (stage >= nlist / 8 && np == 0) {
                np = stage * t->multipler;
                if (np >= nlist) t->t_recalls[id_q] = 1.;
            }
            if (np != 0 && np <= stage) {
                if (t->profile) t->t_recalls[id_q] = true_recall;
                break;
            }
        }
        if (training) {
            size_t stage = ik + 1;
            if (stage > nlist / 8) break;
            if ((stage & (stage - 1)) != 0) continue;
            size_t ind = 0;
            while (stage != (size_t(1) << ind)) ind++;
            memcpy(tmp.data(), simi, sizeof(float) * k);
            std::sort(tmp.begin(), tmp.end());
            if (!IsMax) std::reverse(tmp.begin(), tmp.end());
            size_t count = 0;
            for (size_t ij = 0; ij < k; ij++) {
                float ks = kscaling(tmp[ij], ij, tc->gt_D + id_q * k, k);
                if (ks < 0) break;
                float tval = tmp[ij];
                if (!IsMax) tval = arcos_lut(lut, tval);
                float sum_a = sum_angle(lut, tval, dtb.data(), 15, stage - 1);
                float* slot = tc->raw[ind] + 2 * (id_q * (k / 4) + count++);
                slot[0] = sum_a;
                slot[1] = ks;
                if (count >= k / 4) break;
            }
        }
    }
    st[1] += nscan;
    hreorder<IsMax>(k, simi, idxi);
}

int search_driver(const orc_index_t* ix, size_t n, const float* x, size_t k, size_t nprobe, const int64_t* keys,
                  const float* coarse_dis, float* D, int64_t* I, bool store_pairs, size_t max_codes, orc_tuner_t* t,
                  const TrainCtx* tc, size_t offset, size_t* stats, int nthreads) {
    size_t s0 = 0, s1 = 0, s2 = 0;
    bool failed = false;
    std::string msg;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) reduction(+ : s0, s1, s2) schedule(dynamic, 4)
    for (size_t i = 0; i < n; i++) {
        size_t st[3] = {0, 0, 0};
        try {
            if (ix->metric == ORC_METRIC_L2)
                search_one<true>(ix, i, x, k, nprobe, keys, coarse_dis, D, I, store_pairs, max_codes, t, tc, offset, st);
            else
                search_one<false>(ix, i, x, k, nprobe, keys, coarse_dis, D, I, store_pairs, max_codes, t, tc, offset, st);
        } catch (const OracleError& e) {
#pragma omp critical
            {
                failed = true;
                msg = e.msg;
            }
        }
        s0 += st[0];
        s1 += st[1];
        s2 += st[2];
    }
    if (stats) {
        stats[0] += s0;
        stats[1] += s1;
        stats[2] += s2;
    }
    if (failed) {
        g_err = msg;
        return -1;
    }
    return 0;
}

template <bool IsMax>
void knn_exact(const float* x, const float* y, size_t d, size_t nx, size_t ny, size_t k, float* D, int64_t* I,
               int nthreads) {
#pragma omp parallel for num_threads(nthreads)
    for (size_t i = 0; i < nx; i++) {
        float* simi = D + i * k;
        int64_t* idxi = I + i * k;
        hinit<IsMax>(k, simi, idxi);
        for (size_t j = 0; j < ny; j++) {
            float dis = IsMax ? l2sqr(x + i * d, y + j * d, d) : inner(x + i * d, y + j * d, d);
            if (Ord<IsMax>::cmp(simi[0], dis)) {
                hpop<IsMax>(k, simi, idxi);
                hpush<IsMax>(k, simi, idxi, dis, (int64_t)j);
            }
        }
        hreorder<IsMax>(k, simi, idxi);
    }
}

// norms + dot products (knn_L2sqr_blas, utils.cpp:538-608): dis = |x|^2 + |y|^2 - 2 x.y, clamped at 0.
// The dot product here is a plain k-ordered sum; the reference's comes from the vendor sgemm.
template <bool IsMax>
void knn_gemm(const float* x, const float* y, size_t d, size_t nx, size_t ny, size_t k, float* D, int64_t* I,
              int nthreads) {
    std::vector<float> xn(nx), yn(ny);
    for (size_t i = 0; i < nx; i++) xn[i] = inner(x + i * d, x + i * d, d);
    for (size_t j = 0; j < ny; j++) yn[j] = inner(y + j * d, y + j * d, d);
#pragma omp parallel for num_threads(nthreads)
    for (size_t i = 0; i < nx; i++) {
        float* simi = D + i * k;
        int64_t* idxi = I + i * k;
        hinit<IsMax>(k, simi, idxi);
        for (size_t j = 0; j < ny; j++) {
            float ip = 0;
            for (size_t c = 0; c < d; c++) ip += x[i * d + c] * y[j * d + c];
            float dis = ip;
            if (IsMax) {
                dis = xn[i] + yn[j] - 2 * ip;
                if (dis < 0) dis = 0;
            }
            if (Ord<IsMax>::cmp(simi[0], dis)) {
                hpop<IsMax>(k, simi, idxi);
                hpush<IsMax>(k, simi, idxi, dis, (int64_t)j);
            }
        }
        hreorder<IsMax>(k, simi, idxi);
    }
}

template <bool IsMaxHeapOfShards>
void merge_tables(size_t n, size_t k, size_t nshard, const float* all_D, const int64_t* all_I, float* Dout,
                  int64_t* Iout) {
    // L2 -> CMin heap over (distance, shard) i.e. IsMaxHeapOfShards=false pops the smallest first
    size_t stride = n * k;
    std::vector<int> pointer(nshard), shard_ids(nshard);
    std::vector<float> heap_vals(nshard);
    for (size_t i = 0; i < n; i++) {
        const float* D_in = all_D + i * k;
        const int64_t* I_in = all_I + i * k;
        size_t heap_size = 0;
        for (size_t s = 0; s < nshard; s++) {
            pointer[s] = 0;
            if (I_in[stride * s] >= 0) hpush<IsMaxHeapOfShards, int>(++heap_size, heap_vals.data(), shard_ids.data(), D_in[stride * s], (int)s);
        }
        for (size_t j = 0; j < k; j++) {
            if (heap_size == 0) {
                Iout[i * k + j] = -1;
                Dout[i * k + j] = Ord<IsMaxHeapOfShards>::neutral();
            } else {
                int s = shard_ids[0];
                int& p = pointer[s];
                Dout[i * k + j] = heap_vals[0];
                Iout[i * k + j] = I_in[stride * s + p];
                hpop<IsMaxHeapOfShards, int>(heap_size--, heap_vals.data(), shard_ids.data());
                p++;
                if ((size_t)p < k && I_in[stride * s + p] >= 0)
                    hpush<IsMaxHeapOfShards, int>(++heap_size, heap_vals.data(), shard_ids.data(), D_in[stride * s + p], s);
            }
        }
    }
}

}  // namespace

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

float orc_fvec_L2sqr(const float* x, const float* y, size_t d) { return l2sqr(x, y, d); }
float orc_fvec_inner_product(const float* x, const float* y, size_t d) { return inner(x, y, d); }

void orc_knn(int metric, const float* x, const float* y, size_t d, size_t nx, size_t ny, size_t k, float* D,
             int64_t* I, int gemm, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (metric == ORC_METRIC_L2) {
        if (gemm) knn_gemm<true>(x, y, d, nx, ny, k, D, I, nthreads);
        else knn_exact<true>(x, y, d, nx, ny, k, D, I, nthreads);
    } else {
        if (gemm) knn_gemm<false>(x, y, d, nx, ny, k, D, I, nthreads);
        else knn_exact<false>(x, y, d, nx, ny, k, D, I, nthreads);
    }
}

// ---- Clustering::train (Clustering.cpp:75-226), rand_perm (utils.cpp:229-239), RandomGenerator (utils.cpp:111-137),
//      km_update_centroids (utils.cpp:1078-1159)
namespace {
struct RefRng {
    std::mt19937 mt;
    explicit RefRng(long seed) : mt((unsigned int)seed) {}
    int rand_int(int max) { return mt() % max; }
    float rand_float() { return mt() / float(mt.max()); }
};
void ref_rand_perm(std::vector<int>& perm, size_t n, long seed) {
    perm.resize(n);
    for (size_t i = 0; i < n; i++) perm[i] = (int)i;
    RefRng rng(seed);
    for (size_t i = 0; i + 1 < n; i++) {
        int i2 = (int)i + rng.rand_int((int)(n - i));
        std::swap(perm[i], perm[i2]);
    }
}
int ref_km_update_centroids(const float* x, float* centroids, const int64_t* assign, size_t d, size_t k, size_t n) {
    std::vector<size_t> hassign(k);
    memset(centroids, 0, sizeof(*centroids) * d * k);
    for (size_t i = 0; i < n; i++) {  // per centroid: fp32 sums in point order (the reference splits the centroids over threads)
        const size_t ci = (size_t)assign[i];
        float* c = centroids + ci * d;
        hassign[ci]++;
        for (size_t j = 0; j < d; j++) c[j] += x[i * d + j];
    }
    for (size_t ci = 0; ci < k; ci++) {
        float* c = centroids + ci * d;
        float ni = (float)hassign[ci];
        if (ni != 0)
            for (size_t j = 0; j < d; j++) c[j] /= ni;
    }
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
void ref_post_process(float* c, size_t d, size_t k, int spherical, int int_centroids) {
    if (spherical) {  // fvec_renorm_L2 (utils.cpp): x /= sqrt(|x|^2) when the norm is positive
        for (size_t i = 0; i < k; i++) {
            float* xi = c + i * d;
            float nr = inner(xi, xi, d);  // fvec_norm_L2sqr (utils_simd.cpp:137-155): the same 4-lane sums
            if (nr > 0) {
                const float inv_nr = 1.0 / sqrtf(nr);
                for (size_t j = 0; j < d; j++) xi[j] *= inv_nr;
            }
        }
    }
    if (int_centroids)
        for (size_t i = 0; i < k * d; i++) c[i] = roundf(c[i]);
}
}  // namespace

void orc_kmeans(int metric, size_t d, size_t n, const float* x_in, size_t k, int niter, long seed, size_t max_pts, int spherical,
                int int_centroids, int gemm, float* centroids, float* obj, int nthreads) {
    const float* x = x_in;
    std::vector<float> sub;
    size_t nx = n;
    if (nx > k * max_pts) {
        std::vector<int> perm;
        ref_rand_perm(perm, nx, seed);
        nx = k * max_pts;
        sub.resize(nx * d);
        for (size_t i = 0; i < nx; i++) memcpy(&sub[i * d], x_in + (size_t)perm[i] * d, sizeof(float) * d);
        x = sub.data();
    }
    if (nx == k) {  // corner case of the reference: copy the training set
        memcpy(centroids, x_in, sizeof(float) * d * k);
        return;
    }
    std::vector<int> perm;
    ref_rand_perm(perm, nx, seed + 1);
    for (size_t i = 0; i < k; i++) memcpy(centroids + i * d, x + (size_t)perm[i] * d, d * sizeof(float));
    ref_post_process(centroids, d, k, spherical, int_centroids);
    std::vector<float> dis(nx);
    std::vector<int64_t> assign(nx);
    for (int it = 0; it < niter; it++) {
        orc_knn(metric, x, centroids, d, nx, k, 1, dis.data(), assign.data(), gemm, nthreads);
        float err = 0;
        for (size_t j = 0; j < nx; j++) err += dis[j];
        obj[it] = err;
        ref_km_update_centroids(x, centroids, assign.data(), d, k, nx);
        ref_post_process(centroids, d, k, spherical, int_centroids);
    }
}

void orc_interdis(int metric, float* c, size_t nlist, size_t d, float* out) {
    if (metric == ORC_METRIC_IP) {
        // the reference renormalises centroid 0, nlist times (IndexIVF.cpp:102-107), then takes
        // acos of the raw inner products
        for (size_t i = 0; i < nlist; i++) {
            float norm = sqrtf(inner(c, c, d));
            for (size_t j = 0; j < d; j++) c[j] /= norm;
        }
    }
    for (size_t i = 0; i < nlist; i++)
        for (size_t j = i + 1; j < nlist; j++) {
            float v = metric == ORC_METRIC_L2 ? l2sqr(c + i * d, c + j * d, d) : inner(c + i * d, c + j * d, d);
            if (metric == ORC_METRIC_IP) v = std::acos(v);
            out[(2 * nlist - 1 - i) * i / 2 + j - 1 - i] = v;
        }
}

void orc_arcos_table(float* out) {
    int len = 500;
    float sc = len / 2;
    for (int i = 0; i < len; i++) {
        float x = float(i - sc) / sc;
        out[i] = std::acos(x);
    }
}

int orc_set_online(int metric, size_t nlist, const float* cd, const int64_t* ci, const float* interdis_cem,
                   const float* arcos_list, float* dtb, float* c2c) {
    try {
        set_online(metric, nlist, cd, ci, interdis_cem, arcos_list, dtb, c2c);
    } catch (const OracleError& e) {
        g_err = e.msg;
        return -1;
    }
    return 0;
}

size_t orc_scan_codes(int metric, size_t d, const float* query, size_t list_size, const float* codes,
                      const int64_t* ids, int64_t list_no, int store_pairs, size_t k, float* simi, int64_t* idxi) {
    size_t off[2] = {0, list_size};
    orc_index_t ix{metric, d, 1, off, codes, ids};
    size_t nheap = 0;
    // scan_list looks lists up by key; present the single list as key 0 and patch pair ids
    if (metric == ORC_METRIC_L2) {
        for (size_t j = 0; j < list_size; j++) {
            float dis = l2sqr(query, codes + j * d, d);
            if (simi[0] > dis) {
                hpop<true>(k, simi, idxi);
                hpush<true>(k, simi, idxi, dis, store_pairs ? (list_no << 32 | (int64_t)j) : ids[j]);
                nheap++;
            }
        }
    } else {
        for (size_t j = 0; j < list_size; j++) {
            float dis = inner(query, codes + j * d, d);
            if (simi[0] < dis) {
                hpop<false>(k, simi, idxi);
                hpush<false>(k, simi, idxi, dis, store_pairs ? (list_no << 32 | (int64_t)j) : ids[j]);
                nheap++;
            }
        }
    }
    (void)ix;
    return nheap;
}

// IndexIVF.cpp:759-857 (parallel_mode 0: a query's results are its probes' in order; the per-thread partial results
// are laid out query by query by RangeSearchPartialResult::finalize) + IndexIVFFlat.cpp:139-155
int orc_range_search_preassigned(const orc_index_t* ix, size_t n, const float* x, float radius, size_t nprobe,
                                 const int64_t* keys, size_t* lims, int64_t* labels, float* distances, size_t* stats) {
    const size_t d = ix->d;
    size_t pos = 0, nlistv = 0, ndis = 0;
    for (size_t i = 0; i < n; i++) {
        lims[i] = pos;
        const float* xi = x + i * d;
        for (size_t ik = 0; ik < nprobe; ik++) {
            const int64_t key = keys[i * nprobe + ik];
            if (key < 0) continue;
            if ((size_t)key >= ix->nlist) {
                g_err = "Invalid key=" + std::to_string(key) + " nlist=" + std::to_string(ix->nlist);
                return -1;
            }
            const size_t b = ix->list_off[key], e = ix->list_off[key + 1];
            if (e == b) continue;
            nlistv++;
            ndis += e - b;
            for (size_t j = b; j < e; j++) {
                const float dis = ix->metric == ORC_METRIC_IP ? inner(xi, ix->codes + j * d, d) : l2sqr(xi, ix->codes + j * d, d);
                const bool in = ix->metric == ORC_METRIC_IP ? radius < dis : radius > dis;  // C::cmp(radius, dis)
                if (in) {
                    if (labels) {
                        labels[pos] = ix->ids[j];
                        distances[pos] = dis;
                    }
                    pos++;
                }
            }
        }
    }
    lims[n] = pos;
    if (stats && labels) {
        stats[0] += nlistv;
        stats[1] += ndis;
    }
    return 0;
}

int orc_search_preassigned(const orc_index_t* ix, size_t n, const float* x, size_t k, size_t nprobe,
                           const int64_t* keys, const float* coarse_dis, float* D, int64_t* I, int store_pairs,
                           size_t max_codes, orc_tuner_t* tuner, size_t offset, size_t* stats, int nthreads) {
    return search_driver(ix, n, x, k, nprobe, keys, coarse_dis, D, I, store_pairs != 0, max_codes, tuner, nullptr,
                         offset, stats, nthreads);
}

int orc_train_samples(const orc_index_t* ix, size_t n, const float* x, size_t max_topk, size_t nprobe,
                      const int64_t* keys, const float* coarse_dis, const float* interdis_cem,
                      const float* arcos_list, const float* gt_D, size_t offset, size_t train_num, float** raw_traces,
                      float* D, int64_t* I) {
    TrainCtx tc{interdis_cem, arcos_list, gt_D, train_num, raw_traces};
    return search_driver(ix, n, x, max_topk, nprobe, keys, coarse_dis, D, I, false, 0, nullptr, &tc, offset, nullptr, 1);
}

size_t orc_trace_sb(float* raw_xy, size_t n, size_t bs, float* out_x, float* out_y, float* out_std) {
    // same container, comparator and std::sort as the reference so that equal keys land in
    // the same order (IVF_pro.cpp:110-111)
    std::vector<std::pair<float, float>> trace(n);
    for (size_t i = 0; i < n; i++) trace[i] = std::make_pair(raw_xy[2 * i], raw_xy[2 * i + 1]);
    std::sort(trace.begin(), trace.end(),
              [](std::pair<float, float>& l, std::pair<float, float>& r) { return l.first > r.first; });
    size_t size = 0;
    for (auto& p : trace) size += (p.first < 0 && p.second < 0) ? 0 : 1;
    size_t sz = (size + bs - 1) / bs;
    for (size_t i = 0; i < sz; i++) {
        size_t left = i * bs, right = std::min((i + 1) * bs, size);
        float ave1 = 0, ave2 = 0;
        for (size_t index = left; index < right; index++) {
            size_t j = index - left;
            ave1 = (float)j / (float)(j + 1) * ave1 + trace[index].first / (j + 1);
            ave2 = (float)j / (float)(j + 1) * ave2 + trace[index].second / (j + 1);
        }
        double accum = 0.;
        for (size_t index = left; index < right; index++)
            accum += (trace[index].second - ave2) * (trace[index].second - ave2);
        float sd = std::sqrt(accum / bs);
        // reversed to ascending order at the end (IVF_pro.cpp:147-148)
        out_x[sz - 1 - i] = ave1;
        out_y[sz - 1 - i] = ave2;
        out_std[sz - 1 - i] = sd;
    }
    return sz;
}

void orc_merge_tables(int metric, size_t n, size_t k, size_t nshard, const float* all_D, const int64_t* all_I,
                      float* D, int64_t* I) {
    if (k == 0) return;
    if (metric == ORC_METRIC_L2) merge_tables<false>(n, k, nshard, all_D, all_I, D, I);
    else merge_tables<true>(n, k, nshard, all_D, all_I, D, I);
}

}  // extern "C"
