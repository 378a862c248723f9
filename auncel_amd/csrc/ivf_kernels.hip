// gfx950 (MI355X, CDNA4) kernels of the IVF-Flat search engine.
//
// Arithmetic contract: every query-to-vector distance is computed with exactly the rounding
// sequence of the reference's default (SSE) build of fvec_L2sqr / fvec_inner_product
// (Auncel/utils_simd.cpp:391-443): four running fp32 sums, sum l taking elements 4i+l in order,
// products and sums rounded separately (this file is built with -ffp-contract=off), final
// (s0+s1)+(s2+s3).  That is what makes ids reproducible bit for bit on float data.
//
// Selection contract: the reference keeps its k best in a binary heap that only admits strictly
// better candidates (Heap.h:88-142, IndexIVFFlat.cpp:125-135); which of several equal distances
// survives, and the order of equal distances in the output, depend on the heap's history.  The
// replay kernel therefore runs that very heap, sequentially, over the distance rows in probe
// order -- one wave per query, candidates pre-filtered 64 at a time with a ballot against the
// heap top.
#include "ivf_kernels.h"

#include <float.h>

namespace amdivf {

// =============================================================================================
// K-scan: distance tiles
// =============================================================================================
// Workgroup = 4 waves.  Vector tile staged through LDS in chunks of SCAN_DC dimensions (coalesced
// 16-B-per-lane global reads, 128-B row segments), each lane then owns SCAN_RV vectors and reads
// its rows with conflict-free ds_read_b128 (row stride 36 dwords).  The SCAN_RQ query operands of
// a wave are wave-uniform: they are fetched with scalar loads and used as SGPR operands, so a
// query value costs neither LDS bandwidth nor VGPRs.
constexpr int LDS_ROW = SCAN_DC + 4;                 // dwords per staged row (pad = one 16-B slot)
constexpr int TILE_MAX_VECS = 4 * SCAN_WAVE_VECS;    // 512

template <int METRIC>
__global__ __launch_bounds__(256) void scan_tiles_kernel(ScanArgs a) {
    __shared__ float lds[TILE_MAX_VECS * LDS_ROW];   // 73,728 B

    const ScanItem it = a.items[blockIdx.x];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = (int)it.qg;                      // 1, 2 or 4
    const int vg = 4 / qg;
    const int qgi = wave & (qg - 1);
    const int vgi = wave / qg;
    const int tile_vecs = vg * SCAN_WAVE_VECS;
    const int d = a.d;

    const float* qptr[SCAN_RQ];
#pragma unroll
    for (int r = 0; r < SCAN_RQ; r++) {
        uint32_t local = (uint32_t)(qgi * SCAN_RQ + r);
        uint32_t pi = it.pair_begin + (local < it.npair ? local : 0u);
        qptr[r] = a.queries + (size_t)a.pair_query[pi] * (size_t)d;
    }

    float acc[SCAN_RQ][SCAN_RV][4];
#pragma unroll
    for (int r = 0; r < SCAN_RQ; r++)
#pragma unroll
        for (int v = 0; v < SCAN_RV; v++)
#pragma unroll
            for (int l = 0; l < 4; l++) acc[r][v][l] = 0.f;

    const float* tile_base = a.codes + (size_t)it.vec_base * (size_t)d;

    for (int c0 = 0; c0 < d; c0 += SCAN_DC) {
        const int nslot = (d - c0 >= SCAN_DC ? SCAN_DC : d - c0) >> 2;
        __syncthreads();
        for (int idx = tid; idx < tile_vecs * (SCAN_DC / 4); idx += 256) {
            const int row = idx >> 3, slot = idx & 7;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < (int)it.nvec && slot < nslot)
                val = *reinterpret_cast<const float4*>(tile_base + (size_t)row * d + c0 + slot * 4);
            *reinterpret_cast<float4*>(&lds[row * LDS_ROW + slot * 4]) = val;
        }
        __syncthreads();
        const float* myrow = &lds[(vgi * SCAN_WAVE_VECS + lane) * LDS_ROW];
        for (int s = 0; s < nslot; s++) {
            float4 y[SCAN_RV];
#pragma unroll
            for (int v = 0; v < SCAN_RV; v++)
                y[v] = *reinterpret_cast<const float4*>(myrow + v * 64 * LDS_ROW + s * 4);
#pragma unroll
            for (int r = 0; r < SCAN_RQ; r++) {
                const float4 q = *reinterpret_cast<const float4*>(qptr[r] + c0 + s * 4);
#pragma unroll
                for (int v = 0; v < SCAN_RV; v++) {
                    if (METRIC == METRIC_L2) {
                        float t0 = y[v].x - q.x, t1 = y[v].y - q.y, t2 = y[v].z - q.z, t3 = y[v].w - q.w;
                        acc[r][v][0] += t0 * t0;
                        acc[r][v][1] += t1 * t1;
                        acc[r][v][2] += t2 * t2;
                        acc[r][v][3] += t3 * t3;
                    } else {
                        acc[r][v][0] += y[v].x * q.x;
                        acc[r][v][1] += y[v].y * q.y;
                        acc[r][v][2] += y[v].z * q.z;
                        acc[r][v][3] += y[v].w * q.w;
                    }
                }
            }
        }
    }

#pragma unroll
    for (int r = 0; r < SCAN_RQ; r++) {
        uint32_t local = (uint32_t)(qgi * SCAN_RQ + r);
        if (local < it.npair) {
            float* out = a.dist + a.pair_out[it.pair_begin + local] + it.vec_off;
#pragma unroll
            for (int v = 0; v < SCAN_RV; v++) {
                int lv = vgi * SCAN_WAVE_VECS + v * 64 + lane;
                if (lv < (int)it.nvec) out[lv] = (acc[r][v][0] + acc[r][v][1]) + (acc[r][v][2] + acc[r][v][3]);
            }
        }
    }
}

void launch_scan(const ScanArgs& a, size_t nitems, hipStream_t s) {
    if (nitems == 0) return;
    if (a.metric == METRIC_L2)
        hipLaunchKernelGGL(scan_tiles_kernel<METRIC_L2>, dim3((unsigned)nitems), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(scan_tiles_kernel<METRIC_IP>, dim3((unsigned)nitems), dim3(256), 0, s, a);
}

// =============================================================================================
// K-replay: ordered selection + Auncel stop rule + training samples
// =============================================================================================
// heap entries made by this kernel carry (REF_TAG | list << 32 | position); anything else in the
// id slot is a caller-supplied id (scanner API) or -1 (empty)
constexpr int64_t REF_TAG = 1ll << 62;

template <bool IsMax> __device__ __forceinline__ bool hcmp(float a, float b) { return IsMax ? a > b : a < b; }
template <bool IsMax> __device__ __forceinline__ float hneutral() { return IsMax ? FLT_MAX : -FLT_MAX; }

// Heap.h:88-118 -- executed redundantly by every lane of the wave (uniform control flow)
template <bool IsMax> __device__ inline void heap_pop(int k, float* val, int64_t* ref) {
    val--;
    ref--;
    const float v = val[k];
    int i = 1;
    for (;;) {
        const int i1 = i << 1, i2 = i1 + 1;
        if (i1 > k) break;
        const float c1 = val[i1];
        if (i2 == k + 1 || hcmp<IsMax>(c1, val[i2])) {
            if (hcmp<IsMax>(v, c1)) break;
            val[i] = c1;
            ref[i] = ref[i1];
            i = i1;
        } else {
            const float c2 = val[i2];
            if (hcmp<IsMax>(v, c2)) break;
            val[i] = c2;
            ref[i] = ref[i2];
            i = i2;
        }
    }
    val[i] = val[k];
    ref[i] = ref[k];
}

// Heap.h:125-142
template <bool IsMax> __device__ inline void heap_push(int k, float* val, int64_t* ref, float v, int64_t id) {
    val--;
    ref--;
    int i = k;
    while (i > 1) {
        const int f = i >> 1;
        const float fv = val[f];
        if (!hcmp<IsMax>(v, fv)) break;
        val[i] = fv;
        ref[i] = ref[f];
        i = f;
    }
    val[i] = v;
    ref[i] = id;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    for (int off = 32; off; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)x, off);
        x = x > o ? x : o;
    }
    return x;
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// error_pro::arcos (IVF_pro.cpp:179-184)
__device__ inline float arcos_lut(const float* lut, float x, uint32_t* err) {
    if (!(x <= 1.0 && x >= -1.0)) {
        *err = ERR_ARCOS_DOMAIN;
        return 0.f;
    }
    int index = (int)(x * 500.f / 2.f + 250.f);
    return lut[index];
}

// cosine_theorem (IVF_pro.cpp:41-51): pow(float,int) promotes to double
__device__ inline float cosine_theorem_dev(float a, float b, float c, uint32_t* err) {
    if (!(a <= b)) *err = ERR_COSINE_PRECOND;
    float temp = (float)((double)a * (double)a + (double)c * (double)c - (double)b * (double)b);
    temp = temp / (2 * c);
    return c / 2 - temp;
}

// error_pro::sum_angle (IVF_pro.cpp:162-177), n = 15
__device__ inline float sum_angle_dev(const float* lut, float kdis, const float* dtb, uint32_t start, uint32_t* err) {
    float sum = 0;
    for (uint32_t i = start; i < start + 15; i++) {
        const float b = dtb[i];
        if (b >= kdis) continue;
        sum += arcos_lut(lut, b / kdis, err);
    }
    return sum;
}

// Trace::search (IVF_pro.cpp:84-107)
__device__ inline float trace_search(const float* x, const float* y, const float* sd, uint32_t n, float k, float sc) {
    if (k <= x[0]) return y[0] + sc * sd[0];
    if (k >= x[n - 1]) {
        const float ampli = k / x[n - 1];
        return (y[n - 1] + sc * sd[n - 1]) * ampli;
    }
    unsigned long long high = n - 1, low = 0, middle = 0;
    while (low <= high) {
        middle = (low + high) / 2;
        if (x[middle] < k) low = middle + 1;
        else high = middle - 1;
    }
    if (x[low] > k) low--;
    return y[low] + sc * sd[low];
}

// error_pro::cur_num (IVF_pro.cpp:258-291)
__device__ inline uint32_t cur_num_dev(const TunerDev& t, const float* Ds, const float* dtb, uint32_t index,
                                       uint32_t* err) {
    const uint32_t o = t.trace_off[index], n = t.trace_off[index + 1] - o;
    const float *tx = t.trace_x + o, *ty = t.trace_y + o, *ts = t.trace_std + o;
    const uint32_t start = (1u << index) - 1;
    const unsigned long long query_k = t.query_topk;
    unsigned long long high = query_k - 1, low = 0, middle = 0;
    {
        const float g = trace_search(tx, ty, ts, n, sum_angle_dev(t.arcos, Ds[high], dtb, start, err), t.std_m);
        if ((double)((float)query_k * g) <= (double)query_k * 1.005) return (uint32_t)query_k;
    }
    while (low <= high) {
        middle = (low + high) / 2;
        if (middle <= 0) return 0;
        const float g = trace_search(tx, ty, ts, n, sum_angle_dev(t.arcos, Ds[middle], dtb, start, err), t.std_m);
        if ((float)(middle + 1) * g <= (float)query_k) low = middle + 1;
        else high = middle - 1;
    }
    return (uint32_t)(low + 1);
}

// kscaling (IVF_pro.cpp:72-82)
__device__ inline float kscaling_dev(float kdis, uint32_t in, const float* gt, uint32_t max_topk) {
    uint32_t index = 0;
    for (; index < max_topk; index++) {
        const float df = fabsf(gt[index] - kdis);
        if ((double)(df / kdis) < 1e-5 || (double)df < 1e-5) break;
    }
    if (index >= max_topk) return -1.f;
    return (float)(index + 1) / (float)(in + 1);
}

// error_pro::set_online (IVF_pro.cpp:196-238): lanes split the entries
__device__ inline void set_online_dev(int metric, uint32_t nlist, const float* cd, const int64_t* ci,
                                      const float* interdis, const float* lut, float* dtb, int lane, uint32_t* err) {
    const uint32_t max_num = nlist / 8 + 20;
    const unsigned long long cur = (unsigned long long)ci[0];
    const float a0 = metric == METRIC_IP ? arcos_lut(lut, cd[0], err) : cd[0];
    for (uint32_t k = lane; k < max_num - 1; k += 64) {
        const unsigned long long dst = (unsigned long long)ci[k + 1];
        const unsigned long long i = cur < dst ? cur : dst, j = cur < dst ? dst : cur;
        const float c = interdis[(2ull * nlist - 1 - i) * i / 2 + j - 1 - i];
        const float b = metric == METRIC_IP ? arcos_lut(lut, cd[k + 1], err) : cd[k + 1];
        dtb[k] = cosine_theorem_dev(a0, b, c, err);
    }
    if (lane == 0) dtb[max_num - 1] = 0.f;
    if (metric == METRIC_IP) {
        // the reference converts all max_num coarse values up front (IVF_pro.cpp:208-211)
        for (uint32_t k = lane; k < max_num; k += 64) (void)arcos_lut(lut, cd[k], err);
    }
}

// ascending sort of src[0..k) into dst by ranking (values only matter)
__device__ inline void rank_sort(const float* src, float* dst, int k, int lane) {
    for (int i = lane; i < k; i += 64) {
        const float x = src[i];
        int rank = 0;
        for (int j = 0; j < k; j++) {
            const float y = src[j];
            rank += (y < x) || (y == x && j < i);
        }
        dst[rank] = x;
    }
}

template <bool IsMax>
__global__ __launch_bounds__(256) void replay_kernel(ReplayArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t li = blockIdx.x * 4 + wave;   // position in this launch
    if (li >= a.nq) return;
    const uint32_t qi = a.qsel ? a.qsel[li] : li;  // query slot (state / output row)
    if (a.done[qi]) return;

    const int k = a.k;
    // per-wave LDS: ref[k] (8B) | val[k] | tmp[k] | srt[k]
    unsigned char* base = smem + (size_t)wave * ((size_t)k * 20 + 16);
    int64_t* href = reinterpret_cast<int64_t*>(base);
    float* hval = reinterpret_cast<float*>(base + (size_t)k * 8);
    float* tmp = hval + k;
    float* srt = tmp + k;

    for (int i = lane; i < k; i += 64) {
        hval[i] = a.heap_val[(size_t)qi * k + i];
        href[i] = a.heap_ref[(size_t)qi * k + i];
    }
    wave_sync();

    const unsigned long long id_q = a.id_offset + qi;
    const uint32_t nlist = a.nlist;
    const bool tune = a.tuner.enabled != 0, training = a.train.enabled != 0;
    const uint32_t max_num = nlist / 8 + 20;
    float* dtb = (tune || training) ? a.dtb + (size_t)qi * max_num : nullptr;
    uint32_t err = 0;

    uint32_t ik0 = a.stage[qi];
    const uint32_t cnt = a.seg_count[li];
    unsigned long long nscan = a.nscan[qi];
    float pre_val = a.pre_val ? a.pre_val[qi] : 0.f;
    uint32_t stoped = a.stoped ? a.stoped[qi] : 0u;
    unsigned long long st_nlist = 0, st_nheap = 0, st_ndis = 0;

    if ((tune || training) && ik0 == 0) {
        set_online_dev(a.metric, nlist, a.coarse_dis + (size_t)qi * a.coarse_stride,
                       a.coarse_keys + (size_t)qi * a.coarse_stride, tune ? a.tuner.interdis : a.train.interdis,
                       tune ? a.tuner.arcos : a.train.arcos, dtb, lane, &err);
        __threadfence_block();
        wave_sync();
    }

    uint32_t query_k = 0;
    float true_KD_K = 0.f;
    if (tune) {
        query_k = a.tuner.query_topk;
        if (a.tuner.gt_D) true_KD_K = a.tuner.gt_D[id_q * (unsigned long long)k + query_k - 1];
    }

    bool finished = false;
    uint32_t consumed = 0;
    for (uint32_t p = 0; p < cnt && !finished; p++) {
        const uint32_t ik = ik0 + p;
        consumed = p + 1;
        const int key = a.seg_list[(size_t)li * a.round_probes + p];
        if (key >= 0) {
            if ((uint32_t)key >= nlist) {
                err = ERR_INVALID_KEY;
                finished = true;
                break;
            }
            const uint32_t n = a.identity_ids ? a.nlist : (uint32_t)(a.list_off[key + 1] - a.list_off[key]);
            if (n > 0) {
                st_nlist++;
                const float* seg = a.dist + a.seg_off[(size_t)li * a.round_probes + p];
                const int64_t refbase = REF_TAG | ((int64_t)key << 32);
                for (uint32_t b0 = 0; b0 < n; b0 += 256) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t j = b0 + u * 64 + lane;
                        v[u] = j < n ? seg[j] : hneutral<IsMax>();
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        unsigned long long m = __ballot(hcmp<IsMax>(hval[0], v[u]));
                        while (m) {
                            const int l = __builtin_ctzll(m);
                            m &= m - 1;
                            const float val = __shfl(v[u], l);
                            if (hcmp<IsMax>(hval[0], val)) {
                                heap_pop<IsMax>(k, hval, href);
                                heap_push<IsMax>(k, hval, href, val, refbase | (int64_t)(b0 + u * 64 + l));
                                st_nheap++;
                            }
                        }
                    }
                }
                nscan += n;
                st_ndis += n;
            }
        }
        if (a.max_codes && nscan >= a.max_codes) {
            finished = true;
            break;
        }
        if (a.total_nprobe && ik + 1 >= a.total_nprobe) finished = true;  // end of the probe loop
        wave_sync();
        if (tune) {
            // IndexIVF.cpp:551-638
            const uint32_t stage = ik + 1;
            uint32_t ind = 0;
            const uint32_t tmp_stage = stage >= nlist / 8 ? nlist / 8 - 1 : stage;
            while (tmp_stage > (1u << ind)) ind++;
            if (!IsMax) {
                for (int i = lane; i < k; i += 64) tmp[i] = arcos_lut(a.tuner.arcos, hval[i], &err);
                wave_sync();
                rank_sort(tmp, srt, k, lane);
            } else {
                rank_sort(hval, srt, k, lane);
            }
            wave_sync();
            err = wave_max_u32(err);
            if (err) {
                finished = true;
                break;
            }
            const uint32_t pre_num = cur_num_dev(a.tuner, srt, dtb, ind, &err);
            float recall = (float)pre_num / (float)query_k;
            // heap extreme + ground-truth hit count
            float ext = IsMax ? -1.f : FLT_MAX;
            uint32_t hits = 0;
            for (int i = lane; i < k; i += 64) {
                const float s = hval[i];
                ext = IsMax ? fmaxf(ext, s) : fminf(ext, s);
                if (IsMax ? ((double)s <= (double)true_KD_K * 1.0005) : ((double)s >= (double)true_KD_K * 0.9995)) hits++;
            }
            for (int off = 32; off; off >>= 1) {
                const float o = __shfl_xor(ext, off);
                ext = IsMax ? fmaxf(ext, o) : fminf(ext, o);
                hits += __shfl_xor(hits, off);
            }
            const float max_val = ext;
            const float racc = a.tuner.require_acc[id_q];
            const unsigned long long stops = (unsigned long long)(racc * 12);
            if (stage > 1) {
                if (max_val == pre_val) stoped++;
                else stoped = 0;
                if (stoped >= stops) recall = 1;
            }
            pre_val = max_val;
            const float true_recall = (float)hits / (float)query_k;
            unsigned long long np = a.tuner.my_nprobe[id_q];
            bool np_changed = false;
            if (recall >= racc && np == 0) {
                np = (unsigned long long)((float)stage * a.tuner.multipler);
                np_changed = true;
                if (np >= nlist && lane == 0) a.tuner.t_recalls[id_q] = 1.f;
            }
            if (stage >= nlist / 8 && np == 0) {
                np = (unsigned long long)((float)stage * a.tuner.multipler);
                np_changed = true;
                if (np >= nlist && lane == 0) a.tuner.t_recalls[id_q] = 1.f;
            }
            if (np_changed && lane == 0) a.tuner.my_nprobe[id_q] = np;
            if (np != 0 && np <= stage) {
                if (a.tuner.profile && lane == 0) a.tuner.t_recalls[id_q] = true_recall;
                finished = true;
            }
            if (err) finished = true;
        }
        if (training && !finished) {
            // IndexIVF.cpp:640-673
            const uint32_t stage = ik + 1;
            if (stage > nlist / 8) {
                finished = true;
            } else if ((stage & (stage - 1)) == 0) {
                uint32_t ind = 0;
                while (stage != (1u << ind)) ind++;
                rank_sort(hval, srt, k, lane);
                wave_sync();
                const float* gt = a.train.gt_D + id_q * (unsigned long long)k;
                float* out = a.train.raw[ind] + 2ull * (id_q * (unsigned long long)(k / 4));
                uint32_t count = 0;
                for (int ij = 0; ij < k; ij++) {
                    const float dv = IsMax ? srt[ij] : srt[k - 1 - ij];
                    const float ks = kscaling_dev(dv, (uint32_t)ij, gt, (uint32_t)k);
                    if (ks < 0) break;
                    float tval = dv;
                    if (!IsMax) tval = arcos_lut(a.train.arcos, tval, &err);
                    const float sum_a = sum_angle_dev(a.train.arcos, tval, dtb, stage - 1, &err);
                    if (lane == 0) {
                        out[2 * count] = sum_a;
                        out[2 * count + 1] = ks;
                    }
                    count++;
                    if (count >= (uint32_t)(k / 4)) break;
                }
                if (err) finished = true;
            }
        }
    }
    err = wave_max_u32(err);

    if (lane == 0) {
        a.stage[qi] = ik0 + consumed;
        a.nscan[qi] = nscan;
        if (a.pre_val) a.pre_val[qi] = pre_val;
        if (a.stoped) a.stoped[qi] = stoped;
        if (st_nlist) atomicAdd(&a.stats[0], st_nlist);
        if (st_ndis) atomicAdd(&a.stats[1], st_ndis);
        if (st_nheap) atomicAdd(&a.stats[2], st_nheap);
        if (err) atomicMax(a.error, err);
    }

    wave_sync();
    if (finished || a.finalize_all || err) {
        if (a.raw_heap_out) {
            for (int i = lane; i < k; i += 64) {
                int64_t ref = href[i];
                if (ref >= 0 && (ref & REF_TAG)) {
                    ref &= ~REF_TAG;
                    if (!a.store_pairs) ref = a.ids[a.list_off[ref >> 32] + (uint64_t)(ref & 0xffffffffll)];
                }
                a.D[(size_t)qi * k + i] = hval[i];
                a.I[(size_t)qi * k + i] = ref;
            }
        } else {
            // heap_reorder (Heap.h:295-322)
            int ii = 0;
            for (int i = 0; i < k; i++) {
                const float v = hval[0];
                const int64_t id = href[0];
                heap_pop<IsMax>(k - i, hval, href);
                hval[k - ii - 1] = v;
                href[k - ii - 1] = id;
                if (id != -1) ii++;
            }
            wave_sync();
            // valid entries now sit in [k-ii, k): move to the front, pad the rest
            for (int i = lane; i < k; i += 64) {
                float v = hneutral<IsMax>();
                int64_t id = -1;
                if (i < ii) {
                    v = hval[k - ii + i];
                    int64_t ref = href[k - ii + i];
                    if (ref & REF_TAG) {
                        ref &= ~REF_TAG;
                        if (a.identity_ids) ref &= 0xffffffffll;
                        else if (!a.store_pairs) ref = a.ids[a.list_off[ref >> 32] + (uint64_t)(ref & 0xffffffffll)];
                    }
                    id = ref;
                }
                a.D[(size_t)qi * k + i] = v;
                a.I[(size_t)qi * k + i] = id;
            }
        }
        if (lane == 0) a.done[qi] = 1;
    } else {
        for (int i = lane; i < k; i += 64) {
            a.heap_val[(size_t)qi * k + i] = hval[i];
            a.heap_ref[(size_t)qi * k + i] = href[i];
        }
    }
}

void launch_replay(const ReplayArgs& a, hipStream_t s) {
    if (a.nq == 0) return;
    const size_t shmem = 4 * ((size_t)a.k * 20 + 16);
    const dim3 grid((a.nq + 3) / 4), block(256);
    if (a.metric == METRIC_L2) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(replay_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(replay_kernel<true>, grid, block, shmem, s, a);
    } else {
        hipFuncSetAttribute(reinterpret_cast<const void*>(replay_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(replay_kernel<false>, grid, block, shmem, s, a);
    }
}

// =============================================================================================
// coarse quantiser: full sort of one row of centroid distances per workgroup
// =============================================================================================
// key order: L2 ascending distance, IP descending; equal distances by ascending centroid number
// (the reference's heap order for exactly equal coarse distances depends on its history; rows
// with nprobe <= 128 go through the replay kernel instead, which reproduces it).
template <bool Ascending>
__global__ __launch_bounds__(256) void sort_rows_kernel(const float* dis, uint32_t nlist, uint32_t npow2, uint32_t nprobe,
                                                         float* out_dis, int64_t* out_keys) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* v = reinterpret_cast<float*>(smem);
    uint32_t* ix = reinterpret_cast<uint32_t*>(v + npow2);
    const uint32_t q = blockIdx.x;
    const float* row = dis + (size_t)q * nlist;
    for (uint32_t i = threadIdx.x; i < npow2; i += 256) {
        if (i < nlist) {
            v[i] = row[i];
            ix[i] = i;
        } else {
            v[i] = Ascending ? INFINITY : -INFINITY;
            ix[i] = 0xffffffffu;
        }
    }
    __syncthreads();
    for (uint32_t size = 2; size <= npow2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < npow2 / 2; t += 256) {
                const uint32_t lo = 2 * t - (t & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const float a = v[lo], b = v[hi];
                const uint32_t ia = ix[lo], ib = ix[hi];
                // "a before b" in the final order
                const bool a_first = Ascending ? (a < b || (a == b && ia < ib)) : (a > b || (a == b && ia < ib));
                if (a_first != up) {
                    v[lo] = b;
                    v[hi] = a;
                    ix[lo] = ib;
                    ix[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < nprobe; i += 256) {
        const bool ok = i < nlist;
        out_dis[(size_t)q * nprobe + i] = ok ? v[i] : (Ascending ? FLT_MAX : -FLT_MAX);
        out_keys[(size_t)q * nprobe + i] = ok ? (int64_t)ix[i] : -1;
    }
}

void launch_sort_rows(const float* dis, uint32_t nq, uint32_t nlist, uint32_t nprobe, int metric, float* out_dis,
                      int64_t* out_keys, hipStream_t s) {
    if (nq == 0) return;
    uint32_t npow2 = 2;
    while (npow2 < nlist) npow2 <<= 1;
    const size_t shmem = (size_t)npow2 * 8;
    if (metric == METRIC_L2) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(sort_rows_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(sort_rows_kernel<true>, dim3(nq), dim3(256), shmem, s, dis, nlist, npow2, nprobe, out_dis, out_keys);
    } else {
        hipFuncSetAttribute(reinterpret_cast<const void*>(sort_rows_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(sort_rows_kernel<false>, dim3(nq), dim3(256), shmem, s, dis, nlist, npow2, nprobe, out_dis, out_keys);
    }
}

// =============================================================================================
// pack the upper triangle of the centroid x centroid distance matrix (IVF_pro.cpp:21-39 layout)
// =============================================================================================
__global__ void pack_upper_kernel(const float* full, uint32_t nlist, float* out) {
    const uint32_t i = blockIdx.y;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nlist && j > i) out[(2ull * nlist - 1 - i) * i / 2 + j - 1 - i] = full[(size_t)i * nlist + j];
}

void launch_pack_upper(const float* full, uint32_t nlist, float* out, hipStream_t s) {
    hipLaunchKernelGGL(pack_upper_kernel, dim3((nlist + 255) / 256, nlist), dim3(256), 0, s, full, nlist, out);
}

}  // namespace amdivf

namespace amdivf {
__global__ void fill_f32_kernel(float* p, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void fill_i64_kernel(int64_t* p, size_t n, int64_t v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill_f32(float* p, size_t n, float v, hipStream_t s) {
    if (n) hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}
void launch_fill_i64(int64_t* p, size_t n, int64_t v, hipStream_t s) {
    if (n) hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}
}  // namespace amdivf
