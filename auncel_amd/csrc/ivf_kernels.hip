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
#include <algorithm>
#include <stdexcept>
#include <stdlib.h>
#include <string>
#include <type_traits>
#include <utility>

namespace amdivf {

// =============================================================================================
// K-scan: distance tiles
// =============================================================================================
// Workgroup = 4 waves.  Vector tile staged through LDS in chunks of SCAN_DC dimensions (coalesced
// 16-B-per-lane global reads, 128-B row segments), each lane then owns SCAN_RV vectors and reads
// its rows with conflict-free ds_read_b128 (row stride 36 dwords).  The SCAN_RQ query operands of
// a wave are wave-uniform: they are fetched with scalar loads and used as SGPR operands, so a
// query value costs neither LDS bandwidth nor VGPRs.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int LDS_ROW = SCAN_DC + 4;                 // dwords per staged row (pad = one 16-B slot)

// QG = query groups per workgroup (1, 2, 4 or 8): QG groups of 8 queries x VG blocks of 128 vectors, one wave
// each.  QG 8: 8 waves share one 128-vector tile (64 queries per pass over the list); QG 4: 4 waves, 128 vectors;
// QG 2 / 1: 4 waves over 256 / 512 vectors.  The staged tile (and the LDS footprint) follows.
//
// ARITH selects the arithmetic (the result is the same fp32 number either way):
//   0  the reference's SSE order: four running sums over elements 4i+l, separate multiply and add
//   1  the same order with fma, legal when operands are small integers (see IntRange in the engine)
// (uint8-valued data takes scan_mfma_kernel below instead: exact integer contraction on the i8 matrix cores)
template <int QG> struct ScanShape {
    static constexpr int NT = QG == 8 ? 512 : 256;
    static constexpr int vg = QG >= 4 ? 1 : 4 / QG;
    static constexpr int tile_vecs = vg * SCAN_WAVE_VECS;
    static constexpr int lds_floats = tile_vecs * LDS_ROW;
};

// one tile; every wave of the workgroup takes part in all of its barriers
template <int METRIC, int QG, int ARITH>
__device__ __forceinline__ void scan_tile_one(const ScanArgs& a, const ScanItem it, float (*lds)[ScanShape<QG>::lds_floats]) {
    constexpr bool FUSED = ARITH == 1;
    constexpr int qg = QG;
    constexpr int NT = ScanShape<QG>::NT;
    constexpr int vg = ScanShape<QG>::vg;
    constexpr int tile_vecs = ScanShape<QG>::tile_vecs;
    constexpr int SLOTS = SCAN_DC / 4;                     // 16-B slots per staged row
    constexpr int NLD = tile_vecs * SLOTS / NT;            // fetches per thread per chunk
    static_assert(tile_vecs * SLOTS % NT == 0, "tile must split evenly over the workgroup");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qgi = wave & (qg - 1);
    const int vgi = wave / qg;
    const int d = a.d;

    // the wave's 8 query operands of one 4-dimension step are 128 contiguous bytes (pack_queries): they come
    // in through the scalar cache with two wide scalar loads and stay in SGPRs
    // (constant address space: inside the item loop of the chained rounds the compiler sees the previous item's stores
    // between these loads and the kernel entry, may not assume plain global memory unchanged, and would fetch the
    // operands with per-lane vector loads -- eight 1-KB broadcasts per step through the CU's one texture path, which
    // bounded every shape of this kernel at 11 T element pairs/s; measured with scalar loads: 15-17 T.  Requesting them
    // half a step at a time, or across the chunk boundary, was measured too: 13 and 15 T -- the SGPR file is full
    // either way and the spill traffic decides)
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef const v4f __attribute__((address_space(4)))* const_f4p;
    const const_f4p qtile = (const_f4p)(uintptr_t)(a.qtile + (size_t)(it.qgroup + qgi) * (size_t)d * SCAN_RQ);
    auto qload = [](const_f4p p) {
        const v4f t = *p;
        return make_float4(t.x, t.y, t.z, t.w);
    };

    // running sums (s0, s1) and (s2, s3) of the reference's 4-lane accumulator, as two register pairs
    f2 acc[SCAN_RQ][SCAN_RV][2];
#pragma unroll
    for (int r = 0; r < SCAN_RQ; r++)
#pragma unroll
        for (int v = 0; v < SCAN_RV; v++) acc[r][v][0] = acc[r][v][1] = f2{0.f, 0.f};

    const float* tile_base = a.codes + (size_t)it.vec_base * (size_t)d;
    const bool has_queries = (uint32_t)(qgi * SCAN_RQ) < it.npair;

    // What the epilogue needs per query -- row offset of its distances, |x|^2, threshold -- is fetched now by lane r
    // for query r (and |y|^2 per vector by every lane), so that the loads ride under the tile's arithmetic instead of
    // forming two dependent scalar round trips per query at the end.
    const bool masked = a.thr != nullptr;
    unsigned long long e_row = 0;
    float e_thr = 0.f;
    if (has_queries) {
        const uint32_t local = (uint32_t)(qgi * SCAN_RQ + lane);
        if (lane < SCAN_RQ && local < it.npair) {
            e_row = a.pair_out[it.pair_begin + local] + it.vec_off;
            if (masked) e_thr = a.thr[a.pair_query[it.pair_begin + local]];
        }
    }

    // fetches run two chunks ahead of the compute (register staging, PF sets): at 4 steps per chunk one chunk
    // of lead does not cover an HBM round trip under load
    constexpr int PF = 2;
    float4 pre[PF][NLD];
    auto fetch = [&](int c0, float4 (&dst)[NLD]) {
        const int nslot = (d - c0 >= SCAN_DC ? SCAN_DC : d - c0) >> 2;
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = tid + i * NT;
            const int row = idx / SLOTS, slot = idx % SLOTS;
            dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 < d && row < (int)it.nvec && slot < nslot)
                dst[i] = *reinterpret_cast<const float4*>(tile_base + (size_t)row * d + c0 + slot * 4);
        }
    };

    fetch(0, pre[0]);
    fetch(SCAN_DC, pre[1]);
    int buf = 0;
    // the chunk loop is unrolled by PF so that the staging registers are indexed statically
    for (int c0 = 0; c0 < d; c0 += PF * SCAN_DC) {
#pragma unroll
      for (int ph = 0; ph < PF; ph++, buf ^= 1) {
        const int cc = c0 + ph * SCAN_DC;
        if (cc >= d) break;
        const int nslot = (d - cc >= SCAN_DC ? SCAN_DC : d - cc) >> 2;
        float* stage = lds[buf];
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = tid + i * NT;
            *reinterpret_cast<float4*>(&stage[(idx / SLOTS) * LDS_ROW + (idx % SLOTS) * 4]) = pre[ph][i];
        }
        __syncthreads();
        fetch(cc + PF * SCAN_DC, pre[ph]);
        if (!has_queries) continue;  // wave without queries: staging + barriers only
        const float* myrow = &stage[(vgi * SCAN_WAVE_VECS + lane) * LDS_ROW];
        // operands of step s+1 (queries: scalar loads, vectors: LDS rows) are requested before step s is computed
        const const_f4p qs = qtile + (size_t)(cc >> 2) * SCAN_RQ;
        float4 qn[SCAN_RQ], yn[SCAN_RV];
#pragma unroll
        for (int r = 0; r < SCAN_RQ; r++) qn[r] = qload(qs + r);
#pragma unroll
        for (int v = 0; v < SCAN_RV; v++) yn[v] = *reinterpret_cast<const float4*>(myrow + v * 64 * LDS_ROW);
#pragma unroll
        for (int s = 0; s < SLOTS; s++) {
            if (s >= nslot) break;  // partial last chunk (wave-uniform)
            float4 qc[SCAN_RQ];
            f2 ya[SCAN_RV], yb[SCAN_RV];  // elements (0,1) and (2,3) of the 4-wide step: sums 0,1 and 2,3
#pragma unroll
            for (int r = 0; r < SCAN_RQ; r++) qc[r] = qn[r];
#pragma unroll
            for (int v = 0; v < SCAN_RV; v++) {
                ya[v] = f2{yn[v].x, yn[v].y};
                yb[v] = f2{yn[v].z, yn[v].w};
            }
            if (s + 1 < nslot) {
#pragma unroll
                for (int v = 0; v < SCAN_RV; v++) yn[v] = *reinterpret_cast<const float4*>(myrow + v * 64 * LDS_ROW + (s + 1) * 4);
#pragma unroll
                for (int r = 0; r < SCAN_RQ; r++) qn[r] = qload(qs + (s + 1) * SCAN_RQ + r);
            }
#pragma unroll
            for (int r = 0; r < SCAN_RQ; r++) {
                const f2 qa = f2{qc[r].x, qc[r].y}, qb = f2{qc[r].z, qc[r].w};
#pragma unroll
                for (int v = 0; v < SCAN_RV; v++) {
                    if (METRIC == METRIC_L2) {
                        const f2 ta = ya[v] - qa, tb = yb[v] - qb;
                        if (FUSED) {  // products exactly representable: one rounding either way
                            acc[r][v][0] = __builtin_elementwise_fma(ta, ta, acc[r][v][0]);
                            acc[r][v][1] = __builtin_elementwise_fma(tb, tb, acc[r][v][1]);
                        } else {
                            acc[r][v][0] += ta * ta;
                            acc[r][v][1] += tb * tb;
                        }
                    } else if (FUSED) {
                        acc[r][v][0] = __builtin_elementwise_fma(ya[v], qa, acc[r][v][0]);
                        acc[r][v][1] = __builtin_elementwise_fma(yb[v], qb, acc[r][v][1]);
                    } else {
                        acc[r][v][0] += ya[v] * qa;
                        acc[r][v][1] += yb[v] * qb;
                    }
                }
            }
        }
      }
    }

    // With thresholds (a.thr: the heap top each query had when the round was planned) only the distances that can
    // still enter the heap are stored, and every 64-candidate chunk of a row gets a bit mask of those positions:
    // the replay kernel then reads 1 bit per candidate instead of 4 bytes, and the rest of the row never
    // leaves the chip.  Rows start on multiples of 64 floats in that mode.
    if (!has_queries) return;
    // threshold mode: the 64-bit masks of the wave's SCAN_RQ x SCAN_RV chunks are parked in lanes r * SCAN_RV + v and
    // stored by one instruction at the end (a predicated store per chunk costs more scalar work than the chunk's test)
    uint32_t mk_lo = 0, mk_hi = 0;
#pragma unroll
    for (int r = 0; r < SCAN_RQ; r++) {
        uint32_t local = (uint32_t)(qgi * SCAN_RQ + r);
        if (local < it.npair) {
            const unsigned long long row = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(e_row >> 32), r) << 32) |
                                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e_row, r);
            float* out = a.dist + row;
            const float thr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e_thr), r));
#pragma unroll
            for (int v = 0; v < SCAN_RV; v++) {
                const int lv0 = vgi * SCAN_WAVE_VECS + v * 64;
                const int lv = lv0 + lane;
                const float res = (acc[r][v][0].x + acc[r][v][0].y) + (acc[r][v][1].x + acc[r][v][1].y);
                bool keep = lv < (int)it.nvec;
                if (masked) {
                    keep = keep && (METRIC == METRIC_L2 ? thr > res : thr < res);
                    const unsigned long long m = __ballot(keep);
                    const bool mine = lane == r * SCAN_RV + v;
                    mk_lo = mine ? (uint32_t)m : mk_lo;
                    mk_hi = mine ? (uint32_t)(m >> 32) : mk_hi;
                }
                if (keep) out[lv] = res;
            }
        }
    }
    if (masked && lane < SCAN_RQ * SCAN_RV) {
        const int r = lane / SCAN_RV, v = lane % SCAN_RV;
        const int lv0 = vgi * SCAN_WAVE_VECS + v * 64;
        const unsigned long long row = __shfl(e_row, r);  // lane r holds the row offset of query r
        if ((uint32_t)(qgi * SCAN_RQ + r) < it.npair && lv0 < (int)it.nvec)
            a.mask[(row + lv0) >> 6] = ((unsigned long long)mk_hi << 32) | mk_lo;
    }
}

// Which items a workgroup walks.  Host-sized launches: one item per workgroup (the grid is the item count).  Chained rounds
// (a.dev_counts): a resident grid, every workgroup strides over the items of its shape, whose count the planning kernels
// left on the device.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2); consecutive items are the
// tiles of one list for one block of queries, so with xcd_chunks XCD x takes the x-th eighth of the item list and a block's
// tiles (and its packed query operands) stay in one L2.
struct ItemWalk {
    uint32_t cur, end, step;
    __device__ ItemWalk(uint32_t n, int xcd_chunks) {
        if (xcd_chunks) {
            const uint32_t per = (n + 7) >> 3, x = blockIdx.x & 7;
            cur = x * per + (blockIdx.x >> 3);
            end = (x + 1) * per < n ? (x + 1) * per : n;
            step = gridDim.x >> 3;
        } else {
            cur = blockIdx.x;
            end = n;
            step = gridDim.x;
        }
    }
};

template <int METRIC, int QG, int ARITH>
__global__ __launch_bounds__(QG == 8 ? 512 : 256) void scan_tiles_kernel(ScanArgs a) {
    // two staging buffers: chunk c+1 is fetched into registers while chunk c is being consumed, and written
    // to the other buffer, so a workgroup needs one barrier per chunk and hides its own fetch latency
    __shared__ float lds[2][ScanShape<QG>::lds_floats];
    uint32_t n = a.nitems;
    const ScanItem* items = a.items;
    if (a.dev_counts) {  // shapes are laid out qg 1 | 2 | 4 | 8 in the item array
        const uint32_t n1 = a.dev_counts[CNT_QG1], n2 = a.dev_counts[CNT_QG2], n4 = a.dev_counts[CNT_QG4], n8 = a.dev_counts[CNT_QG8];
        n = QG == 1 ? n1 : QG == 2 ? n2 : QG == 4 ? n4 : n8;
        items += QG == 1 ? 0 : QG == 2 ? n1 : QG == 4 ? n1 + n2 : n1 + n2 + n4;
    }
    ItemWalk w(n, a.xcd_chunks);
    for (uint32_t i = w.cur; i < w.end; i += w.step) {
        scan_tile_one<METRIC, QG, ARITH>(a, items[i], lds);
        __syncthreads();  // the staging buffers are reused by the next tile
    }
}

__global__ __launch_bounds__(256) void pack_queries_kernel(const float* queries, const uint32_t* pair_query, const uint32_t* group_p0,
                                                           const uint32_t* group_cnt, int d, float* qtile, uint32_t ngroups,
                                                           const uint32_t* dev_ngroups) {
    if (dev_ngroups) ngroups = *dev_ngroups;
    for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint32_t p0 = group_p0[g], cnt = group_cnt[g];
        float4* out = reinterpret_cast<float4*>(qtile + (size_t)g * d * SCAN_RQ);
        const int nstep = d >> 2;
        for (int idx = threadIdx.x; idx < nstep * SCAN_RQ; idx += 256) {
            const int step = idx / SCAN_RQ, r = idx % SCAN_RQ;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((uint32_t)r < cnt) v = *reinterpret_cast<const float4*>(queries + (size_t)pair_query[p0 + r] * d + step * 4);
            out[idx] = v;
        }
    }
}

// dev_ngroups: the group count is on the device (chained rounds); ngroups is then only the capacity of the buffers
void launch_pack_queries(const float* queries, const uint32_t* pair_query, const uint32_t* group_p0, const uint32_t* group_cnt,
                         size_t ngroups, int d, float* qtile, hipStream_t s, const uint32_t* dev_ngroups, uint32_t hint) {
    if (!ngroups && !dev_ngroups) return;
    const unsigned grid = dev_ngroups ? (hint ? hint + hint / 8 + 8 : resident_grid(16)) : (unsigned)ngroups;
    LAUNCH(pack_queries_kernel, dim3(grid), dim3(256), 0, s, queries, pair_query, group_p0, group_cnt, d, qtile, (uint32_t)ngroups, dev_ngroups);
}

unsigned resident_grid(unsigned per_cu) {
    static const unsigned cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return (unsigned)n;
    }();
    return ((cus * per_cu + 7) / 8) * 8;  // a multiple of the 8 XCDs
}

// workgroups of `kern` that fit a CU at once (resident grids: more would queue behind the others and start their share late)
template <class K> static unsigned blocks_per_cu(K kern, int threads, size_t dyn_lds = 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, dyn_lds) != hipSuccess || nb <= 0) nb = 1;
    return (unsigned)nb;
}

template <int QG> static void launch_scan_qg(ScanArgs a, size_t first, size_t n, hipStream_t s) {
    if (n == 0 && !a.dev_counts) return;
    if (!a.dev_counts) a.items += first;
    a.nitems = (uint32_t)n;
    // chained rounds: n is only a bound; a resident grid walks the device-side count
    const int threads = QG == 8 ? 512 : 256;
    const uint32_t hint = a.hint_qg[scan_qg_class(QG)];
    auto go = [&](auto kern) {
        static const unsigned per_cu = blocks_per_cu(kern, threads);
        const unsigned hinted = ((unsigned)((size_t)hint + hint / 8 + 7) / 8) * 8 + 8;  // last time's count + 12 %
        const dim3 grid((unsigned)(a.dev_counts ? (hint ? hinted : resident_grid(per_cu)) : a.xcd_chunks ? ((n + 7) / 8) * 8 : n)), block(threads);
        LAUNCH(kern, grid, block, 0, s, a);
    };
    if (a.metric == METRIC_L2) {
        if (a.fused) go(scan_tiles_kernel<METRIC_L2, QG, 1>);
        else go(scan_tiles_kernel<METRIC_L2, QG, 0>);
    } else {
        if (a.fused) go(scan_tiles_kernel<METRIC_IP, QG, 1>);
        else go(scan_tiles_kernel<METRIC_IP, QG, 0>);
    }
}

// items must be grouped by qg: n_qg[0] items with qg 1, then n_qg[1] with qg 2, n_qg[2] with qg 4, n_qg[3] with qg 8
void launch_scan(const ScanArgs& a, const size_t n_qg[4], hipStream_t s, hipStream_t s2, hipStream_t s1, hipStream_t s4) {
    launch_scan_qg<8>(a, n_qg[0] + n_qg[1] + n_qg[2], n_qg[3], s);
    launch_scan_qg<4>(a, n_qg[0] + n_qg[1], n_qg[2], s4 ? s4 : s);
    launch_scan_qg<2>(a, n_qg[0], n_qg[1], s2 ? s2 : s);
    launch_scan_qg<1>(a, 0, n_qg[0], s1 ? s1 : s);
}


// =============================================================================================
// K-scan, byte codes: distance tiles on the i8 matrix cores
// =============================================================================================
// One wave = one work item: up to 32 queries probing a list x a chunk of MFMA_CHUNK consecutive vectors of it.  The query
// bytes are the A operand (rows = queries), gathered once per item straight from the (L2-resident) signed query matrix; the
// list streams through as the B operand in pairs of 32-vector blocks, fetched in fragment order (ivf_kernels.h) with
// coalesced 16-byte-per-lane loads -- every byte of a list is read once per item from HBM, no LDS staging, no barriers;
// the waves of a CU in their load phase are what keeps the memory pipe full.  v_mfma_i32_32x32x32_i8 leaves query
// (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) x vector (lane & 31) in accumulator register `reg`, so a register of a block is two
// 128-byte runs of two distance rows.  Exact integers throughout: t = 2 x.y - |y|^2 (L2) or x.y + cy (IP) is compared with
// the query's threshold in the integer domain (a v_cmp per register is the 64-candidate ballot), res = cx -/+ t is converted
// once.  K order inside the contraction is irrelevant (integers), so both operands use "lane half h owns bytes
// [16 ks h, 16 ks (h + 1))", which makes a lane's ks pieces of a query row contiguous.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// B operand loads.  A chunk is read by one item per block of 32 queries probing its list; those items are neighbouring waves
// of one workgroup, so with the default cache policy the second reader finds the chunk in L2 (a streaming hint sends every
// reader to HBM: 3.3 GB instead of 1.3 GB per round on the bench workload).  AUNCEL_MFMA_NT=1 restores the hint.
#if defined(AUNCEL_MFMA_NT) && AUNCEL_MFMA_NT
#define MFMA_BLOAD(p) __builtin_nontemporal_load(p)
#else
#define MFMA_BLOAD(p) (*(p))
#endif

// reg[lane LANE] = val (val wave-uniform, LANE a compile-time constant: an inline operand, so the one SGPR slot is val's)
template <int LANE> __device__ __forceinline__ void writelane_c(int& reg, uint32_t val) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(reg) : "s"(val), "n"(LANE));
}
template <int... I, class F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// NKS = K-steps (d <= 32 NKS) with the query operand resident in registers; 0 = any d, query pieces re-read per block pair
template <int METRIC, bool MASKED, int NKS>
// (four waves per SIMD: with the register budget stated the compiler accumulates in VGPRs and needs no AGPR copies)
#ifndef AUNCEL_MFMA_WAVES
#define AUNCEL_MFMA_WAVES 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AUNCEL_MFMA_WAVES, 8))) void scan_mfma_kernel(MfmaScanArgs a) {
    __shared__ int s_cx[4][32];
    __shared__ int s_u[4][32];
    __shared__ uint32_t s_row[4][32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int ks = NKS ? NKS : (int)mfma_ksteps(a.d);
    const size_t qstride = (size_t)ks * 32;
    // items of this wave: workgroup w of the walk takes items 4 w .. 4 w + 3 (host-sized launch: one such group per workgroup;
    // chained rounds: resident workgroups stride over the device-side count).  No workgroup barrier anywhere below.
    const uint32_t nitems = a.dev_nitems ? *a.dev_nitems : a.nitems;
    ItemWalk w((nitems + 3) >> 2, a.xcd_chunks);
    for (uint32_t wi = w.cur; wi < w.end; wi += w.step) {
    const uint32_t item_no = wi * 4 + wave;
    if (item_no >= nitems) break;
    const ScanItem it = a.items[item_no];

    // ---- per-query operands, lane m (both halves) for query m of the item
    const bool qok = (uint32_t)m < it.npair;
    uint32_t qrow = 0;
    unsigned long long row = 0;
    int cx = 0, u = 0x7fffffff;
    if (qok) {
        qrow = a.pair_query[it.pair_begin + m];
        row = a.pair_out[it.pair_begin + m] + it.vec_off;
        cx = a.query_cx[qrow];
        if (MASKED) {
            // the reference keeps a candidate iff C::cmp(top, dis): L2 top > dis, IP top < dis.  dis is an integer in
            // [0, 2^24]; the threshold (a heap top, or a range-search radius) need not be: L2 dis < ceil(thr), IP dis > floor(thr).
            const float thr = a.thr[qrow];
            if (METRIC == METRIC_L2) {
                const float c = ceilf(thr);
                const int T = !(c > 0.f) ? 0 : (c >= 1073741824.f ? 1073741824 : (int)c);  // NaN -> nothing passes
                u = cx - T;          // keep <=> 2 x.y - |y|^2 > |x|^2 - T
            } else {
                const float f = floorf(thr);
                const int T = !(f < 1073741824.f) ? 0x7fffffff : (f <= -1073741824.f ? -1073741824 : (int)f);
                u = T == 0x7fffffff ? T : T - cx;  // keep <=> x.y + cy > T - cx
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous item's reads of this wave's LDS rows are done
    __builtin_amdgcn_wave_barrier();
    if (h == 0) {
        s_cx[wave][m] = cx;
        s_u[wave][m] = u;
        s_row[wave][m] = (uint32_t)row;
    }
    const int8_t* qp = a.queries8 + (size_t)qrow * qstride + (size_t)h * (size_t)(16 * ks);
    v4i af[NKS ? NKS : 1];
    if (NKS) {
#pragma unroll
        for (int s = 0; s < NKS; s++) af[s] = qok ? *reinterpret_cast<const v4i*>(qp + 16 * s) : v4i{0, 0, 0, 0};
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // thresholds / cx / row offsets in accumulator layout: register 4 g + i of lane half h belongs to query 8 g + 4 h + i.
    // Threshold mode keeps thresholds and rows (cx is read from LDS for the few values it stores); the dense mode cx and rows.
    int ur[16], cxr[16];
    uint32_t rowr[16];  // in floats: the distance buffer is at most 2^31 floats
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const v4i tu = *reinterpret_cast<const v4i*>(&s_u[wave][8 * g + 4 * h]);
        const v4i tc = *reinterpret_cast<const v4i*>(&s_cx[wave][8 * g + 4 * h]);
        const v4i tr = *reinterpret_cast<const v4i*>(&s_row[wave][8 * g + 4 * h]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ur[4 * g + i] = tu[i];
            cxr[4 * g + i] = tc[i];
            rowr[4 * g + i] = (uint32_t)tr[i];
        }
    }

    const size_t block_bytes = (size_t)ks * 1024;
    const uint32_t nblk = ((it.nvec + 63) >> 6) * 2;  // lists are stored in pairs of blocks: a 64-candidate mask word is two of ours
    uint32_t* mask32 = reinterpret_cast<uint32_t*>(a.mask);
    // What was measured on this loop in round 2 and did not move a launch (0.52 ms by HIP events): two and three blocks
    // requested ahead at three waves per SIMD (0.54 / 0.86), four at two (0.65), five waves per SIMD with the per-register
    // operands re-read from LDS (0.54), the next item's header and first block requested under the current item (0.52),
    // lane writes only for registers with a candidate (0.52), unconditional stores into per-wave trash lines so that the
    // wait at the top is a count instead of vmcnt(0) (0.41 vs 0.42 on a dense round).  Without any epilogue a dense
    // launch takes 0.25 ms (4.6 TB/s; torch's copy reaches 5.5 on this box): the distance stores and the second query
    // block of a list are what the rest pays for.
    // Software pipeline (resident-operand form): the B pieces of block i + 1 are requested right after the MFMAs of block i
    // have consumed their registers, and land while the epilogue of block i runs; the scheduling barriers keep the compiler
    // from sinking the loads back to their uses (it otherwise recycles three registers and keeps three loads in flight).
    v4i b[NKS ? NKS : 1];
    int cyn = 0;
    auto fetch_block = [&](uint32_t i) {
        const uint64_t blk = it.vec_base + i;
        const uint8_t* bp = a.codes_frag + blk * block_bytes + (size_t)lane * 16;
#pragma unroll
        for (int s = 0; s < (NKS ? NKS : 1); s++) b[s] = MFMA_BLOAD(reinterpret_cast<const v4i*>(bp + (size_t)s * 1024));
        cyn = a.code_cy[blk * 32 + m];
    };
    if (NKS) fetch_block(0);
    for (uint32_t i = 0; i < nblk; i++) {
        v16i acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0;
        int cy;
        if (NKS) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NKS; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s], b[s], acc, 0, 0, 0);
            cy = cyn;
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < nblk) fetch_block(i + 1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            const uint64_t blk = it.vec_base + i;
            const uint8_t* bp = a.codes_frag + blk * block_bytes + (size_t)lane * 16;
            cy = a.code_cy[blk * 32 + m];
            for (int s = 0; s < ks; s++) {
                const v4i aq = qok ? *reinterpret_cast<const v4i*>(qp + 16 * s) : v4i{0, 0, 0, 0};
                const v4i b0 = MFMA_BLOAD(reinterpret_cast<const v4i*>(bp + (size_t)s * 1024));
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq, b0, acc, 0, 0, 0);
            }
        }
        const uint32_t lv = i * 32 + m;           // position of this lane's vector in the chunk
        const bool vok = lv < it.nvec;
        const unsigned long long vmask = __ballot(vok);
        int word = 0;  // threshold mode: lane q (< 32) collects the 32-candidate mask word of query q
        static_for(std::make_integer_sequence<int, 16>{}, [&](auto R) {
            constexpr int reg = decltype(R)::value;
            constexpr int q0 = (reg & 3) + 8 * (reg >> 2);  // query of lane half 0; half 1: q0 + 4
            const int t = METRIC == METRIC_L2 ? 2 * acc[reg] - cy : acc[reg] + cy;
            if (MASKED) {
                const bool keep = t > ur[reg];
                const unsigned long long bal = __ballot(keep) & vmask;
                writelane_c<q0>(word, (uint32_t)bal);
                writelane_c<q0 + 4>(word, (uint32_t)(bal >> 32));
                if (bal && keep && vok) {  // rare after round 0: the query's cx comes from LDS only then
                    const int cq = s_cx[wave][q0 + 4 * h];
                    uint32_t off = rowr[reg] + lv;
                    asm volatile("" : "+v"(off));  // 64-bit addresses are formed here, not hoisted as 16 register pairs
                    a.dist[off] = (float)(METRIC == METRIC_L2 ? cq - t : cq + t);
                }
            } else {
                const int res = METRIC == METRIC_L2 ? cxr[reg] - t : cxr[reg] + t;
                uint32_t off = rowr[reg] + lv;
                asm volatile("" : "+v"(off));
                if (vok && (uint32_t)(q0 + 4 * h) < it.npair) a.dist[off] = (float)res;
            }
        });
        // rows start on multiples of 64 floats, chunks on multiples of 64 vectors: word index = (row + position) / 32.
        // The odd block of a 64-candidate chunk past the list's end gets its (zero) word too.
        if (MASKED && lane < 32 && qok) mask32[(row + i * 32) >> 5] = (uint32_t)word;
    }
    }  // items
}

uint32_t mfma_chunk() {
    static const uint32_t v = [] {
        const char* e = getenv("AUNCEL_AMD_MFMA_CHUNK");
        const long x = e ? atol(e) : 0;
        return x >= 64 ? (uint32_t)((x + 63) / 64 * 64) : 256u;
    }();
    return v;
}

void launch_scan_mfma(const MfmaScanArgs& a, hipStream_t s) {
    if (a.nitems == 0 && !a.dev_nitems) return;
    const unsigned nwg = (a.nitems + 3) / 4;
    static const unsigned per_cu_env = getenv("AUNCEL_AMD_MFMA_WG_PER_CU") ? (unsigned)atoi(getenv("AUNCEL_AMD_MFMA_WG_PER_CU")) : 0;
    const bool masked = a.thr != nullptr;
    const int ks = (int)mfma_ksteps(a.d);
    static const bool no_hint = getenv("AUNCEL_AMD_RESIDENT_GRIDS") != nullptr;
    auto go = [&](auto kern) {
        static const unsigned per_cu = blocks_per_cu(kern, 256);
        const size_t hwg = ((size_t)a.hint_nitems + a.hint_nitems / 8 + 3) / 4;  // last time's count + 12 %, four items per workgroup
        const unsigned hinted = (unsigned)((hwg + 7) / 8) * 8 + 8;
        const dim3 grid(a.dev_nitems ? (a.hint_nitems && !no_hint ? hinted : resident_grid(per_cu_env ? per_cu_env : per_cu))
                                     : a.xcd_chunks ? ((nwg + 7) / 8) * 8 : nwg),
                        block(256);
        LAUNCH(kern, grid, block, 0, s, a);
    };
    auto pick_ks = [&](auto metric, auto msk) {
        constexpr int M = decltype(metric)::value;
        constexpr bool K = decltype(msk)::value;
        switch (ks) {
            case 1: return go(scan_mfma_kernel<M, K, 1>);
            case 2: return go(scan_mfma_kernel<M, K, 2>);
            case 3: return go(scan_mfma_kernel<M, K, 3>);
            case 4: return go(scan_mfma_kernel<M, K, 4>);
            default: return go(scan_mfma_kernel<M, K, 0>);
        }
    };
    if (a.metric == METRIC_L2) {
        if (masked) pick_ks(std::integral_constant<int, METRIC_L2>{}, std::true_type{});
        else pick_ks(std::integral_constant<int, METRIC_L2>{}, std::false_type{});
    } else {
        if (masked) pick_ks(std::integral_constant<int, METRIC_IP>{}, std::true_type{});
        else pick_ks(std::integral_constant<int, METRIC_IP>{}, std::false_type{});
    }
}

// fp32 lists -> fragment order (one wave per 32-vector block; the block's list by bisection over block_off)
__global__ __launch_bounds__(64) void frag_from_f32_kernel(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist,
                                                           int d, int dpad, int metric, uint8_t* out, int32_t* cy) {
    const uint64_t blk = blockIdx.x;
    const int lane = threadIdx.x, v = lane & 31, h = lane >> 5;
    uint32_t lo = 0, hi = nlist;  // largest l with block_off[l] <= blk
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (block_off[mid] <= blk) lo = mid;
        else hi = mid;
    }
    const uint64_t pos = (blk - block_off[lo]) * 32 + v, size = list_off[lo + 1] - list_off[lo];
    const bool ok = pos < size;
    const float* src = codes + (list_off[lo] + (ok ? pos : 0)) * (uint64_t)dpad;
    const int ks = (int)mfma_ksteps(d);
    int sq = 0, sum = 0;
    for (int s = 0; s < ks; s++) {
        const int c0 = h * 16 * ks + 16 * s;
        v4i piece;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t word = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = c0 + 4 * w + b;
                int sv = 0;  // padding: signed zero
                if (ok && c < d) {
                    const int uv = (int)src[c];
                    sv = uv - 128;
                    sq += sv * sv;
                    sum += uv;
                }
                word |= (uint32_t)(sv & 0xff) << (8 * b);
            }
            piece[w] = (int)word;
        }
        *reinterpret_cast<v4i*>(out + blk * (uint64_t)ks * 1024 + (uint64_t)s * 1024 + (uint64_t)lane * 16) = piece;
    }
    sq += __shfl_xor(sq, 32);
    sum += __shfl_xor(sum, 32);
    if (h == 0) cy[blk * 32 + v] = !ok ? 0 : metric == METRIC_L2 ? sq : 128 * sum;
}

void launch_frag_from_f32(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist, uint64_t nblocks, int d,
                          int dpad, int metric, uint8_t* out, int32_t* cy, hipStream_t s) {
    if (nblocks == 0) return;
    LAUNCH(frag_from_f32_kernel, dim3((unsigned)nblocks), dim3(64), 0, s, codes, list_off, block_off, nlist, d, dpad, metric, out, cy);
}

// fp32 query rows -> signed byte rows (stride 32 ks, zero padded) + cx; one wave per row
__global__ __launch_bounds__(256) void sbytes_from_f32_kernel(const float* x, size_t n, int d, int dpad, int metric, int8_t* out, int32_t* cx) {
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const int stride = (int)mfma_ksteps(d) * 32;
    int sq = 0, sum = 0;
    for (int c = lane * 4; c < stride; c += 256) {
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int sv = 0;
            if (c + b < d) {
                const int uv = (int)x[row * (size_t)dpad + c + b];
                sv = uv - 128;
                sq += sv * sv;
                sum += uv;
            }
            word |= (uint32_t)(sv & 0xff) << (8 * b);
        }
        *reinterpret_cast<uint32_t*>(out + row * (size_t)stride + c) = word;
    }
    for (int off = 32; off; off >>= 1) {
        sq += __shfl_xor(sq, off);
        sum += __shfl_xor(sum, off);
    }
    if (lane == 0) cx[row] = metric == METRIC_L2 ? sq : 128 * sum - 16384 * d;
}

void launch_sbytes_from_f32(const float* x, size_t n, int d, int dpad, int metric, int8_t* out, int32_t* cx, hipStream_t s) {
    if (n == 0) return;
    LAUNCH(sbytes_from_f32_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, x, n, d, dpad, metric, out, cx);
}

// =============================================================================================
// K-replay: ordered selection + Auncel stop rule + training samples
// =============================================================================================
// heap entries made by this kernel carry (REF_TAG | list << 32 | position); anything else in the
// id slot is a caller-supplied id (scanner API) or -1 (empty)
constexpr int64_t REF_TAG = 1ll << 62;

template <bool IsMax> __device__ __forceinline__ bool hcmp(float a, float b) { return IsMax ? a > b : a < b; }
template <bool IsMax> __device__ __forceinline__ float hneutral() { return IsMax ? FLT_MAX : -FLT_MAX; }

// Heap.h:88-118 -- executed redundantly by every lane of the wave (uniform control flow).  Both children's
// value and id are requested together so that a level costs one LDS round trip.
template <bool IsMax> __device__ inline void heap_pop(int k, float* val, int64_t* ref) {
    val--;
    ref--;
    const float v = val[k];
    int i = 1;
    for (;;) {
        const int i1 = i << 1, i2 = i1 + 1;
        if (i1 > k) break;
        const int j2 = i2 <= k ? i2 : i1;  // i2 == k + 1: there is no right child
        const float c1 = val[i1], c2 = val[j2];
        const int64_t r1 = ref[i1], r2 = ref[j2];
        const bool left = (i2 == k + 1) || hcmp<IsMax>(c1, c2);
        const float c = left ? c1 : c2;
        if (hcmp<IsMax>(v, c)) break;
        val[i] = c;
        ref[i] = left ? r1 : r2;
        i = left ? i1 : i2;
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

// ---------------------------------------------------------------------------------------------
// The same heap, resident in registers (k <= 127): node i (1-based, Heap.h numbering) lives in lane
// i & 63 of register i >> 6, so levels 0-5 (nodes 1..63) are in register 0 and level 6 in register 1.
// A wave replays one query, so every index below is wave-uniform: nodes are read with v_readlane and written
// with v_writelane, and the sift loops run on the scalar unit.  For that the registers hold order keys, not
// floats: key(x) is an unsigned integer with key(a) < key(b) <=> a < b (gfx950 has no scalar float compare),
// and the float comes back bit for bit from the key.  Each node carries the slot (0..k-1) of its 64-bit id in
// an LDS table, so ids never move.  (Keys order -0.0 below +0.0 where floats call them equal; distances from
// the scan kernels are never -0.0.  NaN never enters: admission is tested on the floats.)
struct RegHeap {
    uint32_t v0, v1;  // keys
    uint32_t s0, s1;  // id slots
};

__device__ __forceinline__ uint32_t fkey(float x) {
    const uint32_t u = __float_as_uint(x);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t key) { return __uint_as_float((key & 0x80000000u) ? key ^ 0x80000000u : ~key); }
template <bool IsMax> __device__ __forceinline__ bool kcmp(uint32_t a, uint32_t b) { return IsMax ? a > b : a < b; }

__device__ __forceinline__ float rl_f(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ int rl_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ uint32_t rl_u(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }
// reg[lane l] = val (val and l wave-uniform).  The lane select goes through M0: v_writelane_b32 may name one SGPR.
__device__ __forceinline__ void wl_u(uint32_t& reg, uint32_t val, int l) {
    asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(val), "s"(l) : "m0");
}

// two registers, same lane: one M0 set-up
__device__ __forceinline__ void wl2_u(uint32_t& r0, uint32_t v0, uint32_t& r1, uint32_t v1, int l) {
    asm("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
        : "+v"(r0), "+v"(r1)
        : "s"(v0), "s"(v1), "s"(l)
        : "m0");
}

__device__ __forceinline__ uint32_t rh_key(const RegHeap& h, int node) {
    const uint32_t a = rl_u(h.v0, node & 63), b = rl_u(h.v1, node & 63);
    return node < 64 ? a : b;
}
__device__ __forceinline__ uint32_t rh_slot(const RegHeap& h, int node) {
    const uint32_t a = rl_u(h.s0, node & 63), b = rl_u(h.s1, node & 63);
    return node < 64 ? a : b;
}
// node <- (key, slot); the register is picked by one branch (in C++ the compiler copies both registers around it)
__device__ __forceinline__ void rh_set(RegHeap& h, int node, uint32_t key, uint32_t slot) {
    asm volatile(
        "s_cmp_gt_u32 %[n], 63\n\ts_cbranch_scc1 1f\n\t"
        "s_mov_b32 m0, %[n]\n\ts_nop 0\n\tv_writelane_b32 %[v0], %[k], m0\n\tv_writelane_b32 %[s0], %[s], m0\n\ts_branch 2f\n"
        "1:\n\ts_sub_u32 m0, %[n], 64\n\ts_nop 0\n\tv_writelane_b32 %[v1], %[k], m0\n\tv_writelane_b32 %[s1], %[s], m0\n"
        "2:\n\t"
        : [v0] "+v"(h.v0), [s0] "+v"(h.s0), [v1] "+v"(h.v1), [s1] "+v"(h.s1)
        : [n] "s"(node), [k] "s"(key), [s] "s"(slot)
        : "m0", "scc");
}

// Heap.h:88-118 (the node being removed, k, still takes part in the child comparisons, as there).
// KC != 0: k is the compile-time constant KC and the bounds tests of complete levels fold away.
template <bool IsMax, int KC> __device__ __forceinline__ void rh_pop(RegHeap& h, int krt) {
    const int k = KC ? KC : krt;
    const uint32_t v = rh_key(h, k);
    const uint32_t sv = rh_slot(h, k);
    int i = 1;
#pragma unroll
    for (int lvl = 0; lvl < 6; lvl++) {  // parent on level lvl (node < 64: register 0), children on level lvl + 1
        const bool absent = KC && (2 << lvl) > KC;      // the whole child level lies beyond k
        const bool full = KC && (4 << lvl) - 1 <= KC;   // every node of the child level exists
        if (absent) break;
        const int i1 = i << 1, i2 = i1 + 1;
        if (!full && i1 > k) break;
        const bool only_left = !full && i2 == k + 1;
        const int j2 = only_left ? i1 : i2;
        const uint32_t c1 = lvl < 5 ? rl_u(h.v0, i1) : rl_u(h.v1, i1 - 64);
        const uint32_t c2 = lvl < 5 ? rl_u(h.v0, j2) : rl_u(h.v1, j2 - 64);
        const bool left = only_left || kcmp<IsMax>(c1, c2);
        const uint32_t c = left ? c1 : c2;
        if (kcmp<IsMax>(v, c)) break;
        const int ci = left ? i1 : i2;
        const uint32_t cs = lvl < 5 ? rl_u(h.s0, ci) : rl_u(h.s1, ci - 64);
        wl2_u(h.v0, c, h.s0, cs, i);
        i = ci;
    }
    rh_set(h, i, v, sv);
}

// The same walk for k = 100, written out in assembly: the compiler's structured control flow spends five scalar
// instructions per level on exit flags; here a level is 9 scalar + 5 vector instructions and one branch.  Levels 0-4
// (children in register 0, all present), then level 5 (children 64..100 in register 1; node 50 has the left child only,
// which the equal-keys case of the selection handles: both reads name the same lane and the "right" pick is that lane).
#define RH_ASM_LEVEL(MAXOP, CMPOP)                                                                                      \
    "s_lshl_b32 %[a], %[i], 1\n\ts_or_b32 %[b], %[a], 1\n\tv_readlane_b32 %[k1], %[v0], %[a]\n\tv_readlane_b32 %[k2], %[v0], %[b]\n\t" \
    MAXOP " %[c], %[k1], %[k2]\n\t" CMPOP " %[v], %[c]\n\ts_cbranch_scc1 9f\n\t" CMPOP " %[k1], %[k2]\n\ts_cselect_b32 %[a], %[a], %[b]\n\t" \
    "v_readlane_b32 %[cs], %[s0], %[a]\n\ts_mov_b32 m0, %[i]\n\ts_nop 0\n\tv_writelane_b32 %[v0], %[c], m0\n\t"                       \
    "v_writelane_b32 %[s0], %[cs], m0\n\ts_mov_b32 %[i], %[a]\n\t"
#define RH_ASM_LAST(MAXOP, CMPOP)                                                                                       \
    "s_cmp_gt_u32 %[i], 50\n\ts_cbranch_scc1 9f\n\ts_lshl_b32 %[a], %[i], 1\n\ts_sub_u32 %[a], %[a], 64\n\ts_or_b32 %[b], %[a], 1\n\t"     \
    "s_cmp_eq_u32 %[i], 50\n\ts_cselect_b32 %[b], %[a], %[b]\n\tv_readlane_b32 %[k1], %[v1], %[a]\n\tv_readlane_b32 %[k2], %[v1], %[b]\n\t" \
    MAXOP " %[c], %[k1], %[k2]\n\t" CMPOP " %[v], %[c]\n\ts_cbranch_scc1 9f\n\t" CMPOP " %[k1], %[k2]\n\ts_cselect_b32 %[a], %[a], %[b]\n\t" \
    "v_readlane_b32 %[cs], %[s1], %[a]\n\ts_mov_b32 m0, %[i]\n\ts_nop 0\n\tv_writelane_b32 %[v0], %[c], m0\n\t"                       \
    "v_writelane_b32 %[s0], %[cs], m0\n\ts_add_u32 %[i], %[a], 64\n\t"                                                 \
    "9:\n\t"
template <bool IsMax> __device__ __forceinline__ void rh_pop_k100(RegHeap& h) {
    const uint32_t v = rl_u(h.v1, 100 - 64);
    const uint32_t sv = rl_u(h.s1, 100 - 64);
    int i = 1;
    uint32_t a, b, k1, k2, c, cs;
    if (IsMax) {
        asm volatile(RH_ASM_LEVEL("s_max_u32", "s_cmp_gt_u32") RH_ASM_LEVEL("s_max_u32", "s_cmp_gt_u32") RH_ASM_LEVEL("s_max_u32", "s_cmp_gt_u32")
                         RH_ASM_LEVEL("s_max_u32", "s_cmp_gt_u32") RH_ASM_LEVEL("s_max_u32", "s_cmp_gt_u32") RH_ASM_LAST("s_max_u32", "s_cmp_gt_u32")
                     : [i] "+s"(i), [v0] "+v"(h.v0), [s0] "+v"(h.s0), [a] "=&s"(a), [b] "=&s"(b), [k1] "=&s"(k1), [k2] "=&s"(k2),
                       [c] "=&s"(c), [cs] "=&s"(cs)
                     : [v] "s"(v), [v1] "v"(h.v1), [s1] "v"(h.s1)
                     : "m0", "scc");
    } else {
        asm volatile(RH_ASM_LEVEL("s_min_u32", "s_cmp_lt_u32") RH_ASM_LEVEL("s_min_u32", "s_cmp_lt_u32") RH_ASM_LEVEL("s_min_u32", "s_cmp_lt_u32")
                         RH_ASM_LEVEL("s_min_u32", "s_cmp_lt_u32") RH_ASM_LEVEL("s_min_u32", "s_cmp_lt_u32") RH_ASM_LAST("s_min_u32", "s_cmp_lt_u32")
                     : [i] "+s"(i), [v0] "+v"(h.v0), [s0] "+v"(h.s0), [a] "=&s"(a), [b] "=&s"(b), [k1] "=&s"(k1), [k2] "=&s"(k2),
                       [c] "=&s"(c), [cs] "=&s"(cs)
                     : [v] "s"(v), [v1] "v"(h.v1), [s1] "v"(h.s1)
                     : "m0", "scc");
    }
    rh_set(h, i, v, sv);
}
#undef RH_ASM_LEVEL
#undef RH_ASM_LAST

// Heap.h:125-142
template <bool IsMax, int KC> __device__ __forceinline__ void rh_push(RegHeap& h, int krt, uint32_t v, uint32_t sv) {
    int i = KC ? KC : krt;
    while (i > 1) {
        const int f = i >> 1;  // < 64
        const uint32_t fv = rl_u(h.v0, f);
        if (!kcmp<IsMax>(v, fv)) break;
        const uint32_t fs = rl_u(h.s0, f);
        rh_set(h, i, fv, fs);
        i = f;
    }
    rh_set(h, i, v, sv);
}

// Heap.h:125-142 for k = 100: the ancestors of node 100 are fixed (50, 25, 12, 6, 3, 1), a new value rarely climbs
// past the first
template <bool IsMax> __device__ __forceinline__ void rh_push_k100(RegHeap& h, uint32_t v, uint32_t sv) {
    int i = 100;
#define RH_PUSH_STEP(F)                         \
    {                                           \
        const uint32_t fv = rl_u(h.v0, F);      \
        if (!kcmp<IsMax>(v, fv)) goto done;     \
        const uint32_t fs = rl_u(h.s0, F);      \
        rh_set(h, i, fv, fs);                   \
        i = F;                                  \
    }
    RH_PUSH_STEP(50)
    RH_PUSH_STEP(25)
    RH_PUSH_STEP(12)
    RH_PUSH_STEP(6)
    RH_PUSH_STEP(3)
    RH_PUSH_STEP(1)
#undef RH_PUSH_STEP
done:
    rh_set(h, i, v, sv);
}

// LDS heap arrays (node order) -> registers; slot j holds the id of node j + 1
__device__ __forceinline__ void rh_load(RegHeap& h, const float* hval, int k, int lane) {
    h.v0 = (lane >= 1 && lane <= k) ? fkey(hval[lane - 1]) : 0u;
    h.s0 = (uint32_t)(lane - 1);
    h.v1 = (lane + 64 <= k) ? fkey(hval[lane + 63]) : 0u;
    h.s1 = (uint32_t)(lane + 63);
}

// registers -> LDS heap arrays in node order (ids permuted through registers)
__device__ __forceinline__ void rh_store(const RegHeap& h, float* hval, int64_t* href, int k, int lane, bool with_refs) {
    const bool n0 = lane >= 1 && lane <= k, n1 = lane + 64 <= k;
    int64_t r0 = 0, r1 = 0;
    if (with_refs) {
        if (n0) r0 = href[h.s0];
        if (n1) r1 = href[h.s1];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (n0) hval[lane - 1] = fkey_inv(h.v0);
    if (n1) hval[lane + 63] = fkey_inv(h.v1);
    if (with_refs) {
        if (n0) href[lane - 1] = r0;
        if (n1) href[lane + 63] = r1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    for (int off = 32; off; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)x, off);
        x = x > o ? x : o;
    }
    return x;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
    for (int off = 32; off; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
    return x;
}

// largest error code raised by any lane (0 in the common case: one ballot, no shuffles)
__device__ __forceinline__ uint32_t wave_err(uint32_t err) { return __ballot(err != 0) ? wave_max_u32(err) : 0u; }

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

// Trace::search (IVF_pro.cpp:84-107); z[i] = y[i] + std_m * sd[i] is formed once per cached trace with the
// reference's own expression, so every return value is the same fp32 number
__device__ inline float trace_search(const float* x, const float* z, uint32_t n, float k) {
    if (k <= x[0]) return z[0];
    if (k >= x[n - 1]) {
        const float ampli = k / x[n - 1];
        return z[n - 1] * ampli;
    }
    unsigned long long high = n - 1, low = 0, middle = 0;
    while (low <= high) {
        middle = (low + high) / 2;
        if (x[middle] < k) low = middle + 1;
        else high = middle - 1;
    }
    if (x[low] > k) low--;
    return z[low];
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

// one wave per query: disToBoundary rows for a batch of queries (run once, before the first round)
__global__ __launch_bounds__(256) void set_online_kernel(int metric, uint32_t nlist, uint32_t nq, const float* coarse_dis,
                                                         const int64_t* coarse_keys, uint32_t coarse_stride, const float* interdis,
                                                         const float* arcos, float* dtb, uint32_t* error) {
    __shared__ float lut[500];
    for (int i = threadIdx.x; i < 500; i += 256) lut[i] = arcos[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    uint32_t err = 0;
    set_online_dev(metric, nlist, coarse_dis + (size_t)qi * coarse_stride, coarse_keys + (size_t)qi * coarse_stride, interdis, lut,
                   dtb + (size_t)qi * (nlist / 8 + 20), lane, &err);
    err = wave_max_u32(err);
    if (err && lane == 0) atomicMax(error, err);
}

void launch_set_online(int metric, uint32_t nlist, uint32_t nq, const float* coarse_dis, const int64_t* coarse_keys,
                       uint32_t coarse_stride, const float* interdis, const float* arcos, float* dtb, uint32_t* error, hipStream_t s) {
    if (nq) LAUNCH(set_online_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, metric, nlist, nq, coarse_dis, coarse_keys,
                               coarse_stride, interdis, arcos, dtb, error);
}

// best-first sort of the heap values into srt by ranking (values only matter)
template <bool IsMax> __device__ inline void rank_sort_best_first(const float* src, float* dst, int k, int lane) {
    for (int i = lane; i < k; i += 64) {
        const float x = src[i];
        int rank = 0;
        for (int j = 0; j < k; j++) {
            const float y = src[j];
            rank += (IsMax ? (y < x) : (y > x)) || (y == x && j < i);
        }
        dst[rank] = x;
    }
}

// srt holds the k heap values best first.  A heap update replaces the worst value (the heap top,
// == srt[k-1]) by `val`: shift the worse ones down by one slot and drop val into the gap.
template <bool IsMax> __device__ inline int sorted_replace_worst(float* srt, int k, float val, int lane) {
    int pos = 0;
    for (int c = (k - 1) / 64; c >= 0; c--) {
        const int idx = c * 64 + lane;
        const bool in = idx < k - 1;
        const float s = in ? srt[idx] : 0.f;
        const bool worse = in && (IsMax ? s > val : s < val);
        pos += __builtin_popcountll(__ballot(in && !worse));
        wave_sync();
        if (worse) srt[idx + 1] = s;
        wave_sync();
    }
    srt[pos] = val;
    wave_sync();
    return pos;  // where val landed in the best-first order
}

// error_pro::sum_angle (IVF_pro.cpp:162-177), n = 15: the 15 terms on 15 lanes, then summed in the
// reference's order (a skipped term adds +0, which leaves the non-negative running sum unchanged)
__device__ inline float sum_angle_par(const float* lut, float kdis, const float* dwin, int lane, uint32_t* err) {
    float t = 0.f;
    if (lane < 15) {
        const float b = dwin[lane];  // the 15 boundary distances of this stage (window of disToBoundary)
        if (!(b >= kdis)) t = arcos_lut(lut, b / kdis, err);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 15; i++) sum += __shfl(t, i);
    return sum;
}

struct TraceLds {
    const float *x, *z;
    uint32_t n;
};

// error_pro::cur_num (IVF_pro.cpp:258-291); Ds(m) = m-th best value (IP: its arcos)
template <bool IsMax>
__device__ inline uint32_t cur_num_lds(const TraceLds& tr, const float* lut, const float* srt, const float* dwin,
                                       uint32_t query_topk, int lane, uint32_t* err) {
    const unsigned long long query_k = query_topk;
    unsigned long long high = query_k - 1, low = 0, middle = 0;
    auto Ds = [&](unsigned long long m) { return IsMax ? srt[m] : arcos_lut(lut, srt[m], err); };
    {
        const float g = trace_search(tr.x, tr.z, tr.n, sum_angle_par(lut, Ds(high), dwin, lane, err));
        if ((double)((float)query_k * g) <= (double)query_k * 1.005) return (uint32_t)query_k;
    }
    while (low <= high) {
        middle = (low + high) / 2;
        if (middle <= 0) return 0;
        const float g = trace_search(tr.x, tr.z, tr.n, sum_angle_par(lut, Ds(middle), dwin, lane, err));
        if ((float)(middle + 1) * g <= (float)query_k) low = middle + 1;
        else high = middle - 1;
    }
    return (uint32_t)(low + 1);
}

// The same function with every probe value of the binary search evaluated at once: term (m, i) of
// sum_angle(Ds(m)) on lane m * 15 + i (three passes for query_topk = 10), then lane m adds its 15 terms in the
// reference's order and runs its own Trace::search; the search itself is replayed on the scalar unit over the
// resulting predicate bits.  Each S(Ds(m)) is formed by the same fp32 operations in the same order as above, and
// an acos-domain error only counts if the reference's search would have visited that m.
constexpr uint32_t CURNUM_PAR_MAXK = 10;  // terms[] holds CURNUM_PAR_MAXK * 15 floats per wave
template <bool IsMax>
__device__ inline uint32_t cur_num_par(const TraceLds& tr, const float* lut, const float* srt, const float* dwin, float* terms,
                                       uint32_t query_topk, int lane, uint32_t* err) {
    const int nterm = (int)query_topk * 15;
    unsigned long long errm = 0;  // bit m: evaluating S(Ds(m)) left the acos domain
    for (int base = 0; base < nterm; base += 64) {
        const int idx = base + lane;
        uint32_t e = 0;
        if (idx < nterm) {
            const int m = idx / 15, i = idx - m * 15;
            uint32_t e0 = 0;
            const float kd = IsMax ? srt[m] : arcos_lut(lut, srt[m], &e0);  // IP: the caller has range-checked every srt[]
            const float b = dwin[i];
            float t = 0.f;
            if (!(b >= kd)) t = arcos_lut(lut, b / kd, &e);
            terms[idx] = t;
        }
        unsigned long long eb = __ballot(e != 0);
        while (eb) {
            const int l = __builtin_ctzll(eb);
            eb &= eb - 1;
            errm |= 1ull << ((base + l) / 15);
        }
    }
    wave_sync();
    float g = 0.f;
    if ((uint32_t)lane < query_topk) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 15; i++) sum += terms[lane * 15 + i];
        g = trace_search(tr.x, tr.z, tr.n, sum);
    }
    const unsigned long long query_k = query_topk;
    const bool first_ok = (double)((float)query_k * g) <= (double)query_k * 1.005;
    const bool step_ok = (float)(lane + 1) * g <= (float)query_k;
    const unsigned long long mfirst = __ballot((uint32_t)lane == query_topk - 1 && first_ok);
    const unsigned long long mstep = __ballot((uint32_t)lane < query_topk && step_ok);
    wave_sync();
    unsigned long long high = query_k - 1, low = 0, middle = 0;
    if ((errm >> high) & 1) {
        *err = ERR_ARCOS_DOMAIN;
        return 0;
    }
    if (mfirst) return (uint32_t)query_k;
    while (low <= high) {
        middle = (low + high) / 2;
        if (middle <= 0) return 0;
        if ((errm >> middle) & 1) {
            *err = ERR_ARCOS_DOMAIN;
            return 0;
        }
        if ((mstep >> middle) & 1) low = middle + 1;
        else high = middle - 1;
    }
    return (uint32_t)(low + 1);
}

__host__ __device__ inline size_t replay_wave_bytes(int k, uint32_t nlist, bool geo, bool tune, bool train, uint32_t trace_cap) {
    (void)nlist;
    size_t b = (size_t)k * 16;                       // href | hval | srt
    if (geo) b += 16 * 4 + 16 * 4;                   // window of disToBoundary | values inserted during the current probe
    if (tune) b += (size_t)trace_cap * 8;            // cached trace (x | z)
    if (tune) b += CURNUM_PAR_MAXK * 15 * 4 + 8;     // sum_angle terms of cur_num_par
    if (train) b += (size_t)k * 4;                   // ground-truth row
    return (b + 15) & ~(size_t)15;
}

// RH: the heap lives in registers (k <= 127); otherwise in LDS
// NLD: 64-candidate chunks per trip of the candidate stream (registers for two trips are live)
// KC: compile-time k of the register heap (0: run-time k)
template <bool IsMax, bool RH, int NLD, int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NLD == 16 ? 5 : 3))) void replay_kernel(ReplayArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool tune = a.tuner.enabled != 0, training = a.train.enabled != 0, geo = tune || training;
    const int k = KC ? KC : a.k;
    const uint32_t nlist = a.nlist;
    const uint32_t max_num = nlist / 8 + 20;

    // shared: the acos LUT; per wave: href | hval | srt | dtb | trace cache | gt row
    float* lut = reinterpret_cast<float*>(smem);
    if (geo) {
        const float* g = tune ? a.tuner.arcos : a.train.arcos;
        for (int i = threadIdx.x; i < 500; i += 256) lut[i] = g[i];
        __syncthreads();
    }
    const uint32_t li = blockIdx.x * 4 + wave;   // position in this launch
    if (li >= (a.nq_dev ? *a.nq_dev : a.nq)) return;
    const uint32_t qi = a.qsel ? a.qsel[li] : li;  // query slot (state / output row)
    if (a.done[qi]) return;

    unsigned char* base = smem + (geo ? 2000 : 0) + (size_t)wave * replay_wave_bytes(k, nlist, geo, tune, training, a.trace_cap);
    int64_t* href = reinterpret_cast<int64_t*>(base);
    float* hval = reinterpret_cast<float*>(base + (size_t)k * 8);
    float* srt = hval + k;
    float* dwin = srt + k;                                 // geo only: 16 boundary distances of the current stage
    float* pend = dwin + (geo ? 16 : 0);                   // geo only: values inserted during the current probe
    float* trc = pend + (geo ? 16 : 0);                    // tune only: x | z, trace_cap each
    float* terms = trc + (tune ? 2 * a.trace_cap : 0);     // tune only: cur_num_par scratch
    float* gtrow = terms + (tune ? CURNUM_PAR_MAXK * 15 + 2 : 0);  // training only
    const float* gdtb = geo ? a.dtb + (size_t)qi * max_num : nullptr;  // disToBoundary (set_online_kernel)

    for (int i = lane; i < k; i += 64) {
        hval[i] = a.heap_val[(size_t)qi * k + i];
        href[i] = a.heap_ref[(size_t)qi * k + i];
    }
    wave_sync();

    const unsigned long long id_q = a.id_offset + qi;
    const unsigned long long dbg_t0 = a.dbg ? __builtin_readcyclecounter() : 0;
    unsigned long long dbg_evals = 0, dbg_stream = 0, dbg_rule = 0, dbg_chunks = 0, dbg_probes = 0;
    uint32_t err = 0;
    uint32_t ik0 = a.stage[qi];
    const uint32_t loop_end = a.limit ? a.limit[qi] : a.total_nprobe;
    const uint32_t si = a.seg_by_slot ? qi : li;
    const uint32_t cnt = a.seg_count[si];
    const size_t seg0 = a.seg_begin ? (size_t)a.seg_begin[si] : (size_t)li * a.round_probes;
    unsigned long long nscan = a.nscan[qi];
    float pre_val = a.pre_val ? a.pre_val[qi] : 0.f;
    uint32_t stoped = a.stoped ? a.stoped[qi] : 0u;
    unsigned long long st_nlist = 0, st_nheap = 0, st_ndis = 0;

    constexpr bool asm_off = false;  // true: the C++ walk for k = 100 too
    RegHeap rh{};
    if (RH) rh_load(rh, hval, k, lane);

    int win_start = -1;
    if (geo) {
        rank_sort_best_first<IsMax>(hval, srt, k, lane);
        if (training) {
            const float* gt = a.train.gt_D + id_q * (unsigned long long)k;
            for (int i = lane; i < k; i += 64) gtrow[i] = gt[i];
        }
        wave_sync();
    }

    uint32_t query_k = 0;
    float true_KD_K = 0.f, racc = 0.f;
    unsigned long long np = 0;
    int cached_ind = -1;
    // cur_num is a pure function of the trace / window of `ind` and of the query_k best heap values: its value is kept
    // until one of them changes (in the later rounds most probes leave the best values alone)
    bool have_pre = false, top_changed = true, srt_changed = true;
    uint32_t kept_pre = 0;
    TraceLds tr{trc, trc + a.trace_cap, 0};
    if (tune) {
        query_k = a.tuner.query_topk;
        if (a.tuner.gt_D) true_KD_K = a.tuner.gt_D[id_q * (unsigned long long)k + query_k - 1];
        racc = a.tuner.require_acc[id_q];
        np = a.tuner.my_nprobe[id_q];
    }
    const unsigned long long np_in = np;

    // The candidate stream (this round's distance rows, in probe order) comes in trips of NLD x 64 values.
    // The loads of trip t + 1 are issued before trip t is examined, across probe boundaries: a wave owns one
    // query, so its own loads in flight are all that hides the HBM round trip.
    constexpr uint32_t TRIP = NLD * 64;
    // probe table of the current window of 64 probes, one probe per lane: list number, candidates, row offset
    uint32_t win0 = 0;
    int m_key = -1;
    uint32_t m_n = 0;
    unsigned long long m_off = 0;
    auto load_window = [&](uint32_t w0) {
        win0 = w0;
        m_key = -1;
        m_n = 0;
        m_off = 0;
        const uint32_t pi = w0 + lane;
        if (pi < cnt) {
            m_key = a.seg_list[seg0 + pi];
            m_off = a.seg_off[seg0 + pi];
            if (m_key >= 0 && (uint32_t)m_key < nlist)
                m_n = a.identity_ids ? nlist : (uint32_t)(a.list_off[m_key + 1] - a.list_off[m_key]);
        }
    };
    load_window(0);
    uint32_t fp = 0, fb = 0;  // fetch cursor: next (probe, offset); it never leaves the consumer's window
    auto fetch = [&](float (&dst)[NLD]) {
        for (;;) {
            if (fp >= cnt || fp >= win0 + 64) return;
            const uint32_t fn = (uint32_t)rl_i((int)m_n, (int)(fp - win0));
            if (fb < fn) {
                const unsigned long long fo = ((unsigned long long)(uint32_t)rl_i((int)(m_off >> 32), (int)(fp - win0)) << 32) |
                                              (uint32_t)rl_i((int)(uint32_t)m_off, (int)(fp - win0));
                const float* fseg = a.dist + fo;
#pragma unroll
                for (int u = 0; u < NLD; u++) {
                    const uint32_t j = fb + u * 64 + lane;
                    dst[u] = j < fn ? __builtin_nontemporal_load(fseg + j) : hneutral<IsMax>();
                }
                fb += TRIP;
                if (fb >= fn) {
                    fp++;
                    fb = 0;
                }
                return;
            }
            fp++;
            fb = 0;
        }
    };
    // Masked rounds (a.mask: one bit per candidate, written by the scan kernel for the values that beat the heap
    // top the query had when the round was planned): a row costs one 8-byte load per 64 candidates, and only the
    // chunks with a bit set are fetched.  Rows start on multiples of 64 floats there.
    const bool masked = a.mask != nullptr;
    float v[NLD], nv[NLD];
#pragma unroll
    for (int u = 0; u < NLD; u++) v[u] = nv[u] = hneutral<IsMax>();
    if (!masked) fetch(v);
    auto row_offset = [&](uint32_t p) {
        return ((unsigned long long)rl_u((uint32_t)(m_off >> 32), (int)(p - win0)) << 32) | rl_u((uint32_t)m_off, (int)(p - win0));
    };
    // mask words of chunks 0..63 and 64..127 of probe pre_p (8192 candidates: all but the very longest lists), requested one
    // probe ahead: the row of a probe then costs no memory round trip of its own
    unsigned long long mw_pre = 0, mw_pre2 = 0;
    uint32_t pre_p = 0xffffffffu;
    auto prefetch_masks = [&](uint32_t p) {
        pre_p = 0xffffffffu;
        if (p >= cnt || p >= win0 + 64) return;
        const uint32_t pn = rl_u(m_n, (int)(p - win0));
        if (pn == 0) return;
        const unsigned long long* mr = a.mask + (row_offset(p) >> 6);
        const uint32_t pch = (pn + 63) >> 6;
        mw_pre = (uint32_t)lane < pch ? mr[lane] : 0ull;
        mw_pre2 = (uint32_t)lane + 64 < pch ? mr[lane + 64] : 0ull;
        pre_p = p;
    };

    bool finished = false;
    uint32_t consumed = 0;
    for (uint32_t p = 0; p < cnt && !finished; p++) {
        const uint32_t ik = ik0 + p;
        consumed = p + 1;
        if (p >= win0 + 64) {  // next window of the probe table; the stream restarts behind it
            load_window(p);
            fp = p;
            fb = 0;
            if (!masked) fetch(v);
        }
        const int key = rl_i(m_key, (int)(p - win0));
        if (key >= 0) {
            if ((uint32_t)key >= nlist) {
                err = ERR_INVALID_KEY;
                finished = true;
                break;
            }
            const uint32_t n = rl_u(m_n, (int)(p - win0));
            if (n > 0) {
                st_nlist++;
                const unsigned long long dbg_s0 = a.dbg ? __builtin_readcyclecounter() : 0;
                const int64_t refbase = REF_TAG | ((int64_t)key << 32);
                uint32_t npend = 0;
                const uint32_t nchunk = (n + 63) >> 6;
                const unsigned long long roff = row_offset(p);
                const float* seg = a.dist + roff;
                const unsigned long long* mrow = masked ? a.mask + (roff >> 6) : nullptr;
                unsigned long long mw = 0, nz = 0, mw2 = 0;
                bool have2 = false;
                uint32_t b0 = 0, w0 = 0;
                if (masked) {
                    if (pre_p == p) {  // the first two windows of this row were requested while the previous row ran
                        mw = mw_pre;
                        mw2 = mw_pre2;
                        have2 = true;
                        nz = __ballot(mw != 0);
                        w0 = 64;
                    }
                    prefetch_masks(p + 1);
                }
                for (;;) {
                    unsigned long long bm = 0;  // masked: lanes of `mw` (chunks w0 - 64 + lane) now held in v[0..)
                    if (masked) {
                        while (nz == 0 && w0 < nchunk) {
                            if (w0 == 64 && have2) mw = mw2;
                            else mw = w0 + lane < nchunk ? mrow[w0 + lane] : 0ull;
                            nz = __ballot(mw != 0);
                            w0 += 64;
                        }
                        if (nz == 0) break;
                        if (a.dbg) dbg_chunks += __builtin_popcountll(nz);
#pragma unroll
                        for (int t = 0; t < NLD; t++) {
                            v[t] = hneutral<IsMax>();
                            if (nz) {
                                const int c = __builtin_ctzll(nz);
                                nz &= nz - 1;
                                bm |= 1ull << c;
                                const unsigned long long bits =
                                    ((unsigned long long)rl_u((uint32_t)(mw >> 32), c) << 32) | rl_u((uint32_t)mw, c);
                                if ((bits >> lane) & 1) v[t] = __builtin_nontemporal_load(seg + (size_t)(w0 - 64 + c) * 64 + lane);
                            }
                        }
                    } else {
                        if (b0 >= n) break;
                        fetch(nv);
                    }
                    float top = RH ? fkey_inv(rl_u(rh.v0, 1)) : hval[0];  // heap top, kept in a register between admissions
                    // chunks (64 candidates) holding at least one value that beats the top as it is now
                    uint32_t umask = 0;
#pragma unroll
                    for (int u = 0; u < NLD; u++) umask |= __ballot(hcmp<IsMax>(top, v[u])) ? (1u << u) : 0u;
                    // one copy of the update code for all chunks (32 inlined copies do not fit the instruction cache)
                    while (umask) {
                        const int u = __builtin_ctz(umask);
                        umask &= umask - 1;
                        float x = v[0];
#pragma unroll
                        for (int t = 1; t < NLD; t++) x = t == u ? v[t] : x;
                        uint32_t cbase = b0 + u * 64;  // position of the chunk's first candidate in its list
                        if (masked) {
                            unsigned long long mm = bm;
                            for (int i = 0; i < u; i++) mm &= mm - 1;
                            cbase = (w0 - 64 + (uint32_t)__builtin_ctzll(mm)) * 64;
                        }
                        unsigned long long m = __ballot(hcmp<IsMax>(top, x));
                        while (m) {
                            const int l = __builtin_ctzll(m);
                            m &= m - 1;
                            const float val = rl_f(x, l);
                            if (hcmp<IsMax>(top, val)) {
                                const int64_t nref = refbase | (int64_t)(cbase + l);
                                if (RH) {
                                    const uint32_t sr = rl_u(rh.s0, 1);  // the evicted root's id slot passes to the new entry
                                    if (lane == 0) href[sr] = nref;
                                    if (KC == 100 && !asm_off) rh_pop_k100<IsMax>(rh);
                                    else rh_pop<IsMax, KC>(rh, k);
                                    if (KC == 100 && !asm_off) rh_push_k100<IsMax>(rh, fkey(val), sr);
                                    else rh_push<IsMax, KC>(rh, k, fkey(val), sr);
                                    top = fkey_inv(rl_u(rh.v0, 1));
                                } else {
                                    heap_pop<IsMax>(k, hval, href);
                                    heap_push<IsMax>(k, hval, href, val, nref);
                                    top = hval[0];
                                }
                                st_nheap++;
                                if (geo) {  // the sorted view is only read at the end of the probe: defer
                                    if (npend < 16) pend[npend] = val;
                                    npend++;
                                }
                            }
                        }
                    }
                    if (!masked) {
#pragma unroll
                        for (int u = 0; u < NLD; u++) v[u] = nv[u];
                        b0 += TRIP;
                    }
                }
                if (a.dbg) dbg_stream += __builtin_readcyclecounter() - dbg_s0;
                if (geo && npend) {
                    wave_sync();
                    srt_changed = true;
                    if (npend <= 16) {
                        for (uint32_t u = 0; u < npend; u++)
                            if (sorted_replace_worst<IsMax>(srt, k, pend[u], lane) < (int)query_k) top_changed = true;
                    } else {
                        if (RH) rh_store(rh, hval, href, k, lane, false);
                        rank_sort_best_first<IsMax>(hval, srt, k, lane);
                        wave_sync();
                        top_changed = true;
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
        if (loop_end && ik + 1 >= loop_end) finished = true;  // end of the probe loop
        wave_sync();
        const unsigned long long dbg_r0 = a.dbg ? __builtin_readcyclecounter() : 0;
        if (tune) {
            // IndexIVF.cpp:551-638.  Once my_nprobe is known nothing the rule computes can change the
            // outcome any more (L2: no throwing path left), so only the stop test remains.
            const uint32_t stage = ik + 1;
            const bool overhead = a.tuner.overhead != 0;  // IndexIVF.cpp:614,634-637
            const bool fired = IsMax && np != 0 && !overhead;
            if (!fired) {
                uint32_t ind = 0;
                const uint32_t tmp_stage = stage >= nlist / 8 ? nlist / 8 - 1 : stage;
                while (tmp_stage > (1u << ind)) ind++;
                if ((int)ind != cached_ind) {
                    const uint32_t o = a.tuner.trace_off[ind], n = a.tuner.trace_off[ind + 1] - o;
                    const float sc = a.tuner.std_m;
                    for (uint32_t i = lane; i < n; i += 64) {
                        trc[i] = a.tuner.trace_x[o + i];
                        trc[a.trace_cap + i] = a.tuner.trace_y[o + i] + sc * a.tuner.trace_std[o + i];
                    }
                    if (lane < 15) dwin[lane] = gdtb[(1u << ind) - 1 + lane];  // sum_angle start = 2^ind - 1
                    tr.n = n;
                    cached_ind = (int)ind;
                    have_pre = false;
                    wave_sync();
                }
                if (!IsMax && srt_changed) {
                    // the reference converts all k heap values (IndexIVF.cpp:562-564): any out-of-domain one throws
                    for (int i = lane; i < k; i += 64) (void)arcos_lut(lut, srt[i], &err);
                    err = wave_err(err);
                    if (err) {
                        finished = true;
                        break;
                    }
                }
                srt_changed = false;
                if (!have_pre || top_changed) {
                    dbg_evals++;
                    kept_pre = query_k <= CURNUM_PAR_MAXK ? cur_num_par<IsMax>(tr, lut, srt, dwin, terms, query_k, lane, &err)
                                                          : cur_num_lds<IsMax>(tr, lut, srt, dwin, query_k, lane, &err);
                    have_pre = true;
                    top_changed = false;
                }
                const uint32_t pre_num = kept_pre;
                float recall = (float)pre_num / (float)query_k;
                const float max_val = IsMax ? fmaxf(-1.f, srt[k - 1]) : fminf(FLT_MAX, srt[k - 1]);
                const unsigned long long stops = (unsigned long long)(racc * 12);
                if (stage > 1) {
                    if (max_val == pre_val) stoped++;
                    else stoped = 0;
                    if (stoped >= stops) recall = 1;
                }
                pre_val = max_val;
                if (!overhead) {
                    if (recall >= racc && np == 0) {
                        np = (unsigned long long)((float)stage * a.tuner.multipler);
                        if (np >= nlist && lane == 0) a.tuner.t_recalls[id_q] = 1.f;
                    }
                    if (stage >= nlist / 8 && np == 0) {
                        np = (unsigned long long)((float)stage * a.tuner.multipler);
                        if (np >= nlist && lane == 0) a.tuner.t_recalls[id_q] = 1.f;
                    }
                }
                err = wave_err(err);
                if (err) finished = true;
            }
            if (overhead) {
                if (stage >= nlist / 8) finished = true;
            } else if (np != 0 && np <= stage) {
                if (a.tuner.profile) {
                    if (RH) rh_store(rh, hval, href, k, lane, false);
                    uint32_t hits = 0;
                    for (int i = lane; i < k; i += 64) {
                        const float s = hval[i];
                        if (IsMax ? ((double)s <= (double)true_KD_K * 1.0005) : ((double)s >= (double)true_KD_K * 0.9995)) hits++;
                    }
                    for (int off = 32; off; off >>= 1) hits += __shfl_xor(hits, off);
                    if (lane == 0) a.tuner.t_recalls[id_q] = (float)hits / (float)query_k;
                }
                finished = true;
            }
        }
        if (a.dbg) dbg_rule += __builtin_readcyclecounter() - dbg_r0;
        if (training && !finished) {
            // IndexIVF.cpp:640-673
            const uint32_t stage = ik + 1;
            if (stage > nlist / 8) {
                finished = true;
            } else if ((stage & (stage - 1)) == 0) {
                uint32_t ind = 0;
                while (stage != (1u << ind)) ind++;
                float* out = a.train.raw[ind] + 2ull * (id_q * (unsigned long long)(k / 4));
                if (win_start != (int)(stage - 1)) {
                    wave_sync();
                    if (lane < 15) dwin[lane] = gdtb[stage - 1 + lane];
                    win_start = (int)(stage - 1);
                    wave_sync();
                }
                uint32_t count = 0;
                for (int ij = 0; ij < k; ij++) {
                    const float dv = srt[ij];  // L2 ascending / IP descending, as the reference walks them
                    const float ks = kscaling_dev(dv, (uint32_t)ij, gtrow, (uint32_t)k);
                    if (ks < 0) break;
                    float tval = dv;
                    if (!IsMax) tval = arcos_lut(lut, tval, &err);
                    const float sum_a = sum_angle_par(lut, tval, dwin, lane, &err);
                    if (lane == 0) {
                        out[2 * count] = sum_a;
                        out[2 * count + 1] = ks;
                    }
                    count++;
                    if (count >= (uint32_t)(k / 4)) break;
                }
                err = wave_err(err);
                if (err) finished = true;
            }
        }
    }
    err = wave_err(err);

    if (lane == 0) {
        a.stage[qi] = ik0 + consumed;
        a.nscan[qi] = nscan;
        if (a.pre_val) a.pre_val[qi] = pre_val;
        if (a.stoped) a.stoped[qi] = stoped;
        if (tune && np != np_in) a.tuner.my_nprobe[id_q] = np;
        if (st_nlist) atomicAdd(&a.stats[0], st_nlist);
        if (st_ndis) atomicAdd(&a.stats[1], st_ndis);
        if (st_nheap) atomicAdd(&a.stats[2], st_nheap);
        if (err) atomicMax(a.error, err);
        if (a.dbg) {
            a.dbg[(size_t)li * 8 + 0] = __builtin_readcyclecounter() - dbg_t0;
            a.dbg[(size_t)li * 8 + 1] = st_nheap;
            a.dbg[(size_t)li * 8 + 2] = st_ndis;
            a.dbg[(size_t)li * 8 + 3] = dbg_evals;
            a.dbg[(size_t)li * 8 + 4] = dbg_stream;
            a.dbg[(size_t)li * 8 + 5] = dbg_rule;
            a.dbg[(size_t)li * 8 + 6] = dbg_chunks;
            a.dbg[(size_t)li * 8 + 7] = consumed;
        }
    }

    wave_sync();
    if (a.thr && lane == 0) a.thr[qi] = RH ? fkey_inv(rl_u(rh.v0, 1)) : hval[0];  // next round's scan stores only what beats this
    if (RH) rh_store(rh, hval, href, k, lane, true);  // back to the node-ordered LDS layout
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
    if (a.nq == 0) return;  // (chained rounds: nq is the bound the grid is sized by)
    const bool tune = a.tuner.enabled != 0, train = a.train.enabled != 0, geo = tune || train;
    const size_t shmem = (geo ? 2000 : 0) + 4 * replay_wave_bytes(a.k, a.nlist, geo, tune, train, a.trace_cap);
    const dim3 grid((a.nq + 3) / 4), block(256);
    static const bool no_rh = getenv("AUNCEL_AMD_LDS_HEAP") != nullptr;
    const bool rh = a.k <= 127 && !no_rh;
    // the heap (LDS form) and its sorted view take 16 k bytes per query, four queries per workgroup, of the CU's 160 KiB
    if (shmem > 160 * 1024)
        throw std::runtime_error("k = " + std::to_string(a.k) + " is beyond the selection kernel's LDS heap (" + std::to_string(shmem) +
                                 " bytes of 163840 per workgroup)");
    auto go = [&](auto kern) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) throw std::runtime_error(std::string("selection kernel: cannot reserve LDS: ") + hipGetErrorString(e));
        LAUNCH(kern, grid, block, shmem, s, a);
    };
    // few queries: longer trips (more loads in flight per wave) at the price of fewer resident waves
    const char* nld_s = getenv("AUNCEL_AMD_REPLAY_NLD");  // read per launch: the tests run both variants in one process
    const int nld_env = nld_s ? atoi(nld_s) : 0;
    const bool wide = nld_env ? nld_env >= 32 : (a.nq_hint ? a.nq_hint : a.nq) <= 3072;
    auto pick = [&](auto is_max) {
        constexpr bool M = decltype(is_max)::value;
        if (!rh) return wide ? go(replay_kernel<M, false, 32, 0>) : go(replay_kernel<M, false, 16, 0>);
        if (a.k == 100) return wide ? go(replay_kernel<M, true, 32, 100>) : go(replay_kernel<M, true, 16, 100>);
        if (a.k == 10) return wide ? go(replay_kernel<M, true, 32, 10>) : go(replay_kernel<M, true, 16, 10>);
        return wide ? go(replay_kernel<M, true, 32, 0>) : go(replay_kernel<M, true, 16, 0>);
    };
    if (a.metric == METRIC_L2) pick(std::true_type{});
    else pick(std::false_type{});
}

// =============================================================================================
// Selection in two kernels: compact_kernel (K1) + replay_lanes_kernel (K2)
// =============================================================================================
// replay_kernel spends one wave per query and ~140 scalar + vector instructions per heap update on a serial walk; a round 0
// of the bench workload is ~520 updates per query and the CU's issue ports are the bound.  Here the two halves of that
// work are separated:
//   K1, one wave per query, reads the round's distance rows once and keeps, in stream order, only the candidates that can
//      still enter the heap: those better than a threshold T that is always >= the heap top the reference has at that
//      point.  T starts as the heap top the round begins with and is refreshed every so often (after 128, 256, 512 ...
//      candidates) to the k-th best of a pool of values seen so far -- any k values seen bound the top from above, the
//      best k seen give it exactly.  A dense round 0 of 29 000 candidates leaves ~900, a threshold-mode round ~100.
//   K2, one query per LANE, replays the reference's heap_pop / heap_push (Heap.h:88-142) over those short lists with
//      the heap in LDS ([node][lane]: conflict-free), re-testing each candidate against the current top, and evaluates
//      the stop rule after every probe.  64 queries advance per instruction; a launch is ~80 waves that occupy a
//      fraction of the chip and overlap with other contexts' scans.
// Same state arrays in, same state and results out as replay_kernel: the two can take turns between rounds.
constexpr int POOL_CAP = 1024;             // K1: values per query between two threshold refreshes
constexpr int LANES_MAXK = 200;            // K2: the heap of 64 queries (8 bytes per node and query) + staging must fit 160 KiB
constexpr int LANES_BEST = 10;             // K2: best values kept sorted per query (query_topk <= this in tune mode)
constexpr bool LANES_DEFAULT = false;      // AUNCEL_AMD_LANES=1 / 0 overrides

template <bool IsMax> __device__ __forceinline__ uint32_t okey(float x) {  // smaller key <=> better candidate
    const uint32_t kx = fkey(x);
    return IsMax ? kx : ~kx;
}
template <bool IsMax> __device__ __forceinline__ float okey_inv(uint32_t key) { return fkey_inv(IsMax ? key : ~key); }

template <bool IsMax>
__global__ __launch_bounds__(256) void compact_kernel(ReplayArgs a) {
    __shared__ float s_pool[4][POOL_CAP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t li = blockIdx.x * 4 + wave;
    if (li >= (a.nq_dev ? *a.nq_dev : a.nq)) return;
    const uint32_t qi = a.qsel ? a.qsel[li] : li;
    if (a.done[qi]) return;
    const int k = a.k;
    const uint32_t nlist = a.nlist;
    const uint32_t cnt = a.seg_count[qi];
    const size_t seg0 = a.seg_begin[qi];
    float* pool = s_pool[wave];
    const size_t base = (size_t)qi * a.capq;
    const bool masked = a.mask != nullptr;
    float T = a.heap_val[(size_t)qi * k];  // the root: worst value kept
    uint32_t npool = 0;
    if (!masked) {
        for (int i = lane; i < k; i += 64) pool[i] = a.heap_val[(size_t)qi * k + i];
        npool = (uint32_t)k;
    }
    wave_sync();
    const unsigned long long lt_mask = (1ull << lane) - 1;

    // T <- (about) the k-th best of the pool; the pool keeps what is at least that good.  Bisection on the order keys for the
    // smallest key X with #(key <= X) >= k, left early once the count is within k / 8 of k: any X with at least k pool values
    // at or below it is a valid bound.  Always leaves npool <= POOL_CAP - 64.
    auto refresh = [&]() {
        constexpr int R = POOL_CAP / 64;
        uint32_t key[R];
        wave_sync();  // the pool as the other lanes left it
#pragma unroll
        for (int r = 0; r < R; r++) key[r] = (uint32_t)(r * 64 + lane) < npool ? okey<IsMax>(pool[r * 64 + lane]) : 0xffffffffu;
        auto count_le = [&](uint32_t x) {
            uint32_t c = 0;
#pragma unroll
            for (int r = 0; r < R; r++)
                if ((uint32_t)(r * 64) < npool) c += __builtin_popcountll(__ballot(key[r] <= x && (uint32_t)(r * 64 + lane) < npool));
            return c;
        };
        uint32_t lo = 0, hi = okey<IsMax>(T);
        const uint32_t want = (uint32_t)k, slack = (uint32_t)k / 8;
        uint32_t chi = count_le(hi);
        if (chi >= want) {  // (always: T is a bound already; kept as a guard)
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                const uint32_t c = count_le(mid);
                if (c >= want) {
                    hi = mid;
                    chi = c;
                    if (c <= want + slack) break;
                } else {
                    lo = mid + 1;
                }
            }
            T = okey_inv<IsMax>(hi);
        }
        // keep the values at or below the bound (at least k of them); if equal values still overfill the pool, any k of them do
        wave_sync();
        uint32_t out = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if ((uint32_t)(r * 64) >= npool) break;
            const bool keep = key[r] <= hi && (uint32_t)(r * 64 + lane) < npool;
            const unsigned long long m = __ballot(keep);
            const uint32_t room = (uint32_t)(POOL_CAP - 128) > out ? (uint32_t)(POOL_CAP - 128) - out : 0u;
            const uint32_t rank = __builtin_popcountll(m & lt_mask);
            if (keep && rank < room) pool[out + rank] = okey_inv<IsMax>(key[r]);
            const uint32_t c = __builtin_popcountll(m);
            out += c < room ? c : room;
        }
        npool = out;
        wave_sync();
    };

    uint32_t cursor = 0, since = 0, next_refresh = 128, probes_done = cnt;
    for (uint32_t p = 0; p < cnt; p++) {
        const int key = a.seg_list[seg0 + p];
        uint32_t n = 0;
        if (key >= 0 && (uint32_t)key < nlist) n = (uint32_t)(a.list_off[key + 1] - a.list_off[key]);
        const unsigned long long roff = a.seg_off[seg0 + p];
        const uint32_t c0 = cursor;
        bool overflow = false;
        if (n && masked) {
            const unsigned long long* mrow = a.mask + (roff >> 6);
            const uint32_t nchunk = (n + 63) >> 6;
            for (uint32_t w0 = 0; w0 < nchunk && !overflow; w0 += 64) {
                unsigned long long word = w0 + lane < nchunk ? mrow[w0 + lane] : 0ull;
                const uint32_t pc = (uint32_t)__builtin_popcountll(word);
                if (!__ballot(pc != 0)) continue;
                uint32_t incl = pc;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
                    if (lane >= off) incl += o;
                }
                const uint32_t total = (uint32_t)__shfl((int)incl, 63);
                if (cursor + total > a.capq) {
                    overflow = true;
                    break;
                }
                size_t idx = base + cursor + incl - pc;
                while (word) {
                    const int b = __builtin_ctzll(word);
                    word &= word - 1;
                    const uint32_t pos = (w0 + lane) * 64 + b;
                    a.cand[idx] = make_uint2(__float_as_uint(a.dist[roff + pos]), pos);
                    idx++;
                }
                cursor += total;
            }
        } else if (n) {
            const float* row = a.dist + roff;
            for (uint32_t j0 = 0; j0 < n && !overflow; j0 += 256) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t j = j0 + u * 64 + lane;
                    v[u] = j < n ? __builtin_nontemporal_load(row + j) : hneutral<IsMax>();
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (j0 + u * 64 >= n) break;
                    if (npool + 64 > (uint32_t)POOL_CAP) {
                        refresh();
                        since = 0;
                    }
                    const bool pass = hcmp<IsMax>(T, v[u]);
                    const unsigned long long bal = __ballot(pass);
                    if (bal) {
                        const uint32_t c = (uint32_t)__builtin_popcountll(bal);
                        if (cursor + c > a.capq) {
                            overflow = true;
                            break;
                        }
                        const uint32_t r = (uint32_t)__builtin_popcountll(bal & lt_mask);
                        if (pass) {
                            a.cand[base + cursor + r] = make_uint2(__float_as_uint(v[u]), j0 + u * 64 + lane);
                            pool[npool + r] = v[u];
                        }
                        cursor += c;
                        npool += c;
                    }
                    since += 64;
                    if (since >= next_refresh) {
                        wave_sync();
                        refresh();
                        since = 0;
                        next_refresh = next_refresh < 4096 ? next_refresh * 2 : 4096;
                    }
                }
            }
        }
        if (overflow) {  // never at p == 0 (capq >= list length): the rest of the round is left to the next one
            cursor = c0;
            probes_done = p;
            break;
        }
        if (lane == 0) a.cmeta[seg0 + p] = make_uint4(cursor - c0, n, (uint32_t)key, 0u);
    }
    if (lane == 0) a.cprobes[qi] = probes_done;
}

// Trace::search on a trace held as x | y | std (z = y + std_m * std formed with the reference's expression)
__device__ inline float trace_search_xyz(const float* x, const float* y, const float* sd, float sc, uint32_t n, float kv) {
    if (kv <= x[0]) return y[0] + sc * sd[0];
    if (kv >= x[n - 1]) {
        const float ampli = kv / x[n - 1];
        return (y[n - 1] + sc * sd[n - 1]) * ampli;
    }
    unsigned long long high = n - 1, low = 0, middle = 0;
    while (low <= high) {
        middle = (low + high) / 2;
        if (x[middle] < kv) low = middle + 1;
        else high = middle - 1;
    }
    if (x[low] > kv) low--;
    return y[low] + sc * sd[low];
}

constexpr int LANES_CHUNK = 64;            // K2: candidates per query staged in LDS at a time
constexpr int LANES_CQ_ROW = 65;           // entries per staged row (+1: the transposing writes spread over the banks)
__host__ __device__ inline size_t lanes_lds_bytes(int k, bool tune, uint32_t trace_cap) {
    size_t b = (size_t)k * LANES_CQ_ROW * 8;               // heap: (value, id slot) per node and lane, rows of 65
    b += (size_t)LANES_CHUNK * LANES_CQ_ROW * 8;           // staged candidates: (value, position)
    if (tune) b += 512 * 4 + 16 * 64 * 4 + (size_t)trace_cap * 8;  // acos LUT | disToBoundary windows | cached trace (x | z)
    return b;
}

// One query per lane, `a.lanes` lanes per wave (a launch wants about one wave per CU: the kernel is a chain of dependent LDS
// round trips per query, and work that only some lanes have -- a heap update, a rule evaluation -- costs the wave its full
// latency whatever the number of lanes that take part).
//   * a probe's candidates are staged through LDS 64 per query at a time (coalesced row loads, transposing writes); every
//     lane then runs ahead on its own to its next candidate that beats its heap top, and the lanes that found one update
//     their heaps together: the wave pays for max-over-lanes admissions per chunk, not for every candidate position;
//   * heap nodes are (value, id slot) pairs, one 8-byte LDS access each; heap_pop reads children and grandchildren together
//     and decides two levels per round trip; heap_push reads the fixed ancestor chain of node k in one;
//   * the LANES_BEST best values live in registers (insertion network), the window of disToBoundary and the stage's trace
//     in LDS.
template <bool IsMax>
__global__ __launch_bounds__(64) void replay_lanes_kernel(ReplayArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int k = a.k;
    const uint32_t nlist = a.nlist;
    const bool tune = a.tuner.enabled != 0;
    uint2* h64 = reinterpret_cast<uint2*>(smem);
    uint2* cq = h64 + (size_t)k * LANES_CQ_ROW;
    float* lut = reinterpret_cast<float*>(cq + LANES_CHUNK * LANES_CQ_ROW);
    float* dwin = lut + 512;
    float* trc = dwin + 16 * 64;
#define H(i) h64[(size_t)(i) * LANES_CQ_ROW + lane]
#define HC(i, col) h64[(size_t)(i) * LANES_CQ_ROW + (col)]
#define HVAL(i) __uint_as_float(H(i).x)
#define DWIN(i) dwin[(i) * 64 + lane]
    if (tune) {
        for (int i = lane; i < 500; i += 64) lut[i] = a.tuner.arcos[i];
    }
    const uint32_t nact = a.nq_dev ? *a.nq_dev : a.nq;
    const uint32_t L = a.lanes;
    if (blockIdx.x * L >= nact) return;  // whole wave idle
    const bool dbg = a.dbg != nullptr;
    const unsigned long long t_start = dbg ? __builtin_readcyclecounter() : 0;
    unsigned long long t_stage = 0, t_loop = 0, t_rule = 0, n_upd = 0, n_skip = 0;
    const uint32_t li = blockIdx.x * L + lane;
    const bool mine = (uint32_t)lane < L && li < nact;
    const uint32_t qi = mine ? (a.qsel ? a.qsel[li] : li) : 0u;
    const bool live = mine && !a.done[qi];
    const size_t hb = (size_t)qi * k;
    const uint32_t max_num = nlist / 8 + 20;

    // ---- heap of the query into this lane's column; ids stay in a table indexed by slot
    float b[LANES_BEST];  // best values, best first
#pragma unroll
    for (int i = 0; i < LANES_BEST; i++) b[i] = hneutral<IsMax>();
    auto best_insert = [&](float x) {  // x is known to beat b[LANES_BEST - 1]; equal values keep their order
#pragma unroll
        for (int i = 0; i < LANES_BEST; i++) {
            const bool better = hcmp<IsMax>(b[i], x);
            const float t = better ? x : b[i];
            x = better ? b[i] : x;
            b[i] = t;
        }
    };
    auto best_get = [&](uint32_t m) {
        float r = b[0];
#pragma unroll
        for (int i = 1; i < LANES_BEST; i++) r = m == (uint32_t)i ? b[i] : r;
        return r;
    };
    // (row by row with the whole wave: a lane walking its own k entries would pay a memory round trip per entry)
    for (uint32_t r = 0; r < L; r++) {
        const bool lr = __builtin_amdgcn_readlane((int)live, (int)r) != 0;
        const size_t rb = (size_t)(uint32_t)__builtin_amdgcn_readlane((int)qi, (int)r) * k;
        for (int i = lane; i < k; i += 64) {
            HC(i, r) = make_uint2(__float_as_uint(lr ? a.heap_val[rb + i] : hneutral<IsMax>()), (uint32_t)i);
            if (lr) a.href_tmp[rb + i] = a.heap_ref[rb + i];
        }
    }
    wave_sync();
    for (int i = 0; i < k; i++) {
        const float x = HVAL(i);
        if (hcmp<IsMax>(b[LANES_BEST - 1], x)) best_insert(x);
    }
    // ancestors of node k (heap_push always starts there): k, k/2, ..., 1
    int depth = 0;
    for (int t = k; t >= 1; t >>= 1) depth++;
    wave_sync();
    const unsigned long long t_pro = dbg ? __builtin_readcyclecounter() : 0;

    const unsigned long long id_q = a.id_offset + qi;
    uint32_t err = 0;
    const uint32_t ik0 = live ? a.stage[qi] : 0u;
    const uint32_t loop_end = a.total_nprobe;
    const uint32_t planned = live ? a.seg_count[qi] : 0u, ready = live ? a.cprobes[qi] : 0u;
    const uint32_t cnt = planned < ready ? planned : ready;
    const bool truncated = live && ready < planned;
    const size_t seg0 = live ? (size_t)a.seg_begin[qi] : 0;
    const uint2* cand = a.cand + (size_t)qi * a.capq;
    unsigned long long nscan = live ? a.nscan[qi] : 0ull;
    float pre_val = live && a.pre_val ? a.pre_val[qi] : 0.f;
    uint32_t stoped = live && a.stoped ? a.stoped[qi] : 0u;
    unsigned long long st_nlist = 0, st_nheap = 0, st_ndis = 0;
    float top = HVAL(0);

    const uint32_t query_k = tune ? a.tuner.query_topk : 0u;
    float true_KD_K = 0.f, racc = 0.f;
    unsigned long long np = 0;
    int cached_ind = -1;       // this lane's window / cur_num cache
    int trace_ind = -1;        // trace held in LDS (wave-uniform)
    uint32_t trace_n = 0;
    bool have_pre = false, top_changed = true;
    uint32_t kept_pre = 0;
    if (tune && live) {
        if (a.tuner.gt_D) true_KD_K = a.tuner.gt_D[id_q * (unsigned long long)k + query_k - 1];
        racc = a.tuner.require_acc[id_q];
        np = a.tuner.my_nprobe[id_q];
    }
    const unsigned long long np_in = np;
    const float* gdtb = tune ? a.dtb + (size_t)qi * max_num : nullptr;

    // one heap update: heap_pop (Heap.h:88-118) then heap_push (Heap.h:125-142) of (val, slot of the evicted root)
    auto heap_update = [&](float val, int64_t ref) {
        const uint32_t sr = H(0).y;
        float rootv;  // the root after the pop
        {
            const uint2 ve = H(k - 1);
            const float v = __uint_as_float(ve.x);
            int i = 1;
            bool first = true;
            rootv = v;
            for (;;) {
                const int i1 = i << 1;
                if (i1 > k) break;
                const int i2 = i1 + 1, g = i << 2;
                // children and grandchildren in one round trip (indices past k are clamped and ignored)
                const int n2 = i2 <= k ? i2 : k, g0 = g <= k ? g : k, g1 = g + 1 <= k ? g + 1 : k, g2 = g + 2 <= k ? g + 2 : k,
                          g3 = g + 3 <= k ? g + 3 : k;
                const uint2 c1 = H(i1 - 1), c2 = H(n2 - 1), d0 = H(g0 - 1), d1 = H(g1 - 1), d2 = H(g2 - 1), d3 = H(g3 - 1);
                const bool leftA = (i2 == k + 1) || hcmp<IsMax>(__uint_as_float(c1.x), __uint_as_float(c2.x));
                const uint2 c = leftA ? c1 : c2;
                if (hcmp<IsMax>(v, __uint_as_float(c.x))) break;
                H(i - 1) = c;
                if (first) rootv = __uint_as_float(c.x);
                first = false;
                i = leftA ? i1 : i2;
                const int j1 = i << 1;
                if (j1 > k) break;
                const int j2 = j1 + 1;
                const uint2 e1 = leftA ? d0 : d2, e2 = leftA ? d1 : d3;
                const bool leftB = (j2 == k + 1) || hcmp<IsMax>(__uint_as_float(e1.x), __uint_as_float(e2.x));
                const uint2 e = leftB ? e1 : e2;
                if (hcmp<IsMax>(v, __uint_as_float(e.x))) break;
                H(i - 1) = e;
                i = leftB ? j1 : j2;
            }
            H(i - 1) = ve;
        }
        {
            // the ancestors of node k, read together, then shifted down as far as val climbs
            uint2 fe[8];
#pragma unroll
            for (int t = 1; t < 8; t++) fe[t] = H((t < depth ? k >> t : 1) - 1);
            int i = k;
            bool reached_root = depth == 1;
#pragma unroll
            for (int t = 1; t < 8; t++) {
                if (t >= depth) break;
                if (!hcmp<IsMax>(val, __uint_as_float(fe[t].x))) break;
                H(i - 1) = fe[t];
                i = k >> t;
                if (t == depth - 1) reached_root = true;
            }
            H(i - 1) = make_uint2(__float_as_uint(val), sr);
            top = reached_root ? val : rootv;
        }
        a.href_tmp[hb + sr] = ref;
        st_nheap++;
        if (hcmp<IsMax>(b[LANES_BEST - 1], val)) {
            if (query_k && hcmp<IsMax>(b[query_k - 1], val)) top_changed = true;
            best_insert(val);
        }
    };

    bool finished = false;
    uint32_t consumed = 0, cur = 0;
    uint4 meta_next = make_uint4(0u, 0u, 0xffffffffu, 0u);
    if (cnt) meta_next = a.cmeta[seg0];
    uint32_t maxcnt = cnt;
    for (int off = 32; off; off >>= 1) maxcnt = max(maxcnt, (uint32_t)__shfl_xor((int)maxcnt, off));

    for (uint32_t p = 0; p < maxcnt; p++) {
        const bool on = live && !finished && p < cnt;
        if (!__ballot(on)) break;
        const uint32_t ik = ik0 + p;
        uint32_t ncand = 0, n = 0;
        int key = -1;
        const uint4 meta = meta_next;  // (candidates, list length, key) of this probe, requested one probe ahead
        if (live && p + 1 < cnt) meta_next = a.cmeta[seg0 + p + 1];
        if (on) {
            consumed = p + 1;
            key = (int)meta.z;
            if (key >= 0) {
                if ((uint32_t)key >= nlist) {
                    err = ERR_INVALID_KEY;
                    finished = true;
                } else {
                    n = meta.y;
                    ncand = meta.x;
                }
            }
        }
        const bool scan = on && !finished && n > 0;
        // ---- the probe's candidates (IndexIVFFlat.cpp:125-135: only what is strictly better than the top enters)
        uint32_t maxc = scan ? ncand : 0u;
        for (int off = 32; off; off >>= 1) maxc = max(maxc, (uint32_t)__shfl_xor((int)maxc, off));
        const int64_t refbase = REF_TAG | ((int64_t)key << 32);
        for (uint32_t c0 = 0; c0 < maxc; c0 += LANES_CHUNK) {
            // stage entries [c0, c0 + 64) of every scanning lane's list: row r (lane r's list) is read by the whole wave, entry e
            // by lane e, and lands in column r of the staged chunk
            const uint32_t left = scan && ncand > c0 ? ncand - c0 : 0u;
            const uint2* rowp = cand + cur + c0;
            const unsigned long long t_s0 = dbg ? __builtin_readcyclecounter() : 0;
            wave_sync();
            for (uint32_t r = 0; r < L; r++) {
                const uint32_t rn = (uint32_t)__builtin_amdgcn_readlane((int)left, (int)r);
                if (!rn) continue;
                const unsigned long long rp = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)((unsigned long long)rowp >> 32), (int)r) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(unsigned long long)rowp, (int)r);
                if ((uint32_t)lane < rn) cq[(size_t)lane * LANES_CQ_ROW + r] = reinterpret_cast<const uint2*>(rp)[lane];
            }
            wave_sync();
            const unsigned long long t_s1 = dbg ? __builtin_readcyclecounter() : 0;
            t_stage += t_s1 - t_s0;
            const uint32_t ne = left < (uint32_t)LANES_CHUNK ? left : (uint32_t)LANES_CHUNK;
            uint32_t e = 0;
            for (;;) {
                // every lane runs ahead to its next candidate that beats its top
                bool has = false;
                uint2 ent = make_uint2(0u, 0u);
                while (e < ne) {
                    ent = cq[(size_t)e * LANES_CQ_ROW + lane];
                    e++;
                    if (hcmp<IsMax>(top, __uint_as_float(ent.x))) {
                        has = true;
                        break;
                    }
                }
                n_skip++;
                if (!__ballot(has)) break;
                n_upd++;
                if (has) heap_update(__uint_as_float(ent.x), refbase | (int64_t)ent.y);
            }
            if (dbg) t_loop += __builtin_readcyclecounter() - t_s1;
        }
        const unsigned long long t_r0 = dbg ? __builtin_readcyclecounter() : 0;
        if (scan) {
            cur += ncand;
            st_nlist++;
            nscan += n;
            st_ndis += n;
        }
        bool rule = on && !finished;
        if (rule && a.max_codes && nscan >= a.max_codes) {
            finished = true;
            rule = false;
        }
        if (rule && loop_end && ik + 1 >= loop_end) finished = true;  // end of the probe loop (the rule is still evaluated)
        if (tune) {
            // IndexIVF.cpp:551-638, one query per lane.  Once my_nprobe is known nothing the rule computes can change the
            // outcome any more (L2: no throwing path left), so only the stop test remains.
            const uint32_t stage = ik + 1;
            const bool fired = IsMax && np != 0;
            const bool eval = rule && !fired;
            uint32_t ind = 0;
            {
                const uint32_t tmp_stage = stage >= nlist / 8 ? nlist / 8 - 1 : stage;
                while (tmp_stage > (1u << ind)) ind++;
            }
            // the trace of the first evaluating lane's stage goes to LDS (the lanes of a wave are nearly always at the same
            // stage); a lane at another stage reads its trace from memory
            const unsigned long long em = __ballot(eval);
            if (em) {
                const int ind_u = __shfl((int)ind, __builtin_ctzll(em));
                if (trace_ind != ind_u) {
                    wave_sync();
                    const uint32_t o = a.tuner.trace_off[ind_u], tn = a.tuner.trace_off[ind_u + 1] - o;
                    const float sc = a.tuner.std_m;
                    for (uint32_t i = lane; i < tn; i += 64) {
                        trc[i] = a.tuner.trace_x[o + i];
                        trc[a.trace_cap + i] = a.tuner.trace_y[o + i] + sc * a.tuner.trace_std[o + i];
                    }
                    trace_ind = ind_u;
                    trace_n = tn;
                    wave_sync();
                }
            }
            if (eval) {
                if ((int)ind != cached_ind) {
                    for (int i = 0; i < 15; i++) DWIN(i) = gdtb[(1u << ind) - 1 + i];  // sum_angle start = 2^ind - 1
                    cached_ind = (int)ind;
                    have_pre = false;
                }
                if (!IsMax) {
                    // the reference converts all k heap values (IndexIVF.cpp:562-564): any out-of-domain one throws.  The
                    // smallest is the root, the largest the best value.
                    (void)arcos_lut(lut, top, &err);
                    (void)arcos_lut(lut, b[0], &err);
                }
                if (!err && (!have_pre || top_changed)) {
                    const TraceLds tr{trc, trc + a.trace_cap, trace_n};
                    const bool in_lds = (int)ind == trace_ind;
                    uint32_t go = 0, gn = 0;
                    if (!in_lds) {
                        go = a.tuner.trace_off[ind];
                        gn = a.tuner.trace_off[ind + 1] - go;
                    }
                    float dw[15];
#pragma unroll
                    for (int i = 0; i < 15; i++) dw[i] = DWIN(i);
                    auto S = [&](unsigned long long m) {
                        const float bm = best_get((uint32_t)m);
                        const float kd = IsMax ? bm : arcos_lut(lut, bm, &err);
                        float sum = 0.f;
#pragma unroll
                        for (int i = 0; i < 15; i++) {
                            float t = 0.f;
                            if (!(dw[i] >= kd)) t = arcos_lut(lut, dw[i] / kd, &err);
                            sum += t;
                        }
                        if (in_lds) return trace_search(tr.x, tr.z, tr.n, sum);
                        return trace_search_xyz(a.tuner.trace_x + go, a.tuner.trace_y + go, a.tuner.trace_std + go, a.tuner.std_m, gn, sum);
                    };
                    const unsigned long long qk = query_k;
                    unsigned long long high = qk - 1, low = 0, middle = 0;
                    uint32_t res = 0;
                    bool found = false;
                    {
                        const float g = S(high);
                        if ((double)((float)qk * g) <= (double)qk * 1.005) {
                            res = (uint32_t)qk;
                            found = true;
                        }
                    }
                    while (!found && !err && low <= high) {
                        middle = (low + high) / 2;
                        if (middle <= 0) {
                            res = 0;
                            found = true;
                            break;
                        }
                        const float g = S(middle);
                        if ((float)(middle + 1) * g <= (float)qk) low = middle + 1;
                        else high = middle - 1;
                    }
                    if (!found) res = (uint32_t)(low + 1);
                    kept_pre = res;
                    have_pre = true;
                    top_changed = false;
                }
                if (err) {
                    finished = true;
                } else {
                    float recall = (float)kept_pre / (float)query_k;
                    const float max_val = IsMax ? fmaxf(-1.f, top) : fminf(FLT_MAX, top);
                    const unsigned long long stops = (unsigned long long)(racc * 12);
                    if (stage > 1) {
                        if (max_val == pre_val) stoped++;
                        else stoped = 0;
                        if (stoped >= stops) recall = 1;
                    }
                    pre_val = max_val;
                    if (recall >= racc && np == 0) {
                        np = (unsigned long long)((float)stage * a.tuner.multipler);
                        if (np >= nlist) a.tuner.t_recalls[id_q] = 1.f;
                    }
                    if (stage >= nlist / 8 && np == 0) {
                        np = (unsigned long long)((float)stage * a.tuner.multipler);
                        if (np >= nlist) a.tuner.t_recalls[id_q] = 1.f;
                    }
                }
            }
            if (rule && !err && np != 0 && np <= stage) {
                if (a.tuner.profile) {
                    uint32_t hits = 0;
                    for (int i = 0; i < k; i++) {
                        const float s = HVAL(i);
                        if (IsMax ? ((double)s <= (double)true_KD_K * 1.0005) : ((double)s >= (double)true_KD_K * 0.9995)) hits++;
                    }
                    a.tuner.t_recalls[id_q] = (float)hits / (float)query_k;
                }
                finished = true;
            }
        }
        if (dbg) t_rule += __builtin_readcyclecounter() - t_r0;
    }

    // ---- state out
    const unsigned long long t_epi = dbg ? __builtin_readcyclecounter() : 0;
    unsigned long long tot_nlist = st_nlist, tot_ndis = st_ndis, tot_nheap = st_nheap;
    uint32_t werr = err;
    for (int off = 32; off; off >>= 1) {
        tot_nlist += __shfl_xor(tot_nlist, off);
        tot_ndis += __shfl_xor(tot_ndis, off);
        tot_nheap += __shfl_xor(tot_nheap, off);
        const uint32_t o = (uint32_t)__shfl_xor((int)werr, off);
        werr = werr > o ? werr : o;
    }
    if (lane == 0) {
        if (tot_nlist) atomicAdd(&a.stats[0], tot_nlist);
        if (tot_ndis) atomicAdd(&a.stats[1], tot_ndis);
        if (tot_nheap) atomicAdd(&a.stats[2], tot_nheap);
        if (werr) atomicMax(a.error, werr);
    }
    if (live) {
        a.stage[qi] = ik0 + consumed;
        a.nscan[qi] = nscan;
        if (a.pre_val) a.pre_val[qi] = pre_val;
        if (a.stoped) a.stoped[qi] = stoped;
        if (tune && np != np_in) a.tuner.my_nprobe[id_q] = np;
        if (a.thr) a.thr[qi] = top;  // next round's scan stores only what beats this
    }
    if (truncated && !finished && !err) atomicAdd(&a.stats[3], 1ull);  // the host plans another round for what is left
    const bool finalize = live && (finished || err || (a.finalize_all && !truncated));
    int ii = 0;
    if (finalize) {
        // heap_reorder (Heap.h:295-322): k pops, valid entries collected from the back.  An entry without an id (-1) is one
        // of the initial ones, and those are the only entries holding the neutral value (anything admitted beat a top).
        for (int i = 0; i < k; i++) {
            const uint2 r0 = H(0);
            const int kk = k - i;
            {
                const uint2 ve = H(kk - 1);
                const float v = __uint_as_float(ve.x);
                int n1 = 1;
                for (;;) {
                    const int i1 = n1 << 1, i2 = i1 + 1;
                    if (i1 > kk) break;
                    const int j2 = i2 <= kk ? i2 : i1;
                    const uint2 c1 = H(i1 - 1), c2 = H(j2 - 1);
                    const bool left = (i2 == kk + 1) || hcmp<IsMax>(__uint_as_float(c1.x), __uint_as_float(c2.x));
                    const uint2 c = left ? c1 : c2;
                    if (hcmp<IsMax>(v, __uint_as_float(c.x))) break;
                    H(n1 - 1) = c;
                    n1 = left ? i1 : i2;
                }
                H(n1 - 1) = ve;
            }
            H(k - ii - 1) = r0;
            if (__uint_as_float(r0.x) != hneutral<IsMax>()) ii++;
        }
        a.done[qi] = 1;
    }
    wave_sync();
    // results / carried state leave row by row with the whole wave (ids through the slot table)
    for (uint32_t r = 0; r < L; r++) {
        if (!__builtin_amdgcn_readlane((int)live, (int)r)) continue;
        const size_t rb = (size_t)(uint32_t)__builtin_amdgcn_readlane((int)qi, (int)r) * k;
        if (__builtin_amdgcn_readlane((int)finalize, (int)r)) {
            const int iir = __builtin_amdgcn_readlane(ii, (int)r);  // valid entries sit in [k - ii, k): to the front, pad the rest
            for (int i = lane; i < k; i += 64) {
                float v = hneutral<IsMax>();
                int64_t id = -1;
                if (i < iir) {
                    const uint2 en = HC(k - iir + i, r);
                    v = __uint_as_float(en.x);
                    int64_t ref = a.href_tmp[rb + en.y];
                    if (ref >= 0 && (ref & REF_TAG)) {
                        ref &= ~REF_TAG;
                        if (!a.store_pairs) ref = a.ids[a.list_off[ref >> 32] + (uint64_t)(ref & 0xffffffffll)];
                    }
                    id = ref;
                }
                a.D[rb + i] = v;
                a.I[rb + i] = id;
            }
        } else {
            for (int i = lane; i < k; i += 64) {
                const uint2 en = HC(i, r);
                a.heap_val[rb + i] = __uint_as_float(en.x);
                a.heap_ref[rb + i] = a.href_tmp[rb + en.y];
            }
        }
    }
    if (dbg && lane == 0) {
        unsigned long long* o = a.dbg + (size_t)blockIdx.x * 8;
        const unsigned long long t_end = __builtin_readcyclecounter();
        o[0] = t_end - t_start;  // wave cycles
        o[1] = n_upd;            // wave-level heap updates
        o[2] = n_skip;           // run-ahead rounds
        o[3] = t_pro - t_start;  // prologue cycles
        o[4] = t_loop;           // candidate loop cycles
        o[5] = t_rule;           // rule cycles
        o[6] = t_stage;          // staging cycles
        o[7] = t_end - t_epi;    // epilogue cycles
    }
#undef H
#undef HC
#undef HVAL
#undef DWIN
}

bool select_lanes_supported(const ReplayArgs& a) {
    // read per call: the tests run both selections in one process
    const char* e = getenv("AUNCEL_AMD_LANES");
    if (e ? atoi(e) == 0 : !LANES_DEFAULT) return false;
    if (a.k < 1 || a.k > LANES_MAXK || a.k > POOL_CAP - 192) return false;
    if (a.train.enabled || a.raw_heap_out || a.identity_ids || a.limit) return false;
    if (!a.seg_by_slot || !a.seg_begin || !a.qsel || !a.cand) return false;
    if (a.tuner.enabled && (a.tuner.query_topk > (uint32_t)LANES_BEST || a.tuner.query_topk > (uint32_t)a.k)) return false;
    if (a.tuner.enabled && a.trace_cap > 4096) return false;
    if (a.tuner.enabled && a.tuner.overhead) return false;
    return true;
}

void launch_select_lanes(const ReplayArgs& a, hipStream_t s) {
    if (a.nq == 0) return;
    const bool tune = a.tuner.enabled != 0;
    const size_t shmem = lanes_lds_bytes(a.k, tune, a.trace_cap);
    ReplayArgs b = a;
    // lanes per wave: about one wave per CU (see replay_lanes_kernel); nq_hint = queries expected in this launch
    const uint32_t expect = a.nq_hint ? a.nq_hint : a.nq;
    const char* le = getenv("AUNCEL_AMD_LANES_PER_WAVE");
    uint32_t L = le ? (uint32_t)atoi(le) : 0;
    if (!L) {
        L = 4;
        while (L < 64 && expect / L > resident_grid(1)) L *= 2;
    }
    b.lanes = L < 1 ? 1 : L > 64 ? 64 : L;
    const dim3 g1((a.nq + 3) / 4), g2((a.nq + b.lanes - 1) / b.lanes);
    auto go = [&](auto k1, auto k2) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) throw std::runtime_error(std::string("selection kernel: cannot reserve LDS: ") + hipGetErrorString(e));
        LAUNCH(k1, g1, dim3(256), 0, s, b);
        LAUNCH(k2, g2, dim3(64), shmem, s, b);
    };
    if (a.metric == METRIC_L2) go(compact_kernel<true>, replay_lanes_kernel<true>);
    else go(compact_kernel<false>, replay_lanes_kernel<false>);
}

// =============================================================================================
// range search: count / fill over the threshold masks (RangeArgs in ivf_kernels.h)
// =============================================================================================
template <bool FILL>
__global__ __launch_bounds__(256) void range_collect_kernel(RangeArgs a) {
    const uint32_t li = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (li >= a.nq) return;
    const uint32_t qi = a.qsel[li];
    const uint32_t cnt = a.seg_count[qi];
    const size_t seg0 = a.seg_begin[qi];
    unsigned long long pos = FILL ? a.out_off[qi] : 0ull;  // wave-uniform
    unsigned long long nlistv = 0, ndis = 0;
    uint32_t total = 0, err = 0;
    for (uint32_t p = 0; p < cnt; p++) {
        const int key = a.seg_list[seg0 + p];
        if (key < 0) continue;
        if ((uint32_t)key >= a.nlist) {
            err = ERR_INVALID_KEY;
            break;
        }
        const unsigned long long lb = a.list_off[key];
        const uint32_t n = (uint32_t)(a.list_off[key + 1] - lb);
        if (n == 0) continue;
        nlistv++;
        ndis += n;
        const unsigned long long roff = a.seg_off[seg0 + p];
        const unsigned long long* mrow = a.mask + (roff >> 6);
        const uint32_t nchunk = (n + 63) >> 6;
        for (uint32_t w0 = 0; w0 < nchunk; w0 += 64) {
            const unsigned long long word = w0 + lane < nchunk ? mrow[w0 + lane] : 0ull;
            const uint32_t pc = (uint32_t)__builtin_popcountll(word);
            if (!FILL) {
                total += pc;
                continue;
            }
            uint32_t incl = pc;  // inclusive prefix over the lanes: where this lane's chunk starts in the output
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
                if (lane >= off) incl += o;
            }
            unsigned long long o = pos + incl - pc;
            unsigned long long bits = word;
            while (bits) {
                const int b = __builtin_ctzll(bits);
                bits &= bits - 1;
                const uint32_t cand = (w0 + lane) * 64 + b;
                a.out_dist[o] = a.dist[roff + cand];
                a.out_labels[o] = a.ids[lb + cand];
                o++;
            }
            pos += (uint32_t)__shfl((int)incl, 63);
        }
    }
    if (!FILL) {
        for (int off = 32; off; off >>= 1) total += __shfl_xor(total, off);
        if (lane == 0) {
            a.counts[qi] = total;
            a.stage[qi] += cnt;
            a.done[qi] = 1;
            if (nlistv) atomicAdd(&a.stats[0], nlistv);
            if (ndis) atomicAdd(&a.stats[1], ndis);
            if (err) atomicMax(a.error, err);
        }
    }
}

void launch_range_count(const RangeArgs& a, hipStream_t s) {
    if (a.nq) LAUNCH(range_collect_kernel<false>, dim3((a.nq + 3) / 4), dim3(256), 0, s, a);
}
void launch_range_fill(const RangeArgs& a, hipStream_t s) {
    if (a.nq) LAUNCH(range_collect_kernel<true>, dim3((a.nq + 3) / 4), dim3(256), 0, s, a);
}

// =============================================================================================
// coarse quantiser, GEMM formulation (the reference's knn_L2sqr_blas / knn_inner_product_blas,
// utils.cpp:494-608): dis = |x|^2 + |y|^2 - 2 x.y clamped at 0, x.y on the fp32 matrix cores
// =============================================================================================
// This is the one dense contraction of the hot path.  v_mfma_f32_32x32x2_f32 accumulates a k-ordered fp32 fma
// chain per output element; the reference's sgemm order belongs to the vendor BLAS, so on float data the two
// agree to rounding only (exactly on integer-valued data, where every order is exact).
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void row_norms_kernel(const float* x, size_t n, int d, float* out) {
    // fvec_norm_L2sqr, SSE order (utils_simd.cpp): four running sums, (s0+s1)+(s2+s3)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4* r = reinterpret_cast<const float4*>(x + i * d);
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int c = 0; c < d / 4; c++) {
        const float4 v = r[c];
        s0 += v.x * v.x;
        s1 += v.y * v.y;
        s2 += v.z * v.z;
        s3 += v.w * v.w;
    }
    out[i] = (s0 + s1) + (s2 + s3);
}

// 128 x 128 output tile per workgroup, a 64 x 64 quarter (2 x 2 MFMA tiles, 64 accumulator registers) per wave; K in
// chunks of 16 dimensions, double-buffered through LDS with one barrier per chunk, fetched with 16-byte loads one chunk
// ahead.  An LDS row holds the even dimensions of the chunk, then the odd ones ([row][k & 1][k >> 1], stride 20 floats): lane
// (m, h) of v_mfma_f32_32x32x2_f32 takes A[m][2 s + h], so one ds_read_b128 feeds four consecutive k-steps, and eight
// lanes' rows fall on disjoint banks.  The k-steps run in ascending order as before (same rounding as the first version).
constexpr int GEMM_TK = 16, GEMM_LD = 20;

template <int METRIC>
__global__ __launch_bounds__(256) void coarse_gemm_kernel(const float* X, const float* Y, const float* xn, const float* yn, int nq,
                                                          int ny, int d, float* out) {
    __shared__ __align__(16) float As[2][128 * GEMM_LD], Bs[2][128 * GEMM_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.y * 128, c0 = blockIdx.x * 128;
    const int wq = wave >> 1, wc = wave & 1;
    const int m = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float4 pa[2], pb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + i * 256, row = idx >> 2, c = idx & 3;
            const bool kok = k0 + 4 * c < d;
            pa[i] = q0 + row < nq && kok ? *reinterpret_cast<const float4*>(X + (size_t)(q0 + row) * d + k0 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            pb[i] = c0 + row < ny && kok ? *reinterpret_cast<const float4*>(Y + (size_t)(c0 + row) * d + k0 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + i * 256, row = idx >> 2, c = idx & 3;
            float* ar = &As[buf][row * GEMM_LD + 2 * c];
            float* br = &Bs[buf][row * GEMM_LD + 2 * c];
            *reinterpret_cast<float2*>(ar) = make_float2(pa[i].x, pa[i].z);      // dimensions 4c, 4c+2 -> even half
            *reinterpret_cast<float2*>(ar + 8) = make_float2(pa[i].y, pa[i].w);  // 4c+1, 4c+3 -> odd half
            *reinterpret_cast<float2*>(br) = make_float2(pb[i].x, pb[i].z);
            *reinterpret_cast<float2*>(br + 8) = make_float2(pb[i].y, pb[i].w);
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < d; k0 += GEMM_TK, buf ^= 1) {
        const bool more = k0 + GEMM_TK < d;
        if (more) fetch(k0 + GEMM_TK);
        const float* a_base = &As[buf][(wq * 64 + m) * GEMM_LD + h * 8];
        const float* b_base = &Bs[buf][(wc * 64 + m) * GEMM_LD + h * 8];
#pragma unroll
        for (int j0 = 0; j0 < 8; j0 += 4) {
            const float4 a0 = *reinterpret_cast<const float4*>(a_base + j0), a1 = *reinterpret_cast<const float4*>(a_base + 32 * GEMM_LD + j0);
            const float4 b0 = *reinterpret_cast<const float4*>(b_base + j0), b1 = *reinterpret_cast<const float4*>(b_base + 32 * GEMM_LD + j0);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int s = 0; s < 4; s++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], acc[1][1], 0, 0, 0);
            }
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h, col = m;
                const int q = q0 + wq * 64 + ti * 32 + row, c = c0 + wc * 64 + tj * 32 + col;
                if (q < nq && c < ny) {
                    const float ip = acc[ti][tj][reg];
                    float dis = ip;
                    if (METRIC == METRIC_L2) {
                        dis = xn[q] + yn[c] - 2 * ip;
                        if (dis < 0) dis = 0;  // utils.cpp:593
                    }
                    out[(size_t)q * ny + c] = dis;
                }
            }
}

void launch_row_norms(const float* x, size_t n, int d, float* out, hipStream_t s) {
    if (n) LAUNCH(row_norms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, d, out);
}

void launch_coarse_gemm(int metric, const float* X, const float* Y, const float* xn, const float* yn, int nq, int ny, int d, float* out,
                        hipStream_t s) {
    if (nq == 0 || ny == 0) return;
    const dim3 grid((ny + 127) / 128, (nq + 127) / 128);
    if (metric == METRIC_L2) LAUNCH(coarse_gemm_kernel<METRIC_L2>, grid, dim3(256), 0, s, X, Y, xn, yn, nq, ny, d, out);
    else LAUNCH(coarse_gemm_kernel<METRIC_IP>, grid, dim3(256), 0, s, X, Y, xn, yn, nq, ny, d, out);
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

// Only the first `prefix` entries of the ranking (the adaptive search never probes past (nlist / 8) * multipler):
// a workgroup keeps its row in registers (nlist <= 4096), finds by bisection on the order keys the threshold below
// which at least `prefix` values lie, and sorts just those (S = 1024 or 2048 slots in LDS).  Same total order as
// sort_rows_kernel; entries [prefix, nprobe) come out as (neutral distance, -1).
template <bool Ascending>
__global__ __launch_bounds__(256) void sort_prefix_kernel(const float* dis, uint32_t nlist, uint32_t nprobe, uint32_t prefix, uint32_t S,
                                                           float* out_dis, int64_t* out_keys) {
    __shared__ unsigned long long buf[2048];
    __shared__ uint32_t red[2][4];
    __shared__ uint32_t s_cnt;
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = dis + (size_t)q * nlist;
    constexpr int E = 16;
    uint32_t key[E];
#pragma unroll
    for (int j = 0; j < E; j++) {
        const uint32_t i = tid + 256 * j;
        key[j] = 0xffffffffu;
        if (i < nlist) key[j] = Ascending ? fkey(row[i]) : ~fkey(row[i]);
    }
    if (tid == 0) s_cnt = 0;
    int par = 0;
    // number of row entries for which pred holds (wave-uniform partial counts from ballots, then 4 waves through LDS)
    auto block_count = [&](auto pred) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < E; j++) c += __builtin_popcountll(__ballot(pred(key[j], tid + 256 * j)));
        if (lane == 0) red[par][wave] = c;
        __syncthreads();
        const uint32_t tot = red[par][0] + red[par][1] + red[par][2] + red[par][3];
        par ^= 1;
        return tot;
    };
    // smallest T with #(key <= T) >= prefix
    uint32_t lo = 0, hi = 0xffffffffu, cT = nlist;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        const uint32_t c = block_count([&](uint32_t k, uint32_t) { return k <= mid; });
        if (c >= prefix) {
            hi = mid;
            cT = c;
            if (c <= S) break;  // any threshold that keeps between prefix and S values will do
        } else {
            lo = mid + 1;
        }
    }
    const uint32_t T = hi;
    uint32_t I = 0xffffffffu;  // among the values equal to T keep the centroids numbered <= I
    if (cT > S) {              // a run of exactly equal distances straddles the cut: bisect on the centroid number
        const uint32_t below = block_count([&](uint32_t k, uint32_t) { return k < T; });
        uint32_t a = 0, b = nlist - 1;
        while (a < b) {
            const uint32_t mid = a + (b - a) / 2;
            const uint32_t c = below + block_count([&](uint32_t k, uint32_t i) { return k == T && i <= mid; });
            if (c >= prefix) b = mid;
            else a = mid + 1;
        }
        I = b;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < E; j++) {
        const uint32_t i = tid + 256 * j;
        const bool take = key[j] < T || (key[j] == T && i <= I);
        const unsigned long long m = __ballot(take);
        uint32_t base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_cnt, (uint32_t)__builtin_popcountll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (take) buf[base + __builtin_popcountll(m & ((1ull << lane) - 1))] = ((unsigned long long)key[j] << 32) | i;
    }
    __syncthreads();
    const uint32_t C = s_cnt;  // prefix <= C <= S
    for (uint32_t i = C + tid; i < S; i += 256) buf[i] = ~0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= S; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < S / 2; t += 256) {
                const uint32_t l = 2 * t - (t & (stride - 1)), h = l + stride;
                const bool up = (l & size) == 0;
                const unsigned long long x = buf[l], y = buf[h];
                if ((x < y) != up) {
                    buf[l] = y;
                    buf[h] = x;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < nprobe; i += 256) {
        float dv = Ascending ? FLT_MAX : -FLT_MAX;
        int64_t id = -1;
        if (i < prefix && i < C) {
            const unsigned long long e = buf[i];
            const uint32_t k = (uint32_t)(e >> 32);
            dv = fkey_inv(Ascending ? k : ~k);
            id = (int64_t)(uint32_t)e;
        }
        out_dis[(size_t)q * nprobe + i] = dv;
        out_keys[(size_t)q * nprobe + i] = id;
    }
}

// prefix: 0 = rank all nprobe entries; else only the first `prefix` are needed (see sort_prefix_kernel)
void launch_sort_rows(const float* dis, uint32_t nq, uint32_t nlist, uint32_t nprobe, int metric, float* out_dis,
                      int64_t* out_keys, hipStream_t s, uint32_t prefix) {
    if (nq == 0) return;
    if (prefix && prefix < nprobe && prefix <= 2048 && prefix <= nlist && nlist <= 4096) {
        const uint32_t S = prefix <= 1024 ? 1024u : 2048u;
        if (metric == METRIC_L2) LAUNCH(sort_prefix_kernel<true>, dim3(nq), dim3(256), 0, s, dis, nlist, nprobe, prefix, S, out_dis, out_keys);
        else LAUNCH(sort_prefix_kernel<false>, dim3(nq), dim3(256), 0, s, dis, nlist, nprobe, prefix, S, out_dis, out_keys);
        return;
    }
    uint32_t npow2 = 2;
    while (npow2 < nlist) npow2 <<= 1;
    const size_t shmem = (size_t)npow2 * 8;
    if (metric == METRIC_L2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sort_rows_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        LAUNCH(sort_rows_kernel<true>, dim3(nq), dim3(256), shmem, s, dis, nlist, npow2, nprobe, out_dis, out_keys);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sort_rows_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        LAUNCH(sort_rows_kernel<false>, dim3(nq), dim3(256), shmem, s, dis, nlist, npow2, nprobe, out_dis, out_keys);
    }
}

// =============================================================================================
// reference order inside runs of exactly equal coarse distances (rankings longer than 128)
// =============================================================================================
// knn_L2sqr_sse / knn_inner_product_sse (utils.cpp:417-490) keep a binary heap of nprobe entries over centroids
// 0..nlist-1 and heap-sort it at the end (Heap.h:88-142, 295-322): between exactly equal distances the output order is
// what the heap's history leaves, not a function of (distance, centroid number).  The sort kernels above order such runs
// by centroid number; this kernel re-runs the reference's heap for the rows in which that can show -- a run of equal
// distances inside the first `nout` entries (the part of the ranking the caller reads) or across their end -- and
// overwrites those entries.  One wave per row, heap and distance row in LDS; the heap walk is a chain of dependent
// compares, so lane 0 does it (rows with such runs are rare: two centroids at bit-equal fp32 distance from one query).
struct HeapEnt {
    float v;
    uint32_t id;
};

// heap_pop's walk (Heap.h:88-118) on the 1-based array h[1..k], started at slot `start`: the hole moves to the better child
// (the right one between equals) until `last` beats it, `last` lands in the hole
template <bool IsMax> __device__ inline void lds_sift_down(HeapEnt* h, uint32_t k, uint32_t start, HeapEnt last) {
    uint32_t i = start;
    while (true) {
        const uint32_t i1 = i << 1, i2 = i1 + 1;
        if (i1 > k) break;
        const HeapEnt c1 = h[i1], c2 = h[i2];  // h has k + 2 slots: reading past k is harmless
        if (i2 == k + 1 || hcmp<IsMax>(c1.v, c2.v)) {
            if (hcmp<IsMax>(last.v, c1.v)) break;
            h[i] = c1;
            i = i1;
        } else {
            if (hcmp<IsMax>(last.v, c2.v)) break;
            h[i] = c2;
            i = i2;
        }
    }
    h[i] = last;
}

template <bool IsMax> __device__ inline void lds_heap_pop(HeapEnt* h, uint32_t k) { lds_sift_down<IsMax>(h, k, 1, h[k]); }

// heap_push (Heap.h:125-142)
template <bool IsMax> __device__ inline void lds_heap_push(HeapEnt* h, uint32_t k, float val, uint32_t id) {
    uint32_t i = k;
    while (i > 1) {
        const uint32_t f = i >> 1;
        const HeapEnt fe = h[f];
        if (!hcmp<IsMax>(val, fe.v)) break;
        h[i] = fe;
        i = f;
    }
    h[i] = HeapEnt{val, id};
}

// ---- nprobe == nlist == n = 2^m (what Error_sys::search asks for, profile.cpp:220): the same heap without walking it
// one entry at a time.
//
// Filling.  With k = n every centroid enters: step j pops one of the n initial (FLT_MAX, -1) entries and pushes x_j into
// slot n, and the entry it re-inserts from the root is the one pushed the step before.  All initial entries are equal, so
// the pop's walk from the root runs through them (to the right between two of them, else towards the one that is left) to
// the last one on its way, and from there on it is an ordinary sift-down of x_{j-1} into the finished heaps below.  The
// initial entries therefore disappear in right-to-left post-order of slots 1..n-1, slot p receives x_{j(p)-1} with j(p) its
// place in that order, and every step is a sift-down inside subtree(p) only: steps of disjoint subtrees commute, so whole
// levels go at once, bottom-up (this is Floyd's heap construction with the reference's value-to-slot assignment and its
// tie rules).  Slot n hangs under the leftmost leaf n/2, the last slots in post-order are n/2 and its ancestors; from
// step n - m on the pushes can move up that path, and those m steps are replayed one by one.
//
// Heap sort (heap_reorder).  Pop t takes the entry of slot n - t and walks it down from the root; a walk only ever
// touches the level it is on and reads the one below, so the next pop can start two levels behind it.  Lanes hold the
// walks in flight, one level per tick each; a pop may not start while a walk in flight is above slot n - t (it could
// still change the entry the pop is about to take).  2.4 ticks per pop instead of one walk of m levels (measured, nlist
// 4096: 2.8 ms a row against 12.4 ms for the plain walk below; the filling takes 0.06 ms of that).
// (tests/heap_tie_model.py is the model both halves were checked with, entry for entry, against the literal heap.)
template <bool IsMax> __device__ inline void floyd_fill_pow2(HeapEnt* a, const float* row, uint32_t n, int lane) {
    const int m = 31 - __builtin_clz(n);
    const uint32_t F = n >> 1;
    for (int dp = m - 1; dp >= 1; dp--) {
        const uint32_t base = 1u << dp;
        for (uint32_t o = lane; o < base; o += 64) {
            const uint32_t p = base + o;
            if ((F >> (m - 1 - dp)) == p) continue;  // n/2 and its ancestors: the last m steps, below
            uint32_t j = (1u << (m - dp)) - 1;       // right-to-left post-order place of p: its own subtree ...
            for (int t = 1; t <= dp; t++)            // ... and the right siblings of the left turns on the way to it
                if (!((p >> (dp - t)) & 1)) j += (1u << (m - t)) - 1;
            lds_sift_down<IsMax>(a, n - 1, p, HeapEnt{row[j - 1], j - 1});
        }
        wave_sync();
    }
    if (lane == 0) {
        a[n] = HeapEnt{row[n - m - 1], n - m - 1};
        uint32_t A = F;
        for (uint32_t j = n - m; j < n; j++, A >>= 1) {
            lds_sift_down<IsMax>(a, n, A, a[n]);
            lds_heap_push<IsMax>(a, n, row[j], j);
        }
    }
    wave_sync();
}

template <bool IsMax> __device__ inline void heapsort_pipelined(HeapEnt* a, uint32_t n, uint32_t* out_id, int lane) {
    // one walk per lane: hole = slot the walk stands on (0: lane free), s = heap size of its pop, lvl = depth of hole,
    // (Lv, Lid) = the entry it carries.  Every lane runs the same straight-line tick: free lanes read slot 0 and store nothing.
    uint32_t hole = 0, s = 0, lvl = 0, Lid = 0;
    float Lv = 0.f;
    uint32_t t = 0, since = 2;  // pops started; ticks since the last start
    const uint4* a4 = reinterpret_cast<const uint4*>(a);
    const uint2* a2 = reinterpret_cast<const uint2*>(a);
    while (true) {
        const unsigned long long act = __ballot(hole != 0);
        if (t == n && !act) break;
        const uint32_t sc = n - t, dsc = 31 - __builtin_clz(sc | 1);
        const bool above = hole != 0 && lvl <= dsc && (sc >> ((dsc - lvl) & 31)) == hole;
        const bool create = t < n && since >= 2 && !__ballot(above);
        const bool mine = create && lane == __builtin_ctzll(~act);
        hole = mine ? 1u : hole;
        s = mine ? sc : s;
        lvl = mine ? 0u : lvl;
        const uint32_t i1 = hole << 1;
        const uint4 ch = a4[(i1 < n ? i1 : n) >> 1];  // both children (a has n + 2 slots)
        const uint2 ls = a2[mine ? sc : 0u];
        const uint32_t top_id = a2[1].y;
        if (mine) out_id[sc - 1] = top_id;  // heap_reorder: the top goes behind the shrinking heap
        Lv = mine ? __uint_as_float(ls.x) : Lv;
        Lid = mine ? ls.y : Lid;
        const float c1v = __uint_as_float(ch.x), c2v = __uint_as_float(ch.z);
        const bool left = i1 == s || hcmp<IsMax>(c1v, c2v);  // a single child, or the better one (the right one between equals)
        const float cv = left ? c1v : c2v;
        const uint32_t cid = left ? ch.y : ch.w;
        const bool done = i1 > s || hcmp<IsMax>(Lv, cv);
        if (hole != 0) a[hole] = HeapEnt{done ? Lv : cv, done ? Lid : cid};
        hole = hole != 0 && !done ? (left ? i1 : i1 + 1) : 0u;
        lvl++;
        t += create ? 1u : 0u;
        since = create ? 1u : since + 1;
        wave_sync();
    }
}

template <bool IsMax>
__global__ __launch_bounds__(64) void heap_tie_order_kernel(const float* dis, uint32_t nlist, uint32_t nprobe, uint32_t nout,
                                                            float* out_dis, int64_t* out_keys, unsigned long long* nrows) {
    extern __shared__ __align__(16) unsigned char smem[];
    HeapEnt* h = reinterpret_cast<HeapEnt*>(smem);                           // nprobe + 2 entries, [0] unused
    float* row = reinterpret_cast<float*>(smem + (size_t)(nprobe + 2) * 8);  // nlist
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const float* grow = dis + (size_t)q * nlist;
    float* od = out_dis + (size_t)q * nprobe;
    int64_t* ok = out_keys + (size_t)q * nprobe;
    const float neutral = IsMax ? FLT_MAX : -FLT_MAX;
    const uint32_t nreal = nout < nlist ? nout : nlist;  // leading entries that are centroids (the rest is padding)
    bool tie = false;
    for (uint32_t i = lane; i + 1 < nreal; i += 64) tie |= od[i] == od[i + 1];
    if (nreal && nreal < nlist) {
        // a run across the end of what was ranked: more centroids at the last distance than entries that carry it
        const float T = od[nreal - 1];
        uint32_t in_row = 0, in_out = 0;
        for (uint32_t j = lane; j < nlist; j += 64) in_row += grow[j] == T;
        for (uint32_t i = lane; i < nreal; i += 64) in_out += od[i] == T;
        in_row = wave_sum_u32(in_row);
        in_out = wave_sum_u32(in_out);
        tie |= in_row > in_out;
    }
    if (!__ballot(tie)) return;
    bool enters = true;  // every centroid beats the initial entries (utils.cpp:478: "if (disij < simi[0])")
    for (uint32_t j = lane; j < nlist; j += 64) {
        const float v = grow[j];
        row[j] = v;
        enters &= hcmp<IsMax>(neutral, v);
    }
    for (uint32_t i = lane; i < nprobe + 2; i += 64) h[i] = HeapEnt{neutral, 0xffffffffu};  // heap_heapify, no input
    wave_sync();
    if (lane == 0) atomicAdd(nrows, 1ull);
    if (nprobe == nlist && (nlist & (nlist - 1)) == 0 && nlist >= 64 && !__ballot(!enters)) {
        floyd_fill_pow2<IsMax>(h, row, nlist, (int)lane);
        uint32_t* out_id = reinterpret_cast<uint32_t*>(row);  // the distances are not needed any more
        heapsort_pipelined<IsMax>(h, nlist, out_id, (int)lane);
        // same distances in the same places, only centroid numbers inside runs of equal distances move
        for (uint32_t i = lane; i < nout && i < nprobe; i += 64) ok[i] = (int64_t)out_id[i];
        return;
    }
    uint32_t nvalid = 0;
    if (lane == 0) {
        const uint32_t k = nprobe;
        for (uint32_t j = 0; j < nlist; j++) {
            const float dj = row[j];
            if (hcmp<IsMax>(h[1].v, dj)) {
                lds_heap_pop<IsMax>(h, k);
                lds_heap_push<IsMax>(h, k, dj, j);
            }
        }
        // heap_reorder: k pops, each popped entry stored behind the shrinking heap; entries without an id are dropped
        uint32_t ii = 0;
        for (uint32_t i = 0; i < k; i++) {
            const HeapEnt top = h[1];
            lds_heap_pop<IsMax>(h, k - i);
            h[k - ii] = top;
            if (top.id != 0xffffffffu) ii++;
        }
        nvalid = ii;
    }
    wave_sync();
    nvalid = (uint32_t)__builtin_amdgcn_readfirstlane((int)nvalid);
    for (uint32_t i = lane; i < nout && i < nprobe; i += 64) {
        const bool real = i < nvalid;
        const HeapEnt e = real ? h[nprobe - nvalid + 1 + i] : HeapEnt{neutral, 0xffffffffu};
        od[i] = e.v;
        ok[i] = real ? (int64_t)e.id : -1;
    }
}

__global__ __launch_bounds__(64) void first_tie_kernel(const float* sorted_dis, uint32_t stride, uint32_t nreal, uint32_t* out) {
    const float* od = sorted_dis + (size_t)blockIdx.x * stride;
    uint32_t first = 0xffffffffu;
    for (uint32_t i = threadIdx.x; i + 1 < nreal; i += 64)
        if (od[i] == od[i + 1] && i < first) first = i;
    for (int off = 32; off; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)first, off);
        first = o < first ? o : first;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = first;
}

void launch_first_tie(const float* sorted_dis, uint32_t nq, uint32_t stride, uint32_t nreal, uint32_t* out, hipStream_t s) {
    if (nq) LAUNCH(first_tie_kernel, dim3(nq), dim3(64), 0, s, sorted_dis, stride, nreal, out);
}

size_t heap_tie_order_lds(uint32_t nlist, uint32_t nprobe) { return (size_t)(nprobe + 2) * 8 + (size_t)nlist * 4; }

// nout: leading entries of each ranking the caller reads (<= nprobe); rows without a run of equal distances there are left
// as they are.  Returns false (nothing launched) when heap and row do not fit one workgroup's LDS.
bool launch_heap_tie_order(const float* dis, uint32_t nq, uint32_t nlist, uint32_t nprobe, uint32_t nout, int metric, float* out_dis,
                           int64_t* out_keys, unsigned long long* nrows, hipStream_t s) {
    if (nq == 0) return true;
    const size_t shmem = heap_tie_order_lds(nlist, nprobe);
    if (shmem > 160 * 1024) return false;
    if (metric == METRIC_L2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(heap_tie_order_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        LAUNCH(heap_tie_order_kernel<true>, dim3(nq), dim3(64), shmem, s, dis, nlist, nprobe, nout, out_dis, out_keys, nrows);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(heap_tie_order_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        LAUNCH(heap_tie_order_kernel<false>, dim3(nq), dim3(64), shmem, s, dis, nlist, nprobe, nout, out_dis, out_keys, nrows);
    }
    return true;
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
    LAUNCH(pack_upper_kernel, dim3((nlist + 255) / 256, nlist), dim3(256), 0, s, full, nlist, out);
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
// per-query state of a search in one launch: empty heaps (neutral value, id -1), thresholds, zeroed stage / nscan / done /
// stop-rule state, counters and error word
__global__ __launch_bounds__(256) void init_state_kernel(InitStateArgs a) {
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t nk = a.n * a.k;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nk; i += stride) {
        a.heap_val[i] = a.neutral;
        a.heap_ref[i] = -1;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
        a.thr[i] = a.neutral;
        a.stage[i] = 0;
        a.nscan[i] = 0;
        a.done[i] = 0;
        a.pre_val[i] = 0.f;
        a.stoped[i] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) a.stats[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 4) *a.error = 0;
}

void launch_init_state(const InitStateArgs& a, hipStream_t s) {
    const size_t work = std::max<size_t>(a.n * a.k, 1);
    const unsigned grid = (unsigned)std::min<size_t>((work + 255) / 256, 4096);
    LAUNCH(init_state_kernel, dim3(grid), dim3(256), 0, s, a);
}

void launch_fill_f32(float* p, size_t n, float v, hipStream_t s) {
    if (n) LAUNCH(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}
void launch_fill_i64(int64_t* p, size_t n, int64_t v, hipStream_t s) {
    if (n) LAUNCH(fill_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}
}  // namespace amdivf
