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
#include "ivf_dev.h"

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
#ifndef AUNCEL_SCAN_PF
#define AUNCEL_SCAN_PF 2   // chunks fetched ahead of the compute (register staging)
#endif
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int LDS_ROW = SCAN_DC + 4;                 // dwords per staged row (pad = one 16-B slot)

// QG = query groups per workgroup (1, 2, 4 or 8): QG groups of 8 queries x VG blocks of 128 vectors, one wave
// each.  QG 8: 8 waves share one 128-vector tile (64 queries per pass over the list); QG 4: 4 waves, 128 vectors;
// QG 2: 4 waves over 256 vectors; QG 1: one wave, 128 vectors.  The staged tile (and the LDS footprint) follows.
//
// ARITH selects the arithmetic (the result is the same fp32 number either way):
//   0  the reference's SSE order: four running sums over elements 4i+l, separate multiply and add
//   1  the same order with fma, legal when operands are small integers (see IntRange in the engine)
// (uint8-valued data takes scan_mfma_kernel below instead: exact integer contraction on the i8 matrix cores)
// QG 1 (at most 8 queries on the list): ONE wave per workgroup over 128 vectors -- four waves over 512 vectors left half of
// them staging and waiting at barriers for nothing on lists shorter than that (cfg 5: 244 vectors a list, 7 queries each; the
// dense round of its fp32 search ran at 12 % of the vector ALU's rate), and nothing is shared between waves at QG 1.
template <int QG> struct ScanShape {
    static constexpr int NT = QG == 8 ? 512 : QG == 1 ? 64 : 256;
    static constexpr int vg = QG >= 4 || QG == 1 ? 1 : 4 / QG;
    static constexpr int tile_vecs = vg * SCAN_WAVE_VECS;
    static constexpr int lds_floats = tile_vecs * LDS_ROW;
};

// What a wave does with its finished sums (both scan kernels): distance rows, and in threshold mode the masks.
template <int METRIC>
__device__ __forceinline__ void scan_tile_store(const ScanArgs& a, const ScanItem& it, f2 (&acc)[SCAN_RQ][SCAN_RV][2], bool has_queries, int qgi, int vgi,
                                                int lane, unsigned long long e_row, float e_thr) {
    const bool masked = a.thr != nullptr;
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

    const float* tile_base = a.codes + (size_t)(it.vec_base & SCAN_VB_MASK) * (size_t)d;
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
    constexpr int PF = AUNCEL_SCAN_PF;
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

#pragma unroll
    for (int i = 0; i < PF; i++) fetch(i * SCAN_DC, pre[i]);
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

    scan_tile_store<METRIC>(a, it, acc, has_queries, qgi, vgi, lane, e_row, e_thr);
}

template <int METRIC, int QG, int ARITH>
__global__ __launch_bounds__(ScanShape<QG>::NT) void scan_tiles_kernel(ScanArgs a) {
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

// ---------------------------------------------------------------------------------------------
// The same tiles from the lane-ordered copy of the lists (ScanArgs::lanes).  With rows of d floats a staged chunk of a tile is 64
// bytes out of every row: at d = 960 that is half a cache line from each of 128 rows 3840 bytes apart, fetched again for the other
// half a chunk later -- the dense round of cfg 5 moved 2x its bytes and ran at a quarter of the kernel's own rate.  Here a wave's
// operand of one 4-dimension step is ONE contiguous KiB per 64-vector block (lane l: vector l), consecutive steps are consecutive
// KiB: no LDS, no barrier, nothing shared between the waves of a workgroup but the L2 lines they both read.  A wave keeps
// SCAN_LANE_AHEAD steps of its two blocks in flight in registers.  Arithmetic, operand order and the epilogue are scan_tile_one's.
#ifndef AUNCEL_SCAN_LANE_AHEAD
#define AUNCEL_SCAN_LANE_AHEAD 6
#endif
constexpr int SCAN_LANE_AHEAD = AUNCEL_SCAN_LANE_AHEAD;
#ifndef AUNCEL_LANES_NT
#define AUNCEL_LANES_NT 2   // streaming hint on the block loads -- 1: always; 2: only where one wave reads the block (QG 1: the other shapes'
                            // waves find it in L2); 0: never.  Measured the same within a run's spread (round 5)
#endif

template <int METRIC, int QG, int ARITH>
__device__ __forceinline__ void scan_lanes_one(const ScanArgs& a, const ScanItem it) {
    constexpr bool FUSED = ARITH == 1;
    constexpr int qg = QG;
    constexpr int D = SCAN_LANE_AHEAD;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qgi = wave & (qg - 1);
    const int vgi = wave / qg;
    const int nsteps = a.d >> 2;
    const bool has_queries = (uint32_t)(qgi * SCAN_RQ) < it.npair;
    if (!has_queries) return;  // (nothing to wait for: the waves of a workgroup share no staging)
    const uint32_t v0 = (uint32_t)vgi * SCAN_WAVE_VECS;  // the wave's first vector inside the tile
    if (v0 >= it.nvec) return;
    const bool two = v0 + 64u < it.nvec;                 // its second block holds vectors of the tile

    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef const v4f __attribute__((address_space(4)))* const_f4p;
    typedef const v4f __attribute__((address_space(1)))* glob_f4p;
    const const_f4p qtile = (const_f4p)(uintptr_t)(a.qtile + (size_t)(it.qgroup + qgi) * (size_t)a.d * SCAN_RQ);
    auto qload = [](const_f4p p) {
        const v4f t = *p;
        return make_float4(t.x, t.y, t.z, t.w);
    };
    const uint64_t blk = (it.vec_base >> SCAN_VB_BITS) + (uint64_t)(v0 >> 6);
    const glob_f4p pa = (glob_f4p)(uintptr_t)(a.lanes + (blk * (uint64_t)nsteps) * 256u) + lane;  // piece s: pa[s * 64]
    const glob_f4p pb = pa + (size_t)nsteps * 64;

    f2 acc[SCAN_RQ][SCAN_RV][2];
#pragma unroll
    for (int r = 0; r < SCAN_RQ; r++)
#pragma unroll
        for (int v = 0; v < SCAN_RV; v++) acc[r][v][0] = acc[r][v][1] = f2{0.f, 0.f};
    static_assert(SCAN_RV == 2, "two blocks a wave");

    unsigned long long e_row = 0;
    float e_thr = 0.f;
    {
        const uint32_t local = (uint32_t)(qgi * SCAN_RQ + lane);
        if (lane < SCAN_RQ && local < it.npair) {
            e_row = a.pair_out[it.pair_begin + local] + it.vec_off;
            if (a.thr != nullptr) e_thr = a.thr[a.pair_query[it.pair_begin + local]];
        }
    }

    constexpr bool STREAM = AUNCEL_LANES_NT == 1 || (AUNCEL_LANES_NT == 2 && QG == 1);
#define LANE_LOAD(p) (STREAM ? __builtin_nontemporal_load(p) : *(p))
    v4f ring[D][SCAN_RV];
#pragma unroll
    for (int j = 0; j < D; j++) {
        ring[j][0] = ring[j][1] = v4f{0.f, 0.f, 0.f, 0.f};
        if (j < nsteps) {
            ring[j][0] = LANE_LOAD(pa + (size_t)j * 64);
            if (two) ring[j][1] = LANE_LOAD(pb + (size_t)j * 64);
        }
    }
    float4 qn[SCAN_RQ];
#pragma unroll
    for (int r = 0; r < SCAN_RQ; r++) qn[r] = qload(qtile + r);
    for (int s0 = 0; s0 < nsteps; s0 += D) {
#pragma unroll
        for (int j = 0; j < D; j++) {
            const int s = s0 + j;
            if (s >= nsteps) break;  // (wave-uniform)
            const v4f ya4 = ring[j][0], yb4 = ring[j][1];
            if (s + D < nsteps) {
                ring[j][0] = LANE_LOAD(pa + (size_t)(s + D) * 64);
                if (two) ring[j][1] = LANE_LOAD(pb + (size_t)(s + D) * 64);
            }
            float4 qc[SCAN_RQ];
#pragma unroll
            for (int r = 0; r < SCAN_RQ; r++) qc[r] = qn[r];
            if (s + 1 < nsteps) {
#pragma unroll
                for (int r = 0; r < SCAN_RQ; r++) qn[r] = qload(qtile + (size_t)(s + 1) * SCAN_RQ + r);
            }
            const f2 ya[SCAN_RV] = {f2{ya4.x, ya4.y}, f2{yb4.x, yb4.y}};  // elements (0,1) of the step: sums 0,1
            const f2 yb[SCAN_RV] = {f2{ya4.z, ya4.w}, f2{yb4.z, yb4.w}};  // elements (2,3): sums 2,3
#pragma unroll
            for (int r = 0; r < SCAN_RQ; r++) {
                const f2 qa = f2{qc[r].x, qc[r].y}, qb = f2{qc[r].z, qc[r].w};
#pragma unroll
                for (int v = 0; v < SCAN_RV; v++) {
                    if (METRIC == METRIC_L2) {
                        const f2 ta = ya[v] - qa, tb = yb[v] - qb;
                        if (FUSED) {
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
#undef LANE_LOAD
    scan_tile_store<METRIC>(a, it, acc, true, qgi, vgi, lane, e_row, e_thr);
}

template <int METRIC, int QG, int ARITH>
__global__ __launch_bounds__(ScanShape<QG>::NT) void scan_lanes_kernel(ScanArgs a) {
    uint32_t n = a.nitems;
    const ScanItem* items = a.items;
    if (a.dev_counts) {
        const uint32_t n1 = a.dev_counts[CNT_QG1], n2 = a.dev_counts[CNT_QG2], n4 = a.dev_counts[CNT_QG4], n8 = a.dev_counts[CNT_QG8];
        n = QG == 1 ? n1 : QG == 2 ? n2 : QG == 4 ? n4 : n8;
        items += QG == 1 ? 0 : QG == 2 ? n1 : QG == 4 ? n1 + n2 : n1 + n2 + n4;
    }
    ItemWalk w(n, a.xcd_chunks);
    for (uint32_t i = w.cur; i < w.end; i += w.step) scan_lanes_one<METRIC, QG, ARITH>(a, items[i]);
}

// one wave per (64-vector block, run of pieces): lane l copies vector 64 b + l's elements, 16 bytes a piece
__global__ __launch_bounds__(64) void lanes_from_f32_kernel(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist,
                                                            int dpad, float* out) {
    const uint64_t blk = blockIdx.x;
    const int lane = threadIdx.x;
    uint32_t lo = 0, hi = nlist;  // largest l with block_off[l] / 2 <= blk
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((block_off[mid] >> 1) <= blk) lo = mid;
        else hi = mid;
    }
    const uint64_t pos = (blk - (block_off[lo] >> 1)) * 64 + lane, size = list_off[lo + 1] - list_off[lo];
    const bool ok = pos < size;
    const float4* src = reinterpret_cast<const float4*>(codes + (list_off[lo] + (ok ? pos : 0)) * (uint64_t)dpad);
    const int nsteps = dpad >> 2;
    float4* dst = reinterpret_cast<float4*>(out) + blk * (uint64_t)nsteps * 64 + lane;
    for (int s = 0; s < nsteps; s++) dst[(size_t)s * 64] = ok ? src[s] : make_float4(0.f, 0.f, 0.f, 0.f);
}

void launch_lanes_from_f32(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist, uint64_t nblocks64, int dpad,
                           float* out, hipStream_t s) {
    if (nblocks64 == 0) return;
    LAUNCH(lanes_from_f32_kernel, dim3((unsigned)nblocks64), dim3(64), 0, s, codes, list_off, block_off, nlist, dpad, out);
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
    const int threads = ScanShape<QG>::NT;
    const uint32_t hint = a.hint_qg[scan_qg_class(QG)];
    auto go = [&](auto kern) {
        static const unsigned per_cu = blocks_per_cu(kern, threads);
        const unsigned hinted = ((unsigned)((size_t)hint + hint / 8 + 7) / 8) * 8 + 8;  // last time's count + 12 %
        // (a shape the same round of the previous search had no item of gets one workgroup a CU: a fully resident grid that finds nothing to do
        // still has to be placed on the chip behind the other shapes' workgroups, and the round's join waits for it)
        const dim3 grid((unsigned)(a.dev_counts ? (hint ? hinted : a.hint_valid ? resident_grid(1) : resident_grid(per_cu)) : a.xcd_chunks ? ((n + 7) / 8) * 8 : n)), block(threads);
        LAUNCH(kern, grid, block, 0, s, a);
    };
    if (a.lanes) {
        if (a.metric == METRIC_L2) {
            if (a.fused) go(scan_lanes_kernel<METRIC_L2, QG, 1>);
            else go(scan_lanes_kernel<METRIC_L2, QG, 0>);
        } else {
            if (a.fused) go(scan_lanes_kernel<METRIC_IP, QG, 1>);
            else go(scan_lanes_kernel<METRIC_IP, QG, 0>);
        }
    } else if (a.metric == METRIC_L2) {
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

// ---- threshold rounds, pipelined form (round 4).  The kernel above keeps ONE list block in flight per wave (its loop top is
// `s_waitcnt vmcnt(0)`) and starts every item with three dependent round trips (item -> pair rows -> per-query operands): with
// four waves per SIMD that is 64 KB in flight per CU only while every wave happens to be waiting, and a launch streamed at 0.30
// of the HBM peak with the waves parked in s_waitcnt for half their cycles (profiles/r03_summary.md).  Here
//   * two blocks are in flight per wave at all times, across block AND item boundaries: block i + 2 is requested right behind
//     the MFMAs of block i, and past the item's end that is block 0 / 1 of the wave's NEXT item (its header came in through the
//     scalar cache two items ahead, its pair rows one item ahead), so a wave's list stream never drains between items;
//   * every load of the stream is unconditional straight-line code (a wave without a next item re-requests its own first block),
//     which lets the compiler wait with a count (the stores of an epilogue are younger than the loads it runs under);
//   * the per-query threshold is folded into the contraction: the accumulator starts at -floor(u / 2) (L2; u = |x|^2 - T made
//     even, i.e. T loosened by at most one -- the selection re-tests every marked candidate against the query's current worst
//     anyway) or -u (IP), so a register's test is ONE v_cmp against floor(|y|^2 / 2) (-cy for IP) and the 48 registers of
//     per-query operands in accumulator layout (thresholds, |x|^2, row offsets) shrink to the 16 start values; |x|^2, u and the
//     row offset of a query are read from LDS by the rare register that stores a distance.
// The mask it leaves is a superset of the exact one (never a subset): callers that COUNT mask bits (range search) keep the
// kernel above (MfmaScanArgs::exact_mask).
// DENSE: the same stream for rounds that store every distance (round 0): accumulators start at zero, |x|^2 and the row offset of a
// register's query come from LDS four registers at a time.
// Cycle accounting of the waves of the kernel below (an experiment build: AUNCEL_AMD_CXXFLAGS=-DAUNCEL_AMD_SCAN_PROF; the counters
// are printed per launch by launch_scan_mfma).  [0] waves, [1] lifetime, [2] start -> first block's operands ready, [3] waiting
// for list blocks, [4] MFMAs + epilogue of the blocks, [5] per-item prologue after the first, [6] mask stores, [7] items
#ifdef AUNCEL_AMD_SCAN_PROF
__device__ unsigned long long g_scan_prof[8];
#define SCAN_PROF_NOW() __builtin_amdgcn_s_memtime()
#else
#define SCAN_PROF_NOW() 0ull
#endif
template <int METRIC, int NKS, bool DENSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void scan_mfma_thr_kernel(MfmaScanArgs a) {
    static_assert(NKS >= 1 && NKS <= 4, "query operand resident in registers");
    __shared__ int s_init[4][32];
    __shared__ int s_cx[4][32];
    __shared__ int s_u[4][32];
    __shared__ uint32_t s_row[4][32];
    // the mask words of an item, [block][query]: a block's 32 words are 32 different rows of the mask -- stored per block that
    // is one 4-byte write to each of 32 cache lines per tile, 0.15 of a 0.56 ms launch (measured with the stores left out);
    // gathered here they leave as 16- / 8-byte pieces per query when the item ends
    constexpr uint32_t MAX_BLK = 16;  // (items of up to 512 vectors: launch_scan_mfma)
    __shared__ uint32_t s_mask[DENSE ? 1 : 4][DENSE ? 1 : MAX_BLK][32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    constexpr size_t qstride = (size_t)NKS * 32;
    constexpr size_t block_bytes = (size_t)NKS * 1024;
    [[maybe_unused]] const unsigned long long pt_start = SCAN_PROF_NOW();
    [[maybe_unused]] unsigned long long pt_wait = 0, pt_work = 0, pt_first = 0, pt_pro = 0, pt_mask = 0, pt_items = 0;
    // item headers through the scalar cache (constant address space: nothing this kernel stores aliases them)
    typedef int v8i __attribute__((ext_vector_type(8)));
    typedef const v8i __attribute__((address_space(4)))* item_cp;
    static_assert(sizeof(ScanItem) == 32, "one s_load_dwordx8 per item header");
    const item_cp items_c = (item_cp)(uintptr_t)a.items;
    struct Hdr {  // the fields of a ScanItem this kernel reads (wave-uniform: SGPRs)
        uint64_t vec_base;
        uint32_t nvec, vec_off, pair_begin, npair;
    };
    auto load_item = [&](uint32_t n) {
        const v8i t = items_c[n];
        Hdr r;
        r.vec_base = (uint64_t)(uint32_t)t[0] | ((uint64_t)(uint32_t)t[1] << 32);
        r.nvec = (uint32_t)t[2];
        r.vec_off = (uint32_t)t[3];
        r.pair_begin = (uint32_t)t[4];
        r.npair = (uint32_t)t[5];
        return r;
    };
    const uint32_t nitems = a.dev_nitems ? *a.dev_nitems : a.nitems;
    ItemWalk w((nitems + 3) >> 2, a.xcd_chunks);
    uint32_t wi = w.cur;
    if (wi >= w.end || wi * 4 + (uint32_t)wave >= nitems) return;  // (no workgroup barrier anywhere below)
    auto item_of = [&](uint32_t walk) { return walk * 4 + (uint32_t)wave; };
    auto exists = [&](uint32_t walk) { return walk < w.end && item_of(walk) < nitems; };

    v4i b0[NKS], b1[NKS];
    int cy0 = 0, cy1 = 0;
    auto fetch = [&](v4i (&b)[NKS], int& cy, uint64_t blk) {
        const uint8_t* bp = a.codes_frag + blk * block_bytes + (size_t)lane * 16;
#pragma unroll
        for (int s = 0; s < NKS; s++) b[s] = MFMA_BLOAD(reinterpret_cast<const v4i*>(bp + (size_t)s * 1024));
        cy = a.code_cy[blk * 32 + m];
    };

    Hdr cur = load_item(item_of(wi));
    Hdr nxt = load_item(exists(wi + w.step) ? item_of(wi + w.step) : item_of(wi));
    // (every load of the stream below is unconditional: a slot without a query reads entry 0 of the item and masks the result,
    // so that the compiler's wait counts stay exact)
    uint32_t pq, po;  // query row and distance row (at most 2^31 floats) of this lane's query slot
    {
        const uint32_t e = cur.pair_begin + ((uint32_t)m < cur.npair ? (uint32_t)m : 0u);
        pq = a.pair_query[e];
        po = (uint32_t)a.pair_out[e];
    }
    fetch(b0, cy0, cur.vec_base);
    fetch(b1, cy1, cur.vec_base + 1);
    uint32_t* mask32 = reinterpret_cast<uint32_t*>(a.mask);

    for (;;) {
        [[maybe_unused]] const unsigned long long pt_item = SCAN_PROF_NOW();
        const uint32_t wn = wi + w.step;
        const bool has_next = exists(wn);
        // (header two items ahead: consumed when `nxt` becomes `cur`)
        const Hdr nn = load_item(exists(wn + w.step) ? item_of(wn + w.step) : item_of(wi));
        const bool qok = (uint32_t)m < cur.npair;
        const uint32_t row = po + cur.vec_off;
        const int8_t* qp = a.queries8 + (size_t)pq * qstride + (size_t)h * (size_t)(16 * NKS);
        v4i af[NKS];
#pragma unroll
        for (int s = 0; s < NKS; s++) af[s] = *reinterpret_cast<const v4i*>(qp + 16 * s);
        const int cx = a.query_cx[pq];
        const float thr = DENSE ? 0.f : a.thr[pq];
        // pair rows of the next item (in flight for the whole of this one)
        const uint32_t en = has_next ? nxt.pair_begin + ((uint32_t)m < nxt.npair ? (uint32_t)m : 0u) : cur.pair_begin;
        const uint32_t pqn = a.pair_query[en];
        const uint32_t pon = (uint32_t)a.pair_out[en];
        if (!qok) {
#pragma unroll
            for (int s = 0; s < NKS; s++) af[s] = v4i{0, 0, 0, 0};
        }
        // start value of the accumulators: see the head comment.  Invalid query slots never pass.
        int u = 0, init = -1073741824;
        if (qok && !DENSE) {
            if (METRIC == METRIC_L2) {
                const float c = ceilf(thr);
                const int T = !(c > 0.f) ? 0 : (c >= 1073741824.f ? 1073741824 : (int)c);  // NaN -> nothing passes the selection
                u = (cx - T) & ~1;
                init = -(u >> 1);
            } else {
                const float f = floorf(thr);
                const int T = !(f < 1073741824.f) ? 0x7fffffff : (f <= -1073741824.f ? -1073741824 : (int)f);
                if (T != 0x7fffffff) {  // (else nothing passes: the start value of an invalid slot)
                    u = T - cx;
                    init = -u;
                }
            }
        }
        // (the next item's pair rows are awaited here, with this item's operands: a load left pending across the block loop
        // ages without bound there, the compiler's wait-count analysis does not converge and falls back to vmcnt(0) at the
        // loop head -- one block in flight again)
        asm volatile("" ::"v"(pqn), "v"(pon));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous item's reads of this wave's LDS rows are done
        __builtin_amdgcn_wave_barrier();
        if (h == 0) {
            s_init[wave][m] = init;
            s_cx[wave][m] = cx;
            s_u[wave][m] = u;
            s_row[wave][m] = row;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        v16i cinit;  // register 4 g + i of lane half h belongs to query 8 g + 4 h + i
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const v4i t = DENSE ? v4i{0, 0, 0, 0} : *reinterpret_cast<const v4i*>(&s_init[wave][8 * g + 4 * h]);
#pragma unroll
            for (int i = 0; i < 4; i++) cinit[4 * g + i] = t[i];
        }

        const uint32_t nblk = ((cur.nvec + 63) >> 6) * 2;  // lists are stored in pairs of blocks: a 64-candidate mask word is two of ours
#ifdef AUNCEL_AMD_SCAN_PROF
        {
            asm volatile("" ::"v"(cinit), "v"(af[0]), "v"(af[NKS - 1]));
            const unsigned long long t = SCAN_PROF_NOW();
            if (pt_items == 0) pt_first = t - pt_start;
            else pt_pro += t - pt_item;
            pt_items++;
        }
#endif
        auto step = [&](v4i (&b)[NKS], int& cyv, uint32_t i) {
            __builtin_amdgcn_sched_barrier(0);
#ifdef AUNCEL_AMD_SCAN_PROF
            const unsigned long long t_a = SCAN_PROF_NOW();
            asm volatile("" ::"v"(b[0]), "v"(b[NKS - 1]), "v"(cyv));
            const unsigned long long t_b = SCAN_PROF_NOW();
            pt_wait += t_b - t_a;
#endif
            v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], b[0], cinit, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < NKS; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s], b[s], acc, 0, 0, 0);
            // (what the epilogue needs of |y|^2 is taken before the register is requested again: the old and the new value never
            // live side by side, so the loop carries no copy -- a copy behind the load would wait for it)
            const uint32_t lv = i * 32 + m;  // position of this lane's vector in the chunk
            const bool vok = lv < cur.nvec;
            int hc = !vok ? 0x7fffffff : METRIC == METRIC_L2 ? cyv >> 1 : -cyv;
            int cy;  // (an explicit move: a plain copy would be coalesced with the register the next request writes)
            asm volatile("v_mov_b32 %0, %1" : "=v"(cy) : "v"(cyv));
            asm volatile("" : "+v"(hc));
            __builtin_amdgcn_sched_barrier(0);
            // block i + 2 of the stream: of this item, else block 0 / 1 of the next (nblk is even), else anything valid
            const uint32_t in = i + 2;
            const uint64_t nb = in < nblk ? cur.vec_base + in : has_next ? nxt.vec_base + (in - nblk) : cur.vec_base;
            fetch(b, cyv, nb);
            __builtin_amdgcn_sched_barrier(0);
            if (DENSE) {
                // every distance of the tile: register 4 g + i = (query 8 g + 4 h + i, vector lv): two 128-byte runs per register
                static_for(std::make_integer_sequence<int, 4>{}, [&](auto G) {
                    constexpr int g = decltype(G)::value;
                    const v4i c4 = *reinterpret_cast<const v4i*>(&s_cx[wave][8 * g + 4 * h]);
                    const v4i r4 = *reinterpret_cast<const v4i*>(&s_row[wave][8 * g + 4 * h]);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int t = METRIC == METRIC_L2 ? 2 * acc[4 * g + i] - cy : acc[4 * g + i] + cy;
                        const int res = METRIC == METRIC_L2 ? c4[i] - t : c4[i] + t;
                        uint32_t off = (uint32_t)r4[i] + lv;
                        asm volatile("" : "+v"(off));  // 64-bit addresses are formed here, not hoisted as register pairs
                        if (vok && (uint32_t)(8 * g + 4 * h + i) < cur.npair) a.dist[off] = (float)res;
                    }
                });
#ifdef AUNCEL_AMD_SCAN_PROF
                pt_work += SCAN_PROF_NOW() - t_b;
#endif
                return;
            }
            // One v_cmp per register is the ballot of its 64 candidates.  Everything else -- the two lane writes that park the
            // mask words, the store of a distance -- happens only for a register that HAS a candidate (a few in a hundred), behind
            // a scalar branch on the ballot: the epilogue of a tile was 72 vector + 129 scalar instructions when every register
            // wrote its lanes and walked an exec-masked branch (PMC, profiles/r04a_summary.md: the waves of this kernel were
            // issuing or stalled on issue for 2/3 of their cycles), it is 16 + 32 on the common path now.  Ballots are taken eight
            // registers at a time (16 SGPRs) ahead of their branches, so that the compares issue back to back.
            int word = 0;  // lane q (< 32) collects the 32-candidate mask word of query q
            static_for(std::make_integer_sequence<int, 2>{}, [&](auto H) {
                constexpr int r0 = decltype(H)::value * 8;
                unsigned long long bal[8];
#pragma unroll
                for (int r = 0; r < 8; r++) bal[r] = __ballot(acc[r0 + r] > hc);
                static_for(std::make_integer_sequence<int, 8>{}, [&](auto R) {
                    constexpr int reg = r0 + decltype(R)::value;
                    constexpr int q0 = (reg & 3) + 8 * (reg >> 2);  // query of lane half 0; half 1: q0 + 4
                    const unsigned long long b = bal[reg - r0];
                    if (b != 0) {  // (wave-uniform: a scalar compare and branch)
                        writelane_c<q0>(word, (uint32_t)b);
                        writelane_c<q0 + 4>(word, (uint32_t)(b >> 32));
                        if (acc[reg] > hc && !(a.debug & 2)) {
                            const int cq = s_cx[wave][q0 + 4 * h], uq = s_u[wave][q0 + 4 * h];
                            const uint32_t off = s_row[wave][q0 + 4 * h] + lv;
                            // the contraction itself: acc = x.y - floor(u / 2) (L2) | x.y - u (IP)
                            const int t = METRIC == METRIC_L2 ? 2 * (acc[reg] + (uq >> 1)) - cy : acc[reg] + uq + cy;
                            a.dist[off] = (float)(METRIC == METRIC_L2 ? cq - t : cq + t);
                        }
                    }
                });
            });
            if (lane < 32) s_mask[DENSE ? 0 : wave][DENSE ? 0 : i][lane] = (uint32_t)word;
#ifdef AUNCEL_AMD_SCAN_PROF
            asm volatile("" ::"v"(word));
            pt_work += SCAN_PROF_NOW() - t_b;
#endif
        };
        for (uint32_t i = 0; i < nblk; i += 2) {
            step(b0, cy0, i);
            step(b1, cy1, i + 1);
        }
        [[maybe_unused]] const unsigned long long pt_m0 = SCAN_PROF_NOW();
        if (!DENSE) {
            // rows start on multiples of 64 floats, chunks on multiples of 64 vectors: word index = (row + position) / 32; the
            // item's words of a query are consecutive there (nblk of them, an even number: 8-byte pieces at least)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < 32 && qok && !(a.debug & 1)) {
                uint32_t* dst = mask32 + (row >> 5);
                for (uint32_t i = 0; i < nblk; i += 4) {
                    if (i + 4 <= nblk) {
                        typedef int v4i8 __attribute__((ext_vector_type(4), aligned(8)));  // (rows start on multiples of 64 floats)
                        v4i8 t;
#pragma unroll
                        for (int c = 0; c < 4; c++) t[c] = (int)s_mask[DENSE ? 0 : wave][DENSE ? 0 : i + c][lane];
                        *reinterpret_cast<v4i8*>(dst + i) = t;
                    } else {
                        typedef int v2i __attribute__((ext_vector_type(2)));
                        v2i t;
                        t[0] = (int)s_mask[DENSE ? 0 : wave][DENSE ? 0 : i][lane];
                        t[1] = (int)s_mask[DENSE ? 0 : wave][DENSE ? 0 : i + 1][lane];
                        *reinterpret_cast<v2i*>(dst + i) = t;
                    }
                }
            }
        }
#ifdef AUNCEL_AMD_SCAN_PROF
        pt_mask += SCAN_PROF_NOW() - pt_m0;
#endif
        if (!has_next) break;
        cur = nxt;
        nxt = nn;
        pq = pqn;
        po = pon;
        wi = wn;
    }
#ifdef AUNCEL_AMD_SCAN_PROF
    if (lane == 0) {
        atomicAdd(&g_scan_prof[0], 1ull);
        atomicAdd(&g_scan_prof[1], SCAN_PROF_NOW() - pt_start);
        atomicAdd(&g_scan_prof[2], pt_first);
        atomicAdd(&g_scan_prof[3], pt_wait);
        atomicAdd(&g_scan_prof[4], pt_work);
        atomicAdd(&g_scan_prof[5], pt_pro);
        atomicAdd(&g_scan_prof[6], pt_mask);
        atomicAdd(&g_scan_prof[7], pt_items);
    }
#endif
}

// ---- threshold rounds, TWO query blocks per wave.  A round in which a list is probed by more than 32 queries runs one item per
// (chunk, query block) above, and every one of them fetches the chunk: on the bench workload 3.1 query blocks per chunk, 4.2 GB
// through L2 -> L1 for 1.3 GB of lists, and a CU's stream is served at the HBM rate for the first fetch of a chunk and at the L2
// rate for the others, one after the other (0.38 ms = 0.22 + 0.16: profiles/r04_summary.md).  Here an item carries up to 64
// queries (the planner's mfma_qblock): the list block that is in registers is contracted with both query blocks before it is let
// go, so a chunk is fetched once per 64 queries.  The start values of the accumulators come straight from LDS into the
// accumulator registers (no second copy in registers); everything else is the kernel above with two of each.
template <int METRIC, int NKS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void scan_mfma_pair_kernel(MfmaScanArgs a) {
    static_assert(NKS >= 1 && NKS <= 4, "query operands resident in registers");
    constexpr uint32_t MAX_BLK = 16;  // (items of up to 512 vectors: launch_scan_mfma)
    __shared__ int s_init[4][64];
    __shared__ int s_cx[4][64];
    __shared__ int s_u[4][64];
    __shared__ uint32_t s_row[4][64];
    __shared__ uint32_t s_mask[4][MAX_BLK][64];  // [block][query slot]: leave as 16- / 8-byte pieces per query when the item ends
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    constexpr size_t qstride = (size_t)NKS * 32;
    constexpr size_t block_bytes = (size_t)NKS * 1024;
    typedef int v8i __attribute__((ext_vector_type(8)));
    typedef const v8i __attribute__((address_space(4)))* item_cp;
    const item_cp items_c = (item_cp)(uintptr_t)a.items;
    struct Hdr {
        uint64_t vec_base;
        uint32_t nvec, vec_off, pair_begin, npair;
    };
    auto load_item = [&](uint32_t n) {
        const v8i t = items_c[n];
        Hdr r;
        r.vec_base = (uint64_t)(uint32_t)t[0] | ((uint64_t)(uint32_t)t[1] << 32);
        r.nvec = (uint32_t)t[2];
        r.vec_off = (uint32_t)t[3];
        r.pair_begin = (uint32_t)t[4];
        r.npair = (uint32_t)t[5];
        return r;
    };
    const uint32_t nitems = a.dev_nitems ? *a.dev_nitems : a.nitems;
    ItemWalk w((nitems + 3) >> 2, a.xcd_chunks);
    uint32_t wi = w.cur;
    if (wi >= w.end || wi * 4 + (uint32_t)wave >= nitems) return;  // (no workgroup barrier anywhere below)
    auto item_of = [&](uint32_t walk) { return walk * 4 + (uint32_t)wave; };
    auto exists = [&](uint32_t walk) { return walk < w.end && item_of(walk) < nitems; };

    v4i b0[NKS], b1[NKS];
    int cy0 = 0, cy1 = 0;
    auto fetch = [&](v4i (&b)[NKS], int& cy, uint64_t blk) {
        const uint8_t* bp = a.codes_frag + blk * block_bytes + (size_t)lane * 16;
#pragma unroll
        for (int s = 0; s < NKS; s++) b[s] = MFMA_BLOAD(reinterpret_cast<const v4i*>(bp + (size_t)s * 1024));
        cy = a.code_cy[blk * 32 + m];
    };

    Hdr cur = load_item(item_of(wi));
    Hdr nxt = load_item(exists(wi + w.step) ? item_of(wi + w.step) : item_of(wi));
    uint32_t pq, po;  // query row and distance row of query slot `lane` (every load of the stream is unconditional: see above)
    {
        const uint32_t e = cur.pair_begin + ((uint32_t)lane < cur.npair ? (uint32_t)lane : 0u);
        pq = a.pair_query[e];
        po = (uint32_t)a.pair_out[e];
    }
    fetch(b0, cy0, cur.vec_base);
    fetch(b1, cy1, cur.vec_base + 1);
    uint32_t* mask32 = reinterpret_cast<uint32_t*>(a.mask);

    for (;;) {
        const uint32_t wn = wi + w.step;
        const bool has_next = exists(wn);
        const Hdr nn = load_item(exists(wn + w.step) ? item_of(wn + w.step) : item_of(wi));
        const bool sok = (uint32_t)lane < cur.npair;
        const bool two = cur.npair > 32;  // (wave-uniform)
        const uint32_t row = po + cur.vec_off;
        // the A operands: lane (m, h) of query block g holds the bytes of query slot 32 g + m
        const uint32_t pq0 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * m, (int)pq);
        const uint32_t pq1 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (32 + m), (int)pq);
        const bool qok0 = (uint32_t)m < cur.npair, qok1 = (uint32_t)(32 + m) < cur.npair;
        v4i af0[NKS], af1[NKS];
        {
            const int8_t* q0p = a.queries8 + (size_t)pq0 * qstride + (size_t)h * (size_t)(16 * NKS);
            const int8_t* q1p = a.queries8 + (size_t)pq1 * qstride + (size_t)h * (size_t)(16 * NKS);
#pragma unroll
            for (int s = 0; s < NKS; s++) af0[s] = *reinterpret_cast<const v4i*>(q0p + 16 * s);
#pragma unroll
            for (int s = 0; s < NKS; s++) af1[s] = *reinterpret_cast<const v4i*>(q1p + 16 * s);
        }
        const int cx = a.query_cx[pq];
        const float thr = a.thr[pq];
        const uint32_t en = has_next ? nxt.pair_begin + ((uint32_t)lane < nxt.npair ? (uint32_t)lane : 0u) : cur.pair_begin;
        const uint32_t pqn = a.pair_query[en];
        const uint32_t pon = (uint32_t)a.pair_out[en];
        if (!qok0) {
#pragma unroll
            for (int s = 0; s < NKS; s++) af0[s] = v4i{0, 0, 0, 0};
        }
        if (!qok1) {
#pragma unroll
            for (int s = 0; s < NKS; s++) af1[s] = v4i{0, 0, 0, 0};
        }
        int u = 0, init = -1073741824;  // (start values: scan_mfma_thr_kernel; invalid query slots never pass)
        if (sok) {
            if (METRIC == METRIC_L2) {
                const float c = ceilf(thr);
                const int T = !(c > 0.f) ? 0 : (c >= 1073741824.f ? 1073741824 : (int)c);
                u = (cx - T) & ~1;
                init = -(u >> 1);
            } else {
                const float f = floorf(thr);
                const int T = !(f < 1073741824.f) ? 0x7fffffff : (f <= -1073741824.f ? -1073741824 : (int)f);
                if (T != 0x7fffffff) {
                    u = T - cx;
                    init = -u;
                }
            }
        }
        asm volatile("" ::"v"(pqn), "v"(pon));  // (awaited here, not across the block loop: see scan_mfma_thr_kernel)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous item's reads of this wave's LDS rows are done
        __builtin_amdgcn_wave_barrier();
        s_init[wave][lane] = init;
        s_cx[wave][lane] = cx;
        s_u[wave][lane] = u;
        s_row[wave][lane] = row;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const uint32_t nblk = ((cur.nvec + 63) >> 6) * 2;
        auto step = [&](v4i (&b)[NKS], int& cyv, uint32_t i) {
            __builtin_amdgcn_sched_barrier(0);
            // register 4 g + i of lane half h belongs to query 8 g + 4 h + i of its block: the start values, LDS -> accumulator
            v16i acc0, acc1;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const v4i t0 = *reinterpret_cast<const v4i*>(&s_init[wave][8 * g + 4 * h]);
                const v4i t1 = *reinterpret_cast<const v4i*>(&s_init[wave][32 + 8 * g + 4 * h]);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    acc0[4 * g + c] = t0[c];
                    acc1[4 * g + c] = t1[c];
                }
            }
            if (!(a.debug & 8)) {
#pragma unroll
                for (int s = 0; s < NKS; s++) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af0[s], b[s], acc0, 0, 0, 0);
                if (two) {
#pragma unroll
                    for (int s = 0; s < NKS; s++) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af1[s], b[s], acc1, 0, 0, 0);
                }
            } else {
                asm volatile("" ::"v"(b[0]), "v"(b[NKS - 1]));  // (the blocks are still awaited)
            }
            const uint32_t lv = i * 32 + m;
            const bool vok = lv < cur.nvec;
            int hc = !vok ? 0x7fffffff : METRIC == METRIC_L2 ? cyv >> 1 : -cyv;
            int cy;
            asm volatile("v_mov_b32 %0, %1" : "=v"(cy) : "v"(cyv));
            asm volatile("" : "+v"(hc));
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t in = i + 2;
            const uint64_t nb = in < nblk ? cur.vec_base + in : has_next ? nxt.vec_base + (in - nblk) : cur.vec_base;
            fetch(b, cyv, nb);
            __builtin_amdgcn_sched_barrier(0);
            int word = 0;  // lane 32 g + q collects the 32-candidate mask word of query q of block g
            auto epilogue = [&](const v16i& acc, auto QB) {
                constexpr int qb = decltype(QB)::value;
                static_for(std::make_integer_sequence<int, 2>{}, [&](auto H) {
                    constexpr int r0 = decltype(H)::value * 8;
                    // eight registers at a time: their largest value against the bound first (four v_max3 / v_max and one compare:
                    // most groups of eight hold no candidate, and sixteen compare-and-branch pairs per tile, the branch taken,
                    // were most of a tile's issue time)
                    int mx = max(max(acc[r0], acc[r0 + 1]), acc[r0 + 2]);
                    mx = max(max(mx, acc[r0 + 3]), acc[r0 + 4]);
                    mx = max(max(mx, acc[r0 + 5]), acc[r0 + 6]);
                    mx = max(mx, acc[r0 + 7]);
                    if (__ballot(mx > hc) == 0) return;
                    unsigned long long bal[8];
#pragma unroll
                    for (int r = 0; r < 8; r++) bal[r] = __ballot(acc[r0 + r] > hc);
                    static_for(std::make_integer_sequence<int, 8>{}, [&](auto R) {
                        constexpr int reg = r0 + decltype(R)::value;
                        constexpr int q0 = qb + (reg & 3) + 8 * (reg >> 2);  // query slot of lane half 0; half 1: q0 + 4
                        const unsigned long long bb = bal[reg - r0];
                        if (bb != 0) {  // (wave-uniform: a scalar compare and branch)
                            writelane_c<q0>(word, (uint32_t)bb);
                            writelane_c<q0 + 4>(word, (uint32_t)(bb >> 32));
                            if (acc[reg] > hc && !(a.debug & 2)) {
                                const int cq = s_cx[wave][q0 + 4 * h], uq = s_u[wave][q0 + 4 * h];
                                const uint32_t off = s_row[wave][q0 + 4 * h] + lv;
                                const int t = METRIC == METRIC_L2 ? 2 * (acc[reg] + (uq >> 1)) - cy : acc[reg] + uq + cy;
                                a.dist[off] = (float)(METRIC == METRIC_L2 ? cq - t : cq + t);
                            }
                        }
                    });
                });
            };
            if (!(a.debug & 4)) {  // (timing experiments: 4 = no epilogue, 8 = no contraction; results are wrong)
                epilogue(acc0, std::integral_constant<int, 0>{});
                if (two) epilogue(acc1, std::integral_constant<int, 32>{});
            }
            s_mask[wave][i][lane] = (uint32_t)word;
        };
        for (uint32_t i = 0; i < nblk; i += 2) {
            step(b0, cy0, i);
            step(b1, cy1, i + 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (sok && !(a.debug & 1)) {
            uint32_t* dst = mask32 + (row >> 5);
            for (uint32_t i = 0; i < nblk; i += 4) {
                if (i + 4 <= nblk) {
                    typedef int v4i8 __attribute__((ext_vector_type(4), aligned(8)));  // (rows start on multiples of 64 floats)
                    v4i8 t;
#pragma unroll
                    for (int c = 0; c < 4; c++) t[c] = (int)s_mask[wave][i + c][lane];
                    *reinterpret_cast<v4i8*>(dst + i) = t;
                } else {
                    typedef int v2i __attribute__((ext_vector_type(2)));
                    v2i t;
                    t[0] = (int)s_mask[wave][i][lane];
                    t[1] = (int)s_mask[wave][i + 1][lane];
                    *reinterpret_cast<v2i*>(dst + i) = t;
                }
            }
        }
        if (!has_next) break;
        cur = nxt;
        nxt = nn;
        pq = pqn;
        po = pon;
        wi = wn;
    }
}

uint32_t mfma_chunk() {
    static const uint32_t v = [] {
        const char* e = getenv("AUNCEL_AMD_MFMA_CHUNK");
        const long x = e ? atol(e) : 0;
        return x >= 64 ? (uint32_t)((x + 63) / 64 * 64) : 256u;
    }();
    return v;
}

// ... of a threshold round: an item is longer there (512), where nothing is stored per candidate and the per-item prologue and
// the scattered stores of the mask words weigh more (0.450 -> 0.407 ms per launch on the bench workload; the dense round
// prefers 256: 0.529 vs 0.549)
uint32_t mfma_chunk_thr() {
    static const uint32_t v = [] {
        const char* e = getenv("AUNCEL_AMD_MFMA_CHUNK_THR");
        const long x = e ? atol(e) : 0;
        return x >= 64 ? (uint32_t)((x + 63) / 64 * 64) : (getenv("AUNCEL_AMD_MFMA_CHUNK") ? mfma_chunk() : 512u);
    }();
    return v;
}

// queries per item of a threshold round over byte codes: 64 where scan_mfma_pair_kernel runs it (bit 2 of `pipelined`; masks that
// may be supersets; the query operands of two blocks fit the registers), else one tile's 32
uint32_t mfma_thr_qblock(int d, int pipelined, bool exact_mask) {
    const int ks = (int)mfma_ksteps(d);
    return (pipelined & 4) && !exact_mask && ks >= 1 && ks <= 4 && mfma_chunk() <= 512 && mfma_chunk_thr() <= 512 ? 2 * MFMA_QBLOCK : MFMA_QBLOCK;
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
#ifdef AUNCEL_AMD_SCAN_PROF
        if (getenv("AUNCEL_AMD_SCAN_PROF")) {
            unsigned long long c[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_scan_prof), sizeof c);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_scan_prof), z, sizeof z);
            if (c[0])
                fprintf(stderr, "[scan prof] %s grid %u waves %llu items %llu | per wave (cycles of s_memtime): life %.0f first %.0f wait %.0f work %.0f "
                        "item-prologue %.0f mask %.0f\n", masked ? "thr" : "dense", grid.x, c[0], c[7], (double)c[1] / c[0], (double)c[2] / c[0],
                        (double)c[3] / c[0], (double)c[4] / c[0], (double)c[5] / c[0], (double)c[6] / c[0]);
        }
#endif
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
    // ... with up to 64 queries per item (the planner was told so: mfma_thr_qblock)
    if (masked && mfma_thr_qblock(a.d, a.pipelined, a.exact_mask != 0) == 2 * MFMA_QBLOCK) {
        if (a.metric == METRIC_L2) {
            switch (ks) {
                case 1: return go(scan_mfma_pair_kernel<METRIC_L2, 1>);
                case 2: return go(scan_mfma_pair_kernel<METRIC_L2, 2>);
                case 3: return go(scan_mfma_pair_kernel<METRIC_L2, 3>);
                default: return go(scan_mfma_pair_kernel<METRIC_L2, 4>);
            }
        } else {
            switch (ks) {
                case 1: return go(scan_mfma_pair_kernel<METRIC_IP, 1>);
                case 2: return go(scan_mfma_pair_kernel<METRIC_IP, 2>);
                case 3: return go(scan_mfma_pair_kernel<METRIC_IP, 3>);
                default: return go(scan_mfma_pair_kernel<METRIC_IP, 4>);
            }
        }
    }
    // threshold rounds whose mask may be a superset of the exact one: the pipelined form (scan_mfma_thr_kernel)
    if ((a.pipelined & (masked ? 2 : 1)) && !(masked && a.exact_mask) && ks >= 1 && ks <= 4 && mfma_chunk() <= 512 && mfma_chunk_thr() <= 512) {
        auto pick_thr = [&](auto metric, auto dense) {
            constexpr int M = decltype(metric)::value;
            constexpr bool D = decltype(dense)::value;
            switch (ks) {
                case 1: return go(scan_mfma_thr_kernel<M, 1, D>);
                case 2: return go(scan_mfma_thr_kernel<M, 2, D>);
                case 3: return go(scan_mfma_thr_kernel<M, 3, D>);
                default: return go(scan_mfma_thr_kernel<M, 4, D>);
            }
        };
        if (a.metric == METRIC_L2) {
            if (masked) pick_thr(std::integral_constant<int, METRIC_L2>{}, std::false_type{});
            else pick_thr(std::integral_constant<int, METRIC_L2>{}, std::true_type{});
        } else {
            if (masked) pick_thr(std::integral_constant<int, METRIC_IP>{}, std::false_type{});
            else pick_thr(std::integral_constant<int, METRIC_IP>{}, std::true_type{});
        }
        return;
    }
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
    int c = 0;
    for (; c + 3 < d / 4; c += 4) {  // (four steps per trip: the loads together, the sums in order)
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = r[c + u];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s0 += v[u].x * v[u].x;
            s1 += v[u].y * v[u].y;
            s2 += v[u].z * v[u].z;
            s3 += v[u].w * v[u].w;
        }
    }
    for (; c < d / 4; c++) {
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

// ---- the approximate form of the product for coarse_pick_kernel: fp16 operands.  The ranking only has to be within a known
// bound of the exact distances (the candidates are recomputed exactly), so the rows are scaled by the powers of two of
// FilterParams (ivf_filter.hip: largest magnitudes into [2^14, 2^15), the rule for when one scale per matrix is safe) and rounded to
// fp16 while they are staged; v_mfma_f32_32x32x16_f16 then does in two instructions what sixteen fp32 ones did (the fp32 matrix
// product was 0.9 of the 1.9 ms a coarse ranking took at d = 960).  Same tiling: 128 x 128 per workgroup, 64 x 64 per wave, K in
// chunks of 32 dimensions, double-buffered, one barrier per chunk.  An LDS row holds the chunk's 32 halves (+ 8 of padding: eight
// lanes' 16-byte reads fall on disjoint banks); lane (m, h) of K-step s reads halves 16 s + 8 h .. + 7 of row m.
constexpr int GEMM16_TK = 32, GEMM16_LD = 40;  // halves
template <int METRIC>
__global__ __launch_bounds__(256) void coarse_gemm16_kernel(const float* X, const float* Y, const float* xn, const float* yn, int nq, int ny,
                                                            int d, float* out, const FilterParams* prm) {
    typedef _Float16 v8h __attribute__((ext_vector_type(8)));
    typedef _Float16 v4h __attribute__((ext_vector_type(4)));
    __shared__ __align__(16) _Float16 As[2][128 * GEMM16_LD], Bs[2][128 * GEMM16_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.y * 128, c0 = blockIdx.x * 128;
    const int wq = wave >> 1, wc = wave & 1;
    const int m = lane & 31, h = lane >> 5;
    const float sx = prm->sx, sy = prm->pad, ps = prm->ps;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // a chunk = 128 rows x 32 dimensions of each matrix = 1024 float4 each: four per thread and matrix
    float4 pa[4], pb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = tid + i * 256, row = idx >> 3, c = idx & 7;
            const bool kok = k0 + 4 * c < d;
            pa[i] = q0 + row < nq && kok ? *reinterpret_cast<const float4*>(X + (size_t)(q0 + row) * d + k0 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            pb[i] = c0 + row < ny && kok ? *reinterpret_cast<const float4*>(Y + (size_t)(c0 + row) * d + k0 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = tid + i * 256, row = idx >> 3, c = idx & 7;
            const v4h ha = {(_Float16)(pa[i].x * sx), (_Float16)(pa[i].y * sx), (_Float16)(pa[i].z * sx), (_Float16)(pa[i].w * sx)};
            const v4h hb = {(_Float16)(pb[i].x * sy), (_Float16)(pb[i].y * sy), (_Float16)(pb[i].z * sy), (_Float16)(pb[i].w * sy)};
            *reinterpret_cast<v4h*>(&As[buf][row * GEMM16_LD + 4 * c]) = ha;
            *reinterpret_cast<v4h*>(&Bs[buf][row * GEMM16_LD + 4 * c]) = hb;
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < d; k0 += GEMM16_TK, buf ^= 1) {
        const bool more = k0 + GEMM16_TK < d;
        if (more) fetch(k0 + GEMM16_TK);
        const _Float16* a_base = &As[buf][(wq * 64 + m) * GEMM16_LD + 8 * h];
        const _Float16* b_base = &Bs[buf][(wc * 64 + m) * GEMM16_LD + 8 * h];
#pragma unroll
        for (int st = 0; st < 2; st++) {
            const v8h a0 = *reinterpret_cast<const v8h*>(a_base + 16 * st), a1 = *reinterpret_cast<const v8h*>(a_base + 32 * GEMM16_LD + 16 * st);
            const v8h b0 = *reinterpret_cast<const v8h*>(b_base + 16 * st), b1 = *reinterpret_cast<const v8h*>(b_base + 32 * GEMM16_LD + 16 * st);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
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
                    const float ip = ps * acc[ti][tj][reg];  // (a power of two: exact)
                    float dis = ip;
                    if (METRIC == METRIC_L2) {
                        dis = xn[q] + yn[c] - 2 * ip;
                        if (dis < 0) dis = 0;
                    }
                    out[(size_t)q * ny + c] = dis;
                }
            }
}
void launch_coarse_gemm16(int metric, const float* X, const float* Y, const float* xn, const float* yn, int nq, int ny, int d, float* out,
                          const FilterParams* params, hipStream_t s) {
    if (nq == 0 || ny == 0) return;
    const dim3 grid((ny + 127) / 128, (nq + 127) / 128);
    if (metric == METRIC_L2) LAUNCH(coarse_gemm16_kernel<METRIC_L2>, grid, dim3(256), 0, s, X, Y, xn, yn, nq, ny, d, out, params);
    else LAUNCH(coarse_gemm16_kernel<METRIC_IP>, grid, dim3(256), 0, s, X, Y, xn, yn, nq, ny, d, out, params);
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

// Bitonic sort of S = EPT * 256 distinct 64-bit keys by one workgroup, thread t holding elements EPT t .. EPT t + EPT - 1 in
// registers: compare-exchange distances below EPT stay inside the thread, up to 64 threads apart they are lane exchanges, and only
// the last one or two distances of the last two or three merges cross waves, through `buf` (S slots).  (Every step through LDS
// with a workgroup barrier, as the sort was written first, cost sort_prefix_kernel 0.126 ms per 5000 rankings: 55 barriers and
// strided 8-byte accesses that collide on the banks.)  On return buf[] holds the keys in ascending order.
template <int EPT>
__device__ __forceinline__ void bitonic_sort_wg(unsigned long long* buf, uint32_t tid) {
    constexpr uint32_t S = EPT * 256u;
    unsigned long long x[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) x[j] = buf[EPT * tid + j];
    auto keep = [&](unsigned long long a, unsigned long long b, bool keep_min) { return (a < b) == keep_min ? a : b; };
    for (uint32_t size = 2; size <= S; size <<= 1) {
        const bool up = ((EPT * tid) & size) == 0;  // (merges shorter than EPT alternate inside the thread: upj below)
        for (uint32_t stride = size >> 1; stride >= (uint32_t)EPT; stride >>= 1) {
            if (stride >= EPT * 64u) {
                __syncthreads();
#pragma unroll
                for (int j = 0; j < EPT; j++) buf[EPT * tid + j] = x[j];
                __syncthreads();
                const uint32_t pt = tid ^ (stride / EPT);
                const bool keep_min = ((tid & (stride / EPT)) == 0) == up;
#pragma unroll
                for (int j = 0; j < EPT; j++) x[j] = keep(x[j], buf[EPT * pt + j], keep_min);
            } else {
                const int lm = (int)(stride / EPT);
                const bool keep_min = ((tid & (uint32_t)lm) == 0) == up;
#pragma unroll
                for (int j = 0; j < EPT; j++) {
                    const unsigned long long y = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(x[j] >> 32), lm) << 32) |
                                                 (uint32_t)__shfl_xor((int)(uint32_t)x[j], lm);
                    x[j] = keep(x[j], y, keep_min);
                }
            }
        }
        // inside the thread: element j with j | st; the merge's direction is that of element EPT t + j
#pragma unroll
        for (int st = EPT / 2; st >= 1; st >>= 1) {
            if ((uint32_t)st >= size) continue;
#pragma unroll
            for (int j = 0; j < EPT; j++) {
                if (j & st) continue;
                const bool upj = size < (uint32_t)EPT ? (((uint32_t)j & size) == 0) : up;
                const unsigned long long a = x[j], b = x[j | st];
                x[j] = keep(a, b, upj);
                x[j | st] = keep(a, b, !upj);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; j++) buf[EPT * tid + j] = x[j];
    __syncthreads();
}

// Only the first `prefix` entries of the ranking (the adaptive search never probes past (nlist / 8) * multipler):
// a workgroup keeps its row in registers (nlist <= 4096), finds by bisection on the order keys the threshold below
// which at least `prefix` values lie, and sorts just those (S = 512, 1024 or 2048 slots).  Same total order as
// sort_rows_kernel; entries [prefix, nprobe) come out as (neutral distance, -1).
template <bool Ascending>
__global__ __launch_bounds__(256) void sort_prefix_kernel(const float* dis, uint32_t nlist, uint32_t nprobe, uint32_t prefix, uint32_t S,
                                                           float* out_dis, int64_t* out_keys, uint32_t write_end) {
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
    if (S == 512) bitonic_sort_wg<2>(buf, tid);
    else if (S == 1024) bitonic_sort_wg<4>(buf, tid);
    else bitonic_sort_wg<8>(buf, tid);
    // (write_end < nprobe: the entries behind it hold (neutral, -1) already -- left by the ranking of the search before, which had the
    // same prefix: 36 of the 48 KB a ranking of 4096 takes, 245 MB per 5000 rankings that nothing reads)
    for (uint32_t i = tid; i < write_end; i += 256) {
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

// ---- exact top-nprobe of every ranking from APPROXIMATE distances (coarse_gemm_kernel) + exact recomputation of the few
// centroids that can be among them.  The exact coarse distances of a call cost n x nlist x d operations in the reference's
// rounding sequence on the vector ALU (2.98 ms of a 12.5 ms search at d = 960, 10 000 queries); a fixed-nprobe search reads
// nprobe << nlist of them.  Here the matrix cores rank all centroids approximately, and one workgroup per query
//   1. finds a threshold T with at least nprobe approximate distances at or below it (bisection on order keys, as
//      sort_prefix_kernel), and widens it by 2 eps: |approx - exact| <= eps = C (|x|^2 + max |c|^2) (L2) / C |x| max |c| (IP) with
//      C = (2 d + 32) 2^-24, the bound of the fp32 filter (ivf_filter.hip) -- every centroid whose EXACT distance is among the
//      nprobe best has its approximate one within the widened threshold (exact_(nprobe) <= T + eps, approx <= exact + eps);
//   2. recomputes those candidates' distances in the reference's sequence (exact_distance, one candidate per thread) and
//      sorts them;
//   3. writes the first nprobe -- unless two of the first nprobe + 1 are exactly equal (their order in the reference is its
//      heap's history, utils.cpp:454-490), a distance is not finite, or the candidates overflow the buffer: such a query is
//      FLAGGED and recomputed by the caller the old way (exact tile kernel + the reference's heap).
// Distinct finite distances leave the reference's heap in ascending (L2) / descending (IP) order, so what is written is its
// result bit for bit.
struct CoarsePickArgs {
    const float* approx;      // n x nlist
    const float* x;           // n x dpad
    const float* centroids;   // nlist x dpad
    const float* xn;          // n: |x|^2 (row_norms_kernel)
    float cmax;               // max over centroids of |c|^2
    float C;
    const FilterParams* params;  // approximate distances from fp16 operands: C = params->C (< 0: they mean nothing, every query is flagged)
    uint32_t nlist, nprobe;
    int dpad;
    float* out_dis;           // n x nprobe
    int64_t* out_keys;
    uint32_t* nflag;          // += 1 per flagged query
    uint32_t* flagged;        // [n] their numbers
    uint32_t pick_slack;      // the bisection stops at a threshold with nprobe .. nprobe + pick_slack approximate distances under it
                              // (every candidate under the widened threshold costs a gathered row of d floats: cfg 5, d = 960,
                              // nprobe 32 -- slack 48 / 16 / 8 / 2: 1.07 / 0.93 / 0.89 / 0.87 ms of coarse ranking)
    uint32_t* why;            // [5] (debugging, or null): flagged because of 0 the threshold / scale, 1 too many candidates, 2 too few,
                              // 3 a distance that is not finite, 4 equal distances among the first nprobe + 1
};
constexpr uint32_t PICK_CAP = 1024;
template <int METRIC>
__global__ __launch_bounds__(256) void coarse_pick_kernel(CoarsePickArgs a) {
    constexpr bool Ascending = METRIC == METRIC_L2;
    __shared__ unsigned long long buf[PICK_CAP];
    __shared__ uint32_t red[2][4];
    __shared__ uint32_t s_cnt, s_bad;
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = a.approx + (size_t)q * a.nlist;
    constexpr int E = 16;  // nlist <= 4096
    uint32_t key[E];
#pragma unroll
    for (int j = 0; j < E; j++) {
        const uint32_t i = tid + 256 * j;
        key[j] = 0xffffffffu;
        if (i < a.nlist) key[j] = Ascending ? fkey(row[i]) : ~fkey(row[i]);
    }
    if (tid == 0) s_cnt = 0, s_bad = 0;
    int par = 0;
    auto block_count = [&](uint32_t thr) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < E; j++) c += __builtin_popcountll(__ballot(key[j] <= thr));
        if (lane == 0) red[par][wave] = c;
        __syncthreads();
        const uint32_t tot = red[par][0] + red[par][1] + red[par][2] + red[par][3];
        par ^= 1;
        return tot;
    };
    // a threshold with nprobe .. nprobe + pick_slack approximate distances at or below it (any such T is >= the nprobe-th smallest)
    uint32_t lo = 0, hi = 0xffffffffu;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        const uint32_t c = block_count(mid);
        if (c >= a.nprobe) {
            hi = mid;
            if (c <= a.nprobe + a.pick_slack) break;
        } else {
            lo = mid + 1;
        }
    }
    const float Tv = fkey_inv(Ascending ? hi : ~hi);
    const float xn = a.xn[q];
    const float Cb = a.params ? a.params->C : a.C;
    const float eps = METRIC == METRIC_L2 ? Cb * (xn + a.cmax) : Cb * sqrtf(xn) * sqrtf(a.cmax);
    const float Tw = Ascending ? Tv + 2.f * eps : Tv - 2.f * eps;
    bool bad = !(fabsf(Tw) < 3.0e38f) || Cb < 0.f;  // (NaN, infinities, no usable fp16 scale: the caller's exact path deals with them)
    uint32_t why = bad ? 1u : 0u;
    const uint32_t kw = Ascending ? fkey(Tw) : ~fkey(Tw);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < E; j++) {
        const uint32_t i = tid + 256 * j;
        const bool take = key[j] <= kw && i < a.nlist;
        const unsigned long long m = __ballot(take);
        uint32_t base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_cnt, (uint32_t)__builtin_popcountll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t slot = base + __builtin_popcountll(m & ((1ull << lane) - 1));
        if (take && slot < PICK_CAP) buf[slot] = i;
    }
    __syncthreads();
    const uint32_t C = s_cnt;
    if (C > PICK_CAP || C < a.nprobe) bad = true, why |= C > PICK_CAP ? 2u : 4u;
    const uint32_t n = C < PICK_CAP ? C : PICK_CAP;
    uint32_t S = 64;
    while (S < n) S <<= 1;
    // Exact distances of the candidates in the reference's sequence (utils_simd.cpp:391-443): four running sums over the elements
    // 4 i + l, each accumulated in order, then (s0 + s1) + (s2 + s3).  The four sums are independent chains, so four
    // neighbouring lanes take one candidate (lane l its sum l: same rounding sequence) and a wave-instruction touches 16 rows
    // instead of 64 -- with one thread per candidate the 3840-byte rows of d = 960 made this loop, not the matrix product, the
    // coarse ranking's time.
    const float* xq = a.x + (size_t)q * a.dpad;
    const uint32_t sub = tid & 3;
    for (uint32_t c0 = 0; c0 < S; c0 += 64) {
        const uint32_t c = c0 + (tid >> 2);
        const bool have = c < n;
        const uint32_t i = have ? (uint32_t)buf[c] : 0u;
        const float* y = a.centroids + (size_t)i * a.dpad;
        float sl = 0.f;
        if (have) {
            // (sixteen steps of the chain per trip: their loads are requested together, the sums still run in order -- one step at a
            // time the loop paid a trip to L2 per step of a chain d / 4 steps long; eight a trip: 30 trips at d = 960)
            int e = (int)sub;
            for (; e + 60 < a.dpad; e += 64) {
                float yv[16], xv[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    yv[u] = y[e + 4 * u];
                    xv[u] = xq[e + 4 * u];
                }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    if (METRIC == METRIC_L2) {
                        const float t = yv[u] - xv[u];
                        sl += t * t;
                    } else {
                        sl += yv[u] * xv[u];
                    }
                }
            }
            for (; e < a.dpad; e += 4) {
                if (METRIC == METRIC_L2) {
                    const float t = y[e] - xq[e];
                    sl += t * t;
                } else {
                    sl += y[e] * xq[e];
                }
            }
        }
        const float s1 = __shfl_xor(sl, 1);           // lanes 0/1: s0 + s1, lanes 2/3: s2 + s3 (the sum is commutative bit for bit)
        const float pair = sl + s1;
        const float ex = pair + __shfl_xor(pair, 2);  // (s0 + s1) + (s2 + s3)
        __syncthreads();  // (every thread has read its candidate number before the slots are rewritten)
        if (sub == 0) {
            unsigned long long e = ~0ull;
            if (have) {
                if (!(fabsf(ex) < 3.0e38f)) bad = true, why |= 8u;
                e = ((unsigned long long)(Ascending ? fkey(ex) : ~fkey(ex)) << 32) | i;
            }
            buf[c] = e;
        }
    }
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
    // equal neighbours among the first nprobe + 1: the reference's order there is its heap's
    for (uint32_t i = tid; i < a.nprobe && i + 1 < n; i += 256)
        if ((uint32_t)(buf[i] >> 32) == (uint32_t)(buf[i + 1] >> 32)) bad = true, why |= 16u;
    if (bad) atomicOr(&s_bad, why ? why : 1u);
    __syncthreads();
    if (s_bad) {
        if (tid == 0) {
            a.flagged[atomicAdd(a.nflag, 1u)] = q;
            if (a.why)
                for (int r = 0; r < 5; r++)
                    if (s_bad & (1u << r)) atomicAdd(&a.why[r], 1u);
        }
        return;
    }
    for (uint32_t i = tid; i < a.nprobe; i += 256) {
        const unsigned long long e = buf[i];
        const uint32_t k = (uint32_t)(e >> 32);
        a.out_dis[(size_t)q * a.nprobe + i] = fkey_inv(Ascending ? k : ~k);
        a.out_keys[(size_t)q * a.nprobe + i] = (int64_t)(uint32_t)e;
    }
}
void launch_coarse_pick(int metric, const float* approx, const float* x, const float* centroids, const float* xn, float cmax, uint32_t n,
                        uint32_t nlist, uint32_t nprobe, int dpad, float* out_dis, int64_t* out_keys, uint32_t* nflag, uint32_t* flagged,
                        hipStream_t s, const FilterParams* params, uint32_t* why) {
    if (n == 0) return;
    CoarsePickArgs a{approx, x, centroids, xn, cmax, (2.f * (float)dpad + 32.f) * 5.9604645e-8f, params, nlist, nprobe, dpad, out_dis, out_keys, nflag, flagged,
                     getenv("AUNCEL_AMD_PICK_SLACK") ? (uint32_t)atoi(getenv("AUNCEL_AMD_PICK_SLACK")) : nprobe / 8u + 2u, why};
    if (metric == METRIC_L2) LAUNCH(coarse_pick_kernel<METRIC_L2>, dim3(n), dim3(256), 0, s, a);
    else LAUNCH(coarse_pick_kernel<METRIC_IP>, dim3(n), dim3(256), 0, s, a);
}
// out row idx[j] <- in row j (rows of `words` 4-byte words)
__global__ void scatter_rows_kernel(const uint32_t* in, const uint32_t* idx, uint32_t words, uint32_t* out) {
    const uint32_t j = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) out[(size_t)idx[j] * words + i] = in[(size_t)j * words + i];
}
void launch_scatter_rows(const void* in, const uint32_t* idx, uint32_t m, uint32_t words, void* out, hipStream_t s) {
    if (m == 0) return;
    LAUNCH(scatter_rows_kernel, dim3(m), dim3(64), 0, s, static_cast<const uint32_t*>(in), idx, words, static_cast<uint32_t*>(out));
}

// prefix: 0 = rank all nprobe entries; else only the first `prefix` are needed (see sort_prefix_kernel)
bool sort_rows_ranks_a_prefix(uint32_t nlist, uint32_t nprobe, uint32_t prefix) {
    return prefix && prefix < nprobe && prefix <= 2048 && prefix <= nlist && nlist <= 4096;
}
// tail_is_neutral: the entries [prefix, nprobe) of every row are (neutral, -1) already (the caller's book-keeping: coarse_dev)
void launch_sort_rows(const float* dis, uint32_t nq, uint32_t nlist, uint32_t nprobe, int metric, float* out_dis,
                      int64_t* out_keys, hipStream_t s, uint32_t prefix, bool tail_is_neutral) {
    if (nq == 0) return;
    if (sort_rows_ranks_a_prefix(nlist, nprobe, prefix)) {
        const uint32_t S = prefix <= 384 ? 512u : prefix <= 1024 ? 1024u : 2048u;  // (slots sorted: >= prefix, with room for the bisection to stop early)
        const uint32_t write_end = tail_is_neutral ? prefix : nprobe;
        if (metric == METRIC_L2) LAUNCH(sort_prefix_kernel<true>, dim3(nq), dim3(256), 0, s, dis, nlist, nprobe, prefix, S, out_dis, out_keys, write_end);
        else LAUNCH(sort_prefix_kernel<false>, dim3(nq), dim3(256), 0, s, dis, nlist, nprobe, prefix, S, out_dis, out_keys, write_end);
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
template <bool IsMax> __device__ inline void floyd_fill_pow2(HeapEnt* a, const float* __restrict__ row, uint32_t n, int lane) {
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

// One step of a walk: the hole takes the better child (the right one between equals) unless the entry the walk carries beats it, in
// which case that entry lands in the hole and the walk ends.  A lane without a walk steps on slot 0 (unused) instead of being masked
// off; a hole without children reads the heap's last pair (not used).
struct HeapWalk {
    uint32_t hole, s, lvl, Lid;  // slot the walk stands on (0: none), heap size of its pop, depth of hole, id it carries
    float Lv;
};
template <bool IsMax> __device__ __forceinline__ void heap_walk_step(HeapEnt* a, HeapWalk& w, const uint4 ch) {
    const uint32_t j1 = w.hole << 1;
    const float c1v = __uint_as_float(ch.x), c2v = __uint_as_float(ch.z);
    const bool left = j1 == w.s || hcmp<IsMax>(c1v, c2v);  // a single child, or the better one (the right one between equals)
    const float cv = left ? c1v : c2v;
    const uint32_t cid = left ? ch.y : ch.w;
    const bool done = j1 > w.s || hcmp<IsMax>(w.Lv, cv);
    reinterpret_cast<uint2*>(a)[w.hole] = make_uint2(__float_as_uint(done ? w.Lv : cv), done ? w.Lid : cid);
    const uint32_t nh = left ? j1 : j1 + 1;
    w.hole = done ? 0u : nh;
    w.lvl++;
}

// Two kinds of tick (round 5; the one-kind loop it replaces issued ~85 instructions a tick and was bound by that, not by the LDS
// round trip: 1490 cycles per pop, now 870; scratch/ubench/heap_sort.hip).  A pop can start at most every other tick, so the tick
// behind a start is a plain step of the walks in flight (children -> two compares -> store) and only the other ticks carry the
// start logic; walks sit on a ring of lanes by pop number (at most one per two levels is alive).
template <bool IsMax> __device__ inline void heapsort_pipelined(HeapEnt* a, uint32_t n, int64_t* out_keys, uint32_t nout, int lane) {
    HeapWalk w{0u, 0xffffffffu, 0u, 0u, 0.f};
    uint32_t t = 0;  // pops started
    const uint4* a4 = reinterpret_cast<const uint4*>(a);
    const uint2* a2 = reinterpret_cast<const uint2*>(a);
    while (true) {
        // ---- a tick that may start pop t: it takes the entry of slot n - t and walks it down from the root
        const uint32_t sc = n - t;
        const uint32_t lim = n >> 1;  // (a has n + 2 slots: pair n / 2 is its last)
        const uint4 ch_own = a4[w.hole < lim ? w.hole : lim];
        const unsigned long long act = __ballot(w.hole != 0);
        if (t == n) {
            if (!act) break;
            heap_walk_step<IsMax>(a, w, ch_own);
            wave_sync();
            continue;
        }
        // (no start while a walk in flight is above slot n - t: it could still change the entry the pop is about to take)
        const uint32_t dsc = 31 - __builtin_clz(sc);
        const bool above = w.hole != 0 && w.lvl <= dsc && (sc >> ((dsc - w.lvl) & 31)) == w.hole;
        const bool create = !__ballot(above);
        const bool mine = create && (uint32_t)lane == (t & 31u);
        uint4 ch = ch_own;
        if (mine) {
            const uint4 r01 = a4[0];  // entry 1: the root
            ch = a4[1];               // its children
            const uint2 ls = a2[sc];
            if (sc - 1 < nout) out_keys[sc - 1] = (int64_t)r01.w;  // heap_reorder: the top goes behind the shrinking heap (the places that are read)
            w.hole = 1u;
            w.s = sc;
            w.lvl = 0u;
            w.Lv = __uint_as_float(ls.x);
            w.Lid = ls.y;
        }
        heap_walk_step<IsMax>(a, w, ch);
        wave_sync();
        if (create) {
            t++;
            // ---- the tick behind a start: a plain step
            const uint4 c2 = a4[w.hole < lim ? w.hole : lim];
            heap_walk_step<IsMax>(a, w, c2);
            wave_sync();
        }
    }
}

template <bool IsMax>
__global__ __launch_bounds__(64) void heap_tie_order_kernel(const float* dis, uint32_t nlist, uint32_t nprobe, uint32_t nout,
                                                            float* out_dis, int64_t* out_keys, unsigned long long* nrows,
                                                            const uint32_t* dev_nq) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (dev_nq && blockIdx.x >= *dev_nq) return;  // (rows set aside on the device: launch_spec_collect)
#ifndef AUNCEL_HEAP_WAVE_PRIO
#define AUNCEL_HEAP_WAVE_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(AUNCEL_HEAP_WAVE_PRIO);  // (one wave's dependent chain of a few thousand steps, beside other searches' kernels)
    HeapEnt* h = reinterpret_cast<HeapEnt*>(smem);                           // nprobe + 2 entries, [0] unused
    // (the distance row stays in global memory -- every entry is read once or twice -- and the sorted ids go straight to the ranking:
    // 33 KB of LDS a row instead of 49, with half a thousand rows of four searches resident at a time)
    const bool fast = nprobe == nlist && (nlist & (nlist - 1)) == 0 && nlist >= 64;
    float* row = fast ? nullptr : reinterpret_cast<float*>(smem + (size_t)(nprobe + 2) * 8);  // nlist (the literal heap only)
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
        if (row) row[j] = v;
        enters &= hcmp<IsMax>(neutral, v);
    }
    for (uint32_t i = lane; i < nprobe + 2; i += 64) h[i] = HeapEnt{neutral, 0xffffffffu};  // heap_heapify, no input
    wave_sync();
    if (lane == 0) atomicAdd(nrows, 1ull);
    if (fast && !__ballot(!enters)) {
        floyd_fill_pow2<IsMax>(h, grow, nlist, (int)lane);
        // same distances in the same places, only centroid numbers inside runs of equal distances move
        heapsort_pipelined<IsMax>(h, nlist, ok, nout < nprobe ? nout : nprobe, (int)lane);
        return;
    }
    if (!row) {  // (a row that holds a value the empty heap would not admit: the literal heap, from the row in global memory)
        row = const_cast<float*>(grow);
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

// first_tie_kernel and spec_collect_kernel in one launch (one wave per ranking): the rankings whose first run starts below `window`
// are set aside for the heap right where the run is found -- three launches fewer between the coarse ranking and the start of the
// heap, whose end the first selection waits for
__global__ __launch_bounds__(64) void tie_collect_kernel(const float* sorted_dis, uint32_t nreal, uint32_t window, uint32_t cap, uint32_t nlist,
                                                         uint32_t ncopy, const float* full, const int64_t* ckeys, uint32_t* first_out, uint32_t* count,
                                                         int32_t* slot_of, float* s_full, float* s_dis, int64_t* s_keys, uint32_t* slot_query) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const float* od = sorted_dis + (size_t)q * nlist;
    uint32_t first = 0xffffffffu;
    for (uint32_t i = lane; i + 1 < nreal; i += 64)
        if (od[i] == od[i + 1] && i < first) first = i;
    for (int off = 32; off; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)first, off);
        first = o < first ? o : first;
    }
    if (lane == 0) first_out[q] = first;
    uint32_t slot = 0xffffffffu;
    if (lane == 0 && first < window) slot = atomicAdd(count, 1u);
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    if (slot >= cap) {
        if (lane == 0) slot_of[q] = -1;
        return;
    }
    if (lane == 0) {
        slot_of[q] = (int32_t)slot;
        slot_query[slot] = q;
    }
    const float* fr = full + (size_t)q * nlist;
    float* fo = s_full + (size_t)slot * nlist;
    for (uint32_t j = lane; j < nlist; j += 64) fo[j] = fr[j];
    for (uint32_t j = lane; j < ncopy; j += 64) {
        s_dis[(size_t)slot * nlist + j] = od[j];
        s_keys[(size_t)slot * nlist + j] = ckeys[(size_t)q * nlist + j];
    }
}

void launch_tie_collect(const float* sorted_dis, uint32_t nq, uint32_t nreal, uint32_t window, uint32_t cap, uint32_t nlist, uint32_t ncopy,
                        const float* full, const int64_t* ckeys, uint32_t* first_out, uint32_t* count, int32_t* slot_of, float* s_full, float* s_dis,
                        int64_t* s_keys, uint32_t* slot_query, hipStream_t s) {
    if (nq)
        LAUNCH(tie_collect_kernel, dim3(nq), dim3(64), 0, s, sorted_dis, nreal, window, cap, nlist, ncopy, full, ckeys, first_out, count, slot_of, s_full,
               s_dis, s_keys, slot_query);
}

void launch_first_tie(const float* sorted_dis, uint32_t nq, uint32_t stride, uint32_t nreal, uint32_t* out, hipStream_t s) {
    if (nq) LAUNCH(first_tie_kernel, dim3(nq), dim3(64), 0, s, sorted_dis, stride, nreal, out);
}

__global__ __launch_bounds__(64) void spec_collect_kernel(const uint32_t* first, uint32_t lo, uint32_t window, uint32_t cap, uint32_t nlist, uint32_t ncopy,
                                                          const float* full, const float* cdis, const int64_t* ckeys, uint32_t* count,
                                                          int32_t* slot_of, float* s_full, float* s_dis, int64_t* s_keys, uint32_t* slot_query) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    uint32_t slot = 0xffffffffu;
    if (first[q] < lo) return;  // (dealt with by the launch for the nearer runs)
    if (lane == 0 && first[q] < window) slot = atomicAdd(count, 1u);
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    if (slot >= cap) {
        if (lane == 0) slot_of[q] = -1;
        return;
    }
    if (lane == 0) {
        slot_of[q] = (int32_t)slot;
        if (slot_query) slot_query[slot] = q;
    }
    const float* fr = full + (size_t)q * nlist;
    float* fo = s_full + (size_t)slot * nlist;
    for (uint32_t j = lane; j < nlist; j += 64) fo[j] = fr[j];
    for (uint32_t j = lane; j < ncopy; j += 64) {
        s_dis[(size_t)slot * nlist + j] = cdis[(size_t)q * nlist + j];
        s_keys[(size_t)slot * nlist + j] = ckeys[(size_t)q * nlist + j];
    }
}

void launch_spec_collect(const uint32_t* first, uint32_t nq, uint32_t lo, uint32_t window, uint32_t cap, uint32_t nlist, uint32_t ncopy,
                         const float* full, const float* cdis, const int64_t* ckeys, uint32_t* count, int32_t* slot_of, float* s_full,
                         float* s_dis, int64_t* s_keys, hipStream_t s, uint32_t* slot_query) {
    if (nq)
        LAUNCH(spec_collect_kernel, dim3(nq), dim3(64), 0, s, first, lo, window, cap, nlist, ncopy, full, cdis, ckeys, count, slot_of, s_full, s_dis, s_keys,
               slot_query);
}

// The heap's order of the slots' rankings, applied to the search that is under way (TiePatchArgs, ivf_kernels.h).  One wave per slot.
__global__ __launch_bounds__(64) void tie_patch_kernel(TiePatchArgs a) {
    const uint32_t slot = blockIdx.x, lane = threadIdx.x;
    const uint32_t nslots = *a.count < a.cap ? *a.count : a.cap;
    if (slot >= nslots) return;
    const uint32_t q = a.slot_query[slot];
    const int64_t* nk = a.s_keys + (size_t)slot * a.nlist;
    int64_t* ok = a.ckeys + (size_t)q * a.key_stride;
    const uint32_t cnt0 = a.seg_count[q];
    bool changed = false, in_round = false;
    for (uint32_t j = lane; j < a.ncopy; j += 64) {
        const int64_t want = nk[j];
        if (ok[j] != want) {
            ok[j] = want;
            changed = true;
            in_round = in_round || j < cnt0;
        }
    }
    if (!__ballot(changed)) return;
    if (lane == 0) atomicAdd(a.patched, 1u);
    if (!__ballot(in_round)) return;
    // ---- the rows the round has scanned already are in the old order: put them (and their table entries) in the new one
    auto give_up = [&]() {
        if (lane == 0) a.slot_of[q] = -2 - (int32_t)slot;  // (this query is searched again, its ranking taken from the slot: adaptive_redo_ties)
    };
    if (cnt0 > 64) return give_up();
    const uint32_t seg0 = a.seg_begin[q];
    const bool in = lane < cnt0;
    const int32_t oldkey = in ? a.seg_list[seg0 + lane] : -1;
    const int32_t newkey = in ? (int32_t)nk[lane] : -1;
    const unsigned long long off_old = in ? a.seg_off[seg0 + lane] : 0ull;
    unsigned long long psz = 0;
    if (in && oldkey >= 0 && (uint32_t)oldkey < a.nlist) {
        const unsigned long long sz = a.list_off[oldkey + 1] - a.list_off[oldkey], ra = a.row_align - 1;
        psz = (sz + ra) & ~ra;
    }
    // where the row that belongs at place `lane` is now
    int src = -1;
    for (uint32_t p = 0; p < cnt0; p++)
        if (rl_i(oldkey, (int)p) == newkey) src = (int)p;
    if (__ballot(in && (src < 0 || newkey < 0)) != 0) return give_up();  // (a run that crosses the end of the round's rows)
    const unsigned long long moved = __ballot(in && src != (int)lane);
    if (!moved) return;
    const int lo = __builtin_ctzll(moved), hi = 63 - __builtin_clzll(moved);
    // sizes in the new order, new offsets inside [lo, hi]: exclusive prefix sums
    const uint32_t my_psz = (uint32_t)psz;
    uint32_t new_psz = (uint32_t)__shfl((int)my_psz, src < 0 ? 0 : src);
    const bool span = (int)lane >= lo && (int)lane <= hi;
    uint32_t incl = span ? new_psz : 0u;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
        if ((int)lane >= off) incl += o;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, hi);
    const unsigned long long base = ((unsigned long long)rl_u((uint32_t)(off_old >> 32), lo) << 32) | rl_u((uint32_t)off_old, lo);
    const unsigned long long off_new = base + (incl - (span ? new_psz : 0u));
    // room in the scratch buffer
    unsigned long long at = 0;
    if (lane == 0) at = atomicAdd(a.cursor, (unsigned long long)total);
    at = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)at);
    if (at + total > a.scratch_floats) return give_up();
    float* sc = a.scratch + at;
    for (uint32_t i = lane; i < total; i += 64) sc[i] = a.dist[base + i];
    __threadfence();
    wave_sync();
    for (int p = lo; p <= hi; p++) {
        const int sp = rl_i(src, p);
        const unsigned long long from = (((unsigned long long)rl_u((uint32_t)(off_old >> 32), sp) << 32) | rl_u((uint32_t)off_old, sp)) - base;
        const unsigned long long to = ((unsigned long long)rl_u((uint32_t)(off_new >> 32), p) << 32) | rl_u((uint32_t)off_new, p);
        const uint32_t len = rl_u(new_psz, p);
        for (uint32_t i = lane; i < len; i += 64) a.dist[to + i] = sc[from + i];
    }
    if (span) {
        a.seg_list[seg0 + lane] = newkey;
        a.seg_off[seg0 + lane] = off_new;
    }
}

void launch_tie_patch(const TiePatchArgs& a, hipStream_t s) {
    if (a.cap) LAUNCH(tie_patch_kernel, dim3(a.cap), dim3(64), 0, s, a);
}

__global__ __launch_bounds__(64) void spec_gather_kernel(const int32_t* slots, uint32_t nlist, uint32_t ncopy, const float* s_dis,
                                                         const int64_t* s_keys, float* cdis, int64_t* ckeys) {
    const uint32_t j = blockIdx.x;
    const size_t src = (size_t)slots[j] * nlist, dst = (size_t)j * nlist;
    for (uint32_t i = threadIdx.x; i < ncopy; i += 64) {
        cdis[dst + i] = s_dis[src + i];
        ckeys[dst + i] = s_keys[src + i];
    }
}

void launch_spec_gather(const int32_t* slots, uint32_t m, uint32_t nlist, uint32_t ncopy, const float* s_dis, const int64_t* s_keys,
                        float* cdis, int64_t* ckeys, hipStream_t s) {
    if (m) LAUNCH(spec_gather_kernel, dim3(m), dim3(64), 0, s, slots, nlist, ncopy, s_dis, s_keys, cdis, ckeys);
}

__global__ __launch_bounds__(64) void gather_rows_kernel(const float* x, const uint32_t* idx, uint32_t dpad, float* out) {
    const float* src = x + (size_t)idx[blockIdx.x] * dpad;
    float* dst = out + (size_t)blockIdx.x * dpad;
    for (uint32_t i = threadIdx.x; i < dpad; i += 64) dst[i] = src[i];
}

void launch_gather_rows(const float* x, const uint32_t* idx, uint32_t m, uint32_t dpad, float* out, hipStream_t s) {
    if (m) LAUNCH(gather_rows_kernel, dim3(m), dim3(64), 0, s, x, idx, dpad, out);
}

size_t heap_tie_order_lds(uint32_t nlist, uint32_t nprobe) {
    const bool fast = nprobe == nlist && (nlist & (nlist - 1)) == 0 && nlist >= 64;
    return (size_t)(nprobe + 2) * 8 + (fast ? 0 : (size_t)nlist * 4);
}

// nout: leading entries of each ranking the caller reads (<= nprobe); rows without a run of equal distances there are left
// as they are.  Returns false (nothing launched) when heap and row do not fit one workgroup's LDS.
bool launch_heap_tie_order(const float* dis, uint32_t nq, uint32_t nlist, uint32_t nprobe, uint32_t nout, int metric, float* out_dis,
                           int64_t* out_keys, unsigned long long* nrows, hipStream_t s, const uint32_t* dev_nq) {
    if (nq == 0) return true;
    const size_t shmem = heap_tie_order_lds(nlist, nprobe);
    if (shmem > 160 * 1024) return false;
    if (metric == METRIC_L2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(heap_tie_order_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        LAUNCH(heap_tie_order_kernel<true>, dim3(nq), dim3(64), shmem, s, dis, nlist, nprobe, nout, out_dis, out_keys, nrows, dev_nq);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(heap_tie_order_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        LAUNCH(heap_tie_order_kernel<false>, dim3(nq), dim3(64), shmem, s, dis, nlist, nprobe, nout, out_dis, out_keys, nrows, dev_nq);
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
        if (a.fix_val) {
            a.fix_val[i] = a.neutral;
            a.fix_ref[i] = -1;
        }
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
        a.thr[i] = a.neutral;
        a.stage[i] = 0;
        a.nscan[i] = 0;
        a.done[i] = 0;
        a.pre_val[i] = 0.f;
        a.stoped[i] = 0;
        if (a.qstat) a.qstat[i] = make_uint2(0u, 0u);
        if (a.log_cnt) {
            a.log_cnt[i] = 0;
            a.amb[i] = 0xffffffffu;
            a.tie_flag[i] = 0;
            a.log_snap[i] = 0;
            a.log_snap[a.n + i] = 0;
            a.fin_round[i] = 0xffffffffu;
            a.fix_pos[i] = 0;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 4 * STATS_ROWS) a.stats[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 4) *a.error = 0;
}

__global__ __launch_bounds__(256) void copy_segs_kernel(CopySegs c) {
    const uint32_t seg = blockIdx.y;
    const uint32_t* src = static_cast<const uint32_t*>(c.src[seg]);
    uint32_t* dst = static_cast<uint32_t*>(c.dst[seg]);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < c.words[seg]; i += gridDim.x * 256) dst[i] = src[i];
}
void launch_copy_segs(const CopySegs& c, hipStream_t s) {
    if (c.n == 0) return;
    uint32_t mx = 0;
    for (uint32_t i = 0; i < c.n; i++) mx = c.words[i] > mx ? c.words[i] : mx;
    const unsigned gx = mx <= 256 ? 1u : mx <= 4096 ? 4u : 16u;
    LAUNCH(copy_segs_kernel, dim3(gx, c.n), dim3(256), 0, s, c);
}

void launch_init_state(const InitStateArgs& a, hipStream_t s) {
    const size_t work = std::max<size_t>(a.n * a.k, 1);
    const unsigned grid = (unsigned)std::min<size_t>((work + 255) / 256, 4096);
    LAUNCH(init_state_kernel, dim3(grid), dim3(256), 0, s, a);
}

// InvertedListScanner::scan_codes_range (Auncel/IndexIVFFlat.cpp:139-155) over one row of n distances: the entries with
// C::cmp(radius, dis) -- dis < radius for L2, dis > radius for inner product -- in position order, as RangeQueryResult::add receives
// them.  One workgroup of 256 threads walks the row 256 candidates at a time; a chunk's survivors keep their order through the
// waves' ballots (rank inside the wave) and the four waves' counts (LDS).  out_pos / out_dis hold `*count` entries afterwards.
__global__ __launch_bounds__(256) void range_collect_kernel(const float* __restrict__ dist, uint32_t n, float radius, int metric, uint32_t* __restrict__ count,
                                                            uint32_t* __restrict__ out_pos, float* __restrict__ out_dis) {
    __shared__ uint32_t wave_cnt[4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t base = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += 256) {
        const uint32_t j = c0 + threadIdx.x;
        const float v = j < n ? dist[j] : 0.f;
        const bool keep = j < n && (metric == METRIC_L2 ? radius > v : radius < v);
        const unsigned long long b = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < 4; w++) {
            const uint32_t cw = wave_cnt[w];
            before += w < wave ? cw : 0u;
            total += cw;
        }
        if (keep) {
            const uint32_t o = base + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            out_pos[o] = j;
            out_dis[o] = v;
        }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}

void launch_range_collect(const float* dist, uint32_t n, float radius, int metric, uint32_t* count, uint32_t* out_pos, float* out_dis, hipStream_t s) {
    LAUNCH(range_collect_kernel, dim3(1), dim3(256), 0, s, dist, n, radius, metric, count, out_pos, out_dis);
}

void launch_fill_f32(float* p, size_t n, float v, hipStream_t s) {
    if (n) LAUNCH(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}
void launch_fill_i64(int64_t* p, size_t n, int64_t v, hipStream_t s) {
    if (n) LAUNCH(fill_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}
}  // namespace amdivf
