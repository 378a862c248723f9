// gfx950 (MI355X, CDNA4): the fp32 list scan of threshold rounds as a filter on the matrix cores + exact rescoring.
//
// The reference's distance (Auncel/utils_simd.cpp:391-443: four running fp32 sums, products and sums rounded separately) is
// a rounding *sequence* the matrix cores cannot produce, and computing it for every (query, vector) pair binds the fp32 scan
// to the vector ALU (scan_tiles_kernel: 3 packed instructions per 2 elements).  In a threshold round only the candidates
// that beat the query's threshold matter -- typically a per cent of the pairs.  So:
//   1. scan_filter_kernel   x.y of 32 queries x 32 vectors per v_mfma_f32_32x32x2_f32 chain, lists streamed once from a
//      (d <= 128)           fragment-ordered fp32 copy (coalesced 16-byte loads, no LDS); with the exact |x|^2, |y|^2 this
//      scan_filter_wide_    gives the distance up to a rigorous error bound eps(x, y) (below); a candidate is kept iff even
//      kernel (d > 128)     its most favourable value, approx -/+ eps, beats the threshold.  Kept candidates get their mask
//                           bit and an entry in the survivor list (slots handed out per wave in chunks: SurvChunk).  Beyond
//                           128 dimensions a workgroup shares one chunk of the list among up to 128 queries (operand
//                           through LDS), so that the list is read once however many queries probe it.
//   2. rescore_kernel       four lanes per survivor: the reference's own rounding sequence on the two fp32 rows; the exact
//                           distance goes into the distance row.  A survivor whose exact distance does not beat the threshold
//                           keeps its mask bit: the selection re-tests every candidate against the query's current worst
//                           value anyway, so it costs a comparison, never a wrong result.
// Nothing the selection sees differs from what scan_tiles_kernel would have written for the candidates that can enter.
//
// Error bound (u = 2^-24; S = |x|^2 + |y|^2; all norms accumulated in double and rounded once):
//   reference value r vs the real distance D:  |r - D| <= (d/4 + 4) u * 2 S        (L2: d/4 + 3 roundings per running sum, all
//                                                                                  terms >= 0 and <= 2 S in total)
//   MFMA dot product p vs x.y:                  |p - x.y| <= d u |x||y| <= d u S / 2   (any order of fp32 fused steps)
//   approx = xn + yn - 2 p in fp32:             three more roundings, each <= u * 2 S
//   =>  |approx - r| <= (1.5 d + 20) u S.  The kernel uses C = (2 d + 32) u (a quarter more, and the rounding of the test
//   itself) and keeps iff (1 - C)(xn + yn) - 2 p < thr.  Inner product: |p - r| <= 2 d u |x||y|; kept iff p + C |x||y| > thr.
//
// The fp16 form (HALF: round 4).  The filter only has to be *safe*, so its operands need not be the fp32 values: lists and
// queries are also kept as fp16, scaled by powers of two s_y, s_x that bring their largest magnitudes into [2^14, 2^15) (no
// overflow), 16 dimensions per piece in the operand order of v_mfma_f32_32x32x16_f16.  An element that lands below 2^-14 is
// off by at most 2^-25 absolutely; for a pair whose rows' largest scaled elements are a, b that adds at most 2^-25 d (a + b) to
// the sum, i.e. 2^-25 d (1 / a + 1 / b) of |x16||y16| >= a b -- so one scale per matrix is used only while every row that is not
// all zero keeps its largest element above max(1, d / 1024) after scaling (half_scale_of: rows within 2^-12 of the largest), which
// keeps that term below 2^-14 of |x||y|.  That halves the bytes of a pass and
// replaces sixteen fp32 MFMAs per 32 dimensions by two fp16 ones (a 70-queries-per-list pass was bound by the fp32 matrix pipe).
// Products of two fp16 values are exact in fp32, so the only new error is the operands' rounding: |fl16(x_i) fl16(y_i) - x_i y_i|
// <= (2^-10 + 2^-22) |x_i y_i|, summed <= (2^-10 + 2^-22) |x||y| <= (2^-10 + 2^-22) S / 2, twice that in an L2 distance.  The keep
// test is the same with C16 = C + 2^-10 + 2^-12 and p = ps * (the MFMA sum), ps = 1 / (s_x s_y) (a power of two: exact).
// Where every element on both sides is an integer of magnitude <= 2048 fp16 holds it exactly: s = 1 and C16 = C.  Queries
// with a non-finite element (no usable scale) make the whole call keep every candidate (C < 0 in FilterParams): slow, never wrong.
// Survivors are recomputed from the fp32 rows as before, so results do not depend on which form filtered.
#include "ivf_dev.h"

#include <algorithm>
#include <stdexcept>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <type_traits>
#include <utility>

namespace amdivf {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
// (explicitly global: a generic pointer makes these flat loads, which count on both wait counters and force full waits)
typedef const v4f __attribute__((address_space(1)))* gv4f;

// ---------------------------------------------------------------------------------------------
// fp32 lists (CSR rows, row stride dpad) -> fragment order: a list is a run of 32-vector blocks (block_off[l] = its first, an
// even number of them as for the byte codes); a block is filter_steps(d) pieces of 64 lanes x 16 bytes, lane (v = l & 31,
// h = l >> 5) of piece j holding elements 8 j + 4 h .. + 3 of vector v (zero beyond d).  yn[slot] = |y|^2 (L2) or |y| (IP).
__global__ __launch_bounds__(64) void frag32_from_f32_kernel(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist,
                                                             int d, int dpad, int metric, float* out, float* yn) {
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
    const int J = (int)filter_steps(d);
    double sq = 0.0;
    for (int j = 0; j < J; j++) {
        v4f piece;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int c = 8 * j + 4 * h + w;
            const float val = ok && c < d ? src[c] : 0.f;
            piece[w] = val;
            sq += (double)val * (double)val;
        }
        *reinterpret_cast<v4f*>(out + (blk * (uint64_t)J + (uint64_t)j) * 256 + (uint64_t)lane * 4) = piece;
    }
    sq += __shfl_xor(sq, 32);
    if (h == 0) yn[blk * 32 + v] = !ok ? 0.f : metric == METRIC_L2 ? (float)sq : (float)sqrt(sq);
}

void launch_frag32_from_f32(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist, uint64_t nblocks, int d,
                            int dpad, int metric, float* out, float* yn, hipStream_t s) {
    if (nblocks == 0) return;
    LAUNCH(frag32_from_f32_kernel, dim3((unsigned)nblocks), dim3(64), 0, s, codes, list_off, block_off, nlist, d, dpad, metric, out, yn);
}

// query rows (stride dpad) -> rows of 8 J floats (zero beyond d) + |x|^2 (L2) or |x| (IP); one wave per row
__global__ __launch_bounds__(256) void filter_queries_kernel(const float* x, size_t n, int d, int dpad, int metric, float* xf, float* xn) {
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const int stride = (int)filter_steps(d) * 8;
    double sq = 0.0;
    for (int c = lane; c < stride; c += 64) {
        const float val = c < d ? x[row * (size_t)dpad + c] : 0.f;
        xf[row * (size_t)stride + c] = val;
        sq += (double)val * (double)val;
    }
    for (int off = 32; off; off >>= 1) sq += __shfl_xor(sq, off);
    if (lane == 0) xn[row] = metric == METRIC_L2 ? (float)sq : (float)sqrt(sq);
}

void launch_filter_queries(const float* x, size_t n, int d, int dpad, int metric, float* xf, float* xn, hipStream_t s) {
    if (n == 0) return;
    LAUNCH(filter_queries_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, x, n, d, dpad, metric, xf, xn);
}

// ---------------------------------------------------------------------------------------------
// fp16 form: ranges, lists, queries
// info[0] = max |v| over all rows (bits of a non-negative float; an infinity if a NaN or an infinity was seen), info[1] != 0: some
// element is not an integer of magnitude <= 2048, info[2] = 0x7f800000 - (bits of the smallest row maximum among rows that are not
// all zero): one scale serves a whole matrix only while no row is tiny beside the largest (its elements would underflow)
__global__ __launch_bounds__(256) void amax_kernel(const float* x, size_t rows, int stride, uint32_t* info) {
    const int lane = threadIdx.x & 63;
    uint32_t mx_all = 0, bad = 0, inv_min = 0;
    auto take = [&](float v, uint32_t& mx) {
        const float av = fabsf(v);
        const uint32_t bits = av == av ? __float_as_uint(av) : 0x7f800000u;  // (NaN counts as an infinity)
        mx = bits > mx ? bits : mx;
        bad |= !(v == (float)(int)v && av <= 2048.f);
    };
    for (size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * 4) {
        uint32_t mx = 0;
        const float* row = x + r * (size_t)stride;
        if ((stride & 3) == 0) {  // (rows are padded to multiples of four floats: 16 bytes per lane, four pieces in flight)
            const v4f* row4 = reinterpret_cast<const v4f*>(row);
            const int n4 = stride >> 2;
            for (int c0 = 0; c0 < n4; c0 += 256) {
                v4f t[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int c = c0 + u * 64 + lane;
                    t[u] = c < n4 ? row4[c] : v4f{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int e = 0; e < 4; e++) take(t[u][e], mx);
            }
        } else {
            for (int c = lane; c < stride; c += 64) take(row[c], mx);
        }
        for (int off = 32; off; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)mx, off);
            mx = o > mx ? o : mx;
        }
        mx_all = mx > mx_all ? mx : mx_all;
        if (mx != 0 && 0x7f800000u - mx > inv_min) inv_min = 0x7f800000u - mx;
    }
    for (int off = 32; off; off >>= 1) bad |= (uint32_t)__shfl_xor((int)bad, off);
    // one set of atomics per workgroup, and few workgroups (launch_amax): thousands of waves adding to the same three words are
    // served one after the other at the memory side -- 0.35 ms for 10000 rows when every wave did
    __shared__ uint32_t s_mx, s_bad, s_inv;
    if (threadIdx.x == 0) s_mx = s_bad = s_inv = 0;
    __syncthreads();
    if (lane == 0) {
        atomicMax(&s_mx, mx_all);
        if (bad) atomicOr(&s_bad, 1u);
        atomicMax(&s_inv, inv_min);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(&info[0], s_mx);
        if (s_bad) atomicOr(&info[1], 1u);
        atomicMax(&info[2], s_inv);
    }
}
void launch_amax(const float* x, size_t rows, int stride, uint32_t* info, hipStream_t s) {  // (info: 4 words, zeroed by the caller)
    if (rows == 0) return;
    const unsigned grid = (unsigned)std::min<size_t>((rows + 3) / 4, 512);
    LAUNCH(amax_kernel, dim3(grid), dim3(256), 0, s, x, rows, stride, info);
}
// the power of two that brings a matrix's largest magnitude into [2^14, 2^15); 1 where every element is held exactly as it is, and
// where there is nothing to scale; 0 where no one scale is usable: a non-finite element, magnitudes outside 2^+-24, or a row
// whose largest element is below 2^-12 max(1, d / 1024) of the matrix's (scaled, its largest element must stay above
// max(1, d / 1024) for the absolute error of elements that underflow to be covered: head comment)
__host__ __device__ inline float half_scale_of(uint32_t amax_bits, uint32_t nonexact, uint32_t inv_min, int d) {
    float amax;
    memcpy(&amax, &amax_bits, 4);
    if (!nonexact || amax == 0.f) return 1.f;
    if (!(amax < __builtin_inff())) return 0.f;
    int e;
    (void)frexpf(amax, &e);  // amax in [2^(e-1), 2^e)
    if (e > 24 || e < -24) return 0.f;
    if (inv_min) {
        const uint32_t min_bits = 0x7f800000u - inv_min;
        float minrow;
        memcpy(&minrow, &min_bits, 4);
        const float need = amax * 0.000244140625f * (d > 1024 ? (float)d / 1024.f : 1.f);
        if (minrow < need) return 0.f;
    }
    return ldexpf(1.f, 15 - e);
}
__device__ __forceinline__ float half_scale(const uint32_t* info, int d) { return half_scale_of(info[0], info[1], info[2], d); }
float filter_half_scale(const uint32_t info[4], int d) { return half_scale_of(info[0], info[1], info[2], d); }

// a block = filter_steps16(d) pieces of 64 lanes x 16 bytes, lane (v = l & 31, h = l >> 5) of piece j holding elements
// 16 j + 8 h .. + 7 of vector v as fp16(s_y * value) (zero beyond d); yn as in the fp32 form (from the fp32 values)
__global__ __launch_bounds__(64) void frag16_from_f32_kernel(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist,
                                                             int d, int dpad, int metric, const uint32_t* info, float* out, float* yn) {
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
    const int J = (int)filter_steps16(d);
    const float sy = half_scale(info, d);
    double sq = 0.0;
    for (int j = 0; j < J; j++) {
        v8h piece;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const int c = 16 * j + 8 * h + w;
            const float val = ok && c < d ? src[c] : 0.f;
            piece[w] = (_Float16)(val * sy);
            sq += (double)val * (double)val;
        }
        *reinterpret_cast<v8h*>(out + (blk * (uint64_t)J + (uint64_t)j) * 256 + (uint64_t)lane * 4) = piece;
    }
    sq += __shfl_xor(sq, 32);
    if (h == 0) yn[blk * 32 + v] = !ok ? 0.f : metric == METRIC_L2 ? (float)sq : (float)sqrt(sq);
}
void launch_frag16_from_f32(const float* codes, const uint64_t* list_off, const uint64_t* block_off, uint32_t nlist, uint64_t nblocks, int d,
                            int dpad, int metric, const uint32_t* info, float* out, float* yn, hipStream_t s) {
    if (nblocks == 0) return;
    LAUNCH(frag16_from_f32_kernel, dim3((unsigned)nblocks), dim3(64), 0, s, codes, list_off, block_off, nlist, d, dpad, metric, info, out, yn);
}

// FilterParams of a pair of matrices by themselves (the approximate coarse ranking on fp16 operands: coarse_gemm16_kernel): pad =
// the second matrix's scale
__global__ void half_params_kernel(const uint32_t* qinfo, const uint32_t* yinfo, int d, FilterParams* params) {
    const float sx = half_scale(qinfo, d), sy = half_scale(yinfo, d);
    FilterParams p;
    p.sx = sx;
    p.pad = sy;
    const float C = (float)(2 * d + 32) * 5.9604644775390625e-08f;
    const bool exact = !qinfo[1] && !yinfo[1];
    p.C = sx == 0.f || sy == 0.f ? -1.f : exact ? C : C + 0.0009765625f + 0.000244140625f;
    p.ps = sx == 0.f || sy == 0.f ? 0.f : 1.f / (sx * sy);
    *params = p;
}
void launch_half_params(const uint32_t* qinfo, const uint32_t* yinfo, int d, FilterParams* params, hipStream_t s) {
    LAUNCH(half_params_kernel, dim3(1), dim3(1), 0, s, qinfo, yinfo, d, params);
}

// query rows (stride dpad) -> rows of 16 J halves (J = filter_steps16(d); row stride 8 J floats) + xn; one wave per row; the first
// thread also leaves the search's FilterParams
__global__ __launch_bounds__(256) void filter_queries16_kernel(const float* x, size_t n, int d, int dpad, int metric, const uint32_t* qinfo,
                                                               const uint32_t* yinfo, float* xf, float* xn, FilterParams* params) {
    const float sx = half_scale(qinfo, d), sy = half_scale(yinfo, d);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        FilterParams p;
        p.sx = sx;
        p.pad = 0.f;
        const float C = (float)(2 * d + 32) * 5.9604644775390625e-08f;  // (2 d + 32) 2^-24
        const bool exact = !qinfo[1] && !yinfo[1];                       // every element on both sides held exactly
        p.C = sx == 0.f || sy == 0.f ? -1.f : exact ? C : C + 0.0009765625f + 0.000244140625f;
        p.ps = sx == 0.f || sy == 0.f ? 0.f : 1.f / (sx * sy);
        *params = p;
    }
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const int stride = (int)filter_steps16(d) * 16;
    _Float16* dst = reinterpret_cast<_Float16*>(xf) + row * (size_t)stride;
    double sq = 0.0;
    for (int c = lane; c < stride; c += 64) {
        const float val = c < d ? x[row * (size_t)dpad + c] : 0.f;
        dst[c] = (_Float16)(val * sx);
        sq += (double)val * (double)val;
    }
    for (int off = 32; off; off >>= 1) sq += __shfl_xor(sq, off);
    if (lane == 0) xn[row] = metric == METRIC_L2 ? (float)sq : (float)sqrt(sq);
}
void launch_filter_queries16(const float* x, size_t n, int d, int dpad, int metric, const uint32_t* qinfo, const uint32_t* yinfo, float* xf,
                             float* xn, FilterParams* params, hipStream_t s) {
    LAUNCH(filter_queries16_kernel, dim3((unsigned)std::max<size_t>((n + 3) / 4, 1)), dim3(256), 0, s, x, n, d, dpad, metric, qinfo, yinfo, xf, xn,
           params);
}

// ---------------------------------------------------------------------------------------------
// The survivors of the filter: exact distance into the distance row, in the reference's sequence (utils_simd.cpp:391-443) -- four
// running sums over the elements 4 i + l, each accumulated in order, then (s0 + s1) + (s2 + s3).  The four sums are independent
// chains, so FOUR neighbouring lanes take one survivor (lane l its sum l: the same rounding sequence, as in coarse_pick_kernel)
// and request sixteen steps of their chain together: a survivor's chain of d / 4 steps is what the kernel's time is (a few
// survivors per query, far fewer than lanes) -- with one lane a survivor and four steps a trip cfg 5 (d = 960) took 60 trips to
// memory, 0.26 ms; now 15.
template <int METRIC> __global__ __launch_bounds__(256) void rescore_kernel(FilterScanArgs a) {
    const uint32_t n = *a.surv_count < a.surv_cap ? *a.surv_count : a.surv_cap;
    const uint32_t sub = threadIdx.x & 3;
    for (uint32_t i = (blockIdx.x * 256 + threadIdx.x) >> 2; i < n; i += gridDim.x * 64) {
        const uint4 e = a.surv[i];  // (distance row position, query row, vector, -)
        if (e.x == 0xffffffffu) continue;  // (an unused slot of a wave's chunk; the same for the four lanes of a survivor)
        const float* x = a.queries + (size_t)e.y * a.dpad;
        const float* y = a.codes + (size_t)e.z * a.dpad;
        float sl = 0.f;
        int c = (int)sub;
        for (; c + 60 < a.dpad; c += 64) {
            float xv[16], yv[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                xv[u] = x[c + 4 * u];
                yv[u] = y[c + 4 * u];
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
        for (; c < a.dpad; c += 4) {
            if (METRIC == METRIC_L2) {
                const float t = y[c] - x[c];
                sl += t * t;
            } else {
                sl += y[c] * x[c];
            }
        }
        const float pair = sl + __shfl_xor(sl, 1);         // lanes 0/1: s0 + s1, lanes 2/3: s2 + s3 (the sum is commutative bit for bit)
        const float ex = pair + __shfl_xor(pair, 2);       // (s0 + s1) + (s2 + s3)
        if (sub == 0) a.dist[e.x] = ex;
    }
}

// ---------------------------------------------------------------------------------------------
// Survivor slots are handed out in chunks per wave: one returning atomic per SURV_CHUNK survivors instead of one per tile (a
// tile of 32 x 32 pairs keeps ~15 candidates in a round that lets 1.5 % through, so every tile paid the atomic's round trip --
// about as long as the tile's 64 MFMAs).  Slots of a chunk that stay unused are marked (rescore_kernel skips them).
constexpr uint32_t SURV_CHUNK = 64;  // (a wave of the one-wave form runs one item: what it leaves unused of its last chunk is scanned by rescore_kernel)
constexpr uint32_t SURV_NONE = 0xffffffffu;
struct SurvChunk {
    uint32_t base = 0, left = 0;  // wave-uniform
};
__device__ __forceinline__ void surv_retire(const FilterScanArgs& a, SurvChunk& sc, int lane) {
    for (uint32_t s = (uint32_t)lane; s < sc.left; s += 64)
        if (sc.base + s < a.surv_cap) a.surv[sc.base + s] = make_uint4(SURV_NONE, 0u, 0u, 0u);
    sc.left = 0;
}

// ---- verdicts of block i of an item against its query block qb, whose 32 x 32 dot products are in acc; per-query operands
// of the item's queries in LDS tables (u, c, row position, query row: entry qb * 32 + query of the block); yn = this lane's
// vector's |y|^2 (L2) / |y| (IP), loaded by the caller ahead of the block's MFMAs
// ps: what the accumulators are multiplied by to give x.y (1 in the fp32 form); C < 0: keep every candidate
template <int METRIC, bool HALF>
__device__ __forceinline__ void filter_verdicts(const FilterScanArgs& a, const ScanItem& it, float C, float ps, uint32_t i, int qb, const v16f& acc,
                                                float yn, const float* s_u, const float* s_c, const uint32_t* s_row,
                                                const uint32_t* s_q, uint32_t* mask32, int lane, SurvChunk& sc) {
    const int m = lane & 31, h = lane >> 5;
    float ur[16], cr[16];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const v4f tu = *reinterpret_cast<const v4f*>(&s_u[qb * 32 + 8 * g + 4 * h]);
        const v4f tc = *reinterpret_cast<const v4f*>(&s_c[qb * 32 + 8 * g + 4 * h]);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            ur[4 * g + e] = tu[e];
            cr[4 * g + e] = tc[e];
        }
    }
    const uint32_t lv = i * 32 + m;           // position of this lane's vector in the chunk
    const bool vok = lv < it.nvec;
    const unsigned long long vmask = __ballot(vok);
    const float c1yn = (1.f - C) * yn;
    int word = 0;  // lane q (< 32) collects the 32-candidate mask word of query q of the block
    unsigned long long kept[16];
    uint32_t total = 0;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const float p = HALF ? ps * acc[reg] : acc[reg];  // (a power of two: exact)
        const float t = METRIC == METRIC_L2 ? fmaf(2.f, p, -c1yn) : fmaf(cr[reg], yn, p);
        // (keep-all leaves out what nothing can pass: an absent query slot, a NaN threshold -- their u is +inf)
        const bool keep = HALF ? ((C < 0.f) & (ur[reg] < __builtin_inff())) | (t > ur[reg]) : t > ur[reg];
        kept[reg] = __ballot(keep) & vmask;
        total += (uint32_t)__builtin_popcountll(kept[reg]);
    }
    static_for(std::make_integer_sequence<int, 16>{}, [&](auto R) {
        constexpr int reg = decltype(R)::value;
        constexpr int q0 = (reg & 3) + 8 * (reg >> 2);  // query of lane half 0; half 1: q0 + 4
        writelane_c<q0>(word, (uint32_t)kept[reg]);
        writelane_c<q0 + 4>(word, (uint32_t)(kept[reg] >> 32));
    });
    if (lane < 32 && (uint32_t)(qb * 32 + lane) < it.npair) mask32[(s_row[qb * 32 + lane] + i * 32) >> 5] = (uint32_t)word;
    if (total) {
        // survivors: (distance row position, query row, vector) for rescore_kernel; beyond the list's capacity the exact
        // distance is computed here (slow, never wrong)
        if (total > sc.left) {
            surv_retire(a, sc, lane);
            const uint32_t want = total > SURV_CHUNK ? total : SURV_CHUNK;
            uint32_t nb = 0;
            if (lane == 0) nb = atomicAdd(a.surv_count, want);
            sc.base = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
            sc.left = want;
        }
        uint32_t base = sc.base;
        sc.base += total;
        sc.left -= total;
        const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const unsigned long long kb = kept[reg];
            if (kb == 0) continue;
            if ((kb >> lane) & 1) {
                const int q = qb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const uint32_t slot = base + (uint32_t)__builtin_popcountll(kb & lt);
                const uint32_t off = s_row[q] + lv, qr = s_q[q], vec = it.qgroup + lv;
                if (slot < a.surv_cap) a.surv[slot] = make_uint4(off, qr, vec, 0u);
                else a.dist[off] = exact_distance<METRIC>(a.queries + (size_t)qr * a.dpad, a.codes + (size_t)vec * a.dpad, a.dpad);
            }
            base += (uint32_t)__builtin_popcountll(kb);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One wave = one work item as in scan_mfma_kernel: up to 32 queries probing a list x a chunk of consecutive vectors of it.
// A operand: the queries' rows (lane (m, h): elements 8 j + 4 h .. + 3 of query m, piece j), resident in registers when
// NJ != 0 (d <= 128), re-read from the (L2-resident) packed query matrix otherwise; B operand: the list in fragment order,
// P pieces in flight.  Piece j feeds four v_mfma_f32_32x32x2_f32 (element i of both 4-vectors: k = lane half).
// HALF: the fp16 form -- a piece is 16 dimensions (lane (m, h): elements 16 j + 8 h .. + 7) and feeds ONE
// v_mfma_f32_32x32x16_f16; the bytes of a piece, and so every address below, are the same.
template <bool HALF> __device__ __forceinline__ void filter_mac(v16f& acc, const v4f& av, const v4f& bv) {
    if constexpr (HALF) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, av), __builtin_bit_cast(v8h, bv), acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
    }
}
// (Occupancy is not what bounds this kernel: at NJ = 8 the compiler's free choice is 155 registers + 16 accumulators, two waves a SIMD;
// asked for three it fits 153 without spilling, for four it spills 24 -- measured on the bench workload's fp32 form, threshold passes
// per step: 1.78 / 1.78 / 2.18 ms at 2 / 3 / 4 waves, cfg 3 (NJ = 6) 0.88 / 0.88 / 0.95: profiles/r06_experiments.txt A.)
template <int METRIC, int NJ, bool HALF> __global__ __launch_bounds__(256) void scan_filter_kernel(FilterScanArgs a) {
    __shared__ float s_u[4][32];
    __shared__ float s_c[4][32];
    __shared__ uint32_t s_row[4][32];
    __shared__ uint32_t s_q[4][32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int J = NJ ? NJ : (int)(HALF ? filter_steps16(a.d) : filter_steps(a.d));
    const size_t qstride = (size_t)J * 8;
    const FilterParams fp = HALF ? *reinterpret_cast<const FilterParams*>(a.params) : FilterParams{1.f, 1.f, 0.f, 0.f};
    const float C = HALF ? fp.C : (float)(2 * a.d + 32) * 5.9604644775390625e-08f;  // (2 d + 32) 2^-24
    const float ps = fp.ps;
    const uint32_t nitems = a.dev_nitems ? *a.dev_nitems : a.nitems;
    SurvChunk sc;
    ItemWalk w((nitems + 3) >> 2, a.xcd_chunks);
    for (uint32_t wi = w.cur; wi < w.end; wi += w.step) {
        const uint32_t item_no = wi * 4 + wave;
        if (item_no >= nitems) break;
        const ScanItem it = a.items[item_no];
        // ---- per-query operands, lane m (both halves) for query m of the item
        const bool qok = (uint32_t)m < it.npair;
        uint32_t qrow = 0;
        unsigned long long row = 0;
        float u = __builtin_inff(), cq = 0.f;  // kept iff t > u: nothing passes for an absent query
        if (qok) {
            qrow = a.pair_query[it.pair_begin + m];
            row = a.pair_out[it.pair_begin + m] + it.vec_off;
            const float thr = a.thr[qrow], xn = a.xn[qrow];
            if (METRIC == METRIC_L2) {
                u = (1.f - C) * xn - thr;  // kept iff 2 p - (1 - C) yn > (1 - C) xn - thr   (NaN thr: nothing passes)
                if (!(thr == thr)) u = __builtin_inff();
            } else {
                u = thr;                   // kept iff p + C |x||y| > thr
                if (!(thr == thr)) u = __builtin_inff();
                cq = C * xn;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous item's reads of this wave's LDS rows are done
        __builtin_amdgcn_wave_barrier();
        if (h == 0) {
            s_u[wave][m] = u;
            s_c[wave][m] = cq;
            s_row[wave][m] = (uint32_t)row;
            s_q[wave][m] = qrow;
        }
        const float* qp = a.xf + (size_t)qrow * qstride + (size_t)h * 4;
        v4f af[NJ ? NJ : 1];
        if (NJ) {
#pragma unroll
            for (int j = 0; j < NJ; j++) af[j] = qok ? *reinterpret_cast<const v4f*>(qp + 8 * j) : v4f{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // thresholds in accumulator layout: register 4 g + i of lane half h belongs to query 8 g + 4 h + i
        float ur[16], cr[16];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const v4f tu = *reinterpret_cast<const v4f*>(&s_u[wave][8 * g + 4 * h]);
            const v4f tc = *reinterpret_cast<const v4f*>(&s_c[wave][8 * g + 4 * h]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ur[4 * g + i] = tu[i];
                cr[4 * g + i] = tc[i];
            }
        }
        const uint32_t nblk = ((it.nvec + 63) >> 6) * 2;  // lists are stored in pairs of blocks: a 64-candidate mask word is two of ours
        uint32_t* mask32 = reinterpret_cast<uint32_t*>(a.mask);
        const float* bbase = a.codes_frag + (size_t)it.vec_base * (size_t)J * 256 + (size_t)lane * 4;

        // ---- the tile's verdicts: mask words + survivor entries of block i, whose 32 x 32 dot products are in acc
        auto finish_block = [&](uint32_t i, const v16f& acc, float yn) __attribute__((always_inline)) {
            filter_verdicts<METRIC, HALF>(a, it, C, ps, i, 0, acc, yn, s_u[wave], s_c[wave], s_row[wave], s_q[wave], mask32, lane, sc);
        };
        auto load_yn = [&](uint32_t i) __attribute__((always_inline)) { return a.yn[(it.vec_base + i) * 32 + m]; };

        if constexpr (NJ != 0) {
            // P pieces of the list in flight, across block boundaries: piece t of the chunk's linear sequence lives in register
            // bq[t % P] (P divides NJ, so the index is static in the unrolled block body); the scheduling barriers keep every
            // load where it is written -- right behind the MFMAs that freed its register
            constexpr int P = NJ == 12 || NJ == 6 ? 6 : NJ >= 8 ? 8 : 4;
            v4f bq[P];
#pragma unroll
            for (int j = 0; j < P; j++) bq[j] = *(gv4f)(uintptr_t)(bbase + (size_t)j * 256);
            for (uint32_t i = 0; i < nblk; i++) {
                const float yn = load_yn(i);  // (needed behind the block's MFMAs: its latency hides under them)
                v16f acc;
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] = 0.f;
                const float* bp = bbase + (size_t)i * (size_t)NJ * 256;
                const float* bn = i + 1 < nblk ? bp + (size_t)NJ * 256 : bp;  // (past the chunk's end: re-read this block, unused)
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    __builtin_amdgcn_sched_barrier(0);
                    filter_mac<HALF>(acc, af[j], bq[j % P]);
                    __builtin_amdgcn_sched_barrier(0);
                    const float* src = j + P < NJ ? bp + (size_t)(j + P) * 256 : bn + (size_t)(j + P - NJ) * 256;
                    bq[j % P] = *(gv4f)(uintptr_t)src;  // (cached: the chunk's other query blocks read it too)
                }
                __builtin_amdgcn_sched_barrier(0);
                finish_block(i, acc, yn);
            }
        } else {
            // d > 128: the query operand does not fit the registers; it is re-read (L2) piece by piece, and every piece serves
            // four blocks of the list at once (four accumulator tiles), two pieces in flight
            constexpr int G = 4;
            for (uint32_t i0 = 0; i0 < nblk; i0 += G) {
                v16f acc[G];
#pragma unroll
                for (int g = 0; g < G; g++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[g][r] = 0.f;
                const uint32_t ng = nblk - i0 < (uint32_t)G ? nblk - i0 : (uint32_t)G;  // (2 or 4: lists come in pairs of blocks)
                float yng[G];
#pragma unroll
                for (int g = 0; g < G; g++) yng[g] = load_yn(i0 + ((uint32_t)g < ng ? (uint32_t)g : ng - 1));
                const float* bp = bbase + (size_t)i0 * (size_t)J * 256;
                const size_t bstep = (size_t)J * 256;  // floats from a block's piece j to the next block's
                // straight-line loads (no predicates: an absent query reads query row 0 and a block past the chunk's end re-reads
                // the last one; neither result is looked at); the scheduling barriers keep the loads of step j + 2 where they are
                // written -- behind the MFMAs of step j, in flight under those of step j + 1
                const float* bpg[G];
#pragma unroll
                for (int g = 0; g < G; g++) bpg[g] = bp + (size_t)((uint32_t)g < ng ? (uint32_t)g : ng - 1) * bstep;
                v4f a0, a1, b0[G], b1[G];
#define FILTER_FETCH(JJ, AV, BV)                                                                                 \
    {                                                                                                            \
        const int jj_ = (JJ);                                                                                    \
        AV = *(gv4f)(uintptr_t)(qp + 8 * jj_);                                                                   \
        _Pragma("unroll") for (int g = 0; g < G; g++)                                                            \
            BV[g] = *(gv4f)(uintptr_t)(bpg[g] + (size_t)jj_ * 256);                                            \
    }
#define FILTER_MAC(AV, BV)                                                                                       \
    {                                                                                                            \
        _Pragma("unroll") for (int g = 0; g < G; g++) filter_mac<HALF>(acc[g], AV, BV[g]);                       \
    }
                FILTER_FETCH(0, a0, b0)
                FILTER_FETCH(1, a1, b1)
                for (int j = 0; j < J; j += 2) {  // (J is even beyond 128 dimensions: filter_steps)
                    __builtin_amdgcn_sched_barrier(0);
                    FILTER_MAC(a0, b0)
                    __builtin_amdgcn_sched_barrier(0);
                    FILTER_FETCH(j + 2 < J ? j + 2 : j, a0, b0)
                    __builtin_amdgcn_sched_barrier(0);
                    FILTER_MAC(a1, b1)
                    __builtin_amdgcn_sched_barrier(0);
                    FILTER_FETCH(j + 3 < J ? j + 3 : j + 1, a1, b1)
                }
#undef FILTER_FETCH
#undef FILTER_MAC
                static_for(std::make_integer_sequence<int, G>{}, [&](auto Gi) {
                    constexpr int g = decltype(Gi)::value;
                    if ((uint32_t)g < ng) finish_block(i0 + g, acc[g], yng[g]);
                });
            }
        }
    }  // items
    surv_retire(a, sc, lane);
}

// ---------------------------------------------------------------------------------------------
// d > 128, the workgroup form: one work item = a chunk of <= 128 consecutive vectors of a list (FILTER_WIDE_VECTORS) x up to
// 128 of the queries probing it (FILTER_WIDE_QUERIES), computed by the four waves of a workgroup together.  Wave w owns block
// w of the chunk -- its 32 vectors stream from HBM once, straight into its registers, through a ring of FILTER_WIDE_SP pieces --
// against ALL the item's query blocks (up to 4 accumulator tiles); the query operand, which every wave needs, goes through
// LDS in stages of FILTER_WIDE_SP pieces (wave w fetches the rows of query block w: register staging, the write of stage
// s + 1 behind the MFMAs of stage s, one barrier per stage).  Two workgroups per CU: one's barriers, prologue and epilogue
// run under the other's MFMAs.  Against the one-wave form (32 queries x 128 vectors per pass) this reads the list once
// instead of once per query block and the query rows once per chunk instead of once per wave: PMC on the GIST-like
// configuration (d = 960, 71 queries per list): 16 GB fetched per pass over 3.84 GB of lists before.
constexpr int FILTER_WIDE_QB = FILTER_WIDE_QUERIES / 32;
#ifndef AUNCEL_FILTER_WIDE_SP
#define AUNCEL_FILTER_WIDE_SP 6  // (8 pieces: 376 registers where two waves per SIMD leave 256 -- 120 of them spilled, 6 GB of scratch traffic a pass)
#endif
constexpr int FILTER_WIDE_SP = AUNCEL_FILTER_WIDE_SP;
constexpr size_t FILTER_WIDE_STAGE_FLOATS = (size_t)FILTER_WIDE_QB * FILTER_WIDE_SP * 256;
constexpr size_t FILTER_WIDE_LDS = 2 * FILTER_WIDE_STAGE_FLOATS * sizeof(float) + 4 * FILTER_WIDE_QUERIES * sizeof(float);
static_assert(FILTER_WIDE_VECTORS == 128, "one 32-vector block per wave of the workgroup");

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load in flight (one counter for
// loads and stores on this ISA), which would drain the list prefetch ring at every stage
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int METRIC, bool HALF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void scan_filter_wide_kernel(FilterScanArgs a) {
    extern __shared__ __align__(16) unsigned char wide_smem[];
    float* s_a = reinterpret_cast<float*>(wide_smem);  // [2 buffers][query block][piece of the stage][lane][4]
    float* s_u = s_a + 2 * FILTER_WIDE_STAGE_FLOATS;
    float* s_c = s_u + FILTER_WIDE_QUERIES;
    uint32_t* s_row = reinterpret_cast<uint32_t*>(s_c + FILTER_WIDE_QUERIES);
    uint32_t* s_q = s_row + FILTER_WIDE_QUERIES;
    constexpr int QB = FILTER_WIDE_QB, SP = FILTER_WIDE_SP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int J = (int)(HALF ? filter_steps16(a.d) : filter_steps(a.d));
    const int S = (J + SP - 1) / SP;
    const size_t qstride = (size_t)J * 8;
    const FilterParams fp = HALF ? *reinterpret_cast<const FilterParams*>(a.params) : FilterParams{1.f, 1.f, 0.f, 0.f};
    const float C = HALF ? fp.C : (float)(2 * a.d + 32) * 5.9604644775390625e-08f;  // (2 d + 32) 2^-24
    const float ps = fp.ps;
    const uint32_t nitems = a.dev_nitems ? *a.dev_nitems : a.nitems;
    uint32_t* mask32 = reinterpret_cast<uint32_t*>(a.mask);
    SurvChunk sc;
    ItemWalk w(nitems, a.xcd_chunks);
    for (uint32_t wi = w.cur; wi < w.end; wi += w.step) {
        const ScanItem it = a.items[wi];
        const int nqb = (int)((it.npair + 31) >> 5);
        // ---- per-query operands of query block `wave`, lane m (both halves) for its query m
        const bool qok = (uint32_t)(wave * 32 + m) < it.npair;
        uint32_t qrow = 0;
        unsigned long long row = 0;
        float u = __builtin_inff(), cq = 0.f;  // kept iff t > u: nothing passes for an absent query
        if (qok) {
            qrow = a.pair_query[it.pair_begin + wave * 32 + m];
            row = a.pair_out[it.pair_begin + wave * 32 + m] + it.vec_off;
            const float thr = a.thr[qrow], xn = a.xn[qrow];
            if (METRIC == METRIC_L2) {
                u = (1.f - C) * xn - thr;
                if (!(thr == thr)) u = __builtin_inff();
            } else {
                u = thr;
                if (!(thr == thr)) u = __builtin_inff();
                cq = C * xn;
            }
        }
        lds_barrier();  // the previous item's readers of the LDS tables and stages are done
        if (h == 0) {
            s_u[wave * 32 + m] = u;
            s_c[wave * 32 + m] = cq;
            s_row[wave * 32 + m] = (uint32_t)row;
            s_q[wave * 32 + m] = qrow;
        }
        // Every wave runs the same sequence of loads, stages and barriers whatever its share of the item (a wave without a
        // query block stages row 0, one without a block of the chunk recomputes the first and drops the result): the
        // compiler's wait counts are exact only in straight-line code -- with the loads under conditions it waited for the
        // youngest load of the prefetch ring at every stage.
        const float* qp = a.xf + (size_t)qrow * qstride + (size_t)h * 4;
        v4f st[SP];
        auto stage_load = [&](int s) __attribute__((always_inline)) {
#pragma unroll
            for (int jj = 0; jj < SP; jj++) {
                const int j = s * SP + jj;
                st[jj] = *(gv4f)(uintptr_t)(qp + 8 * (j < J ? j : J - 1));
            }
        };
        auto stage_write = [&](int s, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int jj = 0; jj < SP; jj++)  // (pieces that pad the last stage contribute 0 x y)
                *reinterpret_cast<v4f*>(s_a + ((size_t)(buf * QB + wave) * SP + jj) * 256 + (size_t)lane * 4) =
                    s * SP + jj < J ? st[jj] : v4f{0.f, 0.f, 0.f, 0.f};
        };
        stage_load(0);
        stage_write(0, 0);
        const uint32_t nblk = ((it.nvec + 63) >> 6) * 2;  // (pairs of blocks, as the lists are stored)
        const bool active = (uint32_t)wave < nblk;
        const float* bbase = a.codes_frag + ((size_t)it.vec_base + (active ? (uint32_t)wave : 0u)) * (size_t)J * 256 + (size_t)lane * 4;

        const float yn = a.yn[((size_t)it.vec_base + (active ? (uint32_t)wave : 0u)) * 32 + m];
        auto finish_block = [&](uint32_t i, int qb, const v16f& acc) __attribute__((always_inline)) {
            filter_verdicts<METRIC, HALF>(a, it, C, ps, i, qb, acc, yn, s_u, s_c, s_row, s_q, mask32, lane, sc);
        };

        // ---- the contraction; the K loop is compiled per count of query blocks (their accumulator tiles are registers)
        v16f acc[QB];
#pragma unroll
        for (int q = 0; q < QB; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
        // this wave's block, a ring of SP pieces: the slot of piece j is refilled with piece j + SP right behind its MFMAs, so
        // SP KB per wave are on their way from HBM at any time (one piece ahead left the matrix cores 3/4 idle: that covers a
        // fraction of the memory latency)
        v4f br[SP];
        auto fetch_b = [&](int j, v4f& bv) __attribute__((always_inline)) {
            bv = *(gv4f)(uintptr_t)(bbase + (size_t)(j < J ? j : J - 1) * 256);
        };
#pragma unroll
        for (int jj = 0; jj < SP; jj++) fetch_b(jj, br[jj]);
        lds_barrier();  // stage 0 is in LDS
        auto run = [&](auto NQBc) __attribute__((always_inline)) {
            constexpr int NQB = decltype(NQBc)::value;
            v4f afA[NQB], afB[NQB];  // the query blocks' operands of an even and an odd piece, read from LDS one MFMA group ahead
            auto load_a = [&](int buf, int jj, v4f (&af)[NQB]) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < NQB; q++)
                    af[q] = *reinterpret_cast<const v4f*>(s_a + ((size_t)(buf * QB + q) * SP + jj) * 256 + (size_t)lane * 4);
            };
            auto mac = [&](const v4f (&af)[NQB], const v4f& bv) __attribute__((always_inline)) {
                if constexpr (HALF) {
#pragma unroll
                    for (int q = 0; q < NQB; q++) filter_mac<true>(acc[q], af[q], bv);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
#pragma unroll
                        for (int q = 0; q < NQB; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][e], bv[e], acc[q], 0, 0, 0);
                }
            };
            for (int s = 0; s < S; s++) {
                const int buf = s & 1;
                stage_load(s + 1);  // (behind the last stage: padding, written to the idle buffer and never read)
#pragma unroll
                for (int jj = 0; jj < SP; jj += 2) {
                    const int j = s * SP + jj;
                    if (jj == 0) load_a(buf, 0, afA);
                    load_a(buf, jj + 1, afB);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(afA, br[jj]);
                    __builtin_amdgcn_sched_barrier(0);
                    fetch_b(j + SP, br[jj]);
                    if (jj + 2 < SP) load_a(buf, jj + 2, afA);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(afB, br[jj + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    fetch_b(j + 1 + SP, br[jj + 1]);
                }
                stage_write(s + 1, buf ^ 1);  // (last read in stage s - 1: every wave is past that barrier)
                lds_barrier();
            }
        };
        switch (nqb) {
            case 1: run(std::integral_constant<int, 1>{}); break;
            case 2: run(std::integral_constant<int, 2>{}); break;
            case 3: run(std::integral_constant<int, 3>{}); break;
            default: run(std::integral_constant<int, 4>{}); break;
        }
        if (active) {
            static_for(std::make_integer_sequence<int, QB>{}, [&](auto T) __attribute__((always_inline)) {
                constexpr int q = decltype(T)::value;
                if (q < nqb) finish_block((uint32_t)wave, q, acc[q]);
            });
        }
    }  // items
    surv_retire(a, sc, lane);
}

// Up to 128 dimensions the one-wave form stays: the same workgroup scheme with the whole query operand in LDS (64 KB at d = 128)
// was measured slower in both rounds of an adaptive fp32 search (15 queries per list: 2.36 vs 1.25 ms; 70 per list: 3.90 vs
// 2.95 ms) -- a wave that holds its 32 queries in registers needs no barrier and no LDS traffic, and the re-reads of a chunk by
// the list's other query blocks are L2 hits of neighbouring waves.
static bool filter_narrow() { return getenv("AUNCEL_AMD_FILTER_NARROW") != nullptr; }
uint32_t filter_item_queries(int d) {
    return filter_steps(d) > 16 && !filter_narrow() ? FILTER_WIDE_QUERIES : MFMA_QBLOCK;
}
uint32_t filter_item_vectors(int d) {
    return filter_steps(d) > 16 && !filter_narrow() ? FILTER_WIDE_VECTORS : mfma_chunk();
}

void launch_scan_filter(const FilterScanArgs& a, hipStream_t s) {
    if (a.nitems == 0 && !a.dev_nitems) return;
    const unsigned nwg = (a.nitems + 3) / 4;
    const int J = (int)(a.half ? filter_steps16(a.d) : filter_steps(a.d));
    auto go = [&](auto kern, auto resc) {
        static const unsigned per_cu = 3;
        const size_t hwg = ((size_t)a.hint_nitems + a.hint_nitems / 8 + 3) / 4;
        const unsigned hinted = (unsigned)((hwg + 7) / 8) * 8 + 8;
        const dim3 grid(a.dev_nitems ? (a.hint_nitems ? hinted : resident_grid(per_cu)) : (a.xcd_chunks ? ((nwg + 7) / 8) * 8 : nwg)), block(256);
        LAUNCH(kern, grid, block, 0, s, a);
        LAUNCH(resc, dim3(resident_grid(4)), dim3(256), 0, s, a);
    };
    auto go_wide = [&](auto kern, auto resc) {  // one workgroup per item
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FILTER_WIDE_LDS);
        if (e != hipSuccess) throw std::runtime_error(std::string("filter kernel: cannot reserve LDS: ") + hipGetErrorString(e));
        const size_t hwg = (size_t)a.hint_nitems + a.hint_nitems / 8;
        const unsigned hinted = (unsigned)((hwg + 7) / 8) * 8 + 8;
        const dim3 grid(a.dev_nitems ? (a.hint_nitems ? hinted : resident_grid(2)) : (a.xcd_chunks ? ((a.nitems + 7) / 8) * 8 : a.nitems)), block(256);
        LAUNCH(kern, grid, block, FILTER_WIDE_LDS, s, a);
        LAUNCH(resc, dim3(resident_grid(4)), dim3(256), 0, s, a);
    };
    const bool wide = filter_steps(a.d) > 16 && !filter_narrow();  // (as the planner shaped the items: filter_item_queries)
    auto pick = [&](auto metric) {
        constexpr int M = decltype(metric)::value;
        if (a.half) {
            if (wide) return go_wide(scan_filter_wide_kernel<M, true>, rescore_kernel<M>);
            if (J <= 4) return go(scan_filter_kernel<M, 4, true>, rescore_kernel<M>);
            if (J <= 6) return go(scan_filter_kernel<M, 6, true>, rescore_kernel<M>);
            if (J <= 8) return go(scan_filter_kernel<M, 8, true>, rescore_kernel<M>);
            return go(scan_filter_kernel<M, 0, true>, rescore_kernel<M>);
        }
        if (J <= 4) return go(scan_filter_kernel<M, 4, false>, rescore_kernel<M>);
        if (J <= 8) return go(scan_filter_kernel<M, 8, false>, rescore_kernel<M>);
        if (J <= 12) return go(scan_filter_kernel<M, 12, false>, rescore_kernel<M>);
        if (J <= 16) return go(scan_filter_kernel<M, 16, false>, rescore_kernel<M>);
        if (!wide) return go(scan_filter_kernel<M, 0, false>, rescore_kernel<M>);
        return go_wide(scan_filter_wide_kernel<M, false>, rescore_kernel<M>);
    };
    if (a.metric == METRIC_L2) pick(std::integral_constant<int, METRIC_L2>{});
    else pick(std::integral_constant<int, METRIC_IP>{});
}

}  // namespace amdivf
