// Device-side helpers shared by the scan, selection and coarse-ranking kernels (gfx950).
#pragma once
#include "ivf_kernels.h"

#include <float.h>
#include <utility>

namespace amdivf {

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

__device__ __forceinline__ uint32_t fkey(float x) {
    const uint32_t u = __float_as_uint(x);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t key) { return __uint_as_float((key & 0x80000000u) ? key ^ 0x80000000u : ~key); }
template <bool IsMax> __device__ __forceinline__ bool kcmp(uint32_t a, uint32_t b) { return IsMax ? a > b : a < b; }

__device__ __forceinline__ float rl_f(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ int rl_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ uint32_t rl_u(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }

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

// order keys: smaller key <=> better candidate, whatever the metric
template <bool IsMax> __device__ __forceinline__ uint32_t okey(float x) {
    const uint32_t kx = fkey(x);
    return IsMax ? kx : ~kx;
}
template <bool IsMax> __device__ __forceinline__ float okey_inv(uint32_t key) { return fkey_inv(IsMax ? key : ~key); }

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


// The XCD this wave runs on, and adds that are served by that XCD's L2 (workgroup scope: no trip to the memory side).  Only for
// words that no other XCD touches during the launch (per-XCD tables: ivf_plan.hip, STATS_ROWS).
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ uint32_t xcd_local_add(uint32_t* p, uint32_t v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void xcd_local_add64(unsigned long long* p, unsigned long long v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// reg[lane LANE] = val (val wave-uniform, LANE a compile-time constant: an inline operand, so the one SGPR slot is val's)
template <int LANE> __device__ __forceinline__ void writelane_c(int& reg, uint32_t val) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(reg) : "s"(val), "n"(LANE));
}
// the reference's distance between two fp32 rows (utils_simd.cpp:391-443): sums 0..3 over elements 4 i + l, products and sums
// rounded separately (this file is built with -ffp-contract=off), (s0 + s1) + (s2 + s3); rows are zero-padded to dpad (% 4)
template <int METRIC> __device__ __forceinline__ float exact_distance(const float* x, const float* y, int dpad) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    auto step = [&](const float4& a, const float4& b) {
        if (METRIC == METRIC_L2) {
            const float t0 = b.x - a.x, t1 = b.y - a.y, t2 = b.z - a.z, t3 = b.w - a.w;
            s0 += t0 * t0;
            s1 += t1 * t1;
            s2 += t2 * t2;
            s3 += t3 * t3;
        } else {
            s0 += b.x * a.x;
            s1 += b.y * a.y;
            s2 += b.z * a.z;
            s3 += b.w * a.w;
        }
    };
    // four steps per trip: their eight loads are requested together, the sums run in order (a lane's rows are its own: one step at
    // a time the loop paid a trip to memory per step)
    int c = 0;
    for (; c + 12 < dpad; c += 16) {
        float4 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            av[u] = *reinterpret_cast<const float4*>(x + c + 4 * u);
            bv[u] = *reinterpret_cast<const float4*>(y + c + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) step(av[u], bv[u]);
    }
    for (; c < dpad; c += 4) step(*reinterpret_cast<const float4*>(x + c), *reinterpret_cast<const float4*>(y + c));
    return (s0 + s1) + (s2 + s3);
}


template <int... I, class F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}


}  // namespace amdivf
