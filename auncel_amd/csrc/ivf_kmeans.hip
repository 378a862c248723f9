// k-means centroid update on the device, in the reference's summation order (km_update_centroids, utils.cpp:1087-1124):
// centroid c = fp32 sum of its points in ascending point order, divided by their count.  The points of every centroid are
// brought together by a stable radix sort of (centroid, point index) pairs; then one thread per (centroid, dimension)
// walks its run of points -- a wave reads 64 consecutive dimensions of one point at a time, so the loads are coalesced.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "ivf_kernels.h"

namespace amdivf {

__global__ __launch_bounds__(256) void kmeans_keys_kernel(const int64_t* assign, size_t n, uint32_t* keys, uint32_t* idx, uint32_t* counts) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = (uint32_t)assign[i];
    keys[i] = c;
    idx[i] = (uint32_t)i;
    atomicAdd(&counts[c], 1u);
}

__global__ __launch_bounds__(64) void kmeans_sums_kernel(const float* x, size_t stride, int d, const uint32_t* idx, const uint32_t* seg_off,
                                                         float* centroids) {
    const uint32_t c = blockIdx.x;
    const int j = blockIdx.y * 64 + threadIdx.x;
    if (j >= d) return;
    const uint32_t b = seg_off[c], e = seg_off[c + 1];
    float s = 0.f;
    for (uint32_t p = b; p < e; p++) s += x[(size_t)idx[p] * stride + j];
    const float ni = (float)(e - b);
    centroids[(size_t)c * d + j] = ni != 0 ? s / ni : 0.f;
}

size_t kmeans_sort_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (int)n, 0, 32, (hipStream_t) nullptr);
    return bytes;
}

void launch_kmeans_group(const int64_t* assign, size_t n, uint32_t k, uint32_t* keys_in, uint32_t* keys_out, uint32_t* idx_in,
                         uint32_t* idx_out, uint32_t* counts, void* temp, size_t temp_bytes, hipStream_t s) {
    (void)hipMemsetAsync(counts, 0, (size_t)k * 4, s);
    LAUNCH(kmeans_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, assign, n, keys_in, idx_in, counts);
    int bits = 1;
    while ((1ull << bits) < k) bits++;
    (void)hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, (int)n, 0, bits, s);
}

void launch_kmeans_sums(const float* x, size_t stride, int d, const uint32_t* idx_sorted, const uint32_t* seg_off, uint32_t k, float* centroids,
                        hipStream_t s) {
    LAUNCH(kmeans_sums_kernel, dim3(k, (unsigned)((d + 63) / 64)), dim3(64), 0, s, x, stride, d, idx_sorted, seg_off, centroids);
}

}  // namespace amdivf
