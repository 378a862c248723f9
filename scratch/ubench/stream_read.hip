// What a hand-written streaming read reaches on this device (the ceiling the scans' HBM fractions should be read against):
// every lane 16-byte loads, grid-stride over a buffer far larger than the caches, a few loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench/stream_read.hip -o scratch/ubench/stream_read && scratch/ubench/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT> __global__ __launch_bounds__(256) void rd(const v4f* __restrict__ p, size_t n16, float* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    for (; i < n16; i += stride) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;  // (keeps the loads)
}

template <int UNROLL, bool NT> int run(const v4f* p, size_t n16, float* out, int wg_per_cu) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((rd<UNROLL, NT>), dim3(256 * wg_per_cu), dim3(256), 0, 0, p, n16, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    printf("loads in flight %d, %s, %2d workgroups a CU: %.3f ms = %.2f TB/s\n", UNROLL, NT ? "nontemporal" : "plain      ", wg_per_cu, best,
           (double)n16 * 16 / 1e12 / (best / 1e3));
    return 0;
}
int main() {
    const size_t bytes = (size_t)8 << 30;
    v4f* p; float* out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(p, 1, bytes));
    const size_t n16 = bytes / 16;
    for (int w : {2, 4, 8}) {
        run<1, false>(p, n16, out, w);
        run<4, false>(p, n16, out, w);
        run<8, false>(p, n16, out, w);
        run<4, true>(p, n16, out, w);
        run<8, true>(p, n16, out, w);
    }
    return 0;
}
