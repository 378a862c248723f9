// Dependent LDS reads by one wave (pointer chase): cycles per hop for ds_read_b32 / b64 / b128, alone on a SIMD.
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench/lds_latency.hip -o scratch/ubench/lds_latency && scratch/ubench/lds_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int W> __global__ __launch_bounds__(64) void chase(unsigned long long* out, int hops) {
    __shared__ __align__(16) uint32_t a[4096 * 4];
    for (int i = threadIdx.x; i < 4096; i += 64) {
        a[i * 4] = (uint32_t)((i * 97 + 13) & 4095);
        a[i * 4 + 1] = a[i * 4 + 2] = a[i * 4 + 3] = (uint32_t)i;
    }
    __syncthreads();
    uint32_t p = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int h = 0; h < hops; h++) {
        if (W == 1) p = a[p * 4];
        else if (W == 2) {
            const uint2 v = *reinterpret_cast<const uint2*>(&a[p * 4]);
            p = v.x + (v.y & 0);
        } else {
            const uint4 v = *reinterpret_cast<const uint4*>(&a[p * 4]);
            p = v.x + ((v.y ^ v.w) & 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (p == 0xffffffffu) out[1] = p;
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 64));
    unsigned long long h[2];
    const int hops = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 1; w <= 3; w++) {
        CK(hipEventRecord(e0));
        if (w == 1) chase<1><<<1, 64>>>(d, hops); else if (w == 2) chase<2><<<1, 64>>>(d, hops); else chase<4><<<1, 64>>>(d, hops);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
        printf("ds_read_b%d: %.1f counter ticks per hop, %.1f ns per hop (kernel %.3f ms)\n", w == 3 ? 128 : 32 * w, (double)h[0] / hops, ms * 1e6 / hops, ms);
    }
    return 0;
}
