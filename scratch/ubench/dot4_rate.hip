// peak issue rate of v_dot4_u32_u8 and v_pk_fma_f32 on one chip: independent accumulator chains, no memory traffic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
    uint32_t acc[16];
    float2 facc[16];
    uint32_t a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < 16; i++) { acc[i] = i; facc[i] = make_float2(i, i); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_udot4(a, b + i, acc[i], false);
                else {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 x = {facc[i].x, facc[i].y}, y = {(float)a, (float)b};
                    x = __builtin_elementwise_fma(x, y, x);
                    facc[i].x = x.x; facc[i].y = x.y;
                }
            }
    }
    uint32_t s = 0;
    for (int i = 0; i < 16; i++) s += acc[i] + (uint32_t)facc[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    uint32_t* d; hipMalloc(&d, 4 * 256 * 4096);
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 2000, blocks = 4096;
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 1u, 2u, 10); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 1u, 2u, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 1u, 2u, iters); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 1u, 2u, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst = (double)blocks * 4 /*waves*/ * iters * 64.0;  // wave instructions
        printf("%s: %.3f ms, %.1f G wave-instr/s, %.2f T lane-ops/s\n", mode == 0 ? "v_dot4_u32_u8" : "v_pk_fma_f32", ms, inst / ms / 1e6, inst * 64 / ms / 1e9);
    }
    return 0;
}
