// v_mfma_i32_32x32x32_i8 on gfx950: operand / result lane maps checked with exact integer data (signed bytes,
// asymmetric operands), and the back-to-back issue rate next to v_dot4_u32_u8 and v_pk_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench/mfma_i8.hip -o scratch/ubench/mfma_i8 && scratch/ubench/mfma_i8
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// one wave, one instruction: lane l gives A[l & 31][16 (l >> 5) + j] and B[16 (l >> 5) + j][l & 31], j = 0..15
__global__ void layout_kernel(const int8_t* A, const int8_t* B, int* C) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v4i a, b;
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; j++) {
        ab[j] = A[r * 32 + 16 * h + j];
        bb[j] = B[(16 * h + j) * 32 + r];
    }
    __builtin_memcpy(&a, ab, 16);
    __builtin_memcpy(&b, bb, 16);
    v16i c;
    for (int i = 0; i < 16; i++) c[i] = 0;
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int reg = 0; reg < 16; reg++) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        C[row * 32 + r] = c[reg];
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(int* out, int seed, int iters) {
    v4i a = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7}, b = {seed * 11, seed * 13, (int)threadIdx.x, seed};
    v16i c0, c1, c2, c3;
    for (int i = 0; i < 16; i++) c0[i] = c1[i] = c2[i] = c3[i] = i;
    uint32_t d[16];
    for (int i = 0; i < 16; i++) d[i] = i;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) d[i] = __builtin_amdgcn_udot4((uint32_t)a[0], (uint32_t)b[0] + i, d[i], false);
        }
    }
    int s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i] + (int)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    // ---- lane maps
    std::vector<int8_t> A(32 * 32), B(32 * 32);
    srand(7);
    for (auto& x : A) x = (int8_t)(rand() % 256 - 128);
    for (auto& x : B) x = (int8_t)(rand() % 256 - 128);
    int8_t *dA, *dB;
    int* dC;
    hipMalloc(&dA, 1024);
    hipMalloc(&dB, 1024);
    hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<int> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad_signed = 0, bad_unsigned = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            int s = 0, u = 0;
            for (int k = 0; k < 32; k++) {
                s += (int)A[i * 32 + k] * (int)B[k * 32 + j];
                u += (int)(uint8_t)A[i * 32 + k] * (int)(uint8_t)B[k * 32 + j];
            }
            bad_signed += C[i * 32 + j] != s;
            bad_unsigned += C[i * 32 + j] != u;
        }
    printf("v_mfma_i32_32x32x32_i8 lane maps (A[l&31][16(l>>5)+j], B[16(l>>5)+j][l&31], C row (reg&3)+8(reg>>2)+4(l>>5) col l&31): "
           "%d of 1024 differ from the signed-byte product, %d from the unsigned one\n", bad_signed, bad_unsigned);

    // ---- issue rates
    int* d;
    hipMalloc(&d, 4 * 256 * 4096);
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int iters = mode == 0 ? 4000 : 2000, blocks = 4096;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, d, 3, iters);
            else hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, d, 3, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (mode == 0) {
            const double inst = (double)blocks * 4 * iters * 4.0;  // wave-level MFMAs
            const double macs = inst * 32.0 * 32.0 * 32.0;
            printf("v_mfma_i32_32x32x32_i8: %.3f ms, %.2f G MFMA/s, %.1f T MAC/s = %.1f TOP/s; at d = 128: %.1f G distances/s\n", ms, inst / ms / 1e6,
                   macs / ms / 1e9, 2 * macs / ms / 1e9, macs / 128.0 / ms / 1e6);
        } else {
            const double inst = (double)blocks * 4 * iters * 64.0;
            printf("v_dot4_u32_u8: %.3f ms, %.1f G wave-instr/s, %.1f T MAC/s; at d = 128: %.1f G distances/s\n", ms, inst / ms / 1e6,
                   inst * 64 * 4 / ms / 1e9, inst * 64 / 32.0 / ms / 1e6);
        }
    }
    return 0;
}
