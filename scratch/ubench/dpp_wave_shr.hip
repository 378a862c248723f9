#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(uint32_t* out, uint32_t carry) {
    uint32_t x = threadIdx.x * 3 + 1;
    uint32_t old = carry;
    uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, 0x138, 0xf, 0xf, false);
    out[threadIdx.x] = y;
}
int main() {
    uint32_t* d; hipMalloc(&d, 256);
    k<<<1, 64>>>(d, 777);
    uint32_t h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) { uint32_t w = i == 0 ? 777 : (i - 1) * 3 + 1; if (h[i] != w) bad++; }
    printf("wave_shr1 %s (lane0 %u lane1 %u lane32 %u lane63 %u)\n", bad ? "BAD" : "ok", h[0], h[1], h[32], h[63]);
    return bad;
}
