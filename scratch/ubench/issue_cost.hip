// What one wave alone on a SIMD pays per instruction, by kind (ns per instruction from event time over a long loop):
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench/issue_cost.hip -o scratch/ubench/issue_cost && scratch/ubench/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))

template <int MODE> __global__ void k(uint32_t n, uint32_t* out) {
    uint32_t s = n, t = 1, v = threadIdx.x, w = threadIdx.x * 3;
    for (uint32_t i = 0; i < n; i++) {
        if (MODE == 0) {  // 16 dependent scalar adds
            asm volatile(R16("s_add_u32 %0, %0, %1\n\t") : "+s"(s) : "s"(t) : "scc");
        } else if (MODE == 1) {  // 16 x (compare + taken forward branch)
            asm volatile(R16("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 1\n1:\n\t") : "+s"(s) : : "scc");
        } else if (MODE == 2) {  // 16 x (compare + branch not taken)
            asm volatile(R16("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 0\n1:\n\t") : "+s"(s) : : "scc");
        } else if (MODE == 3) {  // 16 x (readlane -> scalar add -> writelane with it)
            asm volatile(R16("v_readlane_b32 %0, %1, 5\n\ts_add_u32 %0, %0, 1\n\tv_writelane_b32 %1, %0, 7\n\t") : "+s"(s), "+v"(v) : : "scc");
        } else if (MODE == 4) {  // 16 dependent vector adds
            asm volatile(R16("v_add_u32 %0, %0, %1\n\t") : "+v"(v) : "v"(w));
        } else if (MODE == 5) {  // 16 dependent bpermutes
            asm volatile(R16("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(v) : "v"(w));
        } else if (MODE == 6) {  // 16 dependent DPP moves
            asm volatile(R16("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t") : "+v"(v));
        } else if (MODE == 7) {  // 16 x (vector compare to SGPR pair -> scalar popcount -> scalar add)
            unsigned long long m;
            asm volatile(R16("v_cmp_gt_u32 %2, %1, %3\n\ts_bcnt1_i32_b64 %0, %2\n\ts_add_u32 %0, %0, 1\n\t") : "+s"(s), "+v"(v), "=&s"(m) : "v"(w) : "scc");
        } else if (MODE == 8) {  // 16 independent vector moves
            uint32_t a, b, c, d;
            asm volatile(R4("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4\n\t") : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(w));
            v += a + b + c + d;
        } else if (MODE == 9) {  // 16 x (scalar select chain: cmp + cselect)
            asm volatile(R16("s_cmp_gt_u32 %0, %1\n\ts_cselect_b32 %0, %0, %1\n\t") : "+s"(s) : "s"(t) : "scc");
        } else if (MODE == 10) {  // 16 x exec-masked vector move (set exec from a scalar pair, move, restore)
            unsigned long long m = 1ull << (i & 63), sv;
            asm volatile(R16("s_mov_b64 %2, exec\n\ts_mov_b64 exec, %1\n\tv_mov_b32 %0, 7\n\ts_mov_b64 exec, %2\n\t") : "+v"(v), "+s"(m), "=&s"(sv));
        }
    }
    out[threadIdx.x] = s + v;
}

template <int MODE> int run(const char* what, int per_iter) {
    uint32_t* d; CK(hipMalloc(&d, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t n = 200000;
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, n, d);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-70s %6.2f ns per instruction (%d per iteration)\n", what, best * 1e6 / ((double)n * per_iter), per_iter);
    return 0;
}
int main() {
    run<0>("dependent s_add_u32", 16);
    run<1>("s_cmp + s_cbranch TAKEN (skips one instruction)", 32);
    run<2>("s_cmp + s_cbranch not taken + s_add", 48);
    run<3>("v_readlane -> s_add -> v_writelane chain", 48);
    run<4>("dependent v_add_u32", 16);
    run<5>("dependent ds_bpermute_b32 + waitcnt", 16);
    run<6>("dependent v_mov_b32_dpp", 16);
    run<7>("v_cmp -> s_bcnt1 -> s_add chain", 48);
    run<8>("independent v_mov_b32", 16);
    run<9>("s_cmp + s_cselect chain", 32);
    run<10>("save exec, set exec, v_mov, restore exec", 64);
    return 0;
}
