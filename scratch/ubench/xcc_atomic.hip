// Are returning integer atomics at workgroup scope served by the issuing XCD's L2, and are they coherent among the CUs of that XCD?
// Every thread adds 1 to counter[xcc][key] (key = a hash of its index, `nkeys` addresses) and keeps the value returned; per
// (xcc, key) the returned values must be exactly 0 .. count - 1.  Timed against the same adds at agent scope on one shared table.
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench/xcc_atomic.hip -o /tmp/xcc_atomic && /tmp/xcc_atomic
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
template <int SCOPE> __global__ void adds(uint32_t* counters, uint32_t nkeys, uint32_t* slot, uint32_t* xcc_of, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = (i * 2654435761u) % nkeys;
    const uint32_t x = SCOPE ? xcc_id() : 0u;
    uint32_t s;
    if (SCOPE) s = __hip_atomic_fetch_add(&counters[x * nkeys + key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else s = __hip_atomic_fetch_add(&counters[key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    slot[i] = s;
    xcc_of[i] = x;
}
int main() {
    const uint32_t n = 1u << 20, nkeys = 4096;
    uint32_t *c, *slot, *xo;
    CK(hipMalloc(&c, 16 * nkeys * 4)); CK(hipMalloc(&slot, n * 4)); CK(hipMalloc(&xo, n * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int scope = 0; scope < 2; scope++) {
        float best = 1e9f;
        std::vector<uint32_t> hs(n), hx(n), hc(16 * nkeys);
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemset(c, 0, 16 * nkeys * 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            if (scope) adds<1><<<n / 256, 256>>>(c, nkeys, slot, xo, n); else adds<0><<<n / 256, 256>>>(c, nkeys, slot, xo, n);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
        }
        CK(hipMemcpy(hs.data(), slot, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), xo, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc.data(), c, 16 * nkeys * 4, hipMemcpyDeviceToHost));
        // check: per (xcc, key) the slots are a permutation of 0 .. count - 1
        std::vector<std::vector<uint32_t>> got(16 * (size_t)nkeys);
        for (uint32_t i = 0; i < n; i++) got[(size_t)hx[i] * nkeys + (i * 2654435761u) % nkeys].push_back(hs[i]);
        size_t bad = 0, total = 0; uint32_t xccs = 0;
        for (size_t k = 0; k < got.size(); k++) {
            auto& v = got[k]; std::sort(v.begin(), v.end()); total += hc[k];
            if (v.size() != hc[k]) bad++;
            for (size_t j = 0; j < v.size(); j++) if (v[j] != j) { bad++; break; }
            if (!v.empty()) xccs |= 1u << (k / nkeys);
        }
        printf("%s scope: %.3f ms for %u returning adds on %u addresses%s; counters sum %zu (want %u), tables with a wrong slot set %zu, xcc mask 0x%x\n",
               scope ? "workgroup" : "agent", best, n, nkeys, scope ? " x XCDs" : "", total, n, bad, xccs);
    }
    return 0;
}
