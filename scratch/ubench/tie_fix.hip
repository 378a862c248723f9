// tie_fix_kernel alone: cost per replayed admission and per call, k = 100 (and 10), one query and 5000.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 [-DAUNCEL_TH_NS=0] -Iauncel_amd/csrc scratch/ubench/tie_fix.hip -o scratch/ubench/tie_fix
#include "../../auncel_amd/csrc/ivf_select.hip"
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace amdivf;

int main() {
    for (int k : {100, 10}) {
        for (uint32_t nq : {1u, 5000u}) {
            for (uint32_t nadm : {0u, 200u, 700u, 1400u}) {
                const uint32_t cap = 2048;
                std::vector<uint2> log((size_t)nq * cap);
                std::vector<uint32_t> cnt(nq, nadm), flag(nq, 1), pos(nq, 0);
                std::mt19937 rng(5);
                // a decreasing-ish stream: every entry beats the worst of the k before it (what an admission log is)
                for (uint32_t q = 0; q < nq; q++) {
                    std::vector<float> top;
                    for (uint32_t i = 0; i < nadm; i++) {
                        float v;
                        if (top.size() < (size_t)k) v = 1000.f + (float)(rng() % 100000);
                        else {
                            std::sort(top.begin(), top.end());
                            const float worst = top.back();
                            v = worst - 1.f - (float)(rng() % 50);
                            top.pop_back();
                        }
                        top.push_back(v);
                        log[(size_t)q * cap + i] = make_uint2(__builtin_bit_cast(uint32_t, v), i);
                    }
                }
                std::vector<float> fv((size_t)nq * k, FLT_MAX);
                std::vector<int64_t> fr((size_t)nq * k, -1);
                uint2* d_log; uint32_t *d_cnt, *d_flag, *d_pos; float *d_fv, *d_D; int64_t *d_fr, *d_I, *d_ids; uint64_t* d_lo;
                CK(hipMalloc(&d_log, log.size() * 8)); CK(hipMalloc(&d_cnt, nq * 4)); CK(hipMalloc(&d_flag, nq * 4)); CK(hipMalloc(&d_pos, nq * 4));
                CK(hipMalloc(&d_fv, fv.size() * 4)); CK(hipMalloc(&d_fr, fr.size() * 8)); CK(hipMalloc(&d_D, fv.size() * 4)); CK(hipMalloc(&d_I, fr.size() * 8));
                CK(hipMalloc(&d_ids, 8 * 4096)); CK(hipMalloc(&d_lo, 16));
                CK(hipMemcpy(d_log, log.data(), log.size() * 8, hipMemcpyHostToDevice));
                CK(hipMemcpy(d_cnt, cnt.data(), nq * 4, hipMemcpyHostToDevice));
                CK(hipMemcpy(d_pos, pos.data(), nq * 4, hipMemcpyHostToDevice));
                TieFixArgs a{};
                a.metric = METRIC_L2; a.k = k; a.nq = nq; a.nlist = 1; a.log = d_log; a.log_cap = cap; a.round = 0; a.final_pass = 1;
                a.log_cnt = d_cnt; a.log_snap = d_cnt; a.fin_round = d_cnt; a.fix_val = d_fv; a.fix_ref = d_fr; a.fix_pos = d_pos; a.tie_flag = d_flag;
                a.list_off = d_lo; a.ids = d_ids; a.store_pairs = 0; a.identity_ids = 1; a.D = d_D; a.I = d_I;
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                float best = 1e9f;
                for (int rep = 0; rep < 6; rep++) {
                    CK(hipMemcpy(d_flag, flag.data(), nq * 4, hipMemcpyHostToDevice));
                    CK(hipMemcpy(d_fv, fv.data(), fv.size() * 4, hipMemcpyHostToDevice));
                    CK(hipMemcpy(d_fr, fr.data(), fr.size() * 8, hipMemcpyHostToDevice));
                    CK(hipEventRecord(e0, 0));
                    launch_tie_fix(a, 0);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                std::vector<float> D((size_t)nq * k);
                CK(hipMemcpy(D.data(), d_D, D.size() * 4, hipMemcpyDeviceToHost));
                double sum = 0; for (int i = 0; i < k; i++) sum += D[i] == FLT_MAX ? 0 : D[i];
                printf("k %3d nq %4u admissions %4u: %.4f ms  (checksum %.1f)\n", k, nq, nadm, best, sum);
                hipFree(d_log); hipFree(d_cnt); hipFree(d_flag); hipFree(d_pos); hipFree(d_fv); hipFree(d_fr); hipFree(d_D); hipFree(d_I); hipFree(d_ids); hipFree(d_lo);
            }
        }
    }
    return 0;
}
