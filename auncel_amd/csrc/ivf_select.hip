// gfx950 (MI355X, CDNA4) selection kernels of the IVF-Flat search engine.
//
// Selection contract: the reference keeps its k best in a binary heap that only admits strictly
// better candidates (Heap.h:88-142, IndexIVFFlat.cpp:125-135); which of several equal distances
// survives, and the order of equal distances in the output, depend on the heap's history.
#include "ivf_dev.h"

#include <algorithm>
#include <stdexcept>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <type_traits>
#include <utility>

namespace amdivf {

// =============================================================================================
// K-replay: ordered selection + Auncel stop rule + training samples
// =============================================================================================
// ---------------------------------------------------------------------------------------------
// The reference's heap resident in registers (k <= 127): node i (1-based, Heap.h numbering) lives in lane i & 63 of register
// i >> 6, as an order key -- key(x) is an unsigned integer with key(a) < key(b) <=> a < b, and the float comes back bit for bit
// from the key -- next to the slot (0..k-1) of its 64-bit id in an LDS table, so ids never move.  (Keys order -0.0 below +0.0
// where floats call them equal; distances from the scan kernels are never -0.0.  NaN never enters: admission is tested on the
// floats.)  This is the form in which a heap is handed between kernels' phases (rh_store); the walks themselves are VHeap's.
struct RegHeap {
    uint32_t v0, v1;  // keys
    uint32_t s0, s1;  // id slots
};

// reg[lane l] = val (val and l wave-uniform).  The lane select goes through M0: v_writelane_b32 may name one SGPR.  M0 is a
// register the compiler reserves for itself (it cannot be named as clobbered: "reserved registers on the clobber list may not
// be preserved"), so every block below saves it into a scratch SGPR and puts it back -- two scalar moves per block.
__device__ __forceinline__ void wl_u(uint32_t& reg, uint32_t val, int l) {
    uint32_t m0s;
    asm("s_mov_b32 %[m], m0\n\ts_mov_b32 m0, %[l]\n\ts_nop 0\n\tv_writelane_b32 %[r], %[v], m0\n\ts_mov_b32 m0, %[m]"
        : [r] "+v"(reg), [m] "=&s"(m0s)
        : [v] "s"(val), [l] "s"(l));
}

// two registers, same lane: one M0 set-up
__device__ __forceinline__ void wl2_u(uint32_t& r0, uint32_t v0, uint32_t& r1, uint32_t v1, int l) {
    uint32_t m0s;
    asm("s_mov_b32 %[m], m0\n\ts_mov_b32 m0, %[l]\n\ts_nop 0\n\tv_writelane_b32 %[r0], %[v0], m0\n\tv_writelane_b32 %[r1], %[v1], m0\n\t"
        "s_mov_b32 m0, %[m]"
        : [r0] "+v"(r0), [r1] "+v"(r1), [m] "=&s"(m0s)
        : [v0] "s"(v0), [v1] "s"(v1), [l] "s"(l));
}

// ---------------------------------------------------------------------------------------------
// The heap's walks as VECTOR work without a branch (k <= 127): tie_fix_kernel's replay and replay_kernel's admissions.  Walked
// node by node (rounds 2-4: v_readlane / v_writelane on wave-uniform indices, the sift loops on the scalar unit) a level is two
// lane reads, a compare and a taken branch: ~520 ns a pop + push at k = 100 on a wave that runs alone on its SIMD
// (scratch/ubench/tie_fix.hip; issue_cost.hip: ~2.5 ns an instruction, ~11 ns a taken branch); this form takes 290.  Here a pop
// is one pass of all lanes over all nodes:
//   1. every node compares itself with its sibling (one DPP): w(i) = "i is the child its parent would pick" (Heap.h:100: the left
//      one iff it is strictly worse-ranked than the right; absent nodes hold key 0 and never win);
//   2. a node is on the root's path iff w holds for it and all its ancestors: the ballots of w against per-lane constant masks;
//   3. the value v being re-placed stops above the first path node that is better than it (Heap.h:102,109: "if cmp(val, child)
//      break"); keys fall along the path, so the nodes that move up are the path's prefix with key >= v: each path node whose picked
//      child is one of them takes that child's entry (the pair winners gathered to their parents with ds_bpermute), and v lands in
//      the deepest of them (or the root).
// Keys are stored so that the heap is always a max-heap (L2: fkey, IP: ~fkey); payloads are 32-bit global positions (0xffffffff: an
// empty entry, id -1).  Node i: lane i & 63 of register i >> 6.
struct VHeap {
    uint32_t k0, k1, p0, p1;
};
struct VHeapLane {        // per-lane constants
    uint32_t a0lo, a0hi;  // node `lane`: its ancestors and itself, root excluded, as bits over nodes 0..63
    uint32_t a1lo, a1hi;  // node 64 + lane: its ancestors (all below 64)
    uint32_t caddr;       // ds_bpermute address of the lane holding this node's left child (register by lane < 32)
    uint32_t paddr0, paddr1;  // ... of the lane holding the parent of node `lane` / of node 64 + lane
};
__device__ __forceinline__ VHeapLane vh_lane(int lane) {
    VHeapLane c;
    unsigned long long a0 = lane == 0 ? 1ull : 0ull, a1 = 0ull;
    for (int i = lane; i > 1; i >>= 1) a0 |= 1ull << i;
    for (int i = (64 + lane) >> 1; i > 1; i >>= 1) a1 |= 1ull << i;
    c.a0lo = (uint32_t)a0, c.a0hi = (uint32_t)(a0 >> 32), c.a1lo = (uint32_t)a1, c.a1hi = (uint32_t)(a1 >> 32);
    c.caddr = (uint32_t)((2 * lane) & 63) * 4u;
    c.paddr0 = (uint32_t)(lane >> 1) * 4u;
    c.paddr1 = (uint32_t)(32 + (lane >> 1)) * 4u;
    return c;
}
__device__ __forceinline__ uint32_t vh_sel(unsigned long long mask, uint32_t yes, uint32_t no) {  // per lane: mask bit ? yes : no
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(no), "v"(yes), "s"(mask));
    return r;
}
__device__ __forceinline__ uint32_t vh_sib(uint32_t x) {  // the value of lane ^ 1
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);  // (every lane has a source: `old` is never used)
}
template <bool TWO> __device__ __forceinline__ uint32_t vh_key_at(const VHeap& h, int node) {
    if (!TWO) return rl_u(h.k0, node);
    const uint32_t a = rl_u(h.k0, node & 63), b = rl_u(h.k1, node & 63);
    return node < 64 ? a : b;
}
template <bool TWO> __device__ __forceinline__ uint32_t vh_pay_at(const VHeap& h, int node) {
    if (!TWO) return rl_u(h.p0, node);
    const uint32_t a = rl_u(h.p0, node & 63), b = rl_u(h.p1, node & 63);
    return node < 64 ? a : b;
}
// Heap.h:88-118: the entry (v, pv) of the last node is re-placed from the root down (the last node itself still takes part, as there)
template <bool TWO> __device__ __forceinline__ void vh_pop(VHeap& h, const VHeapLane& c, int lane, uint32_t v, uint32_t pv) {
    constexpr unsigned long long EVEN = 0x5555555555555555ull;
    const uint32_t s0 = vh_sib(h.k0), sp0 = vh_sib(h.p0);
    const unsigned long long w0 = (__ballot(h.k0 > s0) & EVEN) | (__ballot(h.k0 >= s0) & ~EVEN);
    const uint32_t t0 = vh_sel(w0, h.k0, s0), tp0 = vh_sel(w0, h.p0, sp0);  // the pair's winner, in both its lanes
    uint32_t gk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.caddr, (int)t0);
    uint32_t gp = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.caddr, (int)tp0);
    unsigned long long w1 = 0, path1 = 0;
    const unsigned long long low32 = 0xffffffffull;
    if (TWO) {
        const uint32_t s1 = vh_sib(h.k1), sp1 = vh_sib(h.p1);
        w1 = (__ballot(h.k1 > s1) & EVEN) | (__ballot(h.k1 >= s1) & ~EVEN);
        const uint32_t t1 = vh_sel(w1, h.k1, s1), tp1 = vh_sel(w1, h.p1, sp1);
        const uint32_t gk1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.caddr, (int)t1);
        const uint32_t gp1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.caddr, (int)tp1);
        gk = vh_sel(low32, gk, gk1);  // nodes 0..31: children in register 0; 32..63: in register 1
        gp = vh_sel(low32, gp, gp1);
    } else {
        gk = vh_sel(low32, gk, 0u);   // (nodes 32..63 have no children below 64)
    }
    const uint32_t nlo = ~(uint32_t)w0, nhi = ~(uint32_t)(w0 >> 32);  // on the path <=> none of the node's ancestor bits is missing from w
    const unsigned long long path0 = __ballot(((nlo & c.a0lo) | (nhi & c.a0hi)) == 0u);
    if (TWO) path1 = __ballot(((nlo & c.a1lo) | (nhi & c.a1hi)) == 0u) & w1;
    const unsigned long long recv = path0 & __ballot(gk >= v);              // its picked child moves up into it
    const unsigned long long land0 = path0 & ~recv & (__ballot(h.k0 >= v) | 2ull);
    h.k0 = vh_sel(land0, v, vh_sel(recv, gk, h.k0));
    h.p0 = vh_sel(land0, pv, vh_sel(recv, gp, h.p0));
    if (TWO) {
        const unsigned long long land1 = path1 & __ballot(h.k1 >= v);
        h.k1 = vh_sel(land1, v, h.k1);
        h.p1 = vh_sel(land1, pv, h.p1);
    }
}
// Heap.h:125-142: (nv, np) enters at node k and climbs while it is worse than its father.  chain0 / chain1: node k and its ancestors
// as lane masks of the two registers (wave-uniform constants of the kernel)
template <bool TWO> __device__ __forceinline__ void vh_push(VHeap& h, const VHeapLane& c, int k, unsigned long long chain0, unsigned long long chain1,
                                                             uint32_t nv, uint32_t np) {
    const unsigned long long kbit0 = k < 64 ? 1ull << k : 0ull, kbit1 = k >= 64 ? 1ull << (k - 64) : 0ull;
    if (!(nv > vh_key_at<false>(h, k >> 1)) || k == 1) {  // (the father of node k is below 64) -- the common case: it stays at node k
        h.k0 = vh_sel(kbit0, nv, h.k0);
        h.p0 = vh_sel(kbit0, np, h.p0);
        if (TWO) {
            h.k1 = vh_sel(kbit1, nv, h.k1);
            h.p1 = vh_sel(kbit1, np, h.p1);
        }
        return;
    }
    const uint32_t f0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.paddr0, (int)h.k0);  // the father's entry, per node
    const uint32_t fp0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.paddr0, (int)h.p0);
    const unsigned long long down0 = chain0 & ~2ull & __ballot(f0 < nv);       // the father comes down into this node
    const unsigned long long land0 = chain0 & (__ballot(h.k0 < nv) | kbit0) & ~down0;
    if (TWO) {
        const uint32_t f1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.paddr1, (int)h.k0);
        const uint32_t fp1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.paddr1, (int)h.p0);
        const unsigned long long down1 = chain1 & __ballot(f1 < nv);
        const unsigned long long land1 = chain1 & (__ballot(h.k1 < nv) | kbit1) & ~down1;
        h.k1 = vh_sel(land1, nv, vh_sel(down1, f1, h.k1));
        h.p1 = vh_sel(land1, np, vh_sel(down1, fp1, h.p1));
    }
    h.k0 = vh_sel(land0, nv, vh_sel(down0, f0, h.k0));
    h.p0 = vh_sel(land0, np, vh_sel(down0, fp0, h.p0));
}

// registers -> LDS heap arrays in node order (ids permuted through registers)
__device__ __forceinline__ void rh_store(const RegHeap& h, float* hval, int64_t* href, int k, int lane, bool with_refs) {
    const bool n0 = lane >= 1 && lane <= k, n1 = lane + 64 <= k;
    int64_t r0 = 0, r1 = 0;
    if (with_refs) {
        if (n0) r0 = href[h.s0];
        if (n1) r1 = href[h.s1];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (n0) hval[lane - 1] = fkey_inv(h.v0);
    if (n1) hval[lane + 63] = fkey_inv(h.v1);
    if (with_refs) {
        if (n0) href[lane - 1] = r0;
        if (n1) href[lane + 63] = r1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// error_pro::arcos (IVF_pro.cpp:179-184)
__device__ inline float arcos_lut(const float* lut, float x, uint32_t* err) {
    if (!(x <= 1.0 && x >= -1.0)) {
        *err = ERR_ARCOS_DOMAIN;
        return 0.f;
    }
    int index = (int)(x * 500.f / 2.f + 250.f);
    return lut[index];
}

// cosine_theorem (IVF_pro.cpp:41-51): pow(float,int) promotes to double
__device__ inline float cosine_theorem_dev(float a, float b, float c, uint32_t* err) {
    if (!(a <= b)) *err = ERR_COSINE_PRECOND;
    float temp = (float)((double)a * (double)a + (double)c * (double)c - (double)b * (double)b);
    temp = temp / (2 * c);
    return c / 2 - temp;
}

// Trace::search (IVF_pro.cpp:84-107); z[i] = y[i] + std_m * sd[i] is formed once per cached trace with the
// reference's own expression, so every return value is the same fp32 number
__device__ inline float trace_search(const float* x, const float* z, uint32_t n, float k) {
    if (k <= x[0]) return z[0];
    if (k >= x[n - 1]) {
        const float ampli = k / x[n - 1];
        return z[n - 1] * ampli;
    }
    unsigned long long high = n - 1, low = 0, middle = 0;
    while (low <= high) {
        middle = (low + high) / 2;
        if (x[middle] < k) low = middle + 1;
        else high = middle - 1;
    }
    if (x[low] > k) low--;
    return z[low];
}

// kscaling (IVF_pro.cpp:72-82)
__device__ inline float kscaling_dev(float kdis, uint32_t in, const float* gt, uint32_t max_topk) {
    uint32_t index = 0;
    for (; index < max_topk; index++) {
        const float df = fabsf(gt[index] - kdis);
        if ((double)(df / kdis) < 1e-5 || (double)df < 1e-5) break;
    }
    if (index >= max_topk) return -1.f;
    return (float)(index + 1) / (float)(in + 1);
}

// error_pro::set_online (IVF_pro.cpp:196-238): lanes split the entries
__device__ inline void set_online_dev(int metric, uint32_t nlist, const float* cd, const int64_t* ci,
                                      const float* interdis, const float* lut, float* dtb, int lane, uint32_t* err) {
    const uint32_t max_num = nlist / 8 + 20;
    const unsigned long long cur = (unsigned long long)ci[0];
    const float a0 = metric == METRIC_IP ? arcos_lut(lut, cd[0], err) : cd[0];
    for (uint32_t k = lane; k < max_num - 1; k += 64) {
        const unsigned long long dst = (unsigned long long)ci[k + 1];
        const unsigned long long i = cur < dst ? cur : dst, j = cur < dst ? dst : cur;
        const float c = interdis[(2ull * nlist - 1 - i) * i / 2 + j - 1 - i];
        const float b = metric == METRIC_IP ? arcos_lut(lut, cd[k + 1], err) : cd[k + 1];
        dtb[k] = cosine_theorem_dev(a0, b, c, err);
    }
    if (lane == 0) dtb[max_num - 1] = 0.f;
    if (metric == METRIC_IP) {
        // the reference converts all max_num coarse values up front (IVF_pro.cpp:208-211)
        for (uint32_t k = lane; k < max_num; k += 64) (void)arcos_lut(lut, cd[k], err);
    }
}

// one wave per query: disToBoundary rows for a batch of queries (run once, before the first round)
__global__ __launch_bounds__(256) void set_online_kernel(int metric, uint32_t nlist, uint32_t nq, const float* coarse_dis,
                                                         const int64_t* coarse_keys, uint32_t coarse_stride, const float* interdis,
                                                         const float* arcos, float* dtb, uint32_t* error, const uint32_t* qsel,
                                                         const uint32_t* nq_dev) {
    __shared__ float lut[500];
    for (int i = threadIdx.x; i < 500; i += 256) lut[i] = arcos[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t li = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (li >= (nq_dev ? *nq_dev : nq)) return;
    const uint32_t qi = qsel ? qsel[li] : li;  // (a subset of the call's queries: the rankings the tie patch rewrote)
    uint32_t err = 0;
    set_online_dev(metric, nlist, coarse_dis + (size_t)qi * coarse_stride, coarse_keys + (size_t)qi * coarse_stride, interdis, lut,
                   dtb + (size_t)qi * (nlist / 8 + 20), lane, &err);
    err = wave_max_u32(err);
    if (err && lane == 0) atomicMax(error, err);
}

void launch_set_online(int metric, uint32_t nlist, uint32_t nq, const float* coarse_dis, const int64_t* coarse_keys,
                       uint32_t coarse_stride, const float* interdis, const float* arcos, float* dtb, uint32_t* error, hipStream_t s,
                       const uint32_t* qsel, const uint32_t* nq_dev) {
    if (nq) LAUNCH(set_online_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, metric, nlist, nq, coarse_dis, coarse_keys,
                               coarse_stride, interdis, arcos, dtb, error, qsel, nq_dev);
}

// The active queries of a round in two lists: those whose coarse ranking waits for the reference's heap order (slot_of >= 0) and the
// others, which need not wait (run_rounds_device: the first selection in two launches).  counts[0 / 1]: others / waiting (zeroed by
// the caller's memset on the same stream).
__global__ __launch_bounds__(256) void partition_qsel_kernel(const uint32_t* qsel, const uint32_t* nq_dev, uint32_t nq, const int32_t* slot_of,
                                                             uint32_t* q_free, uint32_t* q_wait, uint32_t* counts) {
    const uint32_t n = nq_dev ? *nq_dev : nq;
    const int lane = threadIdx.x & 63;
    for (uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) & ~63u; i0 < n; i0 += gridDim.x * 256) {
        const uint32_t i = i0 + lane;
        const bool in = i < n;
        const uint32_t q = in ? (qsel ? qsel[i] : i) : 0u;
        const bool wait = in && slot_of[q] >= 0;
        const unsigned long long mw = __ballot(wait), mf = __ballot(in && !wait);
        uint32_t bw = 0, bf = 0;
        if (lane == 0) {
            if (mw) bw = atomicAdd(&counts[1], (uint32_t)__builtin_popcountll(mw));
            if (mf) bf = atomicAdd(&counts[0], (uint32_t)__builtin_popcountll(mf));
        }
        bw = (uint32_t)__builtin_amdgcn_readfirstlane((int)bw);
        bf = (uint32_t)__builtin_amdgcn_readfirstlane((int)bf);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (wait) q_wait[bw + __builtin_popcountll(mw & below)] = q;
        else if (in) q_free[bf + __builtin_popcountll(mf & below)] = q;
    }
}
void launch_partition_qsel(const uint32_t* qsel, const uint32_t* nq_dev, uint32_t nq, const int32_t* slot_of, uint32_t* q_free, uint32_t* q_wait,
                           uint32_t* counts, hipStream_t s) {
    if (!nq) return;
    const unsigned grid = std::min<unsigned>((nq + 255) / 256, 1024u);
    LAUNCH(partition_qsel_kernel, dim3(grid), dim3(256), 0, s, qsel, nq_dev, nq, slot_of, q_free, q_wait, counts);
}

// init_state_kernel + set_online_kernel + first_tie_kernel + sbytes_from_f32_kernel for at most four queries (SmallStateArgs)
__global__ __launch_bounds__(256) void small_state_kernel(SmallStateArgs a) {
    __shared__ float lut[500];
    for (int i = threadIdx.x; i < 500; i += 256) lut[i] = a.arcos[i];
    {
        const InitStateArgs& ia = a.init;
        const size_t nk = ia.n * ia.k;
        for (size_t i = threadIdx.x; i < nk; i += 256) {
            ia.heap_val[i] = ia.neutral;
            ia.heap_ref[i] = -1;
            if (ia.fix_val) {
                ia.fix_val[i] = ia.neutral;
                ia.fix_ref[i] = -1;
            }
        }
        for (size_t i = threadIdx.x; i < ia.n; i += 256) {
            ia.thr[i] = ia.neutral;
            ia.stage[i] = 0;
            ia.nscan[i] = 0;
            ia.done[i] = 0;
            ia.pre_val[i] = 0.f;
            ia.stoped[i] = 0;
            if (ia.qstat) ia.qstat[i] = make_uint2(0u, 0u);
            if (ia.log_cnt) {
                ia.log_cnt[i] = 0;
                ia.amb[i] = 0xffffffffu;
                ia.tie_flag[i] = 0;
                ia.log_snap[i] = 0;
                ia.log_snap[ia.n + i] = 0;
                ia.fin_round[i] = 0xffffffffu;
                ia.fix_pos[i] = 0;
            }
        }
        if (threadIdx.x < 4 * STATS_ROWS) ia.stats[threadIdx.x] = 0;
        if (threadIdx.x == 4 * STATS_ROWS) *ia.error = 0;
    }
    __syncthreads();  // (the error word is zeroed before set_online may raise it; the lookup table is in LDS)
    const int lane = threadIdx.x & 63;
    const uint32_t qi = threadIdx.x >> 6;
    if (qi >= a.nq) return;
    uint32_t err = 0;
    set_online_dev(a.metric, a.nlist, a.coarse_dis + (size_t)qi * a.coarse_stride, a.coarse_keys + (size_t)qi * a.coarse_stride, a.interdis, lut,
                   a.dtb + (size_t)qi * (a.nlist / 8 + 20), lane, &err);
    err = wave_max_u32(err);
    if (err && lane == 0) atomicMax(a.init.error, err);
    if (a.ft_sorted_dis) {
        const float* od = a.ft_sorted_dis + (size_t)qi * a.ft_stride;
        uint32_t first = 0xffffffffu;
        for (uint32_t i = (uint32_t)lane; i + 1 < a.ft_nreal; i += 64)
            if (od[i] == od[i + 1] && i < first) first = i;
        for (int off = 32; off; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)first, off);
            first = o < first ? o : first;
        }
        if (lane == 0) a.ft_out[qi] = first;
    }
    if (a.bx) {
        const int stride = (int)mfma_ksteps(a.d) * 32;
        int sq = 0, sum = 0;
        for (int c = lane * 4; c < stride; c += 256) {
            uint32_t word = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int sv = 0;
                if (c + b < a.d) {
                    const int uv = (int)a.bx[(size_t)qi * a.dpad + c + b];
                    sv = uv - 128;
                    sq += sv * sv;
                    sum += uv;
                }
                word |= (uint32_t)(sv & 0xff) << (8 * b);
            }
            *reinterpret_cast<uint32_t*>(a.bout + (size_t)qi * stride + c) = word;
        }
        for (int off = 32; off; off >>= 1) {
            sq += __shfl_xor(sq, off);
            sum += __shfl_xor(sum, off);
        }
        if (lane == 0) a.bcx[qi] = a.metric == METRIC_L2 ? sq : 128 * sum - 16384 * a.d;
    }
}
void launch_small_state(const SmallStateArgs& a, hipStream_t s) { LAUNCH(small_state_kernel, dim3(1), dim3(256), 0, s, a); }

// best-first sort of the heap values into srt by ranking (values only matter)
template <bool IsMax> __device__ inline void rank_sort_best_first(const float* src, float* dst, int k, int lane) {
    for (int i = lane; i < k; i += 64) {
        const float x = src[i];
        int rank = 0;
        for (int j = 0; j < k; j++) {
            const float y = src[j];
            rank += (IsMax ? (y < x) : (y > x)) || (y == x && j < i);
        }
        dst[rank] = x;
    }
}

// srt holds the k heap values best first.  A heap update replaces the worst value (the heap top,
// == srt[k-1]) by `val`: shift the worse ones down by one slot and drop val into the gap.
template <bool IsMax> __device__ inline int sorted_replace_worst(float* srt, int k, float val, int lane) {
    int pos = 0;
    for (int c = (k - 1) / 64; c >= 0; c--) {
        const int idx = c * 64 + lane;
        const bool in = idx < k - 1;
        const float s = in ? srt[idx] : 0.f;
        const bool worse = in && (IsMax ? s > val : s < val);
        pos += __builtin_popcountll(__ballot(in && !worse));
        wave_sync();
        if (worse) srt[idx + 1] = s;
        wave_sync();
    }
    srt[pos] = val;
    wave_sync();
    return pos;  // where val landed in the best-first order
}

// error_pro::sum_angle (IVF_pro.cpp:162-177), n = 15: the 15 terms on 15 lanes, then summed in the
// reference's order (a skipped term adds +0, which leaves the non-negative running sum unchanged)
__device__ inline float sum_angle_par(const float* lut, float kdis, const float* dwin, int lane, uint32_t* err) {
    float t = 0.f;
    if (lane < 15) {
        const float b = dwin[lane];  // the 15 boundary distances of this stage (window of disToBoundary)
        if (!(b >= kdis)) t = arcos_lut(lut, b / kdis, err);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 15; i++) sum += __shfl(t, i);
    return sum;
}

struct TraceLds {
    const float *x, *z;
    uint32_t n;
};

// error_pro::cur_num (IVF_pro.cpp:258-291); Ds(m) = m-th best value (IP: its arcos)
template <bool IsMax>
__device__ inline uint32_t cur_num_lds(const TraceLds& tr, const float* lut, const float* srt, const float* dwin,
                                       uint32_t query_topk, int lane, uint32_t* err) {
    const unsigned long long query_k = query_topk;
    unsigned long long high = query_k - 1, low = 0, middle = 0;
    auto Ds = [&](unsigned long long m) { return IsMax ? srt[m] : arcos_lut(lut, srt[m], err); };
    {
        const float g = trace_search(tr.x, tr.z, tr.n, sum_angle_par(lut, Ds(high), dwin, lane, err));
        if ((double)((float)query_k * g) <= (double)query_k * 1.005) return (uint32_t)query_k;
    }
    while (low <= high) {
        middle = (low + high) / 2;
        if (middle <= 0) return 0;
        const float g = trace_search(tr.x, tr.z, tr.n, sum_angle_par(lut, Ds(middle), dwin, lane, err));
        if ((float)(middle + 1) * g <= (float)query_k) low = middle + 1;
        else high = middle - 1;
    }
    return (uint32_t)(low + 1);
}

// The same function with every probe value of the binary search evaluated at once: term (m, i) of
// sum_angle(Ds(m)) on lane m * 15 + i (three passes for query_topk = 10), then lane m adds its 15 terms in the
// reference's order and runs its own Trace::search; the search itself is replayed on the scalar unit over the
// resulting predicate bits.  Each S(Ds(m)) is formed by the same fp32 operations in the same order as above, and
// an acos-domain error only counts if the reference's search would have visited that m.
constexpr uint32_t CURNUM_PAR_MAXK = 10;  // terms[] holds CURNUM_PAR_MAXK * 15 floats per wave
template <bool IsMax>
__device__ inline uint32_t cur_num_par(const TraceLds& tr, const float* lut, const float* srt, const float* dwin, float* terms,
                                       uint32_t query_topk, int lane, uint32_t* err) {
    const int nterm = (int)query_topk * 15;
    unsigned long long errm = 0;  // bit m: evaluating S(Ds(m)) left the acos domain
    for (int base = 0; base < nterm; base += 64) {
        const int idx = base + lane;
        uint32_t e = 0;
        if (idx < nterm) {
            const int m = idx / 15, i = idx - m * 15;
            uint32_t e0 = 0;
            const float kd = IsMax ? srt[m] : arcos_lut(lut, srt[m], &e0);  // IP: the caller has range-checked every srt[]
            const float b = dwin[i];
            float t = 0.f;
            if (!(b >= kd)) t = arcos_lut(lut, b / kd, &e);
            terms[idx] = t;
        }
        unsigned long long eb = __ballot(e != 0);
        while (eb) {
            const int l = __builtin_ctzll(eb);
            eb &= eb - 1;
            errm |= 1ull << ((base + l) / 15);
        }
    }
    wave_sync();
    float g = 0.f;
    if ((uint32_t)lane < query_topk) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 15; i++) sum += terms[lane * 15 + i];
        g = trace_search(tr.x, tr.z, tr.n, sum);
    }
    const unsigned long long query_k = query_topk;
    const bool first_ok = (double)((float)query_k * g) <= (double)query_k * 1.005;
    const bool step_ok = (float)(lane + 1) * g <= (float)query_k;
    const unsigned long long mfirst = __ballot((uint32_t)lane == query_topk - 1 && first_ok);
    const unsigned long long mstep = __ballot((uint32_t)lane < query_topk && step_ok);
    wave_sync();
    unsigned long long high = query_k - 1, low = 0, middle = 0;
    if ((errm >> high) & 1) {
        *err = ERR_ARCOS_DOMAIN;
        return 0;
    }
    if (mfirst) return (uint32_t)query_k;
    while (low <= high) {
        middle = (low + high) / 2;
        if (middle <= 0) return 0;
        if ((errm >> middle) & 1) {
            *err = ERR_ARCOS_DOMAIN;
            return 0;
        }
        if ((mstep >> middle) & 1) low = middle + 1;
        else high = middle - 1;
    }
    return (uint32_t)(low + 1);
}

// ---------------------------------------------------------------------------------------------
// The k best as a SORTED array in registers (k <= 128): entry i lives in lane i & 63 of register i >> 6, as an order key
// (okey: smaller = better) next to the global position of its vector (list_off[list] + position: what ids[] is indexed
// by).  The reference's heap and this array hold the same multiset of values at every moment -- both admit a candidate iff
// it is strictly better than the worst of the k, both evict a worst one -- so the admissions (hence nheap_updates, the
// thresholds, every input of the stop rule) are the reference's.  Which *id* goes with a value is history-free too unless
// equal values meet: (a) a worst value is evicted while an equal one stays (`amb` remembers the value until nothing that
// bad is left), (b) two equal values are both in the final k (their output order is the heap's).  Such a query is flagged
// and tie_fix_kernel replays the reference's heap over the query's admission log; everybody else's result is the array.
constexpr uint32_t SKEY_SENT = 0xff7fffffu;   // okey of the empty entry (FLT_MAX for L2, -FLT_MAX for IP)
constexpr uint32_t SPOS_NONE = 0xffffffffu;

struct SortedRegs {
    uint32_t k0, k1;  // order keys, ascending
    uint32_t g0, g1;  // global positions
};

// lane i <- lane i - 1 across the whole wave; lane 0 <- carry (wave-uniform)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// Two registers (k > 64) hold the array INTERLEAVED: entry i is lane i >> 1 of register i & 1.  An insertion moves every worse
// entry one place down: an odd entry takes its even neighbour's place-mate of the same lane (no lane crossing at all), an even entry
// takes the odd entry of the lane to its left -- one wave shift per payload register instead of two shifts, two carries between the
// registers (v_readlane + v_mov each) and an early-out branch.  One register (k <= 64): entry i is lane i.
template <bool TWO> __device__ __forceinline__ uint32_t sr_key(const SortedRegs& r, int i) {
    if (!TWO) return rl_u(r.k0, i);
    const uint32_t a = rl_u(r.k0, i >> 1), b = rl_u(r.k1, i >> 1);
    return (i & 1) ? b : a;
}
template <bool TWO> __device__ __forceinline__ int sr_e0(int lane) { return TWO ? 2 * lane : lane; }      // entry held in k0 / g0
template <bool TWO> __device__ __forceinline__ int sr_e1(int lane) { return TWO ? 2 * lane + 1 : 128; }   // ... in k1 / g1

// insert (key, gpos) behind the entries that are <= key; everything worse moves down one place (the old entry k - 1 is
// thereby evicted: it slides into the unused tail).  Returns the position taken.  An entry takes the new one iff its own key
// is worse and its predecessor's is not; the entries behind it take their predecessor's.
template <bool TWO> __device__ __forceinline__ int sr_insert(SortedRegs& r, uint32_t key, uint32_t gpos, int lane) {
    (void)lane;
    const bool w0 = r.k0 > key;
    if (TWO) {
        const bool w1 = r.k1 > key;
        const int at = 128 - __builtin_popcountll(__ballot(w0)) - __builtin_popcountll(__ballot(w1));
        // entry 2l: predecessor = entry 2l - 1 = lane l - 1 of register 1 (entry 0: none -- nothing is better than key 0)
        const uint32_t t0 = wave_shr1(r.k1, 0u), u0 = wave_shr1(r.g1, 0u);
        const bool p0 = t0 > key;
        // entry 2l + 1: predecessor = entry 2l = this lane's k0, and "it moves too" is w0
        const uint32_t n1k = w0 ? r.k0 : key, n1g = w0 ? r.g0 : gpos;
        r.k1 = w1 ? n1k : r.k1;
        r.g1 = w1 ? n1g : r.g1;
        r.k0 = w0 ? (p0 ? t0 : key) : r.k0;
        r.g0 = w0 ? (p0 ? u0 : gpos) : r.g0;
        return at;
    }
    const int at = 64 - __builtin_popcountll(__ballot(w0));
    const uint32_t t0 = wave_shr1(r.k0, 0u), u0 = wave_shr1(r.g0, 0u);  // (lane 0's left neighbour: nothing is better than key 0)
    const bool p0 = t0 > key;
    r.k0 = w0 ? (p0 ? t0 : key) : r.k0;
    r.g0 = w0 ? (p0 ? u0 : gpos) : r.g0;
    return at;
}

// (list << 32 | position) of a global position: the list whose range holds it (empty lists skipped by the search)
__device__ inline int64_t pair_of_gpos(const uint64_t* list_off, uint32_t nlist, uint32_t gpos) {
    uint32_t lo = 0, hi = nlist;  // invariant: list_off[lo] <= gpos < list_off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (list_off[mid] <= gpos) lo = mid;
        else hi = mid;
    }
    return ((int64_t)lo << 32) | (int64_t)(gpos - list_off[lo]);
}

__host__ __device__ inline size_t replay_wave_bytes(int k, uint32_t nlist, bool geo, bool tune, bool train, uint32_t trace_cap) {
    (void)nlist;
    size_t b = (size_t)k * 16;                       // href | hval | srt
    if (geo) b += 16 * 4 + 16 * 4;                   // window of disToBoundary | values inserted during the current probe
    if (tune) b += (size_t)trace_cap * 8;            // cached trace (x | z)
    if (tune) b += CURNUM_PAR_MAXK * 15 * 4 + 8;     // sum_angle terms of cur_num_par
    if (train) b += (size_t)k * 4;                   // ground-truth row
    return (b + 15) & ~(size_t)15;
}

// The reference's heap itself, replayed for every query: what the scanner API (raw heap in / out), trace training, the coarse
// quantiser's short rankings, k > 128 and AUNCEL_AMD_SELECT=heap use (everything else: select_sorted_kernel below).
// RH: the heap lives in registers (k <= 127); otherwise in LDS
// NLD: 64-candidate chunks per trip of the candidate stream (registers for two trips are live)
// KC: compile-time k of the register heap (0: run-time k)
template <bool IsMax, bool RH, int NLD, int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NLD == 16 ? 5 : 3))) void replay_kernel(ReplayArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool tune = a.tuner.enabled != 0, training = a.train.enabled != 0, geo = tune || training;
    const int k = KC ? KC : a.k;
    const uint32_t nlist = a.nlist;
    const uint32_t max_num = nlist / 8 + 20;

    // shared: the acos LUT; per wave: href | hval | srt | dtb | trace cache | gt row
    float* lut = reinterpret_cast<float*>(smem);
    if (geo) {
        const float* g = tune ? a.tuner.arcos : a.train.arcos;
        for (int i = threadIdx.x; i < 500; i += 256) lut[i] = g[i];
        __syncthreads();
    }
    const uint32_t li = blockIdx.x * 4 + wave;   // position in this launch
    if (li >= (a.nq_dev ? *a.nq_dev : a.nq)) return;
    const uint32_t qi = a.qsel ? a.qsel[li] : li;  // query slot (state / output row)
    if (a.done[qi]) return;

    unsigned char* base = smem + (geo ? 2000 : 0) + (size_t)wave * replay_wave_bytes(k, nlist, geo, tune, training, a.trace_cap);
    int64_t* href = reinterpret_cast<int64_t*>(base);
    float* hval = reinterpret_cast<float*>(base + (size_t)k * 8);
    float* srt = hval + k;
    float* dwin = srt + k;                                 // geo only: 16 boundary distances of the current stage
    float* pend = dwin + (geo ? 16 : 0);                   // geo only: values inserted during the current probe
    float* trc = pend + (geo ? 16 : 0);                    // tune only: x | z, trace_cap each
    float* terms = trc + (tune ? 2 * a.trace_cap : 0);     // tune only: cur_num_par scratch
    float* gtrow = terms + (tune ? CURNUM_PAR_MAXK * 15 + 2 : 0);  // training only
    const float* gdtb = geo ? a.dtb + (size_t)qi * max_num : nullptr;  // disToBoundary (set_online_kernel)

    for (int i = lane; i < k; i += 64) {
        hval[i] = a.heap_val[(size_t)qi * k + i];
        href[i] = a.heap_ref[(size_t)qi * k + i];
    }
    wave_sync();

    const unsigned long long id_q = a.id_offset + qi;
    const unsigned long long dbg_t0 = a.dbg ? __builtin_readcyclecounter() : 0;
    unsigned long long dbg_evals = 0, dbg_stream = 0, dbg_rule = 0, dbg_chunks = 0, dbg_probes = 0;
    uint32_t err = 0;
    uint32_t ik0 = a.stage[qi];
    const uint32_t loop_end = a.limit ? a.limit[qi] : a.total_nprobe;
    const uint32_t si = a.seg_by_slot ? qi : li;
    const uint32_t cnt = a.seg_count[si];
    const size_t seg0 = a.seg_begin ? (size_t)a.seg_begin[si] : (size_t)li * a.round_probes;
    unsigned long long nscan = a.nscan[qi];
    float pre_val = a.pre_val ? a.pre_val[qi] : 0.f;
    uint32_t stoped = a.stoped ? a.stoped[qi] : 0u;
    unsigned long long st_nlist = 0, st_nheap = 0, st_ndis = 0;

    // RH: the heap in lanes, walked as vector work (VHeap); a node's payload is the slot of its 64-bit id in href[], so ids never move
    constexpr bool VTWO = KC == 0 || KC > 63;
    auto stored = [](float x) { return IsMax ? fkey(x) : ~fkey(x); };
    auto unstored = [](uint32_t sk) { return fkey_inv(IsMax ? sk : ~sk); };
    VHeap vh{0u, 0u, 0u, 0u};
    const VHeapLane vc = vh_lane(lane);
    unsigned long long chain0 = 0, chain1 = 0;
    if (RH) {
        if (lane >= 1 && lane <= k) vh.k0 = stored(hval[lane - 1]);
        if (VTWO && lane + 64 <= k) vh.k1 = stored(hval[lane + 63]);
        vh.p0 = (uint32_t)(lane - 1);
        vh.p1 = (uint32_t)(lane + 63);
        for (int i = k; i >= 1; i >>= 1) {
            if (i < 64) chain0 |= 1ull << i;
            else chain1 |= 1ull << (i - 64);
        }
    }
    // registers -> LDS heap arrays in node order (rh_store's form)
    auto vh_store = [&](bool with_refs) {
        RegHeap rh{IsMax ? vh.k0 : ~vh.k0, IsMax ? vh.k1 : ~vh.k1, vh.p0, vh.p1};
        rh_store(rh, hval, href, k, lane, with_refs);
    };

    int win_start = -1;
    if (geo) {
        rank_sort_best_first<IsMax>(hval, srt, k, lane);
        if (training) {
            const float* gt = a.train.gt_D + id_q * (unsigned long long)k;
            for (int i = lane; i < k; i += 64) gtrow[i] = gt[i];
        }
        wave_sync();
    }

    uint32_t query_k = 0;
    float true_KD_K = 0.f, racc = 0.f;
    unsigned long long np = 0;
    int cached_ind = -1;
    // cur_num is a pure function of the trace / window of `ind` and of the query_k best heap values: its value is kept
    // until one of them changes (in the later rounds most probes leave the best values alone)
    bool have_pre = false, top_changed = true, srt_changed = true;
    uint32_t kept_pre = 0;
    TraceLds tr{trc, trc + a.trace_cap, 0};
    if (tune) {
        query_k = a.tuner.query_topk;
        if (a.tuner.gt_D) true_KD_K = a.tuner.gt_D[id_q * (unsigned long long)k + query_k - 1];
        racc = a.tuner.require_acc[id_q];
        np = a.tuner.my_nprobe[id_q];
    }
    const unsigned long long np_in = np;

    // The candidate stream (this round's distance rows, in probe order) comes in trips of NLD x 64 values.
    // The loads of trip t + 1 are issued before trip t is examined, across probe boundaries: a wave owns one
    // query, so its own loads in flight are all that hides the HBM round trip.
    constexpr uint32_t TRIP = NLD * 64;
    // probe table of the current window of 64 probes, one probe per lane: list number, candidates, row offset
    uint32_t win0 = 0;
    int m_key = -1;
    uint32_t m_n = 0;
    unsigned long long m_off = 0;
    auto load_window = [&](uint32_t w0) {
        win0 = w0;
        m_key = -1;
        m_n = 0;
        m_off = 0;
        const uint32_t pi = w0 + lane;
        if (pi < cnt) {
            m_key = a.seg_list[seg0 + pi];
            m_off = a.seg_off[seg0 + pi];
            if (m_key >= 0 && (uint32_t)m_key < nlist)
                m_n = a.identity_ids ? nlist : (uint32_t)(a.list_off[m_key + 1] - a.list_off[m_key]);
        }
    };
    load_window(0);
    uint32_t fp = 0, fb = 0;  // fetch cursor: next (probe, offset); it never leaves the consumer's window
    auto fetch = [&](float (&dst)[NLD]) {
        for (;;) {
            if (fp >= cnt || fp >= win0 + 64) return;
            const uint32_t fn = (uint32_t)rl_i((int)m_n, (int)(fp - win0));
            if (fb < fn) {
                const unsigned long long fo = ((unsigned long long)(uint32_t)rl_i((int)(m_off >> 32), (int)(fp - win0)) << 32) |
                                              (uint32_t)rl_i((int)(uint32_t)m_off, (int)(fp - win0));
                const float* fseg = a.dist + fo;
#pragma unroll
                for (int u = 0; u < NLD; u++) {
                    const uint32_t j = fb + u * 64 + lane;
                    dst[u] = j < fn ? __builtin_nontemporal_load(fseg + j) : hneutral<IsMax>();
                }
                fb += TRIP;
                if (fb >= fn) {
                    fp++;
                    fb = 0;
                }
                return;
            }
            fp++;
            fb = 0;
        }
    };
    // Masked rounds (a.mask: one bit per candidate, written by the scan kernel for the values that beat the heap
    // top the query had when the round was planned): a row costs one 8-byte load per 64 candidates, and only the
    // chunks with a bit set are fetched.  Rows start on multiples of 64 floats there.
    const bool masked = a.mask != nullptr;
    float v[NLD], nv[NLD];
#pragma unroll
    for (int u = 0; u < NLD; u++) v[u] = nv[u] = hneutral<IsMax>();
    if (!masked) fetch(v);
    auto row_offset = [&](uint32_t p) {
        return ((unsigned long long)rl_u((uint32_t)(m_off >> 32), (int)(p - win0)) << 32) | rl_u((uint32_t)m_off, (int)(p - win0));
    };
    // mask words of chunks 0..63 and 64..127 of probe pre_p (8192 candidates: all but the very longest lists), requested one
    // probe ahead: the row of a probe then costs no memory round trip of its own
    unsigned long long mw_pre = 0, mw_pre2 = 0;
    uint32_t pre_p = 0xffffffffu;
    auto prefetch_masks = [&](uint32_t p) {
        pre_p = 0xffffffffu;
        if (p >= cnt || p >= win0 + 64) return;
        const uint32_t pn = rl_u(m_n, (int)(p - win0));
        if (pn == 0) return;
        const unsigned long long* mr = a.mask + (row_offset(p) >> 6);
        const uint32_t pch = (pn + 63) >> 6;
        mw_pre = (uint32_t)lane < pch ? mr[lane] : 0ull;
        mw_pre2 = (uint32_t)lane + 64 < pch ? mr[lane + 64] : 0ull;
        pre_p = p;
    };

    bool finished = false;
    uint32_t consumed = 0;
    for (uint32_t p = 0; p < cnt && !finished; p++) {
        const uint32_t ik = ik0 + p;
        consumed = p + 1;
        if (p >= win0 + 64) {  // next window of the probe table; the stream restarts behind it
            load_window(p);
            fp = p;
            fb = 0;
            if (!masked) fetch(v);
        }
        const int key = rl_i(m_key, (int)(p - win0));
        if (key >= 0) {
            if ((uint32_t)key >= nlist) {
                err = ERR_INVALID_KEY;
                finished = true;
                break;
            }
            const uint32_t n = rl_u(m_n, (int)(p - win0));
            if (n > 0) {
                st_nlist++;
                const unsigned long long dbg_s0 = a.dbg ? __builtin_readcyclecounter() : 0;
                const int64_t refbase = REF_TAG | ((int64_t)key << 32);
                uint32_t npend = 0;
                const uint32_t nchunk = (n + 63) >> 6;
                const unsigned long long roff = row_offset(p);
                const float* seg = a.dist + roff;
                const unsigned long long* mrow = masked ? a.mask + (roff >> 6) : nullptr;
                unsigned long long mw = 0, nz = 0, mw2 = 0;
                bool have2 = false;
                uint32_t b0 = 0, w0 = 0;
                if (masked) {
                    if (pre_p == p) {  // the first two windows of this row were requested while the previous row ran
                        mw = mw_pre;
                        mw2 = mw_pre2;
                        have2 = true;
                        nz = __ballot(mw != 0);
                        w0 = 64;
                    }
                    prefetch_masks(p + 1);
                }
                for (;;) {
                    unsigned long long bm = 0;  // masked: lanes of `mw` (chunks w0 - 64 + lane) now held in v[0..)
                    if (masked) {
                        while (nz == 0 && w0 < nchunk) {
                            if (w0 == 64 && have2) mw = mw2;
                            else mw = w0 + lane < nchunk ? mrow[w0 + lane] : 0ull;
                            nz = __ballot(mw != 0);
                            w0 += 64;
                        }
                        if (nz == 0) break;
                        if (a.dbg) dbg_chunks += __builtin_popcountll(nz);
#pragma unroll
                        for (int t = 0; t < NLD; t++) {
                            v[t] = hneutral<IsMax>();
                            if (nz) {
                                const int c = __builtin_ctzll(nz);
                                nz &= nz - 1;
                                bm |= 1ull << c;
                                const unsigned long long bits =
                                    ((unsigned long long)rl_u((uint32_t)(mw >> 32), c) << 32) | rl_u((uint32_t)mw, c);
                                if ((bits >> lane) & 1) v[t] = __builtin_nontemporal_load(seg + (size_t)(w0 - 64 + c) * 64 + lane);
                            }
                        }
                    } else {
                        if (b0 >= n) break;
                        fetch(nv);
                    }
                    float top = RH ? unstored(rl_u(vh.k0, 1)) : hval[0];  // heap top, kept in a register between admissions
                    // chunks (64 candidates) holding at least one value that beats the top as it is now
                    uint32_t umask = 0;
#pragma unroll
                    for (int u = 0; u < NLD; u++) umask |= __ballot(hcmp<IsMax>(top, v[u])) ? (1u << u) : 0u;
                    // one copy of the update code for all chunks (32 inlined copies do not fit the instruction cache)
                    while (umask) {
                        const int u = __builtin_ctz(umask);
                        umask &= umask - 1;
                        float x = v[0];
#pragma unroll
                        for (int t = 1; t < NLD; t++) x = t == u ? v[t] : x;
                        uint32_t cbase = b0 + u * 64;  // position of the chunk's first candidate in its list
                        if (masked) {
                            unsigned long long mm = bm;
                            for (int i = 0; i < u; i++) mm &= mm - 1;
                            cbase = (w0 - 64 + (uint32_t)__builtin_ctzll(mm)) * 64;
                        }
                        unsigned long long m = __ballot(hcmp<IsMax>(top, x));
                        while (m) {
                            const int l = __builtin_ctzll(m);
                            m &= m - 1;
                            const float val = rl_f(x, l);
                            if (hcmp<IsMax>(top, val)) {
                                const int64_t nref = refbase | (int64_t)(cbase + l);
                                if (RH) {
                                    const uint32_t sr = rl_u(vh.p0, 1);  // the evicted root's id slot passes to the new entry
                                    if (lane == 0) href[sr] = nref;
                                    vh_pop<VTWO>(vh, vc, lane, vh_key_at<VTWO>(vh, k), vh_pay_at<VTWO>(vh, k));
                                    vh_push<VTWO>(vh, vc, k, chain0, chain1, stored(val), sr);
                                    top = unstored(rl_u(vh.k0, 1));
                                } else {
                                    heap_pop<IsMax>(k, hval, href);
                                    heap_push<IsMax>(k, hval, href, val, nref);
                                    top = hval[0];
                                }
                                st_nheap++;
                                if (geo) {  // the sorted view is only read at the end of the probe: defer
                                    if (npend < 16) pend[npend] = val;
                                    npend++;
                                }
                            }
                        }
                    }
                    if (!masked) {
#pragma unroll
                        for (int u = 0; u < NLD; u++) v[u] = nv[u];
                        b0 += TRIP;
                    }
                }
                if (a.dbg) dbg_stream += __builtin_readcyclecounter() - dbg_s0;
                if (geo && npend) {
                    wave_sync();
                    srt_changed = true;
                    if (npend <= 16) {
                        for (uint32_t u = 0; u < npend; u++)
                            if (sorted_replace_worst<IsMax>(srt, k, pend[u], lane) < (int)query_k) top_changed = true;
                    } else {
                        if (RH) vh_store(false);
                        rank_sort_best_first<IsMax>(hval, srt, k, lane);
                        wave_sync();
                        top_changed = true;
                    }
                }
                nscan += n;
                st_ndis += n;
            }
        }
        if (a.max_codes && nscan >= a.max_codes) {
            finished = true;
            break;
        }
        if (loop_end && ik + 1 >= loop_end) finished = true;  // end of the probe loop
        wave_sync();
        const unsigned long long dbg_r0 = a.dbg ? __builtin_readcyclecounter() : 0;
        if (tune) {
            // IndexIVF.cpp:551-638.  Once my_nprobe is known nothing the rule computes can change the
            // outcome any more (L2: no throwing path left), so only the stop test remains.
            const uint32_t stage = ik + 1;
            const bool overhead = a.tuner.overhead != 0;  // IndexIVF.cpp:614,634-637
            const bool fired = IsMax && np != 0 && !overhead;
            if (!fired) {
                uint32_t ind = 0;
                const uint32_t tmp_stage = stage >= nlist / 8 ? nlist / 8 - 1 : stage;
                while (tmp_stage > (1u << ind)) ind++;
                if ((int)ind != cached_ind) {
                    const uint32_t o = a.tuner.trace_off[ind], n = a.tuner.trace_off[ind + 1] - o;
                    const float sc = a.tuner.std_m;
                    for (uint32_t i = lane; i < n; i += 64) {
                        trc[i] = a.tuner.trace_x[o + i];
                        trc[a.trace_cap + i] = a.tuner.trace_y[o + i] + sc * a.tuner.trace_std[o + i];
                    }
                    if (lane < 15) dwin[lane] = gdtb[(1u << ind) - 1 + lane];  // sum_angle start = 2^ind - 1
                    tr.n = n;
                    cached_ind = (int)ind;
                    have_pre = false;
                    wave_sync();
                }
                if (!IsMax && srt_changed) {
                    // the reference converts all k heap values (IndexIVF.cpp:562-564): any out-of-domain one throws
                    for (int i = lane; i < k; i += 64) (void)arcos_lut(lut, srt[i], &err);
                    err = wave_err(err);
                    if (err) {
                        finished = true;
                        break;
                    }
                }
                srt_changed = false;
                if (!have_pre || top_changed) {
                    dbg_evals++;
                    kept_pre = query_k <= CURNUM_PAR_MAXK ? cur_num_par<IsMax>(tr, lut, srt, dwin, terms, query_k, lane, &err)
                                                          : cur_num_lds<IsMax>(tr, lut, srt, dwin, query_k, lane, &err);
                    have_pre = true;
                    top_changed = false;
                }
                const uint32_t pre_num = kept_pre;
                float recall = (float)pre_num / (float)query_k;
                const float max_val = IsMax ? fmaxf(-1.f, srt[k - 1]) : fminf(FLT_MAX, srt[k - 1]);
                const unsigned long long stops = (unsigned long long)(racc * 12);
                if (stage > 1) {
                    if (max_val == pre_val) stoped++;
                    else stoped = 0;
                    if (stoped >= stops) recall = 1;
                }
                pre_val = max_val;
                if (!overhead) {
                    if (recall >= racc && np == 0) {
                        np = (unsigned long long)((float)stage * a.tuner.multipler);
                        if (np >= nlist && lane == 0) a.tuner.t_recalls[id_q] = 1.f;
                    }
                    if (stage >= nlist / 8 && np == 0) {
                        np = (unsigned long long)((float)stage * a.tuner.multipler);
                        if (np >= nlist && lane == 0) a.tuner.t_recalls[id_q] = 1.f;
                    }
                }
                err = wave_err(err);
                if (err) finished = true;
            }
            if (overhead) {
                if (stage >= nlist / 8) finished = true;
            } else if (np != 0 && np <= stage) {
                if (a.tuner.profile) {
                    if (RH) vh_store(false);
                    uint32_t hits = 0;
                    for (int i = lane; i < k; i += 64) {
                        const float s = hval[i];
                        if (IsMax ? ((double)s <= (double)true_KD_K * 1.0005) : ((double)s >= (double)true_KD_K * 0.9995)) hits++;
                    }
                    for (int off = 32; off; off >>= 1) hits += __shfl_xor(hits, off);
                    if (lane == 0) a.tuner.t_recalls[id_q] = (float)hits / (float)query_k;
                }
                finished = true;
            }
        }
        if (a.dbg) dbg_rule += __builtin_readcyclecounter() - dbg_r0;
        if (training && !finished) {
            // IndexIVF.cpp:640-673
            const uint32_t stage = ik + 1;
            if (stage > nlist / 8) {
                finished = true;
            } else if ((stage & (stage - 1)) == 0) {
                uint32_t ind = 0;
                while (stage != (1u << ind)) ind++;
                float* out = a.train.raw[ind] + 2ull * (id_q * (unsigned long long)(k / 4));
                if (win_start != (int)(stage - 1)) {
                    wave_sync();
                    if (lane < 15) dwin[lane] = gdtb[stage - 1 + lane];
                    win_start = (int)(stage - 1);
                    wave_sync();
                }
                uint32_t count = 0;
                for (int ij = 0; ij < k; ij++) {
                    const float dv = srt[ij];  // L2 ascending / IP descending, as the reference walks them
                    const float ks = kscaling_dev(dv, (uint32_t)ij, gtrow, (uint32_t)k);
                    if (ks < 0) break;
                    float tval = dv;
                    if (!IsMax) tval = arcos_lut(lut, tval, &err);
                    const float sum_a = sum_angle_par(lut, tval, dwin, lane, &err);
                    if (lane == 0) {
                        out[2 * count] = sum_a;
                        out[2 * count + 1] = ks;
                    }
                    count++;
                    if (count >= (uint32_t)(k / 4)) break;
                }
                err = wave_err(err);
                if (err) finished = true;
            }
        }
    }
    err = wave_err(err);

    if (lane == 0) {
        a.stage[qi] = ik0 + consumed;
        a.nscan[qi] = nscan;
        if (a.pre_val) a.pre_val[qi] = pre_val;
        if (a.stoped) a.stoped[qi] = stoped;
        if (tune && np != np_in) a.tuner.my_nprobe[id_q] = np;
        if (a.qstat) {
            uint2 qs = a.qstat[qi];
            qs.x += (uint32_t)st_nlist;
            qs.y += (uint32_t)st_nheap;
            a.qstat[qi] = qs;
        }
        {
            unsigned long long* st = a.stats + 4 * xcc_id();  // (this XCD's row: STATS_ROWS)
            if (st_nlist) xcd_local_add64(&st[0], st_nlist);
            if (st_ndis) xcd_local_add64(&st[1], st_ndis);
            if (st_nheap) xcd_local_add64(&st[2], st_nheap);
        }
        if (err) atomicMax(a.error, err);
        if (a.dbg) {
            a.dbg[(size_t)li * 8 + 0] = __builtin_readcyclecounter() - dbg_t0;
            a.dbg[(size_t)li * 8 + 1] = st_nheap;
            a.dbg[(size_t)li * 8 + 2] = st_ndis;
            a.dbg[(size_t)li * 8 + 3] = dbg_evals;
            a.dbg[(size_t)li * 8 + 4] = dbg_stream;
            a.dbg[(size_t)li * 8 + 5] = dbg_rule;
            a.dbg[(size_t)li * 8 + 6] = dbg_chunks;
            a.dbg[(size_t)li * 8 + 7] = consumed;
        }
    }

    wave_sync();
    if (a.thr && lane == 0) a.thr[qi] = RH ? unstored(rl_u(vh.k0, 1)) : hval[0];  // next round's scan stores only what beats this
    if (RH) vh_store(true);  // back to the node-ordered LDS layout
    if (finished || a.finalize_all || err) {
        if (a.raw_heap_out) {
            for (int i = lane; i < k; i += 64) {
                int64_t ref = href[i];
                if (ref >= 0 && (ref & REF_TAG)) {
                    ref &= ~REF_TAG;
                    if (!a.store_pairs) ref = a.ids[a.list_off[ref >> 32] + (uint64_t)(ref & 0xffffffffll)];
                    else if (a.pair_list >= 0) ref = ((int64_t)a.pair_list << 32) | (ref & 0xffffffffll);
                }
                a.D[(size_t)qi * k + i] = hval[i];
                a.I[(size_t)qi * k + i] = ref;
            }
        } else {
            // heap_reorder (Heap.h:295-322)
            int ii = 0;
            for (int i = 0; i < k; i++) {
                const float v = hval[0];
                const int64_t id = href[0];
                heap_pop<IsMax>(k - i, hval, href);
                hval[k - ii - 1] = v;
                href[k - ii - 1] = id;
                if (id != -1) ii++;
            }
            wave_sync();
            // valid entries now sit in [k-ii, k): move to the front, pad the rest
            for (int i = lane; i < k; i += 64) {
                float v = hneutral<IsMax>();
                int64_t id = -1;
                if (i < ii) {
                    v = hval[k - ii + i];
                    int64_t ref = href[k - ii + i];
                    if (ref & REF_TAG) {
                        ref &= ~REF_TAG;
                        if (a.identity_ids) ref &= 0xffffffffll;
                        else if (!a.store_pairs) ref = a.ids[a.list_off[ref >> 32] + (uint64_t)(ref & 0xffffffffll)];
                    }
                    id = ref;
                }
                a.D[(size_t)qi * k + i] = v;
                a.I[(size_t)qi * k + i] = id;
            }
        }
        if (lane == 0) a.done[qi] = 1;
    } else {
        for (int i = lane; i < k; i += 64) {
            a.heap_val[(size_t)qi * k + i] = hval[i];
            a.heap_ref[(size_t)qi * k + i] = href[i];
        }
        if (lane == 0 && a.unfinished) (void)xcd_local_add(&a.unfinished[xcc_id()], 1u);
    }
}

// =============================================================================================
// select_sorted_kernel: the selection of the device-planned rounds (sorted register array + admission log, see SortedRegs)
// =============================================================================================
// One wave per query.  A query's rows of a round are one contiguous region (row starts on multiples of 1024 floats, rows in
// probe order), so the candidates are read as one linear stream that runs ahead of the consumer across probe boundaries:
//   dense round   blocks of 256 candidates, one 16-byte load per lane (lane l holds candidates 4l .. 4l+3 of the block), in
//                 groups of four (rows start on multiples of 1024 floats) with the next group in flight; a block costs a min3,
//                 a min and a compare unless something in it beats the worst of the k;
//   masked round  the round's mask words (one per 64 candidates, written by the scan for what beats the round's threshold), 64
//                 words per step with the next step's words in flight; the few marked chunks of a step are fetched together.
// After every probe the stop rule is evaluated exactly as IndexIVF.cpp:551-638 does (tune mode); a probe that admitted
// nothing costs a handful of scalar instructions.
// Admissions in batches of up to 64 (flush, below) instead of one sorted insert each: built and measured in round 5 -- every parity
// suite passes with it, and it is NOT faster: round 0 of the bench workload, per wave, 590 k cycles against 535 k one by one (mean;
// slowest wave 1.23 M against 1.15 M), because an admission costs ~50 issue slots, not the 75 instructions it was counted at, and a
// candidate of a batch costs ~20 + its extraction.  Kept behind the switch (-DAUNCEL_SEL_BATCH=1) with its derivation; off.
#ifndef AUNCEL_SEL_BATCH
#define AUNCEL_SEL_BATCH 0
#endif
constexpr size_t SEL_MERGE_BYTES = AUNCEL_SEL_BATCH ? 2 * 128 * 4 : 0;
__host__ __device__ inline size_t select_wave_bytes(int k, bool tune, bool dense, uint32_t trace_cap) {
    // srt | dwin | trace x,z | cur_num_par terms.  Dense rounds rank the first k candidates of a search at once (keys, value bits,
    // positions | sorted keys, positions: 5 x 128 words): that happens before the first probe's stop rule loads a trace, so in tune
    // mode the two share their room (a workgroup's LDS: 28 -> 18 KB at k = 100, which is what lets a 49 KB workgroup -- the heap
    // order of coarse ties -- start on a CU that holds five of these)
    const size_t fill = 5 * 128 * 4;
    size_t b = SEL_MERGE_BYTES;  // keys | positions of the array a batch of admissions is merged into (select_sorted_kernel: flush)
    if (tune) {
        const size_t trace = (size_t)trace_cap * 8;
        b += (size_t)k * 4 + 16 * 4 + (dense && trace < fill ? fill : trace) + CURNUM_PAR_MAXK * 15 * 4 + 8;
    } else if (dense) {
        b += fill;
    }
    return (b + 15) & ~(size_t)15;
}

typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef AUNCEL_SEL_WAVES
#define AUNCEL_SEL_WAVES 5  // waves per SIMD the dense-round kernel is compiled for: 5 keeps a 5000-query batch resident at once
#endif

// ---------------------------------------------------------------------------- threshold rounds: the rows' marked candidates as lists
// (CompactArgs, ivf_kernels.h).  A wave takes eight consecutive rows: their table entries one per lane, the first 64 mask words of
// all eight requested together -- the words' population counts say how long each row's list is and where it goes -- then row by
// row the marked chunks, four at a time.  Entries are written in position order (chunks ascending, lanes ascending), which is the
// order the selection's own walk admits in.  (Requesting the first chunks of all eight rows together as well took 122 registers:
// half the waves per SIMD, and the launch 0.12 instead of 0.07 ms.)
__global__ __launch_bounds__(256) void compact_rows_kernel(CompactArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t nseg = a.nseg_dev ? *a.nseg_dev : a.nseg;
    const uint32_t stride = gridDim.x * 32u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t s0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 8u; s0 < nseg; s0 += stride) {
        uint32_t m_n = 0;
        unsigned long long m_off = 0;
        if (lane < 8 && s0 + lane < nseg) {
            const int key = a.seg_list[s0 + lane];
            m_off = a.seg_off[s0 + lane];
            if (key >= 0 && (uint32_t)key < a.nlist) m_n = (uint32_t)(a.list_off[key + 1] - a.list_off[key]);
        }
        unsigned long long W[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t n = rl_u(m_n, r);
            const unsigned long long off = ((unsigned long long)rl_u((uint32_t)(m_off >> 32), r) << 32) | rl_u((uint32_t)m_off, r);
            W[r] = (uint32_t)lane < ((n + 63u) >> 6) ? __builtin_nontemporal_load(a.mask + (off >> 6) + lane) : 0ull;
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (s0 + r >= nseg) break;
            const uint32_t n = rl_u(m_n, r);
            const unsigned long long off = ((unsigned long long)rl_u((uint32_t)(m_off >> 32), r) << 32) | rl_u((uint32_t)m_off, r);
            const unsigned long long* mw = a.mask + (off >> 6) + lane;
            const float* row = a.dist + off + lane;
            const uint32_t words = (n + 63u) >> 6;
            // how many: the bits of the row's words
            uint32_t total = wave_sum_u32((uint32_t)__builtin_popcountll(W[r]));
            for (uint32_t b = 64; b < words; b += 64u)
                total += wave_sum_u32(b + lane < words ? (uint32_t)__builtin_popcountll(__builtin_nontemporal_load(mw + b)) : 0u);
            total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
            uint2* ent = a.cl_ent + (size_t)(s0 + r) * CL_CAP;
            uint32_t report = total;
            if (total > CL_CAP) {
                // (a place is asked for only while the cursor is inside the arena and the list could fit at all: a cursor that kept
                // growing with every overflowing row could wrap around and hand out places that other rows own)
                uint32_t start = a.arena_per_xcd;
                if (lane == 0 && total <= a.arena_per_xcd &&
                    __hip_atomic_load(&a.cursor[32u * xcc_id()], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < a.arena_per_xcd)
                    start = xcd_local_add(&a.cursor[32u * xcc_id()], total);
                start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
                if (start >= a.arena_per_xcd || total > a.arena_per_xcd - start) {
                    if (lane == 0) a.cl_cnt[s0 + r] = CL_WALK;
                    continue;
                }
                start += xcc_id() * a.arena_per_xcd;
                if (lane == 0) ent[0] = make_uint2(start, 0u);
                ent = a.arena + start;
            }
            uint32_t count = 0;
            auto put = [&](unsigned long long w, uint32_t b, int c, float x) {
                const unsigned long long bits = ((unsigned long long)rl_u((uint32_t)(w >> 32), c) << 32) | rl_u((uint32_t)w, c);
                const bool mine = (bits >> lane) & 1ull;
                const unsigned long long bal = __ballot(mine);
                if (mine) ent[count + (uint32_t)__builtin_popcountll(bal & lt)] = make_uint2((b + (uint32_t)c) * 64u + (uint32_t)lane, __float_as_uint(x));
                count += (uint32_t)__builtin_popcountll(bal);
            };
            for (uint32_t b = 0; b < words; b += 64u) {
                const unsigned long long w = b == 0 ? W[r] : (b + lane < words ? __builtin_nontemporal_load(mw + b) : 0ull);
                unsigned long long nz = __ballot(w != 0ull);
                while (nz) {
                    int c[4] = {-1, -1, -1, -1};
                    float x[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (nz) {
                            c[u] = __builtin_ctzll(nz);
                            nz &= nz - 1;
                            x[u] = __builtin_nontemporal_load(row + (size_t)(b + c[u]) * 64);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (c[u] >= 0) put(w, b, c[u], x[u]);
                }
            }
            if (lane == 0) a.cl_cnt[s0 + r] = report;
        }
    }
}

void launch_compact_rows(const CompactArgs& a, hipStream_t s) {
    const uint32_t want = a.nseg_dev ? a.nseg_hint : a.nseg;
    if (want == 0) return;
    const uint32_t grid = std::min<uint32_t>((want + 31u) / 32u, 8192u);
    LAUNCH(compact_rows_kernel, dim3(grid), dim3(256), 0, s, a);
}

template <bool IsMax, bool MASKED, bool TUNE, int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MASKED ? 4 : AUNCEL_SEL_WAVES))) void select_sorted_kernel(ReplayArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool TWO = KC == 0 || KC > 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = KC ? KC : a.k;
    const uint32_t nlist = a.nlist;
    const uint32_t max_num = nlist / 8 + 20;
    float* lut = reinterpret_cast<float*>(smem);
    if (TUNE) {
        for (int i = threadIdx.x; i < 500; i += 256) lut[i] = a.tuner.arcos[i];
        __syncthreads();
    }
    const uint32_t li = blockIdx.x * 4 + wave;
    if (li >= (a.nq_dev ? *a.nq_dev : a.nq)) return;
    const uint32_t qi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.qsel ? a.qsel[li] : li));
    if (a.done[qi]) return;

    unsigned char* wbase = smem + (TUNE ? 2000 : 0) + (size_t)wave * select_wave_bytes(k, TUNE, !MASKED, a.trace_cap);
    uint32_t* mk = reinterpret_cast<uint32_t*>(wbase);  // batch merge: keys | positions (SEL_MERGE_BYTES)
    uint32_t* mg = mk + 128;
    float* srt = reinterpret_cast<float*>(wbase + SEL_MERGE_BYTES);  // tune: the k values best first, as the rule reads them
    float* dwin = srt + k;                              // 16 boundary distances of the current stage
    float* trc = dwin + 16;                             // x | z of the cached trace, trace_cap each
    float* terms = trc + (!MASKED && 2 * a.trace_cap < 640u ? 640u : 2 * a.trace_cap);  // cur_num_par scratch (behind the trace's room: select_wave_bytes)
    uint32_t* fill = reinterpret_cast<uint32_t*>(TUNE ? reinterpret_cast<unsigned char*>(trc) : wbase + SEL_MERGE_BYTES);  // dense rounds: 5 x 128 words (tune: in the trace's room)
    const float* gdtb = TUNE ? a.dtb + (size_t)qi * max_num : nullptr;

    // ---- state
    SortedRegs sr{0xffffffffu, 0xffffffffu, SPOS_NONE, SPOS_NONE};
    const int e0 = sr_e0<TWO>(lane), e1 = sr_e1<TWO>(lane);  // the array entries this lane holds (in k0 / g0 and k1 / g1)
    if (e0 < k) {
        sr.k0 = okey<IsMax>(a.heap_val[(size_t)qi * k + e0]);
        sr.g0 = (uint32_t)a.heap_ref[(size_t)qi * k + e0];
    }
    if (TWO && e1 < k) {
        sr.k1 = okey<IsMax>(a.heap_val[(size_t)qi * k + e1]);
        sr.g1 = (uint32_t)a.heap_ref[(size_t)qi * k + e1];
    }
    uint32_t amb = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.amb[qi]);
    uint32_t logn = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.log_cnt[qi]);
    uint2* const qlog = a.log + (size_t)qi * a.log_cap;
    uint32_t log_v = 0, log_g = 0;
    if (!AUNCEL_SEL_BATCH && (uint32_t)lane < (logn & 63u)) {  // the open block of the log comes back into the staging registers
        const uint2 e = qlog[(logn & ~63u) + lane];
        log_v = e.x;
        log_g = e.y;
    }
    auto worst_key = [&]() { return sr_key<TWO>(sr, k - 1); };
    uint32_t topk = worst_key();          // the worst of the k: what a candidate has to beat (as an order key, for the scalar unit,
    float top = okey_inv<IsMax>(topk);    // and as the float the vector compares take)
    auto srt_from_regs = [&]() {
        if (e0 < k) srt[e0] = okey_inv<IsMax>(sr.k0);
        if (TWO && e1 < k) srt[e1] = okey_inv<IsMax>(sr.k1);
    };
    if (TUNE) {
        srt_from_regs();
        wave_sync();
    }

    const unsigned long long id_q = a.id_offset + qi;
    uint32_t err = 0;
    const uint32_t ik0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.stage[qi]);
    const uint32_t loop_end = a.limit ? a.limit[qi] : a.total_nprobe;
    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.seg_count[qi]);
    const size_t seg0 = (size_t)a.seg_begin[qi];
    unsigned long long nscan = a.nscan[qi];
    float pre_val = a.pre_val[qi];
    uint32_t stoped = a.stoped[qi];
    unsigned long long st_nlist = 0, st_ndis = 0;
    uint32_t st_nheap = 0;
    const unsigned long long dbg_t0 = a.dbg ? __builtin_readcyclecounter() : 0;
    unsigned long long dbg_rule = 0;
    uint32_t dbg_evals = 0, dbg_chunks = 0, dbg_walked = 0;

    // ---- stop-rule state (IndexIVF.cpp:551-638)
    uint32_t query_k = 0;
    float true_KD_K = 0.f, racc = 0.f;
    unsigned long long np = 0;
    int cached_ind = -1;
    bool have_pre = false, top_changed = true, srt_changed = true;
    uint32_t kept_pre = 0;
    TraceLds tr{trc, trc + a.trace_cap, 0};
    if (TUNE) {
        query_k = a.tuner.query_topk;
        if (a.tuner.gt_D) true_KD_K = a.tuner.gt_D[id_q * (unsigned long long)k + query_k - 1];
        racc = a.tuner.require_acc[id_q];
        np = a.tuner.my_nprobe[id_q];
    }
    const unsigned long long np_in = np;
    const uint32_t nl8 = nlist / 8;
    const bool overhead = TUNE && a.tuner.overhead != 0;  // IndexIVF.cpp:614,634-637
    const float mult = TUNE ? a.tuner.multipler : 1.f;
    const unsigned long long stops = (unsigned long long)(racc * 12);
    float kept_recall = 0.f;

    // ---- one admission (the candidate beats the worst of the k): IndexIVFFlat.cpp:125-135 on the sorted array
    int ins_min = 128;       // best position taken during the current probe
#if AUNCEL_SEL_BATCH
    // Round 5: admissions in batches.  A candidate that beats the worst of the k *as last known* is set aside (one per lane, in scan
    // order); a batch -- 64 of them, or what a probe's row leaves -- is merged into the array at once (flush).  Candidate j is admitted
    // iff fewer than k values seen before it are <= it: #{array entries <= b_j} + #{earlier candidates of the batch <= b_j} < k (values
    // that left the array are >= its worst and so > b_j; a candidate set aside under a threshold that has moved on since fails the
    // count) -- exactly the reference's "dis < simi[0]" at its turn, so the admissions, their order in the log, nheap_updates and the
    // array after the batch are the one-by-one result; what it costs is ~20 instructions a candidate instead of ~75 an admission.
    uint32_t B_k = 0, B_g = 0;  // lane j: order key and global position of the j-th candidate set aside
    uint32_t bn = 0;
    auto flush = [&]() {
        if (bn == 0) return;
        const bool cand = (uint32_t)lane < bn;
        const uint32_t myk = cand ? B_k : 0xffffffffu;
        uint32_t pa = 0, w = 0, later = 0, sh0 = 0, sh1 = 0;
        for (uint32_t i = 0; i < bn; i++) {
            const uint32_t ki = rl_u(B_k, (int)i);
            uint32_t p = (uint32_t)__builtin_popcountll(__ballot(sr.k0 <= ki));
            if (TWO) p += (uint32_t)__builtin_popcountll(__ballot(sr.k1 <= ki));
            wl_u(pa, (uint32_t)__builtin_amdgcn_readfirstlane((int)p), __builtin_amdgcn_readfirstlane((int)i));
            w += (ki <= myk && (uint32_t)lane > i) ? 1u : 0u;
            later += (ki < myk && (uint32_t)lane < i) ? 1u : 0u;
            sh0 += ki < sr.k0 ? 1u : 0u;
            if (TWO) sh1 += ki < sr.k1 ? 1u : 0u;
        }
        const bool adm = cand && pa + w < (uint32_t)k;
        const unsigned long long am = __ballot(adm);
        const uint32_t nadm = (uint32_t)__builtin_popcountll(am);
        if (nadm) {
            const uint32_t rank = pa + w + later;  // place of an admitted candidate among (array + admitted), by (value, arrival)
            const uint32_t r0 = (uint32_t)e0 + sh0, r1 = (uint32_t)e1 + sh1;  // ... of the array's own entries
            wave_sync();
            if (e0 < k && r0 < (uint32_t)k) {
                mk[r0] = sr.k0;
                mg[r0] = sr.g0;
            }
            if (TWO && e1 < k && r1 < (uint32_t)k) {
                mk[r1] = sr.k1;
                mg[r1] = sr.g1;
            }
            if (adm && rank < (uint32_t)k) {
                mk[rank] = B_k;
                mg[rank] = B_g;
            }
            // the admission log: the admitted candidates in scan order
            if (logn + nadm > a.log_cap) {
                err = ERR_LOG_OVERFLOW;
            } else if (adm) {
                const uint32_t at = logn + (uint32_t)__builtin_popcountll(am & ((1ull << lane) - 1ull));
                qlog[at] = make_uint2(__float_as_uint(okey_inv<IsMax>(B_k)), B_g);
            }
            logn += logn + nadm > a.log_cap ? 0u : nadm;
            st_nheap += nadm;
            const unsigned long long top_in = __ballot(adm && rank < query_k);
            if (top_in) ins_min = 0;  // (an admitted value among the first query_k: the rule's inputs changed)
            wave_sync();
            const bool l0 = e0 < k, l1 = TWO && e1 < k;
            // what left: an array entry or an admitted candidate whose place is k or beyond
            const uint32_t wk = mk[k - 1];
            const bool out0 = l0 && r0 >= (uint32_t)k && sr.k0 == wk, out1 = l1 && r1 >= (uint32_t)k && sr.k1 == wk;
            const bool outc = adm && rank >= (uint32_t)k && B_k == wk;
            if (l0) {
                sr.k0 = mk[e0];
                sr.g0 = mg[e0];
            }
            if (l1) {
                sr.k1 = mk[e1];
                sr.g1 = mg[e1];
            }
            // a value of which a copy left while this one stays: the new worst equals something that left (the new values are
            // strictly better than the old worst, so only then can equal worst values have been split)
            if (wk != SKEY_SENT && __ballot(out0 || out1 || outc) != 0) amb = wk;
            topk = wk;
            top = okey_inv<IsMax>(wk);
        }
        bn = 0;
    };
    auto admit = [&](float val, uint32_t ckey, uint32_t gp) {
        (void)val;
        gp = (uint32_t)__builtin_amdgcn_readfirstlane((int)gp);
        wl2_u(B_k, (uint32_t)__builtin_amdgcn_readfirstlane((int)ckey), B_g, gp, __builtin_amdgcn_readfirstlane((int)bn));
        bn++;
        if (bn == 64u) flush();
    };
#else
    auto flush = [&]() {};
    auto admit = [&](float val, uint32_t ckey, uint32_t gp) {
        const uint32_t evicted = topk;
        gp = (uint32_t)__builtin_amdgcn_readfirstlane((int)gp);
        const int at = sr_insert<TWO>(sr, ckey, gp, lane);
        ins_min = at < ins_min ? at : ins_min;
        const uint32_t ln = (uint32_t)__builtin_amdgcn_readfirstlane((int)logn);
        if (ln >= a.log_cap) {
            err = ERR_LOG_OVERFLOW;
        } else {
            const bool me = (uint32_t)lane == (ln & 63u);  // (a compare and two selects: a v_writelane pair through M0 is six instructions)
            log_v = me ? __float_as_uint(val) : log_v;
            log_g = me ? gp : log_g;
            if ((ln & 63u) == 63u) qlog[(ln & ~63u) + lane] = make_uint2(log_v, log_g);
            logn = ln + 1;
        }
        st_nheap++;
        // the entry that left was one of several equal worst values iff the new worst equals it (the new value is strictly better)
        const uint32_t wk = worst_key();
        if (wk == evicted && wk != SKEY_SENT) amb = wk;
        topk = wk;
        top = okey_inv<IsMax>(wk);
    };
#endif

    // ---- probe table: a window of 64 probes, one per lane (list number, list length)
    uint32_t win0 = 0;
    int m_key = -1;
    uint32_t m_n = 0, m_base = 0;  // list length, global position of the list's first vector
    // masked rounds with compact lists (CompactArgs): the window's rows' counts, one per lane, and their lists -- register j, lane l:
    // entry l & 7 of row 8 j + (l >> 3)
    const bool lists = MASKED && a.cl_cnt != nullptr;
    uint32_t m_cc = 0;
    uint2 E[8];
    auto load_window = [&](uint32_t w0) {
        win0 = w0;
        m_key = -1;
        m_n = 0;
        m_base = 0;
        const uint32_t pi = w0 + lane;
        if (lists) {
            m_cc = pi < cnt ? a.cl_cnt[seg0 + pi] : 0u;
            const uint2* ep = a.cl_ent + (seg0 + w0) * CL_CAP + lane;
#pragma unroll
            for (int j = 0; j < 8; j++) E[j] = w0 + 8u * j + ((uint32_t)lane >> 3) < cnt ? ep[64 * j] : make_uint2(0u, 0u);
        }
        if (pi < cnt) {
            m_key = a.seg_list[seg0 + pi];
            if (m_key >= 0 && (uint32_t)m_key < nlist) {
                const uint64_t o0 = a.list_off[m_key], o1 = a.list_off[m_key + 1];
                m_n = (uint32_t)(o1 - o0);
                m_base = a.identity_ids ? 0u : (uint32_t)o0;
            }
        }
    };
    load_window(0);
    // the region of this query's rows: contiguous, every row padded to a multiple of 1024 floats
    unsigned long long region_off = cnt ? a.seg_off[seg0] : 0ull;
    region_off = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(region_off >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)region_off);

    // ---- the candidate stream
    // dense: R0..R3 hold blocks bpos .. bpos + 3 (in ring order starting at `slot`)
    // A block of 256 candidates comes in with four 4-byte loads a lane -- component c of lane l is candidate 64 c + l -- so that the
    // ballot of a component lists its candidates in scan order and a candidate is reached with one ctz and one v_readlane (with one
    // 16-byte load a lane, candidate 4 l + c, the four ballots had to be interleaved lane by lane: ~30 scalar instructions a hit).
    const float* regionf = a.dist + region_off + lane;  // block b, component c: regionf[b * 256 + c * 64]
    auto load_block = [&](const float* p) {
        f4 r;
        r.x = __builtin_nontemporal_load(p);
        r.y = __builtin_nontemporal_load(p + 64);
        r.z = __builtin_nontemporal_load(p + 128);
        r.w = __builtin_nontemporal_load(p + 192);
        return r;
    };
    f4 R0, R1, R2, R3;
    uint32_t gpos = 0;  // group of four blocks held in R0..R3
    // masked: W = the mask words of step `step` (word step * 64 + lane of the region), Wn = the next step's
    const unsigned long long* rmask = MASKED ? a.mask + (region_off >> 6) + lane : nullptr;
    const float* region1 = a.dist + region_off + lane;  // chunk c (64 candidates): region1[c * 64]
    unsigned long long W = 0, Wn = 0;
    uint32_t wstep = 0;           // words [wstep * 64, wstep * 64 + 64) are in W
    if (!MASKED) {
        R0 = load_block(regionf);
        R1 = load_block(regionf + 256);
        R2 = load_block(regionf + 512);
        R3 = load_block(regionf + 768);
    } else if (!lists) {
        W = __builtin_nontemporal_load(rmask);
        Wn = __builtin_nontemporal_load(rmask + 64);
    }
    uint32_t wpos = 0;  // masked: first word of the current row (region word index)

    // a block of 256 candidates (component c of lane l: candidate 64 c + l; the block's first is number c0 of its list) in which
    // something may beat the worst of the k, or that holds the padding behind the row's end
    auto dense_block = [&](const f4 x, uint32_t c0, uint32_t n, uint32_t lbase, uint32_t taken) {
        const uint32_t valid = n > c0 ? n - c0 : 0u;  // candidates of the row in this block (>= 256: all)
        const uint32_t skip = taken > c0 ? taken - c0 : 0u;  // ... of which the first `skip` are in the array already (fill_first_k)
        const uint32_t p0 = (uint32_t)lane, p1 = 64u + lane, p2 = 128u + lane, p3 = 192u + lane;
        const unsigned long long h0 = __ballot(hcmp<IsMax>(top, x.x) && p0 < valid && p0 >= skip);
        const unsigned long long h1 = __ballot(hcmp<IsMax>(top, x.y) && p1 < valid && p1 >= skip);
        const unsigned long long h2 = __ballot(hcmp<IsMax>(top, x.z) && p2 < valid && p2 >= skip);
        const unsigned long long h3 = __ballot(hcmp<IsMax>(top, x.w) && p3 < valid && p3 >= skip);
        auto walk = [&](unsigned long long h, const float xc, uint32_t g0) {  // one component's hits, in scan order
            while (h) {
                const int l = __builtin_ctzll(h);
                h &= h - 1;
                const float val = rl_f(xc, l);
                const uint32_t ck = okey<IsMax>(val);  // (it beat an earlier worst in the vector compare: not a NaN)
                if (ck < topk) admit(val, ck, g0 + (uint32_t)l);
            }
        };
        const uint32_t g0 = lbase + c0;
        walk(h0, x.x, g0);
        walk(h1, x.y, g0 + 64u);
        walk(h2, x.z, g0 + 128u);
        walk(h3, x.w, g0 + 192u);
    };

    // The first min(k, n) candidates of a search all enter (the array is empty): ranked against each other at once instead of
    // inserted one by one.  x: the row's first block; returns how many were taken.  (Equal values keep arrival order, as
    // sr_insert would leave them; the entries that go are empty ones: nothing for `amb` to note.)
    auto fill_first_k = [&](const f4 x, uint32_t n, uint32_t lbase) -> uint32_t {
        const uint32_t m = n < (uint32_t)k ? n : (uint32_t)k;
        uint32_t *fk = fill, *fv = fill + 128, *fg = fill + 256, *sk = fill + 384, *sg = fill + 512;
        const float xs[4] = {x.x, x.y, x.z, x.w};
        // A candidate that is not strictly better than the empty heap's (+/-)FLT_MAX -- +inf, NaN, FLT_MAX itself -- does not
        // enter the reference's heap either (IndexIVFFlat.cpp:129: C::cmp(simi[0], dis)): with one among the first m nothing is
        // taken here and the block goes through the one-by-one admission like any other.
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 4; c++) bad = bad || (64u * c + lane < m && okey<IsMax>(xs[c]) >= SKEY_SENT);
        if (__ballot(bad) != 0) return 0u;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t j = 64u * c + lane;
            if (j < m) {
                fk[j] = okey<IsMax>(xs[c]);
                fv[j] = __float_as_uint(xs[c]);
                fg[j] = lbase + j;
                qlog[j] = make_uint2(__float_as_uint(xs[c]), lbase + j);
            }
        }
        wave_sync();
        const uint32_t j0 = lane, j1 = lane + 64;
        const uint32_t k0v = j0 < m ? fk[j0] : 0u, k1v = j1 < m ? fk[j1] : 0u;
        uint32_t r0 = 0, r1 = 0;
        for (uint32_t i = 0; i < m; i++) {
            const uint32_t ki = fk[i];
            r0 += (ki < k0v || (ki == k0v && i < j0)) ? 1u : 0u;
            r1 += (ki < k1v || (ki == k1v && i < j1)) ? 1u : 0u;
        }
        if (j0 < m) {
            sk[r0] = k0v;
            sg[r0] = fg[j0];
        }
        if (j1 < m) {
            sk[r1] = k1v;
            sg[r1] = fg[j1];
        }
        wave_sync();
        if ((uint32_t)e0 < m) {
            sr.k0 = sk[e0];
            sr.g0 = sg[e0];
        }
        if (TWO && (uint32_t)e1 < m) {
            sr.k1 = sk[e1];
            sr.g1 = sg[e1];
        }
        // the admission log holds them in arrival order; its open block goes to the staging registers
        logn = m;
        if ((uint32_t)lane < (m & 63u)) {
            log_v = fv[(m & ~63u) + lane];
            log_g = fg[(m & ~63u) + lane];
        }
        st_nheap += m;
        ins_min = 0;
        topk = worst_key();
        top = okey_inv<IsMax>(topk);
        wave_sync();
        return m;
    };
    bool fresh = !MASKED && logn == 0 && sr_key<TWO>(sr, 0) == SKEY_SENT;  // nothing admitted yet in this search

    bool finished = false;
    uint32_t consumed = 0;
    for (uint32_t p = 0; p < cnt && !finished; p++) {
        const uint32_t ik = ik0 + p;
        consumed = p + 1;
        if (p >= win0 + 64) load_window(p);
        const int key = rl_i(m_key, (int)(p - win0));
        ins_min = 128;
        const uint32_t nheap0 = st_nheap;
        if (key >= 0) {
            if ((uint32_t)key >= nlist) {
                err = ERR_INVALID_KEY;
                finished = true;
                break;
            }
            const uint32_t n = rl_u(m_n, (int)(p - win0));
            if (n > 0) {
                st_nlist++;
                const uint32_t lbase = rl_u(m_base, (int)(p - win0));
                if (!MASKED) {
                    // The row is a whole number of groups of four blocks (rows start on multiples of 1024 floats): R0..R3 hold the
                    // current group, each register is refilled with its block of the next group (of the region: whatever row that is
                    // in) as soon as it has been looked at.
                    const uint32_t ngrp = (n + 1023u) >> 10;
                    uint32_t taken = 0;
                    if (fresh) {
                        taken = fill_first_k(R0, n, lbase);
                        fresh = false;
                    }
                    for (uint32_t g = 0; g < ngrp; g++) {
                        const float* nxt = regionf + (size_t)(gpos + 1) * 1024;
#define SEL_BLOCK(R, U)                                                                                                       \
    {                                                                                                                         \
        const f4 x = R;                                                                                                       \
        R = load_block(nxt + (U) * 256);                                                                                      \
        const float best = IsMax ? fminf(fminf(fminf(x.x, x.y), x.z), x.w) : fmaxf(fmaxf(fmaxf(x.x, x.y), x.z), x.w);         \
        if (__ballot(hcmp<IsMax>(top, best)) != 0) dense_block(x, g * 1024u + (U) * 256u, n, lbase, taken);                   \
    }
                        SEL_BLOCK(R0, 0)
                        SEL_BLOCK(R1, 1)
                        SEL_BLOCK(R2, 2)
                        SEL_BLOCK(R3, 3)
#undef SEL_BLOCK
                        gpos++;
                    }
                } else {
                    // words [wpos, wend) of the region are this row's chunks (the scan leaves no bit behind the row's end); the
                    // words of the padding up to the next multiple of 256 candidates are never written and never read
                    const uint32_t row_w0 = wpos, wend = wpos + ((n + 63u) >> 6);
                    if (lists) {
                        const int r = (int)(p - win0);
                        const uint32_t cc = rl_u(m_cc, r);
                        if (cc <= CL_CAP) {
                            // the row's marked candidates are in lanes 8 (r & 7) ... of register r >> 3, in position order
                            if (cc) {
                                uint2 e = E[0];
#pragma unroll
                                for (int j = 1; j < 8; j++)
                                    if ((r >> 3) == j) e = E[j];
                                const int l0 = (r & 7) * 8;
                                for (uint32_t t = 0; t < cc; t++) {
                                    const float val = __uint_as_float(rl_u(e.y, l0 + (int)t));
                                    const uint32_t ck = okey<IsMax>(val);
                                    if (hcmp<IsMax>(top, val) && ck < topk) admit(val, ck, lbase + rl_u(e.x, l0 + (int)t));
                                }
                            }
                            wpos = wend;
                        } else if (cc != CL_WALK) {
                            // a long list: in the arena, from the entry the row's first slot names
                            uint2 e = E[0];
#pragma unroll
                            for (int j = 1; j < 8; j++)
                                if ((r >> 3) == j) e = E[j];
                            const uint2* al = a.cl_arena + rl_u(e.x, (r & 7) * 8) + lane;
                            for (uint32_t t0 = 0; t0 < cc; t0 += 64u) {
                                const bool in = t0 + (uint32_t)lane < cc;
                                const uint2 o = in ? al[t0] : make_uint2(0u, 0u);
                                unsigned long long m = __ballot(in && hcmp<IsMax>(top, __uint_as_float(o.y)));
                                while (m) {
                                    const int l = __builtin_ctzll(m);
                                    m &= m - 1;
                                    const float val = __uint_as_float(rl_u(o.y, l));
                                    const uint32_t ck = okey<IsMax>(val);
                                    if (ck < topk) admit(val, ck, lbase + rl_u(o.x, l));
                                }
                            }
                            wpos = wend;
                        } else {
                            // the arena was full: this row through the mask stream, which starts here
                            if (a.dbg) dbg_walked++;
                            wstep = wpos >> 6;
                            W = __builtin_nontemporal_load(rmask + (size_t)wstep * 64);
                            Wn = __builtin_nontemporal_load(rmask + (size_t)(wstep + 1) * 64);
                        }
                    }
                    while (wpos < wend) {
                        const uint32_t s0 = wstep * 64u;              // first word of the step in W
                        if (wpos >= s0 + 64u) {                       // next step
                            W = Wn;
                            wstep++;
                            Wn = __builtin_nontemporal_load(rmask + (size_t)(wstep + 1) * 64);
                            continue;
                        }
                        // this row's words inside the step: lanes [wpos - s0, min(wend - s0, 64))
                        const uint32_t lo = wpos - s0, hi = wend - s0 < 64u ? wend - s0 : 64u;
                        unsigned long long nz = __ballot(W != 0ull);
                        nz &= ~0ull << lo;
                        if (hi < 64u) nz &= (1ull << hi) - 1ull;
                        wpos = s0 + hi;
                        // the marked chunks, four at a time: their values are requested together
                        while (nz) {
                            int c0 = -1, c1 = -1, c2 = -1, c3 = -1;
                            float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
                            c0 = __builtin_ctzll(nz);
                            nz &= nz - 1;
                            x0 = __builtin_nontemporal_load(region1 + (size_t)(s0 + c0) * 64);
                            if (nz) {
                                c1 = __builtin_ctzll(nz);
                                nz &= nz - 1;
                                x1 = __builtin_nontemporal_load(region1 + (size_t)(s0 + c1) * 64);
                            }
                            if (nz) {
                                c2 = __builtin_ctzll(nz);
                                nz &= nz - 1;
                                x2 = __builtin_nontemporal_load(region1 + (size_t)(s0 + c2) * 64);
                            }
                            if (nz) {
                                c3 = __builtin_ctzll(nz);
                                nz &= nz - 1;
                                x3 = __builtin_nontemporal_load(region1 + (size_t)(s0 + c3) * 64);
                            }
                            auto chunk = [&](int c, float x) {
                                if (c < 0) return;
                                if (a.dbg) dbg_chunks++;
                                const unsigned long long bits = ((unsigned long long)rl_u((uint32_t)(W >> 32), c) << 32) | rl_u((uint32_t)W, c);
                                const uint32_t cb = (s0 + (uint32_t)c - row_w0) * 64u;  // position of the chunk's first candidate in its list
                                unsigned long long m = __ballot(((bits >> lane) & 1) && hcmp<IsMax>(top, x));
                                while (m) {
                                    const int l = __builtin_ctzll(m);
                                    m &= m - 1;
                                    const float val = rl_f(x, l);
                                    const uint32_t ck = okey<IsMax>(val);
                                    if (ck < topk) admit(val, ck, lbase + cb + (uint32_t)l);
                                }
                            };
                            chunk(c0, x0);
                            chunk(c1, x1);
                            chunk(c2, x2);
                            chunk(c3, x3);
                        }
                    }
                    wpos = row_w0 + (((n + 1023u) >> 10) << 4);
                }
                nscan += n;
                st_ndis += n;
            }
        }
        flush();  // (the probe's last candidates: the rule below reads the array as the probe leaves it)
        if (TUNE && st_nheap != nheap0) {
            wave_sync();
            srt_from_regs();
            wave_sync();
            srt_changed = true;
            if (ins_min < (int)query_k) top_changed = true;
        }
        if (a.max_codes && nscan >= a.max_codes) {
            finished = true;
            break;
        }
        if (loop_end && ik + 1 >= loop_end) finished = true;  // end of the probe loop
        if (TUNE) {
            const unsigned long long dbg_r0 = a.dbg ? __builtin_readcyclecounter() : 0;
            // IndexIVF.cpp:551-638.  Once my_nprobe is known nothing the rule computes can change the outcome any more (L2: no
            // throwing path left), so only the stop test remains.
            const uint32_t stage = ik + 1;
            const bool fired = IsMax && np != 0 && !overhead;
            if (!fired) {
                const uint32_t tmp_stage = stage >= nl8 ? nl8 - 1 : stage;
                const uint32_t ind = tmp_stage <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(tmp_stage - 1u);  // smallest ind with tmp_stage <= 2^ind
                if ((int)ind != cached_ind) {
                    const uint32_t o = a.tuner.trace_off[ind], tn = a.tuner.trace_off[ind + 1] - o;
                    const float sc = a.tuner.std_m;
                    wave_sync();
                    for (uint32_t i = lane; i < tn; i += 64) {
                        trc[i] = a.tuner.trace_x[o + i];
                        trc[a.trace_cap + i] = a.tuner.trace_y[o + i] + sc * a.tuner.trace_std[o + i];
                    }
                    if (lane < 15) dwin[lane] = gdtb[(1u << ind) - 1 + lane];  // sum_angle start = 2^ind - 1
                    tr.n = tn;
                    cached_ind = (int)ind;
                    have_pre = false;
                    wave_sync();
                }
                if (!IsMax && srt_changed) {
                    // the reference converts all k heap values (IndexIVF.cpp:562-564): any out-of-domain one throws
                    for (int i = lane; i < k; i += 64) (void)arcos_lut(lut, srt[i], &err);
                    err = wave_err(err);
                    if (err) {
                        finished = true;
                        break;
                    }
                }
                srt_changed = false;
                if (!have_pre || top_changed) {
                    dbg_evals++;
                    kept_pre = query_k <= CURNUM_PAR_MAXK ? cur_num_par<IsMax>(tr, lut, srt, dwin, terms, query_k, lane, &err)
                                                          : cur_num_lds<IsMax>(tr, lut, srt, dwin, query_k, lane, &err);
                    have_pre = true;
                    top_changed = false;
                    err = wave_err(err);
                    if (err) finished = true;
                    kept_recall = (float)kept_pre / (float)query_k;
                }
                float recall = kept_recall;
                const float max_val = IsMax ? fmaxf(-1.f, top) : fminf(FLT_MAX, top);  // (top == srt[k - 1])
                if (stage > 1) {
                    if (max_val == pre_val) stoped++;
                    else stoped = 0;
                    if (stoped >= stops) recall = 1;
                }
                pre_val = max_val;
                if (!overhead) {
                    if (recall >= racc && np == 0) {
                        np = (unsigned long long)((float)stage * mult);
                        if (np >= nlist && lane == 0) a.tuner.t_recalls[id_q] = 1.f;
                    }
                    if (stage >= nl8 && np == 0) {
                        np = (unsigned long long)((float)stage * mult);
                        if (np >= nlist && lane == 0) a.tuner.t_recalls[id_q] = 1.f;
                    }
                }
            }
            if (overhead) {
                if (stage >= nl8) finished = true;
            } else if (np != 0 && np <= stage) {
                if (a.tuner.profile) {
                    uint32_t hits = 0;
                    for (int i = lane; i < k; i += 64) {
                        const float s = srt[i];
                        if (IsMax ? ((double)s <= (double)true_KD_K * 1.0005) : ((double)s >= (double)true_KD_K * 0.9995)) hits++;
                    }
                    for (int off = 32; off; off >>= 1) hits += __shfl_xor(hits, off);
                    if (lane == 0) a.tuner.t_recalls[id_q] = (float)hits / (float)query_k;
                }
                finished = true;
            }
            if (a.dbg) dbg_rule += __builtin_readcyclecounter() - dbg_r0;
        }
    }
    err = wave_err(err);

    if (lane == 0) {
        a.stage[qi] = ik0 + consumed;
        a.nscan[qi] = nscan;
        a.pre_val[qi] = pre_val;
        a.stoped[qi] = stoped;
        if (TUNE && np != np_in) a.tuner.my_nprobe[id_q] = np;
        if (a.qstat) {
            uint2 qs = a.qstat[qi];
            qs.x += (uint32_t)st_nlist;
            qs.y += (uint32_t)st_nheap;
            a.qstat[qi] = qs;
        }
        {
            unsigned long long* st = a.stats + 4 * xcc_id();  // (this XCD's row: STATS_ROWS)
            if (st_nlist) xcd_local_add64(&st[0], st_nlist);
            if (st_ndis) xcd_local_add64(&st[1], st_ndis);
            if (st_nheap) xcd_local_add64(&st[2], (unsigned long long)st_nheap);
        }
        if (err) atomicMax(a.error, err);
        if (a.dbg) {
            a.dbg[(size_t)li * 8 + 0] = __builtin_readcyclecounter() - dbg_t0;
            a.dbg[(size_t)li * 8 + 1] = st_nheap;
            a.dbg[(size_t)li * 8 + 2] = st_ndis;
            a.dbg[(size_t)li * 8 + 3] = dbg_evals;
            a.dbg[(size_t)li * 8 + 4] = dbg_walked;  // (rows whose candidates did not fit a list)
            a.dbg[(size_t)li * 8 + 5] = dbg_rule;
            a.dbg[(size_t)li * 8 + 6] = dbg_chunks;
            a.dbg[(size_t)li * 8 + 7] = consumed;
        }
        if (a.thr) a.thr[qi] = top;  // next round's scan stores only what beats this
        a.log_cnt[qi] = logn;
        a.log_snap[(size_t)(a.round & 1u) * a.nq_total + qi] = logn;
        if (finished || a.finalize_all || err) a.fin_round[qi] = a.round;
    }
    // the open block of the admission log
    if (!AUNCEL_SEL_BATCH && (uint32_t)lane < (logn & 63u)) qlog[(logn & ~63u) + lane] = make_uint2(log_v, log_g);
    const bool in0 = e0 < k, in1 = TWO && e1 < k;
    if (finished || a.finalize_all || err) {
        // equal values among the k (their order is the heap's), or a value of which a copy was evicted while this one stayed
        // (an entry's predecessor: one register -- the lane to the left; two -- see sr_insert)
        const uint32_t p0 = wave_shr1(TWO ? sr.k1 : sr.k0, 0xfffffffeu), p1 = sr.k0;
        const bool dup = (in0 && lane >= 1 && sr.k0 == p0 && sr.k0 != SKEY_SENT) || (in1 && sr.k1 == p1 && sr.k1 != SKEY_SENT);
        const bool tainted = __ballot(dup) != 0 || worst_key() == amb;
        if (tainted && !err) {
            if (lane == 0) {
                a.tie_flag[qi] = 1;  // tie_fix_kernel writes this query's (D, I)
                xcd_local_add64(&a.stats[4 * xcc_id() + 3], 1ull);  // (how common that is decides when tie_fix_kernel runs: run_rounds_device)
            }
        } else {
            auto put = [&](int i, uint32_t key, uint32_t g) {
                const bool empty = key == SKEY_SENT && g == SPOS_NONE;
                int64_t id = -1;
                if (!empty) id = a.identity_ids ? (int64_t)g : a.store_pairs ? pair_of_gpos(a.list_off, nlist, g) : a.ids[g];
                a.D[(size_t)qi * k + i] = empty ? hneutral<IsMax>() : okey_inv<IsMax>(key);
                a.I[(size_t)qi * k + i] = id;
            };
            if (in0) put(e0, sr.k0, sr.g0);
            if (in1) put(e1, sr.k1, sr.g1);
        }
        if (lane == 0) a.done[qi] = 1;
    } else {
        if (in0) {
            a.heap_val[(size_t)qi * k + e0] = okey_inv<IsMax>(sr.k0);
            a.heap_ref[(size_t)qi * k + e0] = sr.g0 == SPOS_NONE ? -1 : (int64_t)sr.g0;
        }
        if (in1) {
            a.heap_val[(size_t)qi * k + e1] = okey_inv<IsMax>(sr.k1);
            a.heap_ref[(size_t)qi * k + e1] = sr.g1 == SPOS_NONE ? -1 : (int64_t)sr.g1;
        }
        if (lane == 0) {
            a.amb[qi] = amb;
            if (a.unfinished) (void)xcd_local_add(&a.unfinished[xcc_id()], 1u);
        }
    }
}

void launch_select_sorted(const ReplayArgs& a, hipStream_t s) {
    if (a.nq == 0) return;
    const bool tune = a.tuner.enabled != 0;
    const size_t shmem = (tune ? 2000 : 0) + 4 * select_wave_bytes(a.k, tune, a.mask == nullptr, a.trace_cap);
    if (shmem > 160 * 1024) throw std::runtime_error("selection kernel: trace cache beyond LDS");
    const dim3 grid((a.nq + 3) / 4), block(256);
    auto go = [&](auto kern) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) throw std::runtime_error(std::string("selection kernel: cannot reserve LDS: ") + hipGetErrorString(e));
        LAUNCH(kern, grid, block, shmem, s, a);
    };
    auto pick_k = [&](auto is_max, auto masked, auto tn) {
        constexpr bool M = decltype(is_max)::value, MK = decltype(masked)::value, T = decltype(tn)::value;
        if (a.k == 100) return go(select_sorted_kernel<M, MK, T, 100>);
        if (a.k == 10) return go(select_sorted_kernel<M, MK, T, 10>);
        return go(select_sorted_kernel<M, MK, T, 0>);
    };
    auto pick_t = [&](auto is_max, auto masked) {
        if (tune) pick_k(is_max, masked, std::true_type{});
        else pick_k(is_max, masked, std::false_type{});
    };
    auto pick_m = [&](auto is_max) {
        if (a.mask) pick_t(is_max, std::true_type{});
        else pick_t(is_max, std::false_type{});
    };
    if (a.metric == METRIC_L2) pick_m(std::true_type{});
    else pick_m(std::false_type{});
}

// the sorted-array selection applies when positions fit 32 bits (the caller checks ntotal) and the result is the
// reordered one; the heap kernels remain for the scanner API (raw heap out), trace training, k > 128 and on request
bool replay_sorted_applies(const ReplayArgs& a) {
    // (the engine passes a log only where its "select" option allows the sorted form)
    return a.log != nullptr && a.k >= 1 && a.k <= 128 && !a.train.enabled && !a.raw_heap_out;
}

void launch_replay(const ReplayArgs& a, hipStream_t s) {
    if (a.nq == 0) return;  // (chained rounds: nq is the bound the grid is sized by)
    const bool tune = a.tuner.enabled != 0, train = a.train.enabled != 0, geo = tune || train;
    const size_t shmem = (geo ? 2000 : 0) + 4 * replay_wave_bytes(a.k, a.nlist, geo, tune, train, a.trace_cap);
    const dim3 grid((a.nq + 3) / 4), block(256);
    static const bool no_rh = getenv("AUNCEL_AMD_LDS_HEAP") != nullptr;
    if (replay_sorted_applies(a)) {
        launch_select_sorted(a, s);
        return;
    }
    const bool rh = a.k <= 127 && !no_rh;
    // the heap (LDS form) and its sorted view take 16 k bytes per query, four queries per workgroup, of the CU's 160 KiB
    if (shmem > 160 * 1024)
        throw std::runtime_error("k = " + std::to_string(a.k) + " is beyond the selection kernel's LDS heap (" + std::to_string(shmem) +
                                 " bytes of 163840 per workgroup)");
    auto go = [&](auto kern) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) throw std::runtime_error(std::string("selection kernel: cannot reserve LDS: ") + hipGetErrorString(e));
        LAUNCH(kern, grid, block, shmem, s, a);
    };
    // few queries: longer trips (more loads in flight per wave) at the price of fewer resident waves
    const char* nld_s = getenv("AUNCEL_AMD_REPLAY_NLD");  // read per launch: the tests run both variants in one process
    const int nld_env = nld_s ? atoi(nld_s) : 0;
    const bool wide = nld_env ? nld_env >= 32 : (a.nq_hint ? a.nq_hint : a.nq) <= 3072;
    auto pick = [&](auto is_max) {
        constexpr bool M = decltype(is_max)::value;
        if (!rh) return wide ? go(replay_kernel<M, false, 32, 0>) : go(replay_kernel<M, false, 16, 0>);
        if (a.k == 100) return wide ? go(replay_kernel<M, true, 32, 100>) : go(replay_kernel<M, true, 16, 100>);
        if (a.k == 10) return wide ? go(replay_kernel<M, true, 32, 10>) : go(replay_kernel<M, true, 16, 10>);
        return wide ? go(replay_kernel<M, true, 32, 0>) : go(replay_kernel<M, true, 16, 0>);
    };
    if (a.metric == METRIC_L2) pick(std::true_type{});
    else pick(std::false_type{});
}

// ---------------------------------------------------------------------------------------------
// The reference's heap (Heap.h:88-142), replayed over the admission logs: every log entry was admitted, in this order, so
// each is one heap_pop + heap_push.  Either one launch at the end of the search over the flagged queries (final_pass: several
// search contexts share the GPU, whose other work hides it), or one behind every selection round, on a side stream, so that it
// runs under the next round's scan and selection (a lone context: what is left for the end of the search is the last round's
// handful of entries):
//   a query still searching          the entries the round added are replayed and the heap (node order) is saved;
//   a query that finished this round if the selection flagged it (equal values met: the only case in which the heap's history
//                                    decides an id or an output order), the rest of its log is replayed, then heap_reorder
//                                    (Heap.h:295-322) gives its (D, I); the others' results are the sorted arrays already.
// One wave per query; what is left for the end of the search is the last round's handful of entries.
template <bool IsMax, bool RH, int KC>
__global__ __launch_bounds__(256) void tie_fix_kernel(TieFixArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t qi = blockIdx.x * 4 + wave;
    if (qi >= a.nq) return;
    bool fin;
    uint32_t n;
    const uint32_t pos = a.fix_pos[qi];
    if (a.final_pass) {
        if (a.tie_flag[qi] != 1) return;                // (its sorted array is the result, or it was dealt with in its round)
        fin = true;
        n = a.log_cnt[qi];
    } else {
        const uint32_t fin_r = a.fin_round[qi];
        if (fin_r < a.round) return;                    // finished in an earlier round: dealt with then
        fin = fin_r == a.round;
        if (fin && a.tie_flag[qi] != 1) return;         // finished now, and its sorted array is the result
        n = a.log_snap[(size_t)(a.round & 1u) * a.nq + qi];
        if (!fin && n <= pos) return;                   // nothing new to replay
    }
    const int k = KC ? KC : a.k;
    unsigned char* base = smem + (size_t)wave * (((size_t)k * 12 + 15) & ~(size_t)15);
    int64_t* href = reinterpret_cast<int64_t*>(base);
    float* hval = reinterpret_cast<float*>(base + (size_t)k * 8);
    for (int i = lane; i < k; i += 64) {
        hval[i] = a.fix_val[(size_t)qi * k + i];
        href[i] = a.fix_ref[(size_t)qi * k + i];
    }
    wave_sync();
    // RH (k <= 127): the heap in lanes, walked as vector work (VHeap); else in LDS, walked by every lane redundantly
    constexpr bool TWO = KC == 0 || KC > 63;
    auto stored = [](float x) { return IsMax ? fkey(x) : ~fkey(x); };
    auto unstored = [](uint32_t sk) { return fkey_inv(IsMax ? sk : ~sk); };
    VHeap vh{0u, 0u, 0xffffffffu, 0xffffffffu};
    const VHeapLane vc = vh_lane(lane);
    unsigned long long chain0 = 0, chain1 = 0;
    if (RH) {
        if (lane >= 1 && lane <= k) {
            vh.k0 = stored(hval[lane - 1]);
            vh.p0 = (uint32_t)href[lane - 1];
        }
        if (TWO && lane + 64 <= k) {
            vh.k1 = stored(hval[lane + 63]);
            vh.p1 = (uint32_t)href[lane + 63];
        }
        for (int i = k; i >= 1; i >>= 1) {
            if (i < 64) chain0 |= 1ull << i;
            else chain1 |= 1ull << (i - 64);
        }
    }
    const uint2* qlog = a.log + (size_t)qi * a.log_cap;
    uint2 e = pos + lane < n ? qlog[pos + lane] : make_uint2(0u, 0u);
    for (uint32_t b = pos; b < n; b += 64) {
        const uint2 cur = e;
        if (b + 64 + lane < n) e = qlog[b + 64 + lane];
        const uint32_t cnt = n - b < 64u ? n - b : 64u;
        for (uint32_t l = 0; l < cnt; l++) {
            const float val = __uint_as_float(rl_u(cur.x, (int)l));
            const uint32_t g32 = rl_u(cur.y, (int)l);
            if (RH) {
                vh_pop<TWO>(vh, vc, lane, vh_key_at<TWO>(vh, k), vh_pay_at<TWO>(vh, k));
                vh_push<TWO>(vh, vc, k, chain0, chain1, stored(val), g32);
            } else {
                heap_pop<IsMax>(k, hval, href);
                heap_push<IsMax>(k, hval, href, val, (int64_t)g32);
            }
        }
    }
    wave_sync();
    if (!fin) {
        if (RH) {  // the heap in node order
            if (lane >= 1 && lane <= k) {
                hval[lane - 1] = unstored(vh.k0);
                href[lane - 1] = vh.p0 == 0xffffffffu ? -1 : (int64_t)vh.p0;
            }
            if (TWO && lane + 64 <= k) {
                hval[lane + 63] = unstored(vh.k1);
                href[lane + 63] = vh.p1 == 0xffffffffu ? -1 : (int64_t)vh.p1;
            }
            wave_sync();
        }
        for (int i = lane; i < k; i += 64) {
            a.fix_val[(size_t)qi * k + i] = hval[i];
            a.fix_ref[(size_t)qi * k + i] = href[i];
        }
        if (lane == 0) a.fix_pos[qi] = n;
        return;
    }
    // heap_reorder (Heap.h:295-322): pop everything, worst first, filling the row from its end; empty entries are dropped
    int ii = 0;
    if (RH) {
        // (the popped root's value and position go to a staging pair, one lane per output position)
        uint32_t out_key = 0, out_pay = 0;  // lane j <-> output position k - 1 - j (and k - 65 - j in the second pair)
        uint32_t out_key2 = 0, out_pay2 = 0;
        for (int sz = k; sz >= 1; sz--) {
            const uint32_t rk = rl_u(vh.k0, 1), rp = rl_u(vh.p0, 1);
            vh_pop<TWO>(vh, vc, lane, vh_key_at<TWO>(vh, sz), vh_pay_at<TWO>(vh, sz));
            // the heap is one node shorter: node sz is no node any more
            vh.k0 = lane == sz ? 0u : vh.k0;
            if (TWO) vh.k1 = lane + 64 == sz ? 0u : vh.k1;
            if (rp != 0xffffffffu) {
                if (ii < 64) wl2_u(out_key, rk, out_pay, rp, ii);
                else wl2_u(out_key2, rk, out_pay2, rp, ii - 64);
                ii++;
            }
        }
        // the j-th valid entry popped belongs at position k - 1 - j (the valid ones end up in [k - ii, k), best first)
        wave_sync();
        if (lane < ii && lane < 64) {
            hval[k - 1 - lane] = unstored(out_key);
            href[k - 1 - lane] = (int64_t)out_pay;
        }
        if (lane + 64 < ii) {
            hval[k - 65 - lane] = unstored(out_key2);
            href[k - 65 - lane] = (int64_t)out_pay2;
        }
    } else {
        for (int i = 0; i < k; i++) {
            const float v = hval[0];
            const int64_t id = href[0];
            heap_pop<IsMax>(k - i, hval, href);
            hval[k - ii - 1] = v;
            href[k - ii - 1] = id;
            if (id != -1) ii++;
        }
    }
    wave_sync();
    for (int i = lane; i < k; i += 64) {
        float v = hneutral<IsMax>();
        int64_t id = -1;
        if (i < ii) {
            v = hval[k - ii + i];
            const uint32_t g = (uint32_t)href[k - ii + i];
            id = a.identity_ids ? (int64_t)g : a.store_pairs ? pair_of_gpos(a.list_off, a.nlist, g) : a.ids[g];
        }
        a.D[(size_t)qi * k + i] = v;
        a.I[(size_t)qi * k + i] = id;
    }
    if (lane == 0) a.tie_flag[qi] = 2;  // (counted by the engine's statistics)
}

void launch_tie_fix(const TieFixArgs& a, hipStream_t s) {
    if (a.nq == 0) return;
    const size_t shmem = 4 * (((size_t)a.k * 12 + 15) & ~(size_t)15);
    const dim3 grid((a.nq + 3) / 4), block(256);
    auto go = [&](auto kern) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) throw std::runtime_error(std::string("tie_fix kernel: cannot reserve LDS: ") + hipGetErrorString(e));
        LAUNCH(kern, grid, block, shmem, s, a);
    };
    auto pick = [&](auto is_max) {
        constexpr bool M = decltype(is_max)::value;
        if (a.k > 127) return go(tie_fix_kernel<M, false, 0>);
        if (a.k == 100) return go(tie_fix_kernel<M, true, 100>);
        if (a.k == 10) return go(tie_fix_kernel<M, true, 10>);
        return go(tie_fix_kernel<M, true, 0>);
    };
    if (a.metric == METRIC_L2) pick(std::true_type{});
    else pick(std::false_type{});
}
}  // namespace amdivf
