// Device-side round planning for the adaptive (Auncel) search.
//
// A round scans probes [stage, stage + cnt) of every unfinished query.  Everything the scan and replay kernels
// need for a round -- per-query probe counts, the query-major layout of the distance rows, the (query, probe)
// pairs grouped by inverted list, the packed query groups and the tile list -- is derived here on the GPU from
// the per-query state the replay kernel left behind, so that a round costs the host one 32-byte read-back
// instead of a pass over every pair.
#include "ivf_dev.h"

#include <stdlib.h>

namespace amdivf {

// ---- 1. how many probes does each query run this round, and how many distances is that (one wave per query:
//         the probes of a query are spread over the lanes)
// (the first workgroup, before anything else of a planning pass: what used to be a launch of its own -- nothing in the counts
// phase writes the counters, the prefix phase, one block, does)
__device__ __forceinline__ void plan_begin(const PlanArgs& a, uint32_t tid, uint32_t nthreads) {
    if (tid < 16 && a.history) a.history[tid] = a.counters[tid];
    if (tid == 0 && a.first_plan) {  // the accumulators of a search (bytes, tile slots) and its per-round marks
        a.bytes[0] = 0.0;
        a.acc64[0] = a.acc64[1] = 0ull;
        if (a.min_bytes) a.min_bytes[0] = 0.0;
        if (a.min_bytes_thr) a.min_bytes_thr[0] = 0.0;
    }
    if (a.first_plan && a.round_unfinished)
        for (uint32_t r = tid; r < PLAN_MAX_ROUNDS * 8; r += nthreads) a.round_unfinished[r] = 0;
}
// one wave: query i
__device__ __forceinline__ void plan_counts_query(const PlanArgs& a, uint32_t i, uint32_t lane) {
    const bool have = i < a.nq;
    uint32_t cnt = 0, pad = 0, more = 0;
    unsigned long long need = 0;
    const unsigned long long ra = a.row_align - 1;
    if (have && !a.done[i]) {
        const uint32_t stage = a.stage[i];
        unsigned long long target = (unsigned long long)stage + a.round_len;
        const unsigned long long np = a.my_nprobe ? a.my_nprobe[a.id_offset + i] : 0ull;
        if (a.tune) {
            // an unfired query at stage s cannot stop before floor((s+1) * multipler): that much is waste-free;
            // beyond it `grow` trades over-scan against the number of rounds
            const unsigned long long safe = (unsigned long long)((float)(stage + 1) * a.multipler);
            const unsigned long long g = (unsigned long long)((double)stage * a.grow);
            target = safe > g ? safe : g;
            const unsigned long long inc = stage == 0 ? a.first_round : a.min_inc;
            if (target < (unsigned long long)stage + inc) target = (unsigned long long)stage + inc;
        }
        if (np != 0) target = np > (unsigned long long)stage + 1 ? np : (unsigned long long)stage + 1;
        bool last = false;
        if (a.budget_ms) {
            // The reference leaves the probe loop after probe ik when another one is not expected to fit:
            // el >= 0.95 * budget - el / (ik + 1)  (IndexIVF.cpp:545-549).  A round covers several probes, so the same
            // estimate (el / stage per probe) gives the number that still fit; a query takes them all in this round
            // and ends with it, unless that would more than double its stage -- then it is re-examined after the round.
            target = (unsigned long long)stage + a.first_round;
            if (stage > 0) {
                const float per = a.elapsed_ms / (float)stage, room = 0.95f * a.budget_ms[a.id_offset + i] - a.elapsed_ms;
                const float fit = room > 0.f ? floorf(room / fmaxf(per, 1e-9f)) : 0.f;
                const unsigned long long cap = stage > a.first_round ? stage : a.first_round;
                unsigned long long len = fit >= (float)a.total_nprobe ? a.total_nprobe : (unsigned long long)fit;
                if (len < 1) len = 1;  // a query never ends between rounds: its final state is written by the replay
                last = len <= cap;
                target = (unsigned long long)stage + (last ? len : cap);
            }
        }
        if (target > a.total_nprobe) target = a.total_nprobe;
        if (a.run_dis && target > stage) {
            // (entries behind what was ranked are all (neutral, -1): not a run)
            const float* dq = a.run_dis + (size_t)i * a.key_stride;
            const int64_t* ka = a.keys + (size_t)i * a.key_stride;
            while (target < a.total_nprobe && dq[target - 1] == dq[target] && ka[target] >= 0) target++;
        }
        if (a.limit && lane == 0) a.limit[i] = last ? (uint32_t)target : a.total_nprobe;
        if (target <= stage) target = stage + 1 < a.total_nprobe ? stage + 1 : a.total_nprobe;
        cnt = (uint32_t)(target - stage);
        // queries this round cannot be the last one for: the host skips the next planning pass when there are none
        const bool ends = target >= a.total_nprobe || last || (a.tune && np != 0);
        more = ends ? 0u : 1u;
        const int64_t* kq = a.keys + (size_t)i * a.key_stride + stage;
        for (uint32_t p = lane; p < cnt; p += 64) {
            const int64_t key = kq[p];
            if (key >= 0 && (unsigned long long)key < a.nlist) {
                const unsigned long long sz = a.list_off[key + 1] - a.list_off[key], psz = (sz + ra) & ~ra;
                need += psz;
                pad += (uint32_t)(psz - sz);
            }
        }
        for (int off = 32; off; off >>= 1) {
            need += __shfl_xor(need, off);
            pad += __shfl_xor(pad, off);
        }
    }
    if (have && lane == 0) {
        a.cnt[i] = cnt;
        a.need[i] = need;
        a.pad[i] = pad | (more << 31);  // (top bit: this round cannot be the query's last; plan_prefix_kernel counts and clears it)
    }
}
__global__ __launch_bounds__(256) void plan_counts_kernel(PlanArgs a) {
    // the per-(XCD, list) pair histogram of the round starts from zero (plan_segments_kernel, next on the stream, fills it)
    for (uint32_t l = blockIdx.x * 256 + threadIdx.x; l < 8 * a.nlist; l += gridDim.x * 256) a.xcount[l] = 0;
    if (blockIdx.x == 0) plan_begin(a, threadIdx.x, 256);
    plan_counts_query(a, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// The XCD this wave runs on, and an add that is served by that XCD's L2 (workgroup scope: no trip to the memory side).  Pairs are
// counted per (XCD, list): thousands of returning adds on ONE address from all eight XCDs are served one after the other at the
// memory side, ~50 ns each -- a list probed by half the batch (skewed data), or a cursor every workgroup bumps, cost a planning
// kernel 0.3 ms that way (cfg 5: profiles/r04_timeline_cfg5.txt); eight tables that no two XCDs share are coherent in their L2s
// (scratch/ubench/xcc_atomic.hip: exact slot sets, 0.6 of the time even without contention).
// exclusive prefix sums of N values per thread over a block of 1024 threads (16 waves): shuffles inside the waves, one LDS hop
// for the wave totals; `total` receives the block's sums.  s_wave: 17 x N entries of shared memory, reusable on return.  All N at
// once: the three barriers are what a scan costs a single workgroup that has nothing else to run (six scans one after the other
// were 18 of the ~35 barriers of a planning pass)
template <int N> __device__ __forceinline__ void block_scan_1024_n(const uint32_t (&v)[N], uint32_t (&excl)[N], uint32_t (&total)[N],
                                                                    uint32_t (*s_wave)[N]) {
    // (the workgroup's own size: 1024 threads in the one-workgroup planner of small calls, 256 in the kernels of large ones -- a
    // workgroup of 16 waves waits for a CU with 16 free wave slots, which among other searches' kernels takes longer than its work)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    uint32_t incl[N];
#pragma unroll
    for (int k = 0; k < N; k++) incl[k] = v[k];
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl[k], off);
            if (lane >= off) incl[k] += o;
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < N; k++) s_wave[w][k] = incl[k];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            const uint32_t x = lane < nw ? s_wave[lane][k] : 0u;
            uint32_t inc = x;
            for (int off = 1; off < 16; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
                if (lane >= off) inc += o;
            }
            if (lane < nw) s_wave[lane][k] = inc - x;
            if (lane == nw - 1) s_wave[16][k] = inc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        excl[k] = s_wave[w][k] + incl[k] - v[k];
        total[k] = s_wave[16][k];
    }
    __syncthreads();
}

// ---- 2. one block: prefix sums over the queries, budget cut, list of active queries.  Thread t owns the queries
//         [t * per, (t + 1) * per): sums of its own, one block scan of the 1024 sums, then its queries again.
__device__ __forceinline__ void plan_prefix_body(const PlanArgs& a) {
    __shared__ uint32_t cut;
    const uint32_t t = threadIdx.x;
    if (t == 0) cut = 0xffffffffu;
    const uint32_t nt = blockDim.x;
    const uint32_t per = (a.nq + nt - 1) / nt, q0 = t * per < a.nq ? t * per : a.nq, q1 = q0 + per < a.nq ? q0 + per : a.nq;
    unsigned long long my_need = 0;
    uint32_t my_cnt = 0, my_act = 0;
    for (uint32_t i = q0; i < q1; i++) {
        const uint32_t c = a.cnt[i];
        my_cnt += c;
        my_act += c ? 1u : 0u;
        my_need += a.need[i];
    }
    // one scan for all three; the 64-bit row need travels as (low 20 bits | the rest): the parts' sums over 1024 threads stay
    // below 2^30 and (total >> 20) + 1024, and the sum is linear in the parts
    unsigned long long tot_need;
    uint32_t ecnt, eact;
    unsigned long long eneed;
    {
        __shared__ uint32_t s_w4[17][4];
        const uint32_t v4[4] = {(uint32_t)(my_need & 0xfffffu), (uint32_t)(my_need >> 20), my_cnt, my_act};
        uint32_t e4[4], t4[4];
        block_scan_1024_n<4>(v4, e4, t4, s_w4);
        eneed = (unsigned long long)e4[0] + ((unsigned long long)e4[1] << 20);
        tot_need = (unsigned long long)t4[0] + ((unsigned long long)t4[1] << 20);
        ecnt = e4[2];
        eact = e4[3];
    }
    for (uint32_t i = q0; i < q1; i++) {
        const uint32_t c = a.cnt[i];
        const unsigned long long nd = a.need[i];
        // a query that does not fit the distance / segment budget of this round waits for the next one
        const bool fits = (eneed + nd <= a.dist_budget && ecnt + c <= a.seg_cap) || eact == 0;
        if (c && !fits) atomicMin(&cut, i);
        a.seg_begin[i] = ecnt;
        a.dist_base[i] = eneed;
        // launch positions of the active queries: their rank among the active ones (the deferred ones, all behind the cut, land
        // behind every kept one and are not launched)
        if (c) a.qsel[eact] = i;
        eneed += nd;
        ecnt += c;
        eact += c ? 1u : 0u;
    }
    __syncthreads();
    // queries at or after the cut are deferred; everything before keeps its prefix values
    uint32_t nact = 0, nseg = 0, nmore = 0, ndef = 0;
    unsigned long long ndist = 0;
    __shared__ uint32_t s_nact, s_nseg, s_more, s_def;
    __shared__ unsigned long long s_ndist;
    if (t == 0) {
        s_nact = 0;
        s_nseg = 0;
        s_more = 0;
        s_def = 0;
        s_ndist = 0;
    }
    __syncthreads();
    for (uint32_t i = t; i < a.nq; i += nt) {
        uint32_t c = a.cnt[i];
        const uint32_t pm = a.pad[i], pad = pm & 0x7fffffffu;
        if (pm >> 31) a.pad[i] = pad;
        if (i >= cut && c) {
            a.cnt[i] = c = 0;
            ndef++;  // deferred to the next round
        }
        if (c) {
            nact++;
            nseg += c;
            ndist += a.need[i] - pad;
            nmore += pm >> 31;
        }
    }
    atomicAdd(&s_nact, nact);
    atomicAdd(&s_nseg, nseg);
    atomicAdd(&s_more, nmore);
    atomicAdd(&s_def, ndef);
    atomicAdd(&s_ndist, ndist);
    __syncthreads();
    if (t == 0) {
        a.counters[0] = s_nact;
        a.counters[1] = s_nseg;
        if (a.cl_cursor)  // (compact_rows_kernel's per-XCD places in the arena of long candidate lists: every round starts at 0)
            for (int x = 0; x < 8; x++) a.cl_cursor[32 * x] = 0;
        a.counters[10] = s_more + s_def;     // queries that may still be unfinished after this round
        a.counters[11] = s_def;              // of those: deferred by the budget cut (the selection of this round does not see them)
        a.counters[7] = (uint32_t)(s_ndist >> 20);  // MiB of distances, for bookkeeping
        // Mi-floats of row space the round WANTED (its queries' padded rows, deferred ones included): what the engine sizes its
        // distance workspace by for the next search of the same shape
        a.counters[12] = (uint32_t)((tot_need + (1u << 20) - 1) >> 20);
        a.bytes[0] += (double)s_ndist * (double)a.d * 4.0;
        if (a.min_bytes) a.min_bytes[0] += a.dense_round ? (double)s_ndist * 4.0 : (double)s_ndist / 8.0;
        if (a.min_bytes_thr && !a.dense_round) a.min_bytes_thr[0] += (double)s_ndist / 8.0;
    }
}
__global__ __launch_bounds__(1024) void plan_prefix_kernel(PlanArgs a) { plan_prefix_body(a); }

// ---- 3. segments of every active query + histogram of pairs per list (one wave per query)
// one wave: the segments of active query i (c probes), whose launch position is `slot`
__device__ __forceinline__ void plan_segments_query(const PlanArgs& a, uint32_t i, uint32_t c, uint32_t lane) {
    const uint32_t stage = a.stage[i];
    const uint32_t xcc = xcc_id();
    const int64_t* kq = a.keys + (size_t)i * a.key_stride + stage;
    unsigned long long cur = a.dist_base[i];  // wave-uniform running offset
    const unsigned long long ra = a.row_align - 1;
    const uint32_t sb = a.seg_begin[i];
    for (uint32_t p0 = 0; p0 < c; p0 += 64) {
        const uint32_t p = p0 + lane;
        int64_t key = -1;
        unsigned long long psz = 0;
        if (p < c) {
            key = kq[p];
            if (key >= 0 && (unsigned long long)key < a.nlist) {
                const unsigned long long sz = a.list_off[key + 1] - a.list_off[key];
                if (sz) {
                    // this pair's place among the pairs of its list that were counted on this XCD (plan_scatter_query adds the
                    // list's start and the XCD's offset inside the list)
                    a.seg_slot[sb + p] = (xcc << 28) | xcd_local_add(&a.xcount[xcc * a.nlist + (uint32_t)key], 1u);
                    psz = (sz + ra) & ~ra;
                }
            }
        }
        unsigned long long incl = psz;  // inclusive prefix sum over the lanes
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(incl, off);
            if ((int)lane >= off) incl += o;
        }
        if (p < c) {
            a.seg_list[sb + p] = (int32_t)key;
            a.seg_off[sb + p] = cur + incl - psz;
        }
        cur += __shfl(incl, 63);
    }
}
__global__ __launch_bounds__(256) void plan_segments_kernel(PlanArgs a) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t c = i < a.nq ? a.cnt[i] : 0u;
    if (c) plan_segments_query(a, i, c, threadIdx.x & 63);
}

// ---- 4. one block: per-list pair offsets, query-group bases, tile counts per workgroup shape (thread t owns a run of lists)

__device__ __forceinline__ void plan_lists_body(const PlanArgs& a) {
    __shared__ uint32_t s_w[17][6];
    const uint32_t t = threadIdx.x;
    // pairs, groups, tiles of shape 1, 2, 4, 8 of list l
    auto values = [&](uint32_t l, uint32_t c, uint32_t (&v)[6]) {
#pragma unroll
        for (int k = 0; k < 6; k++) v[k] = 0;
        if (!c) return;
        const unsigned long long sz = a.list_off[l + 1] - a.list_off[l];
        v[0] = c;
        v[1] = (c + SCAN_RQ - 1) / SCAN_RQ;
        if (a.mfma_chunk) {  // byte codes: one item per (chunk of the list, block of 32 queries), all in the last class
            v[5] = ((c + a.mfma_qblock - 1) / a.mfma_qblock) * (uint32_t)((sz + a.mfma_chunk - 1) / a.mfma_chunk);
            return;
        }
        const uint32_t full = c / a.qblock, rem = c % a.qblock;
        if (full) {
            const uint32_t qg = scan_shape_of(a.qblock);
            const uint32_t tv = scan_tile_vecs(qg);
            v[2 + scan_qg_class(qg)] = full * (uint32_t)((sz + tv - 1) / tv);
        }
        if (rem) {
            const uint32_t qg = scan_shape_of(rem);
            const uint32_t tv = scan_tile_vecs(qg);
            v[2 + scan_qg_class(qg)] += (uint32_t)((sz + tv - 1) / tv);
        }
    };
    // tile bookkeeping of a list in closed form (what its items will compute / what was asked for): (sum over its query blocks) x
    // (sum over its chunks) -- here, by the one workgroup that walks every list anyway, not by one atomic pair per wave of the
    // item kernel
    auto book = [&](uint32_t l, uint32_t c, unsigned long long& slots, unsigned long long& useful) {
        if (!c) return;
        const unsigned long long sz = a.list_off[l + 1] - a.list_off[l];
        useful += (unsigned long long)c * sz;
        if (a.mfma_chunk) {
            const unsigned long long nvb = (sz + a.mfma_chunk - 1) / a.mfma_chunk, lastv = sz - (nvb - 1) * a.mfma_chunk;
            const unsigned long long vsum = (nvb - 1) * (((unsigned long long)a.mfma_chunk + 63) / 64 * 64) + (lastv + 63) / 64 * 64;
            const unsigned long long nqb = (c + a.mfma_qblock - 1) / a.mfma_qblock, lastq = c - (nqb - 1) * a.mfma_qblock;
            const unsigned long long qsum = (nqb - 1) * (((unsigned long long)a.mfma_qblock + MFMA_QBLOCK - 1) / MFMA_QBLOCK * MFMA_QBLOCK) +
                                            (lastq + MFMA_QBLOCK - 1) / MFMA_QBLOCK * MFMA_QBLOCK;
            slots += qsum * vsum;
            return;
        }
        const uint32_t full = c / a.qblock, rem = c % a.qblock;
        if (full) {
            const uint32_t tv = scan_tile_vecs(scan_shape_of(a.qblock));
            slots += (unsigned long long)full * ((a.qblock + SCAN_RQ - 1) / SCAN_RQ * SCAN_RQ) * ((sz + tv - 1) / tv) * tv;
        }
        if (rem) {
            const uint32_t tv = scan_tile_vecs(scan_shape_of(rem));
            slots += (unsigned long long)((rem + SCAN_RQ - 1) / SCAN_RQ * SCAN_RQ) * ((sz + tv - 1) / tv) * tv;
        }
    };
    const uint32_t nt = blockDim.x;
    const uint32_t per = (a.nlist + nt - 1) / nt, l0 = t * per < a.nlist ? t * per : a.nlist, l1 = l0 + per < a.nlist ? l0 + per : a.nlist;
    uint32_t mine[6] = {0, 0, 0, 0, 0, 0};
    double list_bytes = 0;
    unsigned long long my_slots = 0, my_useful = 0;
    for (uint32_t l = l0; l < l1; l++) {
        // the list's pairs = the sum of its eight per-XCD counts, which become the XCDs' offsets inside the list
        uint32_t c = 0;
#pragma unroll
        for (uint32_t x = 0; x < 8; x++) {
            const uint32_t cx = a.xcount[x * a.nlist + l];
            a.xcount[x * a.nlist + l] = c;
            c += cx;
        }
        a.lcount[l] = c;
        book(l, c, my_slots, my_useful);
        uint32_t v[6];
        values(l, c, v);
#pragma unroll
        for (int k = 0; k < 6; k++) mine[k] += v[k];
        if (v[0]) list_bytes += (double)(a.list_off[l + 1] - a.list_off[l]) * (double)a.row_bytes;
    }
    for (int off = 32; off; off >>= 1) {
        my_slots += __shfl_xor(my_slots, off);
        my_useful += __shfl_xor(my_useful, off);
    }
    if ((t & 63) == 0 && my_slots) {  // (16 waves)
        atomicAdd(&a.acc64[0], my_slots);
        atomicAdd(&a.acc64[1], my_useful);
    }
    if (a.min_bytes && list_bytes > 0) atomicAdd(a.min_bytes, list_bytes);
    if (a.min_bytes_thr && !a.dense_round && list_bytes > 0) atomicAdd(a.min_bytes_thr, list_bytes);
    uint32_t ex[6], tot[6];
    block_scan_1024_n<6>(mine, ex, tot, s_w);
    for (uint32_t l = l0; l < l1; l++) {
        uint32_t v[6];
        values(l, a.lcount[l], v);
        a.lstart[l] = ex[0];
        a.gbase[l] = ex[1];
        a.ibase[0 * a.nlist + l] = ex[2];
        a.ibase[1 * a.nlist + l] = ex[3];
        a.ibase[2 * a.nlist + l] = ex[4];
        a.ibase[3 * a.nlist + l] = ex[5];
#pragma unroll
        for (int k = 0; k < 6; k++) ex[k] += v[k];
    }
    if (t == 0) {
        // more tiles than the item list holds: the search fails with ERR_ITEM_OVERFLOW when its error word is read; until then
        // the scans of this round, already enqueued, must find nothing to do (the items past the capacity were never written)
        const bool over = (unsigned long long)tot[2] + tot[3] + tot[4] + tot[5] > (unsigned long long)a.item_cap;
        if (over) {
            atomicMax(a.error, ERR_ITEM_OVERFLOW);
            tot[2] = tot[3] = tot[4] = tot[5] = 0;
        }
        a.counters[2] = tot[0];  // pairs
        a.counters[3] = tot[1];  // query groups
        a.counters[4] = tot[2];  // tiles of shape 1
        a.counters[5] = tot[3];  // tiles of shape 2
        a.counters[8] = tot[4];  // tiles of shape 4
        a.counters[9] = tot[5];  // tiles of shape 8
    }
}
__global__ __launch_bounds__(1024) void plan_lists_kernel(PlanArgs a) { plan_lists_body(a); }

// ---- 5. pairs into their list's range (one wave per query)
__device__ __forceinline__ void plan_scatter_query(const PlanArgs& a, uint32_t i, uint32_t lane) {
    if (i >= a.nq) return;
    const uint32_t c = a.cnt[i];
    if (!c) return;
    const uint32_t sb = a.seg_begin[i];
    for (uint32_t p = lane; p < c; p += 64) {
        const int32_t key = a.seg_list[sb + p];
        if (key < 0 || (uint32_t)key >= a.nlist) continue;
        if (a.list_off[key + 1] == a.list_off[key]) continue;
        const uint32_t ss = a.seg_slot[sb + p];
        const uint32_t pos = a.lstart[key] + a.xcount[(ss >> 28) * a.nlist + (uint32_t)key] + (ss & 0x0fffffffu);
        a.pair_query[pos] = a.slot_base + i;
        a.pair_out[pos] = a.seg_off[sb + p];
    }
}

// ---- 6. tiles and query groups of every list (one thread per list), shapes 1 | 2 | 4 in three item ranges
__device__ inline void items_of_list(const PlanArgs& a, uint32_t l, uint32_t c) {
    const uint32_t p0 = a.lstart[l], g0 = a.gbase[l];
    for (uint32_t o = 0, g = 0; o < c; o += SCAN_RQ, g++) {
        a.group_p0[g0 + g] = p0 + o;
        a.group_cnt[g0 + g] = c - o < (uint32_t)SCAN_RQ ? c - o : (uint32_t)SCAN_RQ;
    }
    const uint32_t n1 = a.counters[4], n2 = a.counters[5], n4 = a.counters[8];
    uint32_t cur[4] = {a.ibase[l], n1 + a.ibase[a.nlist + l], n1 + n2 + a.ibase[2 * a.nlist + l],
                       n1 + n2 + n4 + a.ibase[3 * a.nlist + l]};
    const uint32_t sz = (uint32_t)(a.list_off[l + 1] - a.list_off[l]);
    const uint64_t vb0 = a.list_off[l];
    const uint64_t lb0 = a.lane_block_off ? a.lane_block_off[l] : 0ull;  // (the lane-ordered copy's blocks: scan_vec_base)
    if (a.mfma_chunk) {
        // items of one chunk are consecutive (its query blocks): they run close together on one XCD (scan_mfma_kernel's
        // item order), so a chunk fetched for one query block is still in that L2 for the next
        uint32_t ni = cur[3];
        const uint64_t b0 = a.block_off[l];
        // (item_order 1: query block major -- the query blocks of a list pass over it one after the other, each finding it in
        // the L2 / Infinity Cache the previous one left it in, instead of three waves requesting the same lines at once)
        const uint32_t nvb = (sz + a.mfma_chunk - 1) / a.mfma_chunk, nqb = (c + a.mfma_qblock - 1) / a.mfma_qblock;
        for (uint32_t o = 0; o < nvb * nqb; o++) {
            const uint32_t vb = (a.item_order ? o % nvb : o / nqb) * a.mfma_chunk, qb = (a.item_order ? o / nvb : o % nqb) * a.mfma_qblock;
            {
                ScanItem it;
                it.vec_base = b0 + vb / MFMA_BLOCK;
                it.nvec = sz - vb < a.mfma_chunk ? sz - vb : a.mfma_chunk;
                it.vec_off = vb;
                it.pair_begin = p0 + qb;
                it.npair = c - qb < a.mfma_qblock ? c - qb : a.mfma_qblock;
                it.qg = 0;
                it.qgroup = (uint32_t)(vb0 + vb);  // global index of the chunk's first vector (the fp32 filter's exact rescoring)
                if (ni < a.item_cap) a.items[ni] = it;
                ni++;
            }
        }
        return;
    }
    for (uint32_t qb = 0; qb < c; qb += a.qblock) {
        const uint32_t nq_blk = c - qb < a.qblock ? c - qb : a.qblock;
        const uint32_t qg = scan_shape_of(nq_blk);
        const uint32_t tv = scan_tile_vecs(qg);
        uint32_t& ni = cur[scan_qg_class(qg)];
        for (uint32_t vb = 0; vb < sz; vb += tv) {
            ScanItem it;
            it.vec_base = scan_vec_base(vb0 + vb, lb0, vb);
            it.nvec = sz - vb < tv ? sz - vb : tv;
            it.vec_off = vb;
            it.pair_begin = p0 + qb;
            it.npair = nq_blk;
            it.qg = qg;
            it.qgroup = g0 + qb / SCAN_RQ;
            if (ni < a.item_cap) a.items[ni] = it;
            ni++;
        }
    }
}


// the items of list l, one thread (a call of a few queries: plan_small_kernel)
__device__ __forceinline__ void plan_items_lists(const PlanArgs& a, uint32_t l) {
    const uint32_t c = l < a.nlist ? a.lcount[l] : 0u;
    if (c) items_of_list(a, l, c);
}

// ... one WAVE per list, the lanes striding over its query groups and items: with one thread per list the launch took as long as
// its busiest list (a list probed by a fifth of a 10000-query batch: 2500 items written one after the other, 0.29 ms of cfg 1's
// 1.6 ms step).  Same items at the same positions as items_of_list.
__device__ inline void items_of_list_wave(const PlanArgs& a, uint32_t l, uint32_t c, uint32_t lane) {
    const uint32_t p0 = a.lstart[l], g0 = a.gbase[l];
    for (uint32_t g = lane; g * SCAN_RQ < c; g += 64) {
        const uint32_t o = g * SCAN_RQ;
        a.group_p0[g0 + g] = p0 + o;
        a.group_cnt[g0 + g] = c - o < (uint32_t)SCAN_RQ ? c - o : (uint32_t)SCAN_RQ;
    }
    const uint32_t n1 = a.counters[4], n2 = a.counters[5], n4 = a.counters[8];
    const uint32_t cur[4] = {a.ibase[l], n1 + a.ibase[a.nlist + l], n1 + n2 + a.ibase[2 * a.nlist + l],
                             n1 + n2 + n4 + a.ibase[3 * a.nlist + l]};
    const uint32_t sz = (uint32_t)(a.list_off[l + 1] - a.list_off[l]);
    const uint64_t vb0 = a.list_off[l];
    const uint64_t lb0 = a.lane_block_off ? a.lane_block_off[l] : 0ull;  // (the lane-ordered copy's blocks: scan_vec_base)
    if (a.mfma_chunk) {
        const uint64_t b0 = a.block_off[l];
        const uint32_t nvb = (sz + a.mfma_chunk - 1) / a.mfma_chunk, nqb = (c + a.mfma_qblock - 1) / a.mfma_qblock;
        for (uint32_t o = lane; o < nvb * nqb; o += 64) {
            const uint32_t vb = (a.item_order ? o % nvb : o / nqb) * a.mfma_chunk, qb = (a.item_order ? o / nvb : o % nqb) * a.mfma_qblock;
            ScanItem it;
            it.vec_base = b0 + vb / MFMA_BLOCK;
            it.nvec = sz - vb < a.mfma_chunk ? sz - vb : a.mfma_chunk;
            it.vec_off = vb;
            it.pair_begin = p0 + qb;
            it.npair = c - qb < a.mfma_qblock ? c - qb : a.mfma_qblock;
            it.qg = 0;
            it.qgroup = (uint32_t)(vb0 + vb);
            if (cur[3] + o < a.item_cap) a.items[cur[3] + o] = it;
        }
        return;
    }
    // fp32 tiles: the full query blocks (all of one shape) come first, each with its tiles; then the remainder block's
    const uint32_t full = c / a.qblock, rem = c % a.qblock;
    const uint32_t qg_f = scan_shape_of(a.qblock), tv_f = scan_tile_vecs(qg_f), nt_f = (sz + tv_f - 1) / tv_f;
    const int cls_f = scan_qg_class(qg_f);
    for (uint32_t o = lane; o < full * nt_f; o += 64) {
        const uint32_t qb = (o / nt_f) * a.qblock, vb = (o % nt_f) * tv_f;
        ScanItem it;
        it.vec_base = scan_vec_base(vb0 + vb, lb0, vb);
        it.nvec = sz - vb < tv_f ? sz - vb : tv_f;
        it.vec_off = vb;
        it.pair_begin = p0 + qb;
        it.npair = a.qblock;
        it.qg = qg_f;
        it.qgroup = g0 + qb / SCAN_RQ;
        if (cur[cls_f] + o < a.item_cap) a.items[cur[cls_f] + o] = it;
    }
    if (rem) {
        const uint32_t qg_r = scan_shape_of(rem), tv_r = scan_tile_vecs(qg_r), nt_r = (sz + tv_r - 1) / tv_r;
        const int cls_r = scan_qg_class(qg_r);
        const uint32_t base = cur[cls_r] + (cls_r == cls_f ? full * nt_f : 0u), qb = full * a.qblock;
        for (uint32_t o = lane; o < nt_r; o += 64) {
            const uint32_t vb = o * tv_r;
            ScanItem it;
            it.vec_base = scan_vec_base(vb0 + vb, lb0, vb);
            it.nvec = sz - vb < tv_r ? sz - vb : tv_r;
            it.vec_off = vb;
            it.pair_begin = p0 + qb;
            it.npair = rem;
            it.qg = qg_r;
            it.qgroup = g0 + qb / SCAN_RQ;
            if (base + o < a.item_cap) a.items[base + o] = it;
        }
    }
}

// ---- 5 + 6 in one launch (both read what plan_lists_kernel left, neither reads the other's output): the first gq workgroups
//      scatter the pairs of four queries each, the rest make the items of four lists each (a wave per list)
__global__ __launch_bounds__(256) void plan_scatter_items_kernel(PlanArgs a, uint32_t gq) {
    if (blockIdx.x < gq) {
        plan_scatter_query(a, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
        return;
    }
    const uint32_t l = (blockIdx.x - gq) * 4 + (threadIdx.x >> 6);
    const uint32_t c = l < a.nlist ? a.lcount[l] : 0u;
    if (c) items_of_list_wave(a, l, c, threadIdx.x & 63);
}

// ---- a call of a few queries (the reference's callers issue one search per query, eval/bound.cpp:391-396): the whole pass in ONE
//      workgroup of 16 waves -- the five launches above are five dependent kernel boundaries and five dispatch latencies for a
//      few hundred pairs; here the phases are separated by workgroup barriers
constexpr uint32_t PLAN_SMALL_NQ = 32;
__global__ __launch_bounds__(1024) void plan_small_kernel(PlanArgs a) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t l = tid; l < 8 * a.nlist; l += 1024) a.xcount[l] = 0;
    plan_begin(a, tid, 1024);
    __syncthreads();  // (the history copy reads the counters before the prefix phase writes them)
    for (uint32_t i = wave; i < a.nq; i += 16) plan_counts_query(a, i, lane);
    __syncthreads();
    plan_prefix_body(a);
    __syncthreads();
    for (uint32_t i = wave; i < a.nq; i += 16) {
        const uint32_t c = a.cnt[i];
        if (c) plan_segments_query(a, i, c, lane);
    }
    __syncthreads();
    plan_lists_body(a);
    __syncthreads();
    for (uint32_t i = wave; i < a.nq; i += 16) plan_scatter_query(a, i, lane);
    for (uint32_t l0 = 0; l0 < a.nlist; l0 += 1024) plan_items_lists(a, l0 + tid);
}

// ---- ONE query (what the reference's callers issue: eval/bound.cpp:391-396).  plan_small_kernel is the general planner behind
//      workgroup barriers: ~20 phases that each pay a trip to L2, 0.06 ms for a dozen pairs -- a quarter of such a call.  With one
//      query every probed list has exactly one pair, so nothing needs counting per list: one wave walks the probes, 64 at a time, and
//      writes segments, pairs (in probe order), query groups and items with running prefix sums.  Same arrays and counters as the
//      general pass (any consistent layout serves: nothing downstream depends on the order of pairs or items).
__global__ __launch_bounds__(64) void plan_one_kernel(PlanArgs a) {
    const uint32_t lane = threadIdx.x;
    plan_begin(a, lane, 64);
    __syncthreads();  // (the history copy reads the counters before they are written below)
    plan_counts_query(a, 0, lane);
    __syncthreads();
    const uint32_t c = a.cnt[0];
    const unsigned long long nd = a.need[0];
    const uint32_t pm = a.pad[0], pad = pm & 0x7fffffffu;
    const unsigned long long ndist = c ? nd - pad : 0ull;
    if (lane == 0) {
        if (pm >> 31) a.pad[0] = pad;
        a.seg_begin[0] = 0;
        a.dist_base[0] = 0;
        if (c) a.qsel[0] = 0;
        a.counters[0] = c ? 1u : 0u;
        a.counters[1] = c;
        a.counters[10] = c ? pm >> 31 : 0u;
        a.counters[11] = 0;
        a.counters[7] = (uint32_t)(ndist >> 20);
        a.counters[12] = (uint32_t)((nd + (1u << 20) - 1) >> 20);
        a.bytes[0] += (double)ndist * (double)a.d * 4.0;
        if (a.min_bytes) a.min_bytes[0] += a.dense_round ? (double)ndist * 4.0 : (double)ndist / 8.0;
        if (a.min_bytes_thr && !a.dense_round) a.min_bytes_thr[0] += (double)ndist / 8.0;
    }
    const uint32_t stage = a.stage[0];
    const int64_t* kq = a.keys + stage;
    const unsigned long long ra = a.row_align - 1;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    unsigned long long cur = 0, slots = 0, useful = 0;  // wave-uniform running offset of the rows | tile bookkeeping (per lane)
    uint32_t np = 0, ni = 0;                            // pairs (= query groups) and items so far
    double list_bytes = 0;
    const uint32_t tv1 = scan_tile_vecs(1);
    for (uint32_t p0 = 0; p0 < c; p0 += 64) {
        const uint32_t p = p0 + lane;
        int64_t key = -1;
        uint32_t sz = 0;
        if (p < c) {
            key = kq[p];
            if (key >= 0 && (unsigned long long)key < a.nlist) sz = (uint32_t)(a.list_off[key + 1] - a.list_off[key]);
        }
        const bool valid = sz != 0;
        const unsigned long long psz = valid ? ((unsigned long long)sz + ra) & ~ra : 0ull;
        unsigned long long incl = psz;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(incl, off);
            if ((int)lane >= off) incl += o;
        }
        const unsigned long long my_off = cur + incl - psz;
        if (p < c) {
            a.seg_list[p] = (int32_t)key;
            a.seg_off[p] = my_off;
        }
        cur += __shfl(incl, 63);
        const unsigned long long vm = __ballot(valid);
        const uint32_t pidx = np + (uint32_t)__builtin_popcountll(vm & lt);
        // items of this probe's list: one query block x its chunks (byte codes, fp32 filter) or its 128-vector tiles (fp32)
        const uint32_t tile = a.mfma_chunk ? a.mfma_chunk : tv1;
        const uint32_t n_it = valid ? (sz + tile - 1) / tile : 0u;
        uint32_t iincl = n_it;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)iincl, off);
            if ((int)lane >= off) iincl += o;
        }
        const uint32_t ibase = ni + iincl - n_it;
        if (valid) {
            a.pair_query[pidx] = a.slot_base;
            a.pair_out[pidx] = my_off;
            a.group_p0[pidx] = pidx;
            a.group_cnt[pidx] = 1;
            const uint64_t vb0 = a.list_off[key];
            const uint64_t b0 = a.mfma_chunk ? a.block_off[key] : 0ull;
            for (uint32_t o = 0; o < n_it; o++) {
                const uint32_t vb = o * tile;
                ScanItem it;
                it.vec_base = a.mfma_chunk ? b0 + vb / MFMA_BLOCK : scan_vec_base(vb0 + vb, a.lane_block_off ? a.lane_block_off[key] : 0ull, vb);
                it.nvec = sz - vb < tile ? sz - vb : tile;
                it.vec_off = vb;
                it.pair_begin = pidx;
                it.npair = 1;
                it.qg = a.mfma_chunk ? 0u : 1u;
                it.qgroup = a.mfma_chunk ? (uint32_t)(vb0 + vb) : pidx;
                if (ibase + o < a.item_cap) a.items[ibase + o] = it;
                slots += a.mfma_chunk ? (unsigned long long)MFMA_QBLOCK * (((it.nvec + 63) / 64) * 64) : (unsigned long long)SCAN_RQ * tile;
            }
            useful += sz;
            list_bytes += (double)sz * (double)a.row_bytes;
        }
        np += (uint32_t)__builtin_popcountll(vm);
        ni += (uint32_t)__shfl((int)iincl, 63);
    }
    for (int off = 32; off; off >>= 1) {
        slots += __shfl_xor(slots, off);
        useful += __shfl_xor(useful, off);
        list_bytes += __shfl_xor(list_bytes, off);
    }
    if (lane == 0) {
        const bool over = ni > a.item_cap;
        if (over) atomicMax(a.error, ERR_ITEM_OVERFLOW);
        const uint32_t nit = over ? 0u : ni;
        a.counters[2] = np;   // pairs
        a.counters[3] = np;   // query groups (one query each)
        a.counters[4] = a.mfma_chunk ? 0u : nit;  // tiles of shape 1
        a.counters[5] = 0;
        a.counters[8] = 0;
        a.counters[9] = a.mfma_chunk ? nit : 0u;  // matrix-core items
        if (a.min_bytes && list_bytes > 0) a.min_bytes[0] += list_bytes;
        if (a.min_bytes_thr && !a.dense_round && list_bytes > 0) a.min_bytes_thr[0] += list_bytes;
        if (slots) {
            a.acc64[0] += slots;
            a.acc64[1] += useful;
        }
    }
}

// five launches a round (round 3: seven -- a reset kernel of one thread, and scatter / items apart)
void launch_plan(const PlanArgs& a, hipStream_t s) {
    if (a.nq == 0) return;
    static const bool no_one = getenv("AUNCEL_AMD_NO_PLAN_ONE") != nullptr;  // (debugging)
    if (a.nq == 1 && !no_one) {
        LAUNCH(plan_one_kernel, dim3(1), dim3(64), 0, s, a);
        return;
    }
    if (a.nq <= PLAN_SMALL_NQ && a.nlist <= 65536) {
        LAUNCH(plan_small_kernel, dim3(1), dim3(1024), 0, s, a);
        return;
    }
    const unsigned gq = (a.nq + 3) / 4 /* one wave per query */, gl = (a.nlist + 3) / 4 /* one wave per list */;
    // threads of the two one-workgroup kernels (prefix sums over the queries, per-list offsets; AUNCEL_AMD_PLAN_THREADS: 64 ... 1024, a
    // multiple of 64).  A workgroup of sixteen waves waits for a CU with sixteen free wave slots, so 256 threads were tried (round 5,
    // profiles/r05_experiments.txt U): with six searches in flight the same within the spread (3.11-3.14 against 3.09-3.14 M q/s),
    // a search alone 0.08 ms slower (2.60 against 2.52 ms: four times the queries per thread) -- 1024 stays.
    static const unsigned plan_threads = [] {
        const char* e = getenv("AUNCEL_AMD_PLAN_THREADS");
        const int v = e ? atoi(e) : 1024;
        return (unsigned)(v >= 64 && v <= 1024 && v % 64 == 0 ? v : 1024);
    }();
    LAUNCH(plan_counts_kernel, dim3(gq), dim3(256), 0, s, a);
    LAUNCH(plan_prefix_kernel, dim3(1), dim3(plan_threads), 0, s, a);
    LAUNCH(plan_segments_kernel, dim3(gq), dim3(256), 0, s, a);
    LAUNCH(plan_lists_kernel, dim3(1), dim3(plan_threads), 0, s, a);
    LAUNCH(plan_scatter_items_kernel, dim3(gq + gl), dim3(256), 0, s, a, (uint32_t)gq);
}


// ---------------------------------------------------------------------------- self-check of the per-XCD counters
// xcd_local_add (ivf_dev.h) adds at workgroup scope from different workgroups: that is exact only because every global atomic of an
// XCD is executed in that XCD's L2 and the tables it is used on are private to an XCD (row xcc_id()).  Search termination depends
// on it (the per-XCD `unfinished` counts), so the assumption is checked where it can fail: every thread adds 1 to counter
// [xcc][key] under contention and keeps the value returned; per (xcc, key) the returned values must be exactly 0 .. count - 1.
__global__ void xcd_check_kernel(uint32_t* counters, uint32_t nkeys, uint32_t* slot, uint32_t* xcc_of, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = (i * 2654435761u) % nkeys;
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 0xfu;  // (unmasked beyond 8: the check reports an XCD number the engine's 8-row tables do not have)
    slot[i] = x < 8 ? xcd_local_add(&counters[x * nkeys + key], 1u) : 0u;
    xcc_of[i] = x;
}
void launch_xcd_check(uint32_t* counters, uint32_t nkeys, uint32_t* slot, uint32_t* xcc_of, uint32_t n, hipStream_t s) {
    LAUNCH(xcd_check_kernel, dim3((n + 255) / 256), dim3(256), 0, s, counters, nkeys, slot, xcc_of, n);
}

}  // namespace amdivf
