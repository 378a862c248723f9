// Pipelined heap sort of one row per wave (the second half of heap_tie_order_kernel, auncel_amd/csrc/ivf_kernels.hip): how fast can
// a tick be made?  Checked against the literal heap sort (pops with the "right child between equals" rule) on rows with runs of
// equal values.
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench/heap_sort.hip -o scratch/ubench/heap_sort && scratch/ubench/heap_sort
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct HeapEnt { float v; uint32_t id; };

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- the form in the product at the start of round 5
__device__ inline void heapsort_v1(HeapEnt* a, uint32_t n, uint32_t* out_id, int lane) {
    uint32_t hole = 0, s = 0, lvl = 0, Lid = 0;
    float Lv = 0.f;
    uint32_t t = 0, since = 2;
    const uint4* a4 = reinterpret_cast<const uint4*>(a);
    const uint2* a2 = reinterpret_cast<const uint2*>(a);
    while (true) {
        const unsigned long long act = __ballot(hole != 0);
        if (t == n && !act) break;
        const uint32_t sc = n - t, dsc = 31 - __builtin_clz(sc | 1);
        const bool above = hole != 0 && lvl <= dsc && (sc >> ((dsc - lvl) & 31)) == hole;
        const bool create = t < n && since >= 2 && !__ballot(above);
        const bool mine = create && lane == __builtin_ctzll(~act);
        hole = mine ? 1u : hole;
        s = mine ? sc : s;
        lvl = mine ? 0u : lvl;
        const uint32_t i1 = hole << 1;
        const uint4 ch = a4[(i1 < n ? i1 : n) >> 1];
        const uint2 ls = a2[mine ? sc : 0u];
        const uint32_t top_id = a2[1].y;
        if (mine) out_id[sc - 1] = top_id;
        Lv = mine ? __uint_as_float(ls.x) : Lv;
        Lid = mine ? ls.y : Lid;
        const float c1v = __uint_as_float(ch.x), c2v = __uint_as_float(ch.z);
        const bool left = i1 == s || c1v > c2v;
        const float cv = left ? c1v : c2v;
        const uint32_t cid = left ? ch.y : ch.w;
        const bool done = i1 > s || Lv > cv;
        if (hole != 0) a[hole] = HeapEnt{done ? Lv : cv, done ? Lid : cid};
        hole = hole != 0 && !done ? (left ? i1 : i1 + 1) : 0u;
        lvl++;
        t += create ? 1u : 0u;
        since = create ? 1u : since + 1;
        wave_sync();
    }
}

// ---- v2: the loads of a tick do not wait for the decision whether a pop starts (the root's children, the last entry and the root
// itself are requested by every lane next to the lane's own walk), walks sit on a ring of lanes by pop number, and the bookkeeping
// that needs no memory runs while the loads are in flight
__device__ inline void heapsort_v2(HeapEnt* a, uint32_t n, uint32_t* out_id, int lane) {
    uint32_t hole = 0, s = 0, lvl = 0, Lid = 0;
    float Lv = 0.f;
    uint32_t t = 0, since = 2;
    const uint4* a4 = reinterpret_cast<const uint4*>(a);
    const uint2* a2 = reinterpret_cast<const uint2*>(a);
    const uint32_t lim = n >> 1;
    while (true) {
        const uint32_t sc = n - t;
        // requests: own children | root + its children | the entry the next pop would take
        const uint32_t i1 = hole << 1;
        const uint4 ch_own = a4[hole < lim ? hole : lim];
        const uint4 r01 = a4[0];  // entries 0 (unused), 1 (root)
        const uint4 r23 = a4[1];  // the root's children
        const uint2 ls = a2[sc];
        // meanwhile
        const unsigned long long act = __ballot(hole != 0);
        if (t == n && !act) break;
        const uint32_t dsc = 31 - __builtin_clz(sc | 1);
        const bool above = hole != 0 && lvl <= dsc && (sc >> ((dsc - lvl) & 31)) == hole;
        const bool create = t < n && since >= 2 && !__ballot(above);
        const bool mine = create && (uint32_t)lane == (t & 15u);
        // the walk's step
        const uint32_t h = mine ? 1u : hole;
        const uint32_t ss = mine ? sc : s;
        const uint32_t j1 = mine ? 2u : i1;
        const uint4 ch = mine ? r23 : ch_own;
        if (mine) out_id[sc - 1] = r01.w;
        Lv = mine ? __uint_as_float(ls.x) : Lv;
        Lid = mine ? ls.y : Lid;
        const float c1v = __uint_as_float(ch.x), c2v = __uint_as_float(ch.z);
        const bool left = j1 == ss || c1v > c2v;
        const float cv = left ? c1v : c2v;
        const uint32_t cid = left ? ch.y : ch.w;
        const bool done = j1 > ss || Lv > cv;
        if (h != 0) a[h] = HeapEnt{done ? Lv : cv, done ? Lid : cid};
        hole = h != 0 && !done ? (left ? j1 : j1 + 1) : 0u;
        s = ss;
        lvl = mine ? 1u : lvl + 1;
        t += create ? 1u : 0u;
        since = create ? 1u : since + 1;
        wave_sync();
    }
}


// ---- v3: two kinds of tick.  A pop can start at most every other tick, so the tick after a start is a plain step of the walks in
// flight (children -> compare -> store), and only the other ticks carry the start logic.  Lanes without a walk step on slot 0
// (unused) instead of being masked off; nothing is clamped (an address past the heap is read but its result is not used).
struct Walk {
    uint32_t hole, s, lvl, Lid;
    float Lv;
};
__device__ __forceinline__ void walk_step(HeapEnt* a, Walk& w, const uint4 ch) {
    const uint32_t j1 = w.hole << 1;
    const float c1v = __uint_as_float(ch.x), c2v = __uint_as_float(ch.z);
    const bool left = j1 == w.s || c1v > c2v;
    const float cv = left ? c1v : c2v;
    const uint32_t cid = left ? ch.y : ch.w;
    const bool done = j1 > w.s || w.Lv > cv;
    reinterpret_cast<uint2*>(a)[w.hole] = make_uint2(__float_as_uint(done ? w.Lv : cv), done ? w.Lid : cid);  // (hole 0: the unused slot)
    const uint32_t nh = left ? j1 : j1 + 1;
    w.hole = done ? 0u : nh;
    w.lvl++;
}
__device__ inline void heapsort_v3(HeapEnt* a, uint32_t n, uint32_t* out_id, int lane) {
    Walk w{0u, 0xffffffffu, 0u, 0u, 0.f};
    uint32_t t = 0;
    const uint4* a4 = reinterpret_cast<const uint4*>(a);
    const uint2* a2 = reinterpret_cast<const uint2*>(a);
    while (true) {
        // ---- a tick that may start pop t
        const uint32_t sc = n - t;
        const uint4 ch_own = a4[w.hole];
        const uint4 r01 = a4[0];
        const uint4 r23 = a4[1];
        const uint2 ls = a2[sc];
        const unsigned long long act = __ballot(w.hole != 0);
        if (t == n) {
            if (!act) break;
            walk_step(a, w, ch_own);
            wave_sync();
            continue;
        }
        const uint32_t dsc = 31 - __builtin_clz(sc);
        const bool above = w.hole != 0 && w.lvl <= dsc && (sc >> ((dsc - w.lvl) & 31)) == w.hole;
        const bool create = !__ballot(above);
        const bool mine = create && (uint32_t)lane == (t & 15u);
        if (mine) out_id[sc - 1] = r01.w;
        w.hole = mine ? 1u : w.hole;
        w.s = mine ? sc : w.s;
        w.lvl = mine ? 0u : w.lvl;
        w.Lv = mine ? __uint_as_float(ls.x) : w.Lv;
        w.Lid = mine ? ls.y : w.Lid;
        uint4 ch;
        ch.x = mine ? r23.x : ch_own.x;
        ch.y = mine ? r23.y : ch_own.y;
        ch.z = mine ? r23.z : ch_own.z;
        ch.w = mine ? r23.w : ch_own.w;
        walk_step(a, w, ch);
        wave_sync();
        if (create) {
            t++;
            // ---- the tick after a start: a plain step
            const uint4 c2 = a4[w.hole];
            walk_step(a, w, c2);
            wave_sync();
        }
    }
}


// ---- v4: v3 without branches inside a tick: the starting pop's loads are unconditional (uniform addresses), every update is a
// select, the id of the popped root goes to a dummy place when no pop starts; depth from the hole itself (no lvl register)
__device__ __forceinline__ void walk_step4(HeapEnt* a, uint32_t& hole, uint32_t s, float Lv, uint32_t Lid, const uint4 ch) {
    const uint32_t j1 = hole << 1;
    const float c1v = __uint_as_float(ch.x), c2v = __uint_as_float(ch.z);
    const bool left = j1 == s || c1v > c2v;
    const float cv = left ? c1v : c2v;
    const uint32_t cid = left ? ch.y : ch.w;
    const bool done = j1 > s || Lv > cv;
    reinterpret_cast<uint2*>(a)[hole] = make_uint2(__float_as_uint(done ? Lv : cv), done ? Lid : cid);
    const uint32_t nh = left ? j1 : j1 + 1;
    hole = done ? 0u : nh;
}
__device__ inline void heapsort_v4(HeapEnt* a, uint32_t n, uint32_t* out_id, int lane) {
    uint32_t hole = 0, s = 0xffffffffu, Lid = 0;
    float Lv = 0.f;
    uint32_t t = 0;
    const uint4* a4 = reinterpret_cast<const uint4*>(a);
    const uint2* a2 = reinterpret_cast<const uint2*>(a);
    const uint32_t lim = n >> 1;
    while (t < n) {
        const uint32_t sc = n - t;
        const uint4 ch_own = a4[hole < lim ? hole : lim];
        const uint4 r01 = a4[0];
        const uint4 r23 = a4[1];
        const uint2 ls = a2[sc];
        // a walk above slot sc?  depth(hole) = 31 - clz(hole); hole is an ancestor-or-self of sc iff sc >> (depth(sc) - depth(hole)) == hole
        const uint32_t sh = (uint32_t)__builtin_clz(hole | 1u) - (uint32_t)__builtin_clz(sc);  // (hole 0: clz 31 -> a shift that cannot match a non-zero... see below)
        const bool above = hole != 0 && sh < 32u && (sc >> sh) == hole;
        const bool create = !__ballot(above);
        const bool mine = create && (uint32_t)lane == (t & 31u);
        out_id[mine ? sc - 1 : n] = r01.w;  // (slot n: scratch)
        hole = mine ? 1u : hole;
        s = mine ? sc : s;
        Lv = mine ? __uint_as_float(ls.x) : Lv;
        Lid = mine ? ls.y : Lid;
        uint4 ch;
        ch.x = mine ? r23.x : ch_own.x;
        ch.y = mine ? r23.y : ch_own.y;
        ch.z = mine ? r23.z : ch_own.z;
        ch.w = mine ? r23.w : ch_own.w;
        walk_step4(a, hole, s, Lv, Lid, ch);
        wave_sync();
        t += create ? 1u : 0u;
        if (create) {
            const uint4 c2 = a4[hole < lim ? hole : lim];
            walk_step4(a, hole, s, Lv, Lid, c2);
            wave_sync();
        }
    }
    while (__ballot(hole != 0)) {
        const uint4 c2 = a4[hole < lim ? hole : lim];
        walk_step4(a, hole, s, Lv, Lid, c2);
        wave_sync();
    }
}

// ---- v5 (round 6): one kind of walker step without bounds tests, and the start decision taken a tick ahead.
//  * Sentinels instead of bounds: every slot outside the heap holds -inf (the slot a pop takes its entry from is overwritten with it
//    at once; slots n + 1 .. n + 3 hold it from the start), so "a single child", "no child" and "past the array" are what the value
//    compares give by themselves: a missing child loses against any entry, and an entry that meets two missing children ends its walk.
//    The children of a hole beyond n / 2 are read from the pair (n + 2, n + 3).
//  * Lanes without a walk carry +inf on hole 0: they read pair 0, never move, and write slot 0 (unused).
//  * Whether pop t + 1 may start is decided at the END of the tick before (the holes it depends on are final then), in the shadow of
//    that tick's store, instead of between the tick's load and its use.
//  * depth from the hole itself (no level register); the popped root's id is read a tick later from where the step left it.
__device__ __forceinline__ void walk_step5(uint2* a2w, const uint4* a4, uint32_t& hole, float Lv, uint32_t Lid, uint32_t P) {
    const uint32_t pr = hole < P ? hole : P;
    const uint4 ch = a4[pr];
    const float c1v = __uint_as_float(ch.x), c2v = __uint_as_float(ch.z);
    const bool left = c1v > c2v;  // the right one between equals (Heap.h:100)
    const float cv = left ? c1v : c2v;
    const uint32_t cid = left ? ch.y : ch.w;
    const bool done = Lv > cv;
    a2w[hole] = make_uint2(__float_as_uint(done ? Lv : cv), done ? Lid : cid);
    const uint32_t nh = (hole << 1) | (left ? 0u : 1u);
    hole = done ? 0u : nh;
}
__device__ inline void heapsort_v5(HeapEnt* a, uint32_t n, uint32_t* out_id, int lane) {
    const float NEG = -__builtin_inff(), POS = __builtin_inff();
    uint2* a2w = reinterpret_cast<uint2*>(a);
    const uint4* a4 = reinterpret_cast<const uint4*>(a);
    const uint32_t P = (n >> 1) + 1;
    if (lane < 3) a2w[n + 1 + lane] = make_uint2(__float_as_uint(NEG), 0xffffffffu);
    if (lane == 0) a2w[0] = make_uint2(__float_as_uint(POS), 0xffffffffu);
    wave_sync();
    uint32_t hole = 0, Lid = 0;
    float Lv = POS;
    uint32_t t = 0;
    bool create = true;  // (nothing in flight)
    while (t < n) {
        const uint32_t sc = n - t;
        if (create) {
            // ---- the tick that starts pop t: its lane takes the entry of slot sc and stands on the root
            const bool mine = (uint32_t)lane == (t & 31u);
            const uint2 ls = a2w[sc];
            const uint2 root = a2w[1];
            // (slot sc keeps its entry while its own walk is under way -- the walk may meet it as a child, Heap.h:97-107 -- and
            // the slot the pop BEFORE took its entry from leaves the heap for good now: nothing in flight can reach it any more)
            a2w[sc + 1] = make_uint2(__float_as_uint(NEG), 0xffffffffu);
            if (lane == 0) out_id[sc - 1] = root.y;
            hole = mine ? 1u : hole;
            Lv = mine ? __uint_as_float(ls.x) : Lv;
            Lid = mine ? ls.y : Lid;
            // (sc == 1: the last entry is the root itself -- its walk finds no child and puts it back; nothing reads it again)
            walk_step5(a2w, a4, hole, Lv, Lid, P);
            wave_sync();
            t++;
        }
        // ---- a plain tick, and the decision for the next one: no start while a walk stands on slot n - t or above it
        walk_step5(a2w, a4, hole, Lv, Lid, P);
        // (on or above slot n - t, whose entry the next pop takes, or slot n - t + 1, which it declares dead)
        const uint32_t scn = n - t;  // (t == n: the loop ends)
        const uint32_t ch_ = (uint32_t)__builtin_clz(hole | 1u);
        const uint32_t sh0 = ch_ - (uint32_t)__builtin_clz(scn | 1u), sh1 = ch_ - (uint32_t)__builtin_clz(scn + 1u);
        const bool above = (sh0 < 32u && (scn >> sh0) == hole) || (sh1 < 32u && ((scn + 1u) >> sh1) == hole);
        create = !__ballot(above);
        Lv = hole == 0 ? POS : Lv;
        wave_sync();
    }
    while (__ballot(hole != 0)) {
        walk_step5(a2w, a4, hole, Lv, Lid, P);
        Lv = hole == 0 ? POS : Lv;
        wave_sync();
    }
}

// ---- v6: v5 with the ticks ordered by the LDS pipeline alone: a wave's LDS instructions execute in the order they were issued, so a
// later tick's reads see an earlier tick's writes of any lane without waiting for the writes to complete -- only the compiler has to
// keep the order (no fence, no s_waitcnt behind the stores)
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }
__device__ unsigned long long g_ticks6;
__device__ __forceinline__ void walk_step6(uint2* a2w, const uint4* a4, uint32_t& hole, float Lv, uint32_t Lid, uint32_t P) {
    const uint32_t pr = hole < P ? hole : P;
    const uint4 ch = a4[pr];
    const float c1v = __uint_as_float(ch.x), c2v = __uint_as_float(ch.z);
    const bool left = c1v > c2v;  // the right one between equals (Heap.h:100)
    const float cv = left ? c1v : c2v;
    const uint32_t cid = left ? ch.y : ch.w;
    const bool done = Lv > cv;
    a2w[hole] = make_uint2(__float_as_uint(done ? Lv : cv), done ? Lid : cid);
    const uint32_t nh = (hole << 1) | (left ? 0u : 1u);
    hole = done ? 0u : nh;
}
__device__ inline void heapsort_v6(HeapEnt* a, uint32_t n, uint32_t* out_id, int lane) {
    const float NEG = -__builtin_inff(), POS = __builtin_inff();
    uint2* a2w = reinterpret_cast<uint2*>(a);
    const uint4* a4 = reinterpret_cast<const uint4*>(a);
    const uint32_t P = (n >> 1) + 1;
    if (lane < 3) a2w[n + 1 + lane] = make_uint2(__float_as_uint(NEG), 0xffffffffu);
    if (lane == 0) a2w[0] = make_uint2(__float_as_uint(POS), 0xffffffffu);
    lds_order();
    uint32_t hole = 0, Lid = 0;
    float Lv = POS;
    uint32_t t = 0, nticks = 0;
    bool create = true;  // (nothing in flight)
    while (t < n) {
        const uint32_t sc = n - t;
        if (create) {
            // ---- the tick that starts pop t: its lane takes the entry of slot sc and stands on the root
            const bool mine = (uint32_t)lane == (t & 31u);
            const uint2 ls = a2w[sc];
            const uint2 root = a2w[1];
            // (slot sc keeps its entry while its own walk is under way -- the walk may meet it as a child, Heap.h:97-107 -- and
            // the slot the pop BEFORE took its entry from leaves the heap for good now: nothing in flight can reach it any more)
            a2w[sc + 1] = make_uint2(__float_as_uint(NEG), 0xffffffffu);
            if (lane == 0) out_id[sc - 1] = root.y;
            hole = mine ? 1u : hole;
            Lv = mine ? __uint_as_float(ls.x) : Lv;
            Lid = mine ? ls.y : Lid;
            // (sc == 1: the last entry is the root itself -- its walk finds no child and puts it back; nothing reads it again)
            walk_step6(a2w, a4, hole, Lv, Lid, P);
            lds_order();
            t++;
            nticks++;
        }
        // ---- a plain tick, and the decision for the next one: no start while a walk stands on slot n - t or above it
        walk_step6(a2w, a4, hole, Lv, Lid, P);
        // (on or above slot n - t, whose entry the next pop takes, or slot n - t + 1, which it declares dead)
        const uint32_t scn = n - t;  // (t == n: the loop ends)
        const uint32_t ch_ = (uint32_t)__builtin_clz(hole | 1u);
        const uint32_t sh0 = ch_ - (uint32_t)__builtin_clz(scn | 1u), sh1 = ch_ - (uint32_t)__builtin_clz(scn + 1u);
        const bool above = (sh0 < 32u && (scn >> sh0) == hole) || (sh1 < 32u && ((scn + 1u) >> sh1) == hole);
        create = !__ballot(above);
        Lv = hole == 0 ? POS : Lv;
        lds_order();
        nticks++;
    }
    if (lane == 0 && blockIdx.x == 0) g_ticks6 = nticks;
    while (__ballot(hole != 0)) {
        walk_step6(a2w, a4, hole, Lv, Lid, P);
        Lv = hole == 0 ? POS : Lv;
        lds_order();
    }
}

template <int V> __global__ __launch_bounds__(64) void sort_rows(const float* heaps, uint32_t n, uint32_t* out, unsigned long long* cycles) {
    extern __shared__ __align__(16) unsigned char smem[];
    HeapEnt* a = reinterpret_cast<HeapEnt*>(smem);
    uint32_t* out_id = reinterpret_cast<uint32_t*>(smem + (size_t)(n + 4) * 8);
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    // the row is a valid max-heap already (built on the host); ids = original slot
    for (uint32_t i = lane; i < n + 2; i += 64) a[i] = HeapEnt{i >= 1 && i <= n ? heaps[(size_t)blockIdx.x * n + i - 1] : 3.0e38f, i - 1};
    wave_sync();
    const unsigned long long c0 = __builtin_readcyclecounter();
    if (V == 1) heapsort_v1(a, n, out_id, lane);
    else if (V == 2) heapsort_v2(a, n, out_id, lane);
    else if (V == 3) heapsort_v3(a, n, out_id, lane);
    else if (V == 4) heapsort_v4(a, n, out_id, lane);
    else if (V == 5) heapsort_v5(a, n, out_id, lane);
    else heapsort_v6(a, n, out_id, lane);
    const unsigned long long c1 = __builtin_readcyclecounter();
    wave_sync();
    for (uint32_t i = lane; i < n; i += 64) out[(size_t)blockIdx.x * n + i] = out_id[i];
    if (lane == 0) cycles[blockIdx.x] = c1 - c0;
}

// the literal heap sort (pop: Heap.h:88-118 restated)
static void ref_sort(std::vector<float> v, std::vector<uint32_t>& out) {
    const size_t n = v.size();
    std::vector<uint32_t> id(n + 2);
    std::vector<float> a(n + 2, 0.f);
    for (size_t i = 1; i <= n; i++) a[i] = v[i - 1], id[i] = (uint32_t)(i - 1);
    out.assign(n, 0);
    for (size_t t = 0; t < n; t++) {
        const size_t k = n - t;
        const uint32_t top = id[1];
        const float val = a[k];
        const uint32_t vid = id[k];
        size_t i = 1;
        while (true) {
            const size_t i1 = 2 * i, i2 = i1 + 1;
            if (i1 > k) break;
            if (i2 == k + 1 || a[i1] > a[i2]) {
                if (val > a[i1]) break;
                a[i] = a[i1]; id[i] = id[i1]; i = i1;
            } else {
                if (val > a[i2]) break;
                a[i] = a[i2]; id[i] = id[i2]; i = i2;
            }
        }
        a[i] = val; id[i] = vid;
        out[k - 1] = top;
    }
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 4096, rows = argc > 2 ? atoi(argv[2]) : 64;
    std::vector<float> h((size_t)rows * n);
    srand(7);
    for (uint32_t r = 0; r < rows; r++) {
        float* x = h.data() + (size_t)r * n;
        for (uint32_t i = 0; i < n; i++) x[i] = (float)(rand() % (r % 2 ? 3000 : 100000));  // runs of equal values
        std::make_heap(x, x + n);  // any valid max-heap will do as the start
    }
    float* dh; uint32_t* dout; unsigned long long* dcyc;
    CK(hipMalloc(&dh, h.size() * 4)); CK(hipMalloc(&dout, h.size() * 4)); CK(hipMalloc(&dcyc, rows * 8));
    CK(hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const size_t shmem = (size_t)(n + 4) * 8 + (size_t)(n + 1) * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int ver = 3; ver <= 6; ver++) {
        auto kern = ver == 3 ? sort_rows<3> : ver == 4 ? sort_rows<4> : ver == 5 ? sort_rows<5> : sort_rows<6>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            kern<<<rows, 64, shmem>>>(dh, n, dout, dcyc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        std::vector<uint32_t> out(h.size()); std::vector<unsigned long long> cyc(rows);
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(cyc.data(), dcyc, rows * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (uint32_t r = 0; r < rows; r++) {
            std::vector<uint32_t> want;
            ref_sort(std::vector<float>(h.begin() + (size_t)r * n, h.begin() + (size_t)(r + 1) * n), want);
            for (uint32_t i = 0; i < n; i++) bad += want[i] != out[(size_t)r * n + i];
        }
        double mc = 0; for (auto c : cyc) mc += c; mc /= rows;
        if (ver == 6) { unsigned long long tk = 0; CK(hipMemcpyFromSymbol(&tk, HIP_SYMBOL(g_ticks6), 8)); printf("   v6 row 0: %llu ticks for %u pops = %.2f ticks per pop\n", tk, n, (double)tk / n); }
        printf("v%d: n %u, %u rows: %.3f ms per launch, %.0f cycle-counter ticks per row (%.1f per pop), wrong entries %zu\n", ver, n, rows, best, mc, mc / n, bad);
    }
    return 0;
}
