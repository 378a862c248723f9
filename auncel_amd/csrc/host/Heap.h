// Binary-heap helpers with the reference's names and exact index arithmetic (Auncel/Heap.h:88-322).
// Callers of the InvertedListScanner API own a raw heap (tests/test_lowlevel_ivf.cpp:150-175) and use
// these to initialise and finish it; the scan itself runs on the GPU.
#pragma once
#include <cfloat>
#include <cstddef>
#include <cstring>

namespace faiss {

template <bool KeepSmallest> struct HeapOrder {
    // KeepSmallest: max-heap (CMax), the worst = largest value sits on top
    static bool cmp(float a, float b) { return KeepSmallest ? a > b : a < b; }
    static float neutral() { return KeepSmallest ? FLT_MAX : -FLT_MAX; }
};

template <bool S> inline void heap_pop_t(size_t k, float* v, long* id) {
    v--;
    id--;
    float last = v[k];
    size_t i = 1;
    for (;;) {
        size_t a = i << 1, b = a + 1;
        if (a > k) break;
        size_t c = (b == k + 1 || HeapOrder<S>::cmp(v[a], v[b])) ? a : b;
        if (HeapOrder<S>::cmp(last, v[c])) break;
        v[i] = v[c];
        id[i] = id[c];
        i = c;
    }
    v[i] = v[k];
    id[i] = id[k];
}

template <bool S> inline void heap_push_t(size_t k, float* v, long* id, float val, long label) {
    v--;
    id--;
    size_t i = k;
    while (i > 1) {
        size_t f = i >> 1;
        if (!HeapOrder<S>::cmp(val, v[f])) break;
        v[i] = v[f];
        id[i] = id[f];
        i = f;
    }
    v[i] = val;
    id[i] = label;
}

template <bool S> inline void heap_heapify_t(size_t k, float* v, long* id) {
    for (size_t i = 0; i < k; i++) {
        v[i] = HeapOrder<S>::neutral();
        id[i] = -1;
    }
}

template <bool S> inline size_t heap_reorder_t(size_t k, float* v, long* id) {
    size_t kept = 0;
    for (size_t i = 0; i < k; i++) {
        float val = v[0];
        long label = id[0];
        heap_pop_t<S>(k - i, v, id);
        v[k - kept - 1] = val;
        id[k - kept - 1] = label;
        if (label != -1) kept++;
    }
    size_t n = kept;
    memmove(v, v + k - kept, kept * sizeof(*v));
    memmove(id, id + k - kept, kept * sizeof(*id));
    for (; kept < k; kept++) {
        v[kept] = HeapOrder<S>::neutral();
        id[kept] = -1;
    }
    return n;
}

/// n candidates against the heap, each admitted on strict improvement over the top (Heap.h:246-264); null labels: the position
template <bool S> inline void heap_addn_t(size_t k, float* v, long* id, const float* x, const long* labels, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (!HeapOrder<S>::cmp(v[0], x[i])) continue;
        heap_pop_t<S>(k, v, id);
        heap_push_t<S>(k, v, id, x[i], labels ? labels[i] : (long)i);
    }
}
inline void maxheap_addn(size_t k, float* v, long* id, const float* x, const long* labels, size_t n) { heap_addn_t<true>(k, v, id, x, labels, n); }
inline void minheap_addn(size_t k, float* v, long* id, const float* x, const long* labels, size_t n) { heap_addn_t<false>(k, v, id, x, labels, n); }

inline void maxheap_heapify(size_t k, float* v, long* id) { heap_heapify_t<true>(k, v, id); }
inline void minheap_heapify(size_t k, float* v, long* id) { heap_heapify_t<false>(k, v, id); }
inline size_t maxheap_reorder(size_t k, float* v, long* id) { return heap_reorder_t<true>(k, v, id); }
inline size_t minheap_reorder(size_t k, float* v, long* id) { return heap_reorder_t<false>(k, v, id); }
inline void maxheap_pop(size_t k, float* v, long* id) { heap_pop_t<true>(k, v, id); }
inline void minheap_pop(size_t k, float* v, long* id) { heap_pop_t<false>(k, v, id); }
inline void maxheap_push(size_t k, float* v, long* id, float val, long label) { heap_push_t<true>(k, v, id, val, label); }
inline void minheap_push(size_t k, float* v, long* id, float val, long label) { heap_push_t<false>(k, v, id, val, label); }

}  // namespace faiss
