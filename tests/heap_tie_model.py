"""Model of heap_tie_order_kernel's fast path (auncel_amd/csrc/ivf_kernels.hip): the reference's coarse ranking with
nprobe == nlist == 2^m -- a binary heap filled one centroid at a time, then heap-sorted (Auncel/utils.cpp:454-490,
Heap.h:88-142,295-322) -- restated as (1) a level-parallel Floyd-style build with the reference's slot assignment plus the
last m steps one by one, and (2) a heap sort whose pops are pipelined two levels apart.  ref_rank is the literal heap;
tests/test_heap_tie_model.py holds the three to each other and to the oracle's knn on tie-heavy rows."""
import numpy as np, sys

M = float('inf')
def gt(a, b): return a > b   # CMax::cmp

def ref_rank(x):
    n = len(x); k = n
    v = [None] + [M] * k + [M]; ids = [None] + [-1] * k + [-1]
    def pop(k):
        val, vid = v[k], ids[k]; i = 1
        while True:
            i1, i2 = 2 * i, 2 * i + 1
            if i1 > k: break
            if i2 == k + 1 or gt(v[i1], v[i2]):
                if gt(val, v[i1]): break
                v[i], ids[i] = v[i1], ids[i1]; i = i1
            else:
                if gt(val, v[i2]): break
                v[i], ids[i] = v[i2], ids[i2]; i = i2
        v[i], ids[i] = val, vid
    def push(k, val, vid):
        i = k
        while i > 1:
            f = i >> 1
            if not gt(val, v[f]): break
            v[i], ids[i] = v[f], ids[f]; i = f
        v[i], ids[i] = val, vid
    for j in range(n):
        if x[j] < v[1]:
            pop(k); push(k, x[j], j)
    heap_after_build = (list(v), list(ids))
    out = [None] * (k + 1)
    for t in range(k):
        top = (v[1], ids[1]); pop(k - t); out[k - t] = top
    return heap_after_build, [o[1] for o in out[1:]]

def fast_build(x):
    n = len(x); m = n.bit_length() - 1
    assert 1 << m == n
    v = [None] + [M] * n + [M]; ids = [None] + [-1] * n + [-1]
    F = n >> 1
    spine = set()
    a = F
    while a >= 1: spine.add(a); a >>= 1
    def S(t): return (1 << (m - t)) - 1
    def jof(p):
        dp = p.bit_length() - 1
        s = S(dp)
        for t in range(1, dp + 1):
            bit = (p >> (dp - t)) & 1
            if bit == 0: s += S(t)
        return s
    def sift(p, val, vid, k):
        i = p
        while True:
            i1, i2 = 2 * i, 2 * i + 1
            if i1 > k: break
            if i2 == k + 1 or gt(v[i1], v[i2]):
                if gt(val, v[i1]): break
                v[i], ids[i] = v[i1], ids[i1]; i = i1
            else:
                if gt(val, v[i2]): break
                v[i], ids[i] = v[i2], ids[i2]; i = i2
        v[i], ids[i] = val, vid
    # levels bottom-up, every non-spine node (order inside a level is free)
    for dp in range(m - 1, 0, -1):
        for p in range(1 << dp, 1 << (dp + 1)):
            if p in spine: continue
            j = jof(p)
            sift(p, x[j - 1], j - 1, n - 1)   # subtree never contains slot n
    # slot n holds x_{n-m-1} when the tail starts (pushed at step n-m-1)
    v[n], ids[n] = x[n - m - 1], n - m - 1
    def push(k, val, vid):
        i = k
        while i > 1:
            f = i >> 1
            if not gt(val, v[f]): break
            v[i], ids[i] = v[f], ids[f]; i = f
        v[i], ids[i] = val, vid
    A = F
    for j in range(n - m, n):
        # pop: the sentinel path ends at spine node A; val = a[n] sifts down from there (slot n still holds val)
        sift(A, v[n], ids[n], n)
        push(n, x[j], j)
        A >>= 1
    return v, ids

def pipelined_sort(v, ids, n):
    """tick-level model: token = (L, Lid, hole, s); created every >= 2 ticks, stalls while an in-flight hole is an ancestor of slot s"""
    v = list(v); ids = list(ids)
    depth = lambda p: p.bit_length() - 1
    out = [None] * (n + 1)
    tokens = []   # list of dicts, oldest first
    t = 0; tick = 0; last_create = -2; stalls = 0
    while t < n or tokens:
        # advance tokens (all in lockstep, reads before writes: emulate by computing from a snapshot)
        snap_v, snap_i = list(v), list(ids)
        newtok = []
        for tk in tokens:
            i, s = tk['hole'], tk['s']
            i1, i2 = 2 * i, 2 * i + 1
            if i1 > s:
                v[i], ids[i] = tk['L'], tk['Lid']; continue
            if i2 == s + 1 or gt(snap_v[i1], snap_v[i2]): c = i1
            else: c = i2
            if gt(tk['L'], snap_v[c]):
                v[i], ids[i] = tk['L'], tk['Lid']; continue
            v[i], ids[i] = snap_v[c], snap_i[c]
            tk['hole'] = c; newtok.append(tk)
        tokens = newtok
        # create
        if t < n and tick - last_create >= 2:
            s = n - t
            conflict = any((s >> (depth(s) - depth(tk['hole']))) == tk['hole'] for tk in tokens if depth(tk['hole']) <= depth(s))
            if conflict: stalls += 1
            else:
                out[s] = ids[1]
                tokens.append(dict(L=v[s], Lid=ids[s], hole=1, s=s)); last_create = tick; t += 1
        tick += 1
    return out[1:], tick, stalls

if __name__ == '__main__':
    rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    for trial in range(40):
        m = int(rs.choice([3, 4, 5, 6, 8, 10]))
        n = 1 << m
        span = int(rs.choice([3, 10, 1000, 10**6]))
        x = [float(a) for a in rs.randint(0, span, n)]
        (hv, hi), order = ref_rank(x)
        fv, fi = fast_build(x)
        ok_build = hv[1:n + 1] == fv[1:n + 1] and hi[1:n + 1] == fi[1:n + 1]
        po, ticks, stalls = pipelined_sort(hv, hi, n)
        ok_sort = po == order
        print(n, span, 'build', ok_build, 'sort', ok_sort, 'ticks/pop', round(ticks / n, 2), 'stalls', stalls)
        assert ok_build and ok_sort
