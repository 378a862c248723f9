"""The two reformulations behind heap_tie_order_kernel's fast path (Floyd-style level-parallel filling, pipelined heap
sort) against the literal heap and against the pinned oracle's coarse ranking, on rows full of equal distances.  CPU only:
this is the model the kernel was written from (tests/heap_tie_model.py); the kernel itself is held to the oracle by
tests/test_gpu_coarse_ties.py."""
import numpy as np
import pytest

import heap_tie_model as M


@pytest.mark.parametrize("m", [3, 4, 6, 8, 10])
@pytest.mark.parametrize("span", [3, 10, 1000, 10 ** 6])
def test_floyd_build_and_pipelined_sort_equal_the_literal_heap(m, span):
    rs = np.random.RandomState(100 * m + span % 97)
    n = 1 << m
    x = [float(a) for a in rs.randint(0, span, n)]
    (hv, hi), order = M.ref_rank(x)
    fv, fi = M.fast_build(x)
    assert hv[1:n + 1] == fv[1:n + 1] and hi[1:n + 1] == fi[1:n + 1]  # same heap after the filling, entry for entry
    po, ticks, stalls = M.pipelined_sort(hv, hi, n)
    assert po == order
    assert ticks <= 3 * n + 32  # two ticks per pop plus the stalls behind walks above the slot being taken


@pytest.mark.parametrize("seed", range(4))
def test_literal_heap_is_the_oracles_ranking(oracle, seed):
    rs = np.random.RandomState(seed)
    n, d = 256, 8
    cen = rs.randint(0, 6, size=(n, d)).astype(np.float32)
    xq = rs.randint(0, 6, size=(3, d)).astype(np.float32)
    D, I = oracle.knn(1, xq, cen, n)
    for q in range(3):
        x = [float(v) for v in ((xq[q][None, :] - cen) ** 2).sum(1)]  # exact on this grid
        _, order = M.ref_rank(x)
        assert order == [int(i) for i in I[q]]
        assert len(set(x)) < n  # the row does hold runs of equal distances
