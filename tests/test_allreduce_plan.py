"""
tmvb_allreduce_plan (include/tmvb.h): the slab plan of the sliced statistics all-reduce is host arithmetic on the GLOBAL postings per
term -- every rank must derive the same slabs in the same order -- so it runs here without a GPU, through the C ABI.
"""
import ctypes as C

import numpy as np
import pytest


def _plan(tm, counts, want):
    L = tm._lib.lib()
    counts = np.ascontiguousarray(counts, dtype=np.float64)
    cuts = np.zeros(want + 1, dtype=np.int64); order = np.zeros(want, dtype=np.int32); s = C.c_int32(0)
    rc = L.tmvb_allreduce_plan(counts.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(len(counts)), C.c_int32(want),
                               cuts.ctypes.data_as(C.POINTER(C.c_int64)), order.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(s))
    tm._lib.check(rc)
    S = s.value
    return cuts[:S + 1].copy(), order[:S].copy()


@pytest.mark.parametrize("want", [1, 2, 4, 7])
def test_cuts_partition_the_vocabulary_in_equal_shares_of_pass_plus_wire(tmvb, want):
    rng = np.random.default_rng(5)
    V = 3000
    counts = rng.zipf(1.3, V).clip(max=5000).astype(np.float64)
    rng.shuffle(counts)
    cuts, order = _plan(tmvb, counts, want)
    assert cuts[0] == 0 and cuts[-1] == V and np.all(np.diff(cuts) >= 0) and len(cuts) == want + 1
    assert sorted(order.tolist()) == list(range(want))
    w = 0.5 * counts / counts.sum() + 0.5 / V
    share = np.add.reduceat(w, cuts[:-1])
    assert np.all(np.abs(share - 1.0 / want) <= w.max() + 1e-12)          # a slab ends at the first id that completes its share


def test_frequency_sorted_vocabulary_sends_its_light_tail_first(tmvb):
    V = 2000
    counts = 1e6 / (1.0 + np.arange(V)) ** 1.2                             # heavy ids in front
    cuts, order = _plan(tmvb, counts, 4)
    n = counts.sum()
    passes = np.add.reduceat(counts, cuts[:-1]) / n
    wires = np.diff(cuts) / V
    # Johnson's rule: first the slabs with less pass than wire, cheapest pass first; then the others, most wire first
    first = [s for s in order if passes[s] < wires[s]]
    second = [s for s in order if passes[s] >= wires[s]]
    assert order.tolist() == first + second
    assert all(passes[a] <= passes[b] for a, b in zip(first, first[1:]))
    assert all(wires[a] >= wires[b] for a, b in zip(second, second[1:]))
    assert order[0] == 3 and order[-1] == 0                                  # the long light tail first, the heavy head last


def test_every_rank_gets_the_same_plan_and_degenerate_inputs(tmvb):
    counts = np.array([3.0, 0.0, 7.0, 1.0, 0.0, 2.0])
    a = _plan(tmvb, counts, 4); b = _plan(tmvb, counts.copy(), 4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    cuts, order = _plan(tmvb, np.array([1.0, 2.0, 3.0]), 8)                 # more slabs than terms
    assert len(order) == 3 and cuts[0] == 0 and cuts[-1] == 3 and np.all(np.diff(cuts) >= 0)   # (a slab may be empty: [0, 2, 3, 3] here)
    cuts, order = _plan(tmvb, np.zeros(5), 4)                              # an all-empty corpus: one slab
    assert cuts.tolist() == [0, 5] and order.tolist() == [0]
    with pytest.raises(ValueError):
        _plan(tmvb, np.array([1.0, -1.0]), 2)
