"""
model.topics (src/gpuLDA.jl:374: [reverse(sortperm(vec(beta[i,:]))) for i in 1:K]) from tmvb_topic_order, the device's segmented sort,
against the host formula it replaces -- bit-exact index lists, ties included (sortperm is stable; so is the radix sort).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def host_topics(B):
    return [np.argsort(B[i, :], kind="stable")[::-1] + 1 for i in range(B.shape[0])]


@pytest.mark.parametrize("K,V", [(1, 1), (1, 7), (3, 2), (7, 1000), (50, 25319), (124, 333)])
def test_topic_order_equals_the_host_sort(tmvb, K, V):
    from tmvb_amd_pkg.lda import _topic_orders
    rng = np.random.default_rng(100 * K + V)
    B = np.asfortranarray(rng.dirichlet(np.full(V, 0.1), size=K)) if V > 1 else np.ones((K, 1), order="F")
    if V >= 1000:                                            # ties: equal columns, runs of zeros, denormals next to zeros
        B[:, 100:200] = B[:, 300:400]
        B[:, 500:520] = 0.0
        B[:, 520:530] = 5e-324
    ctx = tmvb.DeviceContext(0)
    got = _topic_orders(ctx, B)
    want = host_topics(B)
    assert len(got) == K
    for i in range(K):
        assert got[i].dtype == np.int64 and got[i].shape == (V,)
        assert np.array_equal(got[i], want[i]), (K, V, i, np.flatnonzero(got[i] != want[i])[:5])


def test_train_leaves_the_reference_topics(tmvb):
    g = tmvb.gpuLDA(tmvb.syn_nsf(M=150, V=400), 5)
    g.train(iter=3, checkelbo=np.inf, printelbo=False)
    assert all(np.array_equal(a, b) for a, b in zip(g.topics, host_topics(np.asarray(g.beta))))
    g.close()
    g = tmvb.gpuCTPF(tmvb.syn_citeu(M=120, V=300, U=40), 4)
    g.train(iter=3, checkelbo=np.inf, printelbo=False, recs=False)
    assert all(np.array_equal(a, b) for a, b in zip(g.topics, host_topics(np.asarray(g.alef / g.bet[:, None]))))      # src/gpuCTPF.jl:707-708
    g.close()


def test_argument_errors(tmvb):
    import ctypes as C
    ctx = tmvb.DeviceContext(0)
    B = np.ones((2, 3), order="F")
    out = np.empty((2, 3), dtype=np.int32)
    pd, pi = B.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_int32))
    L = tmvb.lib()
    assert L.tmvb_topic_order(ctx.handle, pd, C.c_int32(0), C.c_int64(3), pi) == 1            # TMVB_EINVAL
    assert L.tmvb_topic_order(ctx.handle, None, C.c_int32(2), C.c_int64(3), pi) == 1
    assert L.tmvb_topic_order(ctx.handle, pd, C.c_int32(2), C.c_int64(0), pi) == 0            # nothing to order
