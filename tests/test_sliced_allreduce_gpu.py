"""
tmvb_lda_estep_allreduce (include/tmvb.h): the sharded E-step with its collective issued in vocabulary slabs under the last
statistics pass (csrc/tmvb_lda.hip, lda_ar_prepare / tmvb_lda_estep_allreduce).  A statistics entry is the fixed-order sum of its
term's chunks whatever the order the chunks are launched in, and an all-reduce sums element by element whatever the slabs, so the
fused call must leave EXACTLY the state of estep + reduce_docs + allreduce:

  * RCCL with nranks = 1 (the call sequence and the event plumbing of the side stream; RCCL short-cuts a one-rank in-place
    all-reduce -- 0.7 us per call, tools/ar_slices_probe.py -- so no collective kernel runs; one-, two- and three-piece plans);
  * two ranks on one GPU through the host transport with gloo carrying the sums: TMVB_AR_SLICES = 4 and 7 against 1 (the default), bit for bit,
    and both against the whole corpus on one context.
"""
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _lda(tm, corpus, K, ctx=None):
    gm = tm.gpuLDA(corpus, K, ctx=ctx)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, corpus.V, seed=3)); gm.beta_old = gm.beta.copy(order="F")
    gm.update_buffer()
    return gm


@pytest.mark.parametrize("pieces,K,slices", [(1, 50, 4), (3, 50, 4), (1, 100, 3), (2, 20, 4), (1, 50, 1), (3, 50, 1)])
def test_fused_call_equals_the_three_calls_single_rank(tmvb, monkeypatch, pieces, K, slices):
    monkeypatch.setenv("TMVB_LDA_PIECES", str(pieces))
    monkeypatch.setenv("TMVB_AR_SLICES", str(slices))          # read when the plan is agreed on, at the first fused call
    corpus = tmvb.syn_nsf(M=4000, V=2500, seed=31)
    a, b = _lda(tmvb, corpus, K), _lda(tmvb, corpus, K)
    ca = tmvb.Communicator.rccl(a.ctx, tmvb.Communicator.unique_id(), 1, 0)
    cb = tmvb.Communicator.rccl(b.ctx, tmvb.Communicator.unique_id(), 1, 0)
    a.set_comm(ca, corpus.M); b.set_comm(cb, corpus.M)
    for _ in range(4):
        a.estep(); a.reduce_docs(); ptr, n = a.stats(); ca.allreduce(ptr, n); a.update_beta(); a.update_alpha()
        b.estep_allreduce(); b.update_beta(); b.update_alpha()
    ea, eb = a.update_elbo(), b.update_elbo()
    a.update_host(); b.update_host()
    assert np.array_equal(a.beta, b.beta) and np.array_equal(a.alpha, b.alpha) and np.array_equal(a.gamma, b.gamma)
    assert ea == eb
    # and the stepwise calls still work on the handle afterwards (the sliced index serves an unsliced pass too)
    b.estep(); b.reduce_docs(); ptr, n = b.stats(); cb.allreduce(ptr, n); b.update_beta(); b.update_alpha()
    a.estep_allreduce(); a.update_beta(); a.update_alpha()
    a.update_host(); b.update_host()
    assert np.array_equal(a.beta, b.beta) and np.array_equal(a.gamma, b.gamma)
    a.set_comm(None, corpus.M); b.set_comm(None, corpus.M)
    a.close(); b.close(); ca.close(); cb.close()


def test_without_a_communicator_it_is_an_error(tmvb):
    corpus = tmvb.syn_nsf(M=300, V=200, seed=2)
    gm = _lda(tmvb, corpus, 50)
    with pytest.raises(ValueError, match="no communicator"):
        gm.estep_allreduce()
    gm.close()


def _gloo_sum(dist):
    import torch

    def fn(a):
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return fn


def _worker(rank, world, initfile, out_dir, slices, M, V, K, iters):
    if slices == 0:                                        # the three-call form inside train! (TMVB_FUSED_ALLREDUCE=0)
        os.environ["TMVB_FUSED_ALLREDUCE"] = "0"; slices = 1
    else:
        os.environ["TMVB_FUSED_ALLREDUCE"] = "1"          # opt-in since round 5 (the default is the single collective)
    os.environ["TMVB_AR_SLICES"] = str(slices)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import tmvb_amd
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    tm = tmvb_amd.pkg
    ctx = tm.DeviceContext(0)
    comm = tm.Communicator.host(ctx, world, rank, _gloo_sum(dist))
    corpus = tm.syn_nsf(M=M, V=V, seed=11)
    d0, d1 = corpus.shard_bounds(world)[rank]
    if M <= 8 and rank == 1:
        d0 = d1                                            # an EMPTY shard still takes part in every collective
    elif M <= 8:
        d1 = corpus.M
    gm = tm.gpuLDA(corpus.shard(d0, d1), K, ctx=ctx)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, corpus.V, seed=3)); gm.beta_old = gm.beta.copy(order="F")
    gm.update_buffer()
    gm.set_comm(comm, corpus.M)
    traj = gm.train(iter=iters, tol=0.0, checkelbo=1, printelbo=False)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), traj=np.array(traj), alpha=gm.alpha, beta=gm.beta, gamma=gm.gamma, d0=d0, d1=d1)
    gm.close(); comm.close()
    dist.barrier()
    dist.destroy_process_group()


def _run(slices, M=5000, V=3000, K=50, iters=4):
    import torch.multiprocessing as mp
    world = 2
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_worker, args=(world, os.path.join(td, "init"), td, slices, M, V, K, iters), nprocs=world, join=True)
        return [dict(np.load(os.path.join(td, f"rank{r}.npz"))) for r in range(world)]


def test_world2_sliced_equals_one_collective_bit_for_bit(tmvb):
    one, four, seven = _run(1), _run(4), _run(7)
    for res in (four, seven):
        for r in range(2):
            for k in ("traj", "alpha", "beta", "gamma"):
                assert np.array_equal(one[r][k], res[r][k]), (r, k)
    assert np.array_equal(four[0]["beta"], four[1]["beta"]) and np.array_equal(four[0]["alpha"], four[1]["alpha"])
    corpus = tmvb.syn_nsf(M=5000, V=3000, seed=11)
    gm = _lda(tmvb, corpus, 50)
    traj = gm.train(iter=4, tol=0.0, checkelbo=1, printelbo=False)
    np.testing.assert_allclose(four[0]["traj"], np.array(traj), rtol=2e-6)


def test_world2_three_call_escape_hatch_equals_the_fused_train(tmvb):
    """TMVB_FUSED_ALLREDUCE=0 makes the sharded train! run estep + reduce_docs + ONE all-reduce on the context's stream (rounds 2-3)."""
    fused, three = _run(1, iters=3), _run(0, iters=3)
    for r in range(2):
        for k in ("traj", "alpha", "beta", "gamma"):
            assert np.array_equal(fused[r][k], three[r][k]), (r, k)


def test_world2_more_slices_than_terms_and_an_empty_shard(tmvb):
    """V = 3 < 4 slices (the plan shrinks to V slabs), and rank 1 holds no document at all."""
    res = _run(4, M=6, V=3, K=5, iters=3)
    assert np.array_equal(res[0]["beta"], res[1]["beta"]) and np.array_equal(res[0]["traj"], res[1]["traj"])
    assert np.all(np.isfinite(res[0]["beta"])) and int(res[1]["d0"]) == int(res[1]["d1"])
