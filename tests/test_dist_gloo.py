"""
World-size-2 test of the document-sharded path (dist.ShardedLDA) on CPU with the gloo backend.

The product engine is the HIP library; here an oracle-backed engine with the same operator interface
is plugged in (tests may use the oracle) so that the host logic -- nnz-balanced shards, ONE all-reduce of
the packed [S | Elogtheta_sum] statistics per outer iteration, identical M-step on every rank, ELBO
all-reduce and the signed stop rule -- is exercised without a GPU and compared with the
single-process oracle.
"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleLDAEngine:
    """dist.ShardedLDA engine interface on top of oracle.LDA (test infrastructure only)."""

    def __init__(self, oc, shard, K, beta0, M_total):
        self.m = oc.LDA(oc.CSR(shard.doc_ptr, shard.terms, shard.counts, shard.V), K, beta0)
        self.K, self.V, self.M_total = K, shard.V, M_total
        self.stats = torch.zeros(K * shard.V + K, dtype=torch.float64)
        self.model = self.m

    def estep(self, viter, vtol): self.m.estep(viter, vtol)

    def reduce_docs(self):
        self.stats[:self.K * self.V] = torch.from_numpy(self.m.beta_temp.ravel(order="F").copy())
        self.stats[self.K * self.V:] = torch.from_numpy(self.m.Elogtheta.sum(axis=1))

    def stats_tensor(self): return self.stats

    def update_beta(self):
        self.m.beta_temp[:] = self.stats[:self.K * self.V].numpy().reshape((self.K, self.V), order="F")
        self.m.update_beta()

    def update_alpha(self, niter, ntol):
        self.m.update_alpha(niter, ntol, Elogtheta_sum=self.stats[self.K * self.V:].numpy().copy(), Mtot=self.M_total)

    def local_elbo(self):
        # Elogptheta's constant is per document, so the shards' sums add up exactly
        return self.m.update_elbo(store=False)


def _worker(rank, world, initfile, out_dir):
    sys.path.insert(0, ROOT)
    import tmvb_amd
    from oracle import oracle as oc
    from tmvb_amd_pkg.dist import ShardedLDA
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    tm = tmvb_amd.pkg
    corpus = tm.syn_nsf(M=240, V=400, seed=9)
    K = 6
    beta0 = tm.dirichlet_rows(K, corpus.V, seed=2)
    d0, d1 = corpus.shard_bounds(world)[rank]
    eng = OracleLDAEngine(oc, corpus.shard(d0, d1), K, beta0, corpus.M)
    tr = ShardedLDA(eng)
    traj = tr.train(iter=6, tol=0.0, checkelbo=1, K=K)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), traj=np.array(traj), alpha=eng.m.alpha, beta=eng.m.beta,
             gamma=eng.m.gamma, d0=d0, d1=d1)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_lda_world2_matches_single_process(oracle, tmvb):
    world = 2
    with tempfile.TemporaryDirectory() as td:
        initfile = os.path.join(td, "init")
        mp.spawn(_worker, args=(world, initfile, td), nprocs=world, join=True)
        res = [dict(np.load(os.path.join(td, f"rank{r}.npz"))) for r in range(world)]
    corpus = tmvb.syn_nsf(M=240, V=400, seed=9)
    K = 6
    beta0 = tmvb.dirichlet_rows(K, corpus.V, seed=2)
    ref = oracle.LDA(oracle.CSR(corpus.doc_ptr, corpus.terms, corpus.counts, corpus.V), K, beta0)
    traj = ref.train(iter=6, tol=0.0, checkelbo=1)
    for r in res:
        np.testing.assert_allclose(r["traj"], traj, rtol=1e-12)
        np.testing.assert_allclose(r["alpha"], ref.alpha, rtol=1e-10)
        np.testing.assert_allclose(r["beta"], ref.beta, rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(r["gamma"], ref.gamma[:, int(r["d0"]):int(r["d1"])], rtol=1e-10)
    # identical M-step on every rank (no broadcast needed)
    assert np.array_equal(res[0]["alpha"], res[1]["alpha"]) and np.array_equal(res[0]["beta"], res[1]["beta"])
    assert int(res[0]["d1"]) == int(res[1]["d0"]) and int(res[1]["d1"]) == corpus.M


def test_sharded_train_argument_errors(tmvb):
    sys.path.insert(0, ROOT)
    from tmvb_amd_pkg.dist import ShardedLDA
    tr = ShardedLDA(engine=None)
    with pytest.raises(ValueError):
        tr.train(iter=-1, K=4)
    with pytest.raises(ValueError):
        tr.train(tol=-1.0, K=4)
    with pytest.raises(ValueError):
        tr.train(checkelbo=0, K=4)
