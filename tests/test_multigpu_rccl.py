"""
REAL multi-rank RCCL (ADVICE r2 / VERDICT r2 item 3a).  These tests need at least two visible MI355X and skip otherwise
(the round-end test box has one; an 8-GPU node runs them).  What they cover that tests/test_comm_gpu.py cannot on one GPU:

  * ncclCommInitRank across distinct devices, one process per GPU (mp.spawn; the unique id travels through the
    torch.distributed store exactly as bench.py --gpus N does it), the sharded tmvb_*_train loop on every rank;
  * ncclCommInitAll over n devices in ONE process and the grouped all-reduce of tmvb_*_train_group (ncclGroupStart/End
    around n collectives issued by one host thread -- the only place a missing group can deadlock);
  * after training: bit-identical globals on every rank (the replicated deterministic M-step), the same ELBO trajectory
    and stop decision on every rank, agreement with the single-context run of the same corpus.
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _ndev():
    sys.path.insert(0, ROOT)
    import tmvb_amd
    try:
        return int(tmvb_amd.pkg.lib().tmvb_device_count())
    except Exception:
        return 0


needs2 = pytest.mark.skipif(_ndev() < 2, reason="needs >= 2 visible GPUs (real multi-rank RCCL)")


def _world():
    return min(_ndev(), 4)


def _init_lda(tm, gm, K, V):
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, V, seed=3)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()


def _init_ctpf(tm, gm, K, V):
    gm.alef = np.asfortranarray(np.exp(tm.dirichlet_rows(K, V, seed=4) - 0.5)); gm.alef_old = gm.alef.copy(order="F"); gm.update_buffer()


def _corpus(tm, model):
    if model == "ctpf":
        return tm.syn_citeu(M=2400, V=900, U=200, seed=13)
    return tm.syn_nsf(M=6000 if model == "lda" else 2400, V=3000 if model == "lda" else 800, seed=11)


K_OF = {"lda": 50, "ctm": 12, "ctpf": 20}
ITERS = 4


def _worker(rank, world, initfile, out_dir, model, fused=None):
    """one process per GPU: device `rank`, RCCL communicator from a unique id published through the torch store"""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if fused is not None:
        os.environ["TMVB_FUSED_ALLREDUCE"] = fused            # read once per process by the library's train! (csrc/tmvb_train.h)
    import torch.distributed as dist
    import tmvb_amd
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    tm = tmvb_amd.pkg
    K = K_OF[model]
    ctx = tm.DeviceContext(rank)
    comm = tm.Communicator.torch_bootstrap(ctx)
    assert comm.info() == {"nranks": world, "rank": rank, "backend": "rccl"}
    corpus = _corpus(tm, model)
    d0, d1 = corpus.shard_bounds(world)[rank]
    sh = corpus.shard(d0, d1)
    if model == "lda":
        gm = tm.gpuLDA(sh, K, ctx=ctx); _init_lda(tm, gm, K, corpus.V); gm.set_comm(comm, corpus.M)
        traj = gm.train(iter=ITERS, tol=0.0, checkelbo=1, printelbo=False)
        out = dict(alpha=gm.alpha, beta=gm.beta, local=gm.gamma)
    elif model == "ctm":
        gm = tm.gpuCTM(sh, K, ctx=ctx); _init_lda(tm, gm, K, corpus.V); gm.set_comm(comm, corpus.M)
        traj = gm.train(iter=ITERS, tol=0.0, checkelbo=1, printelbo=False)
        out = dict(mu=gm.mu, sigma=gm.sigma, beta=gm.beta, local=gm.lam)
    else:
        gm = tm.gpuCTPF(sh, K, ctx=ctx); _init_ctpf(tm, gm, K, corpus.V); gm.set_comm(comm)
        traj = gm.train(iter=ITERS, tol=0.0, checkelbo=1, printelbo=False, recs=False)
        out = dict(alef=gm.alef, he=gm.he, bet=gm.bet, vav=gm.vav, dalet=gm.dalet, het=gm.het, local=gm.gimel)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), traj=np.array(traj), d0=d0, d1=d1, base=gm.elbo_baseline, **out)
    if model == "ctpf":
        gm.set_comm(None)
    else:
        gm.set_comm(None, sh.M)
    gm.close(); comm.close()
    dist.barrier()
    dist.destroy_process_group()


def _single(tm, model):
    K = K_OF[model]
    corpus = _corpus(tm, model)
    if model == "lda":
        gm = tm.gpuLDA(corpus, K); _init_lda(tm, gm, K, corpus.V)
        traj = gm.train(iter=ITERS, tol=0.0, checkelbo=1, printelbo=False)
    elif model == "ctm":
        gm = tm.gpuCTM(corpus, K); _init_lda(tm, gm, K, corpus.V)
        traj = gm.train(iter=ITERS, tol=0.0, checkelbo=1, printelbo=False)
    else:
        gm = tm.gpuCTPF(corpus, K); _init_ctpf(tm, gm, K, corpus.V)
        traj = gm.train(iter=ITERS, tol=0.0, checkelbo=1, printelbo=False, recs=False)
    return corpus, gm, np.asarray(traj)


GLOBALS = {"lda": ("alpha", "beta"), "ctm": ("mu", "sigma", "beta"), "ctpf": ("alef", "he", "bet", "vav", "dalet", "het")}
LOCAL = {"lda": "gamma", "ctm": "lam", "ctpf": "gimel"}


def _check(tm, model, res):
    corpus, ref, traj = _single(tm, model)
    for r in res[1:]:
        for n in GLOBALS[model]:
            assert np.array_equal(res[0][n], r[n]), f"{n} differs between ranks"
        assert np.array_equal(res[0]["traj"], r["traj"])
    assert int(res[-1]["d1"]) == corpus.M and all(int(res[i]["d1"]) == int(res[i + 1]["d0"]) for i in range(len(res) - 1))
    for r in res:
        assert len(r["traj"]) == ITERS
        np.testing.assert_allclose(r["traj"], traj, rtol=5e-6)
        np.testing.assert_allclose(float(r["base"]), ref.elbo_baseline, rtol=1e-9)
        for n in GLOBALS[model]:
            a, b = np.asarray(r[n]), np.asarray(getattr(ref, n))
            if n == "beta":
                big = b > 1e-6
                np.testing.assert_allclose(a[big], b[big], rtol=3e-3)
            else:
                np.testing.assert_allclose(a, b, rtol=3e-3, atol=1e-3 * max(1.0, float(np.abs(b).max())) if n in ("mu", "sigma") else 0.0)
        loc = np.asarray(getattr(ref, LOCAL[model]))[:, int(r["d0"]):int(r["d1"])]
        assert np.quantile(np.abs(r["local"] - loc) / np.maximum(np.abs(loc), 1e-2), 0.999) < 1e-2


@needs2
@pytest.mark.parametrize("model", ["lda", "ctm", "ctpf"])
def test_rccl_init_rank_one_process_per_gpu(tmvb, model):
    import torch.multiprocessing as mp
    world = _world()
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_worker, args=(world, os.path.join(td, "init"), td, model), nprocs=world, join=True)
        res = [dict(np.load(os.path.join(td, f"rank{r}.npz"))) for r in range(world)]
    _check(tmvb, model, res)


@needs2
@pytest.mark.parametrize("model", ["lda", "ctm", "ctpf"])
def test_rccl_init_all_group_train_one_host_thread(tmvb, model):
    """ncclCommInitAll + tmvb_*_train_group: one host thread drives n GPUs; the n all-reduces of an iteration (and of every
    ELBO check) sit inside one RCCL group."""
    tm = tmvb
    world = _world()
    K = K_OF[model]
    corpus = _corpus(tm, model)
    ctxs = [tm.DeviceContext(i) for i in range(world)]
    comms = tm.Communicator.rccl_all(ctxs)
    bounds = corpus.shard_bounds(world)
    gms = []
    for i, (d0, d1) in enumerate(bounds):
        sh = corpus.shard(d0, d1)
        if model == "lda":
            g = tm.gpuLDA(sh, K, ctx=ctxs[i]); _init_lda(tm, g, K, corpus.V); g.set_comm(comms[i], corpus.M)
        elif model == "ctm":
            g = tm.gpuCTM(sh, K, ctx=ctxs[i]); _init_lda(tm, g, K, corpus.V); g.set_comm(comms[i], corpus.M)
        else:
            g = tm.gpuCTPF(sh, K, ctx=ctxs[i]); _init_ctpf(tm, g, K, corpus.V); g.set_comm(comms[i])
        gms.append(g)
    L = tm.lib()
    traj = np.full(ITERS, np.nan); done = C.c_int32(0); base = C.c_double(0.0)
    hs = (C.c_void_p * world)(*[g.handle for g in gms])
    pt = traj.ctypes.data_as(C.POINTER(C.c_double))
    if model == "ctpf":
        rc = L.tmvb_ctpf_train_group(hs, C.c_int32(world), C.c_int32(ITERS), C.c_double(0.0), C.c_int32(10), C.c_double(1.0 / K ** 2),
                                     C.c_int32(1), pt, C.byref(done), C.byref(base))
    else:
        fn = L.tmvb_lda_train_group if model == "lda" else L.tmvb_ctm_train_group
        rc = fn(hs, C.c_int32(world), C.c_int32(ITERS), C.c_double(0.0), C.c_int32(1000), C.c_double(1.0 / K ** 2), C.c_int32(10),
                C.c_double(1.0 / K ** 2), C.c_int32(1), pt, C.byref(done), C.byref(base))
    assert rc == 0, L.tmvb_last_error()
    assert done.value == ITERS
    res = []
    for g, (d0, d1) in zip(gms, bounds):
        g.update_host()
        out = {n: np.asarray(getattr(g, n)) for n in GLOBALS[model]}
        out.update(traj=traj.copy(), d0=d0, d1=d1, base=base.value, local=np.asarray(getattr(g, LOCAL[model])))
        res.append(out)
    _check(tm, model, res)
    for g, c in zip(gms, comms):
        g.set_comm(None) if model == "ctpf" else g.set_comm(None, g.M)
        g.close(); c.close()


@needs2
def test_fused_allreduce_form_over_real_rccl_equals_the_single_collective(tmvb):
    """Round-4 advice: the fused form of the sharded LDA iteration (tmvb_lda_estep_allreduce: the K-float tail all-reduced early on a side stream, the
    statistics behind it -- two streams of collectives on ONE communicator) had only ever run over RCCL with one rank, where RCCL launches no collective
    kernel at all.  It is opt-in (TMVB_FUSED_ALLREDUCE=1) until THIS test has passed on a multi-GPU node: both forms over real RCCL, one process per
    GPU, the same corpus and start; identical globals on every rank within each form, and form against form bit for bit at two ranks (an all-reduce of two
    addends has one summation order), to fp32 summation order beyond."""
    import torch.multiprocessing as mp
    world = _world()
    out = {}
    for fused in ("0", "1"):
        with tempfile.TemporaryDirectory() as td:
            mp.spawn(_worker, args=(world, os.path.join(td, "init"), td, "lda", fused), nprocs=world, join=True)
            out[fused] = [dict(np.load(os.path.join(td, f"rank{r}.npz"))) for r in range(world)]
        _check(tmvb, "lda", out[fused])
    for r in range(world):
        for n in ("alpha", "beta", "traj"):
            a, b = out["0"][r][n], out["1"][r][n]
            if world == 2:
                assert np.array_equal(a, b), (r, n)
            else:
                np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-12)
