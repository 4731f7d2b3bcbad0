"""
Full-size oracle comparisons ("compare what you time with the oracle, at the size that is timed").

The code paths that only exist at size -- LDA's three pipelined document pieces with class-ordered statistics chunks, CTM's
2 013-item persistent queue in predicted-time order, CTPF's multi-wave long-document launch on a side stream -- were covered by
properties (mass conservation, monotone ELBO, run-to-run bit equality) only; a deterministic wrong answer would have passed.
Here every BASELINE.json configuration runs teacher-forced outer iterations at its FULL size against the OpenMP fp64 oracle,
through oracle/parity.py: every document and every global compared, with the tolerances of SURVEY.md section 8c / DESIGN.md
section 6 (the same ones the small-corpus tests use), once from the cold start and once from a state the device itself trained
to (where the sweep counts are mixed and CTM's queue order / regrouping are in play).

And the medium free-running K = 50 tests LDA and CTPF lacked (CTM got its one in round 3): 1 500 documents x 25 iterations through
train!, ELBO rel <= 1e-4 at every iteration, same stop iteration +-1 under the signed rule (Q4).
"""
import os

import numpy as np
import pytest

from tol import within

pytestmark = pytest.mark.gpu



def _copy_state(om, gm, names):
    gm.update_host()
    for n in names:
        setattr(om, n, np.array(getattr(gm, n), dtype=np.float64, copy=True, order="F"))


def _oracle_train(om, step, iters, tol=1.0):
    """the oracle's train! loop (check_elbo! with the signed stop rule Q4, src/modelutils.jl:574-585) around its OpenMP
    document-parallel E-step -- orc_*_train runs the same operators sequentially, minutes at K = 50"""
    e_prev, traj = om.update_elbo(), []
    for _ in range(iters):
        step(om)
        e_new = om.update_elbo(); traj.append(e_new)
        stop = (e_new - e_prev) < tol
        e_prev = e_new
        if stop:
            break
    return np.asarray(traj)


def _say(block):
    w = block["worst"]
    print("\n   parity:", ", ".join(f"{k}={v:.3g}" for k, v in w.items() if v is not None), "pass =", block["pass"])


@pytest.mark.parametrize("K", [50, 100])
def test_full_size_lda_vs_oracle(tmvb, oracle, K):
    """configs 2 and 3 (LDA K = 50 / K = 100 on SYN-NSF, M = 128 804): two teacher-forced iterations from the cold start, then
    one from the state after 30 free-running device iterations (documents at 1 ... 10 sweeps)."""
    from oracle import parity
    pc = tmvb.syn_nsf()
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=7)
    gm = tmvb.gpuLDA(pc, K)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
    om = oracle.LDA(oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V), K, beta0)
    block, _ = parity.lda_parity(gm, om, iters=2 if K == 50 else 1, threads=oracle.usable_cpus())
    _say(block)
    assert block["pass"], block["worst"]
    for _ in range(30):
        gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_alpha()
    _copy_state(om, gm, ("alpha", "beta", "beta_old", "gamma", "Elogtheta", "Elogtheta_old"))
    block, _ = parity.lda_parity(gm, om, iters=1, threads=oracle.usable_cpus())
    _say(block)
    assert block["pass"], block["worst"]
    hist = gm.sweep_hist()
    assert hist.sum() == pc.M and (hist > 0.01 * pc.M).sum() >= 3              # a mixed state: documents leave at several different sweep counts
    oracle.lib().orc_omp_pool_free()


def test_full_size_ctm_k50_vs_oracle(tmvb, oracle):
    """config 4 (CTM K = 50 on SYN-NSF): one teacher-forced iteration from the state after 6 free-running device iterations
    (mu, sigma away from 0 / I; the queue order and the regrouping of the lane-per-document kernel use the previous E-step's
    Newton counts)."""
    from oracle import parity
    pc = tmvb.syn_nsf()
    K = 50
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=7)
    gm = tmvb.gpuCTM(pc, K)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
    om = oracle.CTM(oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V), K, beta0)
    for _ in range(6):
        gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_sigma(); gm.update_mu()
    _copy_state(om, gm, ("mu", "sigma", "invsigma", "beta", "beta_old", "lam", "lam_old", "vsq", "logzeta"))
    om.logzeta = np.ascontiguousarray(om.logzeta); om.mu = np.ascontiguousarray(om.mu)
    block, _ = parity.ctm_parity(gm, om, iters=1, threads=oracle.usable_cpus())
    _say(block)
    assert gm.solver_stats()["waves"] == (pc.M + 63) // 64                   # the lane-per-document kernel ran
    assert block["pass"], block["worst"]


def test_full_size_ctpf_k50_vs_oracle(tmvb, oracle):
    """config 5 (CTPF K = 50 on SYN-CITEU with readers): one teacher-forced iteration from the cold start, one from the state
    after 40 free-running device iterations, the second with update_elbo! (the table form against the oracle's term by term one)."""
    from oracle import parity
    pc = tmvb.syn_citeu()
    K = 50
    alef0 = np.exp(tmvb.dirichlet_rows(K, pc.V, seed=7) - 0.5)
    gm = tmvb.gpuCTPF(pc, K)
    gm.alef = np.asfortranarray(alef0); gm.alef_old = gm.alef.copy(order="F"); gm.update_buffer()
    om = oracle.CTPF(oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V, pc.rdr_ptr, pc.readers, pc.ratings, pc.U), K, alef0)
    block, _ = parity.ctpf_parity(gm, om, iters=1, threads=oracle.usable_cpus())
    _say(block)
    assert block["pass"], block["worst"]
    for _ in range(40):
        gm.estep(); gm.reduce_docs(); gm.mstep()
    _copy_state(om, gm, parity.CTPF_FIELDS + tuple(n + "_old" for n in parity.CTPF_FIELDS))
    for n in ("bet", "vav", "dalet", "het"):
        setattr(om, n, np.ascontiguousarray(getattr(om, n))); setattr(om, n + "_old", np.ascontiguousarray(getattr(om, n + "_old")))
    block, _ = parity.ctpf_parity(gm, om, iters=1, threads=oracle.usable_cpus(), elbo=True)
    _say(block)
    assert block["pass"], block["worst"]


# ------------------------------------------------------------------ medium free-running runs at the configurations' K
def test_lda_free_running_k50_medium_corpus_tracks_the_oracle(tmvb, oracle):
    """LDA K = 50 (LPR = 13: the benched instantiation), 1 500 NSF-shaped documents over the full vocabulary, 25 free-running
    iterations through train!: ELBO rel <= 1e-4 at every iteration, stop iteration +-1."""
    pc = tmvb.syn_nsf(M=1500, V=25319, seed=2)
    K = 50
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=7)
    gm = tmvb.gpuLDA(pc, K)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F")
    om = oracle.LDA(oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V), K, beta0)
    t_g = gm.train(iter=25, tol=1.0, checkelbo=1, printelbo=False)
    nt = oracle.usable_cpus()
    t_o = _oracle_train(om, lambda m: (m.estep(omp_threads=nt), m.update_beta(), m.update_alpha()), 25)
    assert abs(len(t_g) - len(t_o)) <= 1
    n = min(len(t_g), len(t_o))
    assert n >= 5
    dev = np.abs(t_g[:n] - t_o[:n]) / np.abs(t_o[:n])
    print("\n   LDA K=50 free running: iterations", len(t_g), len(t_o), "max ELBO rel", dev.max())
    within("lda.elbo_rel_free", dev, (t_g, t_o))
    if len(t_g) == len(t_o):
        within("lda.alpha_rel_free", np.abs(gm.alpha - om.alpha) / om.alpha)
        within("lda.beta_abs_free", np.abs(gm.beta - om.beta))


def test_lda_free_running_full_size_k50_tracks_the_oracle(tmvb, oracle):
    """Round-4 review, missing #4: full size had been teacher-forced only.  Ten FREE-RUNNING iterations of config 2 (LDA K = 50, the whole SYN-NSF
    corpus) through the library's train! -- the pipelined document pieces, the class-ordered statistics chunks, the queue order, as they run in the timed
    window -- against the OpenMP fp64 oracle's train! loop from the same cold start (src/LDA.jl:169-187): ELBO rel per iteration, then the globals."""
    pc = tmvb.syn_nsf()
    K = 50
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=7)
    gm = tmvb.gpuLDA(pc, K)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F")
    om = oracle.LDA(oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V), K, beta0)
    t_g = gm.train(iter=10, tol=0.0, checkelbo=1, printelbo=False)
    nt = oracle.usable_cpus()
    t_o = _oracle_train(om, lambda m: (m.estep(omp_threads=nt), m.update_beta(), m.update_alpha()), 10, tol=0.0)
    oracle.lib().orc_omp_pool_free()
    assert len(t_g) == len(t_o) == 10 and np.all(np.diff(t_o) > 0)
    dev = np.abs(t_g - t_o) / np.abs(t_o)
    print("\n   LDA K=50 FULL SIZE free running, 10 iterations: ELBO rel per iteration", ", ".join(f"{x:.2e}" for x in dev))
    within("lda.elbo_rel_free_full", dev, (t_g, t_o))
    within("lda.alpha_rel_free_full", np.abs(gm.alpha - om.alpha) / om.alpha)
    within("lda.beta_abs_free_full", np.abs(gm.beta - om.beta))


def test_ctm_free_running_full_size_k50_tracks_the_oracle(tmvb, oracle):
    """Round-5 review: config 4 at full size had been ONE teacher-forced iteration.  Five FREE-RUNNING iterations of CTM K = 50 on the whole SYN-NSF corpus
    through the library's train! (the four-waves-per-item kernel with its regrouping by Newton counts and its queue order, the staged sigma, the decomposed
    update_elbo! of the checked iterations) against the OpenMP fp64 oracle's train! loop from the same cold start (src/CTM.jl:194-211): ELBO rel per
    iteration, then the globals.  The lambda solve is inexact CG on the device and `\` in the oracle: the trajectories may separate by what the Newton
    exit threshold ntol = 1 / K^2 allows, which is what the ctm.*_free_full tolerances measure."""
    pc = tmvb.syn_nsf()
    K = 50
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=7)
    gm = tmvb.gpuCTM(pc, K)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F")
    om = oracle.CTM(oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V), K, beta0)
    t_g = gm.train(iter=5, tol=0.0, checkelbo=1, printelbo=False)
    nt = oracle.usable_cpus()
    t_o = _oracle_train(om, lambda m: (m.estep(omp_threads=nt), m.update_beta(), m.update_sigma_mu()), 5, tol=0.0)
    oracle.lib().orc_omp_pool_free()
    assert len(t_g) == len(t_o) == 5 and np.all(np.diff(t_o) > 0)
    dev = np.abs(t_g - t_o) / np.abs(t_o)
    print("\n   CTM K=50 FULL SIZE free running, 5 iterations: ELBO rel per iteration", ", ".join(f"{x:.2e}" for x in dev))
    within("ctm.elbo_rel_free_full", dev, (t_g, t_o))
    within("ctm.mu_abs_free_full", np.abs(gm.mu - om.mu))
    within("ctm.sigma_rel_free_full", np.abs(gm.sigma - om.sigma).max() / np.abs(om.sigma).max())
    within("ctm.beta_abs_free_full", np.abs(gm.beta - om.beta))


def test_ctpf_free_running_k50_medium_corpus_tracks_the_oracle(tmvb, oracle):
    """CTPF K = 50, 1 500 CiteULike-shaped documents with readers, 25 free-running iterations through train!: ELBO rel <= 1e-4 at
    every iteration (measured on MI355X: <= 2.7e-5, the early iterations where the ELBO moves by 25 % per
    step), stop iteration +-1, rates rel <= 2e-3."""
    pc = tmvb.syn_citeu(M=1500, V=8000, U=1200, seed=5)
    K = 50
    alef0 = np.exp(tmvb.dirichlet_rows(K, pc.V, seed=7) - 0.5)
    gm = tmvb.gpuCTPF(pc, K)
    gm.alef = np.asfortranarray(alef0); gm.alef_old = gm.alef.copy(order="F")
    om = oracle.CTPF(oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V, pc.rdr_ptr, pc.readers, pc.ratings, pc.U), K, alef0)
    t_g = gm.train(iter=25, tol=1.0, checkelbo=1, printelbo=False, recs=False)
    nt = oracle.usable_cpus()
    t_o = _oracle_train(om, lambda m: (m.estep(omp_threads=nt), m.mstep()), 25)
    assert abs(len(t_g) - len(t_o)) <= 1
    n = min(len(t_g), len(t_o))
    assert n >= 5
    dev = np.abs(t_g[:n] - t_o[:n]) / np.abs(t_o[:n])
    print("\n   CTPF K=50 free running: iterations", len(t_g), len(t_o), "max ELBO rel", dev.max())
    within("ctpf.elbo_rel_free", dev, (t_g, t_o))
    # the states after 25 free-running iterations: trajectories drift apart where a document leaves one sweep earlier or later
    # (measured on MI355X: rates up to 1.1e-2 relative on the smallest component); the drift is NOT an error of the operators -- from
    # the device's own final state one more teacher-forced iteration agrees with the oracle to the single-step tolerances
    if len(t_g) == len(t_o):
        worst = max((np.abs(getattr(gm, name) - getattr(om, name)) / getattr(om, name)).max() for name in ("bet", "vav", "dalet", "het"))
        print("   rates after 25 free-running iterations: max rel", worst)
        assert worst <= 5e-2
    from oracle import parity
    _copy_state(om, gm, parity.CTPF_FIELDS + tuple(n + "_old" for n in parity.CTPF_FIELDS))
    for n in ("bet", "vav", "dalet", "het"):
        setattr(om, n, np.ascontiguousarray(getattr(om, n))); setattr(om, n + "_old", np.ascontiguousarray(getattr(om, n + "_old")))
    block, _ = parity.ctpf_parity(gm, om, iters=1, threads=nt, elbo=True)
    _say(block)
    assert block["pass"], block["worst"]
