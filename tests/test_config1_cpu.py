"""
BASELINE.json config 1: "LDA K=10 on CiteULike (16980 docs, V=8000), CPU reference train! path -- plumbing, no GPU".
The reference's CPU path is represented here by the fp64 oracle (the Julia original cannot run in this image); the
corpus is SYN-CITEU's terms/counts.  Runs on CPU in a few seconds with the OpenMP E-step.
Checks the post-conditions of check_model (src/modelutils.jl:39-67), token-mass conservation of gamma, the monotone
ELBO of coordinate ascent while alpha is fixed (niter = 0), and that the serial and OpenMP E-steps agree.
"""
import os

import numpy as np


def test_config1_oracle_lda_k10_on_syn_citeu(tmvb, oracle):
    pc = tmvb.syn_citeu()                                   # M = 16 980, V = 8 000 (U = 5 551 readers unused by LDA)
    assert (pc.M, pc.V) == (16980, 8000)
    K = 10
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=7)
    csr = oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V)
    om = oracle.LDA(csr, K, beta0)
    threads = min(os.cpu_count() or 1, 8)
    elbos = []
    for it in range(4):
        om.estep(omp_threads=threads); om.update_beta(); om.update_alpha(niter=0)     # alpha fixed: pure coordinate ascent
        elbos.append(om.update_elbo())
    assert all(np.isfinite(elbos)) and all(b > a for a, b in zip(elbos, elbos[1:])), elbos
    np.testing.assert_allclose(om.beta.sum(axis=1), 1.0, rtol=1e-12)                  # isprobvec rows
    assert np.all(om.gamma > 0) and np.all(om.Elogtheta <= 0) and np.all(om.alpha > 0)
    # gamma_d = eps + alpha + phi * counts: the phi columns sum to one, so sum_i (gamma - alpha) = C_d
    C = np.add.reduceat(pc.counts, pc.doc_ptr[:-1][np.diff(pc.doc_ptr) > 0])
    mass = (om.gamma - om.alpha[:, None]).sum(axis=0)[np.diff(pc.doc_ptr) > 0]
    np.testing.assert_allclose(mass, C, rtol=1e-9)
    # the full train! loop (alpha Newton on) keeps the state valid and the signed stop rule does not fire early here
    om2 = oracle.LDA(csr, K, beta0)
    om3 = oracle.LDA(csr, K, beta0)
    om2.estep(omp_threads=threads); om3.estep()
    np.testing.assert_allclose(om2.gamma, om3.gamma, rtol=1e-12)                      # per-document work is order independent
    np.testing.assert_allclose(om2.beta_temp, om3.beta_temp, rtol=1e-9)               # statistics: summation order only
    om2.update_beta(); om2.update_alpha()
    assert np.all(np.isfinite(om2.alpha)) and np.all(om2.alpha > 0)
