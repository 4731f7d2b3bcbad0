#!/usr/bin/env python
"""What does update_elbo! cost per model at the benched sizes?  (check_elbo! runs it after EVERY iteration under train!'s default checkelbo = 1,
src/modelutils.jl:574-585, so the time to the ELBO plateau -- the second half of BASELINE.json's metric -- pays it per iteration.)
For each model: a few iterations from the cold start, then 10 x (iteration) and 10 x (iteration + update_elbo!) between synchronisations,
and update_elbo! alone.  LDA and CTM with TMVB_LDA_ELBO_PARTS / TMVB_CTM_ELBO_PARTS = 2 / 0 (decomposed form / token walk, see tests/test_lda_elbo_parts_gpu.py).
Usage: python tools/elbo_cost_probe.py [lda50] [lda100] [ctm] [ctpf] [flda] [fctm]     one JSON line per measurement on stdout."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tmvb_amd

tm = tmvb_amd.pkg


def measure(name, gm, it, burnin=20, n=10):
    for _ in range(burnin):
        it()
    gm.update_elbo(); gm.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        it()
    gm.synchronize(); plain = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        it(); e = gm.update_elbo()
    gm.synchronize(); checked = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        e2 = gm.update_elbo()
    alone = (time.perf_counter() - t0) / n
    out = {"model": name, "ms_per_iteration": 1e3 * plain, "ms_per_checked_iteration": 1e3 * checked, "ms_update_elbo_alone": 1e3 * alone,
           "elbo": e, "elbo_again": e2}
    if hasattr(gm, "elbo_form"):
        out["elbo_form_last"] = gm.elbo_form()
    print(json.dumps(out), flush=True)


def lda(K, parts):
    os.environ["TMVB_LDA_ELBO_PARTS"] = parts
    pc = tm.syn_nsf()
    gm = tm.gpuLDA(pc, K)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=3)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()

    def it():
        gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_alpha()
    measure(f"lda K={K} TMVB_LDA_ELBO_PARTS={parts}", gm, it)
    gm.close()
    del os.environ["TMVB_LDA_ELBO_PARTS"]


def ctm(K=50, parts="2"):
    os.environ["TMVB_CTM_ELBO_PARTS"] = parts
    pc = tm.syn_nsf()
    gm = tm.gpuCTM(pc, K)
    del os.environ["TMVB_CTM_ELBO_PARTS"]
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()

    def it():
        gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_sigma(); gm.update_mu()
    measure(f"ctm K={K} TMVB_CTM_ELBO_PARTS={parts}", gm, it, burnin=10, n=5)
    gm.close()


def ctpf(K=50, parts="2"):
    os.environ["TMVB_CTPF_ELBO_PARTS"] = parts
    pc = tm.syn_citeu()
    gm = tm.gpuCTPF(pc, K)
    del os.environ["TMVB_CTPF_ELBO_PARTS"]

    def it():
        gm.estep(); gm.reduce_docs(); gm.mstep()
    measure(f"ctpf K={K} TMVB_CTPF_ELBO_PARTS={parts}", gm, it, burnin=50, n=50)
    gm.close()


def flda(K=50, parts="2"):
    os.environ["TMVB_FLDA_ELBO_PARTS"] = parts
    pc = tm.syn_nsf()
    gm = tm.gpufLDA(pc, K)
    del os.environ["TMVB_FLDA_ELBO_PARTS"]
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F")
    gm.kappa = tm.dirichlet_rows(1, pc.V, seed=9)[0].copy(); gm.kappa_old = gm.kappa.copy(); gm.update_buffer()

    def it():
        gm.estep(10, 1.0 / K ** 2); gm.reduce_docs(); gm.update_beta(); gm.update_alpha(1000, 1.0 / K ** 2); gm.update_eta()
    measure(f"flda K={K} TMVB_FLDA_ELBO_PARTS={parts}", gm, it, burnin=15, n=10)
    gm.close()


def fctm(K=50, parts="2"):
    os.environ["TMVB_FCTM_ELBO_PARTS"] = parts
    pc = tm.syn_nsf()
    gm = tm.gpufCTM(pc, K)
    del os.environ["TMVB_FCTM_ELBO_PARTS"]
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F")
    gm.kappa = tm.dirichlet_rows(1, pc.V, seed=9)[0].copy(); gm.kappa_old = gm.kappa.copy(); gm.update_buffer()

    def it():
        gm.estep(); gm.reduce_docs(); gm.mstep()
    measure(f"fctm K={K} TMVB_FCTM_ELBO_PARTS={parts}", gm, it, burnin=8, n=4)
    gm.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["lda50", "lda100", "ctm", "ctpf"]
    if "lda50" in which:
        lda(50, "2"); lda(50, "0")
    if "lda100" in which:
        lda(100, "2"); lda(100, "0")
    if "ctm" in which:
        ctm(parts="2"); ctm(parts="0")
    if "flda" in which:
        flda(parts="2"); flda(parts="0")
    if "fctm" in which:
        fctm(parts="2"); fctm(parts="0")
    if "ctpf" in which:
        ctpf(parts="2"); ctpf(parts="0")
