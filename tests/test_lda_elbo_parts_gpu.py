"""
The decomposed update_elbo! of the LDA path (round 5; lda_elbo_doc_kernel in csrc/tmvb_lda.hip, src/LDA.jl:50-93).

An iteration that will be checked leaves the per-token parts of the ELBO behind on its way -- the statistics passes the sum of
c_n log s_n per postings chunk, update_beta! the sum of S (log beta_new - log beta_old) -- and update_elbo! is one per-document
kernel instead of a second walk over the corpus.  Both forms evaluate the reference's sum; they must agree with each other to
fp32 rounding and each with the fp64 oracle inside the frozen tolerance, on every kernel path (statistics-pass instantiations by
K, one pass and pipelined pieces, train! and the stepwise operators).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tol import within

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pair(tmvb, oracle, pc, K, beta0):
    gm = tmvb.gpuLDA(pc, K)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F")
    gm.update_buffer()
    om = oracle.LDA(oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V), K, beta0) if oracle is not None else None
    return gm, om


def _force(gm, om):
    gm.alpha = om.alpha.copy(); gm.beta = om.beta.copy(order="F"); gm.beta_old = om.beta_old.copy(order="F")
    gm.gamma = om.gamma.copy(order="F"); gm.Elogtheta = om.Elogtheta.copy(order="F")
    gm.Elogtheta_old = om.Elogtheta_old.copy(order="F")
    gm.update_buffer()


@pytest.mark.parametrize("K", [1, 3, 7, 50, 70, 100, 120])
def test_both_forms_against_the_oracle_stepwise(tmvb, oracle, monkeypatch, K):
    """TMVB_LDA_ELBO_PARTS=2 (read at model creation): every E-step collects, so the stepwise operators take the decomposed form too."""
    pc = tmvb.syn_nsf(M=300, V=700, seed=5)
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=3)
    monkeypatch.setenv("TMVB_LDA_ELBO_PARTS", "2")
    gp, om = _pair(tmvb, oracle, pc, K, beta0)
    monkeypatch.setenv("TMVB_LDA_ELBO_PARTS", "0")
    gw, _ = _pair(tmvb, None, pc, K, beta0)
    for it in range(3):
        _force(gp, om); _force(gw, om)
        om.estep(viter=4, vtol=0.0); om.update_beta(); om.update_alpha()
        e_o = om.update_elbo()
        vals = []
        for g in (gp, gw):
            g.estep(viter=4, vtol=0.0); g.reduce_docs(); g.update_beta(); g.update_alpha()
            vals.append(g.update_elbo())
        assert gp.elbo_form() == 1 and gw.elbo_form() == 0
        within("lda.elbo_rel_step", abs(vals[0] - e_o) / abs(e_o), (K, it, "decomposed", vals[0], e_o))
        within("lda.elbo_rel_step", abs(vals[1] - e_o) / abs(e_o), (K, it, "token walk", vals[1], e_o))
        within("lda.elbo_forms_rel", abs(vals[0] - vals[1]) / abs(vals[1]), (K, it, vals))


def test_state_set_by_the_host_falls_back_to_the_token_walk(tmvb, oracle, monkeypatch):
    """The parts belong to ONE iteration: after update_buffer() (tmvb_lda_set_state) nothing of them may be used."""
    monkeypatch.setenv("TMVB_LDA_ELBO_PARTS", "2")
    pc = tmvb.syn_nsf(M=200, V=500, seed=8)
    K = 20
    gm, om = _pair(tmvb, oracle, pc, K, tmvb.dirichlet_rows(K, pc.V, seed=3))
    gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_alpha()
    gm.update_elbo()
    assert gm.elbo_form() == 1
    om.estep(); om.update_beta(); om.update_alpha()
    _force(gm, om)
    e_g = gm.update_elbo(); e_o = om.update_elbo()
    assert gm.elbo_form() == 0
    within("lda.elbo_rel_step", abs(e_g - e_o) / abs(e_o), (e_g, e_o))
    # an E-step without the M-step behind it: alpha and the statistics' share are missing
    gm.estep(); gm.reduce_docs()
    gm.update_elbo()
    assert gm.elbo_form() == 0


@pytest.mark.parametrize("K,M", [(50, 60000), (100, 56000)])
def test_train_takes_the_decomposed_form_on_pipelined_pieces(tmvb, monkeypatch, K, M):
    """train!(checkelbo = 1) on a corpus large enough for the pipelined plan (document pieces, the last pass in its own buffer): the checked
    trajectory of the decomposed form against the token walk's, iteration by iteration, and the final state bit for bit (the ELBO form must not
    touch the iteration itself)."""
    pc = tmvb.syn_nsf(M=M, V=8000, seed=17)
    assert pc.nnz >= (1 << 22), "needs the pipelined plan (lda_piece_count)"
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=3)
    out = []
    for env in ("1", "0"):
        monkeypatch.setenv("TMVB_LDA_ELBO_PARTS", env)
        g, _ = _pair(tmvb, None, pc, K, beta0)
        traj = g.train(iter=6, tol=0.0, checkelbo=1, printelbo=False)
        out.append((g, np.asarray(traj, dtype=np.float64)))
    (gp, tp), (gw, tw) = out
    assert gp.elbo_form() == 1 and gw.elbo_form() == 0
    assert len(tp) == len(tw) == 6 and np.all(np.isfinite(tp))
    within("lda.elbo_forms_rel", np.abs(tp - tw) / np.abs(tw), (K, tp, tw))
    assert np.array_equal(gp.beta, gw.beta) and np.array_equal(gp.alpha, gw.alpha)
    assert np.array_equal(gp.gamma, gw.gamma) and np.array_equal(gp.Elogtheta, gw.Elogtheta)


def test_unchecked_iterations_collect_nothing(tmvb):
    """checkelbo = Inf: no iteration is checked, so none pays for the parts; a later update_elbo! takes the token walk."""
    pc = tmvb.syn_nsf(M=300, V=700, seed=5)
    K = 50
    g, _ = _pair(tmvb, None, pc, K, tmvb.dirichlet_rows(K, pc.V, seed=3))
    g.train(iter=3, tol=0.0, checkelbo=np.inf, printelbo=False)
    g.update_elbo()
    assert g.elbo_form() == 0
    g.train(iter=3, tol=0.0, checkelbo=3, printelbo=False)      # only the third iteration is checked
    assert g.elbo_form() == 1


@pytest.mark.parametrize("name", ["lda_m40_v60_k7", "lda_m30_v50_k70_empty"])
def test_golden_fixtures_with_empty_documents_and_unused_terms(tmvb, oracle, monkeypatch, name):
    """Empty documents (gamma = alpha, no postings) and vocabulary entries no document uses (S = 0 meets log beta_old) in the decomposed form; and
    viter = 0 (the E-step leaves no responsibilities behind: the token walk has to take over)."""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    g = {k: z[k] for k in z.files}
    K, V = int(g["K"]), int(g["V"])
    monkeypatch.setenv("TMVB_LDA_ELBO_PARTS", "2")
    pc = tmvb.PackedCorpus(g["doc_ptr"], g["terms"], g["counts"], V)
    gm, om = _pair(tmvb, oracle, pc, K, g["beta0"])
    for it in range(3):
        _force(gm, om)
        om.estep(viter=3, vtol=0.0); om.update_beta(); om.update_alpha(); e_o = om.update_elbo()
        gm.estep(viter=3, vtol=0.0); gm.reduce_docs(); gm.update_beta(); gm.update_alpha(); e_g = gm.update_elbo()
        assert gm.elbo_form() == 1
        within("lda.elbo_rel_step", abs(e_g - e_o) / abs(e_o), (name, it, e_g, e_o))
    _force(gm, om)
    om.estep(viter=0, vtol=0.0); om.update_beta(); om.update_alpha(); e_o = om.update_elbo()
    gm.estep(viter=0, vtol=0.0); gm.reduce_docs(); gm.update_beta(); gm.update_alpha(); e_g = gm.update_elbo()
    assert gm.elbo_form() == 0
    assert np.isfinite(e_g)
