"""
The decomposed update_elbo! of the fCTM path (round 6; fctm_elbo_doc_parts_kernel in csrc/tmvb_ctm.hip, src/fCTM.jl:68-125): as
tests/test_flda_elbo_parts_gpu.py with lambda for Elogtheta -- the E-step kernels' exit test leaves sum_i (phi counts)_i (lambda_i - lambda_old_i) per
document (all three kernels: lane per document K <= 50, register Gauss-Jordan K <= 60, LDS / global Newton beyond), the token phases the per-token
exponent of update_tau!.  Both forms against the fp64 oracle and against each other.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tol import within
from test_fctm_gpu import force, make_pair, step, synth_case


@pytest.mark.parametrize("K", [3, 12, 25, 50, 57, 64, 100, 150])
def test_both_forms_against_the_oracle_stepwise(tmvb, oracle, monkeypatch, K):
    g = synth_case(tmvb, K, M=100, V=300)
    monkeypatch.setenv("TMVB_FCTM_ELBO_PARTS", "2")
    gp, om = make_pair(tmvb, oracle, g)
    monkeypatch.setenv("TMVB_FCTM_ELBO_PARTS", "0")
    gw, _ = make_pair(tmvb, oracle, g)
    for it in range(3):
        force(gp, om); force(gw, om)
        step(om, viter=3, vtol=0.0); e_o = om.update_elbo()
        step(gp, viter=3, vtol=0.0); e_p = gp.update_elbo()
        step(gw, viter=3, vtol=0.0); e_w = gw.update_elbo()
        assert gp.elbo_form() == 1 and gw.elbo_form() == 0
        big = "" if K <= 128 else ".bigk"
        within("fctm.elbo_rel_step" + big, abs(e_p - e_o) / abs(e_o), (K, it, "decomposed", e_p, e_o))
        within("fctm.elbo_rel_step" + big, abs(e_w - e_o) / abs(e_o), (K, it, "token walk", e_w, e_o))
        within("fctm.elbo_forms_rel", abs(e_p - e_w) / abs(e_w), (K, it, e_p, e_w))


def test_default_exit_rule_and_long_documents(tmvb, oracle, monkeypatch):
    rng = np.random.default_rng(3)
    V, K = 3000, 20
    docs = [np.sort(rng.choice(V, size=int(rng.integers(300, 900)) if d % 3 == 0 else int(rng.integers(5, 60)), replace=False)) for d in range(40)]
    docs.insert(7, np.sort(rng.choice(V, size=2500, replace=False)))            # longer than the lane-per-document kernel takes
    doc_ptr = np.concatenate([[0], np.cumsum([len(t) for t in docs])]).astype(np.int64)
    terms = np.concatenate(docs).astype(np.int32); counts = rng.integers(1, 4, size=len(terms)).astype(np.int32)
    g = dict(K=K, V=V, doc_ptr=doc_ptr, terms=terms, counts=counts, beta0=tmvb.dirichlet_rows(K, V, seed=5), kappa0=tmvb.dirichlet_rows(1, V, seed=9)[0])
    monkeypatch.setenv("TMVB_FCTM_ELBO_PARTS", "2")
    gp, om = make_pair(tmvb, oracle, g)
    monkeypatch.setenv("TMVB_FCTM_ELBO_PARTS", "0")
    gw, _ = make_pair(tmvb, oracle, g)
    for it in range(2):
        force(gp, om); force(gw, om)
        step(om)
        step(gp); e_p = gp.update_elbo()
        step(gw); e_w = gw.update_elbo()
        assert gp.elbo_form() == 1 and gw.elbo_form() == 0
        within("fctm.elbo_forms_rel", abs(e_p - e_w) / abs(e_w), (it, e_p, e_w))


def test_state_set_by_the_host_falls_back_to_the_token_walk(tmvb, oracle, monkeypatch):
    monkeypatch.setenv("TMVB_FCTM_ELBO_PARTS", "2")
    g = synth_case(tmvb, 20, M=80, V=300)
    gm, om = make_pair(tmvb, oracle, g)
    step(gm); gm.update_elbo()
    assert gm.elbo_form() == 1
    step(om)
    force(gm, om)
    e_g = gm.update_elbo(); e_o = om.update_elbo()
    assert gm.elbo_form() == 0
    within("fctm.elbo_rel_step", abs(e_g - e_o) / abs(e_o), (e_g, e_o))
    gm.estep(); gm.reduce_docs()                          # an E-step without update_beta! behind it
    gm.update_elbo()
    assert gm.elbo_form() == 0


def test_train_takes_the_decomposed_form_and_tracks_the_walk(tmvb, monkeypatch):
    pc = tmvb.syn_nsf(M=2000, V=1500, seed=17)
    K = 50
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=5); kappa0 = tmvb.dirichlet_rows(1, pc.V, seed=9)[0]
    out = []
    for env in ("1", "0"):
        monkeypatch.setenv("TMVB_FCTM_ELBO_PARTS", env)
        g = tmvb.gpufCTM(pc, K)
        g.beta = np.asfortranarray(beta0); g.beta_old = g.beta.copy(order="F"); g.kappa = kappa0.copy(); g.kappa_old = kappa0.copy(); g.update_buffer()
        traj = g.train(iter=6, tol=0.0, checkelbo=1, printelbo=False)
        out.append((np.asarray(traj), g.elbo_form()))
    (tp, fp), (tw, fw) = out
    assert fp == 1 and fw == 0 and len(tp) == len(tw) == 6
    within("fctm.elbo_forms_rel", np.abs(tp - tw) / np.abs(tw), (tp, tw))
