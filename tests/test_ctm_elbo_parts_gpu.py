"""
The decomposed update_elbo! of the CTM path (round 5; ctm_elbo_kernel<NS, TOK = false> in csrc/tmvb_ctm.hip, src/CTM.jl:56-98), as
tests/test_lda_elbo_parts_gpu.py for LDA: an iteration that will be checked leaves the token terms behind on its way -- the E-step kernels
sum_i (phi counts)_i (lambda_i - lambda_old_i) per document, the statistics pass sum_n c_n log s_n per postings chunk, update_beta!
sum S (log(beta_new + eps) - log beta_old) -- and update_elbo! skips its token loop.  Both forms against the fp64 oracle and against each
other, on every E-step kernel (lane per document K <= 52 with its long-document companion, the Gauss-Jordan and LDS Newton kernels beyond),
teacher-forced stepwise and through train!.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tol import within
from test_ctm_gpu import force, make_pair, step, synth_case


@pytest.mark.parametrize("K", [3, 17, 50, 57, 64, 100, 124])
def test_both_forms_against_the_oracle_stepwise(tmvb, oracle, monkeypatch, K):
    g = synth_case(tmvb, K, M=120, V=400)
    monkeypatch.setenv("TMVB_CTM_ELBO_PARTS", "2")
    gp, om = make_pair(tmvb, oracle, g)
    monkeypatch.setenv("TMVB_CTM_ELBO_PARTS", "0")
    gw, _ = make_pair(tmvb, oracle, g)
    for it in range(3):
        force(gp, om); force(gw, om)
        step(om); e_o = om.update_elbo()
        step(gp); e_p = gp.update_elbo()
        step(gw); e_w = gw.update_elbo()
        assert gp.elbo_form() == 1 and gw.elbo_form() == 0
        within("ctm.elbo_rel_step", abs(e_p - e_o) / abs(e_o), (K, it, "decomposed", e_p, e_o))
        within("ctm.elbo_rel_step", abs(e_w - e_o) / abs(e_o), (K, it, "token walk", e_w, e_o))
        within("ctm.elbo_forms_rel", abs(e_p - e_w) / abs(e_w), (K, it, e_p, e_w))


def test_long_documents_beside_the_lane_kernel(tmvb, oracle, monkeypatch):
    """K = 50 with documents longer than the lane-per-document kernel takes (they run on the wave-per-document kernel beside it): both kernels leave pdot."""
    rng = np.random.default_rng(5)
    V, K = 4000, 50
    docs = [np.sort(rng.choice(V, size=int(rng.integers(5, 120)), replace=False)) for _ in range(200)]
    docs.insert(50, np.sort(rng.choice(V, size=3000, replace=False)))
    docs.insert(120, np.sort(rng.choice(V, size=2500, replace=False)))
    doc_ptr = np.concatenate([[0], np.cumsum([len(t) for t in docs])]).astype(np.int64)
    terms = np.concatenate(docs).astype(np.int32); counts = rng.integers(1, 4, size=len(terms)).astype(np.int32)
    g = dict(K=K, V=V, doc_ptr=doc_ptr, terms=terms, counts=counts, beta0=tmvb.dirichlet_rows(K, V, seed=5))
    monkeypatch.setenv("TMVB_CTM_ELBO_PARTS", "2")
    gp, om = make_pair(tmvb, oracle, g)
    for it in range(2):
        force(gp, om)
        step(om); e_o = om.update_elbo()
        step(gp); e_p = gp.update_elbo()
        assert gp.elbo_form() == 1
        within("ctm.elbo_rel_step", abs(e_p - e_o) / abs(e_o), (it, e_p, e_o))


def test_state_set_by_the_host_falls_back_to_the_token_walk(tmvb, oracle, monkeypatch):
    monkeypatch.setenv("TMVB_CTM_ELBO_PARTS", "2")
    g = synth_case(tmvb, 20, M=80, V=300)
    gm, om = make_pair(tmvb, oracle, g)
    step(gm); gm.update_elbo()
    assert gm.elbo_form() == 1
    step(om)
    force(gm, om)
    e_g = gm.update_elbo(); e_o = om.update_elbo()
    assert gm.elbo_form() == 0
    within("ctm.elbo_rel_step", abs(e_g - e_o) / abs(e_o), (e_g, e_o))
    gm.estep(); gm.reduce_docs()                          # an E-step without update_beta! behind it
    gm.update_elbo()
    assert gm.elbo_form() == 0


def test_train_takes_the_decomposed_form_and_leaves_the_iteration_alone(tmvb, monkeypatch):
    pc = tmvb.syn_nsf(M=3000, V=2000, seed=17)
    K = 50
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=5)
    out = []
    for env in ("1", "0"):
        monkeypatch.setenv("TMVB_CTM_ELBO_PARTS", env)
        g = tmvb.gpuCTM(pc, K)
        g.beta = np.asfortranarray(beta0); g.beta_old = g.beta.copy(order="F"); g.update_buffer()
        traj = np.asarray(g.train(iter=5, tol=0.0, checkelbo=1, printelbo=False), dtype=np.float64)
        out.append((g, traj))
    (gp, tp), (gw, tw) = out
    assert gp.elbo_form() == 1 and gw.elbo_form() == 0
    assert len(tp) == len(tw) == 5 and np.all(np.isfinite(tp))
    within("ctm.elbo_forms_rel", np.abs(tp - tw) / np.abs(tw), (tp, tw))
    assert np.array_equal(gp.beta, gw.beta) and np.array_equal(gp.lam, gw.lam) and np.array_equal(gp.vsq, gw.vsq)
    assert np.array_equal(gp.mu, gw.mu) and np.array_equal(gp.sigma, gw.sigma)
    g = out[0][0]
    g.train(iter=2, tol=0.0, checkelbo=np.inf, printelbo=False)      # unchecked iterations collect nothing
    g.update_elbo()
    assert g.elbo_form() == 0
