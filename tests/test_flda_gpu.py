"""
GPU parity tests for the filtered-LDA path (new device path; oracle src/fLDA.jl): HIP engine through the C ABI vs the
fp64 oracle and the committed golden fixtures.  Tolerances (fp64 -> fp32, one v_exp_f32 per (token, topic) per sweep): the flda.* keys of
tests/tol.py -- round 6: measured on MI355X and frozen at <= 10x like LDA / CTM / CTPF's (round 5 they were literals 25 - 250x looser);
tests/test_mutants_gpu.py holds the negative control (epsilon dropped from update_tau!'s log(beta + eps), src/fLDA.jl:184).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

from tol import within  # noqa: E402  (every comparison below goes through a named tolerance of tests/tol.py, frozen at <= 10x its MI355X measurement)


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: z[k] for k in z.files}


def make_pair(tmvb, oracle, g):
    K, V = int(g["K"]), int(g["V"])
    pc = tmvb.PackedCorpus(g["doc_ptr"], g["terms"], g["counts"], V)
    gm = tmvb.gpufLDA(pc, K)
    gm.beta = np.asfortranarray(g["beta0"]); gm.beta_old = gm.beta.copy(order="F")
    gm.kappa = np.array(g["kappa0"], dtype=np.float64); gm.kappa_old = gm.kappa.copy()
    gm.update_buffer()
    om = oracle.fLDA(oracle.CSR(g["doc_ptr"], g["terms"], g["counts"], V), K, g["beta0"], g["kappa0"])
    return gm, om


def rel(a, b, floor=1e-300):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return (np.abs(a - b) / np.maximum(np.abs(b), floor)).max() if a.size else 0.0


def force(gm, om):
    for n in ("eta", "alpha", "kappa", "kappa_old", "beta", "beta_old", "gamma", "Elogtheta", "Elogtheta_old", "tau", "tau_old"):
        v = getattr(om, n)
        setattr(gm, n, np.array(v, copy=True, order="F") if isinstance(v, np.ndarray) else v)
    gm.update_buffer()


def synth_case(tmvb, K, M=80, V=500, seed=6):
    pc = tmvb.syn_nsf(M=M, V=V, seed=seed)
    return dict(K=K, V=V, doc_ptr=pc.doc_ptr, terms=pc.terms, counts=pc.counts, beta0=tmvb.dirichlet_rows(K, V, seed=5),
                kappa0=tmvb.dirichlet_rows(1, V, seed=9)[0])


@pytest.mark.parametrize("case", ["flda_m40_v60_k5", "flda_m30_v50_k9_empty", "syn_k12", "syn_k50", "syn_k100", "syn_k200", "syn_k300", "syn_k600"])
def test_teacher_forced_fixed_sweeps(tmvb, oracle, case):
    """vtol = 0 pins every document to exactly `viter` sweeps on both sides: pure arithmetic parity of the fused sweep,
    the rebuilt-phi statistics pass, the M-step and the ELBO."""
    g = load(case) if case.startswith("flda_") else synth_case(tmvb, int(case.split("_k")[1]))
    gm, om = make_pair(tmvb, oracle, g)
    for it in range(3):
        force(gm, om)
        gm.estep(viter=3, vtol=0.0); gm.reduce_docs(); gm.mstep()
        om.estep(viter=3, vtol=0.0); om.mstep()
        e_g = gm.update_elbo(); e_o = om.update_elbo()
        gm.update_host()
        within("flda.gamma_rel", rel(gm.gamma, om.gamma), (case, it))
        within("flda.Elogtheta_rel", rel(gm.Elogtheta, om.Elogtheta), (case, it))
        within("flda.tau_abs", np.abs(gm.tau - om.tau).max(initial=0.0), (case, it))
        within("flda.tau_abs", np.abs(gm.tau_old - om.tau_old).max(initial=0.0), (case, it, "tau_old"))
        big = om.beta > 1e-6
        within("flda.beta_rel", rel(gm.beta[big], om.beta[big]), (case, it))
        within("flda.beta_abs", np.abs(gm.beta - om.beta).max(), (case, it))
        bk = om.kappa > 1e-8
        within("flda.kappa_rel", rel(gm.kappa[bk], om.kappa[bk]), (case, it))
        within("flda.alpha_rel", rel(gm.alpha, om.alpha), (case, it))
        within("flda.eta_abs", abs(gm.eta - om.eta), (case, it, gm.eta, om.eta))
        within("flda.elbo_rel_step", abs(e_g - e_o) / abs(e_o), (case, it, e_g, e_o))
        np.testing.assert_allclose(gm.beta.sum(axis=1), 1.0, rtol=1e-5)
        np.testing.assert_allclose(gm.kappa.sum(), 1.0, rtol=1e-5)
        assert np.all((gm.tau >= 0) & (gm.tau <= 1)) and np.all(gm.gamma > 0) and np.all(gm.Elogtheta <= 0)


@pytest.mark.parametrize("case", ["flda_m40_v60_k5", "syn_k12"])
def test_teacher_forced_default_exit_rule(tmvb, oracle, case):
    """Default vtol: documents whose exit sweep agrees with the oracle's are compared one by one; the others counted."""
    g = load(case) if case.startswith("flda_") else synth_case(tmvb, 12)
    gm, om = make_pair(tmvb, oracle, g)
    mism = tot = 0
    for it in range(3):
        force(gm, om)
        gm.estep(); gm.reduce_docs(); gm.mstep()
        sw_o = om.estep(); om.mstep()
        gm.update_host()
        same = gm.doc_sweeps() == np.asarray(sw_o)
        mism += int((~same).sum()); tot += len(sw_o)
        assert same.any()
        within("flda.gamma_rel", rel(gm.gamma[:, same], om.gamma[:, same]), (case, it, "default exit rule"))
    assert mism <= 0.05 * tot, f"{mism}/{tot} documents changed sweep count"


@pytest.mark.parametrize("name", ["flda_m40_v60_k5", "flda_m30_v50_k9_empty"])
def test_free_running_train_vs_golden(tmvb, name):
    g = load(name)
    K, V = int(g["K"]), int(g["V"])
    m = tmvb.fLDA(tmvb.PackedCorpus(g["doc_ptr"], g["terms"], g["counts"], V), K)
    m.beta = np.asfortranarray(g["beta0"]); m.beta_old = m.beta.copy(order="F")
    m.kappa = np.array(g["kappa0"], dtype=np.float64); m.kappa_old = m.kappa.copy()
    traj = tmvb.gpu_train_flda(m, iter=int(g["iters"]), tol=0.0, checkelbo=1, printelbo=False)
    tmvb.check_model_flda(m)
    gold = g["elbo_traj"]
    assert len(traj) == len(gold)
    within("flda.elbo_rel_free", np.abs(traj - gold) / np.abs(gold), (name, traj, gold))
    within("flda.eta_abs_free", abs(m.eta - float(g["eta"])), name)
    within("flda.beta_abs_free", np.abs(m.beta - g["beta"]).max(), name)
    within("flda.kappa_abs_free", np.abs(m.kappa - g["kappa"]).max(), name)
    within("flda.alpha_rel_free", rel(m.alpha, g["alpha"]), name)
    assert sorted(m.topics[0].tolist()) == list(range(1, V + 1))


def test_long_documents_stream_chunks(tmvb, oracle):
    """Documents longer than the LDS window stream their token rows in chunks (tau round-trips through memory)."""
    rng = np.random.default_rng(3)
    V, K, M = 3000, 20, 12
    docs = []
    for d in range(M):
        n = int(rng.integers(300, 900)) if d % 2 == 0 else int(rng.integers(5, 60))
        t = np.sort(rng.choice(V, size=n, replace=False)); c = rng.integers(1, 4, size=n)
        docs.append((t, c))
    doc_ptr = np.concatenate([[0], np.cumsum([len(t) for t, _ in docs])]).astype(np.int64)
    terms = np.concatenate([t for t, _ in docs]).astype(np.int32); counts = np.concatenate([c for _, c in docs]).astype(np.int32)
    g = dict(K=K, V=V, doc_ptr=doc_ptr, terms=terms, counts=counts, beta0=tmvb.dirichlet_rows(K, V, seed=5), kappa0=tmvb.dirichlet_rows(1, V, seed=9)[0])
    gm, om = make_pair(tmvb, oracle, g)
    for it in range(2):
        force(gm, om)
        gm.estep(viter=4, vtol=0.0); gm.reduce_docs(); gm.mstep()
        om.estep(viter=4, vtol=0.0); om.mstep()
        gm.update_host()
        within("flda.gamma_rel", rel(gm.gamma, om.gamma), ("long", it))
        within("flda.tau_abs", np.abs(gm.tau - om.tau).max(), ("long", it))
        big = om.beta > 1e-6
        within("flda.beta_rel", rel(gm.beta[big], om.beta[big]), ("long", it))
        within("flda.eta_abs", abs(gm.eta - om.eta), ("long", it))


def test_errors_and_invariants(tmvb):
    pc = tmvb.syn_nsf(M=200, V=800, seed=2)
    with pytest.raises(ValueError):
        tmvb.gpufLDA(pc, 0)
    with pytest.raises(ValueError):
        tmvb.gpufLDA(pc, 1025)              # K <= 1024, as LDA (round 3; 128 before)
    gm = tmvb.gpufLDA(pc, 16)
    with pytest.raises(ValueError):
        gm.train(viter=-1, printelbo=False)
    e0 = None
    for it in range(3):
        gm.estep(); gm.reduce_docs(); gm.mstep()
        e = gm.update_elbo()
        assert np.isfinite(e) and (e0 is None or e > e0)
        e0 = e
    gm.update_host()
    assert 0.0 < gm.eta < 1.0                                            # topical share of the token mass
    a = gm.alpha.copy()
    gm.estep(); gm.update_host()
    # gamma_d = eps + alpha + phi * counts with stochastic phi columns: sum_i (gamma - alpha) = C_d  (src/fLDA.jl:175)
    np.testing.assert_allclose((gm.gamma - a[:, None]).sum(axis=0), pc.C, rtol=1e-4)


def test_epsilon_keeps_phi_defined_where_a_beta_column_is_zero(tmvb, oracle):
    """update_phi! and update_elbo! take log(@boink beta) (src/fLDA.jl:191, :112, :83): for a term whose beta column is all zero -- a vocabulary entry no
    training document used, met by predict on new text -- every topic has the same log(eps), phi = softmax(Elogtheta) and everything stays finite, where
    log(0) would give -inf - -inf = NaN.  (update_tau! forms beta^-phi from the raw beta, :184: tau of such a token is exactly 0 in the reference and
    ~1e-27 on the device.)  The negative control of tests/test_mutants_gpu.py: the library whose log table has no epsilon must FAIL here."""
    K, V = 5, 7
    rng = np.random.default_rng(3)
    beta0 = rng.random((K, V)); beta0[:, 3] = 0.0; beta0 /= beta0.sum(axis=1, keepdims=True)
    kappa0 = np.full(V, 1.0 / V)
    docs = [([3], [4]), ([0, 3, 5], [2, 1, 3]), ([1, 2, 6], [1, 1, 2])]
    doc_ptr = np.concatenate([[0], np.cumsum([len(t) for t, _ in docs])]).astype(np.int64)
    terms = np.concatenate([t for t, _ in docs]).astype(np.int32); counts = np.concatenate([c for _, c in docs]).astype(np.int32)
    g = dict(K=K, V=V, doc_ptr=doc_ptr, terms=terms, counts=counts, beta0=beta0, kappa0=kappa0)
    gm, om = make_pair(tmvb, oracle, g)
    gm.estep(viter=3, vtol=0.0); gm.reduce_docs(); om.estep(viter=3, vtol=0.0)
    gm.update_host()
    assert np.all(np.isfinite(gm.gamma)) and np.all(np.isfinite(gm.Elogtheta)) and np.all(np.isfinite(gm.tau))
    within("flda.gamma_rel", rel(gm.gamma, om.gamma), "zero column")
    within("flda.tau_abs", np.abs(gm.tau - om.tau).max(), "zero column")
    assert gm.tau[0] <= 1e-20                                                   # the token of the zero column is background
    gm.mstep(); om.mstep(); gm.update_host()
    assert np.all(np.isfinite(gm.beta)) and np.all(np.isfinite(gm.kappa))
    within("flda.beta_abs", np.abs(gm.beta - om.beta).max(), "zero column")
