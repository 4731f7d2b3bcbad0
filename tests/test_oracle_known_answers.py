"""Closed-form known-answer tests and invariants for the oracle."""
import numpy as np

EPS = 2.0 ** -99


def _corp(oracle, docs, V, U=0):
    return oracle.CSR.from_docs(docs, V, U)


def test_lda_k1_closed_form(oracle):
    # K=1: phi == 1, gamma_d = alpha + C_d + eps, beta = empirical unigram distribution.
    rng = np.random.default_rng(0)
    V, M = 9, 12
    docs = []
    for _ in range(M):
        t = np.sort(rng.choice(V, size=rng.integers(1, 6), replace=False)); c = rng.integers(1, 5, size=len(t))
        docs.append((t.tolist(), c.tolist()))
    m = oracle.LDA(_corp(oracle, docs, V), 1, np.full((1, V), 1.0 / V))
    m.estep(viter=10)
    C = np.array([sum(c) for _, c in docs], dtype=float)
    np.testing.assert_allclose(m.gamma[0], 1.0 + C + EPS, rtol=1e-15)
    bt = m.beta_temp.copy()
    m.update_beta()
    emp = np.zeros(V)
    for t, c in docs:
        emp[t] += c
    np.testing.assert_allclose(bt[0], emp, rtol=1e-14)
    np.testing.assert_allclose(m.beta[0], emp / emp.sum(), rtol=1e-14)
    assert np.all(m.beta_temp == 0)


def test_lda_empty_doc_inside_corpus(oracle):
    docs = [([0, 2], [1, 2]), ([], []), ([1], [3])]
    K, V = 4, 3
    rng = np.random.default_rng(1)
    b = rng.exponential(size=(K, V)); b /= b.sum(1, keepdims=True)
    m = oracle.LDA(_corp(oracle, docs, V), K, b)
    sw = m.estep(viter=10)
    np.testing.assert_allclose(m.gamma[:, 1], m.alpha + EPS, rtol=1e-15)   # gamma_d = alpha + eps
    assert sw[1] >= 1


def test_lda_invariants_and_monotone_elbo_fixed_alpha(oracle):
    rng = np.random.default_rng(5)
    V, M, K = 30, 25, 4
    docs = []
    for _ in range(M):
        t = np.sort(rng.choice(V, size=rng.integers(2, 10), replace=False)); c = rng.integers(1, 4, size=len(t))
        docs.append((t.tolist(), c.tolist()))
    b = rng.exponential(size=(K, V)); b /= b.sum(1, keepdims=True)
    m = oracle.LDA(_corp(oracle, docs, V), K, b)
    C = np.array([sum(c) for _, c in docs], dtype=float)
    prev = -np.inf
    for _ in range(6):
        # viter large + vtol=0 => the E-step is (numerically) a full coordinate maximisation
        m.estep(viter=200, vtol=0.0)
        # sum_i (gamma_id - alpha_i - eps) = C_d   (phi columns sum to one)
        np.testing.assert_allclose((m.gamma - m.alpha[:, None] - EPS).sum(0), C, rtol=1e-12)
        m.update_beta()
        np.testing.assert_allclose(m.beta.sum(1), 1.0, rtol=1e-13)        # check_model: right-stochastic
        assert np.all(m.gamma > 0) and np.all(m.Elogtheta <= 0)
        e = m.update_elbo()
        assert np.isfinite(e)
        assert e >= prev - 1e-8 * abs(e)                                   # coordinate ascent, alpha fixed
        prev = e


def test_lda_duplicate_term_overwrite_quirk_q1(oracle):
    # beta_temp[:,terms] += phi .* counts' : with a repeated id the LAST column wins (src/LDA.jl:131)
    docs = [([1, 1], [2, 5])]
    K, V = 2, 3
    b = np.array([[0.2, 0.5, 0.3], [0.6, 0.1, 0.3]])
    m = oracle.LDA(_corp(oracle, docs, V), K, b)
    m.estep(viter=1)
    # both tokens have identical phi (same term) -> column 1 holds phi*5, not phi*(2+5)
    colsum = m.beta_temp[:, 1].sum()
    np.testing.assert_allclose(colsum, 5.0, rtol=1e-14)


def test_signed_stop_rule_q4(oracle):
    # check_elbo! stops when (elbo_new - elbo_old) < tol, signed (src/modelutils.jl:580)
    rng = np.random.default_rng(7)
    V, M, K = 20, 15, 3
    docs = []
    for _ in range(M):
        t = np.sort(rng.choice(V, size=rng.integers(2, 8), replace=False)); c = rng.integers(1, 4, size=len(t))
        docs.append((t.tolist(), c.tolist()))
    b = rng.exponential(size=(K, V)); b /= b.sum(1, keepdims=True)
    m = oracle.LDA(_corp(oracle, docs, V), K, b)
    traj = m.train(iter=50, tol=1e9)          # any finite improvement is < tol -> stops after 1 iteration
    assert len(traj) == 1
    m2 = oracle.LDA(_corp(oracle, docs, V), K, b)
    traj2 = m2.train(iter=8, tol=-1e300)      # never stops early
    assert len(traj2) == 8


def test_ctm_first_sweep_matches_hand_computation(oracle):
    # single doc, K=2, identity sigma, lambda=0, vsq=1: logzeta = log(2) + 0.5
    docs = [([0, 1], [1, 1])]
    b = np.array([[0.5, 0.5], [0.5, 0.5]])
    m = oracle.CTM(_corp(oracle, docs, 2), 2, b)
    m.estep(niter=0, viter=1)                 # niter=0: vsq/lambda Newton loops do not run
    np.testing.assert_allclose(m.logzeta[0], np.log(2.0) + 0.5, rtol=1e-15)
    np.testing.assert_allclose(m.beta_temp, np.full((2, 2), 0.5), rtol=1e-15)


def test_ctpf_mstep_order(oracle):
    # dalet/het use the OLD bet/vav; bet/vav use the NEW dalet/het (src/CTPF.jl:366-371)
    docs = [([0], [1], [0], [1])]
    K, V, U = 2, 1, 1
    m = oracle.CTPF(_corp(oracle, docs, V, U), K, np.ones((K, V)))
    m.alef_temp[:] = np.array([[2.0], [3.0]]); m.he_temp[:] = np.array([[4.0], [5.0]])
    m.bet[:] = [2.0, 4.0]; m.vav[:] = [5.0, 10.0]
    gs = np.array([1.0, 2.0]); zs = np.array([3.0, 4.0])
    m.mstep(gs, zs)
    dalet = 0.1 + np.array([2.0, 3.0]) / [2.0, 4.0] + np.array([4.0, 5.0]) / [5.0, 10.0]
    het = 0.1 + np.array([4.0, 5.0]) / [5.0, 10.0]
    np.testing.assert_allclose(m.dalet, dalet, rtol=1e-15)
    np.testing.assert_allclose(m.het, het, rtol=1e-15)
    np.testing.assert_allclose(m.bet, 0.1 + gs / dalet, rtol=1e-15)
    np.testing.assert_allclose(m.vav, 0.1 + gs / dalet + zs / het, rtol=1e-15)
    assert np.all(m.alef_temp == 0.1) and np.all(m.he_temp == 0.1)
