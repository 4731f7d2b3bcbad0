"""predict / topicdist (src/modelutils.jl:831-855, :886-913, :946-958) on the HIP engine
vs the oracle's E-step with the same frozen globals."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tol import within  # noqa: E402  (named tolerances of tests/tol.py, frozen at <= 10x their MI355X measurement; q90 over the documents and the maximum,
#                                       which a document whose exit sweep flips against the oracle's sets)


def test_lda_predict_matches_oracle_estep(tmvb, oracle):
    K = 12
    train = tmvb.syn_nsf(M=200, V=800, seed=21)
    new = tmvb.syn_nsf(M=60, V=800, seed=22)
    m = tmvb.LDA(train, K)
    tmvb.gpu_train(m, iter=5, tol=0.0, checkelbo=float("inf"), printelbo=False)
    p = tmvb.predict(new, m, iter=10)
    om = oracle.LDA(oracle.CSR(new.doc_ptr, new.terms, new.counts, new.V), K, m.beta)
    om.alpha[:] = m.alpha
    sw = om.estep(viter=10)
    assert p.gamma.shape == (K, new.M)
    same = np.ones(new.M, bool)          # compare documents tightly; sweep-count flips are rare and bounded below
    r = np.abs(p.gamma - om.gamma) / om.gamma
    within("predict.lda.gamma_rel_q90", np.quantile(r.max(axis=0), 0.9)); within("predict.lda.gamma_rel_max", r.max())
    td = tmvb.topicdist(p, 1)
    np.testing.assert_allclose(td, p.gamma[:, 0] / p.gamma[:, 0].sum())
    assert abs(td.sum() - 1.0) < 1e-12 and len(tmvb.topicdist(p, range(1, 4))) == 3
    with pytest.raises(tmvb.CorpusError):
        tmvb.topicdist(p, new.M + 1)
    with pytest.raises(tmvb.CorpusError):
        tmvb.predict(tmvb.syn_nsf(M=10, V=700, seed=1), m)      # different vocabulary


def test_ctm_predict_matches_oracle_estep(tmvb, oracle):
    K = 12
    # CTM's phi has no epsilon (src/CTM.jl:177): a held-out term never seen in training has beta = 0 for every topic
    # and log(0) poisons the document in the reference too -- keep the vocabulary small enough to be covered
    train = tmvb.syn_nsf(M=300, V=120, seed=31)
    new = tmvb.syn_nsf(M=40, V=120, seed=32)
    m = tmvb.CTM(train, K)
    tmvb.gpu_train_ctm(m, iter=3, tol=0.0, checkelbo=float("inf"), printelbo=False)
    p = tmvb.predict_ctm(new, m, iter=10)
    om = oracle.CTM(oracle.CSR(new.doc_ptr, new.terms, new.counts, new.V), K, m.beta)
    om.mu[:] = m.mu; om.sigma[:] = m.sigma; om.invsigma[:] = m.invsigma
    om.estep(viter=10)
    assert np.array_equal(np.isnan(p.lam), np.isnan(om.lam))          # same (absent) NaN pattern as the oracle
    ok = ~np.isnan(om.lam).any(axis=0)
    assert ok.sum() >= 30
    within("predict.ctm.lambda_abs_q90", np.quantile(np.abs(p.lam - om.lam)[:, ok].max(axis=0), 0.9))
    within("predict.ctm.lambda_abs_max", np.abs(p.lam - om.lam)[:, ok].max())
    td = tmvb.topicdist_ctm(p, 2)
    assert abs(td.sum() - 1.0) < 1e-12 and np.all(td > 0)


def test_flda_predict_matches_oracle_estep(tmvb, oracle):
    """src/modelutils.jl:858-883: alpha / beta from the trained model, kappa / eta / tau of the fresh fLDA(corp, K)."""
    K = 9
    train = tmvb.syn_nsf(M=200, V=600, seed=41)
    new = tmvb.syn_nsf(M=50, V=600, seed=42)
    m = tmvb.fLDA(train, K)
    tmvb.gpu_train_flda(m, iter=4, tol=0.0, checkelbo=float("inf"), printelbo=False)
    p = tmvb.predict_flda(new, m, iter=10, seed=13)
    fresh = tmvb.fLDA(new, K, 13)
    assert p.eta == 0.5 and np.array_equal(p.kappa, fresh.kappa)            # not the trained model's
    assert np.array_equal(p.beta, m.beta) and np.array_equal(p.alpha, m.alpha)
    om = oracle.fLDA(oracle.CSR(new.doc_ptr, new.terms, new.counts, new.V), K, m.beta, fresh.kappa)
    om.alpha[:] = m.alpha
    sw = om.estep(viter=10)
    assert p.gamma.shape == (K, new.M) and p.tau.shape == (new.nnz,)
    r = np.abs(p.gamma - om.gamma) / om.gamma
    within("predict.flda.gamma_rel_q90", np.quantile(r.max(axis=0), 0.9)); within("predict.flda.gamma_rel_max", r.max())
    within("predict.flda.tau_abs_q99", np.quantile(np.abs(p.tau - om.tau), 0.99)); assert np.all((p.tau >= 0) & (p.tau <= 1))
    td = tmvb.topicdist(p, 3)
    np.testing.assert_allclose(td, p.gamma[:, 2] / p.gamma[:, 2].sum())
    with pytest.raises(ValueError):
        tmvb.predict_flda(new, m, iter=-1)
    with pytest.raises(tmvb.CorpusError):
        tmvb.predict_flda(tmvb.syn_nsf(M=10, V=500, seed=1), m)


def test_fctm_predict_matches_oracle_estep(tmvb, oracle):
    """src/modelutils.jl:916-943: mu / sigma / invsigma / beta from the trained model, kappa / eta / tau fresh."""
    K = 10
    train = tmvb.syn_nsf(M=300, V=120, seed=51)
    new = tmvb.syn_nsf(M=40, V=120, seed=52)
    m = tmvb.fCTM(train, K)
    tmvb.gpu_train_fctm(m, iter=3, tol=0.0, checkelbo=float("inf"), printelbo=False)
    p = tmvb.predict_fctm(new, m, iter=10, seed=17)
    fresh = tmvb.fCTM(new, K, 17)
    assert p.eta == 0.5 and np.array_equal(p.kappa, fresh.kappa)
    om = oracle.fCTM(oracle.CSR(new.doc_ptr, new.terms, new.counts, new.V), K, m.beta, fresh.kappa)
    om.mu[:] = m.mu; om.sigma[:] = m.sigma; om.invsigma[:] = m.invsigma
    om.estep(viter=10)
    assert np.all(np.isfinite(p.lam)) and np.all(np.isfinite(om.lam))       # the kappa mixture keeps every token's phi finite
    within("predict.fctm.lambda_abs_q90", np.quantile(np.abs(p.lam - om.lam).max(axis=0), 0.9))
    within("predict.fctm.lambda_abs_max", np.abs(p.lam - om.lam).max())
    within("predict.fctm.tau_abs_q99", np.quantile(np.abs(p.tau - om.tau), 0.99))
    td = tmvb.topicdist_ctm(p, 2)
    assert abs(td.sum() - 1.0) < 1e-12 and np.all(td > 0)
    with pytest.raises(ValueError):
        tmvb.predict_fctm(new, m, ntol=-1.0)


def test_ctpf_topicdist(tmvb):
    """src/modelutils.jl:960-965."""
    pc = tmvb.syn_citeu(M=80, V=300, U=60, seed=3)
    g = tmvb.gpuCTPF(pc, 6)
    g.train(iter=3, tol=0.0, checkelbo=float("inf"), printelbo=False, recs=False)
    td = tmvb.topicdist_ctpf(g, 5)
    np.testing.assert_allclose(td, g.gimel[:, 4] / g.gimel[:, 4].sum())
    assert abs(td.sum() - 1.0) < 1e-12 and len(tmvb.topicdist_ctpf(g, [1, 2, 80])) == 3
    with pytest.raises(tmvb.CorpusError):
        tmvb.topicdist_ctpf(g, 81)
