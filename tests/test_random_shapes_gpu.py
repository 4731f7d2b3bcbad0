"""
Randomised shape sweep: teacher-forced LDA / CTM / CTPF steps on small random corpora whose shapes hit the corner
cases together -- empty documents, one-token documents, documents of several hundred terms (multi-tile and LDS-tile
kernels), vocabularies / user sets smaller than K, unused terms, K from 1 to 100 -- each compared with the fp64
oracle at the tolerances of the per-model test files.  Seeds are fixed: failures reproduce.
"""
import numpy as np
import pytest

from tol import within

pytestmark = pytest.mark.gpu


def random_corpus(rng, M, V, U=0, long_docs=0):
    doc_ptr, terms, counts = [0], [], []
    rdr_ptr, readers, ratings = [0], [], []
    for d in range(M):
        kind = rng.integers(0, 10)
        n = 0 if kind == 0 else 1 if kind == 1 else int(min(V, rng.integers(2, 90)))
        if d < long_docs:
            n = int(min(V, rng.integers(140, 420)))
        t = np.sort(rng.choice(V, size=n, replace=False))
        terms += t.tolist(); counts += rng.integers(1, 6, size=n).tolist()
        doc_ptr.append(len(terms))
        if U:
            r = int(min(U, rng.integers(0, 8))) if d >= long_docs else int(min(U, rng.integers(60, 100)))
            u = np.sort(rng.choice(U, size=r, replace=False))
            readers += u.tolist(); ratings += rng.integers(1, 4, size=r).tolist()
        rdr_ptr.append(len(readers))
    a = lambda x, dt: np.asarray(x, dtype=dt)
    return dict(doc_ptr=a(doc_ptr, np.int64), terms=a(terms, np.int32), counts=a(counts, np.int32),
                rdr_ptr=a(rdr_ptr, np.int64), readers=a(readers, np.int32), ratings=a(ratings, np.int32))


def rel(a, b, floor=1e-300):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return (np.abs(a - b) / np.maximum(np.abs(b), floor)).max() if a.size else 0.0


@pytest.mark.parametrize("seed", range(12))
def test_lda_random_shapes(tmvb, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.choice([1, 2, 5, 11, 20, 37, 50, 64, 77, 100]))
    M, V = int(rng.integers(1, 40)), int(rng.integers(1, 500))
    c = random_corpus(rng, M, V, long_docs=int(rng.integers(0, 3)))
    if c["terms"].size == 0:
        c = random_corpus(np.random.default_rng(seed), 5, max(V, 3))
        M = 5
    beta0 = tmvb.dirichlet_rows(K, V, seed=seed)
    gm = tmvb.gpuLDA(tmvb.PackedCorpus(c["doc_ptr"], c["terms"], c["counts"], V), K)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
    om = oracle.LDA(oracle.CSR(c["doc_ptr"], c["terms"], c["counts"], V), K, beta0)
    for it in range(2):
        gm.alpha = om.alpha.copy(); gm.beta = om.beta.copy(order="F"); gm.beta_old = om.beta_old.copy(order="F")
        gm.gamma = om.gamma.copy(order="F"); gm.Elogtheta = om.Elogtheta.copy(order="F")
        gm.Elogtheta_old = om.Elogtheta_old.copy(order="F"); gm.update_buffer()
        gm.estep(viter=3, vtol=0.0); gm.reduce_docs(); om.estep(viter=3, vtol=0.0)
        gm.update_beta(); om.update_beta(); gm.update_alpha(); om.update_alpha()
        e_g, e_o = gm.update_elbo(), om.update_elbo()
        gm.update_host()
        within("rand.lda.gamma_rel", rel(gm.gamma, om.gamma), (K, M, V, it))
        used = om.beta > 1e-6
        within("rand.lda.beta_rel", rel(gm.beta[used], om.beta[used]), (K, M, V, it))
        within("rand.lda.alpha_rel", rel(gm.alpha, om.alpha), (K, M, V, it))
        within("rand.lda.elbo_rel", abs(e_g - e_o) / abs(e_o), (K, M, V, it))


@pytest.mark.parametrize("seed", range(8))
def test_ctm_random_shapes(tmvb, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    K = int(rng.choice([1, 3, 8, 14, 27, 40, 50, 60]))
    M, V = int(rng.integers(2, 30)), int(rng.integers(60, 400))
    c = random_corpus(rng, M, V, long_docs=int(rng.integers(0, 2)))
    # CTM has no epsilon: every term of the vocabulary must occur (beta > 0) or log beta = -inf enters the softmax
    seen = np.zeros(V, bool); seen[c["terms"]] = True
    extra = np.flatnonzero(~seen)
    c["terms"] = np.concatenate([c["terms"], extra.astype(np.int32)]); c["counts"] = np.concatenate([c["counts"], np.ones(extra.size, np.int32)])
    c["doc_ptr"] = np.concatenate([c["doc_ptr"], [c["terms"].size]])
    M += 1
    beta0 = tmvb.dirichlet_rows(K, V, seed=seed)
    gm = tmvb.gpuCTM(tmvb.PackedCorpus(c["doc_ptr"], c["terms"], c["counts"], V), K)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
    om = oracle.CTM(oracle.CSR(c["doc_ptr"], c["terms"], c["counts"], V), K, beta0)
    for it in range(2):
        for n in ("mu", "logzeta"):
            setattr(gm, n, getattr(om, n).copy())
        for n in ("sigma", "invsigma", "beta", "beta_old", "lam", "lam_old", "vsq"):
            setattr(gm, n, getattr(om, n).copy(order="F"))
        gm.update_buffer()
        gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_sigma(); gm.update_mu()
        om.estep(); om.update_beta(); om.update_sigma_mu()
        gm.update_host()
        within("rand.ctm.lambda_err", np.abs(gm.lam - om.lam).max() / (1.0 + np.abs(om.lam).max()), (K, M, V, it))
        within("rand.ctm.vsq_rel", rel(gm.vsq, om.vsq), (K, M, V, it))
        within("rand.ctm.mu_abs", np.abs(gm.mu - om.mu), (K, M, V, it))
        within("rand.ctm.sigma_err", np.abs(gm.sigma - om.sigma).max() / max(1.0, np.abs(om.sigma).max()), (K, M, V, it))


@pytest.mark.parametrize("seed", range(8))
def test_ctpf_random_shapes(tmvb, oracle, seed):
    rng = np.random.default_rng(3000 + seed)
    K = int(rng.choice([1, 4, 9, 22, 35, 50, 60, 64]))
    M, V, U = int(rng.integers(2, 40)), int(rng.integers(5, 500)), int(rng.integers(1, 120))
    c = random_corpus(rng, M, V, U=U, long_docs=int(rng.integers(0, 3)))
    alef0 = np.exp(tmvb.dirichlet_rows(K, V, seed=seed) - 0.5)
    pc = tmvb.PackedCorpus(c["doc_ptr"], c["terms"], c["counts"], V, c["rdr_ptr"], c["readers"], c["ratings"], U)
    gm = tmvb.gpuCTPF(pc, K)
    gm.alef = np.asfortranarray(alef0); gm.alef_old = gm.alef.copy(order="F"); gm.update_buffer()
    om = oracle.CTPF(oracle.CSR(c["doc_ptr"], c["terms"], c["counts"], V, c["rdr_ptr"], c["readers"], c["ratings"], U), K, alef0)
    for it in range(2):
        for n in ("alef", "he", "bet", "vav", "dalet", "het", "gimel", "zayin"):
            setattr(gm, n, np.array(getattr(om, n), copy=True, order="F"))
        gm.update_buffer()
        gm.estep(viter=3, vtol=0.0); gm.reduce_docs(); gm.mstep()
        om.estep(viter=3, vtol=0.0); om.mstep()
        gm.update_host()
        for n in ("gimel", "zayin", "alef", "he"):
            within("rand.ctpf.shape_rel", rel(getattr(gm, n), getattr(om, n)), (K, M, V, U, it, n))
        for n in ("bet", "vav", "dalet", "het"):
            within("rand.ctpf.rates_rel", rel(getattr(gm, n), getattr(om, n)), (K, M, V, U, it, n))


def test_empty_corpora(tmvb):
    """gpuLDA(Corpus(), 1) must be constructible (the @gpu macro builds its device model on an empty corpus,
    src/macros.jl:114) and train! on a corpus whose documents are all empty runs zero iterations (src/LDA.jl:166)."""
    empty = tmvb.PackedCorpus(np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32), 0)
    blank = tmvb.PackedCorpus(np.zeros(4, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32), 7)
    for cls in (tmvb.gpuLDA, tmvb.gpuCTM, tmvb.gpuCTPF):
        m = cls(empty, 1)
        assert len(m.train(iter=2, checkelbo=1, printelbo=False)) == 0
        m = cls(blank, 3)
        assert len(m.train(iter=2, checkelbo=1, printelbo=False)) == 0
