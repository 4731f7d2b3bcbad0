"""C oracle vs the committed golden fixtures (made by the independent NumPy restatement)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: z[k] for k in z.files}


def csr(oracle, g, U=0):
    return oracle.CSR(g["doc_ptr"], g["terms"], g["counts"], int(g["V"]), g["rdr_ptr"], g["readers"], g["ratings"], U)


@pytest.mark.parametrize("name", ["lda_m40_v60_k3", "lda_m40_v60_k7", "lda_m30_v50_k70_empty"])
def test_lda_golden(oracle, name):
    g = load(name)
    m = oracle.LDA(csr(oracle, g), int(g["K"]), g["beta0"])
    traj = m.train(iter=int(g["iters"]), tol=-1e300)
    np.testing.assert_allclose(traj, g["elbo_traj"], rtol=1e-11)
    np.testing.assert_allclose(m.alpha, g["alpha"], rtol=1e-10)
    np.testing.assert_allclose(m.beta, g["beta"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(m.gamma, g["gamma"], rtol=1e-10)
    np.testing.assert_allclose(m.Elogtheta, g["Elogtheta"], rtol=1e-10)
    np.testing.assert_allclose(m.Elogtheta_old, g["Elogtheta_old"], rtol=1e-10)
    hist = np.bincount(g["sweeps"].ravel(), minlength=11)
    assert np.array_equal(hist, m.sweep_hist)


@pytest.mark.parametrize("name", ["flda_m40_v60_k5", "flda_m30_v50_k9_empty"])
def test_flda_golden(oracle, name):
    """filtered LDA (src/fLDA.jl): the C oracle against the fixture of the independent NumPy restatement"""
    g = load(name)
    m = oracle.fLDA(oracle.CSR(g["doc_ptr"], g["terms"], g["counts"], int(g["V"])), int(g["K"]), g["beta0"], g["kappa0"])
    traj = m.train(iter=int(g["iters"]), tol=-1e300)
    np.testing.assert_allclose(traj, g["elbo_traj"], rtol=1e-11)
    np.testing.assert_allclose(m.eta, float(g["eta"]), rtol=1e-12)
    np.testing.assert_allclose(m.alpha, g["alpha"], rtol=1e-10)
    np.testing.assert_allclose(m.kappa, g["kappa"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(m.beta, g["beta"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(m.gamma, g["gamma"], rtol=1e-10)
    np.testing.assert_allclose(m.tau, g["tau"], rtol=1e-10)
    np.testing.assert_allclose(m.tau_old, g["tau_old"], rtol=1e-10)
    assert np.array_equal(np.bincount(g["sweeps"][-1].ravel(), minlength=11), m.sweep_hist)


def test_fctm_golden(oracle):
    """filtered CTM (src/fCTM.jl): the C oracle against the fixture of the independent NumPy restatement"""
    g = load("fctm_m30_v50_k4")
    m = oracle.fCTM(oracle.CSR(g["doc_ptr"], g["terms"], g["counts"], int(g["V"])), int(g["K"]), g["beta0"], g["kappa0"])
    traj = m.train(iter=int(g["iters"]), tol=-1e300)
    np.testing.assert_allclose(traj, g["elbo_traj"], rtol=1e-10)
    np.testing.assert_allclose(m.lam, g["lam"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(m.vsq, g["vsq"], rtol=1e-9)
    np.testing.assert_allclose(m.tau, g["tau"], rtol=1e-9)
    np.testing.assert_allclose(m.kappa, g["kappa"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(m.beta, g["beta"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(m.mu, g["mu"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(m.sigma, g["sigma"], rtol=1e-9, atol=1e-12)
    assert m.eta == 0.5                                   # update_eta! is commented out in the reference's train!


def test_ctm_golden(oracle):
    g = load("ctm_m40_v60_k5")
    m = oracle.CTM(csr(oracle, g), int(g["K"]), g["beta0"])
    traj = m.train(iter=int(g["iters"]), tol=-1e300)
    np.testing.assert_allclose(traj, g["elbo_traj"], rtol=1e-10)
    np.testing.assert_allclose(m.lam, g["lam"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(m.vsq, g["vsq"], rtol=1e-9)
    np.testing.assert_allclose(m.mu, g["mu"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(m.sigma, g["sigma"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(m.invsigma, g["invsigma"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(m.beta, g["beta"], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("name", ["ctpf_m40_v60_u15_k4", "ctpf_m30_v40_u12_k6_r1"])
def test_ctpf_golden(oracle, name):
    g = load(name)
    m = oracle.CTPF(csr(oracle, g, int(g["U"])), int(g["K"]), g["alef0"])
    traj = m.train(iter=int(g["iters"]), tol=-1e300)
    np.testing.assert_allclose(traj, g["elbo_traj"], rtol=1e-10)
    for f in ("alef", "he", "bet", "vav", "dalet", "het", "gimel", "zayin"):
        np.testing.assert_allclose(getattr(m, f), g[f], rtol=1e-9, err_msg=f)
