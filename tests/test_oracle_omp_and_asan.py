"""
CPU tests of the oracle's own plumbing (test infrastructure): the OpenMP document-parallel E-steps (what bench.py's
cpu_baseline times) must agree with the sequential restatement on condensed corpora, and the oracle must run clean under
AddressSanitizer / UBSan (`make -C oracle asan`) on one golden case per model.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _nsf(tmvb, M=120, V=500, seed=5):
    pc = tmvb.syn_nsf(M=M, V=V, seed=seed)
    return pc


def test_omp_estep_matches_sequential_lda_ctm(tmvb, oracle):
    pc = _nsf(tmvb)
    K = 6
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=3)
    csr = oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V)
    a, b = oracle.LDA(csr, K, beta0), oracle.LDA(csr, K, beta0)
    for _ in range(2):
        sa = a.estep(); a.update_beta(); a.update_alpha()
        sb = b.estep(omp_threads=3); b.update_beta(); b.update_alpha()
        assert np.array_equal(sa, sb)                       # per-document sweep counts (the full-size parity checks use them)
    np.testing.assert_allclose(a.gamma, b.gamma, rtol=1e-12)
    np.testing.assert_allclose(a.beta, b.beta, rtol=1e-11, atol=1e-300)
    a, b = oracle.CTM(csr, K, beta0), oracle.CTM(csr, K, beta0)
    for _ in range(2):
        sa = a.estep(); a.update_beta(); a.update_sigma_mu()
        sb = b.estep(omp_threads=3); b.update_beta(); b.update_sigma_mu()
        assert np.array_equal(sa, sb) and a.newton_steps == b.newton_steps == int(b.newton_per_doc.sum())
    np.testing.assert_allclose(a.lam, b.lam, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a.beta, b.beta, rtol=1e-9, atol=1e-300)


def test_omp_estep_matches_sequential_filtered(tmvb, oracle):
    pc = _nsf(tmvb, M=80, V=300, seed=8)
    K = 5
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=3); kappa0 = tmvb.dirichlet_rows(1, pc.V, seed=9)[0]
    csr = oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V)
    a, b = oracle.fLDA(csr, K, beta0, kappa0), oracle.fLDA(csr, K, beta0, kappa0)
    for _ in range(2):
        a.estep(); a.mstep()
        b.estep(omp_threads=3); b.mstep()
    np.testing.assert_allclose(a.tau, b.tau, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(a.beta, b.beta, rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(a.kappa, b.kappa, rtol=1e-10, atol=1e-300)
    assert abs(a.eta - b.eta) <= 1e-13
    a, b = oracle.fCTM(csr, K, beta0, kappa0), oracle.fCTM(csr, K, beta0, kappa0)
    for _ in range(2):
        a.estep(); a.mstep()
        b.estep(omp_threads=3); b.mstep()
    np.testing.assert_allclose(a.tau, b.tau, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(a.lam, b.lam, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a.kappa, b.kappa, rtol=1e-9, atol=1e-300)


ASAN_SCRIPT = r'''
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as oc
oc._LIB_PATH = os.path.join(sys.argv[1], "oracle", "libtmvb_oracle_asan.so")
oc._lib = None
G = os.path.join(sys.argv[1], "tests", "golden")
def load(n):
    z = np.load(os.path.join(G, n + ".npz")); return {k: z[k] for k in z.files}
g = load("lda_m30_v50_k70_empty")
m = oc.LDA(oc.CSR(g["doc_ptr"], g["terms"], g["counts"], int(g["V"])), int(g["K"]), g["beta0"])
t = m.train(iter=3, tol=0.0, checkelbo=1); assert np.all(np.isfinite(t))
m.estep(omp_threads=2); m.update_beta()
g = load("ctm_m40_v60_k5")
m = oc.CTM(oc.CSR(g["doc_ptr"], g["terms"], g["counts"], int(g["V"])), int(g["K"]), g["beta0"])
t = m.train(iter=2, tol=0.0, checkelbo=1); assert np.all(np.isfinite(t))
m.estep(omp_threads=2)
g = load("ctpf_m30_v40_u12_k6_r1")
m = oc.CTPF(oc.CSR(g["doc_ptr"], g["terms"], g["counts"], int(g["V"]), g["rdr_ptr"], g["readers"], g["ratings"], int(g["U"])), int(g["K"]), g["alef0"])
t = m.train(iter=2, tol=0.0, checkelbo=1); assert np.all(np.isfinite(t))
m.estep(omp_threads=2)
g = load("flda_m30_v50_k9_empty")
m = oc.fLDA(oc.CSR(g["doc_ptr"], g["terms"], g["counts"], int(g["V"])), int(g["K"]), g["beta0"], g["kappa0"])
t = m.train(iter=2, tol=0.0, checkelbo=1); assert np.all(np.isfinite(t))
m.estep(omp_threads=2)
g = load("fctm_m30_v50_k4")
m = oc.fCTM(oc.CSR(g["doc_ptr"], g["terms"], g["counts"], int(g["V"])), int(g["K"]), g["beta0"], g["kappa0"])
t = m.train(iter=2, tol=0.0, checkelbo=1); assert np.all(np.isfinite(t))
m.estep(omp_threads=2)
print("asan-ok")
'''


def _libasan():
    try:
        out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        return None
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_oracle_runs_clean_under_asan_ubsan(tmp_path):
    """`make -C oracle asan` builds the oracle with -fsanitize=address,undefined; one golden case per model (train! +
    the OpenMP E-step) must finish without a sanitizer report."""
    asan = _libasan()
    if asan is None:
        pytest.skip("gcc has no libasan.so in this image")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    script = tmp_path / "run_asan.py"
    script.write_text(ASAN_SCRIPT)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0 and "asan-ok" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-4000:]
