"""
CPU test of oracle/parity.py -- the teacher-forced, every-document comparison that bench.py / tools/model_bench.py / the full-size
`-m gpu` tests run against the HIP engine.  Here a second oracle stands in for the device (tests may use the oracle), with an exit
threshold that is deliberately a little different, so that some documents leave one sweep apart: the fix-up (oracle re-run with
the stand-in's sweep count for exactly those documents, their statistics taken out and put back) must bring every document and
every global back to fp64 agreement, and a stand-in that really is wrong must fail.
"""
import numpy as np
import pytest

from oracle import parity


class _Stand:
    """the slice of the gpu* model interface parity.py uses, on top of an oracle model"""

    def __init__(self, m, fields, vtol_scale=1.0, spoil=0.0):
        self.m, self.fields, self.vtol_scale, self.spoil = m, fields, vtol_scale, spoil
        self.K, self.M = m.K, m.M
        self.update_host()

    def update_buffer(self):
        for n in self.fields:
            getattr(self.m, "lam" if n == "lam" else n)[...] = getattr(self, n)

    def update_host(self):
        for n in self.fields:
            setattr(self, n, np.array(getattr(self.m, n), copy=True, order="F"))

    def reduce_docs(self): pass
    def synchronize(self): pass
    def doc_sweeps(self): return np.asarray(self.sw, dtype=np.uint8)
    def update_elbo(self): return self.m.update_elbo()


class StandLDA(_Stand):
    def __init__(self, m, **kw): super().__init__(m, ("alpha", "beta", "beta_old", "gamma", "Elogtheta", "Elogtheta_old"), **kw)

    def estep(self, viter=10, vtol=None):
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        self.sw = self.m.estep(viter, vtol * self.vtol_scale)
        self.m.gamma *= 1.0 + self.spoil

    def update_beta(self): self.m.update_beta()
    def update_alpha(self, niter=1000, ntol=None): self.m.update_alpha(niter, ntol)


class StandCTM(_Stand):
    def __init__(self, m, **kw): super().__init__(m, ("mu", "sigma", "invsigma", "beta", "beta_old", "lam", "lam_old", "vsq", "logzeta"), **kw)

    def estep(self):
        self.sw = self.m.estep(vtol=self.vtol_scale / self.K ** 2)

    def update_beta(self): self.m.update_beta()
    def update_sigma(self): self.m.update_sigma_mu()
    def update_mu(self): pass


class StandCTPF(_Stand):
    def __init__(self, m, **kw):
        super().__init__(m, parity.CTPF_FIELDS, **kw)

    def estep(self):
        self.sw = self.m.estep(vtol=self.vtol_scale / self.K ** 2)

    def mstep(self): self.m.mstep()


def _lda_pair(tmvb, oracle, K=7, **kw):
    pc = tmvb.syn_nsf(M=300, V=400, seed=11)
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=3)
    csr = oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V)
    return StandLDA(oracle.LDA(csr, K, beta0), **kw), oracle.LDA(csr, K, beta0)


def test_lda_sweep_mismatches_are_counted_and_fixed_up(tmvb, oracle):
    gm, om = _lda_pair(tmvb, oracle, vtol_scale=1.005)
    block, secs = parity.lda_parity(gm, om, iters=4, threads=2)
    # (the stand-in's mismatch rate is far above the frozen bound of a real device, 5e-3: everything else must hold)
    assert all(v <= parity.LDA_TOL[k] for k, v in block["worst"].items() if v is not None and k != "sweep_mismatch_frac"), block["worst"]
    assert 0 < block["worst"]["sweep_mismatch_frac"] <= 0.05            # the stand-in did leave some documents a sweep apart
    # after the fix-up both sides ran the same sweeps on every document: fp64 agreement (summation order only)
    assert block["worst"]["gamma_rel_p999"] <= 1e-12 and block["worst"]["beta_rel_max"] <= 1e-10 and block["worst"]["elbo_rel"] <= 1e-13
    assert len(secs) == 4 and all(s > 0 for s in secs)
    assert set(block["tolerances"]) <= set(block["per_iteration"][0]) | {"elbo_rel"}


def test_lda_wrong_device_fails(tmvb, oracle):
    gm, om = _lda_pair(tmvb, oracle, spoil=1e-3)
    block, _ = parity.lda_parity(gm, om, iters=2)
    assert not block["pass"] and block["worst"]["gamma_rel_p999"] > 2e-4


def test_ctm_fix_up(tmvb, oracle):
    pc = tmvb.syn_nsf(M=120, V=300, seed=5)
    K = 6
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=3)
    csr = oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V)
    gm, om = StandCTM(oracle.CTM(csr, K, beta0), vtol_scale=1.003), oracle.CTM(csr, K, beta0)
    block, _ = parity.ctm_parity(gm, om, iters=3, threads=2)
    assert all(v <= parity.CTM_TOL[k] for k, v in block["worst"].items() if v is not None and k != "sweep_mismatch_frac"), block["worst"]
    assert block["worst"]["sweep_mismatch_frac"] > 0
    assert block["worst"]["lambda_err_p999"] <= 1e-6 and block["worst"]["beta_rel_max"] <= 1e-9 and block["worst"]["elbo_rel"] <= 1e-12


def test_ctpf_fix_up(tmvb, oracle):
    pc = tmvb.syn_citeu(M=150, V=300, U=60, seed=4)
    K = 5
    alef0 = np.exp(tmvb.dirichlet_rows(K, pc.V, seed=6) - 0.5)
    csr = oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V, pc.rdr_ptr, pc.readers, pc.ratings, pc.U)
    gm, om = StandCTPF(oracle.CTPF(csr, K, alef0), vtol_scale=1.01), oracle.CTPF(csr, K, alef0)
    block, _ = parity.ctpf_parity(gm, om, iters=4, threads=2, elbo=True)
    assert block["pass"], block["worst"]
    assert block["worst"]["sweep_mismatch_frac"] > 0
    assert block["worst"]["gimel_rel_p999"] <= 1e-12 and block["worst"]["alef_rel_max"] <= 1e-11 and block["worst"]["elbo_rel"] <= 1e-12
