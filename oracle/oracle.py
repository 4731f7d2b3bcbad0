"""
oracle.py -- ctypes loader for the fp64 C oracle (oracle/*.c -> oracle/libtmvb_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  PARITY UNPINNED (see tmvb_oracle.h).

Array conventions: every matrix is a float64 Fortran-ordered numpy array of shape (K, .) so that
indexing reads like the reference (beta[i, j], gamma[i, d]) and the memory is the column-major
K x (.) block the C code expects.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtmvb_oracle.so")
_lib = None

EPSILON = 1.5777218104420236e-30

c_i64 = C.c_int64
c_dbl = C.c_double
P_i64 = C.POINTER(C.c_int64)
P_i32 = C.POINTER(C.c_int32)
P_dbl = C.POINTER(C.c_double)


class Hyper(C.Structure):
    _fields_ = [(n, C.c_double) for n in "abcdefgh"]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("oracle_special.c", "oracle_lda.c", "oracle_ctm.c", "oracle_ctpf.c", "oracle_flda.c", "oracle_fctm.c", "tmvb_oracle.h")]
    if not force and os.path.exists(_LIB_PATH):
        try:
            if all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
                return _LIB_PATH
        except OSError:
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def usable_cpus() -> int:
    """CPUs this process may really use: affinity mask and cgroup quota (a container often sees every host core in
    os.cpu_count() while being limited to a few -- 256 visible, 16 usable on the GPU boxes)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(float(q[0]) / float(q[1]) + 0.5)))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except Exception:
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        # the ELBO evaluations are document-parallel with the OpenMP default team: keep it to the CPUs we may use
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_digamma.restype = c_dbl; _lib.orc_digamma.argtypes = [c_dbl]
        _lib.orc_trigamma.restype = c_dbl; _lib.orc_trigamma.argtypes = [c_dbl]
        _lib.orc_lgamma.restype = c_dbl; _lib.orc_lgamma.argtypes = [c_dbl]
        _lib.orc_lda_update_elbo.restype = c_dbl
        _lib.orc_ctm_update_elbo.restype = c_dbl
        _lib.orc_ctpf_update_elbo.restype = c_dbl
        _lib.orc_flda_update_elbo.restype = c_dbl
        _lib.orc_flda_update_eta.restype = c_dbl
        _lib.orc_fctm_update_elbo.restype = c_dbl
    return _lib


def F64(x, shape=None):
    a = np.asfortranarray(np.array(x, dtype=np.float64, copy=True))
    if shape is not None:
        a = np.asfortranarray(a.reshape(shape, order="F"))
    return a


def _pd(a):
    assert a.dtype == np.float64 and (a.flags.f_contiguous or a.ndim <= 1), "need F-ordered float64"
    return a.ctypes.data_as(P_dbl)


def _pi64(a):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return a.ctypes.data_as(P_i64)


def _pi32(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(P_i32)


def digamma(x):
    flat = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel()); o = np.empty_like(flat)
    lib().orc_digamma_vec(_pd(flat), _pd(o), c_i64(flat.size))
    return o.reshape(np.shape(x))


def trigamma(x):
    flat = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel()); o = np.empty_like(flat)
    lib().orc_trigamma_vec(_pd(flat), _pd(o), c_i64(flat.size))
    return o.reshape(np.shape(x))


# ---- oracle/_ref: the reference's own fp32 digamma (src/utils.jl:21-53, `const DIGAMMA_c`), compiled here from the reference tree
_REF_DIR = os.path.join(_HERE, "_ref")
_REF_LIB_PATH = os.path.join(_REF_DIR, "libref_digamma.so")
_REFERENCE_ROOT = os.environ.get("TMVB_REFERENCE_ROOT", "/root/reference")
_ref_lib = None


def build_ref(force: bool = False):
    """oracle/_ref/Makefile: cuts DIGAMMA_c out of <reference>/src/utils.jl into a temporary directory and compiles it unmodified.
    Returns the library's path, or None when neither the reference tree nor a prebuilt library is there (the GPU box has only
    the prebuilt file: it travels with the snapshot)."""
    src = os.path.join(_REFERENCE_ROOT, "src", "utils.jl")
    if os.path.exists(src) and (force or not os.path.exists(_REF_LIB_PATH)
                                or os.path.getmtime(_REF_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, "ref_digamma_wrap.c")),
                                                                         os.path.getmtime(os.path.join(_REF_DIR, "Makefile")))):
        subprocess.check_call(["make", "-C", _REF_DIR, "-s", "-B", "REF=" + _REFERENCE_ROOT])
    return _REF_LIB_PATH if os.path.exists(_REF_LIB_PATH) else None


def ref_digamma_f32(x):
    """The reference's fp32 digamma helper itself, on an array (float32 in, float32 out)."""
    global _ref_lib
    if _ref_lib is None:
        path = build_ref()
        if path is None:
            raise FileNotFoundError("oracle/_ref/libref_digamma.so is missing and there is no reference tree to build it from")
        _ref_lib = C.CDLL(path)
    flat = np.ascontiguousarray(np.asarray(x, dtype=np.float32).ravel()); o = np.empty_like(flat)
    _ref_lib.ref_digamma_f32_vec(flat.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), c_i64(flat.size))
    return o.reshape(np.shape(x))


class CSR:
    """Packed corpus (0-based ids): doc_ptr int64[M+1], terms/counts int32[nnz], optional readers."""

    def __init__(self, doc_ptr, terms, counts, V, rdr_ptr=None, readers=None, ratings=None, U=0):
        self.doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.int64)
        self.terms = np.ascontiguousarray(terms, dtype=np.int32)
        self.counts = np.ascontiguousarray(counts, dtype=np.int32)
        self.M = len(self.doc_ptr) - 1
        self.V = int(V)
        self.U = int(U)
        if rdr_ptr is None:
            rdr_ptr = np.zeros(self.M + 1, dtype=np.int64)
            readers = np.zeros(0, dtype=np.int32); ratings = np.zeros(0, dtype=np.int32)
        self.rdr_ptr = np.ascontiguousarray(rdr_ptr, dtype=np.int64)
        self.readers = np.ascontiguousarray(readers, dtype=np.int32)
        self.ratings = np.ascontiguousarray(ratings, dtype=np.int32)

    @classmethod
    def from_docs(cls, docs, V, U=0):
        doc_ptr = [0]; terms = []; counts = []; rdr_ptr = [0]; readers = []; ratings = []
        for doc in docs:
            t, c = doc[0], doc[1]
            terms.extend(t); counts.extend(c); doc_ptr.append(len(terms))
            if len(doc) > 2:
                readers.extend(doc[2]); ratings.extend(doc[3])
            rdr_ptr.append(len(readers))
        return cls(doc_ptr, terms, counts, V, rdr_ptr, readers, ratings, U)

    def docs(self):
        out = []
        for d in range(self.M):
            a, b = self.doc_ptr[d], self.doc_ptr[d + 1]
            ra, rb = self.rdr_ptr[d], self.rdr_ptr[d + 1]
            out.append((self.terms[a:b].tolist(), self.counts[a:b].tolist(),
                        self.readers[ra:rb].tolist(), self.ratings[ra:rb].tolist()))
        return out


# ------------------------------------------------------------------------------------- LDA
class LDA:
    """State + operators of src/LDA.jl, backed by the C oracle."""

    def __init__(self, corp: CSR, K: int, beta0):
        self.corp, self.K, self.M, self.V = corp, int(K), corp.M, corp.V
        K = self.K
        self.alpha = np.ones(K)
        self.beta = F64(beta0, (K, self.V))
        self.beta_old = self.beta.copy(order="F")
        self.beta_temp = np.zeros((K, self.V), order="F")
        from scipy.special import digamma as _dg
        e0 = -np.euler_gamma - float(_dg(K))
        self.Elogtheta = np.full((K, self.M), e0, order="F")
        self.Elogtheta_old = self.Elogtheta.copy(order="F")
        self.gamma = np.ones((K, self.M), order="F")
        self.elbo = 0.0

    def _corp_args(self):
        c = self.corp
        return (c_i64(self.M), c_i64(self.V), c_i64(self.K), _pi64(c.doc_ptr), _pi32(c.terms), _pi32(c.counts))

    def estep(self, viter=10, vtol=None, d0=0, d1=None, omp_threads=0):
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        d1 = self.M if d1 is None else d1
        if omp_threads:
            # document-parallel (the per-document results do not depend on the thread count; the statistics' summation order does)
            sw = np.zeros(max(d1 - d0, 1), dtype=np.int32)
            lib().orc_lda_estep_omp_sw(*self._corp_args(), c_i64(d0), c_i64(d1), _pd(self.alpha), _pd(self.beta),
                                       _pd(self.beta_temp), _pd(self.gamma), _pd(self.Elogtheta), _pd(self.Elogtheta_old),
                                       C.c_int(viter), c_dbl(vtol), C.c_int(omp_threads), _pi32(sw))
            return sw[: d1 - d0]
        sw = np.zeros(max(d1 - d0, 1), dtype=np.int32)
        rc = lib().orc_lda_estep(*self._corp_args(), c_i64(d0), c_i64(d1), _pd(self.alpha), _pd(self.beta),
                                 _pd(self.beta_temp), _pd(self.gamma), _pd(self.Elogtheta), _pd(self.Elogtheta_old),
                                 C.c_int(viter), c_dbl(vtol), _pi32(sw))
        assert rc == 0
        return sw[: d1 - d0]

    def update_beta(self):
        lib().orc_lda_update_beta(c_i64(self.V), c_i64(self.K), _pd(self.beta), _pd(self.beta_old), _pd(self.beta_temp))

    def update_alpha(self, niter=1000, ntol=None, Elogtheta_sum=None, Mtot=None):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        if Elogtheta_sum is None:
            Elogtheta_sum = np.zeros(self.K)
            lib().orc_lda_elogtheta_sum(c_i64(self.M), c_i64(self.K), _pd(self.Elogtheta), _pd(Elogtheta_sum))
        Elogtheta_sum = np.ascontiguousarray(Elogtheta_sum, dtype=np.float64)
        return lib().orc_lda_update_alpha(c_i64(self.K), c_i64(self.M if Mtot is None else Mtot), _pd(Elogtheta_sum),
                                          _pd(self.alpha), C.c_int(niter), c_dbl(ntol))

    def update_elbo(self, d0=0, d1=None, store=True):
        d1 = self.M if d1 is None else d1
        e = lib().orc_lda_update_elbo(*self._corp_args(), c_i64(d0), c_i64(d1), _pd(self.alpha), _pd(self.beta),
                                      _pd(self.beta_old), _pd(self.gamma), _pd(self.Elogtheta), _pd(self.Elogtheta_old))
        if store:
            self.elbo = e
        return e

    def train(self, iter=150, tol=1.0, niter=1000, ntol=None, viter=10, vtol=None, checkelbo=1):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        ce = 0 if checkelbo in (None, float("inf")) else int(checkelbo)
        traj = np.full(max(iter, 1), np.nan)
        hist = np.zeros(viter + 1, dtype=np.int64)
        elbo = c_dbl(self.elbo)
        done = lib().orc_lda_train(*self._corp_args(), _pd(self.alpha), _pd(self.beta), _pd(self.beta_old),
                                   _pd(self.gamma), _pd(self.Elogtheta), _pd(self.Elogtheta_old), C.byref(elbo),
                                   C.c_int(iter), c_dbl(tol), C.c_int(niter), c_dbl(ntol), C.c_int(viter), c_dbl(vtol),
                                   C.c_int(ce), _pd(traj), _pi64(hist))
        self.elbo = elbo.value
        self.sweep_hist = hist
        return traj[:done]


# ------------------------------------------------------------------------------------- fLDA
class fLDA(LDA):
    """State + operators of src/fLDA.jl (filtered LDA), backed by the C oracle.  tau / tau_old are flat [nnz] arrays in
    CSR token order.  kappa0: background distribution (the reference draws it from Dirichlet(V, 1), src/fLDA.jl:40)."""

    def __init__(self, corp: CSR, K: int, beta0, kappa0):
        super().__init__(corp, K, beta0)
        self.eta = 0.5                                                    # :38
        self.kappa = np.ascontiguousarray(kappa0, dtype=np.float64).copy()
        self.kappa_old = self.kappa.copy()
        self.kappa_temp = np.zeros(self.V)
        nnz = len(corp.terms)
        self.tau = np.full(nnz, self.eta)                                 # :49
        self.tau_old = self.tau.copy()

    def estep(self, viter=10, vtol=None, d0=0, d1=None, omp_threads=0):
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        d1 = self.M if d1 is None else d1
        if omp_threads:
            return lib().orc_flda_estep_omp(*self._corp_args(), c_i64(d0), c_i64(d1), c_dbl(self.eta), _pd(self.alpha), _pd(self.kappa),
                                            _pd(self.beta), _pd(self.beta_temp), _pd(self.kappa_temp), _pd(self.gamma), _pd(self.Elogtheta),
                                            _pd(self.Elogtheta_old), _pd(self.tau), _pd(self.tau_old), C.c_int(viter), c_dbl(vtol),
                                            C.c_int(omp_threads))
        sw = np.zeros(max(d1 - d0, 1), dtype=np.int32)
        rc = lib().orc_flda_estep(*self._corp_args(), c_i64(d0), c_i64(d1), c_dbl(self.eta), _pd(self.alpha), _pd(self.kappa),
                                  _pd(self.beta), _pd(self.beta_temp), _pd(self.kappa_temp), _pd(self.gamma), _pd(self.Elogtheta),
                                  _pd(self.Elogtheta_old), _pd(self.tau), _pd(self.tau_old), C.c_int(viter), c_dbl(vtol), _pi32(sw))
        assert rc == 0
        return sw[: d1 - d0]

    def update_kappa(self):
        lib().orc_flda_update_kappa(c_i64(self.V), _pd(self.kappa), _pd(self.kappa_old), _pd(self.kappa_temp))

    def update_eta(self):
        c = self.corp
        self.eta = float(lib().orc_flda_update_eta(c_i64(self.M), _pi64(c.doc_ptr), _pi32(c.counts), _pd(self.tau)))
        return self.eta

    def mstep(self, niter=1000, ntol=None):
        """update_beta!, update_kappa!, update_alpha!, update_eta! in the order of src/fLDA.jl:237-240"""
        self.update_beta(); self.update_kappa(); self.update_alpha(niter, ntol); self.update_eta()

    def update_elbo(self, d0=0, d1=None, store=True):
        d1 = self.M if d1 is None else d1
        e = lib().orc_flda_update_elbo(*self._corp_args(), c_i64(d0), c_i64(d1), c_dbl(self.eta), _pd(self.alpha), _pd(self.kappa),
                                       _pd(self.beta), _pd(self.beta_old), _pd(self.gamma), _pd(self.Elogtheta), _pd(self.Elogtheta_old),
                                       _pd(self.tau), _pd(self.tau_old))
        if store:
            self.elbo = e
        return e

    def train(self, iter=150, tol=1.0, niter=1000, ntol=None, viter=10, vtol=None, checkelbo=1):
        """train!  src/fLDA.jl:213-247 incl. check_elbo! (src/modelutils.jl:574-585, signed stop rule)."""
        ce = 0 if checkelbo in (None, float("inf")) else int(checkelbo)
        if len(self.corp.terms) == 0:
            iter = 0
        if ce and ce <= iter:
            self.update_elbo()
        traj = []
        self.sweep_hist = np.zeros(viter + 1, dtype=np.int64)
        for k in range(1, iter + 1):
            sw = self.estep(viter, vtol)
            self.sweep_hist = np.bincount(sw, minlength=viter + 1).astype(np.int64) if self.M else self.sweep_hist
            self.mstep(niter, ntol)
            if ce and k % ce == 0:
                old = self.elbo
                new = self.update_elbo()
                traj.append(new)
                if new - old < tol:
                    break
            else:
                traj.append(float("nan"))
        return np.array(traj)


# ------------------------------------------------------------------------------------- CTM
class CTM:
    """State + operators of src/CTM.jl, backed by the C oracle."""

    def __init__(self, corp: CSR, K: int, beta0):
        self.corp, self.K, self.M, self.V = corp, int(K), corp.M, corp.V
        K = self.K
        self.mu = np.zeros(K)
        self.sigma = np.asfortranarray(np.eye(K))
        self.invsigma = np.asfortranarray(np.eye(K))
        self.beta = F64(beta0, (K, self.V))
        self.beta_old = self.beta.copy(order="F")
        self.beta_temp = np.zeros((K, self.V), order="F")
        self.lam = np.zeros((K, self.M), order="F")
        self.lam_old = np.zeros((K, self.M), order="F")
        self.vsq = np.ones((K, self.M), order="F")
        self.logzeta = np.full(self.M, 0.5)
        self.elbo = 0.0

    def _corp_args(self):
        c = self.corp
        return (c_i64(self.M), c_i64(self.V), c_i64(self.K), _pi64(c.doc_ptr), _pi32(c.terms), _pi32(c.counts))

    def estep(self, niter=1000, ntol=None, viter=10, vtol=None, d0=0, d1=None, omp_threads=0):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        d1 = self.M if d1 is None else d1
        if omp_threads:
            sw = np.zeros(max(d1 - d0, 1), dtype=np.int32); nw = np.zeros(max(d1 - d0, 1), dtype=np.int32)
            lib().orc_ctm_estep_omp_sw(*self._corp_args(), c_i64(d0), c_i64(d1), _pd(self.mu), _pd(self.invsigma),
                                       _pd(self.beta), _pd(self.beta_temp), _pd(self.lam), _pd(self.lam_old),
                                       _pd(self.vsq), _pd(self.logzeta), C.c_int(niter), c_dbl(ntol), C.c_int(viter),
                                       c_dbl(vtol), C.c_int(omp_threads), _pi32(sw), _pi32(nw))
            self.newton_per_doc = nw[: d1 - d0]
            self.newton_steps = int(nw.sum())
            return sw[: d1 - d0]
        sw = np.zeros(max(d1 - d0, 1), dtype=np.int32)
        nst = c_i64(0)
        rc = lib().orc_ctm_estep(*self._corp_args(), c_i64(d0), c_i64(d1), _pd(self.mu), _pd(self.invsigma),
                                 _pd(self.beta), _pd(self.beta_temp), _pd(self.lam), _pd(self.lam_old),
                                 _pd(self.vsq), _pd(self.logzeta), C.c_int(niter), c_dbl(ntol), C.c_int(viter),
                                 c_dbl(vtol), _pi32(sw), C.byref(nst))
        assert rc == 0
        self.newton_steps = nst.value
        return sw[: d1 - d0]

    def update_beta(self):
        lib().orc_lda_update_beta(c_i64(self.V), c_i64(self.K), _pd(self.beta), _pd(self.beta_old), _pd(self.beta_temp))

    def update_sigma_mu(self):
        return lib().orc_ctm_update_sigma_mu(c_i64(self.M), c_i64(self.K), _pd(self.lam), _pd(self.vsq),
                                             _pd(self.mu), _pd(self.sigma), _pd(self.invsigma))

    def update_elbo(self, d0=0, d1=None, store=True):
        d1 = self.M if d1 is None else d1
        e = lib().orc_ctm_update_elbo(*self._corp_args(), c_i64(d0), c_i64(d1), _pd(self.mu), _pd(self.invsigma),
                                      _pd(self.beta), _pd(self.beta_old), _pd(self.lam), _pd(self.lam_old),
                                      _pd(self.vsq), _pd(self.logzeta))
        if store:
            self.elbo = e
        return e

    def train(self, iter=150, tol=1.0, niter=1000, ntol=None, viter=10, vtol=None, checkelbo=1):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        ce = 0 if checkelbo in (None, float("inf")) else int(checkelbo)
        traj = np.full(max(iter, 1), np.nan)
        elbo = c_dbl(self.elbo)
        done = lib().orc_ctm_train(*self._corp_args(), _pd(self.mu), _pd(self.sigma), _pd(self.invsigma), _pd(self.beta),
                                   _pd(self.beta_old), _pd(self.lam), _pd(self.lam_old), _pd(self.vsq), _pd(self.logzeta),
                                   C.byref(elbo), C.c_int(iter), c_dbl(tol), C.c_int(niter), c_dbl(ntol), C.c_int(viter),
                                   c_dbl(vtol), C.c_int(ce), _pd(traj))
        self.elbo = elbo.value
        return traj[:done]


# ------------------------------------------------------------------------------------- fCTM
class fCTM(CTM):
    """State + operators of src/fCTM.jl (filtered CTM), backed by the C oracle.  tau / tau_old: flat [nnz], CSR order."""

    def __init__(self, corp: CSR, K: int, beta0, kappa0):
        super().__init__(corp, K, beta0)
        self.eta = 0.5                                                    # :37
        self.kappa = np.ascontiguousarray(kappa0, dtype=np.float64).copy()
        self.kappa_old = self.kappa.copy()
        self.kappa_temp = np.zeros(self.V)
        nnz = len(corp.terms)
        self.tau = np.full(nnz, self.eta)
        self.tau_old = self.tau.copy()

    def estep(self, niter=1000, ntol=None, viter=10, vtol=None, d0=0, d1=None, omp_threads=0):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        d1 = self.M if d1 is None else d1
        if omp_threads:
            return lib().orc_fctm_estep_omp(*self._corp_args(), c_i64(d0), c_i64(d1), c_dbl(self.eta), _pd(self.kappa), _pd(self.mu),
                                            _pd(self.invsigma), _pd(self.beta), _pd(self.beta_temp), _pd(self.kappa_temp), _pd(self.lam),
                                            _pd(self.lam_old), _pd(self.vsq), _pd(self.logzeta), _pd(self.tau), _pd(self.tau_old),
                                            C.c_int(niter), c_dbl(ntol), C.c_int(viter), c_dbl(vtol), C.c_int(omp_threads))
        sw = np.zeros(max(d1 - d0, 1), dtype=np.int32)
        nst = c_i64(0)
        rc = lib().orc_fctm_estep(*self._corp_args(), c_i64(d0), c_i64(d1), c_dbl(self.eta), _pd(self.kappa), _pd(self.mu),
                                  _pd(self.invsigma), _pd(self.beta), _pd(self.beta_temp), _pd(self.kappa_temp), _pd(self.lam),
                                  _pd(self.lam_old), _pd(self.vsq), _pd(self.logzeta), _pd(self.tau), _pd(self.tau_old),
                                  C.c_int(niter), c_dbl(ntol), C.c_int(viter), c_dbl(vtol), _pi32(sw), C.byref(nst))
        assert rc == 0
        self.newton_steps = nst.value
        return sw[: d1 - d0]

    def update_kappa(self):
        lib().orc_flda_update_kappa(c_i64(self.V), _pd(self.kappa), _pd(self.kappa_old), _pd(self.kappa_temp))

    def mstep(self):
        """update_beta!, update_kappa!, update_sigma!, update_mu! (src/fCTM.jl:249-252; update_eta! is commented out, :253)"""
        self.update_beta(); self.update_kappa(); self.update_sigma_mu()

    def update_elbo(self, d0=0, d1=None, store=True):
        d1 = self.M if d1 is None else d1
        e = lib().orc_fctm_update_elbo(*self._corp_args(), c_i64(d0), c_i64(d1), c_dbl(self.eta), _pd(self.kappa), _pd(self.mu),
                                       _pd(self.invsigma), _pd(self.beta), _pd(self.beta_old), _pd(self.lam), _pd(self.lam_old),
                                       _pd(self.vsq), _pd(self.logzeta), _pd(self.tau), _pd(self.tau_old))
        if store:
            self.elbo = e
        return e

    def train(self, iter=150, tol=1.0, niter=1000, ntol=None, viter=10, vtol=None, checkelbo=1):
        """train!  src/fCTM.jl:226-262 incl. check_elbo! (signed stop rule)."""
        ce = 0 if checkelbo in (None, float("inf")) else int(checkelbo)
        if len(self.corp.terms) == 0:
            iter = 0
        if ce and ce <= iter:
            self.update_elbo()
        traj = []
        for k in range(1, iter + 1):
            self.estep(niter, ntol, viter, vtol)
            self.mstep()
            if ce and k % ce == 0:
                old = self.elbo
                new = self.update_elbo()
                traj.append(new)
                if new - old < tol:
                    break
            else:
                traj.append(float("nan"))
        return np.array(traj)


# ------------------------------------------------------------------------------------ CTPF
class CTPF:
    """State + operators of src/CTPF.jl, backed by the C oracle."""

    def __init__(self, corp: CSR, K: int, alef0):
        self.corp, self.K, self.M, self.V, self.U = corp, int(K), corp.M, corp.V, corp.U
        K, V, U, M = self.K, self.V, self.U, self.M
        self.hp = Hyper(*([0.1] * 8))
        self.alef = F64(alef0, (K, V)); self.alef_old = self.alef.copy(order="F")
        self.alef_temp = np.full((K, V), 0.1, order="F")
        self.he = np.ones((K, U), order="F"); self.he_old = self.he.copy(order="F")
        self.he_temp = np.full((K, U), 0.1, order="F")
        for n in ("bet", "vav", "dalet", "het"):
            setattr(self, n, np.ones(K)); setattr(self, n + "_old", np.ones(K))
        for n in ("gimel", "zayin"):
            setattr(self, n, np.ones((K, M), order="F")); setattr(self, n + "_old", np.ones((K, M), order="F"))
        self.elbo = 0.0

    def _corp_args(self):
        c = self.corp
        return (c_i64(self.M), c_i64(self.V), c_i64(self.U), c_i64(self.K), _pi64(c.doc_ptr), _pi32(c.terms),
                _pi32(c.counts), _pi64(c.rdr_ptr), _pi32(c.readers), _pi32(c.ratings))

    def estep(self, viter=10, vtol=None, d0=0, d1=None, omp_threads=0):
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        d1 = self.M if d1 is None else d1
        common = (*self._corp_args(), c_i64(d0), c_i64(d1), C.byref(self.hp), _pd(self.alef), _pd(self.he),
                  _pd(self.bet), _pd(self.vav), _pd(self.dalet), _pd(self.het), _pd(self.alef_temp), _pd(self.he_temp),
                  _pd(self.gimel), _pd(self.gimel_old), _pd(self.zayin), _pd(self.zayin_old), C.c_int(viter), c_dbl(vtol))
        if omp_threads:
            sw = np.zeros(max(d1 - d0, 1), dtype=np.int32)
            lib().orc_ctpf_estep_omp_sw(*common, C.c_int(omp_threads), _pi32(sw))
            return sw[: d1 - d0]
        sw = np.zeros(max(d1 - d0, 1), dtype=np.int32)
        rc = lib().orc_ctpf_estep(*common, _pi32(sw))
        assert rc == 0
        return sw[: d1 - d0]

    def mstep(self, gimel_sum=None, zayin_sum=None):
        gs = np.ascontiguousarray(self.gimel.sum(axis=1) if gimel_sum is None else gimel_sum)
        zs = np.ascontiguousarray(self.zayin.sum(axis=1) if zayin_sum is None else zayin_sum)
        lib().orc_ctpf_mstep(c_i64(self.V), c_i64(self.U), c_i64(self.K), C.byref(self.hp),
                             _pd(self.alef), _pd(self.alef_old), _pd(self.alef_temp),
                             _pd(self.he), _pd(self.he_old), _pd(self.he_temp), _pd(gs), _pd(zs),
                             _pd(self.bet), _pd(self.bet_old), _pd(self.vav), _pd(self.vav_old),
                             _pd(self.dalet), _pd(self.dalet_old), _pd(self.het), _pd(self.het_old))

    def _state_args(self):
        return (_pd(self.alef), _pd(self.alef_old), _pd(self.he), _pd(self.he_old), _pd(self.bet), _pd(self.bet_old),
                _pd(self.vav), _pd(self.vav_old), _pd(self.dalet), _pd(self.dalet_old), _pd(self.het), _pd(self.het_old),
                _pd(self.gimel), _pd(self.gimel_old), _pd(self.zayin), _pd(self.zayin_old))

    def update_elbo(self, store=True):
        e = lib().orc_ctpf_update_elbo(*self._corp_args(), C.byref(self.hp), *self._state_args())
        if store:
            self.elbo = e
        return e

    def train(self, iter=150, tol=1.0, viter=10, vtol=None, checkelbo=1):
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        ce = 0 if checkelbo in (None, float("inf")) else int(checkelbo)
        traj = np.full(max(iter, 1), np.nan)
        elbo = c_dbl(self.elbo)
        done = lib().orc_ctpf_train(*self._corp_args(), C.byref(self.hp), *self._state_args(), C.byref(elbo),
                                    C.c_int(iter), c_dbl(tol), C.c_int(viter), c_dbl(vtol), C.c_int(ce), _pd(traj))
        self.elbo = elbo.value
        return traj[:done]

    def recommend(self):
        """Tail of train!(model::CTPF), src/CTPF.jl:379-399, restated with NumPy (fp64):
            Eeta = he ./ vav                                                         :379
            scores[d, :] = sum(Eeta .* (gimel[d] ./ dalet + zayin[d] ./ het), dims=1) :380-384
            urecs[u] = findall(ur)[reverse(sortperm(scores[ur, u]))], ur = docs not in libs[u]      :386-391
            drecs[d] = findall(nr)[reverse(sortperm(scores[d, nr]))], nr = users not readers of d   :393-398
        sortperm is stable ascending (Julia's default for vectors), so reverse() yields descending scores with
        equal scores in DESCENDING candidate order.  Indices are 0-based here.  Returns (scores M x U, drecs, urecs)."""
        c = self.corp
        Eeta = self.he / self.vav[:, None]
        X = self.gimel / self.dalet[:, None] + self.zayin / self.het[:, None]        # K x M
        scores = np.empty((self.M, self.U))
        for d in range(self.M):
            scores[d, :] = np.sum(Eeta * X[:, d][:, None], axis=0)
        read = np.zeros((self.M, self.U), dtype=bool)
        for d in range(self.M):
            read[d, c.readers[c.rdr_ptr[d]:c.rdr_ptr[d + 1]]] = True
        urecs, drecs = [], []
        for u in range(self.U):
            cand = np.flatnonzero(~read[:, u])
            urecs.append(cand[np.argsort(scores[cand, u], kind="stable")[::-1]])
        for d in range(self.M):
            cand = np.flatnonzero(~read[d, :])
            drecs.append(cand[np.argsort(scores[d, cand], kind="stable")[::-1]])
        return scores, drecs, urecs
