"""
Host-side mirror of the reference's CTM / gpuCTM interface above the C ABI.

    CTM(corp, K)                      src/CTM.jl:6-53      host fp64 state
    gpuCTM(corp, K)                   src/gpuCTM.jl:6-99   device-backed model
      .update_buffer() / .update_host()      src/modelutils.jl:400-435 / :519-537
      .estep(niter, ntol, viter, vtol)       update_phi!/update_logzeta!/update_vsq!/update_lambda! sweeps +
                                             update_beta!(d), CPU-path semantics src/CTM.jl:194-205
      .update_beta() .update_sigma() .update_mu() .update_elbo()
      .train(iter=150, tol=1.0, niter=1000, ntol=1/K^2, viter=10, vtol=1/K^2, checkelbo=1, printelbo=True)
                                             src/gpuCTM.jl:487-519
    gpu_train_ctm(model, **kwargs)    `@gpu train!(model::CTM; kwargs...)`  src/macros.jl:152-195
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from ._lib import TopicModelError, check, lib, P_dbl, P_i64, VP
from .corpus import dirichlet_rows
from .lda import DeviceContext, DeviceCorpus, _F, _packed, _pd, _print_delbo, _topic_orders, _validate_train_args


class CTM:
    """Host (fp64) CTM state with the reference's field names, src/CTM.jl:6-53."""

    def __init__(self, corp, K: int, seed: int = 7):
        if not (isinstance(K, (int, np.integer)) and K > 0):
            raise ValueError("number of topics must be a positive integer.")
        self.corp = _packed(corp)
        self.K, self.M, self.V = int(K), self.corp.M, self.corp.V
        self.N, self.C = self.corp.N, self.corp.C
        K, M, V = self.K, self.M, self.V
        self.topics = [np.arange(1, V + 1) for _ in range(K)]
        self.mu = np.zeros(K)
        self.sigma = np.asfortranarray(np.eye(K))
        self.invsigma = np.asfortranarray(np.eye(K))
        self.beta = dirichlet_rows(K, V, seed)
        self.beta_old = self.beta.copy(order="F")
        self.beta_temp = np.zeros((K, V), order="F")
        self.lam = np.zeros((K, M), order="F")               # `lambda` is a Python keyword
        self.lam_old = np.zeros((K, M), order="F")
        self.vsq = np.ones((K, M), order="F")
        self.logzeta = np.full(M, 0.5)
        self.elbo = 0.0


def check_model_ctm(model, rtol: float = 1.5e-8):
    """check_model(::CTM) src/modelutils.jl:108-138 (array form)."""
    K, M, V = model.K, model.M, model.V
    if not np.all(np.isfinite(model.mu)):
        raise TopicModelError("mu must be finite.")
    for name in ("sigma", "invsigma"):
        a = getattr(model, name)
        if a.shape != (K, K):
            raise TopicModelError(f"{name} must be of size (K, K).")
        try:
            np.linalg.cholesky(0.5 * (a + a.T))
        except np.linalg.LinAlgError:
            raise TopicModelError(f"{name} must be positive-definite.")
    if model.beta.shape != (K, V):
        raise TopicModelError("beta must be of size (K, V).")
    if V and not (np.all(model.beta >= 0) and np.allclose(model.beta.sum(axis=1), 1.0, rtol=rtol, atol=0)):
        raise TopicModelError("beta must be a right stochastic matrix.")
    if model.lam.shape != (K, M) or not np.all(np.isfinite(model.lam)):
        raise TopicModelError("lambda must be finite.")
    if model.vsq.shape != (K, M) or not np.all(np.isfinite(model.vsq)):
        raise TopicModelError("vsq must be finite.")
    if not np.all(model.vsq > 0):
        raise TopicModelError("vsq must be positive.")
    if model.logzeta.shape != (M,) or not np.all(np.isfinite(model.logzeta)):
        raise TopicModelError("logzeta must be finite.")
    if not math.isfinite(model.elbo):
        raise TopicModelError("elbo must be finite.")


class gpuCTM:
    """GPU accelerated correlated topic model (src/gpuCTM.jl:6-99) on libtmvb_hip.so."""

    _FIELDS = ("corp", "K", "M", "V", "N", "C", "topics", "mu", "sigma", "invsigma", "beta", "beta_old", "lam", "lam_old",
               "vsq", "logzeta", "elbo")

    def __init__(self, corp, K: int, seed: int = 7, ctx: DeviceContext | None = None, device_id: int = 0, stream=None,
                 _from: CTM | None = None):
        host = _from if _from is not None else CTM(corp, K, seed)
        for k in self._FIELDS:
            setattr(self, k, getattr(host, k))
        self.ctx = ctx or DeviceContext(device_id, stream)
        self.dcorp = DeviceCorpus(self.ctx, self.corp)
        self.handle = VP()
        check(lib().tmvb_ctm_create(self.ctx.handle, self.dcorp.handle, C.c_int32(self.K), C.byref(self.handle)))
        self.M_total = self.M
        self.update_buffer()

    def update_buffer(self):
        K, M, V = self.K, self.M, self.V
        mu = np.ascontiguousarray(self.mu, dtype=np.float64)
        elbo = C.c_double(float(self.elbo))
        lz = np.ascontiguousarray(self.logzeta, dtype=np.float64)
        check(lib().tmvb_ctm_set_state(self.handle, _pd(mu), _pd(_F(self.sigma, (K, K))), _pd(_F(self.invsigma, (K, K))),
                                       _pd(_F(self.beta, (K, V))), _pd(_F(self.beta_old, (K, V))), _pd(_F(self.lam, (K, M))),
                                       _pd(_F(self.lam_old, (K, M))), _pd(_F(self.vsq, (K, M))), _pd(lz), C.byref(elbo)))

    def update_host(self):
        K, M, V = self.K, self.M, self.V
        self.mu = np.empty(K)
        self.sigma = np.empty((K, K), order="F"); self.invsigma = np.empty((K, K), order="F")
        self.beta = np.empty((K, V), order="F"); self.beta_old = np.empty((K, V), order="F")
        self.lam = np.empty((K, M), order="F"); self.lam_old = np.empty((K, M), order="F")
        self.vsq = np.empty((K, M), order="F"); self.logzeta = np.empty(M)
        elbo = C.c_double(0.0)
        check(lib().tmvb_ctm_get_state(self.handle, _pd(self.mu), _pd(self.sigma), _pd(self.invsigma), _pd(self.beta),
                                       _pd(self.beta_old), _pd(self.lam), _pd(self.lam_old), _pd(self.vsq), _pd(self.logzeta),
                                       C.byref(elbo)))
        self.elbo = elbo.value

    def estep(self, niter: int = 1000, ntol: float | None = None, viter: int = 10, vtol: float | None = None):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check(lib().tmvb_ctm_estep(self.handle, C.c_int32(niter), C.c_double(ntol), C.c_int32(viter), C.c_double(vtol)))

    def reduce_docs(self): check(lib().tmvb_ctm_reduce_docs(self.handle))
    def update_beta(self): check(lib().tmvb_ctm_update_beta(self.handle))
    def update_sigma(self): check(lib().tmvb_ctm_update_sigma(self.handle))
    def update_mu(self): check(lib().tmvb_ctm_update_mu(self.handle))

    def update_elbo(self) -> float:
        out = C.c_double(0.0)
        check(lib().tmvb_ctm_update_elbo(self.handle, C.byref(out)))
        self.elbo = out.value
        return out.value

    def elbo_form(self) -> int:
        """1 if the last update_elbo! took the decomposed form (parts left behind by the iteration itself), 0 for the token walk."""
        f = C.c_int32(0)
        check(lib().tmvb_ctm_elbo_form(self.handle, C.byref(f)))
        return f.value

    def stats(self):
        p, n = VP(), C.c_int64(0)
        check(lib().tmvb_ctm_stats(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_stats(self, dev_ptr: int, n_f32: int):
        check(lib().tmvb_ctm_bind_stats(self.handle, VP(dev_ptr), C.c_int64(n_f32)))

    def set_distributed(self, M_total: int, distributed: bool = True):
        self.M_total = int(M_total)
        check(lib().tmvb_ctm_set_distributed(self.handle, C.c_int64(M_total), C.c_int32(1 if distributed else 0)))

    def sweep_hist(self, nbins: int = 11):
        h = np.zeros(nbins, dtype=np.int64)
        ns = C.c_int64(0)
        check(lib().tmvb_ctm_sweep_hist(self.handle, h.ctypes.data_as(P_i64), C.c_int32(nbins), C.byref(ns)))
        return h, ns.value

    def solver_stats(self):
        """Lane-per-document kernel diagnostics of the last E-step (tmvb_ctm_solver_stats)."""
        out = np.zeros(12, dtype=np.int64)
        check(lib().tmvb_ctm_solver_stats(self.handle, out.ctypes.data_as(P_i64)))
        names = ("cg_trips", "newton_trips", "waves", "cyc_token", "cyc_logzeta", "cyc_vsq", "cyc_gradient", "cyc_cg", "cyc_gradmv",
                 "cyc_update", "cyc_spare", "cyc_kernel")
        return dict(zip(names, out.tolist()))

    def doc_sweeps(self):
        """Sweeps each document ran in the last E-step (uint8 per document, corpus order)."""
        out = np.zeros(max(self.M, 1), dtype=np.uint8)
        check(lib().tmvb_ctm_doc_sweeps(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out[:self.M]

    def last_estep_ms(self) -> float:
        ms = C.c_float(0.0)
        check(lib().tmvb_ctm_last_estep_ms(self.handle, C.byref(ms)))
        return ms.value

    def synchronize(self):
        self.ctx.synchronize()

    def set_comm(self, comm, M_total: int):
        """Attach a communicator (comm.py): document-sharded train!, every rank calls train() with the same arguments."""
        self.M_total = int(M_total) if comm is not None else self.M
        self._comm = comm
        check(lib().tmvb_ctm_set_comm(self.handle, comm.handle if comm is not None else VP(None), C.c_int64(self.M_total)))

    def train(self, iter: int = 150, tol: float = 1.0, niter: int = 1000, ntol: float | None = None, viter: int = 10,
              vtol: float | None = None, checkelbo=1, printelbo: bool = True):
        """train!(model::gpuCTM; ...) src/gpuCTM.jl:487-519."""
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check_model_ctm(self, rtol=3.5e-4)
        _validate_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
        self.update_buffer()
        ce = 0 if checkelbo == math.inf else int(checkelbo)
        traj = np.full(max(iter, 1), np.nan)
        done, base = C.c_int32(0), C.c_double(float(self.elbo))
        check(lib().tmvb_ctm_train(self.handle, C.c_int32(iter), C.c_double(tol), C.c_int32(niter), C.c_double(ntol),
                                   C.c_int32(viter), C.c_double(vtol), C.c_int32(ce), _pd(traj), C.byref(done), C.byref(base)))
        traj = traj[:done.value]
        self.elbo_baseline = base.value
        if iter > 0:
            self.update_host()
        if printelbo and ce:
            _print_delbo(traj, base.value)
        self.topics = _topic_orders(self.ctx, self.beta)
        return traj

    def close(self):
        if getattr(self, "handle", None):
            lib().tmvb_ctm_destroy(self.handle)
            self.handle = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gpu_train_ctm(model: CTM, device_id: int = 0, **kwargs):
    """`@gpu train!(model::CTM; kwargs...)` (src/macros.jl:152-195)."""
    g = gpuCTM(None, model.K, device_id=device_id, _from=model)
    traj = g.train(**kwargs)
    model.topics, model.mu, model.sigma, model.invsigma = g.topics, g.mu, g.sigma, g.invsigma
    model.beta = g.beta / g.beta.sum(axis=1, keepdims=True)         # src/macros.jl:190
    model.beta_old = model.beta.copy(order="F")
    model.lam = g.lam; model.lam_old = g.lam.copy(order="F")        # :183
    model.vsq, model.logzeta, model.elbo = g.vsq, g.logzeta, g.elbo
    g.close()
    return traj


def predict_ctm(corp, train_model, iter: int = 10, tol: float | None = None, niter: int = 1000, ntol: float | None = None,
                device_id: int = 0) -> CTM:
    """predict(corp, train_model::Union{CTM, gpuCTM}; iter, tol, niter, ntol)  src/modelutils.jl:886-913: one pass of the
    fused CTM E-step with the trained mu / sigma / beta frozen."""
    K = train_model.K
    tol = 1.0 / K ** 2 if tol is None else tol
    ntol = 1.0 / K ** 2 if ntol is None else ntol
    pc = _packed(corp)
    if pc.V != train_model.V:
        from ._lib import CorpusError
        raise CorpusError("predict corpus and train_model corpus must have identical vocabularies.")
    if tol < 0 or ntol < 0:
        raise ValueError("tolerance parameters must be nonnegative.")
    if iter < 0 or niter < 0:
        raise ValueError("iteration parameters must be nonnegative.")
    host = CTM(pc, K)
    host.mu, host.sigma, host.invsigma = np.array(train_model.mu), np.asfortranarray(train_model.sigma), np.asfortranarray(train_model.invsigma)
    host.beta = np.asfortranarray(train_model.beta); host.beta_old = host.beta.copy(order="F")
    g = gpuCTM(None, K, device_id=device_id, _from=host)
    g.estep(niter, ntol, iter, tol)
    g.update_host()
    for n in ("lam", "lam_old", "vsq", "logzeta"):
        setattr(host, n, getattr(g, n))
    host.topics = train_model.topics
    g.close()
    return host


def topicdist_ctm(model, d):
    """topicdist(model::Union{CTM, gpuCTM}, d)  src/modelutils.jl:953-958: additive_logistic(lambda + vsq/2)."""
    if not isinstance(d, (int, np.integer)):
        return [topicdist_ctm(model, int(x)) for x in d]
    if not (1 <= d <= model.M):
        from ._lib import CorpusError
        raise CorpusError("document index outside corpus range.")
    x = model.lam[:, d - 1] + 0.5 * model.vsq[:, d - 1]
    x = np.exp(x - x.max())
    return x / x.sum()
