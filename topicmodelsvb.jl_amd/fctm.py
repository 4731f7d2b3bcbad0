"""
Host-side mirror of the reference's fCTM interface above the C ABI (filtered CTM, src/fCTM.jl).

    fCTM(corp, K)                      src/fCTM.jl:6-65     host fp64 state
    gpufCTM(corp, K)                   NEW: the reference has no device model for the filtered models (`@gpu train!` on an
                                       fCTM is a no-op, src/macros.jl:274-278); this one follows the gpuCTM pattern
      .update_buffer() / .update_host()
      .estep(niter, ntol, viter, vtol) update_phi!/update_tau!/update_logzeta!/update_lambda!/update_vsq! sweeps +
                                       update_beta!(d) + update_kappa!(d), src/fCTM.jl:233-248
      .update_beta() .update_sigma() .update_mu() .update_elbo()       :148/:134, :128, :122, :105
      .train(iter=150, tol=1.0, niter=1000, ntol=1/K^2, viter=10, vtol=1/K^2, checkelbo=1, printelbo=True)   :226-262
    gpu_train_fctm(model, **kwargs)    `@gpu train!(model::fCTM; kwargs...)`

tau / tau_old are flat [nnz] arrays in CSR token order (the reference's tau[d][n]).  eta is a fixed parameter: update_eta!
is commented out of the reference's train! (src/fCTM.jl:253).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from ._lib import TopicModelError, check, lib, P_i64, VP
from .corpus import dirichlet_rows
from .ctm import CTM, check_model_ctm
from .lda import DeviceContext, DeviceCorpus, _F, _pd, _print_delbo, _topic_orders, _validate_train_args


class fCTM(CTM):
    """Host (fp64) fCTM state with the reference's field names, src/fCTM.jl:6-65."""

    def __init__(self, corp, K: int, seed: int = 7):
        super().__init__(corp, K, seed)
        self.eta = 0.5                                                              # :37
        self.kappa = dirichlet_rows(1, self.V, seed + 1)[0].copy() if self.V else np.zeros(0)   # :44 (Julia RNG in the reference)
        self.kappa_old = self.kappa.copy()
        self.kappa_temp = np.zeros(self.V)
        self.tau = np.full(self.corp.nnz, self.eta)                                 # :54
        self.tau_old = self.tau.copy()


def check_model_fctm(model, rtol: float = 1.5e-8):
    """check_model(::fCTM)  src/modelutils.jl:139-178 (array form)."""
    check_model_ctm(model, rtol=rtol)
    if not (0.0 <= model.eta <= 1.0):
        raise TopicModelError("eta must belong to the interval [0,1].")
    for name in ("kappa", "kappa_old"):
        k = getattr(model, name)
        if k.shape != (model.V,):
            raise TopicModelError(f"{name} must be of length V.")
        if model.V and not (np.all(k >= 0) and np.isclose(k.sum(), 1.0, rtol=rtol, atol=0)):
            raise TopicModelError(f"{name} must be a probability vector.")
    if model.tau.shape != (model.corp.nnz,):
        raise TopicModelError("tau must contain vectors of lengths N.")
    if not np.all((model.tau >= 0) & (model.tau <= 1)):
        raise TopicModelError("tau must belong to the interval [0,1].")


class gpufCTM:
    """Device-backed filtered CTM model on libtmvb_hip.so."""

    _FIELDS = ("corp", "K", "M", "V", "N", "C", "topics", "eta", "mu", "sigma", "invsigma", "kappa", "kappa_old", "beta", "beta_old",
               "lam", "lam_old", "vsq", "logzeta", "tau", "tau_old", "elbo")

    def __init__(self, corp, K: int, seed: int = 7, ctx: DeviceContext | None = None, device_id: int = 0, _from: fCTM | None = None):
        if not (isinstance(K, (int, np.integer)) and K > 0):
            raise ValueError("number of topics must be a positive integer.")
        host = _from if _from is not None else fCTM(corp, K, seed)
        for k in self._FIELDS:
            setattr(self, k, getattr(host, k))
        self.ctx = ctx or DeviceContext(device_id)
        self.dcorp = DeviceCorpus(self.ctx, self.corp)
        self.handle = VP()
        check(lib().tmvb_fctm_create(self.ctx.handle, self.dcorp.handle, C.c_int32(self.K), C.byref(self.handle)))
        self.M_total = self.M
        self.update_buffer()

    def update_buffer(self):
        K, M, V, nnz = self.K, self.M, self.V, self.corp.nnz
        vec = lambda x, n: np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(n))
        eta, elbo = C.c_double(float(self.eta)), C.c_double(float(self.elbo))
        check(lib().tmvb_fctm_set_state(self.handle, C.byref(eta), _pd(vec(self.mu, K)), _pd(_F(self.sigma, (K, K))),
                                        _pd(_F(self.invsigma, (K, K))), _pd(vec(self.kappa, V)), _pd(vec(self.kappa_old, V)),
                                        _pd(_F(self.beta, (K, V))), _pd(_F(self.beta_old, (K, V))), _pd(_F(self.lam, (K, M))),
                                        _pd(_F(self.lam_old, (K, M))), _pd(_F(self.vsq, (K, M))), _pd(vec(self.logzeta, M)),
                                        _pd(vec(self.tau, nnz)), _pd(vec(self.tau_old, nnz)), C.byref(elbo)))

    def update_host(self):
        K, M, V, nnz = self.K, self.M, self.V, self.corp.nnz
        eta, elbo = C.c_double(0.0), C.c_double(0.0)
        self.mu = np.empty(K)
        self.sigma = np.empty((K, K), order="F"); self.invsigma = np.empty((K, K), order="F")
        self.kappa = np.empty(V); self.kappa_old = np.empty(V)
        self.beta = np.empty((K, V), order="F"); self.beta_old = np.empty((K, V), order="F")
        self.lam = np.empty((K, M), order="F"); self.lam_old = np.empty((K, M), order="F")
        self.vsq = np.empty((K, M), order="F"); self.logzeta = np.empty(M)
        self.tau = np.empty(nnz); self.tau_old = np.empty(nnz)
        check(lib().tmvb_fctm_get_state(self.handle, C.byref(eta), _pd(self.mu), _pd(self.sigma), _pd(self.invsigma), _pd(self.kappa),
                                        _pd(self.kappa_old), _pd(self.beta), _pd(self.beta_old), _pd(self.lam), _pd(self.lam_old),
                                        _pd(self.vsq), _pd(self.logzeta), _pd(self.tau), _pd(self.tau_old), C.byref(elbo)))
        self.eta, self.elbo = eta.value, elbo.value

    def estep(self, niter: int = 1000, ntol: float | None = None, viter: int = 10, vtol: float | None = None):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check(lib().tmvb_fctm_estep(self.handle, C.c_int32(niter), C.c_double(ntol), C.c_int32(viter), C.c_double(vtol)))

    def reduce_docs(self): check(lib().tmvb_fctm_reduce_docs(self.handle))
    def update_beta(self): check(lib().tmvb_fctm_update_beta(self.handle))          # beta and kappa (src/fCTM.jl:249-250)
    def update_sigma(self): check(lib().tmvb_fctm_update_sigma(self.handle))
    def update_mu(self): check(lib().tmvb_fctm_update_mu(self.handle))

    def mstep(self):
        self.update_beta(); self.update_sigma(); self.update_mu()

    def elbo_form(self) -> int:
        """1 if the last update_elbo! took the decomposed form (tmvb_fctm_elbo_form: nothing per token rebuilt), 0 for the token walk."""
        f = C.c_int32(0)
        check(lib().tmvb_fctm_elbo_form(self.handle, C.byref(f)))
        return f.value

    def update_elbo(self) -> float:
        out = C.c_double(0.0)
        check(lib().tmvb_fctm_update_elbo(self.handle, C.byref(out)))
        self.elbo = out.value
        return out.value

    def sweep_hist(self, nbins: int = 11):
        h = np.zeros(nbins, dtype=np.int64)
        ns = C.c_int64(0)
        check(lib().tmvb_fctm_sweep_hist(self.handle, h.ctypes.data_as(P_i64), C.c_int32(nbins), C.byref(ns)))
        return h, ns.value

    def doc_sweeps(self):
        out = np.zeros(max(self.M, 1), dtype=np.uint8)
        check(lib().tmvb_fctm_doc_sweeps(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out[:self.M]

    def set_comm(self, comm, M_total: int):
        self.M_total = int(M_total) if comm is not None else self.M
        self._comm = comm
        check(lib().tmvb_fctm_set_comm(self.handle, comm.handle if comm is not None else VP(None), C.c_int64(self.M_total)))

    def synchronize(self):
        self.ctx.synchronize()

    def train(self, iter: int = 150, tol: float = 1.0, niter: int = 1000, ntol: float | None = None, viter: int = 10,
              vtol: float | None = None, checkelbo=1, printelbo: bool = True):
        """train!(model::fCTM; ...) src/fCTM.jl:226-262 on the device.  Returns the ELBO trajectory."""
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check_model_fctm(self, rtol=3.5e-4)
        _validate_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
        self.update_buffer()
        ce = 0 if checkelbo == math.inf else int(checkelbo)
        traj = np.full(max(iter, 1), np.nan)
        done, base = C.c_int32(0), C.c_double(float(self.elbo))
        check(lib().tmvb_fctm_train(self.handle, C.c_int32(iter), C.c_double(tol), C.c_int32(niter), C.c_double(ntol), C.c_int32(viter),
                                    C.c_double(vtol), C.c_int32(ce), _pd(traj), C.byref(done), C.byref(base)))
        traj = traj[:done.value]
        self.elbo_baseline = base.value
        if iter > 0:
            self.update_host()
        if printelbo and ce:
            _print_delbo(traj, base.value)
        self.topics = _topic_orders(self.ctx, self.beta)   # :260
        return traj

    def close(self):
        if getattr(self, "handle", None):
            lib().tmvb_fctm_destroy(self.handle)
            self.handle = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gpu_train_fctm(model: fCTM, device_id: int = 0, **kwargs):
    """`@gpu train!(model::fCTM; kwargs...)`: a no-op in the reference (src/macros.jl:274-275); here the model trains on the
    device and comes back like the CTM branch does (beta / kappa re-normalised in fp64, *_old synced)."""
    if not isinstance(model, fCTM):
        raise ValueError("gpu_train_fctm needs an fCTM model.")
    g = gpufCTM(model.corp, model.K, device_id=device_id, _from=model)
    traj = g.train(**kwargs)
    for n in ("topics", "eta", "mu", "sigma", "invsigma", "lam", "vsq", "logzeta", "tau", "elbo"):
        setattr(model, n, getattr(g, n))
    model.beta = g.beta / g.beta.sum(axis=1, keepdims=True); model.beta_old = model.beta.copy(order="F")
    model.kappa = g.kappa / g.kappa.sum(); model.kappa_old = model.kappa.copy()
    model.lam_old = model.lam.copy(order="F"); model.tau_old = model.tau.copy()
    g.close()
    return traj


def predict_fctm(corp, train_model, iter: int = 10, tol: float | None = None, niter: int = 1000, ntol: float | None = None,
                 device_id: int = 0, seed: int = 7) -> fCTM:
    """predict(corp, train_model::fCTM; iter, tol, niter, ntol)  src/modelutils.jl:916-943 on the device: one pass of the
    fused filtered CTM E-step (phi, tau, logzeta, lambda, vsq sweeps) with mu / sigma / invsigma / beta frozen (:924-928);
    kappa and eta are those of the fresh fCTM(corp, K), as in the reference.  `tol` stands for the reference's undefined
    `vtol` (:937)."""
    K = train_model.K
    tol = 1.0 / K ** 2 if tol is None else tol
    ntol = 1.0 / K ** 2 if ntol is None else ntol
    from .lda import _packed
    pc = _packed(corp)
    if pc.V != train_model.V:
        from ._lib import CorpusError
        raise CorpusError("predict corpus and train_model corpus must have identical vocabularies.")
    if tol < 0 or ntol < 0:
        raise ValueError("tolerance parameters must be nonnegative.")
    if iter < 0 or niter < 0:
        raise ValueError("iteration parameters must be nonnegative.")
    host = fCTM(pc, K, seed)
    host.mu, host.sigma, host.invsigma = np.array(train_model.mu), np.asfortranarray(train_model.sigma), np.asfortranarray(train_model.invsigma)
    host.beta = np.asfortranarray(train_model.beta); host.beta_old = host.beta.copy(order="F")
    host.topics = train_model.topics
    g = gpufCTM(None, K, device_id=device_id, _from=host)
    g.estep(niter, ntol, iter, tol)
    g.update_host()
    for n in ("lam", "lam_old", "vsq", "logzeta", "tau", "tau_old"):
        setattr(host, n, getattr(g, n))
    g.close()
    return host
