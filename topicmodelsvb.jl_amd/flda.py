"""
Host-side mirror of the reference's fLDA interface above the C ABI (filtered LDA, src/fLDA.jl).

    fLDA(corp, K)                      src/fLDA.jl:6-60     host fp64 state
    gpufLDA(corp, K)                   NEW: the reference has no device model for the filtered models (`@gpu train!` on an
                                       fLDA is a no-op, src/macros.jl:274-278); this one follows the gpuLDA pattern
      .update_buffer() / .update_host()
      .estep(viter, vtol)              update_phi!/update_tau!/update_gamma!/update_Elogtheta! sweeps + update_beta!(d) +
                                       update_kappa!(d), src/fLDA.jl:222-236
      .update_beta() .update_alpha() .update_eta() .update_elbo()      :152/:138, :128, :122, :108
      .train(iter=150, tol=1.0, niter=1000, ntol=1/K^2, viter=10, vtol=1/K^2, checkelbo=1, printelbo=True)   :213-247
    gpu_train_flda(model, **kwargs)    `@gpu train!(model::fLDA; kwargs...)`

tau / tau_old are flat [nnz] arrays in CSR token order (the reference's tau[d][n]).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from ._lib import TopicModelError, check, lib, VP
from .corpus import dirichlet_rows
from .lda import LDA, DeviceContext, DeviceCorpus, _F, _pd, _print_delbo, _topic_orders, _validate_train_args


class fLDA(LDA):
    """Host (fp64) fLDA state with the reference's field names, src/fLDA.jl:6-60."""

    def __init__(self, corp, K: int, seed: int = 7):
        super().__init__(corp, K, seed)
        self.eta = 0.5                                                              # :38
        self.kappa = dirichlet_rows(1, self.V, seed + 1)[0].copy() if self.V else np.zeros(0)   # :40 (Julia RNG in the reference)
        self.kappa_old = self.kappa.copy()
        self.kappa_temp = np.zeros(self.V)
        self.tau = np.full(self.corp.nnz, self.eta)                                 # :49
        self.tau_old = self.tau.copy()


def check_model_flda(model, rtol: float = 1.5e-8):
    """check_model(::fLDA)  src/modelutils.jl:69-107 (array form)."""
    from .lda import check_model
    check_model(model, rtol=rtol)
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


class gpufLDA:
    """Device-backed filtered LDA model on libtmvb_hip.so."""

    _FIELDS = ("corp", "K", "M", "V", "N", "C", "topics", "eta", "alpha", "kappa", "kappa_old", "beta", "beta_old", "Elogtheta",
               "Elogtheta_old", "gamma", "tau", "tau_old", "elbo")

    def __init__(self, corp, K: int, seed: int = 7, ctx: DeviceContext | None = None, device_id: int = 0, _from: fLDA | None = None):
        if not (isinstance(K, (int, np.integer)) and K > 0):
            raise ValueError("number of topics must be a positive integer.")
        host = _from if _from is not None else fLDA(corp, K, seed)
        for k in self._FIELDS:
            setattr(self, k, getattr(host, k))
        self.ctx = ctx or DeviceContext(device_id)
        self.dcorp = DeviceCorpus(self.ctx, self.corp)
        self.handle = VP()
        check(lib().tmvb_flda_create(self.ctx.handle, self.dcorp.handle, C.c_int32(self.K), C.byref(self.handle)))
        self.update_buffer()

    def update_buffer(self):
        K, M, V, nnz = self.K, self.M, self.V, self.corp.nnz
        vec = lambda x, n: np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(n))
        eta, elbo = C.c_double(float(self.eta)), C.c_double(float(self.elbo))
        check(lib().tmvb_flda_set_state(self.handle, C.byref(eta), _pd(vec(self.alpha, K)), _pd(vec(self.kappa, V)), _pd(vec(self.kappa_old, V)),
                                        _pd(_F(self.beta, (K, V))), _pd(_F(self.beta_old, (K, V))), _pd(_F(self.gamma, (K, M))),
                                        _pd(_F(self.Elogtheta, (K, M))), _pd(_F(self.Elogtheta_old, (K, M))), _pd(vec(self.tau, nnz)),
                                        _pd(vec(self.tau_old, nnz)), C.byref(elbo)))

    def update_host(self):
        K, M, V, nnz = self.K, self.M, self.V, self.corp.nnz
        eta, elbo = C.c_double(0.0), C.c_double(0.0)
        self.alpha = np.empty(K); self.kappa = np.empty(V); self.kappa_old = np.empty(V)
        self.beta = np.empty((K, V), order="F"); self.beta_old = np.empty((K, V), order="F")
        self.gamma = np.empty((K, M), order="F"); self.Elogtheta = np.empty((K, M), order="F"); self.Elogtheta_old = np.empty((K, M), order="F")
        self.tau = np.empty(nnz); self.tau_old = np.empty(nnz)
        check(lib().tmvb_flda_get_state(self.handle, C.byref(eta), _pd(self.alpha), _pd(self.kappa), _pd(self.kappa_old), _pd(self.beta),
                                        _pd(self.beta_old), _pd(self.gamma), _pd(self.Elogtheta), _pd(self.Elogtheta_old), _pd(self.tau),
                                        _pd(self.tau_old), C.byref(elbo)))
        self.eta, self.elbo = eta.value, elbo.value

    def estep(self, viter: int = 10, vtol: float | None = None):
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check(lib().tmvb_flda_estep(self.handle, C.c_int32(viter), C.c_double(vtol)))

    def reduce_docs(self): check(lib().tmvb_flda_reduce_docs(self.handle))
    def update_beta(self): check(lib().tmvb_flda_update_beta(self.handle))          # beta and kappa (src/fLDA.jl:237-238)
    def update_eta(self): check(lib().tmvb_flda_update_eta(self.handle))

    def update_alpha(self, niter: int = 1000, ntol: float | None = None):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        check(lib().tmvb_flda_update_alpha(self.handle, C.c_int32(niter), C.c_double(ntol)))

    def mstep(self, niter: int = 1000, ntol: float | None = None):
        self.update_beta(); self.update_alpha(niter, ntol); self.update_eta()

    def elbo_form(self) -> int:
        """1 if the last update_elbo! took the decomposed form (tmvb_flda_elbo_form: nothing per token rebuilt), 0 for the token walk."""
        f = C.c_int32(0)
        check(lib().tmvb_flda_elbo_form(self.handle, C.byref(f)))
        return f.value

    def update_elbo(self) -> float:
        out = C.c_double(0.0)
        check(lib().tmvb_flda_update_elbo(self.handle, C.byref(out)))
        self.elbo = out.value
        return out.value

    def doc_sweeps(self):
        out = np.zeros(max(self.M, 1), dtype=np.uint8)
        check(lib().tmvb_flda_doc_sweeps(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out[:self.M]

    def last_estep_ms(self) -> float:
        ms = C.c_float(0.0)
        check(lib().tmvb_flda_last_estep_ms(self.handle, C.byref(ms)))
        return ms.value

    def set_comm(self, comm, M_total: int, C_total: int):
        self._comm = comm
        check(lib().tmvb_flda_set_comm(self.handle, comm.handle if comm is not None else VP(None), C.c_int64(M_total), C.c_int64(C_total)))

    def synchronize(self):
        self.ctx.synchronize()

    def train(self, iter: int = 150, tol: float = 1.0, niter: int = 1000, ntol: float | None = None, viter: int = 10,
              vtol: float | None = None, checkelbo=1, printelbo: bool = True):
        """train!(model::fLDA; ...) src/fLDA.jl:213-247 on the device.  Returns the ELBO trajectory."""
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check_model_flda(self, rtol=3.5e-4)
        _validate_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
        self.update_buffer()
        ce = 0 if checkelbo == math.inf else int(checkelbo)
        traj = np.full(max(iter, 1), np.nan)
        done, base = C.c_int32(0), C.c_double(float(self.elbo))
        check(lib().tmvb_flda_train(self.handle, C.c_int32(iter), C.c_double(tol), C.c_int32(niter), C.c_double(ntol), C.c_int32(viter),
                                    C.c_double(vtol), C.c_int32(ce), _pd(traj), C.byref(done), C.byref(base)))
        traj = traj[:done.value]
        self.elbo_baseline = base.value
        if iter > 0:
            self.update_host()
        if printelbo and ce:
            _print_delbo(traj, base.value)
        self.topics = _topic_orders(self.ctx, self.beta)   # :245
        return traj

    def close(self):
        if getattr(self, "handle", None):
            lib().tmvb_flda_destroy(self.handle)
            self.handle = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gpu_train_flda(model: fLDA, device_id: int = 0, **kwargs):
    """`@gpu train!(model::fLDA; kwargs...)`: in the reference this branch does nothing (src/macros.jl:274-275); here the
    model trains on the device and comes back like the LDA branch does (beta / kappa re-normalised in fp64, *_old synced)."""
    if not isinstance(model, fLDA):
        raise ValueError("gpu_train_flda needs an fLDA model.")
    g = gpufLDA(model.corp, model.K, device_id=device_id, _from=model)
    traj = g.train(**kwargs)
    for n in ("topics", "eta", "alpha", "gamma", "Elogtheta", "tau", "elbo"):
        setattr(model, n, getattr(g, n))
    model.beta = g.beta / g.beta.sum(axis=1, keepdims=True); model.beta_old = model.beta.copy(order="F")
    model.kappa = g.kappa / g.kappa.sum(); model.kappa_old = model.kappa.copy()
    model.Elogtheta_old = model.Elogtheta.copy(order="F"); model.tau_old = model.tau.copy()
    g.close()
    return traj


def predict_flda(corp, train_model, iter: int = 10, tol: float | None = None, device_id: int = 0, seed: int = 7) -> fLDA:
    """predict(corp, train_model::fLDA; iter=10, tol=1/K^2)  src/modelutils.jl:858-883 on the device: one pass of the fused
    filtered E-step (phi, tau, gamma, Elogtheta sweeps), no M-step.  As in the reference only alpha, beta and topics come
    from the trained model (:866-868): kappa and eta are those of the fresh fLDA(corp, K) (`seed` stands in for Julia's
    RNG).  The reference's loop tests `vtol`, a name that function never defines (:877); `tol` is used here."""
    K = train_model.K
    tol = 1.0 / K ** 2 if tol is None else tol
    from .lda import _packed
    pc = _packed(corp)
    if pc.V != train_model.V:
        from ._lib import CorpusError
        raise CorpusError("predict corpus and train_model corpus must have identical vocabularies.")
    if tol < 0:
        raise ValueError("tolerance parameter must be nonnegative.")
    if iter < 0:
        raise ValueError("iteration parameter must be nonnegative.")
    host = fLDA(pc, K, seed)
    host.alpha = np.array(train_model.alpha, dtype=np.float64)
    host.beta = np.asfortranarray(train_model.beta); host.beta_old = host.beta.copy(order="F")
    host.topics = train_model.topics
    g = gpufLDA(None, K, device_id=device_id, _from=host)
    g.estep(iter, tol)
    g.update_host()
    for n in ("gamma", "Elogtheta", "Elogtheta_old", "tau", "tau_old"):
        setattr(host, n, getattr(g, n))
    g.close()
    return host
