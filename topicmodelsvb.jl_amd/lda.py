"""
Host-side mirror of the reference's LDA / gpuLDA interface above the C ABI.

    LDA(corp, K)                       src/LDA.jl:6-47     host fp64 state (what @gpu reads/writes)
    gpuLDA(corp, K)                    src/gpuLDA.jl:6-85  device-backed model
      .update_buffer() / .update_host()   src/modelutils.jl:370-397 / :501-516
      .estep(viter, vtol)                 update_phi!/update_gamma!/update_Elogtheta! sweeps + update_beta!(d),
                                          CPU-path semantics src/LDA.jl:170-180
      .update_beta() .update_alpha(niter, ntol) .update_elbo()   src/gpuLDA.jl:201, :132, :120
      .train(iter=150, tol=1.0, niter=1000, ntol=1/K^2, viter=10, vtol=1/K^2, checkelbo=1, printelbo=True)
                                          src/gpuLDA.jl:347-376
    gpu_train(model, **kwargs)         `@gpu train!(model; kwargs...)`  src/macros.jl:106-150

All compute goes through libtmvb_hip.so; there is no CPU fallback in this module.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import TopicModelError, check, lib, P_dbl, P_i32, P_i64, VP
from .corpus import Corpus, PackedCorpus, check_corp, dirichlet_rows

EPSILON = 2.0 ** -99          # src/utils.jl:3
EULERGAMMA = 0.5772156649015329


def _digamma_int(k: int) -> float:
    # psi(k) for a positive integer: -eulergamma + H_{k-1}
    return -EULERGAMMA + sum(1.0 / j for j in range(1, k))


def _pd(a):
    return a.ctypes.data_as(P_dbl)


def _F(a, shape):
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if a.shape != tuple(shape):
        raise TopicModelError(f"expected an array of size {tuple(shape)}, got {a.shape}.")
    return a


def _packed(corp):
    if isinstance(corp, PackedCorpus):
        return corp
    check_corp(corp)
    return PackedCorpus.from_corpus(corp)


class LDA:
    """Host (fp64) LDA state with the reference's field names, src/LDA.jl:6-47."""

    def __init__(self, corp, K: int, seed: int = 7):
        if not (isinstance(K, (int, np.integer)) and K > 0):
            raise ValueError("number of topics must be a positive integer.")
        self.corp = _packed(corp)
        self.K, self.M, self.V = int(K), self.corp.M, self.corp.V
        self.N = self.corp.N
        self.C = self.corp.C
        K, M, V = self.K, self.M, self.V
        self.topics = [np.arange(1, V + 1) for _ in range(K)]
        self.alpha = np.ones(K)
        self.beta = dirichlet_rows(K, V, seed)                       # :35 (Julia RNG in the reference)
        self.beta_old = self.beta.copy(order="F")
        self.beta_temp = np.zeros((K, V), order="F")
        e0 = -EULERGAMMA - _digamma_int(K)                           # :38
        self.Elogtheta = np.full((K, M), e0, order="F")
        self.Elogtheta_old = self.Elogtheta.copy(order="F")
        self.gamma = np.ones((K, M), order="F")
        self.elbo = 0.0


def check_model(model, rtol: float = 1.5e-8):
    """check_model(::LDA) / (::gpuLDA)  src/modelutils.jl:39-67, :255-279 (array form).
    rtol mirrors isapprox's default sqrt(eps(T)): 1.5e-8 for Float64 fields, 3.5e-4 for the
    Float32-derived fields of a gpu model."""
    K, M, V = model.K, model.M, model.V
    if model.alpha.shape != (K,):
        raise TopicModelError("alpha must be of length K.")
    if not np.all(np.isfinite(model.alpha)):
        raise TopicModelError("alpha must be finite.")
    if not np.all(model.alpha > 0):
        raise TopicModelError("alpha must be positive.")
    if model.beta.shape != (K, V):
        raise TopicModelError("beta must be of size (K, V).")
    if V and not (np.all(model.beta >= 0) and np.allclose(model.beta.sum(axis=1), 1.0, rtol=rtol, atol=0)):
        raise TopicModelError("beta must be a right stochastic matrix.")
    for name in ("Elogtheta", "gamma"):
        a = getattr(model, name)
        if a.shape != (K, M):
            raise TopicModelError(f"{name} must contain vectors of length K.")
        if not np.all(np.isfinite(a)):
            raise TopicModelError(f"{name} must be finite.")
    if not np.all(model.Elogtheta <= 0):
        raise TopicModelError("Elogtheta must be nonpositive.")
    if not np.all(model.gamma > 0):
        raise TopicModelError("gamma must be positive.")
    if not math.isfinite(model.elbo):
        raise TopicModelError("elbo must be finite.")


class DeviceContext:
    """One GPU + one stream (replaces cl.create_compute_context(), src/gpuLDA.jl:64)."""

    def __init__(self, device_id: int = 0, stream=None):
        L = lib()
        self.handle = VP()
        check(L.tmvb_ctx_create(C.c_int32(device_id), VP(stream) if stream else VP(None), C.byref(self.handle)))
        self.device_id = device_id

    def synchronize(self):
        check(lib().tmvb_ctx_synchronize(self.handle))

    def timing_event(self):
        """A HIP timing event on this context's stream without the system-scope fence of a default event (tmvb_event_create)."""
        return TimingEvent(self)

    def close(self):
        if self.handle:
            lib().tmvb_ctx_destroy(self.handle)
            self.handle = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TimingEvent:
    """tmvb_event_*: record() on the context's stream; a.elapsed_ms(b) waits for b."""

    def __init__(self, ctx):
        self.handle = VP()
        check(lib().tmvb_event_create(ctx.handle, C.byref(self.handle)))

    def record(self):
        check(lib().tmvb_event_record(self.handle))

    def elapsed_ms(self, stop) -> float:
        ms = C.c_float(0.0)
        check(lib().tmvb_event_elapsed_ms(self.handle, stop.handle, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if self.handle:
            lib().tmvb_event_destroy(self.handle)
            self.handle = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceCorpus:
    """Corpus half of update_buffer! (src/modelutils.jl:370-388)."""

    def __init__(self, ctx: DeviceContext, corp: PackedCorpus):
        self.ctx, self.corp = ctx, corp
        self.handle = VP()
        has_r = corp.U > 0
        check(lib().tmvb_corpus_create(ctx.handle, C.c_int64(corp.M), C.c_int64(corp.V), C.c_int64(corp.U),
                                       corp.doc_ptr.ctypes.data_as(P_i64), corp.terms.ctypes.data_as(P_i32),
                                       corp.counts.ctypes.data_as(P_i32),
                                       corp.rdr_ptr.ctypes.data_as(P_i64) if has_r else None,
                                       corp.readers.ctypes.data_as(P_i32) if has_r else None,
                                       corp.ratings.ctypes.data_as(P_i32) if has_r else None,
                                       C.byref(self.handle)))

    def info(self):
        out = _lib.CorpusInfo()
        check(lib().tmvb_corpus_info(self.handle, C.byref(out)))
        return {n: getattr(out, n) for n, _ in out._fields_}

    def close(self):
        if self.handle:
            lib().tmvb_corpus_destroy(self.handle)
            self.handle = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _validate_train_args(tols, iters, checkelbo):
    # src/gpuLDA.jl:349-351
    if not all(t >= 0 for t in tols):
        raise ValueError("tolerance parameters must be nonnegative.")
    if not all(i >= 0 for i in iters):
        raise ValueError("iteration parameters must be nonnegative.")
    if not ((isinstance(checkelbo, (int, np.integer)) and checkelbo > 0) or checkelbo == math.inf):
        raise ValueError("checkelbo parameter must be a positive integer or Inf.")


def _topic_orders(ctx, B):
    """model.topics = [reverse(sortperm(vec(B[i,:]))) for i in 1:K]  (src/gpuLDA.jl:374 and its siblings), 1-based: tmvb_topic_order, one
    segmented sort on the device (K numpy sorts one after the other were 65 ms of a 209 ms train!(iter=150) call at K = 50, V = 25 319)."""
    B = np.asfortranarray(B, dtype=np.float64)
    K, V = B.shape
    out = np.empty((K, V), dtype=np.int32)
    check(lib().tmvb_topic_order(ctx.handle, _pd(B), C.c_int32(K), C.c_int64(V), out.ctypes.data_as(P_i32)))
    return [out[i].astype(np.int64) for i in range(K)]


def _print_delbo(traj, baseline):
    """The printing half of check_elbo! (src/modelutils.jl:578-579): one line per checked iteration, the first one
    against the ELBO evaluated before the first iteration."""
    prev = baseline
    for k, e in enumerate(traj, start=1):
        if not np.isnan(e):
            print(k, " ∆elbo: ", round(e - prev, 3))
            prev = e


class gpuLDA:
    """GPU accelerated latent Dirichlet allocation model (src/gpuLDA.jl:6-85) on libtmvb_hip.so."""

    def __init__(self, corp, K: int, seed: int = 7, ctx: DeviceContext | None = None, device_id: int = 0, stream=None):
        if not (isinstance(K, (int, np.integer)) and K > 0):
            raise ValueError("number of topics must be a positive integer.")
        host = LDA(corp, K, seed)
        self.__dict__.update({k: getattr(host, k) for k in ("corp", "K", "M", "V", "N", "C", "topics", "alpha", "beta",
                                                             "beta_old", "Elogtheta", "Elogtheta_old", "gamma", "elbo")})
        self.ctx = ctx or DeviceContext(device_id, stream)
        self.dcorp = DeviceCorpus(self.ctx, self.corp)
        self.handle = VP()
        check(lib().tmvb_lda_create(self.ctx.handle, self.dcorp.handle, C.c_int32(self.K), C.byref(self.handle)))
        self.M_total = self.M
        self.update_buffer()

    # ---- marshalling
    def update_buffer(self):
        """State half of update_buffer! (src/modelutils.jl:390-396): host fields -> device."""
        K, M, V = self.K, self.M, self.V
        a = np.ascontiguousarray(self.alpha, dtype=np.float64)
        if a.shape != (K,):
            raise TopicModelError("alpha must be of length K.")
        b, bo = _F(self.beta, (K, V)), _F(self.beta_old, (K, V))
        g, e, eo = _F(self.gamma, (K, M)), _F(self.Elogtheta, (K, M)), _F(self.Elogtheta_old, (K, M))
        elbo = C.c_double(float(self.elbo))
        check(lib().tmvb_lda_set_state(self.handle, _pd(a), _pd(b), _pd(bo), _pd(g), _pd(e), _pd(eo), C.byref(elbo)))

    def update_host(self):
        """update_host! (src/modelutils.jl:501-516): device -> host fields (phi is never materialised)."""
        K, M, V = self.K, self.M, self.V
        self.alpha = np.empty(K)
        self.beta = np.empty((K, V), order="F"); self.beta_old = np.empty((K, V), order="F")
        self.gamma = np.empty((K, M), order="F")
        self.Elogtheta = np.empty((K, M), order="F"); self.Elogtheta_old = np.empty((K, M), order="F")
        elbo = C.c_double(0.0)
        check(lib().tmvb_lda_get_state(self.handle, _pd(self.alpha), _pd(self.beta), _pd(self.beta_old), _pd(self.gamma),
                                       _pd(self.Elogtheta), _pd(self.Elogtheta_old), C.byref(elbo)))
        self.elbo = elbo.value

    # ---- device operators
    def estep(self, viter: int = 10, vtol: float | None = None):
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check(lib().tmvb_lda_estep(self.handle, C.c_int32(viter), C.c_double(vtol)))

    def reduce_docs(self):
        check(lib().tmvb_lda_reduce_docs(self.handle))

    def estep_allreduce(self, viter: int = 10, vtol: float | None = None):
        """estep + reduce_docs + the statistics all-reduce over the attached communicator, the collective issued in vocabulary
        slabs under the last statistics pass (tmvb_lda_estep_allreduce, include/tmvb.h)."""
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check(lib().tmvb_lda_estep_allreduce(self.handle, C.c_int32(viter), C.c_double(vtol)))

    def update_beta(self):
        check(lib().tmvb_lda_update_beta(self.handle))

    def update_alpha(self, niter: int = 1000, ntol: float | None = None):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        check(lib().tmvb_lda_update_alpha(self.handle, C.c_int32(niter), C.c_double(ntol)))

    def update_elbo(self) -> float:
        out = C.c_double(0.0)
        check(lib().tmvb_lda_update_elbo(self.handle, C.byref(out)))
        self.elbo = out.value
        return out.value

    def stats(self):
        """(device pointer, n_float32) of the packed sufficient statistics [S | Elogtheta_sum]."""
        p, n = VP(), C.c_int64(0)
        check(lib().tmvb_lda_stats(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_stats(self, dev_ptr: int, n_f32: int):
        check(lib().tmvb_lda_bind_stats(self.handle, VP(dev_ptr), C.c_int64(n_f32)))

    def set_distributed(self, M_total: int, distributed: bool = True):
        self.M_total = int(M_total)
        check(lib().tmvb_lda_set_distributed(self.handle, C.c_int64(M_total), C.c_int32(1 if distributed else 0)))

    def sweep_hist(self, nbins: int = 11):
        h = np.zeros(nbins, dtype=np.int64)
        check(lib().tmvb_lda_sweep_hist(self.handle, h.ctypes.data_as(P_i64), C.c_int32(nbins)))
        return h

    def estep_launches(self) -> int:
        n = C.c_int32(0)
        check(lib().tmvb_lda_estep_launches(self.handle, C.byref(n)))
        return n.value

    def elbo_form(self) -> int:
        """1 if the last update_elbo! took the decomposed form (parts left behind by the iteration itself), 0 for the token walk."""
        f = C.c_int32(0)
        check(lib().tmvb_lda_elbo_form(self.handle, C.byref(f)))
        return f.value

    def doc_sweeps(self):
        """Sweeps each document ran in the last E-step (uint8 per document, corpus order)."""
        out = np.zeros(max(self.M, 1), dtype=np.uint8)
        check(lib().tmvb_lda_doc_sweeps(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out[:self.M]

    def last_estep_ms(self) -> float:
        ms = C.c_float(0.0)
        check(lib().tmvb_lda_last_estep_ms(self.handle, C.byref(ms)))
        return ms.value

    def synchronize(self):
        self.ctx.synchronize()

    def set_comm(self, comm, M_total: int):
        """Attach a communicator (comm.py): this model's corpus is one document shard of M_total documents and
        train() becomes the sharded train! -- every rank calls it with the same arguments."""
        self.M_total = int(M_total) if comm is not None else self.M
        self._comm = comm
        check(lib().tmvb_lda_set_comm(self.handle, comm.handle if comm is not None else VP(None), C.c_int64(self.M_total)))

    # ---- train!
    def train(self, iter: int = 150, tol: float = 1.0, niter: int = 1000, ntol: float | None = None, viter: int = 10,
              vtol: float | None = None, checkelbo=1, printelbo: bool = True):
        """train!(model::gpuLDA; ...) src/gpuLDA.jl:347-376.  Returns the ELBO trajectory."""
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check_model(self, rtol=3.5e-4)
        _validate_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
        self.update_buffer()
        ce = 0 if checkelbo == math.inf else int(checkelbo)
        traj = np.full(max(iter, 1), np.nan)
        done, base = C.c_int32(0), C.c_double(float(self.elbo))
        check(lib().tmvb_lda_train(self.handle, C.c_int32(iter), C.c_double(tol), C.c_int32(niter), C.c_double(ntol),
                                   C.c_int32(viter), C.c_double(vtol), C.c_int32(ce), _pd(traj), C.byref(done), C.byref(base)))
        traj = traj[:done.value]
        self.elbo_baseline = base.value
        if iter > 0:
            self.update_host()                                       # :373
        if printelbo and ce:
            _print_delbo(traj, base.value)
        self.topics = _topic_orders(self.ctx, self.beta)              # :374
        return traj

    def close(self):
        if getattr(self, "handle", None):
            lib().tmvb_lda_destroy(self.handle)
            self.handle = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gpu_train(model: LDA, device_id: int = 0, **kwargs):
    """`@gpu train!(model; kwargs...)` for LDA (src/macros.jl:113-150): copy the host model onto the
    device, train there, copy back, renormalise beta in fp64."""
    if not isinstance(model, LDA):
        raise ValueError("GPU acceleration only applies to LDA / CTM / CTPF models.")
    g = gpuLDA.__new__(gpuLDA)
    for k in ("corp", "K", "M", "V", "N", "C", "topics", "alpha", "beta", "beta_old", "Elogtheta", "Elogtheta_old", "gamma", "elbo"):
        setattr(g, k, getattr(model, k))
    g.ctx = DeviceContext(device_id)
    g.dcorp = DeviceCorpus(g.ctx, g.corp)
    g.handle = VP()
    check(lib().tmvb_lda_create(g.ctx.handle, g.dcorp.handle, C.c_int32(g.K), C.byref(g.handle)))
    g.M_total = g.M
    traj = g.train(**kwargs)
    model.topics = g.topics
    model.alpha = g.alpha
    model.beta = g.beta / g.beta.sum(axis=1, keepdims=True)          # src/macros.jl:147
    model.beta_old = model.beta.copy(order="F")                      # :148
    model.Elogtheta = g.Elogtheta
    model.Elogtheta_old = g.Elogtheta.copy(order="F")                # :142
    model.gamma = g.gamma
    model.elbo = g.elbo
    g.close()
    return traj


def predict(corp, train_model, iter: int = 10, tol: float | None = None, device_id: int = 0) -> LDA:
    """predict(corp, train_model::Union{LDA, gpuLDA}; iter=10, tol=1/K^2)  src/modelutils.jl:831-855:
    topic distributions of unseen documents with the trained alpha / beta frozen -- one pass of the fused
    E-step kernel (<= iter sweeps per document, same exit rule), no M-step."""
    K = train_model.K
    tol = 1.0 / K ** 2 if tol is None else tol
    pc = _packed(corp)
    if pc.V != train_model.V:
        from ._lib import CorpusError
        raise CorpusError("predict corpus and train_model corpus must have identical vocabularies.")
    if tol < 0:
        raise ValueError("tolerance parameter must be nonnegative.")
    if iter < 0:
        raise ValueError("iteration parameter must be nonnegative.")
    g = gpuLDA(pc, K, device_id=device_id)
    g.alpha = np.array(train_model.alpha, dtype=np.float64)
    g.beta = np.asfortranarray(train_model.beta); g.beta_old = g.beta.copy(order="F")
    g.update_buffer()
    g.estep(iter, tol)
    g.update_host()
    out = LDA(pc, K)
    for n in ("alpha", "beta", "beta_old", "gamma", "Elogtheta", "Elogtheta_old"):
        setattr(out, n, getattr(g, n))
    out.topics = train_model.topics
    g.close()
    return out


def topicdist(model, d):
    """topicdist(model::Union{LDA, gpuLDA}, d)  src/modelutils.jl:946-951 (d is 1-based like the reference;
    a list / range of indices returns a list)."""
    if not isinstance(d, (int, np.integer)):
        return [topicdist(model, int(x)) for x in d]
    if not (1 <= d <= model.M):
        from ._lib import CorpusError
        raise CorpusError("document index outside corpus range.")
    g = model.gamma[:, d - 1]
    return g / g.sum()
