"""
Host-side mirror of the reference's CTPF / gpuCTPF interface above the C ABI.

    CTPF(corp, K)                     src/CTPF.jl:6-108     host fp64 state (Hebrew-letter field names kept)
    gpuCTPF(corp, K)                  src/gpuCTPF.jl:6-153  device-backed model
      .update_buffer() / .update_host()     src/modelutils.jl:438-494 / :539-570
      .estep(viter, vtol)                   update_xi!/update_phi!/update_zayin!/update_gimel! sweeps + update_he!(d),
                                            update_alef!(d); CPU-path semantics src/CTPF.jl:353-365
      .mstep()                              update_he!/alef!/dalet!/het!/bet!/vav! in that order, src/CTPF.jl:366-371
      .train(iter=150, tol=1.0, viter=10, vtol=1/K^2, checkelbo=Inf, printelbo=True)   src/gpuCTPF.jl:677-705
      .recommend(scores=True)               scores / drecs / urecs of the end of train!, src/CTPF.jl:379-399 (device GEMM +
                                            segmented sort; the reference does it on the host, src/gpuCTPF.jl:711-731)
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from ._lib import TopicModelError, check, lib, P_i64, VP
from .corpus import dirichlet_rows
from .lda import DeviceContext, DeviceCorpus, _F, _packed, _pd, _print_delbo, _topic_orders, _validate_train_args


class CTPF:
    """Host (fp64) CTPF state, src/CTPF.jl:6-108.  scores / drecs / urecs are filled by train! (1-based ids, like topics)."""

    def __init__(self, corp, K: int, seed: int = 7):
        if not (isinstance(K, (int, np.integer)) and K > 0):
            raise ValueError("number of topics must be a positive integer.")
        self.corp = _packed(corp)
        self.K, self.M, self.V, self.U = int(K), self.corp.M, self.corp.V, self.corp.U
        self.N, self.C, self.R = self.corp.N, self.corp.C, np.diff(self.corp.rdr_ptr)
        K, M, V, U = self.K, self.M, self.V, self.U
        self.topics = [np.arange(1, V + 1) for _ in range(K)]
        self.a = self.b = self.c = self.d = self.e = self.f = self.g = self.h = 0.1                # :81
        self.alef = np.asfortranarray(np.exp(dirichlet_rows(K, V, seed) - 0.5))                    # :83
        self.alef_old = self.alef.copy(order="F")
        self.he = np.ones((K, U), order="F"); self.he_old = self.he.copy(order="F")
        for n in ("bet", "vav", "dalet", "het"):
            setattr(self, n, np.ones(K)); setattr(self, n + "_old", np.ones(K))
        for n in ("gimel", "zayin"):
            setattr(self, n, np.ones((K, M), order="F")); setattr(self, n + "_old", np.ones((K, M), order="F"))
        self.elbo = 0.0
        # libs[u] = documents user u has read (src/CTPF.jl:62-65), 1-based; recommendations start unranked (:67-79)
        self.libs = [[] for _ in range(U)]
        for d in range(M):
            for u in self.corp.readers[self.corp.rdr_ptr[d]:self.corp.rdr_ptr[d + 1]]:
                self.libs[int(u)].append(d + 1)
        self.scores = None
        self.drecs = None
        self.urecs = None

    def hyper(self):
        return np.array([self.a, self.b, self.c, self.d, self.e, self.f, self.g, self.h], dtype=np.float64)


def check_model_ctpf(model):
    """check_model(::CTPF) src/modelutils.jl:181-253 (array form)."""
    for n in "abcdefgh":
        if not getattr(model, n) > 0:
            raise TopicModelError(f"{n} must be positive.")
    for n in ("alef", "he", "bet", "vav", "dalet", "het", "gimel", "zayin"):
        a = getattr(model, n)
        if not np.all(np.isfinite(a)):
            raise TopicModelError(f"{n} must be finite.")
        if not np.all(a > 0):
            raise TopicModelError(f"{n} must be positive.")
    if not math.isfinite(model.elbo):
        raise TopicModelError("elbo must be finite.")


class gpuCTPF:
    """GPU accelerated collaborative topic Poisson factorization model (src/gpuCTPF.jl:6-153) on libtmvb_hip.so."""

    _FIELDS = ("corp", "K", "M", "V", "U", "N", "C", "R", "topics", "a", "b", "c", "d", "e", "f", "g", "h", "alef", "alef_old",
               "he", "he_old", "bet", "bet_old", "vav", "vav_old", "dalet", "dalet_old", "het", "het_old", "gimel", "gimel_old",
               "zayin", "zayin_old", "elbo", "libs", "scores", "drecs", "urecs")

    def __init__(self, corp, K: int, seed: int = 7, ctx: DeviceContext | None = None, device_id: int = 0, stream=None,
                 _from: CTPF | None = None):
        host = _from if _from is not None else CTPF(corp, K, seed)
        for k in self._FIELDS:
            setattr(self, k, getattr(host, k))
        self.ctx = ctx or DeviceContext(device_id, stream)
        self.dcorp = DeviceCorpus(self.ctx, self.corp)
        self.handle = VP()
        check(lib().tmvb_ctpf_create(self.ctx.handle, self.dcorp.handle, C.c_int32(self.K), C.byref(self.handle)))
        self.update_buffer()

    hyper = CTPF.hyper

    def update_buffer(self):
        K, M, V, U = self.K, self.M, self.V, self.U
        hy = self.hyper()
        elbo = C.c_double(float(self.elbo))
        vec = lambda x: np.ascontiguousarray(x, dtype=np.float64)
        check(lib().tmvb_ctpf_set_state(self.handle, _pd(hy), _pd(_F(self.alef, (K, V))), _pd(_F(self.he, (K, U))),
                                        _pd(vec(self.bet)), _pd(vec(self.vav)), _pd(vec(self.dalet)), _pd(vec(self.het)),
                                        _pd(_F(self.gimel, (K, M))), _pd(_F(self.zayin, (K, M))), C.byref(elbo)))
        # the *_old fields feed update_elbo! (src/CTPF.jl:239-240) and are part of update_buffer! (src/modelutils.jl:474-493)
        check(lib().tmvb_ctpf_set_state_old(self.handle, _pd(_F(self.alef_old, (K, V))), _pd(_F(self.he_old, (K, U))),
                                            _pd(vec(self.bet_old)), _pd(vec(self.vav_old)), _pd(vec(self.dalet_old)),
                                            _pd(vec(self.het_old)), _pd(_F(self.gimel_old, (K, M))), _pd(_F(self.zayin_old, (K, M)))))

    def update_host(self):
        K, M, V, U = self.K, self.M, self.V, self.U
        self.alef = np.empty((K, V), order="F"); self.alef_old = np.empty((K, V), order="F")
        self.he = np.empty((K, U), order="F"); self.he_old = np.empty((K, U), order="F")
        rates = np.empty(8 * K)
        self.gimel = np.empty((K, M), order="F"); self.gimel_old = np.empty((K, M), order="F")
        self.zayin = np.empty((K, M), order="F"); self.zayin_old = np.empty((K, M), order="F")
        elbo = C.c_double(0.0)
        check(lib().tmvb_ctpf_get_state(self.handle, _pd(self.alef), _pd(self.alef_old), _pd(self.he), _pd(self.he_old),
                                        _pd(rates), _pd(self.gimel), _pd(self.gimel_old), _pd(self.zayin), _pd(self.zayin_old),
                                        C.byref(elbo)))
        for q, n in enumerate(("bet", "vav", "dalet", "het", "bet_old", "vav_old", "dalet_old", "het_old")):
            setattr(self, n, rates[q * K:(q + 1) * K].copy())
        self.elbo = elbo.value

    def estep(self, viter: int = 10, vtol: float | None = None):
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check(lib().tmvb_ctpf_estep(self.handle, C.c_int32(viter), C.c_double(vtol)))

    def reduce_docs(self): check(lib().tmvb_ctpf_reduce_docs(self.handle))
    def mstep(self): check(lib().tmvb_ctpf_mstep(self.handle))

    def elbo_form(self) -> int:
        """1 if the last update_elbo! took the decomposed form (parts left behind by the iteration itself), 0 for the table form."""
        f = C.c_int32(0)
        check(lib().tmvb_ctpf_elbo_form(self.handle, C.byref(f)))
        return f.value

    def update_elbo(self) -> float:
        out = C.c_double(0.0)
        check(lib().tmvb_ctpf_update_elbo(self.handle, C.byref(out)))
        self.elbo = out.value
        return out.value

    def update_elbo_parts(self):
        """(per-document part summed over this context's documents, global (beta, eta) part) for document-sharded hosts."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        check(lib().tmvb_ctpf_update_elbo_parts(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stats(self):
        p, n = VP(), C.c_int64(0)
        check(lib().tmvb_ctpf_stats(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_stats(self, dev_ptr: int, n_f32: int):
        check(lib().tmvb_ctpf_bind_stats(self.handle, VP(dev_ptr), C.c_int64(n_f32)))

    def set_distributed(self, distributed: bool = True):
        check(lib().tmvb_ctpf_set_distributed(self.handle, C.c_int32(1 if distributed else 0)))

    def set_comm(self, comm):
        """Attach a communicator (comm.py): document-sharded train!, every rank calls train() with the same arguments."""
        self._comm = comm
        check(lib().tmvb_ctpf_set_comm(self.handle, comm.handle if comm is not None else VP(None)))

    def sweep_hist(self, nbins: int = 11):
        h = np.zeros(nbins, dtype=np.int64)
        check(lib().tmvb_ctpf_sweep_hist(self.handle, h.ctypes.data_as(P_i64), C.c_int32(nbins)))
        return h

    def doc_sweeps(self):
        """Sweeps each document ran in the last E-step (uint8 per document, corpus order)."""
        out = np.zeros(max(self.M, 1), dtype=np.uint8)
        check(lib().tmvb_ctpf_doc_sweeps(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out[:self.M]

    def last_estep_ms(self) -> float:
        ms = C.c_float(0.0)
        check(lib().tmvb_ctpf_last_estep_ms(self.handle, C.byref(ms)))
        return ms.value

    def synchronize(self):
        self.ctx.synchronize()

    def recommend(self, scores: bool = True):
        """scores (M x U fp64), drecs[d], urecs[u] from the device-resident state (src/CTPF.jl:379-399).  Ids are 1-based.
        Returns (ms_scores, ms_rank): device time of the score pass and of the two segmented sorts."""
        M, U = self.M, self.U
        if M == 0 or U == 0:
            self.scores = np.zeros((M, U), order="F"); self.drecs = [np.arange(1, U + 1) for _ in range(M)]
            self.urecs = [np.arange(1, M + 1) for _ in range(U)]
            return 0.0, 0.0
        sc = np.empty((M, U), order="F") if scores else None
        dr = np.empty((M, U), dtype=np.int32); dc = np.empty(M, dtype=np.int32)
        ur = np.empty((U, M), dtype=np.int32); uc = np.empty(U, dtype=np.int32)
        ms0, ms1 = C.c_float(0.0), C.c_float(0.0)
        p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        check(lib().tmvb_ctpf_recommend(self.handle, _pd(sc) if scores else None, p32(dr), p32(dc), p32(ur), p32(uc),
                                        C.byref(ms0), C.byref(ms1)))
        if scores:
            self.scores = sc
        self.drecs = [dr[d, :dc[d]] + 1 for d in range(M)]
        self.urecs = [ur[u, :uc[u]] + 1 for u in range(U)]
        return ms0.value, ms1.value

    def train(self, iter: int = 150, tol: float = 1.0, viter: int = 10, vtol: float | None = None, checkelbo=1,
              printelbo: bool = True, recs: bool = True):
        """train!(model::gpuCTPF; ...) src/gpuCTPF.jl:677-705.  recs=False skips the M x U recommendation tail (:711-731)."""
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        check_model_ctpf(self)
        _validate_train_args([tol, vtol], [iter, viter], checkelbo)
        self.update_buffer()
        ce = 0 if checkelbo == math.inf else int(checkelbo)
        traj = np.full(max(iter, 1), np.nan)
        done, base = C.c_int32(0), C.c_double(float(self.elbo))
        check(lib().tmvb_ctpf_train(self.handle, C.c_int32(iter), C.c_double(tol), C.c_int32(viter), C.c_double(vtol),
                                    C.c_int32(ce), _pd(traj), C.byref(done), C.byref(base)))
        self.elbo_baseline = base.value
        if iter > 0:
            self.update_host()
        if printelbo and ce:
            _print_delbo(traj[:done.value], base.value)
        Ebeta = self.alef / self.bet[:, None]                                                     # :707-708
        self.topics = _topic_orders(self.ctx, Ebeta)   # reverse(sortperm(.))
        if recs:
            self.recommend()                                                                      # :711-731
        return traj[:done.value]

    def close(self):
        if getattr(self, "handle", None):
            lib().tmvb_ctpf_destroy(self.handle)
            self.handle = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gpu_train_ctpf(model: CTPF, device_id: int = 0, **kwargs):
    """`@gpu train!(model::CTPF; kwargs...)` (src/macros.jl:197-272)."""
    g = gpuCTPF(None, model.K, device_id=device_id, _from=model)
    traj = g.train(**kwargs)
    for n in ("topics", "alef", "he", "bet", "vav", "dalet", "het", "gimel", "zayin", "elbo", "scores", "drecs", "urecs"):
        setattr(model, n, getattr(g, n))
    for n in ("alef", "he", "bet", "vav", "dalet", "het", "gimel", "zayin"):
        setattr(model, n + "_old", np.array(getattr(g, n), copy=True, order="F"))
    g.close()
    return traj


def topicdist_ctpf(model, d):
    """topicdist(model::Union{CTPF, gpuCTPF}, d)  src/modelutils.jl:960-965: gimel[d] / sum(gimel[d]) (d is 1-based like the
    reference; a list / range of indices returns a list, :972-983)."""
    if not isinstance(d, (int, np.integer)):
        return [topicdist_ctpf(model, int(x)) for x in d]
    if not (1 <= d <= model.M):
        from ._lib import CorpusError
        raise CorpusError("document index outside corpus range.")
    g = model.gimel[:, d - 1]
    return g / g.sum()
