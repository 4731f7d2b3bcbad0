"""
Document-sharded data parallelism (new functionality: the reference is single-device; closest
precedent is the v0.6 batch accumulation `newbeta +=`, v0.6/src/gpuLDA.jl:200-225).

One process per GPU.  Documents are conditionally independent given the globals, so each rank runs
the fused E-step on its contiguous, nnz-balanced document range; the only exchange per outer
iteration is ONE all-reduce (sum, fp32) of the packed sufficient statistics
[S (K*V) | Elogtheta_sum (K)] -- RCCL over xGMI through torch.distributed's "nccl" backend on GPUs,
gloo in the CPU tests -- after which every rank runs the identical M-step (no broadcast).
The ELBO, when checked, is a second tiny fp64 all-reduce of the ranks' partial sums.

`ShardedLDA` only needs an engine with the operator interface below; the product engine is
`HipLDAEngine` (libtmvb_hip.so).  The CPU tests plug in an oracle-backed engine to exercise this
host logic under gloo without a GPU.
"""
from __future__ import annotations

import math

import numpy as np


class HipLDAEngine:
    """gpuLDA on this rank's shard, statistics buffer owned by torch so the collective can see it."""

    def __init__(self, shard_corpus, K, beta0, M_total, device_index, distributed):
        import torch
        from .lda import DeviceContext, gpuLDA
        self.torch = torch
        torch.cuda.set_device(device_index)
        # a dedicated non-default stream shared by the engine's kernels, torch's collectives and the
        # bench's HIP events (the legacy null stream has handle 0, which the C ABI reads as "create
        # your own stream")
        self.stream = torch.cuda.Stream(device=device_index)
        assert self.stream.cuda_stream != 0
        self.ctx = DeviceContext(device_index, self.stream.cuda_stream)
        self.model = gpuLDA(shard_corpus, K, ctx=self.ctx)
        self.model.beta = np.asfortranarray(beta0); self.model.beta_old = self.model.beta.copy(order="F")
        self.model.update_buffer()
        n = K * shard_corpus.V + K
        # Only a sharded engine binds a torch tensor (the collective has to see the buffer).  One context keeps the library's own buffer: a bound one
        # switches off the E-step's split of the last statistics pass into a buffer of its own (tmvb_lda.hip: split_out needs own_stats -- a host that
        # reads its tensor between estep and update_beta! must see all of S), which is the plan plain gpuLDA + train! runs with.
        self.stats = None
        if distributed:
            self.stats = torch.zeros(n, dtype=torch.float32, device=f"cuda:{device_index}")
            self.model.bind_stats(self.stats.data_ptr(), n)
        self.model.set_distributed(M_total, distributed)
        self.device = torch.device(f"cuda:{device_index}")

    def estep(self, viter, vtol): self.model.estep(viter, vtol)
    def reduce_docs(self): self.model.reduce_docs()
    def stats_tensor(self):
        if self.stats is None:
            raise RuntimeError("HipLDAEngine(distributed=False) has no bound statistics tensor; use model.stats()")
        return self.stats
    def update_beta(self): self.model.update_beta()
    def update_alpha(self, niter, ntol): self.model.update_alpha(niter, ntol)
    def local_elbo(self): return self.model.update_elbo()
    def synchronize(self): self.model.synchronize()


class ShardedLDA:
    """train! (src/LDA.jl:161-187 semantics) over document shards."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.elbo = 0.0
        try:
            import torch.distributed as dist
            self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        except Exception:
            self.dist = None

    @property
    def world_size(self):
        return self.dist.get_world_size(self.group) if self.dist else 1

    def allreduce_stats(self):
        if self.dist and self.world_size > 1:
            st = getattr(self.engine, "stream", None)
            if st is not None:
                with self.engine.torch.cuda.stream(st):      # order the collective with the engine's kernels
                    self.dist.all_reduce(self.engine.stats_tensor(), op=self.dist.ReduceOp.SUM, group=self.group)
            else:
                self.dist.all_reduce(self.engine.stats_tensor(), op=self.dist.ReduceOp.SUM, group=self.group)

    def iterate(self, niter, ntol, viter, vtol):
        """One outer iteration: E-step (local docs) -> all-reduce -> identical M-step on every rank."""
        e = self.engine
        e.estep(viter, vtol)                 # src/LDA.jl:170-180
        e.reduce_docs()                      # :98
        self.allreduce_stats()
        e.update_beta()                      # :181
        e.update_alpha(niter, ntol)          # :182

    def update_elbo(self):
        local = self.engine.local_elbo()
        if self.dist and self.world_size > 1:
            import torch
            t = torch.tensor([local], dtype=torch.float64, device=getattr(self.engine, "device", "cpu"))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)   # local is already synchronised
            local = float(t.item())
        return local

    def train(self, iter=150, tol=1.0, niter=1000, ntol=None, viter=10, vtol=None, checkelbo=1, K=None, on_iter=None):
        K = K or self.engine.model.K
        ntol = 1.0 / K ** 2 if ntol is None else ntol
        vtol = 1.0 / K ** 2 if vtol is None else vtol
        if not all(t >= 0 for t in (tol, ntol, vtol)):
            raise ValueError("tolerance parameters must be nonnegative.")
        if not all(i >= 0 for i in (iter, niter, viter)):
            raise ValueError("iteration parameters must be nonnegative.")
        if not ((isinstance(checkelbo, (int, np.integer)) and checkelbo > 0) or checkelbo == math.inf):
            raise ValueError("checkelbo parameter must be a positive integer or Inf.")
        traj = []
        if checkelbo <= iter:
            self.elbo = self.update_elbo()                              # src/LDA.jl:167
        for k in range(1, iter + 1):
            self.iterate(niter, ntol, viter, vtol)
            if checkelbo != math.inf and k % checkelbo == 0:            # check_elbo! src/modelutils.jl:574-585
                new = self.update_elbo()
                delta = new - self.elbo
                self.elbo = new
                traj.append(new)
                if on_iter:
                    on_iter(k, new)
                if delta < tol:
                    break
            else:
                traj.append(float("nan"))
        return traj


# ---------------------------------------------------------------------------------------------- CTM / CTPF
# Same scheme for the other two models: the packed statistics buffers are
#   CTM   [S (K*V) | sum lambda (K) | sum vsq (K) | scatter (K*K)]     then update_beta!, update_sigma! (previous mu), update_mu!
#   CTPF  [alef_stats (K*V) | he_stats (K*U) | sum gimel (K) | sum zayin (K)]   then the six updates of src/CTPF.jl:366-371
# One all-reduce per outer iteration, identical M-step on every rank.

class _HipEngine:
    """Device-backed model on this rank's shard with a torch-owned statistics buffer on a dedicated stream."""

    def _setup(self, device_index):
        import torch
        from .lda import DeviceContext
        self.torch = torch
        torch.cuda.set_device(device_index)
        self.stream = torch.cuda.Stream(device=device_index)
        assert self.stream.cuda_stream != 0
        self.ctx = DeviceContext(device_index, self.stream.cuda_stream)

    def _bind(self, device_index):
        _, n = self.model.stats()
        self.stats = self.torch.zeros(n, dtype=self.torch.float32, device=f"cuda:{device_index}")
        self.model.bind_stats(self.stats.data_ptr(), n)
        self.device = self.stats.device

    def stats_tensor(self): return self.stats
    def reduce_docs(self): self.model.reduce_docs()
    def synchronize(self): self.model.synchronize()


class HipCTMEngine(_HipEngine):
    def __init__(self, shard_corpus, K, beta0, M_total, device_index, distributed):
        from .ctm import gpuCTM
        self._setup(device_index)
        self.model = gpuCTM(shard_corpus, K, ctx=self.ctx)
        self.model.beta = np.asfortranarray(beta0); self.model.beta_old = self.model.beta.copy(order="F")
        self.model.update_buffer()
        self._bind(device_index)
        self.model.set_distributed(M_total, distributed)

    def estep(self, niter, ntol, viter, vtol): self.model.estep(niter, ntol, viter, vtol)
    def mstep(self):
        self.model.update_beta(); self.model.update_sigma(); self.model.update_mu()     # src/CTM.jl:206-208
    def local_elbo(self): return self.model.update_elbo()


class HipCTPFEngine(_HipEngine):
    def __init__(self, shard_corpus, K, alef0, device_index, distributed):
        from .ctpf import gpuCTPF
        self._setup(device_index)
        self.model = gpuCTPF(shard_corpus, K, ctx=self.ctx)
        self.model.alef = np.asfortranarray(alef0); self.model.alef_old = self.model.alef.copy(order="F")
        self.model.update_buffer()
        self._bind(device_index)
        self.model.set_distributed(distributed)

    def estep(self, viter, vtol): self.model.estep(viter, vtol)
    def mstep(self): self.model.mstep()                                                 # src/CTPF.jl:366-371
    def elbo_parts(self): return self.model.update_elbo_parts()


class _ShardedBase(ShardedLDA):
    def _loop(self, iter, tol, checkelbo, step, on_iter=None):
        if not tol >= 0:
            raise ValueError("tolerance parameters must be nonnegative.")
        if not iter >= 0:
            raise ValueError("iteration parameters must be nonnegative.")
        if not ((isinstance(checkelbo, (int, np.integer)) and checkelbo > 0) or checkelbo == math.inf):
            raise ValueError("checkelbo parameter must be a positive integer or Inf.")
        traj = []
        if checkelbo <= iter:
            self.elbo = self.update_elbo()
        for k in range(1, iter + 1):
            step()
            if checkelbo != math.inf and k % checkelbo == 0:            # check_elbo! src/modelutils.jl:574-585
                new = self.update_elbo()
                delta, self.elbo = new - self.elbo, new
                traj.append(new)
                if on_iter:
                    on_iter(k, new)
                if delta < tol:
                    break
            else:
                traj.append(float("nan"))
        return traj


class ShardedCTM(_ShardedBase):
    """train! (src/CTM.jl:185-213 semantics) over document shards."""

    def iterate(self, niter, ntol, viter, vtol):
        e = self.engine
        e.estep(niter, ntol, viter, vtol)        # src/CTM.jl:194-205
        e.reduce_docs()
        self.allreduce_stats()
        e.mstep()

    def train(self, iter=150, tol=1.0, niter=1000, ntol=None, viter=10, vtol=None, checkelbo=1, on_iter=None):
        K = self.engine.model.K
        ntol = 1.0 / K ** 2 if ntol is None else ntol
        vtol = 1.0 / K ** 2 if vtol is None else vtol
        return self._loop(iter, tol, checkelbo, lambda: self.iterate(niter, ntol, viter, vtol), on_iter)


class ShardedCTPF(_ShardedBase):
    """train! (src/CTPF.jl:344-376 semantics, without the recommendation tail) over document shards."""

    def iterate(self, viter, vtol):
        e = self.engine
        e.estep(viter, vtol)                     # src/CTPF.jl:353-365
        e.reduce_docs()
        self.allreduce_stats()
        e.mstep()

    def update_elbo(self):
        doc_part, global_part = self.engine.elbo_parts()
        if self.dist and self.world_size > 1:
            import torch
            t = torch.tensor([doc_part], dtype=torch.float64, device=getattr(self.engine, "device", "cpu"))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            doc_part = float(t.item())
        return doc_part + global_part

    def train(self, iter=150, tol=1.0, viter=10, vtol=None, checkelbo=math.inf, on_iter=None):
        K = self.engine.model.K
        vtol = 1.0 / K ** 2 if vtol is None else vtol
        return self._loop(iter, tol, checkelbo, lambda: self.iterate(viter, vtol), on_iter)
