"""
Communicators of the document-sharded runs (include/tmvb.h, "communicator"): the ONE all-reduce per outer iteration of the
packed sufficient statistics happens inside libtmvb_hip.so -- RCCL over xGMI, or a host transport -- so that a host in
any language (the Julia shim, this mirror) only has to hand every rank its document shard and call train.

    Communicator.unique_id()                            rank 0; 128 bytes to broadcast by any host channel
    Communicator.rccl(ctx, unique_id, nranks, rank)     one process per GPU (ncclCommInitRank)
    Communicator.rccl_all(ctxs)                         one process, n GPUs (ncclCommInitAll) -> list
    Communicator.host(ctx, nranks, rank, fn)            fn(numpy array) sums in place across ranks (MPI / gloo)
    Communicator.torch_bootstrap(ctx)                   RCCL communicator whose unique id travels through an already
                                                        initialised torch.distributed process group (torchrun)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, VP

UNIQUE_ID_BYTES = 128
F32, F64 = 0, 1
_HOST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class Communicator:
    def __init__(self, handle, ctx, keep=None):
        self.handle, self.ctx, self._keep = handle, ctx, keep

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        check(lib().tmvb_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, ctx, unique_id: bytes, nranks: int, rank: int):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError(f"unique id must be {UNIQUE_ID_BYTES} bytes.")
        h = VP()
        check(lib().tmvb_comm_create_rccl(ctx.handle, C.c_char_p(unique_id), C.c_int32(nranks), C.c_int32(rank), C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def rccl_all(cls, ctxs):
        n = len(ctxs)
        arr = (VP * n)(*[c.handle for c in ctxs])
        out = (VP * n)()
        check(lib().tmvb_comm_create_rccl_all(arr, C.c_int32(n), out))
        return [cls(VP(out[i]), ctxs[i]) for i in range(n)]

    @classmethod
    def host(cls, ctx, nranks: int, rank: int, fn):
        """fn(a: np.ndarray) must leave the element-wise sum over all ranks in `a` (float32 or float64)."""
        def tramp(_user, buf, count, dtype):
            try:
                ct = C.c_float if dtype == F32 else C.c_double
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(ct)), shape=(count,))
                fn(a)
                return 0
            except Exception:                     # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = _HOST_FN(tramp)
        h = VP()
        check(lib().tmvb_comm_create_host(ctx.handle, C.c_int32(nranks), C.c_int32(rank), cb, None, C.byref(h)))
        return cls(h, ctx, keep=cb)

    _boot_count = 0

    @classmethod
    def torch_bootstrap(cls, ctx):
        """One process per GPU under torchrun: rank 0 draws the RCCL unique id, the key-value store of the already
        initialised torch.distributed default group (backend-independent) carries the 128 bytes to the other ranks,
        every rank joins with ncclCommInitRank inside the library."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        store = dist.distributed_c10d._get_default_store()
        key = f"tmvb_rccl_unique_id_{cls._boot_count}"
        cls._boot_count += 1
        if rank == 0:
            store.set(key, cls.unique_id())
        uid = bytes(store.get(key))          # blocks until rank 0 has published it
        return cls.rccl(ctx, uid, world, rank)

    def info(self):
        n, r, b = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(lib().tmvb_comm_info(self.handle, C.byref(n), C.byref(r), C.byref(b)))
        return {"nranks": n.value, "rank": r.value, "backend": "rccl" if b.value == 0 else "host"}

    def allreduce(self, dev_ptr: int, count: int, dtype: int = F32):
        check(lib().tmvb_comm_allreduce(self.handle, VP(dev_ptr), C.c_int64(count), C.c_int32(dtype)))

    def close(self):
        if self.handle:
            lib().tmvb_comm_destroy(self.handle)
            self.handle = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_version() -> int:
    L = lib()
    L.tmvb_rccl_version.restype = C.c_int
    return int(L.tmvb_rccl_version())
