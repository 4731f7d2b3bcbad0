"""
What the vocabulary-sliced all-reduce (tmvb_lda_estep_allreduce) costs and hides on ONE GPU: an 8-GPU rank's shard of SYN-NSF
(16 100 documents by default) with an RCCL communicator of nranks = 1 -- real ncclAllReduce launches on the side stream, but no
wire -- timed for TMVB_AR_SLICES as set in the environment (the library reads it once per process: run once per value).
    python tools/ar_slices_probe.py [docs] [iterations]
Prints one JSON line: ms per iteration of the fused form and of the three-call form (estep, reduce_docs, allreduce) on the same handle.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import torch                # noqa: E402
import tmvb_amd             # noqa: E402

tm = tmvb_amd.pkg
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 16100
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
K = 50
corpus = tm.syn_nsf()
shard = corpus.shard(0, docs)
gm = tm.gpuLDA(shard, K)
gm.beta = np.asfortranarray(tm.dirichlet_rows(K, corpus.V, seed=7)); gm.beta_old = gm.beta.copy(order="F")
gm.update_buffer()
out = {}


def clock0(f, iters=300):
    for _ in range(80):
        f()
    gm.synchronize(); t = time.perf_counter()
    for _ in range(iters):
        f()
    gm.synchronize()
    return 1e3 * (time.perf_counter() - t) / iters


def plain():
    gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_alpha()


out["plain_handle_ms"] = clock0(plain)
gm.set_distributed(shard.M, True)       # a one-rank world: M_total = the shard (with the corpus' M the alpha Newton cannot converge on one shard's sums)
out["distributed_flag_ms"] = clock0(plain)
comm = tm.Communicator.rccl(gm.ctx, tm.Communicator.unique_id(), 1, 0)
gm.set_comm(comm, shard.M)
out["with_comm_ms"] = clock0(plain)
ptr, n = gm.stats()


def fused():
    gm.estep_allreduce(); gm.update_beta(); gm.update_alpha()


def three():
    gm.estep(); gm.reduce_docs(); comm.allreduce(ptr, n); gm.update_beta(); gm.update_alpha()


def clock(f):
    for _ in range(80):
        f()
    gm.synchronize(); t = time.perf_counter()
    for _ in range(iters):
        f()
    gm.synchronize()
    return 1e3 * (time.perf_counter() - t) / iters


out.update({"docs": docs, "nnz": int(shard.nnz), "slices": os.environ.get("TMVB_AR_SLICES", "4 (default)"), "iterations": iters})
def no_collective():
    gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_alpha()


def collective_only():
    comm.allreduce(ptr, n)


out["no_collective_ms"] = clock(no_collective); out["collective_only_ms"] = clock(collective_only)
out["three_call_ms"] = clock(three); out["fused_ms"] = clock(fused); out["three_call_ms_again"] = clock(three); out["fused_ms_again"] = clock(fused)
print(json.dumps(out))
