#!/usr/bin/env python
"""A fixed number of outer iterations of one BASELINE.json configuration, for the rocprofv3 --pmc passes (counters serialise the
kernels, so the runs are short and the iteration count must be known exactly: tools/pmc_summary.py divides by it).
  pmc_window.py <lda50|lda100|ctm|ctpf> [burnin] [iters]      prints "iterations=<burnin + iters>" """
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import tmvb_amd

tm = tmvb_amd.pkg
which = sys.argv[1]
burnin = int(sys.argv[2]) if len(sys.argv) > 2 else 20
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
if which in ("lda50", "lda100"):
    K = 50 if which == "lda50" else 100
    pc = tm.syn_nsf()
    gm = tm.gpuLDA(pc, K)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
    step = lambda: (gm.estep(), gm.reduce_docs(), gm.update_beta(), gm.update_alpha())
elif which == "ctm":
    pc = tm.syn_nsf()
    gm = tm.gpuCTM(pc, 50)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(50, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
    step = lambda: (gm.estep(), gm.reduce_docs(), gm.update_beta(), gm.update_sigma(), gm.update_mu())
elif which == "ctpf":
    pc = tm.syn_citeu()
    gm = tm.gpuCTPF(pc, 50)
    step = lambda: (gm.estep(), gm.reduce_docs(), gm.mstep())
else:
    raise SystemExit(__doc__)
for _ in range(burnin + iters):
    step()
gm.synchronize()
print(f"iterations={burnin + iters}", flush=True)
