#!/usr/bin/env python
"""LDA K=50 on SYN-NSF, checked iterations (checkelbo=1) through the library's train!: what a checked iteration costs (timeline under rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tmvb_amd
tm = tmvb_amd.pkg
K = 50
pc = tm.syn_nsf()
gm = tm.gpuLDA(pc, K)
gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F")
n = int(os.environ.get("ITERS", 200))
t0 = time.perf_counter(); traj = gm.train(iter=n, tol=0.0, checkelbo=1, printelbo=False); gm.synchronize(); t1 = time.perf_counter()
print(f"checked: {1e3 * (t1 - t0) / n:.4f} ms per iteration over {n}")
t0 = time.perf_counter(); gm.train(iter=n, tol=0.0, checkelbo=np.inf, printelbo=False); gm.synchronize(); t1 = time.perf_counter()
print(f"unchecked: {1e3 * (t1 - t0) / n:.4f} ms per iteration over {n}")
