#!/usr/bin/env python
"""Time update_elbo! on SYN-NSF K=50 after a few iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tmvb_amd
tm = tmvb_amd.pkg
K = int(os.environ.get("K", 50))
pc = tm.syn_nsf()
gm = tm.gpuLDA(pc, K)
gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
for _ in range(5):
    gm.estep(10, 1.0 / K ** 2); gm.reduce_docs(); gm.update_beta(); gm.update_alpha(1000, 1.0 / K ** 2)
e = gm.update_elbo(); gm.synchronize()
t0 = time.perf_counter()
for _ in range(10): e = gm.update_elbo()
print(f"update_elbo: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms  elbo={e:.6f}")
