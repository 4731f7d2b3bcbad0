#!/usr/bin/env python
"""Host enqueue time per iteration against the GPU's iteration period (is a configuration launch-bound? would a hipGraph help?).
Enqueue time = wall time of N iterations' API calls WITHOUT waiting for the device (the queues are deep enough to hold them),
period = wall time of the same N iterations including the final synchronisation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tmvb_amd
tm = tmvb_amd.pkg


def probe(name, g, it, n=200):
    for _ in range(60): it()
    g.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): it()
    t1 = time.perf_counter()
    g.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e6 * (t1 - t0) / n:.1f} us / iteration, period {1e6 * (t2 - t0) / n:.1f} us", flush=True)


pc = tm.syn_citeu(); g = tm.gpuCTPF(pc, 50)
probe("CTPF K=50 SYN-CITEU (python, 3 API calls per iteration)", g, lambda: (g.estep(), g.reduce_docs(), g.mstep()))
g.close()
pc = tm.syn_nsf(M=16100); K = 50; g = tm.gpuLDA(pc, K)
g.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); g.beta_old = g.beta.copy(order="F"); g.update_buffer()
probe("LDA K=50, 16 100-document shard (python, 4 API calls per iteration)", g,
      lambda: (g.estep(10, 1.0 / K ** 2), g.reduce_docs(), g.update_beta(), g.update_alpha(1000, 1.0 / K ** 2)))
g.close()
