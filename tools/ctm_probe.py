#!/usr/bin/env python
"""CTM iteration timing probe on SYN-NSF K=50."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tmvb_amd
tm = tmvb_amd.pkg
K = int(os.environ.get("K", 50)); M = int(os.environ.get("M", 128804))
pc = tm.syn_nsf(M=M)
gm = tm.gpuCTM(pc, K)
for it in range(int(os.environ.get("ITERS", 6))):
    t0 = time.perf_counter()
    gm.estep(); ms = gm.last_estep_ms()
    gm.reduce_docs(); gm.update_beta(); gm.update_sigma(); gm.update_mu(); gm.synchronize()
    t1 = time.perf_counter()
    hist, ns = gm.sweep_hist()
    print(f"iter {it}: total {1e3*(t1-t0):.2f} ms estep {ms:.2f} ms sweeps {hist.tolist()} newton {ns} ({ns/M:.1f}/doc)", flush=True)
print("elbo", gm.update_elbo())
