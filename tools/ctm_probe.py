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
    st = gm.solver_stats()
    if st["waves"]:
        w = st["waves"]; tot = max(st["cyc_kernel"], 1)
        print("   per wave: newton trips %.1f cg trips %.1f (%.1f per newton trip); Mcycles token %.2f logzeta %.2f vsq %.2f grad-mv %.2f grad-asm %.2f cg %.2f update %.2f barrier-wait %.2f kernel %.2f" % (
              st["newton_trips"] / w, st["cg_trips"] / w, st["cg_trips"] / max(st["newton_trips"], 1), st["cyc_token"] / w / 1e6, st["cyc_logzeta"] / w / 1e6,
              st["cyc_vsq"] / w / 1e6, st["cyc_gradmv"] / w / 1e6, st["cyc_gradient"] / w / 1e6, st["cyc_cg"] / w / 1e6, st["cyc_update"] / w / 1e6, st["cyc_spare"] / w / 1e6, tot / w / 1e6), flush=True)
    print(f"iter {it}: total {1e3*(t1-t0):.2f} ms estep {ms:.2f} ms sweeps {hist.tolist()} newton {ns} ({ns/M:.1f}/doc)", flush=True)
print("elbo", gm.update_elbo())
