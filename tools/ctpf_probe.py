#!/usr/bin/env python
"""CTPF iteration timing probe on SYN-CITEU K=50."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("TMVB_ESTEP_TIMING", "1")      # last_estep_ms needs the library's timing events
import tmvb_amd
tm = tmvb_amd.pkg
K = int(os.environ.get("K", 50))
pc = tm.syn_citeu()
print("corpus", pc.M, pc.V, pc.U, pc.nnz, pc.nR, flush=True)
gm = tm.gpuCTPF(pc, K)
for it in range(int(os.environ.get("ITERS", 8))):
    t0 = time.perf_counter()
    gm.estep(); ms = gm.last_estep_ms()
    gm.reduce_docs(); gm.mstep(); gm.synchronize()
    t1 = time.perf_counter()
    print(f"iter {it}: total {1e3*(t1-t0):.3f} ms estep {ms:.3f} ms sweeps {gm.sweep_hist().tolist()}", flush=True)
t0 = time.perf_counter(); gm.train(iter=50, checkelbo=float('inf'), printelbo=False, recs=False); t1 = time.perf_counter()
print(f"train 50 iters: {(t1-t0):.3f} s -> {50/(t1-t0):.1f} it/s")
# recommendation tail of train! (src/CTPF.jl:379-399): M x U scores + a full ranking per user and per document
t0 = time.perf_counter(); ms_s, ms_r = gm.recommend(scores=False); t1 = time.perf_counter()
MU = pc.M * pc.U
print(f"recommend: scores+keys {ms_s:.3f} ms ({2 * 4 * MU / ms_s / 1e6:.1f} GB/s of key writes), segmented sorts {ms_r:.3f} ms "
      f"({2 * MU / ms_r / 1e3:.1f} M pairs/s), wall incl. {2 * 4 * MU / 1e6:.0f} MB download {1e3 * (t1 - t0):.1f} ms")
