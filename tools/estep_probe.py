#!/usr/bin/env python
"""E-step timing probe on SYN-NSF: times tmvb_lda_estep alone under different settings."""
import os, sys, time
os.environ.setdefault("TMVB_ESTEP_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tmvb_amd
tm = tmvb_amd.pkg

K = int(os.environ.get("K", 50))
pc = tm.syn_nsf()
gm = tm.gpuLDA(pc, K)
def run(viter, n=5, warm=3):
    for _ in range(warm):
        gm.estep(viter); gm.reduce_docs(); gm.update_beta(); gm.update_alpha()
    ts = []
    for _ in range(n):
        gm.estep(viter); ts.append(gm.last_estep_ms()); gm.reduce_docs(); gm.update_beta(); gm.update_alpha()
    return np.mean(ts), gm.sweep_hist().tolist()
for viter in [int(x) for x in os.environ.get("VITERS", "10").split(",")]:
    ms, hist = run(viter)
    print(f"flags={os.environ.get('TMVB_DEBUG_FLAGS','0')} viter={viter} estep_ms={ms:.3f} sweeps={hist}", flush=True)
