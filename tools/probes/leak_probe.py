#!/usr/bin/env python
"""Create / train / destroy each model many times and watch device memory (hipMemGetInfo through torch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tmvb_amd
tm = tmvb_amd.pkg
pc = tm.syn_citeu(M=400, V=600, U=80, seed=1)
def used(): torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
for name, cls, kw in (("lda", tm.gpuLDA, {}), ("ctm", tm.gpuCTM, {}), ("ctpf", tm.gpuCTPF, {})):
    base = None
    for rep in range(60):
        m = cls(pc, 20)
        m.train(iter=2, checkelbo=1, printelbo=False)
        m.close(); m.dcorp.close() if hasattr(m.dcorp, "close") else None; m.ctx.close() if hasattr(m.ctx, "close") else None
        del m
        if rep == 9: base = used()
    print(f"{name}: device memory in use after 10 reps {base:.1f} MiB, after 60 reps {used():.1f} MiB")
