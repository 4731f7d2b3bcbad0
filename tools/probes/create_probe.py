import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import tmvb_amd
tm = tmvb_amd.pkg
import ctypes as C
from tmvb_amd_pkg.lda import DeviceContext, DeviceCorpus, LDA
pc = tm.syn_nsf()
g0 = tm.gpuLDA(pc, 50); g0.close()
for rep in range(2):
    t0 = time.perf_counter(); host = LDA(pc, 50, 7); t1 = time.perf_counter()
    ctx = DeviceContext(0); t2 = time.perf_counter()
    dc = DeviceCorpus(ctx, pc); t3 = time.perf_counter()
    h = C.c_void_p()
    rc = tm.lib().tmvb_lda_create(ctx.handle, dc.handle, C.c_int32(50), C.byref(h)); t4 = time.perf_counter()
    print(f"host LDA() {1e3*(t1-t0):.1f} ms, ctx {1e3*(t2-t1):.1f}, DeviceCorpus {1e3*(t3-t2):.1f}, tmvb_lda_create {1e3*(t4-t3):.1f} rc={rc}")
    tm.lib().tmvb_lda_destroy(h)
t0=time.perf_counter(); g = tm.gpuLDA(pc, 50); t1=time.perf_counter(); print(f"gpuLDA() {1e3*(t1-t0):.1f}")
t0=time.perf_counter(); tm.check_model(g, rtol=3.5e-4); t1=time.perf_counter(); print(f"check_model {1e3*(t1-t0):.1f}")
t0=time.perf_counter(); tp=[np.argsort(g.beta[i, :], kind="stable")[::-1] + 1 for i in range(g.K)]; t1=time.perf_counter(); print(f"topics argsort {1e3*(t1-t0):.1f}")
