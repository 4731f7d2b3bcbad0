import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import tmvb_amd
tm = tmvb_amd.pkg
from test_ctm_gpu import synth_case
g = synth_case(tm, int(os.environ.get("KK", "3")), M=120, V=400) if False else None
K = int(os.environ.get("KK", "3"))
pc = tm.syn_nsf(M=120, V=400, seed=3)
gm = tm.gpuCTM(pc, K)
gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=5)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
for vi in (1, 2, 10):
    gm.lam[:] = 0.1 * np.arange(K)[:, None]; gm.vsq[:] = 1.0; gm.logzeta[:] = 0.5; gm.update_buffer()
    gm.estep(viter=vi); gm.update_host()
    print("viter", vi, "lam", np.round(gm.lam[:, :3].T.ravel(), 6), "vsq", np.round(gm.vsq[:, 0], 6), "lz", np.round(gm.logzeta[:3], 6), "sweeps", gm.doc_sweeps()[:6])
