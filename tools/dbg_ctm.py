import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ctypes as C
import tmvb_amd
tm = tmvb_amd.pkg
K = int(os.environ.get("K", 5))
pc = tm.syn_nsf(M=200, V=300, seed=2)
def mk(batch):
    os.environ["TMVB_CTM_BATCH"] = batch
    g = tm.gpuCTM(pc, K)
    g.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=5)); g.beta_old = g.beta.copy(order="F"); g.update_buffer()
    return g
gw, gb = mk("0"), mk("1")
def state(g):
    lam = np.empty((g.K, g.M), order="F"); vsq = np.empty((g.K, g.M), order="F"); lz = np.empty(g.M)
    pd = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    tm._lib.check(tm.lib().tmvb_ctm_get_state(g.handle, None, None, None, None, None, pd(lam), None, pd(vsq), pd(lz), None))
    return lam, vsq, lz
for it in range(3):
    for (ni, vi) in ((1, 1), (1000, 1), (1000, 10)):
        gw.update_buffer(); gb.update_buffer()
        gw.estep(niter=ni, viter=vi); gb.estep(niter=ni, viter=vi)
        (lw, vw, zw), (lb, vb, zb) = state(gw), state(gb)
        print(f"it {it} niter={ni} viter={vi}: dlam {np.nanmax(np.abs(lw-lb)):.3e} nan {np.isnan(lb).sum()} dvsq {np.nanmax(np.abs(vw-vb)):.3e} dlz {np.nanmax(np.abs(zw-zb)):.3e} newton {gw.sweep_hist()[1]} {gb.sweep_hist()[1]}")
    gw.update_buffer(); gw.estep(); gw.reduce_docs(); gw.update_beta(); gw.update_sigma(); gw.update_mu(); gw.update_host()
    for n in ("mu", "sigma", "invsigma", "beta", "beta_old", "lam", "lam_old", "vsq", "logzeta"):
        v = getattr(gw, n); setattr(gb, n, v.copy(order="F") if v.ndim > 1 else v.copy())
