import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tmvb_amd
tm = tmvb_amd.pkg
pc = tm.syn_nsf(M=3000, V=25319, seed=2)
os.environ["TMVB_CTM_BATCH"] = "0"
gw = tm.gpuCTM(pc, 50)
os.environ["TMVB_CTM_BATCH"] = "1"
gb = tm.gpuCTM(pc, 50)
gw.estep(); gw.reduce_docs(); gw.update_beta(); gw.update_sigma(); gw.update_mu(); gw.update_host()
for n in ("mu", "sigma", "invsigma", "beta", "beta_old", "lam", "lam_old", "vsq", "logzeta"):
    setattr(gb, n, getattr(gw, n).copy(order="F") if getattr(gw, n).ndim > 1 else getattr(gw, n).copy())
gb.update_buffer()
print("cond invsigma", np.linalg.cond(gw.invsigma), "eig", np.linalg.eigvalsh(gw.invsigma)[[0, -1]])
for (ni, vi) in ((1, 1), (2, 1), (1000, 1), (1000, 2), (1000, 10)):
    gw.update_buffer(); gb.update_buffer()
    gw.estep(niter=ni, viter=vi); gb.estep(niter=ni, viter=vi)
    a = tm.gpuCTM.__dict__  # noqa
    lw = gw.lam.copy(); vw = gw.vsq.copy()
    gw2 = {}; gb2 = {}
    for g, out in ((gw, gw2), (gb, gb2)):
        K, M = g.K, g.M
        lam = np.empty((K, M), order="F"); vsq = np.empty((K, M), order="F"); lz = np.empty(M)
        import ctypes as C
        from tmvb_amd import pkg
        L = pkg.lib()
        pd = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
        pkg._lib.check(L.tmvb_ctm_get_state(g.handle, None, None, None, None, None, pd(lam), None, pd(vsq), pd(lz), None))
        out.update(lam=lam, vsq=vsq, lz=lz, hist=g.sweep_hist())
    dl = np.abs(gw2["lam"] - gb2["lam"]); dv = np.abs(gw2["vsq"] - gb2["vsq"])
    print(f"niter={ni} viter={vi}: max|dlam| {np.nanmax(dl):.3e} nan {np.isnan(gb2['lam']).sum()} max|dvsq| {np.nanmax(dv):.3e} dlz {np.nanmax(np.abs(gw2['lz']-gb2['lz'])):.3e} newton wave {gw2['hist'][1]} batch {gb2['hist'][1]}")
    if np.nanmax(dl) > 1e-2 or np.isnan(gb2["lam"]).any():
        d = int(np.nanargmax(np.where(np.isnan(dl), np.inf, dl).max(axis=0)))
        print(" worst doc", d, "N", np.diff(pc.doc_ptr)[d], "lam wave", gw2["lam"][:6, d], "batch", gb2["lam"][:6, d])
