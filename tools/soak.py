#!/usr/bin/env python
"""Soak run: train!(iter=150, checkelbo=1, tol=0) for every model on its full synthetic corpus; prints wall time, the ELBO at a
few iterations, whether the trajectory is finite and non-decreasing (coordinate ascent: up to fp32 noise for the models
whose M-step is exact), and check_model on the result.  Usage: python tools/soak.py [lda ctm ctpf flda fctm]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tmvb_amd

tm = tmvb_amd.pkg
ITER = int(os.environ.get("ITER", 150))


def report(name, g, traj, secs, check):
    traj = np.asarray(traj)
    d = np.diff(traj)
    print(f"{name}: {len(traj)} iterations in {secs:.2f} s ({1e3 * secs / max(len(traj), 1):.2f} ms per checked iteration); "
          f"elbo[0, 10, 50, -1] = {traj[0]:.1f} {traj[min(10, len(traj) - 1)]:.1f} {traj[min(50, len(traj) - 1)]:.1f} {traj[-1]:.1f}; "
          f"finite {bool(np.all(np.isfinite(traj)))}; decreasing steps {int((d < 0).sum())} (largest drop {float(-d.min()) if len(d) and d.min() < 0 else 0.0:.3g}, "
          f"relative {float(-d.min() / abs(traj[-1])) if len(d) and d.min() < 0 else 0.0:.2e})", flush=True)
    check(g)


def main():
    which = sys.argv[1:] or ["lda", "ctm", "ctpf", "flda", "fctm"]
    K = 50
    for w in which:
        if w == "ctpf":
            pc = tm.syn_citeu()
            g = tm.gpuCTPF(pc, K)
            t0 = time.perf_counter(); traj = g.train(iter=ITER, tol=0.0, checkelbo=1, printelbo=False, recs=False); s = time.perf_counter() - t0
            report("CTPF K=50 SYN-CITEU", g, traj, s, lambda m: tm.check_model_ctpf(m, rtol=1e-3) if "rtol" in tm.check_model_ctpf.__code__.co_varnames else tm.check_model_ctpf(m))
            continue
        pc = tm.syn_nsf()
        cls, chk = {"lda": (tm.gpuLDA, tm.check_model), "ctm": (tm.gpuCTM, tm.check_model_ctm), "flda": (tm.gpufLDA, tm.check_model_flda),
                    "fctm": (tm.gpufCTM, tm.check_model_fctm)}[w]
        g = cls(pc, K)
        t0 = time.perf_counter(); traj = g.train(iter=ITER, tol=0.0, checkelbo=1, printelbo=False); s = time.perf_counter() - t0
        report(f"{w} K=50 SYN-NSF", g, traj, s, lambda m: chk(m, rtol=3.5e-4))


if __name__ == "__main__":
    main()
