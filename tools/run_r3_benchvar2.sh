#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3benchvar; mkdir -p $O; cd $R
python - <<PY 2>&1 | tail -12
import sys, time
sys.path.insert(0, "tools")
import model_bench, numpy as np
tm = model_bench.tm
pc = tm.syn_nsf()
def mk(K=100):
    g = tm.gpuLDA(pc, K)
    g.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); g.beta_old = g.beta.copy(order="F"); g.update_buffer()
    return g
def rate(g, K=100, burn=40, n=20):
    def it(): g.estep(10, 1.0 / K ** 2); g.reduce_docs(); g.update_beta(); g.update_alpha(1000, 1.0 / K ** 2)
    for _ in range(burn): it()
    g.synchronize(); t0 = time.perf_counter()
    for _ in range(n): it()
    g.synchronize(); return n / (time.perf_counter() - t0)
g1 = mk(); print("engine 1", round(rate(g1), 1), flush=True)
g2 = mk(); print("engine 2, engine 1 alive", round(rate(g2), 1), flush=True)
print("engine 1 again", round(rate(g1, burn=5), 1), flush=True)
g1.close(); g3 = mk(); print("engine 3 after closing 1", round(rate(g3), 1), flush=True)
g2.close(); g3.close()
g4 = mk(); print("engine 4 after closing all", round(rate(g4), 1), flush=True)
PY
