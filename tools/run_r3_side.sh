#!/bin/bash
# which stream carries LDA's side chain (Elogtheta sums + alpha Newton) x hardware-queue count: first and later models of a process
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3side; mkdir -p $O; cd $R
for Q in default 8; do
for S in 2 0 1 3; do
if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
TMVB_LDA_SIDE_STREAM=$S python - <<PY 2>&1 | tail -1
import sys, time
sys.path.insert(0, "tools")
import model_bench, numpy as np
tm = model_bench.tm
pc = tm.syn_nsf()
def mk(K):
    g = tm.gpuLDA(pc, K)
    g.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); g.beta_old = g.beta.copy(order="F"); g.update_buffer()
    return g
def rate(g, K, burn=60, n=30):
    def it(): g.estep(10, 1.0 / K ** 2); g.reduce_docs(); g.update_beta(); g.update_alpha(1000, 1.0 / K ** 2)
    for _ in range(burn): it()
    g.synchronize(); t0 = time.perf_counter()
    for _ in range(n): it()
    g.synchronize(); return n / (time.perf_counter() - t0)
r = []
for K in (50, 100, 50, 100):
    g = mk(K); r.append(rate(g, K)); g.close()
print("queues=$Q side=aux[$S]  K=50 first %.0f  K=100 second %.0f  K=50 third %.0f  K=100 fourth %.0f" % tuple(r))
PY
done; done
