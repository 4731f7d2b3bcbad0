#!/bin/bash
# models created after another model of the process was closed run slower: stream -> hardware-queue mapping?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3alloc; mkdir -p $O; cd $R
for Q in default 8; do
if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
python - <<PY 2>&1 | tail -6
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
print("GPU_MAX_HW_QUEUES=$Q")
g1 = mk(); print("  LDA K=100, first model of the process     %.1f it/s" % rate(g1), flush=True)
g1.close(); g3 = mk(); print("  second model after closing the first      %.1f it/s" % rate(g3), flush=True)
g3.close()
g4 = mk(50); print("  LDA K=50 after closing all                %.1f it/s" % rate(g4, 50), flush=True)
g4.close(); g5 = mk(50); print("  LDA K=50 again                            %.1f it/s" % rate(g5, 50), flush=True)
PY
done
