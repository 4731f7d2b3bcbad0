"""
Prints the deviations the free-running parity tests bound (run on the GPU box: python tests/measure_tolerances.py).
The tolerances written in tests/test_ctpf_gpu.py are these figures with a 3-5x margin.  Not collected by pytest.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tmvb_amd                       # noqa: E402
from oracle import oracle             # noqa: E402

tmvb = tmvb_amd.pkg
GOLD = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-300)).max()) if a.size else 0.0


for name in ("ctpf_m40_v60_u15_k4", "ctpf_m30_v40_u12_k6_r1"):
    z = np.load(os.path.join(GOLD, name + ".npz")); g = {k: z[k] for k in z.files}
    K, V, U = int(g["K"]), int(g["V"]), int(g["U"])
    pc = tmvb.PackedCorpus(g["doc_ptr"], g["terms"], g["counts"], V, g["rdr_ptr"], g["readers"], g["ratings"], U)
    m = tmvb.CTPF(pc, K)
    m.alef = np.asfortranarray(g["alef0"]); m.alef_old = m.alef.copy(order="F")
    traj = tmvb.gpu_train_ctpf(m, iter=int(g["iters"]), tol=0.0, checkelbo=1, printelbo=False)
    print(name, "elbo rel", float(np.max(np.abs(traj - g["elbo_traj"]) / np.abs(g["elbo_traj"]))),
          {n: rel(getattr(m, n), g[n]) for n in ("alef", "he", "bet", "vav", "dalet", "het", "gimel", "zayin")})

sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_ctpf_gpu as T            # noqa: E402
g = T.synth_case(tmvb, 100, M=120, V=400, U=70, seed=12)
gm, om = T.make_pair(tmvb, oracle, g)
traj = gm.train(iter=4, tol=0.0, checkelbo=1, printelbo=False, recs=True)
for it in range(4):
    om.estep(); om.mstep()
e_o = om.update_elbo()
sc = (om.gimel / om.dalet[:, None] + om.zayin / om.het[:, None]).T @ (om.he / om.vav[:, None])
print("k100 elbo rel", abs(traj[-1] - e_o) / abs(e_o), {n: rel(getattr(gm, n), getattr(om, n)) for n in ("bet", "vav", "dalet", "het")},
      "scores", float(np.abs(gm.scores - sc).max() / np.abs(sc).max()))
