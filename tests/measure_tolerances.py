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

# ---- CTM: teacher-forced single-step deviations per K (the bounds of tests/test_ctm_gpu.py::test_teacher_forced_step) and the
# free-running K = 50 run on the CG kernel (test_free_running_k50_cg_kernel_tracks_the_oracle)
import test_ctm_gpu as TC            # noqa: E402
print("CTM teacher-forced (max over 3 iterations): K, |dlam|, |dlam|/(2e-3+2e-3|lam|), vsq rel, logzeta abs, beta rel (>1e-6), mu abs, sigma abs/max, elbo rel")
for K in (3, 12, 25, 41, 50, 57, 64, 100, 128):
    g = TC.synth_case(tmvb, K)
    gm, om = TC.make_pair(tmvb, oracle, g)
    worst = np.zeros(8)
    for it in range(3):
        TC.force(gm, om); TC.step(gm); TC.step(om)
        e_g = gm.update_elbo(); e_o = om.update_elbo(); gm.update_host()
        dl = np.abs(gm.lam - om.lam)
        big = om.beta > 1e-6
        cur = np.array([dl.max(), (dl / (2e-3 + 2e-3 * np.abs(om.lam))).max(), (np.abs(gm.vsq - om.vsq) / om.vsq).max(),
                        np.abs(gm.logzeta - om.logzeta).max(), (np.abs(gm.beta[big] - om.beta[big]) / om.beta[big]).max(),
                        np.abs(gm.mu - om.mu).max(), np.abs(gm.sigma - om.sigma).max() / np.abs(om.sigma).max(), abs(e_g - e_o) / abs(e_o)])
        worst = np.maximum(worst, cur)
    print("  K=%d " % K + " ".join("%.3g" % x for x in worst))

pc = tmvb.syn_nsf(M=1500, V=25319, seed=2)
K = 50
beta0 = tmvb.dirichlet_rows(K, pc.V, seed=7)
gm = tmvb.gpuCTM(pc, K)
gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F")
om = oracle.CTM(oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V), K, beta0)
t_g = gm.train(iter=20, tol=1.0, checkelbo=1, printelbo=False)
e_prev, t_o = om.update_elbo(), []
for k in range(20):
    om.estep(omp_threads=os.cpu_count() or 1); om.update_beta(); om.update_sigma_mu()
    e_new = om.update_elbo(); t_o.append(e_new)
    stop = (e_new - e_prev) < 1.0; e_prev = e_new
    if stop:
        break
t_o = np.asarray(t_o); n = min(len(t_g), len(t_o))
print("CTM K=50 free-running (CG kernel), 1500 docs: iterations device/oracle", len(t_g), len(t_o), "elbo rel per iteration",
      " ".join("%.2e" % x for x in np.abs(t_g[:n] - t_o[:n]) / np.abs(t_o[:n])))
if len(t_g) == len(t_o):
    print("  final |dmu|", np.abs(gm.mu - om.mu).max(), "|dsigma|/max", np.abs(gm.sigma - om.sigma).max() / np.abs(om.sigma).max(),
          "|dlam| max", np.abs(gm.lam - om.lam).max(), "99.9 pct", np.quantile(np.abs(gm.lam - om.lam), 0.999))
