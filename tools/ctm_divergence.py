"""
How much of the lane-per-document CTM kernel's work is lane divergence (round-3 review, 4a): a wave of 64 documents runs every loop
until its slowest lane is done.  Config 4 (CTM K = 50, SYN-NSF) is trained to the steady state; then, for the last E-step,
  * Newton: trips the waves ran (tmvb_ctm_solver_stats) x 64 lanes against the Newton steps the documents took (tmvb_ctm_sweep_hist);
  * tokens: sum over waves of the longest document x 64 against the sum of the document lengths, for waves of 64 in length order
    (the launch order up to the regrouping inside chunks of 2 048 documents of alike length);
  * sweeps: sum over waves of the largest sweep count x 64 against the sweeps the documents ran, in the same order.
Run once with the regrouping by last E-step's Newton count (default) and once with TMVB_CTM_REORDER=0.
    python tools/ctm_divergence.py [iterations]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import tmvb_amd             # noqa: E402

tm = tmvb_amd.pkg
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pc = tm.syn_nsf()
gm = tm.gpuCTM(pc, 50)
for _ in range(iters):
    gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_sigma(); gm.update_mu()
gm.synchronize()
hist, newton_steps = gm.sweep_hist()
st = gm.solver_stats()
sweeps = gm.doc_sweeps().astype(np.int64)
ln = np.diff(np.asarray(pc.doc_ptr)).astype(np.int64)
order = np.argsort(-ln, kind="stable")
M = len(ln)
nw = (M + 63) // 64
pad = nw * 64 - M
lw = np.concatenate([ln[order], np.zeros(pad, np.int64)]).reshape(nw, 64)
sw = np.concatenate([sweeps[order], np.zeros(pad, np.int64)]).reshape(nw, 64)
out = {
    "reorder": os.environ.get("TMVB_CTM_REORDER", "1 (default)"), "iterations": iters, "waves": int(st["waves"]),
    "newton_steps_of_the_documents": int(newton_steps), "newton_trips_of_the_waves": int(st["newton_trips"]),
    "newton_lane_efficiency": newton_steps / max(64.0 * st["newton_trips"], 1.0),
    "newton_steps_per_document": newton_steps / M, "newton_trips_per_wave": st["newton_trips"] / max(st["waves"], 1),
    "token_lane_efficiency_length_order": float(ln.sum() / (64.0 * lw.max(axis=1).sum())),
    "token_sweep_lane_efficiency_length_order": float((ln * sweeps).sum() / (64.0 * (lw.max(axis=1) * sw.max(axis=1)).sum())),
    "sweep_lane_efficiency_length_order": float(sweeps.sum() / (64.0 * sw.max(axis=1).sum())),
    "longest_wave_terms": int(lw.max()), "mean_terms": float(ln.mean()), "sweep_hist": hist.tolist(),
}
print(json.dumps(out))
