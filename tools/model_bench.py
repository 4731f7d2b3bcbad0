#!/usr/bin/env python
"""Measured lines for the configurations of BASELINE.json other than the headline one (which bench.py owns):
config 3 LDA K=100 on SYN-NSF (one GPU's view), config 4 CTM K=50 on SYN-NSF, config 5 CTPF K=50 on SYN-CITEU.
One JSON line per configuration with the same roofline vocabulary as bench.py (algorithmic bytes of SURVEY.md
section 8d / DESIGN.md section 3; CTM additionally the flop count of its Newton solves against the fp32 peak).
Usage: python tools/model_bench.py [lda100] [ctm] [ctpf]   (default: all)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tmvb_amd

tm = tmvb_amd.pkg
HBM_PEAK_GBS = 8000.0
F32_PEAK_TFLOPS = 157.3


def timed(fn, steps, warmup, sync):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def lda100():
    K = 100
    pc = tm.syn_nsf()
    gm = tm.gpuLDA(pc, K)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
    def it():
        gm.estep(10, 1.0 / K ** 2); gm.reduce_docs(); gm.update_beta(); gm.update_alpha(1000, 1.0 / K ** 2)
    sec = timed(it, 20, 3, gm.synchronize)
    B = pc.nnz * (8 + 8 * K) + 12 * pc.M * K + 12 * K * pc.V + 4 * (pc.M + 1)
    return {"metric": "VB iters/sec, LDA K=100 on NSF-shaped corpus (config 3, one GPU)", "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LDA K=100, SYN-NSF, train! defaults, cold start, 3 warm-up + 20 timed iterations", "M": pc.M, "V": pc.V, "nnz": pc.nnz},
            "roofline": {"bound": "hbm", "achieved": B / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": B / sec / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_iteration": B, "traffic": None}}


def ctm():
    K = 50
    pc = tm.syn_nsf()
    gm = tm.gpuCTM(pc, K)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
    def it():
        gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_sigma(); gm.update_mu()
    sec = timed(it, 6, 2, gm.synchronize)
    hist, newton = gm.sweep_hist()
    sweeps = int(sum(i * int(h) for i, h in enumerate(hist)))
    B = pc.nnz * (8 + 8 * K) + 16 * pc.M * K + 8 * pc.M + 12 * K * pc.V + 8 * K * K
    F = newton * (K ** 3 / 3.0 + 4 * K * K) + 6.0 * K * pc.nnz * (sweeps / pc.M)
    return {"metric": "VB iters/sec, CTM K=50 on NSF-shaped corpus (config 4)", "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "dtype": "f32 (fp64 gradients / logzeta / vsq)", "data": "synthetic",
            "config": {"workload": "CTM K=50, SYN-NSF, train! defaults (niter=1000 ntol=1/K^2 viter=10 vtol=1/K^2), cold start, 2 warm-up + 6 timed iterations",
                       "M": pc.M, "V": pc.V, "nnz": pc.nnz, "lambda_newton_steps_last_iteration": int(newton), "sweeps_last_iteration": sweeps},
            "roofline": {"bound": "valu (register Gauss-Jordan; f32 MFMA has the same peak)", "achieved": F / sec / 1e12, "peak": F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": F / sec / 1e12 / F32_PEAK_TFLOPS, "flops_per_iteration": F,
                         "hbm_GBs": B / sec / 1e9, "hbm_frac": B / sec / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": B, "traffic": None}}


def ctpf():
    K = 50
    pc = tm.syn_citeu()
    gm = tm.gpuCTPF(pc, K)
    def it():
        gm.estep(); gm.reduce_docs(); gm.mstep()
    sec = timed(it, 50, 5, gm.synchronize)
    B = pc.nnz * (8 + 8 * K) + pc.nR * (8 + 8 * K) + 16 * pc.M * K + 12 * K * (pc.V + pc.U)
    ms_s, ms_r = gm.recommend(scores=False)
    return {"metric": "VB iters/sec, CTPF K=50 on CiteULike-shaped corpus (config 5)", "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CTPF K=50, SYN-CITEU with readers, train! defaults (viter=10 vtol=1/K^2 checkelbo=Inf), cold start, 5 warm-up + 50 timed iterations",
                       "M": pc.M, "V": pc.V, "U": pc.U, "nnz": pc.nnz, "nR": pc.nR},
            "roofline": {"bound": "hbm", "achieved": B / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": B / sec / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_iteration": B, "traffic": None},
            "recommend": {"ms_scores": ms_s, "ms_rank": ms_r, "pairs": pc.M * pc.U}}


if __name__ == "__main__":
    which = sys.argv[1:] or ["lda100", "ctm", "ctpf"]
    for w in which:
        print(json.dumps({"lda100": lda100, "ctm": ctm, "ctpf": ctpf}[w]()), flush=True)
