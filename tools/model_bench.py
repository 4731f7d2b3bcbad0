#!/usr/bin/env python
"""Measured lines for the configurations of BASELINE.json other than the headline one (which bench.py owns):
config 3 LDA K=100 on SYN-NSF (one GPU's view), config 4 CTM K=50 on SYN-NSF, config 5 CTPF K=50 on SYN-CITEU.
One JSON line per configuration with the same roofline vocabulary as bench.py (algorithmic bytes of SURVEY.md
section 8d / DESIGN.md section 3; CTM additionally the flop count of its Newton solves against the fp32 peak).
and the filtered models fLDA / fCTM K=50 on SYN-NSF.
Usage: python tools/model_bench.py [lda100] [ctm] [ctpf] [flda] [fctm]   (default: all)"""
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


def cpu_line(kind, make, step, frac, sample_note, budget_s=12.0):
    """cpu_baseline of a configuration: the fp64 C oracle (port of the reference's CPU path) on a bounded document sample
    of the same workload, OpenMP document-parallel E-step on the CPUs this process may use (cgroup quota / affinity),
    scaled to full-corpus iterations/s by the sample's share of the work (`frac`); one single-thread iteration beside it."""
    from bench import usable_cpus
    threads, info = usable_cpus()
    m = make()
    step(m, threads)                                    # warm-up: OpenMP start-up, first touch
    t0 = time.perf_counter(); n = 0
    while n < 4 and time.perf_counter() - t0 < budget_s / 2:
        step(m, threads); n += 1
    omp = n / (time.perf_counter() - t0)
    m1 = make()
    t1 = time.perf_counter(); step(m1, 0); one = 1.0 / (time.perf_counter() - t1)
    return {"value": omp * frac, "unit": "VB iters/sec", "cores": threads, "host_cpus": info, "kind": "port",
            "single_thread_value": one * frac,
            "sample": f"fp64 C oracle ({kind}), {sample_note}; 1 warm-up + {n} timed iterations on {threads} OpenMP threads, "
                      f"1 single-thread iteration; value = sample iters/s x work fraction {frac:.4f}"}


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
    from oracle import oracle as oc
    sh = pc.shard(0, 24000)
    beta0 = tm.dirichlet_rows(K, pc.V, seed=7)
    def ostep(m, nt):
        m.estep(omp_threads=nt); m.update_beta(); m.update_alpha()
    cpu = cpu_line("port of src/LDA.jl train!", lambda: oc.LDA(oc.CSR(sh.doc_ptr, sh.terms, sh.counts, sh.V), K, beta0), ostep,
                   sh.nnz / pc.nnz, f"first {sh.M} documents of SYN-NSF ({sh.nnz} of {pc.nnz} nnz), K=100, cold start")
    return {"cpu_baseline": cpu, "metric": "VB iters/sec, LDA K=100 on NSF-shaped corpus (config 3, one GPU)", "value": 1.0 / sec, "unit": "VB iters/sec",
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
    from oracle import oracle as oc
    sh = pc.shard(0, 1500)
    beta0 = tm.dirichlet_rows(K, pc.V, seed=7)
    def ostep(m, nt):
        m.estep(omp_threads=nt); m.update_beta(); m.update_sigma_mu()
    cpu = cpu_line("port of src/CTM.jl train!", lambda: oc.CTM(oc.CSR(sh.doc_ptr, sh.terms, sh.counts, sh.V), K, beta0), ostep,
                   sh.M / pc.M, f"first {sh.M} documents of SYN-NSF (the Newton solves scale with the document count), K=50, cold start")
    return {"cpu_baseline": cpu, "metric": "VB iters/sec, CTM K=50 on NSF-shaped corpus (config 4)", "value": 1.0 / sec, "unit": "VB iters/sec",
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
    from oracle import oracle as oc
    sh = pc.shard(0, 3000)
    alef0 = np.exp(tm.dirichlet_rows(K, pc.V, seed=7) - 0.5)
    def ostep(m, nt):
        m.estep(omp_threads=nt); m.mstep()
    cpu = cpu_line("port of src/CTPF.jl train!", lambda: oc.CTPF(oc.CSR(sh.doc_ptr, sh.terms, sh.counts, sh.V, sh.rdr_ptr, sh.readers, sh.ratings, sh.U), K, alef0),
                   ostep, (sh.nnz + sh.nR) / (pc.nnz + pc.nR), f"first {sh.M} documents of SYN-CITEU ({sh.nnz}+{sh.nR} of {pc.nnz}+{pc.nR} term+reader entries), K=50, cold start")
    return {"cpu_baseline": cpu, "metric": "VB iters/sec, CTPF K=50 on CiteULike-shaped corpus (config 5)", "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CTPF K=50, SYN-CITEU with readers, train! defaults (viter=10 vtol=1/K^2 checkelbo=Inf), cold start, 5 warm-up + 50 timed iterations",
                       "M": pc.M, "V": pc.V, "U": pc.U, "nnz": pc.nnz, "nR": pc.nR},
            "roofline": {"bound": "hbm", "achieved": B / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": B / sec / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_iteration": B, "traffic": None},
            "recommend": {"ms_scores": ms_s, "ms_rank": ms_r, "pairs": pc.M * pc.U}}


def cpu_line_1t(kind, make, step, frac, sample_note):
    """Single-thread cpu_baseline (the filtered-model oracles have no OpenMP E-step)."""
    m = make()
    t0 = time.perf_counter(); step(m); one = 1.0 / (time.perf_counter() - t0)
    return {"value": one * frac, "unit": "VB iters/sec", "cores": 1, "kind": "port",
            "sample": f"fp64 C oracle ({kind}), {sample_note}; 1 single-thread iteration; value = sample iters/s x work fraction {frac:.4f}"}


def flda():
    K = 50
    pc = tm.syn_nsf()
    gm = tm.gpufLDA(pc, K)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F")
    gm.kappa = tm.dirichlet_rows(1, pc.V, seed=9)[0].copy(); gm.kappa_old = gm.kappa.copy(); gm.update_buffer()
    def it():
        gm.estep(10, 1.0 / K ** 2); gm.reduce_docs(); gm.update_beta(); gm.update_alpha(1000, 1.0 / K ** 2); gm.update_eta()
    sec = timed(it, 20, 3, gm.synchronize)
    es = gm.last_estep_ms()
    # LDA's bytes + tau / tau_old / lse (read + write per token entry) + the kappa statistics
    B = pc.nnz * (8 + 8 * K + 24) + 12 * pc.M * K + 12 * K * pc.V + 8 * pc.V + 4 * (pc.M + 1)
    from oracle import oracle as oc
    sh = pc.shard(0, 4000)
    beta0 = tm.dirichlet_rows(K, pc.V, seed=7); kappa0 = tm.dirichlet_rows(1, pc.V, seed=9)[0]
    def ostep(m):
        m.estep(); m.mstep()
    cpu = cpu_line_1t("port of src/fLDA.jl train!", lambda: oc.fLDA(oc.CSR(sh.doc_ptr, sh.terms, sh.counts, sh.V), K, beta0, kappa0), ostep,
                      sh.nnz / pc.nnz, f"first {sh.M} documents of SYN-NSF ({sh.nnz} of {pc.nnz} nnz), K=50, cold start")
    return {"cpu_baseline": cpu, "metric": "VB iters/sec, fLDA K=50 on NSF-shaped corpus (section 8 row f4)", "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "estep_ms": es, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "fLDA K=50, SYN-NSF, train! defaults, cold start, 3 warm-up + 20 timed iterations", "M": pc.M, "V": pc.V, "nnz": pc.nnz},
            "roofline": {"bound": "hbm", "achieved": B / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": B / sec / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_iteration": B, "traffic": None}}


def fctm():
    K = 50
    pc = tm.syn_nsf()
    gm = tm.gpufCTM(pc, K)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F")
    gm.kappa = tm.dirichlet_rows(1, pc.V, seed=9)[0].copy(); gm.kappa_old = gm.kappa.copy(); gm.update_buffer()
    def it():
        gm.estep(); gm.reduce_docs(); gm.mstep()
    sec = timed(it, 6, 2, gm.synchronize)
    hist, newton = gm.sweep_hist()
    sweeps = int(sum(i * int(h) for i, h in enumerate(hist)))
    B = pc.nnz * (8 + 8 * K + 24) + 16 * pc.M * K + 8 * pc.M + 12 * K * pc.V + 8 * pc.V + 8 * K * K
    F = newton * (K ** 3 / 3.0 + 4 * K * K) + 8.0 * K * pc.nnz * (sweeps / pc.M)
    from oracle import oracle as oc
    sh = pc.shard(0, 400)
    beta0 = tm.dirichlet_rows(K, pc.V, seed=7); kappa0 = tm.dirichlet_rows(1, pc.V, seed=9)[0]
    def ostep(m):
        m.estep(); m.mstep()
    cpu = cpu_line_1t("port of src/fCTM.jl train!", lambda: oc.fCTM(oc.CSR(sh.doc_ptr, sh.terms, sh.counts, sh.V), K, beta0, kappa0), ostep,
                      sh.M / pc.M, f"first {sh.M} documents of SYN-NSF (the Newton solves scale with the document count), K=50, cold start")
    return {"cpu_baseline": cpu, "metric": "VB iters/sec, fCTM K=50 on NSF-shaped corpus (section 8 row f4)", "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "dtype": "f32 (fp64 gradients / logzeta / vsq)", "data": "synthetic",
            "config": {"workload": "fCTM K=50, SYN-NSF, train! defaults, cold start, 2 warm-up + 6 timed iterations",
                       "M": pc.M, "V": pc.V, "nnz": pc.nnz, "lambda_newton_steps_last_iteration": int(newton), "sweeps_last_iteration": sweeps},
            "roofline": {"bound": "valu (register Gauss-Jordan)", "achieved": F / sec / 1e12, "peak": F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": F / sec / 1e12 / F32_PEAK_TFLOPS, "flops_per_iteration": F,
                         "hbm_GBs": B / sec / 1e9, "hbm_frac": B / sec / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": B, "traffic": None}}


if __name__ == "__main__":
    which = sys.argv[1:] or ["lda100", "ctm", "ctpf", "flda", "fctm"]
    for w in which:
        print(json.dumps({"lda100": lda100, "ctm": ctm, "ctpf": ctpf, "flda": flda, "fctm": fctm}[w]()), flush=True)
