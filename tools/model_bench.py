#!/usr/bin/env python
"""Measured lines for the configurations of BASELINE.json other than the headline one (which bench.py owns):
config 3 LDA K=100 on SYN-NSF (one GPU's view), config 4 CTM K=50 on SYN-NSF, config 5 CTPF K=50 on SYN-CITEU, and the
filtered models fLDA / fCTM K=50 on SYN-NSF.  One JSON object per configuration with the
same vocabulary as bench.py: STEADY-STATE window (untimed burn-in iterations from the cold start first, then warm-up +
timed iterations), roofline on the algorithmic bytes of SURVEY.md section 8d / DESIGN.md section 3 (CTM additionally
the flop counts of its Newton solves against the fp32 peak: nominal = the formula of section 8d, executed = what the
CG kernel really issues), and a cpu_baseline (the fp64 C oracle with its OpenMP document-parallel E-step).

bench.py imports lda100 / ctm / ctpf from here and carries their lines under "other_configs" of its one JSON line, so
that the driver's BENCH record holds them.
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


def usable_cpus():
    from bench import usable_cpus as u
    return u()


def cpu_line(kind, make, step, frac, sample_note, warm=1, timed=2, budget_s=12.0, full=False, check=None):
    """cpu_baseline of a configuration: the fp64 C oracle (port of the reference's CPU path; the reference is Julia and
    cannot run here) with the OpenMP document-parallel E-step on the CPUs this process may use (cgroup quota / affinity).
    full=True: the whole corpus (frac = 1); otherwise a bounded document sample, scaled to full-corpus iterations/s by
    the sample's share of the work (`frac`) and labelled "sample".
    check(om, iters, threads) -> (parity block, oracle seconds per iteration): the warm-up + timed oracle iterations are run
    THROUGH oracle/parity.py, i.e. they are at the same time the checker of the HIP engine on the same full corpus (teacher
    forced, every document and every global compared); the block is returned under "parity"."""
    threads, info = usable_cpus()
    m = make()
    block = None
    if check is not None:
        block, secs = check(m, warm + timed, threads)
        n = timed
        omp = n / sum(secs[warm:])
    else:
        for _ in range(warm):
            step(m, threads)                                # OpenMP start-up, first touch
        t0 = time.perf_counter(); n = 0
        while n < timed and (n == 0 or time.perf_counter() - t0 < budget_s):
            step(m, threads); n += 1
        omp = n / (time.perf_counter() - t0)
    name = "value" if full else "value (sample, scaled)"
    out = {"value": omp * frac, "value_is": "full corpus" if full else "sample scaled by work fraction", "unit": "VB iters/sec",
           "cores": threads, "host_cpus": info, "kind": "port",
           "sample_short": f"fp64 C oracle, {'FULL corpus' if full else f'sample x{frac:.3f}'}, {warm} warm-up + {n} timed iterations, OpenMP x{threads}",
           "sample": f"fp64 C oracle ({kind}), {sample_note}; {warm} warm-up + {n} timed iterations from the cold start on {threads} OpenMP "
                     f"threads" + (" (the oracle's own calls of the parity check's teacher-forced iterations)" if check else "")
                     + ("" if full else f"; {name} = sample iters/s x work fraction {frac:.4f}")}
    if block is not None:
        block["against"] = f"fp64 C oracle ({kind}), the cpu_baseline's own iterations on the FULL workload"
        out["parity"] = block
    return out


def _log(*a):
    print(*a, file=sys.stderr, flush=True)


def pmc_traffic(name):
    """HBM bytes per outer iteration from the committed rocprofv3 PMC summary of this configuration (separate --pmc FETCH_SIZE /
    --pmc WRITE_SIZE passes of tools/pmc_window.py <name>, summarised by tools/pmc_summary.py; 2 x FETCH_SIZE + WRITE_SIZE in KB as
    MI355X_MICROARCH.md's HBM section prescribes for gfx950), summed over every kernel of an iteration.  Only when the summary was
    collected on the kernel sources this build is made of (hash stamped into the summary); otherwise (None, reason)."""
    from bench import kernel_source_hash as _ksh
    fam = "lda" if name.startswith("lda") else name
    kernel_source_hash = lambda: _ksh(fam)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cands = [f"r6_{name}_pmc.json", f"r5_{name}_pmc.json", f"r4_{name}_pmc.json"]
    fname = next((c for c in cands if os.path.exists(os.path.join(root, "profiles", c))), None)
    pmc_traffic.valu_issue_cycles = None
    if fname is None:
        return None, f"no profiles/r6_{name}_pmc.json"
    rows = json.load(open(os.path.join(root, "profiles", fname)))
    meta = rows.get("_meta", {})
    if meta.get("kernel_source_hash") != kernel_source_hash():
        return None, f"profiles/{fname} was collected on kernel sources {meta.get('kernel_source_hash')}, this build is {kernel_source_hash()} (stale)"
    tot = sum((2.0 * r["fetch_kb_per_iteration"] + r["write_kb_per_iteration"]) * 1024.0 for k, r in rows.items() if k != "_meta")
    iss = [r.get("valu_issue_cycles_per_simd_per_iteration") for k, r in rows.items() if k != "_meta"]
    pmc_traffic.valu_issue_cycles = sum(x for x in iss if x) if any(iss) else None     # VALU-issue cycles per SIMD of one iteration's kernels (third pass)
    return tot, f"profiles/{fname} (rocprofv3 --pmc, separate passes; 2*FETCH_SIZE+WRITE_SIZE over every dispatch of {meta.get('iterations')} iterations / that count)"


def _traffic_fields(name, B, sec=None):
    """traffic from the counters, and the SECOND yardstick beside the byte model (round-4 review: LDA K = 100 reports 0.98 of the HBM roofline on
    ALGORITHMIC bytes, more than a copy achieves -- the byte model has saturated): valu_issue_frac = VALU-issue cycles per SIMD of one iteration's kernels
    (4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs, counters collected with the kernels serialised) / (this run's seconds per iteration x 2.4 GHz) -- the share of the
    iteration during which an average SIMD issues a vector instruction.  It is bounded by 1 whatever the caches absorb."""
    t, src = pmc_traffic(name)
    out = {"traffic": t, "traffic_source": src, "traffic_over_algorithmic": (t / B) if t else None}
    iss = getattr(pmc_traffic, "valu_issue_cycles", None)
    out["valu_issue_frac"] = (iss / (sec * 2.4e9)) if (iss and sec) else None
    return out


def window(fn, sync, burnin, warmup, steps):
    """burn-in (state preparation) + warm-up untimed, then `steps` timed iterations between synchronisations."""
    for _ in range(burnin + warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def lda100(burnin=60, warmup=3, steps=20, cpu=True):
    K = 100
    pc = tm.syn_nsf()
    gm = tm.gpuLDA(pc, K)
    beta0 = tm.dirichlet_rows(K, pc.V, seed=7)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()

    def it():
        gm.estep(10, 1.0 / K ** 2); gm.reduce_docs(); gm.update_beta(); gm.update_alpha(1000, 1.0 / K ** 2)
    cold = window(it, gm.synchronize, 0, warmup, steps)
    sec = window(it, gm.synchronize, max(burnin - warmup - steps, 0), warmup, steps)
    hist = gm.sweep_hist(11).tolist()
    B = pc.nnz * (8 + 8 * K) + 12 * pc.M * K + 12 * K * pc.V + 4 * (pc.M + 1)
    line = {"metric": "VB iters/sec, LDA K=100 on NSF-shaped corpus (config 3's model, one GPU)", "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"LDA K=100, SYN-NSF, train! defaults, steady state: {burnin} untimed iterations from the cold start, then "
                                   f"{warmup} warm-up + {steps} timed", "M": pc.M, "V": pc.V, "nnz": pc.nnz, "sweep_hist_last_step": hist},
            "cold_start": {"value": 1.0 / cold, "ms_per_step": 1e3 * cold, "window": f"iterations {warmup + 1}..{warmup + steps} from the cold start"},
            "roofline": {"bound": "hbm", "achieved": B / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": B / sec / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_iteration": B, **_traffic_fields("lda100", B, sec)}}
    gm.close()
    if cpu:
        line["cpu_baseline"] = lda100_cpu(pc)
        line["parity"] = line["cpu_baseline"].pop("parity", None)
    return line


def lda100_cpu(pc=None, parity=True):
    from oracle import oracle as oc
    K = 100
    pc = pc or tm.syn_nsf()
    beta0 = tm.dirichlet_rows(K, pc.V, seed=7)

    def ostep(m, nt):
        m.estep(omp_threads=nt); m.update_beta(); m.update_alpha()
    def check(om, iters, threads):
        from oracle import parity as op
        gm = tm.gpuLDA(pc, K)
        gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
        try:
            return op.lda_parity(gm, om, iters=iters, threads=threads, log=_log)
        finally:
            gm.close()
    out = cpu_line("port of src/LDA.jl train!", lambda: oc.LDA(oc.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V), K, beta0), ostep,
                   1.0, f"FULL SYN-NSF ({pc.M} documents, {pc.nnz} nnz), K=100", warm=1, timed=1, full=True, check=check if parity else None)
    oc.lib().orc_omp_pool_free()
    return out


def ctm(burnin=60, warmup=2, steps=8, cpu=True, K=50):        # 60 burn-in iterations: the same steady-state definition as bench.py's LDA window (30 -> 202, 60 -> 206, 100 -> 214 it/s)
    pc = tm.syn_nsf()
    gm = tm.gpuCTM(pc, K)
    beta0 = tm.dirichlet_rows(K, pc.V, seed=7)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()

    def it():
        gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_sigma(); gm.update_mu()
    cold = window(it, gm.synchronize, 0, warmup, steps)
    sec = window(it, gm.synchronize, max(burnin - warmup - steps, 0), warmup, steps)
    es_ms = gm.last_estep_ms()
    hist, newton = gm.sweep_hist()
    st = gm.solver_stats()
    sweeps = int(sum(i * int(h) for i, h in enumerate(hist)))
    KP = 4 * ((K + 3) // 4); KP += 4 * (1 - (KP // 4) % 2)          # 4 * odd
    lane_kernel = KP <= 52
    B = pc.nnz * (8 + 8 * K) + 16 * pc.M * K + 8 * pc.M + 12 * K * pc.V + 8 * K * K
    tok = 6.0 * K * pc.nnz * (sweeps / pc.M)
    F_nom = newton * (K ** 3 / 3.0 + 4 * K * K) + tok
    # what the lane-per-document kernel issues: a wave runs every loop until its slowest lane is done (64 lanes per trip);
    # one CG trip = one K x K mat-vec + 5 vector updates, one Newton trip = the gradient's mat-vec + exp / assembly
    # (the K > 52 kernel runs one wave per document: its trip counts are per document)
    per_trip = 64.0 if lane_kernel else 1.0
    F_exec = per_trip * (st["cg_trips"] * (2 * KP * KP + 12 * KP) + st["newton_trips"] * (2 * KP * KP + 24 * KP)) + tok if st["waves"] else None
    # a checked iteration as train! runs it (checkelbo = 1): the same model with every E-step collecting the ELBO's parts (TMVB_CTM_ELBO_PARTS=2)
    checked = None
    if K == 50:
        import ctypes as C
        gc = tm.gpuCTM(pc, K)
        gc.beta = np.asfortranarray(beta0); gc.beta_old = gc.beta.copy(order="F"); gc.update_buffer()
        L = tm._lib.lib()

        def train_c(n, ce):                                     # the library's own train! loop (tmvb_ctm_train), no Python between the kernels
            buf = np.full(n, np.nan); done, base = C.c_int32(0), C.c_double(0.0)
            tm._lib.check(L.tmvb_ctm_train(gc.handle, C.c_int32(n), C.c_double(0.0), C.c_int32(1000), C.c_double(1.0 / K ** 2), C.c_int32(10),
                                           C.c_double(1.0 / K ** 2), C.c_int32(ce), buf.ctypes.data_as(C.POINTER(C.c_double)), C.byref(done), C.byref(base)))
        train_c(12, 0)
        gc.synchronize(); t0 = time.perf_counter()
        train_c(8, 0)
        gc.synchronize(); t1 = time.perf_counter()
        train_c(16, 1)                                          # (a call's baseline evaluation and its one doubled evaluation inside: 2 x 1.5 ms over 16 iterations)
        gc.synchronize(); t2 = time.perf_counter()
        checked = {"ms_per_step_same_window": 1e3 * (t1 - t0) / 8, "ms_per_checked_step": 1e3 * (t2 - t1) / 16,
                   "elbo_form": "decomposed" if gc.elbo_form() == 1 else "token walk", "window": "iterations 13..20 unchecked / 21..36 checked from the cold start"}
        gc.close()
    line = {"metric": f"VB iters/sec, CTM K={K} on NSF-shaped corpus" + (" (config 4)" if K == 50 else ""), "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "estep_ms": es_ms, "checked": checked, "dtype": "f32 (fp64 gradients / logzeta / vsq)", "data": "synthetic",
            "config": {"workload": f"CTM K={K}, SYN-NSF, train! defaults (niter=1000 ntol=1/K^2 viter=10 vtol=1/K^2), steady state: {burnin} untimed "
                                   f"iterations from the cold start, then {warmup} warm-up + {steps} timed",
                       "M": pc.M, "V": pc.V, "nnz": pc.nnz, "lambda_newton_steps_last_iteration": int(newton), "sweeps_last_iteration": sweeps,
                       "sweep_hist_last_step": [int(h) for h in hist],
                       "kernel": (("ctm_estep_quad_kernel (lane per document, four waves per 64 documents: topic quarters in the Newton phases, four lanes per document in the token walk; Jacobi-preconditioned CG Newton solves)"
                                   if os.environ.get("TMVB_CTM_QUAD", "1") != "0" and not os.environ.get("TMVB_CTM_PROF") else
                                   "ctm_estep_batch_kernel (lane per document, one wave per 64 documents, Jacobi-preconditioned CG Newton solves)") if lane_kernel else
                                  "ctm_estep_generic_kernel<.., CG> (wave per document, lane = matrix row, CG against invsigma in LDS)") if st["waves"] else "wave per document",
                       "cg_wave_trips": st["cg_trips"], "newton_wave_trips": st["newton_trips"]},
            "cold_start": {"value": 1.0 / cold, "ms_per_step": 1e3 * cold, "window": f"iterations {warmup + 1}..{warmup + steps} from the cold start"},
            "roofline": {"bound": "valu (packed fp32 CG mat-vecs with invsigma streamed through SGPRs; f32 MFMA has the same peak and is not used in the solve)",
                         "achieved": F_nom / sec / 1e12, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": F_nom / sec / 1e12 / F32_PEAK_TFLOPS,
                         "flops_per_iteration_nominal": F_nom,
                         "nominal_is": "SURVEY.md 8d: newton steps x (K^3/3 + 4K^2) + 6 K N_d per sweep -- the flops a direct factorisation per Newton step would execute",
                         "flops_per_iteration_executed": F_exec, "executed_TFLOPs": (F_exec / sec / 1e12) if F_exec else None,
                         "executed_frac": (F_exec / sec / 1e12 / F32_PEAK_TFLOPS) if F_exec else None,
                         "executed_is": "64 lanes x (CG item trips x (2 KP^2 + 12 KP) + Newton item trips x (2 KP^2 + 24 KP)) + token phase; idle lanes of a trip included (the trips are counted per 64-document item in both lane kernels)",
                         "hbm_GBs": B / sec / 1e9, "hbm_frac": B / sec / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": B,
                         **(_traffic_fields("ctm", B, sec) if K == 50 else {"traffic": None})}}
    gm.close()
    if cpu and K == 50:
        line["cpu_baseline"] = ctm_cpu(pc)
        line["parity"] = line["cpu_baseline"].pop("parity", None)
    return line


def ctm100(cpu=False):
    """CTM K = 100 on SYN-NSF (not a BASELINE.json configuration: the K > 52 kernel's line, no CPU baseline)."""
    return ctm(burnin=20, warmup=2, steps=6, cpu=False, K=100)


def ctm_cpu(pc=None, parity=True):
    from oracle import oracle as oc
    K = 50
    pc = pc or tm.syn_nsf()
    beta0 = tm.dirichlet_rows(K, pc.V, seed=7)

    def ostep(m, nt):
        m.estep(omp_threads=nt); m.update_beta(); m.update_sigma_mu()
    def check(om, iters, threads):
        from oracle import parity as op
        gm = tm.gpuCTM(pc, K)
        gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
        try:
            return op.ctm_parity(gm, om, iters=iters, threads=threads, log=_log)
        finally:
            gm.close()
    # two iterations: the cold start (lambda = 0, sigma = I) and the state one iteration later (mu, sigma, lambda moved)
    return cpu_line("port of src/CTM.jl train!", lambda: oc.CTM(oc.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V), K, beta0), ostep,
                    1.0, f"FULL SYN-NSF ({pc.M} documents), K=50; one iteration takes ~5-10 s, so no warm-up iteration", warm=0, timed=2 if parity else 1,
                    full=True, check=check if parity else None)


def ctpf(burnin=300, warmup=10, steps=200, cpu=True):
    K = 50
    pc = tm.syn_citeu()
    gm = tm.gpuCTPF(pc, K)

    def it():
        gm.estep(); gm.reduce_docs(); gm.mstep()
    cold = window(it, gm.synchronize, 0, 5, 50)
    sec = window(it, gm.synchronize, max(burnin - 55 - warmup - steps, 0), warmup, steps)
    hist = gm.sweep_hist().tolist()
    # the E-step's own time: from a second model created with TMVB_ESTEP_TIMING=1 (the library records its two timing events -- default
    # events, a system-scope release each, on the stream the whole iteration runs on -- only then; the timed model runs without them)
    os.environ["TMVB_ESTEP_TIMING"] = "1"
    gt = tm.gpuCTPF(pc, K)
    del os.environ["TMVB_ESTEP_TIMING"]
    es = []
    for i in range(burnin + 10):
        gt.estep()
        if i >= burnin:
            es.append(gt.last_estep_ms())
        gt.reduce_docs(); gt.mstep()
    es_ms = float(np.mean(es))
    gt.close()
    # a checked iteration as train! runs it under its default checkelbo = 1 (update_elbo! on the device after every M-step): the
    # library's own loop (tmvb_ctpf_train, no Python between the kernels) on a second model brought to the same steady state -- 200 unchecked, then 200 checked iterations; a call's baseline evaluation and its one doubled evaluation at the switch of forms (tmvb_train.h) are inside: < 1 %
    import ctypes as C
    gc = tm.gpuCTPF(pc, K)
    L = tm._lib.lib()

    def train_c(n, ce):
        buf = np.full(n, np.nan); done, base = C.c_int32(0), C.c_double(0.0)
        tm._lib.check(L.tmvb_ctpf_train(gc.handle, C.c_int32(n), C.c_double(0.0), C.c_int32(10), C.c_double(1.0 / K ** 2), C.c_int32(ce),
                                        buf.ctypes.data_as(C.POINTER(C.c_double)), C.byref(done), C.byref(base)))
    train_c(burnin, 0); train_c(10, 1)                         # the same steady state as the window above (and the per-document constants)
    gc.synchronize(); t0 = time.perf_counter()
    train_c(200, 0)
    gc.synchronize(); t1 = time.perf_counter()
    train_c(200, 1)
    gc.synchronize(); t2 = time.perf_counter()
    checked, unchecked_same = (t2 - t1) / 200, (t1 - t0) / 200
    elbo_form = gc.elbo_form()
    gc.close()
    B = pc.nnz * (8 + 8 * K) + pc.nR * (8 + 8 * K) + 16 * pc.M * K + 12 * K * (pc.V + pc.U)
    ms_s, ms_r = gm.recommend(scores=False)
    line = {"metric": "VB iters/sec, CTPF K=50 on CiteULike-shaped corpus (config 5)", "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "estep_ms": es_ms, "ms_per_checked_step": 1e3 * checked, "ms_per_step_train_loop": 1e3 * unchecked_same,
            "checked_elbo_form": "decomposed" if elbo_form == 1 else "table form",
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"CTPF K=50, SYN-CITEU with readers, train! defaults (viter=10 vtol=1/K^2 checkelbo=Inf), steady state: {burnin} untimed "
                                   f"iterations from the cold start, then {warmup} warm-up + {steps} timed",
                       "M": pc.M, "V": pc.V, "U": pc.U, "nnz": pc.nnz, "nR": pc.nR, "sweep_hist_last_step": hist},
            "cold_start": {"value": 1.0 / cold, "ms_per_step": 1e3 * cold, "window": "iterations 6..55 from the cold start"},
            "roofline": {"bound": "hbm", "achieved": B / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": B / sec / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_iteration": B, **_traffic_fields("ctpf", B, sec)},
            "recommend": {"ms_scores": ms_s, "ms_rank": ms_r, "pairs": pc.M * pc.U}}
    gm.close()
    if cpu:
        line["cpu_baseline"] = ctpf_cpu(pc)
        line["parity"] = line["cpu_baseline"].pop("parity", None)
    return line


def ctpf_cpu(pc=None, parity=True):
    from oracle import oracle as oc
    K = 50
    pc = pc or tm.syn_citeu()
    alef0 = np.exp(tm.dirichlet_rows(K, pc.V, seed=7) - 0.5)

    def ostep(m, nt):
        m.estep(omp_threads=nt); m.mstep()
    def check(om, iters, threads):
        from oracle import parity as op
        gm = tm.gpuCTPF(pc, K)
        gm.alef = np.asfortranarray(alef0); gm.alef_old = gm.alef.copy(order="F"); gm.update_buffer()
        try:
            return op.ctpf_parity(gm, om, iters=iters, threads=threads, elbo=True, log=_log)      # (round-4 review: elbo_rel was null in the line)
        finally:
            gm.close()
    return cpu_line("port of src/CTPF.jl train!",
                    lambda: oc.CTPF(oc.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V, pc.rdr_ptr, pc.readers, pc.ratings, pc.U), K, alef0),
                    ostep, 1.0, f"FULL SYN-CITEU ({pc.M} documents, {pc.nnz} term + {pc.nR} reader entries), K=50", warm=1, timed=2, full=True,
                    check=check if parity else None)


def flda(burnin=60, warmup=3, steps=20, cpu=True):
    K = 50
    pc = tm.syn_nsf()
    gm = tm.gpufLDA(pc, K)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F")
    gm.kappa = tm.dirichlet_rows(1, pc.V, seed=9)[0].copy(); gm.kappa_old = gm.kappa.copy(); gm.update_buffer()

    def it():
        gm.estep(10, 1.0 / K ** 2); gm.reduce_docs(); gm.update_beta(); gm.update_alpha(1000, 1.0 / K ** 2); gm.update_eta()
    cold = window(it, gm.synchronize, 0, warmup, steps)
    sec = window(it, gm.synchronize, max(burnin - warmup - steps, 0), warmup, steps)
    es = gm.last_estep_ms()
    # LDA's bytes + tau / tau_old / lse (read + write per token entry) + the kappa statistics
    B = pc.nnz * (8 + 8 * K + 24) + 12 * pc.M * K + 12 * K * pc.V + 8 * pc.V + 4 * (pc.M + 1)
    line = {"metric": "VB iters/sec, fLDA K=50 on NSF-shaped corpus (section 8 row f4)", "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "estep_ms": es, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"fLDA K=50, SYN-NSF, train! defaults, steady state: {burnin} untimed iterations, then {warmup} warm-up + {steps} timed",
                       "M": pc.M, "V": pc.V, "nnz": pc.nnz},
            "cold_start": {"value": 1.0 / cold, "ms_per_step": 1e3 * cold},
            "roofline": {"bound": "hbm", "achieved": B / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": B / sec / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_iteration": B, "traffic": None}}
    gm.close()
    if cpu:
        from oracle import oracle as oc
        sh = pc.shard(0, 16000)
        beta0 = tm.dirichlet_rows(K, pc.V, seed=7); kappa0 = tm.dirichlet_rows(1, pc.V, seed=9)[0]

        def ostep(m, nt):
            m.estep(omp_threads=nt); m.mstep()
        line["cpu_baseline"] = cpu_line("port of src/fLDA.jl train!", lambda: oc.fLDA(oc.CSR(sh.doc_ptr, sh.terms, sh.counts, sh.V), K, beta0, kappa0), ostep,
                                        sh.nnz / pc.nnz, f"first {sh.M} documents of SYN-NSF ({sh.nnz} of {pc.nnz} nnz), K=50")
    return line


def fctm(burnin=30, warmup=2, steps=6, cpu=True):
    K = 50
    pc = tm.syn_nsf()
    gm = tm.gpufCTM(pc, K)
    gm.beta = np.asfortranarray(tm.dirichlet_rows(K, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F")
    gm.kappa = tm.dirichlet_rows(1, pc.V, seed=9)[0].copy(); gm.kappa_old = gm.kappa.copy(); gm.update_buffer()

    def it():
        gm.estep(); gm.reduce_docs(); gm.mstep()
    cold = window(it, gm.synchronize, 0, warmup, steps)
    sec = window(it, gm.synchronize, max(burnin - warmup - steps, 0), warmup, steps)
    hist, newton = gm.sweep_hist()
    sweeps = int(sum(i * int(h) for i, h in enumerate(hist)))
    B = pc.nnz * (8 + 8 * K + 24) + 16 * pc.M * K + 8 * pc.M + 12 * K * pc.V + 8 * pc.V + 8 * K * K
    F = newton * (K ** 3 / 3.0 + 4 * K * K) + 8.0 * K * pc.nnz * (sweeps / pc.M)
    line = {"metric": "VB iters/sec, fCTM K=50 on NSF-shaped corpus (section 8 row f4)", "value": 1.0 / sec, "unit": "VB iters/sec",
            "ms_per_step": 1e3 * sec, "dtype": "f32 (fp64 gradients / logzeta / vsq)", "data": "synthetic",
            "config": {"workload": f"fCTM K=50, SYN-NSF, train! defaults, steady state: {burnin} untimed iterations, then {warmup} warm-up + {steps} timed",
                       "M": pc.M, "V": pc.V, "nnz": pc.nnz, "lambda_newton_steps_last_iteration": int(newton), "sweeps_last_iteration": sweeps},
            "cold_start": {"value": 1.0 / cold, "ms_per_step": 1e3 * cold},
            "roofline": {"bound": "valu (packed fp32 CG mat-vecs, K v_exp_f32 per token per sweep)", "achieved": F / sec / 1e12, "peak": F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": F / sec / 1e12 / F32_PEAK_TFLOPS, "flops_per_iteration_nominal": F,
                         "hbm_GBs": B / sec / 1e9, "hbm_frac": B / sec / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": B, "traffic": None}}
    gm.close()
    if cpu:
        from oracle import oracle as oc
        sh = pc.shard(0, 4000)
        beta0 = tm.dirichlet_rows(K, pc.V, seed=7); kappa0 = tm.dirichlet_rows(1, pc.V, seed=9)[0]

        def ostep(m, nt):
            m.estep(omp_threads=nt); m.mstep()
        line["cpu_baseline"] = cpu_line("port of src/fCTM.jl train!", lambda: oc.fCTM(oc.CSR(sh.doc_ptr, sh.terms, sh.counts, sh.V), K, beta0, kappa0), ostep,
                                        sh.M / pc.M, f"first {sh.M} documents of SYN-NSF (the Newton solves scale with the document count), K=50")
    return line


ALL = {"lda100": lda100, "ctm": ctm, "ctpf": ctpf, "flda": flda, "fctm": fctm, "ctm100": ctm100}
CPU = {"lda100": lda100_cpu, "ctm": ctm_cpu, "ctpf": ctpf_cpu}

if __name__ == "__main__":
    argv = sys.argv[1:]
    gpu_only = "--gpu-only" in argv                       # no cpu_baseline (bench.py adds it from its own process)
    which = [a for a in argv if not a.startswith("--")] or list(ALL)
    for w in which:
        print(json.dumps(ALL[w](cpu=False) if gpu_only else ALL[w]()), flush=True)
