#!/usr/bin/env python
"""
bench.py -- VB outer iterations per second of the LDA hot path on SYN-NSF (M=128804, V=25319), K=50
(BASELINE.json configs[1]), document-sharded over N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one outer iteration of train! (src/LDA.jl:169-183) with checkelbo=Inf: fused E-step over this rank's
documents, the Elogtheta_sum reduction, ONE all-reduce of the packed K*V+K statistics (N>1; RCCL inside libtmvb_hip.so
through tmvb_comm_allreduce, the communicator's unique id bootstrapped over torch.distributed), then update_beta! and
update_alpha! on every rank.  The corpus and the state are resident in HBM before the timed region.

STEADY STATE (round-1 review): the per-document sweep counts grow over the first ~50 iterations of a cold start
(5.3 -> 8.3 sweeps per document).  In round 1 the cold-start window overstated the rate by ~15 %; with the register-tile
document kernels the sweeps are no longer what sets the E-step's span (57 % more sweeps cost 3 %: the statistics chain
does, profiles/r6_lda_cold_vs_steady.txt), so today the two windows differ by 2 - 3 %.  The definition stays: the model is
first brought to its operating point with `--burnin` (default 60) untimed iterations -- state preparation, like the
corpus upload -- then W warm-up iterations, then exactly K timed iterations between barrier+synchronize pairs; the time
is the MAX over ranks.  `value` is that steady-state rate; the cold-start rate of the same W+K window (what round 1
reported) is carried in "cold_start".  The corpus is fixed as N grows ("scaling": "strong").

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the E-step against the HBM roofline (algorithmic bytes, HIP-event timing on the launch stream),
  elbo_plateau : wall time until check_elbo!'s own stop rule (delta_elbo < tol = 1.0, src/modelutils.jl:580) fires from
                 a cold start with checkelbo = 1 -- or "reached": false after the iteration cap,
  cpu_baseline : the fp64 oracle (a port of the reference's CPU path -- the reference itself is Julia and cannot run
                 here) timed on this host's cores on the FULL corpus.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the engine's default (tmvb_core.hip: tmvb_env_defaults), set here because torch initialises HIP before the library is loaded
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def estep_bytes(nnz, M, K):
    """Algorithmic bytes of one E-step: per token ids+count (8) +
    one beta-column gather (4K) + one statistics scatter (4K); per document read Elogtheta, write gamma
    and Elogtheta (12K); doc_ptr (4(M+1))."""
    return nnz * (8 + 8 * K) + 12 * M * K + 4 * (M + 1)


def mstep_bytes(K, V):
    return 12 * K * V


KERNEL_SOURCES = {
    # the sources the kernels of one configuration are compiled from (the statistics pass, the reductions and the shared headers
    # belong to every model); a PMC summary is valid for a build whose files of ITS model are unchanged
    "lda": ("tmvb_lda.hip", "tmvb_gridtile.h", "tmvb_regtile.h", "tmvb_termstats.h", "tmvb_common_kernels.h", "tmvb_internal.h"),
    "ctm": ("tmvb_ctm.hip", "tmvb_ctm_batch.h", "tmvb_ctm_quad.h", "tmvb_filtered.h", "tmvb_termstats.h", "tmvb_common_kernels.h", "tmvb_internal.h"),
    "ctpf": ("tmvb_ctpf.hip", "tmvb_gridtile.h", "tmvb_regtile.h", "tmvb_termstats.h", "tmvb_common_kernels.h", "tmvb_internal.h"),
}


def kernel_source_hash(model="lda"):
    """sha256 over the kernel sources of one model family (lda: LDA K=50 / K=100; ctm; ctpf): PMC numbers collected on other
    kernels are stale."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "topicmodelsvb.jl_amd", "csrc")
    for f in KERNEL_SOURCES[model]:
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(K, M, nnz):
    """HBM traffic of one E-step from the committed rocprofv3 PMC summary (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE
    passes of this same command, tools/pmc_summary.py), corrected as MI355X_MICROARCH.md's HBM section prescribes for
    gfx950 (2 x FETCH_SIZE; KB units).  Returned only when the summary was collected on this workload AND on the kernel
    sources this run is built from (source hash stamped into the summary); otherwise null with the reason."""
    for name in ("r6_lda50_pmc.json", "r5_lda50_pmc.json", "r4_lda50_pmc.json", "r3_lda_k50_pmc.json", "r2_lda_k50_pmc.json", "r1_lda_k50_pmc.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            break
    else:
        return None, "no PMC summary under profiles/"
    if not (K == 50 and M == 128804 and nnz == 10929864):
        return None, "PMC summary is for SYN-NSF K=50 on one GPU"
    rows = json.load(open(path))
    meta = rows.get("_meta", {})
    if meta.get("kernel_source_hash") != kernel_source_hash():
        return None, f"{name} was collected on kernel sources {meta.get('kernel_source_hash')}, this build is {kernel_source_hash()} (stale)"
    tot, issue = 0.0, 0.0
    for kname, r in rows.items():
        if "lda_estep" in kname or "termstats" in kname:
            if "fetch_kb_per_iteration" not in r:
                return None, "incomplete PMC summary"
            tot += (2.0 * r["fetch_kb_per_iteration"] + r["write_kb_per_iteration"]) * 1024.0
            issue += r.get("valu_issue_cycles_per_simd_per_iteration", float("nan"))
    pmc_traffic.valu_issue_cycles = issue if issue == issue else None      # VALU-issue cycles per SIMD of one E-step's kernels (third pass)
    return tot, f"profiles/{name} (rocprofv3 --pmc, separate passes; 2*FETCH_SIZE+WRITE_SIZE summed over the E-step's dispatches of one iteration)"


# ---------------------------------------------------------------------------------------------------------------- the ONE line
# Round-4 review: the line had grown to 20 KB (per-iteration parity dumps, clock log, paragraphs) and the driver could not parse
# it.  The contract line on stdout is now SLIM (< 6000 bytes, tests/test_bench_line.py bounds it): the contract fields, and per
# configuration only the numbers a reader checks.  Everything else -- per-iteration parity rows, per-step E-step times, the clock
# log, the ELBO-vs-wall-clock curve, tolerances, the explanatory strings, multi_gpu_check's arrays -- goes to bench_detail.json
# next to this script (and to gpurun_out/ when that directory exists) and to stderr.
LINE_LIMIT = 6000
ROOFLINE_KEEP = ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes", "traffic", "traffic_over_algorithmic", "valu_issue_frac",
                 "issue_model_frac", "estep_ms", "estep_ms_median", "kernel_source_hash", "executed_frac", "hbm_frac")
CPU_KEEP = ("value", "unit", "cores", "kind", "single_thread_value", "sample")
PARITY_KEEP = ("pass", "worst", "documents", "iterations")


def _r(x, sig=7):
    """floats to `sig` significant digits (the detail file keeps them all)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _slim_roofline(rf):
    if not rf:
        return rf
    rf = dict(rf)
    for a in ("algorithmic_bytes_per_estep", "algorithmic_bytes_per_iteration"):
        if a in rf:
            rf["algorithmic_bytes"] = rf[a]
    if rf.get("traffic") and rf.get("algorithmic_bytes") and "traffic_over_algorithmic" not in rf:
        rf["traffic_over_algorithmic"] = rf["traffic"] / rf["algorithmic_bytes"]
    out = {k: rf[k] for k in ROOFLINE_KEEP if k in rf}
    if isinstance(out.get("bound"), str):
        out["bound"] = out["bound"].split(" ")[0]                # "valu (packed fp32 ...)" -> "valu"
    return out


def _slim_cpu(cb):
    if not cb or "error" in cb:
        return cb
    out = {k: cb[k] for k in CPU_KEEP if k in cb}
    if "sample" in out:
        out["sample"] = cb.get("sample_short") or str(out["sample"])[:90]
    return out


def _slim_parity(p):
    if not p:
        return p
    return {k: p[k] for k in PARITY_KEEP if k in p}


def _slim_config(line):
    c = line.get("config") or {}
    out = {"workload": str(c.get("workload", "")).split(";")[0].split(", steady state")[0][:120]}
    for k in ("K", "M", "V", "U", "nnz", "nR", "burnin", "parallelism", "collective"):
        if k in c:
            out[k] = c[k] if not isinstance(c[k], str) else c[k][:70]
    return out


def slim_line(result):
    """The contract line from the full result dict (pure function: tests/test_bench_line.py feeds it a canned result)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: result.get(k) for k in keep}
    out["config"] = _slim_config(result)
    out["roofline"] = _slim_roofline(result.get("roofline"))
    out["cpu_baseline"] = _slim_cpu(result.get("cpu_baseline"))
    out["parity"] = _slim_parity(result.get("parity"))
    if result.get("cold_start"):
        out["cold_start"] = {k: result["cold_start"].get(k) for k in ("value", "ms_per_step")}
    pl = result.get("elbo_plateau")
    if pl:
        out["elbo_plateau"] = {k: pl.get(k) for k in ("reached", "seconds", "iterations", "elbo_first", "elbo_last", "last_delta")}
        if pl.get("seconds_per_checked_iteration") is not None:      # train! checks after every iteration (checkelbo = 1): what one such iteration costs
            out["elbo_plateau"]["ms_per_checked_iteration"] = round(1e3 * pl["seconds_per_checked_iteration"], 4)
        if pl.get("elbo_form") is not None:
            out["elbo_plateau"]["elbo_form"] = pl["elbo_form"]
    mg = result.get("multi_gpu_check")
    if mg:
        out["multi_gpu_check"] = {k: mg[k] for k in ("pass", "iterations", "globals_hash_equal", "elbo_rel_vs_n1", "elbo_rel_tolerance", "form", "fallback", "skipped",
                                                     "shard_nnz_max_over_min", "config3_k100") if k in mg}
    if result.get("other_configs"):
        oc = {}
        for name, ln in result["other_configs"].items():
            if "error" in ln:
                oc[name] = {"error": str(ln["error"])[:200]}
                continue
            o = {k: ln.get(k) for k in ("metric", "value", "unit", "ms_per_step", "estep_ms", "dtype") if k in ln}
            o["dtype"] = str(o.get("dtype", "f32")).split(" ")[0]
            # what an iteration costs with check_elbo! behind it (train!'s default checkelbo = 1), as train! runs it
            if ln.get("ms_per_checked_step") is not None:
                o["ms_per_checked_step"] = ln["ms_per_checked_step"]
            elif isinstance(ln.get("checked"), dict):
                o["ms_per_checked_step"] = ln["checked"].get("ms_per_checked_step")
            o["config"] = _slim_config(ln)
            o["roofline"] = _slim_roofline(ln.get("roofline"))
            o["cpu_baseline"] = _slim_cpu(ln.get("cpu_baseline"))
            o["parity"] = _slim_parity(ln.get("parity"))
            oc[name] = o
        out["other_configs"] = oc
    out["detail"] = "bench_detail.json"
    out = _r(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:                                  # never print a line the driver cannot take: shed the side configurations' extras
        for o in out.get("other_configs", {}).values():
            for k in ("parity", "cpu_baseline", "config"):
                if isinstance(o.get(k), dict):
                    o[k] = {kk: vv for kk, vv in o[k].items() if kk in ("pass", "value", "cores", "workload")}
        line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:
        out.pop("other_configs", None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def write_detail(result):
    """The full record (everything the slim line leaves out) -> bench_detail.json next to the script, gpurun_out/ when present, stderr."""
    text = json.dumps(result, indent=1, default=str)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(text)
            except OSError as e:
                log(f"bench_detail.json not written under {d}: {e}")
    log("bench detail: " + json.dumps(result, default=str))


def usable_cpus():
    """CPUs this process may really use: the affinity mask and the cgroup CPU quota (containers often see every host
    core in os.cpu_count() while being limited to a few)."""
    n = os.cpu_count() or 1
    info = {"os_cpu_count": n}
    try:
        info["sched_affinity"] = len(os.sched_getaffinity(0))
        n = min(n, info["sched_affinity"])
    except Exception:
        pass
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()          # cgroup v2: "<quota> <period>" or "max <period>"
        if q[0] != "max":
            info["cgroup_cpu_quota"] = float(q[0]) / float(q[1])
            n = min(n, max(1, int(info["cgroup_cpu_quota"] + 0.5)))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                info["cgroup_cpu_quota"] = quota / period
                n = min(n, max(1, int(quota / period + 0.5)))
        except Exception:
            pass
    return n, info


def cpu_baseline(tm, corpus, K, beta0, budget_s=24.0):
    """fp64 oracle (oracle/*.c) on the FULL corpus: OpenMP document-parallel E-step on every host core, per-thread
    statistics allocated once; then one single-thread iteration (the reference is single-threaded) if the budget allows.

    PARITY IN THE SAME RUN (round-3 review: "compare what you time with the oracle"): the 1 warm-up + 2 timed oracle
    iterations of the widest team are at the same time the checker of the HIP engine -- a gpuLDA on the same full corpus is
    reset to the oracle's state before each of them (teacher forcing), runs the same outer iteration, and every document and
    every global is compared (oracle/parity.py; tolerances of SURVEY.md section 8c).  Returns (cpu_baseline, parity)."""
    import numpy as np
    from oracle import oracle as oc
    from oracle import parity as op
    oc.build()
    host_cores, cpu_info = usable_cpus()
    threads = host_cores
    csr = oc.CSR(corpus.doc_ptr, corpus.terms, corpus.counts, corpus.V)

    def run(nthreads, warm, max_iters, budget):
        m = oc.LDA(csr, K, beta0)
        for _ in range(warm):
            m.estep(omp_threads=nthreads); m.update_beta(); m.update_alpha()
        t0 = time.perf_counter()
        done = 0
        while done < max_iters:
            m.estep(omp_threads=nthreads); m.update_beta(); m.update_alpha()
            done += 1
            if time.perf_counter() - t0 > budget:
                break
        return done / (time.perf_counter() - t0), done

    t0 = time.perf_counter()
    # the private-statistics reduction grows with the thread count, so the fastest team is not always the widest:
    # try all usable CPUs, a half and a quarter (1 warm-up + 2 timed iterations each) and keep the best
    tried = {}
    # widest team: its three iterations are the parity check's oracle iterations (timed: the oracle's own calls only)
    gm = tm.gpuLDA(corpus, K)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
    block, secs = op.lda_parity(gm, oc.LDA(csr, K, beta0), iters=3, threads=host_cores, log=log)
    gm.close()
    tried[host_cores] = 2.0 / (secs[1] + secs[2])
    oc.lib().orc_omp_pool_free()
    for nt in sorted({max(1, host_cores // 2), max(1, host_cores // 4)} - {host_cores}, reverse=True):
        tried[nt], _ = run(nt, 1, 2, budget_s / 8)
        oc.lib().orc_omp_pool_free()
    threads = max(tried, key=tried.get)
    omp, n_omp = tried[threads], 2
    left = budget_s - (time.perf_counter() - t0)
    single, n_single, single_note = None, 0, "skipped (budget)"
    if host_cores > 1 and left >= 9.0:                           # a full single-thread iteration takes ~6 s at this size
        single, n_single = run(0, 0, 1, left)
        single_note = f"{n_single} full-corpus iteration from the cold start"
    elif host_cores > 1 and left > 2.0:
        sh = corpus.shard(0, min(16000, corpus.M))
        m = oc.LDA(oc.CSR(sh.doc_ptr, sh.terms, sh.counts, sh.V), K, beta0)
        t1 = time.perf_counter()
        m.estep(); m.update_beta(); m.update_alpha()
        single = (sh.nnz / max(corpus.nnz, 1)) / (time.perf_counter() - t1)
        single_note = f"1 iteration on the first {sh.M} documents ({sh.nnz} nnz), scaled by the nnz fraction"
    oc.lib().orc_omp_pool_free()
    block["against"] = "fp64 C oracle (port of src/LDA.jl), the cpu_baseline's own iterations: cold start + 2 more teacher-forced outer iterations on the FULL workload"
    return {
        "value": omp, "unit": "VB iters/sec", "cores": threads, "host_cpus": cpu_info, "kind": "port",
        "omp_threads_tried": {str(k): v for k, v in tried.items()},
        "single_thread_value": single,
        "sample_short": f"fp64 C oracle, FULL corpus, 1 warm-up + {n_omp} timed iterations, OpenMP x{threads}",
        "sample": f"fp64 C oracle (port of src/LDA.jl train!; the Julia reference cannot run here) on the FULL workload "
                  f"({corpus.M} docs, {corpus.nnz} nnz), same cold start (alpha=1, gamma=1, beta0 seed 7): 1 warm-up + {n_omp} timed "
                  f"iterations with the OpenMP document-parallel E-step on {threads} threads (the CPUs usable by this process: affinity mask and cgroup quota, see host_cpus), per-thread "
                  f"statistics allocated once (E-step + update_beta! + update_alpha!; on the widest team these iterations are also the parity check's, see `parity`); single thread: {single_note}.  Cold-start iterations run fewer sweeps per "
                  f"document than the GPU's steady-state window, which favours the CPU figure.",
    }, block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--burnin", type=int, default=60, help="untimed iterations that bring the model to its steady state before warm-up")
    ap.add_argument("--clock-warmup", type=float, default=0.6,
                    help="seconds of untimed iterations on a scratch model before anything is measured (clocks, first touches); 0 = off")
    ap.add_argument("--K", type=int, default=50)
    ap.add_argument("--docs", type=int, default=128804)
    ap.add_argument("--vocab", type=int, default=25319)
    ap.add_argument("--ar-slices", type=int, default=None,
                    help="N > 1, in-library communicator: vocabulary slabs the statistics all-reduce is issued in under the last statistics "
                         "pass (tmvb_lda_estep_allreduce; default: the library's, 1 = the statistics in one collective after the pass, the Elogtheta_sum tail in "
                         "an early one of its own)")
    ap.add_argument("--collective", choices=["lib", "torch"], default="lib",
                    help="lib: RCCL inside libtmvb_hip.so (tmvb_comm_allreduce); torch: torch.distributed all_reduce on the bound buffer")
    ap.add_argument("--plateau-cap", type=int, default=4000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plateau", action="store_true")
    ap.add_argument("--no-cold", action="store_true")
    ap.add_argument("--torch-events", action="store_true", help="diagnostics: torch.cuda.Event (default HIP events) around the E-step instead of the library's")
    ap.add_argument("--no-step-events", action="store_true",
                    help="diagnostics: no HIP events around the E-step inside the timed window (the line then has no valid roofline block); "
                         "what the two events per step cost")
    ap.add_argument("--other-configs-inline", action="store_true", help="measure the side configurations in this process instead of fresh ones")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the lines of BASELINE.json's configs 3-5 (LDA K=100, CTM K=50, CTPF K=50) carried under other_configs at N=1")
    args = ap.parse_args()
    if args.ar_slices is not None:
        os.environ["TMVB_AR_SLICES"] = str(args.ar_slices)       # read by the library when the first plan is agreed on

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")

    import numpy as np
    import torch
    import torch.distributed as dist
    import tmvb_amd
    tm = tmvb_amd.pkg
    from tmvb_amd_pkg.dist import HipLDAEngine, ShardedLDA

    if not torch.cuda.is_available() or tm.lib().tmvb_device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    one_gpu_debug = os.environ.get("TMVB_DIST_BACKEND", "nccl") != "nccl"    # "gloo": smoke-test the N>1 plumbing on ONE GPU
    if one_gpu_debug:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the default group (gloo) carries only host-side control traffic: barriers, the RCCL unique id (through its
        # store), the max over ranks.  The data path is RCCL inside the library; --collective torch (or a failed
        # in-library init) makes a second, nccl (= RCCL) group for torch.distributed's all_reduce instead.
        dist.init_process_group("gloo", rank=rank, world_size=world)

    K, V = args.K, args.vocab
    t0 = time.perf_counter()
    corpus = tm.syn_nsf(M=args.docs, V=V)                 # identical bytes on every rank (seeded)
    bounds = corpus.shard_bounds(world)
    d0, d1 = bounds[rank]
    shard = corpus.shard(d0, d1)
    beta0 = tm.dirichlet_rows(K, V, seed=7)
    log(f"[rank {rank}] corpus M={corpus.M} V={V} nnz={corpus.nnz} sum_counts={int(corpus.counts.sum())} "
        f"shard docs [{d0},{d1}) nnz={shard.nnz}  ({time.perf_counter() - t0:.1f}s to generate)")
    niter, ntol, viter, vtol = 1000, 1.0 / K ** 2, 10, 1.0 / K ** 2   # defaults of train! (src/LDA.jl:161)
    L = tm.lib()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the collective: RCCL inside the library, torch.distributed as the alternative
    collective = "none"
    comm_holder = {}

    def make_engine(Kx=None, beta0x=None):
        eng = HipLDAEngine(shard, K if Kx is None else Kx, beta0 if beta0x is None else beta0x, corpus.M, local_rank, distributed=(world > 1))
        return eng

    def attach_comm(eng):
        """Returns the all-reduce callable for this engine."""
        nonlocal collective
        tr = ShardedLDA(eng)
        if world == 1:
            return lambda: None, None
        use_lib = args.collective == "lib" and not one_gpu_debug
        comm = None
        if use_lib:
            ok = 1
            try:
                comm = tm.Communicator.torch_bootstrap(eng.ctx)
            except Exception as e:                               # agree on the fallback across ranks
                log(f"[rank {rank}] in-library RCCL communicator failed: {e}")
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if comm is not None:
                    comm.close()
                comm, use_lib = None, False
            if comm is not None:
                # the communicator's FIRST collective, checked before anything depends on it (RCCL with more than one rank has never run on the
                # development box): rank r contributes r + 1, every rank must read world (world + 1) / 2; a wrong sum or an error on ANY rank sends
                # all of them to torch.distributed's collective instead
                ok = 1
                try:
                    probe = torch.full((4096,), float(rank + 1), dtype=torch.float32, device=eng.device)
                    torch.cuda.synchronize()
                    comm.allreduce(probe.data_ptr(), probe.numel())
                    eng.synchronize()
                    if not bool((probe == float(world * (world + 1) // 2)).all().item()):
                        log(f"[rank {rank}] in-library RCCL all-reduce returned a wrong sum ({float(probe[0].item())})")
                        ok = 0
                except Exception as e:
                    log(f"[rank {rank}] in-library RCCL all-reduce failed: {e}")
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    comm.close()
                    comm, use_lib = None, False
        fused_form =os.environ.get("TMVB_FUSED_ALLREDUCE", "0") != "0"   # opt-in; DEFAULT = the three-call form, ONE collective per iteration (the library's train! default too)
        if one_gpu_debug and not use_lib:
            def gsum(a):
                t = torch.from_numpy(a); dist.all_reduce(t, op=dist.ReduceOp.SUM)
            comm = tm.Communicator.host(eng.ctx, world, rank, gsum)
            transport = "host transport + gloo (one-GPU plumbing check, not a measurement)"
        elif use_lib:
            transport = f"RCCL {tm.rccl_version()} inside libtmvb_hip.so"
        if comm is not None:
            eng.model.set_comm(comm, corpus.M)
            if fused_form:
                collective = f"{transport}: tmvb_lda_estep_allreduce (TMVB_FUSED_ALLREDUCE=1: Elogtheta_sum tail early on a side stream, statistics in {os.environ.get('TMVB_AR_SLICES', '1')} slab(s))"
                eng.fused_allreduce = True
                return (lambda: None), comm
            collective = f"{transport}: ONE all-reduce of K*V+K f32 per iteration on the context stream (tmvb_comm_allreduce)"
            ptr, n = eng.model.stats()
            return (lambda: comm.allreduce(ptr, n)), comm
        collective = "torch.distributed all_reduce (nccl = RCCL) on the bound statistics buffer"
        if "nccl" not in comm_holder:
            comm_holder["nccl"] = dist.new_group(backend="nccl")
        tr.group = comm_holder["nccl"]
        return tr.allreduce_stats, None

    def run_window(eng, allreduce, burnin, warmup, steps):
        fused = getattr(eng, "fused_allreduce", False)

        def one():
            if fused:
                eng.model.estep_allreduce(viter, vtol)
            else:
                eng.estep(viter, vtol); eng.reduce_docs(); allreduce()
            eng.update_beta(); eng.update_alpha(niter, ntol)
        for _ in range(burnin + warmup):
            one()
        # HIP events on the stream the kernels are launched on.  The library's (tmvb_event_create: hipEventDisableSystemFence, the flag HIP
        # documents for timing); torch.cuda.Event's default events perform a system-scope fence each -- two per step cost the timed
        # window 1.4 % (--torch-events: 1292 / 1295 / 1298 it/s with them, 1321 / 1302 / 1319 with no events at all, run r4bd)
        if args.torch_events:
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            rec = lambda e: e.record(eng.stream)
            ela = lambda a, b2: a.elapsed_time(b2)
        else:
            ev = [(eng.ctx.timing_event(), eng.ctx.timing_event()) for _ in range(steps)]
            rec = lambda e: e.record()
            ela = lambda a, b2: a.elapsed_ms(b2)
        barrier()
        t_start = time.perf_counter()
        for s in range(steps):
            if not args.no_step_events:
                rec(ev[s][0])
            if fused:                             # N > 1: the bracket then holds the E-step AND its (overlapped) collective
                eng.model.estep_allreduce(viter, vtol)
                if not args.no_step_events:
                    rec(ev[s][1])
            else:
                eng.estep(viter, vtol)
                if not args.no_step_events:
                    rec(ev[s][1])
                eng.reduce_docs()
                allreduce()
            eng.update_beta()
            eng.update_alpha(niter, ntol)
        barrier()
        elapsed = time.perf_counter() - t_start
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)          # gloo (CPU tensor)
            elapsed = float(t.item())
        return elapsed, ([ela(a, b2) for a, b2 in ev] if not args.no_step_events else [float('nan')] * steps)

    # ---- clocks and first touches (round-3 review: the driver's first window read 8 % below the same build's rate in a process
    # that had already run for a second, while its SECOND window matched): a scratch model of the same shard runs untimed
    # iterations until the device has been busy for --clock-warmup seconds, in chunks whose rates go to stderr (the ramp is
    # visible there) and into config.clock_warmup.  It shares nothing with the measured model but the device: the workload
    # definition (60 burn-in iterations from the cold start, then W + K) is unchanged.
    clock_log = []
    if args.clock_warmup > 0:
        scratch = HipLDAEngine(shard, K, beta0, corpus.M, local_rank, distributed=False)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < args.clock_warmup and len(clock_log) < 40:
            torch.cuda.synchronize(); t_c = time.perf_counter()
            for _ in range(50):
                scratch.estep(viter, vtol); scratch.reduce_docs(); scratch.update_beta(); scratch.update_alpha(niter, ntol)
            torch.cuda.synchronize()
            clock_log.append(round(50.0 / (time.perf_counter() - t_c), 1))
        scratch.model.close(); del scratch
        log(f"[rank {rank}] clock warm-up: {len(clock_log)} chunks of 50 iterations, it/s per chunk: {clock_log}")

    # ---- steady-state window (the headline)
    eng = make_engine()
    allreduce, comm = attach_comm(eng)
    elapsed, estep_ms = run_window(eng, allreduce, args.burnin, args.warmup, args.steps)
    sweep_hist = eng.model.sweep_hist(viter + 1).tolist()
    n_launch = eng.model.estep_launches()

    # ---- cold-start window (round 1's number), fresh state
    cold = None
    if not args.no_cold:
        eng_c = make_engine()
        ar_c, comm_c = attach_comm(eng_c)
        el_c, ms_c = run_window(eng_c, ar_c, 0, args.warmup, args.steps)
        cold = {"value": args.steps / el_c, "ms_per_step": 1e3 * el_c / args.steps, "estep_ms": float(np.mean(ms_c)),
                "window": f"iterations {args.warmup + 1}..{args.warmup + args.steps} from alpha=1, gamma=1, beta0",
                "sweep_hist_last_step": eng_c.model.sweep_hist(viter + 1).tolist()}
        eng_c.model.set_comm(None, shard.M) if comm_c is not None else None
        if comm_c is not None:
            comm_c.close()
        eng_c.model.close(); del eng_c

    # ---- multi-GPU self-validation (N > 1): the first run on real multi-GPU hardware has to prove itself.  A fresh sharded
    # model runs 5 checked iterations of the library's train! from the cold start; then (i) every rank hashes its replica of the
    # globals (alpha, beta -- the replicated M-step must leave them bit-identical everywhere) and the 64-bit hashes are
    # all-gathered and compared, (ii) rank 0 alone runs the same 5 iterations on the WHOLE corpus on its own GPU (N = 1) and the
    # two ELBO trajectories are compared: they differ by fp32 summation order only.
    mg_check = None
    if world > 1:
        n_chk = 5

        def sharded_check(Kc, beta0c):
            """5 checked iterations of the sharded train! at Kc topics: globals' hashes over the ranks, ELBO trajectory against N = 1 on rank 0"""
            eng_s = make_engine(Kc, beta0c)
            ar_s, comm_s = attach_comm(eng_s)
            if comm_s is None:
                eng_s.model.close()
                return {"skipped": "needs the in-library communicator (tmvb_*_set_comm); --collective torch has no sharded train!"}
            ntol_c = vtol_c = 1.0 / Kc ** 2
            buf = np.full(n_chk, np.nan); done, base = C.c_int32(0), C.c_double(0.0)
            tm._lib.check(L.tmvb_lda_train(eng_s.model.handle, C.c_int32(n_chk), C.c_double(0.0), C.c_int32(niter), C.c_double(ntol_c), C.c_int32(viter),
                                           C.c_double(vtol_c), C.c_int32(1), buf.ctypes.data_as(C.POINTER(C.c_double)), C.byref(done), C.byref(base)))
            traj_n = buf[:done.value].copy()
            eng_s.model.update_host()
            dig = hashlib.sha256(np.ascontiguousarray(eng_s.model.alpha).tobytes() + np.asfortranarray(eng_s.model.beta).tobytes(order="F")).digest()
            mine = torch.tensor([int.from_bytes(dig[:8], "little", signed=True)], dtype=torch.int64)
            allh = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(allh, mine)                                  # gloo
            hashes = [int(t.item()) for t in allh]
            shard_nnz = [int(corpus.doc_ptr[b] - corpus.doc_ptr[a]) for a, b in bounds]
            chk = {"K": Kc, "iterations": int(done.value), "globals_hash_equal": len(set(hashes)) == 1,
                   "form": "fused" if getattr(eng_s, "fused_allreduce", False) else "single", "shard_nnz": shard_nnz,
                   "shard_nnz_max_over_min": max(shard_nnz) / max(min(shard_nnz), 1),
                   "globals_hash_per_rank": [f"{h & 0xFFFFFFFFFFFFFFFF:016x}" for h in hashes]}
            if rank == 0:
                g1 = tm.gpuLDA(corpus, Kc, device_id=local_rank)
                g1.beta = np.asfortranarray(beta0c); g1.beta_old = g1.beta.copy(order="F")
                traj_1 = g1.train(iter=n_chk, tol=0.0, checkelbo=1, printelbo=False)
                g1.close()
                m = min(len(traj_1), len(traj_n))
                relv = float(np.max(np.abs(traj_n[:m] - traj_1[:m]) / np.abs(traj_1[:m]))) if m else float("nan")
                chk.update({"elbo_rel_vs_n1": relv, "elbo_rel_tolerance": 2e-6, "elbo_n": traj_n.tolist(), "elbo_n1": np.asarray(traj_1).tolist(),
                            "pass": bool(chk["globals_hash_equal"] and m == n_chk and relv <= 2e-6)})
                if not chk["pass"]:
                    log(f"MULTI-GPU SELF-CHECK FAILED (K = {Kc}): " + json.dumps(chk))
            eng_s.model.set_comm(None, shard.M); comm_s.close()
            eng_s.model.close(); del eng_s
            barrier()
            return chk

        mg_check = sharded_check(K, beta0)
        # BASELINE.json configs[2] is LDA K = 100 doc-sharded over the 8 GPUs: the headline run (K = 50) also proves THAT model through set_comm + the sharded
        # train! (round-5 review: K = 100 had never been sharded, even in emulation); the result rides in the same block as "config3_k100"
        if K == 50 and "skipped" not in mg_check:
            c3 = sharded_check(100, tm.dirichlet_rows(100, V, seed=7))
            mg_check["config3_k100"] = {k: c3.get(k) for k in ("K", "iterations", "globals_hash_equal", "elbo_rel_vs_n1", "elbo_rel_tolerance", "pass", "form")}
            if rank == 0 and "pass" in mg_check:
                mg_check["pass"] = bool(mg_check["pass"] and c3.get("pass"))

    result = None
    if rank == 0:
        ms = float(np.mean(estep_ms))
        b_e = estep_bytes(shard.nnz, shard.M, K)
        achieved = b_e / (ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(K, shard.M, shard.nnz) if world == 1 else (None, "collected on one GPU only")
        result = {
            "metric": f"VB iters/sec, LDA K={K} on NSF-shaped corpus (M={corpus.M}, V={V})",
            "value": args.steps / elapsed, "unit": "VB iters/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"LDA K={K}, SYN-NSF (synthetic NSF-shaped corpus, seed 20260928), train! defaults "
                                   f"viter=10 vtol=1/K^2 niter=1000 ntol=1/K^2 checkelbo=Inf; steady state: {args.burnin} untimed burn-in "
                                   f"iterations from the cold start (state preparation), then {args.warmup} warm-up + {args.steps} timed",
                       "K": K, "M": corpus.M, "V": V, "nnz": corpus.nnz, "sum_counts": int(corpus.counts.sum()), "burnin": args.burnin,
                       "clock_warmup": {"seconds": args.clock_warmup, "its_per_chunk_of_50": clock_log,
                                        "what": "untimed iterations of a scratch model on the same shard before the measured model is built (device clocks, first touches); not state preparation"},
                       "parallelism": f"doc-shard x{world}, all-reduce of {K * V + K} f32 per iteration" if world > 1 else "single GPU",
                       "collective": collective,
                       "sweep_hist_last_step": sweep_hist},
            "roofline": {"bound": "hbm", "kernel": "LDA E-step = the per-document grid-tile sweep kernels (lda_estep_grid_kernel<LPR, NP>, lda_estep_grid_long_kernel; lda_estep_reg_long_kernel / lda_estep_kernel for documents of more than 768 unique terms) over the document pieces + the gather-side statistics passes (termstats_recompute_kernel, termstats_multi_kernel) of every piece; one 'launch' = one E-step, timed start-to-end with HIP events on the context stream",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_source_hash": kernel_source_hash(),
                         "algorithmic_bytes_per_estep": b_e, "estep_ms": ms,
                         "estep_ms_includes": ("the statistics all-reduce, overlapped slab by slab with the last statistics pass (tmvb_lda_estep_allreduce is one "
                                               "asynchronous call: the events bracket E-step + collective, so frac understates the E-step alone at N > 1)")
                                              if getattr(eng, "fused_allreduce", False) else "the E-step only",
                         "estep_ms_min": float(np.min(estep_ms)), "estep_ms_max": float(np.max(estep_ms)),
                         "estep_ms_median": float(np.median(estep_ms)), "estep_ms_steps": [round(float(x), 4) for x in estep_ms],
                         "frac_at_median": b_e / (float(np.median(estep_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "valu_issue_frac": (getattr(pmc_traffic, "valu_issue_cycles", None) / (ms * 1e-3 * 2.4e9)) if getattr(pmc_traffic, "valu_issue_cycles", None) else None,
                         "valu_issue_frac_is": "VALU-issue cycles per SIMD of the E-step's kernels (4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs, rocprofv3 --pmc, kernels serialised) / (this run's estep_ms x 2.4 GHz): "
                                               "the share of the E-step during which an average SIMD issues a vector instruction -- the limit the byte-count frac saturates against (the byte model overstates what reaches HBM: traffic / algorithmic)",
                         "launches_per_estep": n_launch,
                         "whole_iteration_GBs": (b_e + mstep_bytes(K, V)) / (elapsed / args.steps) / 1e9},
            "cold_start": cold,
            "cpu_baseline": None,
            "parity": None,
        }
        if mg_check is not None:
            result["multi_gpu_check"] = mg_check

    # ---- time to ELBO plateau (second half of the BASELINE metric): fresh cold start, checkelbo=1, tol=1.0, the
    # library's own train! loop (sharded through the communicator when N>1), in chunks so that the wall clock of the
    # trajectory is known; the stop rule carries across chunks (every chunk re-evaluates its baseline ELBO)
    if not args.no_plateau:
        eng2 = make_engine()
        ar2, comm2 = attach_comm(eng2)
        if world > 1 and comm2 is None:
            log("elbo_plateau needs the in-library communicator; skipped")
            if rank == 0:
                result["elbo_plateau"] = None
        else:
            # train! in chunks so that the trajectory gets wall-clock stamps; every chunk is a train! call of its own and pays that call's baseline
            # evaluation (the token-walk form of update_elbo!, twice: tmvb_train.h compares like with like at the switch of forms) -- 250 iterations per
            # chunk keep that measurement overhead below 0.5 % (round 5; it was 50)
            chunk, cap = 250, args.plateau_cap
            h = eng2.model.handle
            traj_all, stamps, reached = [], [], False
            buf = np.full(chunk, np.nan)
            done, base = C.c_int32(0), C.c_double(0.0)
            barrier()
            t1 = time.perf_counter()
            t_prev = 0.0
            while len(traj_all) < cap:
                tm._lib.check(L.tmvb_lda_train(h, C.c_int32(chunk), C.c_double(1.0), C.c_int32(niter), C.c_double(ntol), C.c_int32(viter),
                                               C.c_double(vtol), C.c_int32(1), buf.ctypes.data_as(C.POINTER(C.c_double)), C.byref(done), C.byref(base)))
                now = time.perf_counter() - t1
                nd = done.value
                traj_all.extend(buf[:nd].tolist())
                stamps.extend([t_prev + (now - t_prev) * (i + 1) / max(nd, 1) for i in range(nd)])
                t_prev = now
                if nd < chunk:
                    reached = True
                    break
            barrier()
            t_plateau = time.perf_counter() - t1
            if rank == 0:
                n = len(traj_all)
                idx = sorted(set(list(range(0, n, max(1, n // 14))) + [n - 1]))
                last_delta = traj_all[-1] - (traj_all[-2] if n > 1 else base.value)
                result["elbo_plateau"] = {
                    "reached": reached, "seconds": t_plateau, "iterations": n, "iteration_cap": cap,
                    "stop_rule": "delta_elbo < tol=1.0, signed (check_elbo!, src/modelutils.jl:574-585), checkelbo=1",
                    "last_delta": last_delta, "elbo_first": traj_all[0], "elbo_last": traj_all[-1],
                    "seconds_per_checked_iteration": t_plateau / max(n, 1),
                    "elbo_form": ("decomposed" if eng2.model.elbo_form() == 1 else "token walk"),     # tmvb_lda_elbo_form of the last check
                    "elbo_vs_wallclock": [[round(stamps[i], 4), traj_all[i]] for i in idx],
                    "note": "wall clock includes one update_elbo! per iteration; timestamps are interpolated inside chunks of 250 iterations"}
        if comm2 is not None:
            eng2.model.set_comm(None, shard.M); comm2.close()
        eng2.model.close(); del eng2

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"], result["parity"] = cpu_baseline(tm, corpus, K, beta0)
            if not result["parity"]["pass"]:
                log("PARITY FAILED against the fp64 oracle at the timed size: " + json.dumps(result["parity"]["worst"]))
    if comm is not None:
        eng.model.set_comm(None, shard.M); comm.close(); comm = None
    eng.model.close(); del eng
    if rank == 0:
        # BASELINE.json configs 3-5 (N = 1 only; default sizes only): one measured line each, steady-state window, roofline and a
        # full-corpus OpenMP cpu_baseline -- tools/model_bench.py; the headline fields above stay config 2's
        if world == 1 and not args.no_other_configs and (args.K, args.docs, args.vocab) == (50, 128804, 25319):
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import model_bench
            others = {}
            # The GPU windows first, back to back (the clocks stay up), each in a FRESH process: the iteration time of a model depends
            # on how many hardware queues the process's streams, past and present, have been mapped onto (LDA K = 100: 835 it/s as the
            # first model of a process, 750 behind the models this process has already closed -- DESIGN.md section 4c), and these lines
            # report the configuration, not this process's history.  --other-configs-inline runs them here instead.
            import subprocess
            for name in ("lda100", "ctm", "ctpf"):
                t_c = time.perf_counter()
                try:
                    if args.other_configs_inline:
                        others[name] = model_bench.ALL[name](cpu=False)
                    else:
                        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "model_bench.py"), "--gpu-only", name], capture_output=True,
                                             text=True, timeout=900, cwd=ROOT)
                        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
                        if out.returncode != 0 or not lines:
                            raise RuntimeError(f"model_bench.py {name} failed (rc {out.returncode}): {out.stderr[-300:]}")
                        others[name] = json.loads(lines[-1])
                        others[name]["process"] = "fresh process (python tools/model_bench.py --gpu-only " + name + ")"
                except Exception as e:                       # a failing side line must not cost the headline
                    others[name] = {"error": f"{type(e).__name__}: {e}"}
                log(f"other_configs[{name}] GPU window done in {time.perf_counter() - t_c:.1f}s")
            for name in ("lda100", "ctm", "ctpf"):             # ... then their CPU baselines
                if args.no_cpu_baseline or "error" in others[name]:
                    continue
                t_c = time.perf_counter()
                try:
                    others[name]["cpu_baseline"] = model_bench.CPU[name]()
                    others[name]["parity"] = others[name]["cpu_baseline"].pop("parity", None)
                    if others[name]["parity"] and not others[name]["parity"]["pass"]:
                        log(f"PARITY FAILED for {name}: " + json.dumps(others[name]["parity"]["worst"]))
                except Exception as e:
                    others[name]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
                log(f"other_configs[{name}] cpu_baseline done in {time.perf_counter() - t_c:.1f}s")
            result["other_configs"] = others
        write_detail(result)
        print(slim_line(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
