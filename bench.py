#!/usr/bin/env python
"""
bench.py -- VB outer iterations per second of the LDA hot path on SYN-NSF (M=128804, V=25319), K=50
(BASELINE.json configs[1]), document-sharded over N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one outer iteration of train! (src/LDA.jl:169-183) with checkelbo=Inf: fused E-step over
this rank's documents, the Elogtheta_sum reduction, ONE all-reduce of the packed K*V+K statistics
(N>1), then update_beta! and update_alpha! on every rank.  The corpus and the state are resident in
HBM before the timed region; training starts cold (alpha=1, gamma=1, beta0 ~ Dirichlet(V,1) seed 7),
W warm-up iterations, then exactly K timed iterations between barrier+synchronize pairs; the time is
the MAX over ranks.  The corpus is fixed as N grows ("scaling": "strong").

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the E-step kernel against the HBM roofline (algorithmic bytes, HIP-event timing on
                 the launch stream), and
  cpu_baseline : the fp64 oracle (a port of the reference's CPU path -- the reference itself is Julia
                 and cannot run here) timed on this host's cores on a bounded document sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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


def pmc_traffic(K, M, nnz):
    """HBM traffic of one E-step from the committed rocprofv3 PMC summary (separate --pmc FETCH_SIZE /
    --pmc WRITE_SIZE passes of this same command, profiles/r1_lda_k50_pmc.*), corrected as
    MI355X_MICROARCH.md's HBM section prescribes for gfx950 (2 x FETCH_SIZE; KB units).  Only returned
    when the workload is the one the counters were collected on; otherwise null."""
    path = os.path.join(ROOT, "profiles", "r1_lda_k50_pmc.json")
    if not (K == 50 and M == 128804 and nnz == 10929864 and os.path.exists(path)):
        return None
    rows = json.load(open(path))
    tot = 0.0
    for name, r in rows.items():
        if "lda_estep" in name or "termstats" in name:
            if "fetch_kb_per_iteration" not in r:
                return None
            tot += (2.0 * r["fetch_kb_per_iteration"] + r["write_kb_per_iteration"]) * 1024.0
    return tot


def cpu_baseline(tm, corpus, K, beta0, warmup, budget_s=20.0):
    """fp64 oracle (oracle/*.c) on a bounded sample of the same workload, all host cores (OpenMP
    document-parallel E-step) and single thread; scaled to full-corpus iterations/s by nnz."""
    import numpy as np
    from oracle import oracle as oc
    oc.build()
    ncores = min(os.cpu_count() or 1, 64)      # threads actually used (private K x V statistics per thread)
    sample_docs = min(corpus.M, 16000)
    sh = corpus.shard(0, sample_docs)
    frac = sh.nnz / max(corpus.nnz, 1)

    def run(threads, iters):
        m = oc.LDA(oc.CSR(sh.doc_ptr, sh.terms, sh.counts, sh.V), K, beta0)
        for _ in range(warmup):
            m.estep(omp_threads=threads); m.update_beta(); m.update_alpha()
        t0 = time.perf_counter()
        done = 0
        for _ in range(iters):
            m.estep(omp_threads=threads); m.update_beta(); m.update_alpha()
            done += 1
            if time.perf_counter() - t0 > budget_s / 2:
                break
        return done / (time.perf_counter() - t0)

    run(ncores, 1)                             # OpenMP runtime start-up outside the timing
    omp = run(ncores, 8)
    one = run(1, 2) if ncores > 1 else omp
    return {
        "value": omp * frac, "unit": "VB iters/sec", "cores": ncores, "kind": "port",
        "single_thread_value": one * frac,
        "sample": f"fp64 C oracle (port of src/LDA.jl train!), first {sample_docs} docs of the workload "
                  f"({sh.nnz} of {corpus.nnz} nnz), {warmup} warm-up + timed iterations from the same cold start, "
                  f"OpenMP E-step on {ncores} threads; value = sample iters/s x nnz fraction {frac:.4f}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--K", type=int, default=50)
    ap.add_argument("--docs", type=int, default=128804)
    ap.add_argument("--vocab", type=int, default=25319)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plateau", action="store_true")
    args = ap.parse_args()

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
    backend = os.environ.get("TMVB_DIST_BACKEND", "nccl")          # "gloo" only to smoke-test the N>1 plumbing on one GPU
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    K, V = args.K, args.vocab
    t0 = time.perf_counter()
    corpus = tm.syn_nsf(M=args.docs, V=V)                 # identical bytes on every rank (seeded)
    bounds = corpus.shard_bounds(world)
    d0, d1 = bounds[rank]
    shard = corpus.shard(d0, d1)
    beta0 = tm.dirichlet_rows(K, V, seed=7)
    log(f"[rank {rank}] corpus M={corpus.M} V={V} nnz={corpus.nnz} sum_counts={int(corpus.counts.sum())} "
        f"shard docs [{d0},{d1}) nnz={shard.nnz}  ({time.perf_counter() - t0:.1f}s to generate)")

    eng = HipLDAEngine(shard, K, beta0, corpus.M, local_rank, distributed=(world > 1))
    tr = ShardedLDA(eng)
    niter, ntol, viter, vtol = 1000, 1.0 / K ** 2, 10, 1.0 / K ** 2   # defaults of train! (src/LDA.jl:161)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.iterate(niter, ntol, viter, vtol)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t_start = time.perf_counter()
    for s in range(args.steps):
        e = eng
        ev[s][0].record(eng.stream)           # HIP events on the stream the kernels are launched on
        e.estep(viter, vtol)
        ev[s][1].record(eng.stream)
        e.reduce_docs()
        tr.allreduce_stats()
        e.update_beta()
        e.update_alpha(niter, ntol)
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    estep_ms = [a.elapsed_time(b) for a, b in ev]
    sweep_hist = eng.model.sweep_hist(viter + 1).tolist()
    n_launch = eng.model.estep_launches()

    result = None
    if rank == 0:
        ms = float(np.mean(estep_ms))
        b_e = estep_bytes(shard.nnz, shard.M, K)
        achieved = b_e / (ms * 1e-3) / 1e9
        result = {
            "metric": f"VB iters/sec, LDA K={K} on NSF-shaped corpus (M={corpus.M}, V={V})",
            "value": args.steps / elapsed, "unit": "VB iters/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"LDA K={K}, SYN-NSF (synthetic NSF-shaped corpus, seed 20260928), train! defaults "
                                   "viter=10 vtol=1/K^2 niter=1000 ntol=1/K^2 checkelbo=Inf, cold start",
                       "K": K, "M": corpus.M, "V": V, "nnz": corpus.nnz, "sum_counts": int(corpus.counts.sum()),
                       "parallelism": f"doc-shard x{world}, 1 all-reduce of {K * V + K} f32 per iteration" if world > 1 else "single GPU",
                       "sweep_hist_last_step": sweep_hist},
            "roofline": {"bound": "hbm", "kernel": "LDA E-step = lda_estep_reg_kernel<13,T> / lda_estep_kernel over 4 document pieces on one stream, termstats_recompute_kernel<13> + termstats_multi_kernel of piece p on the context stream under the document kernels of piece p+1; timed start-to-end with events on the context stream",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(K, shard.M, shard.nnz) if world == 1 else None,
                         "traffic_source": "profiles/r1_lda_k50_pmc.txt (rocprofv3 --pmc, separate passes; 2*FETCH_SIZE+WRITE_SIZE summed over all dispatches of the E-step's kernels in one iteration)",
                         "algorithmic_bytes_per_estep": b_e, "estep_ms": ms,
                         "estep_ms_min": float(np.min(estep_ms)), "estep_ms_max": float(np.max(estep_ms)),
                         "launches_per_estep": n_launch,
                         "whole_iteration_GBs": (b_e + mstep_bytes(K, V)) / (elapsed / args.steps) / 1e9},
            "cpu_baseline": None,
        }
    # time to ELBO plateau (second half of the BASELINE metric): fresh cold start, checkelbo=1, tol=1.0
    if not args.no_plateau:
        eng2 = HipLDAEngine(shard, K, beta0, corpus.M, local_rank, distributed=(world > 1))
        tr2 = ShardedLDA(eng2)
        barrier()
        t1 = time.perf_counter()
        stamps = []
        traj = tr2.train(iter=150, tol=1.0, checkelbo=1, K=K, on_iter=lambda k, e: stamps.append(time.perf_counter() - t1))
        barrier()
        t_plateau = time.perf_counter() - t1
        if rank == 0:
            result["elbo_plateau"] = {"seconds": t_plateau, "iterations": len(traj), "stop_rule": "delta_elbo < tol=1.0 (src/modelutils.jl:580)",
                                      "elbo_first": traj[0], "elbo_last": traj[-1],
                                      "elbo_vs_wallclock": [[round(stamps[i], 4), traj[i]] for i in sorted(set(list(range(0, len(traj), max(1, len(traj) // 12))) + [len(traj) - 1]))]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(tm, corpus, K, beta0, args.warmup)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
