// tmvb_train.h -- the outer loop of train! (src/LDA.jl:161-187, src/CTM.jl:185-213, src/CTPF.jl:344-376) with
// check_elbo! (src/modelutils.jl:574-585), written once for the three models and for any number of local handles.
//
// One handle without a communicator is the reference's single-device train!.  With communicators the same loop is the
// document-sharded run of SURVEY.md section 8e: E-step on every local shard, ONE sum-all-reduce of the packed statistics
// per outer iteration (RCCL over xGMI, or the host transport), the identical M-step on every shard, and -- when the ELBO
// is checked -- one 8-byte all-reduce of the per-document part so that every rank takes the same stop decision.
// n > 1 local handles = one host thread driving n GPUs (tmvb_comm_create_rccl_all); the n collectives of a step sit in
// one RCCL group.
#pragma once
#include "tmvb_internal.h"

#include <type_traits>
#include <utility>

// Ops (per model):
//   int      estep(H*)                      E-step sweeps + per-document statistics, asynchronous
//   int      reduce(H*)                     per-document sums into the statistics tail
//   int      before_allreduce(H*)           join side streams so that the context stream owns the whole buffer
//   float*   stats(H*), int64_t stats_len(H*)
//   int      mstep(H*)                      the M-step, identical on every rank
//   int      elbo_local(H*, double* summable, double* once)   synchronous; `summable` adds up over shards, `once` is global
//   double*  elbo_dev(H*)                   one device double usable as all-reduce scratch
//   tmvb_comm* comm(H*), bool distributed(H*), tmvb_ctx* ctx(H*), int64_t nnz(H*), void set_elbo(H*, double), double get_elbo(H*)
//   int      finish(H*)                     join + synchronize

template <class H, class Ops>
static int tmvb_group_sum_f64(H* const* hs, int n, Ops& ops, double local_sum, double* out)
{
    if (!ops.comm(hs[0])) { *out = local_sum; return TMVB_OK; }
    std::vector<tmvb_comm*> comms(n);
    std::vector<void*> ptrs(n);
    std::vector<int64_t> counts(n, 1);
    std::vector<double> vals(n, 0.0);
    vals[0] = local_sum;                    // hs[0] carries this process' sum, the other local handles carry zero
    for (int i = 0; i < n; ++i) {
        tmvb_ctx* ctx = ops.ctx(hs[i]);
        TMVB_HIP(hipSetDevice(ctx->device));
        TMVB_HIP(hipMemcpyAsync(ops.elbo_dev(hs[i]), &vals[i], sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        TMVB_HIP(hipStreamSynchronize(ctx->stream));
        comms[i] = ops.comm(hs[i]); ptrs[i] = ops.elbo_dev(hs[i]);
    }
    int rc = tmvb_comm_allreduce_group(comms.data(), ptrs.data(), counts.data(), n, TMVB_F64);
    if (rc) return rc;
    tmvb_ctx* c0 = ops.ctx(hs[0]);
    TMVB_HIP(hipSetDevice(c0->device));
    double v = 0.0;
    TMVB_HIP(hipMemcpyAsync(&v, ops.elbo_dev(hs[0]), sizeof(double), hipMemcpyDeviceToHost, c0->stream));
    TMVB_HIP(hipStreamSynchronize(c0->stream));
    for (int i = 1; i < n; ++i) {           // the other local streams must also have finished their share of the group
        TMVB_HIP(hipSetDevice(ops.ctx(hs[i])->device));
        TMVB_HIP(hipStreamSynchronize(ops.ctx(hs[i])->stream));
    }
    *out = v;
    return TMVB_OK;
}

// Ops may also offer  int elbo_enqueue(H*, double* once): update_elbo! enqueued on the context's stream, the summable part left in
// elbo_dev(h), no host synchronisation.  A sharded check then costs one collective on the stream and ONE synchronisation (the
// 8-byte read-back) instead of three (read-back of the local value, upload of it, read-back of the sum).
template <class Ops, class H, class = void> struct tmvb_has_elbo_enqueue : std::false_type {};
template <class Ops, class H>
struct tmvb_has_elbo_enqueue<Ops, H, std::void_t<decltype(std::declval<Ops&>().elbo_enqueue((H*)nullptr, (double*)nullptr))>> : std::true_type {};

// Ops may offer  int estep_allreduce(H*): estep + reduce + the statistics all-reduce in one call that overlaps the collective with the
// last statistics pass (LDA).  Used when this process drives ONE handle (one process per GPU); several local handles keep the
// grouped collective below.
template <class Ops, class H, class = void> struct tmvb_has_estep_allreduce : std::false_type {};
template <class Ops, class H>
struct tmvb_has_estep_allreduce<Ops, H, std::void_t<decltype(std::declval<Ops&>().estep_allreduce((H*)nullptr))>> : std::true_type {};

// Ops may offer  void will_check(H*, bool): called before every iteration's E-step with whether check_elbo! will evaluate the ELBO behind it, so that the
// iteration can leave update_elbo!'s per-token parts behind on its way instead of walking the corpus a second time; with it come
// int elbo_form(H*) (1: the last evaluation took the decomposed form) and void force_walk(H*, bool) (evaluate by the token walk whatever is available).
template <class Ops, class H, class = void> struct tmvb_has_will_check : std::false_type {};
template <class Ops, class H>
struct tmvb_has_will_check<Ops, H, std::void_t<decltype(std::declval<Ops&>().will_check((H*)nullptr, true))>> : std::true_type {};

// Ops may offer  bool graph_ok(H*): an UNCHECKED iteration of this handle (estep + reduce + mstep, one context, no communicator) enqueues the same launches with the
// same arguments every time and leaves the handle's host-side flags as it found them, so it may be captured once into a hipGraph and replayed (round 6: CTPF -- seven
// kernels and two cross-queue joins in ~130 us, ~20 us of it launch gaps and event hops).  MEASURED on MI355X / ROCm 7 (profiles/r6_ctpf_graph.txt): the replayed
// iteration is 0.31 - 0.33 ms against 0.130 ms enqueued the ordinary way -- hipGraphLaunch of this ten-node, three-stream graph costs the host more than the launches it
// replaces, and the device then waits for the host.  The replay is therefore OPT-IN (TMVB_TRAIN_GRAPH=1) and bit-identical (tests/test_ctpf_gpu.py).
template <class Ops, class H, class = void> struct tmvb_has_graph_ok : std::false_type {};
template <class Ops, class H>
struct tmvb_has_graph_ok<Ops, H, std::void_t<decltype(std::declval<Ops&>().graph_ok((H*)nullptr))>> : std::true_type {};

struct tmvb_iter_graph {
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    int plain_run = 0;            // unchecked iterations enqueued the ordinary way since the last checked one (lazy allocations and flags settle in the first two)
    bool failed = false;
    ~tmvb_iter_graph() { if (exec) (void)hipGraphExecDestroy(exec); if (graph) (void)hipGraphDestroy(graph); }
};

template <class H, class Ops>
static int tmvb_group_elbo(H* const* hs, int n, Ops& ops, double* out)
{
    if constexpr (tmvb_has_elbo_enqueue<Ops, H>::value) {
        if (ops.comm(hs[0])) {
            double once = 0.0;
            std::vector<tmvb_comm*> comms(n);
            std::vector<void*> ptrs(n);
            std::vector<int64_t> counts(n, 1);
            for (int i = 0; i < n; ++i) {
                double g = 0.0;
                int rc = ops.elbo_enqueue(hs[i], &g);
                if (rc) return rc;
                once = g;
                comms[i] = ops.comm(hs[i]); ptrs[i] = ops.elbo_dev(hs[i]);
            }
            int rc = tmvb_comm_allreduce_group(comms.data(), ptrs.data(), counts.data(), n, TMVB_F64);   // every handle carries its own share
            if (rc) return rc;
            tmvb_ctx* c0 = ops.ctx(hs[0]);
            TMVB_HIP(hipSetDevice(c0->device));
            double v = 0.0;
            TMVB_HIP(hipMemcpyAsync(&v, ops.elbo_dev(hs[0]), sizeof(double), hipMemcpyDeviceToHost, c0->stream));
            TMVB_HIP(hipStreamSynchronize(c0->stream));
            for (int i = 1; i < n; ++i) {
                TMVB_HIP(hipSetDevice(ops.ctx(hs[i])->device));
                TMVB_HIP(hipStreamSynchronize(ops.ctx(hs[i])->stream));
            }
            *out = v + once;
            for (int i = 0; i < n; ++i) ops.set_elbo(hs[i], *out);
            return TMVB_OK;
        }
    }
    double sum = 0.0, once = 0.0;
    for (int i = 0; i < n; ++i) {
        double s = 0.0, g = 0.0;
        int rc = ops.elbo_local(hs[i], &s, &g);
        if (rc) return rc;
        sum += s; once = g;
    }
    double tot = 0.0;
    int rc = tmvb_group_sum_f64(hs, n, ops, sum, &tot);
    if (rc) return rc;
    *out = tot + once;
    for (int i = 0; i < n; ++i) ops.set_elbo(hs[i], *out);
    return TMVB_OK;
}

template <class H, class Ops>
static int tmvb_train_group_loop(const char* who, H* const* hs, int n, int iter, double tol, int checkelbo, double* elbo_traj,
                                 int32_t* iters_done, double* elbo_baseline, Ops& ops)
{
    TMVB_REQUIRE(hs != nullptr && n >= 1, TMVB_EINVAL, "%s: no handles", who);
    for (int i = 0; i < n; ++i) TMVB_REQUIRE(hs[i] != nullptr, TMVB_EINVAL, "%s: handle %d is NULL", who, i);
    if (iters_done) *iters_done = 0;
    const bool sharded = ops.comm(hs[0]) != nullptr;
    for (int i = 0; i < n; ++i) {
        TMVB_REQUIRE((ops.comm(hs[i]) != nullptr) == sharded, TMVB_EINVAL, "%s: some handles have a communicator and some do not", who);
        TMVB_REQUIRE(sharded || !ops.distributed(hs[i]), TMVB_EINVAL,
                     "%s: a document-sharded handle needs a communicator (tmvb_*_set_comm), or the host composes estep/reduce_docs/update_* itself", who);
    }
    TMVB_REQUIRE(sharded || n == 1, TMVB_EINVAL, "%s: several local handles need communicators (tmvb_comm_create_rccl_all)", who);
    int rc;
    {   // iter = 0 when every document of the WHOLE corpus is empty (src/LDA.jl:166)
        double local = 0.0, total = 0.0;
        for (int i = 0; i < n; ++i) local += (double)ops.nnz(hs[i]);
        if ((rc = tmvb_group_sum_f64(hs, n, ops, local, &total))) return rc;
        if (total == 0.0) iter = 0;
    }
    double e_old = ops.get_elbo(hs[0]);
    if (checkelbo > 0 && checkelbo <= iter) {                                   // src/LDA.jl:167
        // The baseline is ALWAYS the token walk (round-5 advice): a second tmvb_*_train on a handle with no set_state in between still has the last
        // checked iteration's parts valid, and would otherwise take the decomposed form here while `old_parts` below says it did not.
        if constexpr (tmvb_has_will_check<Ops, H>::value) for (int i = 0; i < n; ++i) ops.force_walk(hs[i], true);
        rc = tmvb_group_elbo(hs, n, ops, &e_old);
        if constexpr (tmvb_has_will_check<Ops, H>::value) for (int i = 0; i < n; ++i) ops.force_walk(hs[i], false, false);
        if (rc) return rc;
    }
    if (elbo_baseline) *elbo_baseline = e_old;
    std::vector<tmvb_comm*> comms(n);
    std::vector<void*> ptrs(n);
    std::vector<int64_t> counts(n);
    int done = 0;
    tmvb_iter_graph ig;
    bool old_parts = false;                  // e_old was evaluated by the decomposed form of update_elbo! (never true for the baseline: a state this call did not produce)
    for (int k = 1; k <= iter; ++k) {
        ++done;
        if constexpr (tmvb_has_will_check<Ops, H>::value)
            for (int i = 0; i < n; ++i) ops.will_check(hs[i], checkelbo > 0 && (k % checkelbo) == 0);
        bool fused = false;
        if constexpr (tmvb_has_estep_allreduce<Ops, H>::value) {
            // DEFAULT: the three-call form -- estep, reduce_docs, ONE collective of the whole K*V+K buffer on the context's stream (north_star's
            // "single RCCL all-reduce ... per outer iteration").  TMVB_FUSED_ALLREDUCE=1 opts into tmvb_lda_estep_allreduce (the Elogtheta_sum tail
            // all-reduced early on a side stream, the statistics in slabs): two streams of collectives on one communicator, which no multi-rank
            // RCCL run has validated yet (round-4 advice) -- opt-in until tests/test_multigpu_rccl.py has passed on a multi-GPU node.
            static const bool fuse = [] { const char* e = getenv("TMVB_FUSED_ALLREDUCE"); return e && atoi(e) != 0; }();
            if (sharded && n == 1 && fuse) { if ((rc = ops.estep_allreduce(hs[0]))) return rc; fused = true; }
        }
        if constexpr (tmvb_has_graph_ok<Ops, H>::value) {
            // an unchecked iteration of one unsharded handle: captured into a hipGraph at its third occurrence in a row (enough of them left to pay for the
            // capture), replayed from then on; a checked iteration in between runs the ordinary way and the replay resumes behind the next two plain ones
            static const bool graph_env = [] { const char* e = getenv("TMVB_TRAIN_GRAPH"); return e && atoi(e) != 0; }();     // OPT-IN: measured slower (below)
            const bool checked = checkelbo > 0 && (k % checkelbo) == 0;
            if (checked) ig.plain_run = 0;
            if (graph_env && !checked && !sharded && n == 1 && !ig.failed && ops.graph_ok(hs[0])) {
                tmvb_ctx* c0 = ops.ctx(hs[0]);
                TMVB_HIP(hipSetDevice(c0->device));
                if (ig.exec && ig.plain_run >= 2) {
                    TMVB_HIP(hipGraphLaunch(ig.exec, c0->stream));
                    if (elbo_traj) elbo_traj[k - 1] = NAN;
                    continue;
                }
                if (!ig.exec && ig.plain_run >= 2 && iter - k >= 16) {
                    hipError_t ce = hipStreamBeginCapture(c0->stream, hipStreamCaptureModeThreadLocal);
                    if (ce == hipSuccess) {
                        int crc = ops.estep(hs[0]);
                        if (!crc) crc = ops.reduce(hs[0]);
                        if (!crc) crc = ops.mstep(hs[0]);
                        hipGraph_t g = nullptr;
                        ce = hipStreamEndCapture(c0->stream, &g);
                        if (!crc && ce == hipSuccess && g && hipGraphInstantiate(&ig.exec, g, nullptr, nullptr, 0) == hipSuccess) {
                            ig.graph = g;
                            TMVB_HIP(hipGraphLaunch(ig.exec, c0->stream));
                            if (elbo_traj) elbo_traj[k - 1] = NAN;
                            continue;
                        }
                        if (g) (void)hipGraphDestroy(g);
                        ig.exec = nullptr;
                    }
                    (void)hipGetLastError();
                    ig.failed = true;                 // nothing of the attempt has executed: this iteration runs the ordinary way below
                }
                ++ig.plain_run;
            }
        }
        if (!fused) for (int i = 0; i < n; ++i) if ((rc = ops.estep(hs[i]))) return rc;
        if (!fused) for (int i = 0; i < n; ++i) if ((rc = ops.reduce(hs[i]))) return rc;
        if (sharded && !fused) {
            for (int i = 0; i < n; ++i) {
                if ((rc = ops.before_allreduce(hs[i]))) return rc;
                comms[i] = ops.comm(hs[i]); ptrs[i] = ops.stats(hs[i]); counts[i] = ops.stats_len(hs[i]);
            }
            if ((rc = tmvb_comm_allreduce_group(comms.data(), ptrs.data(), counts.data(), n, TMVB_F32))) return rc;
        }
        for (int i = 0; i < n; ++i) if ((rc = ops.mstep(hs[i]))) return rc;
        if (elbo_traj) elbo_traj[k - 1] = NAN;
        if (checkelbo > 0 && (k % checkelbo) == 0) {                            // check_elbo! src/modelutils.jl:574-585
            double e_new;
            if ((rc = tmvb_group_elbo(hs, n, ops, &e_new))) return rc;
            TMVB_REQUIRE(std::isfinite(e_new), TMVB_ENONFINITE, "elbo must be finite.");
            double e_cmp = e_new;
            if constexpr (tmvb_has_will_check<Ops, H>::value) {
                // The two forms of update_elbo! evaluate the same sum with different fp32 roundings (5e-8 relative, i.e. ~5 at an ELBO of 1e8) and the stop
                // rule compares a DIFFERENCE with tol = 1: e_old from the token walk (the baseline of this call, a state set by the host) and e_new from the
                // decomposed form must not meet in one delta.  At the one switch of a call the new state is evaluated both ways: the walk's value against
                // e_old, the decomposed one kept for the next delta (and reported).
                const bool new_parts = ops.elbo_form(hs[0]) == 1;
                if (new_parts && !old_parts) {
                    for (int i = 0; i < n; ++i) ops.force_walk(hs[i], true);
                    rc = tmvb_group_elbo(hs, n, ops, &e_cmp);
                    for (int i = 0; i < n; ++i) { ops.force_walk(hs[i], false); ops.set_elbo(hs[i], e_new); }
                    if (rc) return rc;
                    TMVB_REQUIRE(std::isfinite(e_cmp), TMVB_ENONFINITE, "elbo must be finite.");
                } else if (!new_parts && old_parts) {
                    e_old = NAN;                                    // (cannot happen inside one call: every checked iteration collects; keep the rule visible)
                }
                old_parts = new_parts;
            }
            if (elbo_traj) elbo_traj[k - 1] = e_new;
            const double delta = std::isnan(e_old) ? INFINITY : e_cmp - e_old;
            e_old = e_new;
            if (delta < tol) break;                                             // signed, quirk Q4
        }
    }
    if constexpr (tmvb_has_will_check<Ops, H>::value) for (int i = 0; i < n; ++i) ops.will_check(hs[i], false);
    for (int i = 0; i < n; ++i) if ((rc = ops.finish(hs[i]))) return rc;
    if (iters_done) *iters_done = done;
    return TMVB_OK;
}
