/*
 * tmvb.h -- C ABI of libtmvb_hip.so, the MI355X (gfx950) variational-inference engine that
 * replaces TopicModelsVB.jl's OpenCL backend (gpuLDA / gpuCTM / gpuCTPF) behind the package's
 * `TopicModel / train! / @gpu` surface.
 *
 * Every entry point cites the reference interface it replaces (file:line relative to the
 * reference repository root).  The reference drives its device through one Julia function per
 * kernel pair (update_phi!, update_gamma!, ...); this engine fuses the per-document sweep into one
 * kernel, so the per-document operator triple maps onto a single `*_estep` call that follows the
 * CPU path's per-document semantics (src/LDA.jl:170-180), which is the parity target -- not the
 * OpenCL path's global-median rule (src/gpuLDA.jl:361).
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (streams and device pointers travel
 *     as void*).
 *   - all host matrices are column-major K x (.) Float64, exactly the memory of the reference's
 *     Matrix{Float64} (beta) or of hcat(model.gamma...) (per-document vectors).
 *   - term / reader ids are 0-based (the reference subtracts 1 on upload, src/modelutils.jl:371).
 *   - every call returns a status code; tmvb_last_error() gives the message of the last failure on
 *     the calling thread.  Calls are synchronous on return unless stated otherwise.
 *   - the library copies host data in/out during the call and never retains host pointers.
 *   - quirk Q1: the reference overwrites (does not accumulate) duplicate term ids inside one
 *     document (src/LDA.jl:131); this engine accumulates.  Corpora must be condensed
 *     (src/Corpus.jl:523) for parity; tmvb_corpus_create reports duplicates via
 *     tmvb_corpus_info.
 */
#ifndef TMVB_H
#define TMVB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMVB_ABI_VERSION 2

/* status codes (reference: ArgumentError src/gpuLDA.jl:349-351, TopicModelError
 * src/modelutils.jl:1-5, CorpusError src/Corpus.jl:85-89) */
#define TMVB_OK          0
#define TMVB_EINVAL      1   /* bad argument            -> ArgumentError   */
#define TMVB_ESHAPE      2   /* inconsistent sizes      -> TopicModelError */
#define TMVB_ECORPUS     3   /* corpus failed check_corp -> CorpusError    */
#define TMVB_ENOMEM      4
#define TMVB_EHIP        5   /* HIP runtime failure */
#define TMVB_ENONFINITE  6   /* non-finite statistic / state */
#define TMVB_ENODEVICE   7   /* no usable gfx950 device */
#define TMVB_ERCCL       8   /* RCCL / host-collective failure (document-sharded runs) */

typedef struct tmvb_ctx    tmvb_ctx;
typedef struct tmvb_corpus tmvb_corpus;
typedef struct tmvb_lda    tmvb_lda;
typedef struct tmvb_ctm    tmvb_ctm;
typedef struct tmvb_ctpf   tmvb_ctpf;
typedef struct tmvb_comm   tmvb_comm;
typedef struct tmvb_flda   tmvb_flda;
typedef struct tmvb_fctm   tmvb_fctm;

int         tmvb_abi_version(void);
const char* tmvb_last_error(void);

/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int tmvb_device_count(void);

/* ---- context: replaces cl.create_compute_context() (src/gpuLDA.jl:64) ----
 * One context = one GPU + one in-order stream.  `hip_stream` may be NULL (the library creates its
 * own stream) or an existing hipStream_t owned by the caller (e.g. the host framework's current
 * stream, so that its collectives are ordered with the engine's kernels). */
int tmvb_ctx_create(int32_t device_id, void* hip_stream, tmvb_ctx** out);
int tmvb_ctx_destroy(tmvb_ctx* ctx);
int tmvb_ctx_synchronize(tmvb_ctx* ctx);

/* Timing events on the context's stream for a host that measures the library's asynchronous calls (bench.py's HIP-event bracket around
 * the E-step): HIP events WITH time stamps and WITHOUT the system-scope fence a default event performs when it completes
 * (hipEventDisableSystemFence -- the flag HIP documents for timing; two default events per 0.77 ms LDA iteration cost it 1.4 %).
 * tmvb_event_elapsed_ms waits for `stop`.  Events belong to the device of the context that created them. */
typedef struct tmvb_event tmvb_event;
int tmvb_event_create(tmvb_ctx* ctx, tmvb_event** out);
int tmvb_event_record(tmvb_event* ev);                                  /* on the creating context's stream */
int tmvb_event_elapsed_ms(tmvb_event* start, tmvb_event* stop, float* ms);
int tmvb_event_destroy(tmvb_event* ev);
/* Diagnostics: evaluate the engine's fp32 device special functions on host data (the accuracy pins of the tests).
 * which: 0 = digamma (x > 0; src/utils.jl:21-53's algorithm), 1 = exp as used for exp(Elogtheta) (x <= 0 in exact
 * arithmetic), 2 = the rcp-based reciprocal 1/x. */
int tmvb_special_f32(tmvb_ctx* ctx, int32_t which, const float* x, float* y, int64_t n);
/* model.topics behind every train! of the reference: topics[i] = reverse(sortperm(vec(beta[i,:]))) (src/gpuLDA.jl:374, src/gpuCTM.jl and
 * src/gpuCTPF.jl alike; CTPF passes alef ./ bet).  beta: the model's K x V field, column-major, fp64, on the HOST; topics: K rows of V column
 * ids, row-major, 1-based like the reference's, on the host.  One stable segmented sort on the device, read backwards: ties come out in the
 * order sortperm + reverse gives them.  K * V < 2^31. */
int tmvb_topic_order(tmvb_ctx* ctx, const double* beta, int32_t K, int64_t V, int32_t* topics);


/* ---- communicator: document-sharded multi-GPU behind the C ABI (new: the reference is single-device, src/gpuLDA.jl:64;
 * the closest precedent is the v0.6 batch accumulation `newbeta +=`, v0.6/src/gpuLDA.jl:200-225) ----
 * Documents are conditionally independent given the globals, so every GPU runs the fused E-step on its own contiguous
 * document shard and the ONLY exchange per outer iteration is one all-reduce (sum, f32) of the packed sufficient
 * statistics (tmvb_*_stats); every rank then runs the identical deterministic M-step.  A communicator carries that
 * all-reduce.  Three ways to make one:
 *   tmvb_comm_create_rccl      one process per GPU: rank 0 calls tmvb_comm_unique_id and hands the 128 bytes to the other
 *                              ranks by any host channel (MPI.jl / Distributed.jl / a file / torch.distributed); every
 *                              rank then calls this (ncclCommInitRank) -- RCCL over xGMI.
 *   tmvb_comm_create_rccl_all  one process, one host thread, n GPUs (ncclCommInitAll); the n communicators are used
 *                              together through the *_train_group entry points.
 *   tmvb_comm_create_host      a host-supplied all-reduce on HOST memory (MPI_Allreduce, gloo, ...): the library stages
 *                              the buffer through pinned memory.  For hosts without RCCL connectivity and for tests.
 * All collectives are enqueued on the context's stream. */
#define TMVB_UNIQUE_ID_BYTES 128
#define TMVB_F32 0
#define TMVB_F64 1
int tmvb_comm_unique_id(void* id_out /* TMVB_UNIQUE_ID_BYTES */);
int tmvb_comm_create_rccl(tmvb_ctx* ctx, const void* unique_id, int32_t nranks, int32_t rank, tmvb_comm** out);
int tmvb_comm_create_rccl_all(tmvb_ctx* const* ctxs, int32_t n, tmvb_comm** out /* [n] */);
/* sum-all-reduce `count` elements of `dtype` (TMVB_F32 / TMVB_F64) in place in host_buf across the ranks; 0 = success */
typedef int (*tmvb_host_allreduce_fn)(void* user, void* host_buf, int64_t count, int32_t dtype);
int tmvb_comm_create_host(tmvb_ctx* ctx, int32_t nranks, int32_t rank, tmvb_host_allreduce_fn fn, void* user, tmvb_comm** out);
int tmvb_comm_destroy(tmvb_comm* comm);
/* backend: 0 = RCCL, 1 = host callback */
int tmvb_comm_info(const tmvb_comm* comm, int32_t* nranks, int32_t* rank, int32_t* backend);
/* In-place sum-all-reduce of device memory on the communicator's context stream (asynchronous for RCCL). */
int tmvb_comm_allreduce(tmvb_comm* comm, void* dev_ptr, int64_t count, int32_t dtype);
/* RCCL version code linked into the library (ncclGetVersion), 0 when it cannot be queried. */
int tmvb_rccl_version(void);

/* ---- corpus upload: the corpus half of update_buffer! (src/modelutils.jl:370-388, :438-472) ----
 * doc_ptr[M+1], terms[nnz], counts[nnz]; rdr_ptr/readers/ratings may be NULL when U == 0.
 * Validates the check_doc / check_corp rules (src/Corpus.jl:41-50, :111-122): ids in range,
 * counts and ratings positive, offsets monotone. */
int tmvb_corpus_create(tmvb_ctx* ctx, int64_t M, int64_t V, int64_t U,
                       const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                       const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                       tmvb_corpus** out);
int tmvb_corpus_destroy(tmvb_corpus* corp);

typedef struct {
    int64_t M, V, U, nnz, nR;
    int64_t sum_counts, sum_ratings;
    int64_t max_doc_len, max_readers;
    int64_t n_empty_docs;
    int64_t n_docs_with_duplicate_terms;   /* quirk Q1 */
    int64_t n_docs_with_duplicate_readers;
} tmvb_corpus_info_t;
int tmvb_corpus_info(const tmvb_corpus* corp, tmvb_corpus_info_t* out);

/* ---- docfile ingest: the document part of readcorp (src/Corpus.jl:277-299) straight into the packed CSR ----
 * The reference's text format: one document = a block of 1 + counts + readers + ratings lines of `delim`-separated
 * positive integers (terms; then counts, readers, ratings as switched on).  Missing counts / ratings default to 1
 * (src/Corpus.jl:18-21).  Ids are converted to 0-based.  With condense != 0 equal term ids of a document are merged and
 * their counts added, terms sorted ascending (condense_corp!, src/Corpus.jl:523-531) -- the form the engine needs
 * (quirk Q1).  A block that does not parse fails with TMVB_ECORPUS and the reference's message
 * "document d beginning on line l failed to load." (:295).  The arrays are allocated by the library; release each
 * with tmvb_host_free.  V_seen / U_seen = 1 + the largest 0-based term / reader id read (0 if none). */
typedef struct {
    int64_t M, nnz, nR, V_seen, U_seen;
    int64_t* doc_ptr;  int32_t* terms;   int32_t* counts;     /* [M+1], [nnz], [nnz] */
    int64_t* rdr_ptr;  int32_t* readers; int32_t* ratings;    /* [M+1], [nR], [nR] */
} tmvb_docfile_t;
int tmvb_docfile_read(const char* path, char delim, int32_t counts, int32_t readers, int32_t ratings, int32_t condense,
                      tmvb_docfile_t* out);
void tmvb_docfile_free(tmvb_docfile_t* f);

/* ============================== LDA (src/gpuLDA.jl, oracle src/LDA.jl) ============================== */

/* gpuLDA(corp, K) (src/gpuLDA.jl:45-84).  State is initialised as the constructor does
 * (alpha=1, gamma=1, Elogtheta=psi(1)-psi(K)) except beta, which the reference draws from
 * Dirichlet(V,1) with Julia's RNG (:57): beta is uniform 1/V until tmvb_lda_set_state. */
int tmvb_lda_create(tmvb_ctx* ctx, tmvb_corpus* corp, int32_t K, tmvb_lda** out);
int tmvb_lda_destroy(tmvb_lda* h);

/* State half of update_buffer! (src/modelutils.jl:390-396) / the field copies of @gpu
 * (src/macros.jl:115-134).  NULL pointers leave the field unchanged; beta_old / Elogtheta_old
 * default to copies of beta / Elogtheta (they only feed the pre-training ELBO, src/macros.jl:126-132).
 * alpha[K], beta[K*V], gamma[K*M], Elogtheta[K*M]. */
int tmvb_lda_set_state(tmvb_lda* h, const double* alpha, const double* beta, const double* beta_old,
                       const double* gamma, const double* Elogtheta, const double* Elogtheta_old,
                       const double* elbo);
/* update_host! (src/modelutils.jl:501-516) without phi (never materialised).  NULL = skip. */
int tmvb_lda_get_state(tmvb_lda* h, double* alpha, double* beta, double* beta_old,
                       double* gamma, double* Elogtheta, double* Elogtheta_old, double* elbo);

/* update_phi! / update_gamma! / update_Elogtheta! sweeps + update_beta!(model, d) for every
 * document, with the CPU path's per-document early exit (src/LDA.jl:170-180; replaces the launches
 * at src/gpuLDA.jl:338-339, :294, :263-264, :202).  Accumulates the beta sufficient statistics
 * (the reference's beta_temp) on the device.  Asynchronous on the context's stream. */
int tmvb_lda_estep(tmvb_lda* h, int32_t viter, double vtol);

/* Packed sufficient statistics for the host's collective (new: the reference is single-device).
 * Layout: float32 [ S (K*V, column-major) | Elogtheta_sum (K) ]; *n_f32 = K*V + K.
 * tmvb_lda_reduce_docs must have run after the E-step for the tail to be valid.
 * A multi-process host all-reduces (sum) this buffer between tmvb_lda_reduce_docs and
 * tmvb_lda_update_beta; every rank then runs the identical M-step. */
int tmvb_lda_stats(tmvb_lda* h, void** dev_ptr, int64_t* n_f32);
/* Use caller-owned device memory (>= K*V+K floats) for the packed statistics. */
int tmvb_lda_bind_stats(tmvb_lda* h, void* dev_ptr, int64_t n_f32);
/* Elogtheta_sum = sum_d Elogtheta[:,d] (src/LDA.jl:98; replaces the kernel at src/gpuLDA.jl:264). */
int tmvb_lda_reduce_docs(tmvb_lda* h);
/* Document-sharded runs: M_total = corpus-wide document count used by update_alpha!;
 * distributed != 0 makes update_alpha! read Elogtheta_sum from the (all-reduced) statistics tail. */
int tmvb_lda_set_distributed(tmvb_lda* h, int64_t M_total, int32_t distributed);

/* update_beta!(model) (src/LDA.jl:121-125; replaces src/gpuLDA.jl:201-204): beta_old <- beta,
 * beta <- row-normalised statistics, statistics <- 0. */
int tmvb_lda_update_beta(tmvb_lda* h);
/* update_alpha! (src/LDA.jl:97-118; replaces the host fp32 loop at src/gpuLDA.jl:132-154):
 * fp64 Newton on the device. */
int tmvb_lda_update_alpha(tmvb_lda* h, int32_t niter, double ntol);
/* update_elbo! (src/LDA.jl:83-93) evaluated on the device from the current state; returns the sum
 * over this context's documents (a multi-process host adds the ranks' values) and stores it.  Right after
 * estep -> update_beta the corpus-level term E_q[log p(w)] = sum S .* log(beta + eps) comes from update_beta's statistics;
 * a document-sharded context (global S after the all-reduce) contributes the share M / M_total of it, so the ranks'
 * values still add up to the ELBO. */
int tmvb_lda_update_elbo(tmvb_lda* h, double* elbo);

/* train! (src/gpuLDA.jl:347-376 signature, src/LDA.jl:161-187 semantics incl. check_elbo!
 * src/modelutils.jl:574-585).  checkelbo <= 0 means Inf.  elbo_traj[iter] (may be NULL) receives the
 * ELBO per outer iteration (NaN where not evaluated); *elbo_baseline (may be NULL) the ELBO evaluated before the first
 * iteration (src/LDA.jl:167; the value the first printed delta refers to), or the stored model.elbo when it is not
 * evaluated.  With a communicator attached (tmvb_lda_set_comm) this is the document-sharded train!: every rank calls it
 * with the same arguments; one all-reduce of the packed statistics per iteration, and of the ELBO when it is checked,
 * so all ranks stop at the same iteration. */
int tmvb_lda_train(tmvb_lda* h, int32_t iter, double tol, int32_t niter, double ntol,
                   int32_t viter, double vtol, int32_t checkelbo,
                   double* elbo_traj, int32_t* iters_done, double* elbo_baseline);
/* Attach (comm != NULL) or detach a communicator: the handle's corpus is this rank's document shard of a corpus of
 * M_total documents.  Implies tmvb_lda_set_distributed.  The communicator is not owned by the handle. */
int tmvb_lda_set_comm(tmvb_lda* h, tmvb_comm* comm, int64_t M_total);

/* One rank's E-step, per-document sums and the sum-all-reduce of the packed statistics over the handle's communicator, in one
 * asynchronous call (the sharded train! uses it when the process drives one handle; a host that composes the iteration itself calls
 * it INSTEAD of tmvb_lda_estep + tmvb_lda_reduce_docs + tmvb_comm_allreduce, then tmvb_lda_update_beta / _alpha as before).
 * The collective of the multi-device precedent (v0.6/src/gpuLDA.jl:200-225: the host gathers every device's buffer after the
 * E-step) is issued in pieces, in the same order on every rank: (1) the Elogtheta_sum tail (K floats) on a side stream as soon as
 * the document kernels' column sums exist, i.e. under the statistics pass -- tmvb_lda_update_alpha then starts from that event
 * instead of after the whole buffer; (2) with TMVB_AR_SLICES = S > 1 (default 1) the last statistics pass runs in S vocabulary
 * slices and the slab of S a slice completes is all-reduced on the side stream while the next slice's pass runs; (3) the last
 * (or only) slab on the context's stream.  The first call on a communicator is collective beyond that: the ranks sum their
 * postings per term (V doubles) to agree on the cuts and the order.  Every rank must call it the same number of times.  Results are
 * identical on every rank, and equal to the three-call form's up to the fp32 summation order of the collective: bit for bit with one or two
 * ranks (tests/test_sliced_allreduce_gpu.py), while with three or more the ring / tree chunking of RCCL depends on a buffer's offset and length, so
 * the K-float tail and the slabs reduced as separate collectives may differ from ONE collective of K*V + K floats in the last bit.
 * OPT-IN since round 5: the library's sharded train! and bench.py issue the three-call form (one collective per iteration) unless
 * TMVB_FUSED_ALLREDUCE=1 -- two streams of collectives on one communicator have not run on multi-GPU hardware yet.  TMVB_EINVAL without a communicator. */
int tmvb_lda_estep_allreduce(tmvb_lda* h, int32_t viter, double vtol);

/* The slab plan tmvb_lda_estep_allreduce derives from the GLOBAL postings per term (host arithmetic only, no device: every rank
 * computes the same plan from the same all-reduced counts).  counts[V] >= 0; want >= 1 slabs asked for.  Out: *slices = S (<= want,
 * <= V, 1 if every count is zero), cuts[0..S] (slab s = term ids [cuts[s], cuts[s + 1]), cuts[0] = 0, cuts[S] = V; equal shares of
 * postings(v) / nnz + 1 / V), order[0..S) (the slabs in issue order: Johnson's rule for pass-then-wire).  The caller provides
 * want + 1 and want entries. */
int tmvb_allreduce_plan(const double* counts, int64_t V, int32_t want, int64_t* cuts, int32_t* order, int32_t* slices);
/* One host thread, n GPUs: hs[i] carries the i-th communicator of tmvb_comm_create_rccl_all (n = 1: same as
 * tmvb_lda_train).  The n all-reduces of an iteration are issued as one RCCL group. */
int tmvb_lda_train_group(tmvb_lda* const* hs, int32_t n, int32_t iter, double tol, int32_t niter, double ntol,
                         int32_t viter, double vtol, int32_t checkelbo,
                         double* elbo_traj, int32_t* iters_done, double* elbo_baseline);

/* Diagnostics: histogram of sweeps per document of the last E-step (hist[0..viter]), and the per-document counts
 * themselves (out[M], corpus document order, saturating at 255). */
int tmvb_lda_sweep_hist(tmvb_lda* h, int64_t* hist, int32_t nbins);
int tmvb_lda_doc_sweeps(tmvb_lda* h, uint8_t* out);
/* Which form the last tmvb_lda_update_elbo / check of tmvb_lda_train took (update_elbo!, src/LDA.jl:83-93): *form = 1 the decomposed form -- the
 * iteration's own statistics passes left sum_n c_n log s_n per postings chunk, update_beta! left sum S (log beta_new - log beta_old), and one
 * per-document kernel adds Elogptheta, sum_k (gamma_k - alpha_k)(Elogtheta_k - Elogtheta_old_k) and the Dirichlet entropy: no second walk over the
 * corpus; taken by every iteration tmvb_lda_train checks (K <= 124: the statistics pass recomputes the token weights up to KP = 124; viter > 0) and, with TMVB_LDA_ELBO_PARTS=2 at tmvb_lda_create, by the stepwise operators too;
 * *form = 0 the token walk (phi rebuilt from beta_old / Elogtheta_old per token: any state, e.g. right after tmvb_lda_set_state);
 * TMVB_LDA_ELBO_PARTS=0 forces it.  Both evaluate the same sum; they differ by fp32 rounding only (tests/test_lda_elbo_parts_gpu.py). */
int tmvb_lda_elbo_form(tmvb_lda* h, int32_t* form);
/* Number of kernel launches one tmvb_lda_estep issues (one per document-length bucket). */
int tmvb_lda_estep_launches(tmvb_lda* h, int32_t* n);
/* Timing of the last tmvb_lda_estep on the context's stream, from HIP events (ms). */
int tmvb_lda_last_estep_ms(tmvb_lda* h, float* ms);

/* ============================== fLDA (new device path; oracle src/fLDA.jl) ==============================
 * Filtered LDA: LDA plus a per-token Bernoulli switch tau_n (prior eta) and a background distribution kappa.  The reference
 * has NO accelerator path for it (`@gpu train!` on an fLDA does nothing, src/macros.jl:274-278); SURVEY.md section 8 f4 names
 * it as the next row.  The entry points mirror the LDA ones; the per-document operator chain update_phi! / update_tau! /
 * update_gamma! / update_Elogtheta! (src/fLDA.jl:188, :180, :173, :166) is one fused kernel.  K <= 1024 (1, 2, 4, 8 or 16 topic slots per lane, as LDA).
 * tau / tau_old are flat double[nnz] arrays in the CSR token order of tmvb_corpus_create (= vcat(model.tau...)). */
int tmvb_flda_create(tmvb_ctx* ctx, tmvb_corpus* corp, int32_t K, tmvb_flda** out);           /* fLDA(corp, K), src/fLDA.jl:28-60 */
int tmvb_flda_destroy(tmvb_flda* h);
/* NULL = unchanged; kappa_old / beta_old / Elogtheta_old / tau_old default to the current values.  eta[1], alpha[K], kappa[V],
 * beta[K*V], gamma[K*M], Elogtheta[K*M], tau[nnz]. */
int tmvb_flda_set_state(tmvb_flda* h, const double* eta, const double* alpha, const double* kappa, const double* kappa_old,
                        const double* beta, const double* beta_old, const double* gamma, const double* Elogtheta,
                        const double* Elogtheta_old, const double* tau, const double* tau_old, const double* elbo);
int tmvb_flda_get_state(tmvb_flda* h, double* eta, double* alpha, double* kappa, double* kappa_old, double* beta, double* beta_old,
                        double* gamma, double* Elogtheta, double* Elogtheta_old, double* tau, double* tau_old, double* elbo);
/* sweeps of src/fLDA.jl:224-233 + update_beta!(model, d) (:159) + update_kappa!(model, d) (:145) for every document */
int tmvb_flda_estep(tmvb_flda* h, int32_t viter, double vtol);
int tmvb_flda_reduce_docs(tmvb_flda* h);                                /* Elogtheta_sum, src/fLDA.jl:129 */
/* Packed statistics: float32 [ S (K*V) | kappa_stats (V) | Elogtheta_sum (K) ]. */
int tmvb_flda_stats(tmvb_flda* h, void** dev_ptr, int64_t* n_f32);
int tmvb_flda_update_beta(tmvb_flda* h);                                /* update_beta! (:152) and update_kappa! (:138) */
int tmvb_flda_update_alpha(tmvb_flda* h, int32_t niter, double ntol);   /* update_alpha! (:128-150) */
int tmvb_flda_update_eta(tmvb_flda* h);                                 /* update_eta! (:122-124); after update_beta */
int tmvb_flda_update_elbo(tmvb_flda* h, double* elbo);                  /* update_elbo! (:108-118) */
/* train! (src/fLDA.jl:213-247); arguments as tmvb_lda_train. */
int tmvb_flda_train(tmvb_flda* h, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter, double vtol,
                    int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline);
/* Document-sharded run: the shard belongs to a corpus of M_total documents and C_total tokens (sum of all counts). */
int tmvb_flda_set_comm(tmvb_flda* h, tmvb_comm* comm, int64_t M_total, int64_t C_total);
int tmvb_flda_train_group(tmvb_flda* const* hs, int32_t n, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter,
                          double vtol, int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline);
int tmvb_flda_doc_sweeps(tmvb_flda* h, uint8_t* out);
/* As tmvb_lda_elbo_form (update_elbo!, src/fLDA.jl:108-118): *form = 1 if the last tmvb_flda_update_elbo / check of tmvb_flda_train took the decomposed
 * form -- the checked iteration's document kernel left per token the log-sum-exp of its last phi column and the exponent update_tau! forms
 * (sum_i phi_in log(beta_old + eps), :184), update_beta! left sum S (log(beta_new + eps) - log(beta_old + eps)), and one elementwise pass per document adds
 * c_n [lse_n + (tau_n - tau_old_n) A_n + (1 - tau_n) log(kappa + eps) + H(tau_n)], sum_i (gamma_i - alpha_i)(Elogtheta_i - Elogtheta_old_i), the Dirichlet
 * terms and Elogpc: no phi is rebuilt; *form = 0 the token walk (any state, e.g. after tmvb_flda_set_state).  TMVB_FLDA_ELBO_PARTS at tmvb_flda_create:
 * 0 never, 1 (default) the iterations tmvb_flda_train checks, 2 every E-step collects (the stepwise operators take the form too).  viter > 0. */
int tmvb_flda_elbo_form(tmvb_flda* h, int32_t* form);
int tmvb_flda_last_estep_ms(tmvb_flda* h, float* ms);

/* ============================== CTM (src/gpuCTM.jl, oracle src/CTM.jl) ============================== */

/* gpuCTM(corp, K) (src/gpuCTM.jl:45-98).  Constructor state as src/CTM.jl:37-48 (mu=0, sigma=invsigma=I,
 * lambda=0, vsq=1, logzeta=0.5); beta is uniform until tmvb_ctm_set_state (the reference draws it with
 * Julia's RNG).  K <= 256 (CTM_MAX_K; beyond it the Julia shim trains on the reference's CPU model behind a warning).  E-step kernels behind
 * tmvb_ctm_estep: K <= 52 -- one LANE per
 * document, invsigma streamed through scalar registers, the lambda Newton systems solved by Jacobi-preconditioned CG to
 * max(1e-4 |g|, 5 % of ntol); its documents of more than 2048 unique terms (and everything under TMVB_CTM_BATCH=0) -- one wave
 * per document, Gauss-Jordan in registers (lane = matrix row); 52 < K <= 128 -- one wave per document, lane = matrix row (two
 * topic slots per lane beyond 64), the same CG against one copy of invsigma in LDS per workgroup (TMVB_CTM_GENERIC_CG=0: round 1's
 * Gauss-Jordan through LDS, 30 times slower at K = 100); 128 < K <= 256 -- four topic slots per lane, invsigma read from global memory by
 * columns (L2-resident), the whole LDS for the tile windows, update_sigma!'s fp64 inversion in a global workspace, dense E rows and the
 * scalar statistics kernel beyond KP / 4 = 64 chunks (K > 252).  fCTM: the same limits. */
int tmvb_ctm_create(tmvb_ctx* ctx, tmvb_corpus* corp, int32_t K, tmvb_ctm** out);
int tmvb_ctm_destroy(tmvb_ctm* h);

/* update_buffer! state half (src/modelutils.jl:420-431) / @gpu field copies (src/macros.jl:152-175).
 * mu[K], sigma[K*K], invsigma[K*K], beta[K*V], lambda[K*M], vsq[K*M], logzeta[M]; NULL = unchanged;
 * beta_old / lambda_old default to beta / lambda. */
int tmvb_ctm_set_state(tmvb_ctm* h, const double* mu, const double* sigma, const double* invsigma,
                       const double* beta, const double* beta_old, const double* lambda,
                       const double* lambda_old, const double* vsq, const double* logzeta, const double* elbo);
/* update_host! (src/modelutils.jl:519-537) without phi. */
int tmvb_ctm_get_state(tmvb_ctm* h, double* mu, double* sigma, double* invsigma, double* beta, double* beta_old,
                       double* lambda, double* lambda_old, double* vsq, double* logzeta, double* elbo);

/* update_phi! / update_logzeta! / update_vsq! / update_lambda! sweeps + update_beta!(model, d) for every
 * document with the CPU path's semantics (src/CTM.jl:194-205; replaces the launches at
 * src/gpuCTM.jl:478-479, :425, :390, :342, :254).  Asynchronous on the context's stream. */
int tmvb_ctm_estep(tmvb_ctm* h, int32_t niter, double ntol, int32_t viter, double vtol);
/* sum_d lambda_d, sum_d vsq_d and the scatter matrix sum_d (lambda_d - mu)(lambda_d - mu)^T (f32 MFMA) with
 * the current (= previous-iteration) mu, into the statistics tail.  tmvb_ctm_estep already computes them on a side
 * stream under its statistics pass; the call then only acknowledges that result (it recomputes on the context's stream
 * when lambda / vsq / mu were changed through the API since, or when no E-step came before). */
int tmvb_ctm_reduce_docs(tmvb_ctm* h);
/* Packed statistics for the host's all-reduce: float32 [ S (K*V) | sum_lambda (K) | sum_vsq (K) | scatter (K*K) ]. */
int tmvb_ctm_stats(tmvb_ctm* h, void** dev_ptr, int64_t* n_f32);
int tmvb_ctm_bind_stats(tmvb_ctm* h, void* dev_ptr, int64_t n_f32);
int tmvb_ctm_set_distributed(tmvb_ctm* h, int64_t M_total, int32_t distributed);
/* update_beta! (src/CTM.jl:114-118; replaces src/gpuCTM.jl:253-256). */
int tmvb_ctm_update_beta(tmvb_ctm* h);
/* update_sigma! (src/CTM.jl:108-111; replaces src/gpuCTM.jl:200-206 incl. the host `inv`): uses the
 * PREVIOUS mu (call before tmvb_ctm_update_mu, as train! does, src/CTM.jl:207-208). */
int tmvb_ctm_update_sigma(tmvb_ctm* h);
/* update_mu! (src/CTM.jl:102-104; replaces src/gpuCTM.jl:166-168). */
int tmvb_ctm_update_mu(tmvb_ctm* h);
/* update_elbo! (src/CTM.jl:89-98) on the device; sum over this context's documents. */
int tmvb_ctm_update_elbo(tmvb_ctm* h, double* elbo);
/* As tmvb_lda_elbo_form: *form = 1 if the last tmvb_ctm_update_elbo took the decomposed form (the checked iteration's E-step kernels left
 * sum_i (phi counts)_i (lambda_i - lambda_old_i) per document, its statistics pass sum_n c_n log s_n per postings chunk, update_beta!
 * sum S (log(beta_new + eps) - log beta_old): no token loop), 0 for the token walk (any state).  Every iteration tmvb_ctm_train checks takes
 * it (K <= 124, viter > 0); TMVB_CTM_ELBO_PARTS=2 at tmvb_ctm_create: the stepwise operators too, =0: never. */
int tmvb_ctm_elbo_form(tmvb_ctm* h, int32_t* form);
/* train! (src/gpuCTM.jl:487-519 signature, src/CTM.jl:185-213 semantics); elbo_baseline / communicator as for LDA. */
int tmvb_ctm_train(tmvb_ctm* h, int32_t iter, double tol, int32_t niter, double ntol,
                   int32_t viter, double vtol, int32_t checkelbo, double* elbo_traj, int32_t* iters_done,
                   double* elbo_baseline);
int tmvb_ctm_set_comm(tmvb_ctm* h, tmvb_comm* comm, int64_t M_total);
int tmvb_ctm_train_group(tmvb_ctm* const* hs, int32_t n, int32_t iter, double tol, int32_t niter, double ntol,
                         int32_t viter, double vtol, int32_t checkelbo, double* elbo_traj, int32_t* iters_done,
                         double* elbo_baseline);
/* Diagnostics of the last E-step: sweeps-per-document histogram and total lambda-Newton steps. */
int tmvb_ctm_sweep_hist(tmvb_ctm* h, int64_t* hist, int32_t nbins, int64_t* newton_steps);
int tmvb_ctm_doc_sweeps(tmvb_ctm* h, uint8_t* out);
int tmvb_ctm_last_estep_ms(tmvb_ctm* h, float* ms);
/* Diagnostics of the last E-step when the lane-per-document kernel ran (K <= 52, csrc/tmvb_ctm_batch.h), 12 values: out[0]
 * conjugate-gradient trips, [1] Newton trips, [2] waves (summed over waves); with TMVB_CTM_PROF=1 in the environment also
 * [3..10] shader cycles per phase (token, logzeta, vsq, gradient assembly, CG, gradient mat-vec, lambda update, spare) and
 * [11] whole-kernel cycles.  Zeros otherwise. */
int tmvb_ctm_solver_stats(tmvb_ctm* h, int64_t* out12);

/* ============================== fCTM (new device path; oracle src/fCTM.jl) ==============================
 * Filtered CTM: CTM plus the per-token switch tau_n (prior eta) and the background distribution kappa of fLDA.  No accelerator
 * path in the reference (src/macros.jl:274-278).  Same Newton machinery and K limits as CTM; the per-document chain is
 * update_phi! / update_tau! / update_logzeta! / update_lambda! / update_vsq! (src/fCTM.jl:236-241 -- lambda BEFORE vsq, unlike CTM).
 * eta is a fixed parameter (update_eta! is commented out of train!, src/fCTM.jl:253). */
int tmvb_fctm_create(tmvb_ctx* ctx, tmvb_corpus* corp, int32_t K, tmvb_fctm** out);           /* fCTM(corp, K), src/fCTM.jl:32-65 */
int tmvb_fctm_destroy(tmvb_fctm* h);
/* eta[1], mu[K], sigma/invsigma[K*K], kappa[V], beta[K*V], lambda/vsq[K*M], logzeta[M], tau[nnz] (CSR token order); NULL = unchanged;
 * the *_old arguments default to the current values. */
int tmvb_fctm_set_state(tmvb_fctm* h, const double* eta, const double* mu, const double* sigma, const double* invsigma,
                        const double* kappa, const double* kappa_old, const double* beta, const double* beta_old,
                        const double* lambda, const double* lambda_old, const double* vsq, const double* logzeta,
                        const double* tau, const double* tau_old, const double* elbo);
int tmvb_fctm_get_state(tmvb_fctm* h, double* eta, double* mu, double* sigma, double* invsigma, double* kappa, double* kappa_old,
                        double* beta, double* beta_old, double* lambda, double* lambda_old, double* vsq, double* logzeta,
                        double* tau, double* tau_old, double* elbo);
/* sweeps of src/fCTM.jl:233-248 + update_beta!(model, d) (:155) + update_kappa!(model, d) (:141) for every document.
 * Packed statistics: float32 [ S (K*V) | sum_lambda (K) | sum_vsq (K) | scatter (K*K) | kappa_stats (V) ]. */
int tmvb_fctm_estep(tmvb_fctm* h, int32_t niter, double ntol, int32_t viter, double vtol);
int tmvb_fctm_reduce_docs(tmvb_fctm* h);
int tmvb_fctm_update_beta(tmvb_fctm* h);                                /* update_beta! (:148) and update_kappa! (:134) */
int tmvb_fctm_update_sigma(tmvb_fctm* h);                               /* update_sigma! (:128-131), previous mu */
int tmvb_fctm_update_mu(tmvb_fctm* h);                                  /* update_mu! (:122-124) */
int tmvb_fctm_update_elbo(tmvb_fctm* h, double* elbo);                  /* update_elbo! (:105-115) */
/* train! (src/fCTM.jl:226-262); arguments as tmvb_ctm_train. */
int tmvb_fctm_train(tmvb_fctm* h, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter, double vtol,
                    int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline);
int tmvb_fctm_set_comm(tmvb_fctm* h, tmvb_comm* comm, int64_t M_total);
int tmvb_fctm_train_group(tmvb_fctm* const* hs, int32_t n, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter,
                          double vtol, int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline);
int tmvb_fctm_sweep_hist(tmvb_fctm* h, int64_t* hist, int32_t nbins, int64_t* newton_steps);
int tmvb_fctm_doc_sweeps(tmvb_fctm* h, uint8_t* out);
/* As tmvb_flda_elbo_form for update_elbo! of src/fCTM.jl:105-115 (lambda for Elogtheta: the E-step kernels' exit test leaves
 * sum_i (phi counts)_i (lambda_i - lambda_old_i) per document); TMVB_FCTM_ELBO_PARTS at tmvb_fctm_create. */
int tmvb_fctm_elbo_form(tmvb_fctm* h, int32_t* form);

/* ============================== CTPF (src/gpuCTPF.jl, oracle src/CTPF.jl) ============================== */

/* gpuCTPF(corp, K) (src/gpuCTPF.jl:68-152).  Constructor state as src/CTPF.jl:81-100 (he=1, rates=1, gimel=zayin=1,
 * hyper-parameters a..h = 0.1); alef is 1 until tmvb_ctpf_set_state (the reference draws it with Julia's RNG, :83).
 * K <= 512: the grid-tile fast path (one topic slot per lane of a 16 x 4 lane grid) covers K <= 60; beyond it one wave per document with
 * 2 / 4 / 8 topic slots per lane (lane l owns topics l, l + 64, ...: K <= 128 / 256 / 512), rows of more than 64 chunks through dense E rows
 * and the scalar statistics kernel. */
int tmvb_ctpf_create(tmvb_ctx* ctx, tmvb_corpus* corp, int32_t K, tmvb_ctpf** out);
int tmvb_ctpf_destroy(tmvb_ctpf* h);
/* update_buffer! state half (src/modelutils.jl:474-493).  hyper[8] = a..h; alef[K*V], he[K*U], bet/vav/dalet/het[K],
 * gimel/zayin[K*M]; NULL = unchanged.  The *_old copies are set equal to the new values. */
int tmvb_ctpf_set_state(tmvb_ctpf* h, const double* hyper, const double* alef, const double* he, const double* bet,
                        const double* vav, const double* dalet, const double* het, const double* gimel,
                        const double* zayin, const double* elbo);
/* The *_old fields of update_buffer! (src/modelutils.jl:474-493): update_elbo! rebuilds phi / xi from them
 * (src/CTPF.jl:239-240) and @gpu copies them (src/macros.jl:229-260), so a trained model that is uploaded again must
 * bring them along or its baseline ELBO differs.  Call after tmvb_ctpf_set_state; NULL = leave unchanged. */
int tmvb_ctpf_set_state_old(tmvb_ctpf* h, const double* alef_old, const double* he_old, const double* bet_old,
                            const double* vav_old, const double* dalet_old, const double* het_old,
                            const double* gimel_old, const double* zayin_old);
/* update_host! (src/modelutils.jl:539-570) without phi / xi.  rates[8*K] = bet, vav, dalet, het, then their *_old. */
int tmvb_ctpf_get_state(tmvb_ctpf* h, double* alef, double* alef_old, double* he, double* he_old, double* rates,
                        double* gimel, double* gimel_old, double* zayin, double* zayin_old, double* elbo);
/* update_xi! / update_phi! / update_zayin! / update_gimel! sweeps + update_he!(d) / update_alef!(d) for every
 * document, CPU-path semantics (src/CTPF.jl:353-365; replaces src/gpuCTPF.jl:667-668, :599-600, :511, :382, :448, :315). */
int tmvb_ctpf_estep(tmvb_ctpf* h, int32_t viter, double vtol);
/* sum_d gimel_d, sum_d zayin_d into the statistics tail. */
int tmvb_ctpf_reduce_docs(tmvb_ctpf* h);
/* Packed statistics: float32 [ alef_stats (K*V) | he_stats (K*U) | sum_gimel (K) | sum_zayin (K) ] (priors NOT included). */
int tmvb_ctpf_stats(tmvb_ctpf* h, void** dev_ptr, int64_t* n_f32);
int tmvb_ctpf_bind_stats(tmvb_ctpf* h, void* dev_ptr, int64_t n_f32);
int tmvb_ctpf_set_distributed(tmvb_ctpf* h, int32_t distributed);
/* update_he!, update_alef!, update_dalet!, update_het!, update_bet!, update_vav! in that order
 * (src/CTPF.jl:366-371; replaces src/gpuCTPF.jl:448, :315, :418, :539, :344, :482). */
int tmvb_ctpf_mstep(tmvb_ctpf* h);
/* update_elbo! (src/CTPF.jl:234-247) on the device.  The Binomial sums of Elogpya/Elogpyb/Elogpz and of the Multinomial
 * entropies cancel identically in :243 and are not evaluated.  _parts returns the per-document part (sum over this
 * context's documents) and the global (beta, eta) part separately for document-sharded hosts. */
int tmvb_ctpf_update_elbo(tmvb_ctpf* h, double* elbo);
/* As tmvb_lda_elbo_form: *form = 1 if the last update_elbo! took the decomposed form -- no entry (token / reader) is walked: the checked iteration's document
 * kernels left their softmax shifts, its statistics passes sum_n c_n log s_n per postings chunk, and the entries' remaining terms are sums the M-step already
 * has (sum (alef - a)(psi(alef) - psi(alef_old)), the row sums of alef, sum_d gimel_d); *form = 0 the table form (any state).  Every iteration tmvb_ctpf_train
 * checks on an unsharded handle takes it (K <= 124, viter > 0); TMVB_CTPF_ELBO_PARTS=2 at tmvb_ctpf_create: the stepwise operators too, =0: never. */
int tmvb_ctpf_elbo_form(tmvb_ctpf* h, int32_t* form);
int tmvb_ctpf_update_elbo_parts(tmvb_ctpf* h, double* doc_part, double* global_part);
/* train! (src/gpuCTPF.jl:677-705 signature, src/CTPF.jl:344-376 semantics).  checkelbo <= 0 means Inf.
 * elbo_baseline / communicator as for LDA (the per-document ELBO part is all-reduced, the global part added once). */
int tmvb_ctpf_train(tmvb_ctpf* h, int32_t iter, double tol, int32_t viter, double vtol, int32_t checkelbo,
                    double* elbo_traj, int32_t* iters_done, double* elbo_baseline);
int tmvb_ctpf_set_comm(tmvb_ctpf* h, tmvb_comm* comm);
int tmvb_ctpf_train_group(tmvb_ctpf* const* hs, int32_t n, int32_t iter, double tol, int32_t viter, double vtol,
                          int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline);
int tmvb_ctpf_sweep_hist(tmvb_ctpf* h, int64_t* hist, int32_t nbins);
int tmvb_ctpf_doc_sweeps(tmvb_ctpf* h, uint8_t* out);
int tmvb_ctpf_last_estep_ms(tmvb_ctpf* h, float* ms);
/* Recommendation post-processing at the end of train!(model::CTPF) (src/CTPF.jl:379-399; src/gpuCTPF.jl:711-731 runs the
 * same code on the host).  From the resident state:
 *   scores (or NULL)  double[M*U], column-major M x U as model.scores: sum_k he[k,u]/vav[k] (gimel[k,d]/dalet[k] + zayin[k,d]/het[k])
 *   drecs  (or NULL)  int32[M*U]: row d holds the 0-based users that are not readers of document d, by descending
 *                     score, equal scores in descending index order (= reverse(sortperm(...))); drec_count[d] entries valid
 *   urecs             int32[U*M]: row u holds the 0-based documents outside user u's library, same order; urec_count[u] valid
 * drecs/urecs/drec_count/urec_count are given together or all NULL.  ms_scores / ms_rank (or NULL): device time of the
 * score pass and of the two segmented sorts. */
int tmvb_ctpf_recommend(tmvb_ctpf* h, double* scores, int32_t* drecs, int32_t* drec_count, int32_t* urecs,
                        int32_t* urec_count, float* ms_scores, float* ms_rank);

#ifdef __cplusplus
}
#endif
#endif /* TMVB_H */
