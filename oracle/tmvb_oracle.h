/*
 * tmvb_oracle.h -- fp64 CPU restatement of TopicModelsVB.jl's CPU `train!` path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped HIP path never calls it.
 *
 * PARITY UNPINNED: the reference (Julia) ships no tests / golden vectors for this path
 * (v0.6/test/runtests.jl is empty, no test/ in v1), Julia is not installed in the build
 * image and the NSF/CiteULike docfiles are absent (.MISSING_LARGE_BLOBS), so this
 * restatement could not be checked against outputs of the reference itself.  It is pinned
 * only by (1) an independent NumPy/SciPy restatement (oracle/oracle_np.py) agreeing to
 * <=1e-10, (2) closed-form known-answer tests and (3) mpmath checks of the special
 * functions -- see tests/test_oracle_*.py.
 *
 * Conventions: all matrices column-major K x (.), flat double arrays (the reference's
 * Vector{Vector{Float64}} per-document state is stored as a K x M matrix, column d = doc d).
 * Term / reader ids are 0-based int32; doc_ptr / rdr_ptr are int64 CSR offsets.
 * All citations are file:line relative to the reference repository root.
 */
#ifndef TMVB_ORACLE_H
#define TMVB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/utils.jl:3  EPSILON = eps(1e-14) = 2^-99 */
#define ORC_EPSILON 1.5777218104420236e-30

/* ---- special functions (SpecialFunctions.jl 0.8-0.10; not vendored in the reference) ---- */
double orc_digamma(double x);   /* used at src/LDA.jl:38,103,138  src/CTPF.jl:116..336 */
double orc_trigamma(double x);  /* used at src/LDA.jl:104-105 */
double orc_lgamma(double x);    /* loggamma, src/LDA.jl:51 */
double orc_finite(double x);    /* src/utils.jl:107 */
void   orc_digamma_vec(const double* x, double* out, int64_t n);
void   orc_trigamma_vec(const double* x, double* out, int64_t n);

/* ---- LDA (src/LDA.jl) ---- */

/* One document's sweep loop, src/LDA.jl:171-178 (+ update_phi!/gamma!/Elogtheta! :136-154).
 * phi is a K x N_d workspace that holds the last-sweep phi on return.  Returns #sweeps run. */
int orc_lda_doc_sweeps(int64_t K, int64_t Nd, const int32_t* terms, const int32_t* counts,
                       const double* alpha, const double* beta,
                       double* gamma_d, double* Elogtheta_d, double* Elogtheta_old_d,
                       double* phi, int viter, double vtol);

/* E-step over documents [d0, d1): sweeps + update_beta!(model, d) (src/LDA.jl:170-180).
 * beta_temp (K x V) is accumulated into (not cleared).  sweeps_out[d-d0] optional. */
int orc_lda_estep(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  int64_t d0, int64_t d1,
                  const double* alpha, const double* beta, double* beta_temp,
                  double* gamma, double* Elogtheta, double* Elogtheta_old,
                  int viter, double vtol, int32_t* sweeps_out);

/* Same E-step, OpenMP document-parallel with per-thread beta_temp (for the CPU timing
 * baseline only; summation order differs from the serial path). Returns threads used. */
/* releases the per-thread statistics buffers the OpenMP E-steps keep between calls */
void orc_omp_pool_free(void);
int orc_lda_estep_omp(int64_t M, int64_t V, int64_t K,
                      const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                      int64_t d0, int64_t d1,
                      const double* alpha, const double* beta, double* beta_temp,
                      double* gamma, double* Elogtheta, double* Elogtheta_old,
                      int viter, double vtol, int nthreads);
/* the same with per-document sweep counts (sweeps_out[d - d0], may be NULL): the full-size parity checks */
int orc_lda_estep_omp_sw(int64_t M, int64_t V, int64_t K,
                      const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                      int64_t d0, int64_t d1,
                      const double* alpha, const double* beta, double* beta_temp,
                      double* gamma, double* Elogtheta, double* Elogtheta_old,
                      int viter, double vtol, int nthreads, int32_t* sweeps_out);

/* update_beta!(model), src/LDA.jl:121-125: beta_old <- beta; beta <- rownormalise(beta_temp);
 * beta_temp <- 0. */
void orc_lda_update_beta(int64_t V, int64_t K, double* beta, double* beta_old, double* beta_temp);

/* update_alpha!, src/LDA.jl:97-118.  Elogtheta_sum is K-vector sum_d Elogtheta[:,d] (:98);
 * Mtot is the corpus-wide document count.  Returns Newton iterations taken. */
int orc_lda_update_alpha(int64_t K, int64_t Mtot, const double* Elogtheta_sum,
                         double* alpha, int niter, double ntol);

void orc_lda_elogtheta_sum(int64_t M, int64_t K, const double* Elogtheta, double* out);

/* update_elbo!, src/LDA.jl:83-93 (terms :50-80) over documents [d0,d1).  The
 * corpus-level constants in Elogptheta use the given alpha. */
double orc_lda_update_elbo(int64_t M, int64_t V, int64_t K,
                           const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                           int64_t d0, int64_t d1,
                           const double* alpha, const double* beta, const double* beta_old,
                           const double* gamma, const double* Elogtheta,
                           const double* Elogtheta_old);

/* train!, src/LDA.jl:161-191 + check_elbo! src/modelutils.jl:574-585.
 * checkelbo <= 0 means Inf.  elbo_traj[k] receives the ELBO evaluated at outer iteration
 * k+1 (NaN where not evaluated).  *elbo is the model.elbo field (in/out).
 * Returns the number of outer iterations executed. */
int orc_lda_train(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  double* alpha, double* beta, double* beta_old,
                  double* gamma, double* Elogtheta, double* Elogtheta_old, double* elbo,
                  int iter, double tol, int niter, double ntol, int viter, double vtol,
                  int checkelbo, double* elbo_traj, int64_t* sweep_hist /* viter+1 bins or NULL */);

/* ---- CTM (src/CTM.jl) ---- */

int orc_ctm_estep(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  int64_t d0, int64_t d1,
                  const double* mu, const double* invsigma, const double* beta, double* beta_temp,
                  double* lambda, double* lambda_old, double* vsq, double* logzeta,
                  int niter, double ntol, int viter, double vtol,
                  int32_t* sweeps_out, int64_t* newton_steps_out);

int orc_ctm_estep_omp(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  int64_t d0, int64_t d1,
                  const double* mu, const double* invsigma, const double* beta, double* beta_temp,
                  double* lambda, double* lambda_old, double* vsq, double* logzeta,
                  int niter, double ntol, int viter, double vtol, int nthreads);
/* the same with per-document sweep counts and lambda Newton steps ([d - d0], may be NULL) */
int orc_ctm_estep_omp_sw(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  int64_t d0, int64_t d1,
                  const double* mu, const double* invsigma, const double* beta, double* beta_temp,
                  double* lambda, double* lambda_old, double* vsq, double* logzeta,
                  int niter, double ntol, int viter, double vtol, int nthreads,
                  int32_t* sweeps_out, int32_t* newton_out);

/* update_sigma! then update_mu! (src/CTM.jl:108-111, :102-104, order :207-208). Returns 0, or
 * nonzero if sigma is not positive definite. */
int orc_ctm_update_sigma_mu(int64_t M, int64_t K, const double* lambda, const double* vsq,
                            double* mu, double* sigma, double* invsigma);

double orc_ctm_update_elbo(int64_t M, int64_t V, int64_t K,
                           const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                           int64_t d0, int64_t d1,
                           const double* mu, const double* invsigma,
                           const double* beta, const double* beta_old,
                           const double* lambda, const double* lambda_old,
                           const double* vsq, const double* logzeta);

int orc_ctm_train(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  double* mu, double* sigma, double* invsigma, double* beta, double* beta_old,
                  double* lambda, double* lambda_old, double* vsq, double* logzeta, double* elbo,
                  int iter, double tol, int niter, double ntol, int viter, double vtol,
                  int checkelbo, double* elbo_traj);

/* ---- CTPF (src/CTPF.jl) ---- */

typedef struct {
    double a, b, c, d, e, f, g, h;   /* src/CTPF.jl:81 */
} orc_ctpf_hyper;

int orc_ctpf_estep(int64_t M, int64_t V, int64_t U, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                   int64_t d0, int64_t d1, const orc_ctpf_hyper* hp,
                   const double* alef, const double* he,
                   const double* bet, const double* vav, const double* dalet, const double* het,
                   double* alef_temp, double* he_temp,
                   double* gimel, double* gimel_old, double* zayin, double* zayin_old,
                   int viter, double vtol, int32_t* sweeps_out);

int orc_ctpf_estep_omp(int64_t M, int64_t V, int64_t U, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                   int64_t d0, int64_t d1, const orc_ctpf_hyper* hp,
                   const double* alef, const double* he,
                   const double* bet, const double* vav, const double* dalet, const double* het,
                   double* alef_temp, double* he_temp,
                   double* gimel, double* gimel_old, double* zayin, double* zayin_old,
                   int viter, double vtol, int nthreads);
/* the same with per-document sweep counts ([d - d0], may be NULL) */
int orc_ctpf_estep_omp_sw(int64_t M, int64_t V, int64_t U, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                   int64_t d0, int64_t d1, const orc_ctpf_hyper* hp,
                   const double* alef, const double* he,
                   const double* bet, const double* vav, const double* dalet, const double* het,
                   double* alef_temp, double* he_temp,
                   double* gimel, double* gimel_old, double* zayin, double* zayin_old,
                   int viter, double vtol, int nthreads, int32_t* sweeps_out);

/* Global updates in the reference order (src/CTPF.jl:366-371).  gimel_sum/zayin_sum are the
 * K-vectors sum_d gimel[:,d], sum_d zayin[:,d].  The *_old outputs receive the previous
 * values (alef_old/he_old: K x V / K x U). */
void orc_ctpf_mstep(int64_t V, int64_t U, int64_t K, const orc_ctpf_hyper* hp,
                    double* alef, double* alef_old, double* alef_temp,
                    double* he, double* he_old, double* he_temp,
                    const double* gimel_sum, const double* zayin_sum,
                    double* bet, double* bet_old, double* vav, double* vav_old,
                    double* dalet, double* dalet_old, double* het, double* het_old);

double orc_ctpf_update_elbo(int64_t M, int64_t V, int64_t U, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                   const orc_ctpf_hyper* hp,
                   const double* alef, const double* alef_old, const double* he, const double* he_old,
                   const double* bet, const double* bet_old, const double* vav, const double* vav_old,
                   const double* dalet, const double* dalet_old, const double* het, const double* het_old,
                   const double* gimel, const double* gimel_old,
                   const double* zayin, const double* zayin_old);

int orc_ctpf_train(int64_t M, int64_t V, int64_t U, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                   const orc_ctpf_hyper* hp,
                   double* alef, double* alef_old, double* he, double* he_old,
                   double* bet, double* bet_old, double* vav, double* vav_old,
                   double* dalet, double* dalet_old, double* het, double* het_old,
                   double* gimel, double* gimel_old, double* zayin, double* zayin_old,
                   double* elbo, int iter, double tol, int viter, double vtol,
                   int checkelbo, double* elbo_traj);


/* ---- filtered LDA (src/fLDA.jl) ---- tau / tau_old: flat [nnz] arrays in CSR token order */
int orc_flda_doc_sweeps(int64_t K, int64_t Nd, const int32_t* terms, const int32_t* counts,
                        double eta, const double* alpha, const double* kappa, const double* beta,
                        double* gamma_d, double* Elogtheta_d, double* Elogtheta_old_d,
                        double* tau_d, double* tau_old_d, double* phi, int viter, double vtol);
/* sweeps + update_beta!(model, d) + update_kappa!(model, d) over documents [d0, d1)  (src/fLDA.jl:222-236) */
int orc_flda_estep(int64_t M, int64_t V, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   int64_t d0, int64_t d1, double eta, const double* alpha, const double* kappa, const double* beta,
                   double* beta_temp, double* kappa_temp, double* gamma, double* Elogtheta, double* Elogtheta_old,
                   double* tau, double* tau_old, int viter, double vtol, int32_t* sweeps_out);
/* the same, OpenMP document-parallel (cpu_baseline only; condensed corpora); returns the team size */
int orc_flda_estep_omp(int64_t M, int64_t V, int64_t K,
                       const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                       int64_t d0, int64_t d1, double eta, const double* alpha, const double* kappa, const double* beta,
                       double* beta_temp, double* kappa_temp, double* gamma, double* Elogtheta, double* Elogtheta_old,
                       double* tau, double* tau_old, int viter, double vtol, int nthreads);
void orc_flda_update_kappa(int64_t V, double* kappa, double* kappa_old, double* kappa_temp);       /* :138-142 */
double orc_flda_update_eta(int64_t M, const int64_t* doc_ptr, const int32_t* counts, const double* tau);   /* :122-124 */
double orc_flda_update_elbo(int64_t M, int64_t V, int64_t K,
                            const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                            int64_t d0, int64_t d1, double eta, const double* alpha, const double* kappa,
                            const double* beta, const double* beta_old, const double* gamma, const double* Elogtheta,
                            const double* Elogtheta_old, const double* tau, const double* tau_old);   /* :108-118 */


/* ---- filtered CTM (src/fCTM.jl) ---- */
int orc_fctm_estep(int64_t M, int64_t V, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   int64_t d0, int64_t d1, double eta, const double* kappa, const double* mu, const double* invsigma,
                   const double* beta, double* beta_temp, double* kappa_temp, double* lambda, double* lambda_old,
                   double* vsq, double* logzeta, double* tau, double* tau_old,
                   int niter, double ntol, int viter, double vtol, int32_t* sweeps_out, int64_t* newton_out);
int orc_fctm_estep_omp(int64_t M, int64_t V, int64_t K,
                       const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                       int64_t d0, int64_t d1, double eta, const double* kappa, const double* mu, const double* invsigma,
                       const double* beta, double* beta_temp, double* kappa_temp, double* lambda, double* lambda_old,
                       double* vsq, double* logzeta, double* tau, double* tau_old,
                       int niter, double ntol, int viter, double vtol, int nthreads);
double orc_fctm_update_elbo(int64_t M, int64_t V, int64_t K,
                            const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                            int64_t d0, int64_t d1, double eta, const double* kappa, const double* mu, const double* invsigma,
                            const double* beta, const double* beta_old, const double* lambda, const double* lambda_old,
                            const double* vsq, const double* logzeta, const double* tau, const double* tau_old);

#ifdef __cplusplus
}
#endif
#endif
