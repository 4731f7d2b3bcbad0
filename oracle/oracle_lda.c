/*
 * oracle_lda.c -- fp64 restatement of the reference's CPU LDA path (src/LDA.jl).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see tmvb_oracle.h header).
 * Every function cites the reference lines it follows.
 */
#include "tmvb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const double EPS = ORC_EPSILON;

/* src/LDA.jl:171-178: for v in 1:viter { update_phi!; update_gamma!; update_Elogtheta!;
 * break if norm(Elogtheta - Elogtheta_old) < vtol }.  */
int orc_lda_doc_sweeps(int64_t K, int64_t Nd, const int32_t* terms, const int32_t* counts,
                       const double* alpha, const double* beta,
                       double* gamma_d, double* Elogtheta_d, double* Elogtheta_old_d,
                       double* phi, int viter, double vtol)
{
    int sweeps = 0;
    double* expE = (double*)malloc(sizeof(double) * (size_t)K);
    for (int v = 0; v < viter; ++v) {
        ++sweeps;
        /* update_phi!  src/LDA.jl:150-154:
         *   @positive phi = beta[:,terms] .* exp.(Elogtheta[d]);  phi ./= sum(phi, dims=1) */
        for (int64_t i = 0; i < K; ++i) expE[i] = exp(Elogtheta_d[i]);
        for (int64_t n = 0; n < Nd; ++n) {
            const double* bcol = beta + (int64_t)terms[n] * K;
            double* pcol = phi + n * K;
            double s = 0.0;
            for (int64_t i = 0; i < K; ++i) {
                pcol[i] = EPS + bcol[i] * expE[i];
                s += pcol[i];
            }
            for (int64_t i = 0; i < K; ++i) pcol[i] /= s;
        }
        /* update_gamma!  src/LDA.jl:143-146:  @positive gamma[d] = alpha + phi * counts */
        for (int64_t i = 0; i < K; ++i) {
            double acc = 0.0;
            for (int64_t n = 0; n < Nd; ++n) acc += phi[n * K + i] * (double)counts[n];
            gamma_d[i] = EPS + (alpha[i] + acc);
        }
        /* update_Elogtheta!  src/LDA.jl:136-139 */
        double gsum = 0.0;
        for (int64_t i = 0; i < K; ++i) gsum += gamma_d[i];
        double dg = orc_digamma(gsum);
        double dist2 = 0.0;
        for (int64_t i = 0; i < K; ++i) {
            Elogtheta_old_d[i] = Elogtheta_d[i];
            Elogtheta_d[i] = orc_digamma(gamma_d[i]) - dg;
            double df = Elogtheta_d[i] - Elogtheta_old_d[i];
            dist2 += df * df;
        }
        /* src/LDA.jl:175 */
        if (sqrt(dist2) < vtol) break;
    }
    free(expE);
    return sweeps;
}

/* update_beta!(model, d)  src/LDA.jl:129-132:  beta_temp[:,terms] += phi .* counts'
 * Julia evaluates the right-hand side from the pre-update columns and then assigns column by
 * column, so a term id repeated inside one document is overwritten, not accumulated (quirk Q1).
 * Reproduced here by staging the right-hand side first. */
static void lda_update_beta_doc(int64_t K, int64_t Nd, const int32_t* terms, const int32_t* counts,
                                const double* phi, double* beta_temp, double* stage)
{
    for (int64_t n = 0; n < Nd; ++n) {
        const double* bt = beta_temp + (int64_t)terms[n] * K;
        for (int64_t i = 0; i < K; ++i) stage[n * K + i] = bt[i] + phi[n * K + i] * (double)counts[n];
    }
    for (int64_t n = 0; n < Nd; ++n) {
        double* bt = beta_temp + (int64_t)terms[n] * K;
        memcpy(bt, stage + n * K, sizeof(double) * (size_t)K);
    }
}

static int64_t max_doc_len(const int64_t* doc_ptr, int64_t d0, int64_t d1)
{
    int64_t mx = 1;
    for (int64_t d = d0; d < d1; ++d) {
        int64_t n = doc_ptr[d + 1] - doc_ptr[d];
        if (n > mx) mx = n;
    }
    return mx;
}

/* src/LDA.jl:170-180 over documents [d0,d1) */
int orc_lda_estep(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  int64_t d0, int64_t d1,
                  const double* alpha, const double* beta, double* beta_temp,
                  double* gamma, double* Elogtheta, double* Elogtheta_old,
                  int viter, double vtol, int32_t* sweeps_out)
{
    (void)M; (void)V;
    int64_t mx = max_doc_len(doc_ptr, d0, d1);
    double* phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    double* stage = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    if (!phi || !stage) { free(phi); free(stage); return -1; }
    /* the reference keeps ONE phi workspace (src/LDA.jl:41): with viter == 0 the stale phi of
     * the previous document would be scattered; we initialise to 1/K as the constructor does. */
    for (int64_t q = 0; q < K * mx; ++q) phi[q] = 1.0 / (double)K;
    for (int64_t d = d0; d < d1; ++d) {
        int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
        int sw = orc_lda_doc_sweeps(K, Nd, terms + off, counts + off, alpha, beta,
                                    gamma + d * K, Elogtheta + d * K, Elogtheta_old + d * K,
                                    phi, viter, vtol);
        if (sweeps_out) sweeps_out[d - d0] = sw;
        lda_update_beta_doc(K, Nd, terms + off, counts + off, phi, beta_temp, stage);
    }
    free(phi); free(stage);
    return 0;
}

/* Per-thread private statistics of the OpenMP E-step (K x V doubles each).  They are allocated ONCE and kept across
 * calls: allocating and page-faulting nthreads x K x V x 8 bytes (648 MB at NSF K=50, 64 threads) in every iteration
 * was a fixed cost several times the arithmetic (round-1 review), which made the CPU baseline look slower than it is. */
static double** g_pool = NULL;
static int g_pool_n = 0;
static int64_t g_pool_len = 0;

void orc_omp_pool_free(void)
{
    for (int t = 0; t < g_pool_n; ++t) free(g_pool[t]);
    free(g_pool);
    g_pool = NULL; g_pool_n = 0; g_pool_len = 0;
}

static int pool_reserve(int nt, int64_t len)
{
    if (g_pool && g_pool_n >= nt && g_pool_len >= len) return 0;
    orc_omp_pool_free();
    g_pool = (double**)calloc((size_t)nt, sizeof(double*));
    if (!g_pool) return -1;
    g_pool_n = nt; g_pool_len = len;
    return 0;                            /* the buffers themselves are first-touched by their own threads */
}

/* sweeps_out (may be NULL): sweeps_out[d - d0] = the sweeps document d ran, as orc_lda_estep reports them (each document is
 * written by the one thread that owns it) -- the full-size parity checks compare per document. */
int orc_lda_estep_omp_sw(int64_t M, int64_t V, int64_t K,
                      const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                      int64_t d0, int64_t d1,
                      const double* alpha, const double* beta, double* beta_temp,
                      double* gamma, double* Elogtheta, double* Elogtheta_old,
                      int viter, double vtol, int nthreads, int32_t* sweeps_out)
{
    (void)M; (void)V;
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int64_t mx = max_doc_len(doc_ptr, d0, d1);
    int nt = omp_get_max_threads();
    if (pool_reserve(nt, K * V)) return -1;
    double** bts = g_pool;
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
        int tid = omp_get_thread_num();
        double* phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
        if (!bts[tid]) bts[tid] = (double*)malloc(sizeof(double) * (size_t)(K * V));   /* first call only */
        double* bt = bts[tid];
        memset(bt, 0, sizeof(double) * (size_t)(K * V));
#pragma omp barrier
#pragma omp for schedule(dynamic, 16)
        for (int64_t d = d0; d < d1; ++d) {
            int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
            int sw = orc_lda_doc_sweeps(K, Nd, terms + off, counts + off, alpha, beta,
                               gamma + d * K, Elogtheta + d * K, Elogtheta_old + d * K,
                               phi, viter, vtol);
            if (sweeps_out) sweeps_out[d - d0] = sw;
            for (int64_t n = 0; n < Nd; ++n) {
                double* col = bt + (int64_t)terms[off + n] * K;
                double c = (double)counts[off + n];
                for (int64_t i = 0; i < K; ++i) col[i] += phi[n * K + i] * c;
            }
        }
        free(phi);
        /* parallel reduction of the private statistics over the K*V entries (threads that joined the team) */
        const int team = omp_get_num_threads();
#pragma omp for schedule(static)
        for (int64_t q = 0; q < K * V; ++q) {
            double sacc = 0.0;
            for (int t = 0; t < team; ++t) sacc += bts[t][q];
            beta_temp[q] += sacc;
        }
    }
#else
    (void)nthreads;
    orc_lda_estep(M, V, K, doc_ptr, terms, counts, d0, d1, alpha, beta, beta_temp,
                  gamma, Elogtheta, Elogtheta_old, viter, vtol, sweeps_out);
#endif
    return used;
}

int orc_lda_estep_omp(int64_t M, int64_t V, int64_t K,
                      const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                      int64_t d0, int64_t d1,
                      const double* alpha, const double* beta, double* beta_temp,
                      double* gamma, double* Elogtheta, double* Elogtheta_old,
                      int viter, double vtol, int nthreads)
{
    return orc_lda_estep_omp_sw(M, V, K, doc_ptr, terms, counts, d0, d1, alpha, beta, beta_temp,
                                gamma, Elogtheta, Elogtheta_old, viter, vtol, nthreads, NULL);
}

/* update_beta!(model)  src/LDA.jl:121-125 */
void orc_lda_update_beta(int64_t V, int64_t K, double* beta, double* beta_old, double* beta_temp)
{
    memcpy(beta_old, beta, sizeof(double) * (size_t)(K * V));
    for (int64_t i = 0; i < K; ++i) {
        double s = 0.0;
        for (int64_t j = 0; j < V; ++j) s += beta_temp[j * K + i];
        for (int64_t j = 0; j < V; ++j) beta[j * K + i] = beta_temp[j * K + i] / s;
    }
    memset(beta_temp, 0, sizeof(double) * (size_t)(K * V));
}

void orc_lda_elogtheta_sum(int64_t M, int64_t K, const double* Elogtheta, double* out)
{
    for (int64_t i = 0; i < K; ++i) out[i] = 0.0;
    for (int64_t d = 0; d < M; ++d)
        for (int64_t i = 0; i < K; ++i) out[i] += Elogtheta[d * K + i];
}

/* update_alpha!  src/LDA.jl:97-118 */
int orc_lda_update_alpha(int64_t K, int64_t Mtot, const double* Elogtheta_sum,
                         double* alpha, int niter, double ntol)
{
    double* grad = (double*)malloc(sizeof(double) * (size_t)K);
    double* hinv = (double*)malloc(sizeof(double) * (size_t)K);
    double* p = (double*)malloc(sizeof(double) * (size_t)K);
    double Md = (double)Mtot;
    double nu = (double)K;            /* :100 */
    int it = 0;
    for (int t = 0; t < niter; ++t) {
        ++it;
        double rho = 1.0;
        double asum = 0.0;
        for (int64_t i = 0; i < K; ++i) asum += alpha[i];
        double dgs = orc_digamma(asum);
        double gh = 0.0, hs = 0.0, gn2 = 0.0;
        for (int64_t i = 0; i < K; ++i) {
            grad[i] = nu / alpha[i] + Md * (dgs - orc_digamma(alpha[i])) + Elogtheta_sum[i];   /* :103 */
            hinv[i] = -1.0 / (Md * orc_trigamma(alpha[i]) + nu / (alpha[i] * alpha[i]));         /* :104 */
            gh += grad[i] * hinv[i];
            hs += hinv[i];
            gn2 += grad[i] * grad[i];
        }
        double c = gh / (1.0 / (Md * orc_trigamma(asum)) + hs);                                  /* :105 */
        for (int64_t i = 0; i < K; ++i) p[i] = (grad[i] - c) * hinv[i];
        for (;;) {                                                                               /* :107-109 */
            double mn = INFINITY;
            for (int64_t i = 0; i < K; ++i) {
                double a = alpha[i] - rho * p[i];
                if (a < mn) mn = a;
            }
            if (mn < 0.0) rho *= 0.5; else break;
        }
        /* @finite alpha -= rho*p  (src/macros.jl:53-54): alpha = sign(alpha)*min(|alpha-rho p|, floatmax) */
        for (int64_t i = 0; i < K; ++i) {
            double a = fabs(alpha[i] - rho * p[i]);
            if (a > 1.7976931348623157e308) a = 1.7976931348623157e308;
            double sg = (alpha[i] > 0) ? 1.0 : ((alpha[i] < 0) ? -1.0 : 0.0);
            alpha[i] = sg * a;
        }
        if ((rho * sqrt(gn2) < ntol) && (nu / (double)K < ntol)) break;                          /* :112 */
        nu *= 0.5;                                                                               /* :115 */
    }
    for (int64_t i = 0; i < K; ++i) alpha[i] += EPS;                                             /* :117 */
    free(grad); free(hinv); free(p);
    return it;
}

/* update_elbo!  src/LDA.jl:83-93 with the five terms of :50-80 */
double orc_lda_update_elbo(int64_t M, int64_t V, int64_t K,
                           const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                           int64_t d0, int64_t d1,
                           const double* alpha, const double* beta, const double* beta_old,
                           const double* gamma, const double* Elogtheta,
                           const double* Elogtheta_old)
{
    (void)M; (void)V;
    int64_t mx = max_doc_len(doc_ptr, d0, d1);
    double asum = 0.0, lgsum = 0.0;
    for (int64_t i = 0; i < K; ++i) { asum += alpha[i]; lgsum += orc_lgamma(alpha[i]); }
    double cst = orc_finite(orc_lgamma(asum)) - orc_finite(lgsum);   /* :51 */
    double elbo = 0.0;
    /* documents are independent: evaluated document-parallel (OpenMP, one phi workspace per thread) into ed[d - d0] and then
     * added in document order -- the same sum, bit for bit, as the sequential loop */
    double* ed = (double*)calloc((size_t)(d1 > d0 ? d1 - d0 : 1), sizeof(double));
#pragma omp parallel
    {
    double* phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    double* expE = (double*)malloc(sizeof(double) * (size_t)K);
#pragma omp for schedule(dynamic, 16)
    for (int64_t d = d0; d < d1; ++d) {
        int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
        const int32_t* tm = terms + off;
        const int32_t* ct = counts + off;
        const double* El = Elogtheta + d * K;
        const double* Elo = Elogtheta_old + d * K;
        const double* g = gamma + d * K;
        /* :87-88 phi rebuilt from beta_old / Elogtheta_old */
        for (int64_t i = 0; i < K; ++i) expE[i] = exp(Elo[i]);
        for (int64_t n = 0; n < Nd; ++n) {
            const double* bcol = beta_old + (int64_t)tm[n] * K;
            double s = 0.0;
            for (int64_t i = 0; i < K; ++i) { phi[n * K + i] = EPS + bcol[i] * expE[i]; s += phi[n * K + i]; }
            for (int64_t i = 0; i < K; ++i) phi[n * K + i] /= s;
        }
        /* Elogptheta :51 */
        double t1 = cst;
        for (int64_t i = 0; i < K; ++i) t1 += (alpha[i] - 1.0) * El[i];
        /* Elogpz :58  dot(phi*counts, Elogtheta) */
        double t2 = 0.0;
        for (int64_t i = 0; i < K; ++i) {
            double acc = 0.0;
            for (int64_t n = 0; n < Nd; ++n) acc += phi[n * K + i] * (double)ct[n];
            t2 += acc * El[i];
        }
        /* Elogpw :65  sum(phi .* log.(beta[:,terms] .+ EPS) * counts) */
        double t3 = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            const double* bcol = beta + (int64_t)tm[n] * K;
            double acc = 0.0;
            for (int64_t i = 0; i < K; ++i) acc += phi[n * K + i] * log(bcol[i] + EPS);
            t3 += acc * (double)ct[n];
        }
        /* Elogqtheta :72 = -entropy(Dirichlet(gamma)) with the override of src/utils.jl:163-180 */
        double t4;
        if (K == 1) {
            t4 = -0.0;
        } else {
            double g0 = 0.0, lmnB = 0.0;
            for (int64_t i = 0; i < K; ++i) { g0 += g[i]; lmnB += orc_lgamma(g[i]); }
            lmnB -= orc_lgamma(g0);
            double en = lmnB + (g0 - (double)K) * orc_digamma(g0);
            for (int64_t i = 0; i < K; ++i) en -= (g[i] - 1.0) * orc_digamma(g[i]);
            t4 = -en;
        }
        /* Elogqz :78 = -sum_n c_n * entropy(Categorical(phi[:,n])) */
        double t5 = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            double h = 0.0;
            for (int64_t i = 0; i < K; ++i) {
                double pv = phi[n * K + i];
                if (pv > 0.0) h -= pv * log(pv);
            }
            t5 -= (double)ct[n] * h;
        }
        ed[d - d0] = t1 + t2 + t3 - t4 - t5;   /* :89 */
    }
    free(phi); free(expE);
    }
    for (int64_t d = d0; d < d1; ++d) elbo += ed[d - d0];
    free(ed);
    return elbo;
}

/* train!  src/LDA.jl:161-191; check_elbo!  src/modelutils.jl:574-585 */
int orc_lda_train(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  double* alpha, double* beta, double* beta_old,
                  double* gamma, double* Elogtheta, double* Elogtheta_old, double* elbo,
                  int iter, double tol, int niter, double ntol, int viter, double vtol,
                  int checkelbo, double* elbo_traj, int64_t* sweep_hist)
{
    double* beta_temp = (double*)calloc((size_t)(K * V), sizeof(double));
    double* Esum = (double*)malloc(sizeof(double) * (size_t)K);
    int32_t* sweeps = (int32_t*)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
    /* :166 all documents empty => iter = 0 */
    if (doc_ptr[M] == doc_ptr[0]) iter = 0;
    /* :167 */
    if (checkelbo > 0 && checkelbo <= iter)
        *elbo = orc_lda_update_elbo(M, V, K, doc_ptr, terms, counts, 0, M, alpha, beta, beta_old,
                                    gamma, Elogtheta, Elogtheta_old);
    int done = 0;
    for (int k = 1; k <= iter; ++k) {
        ++done;
        orc_lda_estep(M, V, K, doc_ptr, terms, counts, 0, M, alpha, beta, beta_temp,
                      gamma, Elogtheta, Elogtheta_old, viter, vtol, sweeps);
        if (sweep_hist)
            for (int64_t d = 0; d < M; ++d) sweep_hist[sweeps[d]]++;
        orc_lda_update_beta(V, K, beta, beta_old, beta_temp);              /* :181 */
        orc_lda_elogtheta_sum(M, K, Elogtheta, Esum);                      /* :98 */
        orc_lda_update_alpha(K, M, Esum, alpha, niter, ntol);              /* :182 */
        if (elbo_traj) elbo_traj[k - 1] = NAN;
        if (checkelbo > 0 && (k % checkelbo) == 0) {                       /* modelutils.jl:575 */
            double e_new = orc_lda_update_elbo(M, V, K, doc_ptr, terms, counts, 0, M, alpha, beta,
                                               beta_old, gamma, Elogtheta, Elogtheta_old);
            double delta = -(*elbo - e_new);                               /* :577 */
            *elbo = e_new;
            if (elbo_traj) elbo_traj[k - 1] = e_new;
            if (delta < tol) break;                                        /* :580 (signed) */
        }
    }
    free(beta_temp); free(Esum); free(sweeps);
    return done;
}
