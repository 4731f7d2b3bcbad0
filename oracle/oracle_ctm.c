/*
 * oracle_ctm.c -- fp64 restatement of the reference's CPU CTM path (src/CTM.jl).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see tmvb_oracle.h header).
 *
 * Dense K x K linear algebra: the reference calls LinearAlgebra `\` / `inv` / `logdet` on
 * Symmetric matrices (src/CTM.jl:136,:110,:57; LAPACK Bunch-Kaufman).  All matrices involved
 * are symmetric positive definite, so a Cholesky factorisation is used here; any
 * backward-stable solve agrees to ~1e-12 relative in fp64.
 */
#include "tmvb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const double EPS = ORC_EPSILON;

/* In-place lower Cholesky of column-major n x n A (uses lower triangle). Returns 0 if SPD. */
int orc_chol_lower(double* A, int64_t n)
{
    for (int64_t j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int64_t k = 0; k < j; ++k) d -= A[k * n + j] * A[k * n + j];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int64_t i = j + 1; i < n; ++i) {
            double s = A[j * n + i];
            for (int64_t k = 0; k < j; ++k) s -= A[k * n + i] * A[k * n + j];
            A[j * n + i] = s / d;
        }
    }
    return 0;
}

/* Solve L L^T x = b in place (b -> x). L lower, column-major. */
void orc_chol_solve(const double* L, int64_t n, double* b)
{
    for (int64_t i = 0; i < n; ++i) {
        double s = b[i];
        for (int64_t k = 0; k < i; ++k) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int64_t k = i + 1; k < n; ++k) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
}

/* Gaussian elimination with partial pivoting fallback (A overwritten). */
int orc_lu_solve(double* A, int64_t n, double* b)
{
    for (int64_t c = 0; c < n; ++c) {
        int64_t piv = c;
        double mx = fabs(A[c * n + c]);
        for (int64_t r = c + 1; r < n; ++r)
            if (fabs(A[c * n + r]) > mx) { mx = fabs(A[c * n + r]); piv = r; }
        if (mx == 0.0) return 1;
        if (piv != c) {
            for (int64_t k = 0; k < n; ++k) { double t = A[k * n + c]; A[k * n + c] = A[k * n + piv]; A[k * n + piv] = t; }
            double t = b[c]; b[c] = b[piv]; b[piv] = t;
        }
        for (int64_t r = c + 1; r < n; ++r) {
            double f = A[c * n + r] / A[c * n + c];
            if (f == 0.0) continue;
            for (int64_t k = c; k < n; ++k) A[k * n + r] -= f * A[k * n + c];
            b[r] -= f * b[c];
        }
    }
    for (int64_t r = n - 1; r >= 0; --r) {
        double s = b[r];
        for (int64_t k = r + 1; k < n; ++k) s -= A[k * n + r] * b[k];
        b[r] = s / A[r * n + r];
    }
    return 0;
}

typedef struct {
    double *phi, *stage, *g, *ex, *H, *H2, *phic, *rhs;
} ctm_ws;

static int ws_alloc(ctm_ws* w, int64_t K, int64_t mx)
{
    w->phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    w->stage = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    w->g = (double*)malloc(sizeof(double) * (size_t)K);
    w->ex = (double*)malloc(sizeof(double) * (size_t)K);
    w->H = (double*)malloc(sizeof(double) * (size_t)(K * K));
    w->H2 = (double*)malloc(sizeof(double) * (size_t)(K * K));
    w->phic = (double*)malloc(sizeof(double) * (size_t)K);
    w->rhs = (double*)malloc(sizeof(double) * (size_t)K);
    return !(w->phi && w->stage && w->g && w->ex && w->H && w->H2 && w->phic && w->rhs);
}
static void ws_free(ctm_ws* w)
{
    free(w->phi); free(w->stage); free(w->g); free(w->ex); free(w->H); free(w->H2); free(w->phic); free(w->rhs);
}

/* update_phi!  src/CTM.jl:175-178 : additive_logistic(log.(beta[:,terms]) .+ lambda, dims=1)
 * (src/utils.jl:114-122: subtract column max, exp, normalise) */
static void ctm_phi(int64_t K, int64_t Nd, const int32_t* terms, const double* beta,
                    const double* lambda_d, double* phi)
{
    for (int64_t n = 0; n < Nd; ++n) {
        const double* bcol = beta + (int64_t)terms[n] * K;
        double* pc = phi + n * K;
        double mx = -INFINITY;
        for (int64_t i = 0; i < K; ++i) { pc[i] = log(bcol[i]) + lambda_d[i]; if (pc[i] > mx) mx = pc[i]; }
        double s = 0.0;
        for (int64_t i = 0; i < K; ++i) { pc[i] = exp(pc[i] - mx); s += pc[i]; }
        for (int64_t i = 0; i < K; ++i) pc[i] /= s;
    }
}

/* One document: the sweep loop of src/CTM.jl:195-203.  Returns sweeps; *newton accumulates
 * lambda-Newton steps. */
static int ctm_doc(int64_t K, int64_t Nd, const int32_t* terms, const int32_t* counts,
                   const double* mu, const double* invsigma, const double* beta,
                   double* lam, double* lam_old, double* vsq, double* logzeta,
                   int niter, double ntol, int viter, double vtol, ctm_ws* w, int64_t* newton)
{
    double Cd = 0.0;
    for (int64_t n = 0; n < Nd; ++n) Cd += (double)counts[n];
    int sweeps = 0;
    for (int v = 0; v < viter; ++v) {
        ++sweeps;
        ctm_phi(K, Nd, terms, beta, lam, w->phi);                        /* :196 */
        /* update_logzeta!  :169-171  logsumexp(lambda + 0.5 vsq) */
        {
            double mx = -INFINITY;
            for (int64_t i = 0; i < K; ++i) { double x = lam[i] + 0.5 * vsq[i]; if (x > mx) mx = x; }
            double s = 0.0;
            for (int64_t i = 0; i < K; ++i) s += exp(lam[i] + 0.5 * vsq[i] - mx);
            *logzeta = mx + log(s);
        }
        /* update_vsq!  :146-165 */
        for (int64_t i = 0; i < K; ++i) {
            for (int t = 0; t < niter; ++t) {
                double rho = 1.0;
                double ex = exp(lam[i] + 0.5 * vsq[i] - *logzeta);
                double grad = -0.5 * (invsigma[i * K + i] + Cd * ex - 1.0 / vsq[i]);       /* :150 */
                double ihd = -1.0 / (0.25 * Cd * ex + 0.5 / (vsq[i] * vsq[i]));             /* :151 */
                double p = ihd * grad;
                while (vsq[i] - rho * p <= 0.0) rho *= 0.5;                                 /* :154 */
                vsq[i] -= rho * p;
                if (rho * fabs(grad) < ntol) break;                                         /* :159 */
            }
        }
        for (int64_t i = 0; i < K; ++i) vsq[i] += EPS;                                      /* :164 */
        /* update_lambda!  :129-142 */
        memcpy(lam_old, lam, sizeof(double) * (size_t)K);                                   /* :130 */
        for (int64_t i = 0; i < K; ++i) {
            double acc = 0.0;
            for (int64_t n = 0; n < Nd; ++n) acc += w->phi[n * K + i] * (double)counts[n];
            w->phic[i] = acc;
        }
        for (int t = 0; t < niter; ++t) {
            if (newton) ++*newton;
            for (int64_t i = 0; i < K; ++i) w->ex[i] = exp(lam[i] + 0.5 * vsq[i] - *logzeta);
            double gn2 = 0.0;
            for (int64_t i = 0; i < K; ++i) {
                double acc = 0.0;
                for (int64_t j = 0; j < K; ++j) acc += invsigma[j * K + i] * (mu[j] - lam[j]);
                w->g[i] = acc + w->phic[i] - Cd * w->ex[i];                                 /* :134 */
                gn2 += w->g[i] * w->g[i];
            }
            memcpy(w->H, invsigma, sizeof(double) * (size_t)(K * K));                       /* :135 */
            for (int64_t i = 0; i < K; ++i) w->H[i * K + i] += Cd * w->ex[i];
            memcpy(w->rhs, w->g, sizeof(double) * (size_t)K);
            memcpy(w->H2, w->H, sizeof(double) * (size_t)(K * K));
            if (orc_chol_lower(w->H2, K) == 0) orc_chol_solve(w->H2, K, w->rhs);
            else { memcpy(w->rhs, w->g, sizeof(double) * (size_t)K); orc_lu_solve(w->H, K, w->rhs); }
            for (int64_t i = 0; i < K; ++i) lam[i] += w->rhs[i];                            /* :136 */
            if (sqrt(gn2) < ntol) break;                                                    /* :138 */
        }
        double d2 = 0.0;
        for (int64_t i = 0; i < K; ++i) { double df = lam[i] - lam_old[i]; d2 += df * df; }
        if (sqrt(d2) < vtol) break;                                                         /* :200 */
    }
    return sweeps;
}

/* update_beta!(model, d)  src/CTM.jl:122-125 (same overwrite quirk Q1 as LDA) */
static void ctm_update_beta_doc(int64_t K, int64_t Nd, const int32_t* terms, const int32_t* counts,
                                const double* phi, double* beta_temp, double* stage)
{
    for (int64_t n = 0; n < Nd; ++n) {
        const double* bt = beta_temp + (int64_t)terms[n] * K;
        for (int64_t i = 0; i < K; ++i) stage[n * K + i] = bt[i] + phi[n * K + i] * (double)counts[n];
    }
    for (int64_t n = 0; n < Nd; ++n)
        memcpy(beta_temp + (int64_t)terms[n] * K, stage + n * K, sizeof(double) * (size_t)K);
}

static int64_t max_len(const int64_t* doc_ptr, int64_t d0, int64_t d1)
{
    int64_t mx = 1;
    for (int64_t d = d0; d < d1; ++d) { int64_t n = doc_ptr[d + 1] - doc_ptr[d]; if (n > mx) mx = n; }
    return mx;
}

int orc_ctm_estep(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  int64_t d0, int64_t d1,
                  const double* mu, const double* invsigma, const double* beta, double* beta_temp,
                  double* lambda, double* lambda_old, double* vsq, double* logzeta,
                  int niter, double ntol, int viter, double vtol,
                  int32_t* sweeps_out, int64_t* newton_steps_out)
{
    (void)M; (void)V;
    ctm_ws w;
    int64_t mx = max_len(doc_ptr, d0, d1);
    if (ws_alloc(&w, K, mx)) return -1;
    for (int64_t q = 0; q < K * mx; ++q) w.phi[q] = 1.0 / (double)K;
    for (int64_t d = d0; d < d1; ++d) {
        int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
        int sw = ctm_doc(K, Nd, terms + off, counts + off, mu, invsigma, beta,
                         lambda + d * K, lambda_old + d * K, vsq + d * K, logzeta + d,
                         niter, ntol, viter, vtol, &w, newton_steps_out);
        if (sweeps_out) sweeps_out[d - d0] = sw;
        ctm_update_beta_doc(K, Nd, terms + off, counts + off, w.phi, beta_temp, w.stage);
    }
    ws_free(&w);
    return 0;
}

/* sweeps_out / newton_out (may be NULL): per-document sweep counts and lambda Newton steps, [d - d0] */
int orc_ctm_estep_omp_sw(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  int64_t d0, int64_t d1,
                  const double* mu, const double* invsigma, const double* beta, double* beta_temp,
                  double* lambda, double* lambda_old, double* vsq, double* logzeta,
                  int niter, double ntol, int viter, double vtol, int nthreads,
                  int32_t* sweeps_out, int32_t* newton_out)
{
    (void)M;
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int64_t mx = max_len(doc_ptr, d0, d1);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
        ctm_ws w;
        ws_alloc(&w, K, mx);
        double* bt = (double*)calloc((size_t)(K * V), sizeof(double));
#pragma omp for schedule(dynamic, 16)
        for (int64_t d = d0; d < d1; ++d) {
            int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
            int64_t nst = 0;
            int sw = ctm_doc(K, Nd, terms + off, counts + off, mu, invsigma, beta,
                    lambda + d * K, lambda_old + d * K, vsq + d * K, logzeta + d,
                    niter, ntol, viter, vtol, &w, &nst);
            if (sweeps_out) sweeps_out[d - d0] = sw;
            if (newton_out) newton_out[d - d0] = (int32_t)nst;
            ctm_update_beta_doc(K, Nd, terms + off, counts + off, w.phi, bt, w.stage);
        }
#pragma omp critical
        for (int64_t q = 0; q < K * V; ++q) beta_temp[q] += bt[q];
        free(bt);
        ws_free(&w);
    }
#else
    (void)nthreads;
    (void)newton_out;
    orc_ctm_estep(M, V, K, doc_ptr, terms, counts, d0, d1, mu, invsigma, beta, beta_temp,
                  lambda, lambda_old, vsq, logzeta, niter, ntol, viter, vtol, sweeps_out, NULL);
#endif
    return used;
}

int orc_ctm_estep_omp(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  int64_t d0, int64_t d1,
                  const double* mu, const double* invsigma, const double* beta, double* beta_temp,
                  double* lambda, double* lambda_old, double* vsq, double* logzeta,
                  int niter, double ntol, int viter, double vtol, int nthreads)
{
    return orc_ctm_estep_omp_sw(M, V, K, doc_ptr, terms, counts, d0, d1, mu, invsigma, beta, beta_temp,
                                lambda, lambda_old, vsq, logzeta, niter, ntol, viter, vtol, nthreads, NULL, NULL);
}

/* update_sigma! (src/CTM.jl:108-111) THEN update_mu! (:102-104): sigma uses the previous mu
 * (order at :207-208, quirk Q2). */
int orc_ctm_update_sigma_mu(int64_t M, int64_t K, const double* lambda, const double* vsq,
                            double* mu, double* sigma, double* invsigma)
{
    double Md = (double)M;
    memset(sigma, 0, sizeof(double) * (size_t)(K * K));
    for (int64_t d = 0; d < M; ++d) {
        const double* l = lambda + d * K;
        for (int64_t j = 0; j < K; ++j) {
            double cj = l[j] - mu[j];
            for (int64_t i = 0; i < K; ++i) sigma[j * K + i] += (l[i] - mu[i]) * cj;
        }
        for (int64_t i = 0; i < K; ++i) sigma[i * K + i] += vsq[d * K + i];
    }
    for (int64_t q = 0; q < K * K; ++q) sigma[q] /= Md;
    /* Symmetric(...) reads the upper triangle */
    for (int64_t j = 0; j < K; ++j)
        for (int64_t i = j + 1; i < K; ++i) sigma[j * K + i] = sigma[i * K + j];
    /* invsigma = inv(sigma) via Cholesky */
    double* L = (double*)malloc(sizeof(double) * (size_t)(K * K));
    double* col = (double*)malloc(sizeof(double) * (size_t)K);
    memcpy(L, sigma, sizeof(double) * (size_t)(K * K));
    int bad = orc_chol_lower(L, K);
    if (!bad) {
        for (int64_t j = 0; j < K; ++j) {
            for (int64_t i = 0; i < K; ++i) col[i] = (i == j) ? 1.0 : 0.0;
            orc_chol_solve(L, K, col);
            for (int64_t i = 0; i < K; ++i) invsigma[j * K + i] = col[i];
        }
        /* symmetrise (inv(::Symmetric) returns Symmetric) */
        for (int64_t j = 0; j < K; ++j)
            for (int64_t i = j + 1; i < K; ++i) {
                double a = 0.5 * (invsigma[j * K + i] + invsigma[i * K + j]);
                invsigma[j * K + i] = a; invsigma[i * K + j] = a;
            }
    }
    free(L); free(col);
    /* update_mu! */
    for (int64_t i = 0; i < K; ++i) mu[i] = 0.0;
    for (int64_t d = 0; d < M; ++d)
        for (int64_t i = 0; i < K; ++i) mu[i] += lambda[d * K + i];
    for (int64_t i = 0; i < K; ++i) mu[i] /= Md;
    return bad;
}

/* update_elbo!  src/CTM.jl:89-98 with terms :56-86 */
double orc_ctm_update_elbo(int64_t M, int64_t V, int64_t K,
                           const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                           int64_t d0, int64_t d1,
                           const double* mu, const double* invsigma,
                           const double* beta, const double* beta_old,
                           const double* lambda, const double* lambda_old,
                           const double* vsq, const double* logzeta)
{
    (void)M; (void)V;
    int64_t mx = max_len(doc_ptr, d0, d1);
    double* L = (double*)malloc(sizeof(double) * (size_t)(K * K));
    memcpy(L, invsigma, sizeof(double) * (size_t)(K * K));
    double logdet = NAN;
    if (orc_chol_lower(L, K) == 0) {
        logdet = 0.0;
        for (int64_t i = 0; i < K; ++i) logdet += 2.0 * log(L[i * K + i]);
    }
    double elbo = 0.0;
    /* documents are independent: evaluated document-parallel (OpenMP, one workspace per thread) into ed[d - d0] and then added in
     * document order -- the same sum, bit for bit, as the sequential loop */
    double* ed = (double*)calloc((size_t)(d1 > d0 ? d1 - d0 : 1), sizeof(double));
#pragma omp parallel
    {
    double* phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    double* df = (double*)malloc(sizeof(double) * (size_t)K);
#pragma omp for schedule(dynamic, 16)
    for (int64_t d = d0; d < d1; ++d) {
        int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
        const int32_t* tm = terms + off; const int32_t* ct = counts + off;
        const double* l = lambda + d * K; const double* v = vsq + d * K;
        double lz = logzeta[d];
        double Cd = 0.0;
        for (int64_t n = 0; n < Nd; ++n) Cd += (double)ct[n];
        ctm_phi(K, Nd, tm, beta_old, lambda_old + d * K, phi);                 /* :93 */
        /* Elogpeta :57 */
        double dv = 0.0, q = 0.0;
        for (int64_t i = 0; i < K; ++i) { dv += invsigma[i * K + i] * v[i]; df[i] = l[i] - mu[i]; }
        for (int64_t i = 0; i < K; ++i) {
            double acc = 0.0;
            for (int64_t j = 0; j < K; ++j) acc += invsigma[j * K + i] * df[j];
            q += df[i] * acc;
        }
        double t1 = 0.5 * (logdet - (double)K * log(2.0 * M_PI) - dv - q);
        /* Elogpz :64 */
        double a = 0.0, se = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            double acc = 0.0;
            for (int64_t i = 0; i < K; ++i) acc += phi[n * K + i] * l[i];
            a += acc * (double)ct[n];
        }
        for (int64_t i = 0; i < K; ++i) se += exp(l[i] + 0.5 * v[i] - lz);
        double t2 = a - Cd * (se + lz - 1.0);
        /* Elogpw :71 */
        double t3 = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            const double* bcol = beta + (int64_t)tm[n] * K;
            double acc = 0.0;
            for (int64_t i = 0; i < K; ++i) acc += phi[n * K + i] * log(bcol[i] + EPS);
            t3 += acc * (double)ct[n];
        }
        /* Elogqeta :77 = -entropy(MvNormal(lambda, diagm(vsq))) */
        double slv = 0.0;
        for (int64_t i = 0; i < K; ++i) slv += log(v[i]);
        double t4 = -0.5 * ((double)K * (1.0 + log(2.0 * M_PI)) + slv);
        /* Elogqz :84 */
        double t5 = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            double h = 0.0;
            for (int64_t i = 0; i < K; ++i) { double pv = phi[n * K + i]; if (pv > 0.0) h -= pv * log(pv); }
            t5 -= (double)ct[n] * h;
        }
        ed[d - d0] = t1 + t2 + t3 - t4 - t5;                                   /* :94 */
    }
    free(phi); free(df);
    }
    for (int64_t d = d0; d < d1; ++d) elbo += ed[d - d0];
    free(ed); free(L);
    return elbo;
}

static void ctm_update_beta(int64_t V, int64_t K, double* beta, double* beta_old, double* beta_temp)
{
    /* src/CTM.jl:114-118 */
    memcpy(beta_old, beta, sizeof(double) * (size_t)(K * V));
    for (int64_t i = 0; i < K; ++i) {
        double s = 0.0;
        for (int64_t j = 0; j < V; ++j) s += beta_temp[j * K + i];
        for (int64_t j = 0; j < V; ++j) beta[j * K + i] = beta_temp[j * K + i] / s;
    }
    memset(beta_temp, 0, sizeof(double) * (size_t)(K * V));
}

/* train!  src/CTM.jl:185-217 */
int orc_ctm_train(int64_t M, int64_t V, int64_t K,
                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                  double* mu, double* sigma, double* invsigma, double* beta, double* beta_old,
                  double* lambda, double* lambda_old, double* vsq, double* logzeta, double* elbo,
                  int iter, double tol, int niter, double ntol, int viter, double vtol,
                  int checkelbo, double* elbo_traj)
{
    double* beta_temp = (double*)calloc((size_t)(K * V), sizeof(double));
    if (doc_ptr[M] == doc_ptr[0]) iter = 0;                                    /* :190 */
    if (checkelbo > 0 && checkelbo <= iter)                                    /* :191 */
        *elbo = orc_ctm_update_elbo(M, V, K, doc_ptr, terms, counts, 0, M, mu, invsigma, beta, beta_old,
                                    lambda, lambda_old, vsq, logzeta);
    int done = 0;
    for (int k = 1; k <= iter; ++k) {
        ++done;
        orc_ctm_estep(M, V, K, doc_ptr, terms, counts, 0, M, mu, invsigma, beta, beta_temp,
                      lambda, lambda_old, vsq, logzeta, niter, ntol, viter, vtol, NULL, NULL);
        ctm_update_beta(V, K, beta, beta_old, beta_temp);                      /* :206 */
        orc_ctm_update_sigma_mu(M, K, lambda, vsq, mu, sigma, invsigma);       /* :207-208 */
        if (elbo_traj) elbo_traj[k - 1] = NAN;
        if (checkelbo > 0 && (k % checkelbo) == 0) {
            double e_new = orc_ctm_update_elbo(M, V, K, doc_ptr, terms, counts, 0, M, mu, invsigma,
                                               beta, beta_old, lambda, lambda_old, vsq, logzeta);
            double delta = -(*elbo - e_new);
            *elbo = e_new;
            if (elbo_traj) elbo_traj[k - 1] = e_new;
            if (delta < tol) break;
        }
    }
    free(beta_temp);
    return done;
}
