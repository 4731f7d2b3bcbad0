/*
 * oracle_fctm.c -- fp64 restatement of the reference's CPU filtered-CTM path (src/fCTM.jl).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see tmvb_oracle.h header).
 * fCTM = CTM (logistic-normal topic proportions) + the per-token Bernoulli switch tau_n / background distribution
 * kappa of fLDA.  Differences from src/CTM.jl that matter here:
 *   - phi = softmax_i(tau_n log(beta + eps) + lambda_i)                      (:216-219; CTM has no eps and no tau)
 *   - sweep order phi, tau, logzeta, LAMBDA, then VSQ                        (:236-241; CTM: phi, logzeta, vsq, lambda)
 *   - update_eta! is commented out in train! (:253): eta keeps its value     (0.5 from the constructor)
 *   - update_sigma! then update_mu! as in CTM (sigma uses the previous mu)   (:251-252)
 * tau / tau_old: flat [nnz] arrays in CSR token order.
 */
#include "tmvb_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_chol_lower(double* A, int64_t n);
void orc_chol_solve(const double* L, int64_t n, double* b);
int orc_lu_solve(double* A, int64_t n, double* b);

static const double EPS = ORC_EPSILON;

static int64_t fctm_max_len(const int64_t* doc_ptr, int64_t d0, int64_t d1)
{
    int64_t mx = 1;
    for (int64_t d = d0; d < d1; ++d) if (doc_ptr[d + 1] - doc_ptr[d] > mx) mx = doc_ptr[d + 1] - doc_ptr[d];
    return mx;
}

/* update_phi!  :216-219 */
static void fctm_phi(int64_t K, int64_t Nd, const int32_t* terms, const double* beta, const double* tau_d,
                     const double* lambda_d, double* phi)
{
    for (int64_t n = 0; n < Nd; ++n) {
        const double* bcol = beta + (int64_t)terms[n] * K;
        double* pc = phi + n * K;
        double mx = -INFINITY;
        for (int64_t i = 0; i < K; ++i) { pc[i] = tau_d[n] * log(bcol[i] + EPS) + lambda_d[i]; if (pc[i] > mx) mx = pc[i]; }
        double s = 0.0;
        for (int64_t i = 0; i < K; ++i) { pc[i] = exp(pc[i] - mx); s += pc[i]; }
        for (int64_t i = 0; i < K; ++i) pc[i] /= s;
    }
}

/* One document: the sweep loop of src/fCTM.jl:235-245.  ws: K*K*2 + 4K doubles. */
static int fctm_doc(int64_t K, int64_t Nd, const int32_t* terms, const int32_t* counts, double eta, const double* kappa,
                    const double* mu, const double* invsigma, const double* beta, double* lam, double* lam_old, double* vsq,
                    double* logzeta, double* tau_d, double* tau_old_d, double* phi, double* ws,
                    int niter, double ntol, int viter, double vtol, int64_t* newton)
{
    double* H = ws; double* H2 = H + K * K; double* g = H2 + K * K; double* ex = g + K; double* phic = ex + K; double* rhs = phic + K;
    double Cd = 0.0;
    for (int64_t n = 0; n < Nd; ++n) Cd += (double)counts[n];
    int sweeps = 0;
    for (int v = 0; v < viter; ++v) {
        ++sweeps;
        fctm_phi(K, Nd, terms, beta, tau_d, lam, phi);                                         /* :236 */
        /* update_tau!  :208-213 */
        for (int64_t n = 0; n < Nd; ++n) {
            const double* bcol = beta + (int64_t)terms[n] * K;
            tau_old_d[n] = tau_d[n];
            double prod = 1.0;
            for (int64_t i = 0; i < K; ++i) prod *= pow(bcol[i], -phi[n * K + i]);
            tau_d[n] = eta / (EPS + (eta + (1.0 - eta) * (kappa[terms[n]] * prod)));
        }
        /* update_logzeta!  :202-204 */
        {
            double mx = -INFINITY;
            for (int64_t i = 0; i < K; ++i) { double x = lam[i] + 0.5 * vsq[i]; if (x > mx) mx = x; }
            double s = 0.0;
            for (int64_t i = 0; i < K; ++i) s += exp(lam[i] + 0.5 * vsq[i] - mx);
            *logzeta = mx + log(s);
        }
        /* update_lambda!  :162-176 */
        memcpy(lam_old, lam, sizeof(double) * (size_t)K);
        for (int64_t i = 0; i < K; ++i) {
            double acc = 0.0;
            for (int64_t n = 0; n < Nd; ++n) acc += phi[n * K + i] * (double)counts[n];
            phic[i] = acc;
        }
        for (int t = 0; t < niter; ++t) {
            if (newton) ++*newton;
            for (int64_t i = 0; i < K; ++i) ex[i] = exp(lam[i] + 0.5 * vsq[i] - *logzeta);
            double gn2 = 0.0;
            for (int64_t i = 0; i < K; ++i) {
                double acc = 0.0;
                for (int64_t j = 0; j < K; ++j) acc += invsigma[j * K + i] * (mu[j] - lam[j]);
                g[i] = acc + phic[i] - Cd * ex[i];                                             /* :167 */
                gn2 += g[i] * g[i];
            }
            memcpy(H, invsigma, sizeof(double) * (size_t)(K * K));                             /* :168 */
            for (int64_t i = 0; i < K; ++i) H[i * K + i] += Cd * ex[i];
            memcpy(rhs, g, sizeof(double) * (size_t)K);
            memcpy(H2, H, sizeof(double) * (size_t)(K * K));
            if (orc_chol_lower(H2, K) == 0) orc_chol_solve(H2, K, rhs);
            else { memcpy(rhs, g, sizeof(double) * (size_t)K); orc_lu_solve(H, K, rhs); }
            for (int64_t i = 0; i < K; ++i) lam[i] += rhs[i];                                  /* :169 */
            if (sqrt(gn2) < ntol) break;                                                       /* :171 */
        }
        /* update_vsq!  :180-198 (after lambda in this model) */
        for (int64_t i = 0; i < K; ++i) {
            for (int t = 0; t < niter; ++t) {
                double rho = 1.0;
                double e1 = exp(lam[i] + 0.5 * vsq[i] - *logzeta);
                double grad = -0.5 * (invsigma[i * K + i] + Cd * e1 - 1.0 / vsq[i]);
                double ihd = -1.0 / (0.25 * Cd * e1 + 0.5 / (vsq[i] * vsq[i]));
                double p = ihd * grad;
                while (vsq[i] - rho * p <= 0.0) rho *= 0.5;
                vsq[i] -= rho * p;
                if (rho * fabs(grad) < ntol) break;
            }
        }
        for (int64_t i = 0; i < K; ++i) vsq[i] += EPS;                                         /* :197 */
        double d2 = 0.0;
        for (int64_t i = 0; i < K; ++i) { double df = lam[i] - lam_old[i]; d2 += df * df; }
        if (sqrt(d2) < vtol) break;                                                            /* :242 */
    }
    return sweeps;
}

/* E-step over documents [d0, d1): sweeps + update_beta!(model, d) (:155-158) + update_kappa!(model, d) (:141-144) */
int orc_fctm_estep(int64_t M, int64_t V, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   int64_t d0, int64_t d1, double eta, const double* kappa, const double* mu, const double* invsigma,
                   const double* beta, double* beta_temp, double* kappa_temp, double* lambda, double* lambda_old,
                   double* vsq, double* logzeta, double* tau, double* tau_old,
                   int niter, double ntol, int viter, double vtol, int32_t* sweeps_out, int64_t* newton_out)
{
    (void)M; (void)V;
    int64_t mx = fctm_max_len(doc_ptr, d0, d1);
    double* phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    double* stage = (double*)malloc(sizeof(double) * (size_t)((K + 1) * mx));
    double* ws = (double*)malloc(sizeof(double) * (size_t)(2 * K * K + 4 * K));
    if (!phi || !stage || !ws) { free(phi); free(stage); free(ws); return -1; }
    for (int64_t q = 0; q < K * mx; ++q) phi[q] = 1.0 / (double)K;
    int64_t newton = 0;
    for (int64_t d = d0; d < d1; ++d) {
        int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
        const int32_t* tm = terms + off; const int32_t* ct = counts + off;
        int sw = fctm_doc(K, Nd, tm, ct, eta, kappa, mu, invsigma, beta, lambda + d * K, lambda_old + d * K, vsq + d * K,
                          logzeta + d, tau + off, tau_old + off, phi, ws, niter, ntol, viter, vtol, &newton);
        if (sweeps_out) sweeps_out[d - d0] = sw;
        for (int64_t n = 0; n < Nd; ++n) {                                                     /* quirk Q1: staged overwrite */
            const double* bt = beta_temp + (int64_t)tm[n] * K;
            const double wn = tau[off + n] * (double)ct[n];
            for (int64_t i = 0; i < K; ++i) stage[n * (K + 1) + i] = bt[i] + phi[n * K + i] * wn;
            stage[n * (K + 1) + K] = kappa_temp[tm[n]] + (1.0 - tau[off + n]) * (double)ct[n];
        }
        for (int64_t n = 0; n < Nd; ++n) {
            memcpy(beta_temp + (int64_t)tm[n] * K, stage + n * (K + 1), sizeof(double) * (size_t)K);
            kappa_temp[tm[n]] = stage[n * (K + 1) + K];
        }
    }
    if (newton_out) *newton_out = newton;
    free(phi); free(stage); free(ws);
    return 0;
}

/* The same E-step, document-parallel with OpenMP (cpu_baseline only; private statistics ADD duplicate ids: condensed corpora) */
int orc_fctm_estep_omp(int64_t M, int64_t V, int64_t K,
                       const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                       int64_t d0, int64_t d1, double eta, const double* kappa, const double* mu, const double* invsigma,
                       const double* beta, double* beta_temp, double* kappa_temp, double* lambda, double* lambda_old,
                       double* vsq, double* logzeta, double* tau, double* tau_old,
                       int niter, double ntol, int viter, double vtol, int nthreads)
{
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int64_t mx = fctm_max_len(doc_ptr, d0, d1);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
        double* phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
        double* ws = (double*)malloc(sizeof(double) * (size_t)(2 * K * K + 4 * K));
        double* bt = (double*)calloc((size_t)(K * V), sizeof(double));
        double* kt = (double*)calloc((size_t)V, sizeof(double));
        for (int64_t q = 0; q < K * mx; ++q) phi[q] = 1.0 / (double)K;
        int64_t newton = 0;
#pragma omp for schedule(dynamic, 16)
        for (int64_t d = d0; d < d1; ++d) {
            int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
            const int32_t* tm = terms + off; const int32_t* ct = counts + off;
            fctm_doc(K, Nd, tm, ct, eta, kappa, mu, invsigma, beta, lambda + d * K, lambda_old + d * K, vsq + d * K,
                     logzeta + d, tau + off, tau_old + off, phi, ws, niter, ntol, viter, vtol, &newton);
            for (int64_t n = 0; n < Nd; ++n) {
                double* col = bt + (int64_t)tm[n] * K;
                const double wn = tau[off + n] * (double)ct[n];
                for (int64_t i = 0; i < K; ++i) col[i] += phi[n * K + i] * wn;
                kt[tm[n]] += (1.0 - tau[off + n]) * (double)ct[n];
            }
        }
#pragma omp critical
        {
            for (int64_t q = 0; q < K * V; ++q) beta_temp[q] += bt[q];
            for (int64_t j = 0; j < V; ++j) kappa_temp[j] += kt[j];
        }
        free(phi); free(ws); free(bt); free(kt);
    }
#else
    (void)nthreads;
    orc_fctm_estep(M, V, K, doc_ptr, terms, counts, d0, d1, eta, kappa, mu, invsigma, beta, beta_temp, kappa_temp, lambda,
                   lambda_old, vsq, logzeta, tau, tau_old, niter, ntol, viter, vtol, NULL, NULL);
#endif
    (void)M;
    return used;
}

/* update_elbo!  :105-115 with the seven terms of :68-102 */
double orc_fctm_update_elbo(int64_t M, int64_t V, int64_t K,
                            const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                            int64_t d0, int64_t d1, double eta, const double* kappa, const double* mu, const double* invsigma,
                            const double* beta, const double* beta_old, const double* lambda, const double* lambda_old,
                            const double* vsq, const double* logzeta, const double* tau, const double* tau_old)
{
    (void)M; (void)V;
    int64_t mx = fctm_max_len(doc_ptr, d0, d1);
    double* phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    double* L = (double*)malloc(sizeof(double) * (size_t)(K * K));
    double* df = (double*)malloc(sizeof(double) * (size_t)K);
    memcpy(L, invsigma, sizeof(double) * (size_t)(K * K));
    double logdet = NAN;
    if (orc_chol_lower(L, K) == 0) {
        logdet = 0.0;
        for (int64_t i = 0; i < K; ++i) logdet += 2.0 * log(L[i * K + i]);
    }
    double elbo = 0.0;
    for (int64_t d = d0; d < d1; ++d) {
        int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
        const int32_t* tm = terms + off; const int32_t* ct = counts + off;
        const double* l = lambda + d * K; const double* v = vsq + d * K;
        const double lz = logzeta[d];
        double Cd = 0.0, a = 0.0;
        for (int64_t n = 0; n < Nd; ++n) { Cd += (double)ct[n]; a += tau[off + n] * (double)ct[n]; }
        fctm_phi(K, Nd, tm, beta_old, tau_old + off, lambda_old + d * K, phi);                 /* :109 */
        /* Elogpeta :69 */
        double dv = 0.0, q = 0.0;
        for (int64_t i = 0; i < K; ++i) { dv += invsigma[i * K + i] * v[i]; df[i] = l[i] - mu[i]; }
        for (int64_t i = 0; i < K; ++i) {
            double acc = 0.0;
            for (int64_t j = 0; j < K; ++j) acc += invsigma[j * K + i] * df[j];
            q += df[i] * acc;
        }
        const double t1 = 0.5 * (logdet - (double)K * log(2.0 * M_PI) - dv - q);
        /* Elogpc :74-77 */
        const double t2 = log(EPS + pow(eta, a) * pow(1.0 - eta, Cd - a));
        /* Elogpz :81-84 */
        double z = 0.0, se = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            double acc = 0.0;
            for (int64_t i = 0; i < K; ++i) acc += phi[n * K + i] * l[i];
            z += acc * (double)ct[n];
        }
        for (int64_t i = 0; i < K; ++i) se += exp(l[i] + 0.5 * v[i] - lz);
        const double t3 = z - Cd * (se + lz - 1.0);
        /* Elogpw :88-91 */
        double t4 = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            const double* bcol = beta + (int64_t)tm[n] * K;
            double acc = 0.0;
            for (int64_t i = 0; i < K; ++i) acc += phi[n * K + i] * log(bcol[i] + EPS);
            t4 += acc * ((double)ct[n] * tau[off + n]) + (double)ct[n] * (1.0 - tau[off + n]) * log(kappa[tm[n]] + EPS);
        }
        /* Elogqeta :95-97 */
        double slv = 0.0;
        for (int64_t i = 0; i < K; ++i) slv += log(v[i]);
        const double t5 = -0.5 * ((double)K * (1.0 + log(2.0 * M_PI)) + slv);
        /* Elogqc :101-104 */
        double t6 = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            const double p = tau[off + n];
            if (p > 0.0 && p < 1.0) t6 += (double)ct[n] * (p * log(p) + (1.0 - p) * log(1.0 - p));
        }
        /* Elogqz :108-111 */
        double t7 = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            double h = 0.0;
            for (int64_t i = 0; i < K; ++i) { const double pv = phi[n * K + i]; if (pv > 0.0) h -= pv * log(pv); }
            t7 -= (double)ct[n] * h;
        }
        elbo += t1 + t2 + t3 + t4 - t5 - t6 - t7;                                              /* :110 */
    }
    free(phi); free(L); free(df);
    return elbo;
}
