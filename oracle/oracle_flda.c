/*
 * oracle_flda.c -- fp64 restatement of the reference's CPU filtered-LDA path (src/fLDA.jl).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see tmvb_oracle.h header).
 * Every function cites the reference lines it follows.  fLDA adds to LDA a per-token Bernoulli switch tau_n
 * ("is this token topical?") with prior eta and a corpus-wide background distribution kappa over the vocabulary.
 *
 * Storage: tau / tau_old are flat [nnz] arrays in CSR token order (the reference's Vector{Vector} tau[d][n]).
 */
#include "tmvb_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const double EPS = ORC_EPSILON;

/* One document's sweep loop, src/fLDA.jl:224-233:
 *   update_phi! (:188-191), update_tau! (:180-185), update_gamma! (:173-176), update_Elogtheta! (:166-169),
 *   break if norm(Elogtheta - Elogtheta_old) < vtol.
 * phi (K x N_d) holds the last-sweep phi on return. */
int orc_flda_doc_sweeps(int64_t K, int64_t Nd, const int32_t* terms, const int32_t* counts,
                        double eta, const double* alpha, const double* kappa, const double* beta,
                        double* gamma_d, double* Elogtheta_d, double* Elogtheta_old_d,
                        double* tau_d, double* tau_old_d, double* phi, int viter, double vtol)
{
    int sweeps = 0;
    for (int v = 0; v < viter; ++v) {
        ++sweeps;
        /* update_phi!  :188-191: phi = additive_logistic(tau' .* log.(beta[:,terms] .+ EPS) .+ Elogtheta, dims=1)
         * (column softmax with max subtraction, src/utils.jl:114-122) */
        for (int64_t n = 0; n < Nd; ++n) {
            const double* bcol = beta + (int64_t)terms[n] * K;
            double* pcol = phi + n * K;
            double mx = -INFINITY;
            for (int64_t i = 0; i < K; ++i) {
                pcol[i] = tau_d[n] * log(bcol[i] + EPS) + Elogtheta_d[i];
                if (pcol[i] > mx) mx = pcol[i];
            }
            double s = 0.0;
            for (int64_t i = 0; i < K; ++i) { pcol[i] = exp(pcol[i] - mx); s += pcol[i]; }
            for (int64_t i = 0; i < K; ++i) pcol[i] /= s;
        }
        /* update_tau!  :180-185: tau_old = tau;
         *   tau = eta ./ (EPS .+ eta .+ (1 - eta) * (kappa[terms] .* vec(prod(beta[:,terms].^-phi, dims=1)))) */
        for (int64_t n = 0; n < Nd; ++n) {
            const double* bcol = beta + (int64_t)terms[n] * K;
            const double* pcol = phi + n * K;
            tau_old_d[n] = tau_d[n];
            double prod = 1.0;
            for (int64_t i = 0; i < K; ++i) prod *= pow(bcol[i], -pcol[i]);     /* Julia: 0.0^-0.0 = 1, 0.0^-p = Inf */
            tau_d[n] = eta / (EPS + (eta + (1.0 - eta) * (kappa[terms[n]] * prod)));
        }
        /* update_gamma!  :173-176:  @positive gamma[d] = alpha + phi * counts   (NOT weighted by tau) */
        for (int64_t i = 0; i < K; ++i) {
            double acc = 0.0;
            for (int64_t n = 0; n < Nd; ++n) acc += phi[n * K + i] * (double)counts[n];
            gamma_d[i] = EPS + (alpha[i] + acc);
        }
        /* update_Elogtheta!  :166-169 */
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
        if (sqrt(dist2) < vtol) break;                                           /* :230 */
    }
    return sweeps;
}

static int64_t flda_max_len(const int64_t* doc_ptr, int64_t d0, int64_t d1)
{
    int64_t mx = 1;
    for (int64_t d = d0; d < d1; ++d) if (doc_ptr[d + 1] - doc_ptr[d] > mx) mx = doc_ptr[d + 1] - doc_ptr[d];
    return mx;
}

/* E-step over documents [d0, d1): sweeps + update_beta!(model, d) (:159-162) + update_kappa!(model, d) (:145-148).
 * Both scatters are `X[terms] += ...` assignments: duplicate term ids in one document are overwritten (quirk Q1),
 * reproduced by staging.  beta_temp (K x V) and kappa_temp (V) are accumulated into. */
int orc_flda_estep(int64_t M, int64_t V, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   int64_t d0, int64_t d1, double eta, const double* alpha, const double* kappa, const double* beta,
                   double* beta_temp, double* kappa_temp, double* gamma, double* Elogtheta, double* Elogtheta_old,
                   double* tau, double* tau_old, int viter, double vtol, int32_t* sweeps_out)
{
    (void)M; (void)V;
    int64_t mx = flda_max_len(doc_ptr, d0, d1);
    double* phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    double* stage = (double*)malloc(sizeof(double) * (size_t)((K + 1) * mx));
    if (!phi || !stage) { free(phi); free(stage); return -1; }
    for (int64_t q = 0; q < K * mx; ++q) phi[q] = 1.0 / (double)K;               /* constructor phi, :59 */
    for (int64_t d = d0; d < d1; ++d) {
        int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
        const int32_t* tm = terms + off;
        const int32_t* ct = counts + off;
        int sw = orc_flda_doc_sweeps(K, Nd, tm, ct, eta, alpha, kappa, beta, gamma + d * K, Elogtheta + d * K,
                                     Elogtheta_old + d * K, tau + off, tau_old + off, phi, viter, vtol);
        if (sweeps_out) sweeps_out[d - d0] = sw;
        /* :161  beta_temp[:,terms] += phi .* (tau .* counts)'   and  :147  kappa_temp[terms] += (1 .- tau) .* counts */
        for (int64_t n = 0; n < Nd; ++n) {
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
    free(phi); free(stage);
    return 0;
}

/* The same E-step, document-parallel with OpenMP (bench.py / tools/model_bench.py cpu_baseline only): every thread runs whole
 * documents and accumulates private statistics, reduced at the end.  Private accumulation ADDS duplicate term ids of one
 * document (quirk Q1 overwrites them), so this variant is for condensed corpora -- which every corpus the engine accepts is. */
int orc_flda_estep_omp(int64_t M, int64_t V, int64_t K,
                       const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                       int64_t d0, int64_t d1, double eta, const double* alpha, const double* kappa, const double* beta,
                       double* beta_temp, double* kappa_temp, double* gamma, double* Elogtheta, double* Elogtheta_old,
                       double* tau, double* tau_old, int viter, double vtol, int nthreads)
{
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int64_t mx = flda_max_len(doc_ptr, d0, d1);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
        double* phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
        double* bt = (double*)calloc((size_t)(K * V), sizeof(double));
        double* kt = (double*)calloc((size_t)V, sizeof(double));
        for (int64_t q = 0; q < K * mx; ++q) phi[q] = 1.0 / (double)K;
#pragma omp for schedule(dynamic, 16)
        for (int64_t d = d0; d < d1; ++d) {
            int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
            const int32_t* tm = terms + off;
            const int32_t* ct = counts + off;
            orc_flda_doc_sweeps(K, Nd, tm, ct, eta, alpha, kappa, beta, gamma + d * K, Elogtheta + d * K,
                                Elogtheta_old + d * K, tau + off, tau_old + off, phi, viter, vtol);
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
        free(phi); free(bt); free(kt);
    }
#else
    (void)nthreads;
    orc_flda_estep(M, V, K, doc_ptr, terms, counts, d0, d1, eta, alpha, kappa, beta, beta_temp, kappa_temp, gamma, Elogtheta,
                   Elogtheta_old, tau, tau_old, viter, vtol, NULL);
#endif
    (void)M;
    return used;
}

/* update_kappa!(model)  :138-142 */
void orc_flda_update_kappa(int64_t V, double* kappa, double* kappa_old, double* kappa_temp)
{
    memcpy(kappa_old, kappa, sizeof(double) * (size_t)V);
    double s = 0.0;
    for (int64_t j = 0; j < V; ++j) s += kappa_temp[j];
    for (int64_t j = 0; j < V; ++j) kappa[j] = kappa_temp[j] / s;
    memset(kappa_temp, 0, sizeof(double) * (size_t)V);
}

/* update_eta!  :122-124: eta = sum_d dot(tau[d], counts_d) / sum(C) */
double orc_flda_update_eta(int64_t M, const int64_t* doc_ptr, const int32_t* counts, const double* tau)
{
    double num = 0.0, den = 0.0;
    for (int64_t d = 0; d < M; ++d) {
        double a = 0.0, c = 0.0;
        for (int64_t q = doc_ptr[d]; q < doc_ptr[d + 1]; ++q) { a += tau[q] * (double)counts[q]; c += (double)counts[q]; }
        num += a; den += c;
    }
    return num / den;
}

/* update_elbo!  :108-118 with the seven terms of :62-105 over documents [d0, d1) */
double orc_flda_update_elbo(int64_t M, int64_t V, int64_t K,
                            const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                            int64_t d0, int64_t d1, double eta, const double* alpha, const double* kappa,
                            const double* beta, const double* beta_old, const double* gamma, const double* Elogtheta,
                            const double* Elogtheta_old, const double* tau, const double* tau_old)
{
    (void)M; (void)V;
    int64_t mx = flda_max_len(doc_ptr, d0, d1);
    double* phi = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    double asum = 0.0, lgsum = 0.0;
    for (int64_t i = 0; i < K; ++i) { asum += alpha[i]; lgsum += orc_lgamma(alpha[i]); }
    const double cst = orc_finite(orc_lgamma(asum)) - orc_finite(lgsum);        /* :63 */
    double elbo = 0.0;
    for (int64_t d = d0; d < d1; ++d) {
        int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
        const int32_t* tm = terms + off;
        const int32_t* ct = counts + off;
        const double* El = Elogtheta + d * K;
        const double* Elo = Elogtheta_old + d * K;
        const double* g = gamma + d * K;
        /* :112 phi rebuilt from tau_old / beta_old / Elogtheta_old */
        for (int64_t n = 0; n < Nd; ++n) {
            const double* bcol = beta_old + (int64_t)tm[n] * K;
            double* pcol = phi + n * K;
            double m = -INFINITY;
            for (int64_t i = 0; i < K; ++i) { pcol[i] = tau_old[off + n] * log(bcol[i] + EPS) + Elo[i]; if (pcol[i] > m) m = pcol[i]; }
            double s = 0.0;
            for (int64_t i = 0; i < K; ++i) { pcol[i] = exp(pcol[i] - m); s += pcol[i]; }
            for (int64_t i = 0; i < K; ++i) pcol[i] /= s;
        }
        /* Elogptheta :63 */
        double t1 = cst;
        for (int64_t i = 0; i < K; ++i) t1 += (alpha[i] - 1.0) * El[i];
        /* Elogpc :69-71: log(EPS + eta^dot(tau,counts) * (1-eta)^(C - dot(tau,counts))) -- the powers are formed
         * first, so the term saturates at log(EPS) once the product drops below EPS (C_d >~ 100 at eta = 0.5) */
        double a = 0.0, C = 0.0;
        for (int64_t n = 0; n < Nd; ++n) { a += tau[off + n] * (double)ct[n]; C += (double)ct[n]; }
        double t2 = log(EPS + pow(eta, a) * pow(1.0 - eta, C - a));
        /* Elogpz :76-78  dot(phi*counts, Elogtheta) */
        double t3 = 0.0;
        for (int64_t i = 0; i < K; ++i) {
            double acc = 0.0;
            for (int64_t n = 0; n < Nd; ++n) acc += phi[n * K + i] * (double)ct[n];
            t3 += acc * El[i];
        }
        /* Elogpw :82-84 */
        double t4 = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            const double* bcol = beta + (int64_t)tm[n] * K;
            double acc = 0.0;
            for (int64_t i = 0; i < K; ++i) acc += phi[n * K + i] * log(bcol[i] + EPS);
            t4 += acc * ((double)ct[n] * tau[off + n]) + (double)ct[n] * (1.0 - tau[off + n]) * log(kappa[tm[n]] + EPS);
        }
        /* Elogqtheta :88-90 = -entropy(Dirichlet(gamma)), override of src/utils.jl:163-180 */
        double t5;
        if (K == 1) {
            t5 = -0.0;
        } else {
            double g0 = 0.0, lmnB = 0.0;
            for (int64_t i = 0; i < K; ++i) { g0 += g[i]; lmnB += orc_lgamma(g[i]); }
            lmnB -= orc_lgamma(g0);
            double en = lmnB + (g0 - (double)K) * orc_digamma(g0);
            for (int64_t i = 0; i < K; ++i) en -= (g[i] - 1.0) * orc_digamma(g[i]);
            t5 = -en;
        }
        /* Elogqc :94-97 = -sum c_n entropy(Bernoulli(tau_n)) ; entropy = -(p log p + (1-p) log(1-p)), 0 at p in {0,1} */
        double t6 = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            const double p = tau[off + n];
            double h = 0.0;
            if (p > 0.0 && p < 1.0) h = -(p * log(p) + (1.0 - p) * log(1.0 - p));
            t6 -= (double)ct[n] * h;
        }
        /* Elogqz :101-104 = -sum c_n entropy(Categorical(phi[:,n])) */
        double t7 = 0.0;
        for (int64_t n = 0; n < Nd; ++n) {
            double h = 0.0;
            for (int64_t i = 0; i < K; ++i) { const double p = phi[n * K + i]; if (p > 0.0) h -= p * log(p); }
            t7 -= (double)ct[n] * h;
        }
        elbo += t1 + t2 + t3 + t4 - t5 - t6 - t7;                              /* :114 */
    }
    free(phi);
    return elbo;
}
