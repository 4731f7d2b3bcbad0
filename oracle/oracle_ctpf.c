/*
 * oracle_ctpf.c -- fp64 restatement of the reference's CPU CTPF path (src/CTPF.jl).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see tmvb_oracle.h header).
 * Variable names keep the reference's Hebrew-letter names:
 *   alef/bet   shape/rate of q(beta)     K x V / K
 *   gimel/dalet shape/rate of q(theta)   K x M / K
 *   he/vav     shape/rate of q(eta)      K x U / K
 *   zayin/het  shape/rate of q(epsilon)  K x M / K
 */
#include "tmvb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double *phi, *xi, *stage, *base_t, *base_b, *base_p; } ctpf_ws;

static int ws_alloc(ctpf_ws* w, int64_t K, int64_t mxN, int64_t mxR)
{
    int64_t mx = mxN > 2 * mxR ? mxN : 2 * mxR;
    w->phi = (double*)malloc(sizeof(double) * (size_t)(K * mxN));
    w->xi = (double*)malloc(sizeof(double) * (size_t)(2 * K * mxR));
    w->stage = (double*)malloc(sizeof(double) * (size_t)(K * mx));
    w->base_t = (double*)malloc(sizeof(double) * (size_t)K);
    w->base_b = (double*)malloc(sizeof(double) * (size_t)K);
    w->base_p = (double*)malloc(sizeof(double) * (size_t)K);
    return !(w->phi && w->xi && w->stage && w->base_t && w->base_b && w->base_p);
}
static void ws_free(ctpf_ws* w)
{
    free(w->phi); free(w->xi); free(w->stage); free(w->base_t); free(w->base_b); free(w->base_p);
}

static void max_lens(const int64_t* doc_ptr, const int64_t* rdr_ptr, int64_t d0, int64_t d1,
                     int64_t* mxN, int64_t* mxR)
{
    *mxN = 1; *mxR = 1;
    for (int64_t d = d0; d < d1; ++d) {
        int64_t n = doc_ptr[d + 1] - doc_ptr[d], r = rdr_ptr[d + 1] - rdr_ptr[d];
        if (n > *mxN) *mxN = n;
        if (r > *mxR) *mxR = r;
    }
}

/* update_xi!  src/CTPF.jl:334-337: 2K-way softmax per reader column */
static void ctpf_xi(int64_t K, int64_t Rd, const int32_t* readers, const double* he,
                    const double* gimel_d, const double* zayin_d,
                    const double* dalet, const double* het, const double* vav,
                    double* xi, double* bt, double* bb)
{
    for (int64_t i = 0; i < K; ++i) {
        bt[i] = orc_digamma(gimel_d[i]) - log(dalet[i]) - log(vav[i]);
        bb[i] = orc_digamma(zayin_d[i]) - log(het[i]) - log(vav[i]);
    }
    for (int64_t u = 0; u < Rd; ++u) {
        const double* hc = he + (int64_t)readers[u] * K;
        double* xc = xi + u * 2 * K;
        double mx = -INFINITY;
        for (int64_t i = 0; i < K; ++i) {
            double dh = orc_digamma(hc[i]);
            xc[i] = bt[i] + dh; xc[K + i] = bb[i] + dh;
            if (xc[i] > mx) mx = xc[i];
            if (xc[K + i] > mx) mx = xc[K + i];
        }
        double s = 0.0;
        for (int64_t i = 0; i < 2 * K; ++i) { xc[i] = exp(xc[i] - mx); s += xc[i]; }
        for (int64_t i = 0; i < 2 * K; ++i) xc[i] /= s;
    }
}

/* update_phi!  src/CTPF.jl:327-330 */
static void ctpf_phi(int64_t K, int64_t Nd, const int32_t* terms, const double* alef,
                     const double* gimel_d, const double* dalet, const double* bet,
                     double* phi, double* bp)
{
    for (int64_t i = 0; i < K; ++i) bp[i] = orc_digamma(gimel_d[i]) - log(dalet[i]) - log(bet[i]);
    for (int64_t n = 0; n < Nd; ++n) {
        const double* ac = alef + (int64_t)terms[n] * K;
        double* pc = phi + n * K;
        double mx = -INFINITY;
        for (int64_t i = 0; i < K; ++i) { pc[i] = bp[i] + orc_digamma(ac[i]); if (pc[i] > mx) mx = pc[i]; }
        double s = 0.0;
        for (int64_t i = 0; i < K; ++i) { pc[i] = exp(pc[i] - mx); s += pc[i]; }
        for (int64_t i = 0; i < K; ++i) pc[i] /= s;
    }
}

/* one document: src/CTPF.jl:354-364 */
static int ctpf_doc(int64_t K, int64_t Nd, const int32_t* terms, const int32_t* counts,
                    int64_t Rd, const int32_t* readers, const int32_t* ratings,
                    const orc_ctpf_hyper* hp, const double* alef, const double* he,
                    const double* bet, const double* vav, const double* dalet, const double* het,
                    double* gimel_d, double* gimel_old_d, double* zayin_d, double* zayin_old_d,
                    int viter, double vtol, ctpf_ws* w)
{
    int sweeps = 0;
    for (int v = 0; v < viter; ++v) {
        ++sweeps;
        ctpf_xi(K, Rd, readers, he, gimel_d, zayin_d, dalet, het, vav, w->xi, w->base_t, w->base_b);  /* :355 */
        ctpf_phi(K, Nd, terms, alef, gimel_d, dalet, bet, w->phi, w->base_p);                         /* :356 */
        /* update_zayin!  :318-323 */
        for (int64_t i = 0; i < K; ++i) {
            double acc = 0.0;
            for (int64_t u = 0; u < Rd; ++u) acc += w->xi[u * 2 * K + K + i] * (double)ratings[u];
            zayin_old_d[i] = zayin_d[i];
            zayin_d[i] = hp->g + acc;
        }
        /* update_gimel!  :309-314 */
        double d2 = 0.0;
        for (int64_t i = 0; i < K; ++i) {
            double a1 = 0.0, a2 = 0.0;
            for (int64_t n = 0; n < Nd; ++n) a1 += w->phi[n * K + i] * (double)counts[n];
            for (int64_t u = 0; u < Rd; ++u) a2 += w->xi[u * 2 * K + i] * (double)ratings[u];
            gimel_old_d[i] = gimel_d[i];
            gimel_d[i] = (hp->c + a1) + a2;
            double df = gimel_d[i] - gimel_old_d[i];
            d2 += df * df;
        }
        if (sqrt(d2) < vtol) break;                                                                  /* :359 */
    }
    return sweeps;
}

/* update_he!(d) :274-277 and update_alef!(d) :259-262, with the overwrite quirk Q1 */
static void ctpf_scatter(int64_t K, int64_t Nd, const int32_t* terms, const int32_t* counts,
                         int64_t Rd, const int32_t* readers, const int32_t* ratings,
                         const ctpf_ws* w, double* alef_temp, double* he_temp)
{
    for (int64_t u = 0; u < Rd; ++u) {
        const double* hc = he_temp + (int64_t)readers[u] * K;
        for (int64_t i = 0; i < K; ++i)
            w->stage[u * K + i] = hc[i] + (w->xi[u * 2 * K + i] + w->xi[u * 2 * K + K + i]) * (double)ratings[u];
    }
    for (int64_t u = 0; u < Rd; ++u)
        memcpy(he_temp + (int64_t)readers[u] * K, w->stage + u * K, sizeof(double) * (size_t)K);
    for (int64_t n = 0; n < Nd; ++n) {
        const double* ac = alef_temp + (int64_t)terms[n] * K;
        for (int64_t i = 0; i < K; ++i) w->stage[n * K + i] = ac[i] + w->phi[n * K + i] * (double)counts[n];
    }
    for (int64_t n = 0; n < Nd; ++n)
        memcpy(alef_temp + (int64_t)terms[n] * K, w->stage + n * K, sizeof(double) * (size_t)K);
}

int orc_ctpf_estep(int64_t M, int64_t V, int64_t U, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                   int64_t d0, int64_t d1, const orc_ctpf_hyper* hp,
                   const double* alef, const double* he,
                   const double* bet, const double* vav, const double* dalet, const double* het,
                   double* alef_temp, double* he_temp,
                   double* gimel, double* gimel_old, double* zayin, double* zayin_old,
                   int viter, double vtol, int32_t* sweeps_out)
{
    (void)M; (void)V; (void)U;
    int64_t mxN, mxR;
    max_lens(doc_ptr, rdr_ptr, d0, d1, &mxN, &mxR);
    ctpf_ws w;
    if (ws_alloc(&w, K, mxN, mxR)) return -1;
    for (int64_t q = 0; q < K * mxN; ++q) w.phi[q] = 1.0 / (double)K;
    for (int64_t q = 0; q < 2 * K * mxR; ++q) w.xi[q] = 0.5 / (double)K;
    for (int64_t d = d0; d < d1; ++d) {
        int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
        int64_t roff = rdr_ptr[d], Rd = rdr_ptr[d + 1] - roff;
        int sw = ctpf_doc(K, Nd, terms + off, counts + off, Rd, readers + roff, ratings + roff, hp,
                          alef, he, bet, vav, dalet, het,
                          gimel + d * K, gimel_old + d * K, zayin + d * K, zayin_old + d * K,
                          viter, vtol, &w);
        if (sweeps_out) sweeps_out[d - d0] = sw;
        ctpf_scatter(K, Nd, terms + off, counts + off, Rd, readers + roff, ratings + roff, &w, alef_temp, he_temp);
    }
    ws_free(&w);
    return 0;
}

/* sweeps_out (may be NULL): per-document sweep counts, [d - d0] */
int orc_ctpf_estep_omp_sw(int64_t M, int64_t V, int64_t U, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                   int64_t d0, int64_t d1, const orc_ctpf_hyper* hp,
                   const double* alef, const double* he,
                   const double* bet, const double* vav, const double* dalet, const double* het,
                   double* alef_temp, double* he_temp,
                   double* gimel, double* gimel_old, double* zayin, double* zayin_old,
                   int viter, double vtol, int nthreads, int32_t* sweeps_out)
{
    (void)M;
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int64_t mxN, mxR;
    max_lens(doc_ptr, rdr_ptr, d0, d1, &mxN, &mxR);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
        ctpf_ws w;
        ws_alloc(&w, K, mxN, mxR);
        double* at = (double*)calloc((size_t)(K * V), sizeof(double));
        double* ht = (double*)calloc((size_t)(K * U), sizeof(double));
#pragma omp for schedule(dynamic, 32)
        for (int64_t d = d0; d < d1; ++d) {
            int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
            int64_t roff = rdr_ptr[d], Rd = rdr_ptr[d + 1] - roff;
            int sw = ctpf_doc(K, Nd, terms + off, counts + off, Rd, readers + roff, ratings + roff, hp,
                     alef, he, bet, vav, dalet, het,
                     gimel + d * K, gimel_old + d * K, zayin + d * K, zayin_old + d * K, viter, vtol, &w);
            if (sweeps_out) sweeps_out[d - d0] = sw;
            ctpf_scatter(K, Nd, terms + off, counts + off, Rd, readers + roff, ratings + roff, &w, at, ht);
        }
#pragma omp critical
        {
            for (int64_t q = 0; q < K * V; ++q) alef_temp[q] += at[q];
            for (int64_t q = 0; q < K * U; ++q) he_temp[q] += ht[q];
        }
        free(at); free(ht);
        ws_free(&w);
    }
#else
    (void)nthreads;
    orc_ctpf_estep(M, V, U, K, doc_ptr, terms, counts, rdr_ptr, readers, ratings, d0, d1, hp, alef, he,
                   bet, vav, dalet, het, alef_temp, he_temp, gimel, gimel_old, zayin, zayin_old, viter, vtol, sweeps_out);
#endif
    return used;
}

int orc_ctpf_estep_omp(int64_t M, int64_t V, int64_t U, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                   int64_t d0, int64_t d1, const orc_ctpf_hyper* hp,
                   const double* alef, const double* he,
                   const double* bet, const double* vav, const double* dalet, const double* het,
                   double* alef_temp, double* he_temp,
                   double* gimel, double* gimel_old, double* zayin, double* zayin_old,
                   int viter, double vtol, int nthreads)
{
    return orc_ctpf_estep_omp_sw(M, V, U, K, doc_ptr, terms, counts, rdr_ptr, readers, ratings, d0, d1, hp, alef, he,
                                 bet, vav, dalet, het, alef_temp, he_temp, gimel, gimel_old, zayin, zayin_old,
                                 viter, vtol, nthreads, NULL);
}

/* src/CTPF.jl:366-371 in the reference's order:
 * update_he! :266, update_alef! :251, update_dalet! :295, update_het! :302, update_bet! :281,
 * update_vav! :288 */
void orc_ctpf_mstep(int64_t V, int64_t U, int64_t K, const orc_ctpf_hyper* hp,
                    double* alef, double* alef_old, double* alef_temp,
                    double* he, double* he_old, double* he_temp,
                    const double* gimel_sum, const double* zayin_sum,
                    double* bet, double* bet_old, double* vav, double* vav_old,
                    double* dalet, double* dalet_old, double* het, double* het_old)
{
    memcpy(he_old, he, sizeof(double) * (size_t)(K * U));
    memcpy(he, he_temp, sizeof(double) * (size_t)(K * U));
    for (int64_t q = 0; q < K * U; ++q) he_temp[q] = hp->e;
    memcpy(alef_old, alef, sizeof(double) * (size_t)(K * V));
    memcpy(alef, alef_temp, sizeof(double) * (size_t)(K * V));
    for (int64_t q = 0; q < K * V; ++q) alef_temp[q] = hp->a;
    for (int64_t i = 0; i < K; ++i) {
        double sa = 0.0, sh = 0.0;
        for (int64_t j = 0; j < V; ++j) sa += alef[j * K + i];
        for (int64_t u = 0; u < U; ++u) sh += he[u * K + i];
        dalet_old[i] = dalet[i];
        dalet[i] = (hp->d + sa / bet[i]) + sh / vav[i];          /* :297 (bet, vav not yet updated) */
        het_old[i] = het[i];
        het[i] = hp->h + sh / vav[i];                            /* :304 */
    }
    for (int64_t i = 0; i < K; ++i) {
        bet_old[i] = bet[i];
        bet[i] = hp->b + gimel_sum[i] / dalet[i];                /* :283 (new dalet) */
        vav_old[i] = vav[i];
        vav[i] = (hp->f + gimel_sum[i] / dalet[i]) + zayin_sum[i] / het[i];   /* :290 */
    }
}

/* sum_{y=0}^{n} pdf(Binomial(n,p), y) * loggamma(y+1)   (src/CTPF.jl:116,127,138 and inside
 * entropy(Multinomial)); pdf via the reference's overridden log-pdf (src/utils.jl:159-160) */
static double binom_lgamma_sum(int n, double p)
{
    if (n <= 1) return 0.0;                /* lgamma(1) = lgamma(2) = 0 */
    double s = 0.0;
    double ln1 = orc_lgamma((double)n + 1.0);
    for (int y = 2; y <= n; ++y) {
        double lp = ln1 - orc_lgamma((double)y + 1.0) - orc_lgamma((double)(n - y) + 1.0)
                  + ((y != 0) ? (double)y * log(p) : 0.0)
                  + ((n - y != 0) ? (double)(n - y) * log(1.0 - p) : 0.0);
        s += exp(lp) * orc_lgamma((double)y + 1.0);
    }
    return s;
}

/* entropy(Multinomial(n, p)) (Distributions.jl): -lgamma(n+1) + n*H(p) + sum_i sum_x Binom(x;n,p_i) lgamma(x+1) */
static double multinomial_entropy(int n, const double* p, int64_t len)
{
    double h = 0.0, s;
    for (int64_t i = 0; i < len; ++i) if (p[i] > 0.0) h -= p[i] * log(p[i]);
    s = -orc_lgamma((double)n + 1.0) + (double)n * h;
    for (int64_t i = 0; i < len; ++i) s += binom_lgamma_sum(n, p[i]);
    return s;
}

/* entropy(Gamma(shape, scale)) */
static double gamma_entropy(double a, double scale)
{
    return a + log(scale) + orc_lgamma(a) + (1.0 - a) * orc_digamma(a);
}

/* update_elbo!  src/CTPF.jl:234-247 with terms :111-231 */
double orc_ctpf_update_elbo(int64_t M, int64_t V, int64_t U, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                   const orc_ctpf_hyper* hp,
                   const double* alef, const double* alef_old, const double* he, const double* he_old,
                   const double* bet, const double* bet_old, const double* vav, const double* vav_old,
                   const double* dalet, const double* dalet_old, const double* het, const double* het_old,
                   const double* gimel, const double* gimel_old,
                   const double* zayin, const double* zayin_old)
{
    int64_t mxN, mxR;
    max_lens(doc_ptr, rdr_ptr, 0, M, &mxN, &mxR);
    double* rs_he = (double*)calloc((size_t)K, sizeof(double));
    double* rs_alef = (double*)calloc((size_t)K, sizeof(double));
    double elbo = 0.0;
    /* Elogpbeta :144-150, Elogqbeta :198-204 */
    double x = (double)V * (double)K * (hp->a * log(hp->b) - orc_lgamma(hp->a));
    for (int64_t j = 0; j < V; ++j)
        for (int64_t i = 0; i < K; ++i) {
            double al = alef[j * K + i];
            rs_alef[i] += al;
            x += (hp->a - 1.0) * (orc_digamma(al) - log(bet[i])) - hp->b * al / bet[i];
            x += gamma_entropy(al, 1.0 / bet[i]);       /* - Elogqbeta */
        }
    elbo += x;
    /* Elogpeta :162-168, Elogqeta :216-222 */
    x = (double)U * (double)K * (hp->e * log(hp->f) - orc_lgamma(hp->e));
    for (int64_t u = 0; u < U; ++u)
        for (int64_t i = 0; i < K; ++i) {
            double hv = he[u * K + i];
            rs_he[i] += hv;
            x += (hp->e - 1.0) * (orc_digamma(hv) - log(vav[i])) - hp->f * hv / vav[i];
            x += gamma_entropy(hv, 1.0 / vav[i]);
        }
    elbo += x;
    /* the per-document terms: documents are independent, so they are evaluated document-parallel (OpenMP, one workspace per
     * thread) into ed[d] and then added in document order -- the same sum, bit for bit, as the sequential loop */
    double* ed = (double*)calloc((size_t)(M > 0 ? M : 1), sizeof(double));
#pragma omp parallel
    {
    ctpf_ws w;
    ws_alloc(&w, K, mxN, mxR);
#pragma omp for schedule(dynamic, 16)
    for (int64_t d = 0; d < M; ++d) {
        int64_t off = doc_ptr[d], Nd = doc_ptr[d + 1] - off;
        int64_t roff = rdr_ptr[d], Rd = rdr_ptr[d + 1] - roff;
        const int32_t* tm = terms + off; const int32_t* ct = counts + off;
        const int32_t* rd = readers + roff; const int32_t* rt = ratings + roff;
        const double* gi = gimel + d * K; const double* za = zayin + d * K;
        /* :240-241 phi / xi rebuilt from the *_old variables */
        ctpf_phi(K, Nd, tm, alef_old, gimel_old + d * K, dalet_old, bet_old, w.phi, w.base_p);
        ctpf_xi(K, Rd, rd, he_old, gimel_old + d * K, zayin_old + d * K, dalet_old, het_old, vav_old,
                w.xi, w.base_t, w.base_b);
        double e = 0.0;
        /* Elogpya :111-119 */
        for (int64_t i = 0; i < K; ++i) e -= gi[i] / (dalet[i] * vav[i]) * rs_he[i];
        for (int64_t u = 0; u < Rd; ++u)
            for (int64_t i = 0; i < K; ++i) {
                double xv = w.xi[u * 2 * K + i];
                e += (double)rt[u] * xv * (orc_digamma(gi[i]) - log(dalet[i]) + orc_digamma(he[(int64_t)rd[u] * K + i]) - log(vav[i]))
                     - binom_lgamma_sum(rt[u], xv);
            }
        /* Elogpyb :122-130 */
        for (int64_t i = 0; i < K; ++i) e -= za[i] / (het[i] * vav[i]) * rs_he[i];
        for (int64_t u = 0; u < Rd; ++u)
            for (int64_t i = 0; i < K; ++i) {
                double xv = w.xi[u * 2 * K + K + i];
                e += (double)rt[u] * xv * (orc_digamma(za[i]) - log(het[i]) + orc_digamma(he[(int64_t)rd[u] * K + i]) - log(vav[i]))
                     - binom_lgamma_sum(rt[u], xv);
            }
        /* Elogpz :133-141 */
        for (int64_t i = 0; i < K; ++i) e -= gi[i] / (dalet[i] * bet[i]) * rs_alef[i];
        for (int64_t n = 0; n < Nd; ++n)
            for (int64_t i = 0; i < K; ++i) {
                double pv = w.phi[n * K + i];
                e += (double)ct[n] * pv * (orc_digamma(gi[i]) - log(dalet[i]) + orc_digamma(alef[(int64_t)tm[n] * K + i]) - log(bet[i]))
                     - binom_lgamma_sum(ct[n], pv);
            }
        /* Elogptheta :153-159 */
        e += (double)K * (hp->c * log(hp->d) - orc_lgamma(hp->c));
        for (int64_t i = 0; i < K; ++i)
            e += (hp->c - 1.0) * (orc_digamma(gi[i]) - log(dalet[i])) - hp->d * gi[i] / dalet[i];
        /* Elogpepsilon :171-177 */
        e += (double)K * (hp->g * log(hp->h) - orc_lgamma(hp->g));
        for (int64_t i = 0; i < K; ++i)
            e += (hp->g - 1.0) * (orc_digamma(za[i]) - log(het[i])) - hp->h * za[i] / het[i];
        /* - Elogqy :180-186, - Elogqz :189-195 */
        for (int64_t u = 0; u < Rd; ++u) e += multinomial_entropy(rt[u], w.xi + u * 2 * K, 2 * K);
        for (int64_t n = 0; n < Nd; ++n) e += multinomial_entropy(ct[n], w.phi + n * K, K);
        /* - Elogqtheta :207-213, - Elogqepsilon :225-231 */
        for (int64_t i = 0; i < K; ++i) e += gamma_entropy(gi[i], 1.0 / dalet[i]);
        for (int64_t i = 0; i < K; ++i) e += gamma_entropy(za[i], 1.0 / het[i]);
        ed[d] = e;
    }
    ws_free(&w);
    }
    for (int64_t d = 0; d < M; ++d) elbo += ed[d];
    free(ed);
    free(rs_he); free(rs_alef);
    return elbo;
}

/* train!  src/CTPF.jl:344-376 (the post-training scores/recommendations :378-400 are out of scope) */
int orc_ctpf_train(int64_t M, int64_t V, int64_t U, int64_t K,
                   const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                   const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                   const orc_ctpf_hyper* hp,
                   double* alef, double* alef_old, double* he, double* he_old,
                   double* bet, double* bet_old, double* vav, double* vav_old,
                   double* dalet, double* dalet_old, double* het, double* het_old,
                   double* gimel, double* gimel_old, double* zayin, double* zayin_old,
                   double* elbo, int iter, double tol, int viter, double vtol,
                   int checkelbo, double* elbo_traj)
{
    double* alef_temp = (double*)malloc(sizeof(double) * (size_t)(K * V));
    double* he_temp = (double*)malloc(sizeof(double) * (size_t)(K * (U > 0 ? U : 1)));
    double* gs = (double*)malloc(sizeof(double) * (size_t)K);
    double* zs = (double*)malloc(sizeof(double) * (size_t)K);
    for (int64_t q = 0; q < K * V; ++q) alef_temp[q] = hp->a;     /* :85 */
    for (int64_t q = 0; q < K * U; ++q) he_temp[q] = hp->e;       /* :88 */
    if (doc_ptr[M] == doc_ptr[0]) iter = 0;                        /* :349 */
#define CTPF_ELBO() orc_ctpf_update_elbo(M, V, U, K, doc_ptr, terms, counts, rdr_ptr, readers, ratings, hp, \
        alef, alef_old, he, he_old, bet, bet_old, vav, vav_old, dalet, dalet_old, het, het_old, \
        gimel, gimel_old, zayin, zayin_old)
    if (checkelbo > 0 && checkelbo <= iter) *elbo = CTPF_ELBO();   /* :350 */
    int done = 0;
    for (int k = 1; k <= iter; ++k) {
        ++done;
        orc_ctpf_estep(M, V, U, K, doc_ptr, terms, counts, rdr_ptr, readers, ratings, 0, M, hp, alef, he,
                       bet, vav, dalet, het, alef_temp, he_temp, gimel, gimel_old, zayin, zayin_old,
                       viter, vtol, NULL);
        for (int64_t i = 0; i < K; ++i) { gs[i] = 0.0; zs[i] = 0.0; }
        for (int64_t d = 0; d < M; ++d)
            for (int64_t i = 0; i < K; ++i) { gs[i] += gimel[d * K + i]; zs[i] += zayin[d * K + i]; }
        orc_ctpf_mstep(V, U, K, hp, alef, alef_old, alef_temp, he, he_old, he_temp, gs, zs,
                       bet, bet_old, vav, vav_old, dalet, dalet_old, het, het_old);
        if (elbo_traj) elbo_traj[k - 1] = NAN;
        if (checkelbo > 0 && (k % checkelbo) == 0) {
            double e_new = CTPF_ELBO();
            double delta = -(*elbo - e_new);
            *elbo = e_new;
            if (elbo_traj) elbo_traj[k - 1] = e_new;
            if (delta < tol) break;
        }
    }
#undef CTPF_ELBO
    free(alef_temp); free(he_temp); free(gs); free(zs);
    return done;
}
