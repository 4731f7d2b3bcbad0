// tmvb_dirichlet.h -- the Dirichlet-prior pieces shared by LDA (tmvb_lda.hip) and filtered LDA (tmvb_flda.hip):
// update_alpha! (src/LDA.jl:97-118 = src/fLDA.jl:128-150, the same log-barrier Newton iteration) and the final ELBO
// reduction with the corpus-level constant of Elogptheta (src/LDA.jl:51 = src/fLDA.jl:63).
#pragma once
#include "tmvb_common_kernels.h"

// update_alpha!  src/LDA.jl:97-118, fp64, one wave (lane = topic).
template <int NSLOT>
static __global__ __launch_bounds__(64) void lda_alpha_kernel(int K, double Md, const double* __restrict__ esum_d,
                                                       const float* __restrict__ esum_f, double* __restrict__ alpha_d,
                                                       float* __restrict__ alpha_f, int niter, double ntol,
                                                       int* __restrict__ iters_out)
{
    const int lane = threadIdx.x;
    double a[NSLOT], es[NSLOT], grad[NSLOT], hinv[NSLOT], pp[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        int i = lane + 64 * s;
        a[s] = (i < K) ? alpha_d[i] : 1.0;
        es[s] = (i < K) ? (esum_f ? (double)esum_f[i] : esum_d[i]) : 0.0;
    }
    double nu = (double)K;
    int it = 0;
    for (int t = 0; t < niter; ++t) {
        ++it;
        double rho = 1.0;
        double l = 0.0;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) if (lane + 64 * s < K) l += a[s];
        const double asum = wave_sum_d(l);
        double dgs, tgs;
        digamma_trigamma_d(asum, dgs, tgs);
        double gh = 0.0, hs = 0.0, gn2 = 0.0;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            if (lane + 64 * s < K) {
                double dg, tg;
                digamma_trigamma_d(a[s], dg, tg);
                const double ra = tmvb_rcp_d(a[s]);
                grad[s] = nu * ra + Md * (dgs - dg) + es[s];                         // :103
                hinv[s] = -tmvb_rcp_d(fma(Md, tg, nu * ra * ra));                    // :104
                gh += grad[s] * hinv[s]; hs += hinv[s]; gn2 += grad[s] * grad[s];
            } else { grad[s] = 0.0; hinv[s] = 0.0; }
        }
        gh = wave_sum_d(gh); hs = wave_sum_d(hs); gn2 = wave_sum_d(gn2);
        const double c = gh / (1.0 / (Md * tgs) + hs);                               // :105
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) pp[s] = (grad[s] - c) * hinv[s];
        for (int guard = 0; guard < 1200; ++guard) {                                 // :107-109
            double mn = INFINITY;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) if (lane + 64 * s < K) mn = fmin(mn, a[s] - rho * pp[s]);
            mn = wave_min_d(mn);
            if (mn < 0.0) rho *= 0.5; else break;
        }
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {                                            // :110 @finite
            if (lane + 64 * s < K) {
                double na = fabs(a[s] - rho * pp[s]);
                na = fmin(na, 1.7976931348623157e308);
                a[s] = (a[s] > 0.0) ? na : ((a[s] < 0.0) ? -na : 0.0);
            }
        }
        if ((rho * sqrt(gn2) < ntol) && (nu / (double)K < ntol)) break;              // :112
        nu *= 0.5;                                                                   // :115
    }
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        int i = lane + 64 * s;
        if (i < K) {
            double v = a[s] + TMVB_EPS_D;                                            // :117
            alpha_d[i] = v;
            alpha_f[i] = (float)v;
        }
    }
    if (lane == 0 && iters_out) *iters_out = it;
}

// elbo = sum_d doc_val[d] + M * (lgamma(sum alpha) - sum lgamma(alpha))   (src/LDA.jl:51, finite())
// n_vals > 0: doc_val holds n_vals partial sums (lda_elbo_doc_kernel: one per block of documents) instead of one value per document
static __global__ __launch_bounds__(1024) void lda_elbo_final_kernel(const double* __restrict__ doc_val, int64_t M, int K,
                                                              const double* __restrict__ alpha_d, double* __restrict__ out,
                                                              const double* __restrict__ pw_partial, int pw_blocks, double pw_share,
                                                              int64_t n_vals = 0,
                                                              const double* __restrict__ esum = nullptr,        // [K] sum_d Elogtheta_d of these documents, or NULL
                                                              const double* __restrict__ lz_partial = nullptr)  // [pw_blocks] sums of c log2 s (beta_norm_kernel), or NULL
{
    const int64_t NV = n_vals > 0 ? n_vals : M;
    __shared__ double red[1024];
    // 16 loads in flight per thread, unconditional (a slot past the end reads element 0 and adds 0): the one block's 126 dependent
    // trips of one load each were most of this kernel's 40 us at M = 128 804 (round 4; fixed order: run-to-run bitwise)
    double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t d0 = threadIdx.x; d0 < NV; d0 += 16 * 1024) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int64_t d = d0 + (int64_t)u * 1024; v[u] = doc_val[d < NV ? d : 0]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int64_t d = d0 + (int64_t)u * 1024; s8[u & 7] += (d < NV) ? v[u] : 0.0; }
    }
    double pwl = 0.0;                                             // E_q[log p(w)] partials of update_beta!, if in use
    if (pw_partial) for (int b = threadIdx.x; b < pw_blocks; b += 1024) pwl += pw_share * pw_partial[b];
    if (lz_partial) for (int b = threadIdx.x; b < pw_blocks; b += 1024) pwl += 0.6931471805599453 * lz_partial[b];
    if (esum && (int)threadIdx.x < K) pwl += (alpha_d[threadIdx.x] - 1.0) * esum[threadIdx.x];       // Elogptheta's dot product over the documents (src/LDA.jl:51)
    red[threadIdx.x] = (((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]))) + pwl;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    // lgamma(alpha_i) in parallel (K <= 1024 = blockDim), summed in index order by thread 0
    __shared__ double lga[1024];
    if ((int)threadIdx.x < K) lga[threadIdx.x] = lgamma(alpha_d[threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        double asum = 0.0, lg = 0.0;
        for (int i = 0; i < K; ++i) { asum += alpha_d[i]; lg += lga[i]; }
        double a = lgamma(asum);
        a = fmin(fmax(a, -1.7976931348623157e308), 1.7976931348623157e308);
        lg = fmin(fmax(lg, -1.7976931348623157e308), 1.7976931348623157e308);
        out[0] = red[0] + (double)M * (a - lg);
    }
}

