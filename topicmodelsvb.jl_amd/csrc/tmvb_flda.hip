// tmvb_flda.hip -- filtered latent Dirichlet allocation (fLDA) engine for gfx950 (MI355X).
//
// Path: the per-document coordinate ascent of src/fLDA.jl:222-236 (update_phi! :188, update_tau! :180, update_gamma!
// :173, update_Elogtheta! :166, exit test :230, update_beta!(d) :159, update_kappa!(d) :145) fused into ONE kernel per
// document, the M-step (update_beta! :152, update_kappa! :138, update_alpha! :128, update_eta! :122) and update_elbo!
// (:108) on the device.  The reference has NO accelerator path for the filtered models (`@gpu` does nothing for them,
// src/macros.jl:274-278; SURVEY.md section 8 f4 lists them as the next row after the docfile ingest): this is new.
//
// fLDA adds to LDA a Bernoulli switch tau_n per token ("topical or background?", prior eta) and a background
// distribution kappa over the vocabulary.  The switch enters phi as an EXPONENT,
//       phi[i,n] = softmax_i( tau_n log(beta[i,t_n] + eps) + Elogtheta_i ),
// so phi no longer factors as (beta row) x (document vector) / normaliser: every (token, topic) pair costs an exp per
// sweep, and the register-tile / gather-side factorisations of tmvb_lda.hip do not apply.  The kernels below work on
// the table L = log(beta + eps) ([V][KP], refreshed by update_beta!):
//   * document kernel: the document's L rows sit in an LDS tile; phase A (lane = token) runs the column softmax in two
//     passes over the row (max, then sum and sum p L) -- no cross-lane traffic -- and updates tau_n from
//     prod_i beta^-phi = exp(-sum_i phi_i L_i); phase B (lane = topic quad x token residue) accumulates
//     (phi * counts)_i with one v_exp_f32 per pair; gamma / Elogtheta as in the LDA kernel.
//   * statistics: a gather over the term-major inverted index that REBUILDS phi of the last sweep from
//     (tau_old_n, Elogtheta_old_d, log-sum-exp_n): S[i,j] = sum_n tau_n c_n phi_in, kappa_stat[j] = sum_n (1 - tau_n) c_n.
//     No atomics, fixed summation order.
// Stated deviation: the reference forms prod_i beta[i,t]^-phi[i,n] from beta WITHOUT epsilon (:184); the device uses
// the one table log(beta + eps).  The two differ only for entries beta < ~1e-23 that still carry phi > 0 (an exact 0
// makes the reference's tau_n exactly 0); fp32 beta cannot represent the distinction below 1e-38 anyway.
#include "tmvb_common_kernels.h"
#include "tmvb_train.h"
#include "tmvb_dirichlet.h"
#include "tmvb_filtered.h"

struct FldaParams {
    int K, KP, LPR;
    unsigned lpr_magic;
    const int64_t* doc_ptr;
    const int32_t* terms;
    const int32_t* counts;
    const int32_t* doc_order;
    const float* L;            // [V][KP] log(beta + eps), pads 0
    const float* kappa;        // [V]
    const double* eta;         // [1] device resident (update_eta! never leaves the GPU)
    const float* alpha;        // [K]
    float* gamma; float* elog; float* elog_old;    // [M][K]
    float* tau; float* tau_old;                    // [nnz] CSR order
    float* lse;                // [nnz] log-sum-exp of the last sweep's phi column (input of the statistics pass)
    float* aold;               // [nnz] sum_i phi_in log(beta[i, t_n] + eps) of the last sweep (the exponent update_tau! forms): the decomposed update_elbo!'s per-token input
    uint8_t* sweeps;
    int viter;
    float vtol;
};

static size_t flda_tile_bytes(int rows, int KP) { return ((size_t)rows * KP + KP + 7 * (size_t)rows) * sizeof(float); }

template <int NS>
__global__ __launch_bounds__(64) void flda_estep_kernel(FldaParams p, int64_t first, int tile_rows)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int K = p.K, KP = p.KP, LPR = p.LPR;
    float* Bt = lds;                                   // [tile_rows][KP] rows of L
    float* el_l = Bt + (size_t)tile_rows * KP;         // [KP] Elogtheta (pads -inf)
    float* tp_l = el_l + KP;                           // [tile_rows] tau before this sweep's update_tau! (phi uses it)
    float* tn_l = tp_l + tile_rows;                    // [tile_rows] tau after it
    float* m_l = tn_l + tile_rows;                     // [tile_rows] column maximum
    float* w_l = m_l + tile_rows;                      // [tile_rows] c_n / s_n
    float* c_l = w_l + tile_rows;                      // [tile_rows] counts
    float* a_l = c_l + tile_rows;                      // [tile_rows] sum_i phi_i log(beta_i + eps)
    int* t_l = (int*)(a_l + tile_rows);                // [tile_rows] term ids

    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d];
    const int N = (int)(p.doc_ptr[d + 1] - off);
    const bool single = N <= tile_rows;
    const float eta = (float)p.eta[0];

    bool on[NS];
    float alpha[NS], elog[NS], elog_old[NS], gam[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = lane + 64 * s;
        on[s] = i < K;
        alpha[s] = on[s] ? p.alpha[i] : 0.0f;
        elog[s] = on[s] ? p.elog[(int64_t)d * K + i] : 0.0f;
        elog_old[s] = elog[s]; gam[s] = 0.0f;
    }

    auto load_chunk = [&](int c0, int rows, bool with_tau) {
        for (int n = lane; n < rows; n += 64) {
            t_l[n] = p.terms[off + c0 + n];
            c_l[n] = (float)p.counts[off + c0 + n];
            if (with_tau) tn_l[n] = p.tau[off + c0 + n];
        }
        WAVE_LDS_FENCE();
        const int nch = rows * LPR;
        for (int f0 = 0; f0 < nch; f0 += 64) {
            const int f = f0 + lane;
            if (f < nch) {
                const int n = (LPR == 1) ? f : (int)__umulhi((unsigned)f, p.lpr_magic);
                const int c = f - n * LPR;
                const float* src = p.L + ((int64_t)t_l[n] * KP + 4 * c);
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(Bt + (size_t)f0 * 4), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WAVE_LDS_FENCE();
    };
    // update_phi! (:188-191) normalisers and update_tau! (:180-185), lane = token
    auto phase_a = [&](int c0, int rows, bool store) {
        const float4* er = (const float4*)el_l;
        for (int n = lane; n < rows; n += 64) {
            const float4* br = (const float4*)(Bt + (size_t)n * KP);
            const float tp = tn_l[n];                                   // tau entering this sweep
            float m = -INFINITY;
            for (int q = 0; q < LPR; ++q) {
                const float4 b = br[q], ev = er[q];
                m = fmaxf(m, fmaxf(fmaxf(fmaf(tp, b.x, ev.x), fmaf(tp, b.y, ev.y)), fmaxf(fmaf(tp, b.z, ev.z), fmaf(tp, b.w, ev.w))));
            }
            float s = 0.0f, a = 0.0f;
            for (int q = 0; q < LPR; ++q) {
                const float4 b = br[q], ev = er[q];
                const float p0 = __expf(fmaf(tp, b.x, ev.x) - m), p1 = __expf(fmaf(tp, b.y, ev.y) - m);
                const float p2 = __expf(fmaf(tp, b.z, ev.z) - m), p3 = __expf(fmaf(tp, b.w, ev.w) - m);
                s += (p0 + p1) + (p2 + p3);
                a = fmaf(p0, b.x, a); a = fmaf(p1, b.y, a); a = fmaf(p2, b.z, a); a = fmaf(p3, b.w, a);   // pads: p = 0
            }
            const float A = a / s;                                       // sum_i phi_i log(beta_i + eps) <= 0
            const float prod = __expf(fminf(-A, 87.0f));                 // prod_i beta^-phi  (:184)
            const float tnew = eta / (TMVB_EPS_F + (eta + (1.0f - eta) * (p.kappa[t_l[n]] * prod)));
            tp_l[n] = tp; tn_l[n] = tnew; m_l[n] = m; w_l[n] = c_l[n] / s; a_l[n] = A;
            if (store) {
                p.tau_old[off + c0 + n] = tp; p.tau[off + c0 + n] = tnew; p.lse[off + c0 + n] = m + __logf(s); p.aold[off + c0 + n] = A;
            }
        }
        WAVE_LDS_FENCE();
    };
    const int r4 = lane & 3, ql = lane >> 2;
    // (phi * counts)_i  (update_gamma!, :175), lane = (topic quad, token residue)
    auto phase_b = [&](int rows, float4 (&acc)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int q = ql + 16 * s;
            if (q >= LPR) continue;
            const float4 ev = *(const float4*)(el_l + 4 * q);
            for (int n = r4; n < rows; n += 4) {
                const float tp = tp_l[n], m = m_l[n], w = w_l[n];
                const float4 b = *(const float4*)(Bt + (size_t)n * KP + 4 * q);
                acc[s].x = fmaf(w, __expf(fmaf(tp, b.x, ev.x) - m), acc[s].x);
                acc[s].y = fmaf(w, __expf(fmaf(tp, b.y, ev.y) - m), acc[s].y);
                acc[s].z = fmaf(w, __expf(fmaf(tp, b.z, ev.z) - m), acc[s].z);
                acc[s].w = fmaf(w, __expf(fmaf(tp, b.w, ev.w) - m), acc[s].w);
            }
        }
    };

    int sweeps = 0;
    for (int v = 0; v < p.viter; ++v) {
        ++sweeps;
#pragma unroll
        for (int s = 0; s < NS; ++s) if (on[s]) el_l[lane + 64 * s] = elog[s];
        for (int i = K + lane; i < KP; i += 64) el_l[i] = -INFINITY;    // pads: exp(-inf) = 0
        WAVE_LDS_FENCE();
        float4 acc[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (single) {
            if (v == 0 && N) load_chunk(0, N, true);
            phase_a(0, N, false);                      // tau lives in the tile; stored once at the end
            phase_b(N, acc);
        } else {
            for (int c0 = 0; c0 < N; c0 += tile_rows) {
                const int rows = min(tile_rows, N - c0);
                load_chunk(c0, rows, true);
                phase_a(c0, rows, true);               // the last executed sweep's values remain
                phase_b(rows, acc);
                WAVE_LDS_FENCE();
            }
        }
        float gl = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float4 a = dpp_add4<0xB1>(acc[s]);
            a = dpp_add4<0x4E>(a);
            const float g = (r4 == 0) ? a.x : (r4 == 1) ? a.y : (r4 == 2) ? a.z : a.w;
            gam[s] = TMVB_EPS_F + (alpha[s] + g);                       // update_gamma!  :175
            if (on[s]) gl += gam[s];
        }
        const float dgs = digamma_f(wave_sum(gl));
        float dl = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            elog_old[s] = elog[s];                                      // update_Elogtheta!  :167-168
            if (on[s]) {
                elog[s] = digamma_f(gam[s]) - dgs;
                const float df = elog[s] - elog_old[s];
                dl = fmaf(df, df, dl);
            }
        }
        if (__builtin_amdgcn_sqrtf(wave_sum(dl)) < p.vtol) break;       // :230
    }

    if (sweeps > 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = lane + 64 * s;
            if (on[s]) {
                p.gamma[(int64_t)d * K + i] = gam[s];
                p.elog[(int64_t)d * K + i] = elog[s];
                p.elog_old[(int64_t)d * K + i] = elog_old[s];
            }
        }
        if (single)
            for (int n = lane; n < N; n += 64) {
                p.tau_old[off + n] = tp_l[n]; p.tau[off + n] = tn_l[n];
                p.lse[off + n] = m_l[n] + __logf(c_l[n] / w_l[n]);      // m + log s
                p.aold[off + n] = a_l[n];
            }
    } else {
        for (int n = lane; n < N; n += 64) p.lse[off + n] = INFINITY;   // viter = 0: no responsibilities (phi = 0 in the statistics)
    }
    if (lane == 0) p.sweeps[d] = (uint8_t)min(sweeps, 255);
}

// ------------------------------------------------------------------------------ ELBO
// update_elbo!  src/fLDA.jl:108-118 per document (terms :62-105 without the corpus-level constant of Elogptheta, added by
// lda_elbo_final_kernel).  One wave per document, lane l owns topics l + 64 s.  phi is rebuilt from tau_old, beta_old,
// Elogtheta_old (:112).
template <int NS>
__global__ __launch_bounds__(64) void flda_elbo_kernel(int K, int KP, const int64_t* __restrict__ doc_ptr,
                                                       const int32_t* __restrict__ terms, const int32_t* __restrict__ counts,
                                                       const double* __restrict__ alpha_d, const double* __restrict__ eta_d,
                                                       const float* __restrict__ kappa, const float* __restrict__ beta,
                                                       const float* __restrict__ beta_old, const float* __restrict__ gamma,
                                                       const float* __restrict__ elog, const float* __restrict__ elog_old,
                                                       const float* __restrict__ tau, const float* __restrict__ tau_old,
                                                       double* __restrict__ doc_val)
{
    const int lane = threadIdx.x;
    const int d = blockIdx.x;
    const int64_t off = doc_ptr[d];
    const int N = (int)(doc_ptr[d + 1] - off);
    bool on[NS]; int ix[NS];
    float eo[NS];
    double pc[NS], acc = 0.0, ta = 0.0, Cd = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        on[s] = lane + 64 * s < K;
        ix[s] = on[s] ? lane + 64 * s : 0;
        eo[s] = on[s] ? elog_old[(int64_t)d * K + ix[s]] : -INFINITY;
        pc[s] = 0.0;
    }
    for (int n = 0; n < N; ++n) {
        const int t = terms[off + n];
        const float c = (float)counts[off + n];
        const float tn = tau[off + n], to = tau_old[off + n];
        float x[NS], ml = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            x[s] = on[s] ? fmaf(to, logf(beta_old[(int64_t)t * KP + ix[s]] + TMVB_EPS_F), eo[s]) : -INFINITY;   // :112
            ml = fmaxf(ml, x[s]);
        }
        const float mx = wave_max(ml);
        float ex[NS], sl = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s) { ex[s] = on[s] ? expf(x[s] - mx) : 0.0f; sl += ex[s]; }
        const float inv = 1.0f / wave_sum(sl);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (!on[s]) continue;
            const float ph = ex[s] * inv;
            const double cp = (double)(c * ph);
            pc[s] += cp;
            acc += cp * (double)tn * (double)logf(beta[(int64_t)t * KP + ix[s]] + TMVB_EPS_F);     // Elogpw :83, topical part
            if (ph > 0.0f) acc -= cp * (double)logf(ph);                                           // -Elogqz :101-104
        }
        if (lane == 0) {
            acc += (double)c * (1.0 - (double)tn) * (double)logf(kappa[t] + TMVB_EPS_F);           // Elogpw :83, background part
            if (tn > 0.0f && tn < 1.0f) acc -= (double)c * ((double)tn * log((double)tn) + (1.0 - (double)tn) * log(1.0 - (double)tn));   // -Elogqc :94-97
        }
        ta += (double)tn * (double)c; Cd += (double)c;
    }
    double gl = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (!on[s]) continue;
        const double el = (double)elog[(int64_t)d * K + ix[s]];
        const double g = (double)gamma[(int64_t)d * K + ix[s]];
        acc += (alpha_d[ix[s]] - 1.0) * el;                                                        // Elogptheta :63 (dot part)
        acc += pc[s] * el;                                                                         // Elogpz :77
        if (K > 1) { double ps, lg; digamma_lgamma_d(g, ps, lg); acc += lg - (g - 1.0) * ps; }                                    // -Elogqtheta :89 (utils.jl:172-176)
        gl += g;
    }
    const double g0 = wave_sum_d(gl);
    double tot = wave_sum_d(acc);
    if (K > 1) { double ps0, lg0; digamma_lgamma_d(g0, ps0, lg0); tot += -lg0 + (g0 - (double)K) * ps0; }
    // Elogpc :69-71: the powers are formed first, so the term saturates at log(eps) for documents of more than ~100 tokens
    const double eta = eta_d[0];
    tot += log(TMVB_EPS_D + pow(eta, ta) * pow(1.0 - eta, Cd - ta));
    if (lane == 0) doc_val[d] = tot;
}

// ------------------------------------------------------------------------------ ELBO, decomposed (round 6)
// update_elbo! without rebuilding phi.  With x_in = tau_old_n L_old[i, t_n] + Elogtheta_old_i, lse_n = log sum_i e^{x_in} and
// phi_in = e^{x_in - lse_n} (src/fLDA.jl:112), the token terms of Elogpz (:77), Elogpw (:83) and -Elogqz (:101-104) are
//     c_n [ sum_i phi_in (Elogtheta_i - Elogtheta_old_i) + tau_n sum_i phi_in L_new[i, t_n] - tau_old_n A_n + lse_n ],   A_n = sum_i phi_in L_old[i, t_n],
// and they separate into what the checked iteration already holds: sum_n c_n phi_in = gamma_i - alpha_i - eps (update_gamma!, :175, with the alpha the
// E-step read), sum_n tau_n c_n phi_in = S (update_beta!(d), :161: update_beta! leaves sum S (log(beta_new + eps) - log(beta_old + eps)), which carries
// -tau_n A_n as well), and per token the two floats the E-step kernel stores anyway (lse_n for the statistics pass, A_n = the exponent update_tau!
// forms, :184).  What is left is elementwise: one wave per document adds c_n [lse_n + (tau_n - tau_old_n) A_n + (1 - tau_n) log(kappa + eps) + H(tau_n)]
// over its tokens (lane = token), the Dirichlet terms over its topics (lane = topic), and Elogpc (:69-71).
template <int NS>
__global__ __launch_bounds__(64) void flda_elbo_doc_parts_kernel(int K, const int64_t* __restrict__ doc_ptr, const int32_t* __restrict__ terms,
                                                                 const int32_t* __restrict__ counts, const double* __restrict__ alpha_d,
                                                                 const float* __restrict__ alpha_e, const double* __restrict__ eta_d,
                                                                 const float* __restrict__ kappa, const float* __restrict__ gamma,
                                                                 const float* __restrict__ elog, const float* __restrict__ elog_old,
                                                                 const float* __restrict__ tau, const float* __restrict__ tau_old,
                                                                 const float* __restrict__ lse, const float* __restrict__ aold,
                                                                 double* __restrict__ doc_val)
{
    const int lane = threadIdx.x;
    const int d = blockIdx.x;
    const int64_t off = doc_ptr[d];
    const int N = (int)(doc_ptr[d + 1] - off);
    double acc = 0.0, ta = 0.0, Cd = 0.0;
    for (int n = lane; n < N; n += 64) {
        const double c = (double)counts[off + n];
        const double tn = (double)tau[off + n], to = (double)tau_old[off + n];
        double v = (double)lse[off + n] + (tn - to) * (double)aold[off + n];
        v += (1.0 - tn) * (double)logf(kappa[terms[off + n]] + TMVB_EPS_F);                         // Elogpw :83, background part
        if (tn > 0.0 && tn < 1.0) v -= tn * log(tn) + (1.0 - tn) * log(1.0 - tn);                  // -Elogqc :94-97
        acc += c * v;
        ta += tn * c; Cd += c;
    }
    double gl = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = lane + 64 * s;
        if (i >= K) continue;
        const double el = (double)elog[(int64_t)d * K + i], elo = (double)elog_old[(int64_t)d * K + i];
        const double g = (double)gamma[(int64_t)d * K + i];
        acc += (alpha_d[i] - 1.0) * el;                                                             // Elogptheta :63 (dot part)
        acc += (g - (double)alpha_e[i] - TMVB_EPS_D) * (el - elo);                                   // sum_n c_n phi_in (Elogtheta_i - Elogtheta_old_i)
        if (K > 1) { double ps, lg; digamma_lgamma_d(g, ps, lg); acc += lg - (g - 1.0) * ps; }     // -Elogqtheta :89
        gl += g;
    }
    const double g0 = wave_sum_d(gl);
    double tot = wave_sum_d(acc);
    ta = wave_sum_d(ta); Cd = wave_sum_d(Cd);
    if (K > 1) { double ps0, lg0; digamma_lgamma_d(g0, ps0, lg0); tot += -lg0 + (g0 - (double)K) * ps0; }
    const double eta = eta_d[0];
    tot += log(TMVB_EPS_D + pow(eta, ta) * pow(1.0 - eta, Cd - ta));                                // Elogpc :69-71
    if (lane == 0) doc_val[d] = tot;
}

// ------------------------------------------------------------------------------ host side
struct tmvb_flda {
    tmvb_ctx* ctx = nullptr;
    tmvb_corpus* corp = nullptr;
    int K = 0, KP = 0, nslot = 1;
    int64_t M = 0, V = 0, M_total = 0, nnz = 0;
    double C_total = 0.0;
    bool distributed = false;
    tmvb_comm* comm = nullptr;
    double* d_alpha_d = nullptr; float* d_alpha_f = nullptr;
    float* d_beta[2] = {nullptr, nullptr}; int cur = 0;        // padded [V][KP]
    float* d_L = nullptr;                                      // log(beta + eps)
    float* d_kappa = nullptr; float* d_kappa_old = nullptr;    // [V]
    double* d_eta = nullptr; double* d_ksum = nullptr;
    float* d_stats = nullptr;                                  // S (K*V) | kstat (V) | Elogtheta_sum (K)
    float* d_gamma = nullptr; float* d_elog = nullptr; float* d_elog_old = nullptr;
    float* d_tau = nullptr; float* d_tau_old = nullptr; float* d_lse = nullptr;
    // decomposed update_elbo! (flda_elbo_doc_parts_kernel): the per-token exponent of the last sweep, the alpha the collecting E-step read, update_beta!'s
    // partial sums of S (log beta_new - log beta_old)
    float* d_aold = nullptr; float* d_alpha_e = nullptr; double* d_pw_partial = nullptr; int pw_blocks = 0;
    int parts_env = 1;                 // TMVB_FLDA_ELBO_PARTS at creation: 0 never, 1 the iterations train! checks, 2 every E-step collects
    bool want_parts = false;           // train!: the coming iteration ends in check_elbo!
    bool parts_pending = false;        // the last E-step collected (aold, alpha_e belong to it) and update_beta! has not run behind it yet
    bool parts_valid = false;          // ... it has: d_pw_partial belongs to that E-step too and nothing was set from outside since
    bool force_walk = false;           // train!'s like-with-like evaluation at the switch of forms (tmvb_train.h)
    int elbo_form = 0;                 // the last update_elbo!: 1 decomposed, 0 token walk (tmvb_flda_elbo_form)
    float* d_ts_partial = nullptr;
    uint8_t* d_sweeps = nullptr; int32_t* d_doc_order = nullptr;
    double* d_partial = nullptr; double* d_rowsum = nullptr; double* d_esum = nullptr; double* d_doc_val = nullptr; double* d_elbo = nullptr;
    int* d_iters = nullptr;
    double elbo = 0.0;
    std::vector<tmvb_bucket> buckets;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int64_t stats_len() const { return (int64_t)K * V + V + K; }
    float* kstat() const { return d_stats + (size_t)K * V; }
    float* esum_f() const { return d_stats + (size_t)K * V + (size_t)V; }
};

extern "C" int tmvb_flda_destroy(tmvb_flda* h)
{
    if (!h) return TMVB_OK;
    if (h->ctx) { (void)hipSetDevice(h->ctx->device); (void)hipStreamSynchronize(h->ctx->stream); }
    (void)hipFree(h->d_aold); (void)hipFree(h->d_alpha_e); (void)hipFree(h->d_pw_partial);
    (void)hipFree(h->d_alpha_d); (void)hipFree(h->d_alpha_f); (void)hipFree(h->d_beta[0]); (void)hipFree(h->d_beta[1]); (void)hipFree(h->d_L);
    (void)hipFree(h->d_kappa); (void)hipFree(h->d_kappa_old); (void)hipFree(h->d_eta); (void)hipFree(h->d_ksum); (void)hipFree(h->d_stats);
    (void)hipFree(h->d_gamma); (void)hipFree(h->d_elog); (void)hipFree(h->d_elog_old); (void)hipFree(h->d_tau); (void)hipFree(h->d_tau_old);
    (void)hipFree(h->d_lse); (void)hipFree(h->d_ts_partial); (void)hipFree(h->d_sweeps); (void)hipFree(h->d_doc_order); (void)hipFree(h->d_partial);
    (void)hipFree(h->d_rowsum); (void)hipFree(h->d_esum); (void)hipFree(h->d_doc_val); (void)hipFree(h->d_elbo); (void)hipFree(h->d_iters);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
    return TMVB_OK;
}

static int flda_upload_beta(tmvb_flda* h, float* dst, const double* src)
{
    const size_t K = h->K, KP = h->KP, V = h->V;
    std::vector<float> tmp(V * KP + 4, 0.0f);
    for (size_t j = 0; j < V; ++j)
        for (size_t i = 0; i < K; ++i) tmp[j * KP + i] = (float)src[j * K + i];
    TMVB_HIP(hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    return TMVB_OK;
}

static int flda_download_beta(tmvb_flda* h, double* dst, const float* src)
{
    const size_t K = h->K, KP = h->KP, V = h->V;
    std::vector<float> tmp(V * KP);
    TMVB_HIP(hipMemcpyAsync(tmp.data(), src, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    for (size_t j = 0; j < V; ++j)
        for (size_t i = 0; i < K; ++i) dst[j * K + i] = (double)tmp[j * KP + i];
    return TMVB_OK;
}

static int flda_refresh_L(tmvb_flda* h)
{
    const int64_t total = (int64_t)h->KP * h->V;
    const int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (total + 255) / 256));
    hipLaunchKernelGGL(flda_logbeta_kernel, dim3(nb), dim3(256), 0, h->ctx->stream, h->d_beta[h->cur], h->d_L, h->K, h->KP, h->V);
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

extern "C" int tmvb_flda_set_state(tmvb_flda* h, const double* eta, const double* alpha, const double* kappa, const double* kappa_old,
                                   const double* beta, const double* beta_old, const double* gamma, const double* Elogtheta,
                                   const double* Elogtheta_old, const double* tau, const double* tau_old, const double* elbo);

extern "C" int tmvb_flda_create(tmvb_ctx* ctx, tmvb_corpus* corp, int32_t K, tmvb_flda** out)
{
    TMVB_REQUIRE(out != nullptr, TMVB_EINVAL, "tmvb_flda_create: out is NULL");
    *out = nullptr;
    TMVB_REQUIRE(ctx && corp, TMVB_EINVAL, "tmvb_flda_create: NULL context or corpus");
    TMVB_REQUIRE(K > 0, TMVB_EINVAL, "number of topics must be a positive integer.");      // src/fLDA.jl:30
    TMVB_REQUIRE(K <= 1024, TMVB_EINVAL, "tmvb_flda_create: K <= 1024 (sixteen topic slots per lane); got K=%d", K);
    TMVB_HIP(hipSetDevice(ctx->device));
    tmvb_flda* h = new tmvb_flda();
    tmvb_create_guard<tmvb_flda, tmvb_flda_destroy> guard{h};
    h->ctx = ctx; h->corp = corp; h->K = K; h->KP = tmvb_kpad(K); h->nslot = (K + 63) / 64;
    h->M = corp->info.M; h->V = corp->info.V; h->M_total = h->M; h->nnz = corp->info.nnz;
    h->C_total = (double)corp->info.sum_counts;
    const size_t KM = (size_t)K * h->M, KPV = (size_t)h->KP * h->V + 4, NZ = (size_t)h->nnz;
    int rc;
    if ((rc = tmvb_corpus_term_index(corp))) return rc;
    if ((rc = dmalloc(&h->d_alpha_d, K)) || (rc = dmalloc(&h->d_alpha_f, K)) || (rc = dmalloc(&h->d_beta[0], KPV)) || (rc = dmalloc(&h->d_beta[1], KPV)) ||
        (rc = dmalloc(&h->d_L, KPV)) || (rc = dmalloc(&h->d_kappa, (size_t)h->V)) || (rc = dmalloc(&h->d_kappa_old, (size_t)h->V)) ||
        (rc = dmalloc(&h->d_eta, 1)) || (rc = dmalloc(&h->d_ksum, 1)) || (rc = dmalloc(&h->d_stats, (size_t)h->stats_len())) ||
        (rc = dmalloc(&h->d_gamma, KM)) || (rc = dmalloc(&h->d_elog, KM)) || (rc = dmalloc(&h->d_elog_old, KM)) ||
        (rc = dmalloc(&h->d_tau, NZ)) || (rc = dmalloc(&h->d_tau_old, NZ)) || (rc = dmalloc(&h->d_lse, NZ)) ||
        (rc = dmalloc(&h->d_aold, std::max<size_t>(NZ, 1))) || (rc = dmalloc(&h->d_alpha_e, K)) || (rc = dmalloc(&h->d_pw_partial, 2048)) ||
        (rc = dmalloc(&h->d_ts_partial, (size_t)corp->term_index.n_slots * (K + 1))) || (rc = dmalloc(&h->d_sweeps, (size_t)h->M)) ||
        (rc = dmalloc(&h->d_doc_order, (size_t)h->M)) || (rc = dmalloc(&h->d_partial, (size_t)TMVB_REDUCE_BLOCKS * K)) ||
        (rc = dmalloc(&h->d_rowsum, K)) || (rc = dmalloc(&h->d_esum, K)) || (rc = dmalloc(&h->d_doc_val, (size_t)h->M)) ||
        (rc = dmalloc(&h->d_elbo, 1)) || (rc = dmalloc(&h->d_iters, 1)))
        return rc;
    { const char* e = getenv("TMVB_FLDA_ELBO_PARTS"); h->parts_env = e ? atoi(e) : 1; }
    std::vector<int32_t> order((size_t)h->M);
    std::iota(order.begin(), order.end(), 0);
    const std::vector<int64_t>& len = corp->h_doc_len;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return len[x] > len[y]; });
    {   // LDS tile buckets with this kernel's per-row footprint (6 floats of per-token scratch): cap the tile at 32 KB
        std::vector<tmvb_bucket> bk;
        tmvb_build_lds_buckets(len, order, h->M, h->KP + 4, -1, 3, bk, 32 * 1024);
        h->buckets = bk;
    }
    if (h->M) TMVB_HIP(hipMemcpyAsync(h->d_doc_order, order.data(), (size_t)h->M * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_stats, 0, (size_t)h->stats_len() * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_sweeps, 0, std::max<size_t>((size_t)h->M, 1), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_beta[0], 0, KPV * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_beta[1], 0, KPV * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_lse, 0, std::max<size_t>(NZ, 1) * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_ksum, 0, sizeof(double), ctx->stream));
    TMVB_HIP(hipEventCreate(&h->ev0));
    TMVB_HIP(hipEventCreate(&h->ev1));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    // constructor state, src/fLDA.jl:38-52: eta = 0.5, alpha = 1, gamma = 1, Elogtheta = psi(1) - psi(K), tau = eta;
    // kappa and beta are drawn from Dirichlet(V, 1) with Julia's RNG in the reference: uniform until tmvb_flda_set_state
    const double eta0 = 0.5;
    std::vector<double> alpha(K, 1.0), kap((size_t)h->V, h->V ? 1.0 / (double)h->V : 0.0), beta((size_t)K * h->V, h->V ? 1.0 / (double)h->V : 0.0);
    std::vector<double> gamma(KM, 1.0), elog(KM, -0.5772156649015329 - tmvb_digamma_host((double)K)), tau(NZ, eta0);
    rc = tmvb_flda_set_state(h, &eta0, alpha.data(), kap.data(), nullptr, beta.data(), nullptr, gamma.data(), elog.data(), nullptr, tau.data(), nullptr, nullptr);
    if (rc) return rc;
    guard.release();
    *out = h;
    return TMVB_OK;
}

// update_buffer!-style state upload (the reference has no device model for fLDA; the fields are those of src/fLDA.jl:6-29).
// NULL = unchanged; *_old default to the current values.  tau / tau_old: [nnz] in CSR token order.
extern "C" int tmvb_flda_set_state(tmvb_flda* h, const double* eta, const double* alpha, const double* kappa, const double* kappa_old,
                                   const double* beta, const double* beta_old, const double* gamma, const double* Elogtheta,
                                   const double* Elogtheta_old, const double* tau, const double* tau_old, const double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_flda_set_state: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const size_t K = h->K, KM = K * (size_t)h->M, V = (size_t)h->V, NZ = (size_t)h->nnz;
    int rc;
    if (eta) {
        TMVB_REQUIRE(*eta >= 0.0 && *eta <= 1.0, TMVB_ESHAPE, "eta must belong to the interval [0,1].");     // src/modelutils.jl:75
        TMVB_HIP(hipMemcpyAsync(h->d_eta, eta, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        TMVB_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (alpha) {
        for (size_t i = 0; i < K; ++i) TMVB_REQUIRE(std::isfinite(alpha[i]) && alpha[i] > 0.0, TMVB_ENONFINITE, "alpha must be finite and positive.");
        TMVB_HIP(hipMemcpyAsync(h->d_alpha_d, alpha, K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if ((rc = upload_f32(ctx, h->d_alpha_f, alpha, K))) return rc;
    }
    if (kappa) {
        if ((rc = upload_f32(ctx, h->d_kappa, kappa, V))) return rc;
        if (!kappa_old && (rc = upload_f32(ctx, h->d_kappa_old, kappa, V))) return rc;
    }
    if (kappa_old && (rc = upload_f32(ctx, h->d_kappa_old, kappa_old, V))) return rc;
    if (beta) {
        if ((rc = flda_upload_beta(h, h->d_beta[h->cur], beta))) return rc;
        if (!beta_old && (rc = flda_upload_beta(h, h->d_beta[h->cur ^ 1], beta))) return rc;
        if ((rc = flda_refresh_L(h))) return rc;
    }
    if (beta_old && (rc = flda_upload_beta(h, h->d_beta[h->cur ^ 1], beta_old))) return rc;
    if (gamma && (rc = upload_f32(ctx, h->d_gamma, gamma, KM))) return rc;
    if (Elogtheta) {
        if ((rc = upload_f32(ctx, h->d_elog, Elogtheta, KM))) return rc;
        if (!Elogtheta_old && (rc = upload_f32(ctx, h->d_elog_old, Elogtheta, KM))) return rc;
    }
    if (Elogtheta_old && (rc = upload_f32(ctx, h->d_elog_old, Elogtheta_old, KM))) return rc;
    if (tau) {
        for (size_t q = 0; q < NZ; ++q) TMVB_REQUIRE(tau[q] >= 0.0 && tau[q] <= 1.0, TMVB_ESHAPE, "tau must belong to the interval [0,1].");
        if ((rc = upload_f32(ctx, h->d_tau, tau, NZ))) return rc;
        if (!tau_old && (rc = upload_f32(ctx, h->d_tau_old, tau, NZ))) return rc;
    }
    if (tau_old && (rc = upload_f32(ctx, h->d_tau_old, tau_old, NZ))) return rc;
    if (elbo) h->elbo = *elbo;
    h->parts_pending = false; h->parts_valid = false;      // a state set by the host: update_elbo! takes the token walk
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_flda_get_state(tmvb_flda* h, double* eta, double* alpha, double* kappa, double* kappa_old, double* beta, double* beta_old,
                                   double* gamma, double* Elogtheta, double* Elogtheta_old, double* tau, double* tau_old, double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_flda_get_state: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const size_t K = h->K, KM = K * (size_t)h->M, V = (size_t)h->V, NZ = (size_t)h->nnz;
    int rc;
    if (eta) TMVB_HIP(hipMemcpyAsync(eta, h->d_eta, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (alpha) TMVB_HIP(hipMemcpyAsync(alpha, h->d_alpha_d, K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    if (kappa && (rc = download_f32(ctx, kappa, h->d_kappa, V))) return rc;
    if (kappa_old && (rc = download_f32(ctx, kappa_old, h->d_kappa_old, V))) return rc;
    if (beta && (rc = flda_download_beta(h, beta, h->d_beta[h->cur]))) return rc;
    if (beta_old && (rc = flda_download_beta(h, beta_old, h->d_beta[h->cur ^ 1]))) return rc;
    if (gamma && (rc = download_f32(ctx, gamma, h->d_gamma, KM))) return rc;
    if (Elogtheta && (rc = download_f32(ctx, Elogtheta, h->d_elog, KM))) return rc;
    if (Elogtheta_old && (rc = download_f32(ctx, Elogtheta_old, h->d_elog_old, KM))) return rc;
    if (tau && (rc = download_f32(ctx, tau, h->d_tau, NZ))) return rc;
    if (tau_old && (rc = download_f32(ctx, tau_old, h->d_tau_old, NZ))) return rc;
    if (elbo) *elbo = h->elbo;
    return TMVB_OK;
}

// update_phi! / update_tau! / update_gamma! / update_Elogtheta! sweeps + update_beta!(model, d) + update_kappa!(model, d) for
// every document (src/fLDA.jl:222-236).  Asynchronous on the context's stream.
extern "C" int tmvb_flda_estep(tmvb_flda* h, int32_t viter, double vtol)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_flda_estep: handle is NULL");
    TMVB_REQUIRE(viter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative.");     // src/fLDA.jl:216
    TMVB_REQUIRE(vtol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");      // :215
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    FldaParams p;
    p.K = h->K; p.KP = h->KP; p.LPR = h->KP / 4; p.lpr_magic = (unsigned)(0x100000000ull / (unsigned)p.LPR) + 1u;
    p.doc_ptr = h->corp->d_doc_ptr; p.terms = h->corp->d_terms; p.counts = h->corp->d_counts; p.doc_order = h->d_doc_order;
    p.L = h->d_L; p.kappa = h->d_kappa; p.eta = h->d_eta; p.alpha = h->d_alpha_f;
    p.gamma = h->d_gamma; p.elog = h->d_elog; p.elog_old = h->d_elog_old; p.tau = h->d_tau; p.tau_old = h->d_tau_old; p.lse = h->d_lse; p.aold = h->d_aold;
    p.sweeps = h->d_sweeps; p.viter = viter; p.vtol = (float)vtol;
    // decomposed update_elbo!: an iteration that will be checked keeps the alpha its update_gamma! adds (viter = 0 leaves no responsibilities: token walk)
    const bool collect = (h->parts_env == 2 || (h->parts_env != 0 && h->want_parts)) && viter > 0;
    h->parts_pending = false; h->parts_valid = false;
    if (collect) TMVB_HIP(hipMemcpyAsync(h->d_alpha_e, h->d_alpha_f, (size_t)h->K * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    TMVB_HIP(hipEventRecord(h->ev0, ctx->stream));
    for (const tmvb_bucket& b : h->buckets) {
        const size_t lds = flda_tile_bytes(b.tile_rows, h->KP);
        const dim3 grid((unsigned)b.count), block(64);
        int lrc = dispatch_nslot(h->nslot, [&](auto ns) -> int {            // 1, 2, 4, 8 or 16 topic slots per lane (K <= 1024, as LDA)
            constexpr int NS = decltype(ns)::value;
            if (lds > 48 * 1024) TMVB_HIP(hipFuncSetAttribute((const void*)flda_estep_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((flda_estep_kernel<NS>), grid, block, lds, ctx->stream, p, b.first, b.tile_rows);
            return TMVB_OK;
        });
        if (lrc) return lrc;
        TMVB_HIP(hipGetLastError());
    }
    {
        FldaStatsParams sp;
        sp.K = h->K; sp.KP = h->KP; sp.L = h->d_L; sp.elog_old = h->d_elog_old; sp.tau = h->d_tau; sp.tau_old = h->d_tau_old; sp.lse = h->d_lse;
        sp.S = h->d_stats; sp.kstat = h->kstat(); sp.partial = h->d_ts_partial;
        int rc = tmvb_launch_filtered_stats(ctx, h->nslot, h->corp->term_index, sp);
        if (rc) return rc;
    }
    TMVB_HIP(hipEventRecord(h->ev1, ctx->stream));
    h->timed = true;
    h->parts_pending = collect;
    return TMVB_OK;
}

// Elogtheta_sum (update_alpha!'s input, src/fLDA.jl:129) into the statistics tail
extern "C" int tmvb_flda_reduce_docs(tmvb_flda* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_flda_reduce_docs: handle is NULL");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    return tmvb_colsum(h->ctx, h->nslot, h->K, h->d_elog, h->M, h->d_partial, h->d_esum, h->esum_f());
}

extern "C" int tmvb_flda_stats(tmvb_flda* h, void** dev_ptr, int64_t* n_f32)
{
    TMVB_REQUIRE(h && dev_ptr && n_f32, TMVB_EINVAL, "tmvb_flda_stats: NULL argument");
    *dev_ptr = h->d_stats;
    *n_f32 = h->stats_len();
    return TMVB_OK;
}

// Attach / detach a communicator: this handle holds one document shard of a corpus of M_total documents and C_total tokens.
extern "C" int tmvb_flda_set_comm(tmvb_flda* h, tmvb_comm* comm, int64_t M_total, int64_t C_total)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_flda_set_comm: handle is NULL");
    if (comm) {
        TMVB_REQUIRE(M_total >= h->M && C_total >= h->corp->info.sum_counts, TMVB_ESHAPE, "tmvb_flda_set_comm: totals smaller than the local shard");
        h->M_total = M_total; h->C_total = (double)C_total;
    } else {
        h->M_total = h->M; h->C_total = (double)h->corp->info.sum_counts;
    }
    h->distributed = comm != nullptr;
    h->comm = comm;
    return TMVB_OK;
}

// update_beta!(model) (src/fLDA.jl:152-156) and update_kappa!(model) (:138-142), which train! runs back to back (:237-238);
// refreshes the log table.
extern "C" int tmvb_flda_update_beta(tmvb_flda* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_flda_update_beta: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    int rc = tmvb_colsum(ctx, h->nslot, h->K, h->d_stats, h->V, h->d_partial, h->d_rowsum, nullptr);
    if (rc) return rc;
    const int64_t total = (int64_t)h->KP * h->V;
    const int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (total + 255) / 256));
    // behind a collecting E-step: the partial sums of S (log(beta_new + eps) - log(beta_old + eps)), both logarithms in fp64 (tmvb_common_kernels.h);
    // a sharded handle holds the all-reduced S here and contributes the share M / M_total of the global term (tmvb_flda_update_elbo)
    const bool pw = h->parts_pending;
    hipLaunchKernelGGL(beta_norm_kernel, dim3(nb), dim3(256), (size_t)h->K * sizeof(double), ctx->stream,
                       h->d_stats, h->d_rowsum, h->d_beta[h->cur ^ 1], h->K, h->KP, h->V, pw ? h->d_pw_partial : (double*)nullptr, pw ? TMVB_EPS_F : 0.0f,
                       pw ? (const float*)h->d_beta[h->cur] : (const float*)nullptr);
    TMVB_HIP(hipGetLastError());
    h->pw_blocks = nb; h->parts_valid = pw; h->parts_pending = false;
    h->cur ^= 1;                                            // beta_old <- beta, beta <- new
    if ((rc = flda_refresh_L(h))) return rc;
    hipLaunchKernelGGL(flda_kappa_eta_kernel, dim3(1), dim3(1024), 0, ctx->stream, h->kstat(), h->d_kappa, h->d_kappa_old, h->V, h->C_total,
                       h->d_eta, 1, 0, h->d_ksum);
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

// update_alpha! (src/fLDA.jl:128-150): the LDA Newton iteration
extern "C" int tmvb_flda_update_alpha(tmvb_flda* h, int32_t niter, double ntol)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_flda_update_alpha: handle is NULL");
    TMVB_REQUIRE(niter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative.");
    TMVB_REQUIRE(ntol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const float* ef = h->distributed ? h->esum_f() : nullptr;      // sharded: the all-reduced fp32 tail
    (void)dispatch_nslot(h->nslot, [&](auto ns) -> int {
        hipLaunchKernelGGL((lda_alpha_kernel<decltype(ns)::value>), dim3(1), dim3(64), 0, ctx->stream, h->K, (double)h->M_total, h->d_esum, ef, h->d_alpha_d, h->d_alpha_f, niter, ntol, h->d_iters);
        return TMVB_OK;
    });
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

// update_eta! (src/fLDA.jl:122-124), from the background mass that update_kappa! summed
extern "C" int tmvb_flda_update_eta(tmvb_flda* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_flda_update_eta: handle is NULL");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    hipLaunchKernelGGL(flda_kappa_eta_kernel, dim3(1), dim3(1024), 0, h->ctx->stream, h->kstat(), h->d_kappa, h->d_kappa_old, h->V, h->C_total,
                       h->d_eta, 0, 1, h->d_ksum);
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

// update_elbo! (src/fLDA.jl:108-118) on the device; the sum over this context's documents
extern "C" int tmvb_flda_update_elbo(tmvb_flda* h, double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_flda_update_elbo: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const bool parts = h->parts_valid && !h->force_walk && h->parts_env != 0;
    h->elbo_form = parts ? 1 : 0;
    if (parts) {
        if (h->M > 0) {
            const dim3 grid((unsigned)h->M), block(64);
            (void)dispatch_nslot(h->nslot, [&](auto ns) -> int {
                hipLaunchKernelGGL((flda_elbo_doc_parts_kernel<decltype(ns)::value>), grid, block, 0, ctx->stream, h->K, h->corp->d_doc_ptr, h->corp->d_terms,
                                   h->corp->d_counts, h->d_alpha_d, h->d_alpha_e, h->d_eta, h->d_kappa, h->d_gamma, h->d_elog, h->d_elog_old, h->d_tau,
                                   h->d_tau_old, h->d_lse, h->d_aold, h->d_doc_val);
                return TMVB_OK;
            });
            TMVB_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(lda_elbo_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, h->d_doc_val, h->M, h->K, h->d_alpha_d, h->d_elbo,
                           (const double*)h->d_pw_partial, h->pw_blocks, h->distributed && h->M_total > 0 ? (double)h->M / (double)h->M_total : 1.0);
        TMVB_HIP(hipGetLastError());
        double v = 0.0;
        TMVB_HIP(hipMemcpyAsync(&v, h->d_elbo, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        TMVB_HIP(hipStreamSynchronize(ctx->stream));
        h->elbo = v;
        if (elbo) *elbo = v;
        return TMVB_OK;
    }
    if (h->M > 0) {
        const dim3 grid((unsigned)h->M), block(64);
        (void)dispatch_nslot(h->nslot, [&](auto ns) -> int {
            hipLaunchKernelGGL((flda_elbo_kernel<decltype(ns)::value>), grid, block, 0, ctx->stream, h->K, h->KP, h->corp->d_doc_ptr, h->corp->d_terms, h->corp->d_counts,
                               h->d_alpha_d, h->d_eta, h->d_kappa, h->d_beta[h->cur], h->d_beta[h->cur ^ 1], h->d_gamma, h->d_elog, h->d_elog_old,
                               h->d_tau, h->d_tau_old, h->d_doc_val);
            return TMVB_OK;
        });
        TMVB_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(lda_elbo_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, h->d_doc_val, h->M, h->K, h->d_alpha_d, h->d_elbo,
                       (const double*)nullptr, 0, 1.0);
    TMVB_HIP(hipGetLastError());
    double v = 0.0;
    TMVB_HIP(hipMemcpyAsync(&v, h->d_elbo, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    h->elbo = v;
    if (elbo) *elbo = v;
    return TMVB_OK;
}

namespace {
struct FldaTrainOps {
    int niter, viter; double ntol, vtol;
    int estep(tmvb_flda* h) { return tmvb_flda_estep(h, viter, vtol); }                  // src/fLDA.jl:222-236
    int reduce(tmvb_flda* h) { return tmvb_flda_reduce_docs(h); }
    int before_allreduce(tmvb_flda*) { return TMVB_OK; }
    float* stats(tmvb_flda* h) { return h->d_stats; }
    int64_t stats_len(tmvb_flda* h) { return h->stats_len(); }
    int mstep(tmvb_flda* h)
    {
        int rc = tmvb_flda_update_beta(h);                                               // :237-238 (beta, kappa)
        if (!rc) rc = tmvb_flda_update_alpha(h, niter, ntol);                            // :239
        if (!rc) rc = tmvb_flda_update_eta(h);                                           // :240
        return rc;
    }
    int elbo_local(tmvb_flda* h, double* s, double* once) { *once = 0.0; return tmvb_flda_update_elbo(h, s); }
    int elbo_form(tmvb_flda* h) { return h->elbo_form; }
    void force_walk(tmvb_flda* h, bool on, bool doubled = true) { h->force_walk = on; if (!on && doubled) h->elbo_form = 1; }   // (as LdaTrainOps)
    void will_check(tmvb_flda* h, bool checked) { h->want_parts = checked; }
    double* elbo_dev(tmvb_flda* h) { return h->d_elbo; }
    tmvb_comm* comm(tmvb_flda* h) { return h->comm; }
    bool distributed(tmvb_flda* h) { return h->distributed; }
    tmvb_ctx* ctx(tmvb_flda* h) { return h->ctx; }
    int64_t nnz(tmvb_flda* h) { return h->corp->info.nnz; }
    void set_elbo(tmvb_flda* h, double v) { h->elbo = v; }
    double get_elbo(tmvb_flda* h) { return h->elbo; }
    int finish(tmvb_flda* h)
    {
        TMVB_HIP(hipSetDevice(h->ctx->device));
        TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
        return TMVB_OK;
    }
};
}  // namespace

// NOTE on the sharded ELBO: lda_elbo_final_kernel adds M (lgamma(sum alpha) - sum lgamma(alpha)) with the LOCAL M, so the
// ranks' values add up to the corpus ELBO.
extern "C" int tmvb_flda_train_group(tmvb_flda* const* hs, int32_t n, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter,
                                     double vtol, int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline)
{
    TMVB_REQUIRE(tol >= 0 && ntol >= 0 && vtol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");   // src/fLDA.jl:215
    TMVB_REQUIRE(iter >= 0 && niter >= 0 && viter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative."); // :216
    FldaTrainOps ops{niter, viter, ntol, vtol};
    return tmvb_train_group_loop("tmvb_flda_train", hs, n, iter, tol, checkelbo, elbo_traj, iters_done, elbo_baseline, ops);
}

// train! (src/fLDA.jl:213-247)
extern "C" int tmvb_flda_train(tmvb_flda* h, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter, double vtol,
                               int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_flda_train: handle is NULL");
    return tmvb_flda_train_group(&h, 1, iter, tol, niter, ntol, viter, vtol, checkelbo, elbo_traj, iters_done, elbo_baseline);
}

extern "C" int tmvb_flda_elbo_form(tmvb_flda* h, int32_t* form)
{
    TMVB_REQUIRE(h && form, TMVB_EINVAL, "tmvb_flda_elbo_form: NULL argument");
    *form = h->elbo_form;
    return TMVB_OK;
}

extern "C" int tmvb_flda_doc_sweeps(tmvb_flda* h, uint8_t* out)
{
    TMVB_REQUIRE(h && (out || h->M == 0), TMVB_EINVAL, "tmvb_flda_doc_sweeps: NULL argument");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    if (h->M) TMVB_HIP(hipMemcpyAsync(out, h->d_sweeps, (size_t)h->M, hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_flda_last_estep_ms(tmvb_flda* h, float* ms)
{
    TMVB_REQUIRE(h && ms, TMVB_EINVAL, "tmvb_flda_last_estep_ms: NULL argument");
    TMVB_REQUIRE(h->timed, TMVB_EINVAL, "tmvb_flda_last_estep_ms: no E-step has run");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    TMVB_HIP(hipEventSynchronize(h->ev1));
    TMVB_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return TMVB_OK;
}
