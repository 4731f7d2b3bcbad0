// tmvb_ctm.hip -- correlated topic model (CTM) variational-Bayes engine for gfx950 (MI355X).
//
// Path: the per-document coordinate ascent of src/CTM.jl:194-205 (update_phi! :175, update_logzeta!
// :169, update_vsq! :146, update_lambda! :129, exit test :200, update_beta!(d) :122) fused into ONE
// kernel per document, plus the M-step (update_beta! :114, update_sigma! :108 -- which uses the
// PREVIOUS mu, quirk Q2 -- update_mu! :102) and update_elbo! (:89) on the device.  It replaces the
// nine OpenCL kernels of src/gpuCTM.jl:144-473 but follows the CPU path's semantics (per-document
// exit rule, exact Newton steps), not the OpenCL path's (global median rule :503, softmax max
// initialised to 0 :404,:456, host-side fp32 `inv` :203-205).
//
// Per-document kernel (one 64-lane wave = one document, lane = topic = matrix row):
//   * phi = softmax_K(log beta[:,terms] + lambda) is evaluated in linear space through the same LDS
//     topic tile as the LDA fallback kernel: with e_i = exp(lambda_i - max lambda),
//     s_n = sum_i B[n][i] e_i, w_n = c_n / s_n, (phi*counts)_i = e_i sum_n w_n B[n][i]; the tile is
//     gathered once per outer iteration by LDS-DMA and reused by every sweep.
//   * logzeta, the K scalar Newton iterations on vsq and the gradient of the lambda Newton step are
//     evaluated in fp64 (the exit tests ||g|| < ntol sit at the fp32 noise floor for large documents).
//   * the K x K Newton system (invsigma + C_d diag(e^{...})) delta = g is solved by Gauss-Jordan
//     elimination held ENTIRELY IN REGISTERS: lane i owns row i (KP VGPRs), the pivot row is broadcast
//     with v_readlane (SGPR operands of the FMAs), ~KP^2 VALU instructions per solve, no LDS, no
//     pivoting (the matrix is SPD).  The reference solves this with LAPACK on the CPU (:136) or a
//     K-work-item Gauss-Jordan in OpenCL local memory (src/utils.jl:60-90).
//   * the statistics scatter update_beta!(d) is deferred to the gather-side pass (tmvb_termstats.h).
// M-step: the K x M * M x K scatter matrix sum_d (lambda_d - mu)(lambda_d - mu)^T runs on f32 MFMA
// (v_mfma_f32_32x32x2_f32); sigma assembly, its inverse and log-determinant run in fp64 in one
// workgroup.
#define TMVB_TS_LOGZ 1            // the log-normaliser forms of the statistics pass (decomposed update_elbo!, see ctm_elbo_kernel)
#include "tmvb_common_kernels.h"
#include "tmvb_train.h"
#include "tmvb_filtered.h"

#include <utility>

#define CTM_MAX_K 256

struct CtmParams {
    int K, KP, LPR;
    unsigned lpr_magic;
    const int64_t* doc_ptr;
    const int32_t* terms;
    const int32_t* counts;
    const int32_t* doc_order;
    const int32_t* tok_inv;
    const float* beta;        // [V][KP] padded gather layout
    const float* invsigma;    // [KP][KP] fp32, pads zero (symmetric)
    const float* mu;          // [K]
    float* lambda;            // [M][K]
    float* lambda_old;        // [M][K]
    float* vsq;               // [M][K]
    float* logzeta;           // [M]
    float* wtok;              // [nnz] term-major
    float* E;                 // [M][estride]
    int estride;              // KP (zero pads) while a row has at most 64 sixteen-byte chunks, K beyond (ctm_estep_generic_kernel honours it; the
                              // K <= 60 kernels write KP-strided rows, which is the same thing there)
    uint8_t* sweeps;
    unsigned long long* newton_steps;
    int niter;
    double ntol;
    int viter;
    double vtol;
    int debug;                // TMVB_DEBUG_FLAGS (profiling experiments only): 1 = skip vsq Newton, 2 = diagonal solve instead of GJ
    int store_w;              // KP > 128: the statistics pass reads stored per-token weights instead of recomputing them
    // filtered CTM only (src/fCTM.jl)
    const float* L = nullptr;         // [V][KP] log(beta + eps), pads 0
    const float* kappa = nullptr;     // [V]
    float eta = 0.5f;
    float* tau = nullptr; float* tau_old = nullptr; float* lse = nullptr;     // [nnz], CSR order
    float* aold = nullptr;            // [nnz] or NULL: sum_i phi_in log(beta[i, t_n] + eps) of each sweep (the exponent update_tau! forms), for the decomposed update_elbo! of fCTM
    uint16_t* doc_newton = nullptr;   // [M] lambda-Newton steps of each document in this E-step (lane-per-document kernel: next E-step's grouping key)
    // ctm_estep_generic_kernel<.., CG = true> only: the document queue of the persistent launch and the CG stopping rule
    unsigned* queue = nullptr;        // next position of doc_order to take (zeroed before the launch)
    int64_t n_docs = 0;
    float cg_tol2 = 1e-8f, cg_abs2 = 0.0f;
    int cg_maxit = 200;
    int tile_rows = 32;               // rows of a wave's topic tile window
    unsigned long long* cg_diag = nullptr;   // [0] CG trips, [1] Newton trips, [2] documents
    // decomposed update_elbo! (ctm_elbo_doc_kernel): [M] or NULL -- sum_i (phi counts)_i (lambda_i - lambda_old_i) of each document's LAST executed sweep,
    // the one quantity of the token terms that only the E-step kernel has (phi counts lives in its registers); CTM only
    float* pdot = nullptr;
};

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double readlane_d(double v, int l)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// Gauss-Jordan solve of H x = g, lane i owns row i (registers H[0..R)), SPD, no pivoting.
// At pivot j the pivot row's entries reach every lane as SGPR operands (v_readlane of lane j).
typedef float gj_v2f __attribute__((ext_vector_type(2)));

// One pivot of the elimination with the pivot index a template constant (a `#pragma unroll` loop over 52 pivots with
// blocked broadcasts inside is not fully unrolled by hipcc 7.2, and a dynamic pivot index sends the row to scratch).
// The row is held as column pairs so that the rank-1 update runs on v_pk_fma_f32 with the pivot row's entries as
// SGPR pairs (the kernel is VALU-issue bound: packed fp32 halves the update's instruction count).  The pivot row is
// broadcast in blocks of <= 16 pairs: back-to-back v_readlane into SGPRs, then the FMAs that consume them (a VALU
// that reads an SGPR written by the immediately preceding v_readlane costs a wait state per pair; 52 live SGPRs
// would spill to VGPR lanes).
// (lane == J) ? a : b with the lane mask built in VCC by two SALU instructions inside the asm statement.  Written as
// `lane == J` the 52 + 52 compare results are loop invariant: the compiler hoists them out of the Newton loop, spills
// the 208 mask SGPRs to VGPR lanes and reloads each with two v_readlane -- ~150 extra VALU instructions per solve (a
// literal "s" operand is hoisted and spilled the same way).
template <int J>
__device__ __forceinline__ float select_lane(float a, float b)
{
    float r;
    asm("s_mov_b64 vcc, 0\n\ts_bitset1_b64 vcc, %3\n\tv_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(r) : "v"(b), "v"(a), "n"(J) : "vcc");
    return r;
}

template <int R, int J>
__device__ __forceinline__ void gj_pivot(gj_v2f (&H2)[R / 2], float& g, float& dinv, int lane)
{
    typedef gj_v2f v2f;
    constexpr int jp = J / 2;                               // pair holding column J
    constexpr int NQ = R / 2 - (jp + 1);                    // pairs right of the pivot's pair
    const float pj = readlane_f((J & 1) ? H2[jp].y : H2[jp].x, J);
    const float pj1 = (J & 1) ? 0.0f : readlane_f(H2[jp].y, J);
    const float sg = readlane_f(g, J);
    // v_rcp_f32 is within 1 ulp: the multipliers carry a 1e-7 relative error like every other fp32 operation of the
    // elimination, and the two refinement instructions per pivot are saved
    const float rp = __builtin_amdgcn_rcpf(pj);
    const float hj = (J & 1) ? H2[jp].y : H2[jp].x;
    // Gauss-JORDAN: row J stays live (later pivots keep eliminating their columns from it), so lane J sits out its own
    // step (f = 0) and remembers the reciprocal of its pivot for the final x_J = g_J / pivot_J
    const float f = select_lane<J>(0.0f, hj * rp);
    dinv = select_lane<J>(rp, dinv);
    const float nf = -f;
    if ((J & 1) == 0) H2[jp].y = fmaf(nf, pj1, H2[jp].y);           // column J + 1 shares J's pair
    g = fmaf(nf, sg, g);
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += 16) {
        float sx[16], sy[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (q0 + q < NQ) { sx[q] = readlane_f(H2[jp + 1 + q0 + q].x, J); sy[q] = readlane_f(H2[jp + 1 + q0 + q].y, J); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (q0 + q < NQ) H2[jp + 1 + q0 + q] = __builtin_elementwise_fma(v2f{nf, nf}, v2f{sx[q], sy[q]}, H2[jp + 1 + q0 + q]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int R, int... J>
__device__ __forceinline__ void gj_all_pivots(gj_v2f (&H2)[R / 2], float& g, float& dinv, int lane, std::integer_sequence<int, J...>)
{
    (gj_pivot<R, J>(H2, g, dinv, lane), ...);
}

template <int R, int... Q>
__device__ __forceinline__ void gj_patch_diag(const float (&H)[R], float diag, gj_v2f (&H2)[R / 2], std::integer_sequence<int, Q...>)
{
    ((H2[Q] = gj_v2f{select_lane<2 * Q>(diag, H[2 * Q]), select_lane<2 * Q + 1>(diag, H[2 * Q + 1])}), ...);
}

// H: row `lane` of the matrix WITHOUT its diagonal element, which arrives separately (diag) and is patched in here.
template <int R>
__device__ __forceinline__ float gj_solve_rows(float (&H)[R], float diag, float g, int lane)
{
    static_assert(R % 2 == 0, "gj_solve_rows: R must be even");
    gj_v2f H2[R / 2];
    gj_patch_diag<R>(H, diag, H2, std::make_integer_sequence<int, R / 2>{});
    float dinv = 0.0f;
    gj_all_pivots<R>(H2, g, dinv, lane, std::make_integer_sequence<int, R>{});
    return g * dinv;
}

// ------------------------------------------------------------------------------ E-step kernel
// FILT = true is the filtered CTM (src/fCTM.jl, new device path -- the reference has none): phi carries the per-token
// switch as an exponent, phi[i,n] = softmax_i(tau_n log(beta[i,t_n] + eps) + lambda_i) (:216-219), so the tile holds rows
// of L = log(beta + eps) and the token phase is the two-pass column softmax of tmvb_flda.hip with update_tau! (:208-213)
// fused in; the sweep order is phi, tau, logzeta, LAMBDA, VSQ (:236-241) instead of CTM's phi, logzeta, vsq, lambda.
// The Newton machinery is shared.
static size_t ctm_tile_bytes(int rows, int KP, bool filt) { return tmvb_tile_bytes(rows, KP) + (filt ? 3 * (size_t)rows * sizeof(float) : 0); }

template <int R, bool FILT>
__global__ __launch_bounds__(64) void ctm_estep_kernel(CtmParams p, int64_t first, int tile_rows)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LPR = R / 4, KP = R;
    const int lane = threadIdx.x;
    const int K = p.K;
    float* Bt = lds;
    float* e_l = Bt + (size_t)tile_rows * KP;     // CTM: exp(lambda - max); fCTM: lambda itself (pads -inf)
    float* w_l = e_l + KP;
    float* c_l = w_l + tile_rows;
    int* t_l = (int*)(c_l + tile_rows);
    float* tp_l = (float*)(t_l + tile_rows);      // FILT only: tau entering the sweep, tau after update_tau!, column maximum
    float* tn_l = tp_l + tile_rows;
    float* m_l = tn_l + tile_rows;

    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d];
    const int N = (int)(p.doc_ptr[d + 1] - off);
    const bool single = N <= tile_rows;
    const bool on = lane < K;

    // C_d = sum of counts (src/CTM.jl:33)
    float cl = 0.0f;
    for (int n = lane; n < N; n += 64) cl += (float)p.counts[off + n];
    const double Cd = (double)wave_sum(cl);

    // row `lane` of invsigma is re-read (L1/L2 resident, 10 KB) at every Newton step instead of being
    // pinned in KP registers: the register budget decides the waves per SIMD
    const float* isrow = p.invsigma + (size_t)min(lane, KP - 1) * KP;
    const double isdiag = on ? (double)p.invsigma[(size_t)lane * KP + lane] : 1.0;
    const float isdiag_f = (lane < KP) ? p.invsigma[(size_t)lane * KP + lane] : 0.0f;   // the fp32 entry the row load would bring (pads: 0)
    const double mu = on ? (double)p.mu[lane] : 0.0;
    double lam = on ? (double)p.lambda[(int64_t)d * K + lane] : 0.0;
    double vs = on ? (double)p.vsq[(int64_t)d * K + lane] : 1.0;
    double lam_old = lam, lz = (double)p.logzeta[d];
    float e = 0.0f;

    auto load_chunk = [&](int c0, int rows) {
        for (int n = lane; n < rows; n += 64) {
            t_l[n] = p.terms[off + c0 + n];
            c_l[n] = (float)p.counts[off + c0 + n];
            if constexpr (FILT) tn_l[n] = p.tau[off + c0 + n];
        }
        WAVE_LDS_FENCE();
        const float* table = FILT ? p.L : p.beta;
        const int nch = rows * LPR;
#pragma unroll 4
        for (int f0 = 0; f0 < nch; f0 += 64) {
            const int f = f0 + lane;
            if (f < nch) {
                const int n = (LPR == 1) ? f : (int)__umulhi((unsigned)f, p.lpr_magic);
                const int c = f - n * LPR;
                const float* src = table + ((int64_t)t_l[n] * KP + 4 * c);
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(Bt + (size_t)f0 * 4), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WAVE_LDS_FENCE();
    };
    auto phase1 = [&](int rows) {
        const float4* er = (const float4*)e_l;
        for (int n = lane; n < rows; n += 64) {
            const float4* br = (const float4*)(Bt + (size_t)n * KP);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int q = 0; q < LPR; ++q) {
                float4 b = br[q], ev = er[q];
                s0 = fmaf(b.x, ev.x, s0); s1 = fmaf(b.y, ev.y, s1);
                s2 = fmaf(b.z, ev.z, s2); s3 = fmaf(b.w, ev.w, s3);
            }
            w_l[n] = c_l[n] / ((s0 + s1) + (s2 + s3));
        }
        WAVE_LDS_FENCE();
    };
    const int r4 = lane & 3, ql = lane >> 2;
    auto phase2 = [&](int rows, float4& acc) {
        const int nfull = rows >> 2;
        if (ql < LPR) {
#pragma unroll 4
            for (int m = 0; m < nfull; ++m) {
                const int n = 4 * m + r4;
                const float w = w_l[n];
                const float4 b = *(const float4*)(Bt + (size_t)n * KP + 4 * ql);
                acc.x = fmaf(w, b.x, acc.x); acc.y = fmaf(w, b.y, acc.y);
                acc.z = fmaf(w, b.z, acc.z); acc.w = fmaf(w, b.w, acc.w);
            }
            const int n = 4 * nfull + r4;
            if (n < rows) {
                const float w = w_l[n];
                const float4 b = *(const float4*)(Bt + (size_t)n * KP + 4 * ql);
                acc.x = fmaf(w, b.x, acc.x); acc.y = fmaf(w, b.y, acc.y);
                acc.z = fmaf(w, b.z, acc.z); acc.w = fmaf(w, b.w, acc.w);
            }
        }
    };
    // fCTM: column softmax normalisers (update_phi!, src/fCTM.jl:216-219) and update_tau! (:208-213), lane = token
    auto filt_a = [&](int c0, int rows, bool store) {
        const float4* er = (const float4*)e_l;
        for (int n = lane; n < rows; n += 64) {
            const float4* br = (const float4*)(Bt + (size_t)n * KP);
            const float tp = tn_l[n];
            float m = -INFINITY;
#pragma unroll
            for (int q = 0; q < LPR; ++q) {
                const float4 b = br[q], ev = er[q];
                m = fmaxf(m, fmaxf(fmaxf(fmaf(tp, b.x, ev.x), fmaf(tp, b.y, ev.y)), fmaxf(fmaf(tp, b.z, ev.z), fmaf(tp, b.w, ev.w))));
            }
            float s = 0.0f, a = 0.0f;
#pragma unroll
            for (int q = 0; q < LPR; ++q) {
                const float4 b = br[q], ev = er[q];
                const float p0 = __expf(fmaf(tp, b.x, ev.x) - m), p1 = __expf(fmaf(tp, b.y, ev.y) - m);
                const float p2 = __expf(fmaf(tp, b.z, ev.z) - m), p3 = __expf(fmaf(tp, b.w, ev.w) - m);
                s += (p0 + p1) + (p2 + p3);
                a = fmaf(p0, b.x, a); a = fmaf(p1, b.y, a); a = fmaf(p2, b.z, a); a = fmaf(p3, b.w, a);
            }
            const float prod = __expf(fminf(-(a / s), 87.0f));           // prod_i beta^-phi  (:212)
            const float tnew = p.eta / (TMVB_EPS_F + (p.eta + (1.0f - p.eta) * (p.kappa[t_l[n]] * prod)));
            tp_l[n] = tp; tn_l[n] = tnew; m_l[n] = m; w_l[n] = c_l[n] / s;
            if (store) { p.tau_old[off + c0 + n] = tp; p.tau[off + c0 + n] = tnew; p.lse[off + c0 + n] = m + __logf(s); }
            if (p.aold) p.aold[off + c0 + n] = a / s;                    // every sweep (the last executed one remains): one float per token
        }
        WAVE_LDS_FENCE();
    };
    // fCTM: (phi * counts)_i, lane = (topic quad, token residue), one v_exp_f32 per (token, topic)
    auto filt_b = [&](int rows, float4& acc) {
        if (ql >= LPR) return;
        const float4 ev = *(const float4*)(e_l + 4 * ql);
        for (int n = r4; n < rows; n += 4) {
            const float tp = tp_l[n], m = m_l[n], w = w_l[n];
            const float4 b = *(const float4*)(Bt + (size_t)n * KP + 4 * ql);
            acc.x = fmaf(w, __expf(fmaf(tp, b.x, ev.x) - m), acc.x);
            acc.y = fmaf(w, __expf(fmaf(tp, b.y, ev.y) - m), acc.y);
            acc.z = fmaf(w, __expf(fmaf(tp, b.z, ev.z) - m), acc.z);
            acc.w = fmaf(w, __expf(fmaf(tp, b.w, ev.w) - m), acc.w);
        }
    };

    unsigned nsteps = 0;
    // update_vsq!  src/CTM.jl:146-165 = src/fCTM.jl:180-198  (one scalar Newton iteration per topic = per lane)
    auto run_vsq = [&]() {
        if (on && !(p.debug & 1)) {
            for (int t = 0; t < p.niter; ++t) {
                double rho = 1.0;
                const double ex = exp(lam + 0.5 * vs - lz);
                const double grad = -0.5 * (isdiag + Cd * ex - 1.0 / vs);                  // :150
                const double ihd = -1.0 / (0.25 * Cd * ex + 0.5 / (vs * vs));             // :151
                const double pp = ihd * grad;
                while (vs - rho * pp <= 0.0) rho *= 0.5;                                   // :154
                vs -= rho * pp;
                if (rho * fabs(grad) < p.ntol) break;                                      // :159
            }
            vs += TMVB_EPS_D;                                                              // :164
        }
    };
    // update_lambda!  src/CTM.jl:129-142 = src/fCTM.jl:162-176
    auto run_lambda = [&](const double phic) {
        lam_old = lam;
        for (int t = 0; t < p.niter; ++t) {
            ++nsteps;
            const double ex = on ? exp(lam + 0.5 * vs - lz) : 0.0;
            // -H = invsigma + C_d Diag(e^{...})   :135 ; pad rows are unit rows
            float H[R];
#pragma unroll
            for (int q = 0; q < LPR; ++q) {
                float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane < KP) hv = *(const float4*)(isrow + 4 * q);
                H[4 * q] = hv.x; H[4 * q + 1] = hv.y; H[4 * q + 2] = hv.z; H[4 * q + 3] = hv.w;
            }
            // invsigma * (mu - lambda): row `lane` dot the broadcast vector, fp64.  (Carrying the product across the
            // Newton steps of a sweep -- invsigma delta = g - D delta -- saves 4 % but feeds every solve's residual,
            // ~K eps |A| |delta|, straight into the next gradient, at the size of ntol for the first step of a sweep;
            // measured and dropped.)
            const double dm = mu - lam;
            double mv = 0.0;
#pragma unroll
            for (int j = 0; j < R; ++j) mv = fma((double)H[j], readlane_d(dm, j), mv);
            const double gd = on ? (mv + phic - Cd * ex) : 0.0;                            // :134
            const double gn2 = wave_sum_d(gd * gd);
            const float dval = on ? (float)(Cd * ex) : 1.0f;
            const float diag = (lane < KP ? (float)isdiag_f : 0.0f) + dval;     // -H_ii = invsigma_ii + C_d e^{...}; pad rows: unit
            float delta;
            if (p.debug & 2) delta = (float)gd / (float)(isdiag + Cd * ex);
            else delta = gj_solve_rows<R>(H, diag, (float)gd, lane);
            if (on) lam += (double)delta;                                                  // :136
            if (sqrt(gn2) < p.ntol) break;                                                 // :138
        }
    };

    int sweeps = 0;
    double pd_last = 0.0;                 // sum_i (phi counts)_i (lambda_i - lambda_old_i) of the last executed sweep (CtmParams::pdot)
    for (int v = 0; v < p.viter; ++v) {
        ++sweeps;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        double phic;
        if constexpr (FILT) {
            if (lane < KP) e_l[lane] = on ? (float)lam : -INFINITY;
            WAVE_LDS_FENCE();
            for (int c0 = 0; c0 < N; c0 += tile_rows) {
                const int rows = min(tile_rows, N - c0);
                if (!(single && v > 0)) load_chunk(c0, rows);
                filt_a(c0, rows, !single);
                filt_b(rows, acc);
                if (!single) WAVE_LDS_FENCE();
            }
            acc = dpp_add4<0xB1>(acc);
            acc = dpp_add4<0x4E>(acc);
            const float gsel = (r4 == 0) ? acc.x : (r4 == 1) ? acc.y : (r4 == 2) ? acc.z : acc.w;
            phic = on ? (double)gsel : 0.0;                                     // (phi * counts)_i
        } else {
            // update_phi!  src/CTM.jl:175-178 (additive_logistic, src/utils.jl:114-122), linear space
            const float lmax = wave_max(on ? (float)lam : -INFINITY);
            e = on ? expf((float)lam - lmax) : 0.0f;
            if (lane < KP) e_l[lane] = e;
            WAVE_LDS_FENCE();
            for (int c0 = 0; c0 < N; c0 += tile_rows) {
                const int rows = min(tile_rows, N - c0);
                if (!(single && v > 0)) load_chunk(c0, rows);
                phase1(rows);
                phase2(rows, acc);
                if (!single) WAVE_LDS_FENCE();
            }
            acc = dpp_add4<0xB1>(acc);
            acc = dpp_add4<0x4E>(acc);
            const float gsel = (r4 == 0) ? acc.x : (r4 == 1) ? acc.y : (r4 == 2) ? acc.z : acc.w;
            phic = on ? (double)(e * gsel) : 0.0;                               // (phi * counts)_i
        }
        // update_logzeta!  src/CTM.jl:169-171 = src/fCTM.jl:202-204
        {
            const double x = on ? lam + 0.5 * vs : -INFINITY;
            const double m = wave_max_d(x);
            lz = m + log(wave_sum_d(on ? exp(x - m) : 0.0));
        }
#ifdef TMVB_MUTANT_FCTM_VSQ_FIRST
        run_vsq(); run_lambda(phic);                                            // MUTANT: fCTM's sweep in CTM's order (tests/test_mutants_gpu.py)
#else
        if constexpr (FILT) { run_lambda(phic); run_vsq(); }                    // src/fCTM.jl:239-240
        else { run_vsq(); run_lambda(phic); }                                   // src/CTM.jl:198-199
#endif
        const double df = on ? lam - lam_old : 0.0;
        const double dist2 = wave_sum_d(df * df);
        if (p.pdot) pd_last = wave_sum_d(phic * df);
        if (sqrt(dist2) < p.vtol) break;                                        // :200 / :242
    }
    if (p.pdot && lane == 0) p.pdot[d] = (float)pd_last;

    if (sweeps > 0) {
        if (on) {
            p.lambda[(int64_t)d * K + lane] = (float)lam;
            p.lambda_old[(int64_t)d * K + lane] = (float)lam_old;
            p.vsq[(int64_t)d * K + lane] = (float)vs;
        }
        if (lane == 0) p.logzeta[d] = (float)lz;
        if constexpr (FILT) {
            if (single)
                for (int n = lane; n < N; n += 64) {
                    p.tau_old[off + n] = tp_l[n]; p.tau[off + n] = tn_l[n];
                    p.lse[off + n] = m_l[n] + __logf(c_l[n] / w_l[n]);          // m + log s
                }
        } else {
            if (lane < KP) p.E[(int64_t)d * KP + lane] = e;   // e = exp(lambda_old - max): last-sweep phi factor
        }
    } else {
        if constexpr (FILT) { for (int n = lane; n < N; n += 64) p.lse[off + n] = INFINITY; }   // viter = 0: phi = 0 in the statistics
        else if (lane < KP) p.E[(int64_t)d * KP + lane] = 0.0f;                                 // viter = 0: no responsibilities
    }
    if (lane == 0) {
        p.sweeps[d] = (uint8_t)min(sweeps, 255);
        if (p.newton_steps) atomicAdd(p.newton_steps, (unsigned long long)nsteps);
    }
}

#include "tmvb_ctm_batch.h"
#include "tmvb_ctm_quad.h"

// 64-float padded copies of mu and of invsigma's diagonal for the batched kernel's block scalar loads
// (and the launch's counters / work queue zeroed: one launch instead of a memset beside it)
__global__ __launch_bounds__(64) void ctm_batch_tabs_kernel(int K, int KP, const float* __restrict__ invsigma_f, const float* __restrict__ mu_f,
                                                            float* __restrict__ sdiag, float* __restrict__ muf, unsigned long long* __restrict__ counters16)
{
    const int i = threadIdx.x;
    sdiag[i] = (i < KP) ? invsigma_f[i * KP + i] : 0.0f;
    muf[i] = (i < K) ? mu_f[i] : 0.0f;
    if (i < 16) counters16[i] = 0ull;
}

// Tables of the four-waves-per-item kernel (tmvb_ctm_quad.h): for wave w the H = KP / 4 rows [H w, H w + H) of invsigma as the flat sequence of
// pairs t = jp * H + i -> {S[2 jp][H w + i], S[2 jp + 1][H w + i]} (one v_pk_fma_f32 each), NB 16-float blocks per wave, zero filled; the block's
// diagonal and mu, 16 floats per wave (pads 0); and the launch's counters / work queue zeroed.
__global__ __launch_bounds__(256) void ctm_quad_tabs_kernel(int K, int KP, const float* __restrict__ invsigma_f, const float* __restrict__ mu_f,
                                                            float* __restrict__ Sq, float* __restrict__ sdq, float* __restrict__ muq,
                                                            unsigned long long* __restrict__ counters16)
{
    const int H = KP / 4, npair = (KP / 2) * H, stride = ((2 * npair + 15) / 16) * 16;
    for (int idx = threadIdx.x; idx < 4 * stride; idx += 256) {
        const int w = idx / stride, f = idx - w * stride, t = f >> 1, e = f & 1;
        float v = 0.0f;
        if (t < npair) { const int jp = t / H, i = t - jp * H; v = invsigma_f[(2 * jp + e) * KP + H * w + i]; }
        Sq[idx] = v;
    }
    if (threadIdx.x < 64) {
        const int w = threadIdx.x >> 4, i = threadIdx.x & 15, g = H * w + i;
        sdq[threadIdx.x] = (i < H) ? invsigma_f[g * KP + g] : 0.0f;
        muq[threadIdx.x] = (i < H && g < K) ? mu_f[g] : 0.0f;
    }
    if (threadIdx.x < 16) counters16[threadIdx.x] = 0ull;
}

// tmvb_ctm_update_sigma with a staged result: the staging buffers over sigma / invsigma (fp64), invsigma (fp32, padded), its log-determinant
// and the status flag in one launch (five device-to-device copies were 30 us of launches)
__global__ __launch_bounds__(256) void ctm_commit_sigma_kernel(int KK, int KPKP, const double* __restrict__ sigma_s, const double* __restrict__ invsigma_s,
                                                               const float* __restrict__ invsigma_f_s, const double* __restrict__ logdet_s,
                                                               const int* __restrict__ status_s, double* __restrict__ sigma, double* __restrict__ invsigma,
                                                               float* __restrict__ invsigma_f, double* __restrict__ logdet, int* __restrict__ status)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < KK) { sigma[q] = sigma_s[q]; invsigma[q] = invsigma_s[q]; }
    if (q < KPKP) invsigma_f[q] = invsigma_f_s[q];
    if (q == 0) { *logdet = *logdet_s; *status = *status_s; }
}

// Regroup the documents of the lane-per-document kernel.  A wave runs every loop until its slowest lane is done: on SYN-NSF
// a wave makes 29 Newton trips per E-step while its documents need 21 on average (iterations 30 - 60).  Documents are
// independent, so any grouping gives bit-identical per-document results (a finished lane only idles); grouping documents that
// needed similar numbers of Newton steps LAST iteration cuts the idling.  The static order is by length (the token phase runs
// to the longest document of the wave); inside chunks of CTM_REORDER_CHUNK consecutive documents of that order -- similar
// lengths -- the documents are sorted by last iteration's step count, descending, stable (bitonic sort of key << 16 | rank
// in LDS, one workgroup per chunk).
#define CTM_REORDER_CHUNK 2048
#define CTM_BATCH_MAX_LEN 2048     // unique terms per document the lane-per-document kernel takes
__global__ __launch_bounds__(1024) void ctm_reorder_kernel(const int32_t* __restrict__ order0, const uint16_t* __restrict__ key,
                                                           int32_t* __restrict__ order, int64_t M)
{
    __shared__ unsigned v[CTM_REORDER_CHUNK];
    const int64_t base = (int64_t)blockIdx.x * CTM_REORDER_CHUNK;
    for (int i = threadIdx.x; i < CTM_REORDER_CHUNK; i += 1024) {
        const int64_t q = base + i;
        // composite: larger key first; equal keys keep the static order (smaller i first -> larger CHUNK - 1 - i first)
        v[i] = (q < M) ? (((unsigned)key[order0[q]] << 16) | (unsigned)(CTM_REORDER_CHUNK - 1 - i)) : 0u;
    }
    __syncthreads();
    for (int k = 2; k <= CTM_REORDER_CHUNK; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < CTM_REORDER_CHUNK; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned a = v[i], b = v[l];
                    const bool desc = (i & k) == 0;            // descending overall
                    if (desc ? (a < b) : (a > b)) { v[i] = b; v[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < CTM_REORDER_CHUNK; i += 1024) {
        const int64_t q = base + i;
        if (q < M) order[q] = order0[base + (CTM_REORDER_CHUNK - 1 - (int)(v[i] & 0xffffu))];
    }
}

// Queue order of the lane-per-document launch: whole waves-of-documents (64 consecutive entries of the regrouped order) sorted by
// predicted time, longest first -- item time = 8.4 us x longest document + 37 us x Newton trips + const on SYN-NSF
// (profiles/r3_ctm_wave_log.txt), the trips taken from last E-step's per-document step counts.  List scheduling wants the longest
// items first; the length order alone leaves the Newton counts out (measured item times in queue order: 4456 us, longest first: 4382).
// Per-document results do not depend on the grouping, and waves move whole.  Up to CTM_WAVESORT_MAX waves.
#define CTM_WAVESORT_MAX 8192
// key of one wave-of-documents (one wave of the launch per wave-of-documents, lane = document)
__global__ __launch_bounds__(64) void ctm_wave_key_kernel(const int32_t* __restrict__ order, const int64_t* __restrict__ doc_ptr,
                                                          const uint16_t* __restrict__ newton, unsigned* __restrict__ keys)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    const int d = order[(int64_t)w * 64 + lane];
    const int len = wave_max_i((int)(doc_ptr[d + 1] - doc_ptr[d])), nt = wave_max_i((int)newton[d]);
    const float t = 8.4f * (float)len + 37.0f * (float)nt;
    if (lane == 0) keys[w] = (min((unsigned)t, (1u << 19) - 1u) << 13) | (unsigned)(CTM_WAVESORT_MAX - 1 - w);     // ties keep the given order
}
// keys[0, n) sorted descending in place (bitonic, one workgroup; P = power of two >= n)
__global__ __launch_bounds__(1024) void ctm_wave_sort_kernel(unsigned* __restrict__ keys, int n, int P)
{
    __shared__ unsigned v[CTM_WAVESORT_MAX];
    for (int w = threadIdx.x; w < P; w += 1024) v[w] = w < n ? keys[w] : 0u;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned a = v[i], b = v[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { v[i] = b; v[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int w = threadIdx.x; w < n; w += 1024) keys[w] = v[w];
}
// out = the waves-of-documents of `order` in the sorted order; a trailing partial wave stays last
__global__ __launch_bounds__(256) void ctm_wave_permute_kernel(const int32_t* __restrict__ order, const unsigned* __restrict__ keys, int n, int64_t Mb,
                                                               int32_t* __restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= Mb) return;
    const int q = (int)(idx >> 6);
    const int src = q < n ? (CTM_WAVESORT_MAX - 1 - (int)(keys[q] & (CTM_WAVESORT_MAX - 1))) : q;
    out[idx] = order[(int64_t)src * 64 + (idx & 63)];
}

// ------------------------------------------------------------------------------ E-step kernel, any K <= 128
// The register Gauss-Jordan above needs one lane per matrix row and KP VGPRs per row (K <= 60).  Larger models keep
// the Newton matrix in LDS instead: lane l owns the topics / matrix rows l and l + 64 (NS = 2 slots), the K x K matrix
// -H = invsigma + C_d Diag(e^{...}) (src/CTM.jl:135) is rebuilt in LDS at every Newton step (row stride KP = 4 * odd:
// the lanes' ds_read_b128 of their own rows are conflict free, the pivot row is a uniform-address broadcast read) and
// eliminated in place, one wave per document.  The topic tile is streamed through a fixed 32-row LDS window (gathered
// again every sweep).  Same arithmetic, exit tests and fp64 gradients as ctm_estep_kernel; ~K^3 LDS traffic per solve
// makes it several times slower per flop (K = 64: 170 ms per E-step on SYN-NSF, K = 100: 1.48 s) -- round 1's K > 60 path, now CG = false.
//
// CG = true (round 3, the default for K > 60): the Newton matrix is never formed.  invsigma is the same for every document, so ONE
// copy sits in LDS for a workgroup of several waves (one document per wave at a time, each wave pulling documents from a queue for
// the lifetime of the launch: the copy is loaded once), and (invsigma + C_d Diag(e^{...})) x = g is solved by the Jacobi-
// preconditioned conjugate gradients of the lane-per-document kernel (tmvb_ctm_batch.h; same stopping rule) with lane = matrix row:
// a mat-vec is LPR ds_read_b128 of the lane's own row (stride KP = 4 * odd: conflict free) against LPR broadcast reads of the
// direction, ~5 mat-vecs per Newton step instead of K^3 / 3 eliminations through LDS.  The gradient's invsigma (mu - lambda) uses
// the same fp32 mat-vec (as in the lane-per-document kernel: the rounding is 10 - 100 times below ntol).
#define CTM_GENERIC_TILE_ROWS 32
// waves per workgroup of the CG form (= per CU): the latency of the token gathers and of the CG's reduction chains is hidden by other
// waves only, so as many as registers and LDS allow -- K <= 64 (one topic slot): 12 (168 VGPRs); K > 64: 12, or 8 (256 VGPRs) for KP > 108
static size_t ctm_generic_wave_floats(int KP, int tile_rows = CTM_GENERIC_TILE_ROWS) { return (size_t)tile_rows * KP + 2 * (size_t)KP + 6 * (size_t)tile_rows + 2 * (size_t)KP; }
static size_t ctm_generic_lds_bytes(int KP, int waves = 1, int tile_rows = CTM_GENERIC_TILE_ROWS)
{
    return ((size_t)KP * KP + (size_t)waves * ctm_generic_wave_floats(KP, tile_rows)) * sizeof(float);
}
// Shape of the CG form's workgroup: waves (at most two per SIMD) and the rows of each wave's tile window, from what the CU's LDS holds
// next to the one copy of invsigma.  max_waves (the instantiation's bound) while that leaves a window of 24 rows or more, else fewer
// (never below four);
// the window is capped at 128 rows.  TMVB_CTM_CG_WAVES / TMVB_CTM_CG_TILE override.
static void ctm_generic_cg_shape(int KP, int max_waves, int* waves, int* tile_rows, bool global_a = false)
{
    const size_t room = (160 * 1024 - (global_a ? 0 : (size_t)KP * KP * sizeof(float))) / sizeof(float);
    auto rows_for = [&](int w) { const int64_t f = (int64_t)(room / (size_t)w) - 4 * KP; return (int)std::max<int64_t>(0, f / (KP + 6)); };
    int w = max_waves;
    while (w > 4 && rows_for(w) < 24) --w;
    if (const char* e = getenv("TMVB_CTM_CG_WAVES")) w = std::min(max_waves, std::max(1, atoi(e)));
    int t = std::min(128, rows_for(w)) & ~3;
    if (const char* e = getenv("TMVB_CTM_CG_TILE")) t = std::min(t, std::max(4, atoi(e) & ~3));
    *waves = w; *tile_rows = std::max(4, t);
}

// GA = true (round 4, K > 128: four topic slots per lane): a 256 x 256 invsigma does not fit the LDS, so it STAYS IN GLOBAL MEMORY (L2-resident,
// 260 KB at K = 256) and the mat-vec reads it by columns -- invsigma is symmetric, y_i = sum_c S[c][i] v_c, and row c is one coalesced read
// of the lanes (the lane's own row would be 64 cache lines per load instruction).  The whole LDS goes to the waves' tile windows.
template <int NS, bool FILT, bool CG = false, int MAXW = 1, bool GA = false>
__global__ __launch_bounds__(64 * MAXW) void ctm_estep_generic_kernel(CtmParams p, int64_t first)
{
    static_assert(!GA || CG, "the global-memory invsigma exists for the conjugate-gradient form only");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int K = p.K, KP = p.KP, LPR = p.LPR;
    const int tile_rows = CG ? p.tile_rows : CTM_GENERIC_TILE_ROWS;   // CG: as many rows as the CU's LDS gives each wave (a document that fits stays resident over its sweeps)
    // CG: waves of a workgroup run different documents with different trip counts -- a wave-level fence (LDS operations of one wave
    // complete in program order) where the one-wave form uses the workgroup barrier
#define GEN_FENCE() do { if constexpr (CG) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } else { __syncthreads(); } } while (0)
    float* A = lds;                                     // [KP][KP] Newton matrix; CG: invsigma, read-only, shared by the workgroup's waves
    const size_t wave_floats = (size_t)tile_rows * KP + 2 * (size_t)KP + 6 * tile_rows + 2 * (size_t)KP;
    float* Bt = A + (GA ? 0 : (size_t)KP * KP) + (CG ? (size_t)(threadIdx.x >> 6) * wave_floats : 0);   // [tile_rows][KP] topic tile window
    float* e_l = Bt + (size_t)tile_rows * KP;           // [KP]
    float* g_l = e_l + KP;                              // [KP] right-hand side of the solve
    float* w_l = g_l + KP;                              // [tile_rows]
    float* c_l = w_l + tile_rows;
    int* t_l = (int*)(c_l + tile_rows);
    float* tp_l = (float*)(t_l + tile_rows);            // FILT: tau entering the sweep / after update_tau! / column maximum
    float* tn_l = tp_l + tile_rows;
    float* m_l = tn_l + tile_rows;
    double* dm_l = (double*)(m_l + tile_rows);          // [KP] mu - lambda (8-byte aligned: KP is a multiple of 4, so the float blocks above total a multiple of 8 floats)

    if constexpr (CG && !GA) {
        for (int q = threadIdx.x; q < KP * KP / 4; q += blockDim.x) ((float4*)A)[q] = ((const float4*)p.invsigma)[q];
        __syncthreads();
    }
    float* p_l = g_l;                                   // CG: the vector of the mat-vec in flight (the direction, or mu - lambda)
    unsigned ncg_w = 0, nnewt_w = 0, ndocs_w = 0;
    auto run_doc = [&](const int d) {
    const int64_t off = p.doc_ptr[d];
    const int N = (int)(p.doc_ptr[d + 1] - off);
    const bool resident = CG && N <= tile_rows;         // the document's rows are gathered once, in its first sweep

    float cl = 0.0f;
    for (int n = lane; n < N; n += 64) cl += (float)p.counts[off + n];
    const double Cd = (double)wave_sum(cl);

    bool on[NS], row[NS];
    double isdiag[NS], mu[NS], lam[NS], vs[NS], lam_old[NS], phic[NS];
    float e[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = lane + 64 * s;
        on[s] = i < K; row[s] = on[s];          // the K x K system only: pad columns of invsigma are zero, pad rows are never touched
        isdiag[s] = on[s] ? (double)p.invsigma[(size_t)i * KP + i] : 1.0;
        mu[s] = on[s] ? (double)p.mu[i] : 0.0;
        lam[s] = on[s] ? (double)p.lambda[(int64_t)d * K + i] : 0.0;
        vs[s] = on[s] ? (double)p.vsq[(int64_t)d * K + i] : 1.0;
        lam_old[s] = lam[s]; phic[s] = 0.0; e[s] = 0.0f;
    }
    double lz = (double)p.logzeta[d];

    auto load_chunk = [&](int c0, int rows) {
        for (int n = lane; n < rows; n += 64) {
            t_l[n] = p.terms[off + c0 + n];
            c_l[n] = (float)p.counts[off + c0 + n];
            if constexpr (FILT) tn_l[n] = p.tau[off + c0 + n];
        }
        GEN_FENCE();
        const float* table = FILT ? p.L : p.beta;
        const int nch = rows * LPR;
        for (int f0 = 0; f0 < nch; f0 += 64) {
            const int f = f0 + lane;
            if (f < nch) {
                const int n = (LPR == 1) ? f : (int)__umulhi((unsigned)f, p.lpr_magic);
                const int c = f - n * LPR;
                const float* src = table + ((int64_t)t_l[n] * KP + 4 * c);
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(Bt + (size_t)f0 * 4), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GEN_FENCE();
    };
    auto phase1 = [&](int rows) {
        const float4* er = (const float4*)e_l;
        for (int n = lane; n < rows; n += 64) {
            const float4* br = (const float4*)(Bt + (size_t)n * KP);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int q = 0; q < LPR; ++q) {
                float4 b = br[q], ev = er[q];
                s0 = fmaf(b.x, ev.x, s0); s1 = fmaf(b.y, ev.y, s1);
                s2 = fmaf(b.z, ev.z, s2); s3 = fmaf(b.w, ev.w, s3);
            }
            w_l[n] = c_l[n] / ((s0 + s1) + (s2 + s3));
        }
        GEN_FENCE();
    };
    const int r4 = lane & 3, ql = lane >> 2;
    auto phase2 = [&](int rows, float4 (&acc)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int q = ql + 16 * s;
            if (q >= LPR) continue;
            for (int n = r4; n < rows; n += 4) {
                const float w = w_l[n];
                const float4 b = *(const float4*)(Bt + (size_t)n * KP + 4 * q);
                acc[s].x = fmaf(w, b.x, acc[s].x); acc[s].y = fmaf(w, b.y, acc[s].y);
                acc[s].z = fmaf(w, b.z, acc[s].z); acc[s].w = fmaf(w, b.w, acc[s].w);
            }
        }
    };

    // filtered CTM (src/fCTM.jl:216-219, :208-213): see ctm_estep_kernel
    auto filt_a = [&](int c0, int rows) {
        const float4* er = (const float4*)e_l;
        for (int n = lane; n < rows; n += 64) {
            const float4* br = (const float4*)(Bt + (size_t)n * KP);
            const float tp = tn_l[n];
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
                a = fmaf(p0, b.x, a); a = fmaf(p1, b.y, a); a = fmaf(p2, b.z, a); a = fmaf(p3, b.w, a);
            }
            const float prod = __expf(fminf(-(a / s), 87.0f));
            const float tnew = p.eta / (TMVB_EPS_F + (p.eta + (1.0f - p.eta) * (p.kappa[t_l[n]] * prod)));
            tp_l[n] = tp; m_l[n] = m; w_l[n] = c_l[n] / s;
            tn_l[n] = tnew;                                                                                 // (a resident tile is not loaded again)
            p.tau_old[off + c0 + n] = tp; p.tau[off + c0 + n] = tnew; p.lse[off + c0 + n] = m + __logf(s);   // streamed window: always stored
            if (p.aold) p.aold[off + c0 + n] = a / s;
        }
        GEN_FENCE();
    };
    auto filt_b = [&](int rows, float4 (&acc)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int q = ql + 16 * s;
            if (q >= LPR) continue;
            const float4 ev = *(const float4*)(e_l + 4 * q);
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
    unsigned nsteps = 0;
    double pd_last = 0.0;                 // CtmParams::pdot
    for (int v = 0; v < p.viter; ++v) {
        ++sweeps;
        // update_phi!  src/CTM.jl:175-178, linear space (no epsilon)
        float4 acc[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (FILT) {
#pragma unroll
            for (int s = 0; s < NS; ++s) if (on[s]) e_l[lane + 64 * s] = (float)lam[s];
            for (int i = K + lane; i < KP; i += 64) e_l[i] = -INFINITY;                 // pads: exp(-inf) = 0
            GEN_FENCE();
            for (int c0 = 0; c0 < N; c0 += tile_rows) {
                const int rows = min(tile_rows, N - c0);
                if (!(resident && v > 0)) load_chunk(c0, rows);
                filt_a(c0, rows);
                filt_b(rows, acc);
                GEN_FENCE();
            }
        } else {
            float lml = -INFINITY;
#pragma unroll
            for (int s = 0; s < NS; ++s) if (on[s]) lml = fmaxf(lml, (float)lam[s]);
            const float lmax = wave_max(lml);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                e[s] = on[s] ? expf((float)lam[s] - lmax) : 0.0f;
                if (on[s]) e_l[lane + 64 * s] = e[s];
            }
            for (int i = K + lane; i < KP; i += 64) e_l[i] = 0.0f;                      // pads
            GEN_FENCE();
            for (int c0 = 0; c0 < N; c0 += tile_rows) {
                const int rows = min(tile_rows, N - c0);
                if (!(resident && v > 0)) load_chunk(c0, rows);
                phase1(rows);
                if (p.store_w)                       // last executed sweep wins (the exit sweep is not known in advance)
                    for (int n = lane; n < rows; n += 64) p.wtok[p.tok_inv[off + c0 + n]] = w_l[n];
                phase2(rows, acc);
                GEN_FENCE();
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float4 a = dpp_add4<0xB1>(acc[s]);
            a = dpp_add4<0x4E>(a);
            const float gsel = (r4 == 0) ? a.x : (r4 == 1) ? a.y : (r4 == 2) ? a.z : a.w;
            phic[s] = on[s] ? (double)((FILT ? 1.0f : e[s]) * gsel) : 0.0;             // (phi * counts)_i
        }
        // update_logzeta!  :169-171
        {
            double xm = -INFINITY;
#pragma unroll
            for (int s = 0; s < NS; ++s) if (on[s]) xm = fmax(xm, lam[s] + 0.5 * vs[s]);
            const double m = wave_max_d(xm);
            double sl = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) if (on[s]) sl += exp(lam[s] + 0.5 * vs[s] - m);
            lz = m + log(wave_sum_d(sl));
        }
        // update_vsq!  src/CTM.jl:146-165 = src/fCTM.jl:180-198 (before update_lambda! in CTM, after it in fCTM)
        auto run_vsq = [&]() {
        if constexpr (CG) {
            // the slots' Newton iterations side by side (two independent fp64 chains per lane), exp and reciprocals as in the
            // lane-per-document kernel (cb_exp_n, cb_rcp_n); a lane's slot stops where the one-slot-at-a-time loop below would
            bool act[NS];
            double vv[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) { act[s] = on[s]; vv[s] = vs[s]; }
            for (int t = 0; t < p.niter; ++t) {
                bool any_act = false;
#pragma unroll
                for (int s = 0; s < NS; ++s) any_act = any_act || act[s];
                if (!__any(any_act)) break;
                double ex[NS], rv[NS], den[NS], ihd[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) ex[s] = lam[s] + 0.5 * vv[s] - lz;
                cb_exp_n<NS>(ex);
                cb_rcp_n<NS>(vv, rv);
#pragma unroll
                for (int s = 0; s < NS; ++s) den[s] = 0.25 * Cd * ex[s] + 0.5 * rv[s] * rv[s];
                cb_rcp_n<NS>(den, ihd);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double grad = -0.5 * (isdiag[s] + Cd * ex[s] - rv[s]);
                    const double pp = -ihd[s] * grad;
                    double rho = 1.0;
                    if (act[s]) {
                        while (vv[s] - rho * pp <= 0.0) rho *= 0.5;
                        vv[s] -= rho * pp;
                        if (rho * fabs(grad) < p.ntol) act[s] = false;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) if (on[s]) vs[s] = vv[s] + TMVB_EPS_D;
        } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (!on[s]) continue;
            for (int t = 0; t < p.niter; ++t) {
                double rho = 1.0;
                const double ex = exp(lam[s] + 0.5 * vs[s] - lz);
                const double grad = -0.5 * (isdiag[s] + Cd * ex - 1.0 / vs[s]);
                const double ihd = -1.0 / (0.25 * Cd * ex + 0.5 / (vs[s] * vs[s]));
                const double pp = ihd * grad;
                while (vs[s] - rho * pp <= 0.0) rho *= 0.5;
                vs[s] -= rho * pp;
                if (rho * fabs(grad) < p.ntol) break;
            }
            vs[s] += TMVB_EPS_D;
        }
        }
        };
        if constexpr (!FILT) run_vsq();                                                     // src/CTM.jl:198
        // update_lambda!  :129-142
#pragma unroll
        for (int s = 0; s < NS; ++s) lam_old[s] = lam[s];
        for (int t = 0; t < p.niter; ++t) {
            ++nsteps;
            double ex[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                ex[s] = on[s] ? exp(lam[s] + 0.5 * vs[s] - lz) : 0.0;
                if constexpr (!CG) { if (on[s]) dm_l[lane + 64 * s] = mu[s] - lam[s]; }
            }
            double gn2;
            if constexpr (CG) {
                // y = invsigma v for the vector in p_l (pads zero), lane = row
                // (one broadcast read of the vector per four columns for both slots, packed FMAs)
                auto matvec = [&](float (&y)[NS]) {
                    if constexpr (GA) {
                        const float* __restrict__ S = p.invsigma;
                        int col[NS];
                        float a0[NS], a1[NS];
#pragma unroll
                        for (int s = 0; s < NS; ++s) { col[s] = row[s] ? lane + 64 * s : 0; a0[s] = 0.0f; a1[s] = 0.0f; }
#pragma unroll 2
                        for (int c = 0; c < KP; c += 4) {                    // rows c .. c + 3 (rows past K are zero)
                            const float4 b = *(const float4*)(p_l + c);
                            float v0[NS], v1[NS], v2[NS], v3[NS];
#pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                const float* sc = S + (size_t)c * KP + col[s];
                                v0[s] = sc[0]; v1[s] = sc[KP]; v2[s] = sc[2 * KP]; v3[s] = sc[3 * KP];
                            }
#pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                a0[s] = fmaf(v0[s], b.x, a0[s]); a1[s] = fmaf(v1[s], b.y, a1[s]);
                                a0[s] = fmaf(v2[s], b.z, a0[s]); a1[s] = fmaf(v3[s], b.w, a1[s]);
                            }
                        }
#pragma unroll
                        for (int s = 0; s < NS; ++s) y[s] = row[s] ? a0[s] + a1[s] : 0.0f;
                        return;
                    }
                    const float4* pr = (const float4*)p_l;
                    const float4* sr[NS];
                    cb_v2f a01[NS], a23[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        sr[s] = (const float4*)(A + (size_t)(row[s] ? lane + 64 * s : 0) * KP);
                        a01[s] = cb_v2f{0.f, 0.f}; a23[s] = cb_v2f{0.f, 0.f};
                    }
#pragma unroll 2
                    for (int q = 0; q < LPR; ++q) {
                        const float4 b = pr[q];
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            const float4 a = sr[s][q];
                            a01[s] = __builtin_elementwise_fma(cb_v2f{a.x, a.y}, cb_v2f{b.x, b.y}, a01[s]);
                            a23[s] = __builtin_elementwise_fma(cb_v2f{a.z, a.w}, cb_v2f{b.z, b.w}, a23[s]);
                        }
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s) { const cb_v2f t = a01[s] + a23[s]; y[s] = row[s] ? t.x + t.y : 0.0f; }
                };
                float g[NS], Dg[NS], dinv[NS], x[NS], r[NS], pv[NS], y[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) if (row[s]) p_l[lane + 64 * s] = (float)(mu[s] - lam[s]);
                for (int i = K + lane; i < KP; i += 64) p_l[i] = 0.0f;
                GEN_FENCE();
                matvec(y);                                                                  // invsigma (mu - lambda)
                GEN_FENCE();
                double gn2l = 0.0;
                float ggl = 0.0f, rzl = 0.0f;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double gd = on[s] ? ((double)y[s] + phic[s] - Cd * ex[s]) : 0.0;  // :134
                    gn2l += gd * gd;
                    g[s] = (float)gd;
                    Dg[s] = on[s] ? (float)(Cd * ex[s]) : 1.0f;                             // -H = invsigma + C_d Diag(.)  :135
                    dinv[s] = 1.0f / ((float)isdiag[s] + Dg[s]);
                    r[s] = g[s]; pv[s] = on[s] ? g[s] * dinv[s] : 0.0f; x[s] = 0.0f;
                    ggl = fmaf(g[s], g[s], ggl); rzl = fmaf(r[s], pv[s], rzl);
                }
                gn2 = wave_sum_d(gn2l);
                const float gg = wave_sum(ggl);
                float rz = wave_sum(rzl);
                const float thr = fmaxf(p.cg_tol2 * gg, p.cg_abs2);
                int trips = 0;
                if (gg > thr) {
                    while (trips < p.cg_maxit) {
                        ++trips;
#pragma unroll
                        for (int s = 0; s < NS; ++s) if (row[s]) p_l[lane + 64 * s] = pv[s];
                        GEN_FENCE();
                        matvec(y);
                        GEN_FENCE();
                        float pHpl = 0.0f;
#pragma unroll
                        for (int s = 0; s < NS; ++s) { y[s] = fmaf(Dg[s], pv[s], y[s]); pHpl = fmaf(pv[s], y[s], pHpl); }
                        const float pHp = wave_sum(pHpl);
                        const float alpha = (pHp > 0.0f) ? rz / pHp : 0.0f;
                        float rrl = 0.0f, rznl = 0.0f;
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            x[s] = fmaf(alpha, pv[s], x[s]);
                            r[s] = fmaf(-alpha, y[s], r[s]);
                            y[s] = r[s] * dinv[s];                                           // z
                            rrl = fmaf(r[s], r[s], rrl); rznl = fmaf(r[s], y[s], rznl);
                        }
                        const float rr = wave_sum(rrl), rzn = wave_sum(rznl);
                        if (rr <= thr) break;
                        const float beta = (rz > 0.0f) ? rzn / rz : 0.0f;
                        rz = rzn;
#pragma unroll
                        for (int s = 0; s < NS; ++s) pv[s] = on[s] ? fmaf(beta, pv[s], y[s]) : 0.0f;
                    }
                }
                ncg_w += (unsigned)trips; ++nnewt_w;
#pragma unroll
                for (int s = 0; s < NS; ++s) if (on[s]) lam[s] += (double)x[s];             // :136
            } else {
            for (int i = K + lane; i < KP; i += 64) dm_l[i] = 0.0;
            // A <- invsigma (pads zero), row by row
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (!row[s]) continue;
                const float4* src = (const float4*)(p.invsigma + (size_t)(lane + 64 * s) * KP);
                float4* dst = (float4*)(A + (size_t)(lane + 64 * s) * KP);
                for (int q = 0; q < LPR; ++q) dst[q] = src[q];
            }
            GEN_FENCE();
            double gn2l = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (!row[s]) continue;
                const int i = lane + 64 * s;
                const float* ar = A + (size_t)i * KP;
                double mv = 0.0;
                for (int c = 0; c < KP; ++c) mv = fma((double)ar[c], dm_l[c], mv);      // invsigma (mu - lambda), fp64
                const double gd = on[s] ? (mv + phic[s] - Cd * ex[s]) : 0.0;            // :134
                gn2l += gd * gd;
                g_l[i] = (float)gd;
                A[(size_t)i * KP + i] += (float)(Cd * ex[s]);                           // -H = invsigma + C_d Diag(.)  :135
            }
            gn2 = wave_sum_d(gn2l);
            GEN_FENCE();
            // Gauss-Jordan in LDS, no pivoting (SPD)
            for (int j = 0; j < K; ++j) {
                const float rp = fast_rcp(A[(size_t)j * KP + j]);
                const float gj = g_l[j];
                const int c0 = (j + 1) & ~3;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int i = lane + 64 * s;
                    if (!row[s] || i == j) continue;
                    float* ar = A + (size_t)i * KP;
                    const float nf = -ar[j] * rp;
                    for (int c = c0; c < KP; c += 4) {
                        const float4 pv = *(const float4*)(A + (size_t)j * KP + c);
                        float4 av = *(float4*)(ar + c);
                        av.x = fmaf(nf, pv.x, av.x); av.y = fmaf(nf, pv.y, av.y); av.z = fmaf(nf, pv.z, av.z); av.w = fmaf(nf, pv.w, av.w);
                        *(float4*)(ar + c) = av;
                    }
                    for (int c = j + 1; c < c0; ++c) ar[c] = fmaf(nf, A[(size_t)j * KP + c], ar[c]);   // head of the row up to the aligned part
                    g_l[i] = fmaf(nf, gj, g_l[i]);
                }
                GEN_FENCE();
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int i = lane + 64 * s;
                if (on[s]) lam[s] += (double)(g_l[i] * fast_rcp(A[(size_t)i * KP + i]));   // :136
            }
            GEN_FENCE();
            }
            if (sqrt(gn2) < p.ntol) break;                                                  // :138
        }
        if constexpr (FILT) run_vsq();                                                      // src/fCTM.jl:240
        double d2l = 0.0, pdl = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) if (on[s]) { const double df = lam[s] - lam_old[s]; d2l += df * df; pdl = fma(phic[s], df, pdl); }
        if (p.pdot) pd_last = wave_sum_d(pdl);
        if (sqrt(wave_sum_d(d2l)) < p.vtol) break;                                          // :200
    }
    if (p.pdot && lane == 0) p.pdot[d] = (float)pd_last;

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = lane + 64 * s;
        if (sweeps > 0 && on[s]) {
            p.lambda[(int64_t)d * K + i] = (float)lam[s];
            p.lambda_old[(int64_t)d * K + i] = (float)lam_old[s];
            p.vsq[(int64_t)d * K + i] = (float)vs[s];
        }
        if constexpr (!FILT) { if (on[s]) p.E[(int64_t)d * p.estride + i] = (sweeps > 0) ? e[s] : 0.0f; }
    }
    if constexpr (FILT) {
        if (sweeps == 0) for (int n = lane; n < N; n += 64) p.lse[off + n] = INFINITY;     // viter = 0: phi = 0 in the statistics
    } else {
        if (p.estride >= KP)                                                               // dense rows (KP / 4 > 64) have no pads
            for (int i = K + lane; i < KP; i += 64) p.E[(int64_t)d * p.estride + i] = 0.0f;   // pads
        if (sweeps == 0 && p.store_w)
            for (int n = lane; n < N; n += 64) p.wtok[p.tok_inv[off + n]] = 0.0f;         // viter = 0: no responsibilities
    }
    if (lane == 0) {
        if (sweeps > 0) p.logzeta[d] = (float)lz;
        p.sweeps[d] = (uint8_t)min(sweeps, 255);
        if (p.newton_steps) atomicAdd(p.newton_steps, (unsigned long long)nsteps);
    }
    ++ndocs_w;
    };   // run_doc
    if constexpr (CG) {
        for (;;) {
            unsigned v = 0;
            if (lane == 0) v = atomicAdd(p.queue, 1u);
            const int64_t q = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)v);
            if (q >= p.n_docs) break;
            run_doc(p.doc_order[first + q]);
            GEN_FENCE();
        }
        if (lane == 0 && p.cg_diag) {
            atomicAdd(p.cg_diag, (unsigned long long)ncg_w); atomicAdd(p.cg_diag + 1, (unsigned long long)nnewt_w); atomicAdd(p.cg_diag + 2, (unsigned long long)ndocs_w);
        }
    } else run_doc(p.doc_order[first + blockIdx.x]);
#undef GEN_FENCE
}

// ------------------------------------------------------------------------------ M-step: sigma
// Scatter matrix sum_d (lambda_d - mu)(lambda_d - mu)^T on f32 MFMA.  X = Lambda - mu is K x M; a
// wave owns a slab of documents and accumulates the 64 x 64 (padded) product in four
// v_mfma_f32_32x32x2_f32 accumulators; the contraction index of the MFMA is the document.
// A operand: lane l holds A[i = l & 31][k = l >> 5]; B operand: B[k = l >> 5][j = l & 31].
typedef float f32x16 __attribute__((ext_vector_type(16)));

// blockIdx.y = 32 x 32 tile (bi, bj) of the K x K product, NB = ceil(K / 32) tiles per side (any K <= 128)
__global__ __launch_bounds__(64) void ctm_scatter_mfma_kernel(const float* __restrict__ lambda, const float* __restrict__ mu,
                                                              int K, int NB, int64_t M, int64_t docs_per_wave,
                                                              float* __restrict__ partial /* [nwaves][NB*NB][32*32] */)
{
    const int lane = threadIdx.x;
    const int bi = blockIdx.y / NB, bj = blockIdx.y - bi * NB;
    const int64_t d0 = (int64_t)blockIdx.x * docs_per_wave;
    const int64_t d1 = min(M, d0 + docs_per_wave);
    const int i = lane & 31, kk = lane >> 5;
    const int ri = 32 * bi + i, rj = 32 * bj + i;
    const float mui = (ri < K) ? mu[ri] : 0.0f;
    const float muj = (rj < K) ? mu[rj] : 0.0f;
    f32x16 c = {0};
    // eight document pairs per round, their 16 loads in flight together (one pair per round left the wave waiting a memory latency
    // per MFMA: 179 us for the 128 804 x 50 lambda of SYN-NSF); rows past the slab / topics past K read a valid address and are zeroed
    const bool oi = ri < K, oj = rj < K;
    const int ci = oi ? ri : 0, cj = oj ? rj : 0;
    for (int64_t d = d0; d < d1; d += 16) {
        float a[8], bq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t dd = d + 2 * u + kk;
            const int64_t dc = dd < d1 ? dd : d0;
            a[u] = lambda[dc * K + ci]; bq[u] = lambda[dc * K + cj];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = d + 2 * u + kk < d1;
            const float av = (in && oi) ? a[u] - mui : 0.0f, bv = (in && oj) ? bq[u] - muj : 0.0f;
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
        }
    }
    // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* out = partial + ((size_t)blockIdx.x * NB * NB + blockIdx.y) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        out[row * 32 + col] = c[r];
    }
}

// scatter[i*K + j] = sum_w partial[w][tile(i,j)][(i % 32) * 32 + j % 32]  (fixed order)
// 16 elements x 16 interleaved wave groups per workgroup (one thread per element summed 500 partials in a row on ten CUs: 136 us)
__global__ __launch_bounds__(256) void ctm_scatter_reduce_kernel(const float* __restrict__ partial, int nwaves, int K, int NB,
                                                                 float* __restrict__ scatter)
{
    __shared__ double red[16][17];
    const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int q = blockIdx.x * 16 + e;
    double s = 0.0;
    if (q < K * K) {
        const int i = q / K, j = q - i * K;
        const size_t tile = (size_t)(i >> 5) * NB + (j >> 5), in = (size_t)(i & 31) * 32 + (j & 31);
        for (int w = g; w < nwaves; w += 16) s += (double)partial[((size_t)w * NB * NB + tile) * 1024 + in];
    }
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && q < K * K) {
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += red[u][e];
        scatter[q] = (float)t;
    }
}

// update_sigma! (src/CTM.jl:108-111) then update_mu! (:102-104), one workgroup, fp64:
// sigma = (diagm(sum vsq) + scatter) / M  (Symmetric() reads the upper triangle), invsigma = inv(sigma)
// by IN-PLACE Gauss-Jordan inversion in LDS (K x K doubles: 128 KiB at K = 128), logdet(invsigma) = -sum log pivots;
// mu = sum lambda / M.
__global__ __launch_bounds__(256) void ctm_sigma_mu_kernel(int K, int KP, double Md, const float* __restrict__ stats_tail,
                                                           double* __restrict__ sigma_d, double* __restrict__ invsigma_d,
                                                           float* __restrict__ invsigma_f, double* __restrict__ mu_d,
                                                           float* __restrict__ mu_f, double* __restrict__ logdet_inv,
                                                           int* __restrict__ status, int do_sigma, int do_mu, double* __restrict__ work)
{
    // [K][K]: LDS up to K = 128 (128 KiB); beyond (round 4, K <= 256) a global-memory workspace -- the one workgroup's __syncthreads
    // order its global reads and writes as they order the LDS ones; an inversion then runs at L2 speed (milliseconds at K = 256,
    // next to an E-step of hundreds)
    extern __shared__ double sm_lds[];
    double* __restrict__ sm = work ? work : sm_lds;
    const float* sum_lambda = stats_tail;
    const float* sum_vsq = stats_tail + K;
    const float* scatter = stats_tail + 2 * K;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (do_sigma) {
        for (int q = tid; q < K * K; q += nt) {
            const int i = q / K, j = q - i * K;
            const int a = min(i, j), b = max(i, j);            // upper triangle (a <= b)
            double v = (double)scatter[a * K + b];
            if (i == j) v += (double)sum_vsq[i];
#ifdef TMVB_MUTANT_CTM_SIGMA_NEW_MU
            // MUTANT (tests/test_mutants_gpu.py, never in a shipped build): the scatter matrix re-centred on the NEW mu = sum lambda / M, i.e. update_mu!
            // before update_sigma! -- the reference runs them the other way round (quirk Q2, src/CTM.jl:207-208)
            v -= Md * ((double)sum_lambda[i] / Md - mu_d[i]) * ((double)sum_lambda[j] / Md - mu_d[j]);
#endif
            v /= Md;
            sm[i * K + j] = v;
            sigma_d[j * K + i] = v;                            // column-major (symmetric anyway)
        }
        __syncthreads();
        __shared__ double piv_s;
        __shared__ double ld_s;
        __shared__ int bad_s;
        if (tid == 0) { ld_s = 0.0; bad_s = 0; }
        __syncthreads();
        for (int j = 0; j < K; ++j) {
            if (tid == 0) {
                const double pv = sm[j * K + j];
                piv_s = pv;
                if (!(pv > 0.0)) bad_s = 1;
                ld_s += log(pv);
                sm[j * K + j] = 1.0;                            // the pivot column becomes the inverse's column
            }
            __syncthreads();
            const double rp = 1.0 / piv_s;
            for (int c = tid; c < K; c += nt) sm[j * K + c] *= rp;          // normalise pivot row
            __syncthreads();
            // row_i -= f_i row_j with f_i = a_ij taken before it is cleared; one thread owns (i, c) for all c of its stride
            for (int q = tid; q < K * K; q += nt) {
                const int i = q / K, c = q - i * K;
                if (i != j && c != j) sm[i * K + c] -= sm[i * K + j] * sm[j * K + c];
            }
            __syncthreads();
            for (int i = tid; i < K; i += nt) if (i != j) sm[i * K + j] = -sm[i * K + j] * sm[j * K + j];
            __syncthreads();
        }
        for (int q = tid; q < KP * KP; q += nt) {
            const int i = q / KP, j = q - i * KP;
            float v = 0.0f;
            if (i < K && j < K) {
                const double s = 0.5 * (sm[i * K + j] + sm[j * K + i]);   // inv(::Symmetric) is Symmetric
                invsigma_d[j * K + i] = s;
                v = (float)s;
            }
            invsigma_f[q] = v;
        }
        if (tid == 0) { *logdet_inv = -ld_s; *status = bad_s; }
    }
    if (do_mu) {
        for (int i = tid; i < K; i += nt) {
            const double m = (double)sum_lambda[i] / Md;
            mu_d[i] = m;
            mu_f[i] = (float)m;
        }
    }
}

// ------------------------------------------------------------------------------ ELBO
// update_elbo!  src/CTM.jl:89-98 per document (terms :56-86).  One wave per document, lane l owns topics l + 64 s.
// TOK = false -- the decomposed form (round 5).  With x_ni = beta_old[i, v_n] e_i, e_i = exp(lambda_old_i - max lambda_old), s_n = sum_i x_ni and
// phi_ni = x_ni / s_n (:93), the token terms  sum_n c_n sum_i phi_ni (lambda_i + log(beta_iv + eps) - log phi_ni)  (:64, :71, :84) are
//     sum_i (phi counts)_i (lambda_i - lambda_old_i)                    pdot[d]: left by the E-step kernel's exit test (CtmParams::pdot)
//   + C_d max lambda_old + sum_n c_n log s_n                            the statistics pass recomputes s_n from the same e: one partial per chunk (logz)
//   + sum_{v,i} S_vi (log(beta_new + eps) - log beta_old)               beta_norm_kernel's partial (update_beta! reads S and writes beta_new)
// so this kernel skips its token loop (95 % of it: 1.51 -> see profiles/r5_elbo_forms.txt) and only sums the document's counts.
template <int NS, bool TOK = true>
__global__ __launch_bounds__(64) void ctm_elbo_kernel(int K, int KP, const int64_t* __restrict__ doc_ptr,
                                                      const int32_t* __restrict__ terms, const int32_t* __restrict__ counts,
                                                      const double* __restrict__ mu_d, const double* __restrict__ invsigma_d,
                                                      const double* __restrict__ logdet_inv,
                                                      const float* __restrict__ beta, const float* __restrict__ beta_old,
                                                      const float* __restrict__ lambda, const float* __restrict__ lambda_old,
                                                      const float* __restrict__ vsq, const float* __restrict__ logzeta,
                                                      double* __restrict__ doc_val, const float* __restrict__ pdot = nullptr)
{
    const int lane = threadIdx.x;
    const int d = blockIdx.x;
    const int64_t off = doc_ptr[d];
    const int N = (int)(doc_ptr[d + 1] - off);
    bool on[NS]; int ix[NS];
    float eo[NS];
    double l[NS], v[NS], df[NS];
    float lml = -INFINITY;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        on[s] = lane + 64 * s < K;
        ix[s] = on[s] ? lane + 64 * s : 0;
        eo[s] = on[s] ? lambda_old[(int64_t)d * K + ix[s]] : -INFINITY;
        lml = fmaxf(lml, eo[s]);
    }
    const float lmax = wave_max(lml);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        eo[s] = on[s] ? expf(eo[s] - lmax) : 0.0f;
        l[s] = on[s] ? (double)lambda[(int64_t)d * K + ix[s]] : 0.0;
        v[s] = on[s] ? (double)vsq[(int64_t)d * K + ix[s]] : 1.0;
        df[s] = on[s] ? l[s] - mu_d[ix[s]] : 0.0;
    }
    const double lz = (double)logzeta[d];
    double acc = 0.0, Cd = 0.0;
    if constexpr (!TOK) {
        float cl = 0.0f;                                                       // C_d (exact in fp32 up to 2^24, as the E-step kernels sum it)
        for (int n = lane; n < N; n += 64) cl += (float)counts[off + n];
        Cd = (double)wave_sum(cl);
        if (lane == 0 && N > 0) acc = (double)pdot[d] + Cd * (double)lmax;
    } else
    for (int n = 0; n < N; ++n) {
        const int t = terms[off + n];
        const float c = (float)counts[off + n];
        Cd += (double)c;
        float x[NS], xl = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s) { x[s] = on[s] ? beta_old[(int64_t)t * KP + ix[s]] * eo[s] : 0.0f; xl += x[s]; }   // :93
        const float inv = 1.0f / wave_sum(xl);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (!on[s]) continue;
            const float ph = x[s] * inv;
            const double cp = (double)(c * ph);
            acc += cp * l[s];                                                             // Elogpz :64 (first part)
            acc += cp * (double)logf(beta[(int64_t)t * KP + ix[s]] + TMVB_EPS_F);         // Elogpw :71
            if (ph > 0.0f) acc -= cp * (double)logf(ph);                                  // -Elogqz :84
        }
    }
    // Elogpeta :57
    double mv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) mv[s] = 0.0;
    for (int j = 0; j < K; ++j) {
        double dj = 0.0;                                                      // df of topic j: slot j / 64 (uniform), lane j % 64
#pragma unroll
        for (int s = 0; s < NS; ++s) if ((j >> 6) == s) dj = readlane_d(df[s], j & 63);
#pragma unroll
        for (int s = 0; s < NS; ++s) mv[s] = fma(on[s] ? invsigma_d[(int64_t)j * K + ix[s]] : 0.0, dj, mv[s]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (!on[s]) continue;
        acc += -0.5 * (invsigma_d[(int64_t)ix[s] * K + ix[s]] * v[s] + df[s] * mv[s]);
        acc -= Cd * exp(l[s] + 0.5 * v[s] - lz);                                          // Elogpz :64 (second part)
        acc += 0.5 * log(v[s]);                                                           // -Elogqeta :77
    }
    double tot = wave_sum_d(acc);
    const double TWO_PI_LOG = 1.8378770664093453;   // log(2 pi)
    tot += 0.5 * (*logdet_inv - (double)K * TWO_PI_LOG);
    tot -= Cd * (lz - 1.0);
    tot += 0.5 * (double)K * (1.0 + TWO_PI_LOG);
    if (lane == 0) doc_val[d] = tot;
}

// the decomposed update_elbo!: sum of the per-document values + ln 2 * the chunks' log2-normaliser sums + this shard's share of update_beta!'s partial
__global__ __launch_bounds__(1024) void ctm_elbo_final_kernel(const double* __restrict__ doc_val, int64_t M, const double* __restrict__ logz, int64_t n_logz,
                                                              const double* __restrict__ pw_partial, int pw_blocks, double pw_share, double* __restrict__ out)
{
    __shared__ double red[1024];
    double s = 0.0, lz = 0.0, pw = 0.0;
    for (int64_t d = threadIdx.x; d < M; d += 1024) s += doc_val[d];
    for (int64_t i = threadIdx.x; i < n_logz; i += 1024) lz += logz[i];
    for (int b = threadIdx.x; b < pw_blocks; b += 1024) pw += pw_partial[b];
    red[threadIdx.x] = s + 0.6931471805599453 * lz + pw_share * pw;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

__global__ __launch_bounds__(1024) void sum_docs_kernel(const double* __restrict__ doc_val, int64_t M, double* __restrict__ out)
{
    __shared__ double red[1024];
    double s = 0.0;
    for (int64_t d = threadIdx.x; d < M; d += 1024) s += doc_val[d];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// ------------------------------------------------------------------------------ host side
struct tmvb_ctm {
    tmvb_ctx* ctx = nullptr;
    tmvb_corpus* corp = nullptr;
    int K = 0, KP = 0;
    int64_t M = 0, V = 0, M_total = 0;
    bool distributed = false;
    tmvb_comm* comm = nullptr;         // document-sharded train!: the all-reduce of the packed statistics (not owned)
    int nslot = 1;                     // topic slots per lane of the lane = topic kernels ((K + 63) / 64)
    int NB = 2;                        // 32 x 32 tiles per side of the MFMA scatter product
    bool generic = false;              // K > 60: ctm_estep_generic_kernel
    bool generic_cg = false;           // ... in its conjugate-gradient form (d_cg_iters then carries its trip counts)
    float* d_beta[2] = {nullptr, nullptr};
    float* d_beta_pad = nullptr;       // TMVB_CTM_ROWPAD: [V][64] copy of beta (KP = 52) for the lane-per-document kernel's row gather
    int cur = 0;
    float* d_stats = nullptr;          // S (K*V) | sum_lambda (K) | sum_vsq (K) | scatter (K*K)
    bool own_stats = true;
    float* d_lambda = nullptr; float* d_lambda_old = nullptr; float* d_vsq = nullptr; float* d_logzeta = nullptr;
    float* d_wtok = nullptr; float* d_E = nullptr; float* d_ts_partial = nullptr;
    double* d_sigma_work = nullptr; double* d_sigma_work_s = nullptr;   // K > 128: [K][K] workspaces of the sigma inversion (and of the staged one)
    float* d_invsigma_f = nullptr; float* d_mu_f = nullptr;
    bool batch = false;                // lane-per-document kernel (tmvb_ctm_batch.h), KP <= 52
    int64_t n_long = 0;                // ... except the first n_long documents of the processing order (> CTM_BATCH_MAX_LEN unique terms each):
                                       // a lane walks its document's tokens one after the other, so those keep the wave-per-document kernel
    bool reorder = false, keys_valid = false;      // regroup its documents by last E-step's Newton step counts (ctm_reorder_kernel)
    uint16_t* d_doc_newton = nullptr; int32_t* d_doc_order0 = nullptr;
    int32_t* d_doc_order_q = nullptr; unsigned* d_wave_keys = nullptr; bool queue_sorted = false; hipEvent_t ev_spec2 = nullptr; bool spec2_pending = false;      // the lane-per-document launch's queue order (ctm_wave_sort_kernel), valid for the next E-step
    float* d_bt_sdiag = nullptr; float* d_bt_muf = nullptr; unsigned long long* d_cg_iters = nullptr;
    bool filt_parts = false, filt_pw_valid = false;   // fCTM's collecting E-step ran / update_beta! left sum S (log(beta_new + eps) - log(beta_old + eps)) behind it
    float* d_q_S = nullptr; float* d_q_sd = nullptr; float* d_q_mu = nullptr; int beta_pad_cpr = 0; bool quad_attr_set = false; int lds_limit = -1;   // tables of the four-waves-per-item kernel (tmvb_ctm_quad.h)
    float cg_tol = 1e-4f, cg_abs = 0.05f;   // CG exit: relative residual, and the fraction of ntol it may stop at (TMVB_CTM_CG_TOL / _ABS)
    double* d_sigma = nullptr; double* d_invsigma = nullptr; double* d_mu = nullptr; double* d_logdet = nullptr;
    float* d_scatter_partial = nullptr; int n_scatter_waves = 0; int64_t docs_per_wave = 0;
    uint8_t* d_sweeps = nullptr; int32_t* d_doc_order = nullptr;
    double* d_partial = nullptr; double* d_partial_docs = nullptr; double* d_partial_docs2 = nullptr; double* d_rowsum = nullptr; double* d_doc_val = nullptr; double* d_elbo = nullptr;
    unsigned long long* d_newton = nullptr; int* d_status = nullptr;
    double elbo = 0.0;
    std::vector<tmvb_bucket> buckets;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool tail_fresh = false;           // the statistics tail (sum lambda | sum vsq | scatter) was computed by the last tmvb_ctm_estep
    // ... and, one process, behind it on the same side stream: update_sigma!'s result into staging buffers (tmvb_ctm_update_sigma then
    // only copies them over -- the 115 us fp64 inversion leaves the iteration's critical path) and the regrouping of the documents for
    // the NEXT E-step (ctm_reorder_kernel; it used to run in front of that E-step).  ev_spec marks their end.
    bool sigma_staged = false, reorder_staged = false, spec_pending = false;
    double* d_sigma_s = nullptr; double* d_invsigma_s = nullptr; float* d_invsigma_f_s = nullptr; double* d_logdet_s = nullptr; int* d_status_s = nullptr;
    hipEvent_t ev_spec = nullptr;
    static constexpr int NAUX = 4;
    hipStream_t aux[NAUX] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[NAUX] = {nullptr, nullptr, nullptr, nullptr};
    // decomposed update_elbo! (as LDA's, tmvb_lda.hip): an iteration that train! will check (or every one, TMVB_CTM_ELBO_PARTS=2 at tmvb_ctm_create;
    // 0: never) has its statistics pass leave sum c log2 s per chunk (d_logz), its document kernels sum_i (phi counts)_i (lambda_i - lambda_old_i) per
    // document (d_pdot), and update_beta! sum S (log(beta_new + eps) - log beta_old) (d_pw_partial); update_elbo! then skips its token loop.
    int parts_env = 1; bool want_parts = false;
    double* d_logz = nullptr; size_t logz_cap = 0; int64_t n_logz = 0; float* d_pdot = nullptr; double* d_pw_partial = nullptr; int pw_blocks = 0;
    bool logz_valid = false, stats_fresh = false, pw_valid = false;
    int elbo_form = 0; bool force_walk = false;
    int64_t stats_len() const { return (int64_t)K * V + 2 * K + (int64_t)K * K; }
    float* tail() const { return d_stats + (size_t)K * V; }
};

// KP = 4 * odd <= 52 (K <= 52): the lane-per-document kernel and, for its long documents / TMVB_CTM_BATCH=0, the register Gauss-Jordan
// kernel.  Beyond (round 3): ctm_estep_generic_kernel in its conjugate-gradient form -- at K = 53 ... 60 it measured 11.9 ms per E-step
// on SYN-NSF against 16.5 ms for a KP = 60 instantiation of the lane-per-document kernel (three waves per CU by LDS, 330 scratch
// reloads) and 40.7 ms for the register Gauss-Jordan kernel that round 2 ran there.
static bool ctm_kp_supported(int kp) { return kp >= 4 && kp <= 52 && kp % 8 == 4; }

extern "C" int tmvb_ctm_destroy(tmvb_ctm* h)
{
    if (!h) return TMVB_OK;
    if (h->ctx) (void)hipSetDevice(h->ctx->device);
    (void)hipFree(h->d_beta[0]); (void)hipFree(h->d_beta[1]); (void)hipFree(h->d_beta_pad);
    if (h->own_stats) (void)hipFree(h->d_stats);
    (void)hipFree(h->d_lambda); (void)hipFree(h->d_lambda_old); (void)hipFree(h->d_vsq); (void)hipFree(h->d_logzeta);
    (void)hipFree(h->d_sigma_work); (void)hipFree(h->d_sigma_work_s);
    (void)hipFree(h->d_wtok); (void)hipFree(h->d_E); (void)hipFree(h->d_ts_partial); (void)hipFree(h->d_invsigma_f);
    (void)hipFree(h->d_q_S); (void)hipFree(h->d_q_sd); (void)hipFree(h->d_q_mu);
    (void)hipFree(h->d_bt_sdiag); (void)hipFree(h->d_bt_muf); (void)hipFree(h->d_cg_iters); (void)hipFree(h->d_doc_newton); (void)hipFree(h->d_doc_order0); (void)hipFree(h->d_doc_order_q); (void)hipFree(h->d_wave_keys);
    (void)hipFree(h->d_mu_f); (void)hipFree(h->d_sigma); (void)hipFree(h->d_invsigma); (void)hipFree(h->d_mu);
    (void)hipFree(h->d_logdet); (void)hipFree(h->d_scatter_partial); (void)hipFree(h->d_sweeps); (void)hipFree(h->d_doc_order);
    (void)hipFree(h->d_partial); (void)hipFree(h->d_partial_docs); (void)hipFree(h->d_partial_docs2); (void)hipFree(h->d_rowsum); (void)hipFree(h->d_doc_val); (void)hipFree(h->d_elbo);
    (void)hipFree(h->d_newton); (void)hipFree(h->d_status); (void)hipFree(h->d_logz); (void)hipFree(h->d_pdot); (void)hipFree(h->d_pw_partial);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_spec) (void)hipEventDestroy(h->ev_spec);
    if (h->ev_spec2) (void)hipEventDestroy(h->ev_spec2);
    (void)hipFree(h->d_sigma_s); (void)hipFree(h->d_invsigma_s); (void)hipFree(h->d_invsigma_f_s); (void)hipFree(h->d_logdet_s); (void)hipFree(h->d_status_s);
    for (int a = 0; a < tmvb_ctm::NAUX; ++a) {
        if (h->ev_join[a]) (void)hipEventDestroy(h->ev_join[a]);
        tmvb_release_stream(h->aux[a]); h->aux[a] = nullptr;        // pooled streams stay (tmvb_pool_stream)
    }
    delete h;
    return TMVB_OK;
}

static int ctm_upload_beta(tmvb_ctm* h, float* dst, const double* src)
{
    const size_t K = h->K, KP = h->KP, V = h->V;
    std::vector<float> tmp(V * KP + 4, 0.0f);
    for (size_t j = 0; j < V; ++j)
        for (size_t i = 0; i < K; ++i) tmp[j * KP + i] = (float)src[j * K + i];
    TMVB_HIP(hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    return TMVB_OK;
}

static int ctm_download_beta(tmvb_ctm* h, double* dst, const float* src)
{
    const size_t K = h->K, KP = h->KP, V = h->V;
    std::vector<float> tmp(V * KP);
    TMVB_HIP(hipMemcpyAsync(tmp.data(), src, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    for (size_t j = 0; j < V; ++j)
        for (size_t i = 0; i < K; ++i) dst[j * K + i] = (double)tmp[j * KP + i];
    return TMVB_OK;
}

extern "C" int tmvb_ctm_set_state(tmvb_ctm* h, const double* mu, const double* sigma, const double* invsigma,
                                  const double* beta, const double* beta_old, const double* lambda,
                                  const double* lambda_old, const double* vsq, const double* logzeta, const double* elbo);

static int ctm_reduce_docs_on(tmvb_ctm* h, hipStream_t st);
static int ctm_join_spec(tmvb_ctm* h);
static int ctm_sigma_mu(tmvb_ctm* h, int do_sigma, int do_mu, bool staged, hipStream_t st);

extern "C" int tmvb_ctm_create(tmvb_ctx* ctx, tmvb_corpus* corp, int32_t K, tmvb_ctm** out)
{
    TMVB_REQUIRE(out != nullptr, TMVB_EINVAL, "tmvb_ctm_create: out is NULL");
    *out = nullptr;
    TMVB_REQUIRE(ctx && corp, TMVB_EINVAL, "tmvb_ctm_create: NULL context or corpus");
    TMVB_REQUIRE(K > 0, TMVB_EINVAL, "number of topics must be a positive integer.");       // src/gpuCTM.jl constructor
    TMVB_REQUIRE(K <= CTM_MAX_K, TMVB_EINVAL, "tmvb_ctm_create: K <= %d (four topic slots per lane); got K=%d", CTM_MAX_K, K);
    TMVB_HIP(hipSetDevice(ctx->device));
    tmvb_ctm* h = new tmvb_ctm();
    tmvb_create_guard<tmvb_ctm, tmvb_ctm_destroy> guard{h};      // every early return below destroys h
    h->ctx = ctx; h->corp = corp; h->K = K; h->KP = tmvb_kpad(K);
    h->nslot = (K + 63) / 64; h->NB = (K + 31) / 32;
    { const char* e = getenv("TMVB_CTM_ELBO_PARTS"); h->parts_env = e ? atoi(e) : 1; }
    h->generic = !ctm_kp_supported(h->KP) || [] { const char* e = getenv("TMVB_CTM_FORCE_GENERIC"); return e && atoi(e) != 0; }();   // (the env: measurements)
    h->M = corp->info.M; h->V = corp->info.V; h->M_total = h->M;
    const size_t KM = (size_t)K * h->M, KPV = (size_t)h->KP * h->V + 4;
    h->docs_per_wave = 256;
    h->n_scatter_waves = (int)std::max<int64_t>(1, (h->M + h->docs_per_wave - 1) / h->docs_per_wave);
    int rc;
    if ((rc = dmalloc(&h->d_beta[0], KPV)) || (rc = dmalloc(&h->d_beta[1], KPV)) || (rc = dmalloc(&h->d_stats, (size_t)h->stats_len())) ||
        (rc = dmalloc(&h->d_lambda, KM)) || (rc = dmalloc(&h->d_lambda_old, KM)) || (rc = dmalloc(&h->d_vsq, KM)) ||
        (rc = dmalloc(&h->d_logzeta, (size_t)h->M)) || (rc = dmalloc(&h->d_wtok, (size_t)corp->info.nnz)) ||
        (rc = dmalloc(&h->d_E, (size_t)h->KP * h->M + 4)) || (rc = tmvb_corpus_term_index(corp)) ||
        (rc = dmalloc(&h->d_ts_partial, (size_t)corp->term_index.n_slots * (K + 1))) ||
        (rc = dmalloc(&h->d_invsigma_f, (size_t)h->KP * h->KP + 64)) || (rc = dmalloc(&h->d_mu_f, K)) ||
        (rc = dmalloc(&h->d_sigma, (size_t)K * K)) || (rc = dmalloc(&h->d_invsigma, (size_t)K * K)) || (rc = dmalloc(&h->d_mu, K)) ||
        (rc = dmalloc(&h->d_sigma_s, (size_t)K * K)) || (rc = dmalloc(&h->d_invsigma_s, (size_t)K * K)) || (rc = dmalloc(&h->d_invsigma_f_s, (size_t)h->KP * h->KP + 64)) ||
        (rc = dmalloc(&h->d_logdet_s, 1)) || (rc = dmalloc(&h->d_status_s, 1)) ||
        (rc = dmalloc(&h->d_logdet, 1)) || (rc = dmalloc(&h->d_scatter_partial, (size_t)h->n_scatter_waves * h->NB * h->NB * 1024)) ||
        (rc = dmalloc(&h->d_sweeps, (size_t)h->M)) || (rc = dmalloc(&h->d_doc_order, (size_t)h->M)) ||
        (rc = dmalloc(&h->d_partial, (size_t)TMVB_REDUCE_BLOCKS * K)) || (rc = dmalloc(&h->d_partial_docs, (size_t)TMVB_REDUCE_BLOCKS * K)) || (rc = dmalloc(&h->d_partial_docs2, (size_t)TMVB_REDUCE_BLOCKS * K)) ||
        (rc = dmalloc(&h->d_rowsum, K)) ||
        (rc = dmalloc(&h->d_doc_val, (size_t)h->M)) || (rc = dmalloc(&h->d_pdot, (size_t)std::max<int64_t>(h->M, 1))) || (rc = dmalloc(&h->d_pw_partial, 2048)) || (rc = dmalloc(&h->d_elbo, 1)) || (rc = dmalloc(&h->d_newton, 1)) ||
        (rc = dmalloc(&h->d_status, 1)) ||
        (K > 128 && ((rc = dmalloc(&h->d_sigma_work, (size_t)K * K)) || (rc = dmalloc(&h->d_sigma_work_s, (size_t)K * K))))) {
        return rc;
    }
    {
        // lane-per-document kernel: the default for KP <= 52 (K <= 50); TMVB_CTM_BATCH=0 selects the wave-per-document kernels
        const char* e = getenv("TMVB_CTM_BATCH");
        // (a lane walks its document's tokens one after the other: documents of thousands of unique terms would serialise their wave
        // on them, so THOSE documents -- not the corpus: round 2 sent every document of a corpus with one such document there -- keep
        // the wave-per-document kernel, which spreads a document's tokens over the lanes; h->n_long below)
        // (its token phase addresses the token arrays and the topic table with unsigned 32-bit byte offsets)
        h->batch = !h->generic && h->KP <= 52 && corp->info.nnz < (1ll << 30) && (int64_t)h->KP * h->V < (1ll << 30) && h->V < (1 << 24) &&
                   !(e && atoi(e) == 0);
        if (const char* t = getenv("TMVB_CTM_CG_TOL")) h->cg_tol = std::max(1e-7f, (float)atof(t));
        if (const char* t = getenv("TMVB_CTM_CG_ABS")) h->cg_abs = std::max(0.0f, (float)atof(t));
        if ((rc = dmalloc(&h->d_bt_sdiag, 64)) || (rc = dmalloc(&h->d_bt_muf, 64)) || (rc = dmalloc(&h->d_cg_iters, 16)) ||
            (rc = dmalloc(&h->d_q_S, (size_t)4 * (((size_t)h->KP * h->KP / 4 + 15) / 16 * 16) + 64)) || (rc = dmalloc(&h->d_q_sd, 64)) || (rc = dmalloc(&h->d_q_mu, 64)) ||
            (rc = dmalloc(&h->d_doc_newton, (size_t)std::max<int64_t>(h->M, 1))) || (rc = dmalloc(&h->d_doc_order0, (size_t)std::max<int64_t>(h->M, 1))) || (rc = dmalloc(&h->d_doc_order_q, (size_t)std::max<int64_t>(h->M, 1))) || (rc = dmalloc(&h->d_wave_keys, CTM_WAVESORT_MAX)))
            return rc;
        // regrouping pays when a chunk of the length-sorted order is still homogeneous in length: corpora of >= 4 chunks
        const char* r = getenv("TMVB_CTM_REORDER");
        h->reorder = h->batch && h->M >= 4 * CTM_REORDER_CHUNK && !(r && atoi(r) == 0);
    }
    std::vector<int32_t> order((size_t)h->M);
    std::iota(order.begin(), order.end(), 0);
    const std::vector<int64_t>& len = corp->h_doc_len;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return len[x] > len[y]; });
    h->n_long = 0;
    if (h->batch) {
        while (h->n_long < h->M && len[order[(size_t)h->n_long]] > CTM_BATCH_MAX_LEN) ++h->n_long;
    }
    {
        // The resident topic tile decides the waves per CU of the Newton-bound kernel (K = 50: a 96-row tile is 21 KB,
        // 7 waves per CU), and the Newton steps -- 97 % of the kernel -- never touch it.  A 32-row window (8 KB) that
        // longer documents stream through every sweep (L2-resident beta rows, LDS-DMA) lets the register file decide
        // the occupancy instead: 7.76 -> 5.97 ms per E-step on 32 000 SYN-NSF documents (64 / 24 / 16 / 8 KB: 7.76 / 6.80 /
        // 6.03 / 5.94 ms).  TMVB_CTM_MAX_TILE_KB overrides.
        size_t cap = 8 * 1024;
        if (const char* e = getenv("TMVB_CTM_MAX_TILE_KB")) cap = std::max<size_t>(4, (size_t)atoi(e)) * 1024;
        tmvb_build_lds_buckets(len, order, h->batch ? h->n_long : h->M, h->KP, -1, 3, h->buckets, cap);   // batch: the long documents only
    }
    if (h->M) TMVB_HIP(hipMemcpyAsync(h->d_doc_order, order.data(), (size_t)h->M * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (h->M) TMVB_HIP(hipMemcpyAsync(h->d_doc_order0, order.data(), (size_t)h->M * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_doc_newton, 0, std::max<size_t>((size_t)h->M, 1) * sizeof(uint16_t), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_cg_iters, 0, 16 * sizeof(unsigned long long), ctx->stream));      // tmvb_ctm_solver_stats before the first E-step
    TMVB_HIP(hipMemsetAsync(h->d_stats, 0, (size_t)h->stats_len() * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_sweeps, 0, std::max<size_t>((size_t)h->M, 1), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_newton, 0, sizeof(unsigned long long), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_status, 0, sizeof(int), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_beta[0], 0, KPV * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_beta[1], 0, KPV * sizeof(float), ctx->stream));
    TMVB_HIP(hipEventCreate(&h->ev0));
    TMVB_HIP(hipEventCreate(&h->ev1));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_fork, tmvb_event_flags()));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_spec, tmvb_event_flags()));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_spec2, tmvb_event_flags()));
    for (int a = 0; a < tmvb_ctm::NAUX; ++a) {
        h->aux[a] = tmvb_pool_stream(ctx->device, 1 + a);
        TMVB_REQUIRE(h->aux[a] != nullptr, TMVB_EHIP, "hipStreamCreate failed");
        TMVB_HIP(hipEventCreateWithFlags(&h->ev_join[a], tmvb_event_flags()));
    }
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    // constructor state, src/CTM.jl:37-48: mu = 0, sigma = invsigma = I, lambda = 0, vsq = 1, logzeta = 0.5; beta uniform
    std::vector<double> mu(K, 0.0), eye((size_t)K * K, 0.0), beta((size_t)K * h->V, h->V ? 1.0 / (double)h->V : 0.0);
    for (int i = 0; i < K; ++i) eye[(size_t)i * K + i] = 1.0;
    std::vector<double> lam(KM, 0.0), vsq(KM, 1.0), lz((size_t)h->M, 0.5);
    rc = tmvb_ctm_set_state(h, mu.data(), eye.data(), eye.data(), beta.data(), nullptr, lam.data(), nullptr, vsq.data(), lz.data(), nullptr);
    if (rc) return rc;
    guard.release();
    *out = h;
    return TMVB_OK;
}

static double host_logdet_spd(const double* A, int K, bool* ok)
{
    std::vector<double> L(A, A + (size_t)K * K);
    double ld = 0.0;
    *ok = true;
    for (int j = 0; j < K; ++j) {
        double dd = L[(size_t)j * K + j];
        for (int k = 0; k < j; ++k) dd -= L[(size_t)k * K + j] * L[(size_t)k * K + j];
        if (!(dd > 0.0)) { *ok = false; return NAN; }
        dd = std::sqrt(dd);
        L[(size_t)j * K + j] = dd;
        ld += 2.0 * std::log(dd);
        for (int i = j + 1; i < K; ++i) {
            double s = L[(size_t)j * K + i];
            for (int k = 0; k < j; ++k) s -= L[(size_t)k * K + i] * L[(size_t)k * K + j];
            L[(size_t)j * K + i] = s / dd;
        }
    }
    return ld;
}

extern "C" int tmvb_ctm_set_state(tmvb_ctm* h, const double* mu, const double* sigma, const double* invsigma,
                                  const double* beta, const double* beta_old, const double* lambda,
                                  const double* lambda_old, const double* vsq, const double* logzeta, const double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_set_state: handle is NULL");
    h->tail_fresh = false; h->sigma_staged = false;
    h->logz_valid = false; h->stats_fresh = false; h->pw_valid = false;
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const size_t K = h->K, KP = h->KP, KM = K * (size_t)h->M;
    int rc;
    if (mu) {
        for (size_t i = 0; i < K; ++i) TMVB_REQUIRE(std::isfinite(mu[i]), TMVB_ENONFINITE, "mu must be finite.");
        TMVB_HIP(hipMemcpyAsync(h->d_mu, mu, K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if ((rc = upload_f32(ctx, h->d_mu_f, mu, K))) return rc;
    }
    if (sigma) TMVB_HIP(hipMemcpyAsync(h->d_sigma, sigma, K * K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (invsigma) {
        bool ok;
        double ld = host_logdet_spd(invsigma, (int)K, &ok);
        TMVB_REQUIRE(ok, TMVB_ESHAPE, "invsigma must be positive-definite.");               // check_model, src/modelutils.jl:116
        std::vector<float> pad(KP * KP, 0.0f);
        for (size_t i = 0; i < K; ++i)
            for (size_t j = 0; j < K; ++j) pad[i * KP + j] = (float)invsigma[j * K + i];
        TMVB_HIP(hipMemcpyAsync(h->d_invsigma, invsigma, K * K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        TMVB_HIP(hipMemcpyAsync(h->d_invsigma_f, pad.data(), pad.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        TMVB_HIP(hipMemcpyAsync(h->d_logdet, &ld, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        TMVB_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (beta) {
        if ((rc = ctm_upload_beta(h, h->d_beta[h->cur], beta))) return rc;
        if (!beta_old && (rc = ctm_upload_beta(h, h->d_beta[h->cur ^ 1], beta))) return rc;
    }
    if (beta_old && (rc = ctm_upload_beta(h, h->d_beta[h->cur ^ 1], beta_old))) return rc;
    if (lambda) {
        if ((rc = upload_f32(ctx, h->d_lambda, lambda, KM))) return rc;
        if (!lambda_old && (rc = upload_f32(ctx, h->d_lambda_old, lambda, KM))) return rc;
    }
    if (lambda_old && (rc = upload_f32(ctx, h->d_lambda_old, lambda_old, KM))) return rc;
    if (vsq) {
        for (size_t q = 0; q < KM; ++q) TMVB_REQUIRE(vsq[q] > 0.0 && std::isfinite(vsq[q]), TMVB_ENONFINITE, "vsq must be positive.");
        if ((rc = upload_f32(ctx, h->d_vsq, vsq, KM))) return rc;
    }
    if (logzeta && (rc = upload_f32(ctx, h->d_logzeta, logzeta, (size_t)h->M))) return rc;
    if (elbo) h->elbo = *elbo;
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_ctm_get_state(tmvb_ctm* h, double* mu, double* sigma, double* invsigma, double* beta, double* beta_old,
                                  double* lambda, double* lambda_old, double* vsq, double* logzeta, double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_get_state: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const size_t K = h->K, KM = K * (size_t)h->M;
    int rc;
    if (mu) TMVB_HIP(hipMemcpyAsync(mu, h->d_mu, K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (sigma) TMVB_HIP(hipMemcpyAsync(sigma, h->d_sigma, K * K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (invsigma) TMVB_HIP(hipMemcpyAsync(invsigma, h->d_invsigma, K * K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    if (beta && (rc = ctm_download_beta(h, beta, h->d_beta[h->cur]))) return rc;
    if (beta_old && (rc = ctm_download_beta(h, beta_old, h->d_beta[h->cur ^ 1]))) return rc;
    if (lambda && (rc = download_f32(ctx, lambda, h->d_lambda, KM))) return rc;
    if (lambda_old && (rc = download_f32(ctx, lambda_old, h->d_lambda_old, KM))) return rc;
    if (vsq && (rc = download_f32(ctx, vsq, h->d_vsq, KM))) return rc;
    if (logzeta && (rc = download_f32(ctx, logzeta, h->d_logzeta, (size_t)h->M))) return rc;
    if (elbo) *elbo = h->elbo;
    return TMVB_OK;
}

// launch of the K > 60 kernel for CTM (FILT = false) and fCTM (FILT = true): the conjugate-gradient form as one persistent workgroup
// of several waves per CU (TMVB_CTM_GENERIC_CG=0: round 1's Gauss-Jordan form, one wave per document)
template <bool FILT>
static int ctm_launch_generic(tmvb_ctm* h, CtmParams p, double ntol)
{
    tmvb_ctx* ctx = h->ctx;
    const char* ecg = getenv("TMVB_CTM_GENERIC_CG");          // (read per launch: the tests run both forms in one process)
    const bool use_cg = !(ecg && atoi(ecg) == 0);
    h->generic_cg = use_cg;
    if (use_cg) {
        // instantiation by register budget: TMVB_CTM_CG_MAXW = 8 / 12 / 16 overrides the choice (measurements)
        // (measured on SYN-NSF, ms per E-step at iteration 30: K = 53: 16 waves / 128 VGPRs 7.9, 12 / 168 8.1, 8 / 256 9.8;  K = 64: 9.0, 9.1, 11.3;
        //  K = 80: 12 waves 15.4, 8 waves 18.0;  K = 100: 10 waves / 168 VGPRs 20.9, 8 / 256 22.3;  K = 128: 6 waves / 168 VGPRs 39.1, 6 / 256 37.1)
        int maxw = (h->K <= 64 || h->KP <= 108) ? 12 : 8;
        if (const char* e = getenv("TMVB_CTM_CG_MAXW")) { if (h->K > 64) maxw = (atoi(e) >= 12) ? 12 : 8; }
        const bool ga = h->K > 128;                            // four topic slots per lane, invsigma read from global memory (L2)
        if (ga) maxw = 8;
        int waves, tile_rows;
        ctm_generic_cg_shape(h->KP, maxw, &waves, &tile_rows, ga);
        p.tile_rows = tile_rows;
        const size_t lds = ctm_generic_lds_bytes(h->KP, waves, tile_rows) - (ga ? (size_t)h->KP * h->KP * sizeof(float) : 0);
        TMVB_HIP(hipMemsetAsync(h->d_cg_iters, 0, 16 * sizeof(unsigned long long), ctx->stream));
        p.queue = (unsigned*)(h->d_cg_iters + 12); p.n_docs = h->M; p.cg_diag = h->d_cg_iters;
        p.cg_tol2 = h->cg_tol * h->cg_tol; p.cg_maxit = 4 * h->KP;
        { const double fl = h->cg_abs * std::min(ntol, 4e-4); p.cg_abs2 = (float)(fl * fl); }
        const unsigned nwg = (unsigned)std::min<int64_t>(ctx->num_cu, (h->M + waves - 1) / waves);
        auto launch = [&](auto kern) -> int {
            TMVB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * waves), lds, ctx->stream, p, (int64_t)0);
            return TMVB_OK;
        };
        int lrc;
        if (ga) lrc = launch(ctm_estep_generic_kernel<4, FILT, true, 8, true>);
        else if (h->K <= 64) lrc = launch(ctm_estep_generic_kernel<1, FILT, true, 12>);   // (K = 61 ... 64: KP = 68, but one topic slot per lane)
        else lrc = (maxw == 12) ? launch(ctm_estep_generic_kernel<2, FILT, true, 12>) : launch(ctm_estep_generic_kernel<2, FILT, true, 8>);
        if (lrc) return lrc;
    } else {
        const size_t lds = ctm_generic_lds_bytes(h->KP);
        auto launch = [&](auto kern) -> int {
            TMVB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)h->M), dim3(64), lds, ctx->stream, p, (int64_t)0);
            return TMVB_OK;
        };
        TMVB_REQUIRE(h->K <= 128, TMVB_EINVAL, "TMVB_CTM_GENERIC_CG=0 (the Gauss-Jordan form through LDS) exists for K <= 128 only");
        int lrc = (h->KP > 64) ? launch(ctm_estep_generic_kernel<2, FILT, false, 1>) : launch(ctm_estep_generic_kernel<1, FILT, false, 1>);
        if (lrc) return lrc;
    }
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

// TMVB_CTM_ROWPAD: rows of 52 floats (208 B) straddle two or three 128-byte lines, rows at a 256-byte stride exactly two
__global__ __launch_bounds__(256) void ctm_rowpad_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t V)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;            // one 16-byte chunk of the padded table
    if (q >= V * 16) return;
    const int64_t v = q >> 4; const int c = (int)(q & 15);
    dst[q] = c < 13 ? src[v * 13 + c] : float4{0.f, 0.f, 0.f, 0.f};
}

// the same for any KP and any padded width: LPR 16-byte chunks per row in, CPR >= LPR chunks per row out, pads 0 (tmvb_ctm_quad.h: CPR = 4 ceil(LPR / 4),
// a row is whole 64-byte slots, one 16-byte chunk per lane of a quad)
__global__ __launch_bounds__(256) void ctm_rowpad_generic_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t V, int LPR, int CPR)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= V * CPR) return;
    const int64_t v = q / CPR; const int c = (int)(q - v * CPR);
    dst[q] = c < LPR ? src[v * LPR + c] : float4{0.f, 0.f, 0.f, 0.f};
}

// launch of the four-waves-per-item kernel (tmvb_ctm_quad.h): CTM, KP <= 52
static int ctm_launch_quad(tmvb_ctm* h, const CtmParams& p, double ntol)
{
    tmvb_ctx* ctx = h->ctx;
    hipLaunchKernelGGL(ctm_quad_tabs_kernel, dim3(1), dim3(256), 0, ctx->stream, h->K, h->KP, h->d_invsigma_f, h->d_mu_f, h->d_q_S, h->d_q_sd, h->d_q_mu, h->d_cg_iters);
    TMVB_HIP(hipGetLastError());
    CtmBatchTabs tb;
    tb.S = h->d_invsigma_f; tb.sdiag = h->d_bt_sdiag; tb.muf = h->d_bt_muf;
    tb.Sq = h->d_q_S; tb.sdq = h->d_q_sd; tb.muq = h->d_q_mu;
    tb.cg_tol2 = h->cg_tol * h->cg_tol; tb.cg_maxit = 4 * h->KP; tb.cg_iters = h->d_cg_iters;
    { const double fl = h->cg_abs * std::min(ntol, 4e-4); tb.cg_abs2 = (float)(fl * fl); }
    const int64_t Mb = h->M - h->n_long;
    if (Mb <= 0) return TMVB_OK;
    const int n_items = (int)((Mb + 63) / 64);
    const int LPRv = h->KP / 4, CPR = 4 * ((LPRv + 3) / 4);
    const size_t lds = (size_t)(5 * h->KP * 64 + 2 * 3 * 4 * 64) * 4 + (size_t)(2 * 4 * 64) * 8 + (5 * 64 + 4) * 4;      // = cq_dim<KP>::lds_bytes
    // two workgroups per CU: two waves per SIMD (TMVB_CTM_QUAD_PER_CU: diagnostics)
    int per_cu = TMVB_CTM_QWAVES;
    if (const char* e = getenv("TMVB_CTM_QUAD_PER_CU")) per_cu = std::max(1, std::min(4, atoi(e)));
    const dim3 grid((unsigned)std::min(n_items, per_cu * ctx->num_cu)), block(256);
    tb.next_item = (unsigned*)(h->d_cg_iters + 12); tb.n_items = n_items;
    if (h->reorder && h->keys_valid && !h->reorder_staged) {
        hipLaunchKernelGGL(ctm_reorder_kernel, dim3((unsigned)((Mb + CTM_REORDER_CHUNK - 1) / CTM_REORDER_CHUNK)), dim3(1024), 0, ctx->stream,
                           h->d_doc_order0 + h->n_long, h->d_doc_newton, h->d_doc_order + h->n_long, Mb);
        TMVB_HIP(hipGetLastError());
    }
    h->reorder_staged = false;
    h->keys_valid = true;
    CtmBatchArgs ba;
    ba.p = p; ba.p.doc_order = h->queue_sorted ? h->d_doc_order_q : p.doc_order + h->n_long; ba.p.doc_newton = h->d_doc_newton; ba.tb = tb; ba.M = Mb;
    if (!h->d_beta_pad || h->beta_pad_cpr != CPR) {
        (void)hipFree(h->d_beta_pad); h->d_beta_pad = nullptr;
        int prc = dmalloc(&h->d_beta_pad, (size_t)h->V * CPR * 4 + 64); if (prc) return prc;
        h->beta_pad_cpr = CPR;
    }
    hipLaunchKernelGGL(ctm_rowpad_generic_kernel, dim3((unsigned)((h->V * CPR + 255) / 256)), dim3(256), 0, ctx->stream, (const float4*)p.beta, (float4*)h->d_beta_pad, h->V, LPRv, CPR);
    TMVB_HIP(hipGetLastError());
    ba.p.beta = h->d_beta_pad;
    h->queue_sorted = false;
    // (the dynamic-LDS attribute once per HANDLE, not per process: one host thread may drive several devices, tmvb_ctm_train_group)
#define CTM_QCASE(KPV) case KPV: { if (!h->quad_attr_set && lds > 48 * 1024) { TMVB_HIP(hipFuncSetAttribute((const void*)ctm_estep_quad_kernel<KPV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); h->quad_attr_set = true; } \
                                   hipLaunchKernelGGL((ctm_estep_quad_kernel<KPV, false>), grid, block, lds, ctx->stream, ba); } break;
    static const bool qprof = [] { const char* e = getenv("TMVB_CTM_QPROF"); return e && atoi(e) != 0; }();
    if (qprof && h->KP == 52) {
        TMVB_HIP(hipFuncSetAttribute((const void*)ctm_estep_quad_kernel<52, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((ctm_estep_quad_kernel<52, true>), grid, block, lds, ctx->stream, ba);
    } else
    switch (h->KP) { CTM_QCASE(4) CTM_QCASE(12) CTM_QCASE(20) CTM_QCASE(28) CTM_QCASE(36) CTM_QCASE(44) CTM_QCASE(52)
                     default: TMVB_REQUIRE(false, TMVB_EINVAL, "ctm_launch_quad: KP not instantiated"); }
#undef CTM_QCASE
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

// launch of the lane-per-document kernel (tmvb_ctm_batch.h) for CTM (FILT = false) and fCTM (FILT = true)
template <bool FILT>
static int ctm_launch_batch(tmvb_ctm* h, const CtmParams& p, double ntol)
{
    tmvb_ctx* ctx = h->ctx;
    hipLaunchKernelGGL(ctm_batch_tabs_kernel, dim3(1), dim3(64), 0, ctx->stream, h->K, h->KP, h->d_invsigma_f, h->d_mu_f, h->d_bt_sdiag, h->d_bt_muf, h->d_cg_iters);
    TMVB_HIP(hipGetLastError());
    CtmBatchTabs tb;
    tb.S = h->d_invsigma_f; tb.sdiag = h->d_bt_sdiag; tb.muf = h->d_bt_muf;
    tb.cg_tol2 = h->cg_tol * h->cg_tol; tb.cg_maxit = 4 * h->KP; tb.cg_iters = h->d_cg_iters;
    { const double fl = h->cg_abs * std::min(ntol, 4e-4); tb.cg_abs2 = (float)(fl * fl); }   // never looser than at the reference K = 50 (ntol = 1/K^2)
    const int64_t Mb = h->M - h->n_long;                 // the documents behind the long ones in the processing order
    if (Mb <= 0) return TMVB_OK;
    // persistent launch: one 64-lane workgroup per SIMD of the device (the kernel takes a whole SIMD's register file), each pulling
    // waves-of-documents from the queue at d_cg_iters[12] (zeroed by ctm_batch_tabs_kernel); the order of the queue is the processing order, longest
    // documents first.  TMVB_CTM_PERSISTENT=0: one workgroup per wave-of-documents, placed by the hardware dispatcher.
    const int n_items = (int)((Mb + 63) / 64);
    static const bool persistent = [] { const char* e = getenv("TMVB_CTM_PERSISTENT"); return !(e && atoi(e) == 0); }();
    const size_t lds = (size_t)h->KP * 64 * (sizeof(double) + sizeof(float)) + 64 * sizeof(int32_t);   // vsq, CG solution / row staging, row ids
    int per_cu = (int)std::min<size_t>(4, (160 * 1024) / lds);            // one wave per SIMD, and what the CU's LDS holds
    if (const char* e = getenv("TMVB_CTM_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(e)));   // diagnostics: fewer resident waves (contention experiments, profiles/r5_ctm_token_experiments.txt)
    const dim3 grid((unsigned)(persistent ? std::min(n_items, per_cu * ctx->num_cu) : n_items)), block(64);
    tb.next_item = (unsigned*)(h->d_cg_iters + 12); tb.n_items = n_items;
    if (h->reorder && h->keys_valid && !h->reorder_staged) {      // (staged: the last tmvb_ctm_estep already regrouped behind its statistics tail)
        hipLaunchKernelGGL(ctm_reorder_kernel, dim3((unsigned)((Mb + CTM_REORDER_CHUNK - 1) / CTM_REORDER_CHUNK)), dim3(1024), 0, ctx->stream,
                           h->d_doc_order0 + h->n_long, h->d_doc_newton, h->d_doc_order + h->n_long, Mb);
        TMVB_HIP(hipGetLastError());
    }
    h->reorder_staged = false;
    h->keys_valid = true;
    CtmBatchArgs ba;
    ba.p = p; ba.p.doc_order = (h->queue_sorted && !FILT) ? h->d_doc_order_q : p.doc_order + h->n_long; ba.p.doc_newton = h->d_doc_newton; ba.tb = tb; ba.M = Mb;
    if (cb_rowb<52, FILT>::value == 256u && h->KP == 52) {
        if (!h->d_beta_pad) { int prc = dmalloc(&h->d_beta_pad, (size_t)h->V * 64); if (prc) return prc; }
        hipLaunchKernelGGL(ctm_rowpad_kernel, dim3((unsigned)((h->V * 16 + 255) / 256)), dim3(256), 0, ctx->stream, (const float4*)p.beta, (float4*)h->d_beta_pad, h->V);
        TMVB_HIP(hipGetLastError());
        ba.p.beta = h->d_beta_pad;
    }
    h->queue_sorted = false;
    static const bool prof = [] { const char* e = getenv("TMVB_CTM_PROF"); return e && atoi(e) != 0; }();
    // TMVB_CTM_WAVE_LOG=<file> (with TMVB_CTM_PROF=1): per-item start / end / placement of this launch, written after a synchronisation (diagnostics only)
    static const char* wave_log = getenv("TMVB_CTM_WAVE_LOG");
    static unsigned long long* d_wl = nullptr; static size_t wl_cap = 0;
    const bool logging = prof && h->KP == 52 && wave_log;
    if (logging) {
        if (wl_cap < (size_t)n_items * 4) { (void)hipFree(d_wl); wl_cap = (size_t)n_items * 4; TMVB_HIP(hipMalloc((void**)&d_wl, wl_cap * sizeof(unsigned long long))); }
        ba.tb.wave_log = d_wl;
    }
#define CTM_BCASE(KPV) case KPV: hipLaunchKernelGGL((ctm_estep_batch_kernel<KPV, false, FILT>), grid, block, lds, ctx->stream, ba); break;
    if (prof && h->KP == 52) hipLaunchKernelGGL((ctm_estep_batch_kernel<52, true, FILT>), grid, block, lds, ctx->stream, ba);
    else switch (h->KP) { CTM_BCASE(4) CTM_BCASE(12) CTM_BCASE(20) CTM_BCASE(28) CTM_BCASE(36) CTM_BCASE(44)
                          default: hipLaunchKernelGGL((ctm_estep_batch_kernel<52, false, FILT>), grid, block, lds, ctx->stream, ba); break; }
#undef CTM_BCASE
    TMVB_HIP(hipGetLastError());
    if (logging) {
        std::vector<unsigned long long> wl((size_t)n_items * 4);
        TMVB_HIP(hipMemcpyAsync(wl.data(), d_wl, wl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        TMVB_HIP(hipStreamSynchronize(ctx->stream));
        if (FILE* f = fopen(wave_log, "wb")) { fwrite(wl.data(), sizeof(unsigned long long), wl.size(), f); fclose(f); }
    }
    return TMVB_OK;
}

extern "C" int tmvb_ctm_estep(tmvb_ctm* h, int32_t niter, double ntol, int32_t viter, double vtol)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_estep: handle is NULL");
    TMVB_REQUIRE(viter >= 0 && niter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative.");   // src/gpuCTM.jl:490
    TMVB_REQUIRE(vtol >= 0 && ntol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");     // src/gpuCTM.jl:489
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    CtmParams p;
    p.K = h->K; p.KP = h->KP; p.LPR = h->KP / 4; p.lpr_magic = (unsigned)(0x100000000ull / (unsigned)p.LPR) + 1u;
    p.doc_ptr = h->corp->d_doc_ptr; p.terms = h->corp->d_terms; p.counts = h->corp->d_counts;
    p.doc_order = h->d_doc_order; p.tok_inv = h->corp->term_index.d_inv;
    p.beta = h->d_beta[h->cur]; p.invsigma = h->d_invsigma_f; p.mu = h->d_mu_f;
    p.lambda = h->d_lambda; p.lambda_old = h->d_lambda_old; p.vsq = h->d_vsq; p.logzeta = h->d_logzeta;
    p.wtok = h->d_wtok; p.E = h->d_E; p.estride = (h->KP / 4 <= 64) ? h->KP : h->K; p.sweeps = h->d_sweeps; p.newton_steps = h->d_newton;
    p.niter = niter; p.ntol = ntol; p.viter = viter; p.vtol = vtol;
    { const char* dbg = getenv("TMVB_DEBUG_FLAGS"); p.debug = dbg ? atoi(dbg) : 0; }
    p.store_w = tmvb_termstats_recomputes(h->KP, h->KP / 4 <= 64) ? 0 : 1;
    // decomposed update_elbo!: this iteration will be checked -- the document kernels leave pdot, the statistics pass the log-normaliser sums
    const bool collect = (h->parts_env == 2 || (h->parts_env != 0 && h->want_parts)) && p.store_w == 0 && viter > 0 && !(p.debug & 1);
    h->logz_valid = false; h->pw_valid = false; h->stats_fresh = false;
    if (collect) {
        const size_t need = (size_t)std::max<int64_t>(h->corp->term_index.n_chunks, 1);
        if (need > h->logz_cap) {
            (void)hipFree(h->d_logz); h->d_logz = nullptr; h->logz_cap = 0;
            int arc = dmalloc(&h->d_logz, need);
            if (arc) return arc;
            h->logz_cap = need;
        }
        p.pdot = h->d_pdot;
    }
    { int jrc = ctm_join_spec(h); if (jrc) return jrc; }        // the regrouped document order of the last E-step's side stream
    TMVB_HIP(hipEventRecord(h->ev0, ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_newton, 0, sizeof(unsigned long long), ctx->stream));
    if (h->generic && h->M > 0) {
        int lrc = ctm_launch_generic<false>(h, p, ntol);
        if (lrc) return lrc;
    }
    // buckets of the wave-per-document kernel: every document (no lane-per-document kernel for this K) or the long ones next to the
    // lane-per-document launch; on auxiliary streams whenever something else runs beside them
    const int nb = h->generic ? 0 : (int)h->buckets.size();
    const int naux = (nb > 1 || (nb > 0 && h->batch)) ? std::min(std::max(nb, 1), (int)tmvb_ctm::NAUX) : 0;
    if (naux > 0) {
        TMVB_HIP(hipEventRecord(h->ev_fork, ctx->stream));
        for (int a = 0; a < naux; ++a) TMVB_HIP(hipStreamWaitEvent(h->aux[a], h->ev_fork, 0));
    }
    for (int bi = 0; bi < nb; ++bi) {
        const tmvb_bucket& b = h->buckets[bi];
        hipStream_t st = (naux > 0) ? h->aux[bi % naux] : ctx->stream;
        const size_t lds = tmvb_tile_bytes(b.tile_rows, h->KP);
        const dim3 grid((unsigned)b.count), block(64);
#define CTM_CASE(KPV) case KPV: hipLaunchKernelGGL((ctm_estep_kernel<KPV, false>), grid, block, lds, st, p, b.first, b.tile_rows); break;
        // (measured: compiling for 4 waves per SIMD -- 128 VGPRs, 17 spilled -- changes nothing: 5.98 vs 5.97 ms)
        switch (h->KP) { CTM_CASE(4) CTM_CASE(12) CTM_CASE(20) CTM_CASE(28) CTM_CASE(36) CTM_CASE(44)
                         default: hipLaunchKernelGGL((ctm_estep_kernel<52, false>), grid, block, lds, st, p, b.first, b.tile_rows); break; }
#undef CTM_CASE
        TMVB_HIP(hipGetLastError());
    }
    if (h->batch && h->M > 0) {
        // round 6: four waves per wave-of-documents (tmvb_ctm_quad.h), two workgroups per CU; TMVB_CTM_QUAD=0 (or the profiling build of the
        // one-wave kernel, TMVB_CTM_PROF=1) selects round 3's one-wave kernel
        static const bool quad = [] { const char* e = getenv("TMVB_CTM_QUAD"); const char* pr = getenv("TMVB_CTM_PROF"); return !(e && atoi(e) == 0) && !(pr && atoi(pr) != 0); }();
        // (the four-waves kernel wants 78 KB of LDS per workgroup at KP = 52: on a device whose per-workgroup limit is below that the one-wave kernel -- 40 KB -- runs)
        if (h->lds_limit < 0) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 64 * 1024; }
            h->lds_limit = v;
        }
        const size_t quad_lds = (size_t)(5 * h->KP * 64 + 2 * 3 * 4 * 64) * 4 + (size_t)(2 * 4 * 64) * 8 + (5 * 64 + 4) * 4;
        int brc = (quad && quad_lds <= (size_t)h->lds_limit) ? ctm_launch_quad(h, p, ntol) : ctm_launch_batch<false>(h, p, ntol);
        if (brc) return brc;
    }
    if (naux > 0) {
        for (int a = 0; a < naux; ++a) {
            TMVB_HIP(hipEventRecord(h->ev_join[a], h->aux[a]));
            TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_join[a], 0));
        }
    }
    // The document reductions of tmvb_ctm_reduce_docs (sum lambda, sum vsq, the scatter matrix with the current mu: 0.15 ms of small
    // kernels) depend only on what the document kernels just wrote: they run here on a side stream under the statistics pass, and
    // tmvb_ctm_reduce_docs finds them done (tail_fresh; any change of lambda / vsq / mu through the API clears the flag).
    static const bool fold = [] { const char* e = getenv("TMVB_CTM_FOLD_REDUCE"); return !(e && atoi(e) == 0); }();
    const bool folded = fold && h->M > 0 && h->aux[0] != nullptr;
    if (folded) {
        TMVB_HIP(hipEventRecord(h->ev_fork, ctx->stream));
        TMVB_HIP(hipStreamWaitEvent(h->aux[0], h->ev_fork, 0));
        int frc = ctm_reduce_docs_on(h, h->aux[0]);
        if (frc) return frc;
        TMVB_HIP(hipEventRecord(h->ev_join[0], h->aux[0]));
        // behind the tail, still under the statistics pass: update_sigma! staged (single process only: a sharded run all-reduces the
        // tail first) and the documents regrouped for the next E-step
        const char* espec = getenv("TMVB_CTM_SPECULATE");     // (read per call: the tests run both ways in one process)
        const bool spec = !(espec && atoi(espec) == 0);
        if (spec) {
            if (!h->distributed) { frc = ctm_sigma_mu(h, 1, 0, true, h->aux[0]); if (frc) return frc; h->sigma_staged = true; }
            if (h->batch && h->reorder && h->keys_valid) {
                // on a second side stream (nothing but the E-step kernels to wait for): regroup, then the queue order of the launch
                const int64_t Mb = h->M - h->n_long;
                if (Mb > 0) {
                    hipStream_t s2 = h->aux[1];
                    TMVB_HIP(hipStreamWaitEvent(s2, h->ev_fork, 0));
                    hipLaunchKernelGGL(ctm_reorder_kernel, dim3((unsigned)((Mb + CTM_REORDER_CHUNK - 1) / CTM_REORDER_CHUNK)), dim3(1024), 0, s2,
                                       h->d_doc_order0 + h->n_long, h->d_doc_newton, h->d_doc_order + h->n_long, Mb);
                    TMVB_HIP(hipGetLastError());
                    h->reorder_staged = true;
                    const char* ews = getenv("TMVB_CTM_WAVESORT");
                    const bool wsort = !(ews && atoi(ews) == 0);
                    const int64_t nfull = Mb / 64;
                    if (wsort && nfull >= 2 && nfull <= CTM_WAVESORT_MAX) {
                        int P = 2; while (P < nfull) P <<= 1;
                        hipLaunchKernelGGL(ctm_wave_key_kernel, dim3((unsigned)nfull), dim3(64), 0, s2, h->d_doc_order + h->n_long, h->corp->d_doc_ptr,
                                           h->d_doc_newton, h->d_wave_keys);
                        hipLaunchKernelGGL(ctm_wave_sort_kernel, dim3(1), dim3(1024), 0, s2, h->d_wave_keys, (int)nfull, P);
                        hipLaunchKernelGGL(ctm_wave_permute_kernel, dim3((unsigned)((Mb + 255) / 256)), dim3(256), 0, s2, h->d_doc_order + h->n_long,
                                           h->d_wave_keys, (int)nfull, Mb, h->d_doc_order_q);
                        TMVB_HIP(hipGetLastError());
                        h->queue_sorted = true;
                    }
                    TMVB_HIP(hipEventRecord(h->ev_spec2, s2));
                    h->spec2_pending = true;
                }
            }
            TMVB_HIP(hipEventRecord(h->ev_spec, h->aux[0]));
            h->spec_pending = true;
        }
    }
    // update_beta!(model, d)  src/CTM.jl:122-125 as the gather-side statistics pass (no epsilon in CTM's phi)
    TermStatsParams tp;
    tp.K = h->K; tp.tstride = h->KP; tp.ostride = h->K;
    tp.w = h->d_wtok; tp.E = h->d_E; tp.T = h->d_beta[h->cur]; tp.eps = 0.0f; tp.base = 0.0f; tp.keps = 0.0f;
    tp.out = h->d_stats; tp.partial = h->d_ts_partial;
    tp.logz = collect ? h->d_logz : nullptr;
    int rc = tmvb_launch_termstats(ctx, h->nslot, h->KP, h->KP / 4 <= 64, h->corp->term_index, tp);
    if (folded) { TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_join[0], 0)); h->tail_fresh = (rc == TMVB_OK); }
    if (rc) return rc;
    h->logz_valid = collect; h->n_logz = collect ? h->corp->term_index.n_chunks : 0; h->stats_fresh = true;
    TMVB_HIP(hipEventRecord(h->ev1, ctx->stream));
    h->timed = true;
    return TMVB_OK;
}

// sum_d lambda_d, sum_d vsq_d and the scatter matrix (with the CURRENT = previous-iteration mu) into the statistics tail
static int ctm_reduce_docs_on(tmvb_ctm* h, hipStream_t st)
{
    tmvb_ctx* ctx = h->ctx;
    int rc;
    if (h->K <= 64) {       // sum lambda and sum vsq in one pair of launches (same blocks, same summation order as tmvb_colsum)
        if ((rc = tmvb_colsum2(ctx, h->K, {h->d_lambda, h->M, h->d_partial_docs, nullptr, h->tail()},
                               {h->d_vsq, h->M, h->d_partial_docs2, nullptr, h->tail() + h->K}, st))) return rc;
    } else {
        if ((rc = tmvb_colsum(ctx, h->nslot, h->K, h->d_lambda, h->M, h->d_partial_docs, nullptr, h->tail(), st))) return rc;
        if ((rc = tmvb_colsum(ctx, h->nslot, h->K, h->d_vsq, h->M, h->d_partial_docs, nullptr, h->tail() + h->K, st))) return rc;
    }
    if (h->M > 0) {
        hipLaunchKernelGGL(ctm_scatter_mfma_kernel, dim3(h->n_scatter_waves, h->NB * h->NB), dim3(64), 0, st, h->d_lambda, h->d_mu_f,
                           h->K, h->NB, h->M, h->docs_per_wave, h->d_scatter_partial);
        TMVB_HIP(hipGetLastError());
        hipLaunchKernelGGL(ctm_scatter_reduce_kernel, dim3((h->K * h->K + 15) / 16), dim3(256), 0, st, h->d_scatter_partial,
                           h->n_scatter_waves, h->K, h->NB, h->tail() + 2 * h->K);
        TMVB_HIP(hipGetLastError());
    } else {
        TMVB_HIP(hipMemsetAsync(h->tail() + 2 * h->K, 0, (size_t)h->K * h->K * sizeof(float), st));
    }
    return TMVB_OK;
}

extern "C" int tmvb_ctm_reduce_docs(tmvb_ctm* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_reduce_docs: handle is NULL");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    // tmvb_ctm_estep already computed the tail (on a side stream, under its statistics pass) from the state it left behind
    if (h->tail_fresh) { h->tail_fresh = false; return TMVB_OK; }
    // recomputing the tail: a sigma staged from the previous tail is not this tail's, and the staging kernel may still be reading it
    h->sigma_staged = false;
    { int jrc = ctm_join_spec(h); if (jrc) return jrc; }
    return ctm_reduce_docs_on(h, h->ctx->stream);
}

extern "C" int tmvb_ctm_stats(tmvb_ctm* h, void** dev_ptr, int64_t* n_f32)
{
    TMVB_REQUIRE(h && dev_ptr && n_f32, TMVB_EINVAL, "tmvb_ctm_stats: NULL argument");
    // the caller may rewrite the tail through this pointer (a host-side all-reduce): a sigma that tmvb_ctm_estep inverted
    // speculatively from the tail as it is NOW must not be committed afterwards (round-3 advice), and whatever the caller
    // enqueues on the context's stream must come after the staging kernel's reads
    h->sigma_staged = false;
    TMVB_HIP(hipSetDevice(h->ctx->device));
    { int jrc = ctm_join_spec(h); if (jrc) return jrc; }
    *dev_ptr = h->d_stats;
    *n_f32 = h->stats_len();
    return TMVB_OK;
}

extern "C" int tmvb_ctm_bind_stats(tmvb_ctm* h, void* dev_ptr, int64_t n_f32)
{
    TMVB_REQUIRE(h && dev_ptr, TMVB_EINVAL, "tmvb_ctm_bind_stats: NULL argument");
    TMVB_REQUIRE(n_f32 >= h->stats_len(), TMVB_ESHAPE, "tmvb_ctm_bind_stats: buffer holds %lld floats, need %lld", (long long)n_f32, (long long)h->stats_len());
    TMVB_HIP(hipSetDevice(h->ctx->device));
    h->sigma_staged = false;                                    // as tmvb_ctm_stats: the tail leaves the library's hands
    { int jrc = ctm_join_spec(h); if (jrc) return jrc; }
    TMVB_HIP(hipMemcpyAsync(dev_ptr, h->d_stats, (size_t)h->stats_len() * sizeof(float), hipMemcpyDeviceToDevice, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    if (h->own_stats) (void)hipFree(h->d_stats);
    h->d_stats = (float*)dev_ptr;
    h->own_stats = false;
    return TMVB_OK;
}

extern "C" int tmvb_ctm_set_distributed(tmvb_ctm* h, int64_t M_total, int32_t distributed)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_set_distributed: handle is NULL");
    TMVB_REQUIRE(M_total >= h->M, TMVB_ESHAPE, "tmvb_ctm_set_distributed: M_total < local M");
    h->M_total = M_total;
    h->distributed = distributed != 0;
    return TMVB_OK;
}

extern "C" int tmvb_ctm_update_beta(tmvb_ctm* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_update_beta: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    int rc = tmvb_colsum(ctx, h->nslot, h->K, h->d_stats, h->V, h->d_partial, h->d_rowsum, nullptr);
    if (rc) return rc;
    const int64_t total = (int64_t)h->KP * h->V;
    int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (total + 255) / 256));
    const bool fparts = h->filt_parts;                      // fCTM (src/fCTM.jl:89, :109): both logarithms carry the epsilon
    const bool parts = (h->logz_valid && h->stats_fresh) || fparts;     // the coming update_elbo! is the decomposed one (src/CTM.jl:71 has the epsilon, :93 has none)
    hipLaunchKernelGGL(beta_norm_kernel, dim3(nb), dim3(256), (size_t)h->K * sizeof(double), ctx->stream,
                       h->d_stats, h->d_rowsum, h->d_beta[h->cur ^ 1], h->K, h->KP, h->V, parts ? h->d_pw_partial : (double*)nullptr, parts ? TMVB_EPS_F : 0.0f,
                       parts ? (const float*)h->d_beta[h->cur] : (const float*)nullptr, fparts ? TMVB_EPS_F : 0.0f);
    TMVB_HIP(hipGetLastError());
    h->pw_blocks = nb; h->pw_valid = parts && !fparts; h->stats_fresh = false; h->filt_pw_valid = fparts; h->filt_parts = false;
    h->cur ^= 1;
    return TMVB_OK;
}

// staged = true: update_sigma! into the staging buffers on stream st (tmvb_ctm_estep's side stream)
static int ctm_sigma_mu(tmvb_ctm* h, int do_sigma, int do_mu, bool staged, hipStream_t st)
{
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const bool in_lds = h->K <= 128;
    const size_t lds = in_lds ? (size_t)h->K * h->K * sizeof(double) : 0;
    if (lds > 48 * 1024) TMVB_HIP(hipFuncSetAttribute((const void*)ctm_sigma_mu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (staged)
        hipLaunchKernelGGL(ctm_sigma_mu_kernel, dim3(1), dim3(256), lds, st, h->K, h->KP, (double)h->M_total, h->tail(),
                           h->d_sigma_s, h->d_invsigma_s, h->d_invsigma_f_s, h->d_mu, h->d_mu_f, h->d_logdet_s, h->d_status_s, 1, 0,
                           in_lds ? (double*)nullptr : h->d_sigma_work_s);
    else
        hipLaunchKernelGGL(ctm_sigma_mu_kernel, dim3(1), dim3(256), lds, ctx->stream, h->K, h->KP, (double)h->M_total, h->tail(),
                           h->d_sigma, h->d_invsigma, h->d_invsigma_f, h->d_mu, h->d_mu_f, h->d_logdet, h->d_status, do_sigma, do_mu,
                           in_lds ? (double*)nullptr : h->d_sigma_work);
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

// the context's stream waits for what tmvb_ctm_estep left running on its side stream behind the statistics tail
static int ctm_join_spec(tmvb_ctm* h)
{
    if (h->spec_pending) { TMVB_HIP(hipStreamWaitEvent(h->ctx->stream, h->ev_spec, 0)); h->spec_pending = false; }
    if (h->spec2_pending) { TMVB_HIP(hipStreamWaitEvent(h->ctx->stream, h->ev_spec2, 0)); h->spec2_pending = false; }
    return TMVB_OK;
}

extern "C" int tmvb_ctm_update_sigma(tmvb_ctm* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_update_sigma: handle is NULL");
    if (h->sigma_staged) {
        // tmvb_ctm_estep already inverted: sigma of exactly this statistics tail (any change of lambda / vsq / mu / the tail through the
        // API clears the flag) sits in the staging buffers
        tmvb_ctx* ctx = h->ctx;
        TMVB_HIP(hipSetDevice(ctx->device));
        int jrc = ctm_join_spec(h);
        if (jrc) return jrc;
        const int KK = h->K * h->K, KPKP = h->KP * h->KP;
        hipLaunchKernelGGL(ctm_commit_sigma_kernel, dim3((unsigned)((KPKP + 255) / 256)), dim3(256), 0, ctx->stream, KK, KPKP, h->d_sigma_s, h->d_invsigma_s,
                           h->d_invsigma_f_s, h->d_logdet_s, h->d_status_s, h->d_sigma, h->d_invsigma, h->d_invsigma_f, h->d_logdet, h->d_status);
        TMVB_HIP(hipGetLastError());
        h->sigma_staged = false;
        return TMVB_OK;
    }
    return ctm_sigma_mu(h, 1, 0, false, nullptr);
}

extern "C" int tmvb_ctm_update_mu(tmvb_ctm* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_update_mu: handle is NULL");
    h->tail_fresh = false; h->sigma_staged = false;     // the scatter matrix of the tail was taken about the mu that changes now
    TMVB_HIP(hipSetDevice(h->ctx->device));
    { int jrc = ctm_join_spec(h); if (jrc) return jrc; }        // a staging kernel still running on the side stream reads d_mu / d_mu_f
    return ctm_sigma_mu(h, 0, 1, false, nullptr);
}

extern "C" int tmvb_ctm_update_elbo(tmvb_ctm* h, double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_update_elbo: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const bool parts = h->logz_valid && h->pw_valid && !h->force_walk;         // everything per token was left behind by the iteration itself
    h->elbo_form = parts ? 1 : 0;
    if (parts && h->M > 0) {
#define CTM_ELBO_DOC(NSV) hipLaunchKernelGGL((ctm_elbo_kernel<NSV, false>), dim3((unsigned)h->M), dim3(64), 0, ctx->stream, h->K, h->KP, h->corp->d_doc_ptr, \
                               h->corp->d_terms, h->corp->d_counts, h->d_mu, h->d_invsigma, h->d_logdet, h->d_beta[h->cur], \
                               h->d_beta[h->cur ^ 1], h->d_lambda, h->d_lambda_old, h->d_vsq, h->d_logzeta, h->d_doc_val, h->d_pdot)
        if (h->nslot == 1) CTM_ELBO_DOC(1); else if (h->nslot == 2) CTM_ELBO_DOC(2); else CTM_ELBO_DOC(4);
#undef CTM_ELBO_DOC
        TMVB_HIP(hipGetLastError());
    } else
    if (h->M > 0) {
        if (h->nslot == 1)
            hipLaunchKernelGGL((ctm_elbo_kernel<1>), dim3((unsigned)h->M), dim3(64), 0, ctx->stream, h->K, h->KP, h->corp->d_doc_ptr,
                               h->corp->d_terms, h->corp->d_counts, h->d_mu, h->d_invsigma, h->d_logdet, h->d_beta[h->cur],
                               h->d_beta[h->cur ^ 1], h->d_lambda, h->d_lambda_old, h->d_vsq, h->d_logzeta, h->d_doc_val);
        else if (h->nslot == 2)
            hipLaunchKernelGGL((ctm_elbo_kernel<2>), dim3((unsigned)h->M), dim3(64), 0, ctx->stream, h->K, h->KP, h->corp->d_doc_ptr,
                               h->corp->d_terms, h->corp->d_counts, h->d_mu, h->d_invsigma, h->d_logdet, h->d_beta[h->cur],
                               h->d_beta[h->cur ^ 1], h->d_lambda, h->d_lambda_old, h->d_vsq, h->d_logzeta, h->d_doc_val);
        else
            hipLaunchKernelGGL((ctm_elbo_kernel<4>), dim3((unsigned)h->M), dim3(64), 0, ctx->stream, h->K, h->KP, h->corp->d_doc_ptr,
                               h->corp->d_terms, h->corp->d_counts, h->d_mu, h->d_invsigma, h->d_logdet, h->d_beta[h->cur],
                               h->d_beta[h->cur ^ 1], h->d_lambda, h->d_lambda_old, h->d_vsq, h->d_logzeta, h->d_doc_val);
        TMVB_HIP(hipGetLastError());
    }
    if (parts)
        hipLaunchKernelGGL(ctm_elbo_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, h->d_doc_val, h->M, h->d_logz, h->n_logz, h->d_pw_partial, h->pw_blocks,
                           h->distributed && h->M_total > 0 ? (double)h->M / (double)h->M_total : 1.0, h->d_elbo);
    else
        hipLaunchKernelGGL(sum_docs_kernel, dim3(1), dim3(1024), 0, ctx->stream, h->d_doc_val, h->M, h->d_elbo);
    TMVB_HIP(hipGetLastError());
    double v = 0.0;
    int st = 0;
    TMVB_HIP(hipMemcpyAsync(&v, h->d_elbo, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TMVB_HIP(hipMemcpyAsync(&st, h->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    TMVB_REQUIRE(st == 0, TMVB_ENONFINITE, "sigma must be positive-definite.");
    h->elbo = v;
    if (elbo) *elbo = v;
    return TMVB_OK;
}

extern "C" int tmvb_ctm_elbo_form(tmvb_ctm* h, int32_t* form)
{
    TMVB_REQUIRE(h && form, TMVB_EINVAL, "tmvb_ctm_elbo_form: NULL argument");
    *form = h->elbo_form;
    return TMVB_OK;
}

extern "C" int tmvb_ctm_set_comm(tmvb_ctm* h, tmvb_comm* comm, int64_t M_total)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_set_comm: handle is NULL");
    int rc = tmvb_ctm_set_distributed(h, comm ? M_total : h->M, comm != nullptr);
    if (rc) return rc;
    h->comm = comm;
    return TMVB_OK;
}

namespace {
struct CtmTrainOps {
    int niter, viter; double ntol, vtol;
    int estep(tmvb_ctm* h) { return tmvb_ctm_estep(h, niter, ntol, viter, vtol); }     // src/CTM.jl:194-205
    int reduce(tmvb_ctm* h) { return tmvb_ctm_reduce_docs(h); }
    int before_allreduce(tmvb_ctm*) { return TMVB_OK; }
    float* stats(tmvb_ctm* h) { return h->d_stats; }
    int64_t stats_len(tmvb_ctm* h) { return h->stats_len(); }
    int mstep(tmvb_ctm* h)
    {
        int rc = tmvb_ctm_update_beta(h);                                              // :206
        if (!rc) rc = tmvb_ctm_update_sigma(h);                                        // :207 (previous mu, quirk Q2)
        if (!rc) rc = tmvb_ctm_update_mu(h);                                           // :208
        return rc;
    }
    int elbo_local(tmvb_ctm* h, double* s, double* once) { *once = 0.0; return tmvb_ctm_update_elbo(h, s); }
    int elbo_form(tmvb_ctm* h) { return h->elbo_form; }
    void force_walk(tmvb_ctm* h, bool on, bool doubled = true) { h->force_walk = on; if (!on && doubled) h->elbo_form = 1; }   // (switched off behind the one evaluation that doubled a decomposed one)
    void will_check(tmvb_ctm* h, bool checked) { h->want_parts = checked; }             // the coming iteration ends in check_elbo!
    double* elbo_dev(tmvb_ctm* h) { return h->d_elbo; }
    tmvb_comm* comm(tmvb_ctm* h) { return h->comm; }
    bool distributed(tmvb_ctm* h) { return h->distributed; }
    tmvb_ctx* ctx(tmvb_ctm* h) { return h->ctx; }
    int64_t nnz(tmvb_ctm* h) { return h->corp->info.nnz; }
    void set_elbo(tmvb_ctm* h, double v) { h->elbo = v; }
    double get_elbo(tmvb_ctm* h) { return h->elbo; }
    int finish(tmvb_ctm* h)
    {
        TMVB_HIP(hipSetDevice(h->ctx->device));
        TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
        return TMVB_OK;
    }
};
}  // namespace

extern "C" int tmvb_ctm_train_group(tmvb_ctm* const* hs, int32_t n, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter,
                                    double vtol, int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline)
{
    TMVB_REQUIRE(tol >= 0 && ntol >= 0 && vtol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");   // src/gpuCTM.jl:489
    TMVB_REQUIRE(iter >= 0 && niter >= 0 && viter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative."); // :490
    CtmTrainOps ops{niter, viter, ntol, vtol};
    return tmvb_train_group_loop("tmvb_ctm_train", hs, n, iter, tol, checkelbo, elbo_traj, iters_done, elbo_baseline, ops);
}

extern "C" int tmvb_ctm_train(tmvb_ctm* h, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter,
                              double vtol, int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctm_train: handle is NULL");
    return tmvb_ctm_train_group(&h, 1, iter, tol, niter, ntol, viter, vtol, checkelbo, elbo_traj, iters_done, elbo_baseline);
}

// per-document sweep counts of the last E-step (document order of the corpus), for parity tests that compare the
// state of exactly those documents whose exit sweep agrees with the oracle's
// Diagnostics of the last E-step of the lane-per-document kernel (12 values): [0] CG trips, [1] Newton trips, [2] waves (all
// summed over waves); with TMVB_CTM_PROF=1 also [3..10] shader cycles per phase (token, logzeta, vsq, gradient assembly, CG,
// gradient mat-vec, lambda update, spare) and [11] whole-kernel cycles.  All zero when the wave-per-document kernels ran.
extern "C" int tmvb_ctm_solver_stats(tmvb_ctm* h, int64_t* out9)
{
    TMVB_REQUIRE(h != nullptr && out9 != nullptr, TMVB_EINVAL, "tmvb_ctm_solver_stats: NULL argument");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    unsigned long long v[12] = {0};
    if (h->batch || (h->generic && h->generic_cg)) {
        TMVB_HIP(hipMemcpyAsync(v, h->d_cg_iters, sizeof(v), hipMemcpyDeviceToHost, h->ctx->stream));
        TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    }
    for (int q = 0; q < 12; ++q) out9[q] = (int64_t)v[q];
    return TMVB_OK;
}

extern "C" int tmvb_ctm_doc_sweeps(tmvb_ctm* h, uint8_t* out)
{
    TMVB_REQUIRE(h && (out || h->M == 0), TMVB_EINVAL, "tmvb_ctm_doc_sweeps: NULL argument");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    if (h->M) TMVB_HIP(hipMemcpyAsync(out, h->d_sweeps, (size_t)h->M, hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_ctm_sweep_hist(tmvb_ctm* h, int64_t* hist, int32_t nbins, int64_t* newton_steps)
{
    TMVB_REQUIRE(h && hist && nbins > 0, TMVB_EINVAL, "tmvb_ctm_sweep_hist: bad argument");
    std::vector<uint8_t> sw((size_t)h->M);
    unsigned long long ns = 0;
    TMVB_HIP(hipSetDevice(h->ctx->device));
    if (h->M) TMVB_HIP(hipMemcpyAsync(sw.data(), h->d_sweeps, (size_t)h->M, hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipMemcpyAsync(&ns, h->d_newton, sizeof(ns), hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    for (int b = 0; b < nbins; ++b) hist[b] = 0;
    for (uint8_t s : sw) hist[std::min<int>(s, nbins - 1)]++;
    if (newton_steps) *newton_steps = (int64_t)ns;
    return TMVB_OK;
}

extern "C" int tmvb_ctm_last_estep_ms(tmvb_ctm* h, float* ms)
{
    TMVB_REQUIRE(h && ms, TMVB_EINVAL, "tmvb_ctm_last_estep_ms: NULL argument");
    TMVB_REQUIRE(h->timed, TMVB_EINVAL, "tmvb_ctm_last_estep_ms: no E-step has run");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    TMVB_HIP(hipEventSynchronize(h->ev1));
    TMVB_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return TMVB_OK;
}

// ====================================================================================================================
// filtered CTM (src/fCTM.jl) -- new device path; the reference's `@gpu train!` does nothing for an fCTM (src/macros.jl:277).
// The handle wraps a CTM handle (same Newton machinery, M-step for sigma / mu, MFMA scatter) and adds the per-token switch
// tau, the background distribution kappa, the log table L = log(beta + eps) and the rebuilt-phi statistics pass shared
// with fLDA (tmvb_filtered.h).  Statistics buffer: [ S (K*V) | sum lambda (K) | sum vsq (K) | scatter (K*K) | kappa_stats (V) ].
// ====================================================================================================================

// update_elbo!  src/fCTM.jl:105-115 per document (terms :68-102); phi rebuilt from tau_old, beta_old, lambda_old (:109)
template <int NS>
__global__ __launch_bounds__(64) void fctm_elbo_kernel(int K, int KP, const int64_t* __restrict__ doc_ptr,
                                                       const int32_t* __restrict__ terms, const int32_t* __restrict__ counts,
                                                       const double* __restrict__ mu_d, const double* __restrict__ invsigma_d,
                                                       const double* __restrict__ logdet_inv, double eta, const float* __restrict__ kappa,
                                                       const float* __restrict__ beta, const float* __restrict__ beta_old,
                                                       const float* __restrict__ lambda, const float* __restrict__ lambda_old,
                                                       const float* __restrict__ vsq, const float* __restrict__ logzeta,
                                                       const float* __restrict__ tau, const float* __restrict__ tau_old,
                                                       double* __restrict__ doc_val)
{
    const int lane = threadIdx.x;
    const int d = blockIdx.x;
    const int64_t off = doc_ptr[d];
    const int N = (int)(doc_ptr[d + 1] - off);
    bool on[NS]; int ix[NS];
    float lo[NS];
    double l[NS], v[NS], df[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        on[s] = lane + 64 * s < K;
        ix[s] = on[s] ? lane + 64 * s : 0;
        lo[s] = on[s] ? lambda_old[(int64_t)d * K + ix[s]] : -INFINITY;
        l[s] = on[s] ? (double)lambda[(int64_t)d * K + ix[s]] : 0.0;
        v[s] = on[s] ? (double)vsq[(int64_t)d * K + ix[s]] : 1.0;
        df[s] = on[s] ? l[s] - mu_d[ix[s]] : 0.0;
    }
    const double lz = (double)logzeta[d];
    double acc = 0.0, Cd = 0.0, ta = 0.0;
    for (int n = 0; n < N; ++n) {
        const int t = terms[off + n];
        const float c = (float)counts[off + n];
        const float tn = tau[off + n], to = tau_old[off + n];
        Cd += (double)c; ta += (double)tn * (double)c;
        float x[NS], ml = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            x[s] = on[s] ? fmaf(to, logf(beta_old[(int64_t)t * KP + ix[s]] + TMVB_EPS_F), lo[s]) : -INFINITY;   // :109
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
            acc += cp * l[s];                                                                  // Elogpz :82 (first part)
            acc += cp * (double)tn * (double)logf(beta[(int64_t)t * KP + ix[s]] + TMVB_EPS_F);  // Elogpw :89, topical part
            if (ph > 0.0f) acc -= cp * (double)logf(ph);                                       // -Elogqz :108-111
        }
        if (lane == 0) {
            acc += (double)c * (1.0 - (double)tn) * (double)logf(kappa[t] + TMVB_EPS_F);       // Elogpw :89, background part
            if (tn > 0.0f && tn < 1.0f) acc -= (double)c * ((double)tn * log((double)tn) + (1.0 - (double)tn) * log(1.0 - (double)tn));   // -Elogqc :101-104
        }
    }
    // Elogpeta :69
    double mv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) mv[s] = 0.0;
    for (int j = 0; j < K; ++j) {
        double dj = 0.0;                                                      // df of topic j: slot j / 64 (uniform), lane j % 64
#pragma unroll
        for (int s = 0; s < NS; ++s) if ((j >> 6) == s) dj = readlane_d(df[s], j & 63);
#pragma unroll
        for (int s = 0; s < NS; ++s) mv[s] = fma(on[s] ? invsigma_d[(int64_t)j * K + ix[s]] : 0.0, dj, mv[s]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (!on[s]) continue;
        acc += -0.5 * (invsigma_d[(int64_t)ix[s] * K + ix[s]] * v[s] + df[s] * mv[s]);
        acc -= Cd * exp(l[s] + 0.5 * v[s] - lz);                                               // Elogpz :82 (second part)
        acc += 0.5 * log(v[s]);                                                                // -Elogqeta :95-97
    }
    double tot = wave_sum_d(acc);
    const double TWO_PI_LOG = 1.8378770664093453;
    tot += 0.5 * (*logdet_inv - (double)K * TWO_PI_LOG);
    tot -= Cd * (lz - 1.0);
    tot += 0.5 * (double)K * (1.0 + TWO_PI_LOG);
    tot += log(TMVB_EPS_D + pow(eta, ta) * pow(1.0 - eta, Cd - ta));                           // Elogpc :74-77
    if (lane == 0) doc_val[d] = tot;
}

// update_elbo! of fCTM without rebuilding phi (round 6; the fLDA form, flda_elbo_doc_parts_kernel in tmvb_flda.hip, with lambda for Elogtheta): the token
// terms of Elogpz (:82), Elogpw (:89) and -Elogqz (:108-111) are c_n [sum_i phi_in (lambda_i - lambda_old_i) + tau_n sum_i phi_in L_new - tau_old_n A_n + lse_n];
// the first sum is pdot[d] of the E-step kernels' exit test (CtmParams::pdot), the second sum S L_new of update_beta! (its difference form also carries
// -tau_n A_n), lse_n and A_n are per-token floats the E-step stores.  One wave per document: lane = token for the elementwise sum, lane = topic for
// Elogpeta / the logzeta bound / the Gaussian entropy (as fctm_elbo_kernel).
template <int NS>
__global__ __launch_bounds__(64) void fctm_elbo_doc_parts_kernel(int K, const int64_t* __restrict__ doc_ptr, const int32_t* __restrict__ terms,
                                                                 const int32_t* __restrict__ counts, const double* __restrict__ mu_d,
                                                                 const double* __restrict__ invsigma_d, const double* __restrict__ logdet_inv, double eta,
                                                                 const float* __restrict__ kappa, const float* __restrict__ lambda,
                                                                 const float* __restrict__ vsq, const float* __restrict__ logzeta,
                                                                 const float* __restrict__ tau, const float* __restrict__ tau_old,
                                                                 const float* __restrict__ lse, const float* __restrict__ aold,
                                                                 const float* __restrict__ pdot, double* __restrict__ doc_val)
{
    const int lane = threadIdx.x;
    const int d = blockIdx.x;
    const int64_t off = doc_ptr[d];
    const int N = (int)(doc_ptr[d + 1] - off);
    bool on[NS]; int ix[NS];
    double l[NS], v[NS], df[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        on[s] = lane + 64 * s < K;
        ix[s] = on[s] ? lane + 64 * s : 0;
        l[s] = on[s] ? (double)lambda[(int64_t)d * K + ix[s]] : 0.0;
        v[s] = on[s] ? (double)vsq[(int64_t)d * K + ix[s]] : 1.0;
        df[s] = on[s] ? l[s] - mu_d[ix[s]] : 0.0;
    }
    const double lz = (double)logzeta[d];
    double acc = 0.0, Cd = 0.0, ta = 0.0;
    for (int n = lane; n < N; n += 64) {
        const double c = (double)counts[off + n];
        const double tn = (double)tau[off + n], to = (double)tau_old[off + n];
        double x = (double)lse[off + n] + (tn - to) * (double)aold[off + n];
        x += (1.0 - tn) * (double)logf(kappa[terms[off + n]] + TMVB_EPS_F);                    // Elogpw :89, background part
        if (tn > 0.0 && tn < 1.0) x -= tn * log(tn) + (1.0 - tn) * log(1.0 - tn);             // -Elogqc :101-104
        acc += c * x;
        Cd += c; ta += tn * c;
    }
    Cd = wave_sum_d(Cd); ta = wave_sum_d(ta);
    if (lane == 0 && N > 0) acc += (double)pdot[d];
    // Elogpeta :69
    double mv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) mv[s] = 0.0;
    for (int j = 0; j < K; ++j) {
        double dj = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) if ((j >> 6) == s) dj = readlane_d(df[s], j & 63);
#pragma unroll
        for (int s = 0; s < NS; ++s) mv[s] = fma(on[s] ? invsigma_d[(int64_t)j * K + ix[s]] : 0.0, dj, mv[s]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (!on[s]) continue;
        acc += -0.5 * (invsigma_d[(int64_t)ix[s] * K + ix[s]] * v[s] + df[s] * mv[s]);
        acc -= Cd * exp(l[s] + 0.5 * v[s] - lz);                                               // Elogpz :82 (second part)
        acc += 0.5 * log(v[s]);                                                                // -Elogqeta :95-97
    }
    double tot = wave_sum_d(acc);
    const double TWO_PI_LOG = 1.8378770664093453;
    tot += 0.5 * (*logdet_inv - (double)K * TWO_PI_LOG);
    tot -= Cd * (lz - 1.0);
    tot += 0.5 * (double)K * (1.0 + TWO_PI_LOG);
    tot += log(TMVB_EPS_D + pow(eta, ta) * pow(1.0 - eta, Cd - ta));                           // Elogpc :74-77
    if (lane == 0) doc_val[d] = tot;
}

struct tmvb_fctm {
    tmvb_ctm* base = nullptr;
    int64_t nnz = 0;
    double eta = 0.5;                     // update_eta! is commented out in the reference's train! (src/fCTM.jl:253): a parameter
    float* d_L = nullptr; float* d_kappa = nullptr; float* d_kappa_old = nullptr;
    float* d_tau = nullptr; float* d_tau_old = nullptr; float* d_lse = nullptr;
    float* d_aold = nullptr;              // [nnz] decomposed update_elbo!: the per-token exponent of the last sweep (CtmParams::aold)
    int parts_env = 1;                    // TMVB_FCTM_ELBO_PARTS at creation: 0 never, 1 the iterations train! checks, 2 every E-step collects
    bool want_parts = false, force_walk = false; int elbo_form = 0;      // as tmvb_flda
    float* d_stats = nullptr;             // the base handle is bound to this buffer
    double* d_eta_scratch = nullptr; double* d_ksum = nullptr;
    int64_t stats_len() const { return base->stats_len() + base->V; }
    float* kstat() const { return d_stats + (size_t)base->stats_len(); }
};

extern "C" int tmvb_fctm_destroy(tmvb_fctm* h)
{
    if (!h) return TMVB_OK;
    if (h->base && h->base->ctx) { (void)hipSetDevice(h->base->ctx->device); (void)hipStreamSynchronize(h->base->ctx->stream); }
    (void)tmvb_ctm_destroy(h->base);      // does not free the bound statistics buffer
    (void)hipFree(h->d_aold); (void)hipFree(h->d_L); (void)hipFree(h->d_kappa); (void)hipFree(h->d_kappa_old); (void)hipFree(h->d_tau); (void)hipFree(h->d_tau_old);
    (void)hipFree(h->d_lse); (void)hipFree(h->d_stats); (void)hipFree(h->d_eta_scratch); (void)hipFree(h->d_ksum);
    delete h;
    return TMVB_OK;
}

static int fctm_refresh_L(tmvb_fctm* h)
{
    tmvb_ctm* b = h->base;
    const int64_t total = (int64_t)b->KP * b->V;
    const int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (total + 255) / 256));
    hipLaunchKernelGGL(flda_logbeta_kernel, dim3(nb), dim3(256), 0, b->ctx->stream, b->d_beta[b->cur], h->d_L, b->K, b->KP, b->V);
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

extern "C" int tmvb_fctm_set_state(tmvb_fctm* h, const double* eta, const double* mu, const double* sigma, const double* invsigma,
                                   const double* kappa, const double* kappa_old, const double* beta, const double* beta_old,
                                   const double* lambda, const double* lambda_old, const double* vsq, const double* logzeta,
                                   const double* tau, const double* tau_old, const double* elbo);

extern "C" int tmvb_fctm_create(tmvb_ctx* ctx, tmvb_corpus* corp, int32_t K, tmvb_fctm** out)
{
    TMVB_REQUIRE(out != nullptr, TMVB_EINVAL, "tmvb_fctm_create: out is NULL");
    *out = nullptr;
    tmvb_fctm* h = new tmvb_fctm();
    tmvb_create_guard<tmvb_fctm, tmvb_fctm_destroy> guard{h};
    int rc = tmvb_ctm_create(ctx, corp, K, &h->base);                     // fCTM(corp, K), src/fCTM.jl:32-65: same constructor state
    if (rc) return rc;
    tmvb_ctm* b = h->base;
    h->nnz = corp->info.nnz;
    const size_t V = (size_t)b->V, NZ = (size_t)h->nnz, KPV = (size_t)b->KP * b->V + 4;
    if ((rc = dmalloc(&h->d_L, KPV)) || (rc = dmalloc(&h->d_kappa, V)) || (rc = dmalloc(&h->d_kappa_old, V)) || (rc = dmalloc(&h->d_tau, NZ)) ||
        (rc = dmalloc(&h->d_tau_old, NZ)) || (rc = dmalloc(&h->d_lse, NZ)) || (rc = dmalloc(&h->d_stats, (size_t)h->stats_len())) ||
        (rc = dmalloc(&h->d_aold, std::max<size_t>(NZ, 1))) || (rc = dmalloc(&h->d_eta_scratch, 1)) || (rc = dmalloc(&h->d_ksum, 1)))
        return rc;
    { const char* e = getenv("TMVB_FCTM_ELBO_PARTS"); h->parts_env = e ? atoi(e) : 1; }
    TMVB_HIP(hipMemsetAsync(h->d_stats, 0, (size_t)h->stats_len() * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_lse, 0, std::max<size_t>(NZ, 1) * sizeof(float), ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    if ((rc = tmvb_ctm_bind_stats(b, h->d_stats, h->stats_len()))) return rc;
    const double eta0 = 0.5;                                              // :37
    std::vector<double> kap(V, V ? 1.0 / (double)V : 0.0), tau(NZ, eta0);
    if ((rc = tmvb_fctm_set_state(h, &eta0, nullptr, nullptr, nullptr, kap.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  tau.data(), nullptr, nullptr)))
        return rc;
    if ((rc = fctm_refresh_L(h))) return rc;
    guard.release();
    *out = h;
    return TMVB_OK;
}

extern "C" int tmvb_fctm_set_state(tmvb_fctm* h, const double* eta, const double* mu, const double* sigma, const double* invsigma,
                                   const double* kappa, const double* kappa_old, const double* beta, const double* beta_old,
                                   const double* lambda, const double* lambda_old, const double* vsq, const double* logzeta,
                                   const double* tau, const double* tau_old, const double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_set_state: handle is NULL");
    tmvb_ctm* b = h->base;
    tmvb_ctx* ctx = b->ctx;
    int rc = tmvb_ctm_set_state(b, mu, sigma, invsigma, beta, beta_old, lambda, lambda_old, vsq, logzeta, elbo);
    if (rc) return rc;
    b->filt_parts = false; b->filt_pw_valid = false;       // a state set by the host: update_elbo! takes the token walk
    const size_t V = (size_t)b->V, NZ = (size_t)h->nnz;
    if (eta) {
        TMVB_REQUIRE(*eta >= 0.0 && *eta <= 1.0, TMVB_ESHAPE, "eta must belong to the interval [0,1].");    // src/modelutils.jl:145
        h->eta = *eta;
    }
    if (kappa) {
        if ((rc = upload_f32(ctx, h->d_kappa, kappa, V))) return rc;
        if (!kappa_old && (rc = upload_f32(ctx, h->d_kappa_old, kappa, V))) return rc;
    }
    if (kappa_old && (rc = upload_f32(ctx, h->d_kappa_old, kappa_old, V))) return rc;
    if (tau) {
        for (size_t q = 0; q < NZ; ++q) TMVB_REQUIRE(tau[q] >= 0.0 && tau[q] <= 1.0, TMVB_ESHAPE, "tau must belong to the interval [0,1].");
        if ((rc = upload_f32(ctx, h->d_tau, tau, NZ))) return rc;
        if (!tau_old && (rc = upload_f32(ctx, h->d_tau_old, tau, NZ))) return rc;
    }
    if (tau_old && (rc = upload_f32(ctx, h->d_tau_old, tau_old, NZ))) return rc;
    if (beta && (rc = fctm_refresh_L(h))) return rc;
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_fctm_get_state(tmvb_fctm* h, double* eta, double* mu, double* sigma, double* invsigma, double* kappa, double* kappa_old,
                                   double* beta, double* beta_old, double* lambda, double* lambda_old, double* vsq, double* logzeta,
                                   double* tau, double* tau_old, double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_get_state: handle is NULL");
    tmvb_ctm* b = h->base;
    int rc = tmvb_ctm_get_state(b, mu, sigma, invsigma, beta, beta_old, lambda, lambda_old, vsq, logzeta, elbo);
    if (rc) return rc;
    const size_t V = (size_t)b->V, NZ = (size_t)h->nnz;
    if (eta) *eta = h->eta;
    if (kappa && (rc = download_f32(b->ctx, kappa, h->d_kappa, V))) return rc;
    if (kappa_old && (rc = download_f32(b->ctx, kappa_old, h->d_kappa_old, V))) return rc;
    if (tau && (rc = download_f32(b->ctx, tau, h->d_tau, NZ))) return rc;
    if (tau_old && (rc = download_f32(b->ctx, tau_old, h->d_tau_old, NZ))) return rc;
    return TMVB_OK;
}

// update_phi! / update_tau! / update_logzeta! / update_lambda! / update_vsq! sweeps + update_beta!(model, d) + update_kappa!(model, d)
// for every document (src/fCTM.jl:233-248).  Asynchronous on the context's stream.
extern "C" int tmvb_fctm_estep(tmvb_fctm* h, int32_t niter, double ntol, int32_t viter, double vtol)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_estep: handle is NULL");
    TMVB_REQUIRE(viter >= 0 && niter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative.");   // src/fCTM.jl:229
    TMVB_REQUIRE(vtol >= 0 && ntol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");     // :228
    tmvb_ctm* b = h->base;
    tmvb_ctx* ctx = b->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    CtmParams p;
    p.K = b->K; p.KP = b->KP; p.LPR = b->KP / 4; p.lpr_magic = (unsigned)(0x100000000ull / (unsigned)p.LPR) + 1u;
    p.doc_ptr = b->corp->d_doc_ptr; p.terms = b->corp->d_terms; p.counts = b->corp->d_counts;
    p.doc_order = b->d_doc_order; p.tok_inv = b->corp->term_index.d_inv;
    p.beta = b->d_beta[b->cur]; p.invsigma = b->d_invsigma_f; p.mu = b->d_mu_f;
    p.lambda = b->d_lambda; p.lambda_old = b->d_lambda_old; p.vsq = b->d_vsq; p.logzeta = b->d_logzeta;
    p.wtok = b->d_wtok; p.E = b->d_E; p.estride = (b->KP / 4 <= 64) ? b->KP : b->K; p.sweeps = b->d_sweeps; p.newton_steps = b->d_newton;
    p.niter = niter; p.ntol = ntol; p.viter = viter; p.vtol = vtol; p.debug = 0; p.store_w = 0;
    p.L = h->d_L; p.kappa = h->d_kappa; p.eta = (float)h->eta; p.tau = h->d_tau; p.tau_old = h->d_tau_old; p.lse = h->d_lse;
    // decomposed update_elbo! (fctm_elbo_doc_parts_kernel): an iteration that will be checked stores the per-token exponent and the per-document
    // sum_i (phi counts)_i (lambda_i - lambda_old_i); viter = 0 leaves no responsibilities (token walk)
    const bool collect = (h->parts_env == 2 || (h->parts_env != 0 && h->want_parts)) && viter > 0;
    b->filt_parts = false; b->filt_pw_valid = false;
    if (collect) { p.aold = h->d_aold; p.pdot = b->d_pdot; }
    TMVB_HIP(hipEventRecord(b->ev0, ctx->stream));
    TMVB_HIP(hipMemsetAsync(b->d_newton, 0, sizeof(unsigned long long), ctx->stream));
    if (b->generic && b->M > 0) {
        int lrc = ctm_launch_generic<true>(b, p, ntol);
        if (lrc) return lrc;
    } else {
        // lane-per-document kernel for all but the long documents (b->n_long; the buckets then hold only those), else every bucket
        if (b->batch && b->M > 0) { int brc = ctm_launch_batch<true>(b, p, ntol); if (brc) return brc; }
        for (const tmvb_bucket& bk : b->buckets) {
            const size_t lds = ctm_tile_bytes(bk.tile_rows, b->KP, true);
            const dim3 grid((unsigned)bk.count), block(64);
#define FCTM_CASE(KPV) case KPV: hipLaunchKernelGGL((ctm_estep_kernel<KPV, true>), grid, block, lds, ctx->stream, p, bk.first, bk.tile_rows); break;
            switch (b->KP) { FCTM_CASE(4) FCTM_CASE(12) FCTM_CASE(20) FCTM_CASE(28) FCTM_CASE(36) FCTM_CASE(44)
                             default: hipLaunchKernelGGL((ctm_estep_kernel<52, true>), grid, block, lds, ctx->stream, p, bk.first, bk.tile_rows); break; }
#undef FCTM_CASE
            TMVB_HIP(hipGetLastError());
        }
    }
    FldaStatsParams sp;
    sp.K = b->K; sp.KP = b->KP; sp.L = h->d_L; sp.elog_old = b->d_lambda_old; sp.tau = h->d_tau; sp.tau_old = h->d_tau_old; sp.lse = h->d_lse;
    sp.S = h->d_stats; sp.kstat = h->kstat(); sp.partial = b->d_ts_partial;
    int rc = tmvb_launch_filtered_stats(ctx, b->nslot, b->corp->term_index, sp);
    if (rc) return rc;
    TMVB_HIP(hipEventRecord(b->ev1, ctx->stream));
    b->timed = true;
    b->filt_parts = collect;
    return TMVB_OK;
}

extern "C" int tmvb_fctm_reduce_docs(tmvb_fctm* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_reduce_docs: handle is NULL");
    return tmvb_ctm_reduce_docs(h->base);
}

// update_beta!(model) (src/fCTM.jl:148-152) and update_kappa!(model) (:134-138), run back to back by train! (:249-250)
extern "C" int tmvb_fctm_update_beta(tmvb_fctm* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_update_beta: handle is NULL");
    tmvb_ctm* b = h->base;
    int rc = tmvb_ctm_update_beta(b);
    if (!rc) rc = fctm_refresh_L(h);
    if (rc) return rc;
    hipLaunchKernelGGL(flda_kappa_eta_kernel, dim3(1), dim3(1024), 0, b->ctx->stream, h->kstat(), h->d_kappa, h->d_kappa_old, b->V, 1.0,
                       h->d_eta_scratch, 1, 0, h->d_ksum);
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

extern "C" int tmvb_fctm_update_sigma(tmvb_fctm* h) { TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_update_sigma: handle is NULL"); return tmvb_ctm_update_sigma(h->base); }
extern "C" int tmvb_fctm_update_mu(tmvb_fctm* h) { TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_update_mu: handle is NULL"); return tmvb_ctm_update_mu(h->base); }

extern "C" int tmvb_fctm_update_elbo(tmvb_fctm* h, double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_update_elbo: handle is NULL");
    tmvb_ctm* b = h->base;
    tmvb_ctx* ctx = b->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const bool parts = b->filt_pw_valid && !h->force_walk && h->parts_env != 0;
    h->elbo_form = parts ? 1 : 0;
    if (parts) {
        if (b->M > 0) {
            const dim3 grid((unsigned)b->M), block(64);
#define FCTM_PARTS(NSV) hipLaunchKernelGGL((fctm_elbo_doc_parts_kernel<NSV>), grid, block, 0, ctx->stream, b->K, b->corp->d_doc_ptr, b->corp->d_terms, b->corp->d_counts, \
                               b->d_mu, b->d_invsigma, b->d_logdet, h->eta, h->d_kappa, b->d_lambda, b->d_vsq, b->d_logzeta, h->d_tau, h->d_tau_old, h->d_lse, \
                               h->d_aold, b->d_pdot, b->d_doc_val)
            if (b->nslot == 1) FCTM_PARTS(1); else if (b->nslot == 2) FCTM_PARTS(2); else FCTM_PARTS(4);
#undef FCTM_PARTS
            TMVB_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(ctm_elbo_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, b->d_doc_val, b->M, (const double*)nullptr, (int64_t)0, b->d_pw_partial, b->pw_blocks,
                           b->distributed && b->M_total > 0 ? (double)b->M / (double)b->M_total : 1.0, b->d_elbo);
        TMVB_HIP(hipGetLastError());
    } else {
    if (b->M > 0) {
        const dim3 grid((unsigned)b->M), block(64);
        if (b->nslot == 1)
            hipLaunchKernelGGL((fctm_elbo_kernel<1>), grid, block, 0, ctx->stream, b->K, b->KP, b->corp->d_doc_ptr, b->corp->d_terms, b->corp->d_counts,
                               b->d_mu, b->d_invsigma, b->d_logdet, h->eta, h->d_kappa, b->d_beta[b->cur], b->d_beta[b->cur ^ 1], b->d_lambda,
                               b->d_lambda_old, b->d_vsq, b->d_logzeta, h->d_tau, h->d_tau_old, b->d_doc_val);
        else if (b->nslot == 2)
            hipLaunchKernelGGL((fctm_elbo_kernel<2>), grid, block, 0, ctx->stream, b->K, b->KP, b->corp->d_doc_ptr, b->corp->d_terms, b->corp->d_counts,
                               b->d_mu, b->d_invsigma, b->d_logdet, h->eta, h->d_kappa, b->d_beta[b->cur], b->d_beta[b->cur ^ 1], b->d_lambda,
                               b->d_lambda_old, b->d_vsq, b->d_logzeta, h->d_tau, h->d_tau_old, b->d_doc_val);
        else
            hipLaunchKernelGGL((fctm_elbo_kernel<4>), grid, block, 0, ctx->stream, b->K, b->KP, b->corp->d_doc_ptr, b->corp->d_terms, b->corp->d_counts,
                               b->d_mu, b->d_invsigma, b->d_logdet, h->eta, h->d_kappa, b->d_beta[b->cur], b->d_beta[b->cur ^ 1], b->d_lambda,
                               b->d_lambda_old, b->d_vsq, b->d_logzeta, h->d_tau, h->d_tau_old, b->d_doc_val);
        TMVB_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(sum_docs_kernel, dim3(1), dim3(1024), 0, ctx->stream, b->d_doc_val, b->M, b->d_elbo);
    TMVB_HIP(hipGetLastError());
    }
    double v = 0.0;
    int st = 0;
    TMVB_HIP(hipMemcpyAsync(&v, b->d_elbo, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TMVB_HIP(hipMemcpyAsync(&st, b->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    TMVB_REQUIRE(st == 0, TMVB_ENONFINITE, "sigma must be positive-definite.");
    b->elbo = v;
    if (elbo) *elbo = v;
    return TMVB_OK;
}

extern "C" int tmvb_fctm_elbo_form(tmvb_fctm* h, int32_t* form)
{
    TMVB_REQUIRE(h && form, TMVB_EINVAL, "tmvb_fctm_elbo_form: NULL argument");
    *form = h->elbo_form;
    return TMVB_OK;
}

extern "C" int tmvb_fctm_set_comm(tmvb_fctm* h, tmvb_comm* comm, int64_t M_total)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_set_comm: handle is NULL");
    return tmvb_ctm_set_comm(h->base, comm, M_total);
}

namespace {
struct FctmTrainOps {
    int niter, viter; double ntol, vtol;
    int estep(tmvb_fctm* h) { return tmvb_fctm_estep(h, niter, ntol, viter, vtol); }   // src/fCTM.jl:233-248
    int reduce(tmvb_fctm* h) { return tmvb_fctm_reduce_docs(h); }
    int before_allreduce(tmvb_fctm*) { return TMVB_OK; }
    float* stats(tmvb_fctm* h) { return h->d_stats; }
    int64_t stats_len(tmvb_fctm* h) { return h->stats_len(); }
    int mstep(tmvb_fctm* h)
    {
        int rc = tmvb_fctm_update_beta(h);                                             // :249-250 (beta, kappa)
        if (!rc) rc = tmvb_fctm_update_sigma(h);                                       // :251 (previous mu)
        if (!rc) rc = tmvb_fctm_update_mu(h);                                          // :252;  update_eta! is commented out (:253)
        return rc;
    }
    int elbo_local(tmvb_fctm* h, double* s, double* once) { *once = 0.0; return tmvb_fctm_update_elbo(h, s); }
    int elbo_form(tmvb_fctm* h) { return h->elbo_form; }
    void force_walk(tmvb_fctm* h, bool on, bool doubled = true) { h->force_walk = on; if (!on && doubled) h->elbo_form = 1; }   // (as LdaTrainOps)
    void will_check(tmvb_fctm* h, bool checked) { h->want_parts = checked; }
    double* elbo_dev(tmvb_fctm* h) { return h->base->d_elbo; }
    tmvb_comm* comm(tmvb_fctm* h) { return h->base->comm; }
    bool distributed(tmvb_fctm* h) { return h->base->distributed; }
    tmvb_ctx* ctx(tmvb_fctm* h) { return h->base->ctx; }
    int64_t nnz(tmvb_fctm* h) { return h->nnz; }
    void set_elbo(tmvb_fctm* h, double v) { h->base->elbo = v; }
    double get_elbo(tmvb_fctm* h) { return h->base->elbo; }
    int finish(tmvb_fctm* h)
    {
        TMVB_HIP(hipSetDevice(h->base->ctx->device));
        TMVB_HIP(hipStreamSynchronize(h->base->ctx->stream));
        return TMVB_OK;
    }
};
}  // namespace

extern "C" int tmvb_fctm_train_group(tmvb_fctm* const* hs, int32_t n, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter,
                                     double vtol, int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline)
{
    TMVB_REQUIRE(tol >= 0 && ntol >= 0 && vtol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");   // src/fCTM.jl:228
    TMVB_REQUIRE(iter >= 0 && niter >= 0 && viter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative."); // :229
    FctmTrainOps ops{niter, viter, ntol, vtol};
    return tmvb_train_group_loop("tmvb_fctm_train", hs, n, iter, tol, checkelbo, elbo_traj, iters_done, elbo_baseline, ops);
}

// train! (src/fCTM.jl:226-262)
extern "C" int tmvb_fctm_train(tmvb_fctm* h, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter, double vtol,
                               int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_train: handle is NULL");
    return tmvb_fctm_train_group(&h, 1, iter, tol, niter, ntol, viter, vtol, checkelbo, elbo_traj, iters_done, elbo_baseline);
}

extern "C" int tmvb_fctm_doc_sweeps(tmvb_fctm* h, uint8_t* out)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_doc_sweeps: handle is NULL");
    return tmvb_ctm_doc_sweeps(h->base, out);
}

extern "C" int tmvb_fctm_sweep_hist(tmvb_fctm* h, int64_t* hist, int32_t nbins, int64_t* newton_steps)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_fctm_sweep_hist: handle is NULL");
    return tmvb_ctm_sweep_hist(h->base, hist, nbins, newton_steps);
}
