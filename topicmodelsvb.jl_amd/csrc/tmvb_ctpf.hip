// tmvb_ctpf.hip -- collaborative topic Poisson factorization (CTPF) engine for gfx950 (MI355X).
//
// Path: the per-document coordinate ascent of src/CTPF.jl:353-365 (update_xi! :334, update_phi! :327,
// update_zayin! :318, update_gimel! :309, exit test :359, update_he!(d) :274, update_alef!(d) :259) fused
// into ONE kernel per document, and the global updates in the reference's order (:366-371: he, alef,
// dalet, het, bet, vav).  It replaces the twelve OpenCL kernels of src/gpuCTPF.jl:288-661 but follows
// the CPU path (per-document exit rule; `log vav` in xi, src/CTPF.jl:336, where the OpenCL kernel uses
// `log bet`, src/gpuCTPF.jl:624).
//
// Both softmaxes factor through per-iteration tables (the digammas of alef / he do not change inside
// an outer iteration):  TA[i,j] = exp(psi(alef[i,j])),  TH[i,u] = exp(psi(he[i,u])).
//   phi[i,n]      = TA[i,t_n] e_i / s_n,           e_i  = exp(psi(gimel_i) - log dalet_i - log bet_i - max)
//   xi_top[i,u]   = TH[i,r_u] ea_i / s'_u,         ea_i = exp(psi(gimel_i) - log dalet_i - log vav_i - max')
//   xi_bot[i,u]   = TH[i,r_u] eb_i / s'_u,         eb_i = exp(psi(zayin_i) - log het_i   - log vav_i - max')
// with s_n = sum_i TA[i,t_n] e_i and s'_u = sum_i TH[i,r_u] (ea_i + eb_i): the same two matrix-vector
// products through an LDS tile as the LDA kernel, once over the document's term rows and once over
// its reader rows; 2 fp32 digammas per topic per sweep instead of K (N_d + 2 R_d).
// The scatters update_alef!(d) / update_he!(d) are gather-side statistics passes over the term and
// the reader inverted indices (tmvb_termstats.h); HBM-bound like LDA.
#define TMVB_TS_LOGZ 1            // the log-normaliser forms of the statistics pass (decomposed update_elbo!, ctpf_elbo_doc_parts_kernel)
#include "tmvb_common_kernels.h"

// xi's rate is `log vav` (src/CTPF.jl:336).  -DTMVB_MUTANT_CTPF_LOG_BET (tests/test_mutants_gpu.py, never in a shipped build) stores log bet in its place --
// the reference's own OpenCL bug (src/gpuCTPF.jl:624) -- to prove that the parity tests notice.
#ifdef TMVB_MUTANT_CTPF_LOG_BET
#define TMVB_CTPF_XI_RATE(bet, vav) (bet)
#else
#define TMVB_CTPF_XI_RATE(bet, vav) (vav)
#endif
#include "tmvb_train.h"
#include "tmvb_regtile.h"
#include "tmvb_gridtile.h"

struct CtpfParams {
    int K, KP, LPR;
    unsigned lpr_magic;
    const int64_t* doc_ptr;
    const int32_t* terms;
    const int32_t* counts;
    const int64_t* rdr_ptr;
    const int32_t* readers;
    const int32_t* ratings;
    const int32_t* doc_order;
    const int32_t* tok_inv;     // term-major position of each CSR token
    const int32_t* rdr_inv;     // reader-major position of each CSR reader entry
    const float* TA;            // [V][KP] exp(psi(alef)), pads zero
    const float* TH;            // [U][KP] exp(psi(he)), pads zero
    const float* lrates;        // [4][K]: log bet, log vav, log dalet, log het
    float hc, hg;               // hyper-parameters c, g (src/CTPF.jl:81)
    float* gimel; float* gimel_old; float* zayin; float* zayin_old;   // [M][K]
    float* wtok;                // [nnz] term-major   c_n / s_n
    float* wrdr;                // [nR]  reader-major rating_u / s'_u
    float* E1;                  // [M][estride] e      (phi factor)
    float* E2;                  // [M][estride] ea+eb  (xi_top + xi_bot factor)
    int estride;                // row stride of E1 / E2: KP rounded up to 64 floats for 32 < KP <= 64 (zero padded: the statistics pass's
                                // fast form, tmvb_termstats.h), KP otherwise
    uint8_t* sweeps;
    int viter;
    float vtol;
    int store_w;                // KP > 128: the statistics passes read stored weights instead of recomputing them
    float* shift = nullptr;     // [M][2] or NULL: the softmax shifts (max of the phi arguments, max of the xi arguments) behind E1 / E2 of the last executed
                                // sweep -- the decomposed update_elbo! needs log of the UNSHIFTED normalisers (ctpf_elbo_doc_parts_kernel)
};

// NS = topic slots per lane (lane l owns topics l, l + 64, ...): NS = 1 for K <= 64, NS = 2 for K <= 128.
template <int LPR_T, int NS>
__global__ __launch_bounds__(64) void ctpf_estep_kernel(CtpfParams p, int64_t first, int tile_rows)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int K = p.K;
    const int LPR = LPR_T ? LPR_T : p.LPR;
    const int KP = 4 * LPR;
    float* Bt = lds;                           // [tile_rows][KP] term rows then reader rows
    float* e_l = Bt + (size_t)tile_rows * KP;  // [KP] phi factor e
    float* f_l = e_l + KP;                     // [KP] xi factor ea + eb
    float* w_l = f_l + KP;                     // [tile_rows]
    float* c_l = w_l + tile_rows;              // [tile_rows] counts / ratings as float
    int* t_l = (int*)(c_l + tile_rows);        // [tile_rows] term / reader ids

    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d], roff = p.rdr_ptr[d];
    const int N = (int)(p.doc_ptr[d + 1] - off), Rd = (int)(p.rdr_ptr[d + 1] - roff);
    const bool single = (N + Rd) <= tile_rows;     // both row sets stay resident in the tile

    bool on[NS];
    float lb[NS], lv[NS], ld[NS], lh[NS], gim[NS], zay[NS], gim_old[NS], zay_old[NS], e[NS], ea[NS], eb[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = lane + 64 * s;
        on[s] = i < K;
        lb[s] = on[s] ? p.lrates[i] : 0.f; lv[s] = on[s] ? p.lrates[K + i] : 0.f;
        ld[s] = on[s] ? p.lrates[2 * K + i] : 0.f; lh[s] = on[s] ? p.lrates[3 * K + i] : 0.f;
        gim[s] = on[s] ? p.gimel[(int64_t)d * K + i] : 1.0f; zay[s] = on[s] ? p.zayin[(int64_t)d * K + i] : 1.0f;
        gim_old[s] = gim[s]; zay_old[s] = zay[s]; e[s] = 0.f; ea[s] = 0.f; eb[s] = 0.f;
    }

    // rows [r0, r0+rows) of the tile <- table rows of ids[c0..c0+rows)
    auto load_rows = [&](const float* table, const int32_t* ids, const int32_t* vals, int64_t base, int c0, int rows, int r0) {
        for (int n = lane; n < rows; n += 64) {
            t_l[r0 + n] = ids[base + c0 + n];
            c_l[r0 + n] = (float)vals[base + c0 + n];
        }
        WAVE_LDS_FENCE();
        const int nch = rows * LPR;
#pragma unroll 4
        for (int f0 = 0; f0 < nch; f0 += 64) {
            const int f = f0 + lane;
            if (f < nch) {
                const int n = (LPR == 1) ? f : (int)__umulhi((unsigned)f, p.lpr_magic);
                const int c = f - n * LPR;
                const float* src = table + ((int64_t)t_l[r0 + n] * KP + 4 * c);
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(Bt + ((size_t)r0 * KP + (size_t)f0 * 4)), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WAVE_LDS_FENCE();
    };
    // w[r0+n] = val_n / (row_n . fac)
    auto phase1 = [&](const float* fac, int rows, int r0) {
        const float4* er = (const float4*)fac;
        for (int n = lane; n < rows; n += 64) {
            const float4* br = (const float4*)(Bt + (size_t)(r0 + n) * KP);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
            for (int q = 0; q < LPR; ++q) {
                float4 b = br[q], ev = er[q];
                s0 = fmaf(b.x, ev.x, s0); s1 = fmaf(b.y, ev.y, s1);
                s2 = fmaf(b.z, ev.z, s2); s3 = fmaf(b.w, ev.w, s3);
            }
            w_l[r0 + n] = c_l[r0 + n] / ((s0 + s1) + (s2 + s3));
        }
        WAVE_LDS_FENCE();
    };
    const int r4 = lane & 3, ql = lane >> 2;
    auto phase2 = [&](int rows, int r0, float4 (&acc)[NS]) {
        const int nfull = rows >> 2;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int q = ql + 16 * s;                 // topic quad of slot s: topics 4 q .. 4 q + 3
            if (q >= LPR) continue;
#pragma unroll 4
            for (int m = 0; m < nfull; ++m) {
                const int n = r0 + 4 * m + r4;
                const float w = w_l[n];
                const float4 b = *(const float4*)(Bt + (size_t)n * KP + 4 * q);
                acc[s].x = fmaf(w, b.x, acc[s].x); acc[s].y = fmaf(w, b.y, acc[s].y); acc[s].z = fmaf(w, b.z, acc[s].z); acc[s].w = fmaf(w, b.w, acc[s].w);
            }
            const int n = 4 * nfull + r4;
            if (n < rows) {
                const float w = w_l[r0 + n];
                const float4 b = *(const float4*)(Bt + (size_t)(r0 + n) * KP + 4 * q);
                acc[s].x = fmaf(w, b.x, acc[s].x); acc[s].y = fmaf(w, b.y, acc[s].y); acc[s].z = fmaf(w, b.z, acc[s].z); acc[s].w = fmaf(w, b.w, acc[s].w);
            }
        }
    };
    auto quad_select = [&](float4 a) -> float {      // lane l ends with the sum for topic 4 (l >> 2) + (l & 3) = l of its slot
        a = dpp_add4<0xB1>(a);
        a = dpp_add4<0x4E>(a);
        return (r4 == 0) ? a.x : (r4 == 1) ? a.y : (r4 == 2) ? a.z : a.w;
    };
    // KP > 128 only: per-token / per-reader weights of the last executed sweep for the stored-weight statistics kernels
    auto store_w = [&](float* dst, const int32_t* inv, int64_t base, int c0, int rows, int r0) {
        if (!p.store_w) return;
        for (int n = lane; n < rows; n += 64) dst[inv[base + c0 + n]] = w_l[r0 + n];
    };

    int sweeps = 0;
    float mx_last = 0.f, mab_last = 0.f;                 // CtpfParams::shift
    for (int v = 0; v < p.viter; ++v) {
        ++sweeps;
        float x[NS], a[NS], b[NS], mxl = -INFINITY, mabl = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float dg = on[s] ? digamma_f(gim[s]) : 0.f, dz = on[s] ? digamma_f(zay[s]) : 0.f;
            x[s] = dg - ld[s] - lb[s];                                  // update_phi!  src/CTPF.jl:329
            a[s] = dg - ld[s] - lv[s]; b[s] = dz - lh[s] - lv[s];       // update_xi!  :336
            if (on[s]) { mxl = fmaxf(mxl, x[s]); mabl = fmaxf(mabl, fmaxf(a[s], b[s])); }
        }
        const float mx = wave_max(mxl);
        const float mab = wave_max(mabl);
        mx_last = mx; mab_last = mab;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            e[s] = on[s] ? expf(x[s] - mx) : 0.f;
            ea[s] = on[s] ? expf(a[s] - mab) : 0.f;
            eb[s] = on[s] ? expf(b[s] - mab) : 0.f;
            if (on[s]) { e_l[lane + 64 * s] = e[s]; f_l[lane + 64 * s] = ea[s] + eb[s]; }
        }
        for (int i = K + lane; i < KP; i += 64) { e_l[i] = 0.f; f_l[i] = 0.f; }           // pads
        WAVE_LDS_FENCE();
        float4 accG[NS], accH[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { accG[s] = make_float4(0.f, 0.f, 0.f, 0.f); accH[s] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (single) {
            if (v == 0) {
                if (N) load_rows(p.TA, p.terms, p.counts, off, 0, N, 0);
                if (Rd) load_rows(p.TH, p.readers, p.ratings, roff, 0, Rd, N);
            }
            phase1(e_l, N, 0);
            phase1(f_l, Rd, N);
            phase2(N, 0, accG);
            phase2(Rd, N, accH);
        } else {
            for (int c0 = 0; c0 < N; c0 += tile_rows) {
                const int rows = min(tile_rows, N - c0);
                load_rows(p.TA, p.terms, p.counts, off, c0, rows, 0);
                phase1(e_l, rows, 0);
                store_w(p.wtok, p.tok_inv, off, c0, rows, 0);
                phase2(rows, 0, accG);
                WAVE_LDS_FENCE();
            }
            for (int c0 = 0; c0 < Rd; c0 += tile_rows) {
                const int rows = min(tile_rows, Rd - c0);
                load_rows(p.TH, p.readers, p.ratings, roff, c0, rows, 0);
                phase1(f_l, rows, 0);
                store_w(p.wrdr, p.rdr_inv, roff, c0, rows, 0);
                phase2(rows, 0, accH);
                WAVE_LDS_FENCE();
            }
        }
        float dl = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float G = quad_select(accG[s]), Hh = quad_select(accH[s]);
            zay_old[s] = zay[s]; gim_old[s] = gim[s];
            if (on[s]) {
                zay[s] = p.hg + eb[s] * Hh;                       // update_zayin!  :322
                gim[s] = (p.hc + e[s] * G) + ea[s] * Hh;          // update_gimel!  :313
                const float df = gim[s] - gim_old[s];
                dl = fmaf(df, df, dl);
            }
        }
        if (__builtin_amdgcn_sqrtf(wave_sum(dl)) < p.vtol) break;       // :359
    }

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = lane + 64 * s;
        if (sweeps > 0) {
            if (on[s]) {
                p.gimel[(int64_t)d * K + i] = gim[s]; p.gimel_old[(int64_t)d * K + i] = gim_old[s];
                p.zayin[(int64_t)d * K + i] = zay[s]; p.zayin_old[(int64_t)d * K + i] = zay_old[s];
            }
            if (on[s]) { p.E1[(int64_t)d * p.estride + i] = e[s]; p.E2[(int64_t)d * p.estride + i] = ea[s] + eb[s]; }
        } else {
            if (on[s]) { p.E1[(int64_t)d * p.estride + i] = 0.f; p.E2[(int64_t)d * p.estride + i] = 0.f; }   // viter = 0
        }
    }
    if (p.shift && lane == 0) { p.shift[2 * (int64_t)d] = mx_last; p.shift[2 * (int64_t)d + 1] = mab_last; }
    if (p.estride >= KP)                                                                        // dense rows (KP / 4 > 64) have no pads
        for (int i = K + lane; i < KP; i += 64) { p.E1[(int64_t)d * p.estride + i] = 0.f; p.E2[(int64_t)d * p.estride + i] = 0.f; }   // pads
    if (sweeps > 0 && single) {
        store_w(p.wtok, p.tok_inv, off, 0, N, 0);
        store_w(p.wrdr, p.rdr_inv, roff, 0, Rd, N);
    } else if (sweeps == 0 && p.store_w) {
        for (int n = lane; n < N; n += 64) p.wtok[p.tok_inv[off + n]] = 0.f;
        for (int n = lane; n < Rd; n += 64) p.wrdr[p.rdr_inv[roff + n]] = 0.f;
    }
    if (lane == 0) p.sweeps[d] = (uint8_t)min(sweeps, 255);
}

// ------------------------------------------------------------------------------ register-tile E-step
// Documents with at most 64 T unique terms and at most 64 readers: the TA rows of the terms (T tiles) and the TH
// rows of the readers (one tile) live in VGPRs as topic pairs, exactly as in lda_estep_reg_kernel (tmvb_lda.hip):
// per sweep two packed matrix-vector passes per tile set and two cross-lane reduce-scatters
//   s_n = sum_i TA[n][i] e_i,  w_n = c_n / s_n,  G_i = sum_n w_n TA[n][i]          (update_phi!,  src/CTPF.jl:327-330)
//   s_u = sum_i TH[u][i] f_i,  w_u = r_u / s_u,  H_i = sum_u w_u TH[u][i], f = ea + eb  (update_xi!, :334-337)
//   zayin_i = g + eb_i H_i ;  gimel_i = c + e_i G_i + ea_i H_i                      (:318-323, :309-314)
// with e, ea, eb = exp(psi(.) - log rates - max) as in ctpf_estep_kernel.  A lane past the last term / reader
// carries count 0 (its weight is exactly 0), so the tiles are loaded without branches.
template <int LPR, int T>
__device__ __forceinline__ void ctpf_estep_reg_body(const CtpfParams& p, const int d, const int* __restrict__ topic_of_lane,
                                                    float (&ef_lds)[2][4 * LPR])
{
    constexpr int R = 4 * LPR;
    static_assert(R <= 64, "ctpf_estep_reg_kernel: one result slot per lane");
    const int lane = threadIdx.x;
    const int K = p.K;
    const int64_t off = p.doc_ptr[d], roff = p.rdr_ptr[d];
    const int N = (int)(p.doc_ptr[d + 1] - off), Rd = (int)(p.rdr_ptr[d + 1] - roff);

    v2f A2[T][R / 2], H2[1][R / 2];
    float c[T], rr;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int n = lane + 64 * t;
        const bool in = n < N;
        const int term = in ? p.terms[off + n] : 0;
        c[t] = in ? (float)p.counts[off + n] : 0.0f;
        const float4* row = (const float4*)(p.TA + (int64_t)term * R);
#pragma unroll
        for (int q = 0; q < LPR; ++q) {
            const float4 v = row[q];
            A2[t][2 * q] = v2f{v.x, v.y}; A2[t][2 * q + 1] = v2f{v.z, v.w};
        }
    }
    {
        const bool in = lane < Rd;
        const int u = in ? p.readers[roff + lane] : 0;
        rr = in ? (float)p.ratings[roff + lane] : 0.0f;
        if (Rd > 0) {                                   // uniform: a corpus without readers has no TH table
            const float4* row = (const float4*)(p.TH + (int64_t)u * R);
#pragma unroll
            for (int q = 0; q < LPR; ++q) {
                const float4 v = row[q];
                H2[0][2 * q] = v2f{v.x, v.y}; H2[0][2 * q + 1] = v2f{v.z, v.w};
            }
        } else {
#pragma unroll
            for (int q = 0; q < R / 2; ++q) H2[0][q] = v2f{1.0f, 1.0f};
        }
    }
    // topic role: this lane owns topic pi(lane) after the reduce-scatter (-1: duplicate / pad)
    const int mytopic = topic_of_lane[lane];
    const bool on = mytopic >= 0 && mytopic < K;
    const int mt = on ? mytopic : 0;
    const float lb = on ? p.lrates[mt] : 0.f, lv = on ? p.lrates[K + mt] : 0.f;
    const float ld = on ? p.lrates[2 * K + mt] : 0.f, lh = on ? p.lrates[3 * K + mt] : 0.f;
    float gim = on ? p.gimel[(int64_t)d * K + mt] : 1.0f, zay = on ? p.zayin[(int64_t)d * K + mt] : 1.0f;
    float gim_old = gim, zay_old = zay;
    float e[1] = {0.f}, f[1] = {0.f}, ea = 0.f, eb = 0.f;
    float mx_last = 0.f, mab_last = 0.f;                 // CtpfParams::shift

    int sweeps = 0;
    for (int v = 0; v < p.viter; ++v) {
        ++sweeps;
        const float dg = digamma_f(gim), dz = digamma_f(zay);
        const float x = dg - ld - lb;                     // update_phi!  src/CTPF.jl:329
        const float a = dg - ld - lv, b = dz - lh - lv;   // update_xi!  :336
        const float mx = wave_max(on ? x : -INFINITY);
        const float mab = wave_max(on ? fmaxf(a, b) : -INFINITY);
        mx_last = mx; mab_last = mab;
        e[0] = on ? fast_exp(x - mx) : 0.f;
        ea = on ? fast_exp(a - mab) : 0.f;
        eb = on ? fast_exp(b - mab) : 0.f;
        f[0] = ea + eb;
        // The factors e, f = ea + eb reach all lanes through LDS (one ds_write each, uniform-address ds_read_b128) instead of two
        // v_readlane per topic: the broadcast leaves the saturated VALU (113 of the sweep's 844 instructions) for the idle LDS pipe,
        // as in lda_estep_reg_body.  Pad topics carry 0.
        if (mytopic >= 0 && mytopic < R) { ef_lds[0][mytopic] = e[0]; ef_lds[1][mytopic] = f[0]; }
        WAVE_LDS_FENCE();
        // terms
        v2f sacc[T][2];
#pragma unroll
        for (int t = 0; t < T; ++t) { sacc[t][0] = v2f{0.f, 0.f}; sacc[t][1] = v2f{0.f, 0.f}; }
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            const float4 ev = ((const float4*)ef_lds[0])[j];
            const v2f e0 = v2f{ev.x, ev.y}, e1 = v2f{ev.z, ev.w};
#pragma unroll
            for (int t = 0; t < T; ++t) {
                sacc[t][0] = __builtin_elementwise_fma(A2[t][2 * j], e0, sacc[t][0]);
                sacc[t][1] = __builtin_elementwise_fma(A2[t][2 * j + 1], e1, sacc[t][1]);
            }
        }
        float w[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const v2f s2 = sacc[t][0] + sacc[t][1];
            w[t] = fast_div(c[t], s2.x + s2.y);
        }
        float G[1];
        lane_reduce_scatter<R>([&](int q) {
            v2f acc = A2[0][q] * v2f{w[0], w[0]};
#pragma unroll
            for (int t = 1; t < T; ++t) acc = __builtin_elementwise_fma(A2[t][q], v2f{w[t], w[t]}, acc);
            return acc;
        }, G, lane);
        // readers
        v2f hacc[1][2] = {{v2f{0.f, 0.f}, v2f{0.f, 0.f}}};
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            const float4 fv = ((const float4*)ef_lds[1])[j];
            hacc[0][0] = __builtin_elementwise_fma(H2[0][2 * j], v2f{fv.x, fv.y}, hacc[0][0]);
            hacc[0][1] = __builtin_elementwise_fma(H2[0][2 * j + 1], v2f{fv.z, fv.w}, hacc[0][1]);
        }
        const v2f hs = hacc[0][0] + hacc[0][1];
        const float wr = fast_div(rr, hs.x + hs.y);
        float Hh[1];
        lane_reduce_scatter<R>([&](int q) { return H2[0][q] * v2f{wr, wr}; }, Hh, lane);
        zay_old = zay; gim_old = gim;
        float dl = 0.f;
        if (on) {
            zay = p.hg + eb * Hh[0];                      // update_zayin!  :322
            gim = (p.hc + e[0] * G[0]) + ea * Hh[0];      // update_gimel!  :313
            const float df = gim - gim_old;
            dl = df * df;
        }
        if (__builtin_amdgcn_sqrtf(wave_sum(dl)) < p.vtol) break;          // :359
    }
    const bool mine = mytopic >= 0 && mytopic < R;
    if (sweeps > 0) {
        if (on) {
            p.gimel[(int64_t)d * K + mt] = gim; p.gimel_old[(int64_t)d * K + mt] = gim_old;
            p.zayin[(int64_t)d * K + mt] = zay; p.zayin_old[(int64_t)d * K + mt] = zay_old;
        }
        if (mine) { p.E1[(int64_t)d * p.estride + mytopic] = e[0]; p.E2[(int64_t)d * p.estride + mytopic] = f[0]; }
    } else {
        if (mine) { p.E1[(int64_t)d * p.estride + mytopic] = 0.f; p.E2[(int64_t)d * p.estride + mytopic] = 0.f; }   // viter = 0
    }
    if (lane == 0) {
        p.sweeps[d] = (uint8_t)min(sweeps, 255);
        if (p.shift) { p.shift[2 * (int64_t)d] = mx_last; p.shift[2 * (int64_t)d + 1] = mab_last; }
    }
}

template <int LPR, int T>
__global__ __launch_bounds__(64) void ctpf_estep_reg_kernel(CtpfParams p, int64_t first, const int* __restrict__ topic_of_lane)
{
    __shared__ __attribute__((aligned(16))) float ef_lds[2][4 * LPR];
    ctpf_estep_reg_body<LPR, T>(p, p.doc_order[first + blockIdx.x], topic_of_lane, ef_lds);
}
// Both register-tile buckets in one launch, the tile count read per document (wave-uniform): one kernel tail and one kernel
// boundary fewer in a 0.3 ms iteration; both bodies run two waves per SIMD, so the wider allocation costs nothing.
template <int LPR>
__global__ __launch_bounds__(64) void ctpf_estep_reg_any_kernel(CtpfParams p, int64_t first, const int* __restrict__ topic_of_lane)
{
    __shared__ __attribute__((aligned(16))) float ef_lds[2][4 * LPR];
    const int d = p.doc_order[first + blockIdx.x];
    const int N = __builtin_amdgcn_readfirstlane((int)(p.doc_ptr[d + 1] - p.doc_ptr[d]));
    if (N > 64) ctpf_estep_reg_body<LPR, 2>(p, d, topic_of_lane, ef_lds);
    else ctpf_estep_reg_body<LPR, 1>(p, d, topic_of_lane, ef_lds);
}

template <int LPR>
static void ctpf_launch_reg(int tiles, dim3 grid, hipStream_t st, const CtpfParams& p, int64_t first, const int* tol)
{
    if (tiles == 99) hipLaunchKernelGGL((ctpf_estep_reg_any_kernel<LPR>), grid, dim3(64), 0, st, p, first, tol);
    else if (tiles <= 1) hipLaunchKernelGGL((ctpf_estep_reg_kernel<LPR, 1>), grid, dim3(64), 0, st, p, first, tol);
    else hipLaunchKernelGGL((ctpf_estep_reg_kernel<LPR, 2>), grid, dim3(64), 0, st, p, first, tol);
}

// ------------------------------------------------------------------------------ grid-tile E-step (tmvb_gridtile.h)
// The register-tile arithmetic above on the 16 x 4 lane grid of tmvb_gridtile.h: lane (a, b) holds {terms 16 s + a} x {topics
// 4 j + b} of the TA rows (NPT pairs of term slots) and the same of the TH rows of the readers (NPR pairs); per sweep four
// matrix-vector products on packed fmas, two 4-lane all-reduces and two 16-lane reduce-scatters of LPR values instead of two
// 64-lane reduce-scatters of KP values.  The softmax shifts (the maxima of x and of (a, b), src/CTPF.jl:329, :336) cancel in
// phi / xi, so they are taken once, in the document's first sweep, and kept (the factors move by a few units per sweep; fp32
// exp has 80 to spare).  W > 1: W waves share a long document (terms 32 NPT w + ..., readers 32 NPR w + ...), partial sums
// through LDS once per sweep, every wave runs the identical tail.
// BW = waves per workgroup: W (the W waves of one long document); with W = 1 and BW > 1 the waves of a workgroup would be
// independent single-wave documents sharing nothing but the LDS array, sliced by wave (tried as one launch for the whole corpus:
// slower, see tmvb_ctpf_estep).
template <int LPR, int NPT, int NPR, int W = 1, int BW = W>
__device__ __forceinline__ void ctpf_estep_grid_body(const CtpfParams& p, const int d, const int* __restrict__ topic_of_lane,
                                                    float (*ef_all)[2][4][16], float (*xch)[4][128])
{
    constexpr int R = 4 * LPR;
    static_assert(LPR <= 16, "ctpf_estep_grid_body: one result slot per lane");
    static_assert(W == 1 || W == BW, "ctpf_estep_grid_body: a multi-wave document owns its workgroup");
    static_assert(W <= 4, "ctpf_estep_grid_body: the exchange buffer holds four waves");
    const int lane = threadIdx.x & 63;
    const int bwave = (BW > 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;   // LDS slice
    const int wave = (W > 1) ? bwave : 0;                                                       // share of the document
    const int a = lane >> 2, b = lane & 3;
    const int K = p.K;
    const int64_t off = p.doc_ptr[d], roff = p.rdr_ptr[d];
    const int N = (int)(p.doc_ptr[d + 1] - off), Rd = (int)(p.rdr_ptr[d + 1] - roff);
    float (*ef)[4][16] = ef_all[bwave];                 // e / f by class: [b][j], one copy per wave

    gv2f A[NPT][LPR], ct[NPT], H[NPR][LPR], rr[NPR];
    {
        const int64_t off0 = N > 0 ? off : 0;
        grid_load_tile<LPR, NPT>(A, ct, p.TA, p.terms + off0, p.counts + off0, N, 32 * NPT * wave, a, b);
    }
    if (Rd > 0) {                                       // uniform: a corpus without readers has no TH table
        grid_load_tile<LPR, NPR>(H, rr, p.TH, p.readers + roff, p.ratings + roff, Rd, 32 * NPR * wave, a, b);
    } else {
#pragma unroll
        for (int q = 0; q < NPR; ++q) {
            rr[q] = gv2f{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < LPR; ++j) H[q][j] = gv2f{1.0f, 1.0f};
        }
    }
    // topic role: this lane owns topic mytopic after the reduce-scatters (-1: not a primary owner)
    const int mytopic = topic_of_lane[lane];
    const bool on = mytopic >= 0 && mytopic < K;
    const int mt = on ? mytopic : 0;
    const float lb_l = p.lrates[mt], lv_l = p.lrates[K + mt], ld_l = p.lrates[2 * K + mt], lh_l = p.lrates[3 * K + mt];
    const float g_l = p.gimel[(int64_t)d * K + mt], z_l = p.zayin[(int64_t)d * K + mt];
    const float lb = on ? lb_l : 0.f, lv = on ? lv_l : 0.f, ld = on ? ld_l : 0.f, lh = on ? lh_l : 0.f;
    float gim = on ? g_l : 1.0f, zay = on ? z_l : 1.0f;
    float gim_old = gim, zay_old = zay;
    float e = 0.f, f = 0.f, ea = 0.f, eb = 0.f, mx = 0.f, mab = 0.f;
    const float vtol2 = p.vtol * p.vtol;

    int sweeps = 0;
    for (int v = 0; v < p.viter; ++v) {
        ++sweeps;
        const float dg = digamma_sweep_f(gim), dz = digamma_sweep_f(zay);
        const float x = dg - ld - lb;                     // update_phi!  src/CTPF.jl:329
        const float xa = dg - ld - lv, xb = dz - lh - lv; // update_xi!  :336
        if (v == 0) {
            mx = wave_max(on ? x : -INFINITY);
            mab = wave_max(on ? fmaxf(xa, xb) : -INFINITY);
        }
        e = on ? sweep_exp(x - mx) : 0.f;
        ea = on ? sweep_exp(xa - mab) : 0.f;
        eb = on ? sweep_exp(xb - mab) : 0.f;
        f = ea + eb;
        if (mytopic >= 0) { ef[0][mytopic & 3][mytopic >> 2] = e; ef[1][mytopic & 3][mytopic >> 2] = f; }
        if constexpr (BW > 1) WAVE_PRIVATE_LDS_FENCE(); else WAVE_LDS_FENCE();
        float G[1], Hh[1];
        {   // terms: s_n = sum_i TA[n][i] e_i, w_n = c_n / s_n, G_i = sum_n w_n TA[n][i]
            gv2f w[NPT];
            grid_phase1<LPR, NPT>(A, ef[0][b], 0.0f, w);
#pragma unroll
            for (int q = 0; q < NPT; ++q) w[q] = ct[q] * gv2f{__builtin_amdgcn_rcpf(w[q].x), __builtin_amdgcn_rcpf(w[q].y)};
            float gG[LPR];
            grid_phase2<LPR, NPT>(A, w, gG);
            grid_reduce_scatter<LPR>(gG, G);
        }
        {   // readers: s_u = sum_i TH[u][i] f_i, w_u = r_u / s_u, H_i = sum_u w_u TH[u][i]
            gv2f w[NPR];
            grid_phase1<LPR, NPR>(H, ef[1][b], 0.0f, w);
#pragma unroll
            for (int q = 0; q < NPR; ++q) w[q] = rr[q] * gv2f{__builtin_amdgcn_rcpf(w[q].x), __builtin_amdgcn_rcpf(w[q].y)};
            float gH[LPR];
            grid_phase2<LPR, NPR>(H, w, gH);
            grid_reduce_scatter<LPR>(gH, Hh);
        }
        if constexpr (W > 1) {
            float (*xb2)[128] = xch[v & 1];
            xb2[wave][lane] = G[0]; xb2[wave][64 + lane] = Hh[0];
            __syncthreads();
            float tg = 0.0f, th = 0.0f;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) { tg += xb2[ww][lane]; th += xb2[ww][64 + lane]; }   // fixed order: identical in every wave
            G[0] = tg; Hh[0] = th;
        }
        zay_old = zay; gim_old = gim;
        float dl = 0.f;
        if (on) {
            zay = fmaf(eb, Hh[0], p.hg);                  // update_zayin!  :322
            gim = fmaf(ea, Hh[0], fmaf(e, G[0], p.hc));   // update_gimel!  :313
            const float df = gim - gim_old;
            dl = df * df;
        }
        if (wave_sum(dl) < vtol2) break;                  // :359, norm < vtol on the squares
    }
    if (wave == 0) {
        const bool mine = mytopic >= 0 && mytopic < R;
        if (sweeps > 0) {
            if (on) {
                p.gimel[(int64_t)d * K + mt] = gim; p.gimel_old[(int64_t)d * K + mt] = gim_old;
                p.zayin[(int64_t)d * K + mt] = zay; p.zayin_old[(int64_t)d * K + mt] = zay_old;
            }
            if (mine) { p.E1[(int64_t)d * p.estride + mytopic] = e; p.E2[(int64_t)d * p.estride + mytopic] = f; }
        } else {
            if (mine) { p.E1[(int64_t)d * p.estride + mytopic] = 0.f; p.E2[(int64_t)d * p.estride + mytopic] = 0.f; }   // viter = 0
        }
        if (lane == 0) {
            p.sweeps[d] = (uint8_t)min(sweeps, 255);
            if (p.shift) { p.shift[2 * (int64_t)d] = mx; p.shift[2 * (int64_t)d + 1] = mab; }
        }
    }
}

// length classes of the grid-tile kernel (term pairs, reader pairs per lane): documents of <= 32 readers by their terms
// (<= 64 / 96 / 128 / 192), <= 128 terms with <= 64 readers, and -- eight waves per document -- <= 512 terms with <= 512 readers
#define CTPF_GRID_CLASSES(X) X(2, 1) X(3, 1) X(4, 1) X(6, 1) X(4, 2)
// multi-wave classes, four waves per document: (2 term pairs, 4 reader pairs) = <= 256 terms, <= 512 readers, and (3, 3) = <= 384, <= 384
static inline bool ctpf_grid_class(int64_t n, int64_t r, int* npt, int* npr, int* waves)
{
    *waves = 1;
    if (r <= 32 && n <= 192) { *npr = 1; *npt = n <= 64 ? 2 : n <= 96 ? 3 : n <= 128 ? 4 : 6; return true; }
    if (r <= 64 && n <= 128) { *npt = 4; *npr = 2; return true; }
    if (n <= 256 && r <= 512) { *npt = 2; *npr = 4; *waves = 4; return true; }
    if (n <= 384 && r <= 384) { *npt = 3; *npr = 3; *waves = 4; return true; }
    return false;
}

#define CTPF_GRID_LDS(BWV)                                                                                              \
    __shared__ float xch[2][4][128];                    /* multi-wave documents: per-wave partial (G | H), by sweep parity */ \
    __shared__ __attribute__((aligned(16))) float ef_all[BWV][2][4][16]

template <int LPR, int NPT, int NPR>
__global__ __launch_bounds__(64) void ctpf_estep_grid_kernel(CtpfParams p, int64_t first, const int* __restrict__ topic_of_lane)
{
    CTPF_GRID_LDS(1);
    ctpf_estep_grid_body<LPR, NPT, NPR>(p, p.doc_order[first + blockIdx.x], topic_of_lane, ef_all, xch);
}
template <int LPR, int NPT, int NPR, int W>
__global__ __launch_bounds__(64 * W) void ctpf_estep_grid_long_kernel(CtpfParams p, int64_t first, const int* __restrict__ topic_of_lane)
{
    CTPF_GRID_LDS(W);
    ctpf_estep_grid_body<LPR, NPT, NPR, W>(p, p.doc_order[first + blockIdx.x], topic_of_lane, ef_all, xch);
}
// The two four-wave classes in ONE launch (round 4): the first `count_a` workgroups run the (3, 3) body, the others the (2, 4) body -- the
// host's own assignment, so every document's arithmetic is what the two launches did.  SYN-CITEU has 3 documents of the first class and
// 436 of the second; as two launches on one stream the three ran 39 us of pure latency IN FRONT of the 436 (E-step 74 us where the widest
// single-wave launch ends at 55), and a stream of their own cost more than it brought (a fifth active stream: 0.204 -> 0.235 / 0.255 ms
// per iteration).  Both bodies need 116 VGPRs, so the mixed launch costs no occupancy.
template <int LPR>
__global__ __launch_bounds__(256) void ctpf_estep_grid_long2_kernel(CtpfParams p, int64_t first, const int* __restrict__ topic_of_lane, int count_a)
{
    CTPF_GRID_LDS(4);
    const int d = p.doc_order[first + blockIdx.x];
    if ((int)blockIdx.x < count_a) ctpf_estep_grid_body<LPR, 3, 3, 4>(p, d, topic_of_lane, ef_all, xch);
    else ctpf_estep_grid_body<LPR, 2, 4, 4>(p, d, topic_of_lane, ef_all, xch);
}
// single-wave class of a document (wave-uniform)
template <int LPR, int BW>
__device__ __forceinline__ void ctpf_estep_grid_any(const CtpfParams& p, const int d, const int* __restrict__ topic_of_lane,
                                                   float (*ef_all)[2][4][16], float (*xch)[4][128])
{
    const int N = __builtin_amdgcn_readfirstlane((int)(p.doc_ptr[d + 1] - p.doc_ptr[d]));
    const int Rd = __builtin_amdgcn_readfirstlane((int)(p.rdr_ptr[d + 1] - p.rdr_ptr[d]));
    if (Rd > 32) ctpf_estep_grid_body<LPR, 4, 2, 1, BW>(p, d, topic_of_lane, ef_all, xch);
    else if (N > 128) ctpf_estep_grid_body<LPR, 6, 1, 1, BW>(p, d, topic_of_lane, ef_all, xch);
    else if (N > 96) ctpf_estep_grid_body<LPR, 4, 1, 1, BW>(p, d, topic_of_lane, ef_all, xch);
    else if (N > 64) ctpf_estep_grid_body<LPR, 3, 1, 1, BW>(p, d, topic_of_lane, ef_all, xch);
    else ctpf_estep_grid_body<LPR, 2, 1, 1, BW>(p, d, topic_of_lane, ef_all, xch);
}
// every single-wave class in one launch, the class read per document (wave-uniform): one kernel tail per iteration
template <int LPR>
__global__ __launch_bounds__(64) void ctpf_estep_grid_any_kernel(CtpfParams p, int64_t first, const int* __restrict__ topic_of_lane)
{
    CTPF_GRID_LDS(1);
    ctpf_estep_grid_any<LPR, 1>(p, p.doc_order[first + blockIdx.x], topic_of_lane, ef_all, xch);
}
// Round 4: that one launch ran EVERY document at ONE wave per SIMD -- the (6, 1) body needs 258 registers, and a kernel's allocation is
// its widest path's (the ISA said so; the kernel issued a vector instruction 42 % of the time).  Two launches by register need instead:
//   NARROW  (2, 1) and (3, 1): documents of <= 96 terms and <= 32 readers (most of a CiteULike-shaped corpus), 168 VGPRs = three waves per SIMD,
//           on the context's stream;
//   WIDE    (4, 1), (6, 1), (4, 2): the rest of the single-wave classes, on aux[0] beside it (the multi-wave long documents keep aux[1]).
template <int LPR>
__global__ __launch_bounds__(64, 3) void ctpf_estep_grid_narrow_kernel(CtpfParams p, int64_t first, const int* __restrict__ topic_of_lane)
{
    CTPF_GRID_LDS(1);
    const int d = p.doc_order[first + blockIdx.x];
    const int N = __builtin_amdgcn_readfirstlane((int)(p.doc_ptr[d + 1] - p.doc_ptr[d]));
    if (N > 64) ctpf_estep_grid_body<LPR, 3, 1, 1, 1>(p, d, topic_of_lane, ef_all, xch);
    else ctpf_estep_grid_body<LPR, 2, 1, 1, 1>(p, d, topic_of_lane, ef_all, xch);
}
template <int LPR>
__global__ __launch_bounds__(64) void ctpf_estep_grid_wide_kernel(CtpfParams p, int64_t first, const int* __restrict__ topic_of_lane)
{
    CTPF_GRID_LDS(1);
    const int d = p.doc_order[first + blockIdx.x];
    const int N = __builtin_amdgcn_readfirstlane((int)(p.doc_ptr[d + 1] - p.doc_ptr[d]));
    const int Rd = __builtin_amdgcn_readfirstlane((int)(p.rdr_ptr[d + 1] - p.rdr_ptr[d]));
    if (Rd > 32) ctpf_estep_grid_body<LPR, 4, 2, 1, 1>(p, d, topic_of_lane, ef_all, xch);
    else if (N > 128) ctpf_estep_grid_body<LPR, 6, 1, 1, 1>(p, d, topic_of_lane, ef_all, xch);
    else ctpf_estep_grid_body<LPR, 4, 1, 1, 1>(p, d, topic_of_lane, ef_all, xch);
}
template <int LPR>
static void ctpf_launch_grid(const tmvb_bucket& b, dim3 grid, hipStream_t st, const CtpfParams& p, int64_t first, const int* tol)
{

    if (b.waves == 4 && b.grid_np == 96) { hipLaunchKernelGGL((ctpf_estep_grid_long2_kernel<LPR>), grid, dim3(256), 0, st, p, first, tol, (int)b.grid_np2); return; }
    if (b.waves == 4 && b.grid_np == 2) { hipLaunchKernelGGL((ctpf_estep_grid_long_kernel<LPR, 2, 4, 4>), grid, dim3(256), 0, st, p, first, tol); return; }
    if (b.waves == 4) { hipLaunchKernelGGL((ctpf_estep_grid_long_kernel<LPR, 3, 3, 4>), grid, dim3(256), 0, st, p, first, tol); return; }
    if (b.grid_np == 99) { hipLaunchKernelGGL((ctpf_estep_grid_any_kernel<LPR>), grid, dim3(64), 0, st, p, first, tol); return; }
    if (b.grid_np == 98) { hipLaunchKernelGGL((ctpf_estep_grid_wide_kernel<LPR>), grid, dim3(64), 0, st, p, first, tol); return; }
    if (b.grid_np == 97) { hipLaunchKernelGGL((ctpf_estep_grid_narrow_kernel<LPR>), grid, dim3(64), 0, st, p, first, tol); return; }
#define CTPF_GRID_LAUNCH(T_, R_) if (b.grid_np == T_ && b.grid_np2 == R_) { hipLaunchKernelGGL((ctpf_estep_grid_kernel<LPR, T_, R_>), grid, dim3(64), 0, st, p, first, tol); return; }
    CTPF_GRID_CLASSES(CTPF_GRID_LAUNCH)
#undef CTPF_GRID_LAUNCH
}
// shape update + table refresh:  X[id][i] = prior + stats[id][i];  T[id][i] = exp(psi(X));  stats <- 0
// (update_alef!/update_he! src/CTPF.jl:251-255, :266-270: X <- X_temp, X_temp <- prior)
__global__ __launch_bounds__(256) void ctpf_shape_kernel(float* __restrict__ stats, float prior, float* __restrict__ X,
                                                         float* __restrict__ X_old /* or NULL */,
                                                         float* __restrict__ T, int K, int KP, int64_t n_ids, int refresh_only)
{
    const int64_t total = n_ids * KP;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const int64_t j = q / KP;
        const int i = (int)(q - j * KP);
        float t = 0.0f;
        if (i < K) {
            float x;
            if (refresh_only) x = X[j * K + i];
            else {
                if (X_old) X_old[j * K + i] = X[j * K + i];               // X_old <- X  (src/CTPF.jl:252, :267)
                x = prior + stats[j * K + i]; X[j * K + i] = x; stats[j * K + i] = 0.0f;
            }
            t = expf(digamma_f(x));
        }
        T[q] = t;
    }
}

struct CtpfShapeJob { float* stats; float prior; float* X; float* X_old; float* T; int64_t n_ids; };
// the he and the alef update of one M-step in one launch (blockIdx.y selects the job): a 5 us kernel fewer in a 0.3 ms iteration
__global__ __launch_bounds__(256) void ctpf_shape2_kernel(CtpfShapeJob j0, CtpfShapeJob j1, int K, int KP)
{
    const CtpfShapeJob j = blockIdx.y ? j1 : j0;
    const int64_t total = j.n_ids * KP;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const int64_t id = q / KP;
        const int i = (int)(q - id * KP);
        float t = 0.0f;
        if (i < K) {
            if (j.X_old) j.X_old[id * K + i] = j.X[id * K + i];               // X_old <- X  (src/CTPF.jl:252, :267)
            const float x = j.prior + j.stats[id * K + i];
            j.X[id * K + i] = x; j.stats[id * K + i] = 0.0f;
            t = expf(digamma_f(x));
        }
        j.T[q] = t;
    }
}

// dalet, het, bet, vav in the reference's order (src/CTPF.jl:368-371), fp64, one wave
template <int NSLOT>
__global__ __launch_bounds__(64) void ctpf_rates_kernel(int K, double hb, double hd, double hf, double hh,
                                                        const double* __restrict__ rs_alef, const double* __restrict__ rs_he,
                                                        const double* __restrict__ sum_gimel, const double* __restrict__ sum_zayin,
                                                        double* __restrict__ rates /* [8][K]: bet,vav,dalet,het, then *_old */,
                                                        float* __restrict__ lrates /* [4][K] logs */)
{
    for (int i = threadIdx.x; i < K; i += 64) {
        const double bet = rates[i], vav = rates[K + i], dalet = rates[2 * K + i], het = rates[3 * K + i];
        const double dalet_n = (hd + rs_alef[i] / bet) + rs_he[i] / vav;          // :297 (old bet, vav)
        const double het_n = hh + rs_he[i] / vav;                                 // :304
        const double bet_n = hb + sum_gimel[i] / dalet_n;                         // :283 (new dalet)
        const double vav_n = (hf + sum_gimel[i] / dalet_n) + sum_zayin[i] / het_n;   // :290
        rates[4 * K + i] = bet; rates[5 * K + i] = vav; rates[6 * K + i] = dalet; rates[7 * K + i] = het;
        rates[i] = bet_n; rates[K + i] = vav_n; rates[2 * K + i] = dalet_n; rates[3 * K + i] = het_n;
        lrates[i] = (float)log(bet_n); lrates[K + i] = (float)log(TMVB_CTPF_XI_RATE(bet_n, vav_n));
        lrates[2 * K + i] = (float)log(dalet_n); lrates[3 * K + i] = (float)log(het_n);
    }
}

// ---- the fused M-step (K <= 64; round 4).  The M-step used to be six dependent launches of 4-7 us each (column sums of gimel / zayin in
// two stages, the shape update, column sums of alef / he in two stages, the rates): 36 us of a 185 us iteration, nearly all of it
// kernel boundaries and single-trip latencies.  Now ONE launch, grid (NB, 4) x 16 waves, lane = topic, a wave per id / document:
//     y = 0  he:   he_old <- he; he <- e + stats; stats <- 0; TH <- exp(psi(he))        (update_he!   src/CTPF.jl:266-270)
//     y = 1  alef: the same with a, TA                                                   (update_alef! :251-255)
//     y = 2  sum_d gimel_d      y = 3  sum_d zayin_d      (only when tmvb_ctpf_reduce_docs deferred them to here; a sharded run
//                                                          all-reduces them first and launches y = 0, 1 only)
// Blocks own contiguous id ranges, every wave has all its (<= 8 / <= 16) rows in flight at once; each block leaves the fp64 column
// sums of what it wrote / read in partial[y][block][topic] (ids ascending per wave, then the sixteen waves in order), and the block
// that finishes LAST (a device-scope counter; no block ever waits for another) adds the partial rows in a fixed order and runs
// update_dalet! / update_het! / update_bet! / update_vav! (:295-305, :281-291) -- every rate depends on its own topic's sums only.
// Deterministic: which block happens to be last changes who adds, never the order of the additions; run-to-run bitwise reproducible
// like the two-stage column sums it replaces.
struct CtpfMstepJob { float* stats; float prior; float* X; float* X_old; float* T; int64_t n_ids; };
struct CtpfMstepTail {
    int with_docs;                     // 1: y = 2, 3 are part of this launch
    double hb, hd, hf, hh;
    double* partial;                   // [4][NB][K]
    unsigned int* counter;             // zero between launches (the last block resets it)
    double* rs_he; double* rs_alef; double* sum_gimel; double* sum_zayin;
    float* tail;                       // [2][K] fp32 copies of the document sums (statistics tail)
    double* rates; float* lrates;
};

__device__ __forceinline__ double ctpf_coherent_load(const double* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(1024) void ctpf_mstep_kernel(CtpfMstepJob jh, CtpfMstepJob ja, const float* __restrict__ gimel,
                                                          const float* __restrict__ zayin, int64_t M, CtpfMstepTail t, int K, int KP)
{
    __shared__ double red[16][64];
    __shared__ unsigned int last_flag;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nbk = (int)gridDim.x, bx = (int)blockIdx.x, y = (int)blockIdx.y;
    const bool on = lane < K;
    double a0 = 0.0;
    if (y < 2) {
        const CtpfMstepJob j = y ? ja : jh;
        const int64_t per = (j.n_ids + nbk - 1) / nbk;
        const int64_t lo = (int64_t)bx * per, hi = min(lo + per, j.n_ids);
        for (int64_t id0 = lo + wv; id0 < hi; id0 += 16 * 8) {        // eight ids of this wave per trip, their loads in flight together
            float xo[8], sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t q = min(id0 + 16 * u, hi - 1) * K + (on ? lane : 0);
                xo[u] = j.X[q]; sv[u] = j.stats[q];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t id = id0 + 16 * u;
                if (id >= hi) break;                                  // wave-uniform
                float tv = 0.0f;
                if (on) {
                    const float x = j.prior + sv[u];
                    if (j.X_old) j.X_old[id * K + lane] = xo[u];      // X_old <- X
                    j.X[id * K + lane] = x; j.stats[id * K + lane] = 0.0f;
                    tv = expf(digamma_f(x));
                    a0 += (double)x;
                }
                if (lane < KP) j.T[id * KP + lane] = tv;
            }
        }
    } else {
        const float* __restrict__ X = y == 2 ? gimel : zayin;
        const int64_t per = (M + nbk - 1) / nbk;
        const int64_t lo = (int64_t)bx * per, hi = min(lo + per, M);
        for (int64_t d0 = lo + wv; d0 < hi; d0 += 16 * 16) {
            float g[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) g[u] = X[min(d0 + 16 * u, hi - 1) * K + (on ? lane : 0)];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (d0 + 16 * u >= hi) break;
                a0 += (double)g[u];
            }
        }
    }
    red[wv][lane] = a0;
    __syncthreads();
    // ---- last block done: every block publishes its partial row, then counts itself.  ONE wave writes the row, fences (release at
    // device scope: a wave-level L2 write-back -- with all sixteen waves of all 256 blocks fencing, the kernel took 82 us) and counts;
    // the last block reads the rows with device-coherent loads.
    if (wv == 0) {
        if (on) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < 16; ++w) s += red[w][lane];
            __hip_atomic_store(&t.partial[((int64_t)y * nbk + bx) * K + lane], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
        if (lane == 0) {
            const unsigned int total = gridDim.x * gridDim.y;
            const unsigned int prev = atomicAdd(t.counter, 1u);
            last_flag = (prev == total - 1u) ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!last_flag) return;
    if (wv == 0) __threadfence();
    __syncthreads();
    // wave (m, part): rows b = part, part + 4, ... of matrix m, all in flight
    const int m = wv >> 2, part = wv & 3;
    double acc = 0.0;
    if (m < 2 || t.with_docs) {
        const double* src = t.partial + (int64_t)m * nbk * K + (on ? lane : 0);
        for (int b0 = part; b0 < nbk; b0 += 64) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = ctpf_coherent_load(src + (int64_t)min(b0 + 4 * u, nbk - 1) * K);
#pragma unroll
            for (int u = 0; u < 16; ++u) if (b0 + 4 * u < nbk) acc += v[u];
        }
    }
    __syncthreads();                                                  // red[] is reused
    red[wv][lane] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *t.counter = 0u;                            // ready for the next launch (stream order)
    if (wv != 0 || !on) return;
    double tot[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tot[q] = (red[4 * q][lane] + red[4 * q + 1][lane]) + (red[4 * q + 2][lane] + red[4 * q + 3][lane]);
    const int i = lane;
    t.rs_he[i] = tot[0]; t.rs_alef[i] = tot[1];
    if (t.with_docs) { t.sum_gimel[i] = tot[2]; t.sum_zayin[i] = tot[3]; t.tail[i] = (float)tot[2]; t.tail[K + i] = (float)tot[3]; }
    else { tot[2] = t.sum_gimel[i]; tot[3] = t.sum_zayin[i]; }
    double* rates = t.rates; float* lrates = t.lrates;
    const double bet = rates[i], vav = rates[K + i], dalet = rates[2 * K + i], het = rates[3 * K + i];
    const double dalet_n = (t.hd + tot[1] / bet) + tot[0] / vav;                  // :297 (old bet, vav)
    const double het_n = t.hh + tot[0] / vav;                                     // :304
    const double bet_n = t.hb + tot[2] / dalet_n;                                 // :283 (new dalet)
    const double vav_n = (t.hf + tot[2] / dalet_n) + tot[3] / het_n;              // :290
    rates[4 * K + i] = bet; rates[5 * K + i] = vav; rates[6 * K + i] = dalet; rates[7 * K + i] = het;
    rates[i] = bet_n; rates[K + i] = vav_n; rates[2 * K + i] = dalet_n; rates[3 * K + i] = het_n;
    lrates[i] = (float)log(bet_n); lrates[K + i] = (float)log(TMVB_CTPF_XI_RATE(bet_n, vav_n));
    lrates[2 * K + i] = (float)log(dalet_n); lrates[3 * K + i] = (float)log(het_n);
}

// ------------------------------------------------------------------------------ ELBO (src/CTPF.jl:111-247)
// The Binomial sums  sum_y pdf(Binomial(n,p),y) lgamma(y+1)  enter Elogpya/Elogpyb/Elogpz with a minus sign
// (src/CTPF.jl:116,:127,:138) and -Elogqy/-Elogqz through entropy(Multinomial) with a plus sign (:183,:192): they
// cancel identically in update_elbo! (:243), so the device evaluates the remaining closed form
//   sum_u [ ra <xi_u, log-rates + psi(he)> - lgamma(ra+1) + ra H(xi_u) ] + sum_n [ c <phi_n, ...> - lgamma(c+1) + c H(phi_n) ].
__device__ __forceinline__ double gamma_entropy_d(double a, double rate)   // entropy(Gamma(shape a, scale 1/rate))
{
    return a - log(rate) + lgamma(a) + (1.0 - a) * digamma_d(a);
}

// per-document part; one wave per document, lane l owns topics l + 64 s (NS slots)
template <int NS>
__global__ __launch_bounds__(64) void ctpf_elbo_doc_kernel(int K, const int64_t* __restrict__ doc_ptr, const int32_t* __restrict__ terms,
                                                           const int32_t* __restrict__ counts, const int64_t* __restrict__ rdr_ptr,
                                                           const int32_t* __restrict__ readers, const int32_t* __restrict__ ratings,
                                                           const float* __restrict__ alef, const float* __restrict__ alef_old,
                                                           const float* __restrict__ he, const float* __restrict__ he_old,
                                                           const double* __restrict__ rates /* [8][K] */, const double* __restrict__ rs_alef,
                                                           const double* __restrict__ rs_he, const float* __restrict__ gimel,
                                                           const float* __restrict__ gimel_old, const float* __restrict__ zayin,
                                                           const float* __restrict__ zayin_old, double hc, double hd, double hg, double hh,
                                                           double* __restrict__ doc_val)
{
    const int lane = threadIdx.x;
    const int d = blockIdx.x;
    bool on[NS]; int ix[NS];
    double bet[NS], vav[NS], dalet[NS], het[NS], gi[NS], za[NS], lt[NS], le[NS], lbet[NS], lvav[NS];
    float xo[NS], ao[NS], bo[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        on[s] = lane + 64 * s < K;
        const int i = on[s] ? lane + 64 * s : 0;
        ix[s] = i;
        bet[s] = rates[i]; vav[s] = rates[K + i]; dalet[s] = rates[2 * K + i]; het[s] = rates[3 * K + i];
        const double bet_o = rates[4 * K + i], vav_o = rates[5 * K + i], dalet_o = rates[6 * K + i], het_o = rates[7 * K + i];
        gi[s] = on[s] ? (double)gimel[(int64_t)d * K + i] : 1.0; za[s] = on[s] ? (double)zayin[(int64_t)d * K + i] : 1.0;
        const double gio = on[s] ? (double)gimel_old[(int64_t)d * K + i] : 1.0, zao = on[s] ? (double)zayin_old[(int64_t)d * K + i] : 1.0;
        // softmax arguments rebuilt from the *_old variables (:240-241)
        xo[s] = (float)(digamma_d(gio) - log(dalet_o) - log(bet_o));
        ao[s] = (float)(digamma_d(gio) - log(dalet_o) - log(vav_o));
        bo[s] = (float)(digamma_d(zao) - log(het_o) - log(vav_o));
        // log-rate parts of the current variables
        lt[s] = digamma_d(gi[s]) - log(dalet[s]);       // E[log theta]
        le[s] = digamma_d(za[s]) - log(het[s]);         // E[log epsilon]
        lbet[s] = log(bet[s]); lvav[s] = log(vav[s]);
    }
    double acc = 0.0;
    // tokens: Elogpz - Elogqz (without the cancelling Binomial sums)
    const int64_t off = doc_ptr[d];
    const int N = (int)(doc_ptr[d + 1] - off);
    for (int n = 0; n < N; ++n) {
        const int t = terms[off + n];
        const double c = (double)counts[off + n];
        float x[NS], ml = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            x[s] = on[s] ? xo[s] + digamma_f(alef_old[(int64_t)t * K + ix[s]]) : -INFINITY;
            ml = fmaxf(ml, x[s]);
        }
        const float mx = wave_max(ml);
        float ex[NS], sl = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s) { ex[s] = on[s] ? expf(x[s] - mx) : 0.0f; sl += ex[s]; }
        const float inv = 1.0f / wave_sum(sl);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float ph = ex[s] * inv;
            if (on[s]) {
                acc += c * (double)ph * (lt[s] + (double)digamma_f(alef[(int64_t)t * K + ix[s]]) - lbet[s]);
                if (ph > 0.0f) acc -= c * (double)ph * (double)logf(ph);          // + c H(phi_n)
            }
        }
        if (lane == 0) acc -= lgamma(c + 1.0);
    }
    // readers: Elogpya + Elogpyb - Elogqy
    const int64_t roff = rdr_ptr[d];
    const int Rd = (int)(rdr_ptr[d + 1] - roff);
    for (int u = 0; u < Rd; ++u) {
        const int r = readers[roff + u];
        const double ra = (double)ratings[roff + u];
        float a[NS], b[NS], ml = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float dho = on[s] ? digamma_f(he_old[(int64_t)r * K + ix[s]]) : 0.0f;
            a[s] = on[s] ? ao[s] + dho : -INFINITY; b[s] = on[s] ? bo[s] + dho : -INFINITY;
            ml = fmaxf(ml, fmaxf(a[s], b[s]));
        }
        const float mx = wave_max(ml);
        float ea[NS], eb[NS], sl = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            ea[s] = on[s] ? expf(a[s] - mx) : 0.0f; eb[s] = on[s] ? expf(b[s] - mx) : 0.0f;
            sl += ea[s] + eb[s];
        }
        const float inv = 1.0f / wave_sum(sl);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (on[s]) {
                const double xt = (double)(ea[s] * inv), xb = (double)(eb[s] * inv);
                const double dh = (double)digamma_f(he[(int64_t)r * K + ix[s]]);
                acc += ra * xt * (lt[s] + dh - lvav[s]) + ra * xb * (le[s] + dh - lvav[s]);
                if (xt > 0.0) acc -= ra * xt * log(xt);
                if (xb > 0.0) acc -= ra * xb * log(xb);
            }
        }
        if (lane == 0) acc -= lgamma(ra + 1.0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (on[s]) {
            const int i = ix[s];
            acc -= gi[s] / (dalet[s] * vav[s]) * rs_he[i] + za[s] / (het[s] * vav[s]) * rs_he[i] + gi[s] / (dalet[s] * bet[s]) * rs_alef[i];   // :112,:123,:134
            acc += (hc - 1.0) * lt[s] - hd * gi[s] / dalet[s];                    // Elogptheta :156
            acc += (hg - 1.0) * le[s] - hh * za[s] / het[s];                      // Elogpepsilon :174
            acc += gamma_entropy_d(gi[s], dalet[s]) + gamma_entropy_d(za[s], het[s]);   // -Elogqtheta, -Elogqepsilon
        }
    }
    double tot = wave_sum_d(acc);
    tot += (double)K * (hc * log(hd) - lgamma(hc)) + (double)K * (hg * log(hh) - lgamma(hg));
    if (lane == 0) doc_val[d] = tot;
}

// ---- update_elbo!'s per-document part, fast form (round 3) ------------------------------------------------------------------
// The kernel above spends its time on 2 K digammas per token and per reader (psi(alef_old), psi(alef) of the entry's row) and on
// three wave reductions per entry.  Both go away with two per-call tables per shape matrix,
//     To[id][i] = exp(psi(X_old[id][i])),    D[id][i] = psi(X[id][i]) - psi(X_old[id][i])       (ctpf_elbo_tables_kernel),
// because an entry's contribution factors through per-document vectors Q (the softmax factors) and P:
//     phi_in = To[t_n][i] e_i / s_n,  s_n = sum_i To[t_n][i] e_i,  e_i = exp(x_i - max x)
//     c_n sum_i phi_in (lt_i + psi(alef[i,t_n]) - log bet_i - log phi_in) = c_n [ (sum_i To_i P_i + sum_i To_i D_i Q_i) / s_n + log s_n ]
//     with Q_i = e_i, P_i = e_i (lt_i - log bet_i - x_i + max x); the readers the same with Q_i = ea_i + eb_i and
//     P_i = ea_i (lt_i - log vav_i - a_i + m) + eb_i (le_i - log vav_i - b_i + m), m = max(a, b).
// One wave per document: lane = topic for the per-document vectors (fp64, as above), then lane = token / reader for the entries:
// two row reads, 3 K fmas, one reciprocal and one logarithm per entry.  sum_n lgamma(c_n + 1) + sum_u lgamma(r_u + 1) is a
// per-document constant of the corpus (lg_doc, computed once on the host).
__global__ __launch_bounds__(256) void ctpf_elbo_tables_kernel(const float* __restrict__ X, const float* __restrict__ X_old,
                                                               float* __restrict__ To, float* __restrict__ D, int K, int KP, int64_t n_ids,
                                                               const double* __restrict__ rates, double* __restrict__ lrates_d)
{
    // log of the eight rate vectors (bet, vav, dalet, het and their *_old) in fp64, once per call instead of per document and lane
    if (lrates_d != nullptr && blockIdx.x == 0)
        for (int q = threadIdx.x; q < 8 * K; q += blockDim.x) lrates_d[q] = log(rates[q]);
    const int64_t total = n_ids * KP;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = q / KP;
        const int i = (int)(q - id * KP);
        float to = 0.0f, dd = 0.0f;
        if (i < K) {
            const float po = digamma_f(X_old[id * K + i]);
            to = expf(po);
            dd = digamma_f(X[id * K + i]) - po;
        }
        To[q] = to; D[q] = dd;
    }
}

// sum over the entries [0, n_ent) of one document of val[entry] * ((sum_i To_i P_i + sum_i To_i D_i Q_i) / s + log s), lane = entry
__device__ __forceinline__ float ctpf_elbo_entries(const int* __restrict__ ids, const int* __restrict__ vals, const int n_ent,
                                                   const float* __restrict__ To, const float* __restrict__ D, const int KP,
                                                   const float* __restrict__ Ql, const float* __restrict__ Pl, const int lane)
{
    float acc = 0.0f;
    const int lpr = KP >> 2;
    for (int n0 = 0; n0 < n_ent; n0 += 64) {
        const int n = n0 + lane;
        const bool in = n < n_ent;
        const int id = ids[in ? n : 0];
        const float cv = in ? (float)vals[in ? n : 0] : 0.0f;
        const float4* __restrict__ rt = (const float4*)(To + (int64_t)id * KP);
        const float4* __restrict__ rd = (const float4*)(D + (int64_t)id * KP);
        float s = 0.0f, u = 0.0f, w = 0.0f;
#pragma unroll 4
        for (int q = 0; q < lpr; ++q) {
            const float4 t = rt[q], dd = rd[q];
            const float4 qq = ((const float4*)Ql)[q], pp = ((const float4*)Pl)[q];
            s = fmaf(t.x, qq.x, fmaf(t.y, qq.y, fmaf(t.z, qq.z, fmaf(t.w, qq.w, s))));
            u = fmaf(t.x, pp.x, fmaf(t.y, pp.y, fmaf(t.z, pp.z, fmaf(t.w, pp.w, u))));
            w = fmaf(t.x * dd.x, qq.x, fmaf(t.y * dd.y, qq.y, fmaf(t.z * dd.z, qq.z, fmaf(t.w * dd.w, qq.w, w))));
        }
        acc = fmaf(cv, (u + w) / s + logf(s), acc);       // s > 0: To > 0 and the largest factor of Q is 1
    }
    return acc;
}

// The same sum with FOUR lanes per entry (lane = 4 a + b: entry n0 + a, chunks b, b + 4, b + 8, b + 12 of the entry's two rows): a load
// instruction then touches 16 rows, one 64-byte run each, where lane = entry touches 64 rows in 64 different cache lines -- the CU's vector
// L1 looks up one line per cycle (the CTM token phase's lesson).  The partial sums meet inside the quad (two DPP adds each).
#ifndef TMVB_CTPF_ELBO_QUAD
#define TMVB_CTPF_ELBO_QUAD 1
#endif
__device__ __forceinline__ float ctpf_elbo_entries_quad(const int* __restrict__ ids, const int* __restrict__ vals, const int n_ent,
                                                        const float* __restrict__ To, const float* __restrict__ D, const int KP,
                                                        const float* __restrict__ Ql, const float* __restrict__ Pl, const int lane)
{
    float acc = 0.0f;
    const int lpr = KP >> 2;
    const int a = lane >> 2, b = lane & 3;
    for (int n0 = 0; n0 < n_ent; n0 += 16) {
        const int n = n0 + a;
        const bool in = n < n_ent;
        const int id = ids[in ? n : 0];
        const float cv = in ? (float)vals[in ? n : 0] : 0.0f;
        const float4* __restrict__ rt = (const float4*)(To + (int64_t)id * KP);
        const float4* __restrict__ rd = (const float4*)(D + (int64_t)id * KP);
        float s = 0.0f, u = 0.0f, w = 0.0f;
        for (int q = b; q < lpr; q += 4) {
            const float4 t = rt[q], dd = rd[q];
            const float4 qq = ((const float4*)Ql)[q], pp = ((const float4*)Pl)[q];
            s = fmaf(t.x, qq.x, fmaf(t.y, qq.y, fmaf(t.z, qq.z, fmaf(t.w, qq.w, s))));
            u = fmaf(t.x, pp.x, fmaf(t.y, pp.y, fmaf(t.z, pp.z, fmaf(t.w, pp.w, u))));
            w = fmaf(t.x * dd.x, qq.x, fmaf(t.y * dd.y, qq.y, fmaf(t.z * dd.z, qq.z, fmaf(t.w * dd.w, qq.w, w))));
        }
        float uw = u + w;
        s += dpp_f<0xB1>(s); uw += dpp_f<0xB1>(uw);
        s += dpp_f<0x4E>(s); uw += dpp_f<0x4E>(uw);
        if (b == 0) acc = fmaf(cv, uw / s + logf(s), acc);   // s > 0: To > 0 and the largest factor of Q is 1
    }
    return acc;
}

template <int NS>
__global__ __launch_bounds__(64) void ctpf_elbo_doc_fast_kernel(int K, int KP, const int64_t* __restrict__ doc_ptr, const int32_t* __restrict__ terms,
                                                                const int32_t* __restrict__ counts, const int64_t* __restrict__ rdr_ptr,
                                                                const int32_t* __restrict__ readers, const int32_t* __restrict__ ratings,
                                                                const float* __restrict__ TAo, const float* __restrict__ DA,
                                                                const float* __restrict__ THo, const float* __restrict__ DH,
                                                                const double* __restrict__ rates /* [8][K] */, const double* __restrict__ lrates_d /* their logs */,
                                                                const double* __restrict__ rs_alef,
                                                                const double* __restrict__ rs_he, const float* __restrict__ gimel,
                                                                const float* __restrict__ gimel_old, const float* __restrict__ zayin,
                                                                const float* __restrict__ zayin_old, const double* __restrict__ lg_doc,
                                                                double hc, double hd, double hg, double hh, double doc_const, double* __restrict__ doc_val)
{
    // doc_const = K (c log d - lgamma(c)) + K (g log h - lgamma(g)), from the host: the library's fp64 lgamma / log of four hyperparameters
    // were evaluated here by every wave (round 4: ~2 us of a document's ~7)
    __shared__ __attribute__((aligned(16))) float PQ[4][64 * NS + 8];  // Q1 | P1 | Q2 | P2, pads zero (KP = 4 * odd can exceed 64 NS by 4)
    const int lane = threadIdx.x;
    const int d = blockIdx.x;
    double acc = 0.0;
    double xo[NS], ao[NS], bo[NS], lt[NS], le[NS], lbet[NS], lvav[NS];
    bool on[NS];
    double mx = -INFINITY, mab = -INFINITY;
    // (the logs of the rates come from lrates_d, psi and lgamma of a shape from one evaluation -- digamma_lgamma_d: this prelude was
    //  ~4000 fp64 instructions per document with the library's lgamma / log and the loop form of digamma, and the kernel's whole time)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        on[s] = lane + 64 * s < K;
        const int i = on[s] ? lane + 64 * s : 0;
        const double bet = rates[i], vav = rates[K + i], dalet = rates[2 * K + i], het = rates[3 * K + i];
        const double l_bet = lrates_d[i], l_vav = lrates_d[K + i], l_dalet = lrates_d[2 * K + i], l_het = lrates_d[3 * K + i];
        const double l_bet_o = lrates_d[4 * K + i], l_vav_o = lrates_d[5 * K + i], l_dalet_o = lrates_d[6 * K + i], l_het_o = lrates_d[7 * K + i];
        const double gi = on[s] ? (double)gimel[(int64_t)d * K + i] : 1.0, za = on[s] ? (double)zayin[(int64_t)d * K + i] : 1.0;
        const double gio = on[s] ? (double)gimel_old[(int64_t)d * K + i] : 1.0, zao = on[s] ? (double)zayin_old[(int64_t)d * K + i] : 1.0;
        const double dgo = digamma_shift8_d(gio);
        xo[s] = dgo - l_dalet_o - l_bet_o;                 // softmax arguments rebuilt from the *_old variables (:240-241)
        ao[s] = dgo - l_dalet_o - l_vav_o;
        bo[s] = digamma_shift8_d(zao) - l_het_o - l_vav_o;
        double psi_g, lg_g, psi_z, lg_z;
        digamma_lgamma_d(gi, psi_g, lg_g);
        digamma_lgamma_d(za, psi_z, lg_z);
        lt[s] = psi_g - l_dalet;                          // E[log theta]
        le[s] = psi_z - l_het;                            // E[log epsilon]
        lbet[s] = l_bet; lvav[s] = l_vav;
        if (on[s]) {
            const double r_dalet = tmvb_rcp_d(dalet), r_het = tmvb_rcp_d(het), r_vav = tmvb_rcp_d(vav), r_bet = tmvb_rcp_d(bet);
            mx = fmax(mx, xo[s]); mab = fmax(mab, fmax(ao[s], bo[s]));
            acc -= (gi * r_dalet * r_vav + za * r_het * r_vav) * rs_he[i] + gi * r_dalet * r_bet * rs_alef[i];   // :112,:123,:134
            acc += (hc - 1.0) * lt[s] - hd * gi * r_dalet;                        // Elogptheta :156
            acc += (hg - 1.0) * le[s] - hh * za * r_het;                          // Elogpepsilon :174
            acc += (gi - l_dalet + lg_g + (1.0 - gi) * psi_g) + (za - l_het + lg_z + (1.0 - za) * psi_z);   // -Elogqtheta, -Elogqepsilon: Gamma entropies
        }
    }
    mx = -wave_min_d(-mx); mab = -wave_min_d(-mab);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float q1 = 0.f, p1 = 0.f, q2 = 0.f, p2 = 0.f;
        if (on[s]) {
            const double e = exp(xo[s] - mx), ea = exp(ao[s] - mab), eb = exp(bo[s] - mab);
            q1 = (float)e; p1 = (float)(e * (lt[s] - lbet[s] - xo[s] + mx));
            q2 = (float)(ea + eb);
            p2 = (float)(ea * (lt[s] - lvav[s] - ao[s] + mab) + eb * (le[s] - lvav[s] - bo[s] + mab));
        }
        PQ[0][lane + 64 * s] = q1; PQ[1][lane + 64 * s] = p1; PQ[2][lane + 64 * s] = q2; PQ[3][lane + 64 * s] = p2;
    }
    if (lane < 8) { PQ[0][64 * NS + lane] = 0.f; PQ[1][64 * NS + lane] = 0.f; PQ[2][64 * NS + lane] = 0.f; PQ[3][64 * NS + lane] = 0.f; }
    WAVE_LDS_FENCE();
    const int64_t off = doc_ptr[d], roff = rdr_ptr[d];
    const int N = (int)(doc_ptr[d + 1] - off), Rd = (int)(rdr_ptr[d + 1] - roff);
    float ent = 0.0f;
    // (TMVB_CTPF_ELBO_SKIP: timing by elimination, tools/build_variant.sh -- 1 = no entry sums, 2 = no term entries, 3 = no reader entries;
    //  wrong results, never the shipped build)
#ifndef TMVB_CTPF_ELBO_SKIP
#define TMVB_CTPF_ELBO_SKIP 0
#endif
#if TMVB_CTPF_ELBO_QUAD
#define CTPF_ELBO_ENTRIES ctpf_elbo_entries_quad
#else
#define CTPF_ELBO_ENTRIES ctpf_elbo_entries
#endif
    if (N > 0 && TMVB_CTPF_ELBO_SKIP != 1 && TMVB_CTPF_ELBO_SKIP != 2) ent += CTPF_ELBO_ENTRIES(terms + off, counts + off, N, TAo, DA, KP, PQ[0], PQ[1], lane);
    if (Rd > 0 && TMVB_CTPF_ELBO_SKIP != 1 && TMVB_CTPF_ELBO_SKIP != 3) ent += CTPF_ELBO_ENTRIES(readers + roff, ratings + roff, Rd, THo, DH, KP, PQ[2], PQ[3], lane);
#undef CTPF_ELBO_ENTRIES
    acc += (double)ent;
    double tot = wave_sum_d(acc);
    tot += doc_const - lg_doc[d];
    if (lane == 0) doc_val[d] = tot;
}

// global part over the entries of a K x n shape matrix X with rate vector `rate`:
//   sum [ (prior_shape - 1)(psi(x) - log rate_i) - prior_rate x / rate_i + entropy(Gamma(x, 1/rate_i)) ]
// (Elogpbeta - Elogqbeta :144-150,:198-204 with (a, b, alef, bet); Elogpeta - Elogqeta :162-168,:216-222 with (e, f, he, vav))
// X_old (or NULL; the decomposed update_elbo!): + sum (x - prior_shape)(psi(x) - psi(x_old)) -- after the M-step x - prior_shape IS the statistics entry
// sum_entries val * responsibility, so this is the entries' sum of val * sum_i phi_i (psi(X[i, id]) - psi(X_old[i, id])) without walking them.
// all_rates / lrates_d (or NULL): block 0 also leaves the logarithms of the eight rate vectors for the per-document kernel behind it.
__global__ __launch_bounds__(256) void ctpf_elbo_global_kernel(const float* __restrict__ X, int64_t n_ids, int K, const double* __restrict__ rate,
                                                               double ps, double pr, double* __restrict__ partial, const float* __restrict__ X_old = nullptr,
                                                               const double* __restrict__ all_rates = nullptr, double* __restrict__ lrates_d = nullptr)
{
    __shared__ double red[256];
    if (lrates_d != nullptr && blockIdx.x == 0)
        for (int q = threadIdx.x; q < 8 * K; q += blockDim.x) lrates_d[q] = log(all_rates[q]);
    double s = 0.0;
    const int64_t total = n_ids * K;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(q % K);
        const double x = (double)X[q], r = rate[i], lr = log(r);
        double psi, lg;
        digamma_lgamma_d(x, psi, lg);
        s += (ps - 1.0) * (psi - lr) - pr * x * tmvb_rcp_d(r) + (x - lr + lg + (1.0 - x) * psi);
        if (X_old) s += (x - ps) * (psi - digamma_shift8_d((double)X_old[q]));
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(1024) void ctpf_elbo_final_kernel(const double* __restrict__ doc_val, int64_t M, const double* __restrict__ partial,
                                                               int npartial, double constant, double* __restrict__ out)
{
    __shared__ double red[1024];
    double s = 0.0;
    for (int64_t d = threadIdx.x; d < M; d += 1024) s += doc_val[d];
    for (int q = threadIdx.x; q < npartial; q += 1024) s += partial[q];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] + constant;
}

// out[0] = sum of doc_val[0, M), out[1] = sum of partial[0, npartial) + constant: both parts of update_elbo! behind one launch
__global__ __launch_bounds__(1024) void ctpf_elbo_final2_kernel(const double* __restrict__ doc_val, int64_t M, const double* __restrict__ partial,
                                                                int npartial, double constant, double* __restrict__ out)
{
    __shared__ double red[2][1024];
    double s = 0.0, g = 0.0;
    for (int64_t d = threadIdx.x; d < M; d += 1024) s += doc_val[d];
    for (int q = threadIdx.x; q < npartial; q += 1024) g += partial[q];
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = g;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = red[0][0]; out[1] = red[1][0] + constant; }
}

// ---- the decomposed update_elbo! (round 5), as LDA's and CTM's (tmvb_lda.hip: lda_elbo_doc_kernel) -----------------------------------------------
// After the cancellation above a term entry contributes  c [ sum_i phi_i (lt_i - log bet_i + psi(alef[i, t])) - sum_i phi_i log phi_i ] - lgamma(c + 1)  with
// phi_i = TA_old[t][i] e_i / s_n, log(TA_old e_i) = psi(alef_old[i, t]) + x_i - mx (x = the sweep's softmax argument from gimel_old and the OLD rates, mx its
// shift), and a reader entry the same over the 2 K components of xi.  Summed over a document's entries, with sum_n c_n phi_in + sum_u r_u xi_iu = gimel_i - c
// and sum_u r_u xi_{K+i,u} = zayin_i - g (update_gimel! / update_zayin!, src/CTPF.jl:313, :322):
//     per document   sum_i (gimel_i - c)(psi(gimel_i) - psi(gimel_old_i) - log dalet_i + log dalet_old_i)
//                  + sum_i (zayin_i - g)(psi(zayin_i) - psi(zayin_old_i) - log het_i + log het_old_i - log vav_i + log vav_old_i)
//                  + C_d mx + R_d mab                                       (the shifts the E-step kernel left in CtpfParams::shift; C_d, R_d = sums of counts / ratings)
//     per chunk      sum c log s_n, sum r log s_u                            (the statistics passes' log-normaliser sums, TermStatsParams::logz)
//     global         sum (alef - a)(psi(alef) - psi(alef_old)) + the same for he                          (ctpf_elbo_global_kernel)
//                  - sum_i (log bet_i - log bet_old_i) PC_i - sum_i (log vav_i - log vav_old_i) PA_i      (ctpf_elbo_final_parts_kernel)
//                    with PC_i = sum_d sum_n c phi = rowsum(alef)_i - V a  and  PA_i = sum_d (gimel_di - c) - PC_i  -- the split of gimel - c into its
//                    term and reader halves is only needed summed over the documents, where the M-step has it.
// No entry is walked: the two table kernels and the entry sums of ctpf_elbo_doc_fast_kernel (70 of its 109 us on SYN-CITEU) go away.
// Everything per document that depends on the RATES is linear in per-topic coefficients times sum_d gimel_d / sum_d zayin_d, which the M-step has; what is left
// per (document, topic) depends on gimel / gimel_old / zayin / zayin_old only:
//     R = sum_{d,i} [ (c - 1) psi(gi) + (g - 1) psi(za) + gi + lgamma(gi) + (1 - gi) psi(gi) + za + lgamma(za) + (1 - za) psi(za)
//                     + (gi - c)(psi(gi) - psi(gi_old)) + (za - g)(psi(za) - psi(za_old)) ]  +  sum_d [ doc_const - lg_doc[d] + C_d mx_d + R_d mab_d ]
// so this kernel needs nothing of the M-step and the E-step enqueues it on a stream of its own behind its document kernels, under the statistics pass (as
// lda_elbo_doc_kernel); ctpf_elbo_final_parts_kernel adds the rate terms in closed form.
// Layout: CTPF_ELBO_DPB = 16 documents per block of 128 threads, EIGHT LANES PER DOCUMENT (topics q, q + 8, ...: four fp64 special functions per (document, topic)
// are latency-bound chains -- with four lanes per document and one wave per SIMD the kernel took 40 us on SYN-CITEU); rows staged through LDS with coalesced
// loads.  One fp64 value per block, fixed order.
constexpr int CTPF_ELBO_LPD = 8, CTPF_ELBO_DPB = 128 / CTPF_ELBO_LPD;
__global__ __launch_bounds__(128) void ctpf_elbo_doc_parts_kernel(int K, int64_t M, const float* __restrict__ gimel, const float* __restrict__ gimel_old,
                                                                  const float* __restrict__ zayin, const float* __restrict__ zayin_old,
                                                                  const double* __restrict__ lg_doc, const double* __restrict__ crd /* [M][2] */,
                                                                  const float* __restrict__ shift /* [M][2] */,
                                                                  double hc, double hg, double doc_const, double* __restrict__ block_val)
{
    extern __shared__ __attribute__((aligned(16))) float smf[];   // [4][DPB K]
    __shared__ double red[2];
    constexpr int DPB = CTPF_ELBO_DPB, LPD = CTPF_ELBO_LPD;
    float* sg = smf; float* sgo = sg + DPB * K; float* sz = sg + 2 * DPB * K; float* szo = sg + 3 * DPB * K;
    const int tid = threadIdx.x;
    const int64_t d0 = (int64_t)blockIdx.x * DPB;
    const int nd = (int)min((int64_t)DPB, M - d0);
    {
        const int n = nd * K, n4 = n >> 2;                     // DPB K floats per block: 16-byte aligned
        const float4* a4 = (const float4*)(gimel + d0 * K); const float4* b4 = (const float4*)(gimel_old + d0 * K);
        const float4* c4 = (const float4*)(zayin + d0 * K); const float4* e4 = (const float4*)(zayin_old + d0 * K);
        for (int i = tid; i < n4; i += 128) { ((float4*)sg)[i] = a4[i]; ((float4*)sgo)[i] = b4[i]; ((float4*)sz)[i] = c4[i]; ((float4*)szo)[i] = e4[i]; }
        for (int i = 4 * n4 + tid; i < n; i += 128) { sg[i] = gimel[d0 * K + i]; sgo[i] = gimel_old[d0 * K + i]; sz[i] = zayin[d0 * K + i]; szo[i] = zayin_old[d0 * K + i]; }
    }
    __syncthreads();
    const int dl = tid / LPD, q = tid % LPD;
    double acc = 0.0;
    if (dl < nd) {
        for (int k = q; k < K; k += LPD) {
            const double gi = (double)sg[dl * K + k], za = (double)sz[dl * K + k];
            const double dgo = digamma_shift8_d((double)sgo[dl * K + k]), dzo = digamma_shift8_d((double)szo[dl * K + k]);
            double psi_g, lg_g, psi_z, lg_z;
            digamma_lgamma_d(gi, psi_g, lg_g);
            digamma_lgamma_d(za, psi_z, lg_z);
            acc += (hc - 1.0) * psi_g + (hg - 1.0) * psi_z;                                   // Elogptheta :156, Elogpepsilon :174 (the psi terms)
            acc += (gi + lg_g + (1.0 - gi) * psi_g) + (za + lg_z + (1.0 - za) * psi_z);       // -Elogqtheta, -Elogqepsilon: Gamma entropies without the rates
            acc += (gi - hc) * (psi_g - dgo) + (za - hg) * (psi_z - dzo);                     // the entries' per-document share (see above)
        }
        if (q == 0) {
            const int64_t d = d0 + dl;
            acc += doc_const - lg_doc[d] + crd[2 * d] * (double)shift[2 * d] + crd[2 * d + 1] * (double)shift[2 * d + 1];
        }
    }
    acc = wave_sum_d(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) block_val[blockIdx.x] = red[0] + red[1];
}

// out[0] = sum of the blocks' values + ln 2 * sum logz + the documents' rate terms in closed form (the documents' part),
// out[1] = sum partial + constant - sum_i (dlog bet_i PC_i + dlog vav_i PA_i) (the global part).  With SG = sum_d gimel_d, SZ = sum_d zayin_d:
//   - sum_i [ (SG_i / dalet_i)(rs_he_i / vav_i + rs_alef_i / bet_i + d) + (SZ_i / het_i)(rs_he_i / vav_i + h) ]              :112,:123,:134, :156, :174
//   - M sum_i [ c log dalet_i + g log het_i ]                                        (Elogptheta / Elogpepsilon's and the Gamma entropies' log-rate terms)
//   - sum_i (SG_i - M c)(log dalet_i - log dalet_old_i) - sum_i (SZ_i - M g)(log het_i - log het_old_i + log vav_i - log vav_old_i)
__global__ __launch_bounds__(1024) void ctpf_elbo_final_parts_kernel(const double* __restrict__ block_val, int64_t n_blocks, double Md, const double* __restrict__ partial,
                                                                     int npartial, double constant, const double* __restrict__ logz, int64_t n_logz,
                                                                     const double* __restrict__ rates, const double* __restrict__ lrates_d,
                                                                     const double* __restrict__ rs_alef, const double* __restrict__ rs_he,
                                                                     const double* __restrict__ sum_g, const double* __restrict__ sum_z, int K, double Va,
                                                                     double hc, double hd, double hg, double hh, double* __restrict__ out, int sharded = 0)
{
    // sharded (round 6): the handle holds one document shard of M = Md documents, SG / SZ are the CORPUS' sums (the all-reduced statistics tail).  Every
    // closed-form term is linear in (SG, SZ, Md): the SG / SZ shares go to the global part (identical on every rank, added once), the Md shares stay in the
    // documents' part (the ranks' values add up to M_total's) -- the sum over the ranks of out[0] plus out[1] is the unsharded value.
    __shared__ double red[2][1024];
    double s = 0.0, g = 0.0, lz = 0.0;
    for (int64_t b = threadIdx.x; b < n_blocks; b += 1024) s += block_val[b];
    for (int64_t i = threadIdx.x; i < n_logz; i += 1024) lz += logz[i];
    for (int q = threadIdx.x; q < npartial; q += 1024) g += partial[q];
    for (int i = threadIdx.x; i < K; i += 1024) {
        const double r_bet = tmvb_rcp_d(rates[i]), r_vav = tmvb_rcp_d(rates[K + i]), r_dalet = tmvb_rcp_d(rates[2 * K + i]), r_het = tmvb_rcp_d(rates[3 * K + i]);
        const double l_bet = lrates_d[i], l_vav = lrates_d[K + i], l_dalet = lrates_d[2 * K + i], l_het = lrates_d[3 * K + i];
        const double SG = sum_g[i], SZ = sum_z[i];
        const double dl_dalet = l_dalet - lrates_d[6 * K + i], dl_hv = (l_het - lrates_d[7 * K + i]) + (l_vav - lrates_d[5 * K + i]);
        const double dl_bet = l_bet - lrates_d[4 * K + i], dl_vav = l_vav - lrates_d[5 * K + i];
        const double pc = rs_alef[i] - Va;
        if (sharded) {
            g -= SG * r_dalet * (rs_he[i] * r_vav + rs_alef[i] * r_bet + hd) + SZ * r_het * (rs_he[i] * r_vav + hh);
            g -= SG * dl_dalet + SZ * dl_hv;
            g -= dl_bet * pc + dl_vav * (SG - pc);
            s -= Md * (hc * l_dalet + hg * l_het);
            s += Md * (hc * dl_dalet + hg * dl_hv) + Md * hc * dl_vav;
        } else {
            s -= SG * r_dalet * (rs_he[i] * r_vav + rs_alef[i] * r_bet + hd) + SZ * r_het * (rs_he[i] * r_vav + hh);
            s -= Md * (hc * l_dalet + hg * l_het);
            s -= (SG - Md * hc) * dl_dalet + (SZ - Md * hg) * dl_hv;
            const double pa = (SG - Md * hc) - pc;
            g -= dl_bet * pc + dl_vav * pa;
        }
    }
    red[0][threadIdx.x] = s + 0.6931471805599453 * lz; red[1][threadIdx.x] = g;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = red[0][0]; out[1] = red[1][0] + constant; }
}

// ------------------------------------------------------------------------------ host side
struct tmvb_ctpf {
    tmvb_ctx* ctx = nullptr;
    tmvb_corpus* corp = nullptr;
    int K = 0, KP = 0, nslot = 1;
    int64_t M = 0, V = 0, U = 0;
    bool distributed = false;
    tmvb_comm* comm = nullptr;          // document-sharded train!: the all-reduce of the packed statistics (not owned)
    double hyper[8] = {0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1};   // a..h, src/CTPF.jl:81
    float* d_alef = nullptr; float* d_alef_old = nullptr;          // [V][K] dense
    float* d_he = nullptr; float* d_he_old = nullptr;              // [U][K] dense
    float* d_TA = nullptr; float* d_TH = nullptr;                  // padded tables
    float* d_stats = nullptr; bool own_stats = true;               // alef_stats (K*V) | he_stats (K*U) | sum_gimel (K) | sum_zayin (K)
    double* d_rates = nullptr; float* d_lrates = nullptr;          // [8][K], [4][K]
    double* d_lrates_d = nullptr;                                  // [8][K] log(rates) in fp64 for update_elbo! (filled per call)
    bool rs_fresh = false;                                         // d_rs_alef / d_rs_he are the row sums of the current alef / he
    // decomposed update_elbo! (ctpf_elbo_doc_parts_kernel): TMVB_CTPF_ELBO_PARTS at creation -- 1 (default) the iterations train! will check, 2 every E-step,
    // 0 never.  A collecting E-step leaves the softmax shifts (d_shift) and the statistics passes' log-normaliser sums (d_logz: term chunks, then reader chunks).
    int lds_limit = -1; int elbo_lds_set = 0;                      // the device's per-workgroup LDS limit (read once); dynamic-LDS attribute already set
    int parts_env = 1; bool want_parts = false; bool logz_valid = false; int msteps_after = 0; int elbo_form = 0; bool force_walk = false;
    float* d_shift = nullptr; double* d_logz = nullptr; int64_t n_logz = 0; double* d_crd = nullptr;
    static constexpr int ELBO = 0;                                 // aux[0]: ctpf_elbo_doc_parts_kernel's stream (enqueued by the collecting E-step behind the join of
                                                                   // the document kernels: aux[0] is idle by then.  aux[2] landed on the context stream's own hardware
                                                                   // queue -- rocprofv3: the kernel ran IN FRONT of the statistics pass instead of beside it)
    hipEvent_t ev_docs_done = nullptr, ev_elbo = nullptr; bool elbo_pending = false; int64_t n_elbo_blocks = 0;
    unsigned int* d_mstep_counter = nullptr;                        // last-block-done counter of the fused M-step (zero between launches)
    bool docs_pending = false;                                     // tmvb_ctpf_reduce_docs was asked for and deferred into the fused M-step (K <= 64, one context)
    float* d_gimel = nullptr; float* d_gimel_old = nullptr; float* d_zayin = nullptr; float* d_zayin_old = nullptr;
    float* d_wtok = nullptr; float* d_wrdr = nullptr; float* d_E1 = nullptr; float* d_E2 = nullptr;
    int estride = 0;                                                // row stride of d_E1 / d_E2 (CtpfParams::estride)
    bool e_padded = true;                                           // rows of KP (or 64) floats with zero pads; false: dense K floats (KP / 4 > 64)
    float* d_ts_partial = nullptr; float* d_ts_partial2 = nullptr;   // multi-chunk partials of the term / reader statistics passes
    uint8_t* d_sweeps = nullptr; int32_t* d_doc_order = nullptr;
    double* d_doc_val = nullptr; double* d_elbo_partial = nullptr; double* d_elbo = nullptr;
    // fast update_elbo!: per-call tables exp(psi(X_old)) / psi(X) - psi(X_old) of both shape matrices (allocated at the first call)
    // and the per-document constants sum lgamma(count + 1) + sum lgamma(rating + 1)
    float* d_TAo = nullptr; float* d_DA = nullptr; float* d_THo = nullptr; float* d_DH = nullptr; double* d_lg_doc = nullptr;
    double* d_partial = nullptr; double* d_partial2 = nullptr; double* d_rs_alef = nullptr; double* d_rs_he = nullptr; double* d_sum_g = nullptr; double* d_sum_z = nullptr;
    double elbo = 0.0;
    int* d_topic_of_lane = nullptr;     // register-tile kernel: topic owned by each lane after the reduce-scatter
    int* d_grid_topic_of_lane = nullptr;   // grid-tile kernel (tmvb_gridtile.h): the same for its 16-lane reduce-scatter
    bool grid_path = false;             // KP <= 60: documents of <= 512 terms and <= 512 readers use ctpf_estep_grid_kernel
    bool reg_path = false;              // KP = 4 * odd <= 60: short documents (<= 128 terms, <= 64 readers) use ctpf_estep_reg_kernel
    std::vector<tmvb_bucket> buckets;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool timing = false;               // TMVB_ESTEP_TIMING=1 (read at creation): record the events behind tmvb_ctpf_last_estep_ms
    static constexpr int NAUX = 4;
    hipStream_t aux[NAUX] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[NAUX] = {nullptr, nullptr, nullptr, nullptr};
    int64_t stats_len() const { return (int64_t)K * V + (int64_t)K * U + 2 * K; }
    float* he_stats() const { return d_stats + (size_t)K * V; }
    float* tail() const { return d_stats + (size_t)K * V + (size_t)K * U; }
};

static size_t ctpf_tile_bytes(int rows, int KP) { return ((size_t)rows * KP + 2 * (size_t)KP + 3 * (size_t)rows) * sizeof(float); }

extern "C" int tmvb_ctpf_destroy(tmvb_ctpf* h)
{
    if (!h) return TMVB_OK;
    if (h->ctx) (void)hipSetDevice(h->ctx->device);
    (void)hipFree(h->d_alef); (void)hipFree(h->d_alef_old); (void)hipFree(h->d_he); (void)hipFree(h->d_he_old);
    (void)hipFree(h->d_TA); (void)hipFree(h->d_TH); (void)hipFree(h->d_topic_of_lane); (void)hipFree(h->d_grid_topic_of_lane);
    if (h->own_stats) (void)hipFree(h->d_stats);
    (void)hipFree(h->d_rates); (void)hipFree(h->d_lrates); (void)hipFree(h->d_lrates_d); (void)hipFree(h->d_gimel); (void)hipFree(h->d_gimel_old);
    (void)hipFree(h->d_zayin); (void)hipFree(h->d_zayin_old); (void)hipFree(h->d_wtok); (void)hipFree(h->d_wrdr);
    (void)hipFree(h->d_E1); (void)hipFree(h->d_E2); (void)hipFree(h->d_ts_partial); (void)hipFree(h->d_ts_partial2); (void)hipFree(h->d_sweeps);
    (void)hipFree(h->d_doc_order); (void)hipFree(h->d_partial); (void)hipFree(h->d_partial2); (void)hipFree(h->d_rs_alef); (void)hipFree(h->d_rs_he);
    (void)hipFree(h->d_mstep_counter);
    (void)hipFree(h->d_sum_g); (void)hipFree(h->d_sum_z); (void)hipFree(h->d_doc_val); (void)hipFree(h->d_elbo_partial); (void)hipFree(h->d_elbo);
    (void)hipFree(h->d_TAo); (void)hipFree(h->d_DA); (void)hipFree(h->d_THo); (void)hipFree(h->d_DH); (void)hipFree(h->d_lg_doc);
    (void)hipFree(h->d_shift); (void)hipFree(h->d_logz); (void)hipFree(h->d_crd);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_docs_done) (void)hipEventDestroy(h->ev_docs_done);
    if (h->ev_elbo) (void)hipEventDestroy(h->ev_elbo);
    for (int a = 0; a < tmvb_ctpf::NAUX; ++a) {
        if (h->ev_join[a]) (void)hipEventDestroy(h->ev_join[a]);
        tmvb_release_stream(h->aux[a]); h->aux[a] = nullptr;        // pooled streams stay (tmvb_pool_stream)
    }
    delete h;
    return TMVB_OK;
}

static int ctpf_refresh_tables(tmvb_ctpf* h)
{
    tmvb_ctx* ctx = h->ctx;
    int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ((int64_t)h->KP * h->V + 255) / 256));
    hipLaunchKernelGGL(ctpf_shape_kernel, dim3(nb), dim3(256), 0, ctx->stream, (float*)nullptr, 0.0f, h->d_alef, (float*)nullptr, h->d_TA, h->K, h->KP, h->V, 1);
    TMVB_HIP(hipGetLastError());
    if (h->U > 0) {
        nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ((int64_t)h->KP * h->U + 255) / 256));
        hipLaunchKernelGGL(ctpf_shape_kernel, dim3(nb), dim3(256), 0, ctx->stream, (float*)nullptr, 0.0f, h->d_he, (float*)nullptr, h->d_TH, h->K, h->KP, h->U, 1);
        TMVB_HIP(hipGetLastError());
    }
    return TMVB_OK;
}

extern "C" int tmvb_ctpf_set_state(tmvb_ctpf* h, const double* hyper, const double* alef, const double* he, const double* bet,
                                   const double* vav, const double* dalet, const double* het, const double* gimel,
                                   const double* zayin, const double* elbo);

extern "C" int tmvb_ctpf_create(tmvb_ctx* ctx, tmvb_corpus* corp, int32_t K, tmvb_ctpf** out)
{
    TMVB_REQUIRE(out != nullptr, TMVB_EINVAL, "tmvb_ctpf_create: out is NULL");
    *out = nullptr;
    TMVB_REQUIRE(ctx && corp, TMVB_EINVAL, "tmvb_ctpf_create: NULL context or corpus");
    TMVB_REQUIRE(K > 0, TMVB_EINVAL, "number of topics must be a positive integer.");
    TMVB_REQUIRE(K <= 512, TMVB_EINVAL, "tmvb_ctpf_create: K <= 512 (eight topic slots per lane); got K=%d", K);
    TMVB_HIP(hipSetDevice(ctx->device));
    tmvb_ctpf* h = new tmvb_ctpf();
    tmvb_create_guard<tmvb_ctpf, tmvb_ctpf_destroy> guard{h};      // every early return below destroys h
    h->ctx = ctx; h->corp = corp; h->K = K; h->KP = tmvb_kpad(K); h->nslot = (K + 63) / 64;
    // E rows: padded to KP floats (64 for 32 < KP <= 64: the statistics pass's fast form) while a row fits 64 sixteen-byte chunks,
    // dense K floats beyond (the scalar statistics kernel; round 4: K <= 512)
    h->e_padded = h->KP / 4 <= 64;
    h->estride = !h->e_padded ? K : (h->KP > 32 && h->KP <= 64) ? 64 : h->KP;
    h->M = corp->info.M; h->V = corp->info.V; h->U = corp->info.U;
    const size_t KM = (size_t)K * h->M, KV = (size_t)K * h->V, KU = (size_t)K * h->U;
    int rc;
    if ((rc = tmvb_corpus_term_index(corp)) || (rc = tmvb_corpus_reader_index(corp))) return rc;
    const size_t slots = (size_t)std::max(corp->term_index.n_slots, corp->reader_index.n_slots);
    if ((rc = dmalloc(&h->d_alef, KV)) || (rc = dmalloc(&h->d_alef_old, KV)) || (rc = dmalloc(&h->d_he, KU)) || (rc = dmalloc(&h->d_he_old, KU)) ||
        (rc = dmalloc(&h->d_TA, (size_t)h->KP * h->V + 4)) || (rc = dmalloc(&h->d_TH, (size_t)h->KP * h->U + 4)) ||
        (rc = dmalloc(&h->d_stats, (size_t)h->stats_len())) || (rc = dmalloc(&h->d_rates, 8 * (size_t)K)) || (rc = dmalloc(&h->d_lrates, 4 * (size_t)K)) ||
        (rc = dmalloc(&h->d_gimel, KM)) || (rc = dmalloc(&h->d_gimel_old, KM)) || (rc = dmalloc(&h->d_zayin, KM)) || (rc = dmalloc(&h->d_zayin_old, KM)) ||
        (rc = dmalloc(&h->d_wtok, (size_t)corp->info.nnz)) || (rc = dmalloc(&h->d_wrdr, (size_t)corp->info.nR)) ||
        (rc = dmalloc(&h->d_E1, (size_t)h->estride * h->M + 4)) || (rc = dmalloc(&h->d_E2, (size_t)h->estride * h->M + 4)) ||
        (rc = dmalloc(&h->d_ts_partial, slots * (K + 1))) || (rc = dmalloc(&h->d_ts_partial2, slots * (K + 1))) || (rc = dmalloc(&h->d_sweeps, (size_t)h->M)) || (rc = dmalloc(&h->d_doc_order, (size_t)h->M)) ||
        (rc = dmalloc(&h->d_partial, (size_t)TMVB_REDUCE_BLOCKS * K)) || (rc = dmalloc(&h->d_partial2, (size_t)TMVB_REDUCE_BLOCKS * K)) || (rc = dmalloc(&h->d_rs_alef, K)) || (rc = dmalloc(&h->d_rs_he, K)) ||
        (rc = dmalloc(&h->d_mstep_counter, 1)) ||
        (rc = dmalloc(&h->d_sum_g, K)) || (rc = dmalloc(&h->d_sum_z, K)) || (rc = dmalloc(&h->d_doc_val, (size_t)h->M)) ||
        (rc = dmalloc(&h->d_elbo_partial, 1024)) || (rc = dmalloc(&h->d_elbo, 2))) {
        return rc;
    }
    TMVB_HIP(hipMemset(h->d_mstep_counter, 0, sizeof(unsigned int)));
    TMVB_HIP(hipMemset(h->d_E1, 0, ((size_t)h->estride * h->M + 4) * sizeof(float)));      // the pad columns stay zero for good
    TMVB_HIP(hipMemset(h->d_E2, 0, ((size_t)h->estride * h->M + 4) * sizeof(float)));
    // processing order: first the documents of the LDS-tile kernel by rows (terms + readers), longest first, in LDS
    // buckets on the combined row count; then the register-tile documents (<= 128 terms and <= 64 readers) by tiles
    h->reg_path = (h->KP / 4) <= 15 && ((h->KP / 4) & 1);
    if (h->reg_path) {
        std::vector<int> tol, lot;
        tmvb_reg_lane_maps(h->KP, tol, lot);
        if ((rc = dmalloc(&h->d_topic_of_lane, tol.size()))) return rc;
        TMVB_HIP(hipMemcpy(h->d_topic_of_lane, tol.data(), tol.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    // grid-tile kernel: KP <= 60, both tables addressable by 32-bit byte offsets; TMVB_CTPF_GRID=0 keeps the lane = token tiles
    h->grid_path = h->reg_path && (uint64_t)std::max(h->V, h->U) * (uint64_t)h->KP * 4u < (1ull << 32) &&
                   !(getenv("TMVB_CTPF_GRID") && atoi(getenv("TMVB_CTPF_GRID")) == 0);
    if (h->grid_path) {
        std::vector<int> tol;
        switch (h->KP / 4) {
#define CTPF_GRID_MAP_CASE(LPRV) case LPRV: tmvb_grid_lane_map_fill<LPRV, 0>(tol); break;
            CTPF_GRID_MAP_CASE(1) CTPF_GRID_MAP_CASE(3) CTPF_GRID_MAP_CASE(5) CTPF_GRID_MAP_CASE(7) CTPF_GRID_MAP_CASE(9) CTPF_GRID_MAP_CASE(11)
            CTPF_GRID_MAP_CASE(13) CTPF_GRID_MAP_CASE(15)
#undef CTPF_GRID_MAP_CASE
            default: h->grid_path = false;
        }
        if (h->grid_path) {
            if ((rc = dmalloc(&h->d_grid_topic_of_lane, tol.size()))) return rc;
            TMVB_HIP(hipMemcpy(h->d_grid_topic_of_lane, tol.data(), tol.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    std::vector<int64_t> len((size_t)h->M);
    std::vector<int32_t> order, reg2, reg1;
    // grid-tile classes (ctpf_grid_class): key = waves * 10000 + npt * 100 + npr, longest class first
    std::vector<std::pair<int, int32_t>> gdocs;
    for (int64_t d = 0; d < h->M; ++d) {
        len[d] = corp->h_doc_len[d] + corp->h_rdr_len[d];
        int npt = 0, npr = 0, wv = 1;
        if (h->grid_path && ctpf_grid_class(corp->h_doc_len[d], corp->h_rdr_len[d], &npt, &npr, &wv)) {
            gdocs.emplace_back(wv * 10000 + npt * 100 + npr, (int32_t)d);
            continue;
        }
        const bool reg = h->reg_path && !h->grid_path && corp->h_doc_len[d] <= 128 && corp->h_rdr_len[d] <= 64;
        if (!reg) order.push_back((int32_t)d);
        else if (corp->h_doc_len[d] > 64) reg2.push_back((int32_t)d);
        else reg1.push_back((int32_t)d);
    }
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return len[x] > len[y]; });
    const int64_t n_lds = (int64_t)order.size();
    tmvb_build_lds_buckets(len, order, n_lds, h->KP, -1, 3, h->buckets, h->reg_path ? TMVB_BIG_TILE_BYTES : TMVB_MAX_TILE_BYTES);
    auto by_terms = [&](int32_t x, int32_t y) { return corp->h_doc_len[x] > corp->h_doc_len[y]; };
    std::stable_sort(reg2.begin(), reg2.end(), by_terms);
    std::stable_sort(reg1.begin(), reg1.end(), by_terms);
    if (!reg2.empty()) h->buckets.push_back({(int64_t)order.size(), (int64_t)reg2.size(), 0, 2});
    order.insert(order.end(), reg2.begin(), reg2.end());
    if (!reg1.empty()) h->buckets.push_back({(int64_t)order.size(), (int64_t)reg1.size(), 0, 1});
    order.insert(order.end(), reg1.begin(), reg1.end());
    if (!gdocs.empty()) {
        std::stable_sort(gdocs.begin(), gdocs.end(), [&](const std::pair<int, int32_t>& x, const std::pair<int, int32_t>& y) {
            return x.first != y.first ? x.first > y.first : len[x.second] > len[y.second]; });
        size_t q = 0;
        while (q < gdocs.size()) {
            size_t e = q;
            while (e < gdocs.size() && gdocs[e].first == gdocs[q].first) ++e;
            tmvb_bucket b{(int64_t)order.size(), (int64_t)(e - q), 0, 1};
            b.waves = gdocs[q].first / 10000; b.grid_np = (gdocs[q].first / 100) % 100; b.grid_np2 = gdocs[q].first % 100;
            h->buckets.push_back(b);
            for (size_t u = q; u < e; ++u) order.push_back(gdocs[u].second);
            q = e;
        }
    }
    if (h->M) TMVB_HIP(hipMemcpyAsync(h->d_doc_order, order.data(), (size_t)h->M * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_stats, 0, (size_t)h->stats_len() * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_sweeps, 0, std::max<size_t>((size_t)h->M, 1), ctx->stream));
    // (the two timing events are default events -- time stamps and a system-scope release each -- on the stream the whole iteration runs on:
    //  recorded unconditionally they cost every 0.15 ms iteration two barrier packets; round 4, as LDA has it)
    { const char* t = getenv("TMVB_ESTEP_TIMING"); h->timing = t && atoi(t) != 0; }
    { const char* e = getenv("TMVB_CTPF_ELBO_PARTS"); h->parts_env = e ? atoi(e) : 1; }
    TMVB_HIP(hipEventCreate(&h->ev0));
    TMVB_HIP(hipEventCreate(&h->ev1));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_fork, tmvb_event_flags()));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_docs_done, tmvb_event_flags()));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_elbo, tmvb_event_flags()));
    for (int a = 0; a < tmvb_ctpf::NAUX; ++a) {
        // aux[1] carries the few multi-wave (long) documents next to the chain's one big launch: at the default priority its 512-thread
        // workgroups find no CU with eight free wave slots until the chain's launch drains and then add their whole run time to the
        // E-step; on a high-priority queue they are placed first (TMVB_CTPF_LONG_PRIO=0: default priority)
        static const bool long_prio = [] { const char* e = getenv("TMVB_CTPF_LONG_PRIO"); return !(e && atoi(e) == 0); }();
        h->aux[a] = tmvb_pool_stream(ctx->device, 1 + a, a == 1 && long_prio);
        TMVB_REQUIRE(h->aux[a] != nullptr, TMVB_EHIP, "hipStreamCreate failed");
        TMVB_HIP(hipEventCreateWithFlags(&h->ev_join[a], tmvb_event_flags()));
    }
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    // constructor state src/CTPF.jl:81-100: he = 1, rates = 1, gimel = zayin = 1; alef is drawn with Julia's RNG (:83): 1 here
    std::vector<double> alef(KV, 1.0), he(KU, 1.0), ones(K, 1.0), gz(KM, 1.0);
    rc = tmvb_ctpf_set_state(h, nullptr, alef.data(), he.data(), ones.data(), ones.data(), ones.data(), ones.data(), gz.data(), gz.data(), nullptr);
    if (rc) return rc;
    guard.release();
    *out = h;
    return TMVB_OK;
}

static int positive_finite(const double* x, size_t n) { for (size_t q = 0; q < n; ++q) if (!(x[q] > 0.0) || !std::isfinite(x[q])) return 0; return 1; }

extern "C" int tmvb_ctpf_set_state(tmvb_ctpf* h, const double* hyper, const double* alef, const double* he, const double* bet,
                                   const double* vav, const double* dalet, const double* het, const double* gimel,
                                   const double* zayin, const double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctpf_set_state: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const size_t K = h->K, KM = K * (size_t)h->M, KV = K * (size_t)h->V, KU = K * (size_t)h->U;
    int rc;
    if (hyper) {
        const char* nm = "abcdefgh";
        for (int q = 0; q < 8; ++q) TMVB_REQUIRE(hyper[q] > 0.0, TMVB_ESHAPE, "%c must be positive.", nm[q]);   // src/modelutils.jl:188-195
        memcpy(h->hyper, hyper, sizeof(h->hyper));
    }
    if (alef || he) h->rs_fresh = false;
    h->logz_valid = false;
    if (h->elbo_pending) { TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_elbo, 0)); h->elbo_pending = false; }   // (it reads gimel / zayin)
    if (alef) {
        TMVB_REQUIRE(positive_finite(alef, KV), TMVB_ENONFINITE, "alef must be positive.");
        if ((rc = upload_f32(ctx, h->d_alef, alef, KV)) || (rc = upload_f32(ctx, h->d_alef_old, alef, KV))) return rc;
    }
    if (he) {
        TMVB_REQUIRE(positive_finite(he, KU), TMVB_ENONFINITE, "he must be positive.");
        if ((rc = upload_f32(ctx, h->d_he, he, KU)) || (rc = upload_f32(ctx, h->d_he_old, he, KU))) return rc;
    }
    const double* rv[4] = {bet, vav, dalet, het};
    const char* rn[4] = {"bet", "vav", "dalet", "het"};
    for (int r = 0; r < 4; ++r) {
        if (!rv[r]) continue;
        TMVB_REQUIRE(positive_finite(rv[r], K), TMVB_ENONFINITE, "%s must be positive.", rn[r]);
        std::vector<float> lg(K);
        for (size_t i = 0; i < K; ++i) lg[i] = (float)std::log(rv[r][i]);
#ifdef TMVB_MUTANT_CTPF_LOG_BET
        if (r == 1 && bet) for (size_t i = 0; i < K; ++i) lg[i] = (float)std::log(bet[i]);
#endif
        TMVB_HIP(hipMemcpyAsync(h->d_rates + r * K, rv[r], K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        TMVB_HIP(hipMemcpyAsync(h->d_rates + (4 + r) * K, rv[r], K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        TMVB_HIP(hipMemcpyAsync(h->d_lrates + r * K, lg.data(), K * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        TMVB_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (gimel) {
        TMVB_REQUIRE(positive_finite(gimel, KM), TMVB_ENONFINITE, "gimel must be positive.");
        if ((rc = upload_f32(ctx, h->d_gimel, gimel, KM)) || (rc = upload_f32(ctx, h->d_gimel_old, gimel, KM))) return rc;
    }
    if (zayin) {
        TMVB_REQUIRE(positive_finite(zayin, KM), TMVB_ENONFINITE, "zayin must be positive.");
        if ((rc = upload_f32(ctx, h->d_zayin, zayin, KM)) || (rc = upload_f32(ctx, h->d_zayin_old, zayin, KM))) return rc;
    }
    if (elbo) h->elbo = *elbo;
    if (alef || he) { if ((rc = ctpf_refresh_tables(h))) return rc; }
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    return TMVB_OK;
}

// The *_old fields of update_buffer! (src/modelutils.jl:474-493).  update_elbo! rebuilds phi / xi from them
// (src/CTPF.jl:239-240), so a model that is uploaded again after training must bring them along: tmvb_ctpf_set_state sets
// every *_old equal to the current value (the constructor state), this call overrides them.  NULL = leave unchanged.
extern "C" int tmvb_ctpf_set_state_old(tmvb_ctpf* h, const double* alef_old, const double* he_old, const double* bet_old,
                                       const double* vav_old, const double* dalet_old, const double* het_old,
                                       const double* gimel_old, const double* zayin_old)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctpf_set_state_old: handle is NULL");
    h->logz_valid = false;
    tmvb_ctx* ctx = h->ctx;
    if (h->elbo_pending) { TMVB_HIP(hipSetDevice(ctx->device)); TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_elbo, 0)); h->elbo_pending = false; }
    TMVB_HIP(hipSetDevice(ctx->device));
    const size_t K = h->K, KM = K * (size_t)h->M, KV = K * (size_t)h->V, KU = K * (size_t)h->U;
    int rc;
    if (alef_old) {
        TMVB_REQUIRE(positive_finite(alef_old, KV), TMVB_ENONFINITE, "alef_old must be positive.");
        if ((rc = upload_f32(ctx, h->d_alef_old, alef_old, KV))) return rc;
    }
    if (he_old) {
        TMVB_REQUIRE(positive_finite(he_old, KU), TMVB_ENONFINITE, "he_old must be positive.");
        if ((rc = upload_f32(ctx, h->d_he_old, he_old, KU))) return rc;
    }
    const double* rv[4] = {bet_old, vav_old, dalet_old, het_old};
    const char* rn[4] = {"bet_old", "vav_old", "dalet_old", "het_old"};
    for (int r = 0; r < 4; ++r) {
        if (!rv[r]) continue;
        TMVB_REQUIRE(positive_finite(rv[r], K), TMVB_ENONFINITE, "%s must be positive.", rn[r]);
        TMVB_HIP(hipMemcpyAsync(h->d_rates + (4 + r) * K, rv[r], K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        TMVB_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (gimel_old) {
        TMVB_REQUIRE(positive_finite(gimel_old, KM), TMVB_ENONFINITE, "gimel_old must be positive.");
        if ((rc = upload_f32(ctx, h->d_gimel_old, gimel_old, KM))) return rc;
    }
    if (zayin_old) {
        TMVB_REQUIRE(positive_finite(zayin_old, KM), TMVB_ENONFINITE, "zayin_old must be positive.");
        if ((rc = upload_f32(ctx, h->d_zayin_old, zayin_old, KM))) return rc;
    }
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_ctpf_get_state(tmvb_ctpf* h, double* alef, double* alef_old, double* he, double* he_old,
                                   double* rates /* [8][K]: bet,vav,dalet,het,bet_old,vav_old,dalet_old,het_old */,
                                   double* gimel, double* gimel_old, double* zayin, double* zayin_old, double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctpf_get_state: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const size_t K = h->K, KM = K * (size_t)h->M, KV = K * (size_t)h->V, KU = K * (size_t)h->U;
    int rc;
    if (alef && (rc = download_f32(ctx, alef, h->d_alef, KV))) return rc;
    if (alef_old && (rc = download_f32(ctx, alef_old, h->d_alef_old, KV))) return rc;
    if (he && KU && (rc = download_f32(ctx, he, h->d_he, KU))) return rc;
    if (he_old && KU && (rc = download_f32(ctx, he_old, h->d_he_old, KU))) return rc;
    if (rates) {
        TMVB_HIP(hipMemcpyAsync(rates, h->d_rates, 8 * K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        TMVB_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (gimel && (rc = download_f32(ctx, gimel, h->d_gimel, KM))) return rc;
    if (gimel_old && (rc = download_f32(ctx, gimel_old, h->d_gimel_old, KM))) return rc;
    if (zayin && (rc = download_f32(ctx, zayin, h->d_zayin, KM))) return rc;
    if (zayin_old && (rc = download_f32(ctx, zayin_old, h->d_zayin_old, KM))) return rc;
    if (elbo) *elbo = h->elbo;
    return TMVB_OK;
}

// the corpus constants of update_elbo!: sum lgamma(count + 1) + sum lgamma(rating + 1), C_d, R_d per document (first use)
static int ctpf_elbo_consts(tmvb_ctpf* h)
{
    if (h->d_lg_doc) return TMVB_OK;
    tmvb_ctx* ctx = h->ctx;
    int rc;
    // (a shard without documents -- sharded handles take the decomposed form whatever their shard holds -- still needs the rate logarithms: one element each)
    if ((rc = dmalloc(&h->d_lrates_d, 8 * (size_t)h->K)) || (rc = dmalloc(&h->d_lg_doc, (size_t)std::max<int64_t>(h->M, 1))) ||
        (rc = dmalloc(&h->d_crd, 2 * (size_t)std::max<int64_t>(h->M, 1)))) return rc;
    if (h->M == 0) return TMVB_OK;
    std::vector<double> lg((size_t)h->M, 0.0), crd(2 * (size_t)h->M, 0.0);
    const tmvb_corpus* c = h->corp;
    for (int64_t d = 0; d < h->M; ++d) {
        double v = 0.0, cd = 0.0, rd = 0.0;
        for (int64_t q = c->h_doc_ptr[d]; q < c->h_doc_ptr[d + 1]; ++q) { v += std::lgamma((double)c->h_counts[q] + 1.0); cd += (double)c->h_counts[q]; }
        for (int64_t q = c->h_rdr_ptr[d]; q < c->h_rdr_ptr[d + 1]; ++q) { v += std::lgamma((double)c->h_ratings[q] + 1.0); rd += (double)c->h_ratings[q]; }
        lg[(size_t)d] = v; crd[2 * (size_t)d] = cd; crd[2 * (size_t)d + 1] = rd;
    }
    TMVB_HIP(hipMemcpyAsync(h->d_lg_doc, lg.data(), lg.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TMVB_HIP(hipMemcpyAsync(h->d_crd, crd.data(), crd.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_ctpf_estep(tmvb_ctpf* h, int32_t viter, double vtol)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctpf_estep: handle is NULL");
    TMVB_REQUIRE(viter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative.");     // src/gpuCTPF.jl:680
    TMVB_REQUIRE(vtol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");      // src/gpuCTPF.jl:679
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    CtpfParams p;
    p.K = h->K; p.KP = h->KP; p.LPR = h->KP / 4; p.lpr_magic = (unsigned)(0x100000000ull / (unsigned)p.LPR) + 1u;
    p.doc_ptr = h->corp->d_doc_ptr; p.terms = h->corp->d_terms; p.counts = h->corp->d_counts;
    p.rdr_ptr = h->corp->d_rdr_ptr; p.readers = h->corp->d_readers; p.ratings = h->corp->d_ratings;
    p.doc_order = h->d_doc_order; p.tok_inv = h->corp->term_index.d_inv; p.rdr_inv = h->corp->reader_index.d_inv;
    p.TA = h->d_TA; p.TH = h->d_TH; p.lrates = h->d_lrates; p.hc = (float)h->hyper[2]; p.hg = (float)h->hyper[6];
    p.gimel = h->d_gimel; p.gimel_old = h->d_gimel_old; p.zayin = h->d_zayin; p.zayin_old = h->d_zayin_old;
    p.wtok = h->d_wtok; p.wrdr = h->d_wrdr; p.E1 = h->d_E1; p.E2 = h->d_E2; p.estride = h->estride; p.sweeps = h->d_sweeps;
    p.viter = viter; p.vtol = (float)vtol;
    p.store_w = tmvb_termstats_recomputes(h->KP, h->e_padded) ? 0 : 1;
    // decomposed update_elbo!: this iteration will be checked -- the document kernels leave their softmax shifts, the statistics passes the log-normaliser sums
    // ctpf_elbo_doc_parts_kernel stages 4 x CTPF_ELBO_DPB rows of K floats in dynamic LDS; a device whose per-workgroup limit is below that (K > 256 on a
    // 64 KB part; gfx950: 160 KB) does not collect, and update_elbo! takes the table form (round-5 advice)
    if (h->lds_limit < 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 64 * 1024; }
        h->lds_limit = v;
    }
    // (round 6: sharded handles collect too -- the documents' part adds up over the shards, the closed-form rate terms go to the global part; which form a
    //  handle takes must not depend on its shard: a shard WITHOUT documents has nothing to collect and takes the decomposed form with empty parts)
    const bool would_collect = (h->parts_env == 2 || (h->parts_env != 0 && h->want_parts)) && p.store_w == 0 && viter > 0 &&
                               (size_t)4 * CTPF_ELBO_DPB * h->K * sizeof(float) <= (size_t)h->lds_limit;
    const bool collect = would_collect && h->M > 0;
    h->logz_valid = false; h->msteps_after = 0;
    if (would_collect && h->M == 0 && h->distributed) { h->logz_valid = true; h->n_logz = 0; h->n_elbo_blocks = 0; }
    const int64_t nct = h->corp->term_index.n_chunks, ncr = h->U > 0 ? h->corp->reader_index.n_chunks : 0;
    if (h->elbo_pending) {                                   // ctpf_elbo_doc_parts_kernel of the last collecting E-step still reads gimel / zayin
        TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_elbo, 0));
        h->elbo_pending = false;
    }
    if (collect) {
        { int crc = ctpf_elbo_consts(h); if (crc) return crc; }
        if (!h->d_shift) { int arc = dmalloc(&h->d_shift, 2 * (size_t)h->M); if (arc) return arc; }
        if (!h->d_logz) { int arc = dmalloc(&h->d_logz, (size_t)std::max<int64_t>(nct + ncr, 1)); if (arc) return arc; }
        p.shift = h->d_shift;
    }
    if (h->timing) TMVB_HIP(hipEventRecord(h->ev0, ctx->stream));
    // stream plan as in tmvb_lda_estep for one statistics pass: the register-tile buckets run back to back on the
    // context's stream (the critical chain document kernels -> statistics -> M-step pays kernel boundaries, not ~20 us
    // cross-stream hops: the whole iteration is 0.3 ms), the LDS-tile buckets (long documents) on aux[1]
    const int nb = (int)h->buckets.size();
    hipStream_t chain_st = ctx->stream;                          // only used when h->reg_path
    TMVB_HIP(hipEventRecord(h->ev_fork, ctx->stream));
    for (int a = 0; a < 2; ++a) TMVB_HIP(hipStreamWaitEvent(h->aux[a], h->ev_fork, 0));
    bool aux0_used = !h->reg_path;
    for (int bi = 0; bi < nb; ++bi) {
        const tmvb_bucket& b = h->buckets[bi];
        dim3 grid((unsigned)b.count), block(64);
        if (b.grid_np > 0) {
            // the multi-wave (long) documents on aux[1] next to the chain, the single-wave classes on the context's stream -- as ONE
            // mixed-class launch unless TMVB_CTPF_GRID_ANY=0 (one kernel tail per iteration; the price is the widest body's registers).
            // (Measured and dropped: everything in one launch of four-wave workgroups, long documents first and four single-wave
            //  documents per later workgroup: 0.256 ms per iteration against 0.193 ms.)
            static const bool gany = [] { const char* e = getenv("TMVB_CTPF_GRID_ANY"); return !(e && atoi(e) == 0); }();
            hipStream_t st = b.waves > 1 ? h->aux[1] : chain_st;
            tmvb_bucket bb = b;
            // the (3, 3) four-wave bucket and the (2, 4) one behind it as one launch (ctpf_estep_grid_long2_kernel); TMVB_CTPF_LONG_MERGE=0: two
            const char* elm = getenv("TMVB_CTPF_LONG_MERGE");
            if (b.waves == 4 && b.grid_np == 3 && b.grid_np2 == 3 && bi + 1 < nb && h->buckets[bi + 1].waves == 4 && h->buckets[bi + 1].grid_np == 2 &&
                h->buckets[bi + 1].grid_np2 == 4 && h->buckets[bi + 1].first == b.first + b.count && b.count < (1 << 30) && !(elm && atoi(elm) == 0)) {
                bb.grid_np = 96; bb.grid_np2 = (int)b.count; bb.count = b.count + h->buckets[bi + 1].count;
                grid = dim3((unsigned)bb.count);
                ++bi;
            }
            // (round 4) the single-wave classes as TWO launches by register need (ctpf_estep_grid_narrow_kernel / _wide_kernel) unless
            // TMVB_CTPF_SPLIT=0 (read per call: the tests run both)
            const char* es = getenv("TMVB_CTPF_SPLIT");
            const bool split = !(es && atoi(es) == 0);
            if (gany && b.waves == 1) {
                const bool wide = b.grid_np >= 4;
                int64_t cnt = b.count;
                while (bi + 1 < nb && h->buckets[bi + 1].grid_np > 0 && h->buckets[bi + 1].waves == 1 &&
                       h->buckets[bi + 1].first == b.first + cnt && (!split || (h->buckets[bi + 1].grid_np >= 4) == wide)) { cnt += h->buckets[bi + 1].count; ++bi; }
                if (split) { bb.count = cnt; bb.grid_np = wide ? 98 : 97; grid = dim3((unsigned)cnt); if (wide) { st = h->aux[0]; aux0_used = true; } }
                else if (cnt > b.count) { bb.count = cnt; bb.grid_np = 99; grid = dim3((unsigned)cnt); }
            }
            switch (p.LPR) {
#define CTPF_GRID_CASE(LPRV) case LPRV: ctpf_launch_grid<LPRV>(bb, grid, st, p, b.first, h->d_grid_topic_of_lane); break;
                CTPF_GRID_CASE(1) CTPF_GRID_CASE(3) CTPF_GRID_CASE(5) CTPF_GRID_CASE(7) CTPF_GRID_CASE(9) CTPF_GRID_CASE(11) CTPF_GRID_CASE(13)
                CTPF_GRID_CASE(15)
#undef CTPF_GRID_CASE
                default: TMVB_REQUIRE(false, TMVB_EINVAL, "tmvb_ctpf_estep: no grid-tile kernel for KP=%d", h->KP);
            }
            TMVB_HIP(hipGetLastError());
            continue;
        }
        if (b.reg_tiles > 0) {
            hipStream_t st = chain_st;
            // two adjacent register-tile buckets (T = 2 then T = 1 in processing order): one mixed-tile launch
            static const bool any_env = [] { const char* e = getenv("TMVB_CTPF_REG_ANY"); return !(e && atoi(e) == 0); }();
            int tiles = b.reg_tiles;
            if (any_env && bi + 1 < nb && h->buckets[bi + 1].reg_tiles > 0 && h->buckets[bi + 1].first == b.first + b.count) {
                grid = dim3((unsigned)(b.count + h->buckets[bi + 1].count));
                tiles = 99;
                ++bi;
            }
            switch (p.LPR) {
#define CTPF_REG_CASE(LPRV) case LPRV: ctpf_launch_reg<LPRV>(tiles, grid, st, p, b.first, h->d_topic_of_lane); break;
                CTPF_REG_CASE(1) CTPF_REG_CASE(3) CTPF_REG_CASE(5) CTPF_REG_CASE(7) CTPF_REG_CASE(9) CTPF_REG_CASE(11) CTPF_REG_CASE(13)
                CTPF_REG_CASE(15)
#undef CTPF_REG_CASE
                default: TMVB_REQUIRE(false, TMVB_EINVAL, "tmvb_ctpf_estep: no register-tile kernel for KP=%d", h->KP);
            }
            TMVB_HIP(hipGetLastError());
            continue;
        }
        hipStream_t st = h->reg_path ? h->aux[1] : h->aux[(bi & 1) ^ 1];
        const size_t lds = ctpf_tile_bytes(b.tile_rows, h->KP);
        auto launch = [&](auto kern) -> int {
            if (lds > TMVB_MAX_TILE_BYTES)                // long documents: the whole row set resident in up to 156 KiB of LDS
                TMVB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, grid, block, lds, st, p, b.first, b.tile_rows);
            return TMVB_OK;
        };
        int lrc = (h->nslot > 4) ? launch(ctpf_estep_kernel<0, 8>) : (h->nslot > 2) ? launch(ctpf_estep_kernel<0, 4>)
                  : (h->nslot == 2) ? ((p.LPR == 25) ? launch(ctpf_estep_kernel<25, 2>) : launch(ctpf_estep_kernel<0, 2>))
                  : (p.LPR == 13) ? launch(ctpf_estep_kernel<13, 1>) : (p.LPR == 3) ? launch(ctpf_estep_kernel<3, 1>) : launch(ctpf_estep_kernel<0, 1>);
        if (lrc) return lrc;
        TMVB_HIP(hipGetLastError());
    }
    for (int a = 0; a < 2; ++a) {
        if (a == 0 && !aux0_used) continue;                   // aux[0] carried nothing in this plan
        TMVB_HIP(hipEventRecord(h->ev_join[a], h->aux[a]));
        TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_join[a], 0));
    }
    h->n_elbo_blocks = 0;
    if (collect) {                                           // update_elbo!'s per-document sums: everything they need is there now (every document kernel joined)
        const double* hy = h->hyper;
        const unsigned nblk = (unsigned)((h->M + CTPF_ELBO_DPB - 1) / CTPF_ELBO_DPB);
        const size_t lds = (size_t)4 * CTPF_ELBO_DPB * h->K * sizeof(float);
        hipStream_t es = h->aux[tmvb_ctpf::ELBO];
        TMVB_HIP(hipEventRecord(h->ev_docs_done, ctx->stream));
        TMVB_HIP(hipStreamWaitEvent(es, h->ev_docs_done, 0));
        if (lds > 48 * 1024 && (int)lds > h->elbo_lds_set) {
            TMVB_HIP(hipFuncSetAttribute((const void*)ctpf_elbo_doc_parts_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            h->elbo_lds_set = (int)lds;
        }
        hipLaunchKernelGGL(ctpf_elbo_doc_parts_kernel, dim3(nblk), dim3(128), lds, es, h->K, h->M, (const float*)h->d_gimel, (const float*)h->d_gimel_old,
                           (const float*)h->d_zayin, (const float*)h->d_zayin_old, (const double*)h->d_lg_doc, (const double*)h->d_crd, (const float*)h->d_shift,
                           hy[2], hy[6], (double)h->K * (hy[2] * std::log(hy[3]) - std::lgamma(hy[2])) + (double)h->K * (hy[6] * std::log(hy[7]) - std::lgamma(hy[6])),
                           h->d_doc_val);
        TMVB_HIP(hipGetLastError());
        TMVB_HIP(hipEventRecord(h->ev_elbo, es));
        h->elbo_pending = true; h->n_elbo_blocks = nblk;
    }
    // update_alef!(model, d) / update_he!(model, d) (src/CTPF.jl:259-262, :274-277) as gather-side statistics
    TermStatsParams tp;
    tp.K = h->K; tp.tstride = h->KP; tp.ostride = h->K; tp.eps = 0.0f; tp.base = 0.0f; tp.keps = 0.0f; tp.partial = h->d_ts_partial;
    tp.w = h->d_wtok; tp.E = h->d_E1; tp.T = h->d_TA; tp.out = h->d_stats; tp.estride = h->estride;
    // the two passes are independent (alef / he statistics): one launch with the pass as blockIdx.y (recompute variant), else the
    // reader pass on aux[0] under the term pass
    TermStatsParams tr = tp;
    tr.w = h->d_wrdr; tr.E = h->d_E2; tr.T = h->d_TH; tr.out = h->he_stats(); tr.partial = h->d_ts_partial2;
    if (collect) { tp.logz = h->d_logz; tr.logz = h->d_logz + nct; }
    static const bool fuse_env = [] { const char* e = getenv("TMVB_CTPF_FUSE_STATS"); return !(e && atoi(e) == 0); }();
    int rc = TMVB_EINVAL;
    if (fuse_env && h->U > 0 && tmvb_termstats_recomputes(h->KP, h->e_padded))
        rc = tmvb_launch_termstats2(ctx, h->nslot, h->KP, h->corp->term_index, tp, h->corp->reader_index, tr);
    if (rc == TMVB_EINVAL) {
        TMVB_HIP(hipEventRecord(h->ev_fork, ctx->stream));
        rc = tmvb_launch_termstats(ctx, h->nslot, h->KP, h->e_padded, h->corp->term_index, tp);
        if (rc) return rc;
        if (h->U > 0) {
            TMVB_HIP(hipStreamWaitEvent(h->aux[0], h->ev_fork, 0));
            if ((rc = tmvb_launch_termstats(ctx, h->nslot, h->KP, h->e_padded, h->corp->reader_index, tr, h->aux[0]))) return rc;
            TMVB_HIP(hipEventRecord(h->ev_join[0], h->aux[0]));
            TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_join[0], 0));
        }
    } else if (rc) {
        return rc;
    }
    h->logz_valid = collect || (would_collect && h->M == 0 && h->distributed); h->n_logz = collect ? nct + ncr : 0;
    if (h->timing) TMVB_HIP(hipEventRecord(h->ev1, ctx->stream));
    h->timed = true;
    return TMVB_OK;
}

// sum_d gimel_d, sum_d zayin_d into the statistics tail (src/CTPF.jl:283, :290)
static int ctpf_reduce_docs_on(tmvb_ctpf* h, hipStream_t st)
{
    if (h->K > 64) {           // the paired kernel holds one topic per lane
        int rc = tmvb_colsum(h->ctx, h->nslot, h->K, h->d_gimel, h->M, h->d_partial, h->d_sum_g, h->tail(), st);
        return rc ? rc : tmvb_colsum(h->ctx, h->nslot, h->K, h->d_zayin, h->M, h->d_partial2, h->d_sum_z, h->tail() + h->K, st);
    }
    return tmvb_colsum2(h->ctx, h->K, {h->d_gimel, h->M, h->d_partial, h->d_sum_g, h->tail()},
                        {h->d_zayin, h->M, h->d_partial2, h->d_sum_z, h->tail() + h->K}, st);
}

static bool ctpf_mstep_fused(const tmvb_ctpf* h)
{
    static const bool env = [] { const char* e = getenv("TMVB_CTPF_FUSED_MSTEP"); return !(e && atoi(e) == 0); }();
    return env && h->K <= 64;
}

// the document sums a deferred tmvb_ctpf_reduce_docs still owes (anything that hands the statistics tail out needs them now)
static int ctpf_flush_docs(tmvb_ctpf* h)
{
    if (!h->docs_pending) return TMVB_OK;
    h->docs_pending = false;
    return ctpf_reduce_docs_on(h, h->ctx->stream);
}

extern "C" int tmvb_ctpf_reduce_docs(tmvb_ctpf* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctpf_reduce_docs: handle is NULL");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    // one context, K <= 64: the sums ride in the fused M-step's first launch (tmvb_ctpf_mstep) instead of two launches of their own;
    // a sharded handle needs them in the statistics tail before its all-reduce
    if (!h->distributed && ctpf_mstep_fused(h)) { h->docs_pending = true; return TMVB_OK; }
    h->docs_pending = false;
    return ctpf_reduce_docs_on(h, h->ctx->stream);
}

extern "C" int tmvb_ctpf_stats(tmvb_ctpf* h, void** dev_ptr, int64_t* n_f32)
{
    TMVB_REQUIRE(h && dev_ptr && n_f32, TMVB_EINVAL, "tmvb_ctpf_stats: NULL argument");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    { int frc = ctpf_flush_docs(h); if (frc) return frc; }
    *dev_ptr = h->d_stats;
    *n_f32 = h->stats_len();
    return TMVB_OK;
}

extern "C" int tmvb_ctpf_bind_stats(tmvb_ctpf* h, void* dev_ptr, int64_t n_f32)
{
    TMVB_REQUIRE(h && dev_ptr, TMVB_EINVAL, "tmvb_ctpf_bind_stats: NULL argument");
    TMVB_REQUIRE(n_f32 >= h->stats_len(), TMVB_ESHAPE, "tmvb_ctpf_bind_stats: buffer holds %lld floats, need %lld", (long long)n_f32, (long long)h->stats_len());
    TMVB_HIP(hipSetDevice(h->ctx->device));
    { int frc = ctpf_flush_docs(h); if (frc) return frc; }
    TMVB_HIP(hipMemcpyAsync(dev_ptr, h->d_stats, (size_t)h->stats_len() * sizeof(float), hipMemcpyDeviceToDevice, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    if (h->own_stats) (void)hipFree(h->d_stats);
    h->d_stats = (float*)dev_ptr;
    h->own_stats = false;
    return TMVB_OK;
}

extern "C" int tmvb_ctpf_set_distributed(tmvb_ctpf* h, int32_t distributed)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctpf_set_distributed: handle is NULL");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    { int frc = ctpf_flush_docs(h); if (frc) return frc; }
    h->distributed = distributed != 0;
    return TMVB_OK;
}

// update_he!, update_alef!, update_dalet!, update_het!, update_bet!, update_vav! in the reference's order
// (src/CTPF.jl:366-371); the gimel / zayin sums come from the (all-reduced) statistics tail when distributed.
extern "C" int tmvb_ctpf_mstep(tmvb_ctpf* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctpf_mstep: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    ++h->msteps_after;                                    // (the decomposed update_elbo! wants exactly one M-step behind the collecting E-step)
    int rc;
    if (h->distributed) {     // tail -> fp64 sums
        h->docs_pending = false;
        if ((rc = tmvb_colsum(ctx, h->nslot, h->K, h->tail(), 1, h->d_partial, h->d_sum_g, nullptr))) return rc;
        if ((rc = tmvb_colsum(ctx, h->nslot, h->K, h->tail() + h->K, 1, h->d_partial, h->d_sum_z, nullptr))) return rc;
    }
    if (ctpf_mstep_fused(h)) {
        constexpr int NB = 64;
        static_assert(4 * NB <= TMVB_REDUCE_BLOCKS, "partial buffer");
        const int with_docs = h->docs_pending ? 1 : 0;
        h->docs_pending = false;
        const CtpfMstepJob jh{h->he_stats(), (float)h->hyper[4], h->d_he, h->d_he_old, h->d_TH, h->U};
        const CtpfMstepJob ja{h->d_stats, (float)h->hyper[0], h->d_alef, h->d_alef_old, h->d_TA, h->V};
        const CtpfMstepTail tl{with_docs, h->hyper[1], h->hyper[3], h->hyper[5], h->hyper[7], h->d_partial, h->d_mstep_counter,
                               h->d_rs_he, h->d_rs_alef, h->d_sum_g, h->d_sum_z, h->tail(), h->d_rates, h->d_lrates};
        hipLaunchKernelGGL(ctpf_mstep_kernel, dim3(NB, with_docs ? 4 : 2), dim3(1024), 0, ctx->stream, jh, ja, h->d_gimel, h->d_zayin, h->M, tl, h->K, h->KP);
        TMVB_HIP(hipGetLastError());
        h->rs_fresh = true;
        return TMVB_OK;
    }
    // he_old <- he; he <- e + stats; TH refresh; rowsum(he)    (:266-270)
    // (measured: putting this branch on a second stream gains nothing -- the kernels are 5-8 us each and a
    //  cross-stream dependency costs as much)
    // alef_old <- alef; alef <- a + stats; TA refresh   (:251-255) -- both shape updates in one launch (blockIdx.y)
    if (h->U > 0) {
        int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ((int64_t)h->KP * std::max(h->U, h->V) + 255) / 256));
        const CtpfShapeJob jh{h->he_stats(), (float)h->hyper[4], h->d_he, h->d_he_old, h->d_TH, h->U};
        const CtpfShapeJob ja{h->d_stats, (float)h->hyper[0], h->d_alef, h->d_alef_old, h->d_TA, h->V};
        hipLaunchKernelGGL(ctpf_shape2_kernel, dim3(nb, 2), dim3(256), 0, ctx->stream, jh, ja, h->K, h->KP);
        TMVB_HIP(hipGetLastError());
    } else {
        TMVB_HIP(hipMemsetAsync(h->d_rs_he, 0, (size_t)h->K * sizeof(double), ctx->stream));
        int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ((int64_t)h->KP * h->V + 255) / 256));
        hipLaunchKernelGGL(ctpf_shape_kernel, dim3(nb), dim3(256), 0, ctx->stream, h->d_stats, (float)h->hyper[0], h->d_alef, h->d_alef_old, h->d_TA, h->K, h->KP, h->V, 0);
        TMVB_HIP(hipGetLastError());
    }
    if (h->U > 0 && h->K <= 64) {       // rowsum(alef) and rowsum(he) in one pair of launches
        if ((rc = tmvb_colsum2(ctx, h->K, {h->d_alef, h->V, h->d_partial, h->d_rs_alef, nullptr}, {h->d_he, h->U, h->d_partial2, h->d_rs_he, nullptr}))) return rc;
    } else {
        if ((rc = tmvb_colsum(ctx, h->nslot, h->K, h->d_alef, h->V, h->d_partial, h->d_rs_alef, nullptr))) return rc;
        if (h->U > 0 && (rc = tmvb_colsum(ctx, h->nslot, h->K, h->d_he, h->U, h->d_partial2, h->d_rs_he, nullptr))) return rc;
    }
    hipLaunchKernelGGL((ctpf_rates_kernel<1>), dim3(1), dim3(64), 0, ctx->stream, h->K, h->hyper[1], h->hyper[3], h->hyper[5], h->hyper[7],
                       h->d_rs_alef, h->d_rs_he, h->d_sum_g, h->d_sum_z, h->d_rates, h->d_lrates);
    TMVB_HIP(hipGetLastError());
    h->rs_fresh = true;                                   // d_rs_alef / d_rs_he = row sums of the alef / he just written
    return TMVB_OK;
}

// update_elbo! (src/CTPF.jl:234-247) on the device; returns the sum over this context's documents plus -- on every
// rank -- the global (beta, eta) terms, so a multi-process host must add the ranks' per-document parts only once
// with the global part (tmvb_ctpf_update_elbo_parts).
extern "C" int tmvb_ctpf_update_elbo_parts(tmvb_ctpf* h, double* doc_part, double* global_part)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctpf_update_elbo: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    int rc;
    // rowsums of the CURRENT shapes (Elogpya/Elogpyb/Elogpz use sum(he, dims=2), sum(alef, dims=2)): the M-step has just left them in
    // d_rs_alef / d_rs_he unless the state was set through the API since (rs_fresh)
    if (!h->rs_fresh) {
        if ((rc = tmvb_colsum(ctx, h->nslot, h->K, h->d_alef, h->V, h->d_partial, h->d_rs_alef, nullptr))) return rc;
        if (h->U > 0) { if ((rc = tmvb_colsum(ctx, h->nslot, h->K, h->d_he, h->U, h->d_partial, h->d_rs_he, nullptr))) return rc; }
        else TMVB_HIP(hipMemsetAsync(h->d_rs_he, 0, (size_t)h->K * sizeof(double), ctx->stream));
        h->rs_fresh = true;
    }
    const double* hy = h->hyper;
    double res[2] = {0.0, 0.0};
    static const bool legacy_elbo = [] { const char* e = getenv("TMVB_CTPF_ELBO_LEGACY"); return e && atoi(e) != 0; }();
    // the decomposed form: the last E-step collected its parts and exactly one M-step ran behind it (alef_old / he_old / the old rates are that E-step's)
    const bool parts = (h->M > 0 || h->distributed) && !legacy_elbo && h->logz_valid && h->msteps_after == 1 && !h->force_walk;
    h->elbo_form = parts ? 1 : 0;
    if ((h->M > 0 || h->distributed) && !legacy_elbo && (rc = ctpf_elbo_consts(h))) return rc;
    // global part: partial sums now, added up by the one final kernel behind the per-document part (one copy, one synchronisation)
    const int nb = 256;
    hipLaunchKernelGGL(ctpf_elbo_global_kernel, dim3(nb), dim3(256), 0, ctx->stream, h->d_alef, h->V, h->K, h->d_rates, hy[0], hy[1], h->d_elbo_partial,
                       parts ? (const float*)h->d_alef_old : (const float*)nullptr, parts ? (const double*)h->d_rates : (const double*)nullptr,
                       parts ? h->d_lrates_d : (double*)nullptr);
    hipLaunchKernelGGL(ctpf_elbo_global_kernel, dim3(nb), dim3(256), 0, ctx->stream, h->d_he, h->U, h->K, h->d_rates + h->K, hy[4], hy[5], h->d_elbo_partial + nb,
                       parts ? (const float*)h->d_he_old : (const float*)nullptr, (const double*)nullptr, (double*)nullptr);
    TMVB_HIP(hipGetLastError());
    const double cst = (double)h->V * h->K * (hy[0] * std::log(hy[1]) - std::lgamma(hy[0])) + (double)h->U * h->K * (hy[4] * std::log(hy[5]) - std::lgamma(hy[4]));
    // per-document part
    if (parts) {
        if (h->elbo_pending) {                               // the per-document sums, enqueued by the E-step on aux[ELBO]
            TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_elbo, 0));
            h->elbo_pending = false;
        }
        hipLaunchKernelGGL(ctpf_elbo_final_parts_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double*)h->d_doc_val, h->n_elbo_blocks, (double)h->M,
                           (const double*)h->d_elbo_partial, 2 * nb, cst, (const double*)h->d_logz, h->n_logz, (const double*)h->d_rates, (const double*)h->d_lrates_d,
                           (const double*)h->d_rs_alef, (const double*)h->d_rs_he, (const double*)h->d_sum_g, (const double*)h->d_sum_z, h->K,
                           (double)h->V * hy[0], hy[2], hy[3], hy[6], hy[7], h->d_elbo, h->distributed ? 1 : 0);
        TMVB_HIP(hipGetLastError());
        TMVB_HIP(hipMemcpyAsync(res, h->d_elbo, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        TMVB_HIP(hipStreamSynchronize(ctx->stream));
        if (doc_part) *doc_part = res[0];
        if (global_part) *global_part = res[1];
        return TMVB_OK;
    }
    if (h->M > 0 && !legacy_elbo) {
        if (!h->d_TAo) {                                  // first call of the table form: its tables
            const size_t na = (size_t)h->V * h->KP + 4, nh = (size_t)std::max<int64_t>(h->U, 1) * h->KP + 4;
            if ((rc = dmalloc(&h->d_TAo, na)) || (rc = dmalloc(&h->d_DA, na)) || (rc = dmalloc(&h->d_THo, nh)) || (rc = dmalloc(&h->d_DH, nh))) return rc;
        }
        int nbt = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ((int64_t)h->KP * h->V + 255) / 256));
        hipLaunchKernelGGL(ctpf_elbo_tables_kernel, dim3(nbt), dim3(256), 0, ctx->stream, h->d_alef, h->d_alef_old, h->d_TAo, h->d_DA, h->K, h->KP, h->V,
                           (const double*)h->d_rates, h->d_lrates_d);
        if (h->U > 0) {
            nbt = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ((int64_t)h->KP * h->U + 255) / 256));
            hipLaunchKernelGGL(ctpf_elbo_tables_kernel, dim3(nbt), dim3(256), 0, ctx->stream, h->d_he, h->d_he_old, h->d_THo, h->d_DH, h->K, h->KP, h->U,
                               (const double*)nullptr, (double*)nullptr);
        }
        TMVB_HIP(hipGetLastError());
        auto fast = [&](auto ns) {
            constexpr int NS = decltype(ns)::value;
            hipLaunchKernelGGL((ctpf_elbo_doc_fast_kernel<NS>), dim3((unsigned)h->M), dim3(64), 0, ctx->stream, h->K, h->KP, h->corp->d_doc_ptr, h->corp->d_terms,
                               h->corp->d_counts, h->corp->d_rdr_ptr, h->corp->d_readers, h->corp->d_ratings, h->d_TAo, h->d_DA, h->d_THo, h->d_DH,
                               h->d_rates, h->d_lrates_d, h->d_rs_alef, h->d_rs_he, h->d_gimel, h->d_gimel_old, h->d_zayin, h->d_zayin_old, h->d_lg_doc,
                               hy[2], hy[3], hy[6], hy[7],
                               (double)h->K * (hy[2] * std::log(hy[3]) - std::lgamma(hy[2])) + (double)h->K * (hy[6] * std::log(hy[7]) - std::lgamma(hy[6])), h->d_doc_val);
        };
        if (h->nslot == 1) fast(std::integral_constant<int, 1>()); else if (h->nslot == 2) fast(std::integral_constant<int, 2>());
        else if (h->nslot <= 4) fast(std::integral_constant<int, 4>()); else fast(std::integral_constant<int, 8>());
        TMVB_HIP(hipGetLastError());
    } else if (h->M > 0) {
        auto slow = [&](auto ns) {
            constexpr int NS = decltype(ns)::value;
            hipLaunchKernelGGL((ctpf_elbo_doc_kernel<NS>), dim3((unsigned)h->M), dim3(64), 0, ctx->stream, h->K, h->corp->d_doc_ptr, h->corp->d_terms,
                               h->corp->d_counts, h->corp->d_rdr_ptr, h->corp->d_readers, h->corp->d_ratings, h->d_alef, h->d_alef_old, h->d_he,
                               h->d_he_old, h->d_rates, h->d_rs_alef, h->d_rs_he, h->d_gimel, h->d_gimel_old, h->d_zayin, h->d_zayin_old,
                               hy[2], hy[3], hy[6], hy[7], h->d_doc_val);
        };
        if (h->nslot == 1) slow(std::integral_constant<int, 1>()); else if (h->nslot == 2) slow(std::integral_constant<int, 2>());
        else if (h->nslot <= 4) slow(std::integral_constant<int, 4>()); else slow(std::integral_constant<int, 8>());
        TMVB_HIP(hipGetLastError());
    }
    // d_elbo[0] = sum over the documents (the part that adds up over shards), d_elbo[1] = the global part
    hipLaunchKernelGGL(ctpf_elbo_final2_kernel, dim3(1), dim3(1024), 0, ctx->stream, h->d_doc_val, h->M, h->d_elbo_partial, 2 * nb, cst, h->d_elbo);
    TMVB_HIP(hipGetLastError());
    TMVB_HIP(hipMemcpyAsync(res, h->d_elbo, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    if (doc_part) *doc_part = res[0];
    if (global_part) *global_part = res[1];
    return TMVB_OK;
}

extern "C" int tmvb_ctpf_elbo_form(tmvb_ctpf* h, int32_t* form)
{
    TMVB_REQUIRE(h && form, TMVB_EINVAL, "tmvb_ctpf_elbo_form: NULL argument");
    *form = h->elbo_form;
    return TMVB_OK;
}

extern "C" int tmvb_ctpf_update_elbo(tmvb_ctpf* h, double* elbo)
{
    double dp = 0.0, gp = 0.0;
    int rc = tmvb_ctpf_update_elbo_parts(h, &dp, &gp);
    if (rc) return rc;
    h->elbo = dp + gp;
    if (elbo) *elbo = h->elbo;
    return TMVB_OK;
}

// train! (src/gpuCTPF.jl:677-705 signature, src/CTPF.jl:344-376 semantics incl. check_elbo!).  checkelbo <= 0 means Inf.
extern "C" int tmvb_ctpf_set_comm(tmvb_ctpf* h, tmvb_comm* comm)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctpf_set_comm: handle is NULL");
    int rc = tmvb_ctpf_set_distributed(h, comm != nullptr);
    if (rc) return rc;
    h->comm = comm;
    return TMVB_OK;
}

namespace {
struct CtpfTrainOps {
    int viter; double vtol;
    int estep(tmvb_ctpf* h) { return tmvb_ctpf_estep(h, viter, vtol); }               // src/CTPF.jl:353-365
    int reduce(tmvb_ctpf* h) { return tmvb_ctpf_reduce_docs(h); }
    int before_allreduce(tmvb_ctpf*) { return TMVB_OK; }
    float* stats(tmvb_ctpf* h) { return h->d_stats; }
    int64_t stats_len(tmvb_ctpf* h) { return h->stats_len(); }
    int mstep(tmvb_ctpf* h) { return tmvb_ctpf_mstep(h); }                            // :366-371
    // the per-document part adds up over the shards; the (beta, eta) part is global and identical on every rank
    int elbo_local(tmvb_ctpf* h, double* s, double* once) { return tmvb_ctpf_update_elbo_parts(h, s, once); }
    int elbo_form(tmvb_ctpf* h) { return h->elbo_form; }
    void force_walk(tmvb_ctpf* h, bool on, bool doubled = true) { h->force_walk = on; if (!on && doubled) h->elbo_form = 1; }   // (switched off behind the one evaluation that doubled a decomposed one)
    void will_check(tmvb_ctpf* h, bool checked) { h->want_parts = checked; }          // the coming iteration ends in check_elbo!
    // an unchecked iteration may be replayed from a hipGraph (tmvb_train.h): no pointer of the iteration alternates, no flag outlives it; not while an
    // update_elbo! kernel of a checked iteration is still pending on its side stream (the E-step would wait for an event recorded outside the capture)
    bool graph_ok(tmvb_ctpf* h) { return !h->distributed && !h->elbo_pending && !h->docs_pending && h->M > 0 && !h->timing; }
    double* elbo_dev(tmvb_ctpf* h) { return h->d_elbo; }
    tmvb_comm* comm(tmvb_ctpf* h) { return h->comm; }
    bool distributed(tmvb_ctpf* h) { return h->distributed; }
    tmvb_ctx* ctx(tmvb_ctpf* h) { return h->ctx; }
    int64_t nnz(tmvb_ctpf* h) { return h->corp->info.nnz; }
    void set_elbo(tmvb_ctpf* h, double v) { h->elbo = v; }
    double get_elbo(tmvb_ctpf* h) { return h->elbo; }
    int finish(tmvb_ctpf* h)
    {
        TMVB_HIP(hipSetDevice(h->ctx->device));
        TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
        return TMVB_OK;
    }
};
}  // namespace

extern "C" int tmvb_ctpf_train_group(tmvb_ctpf* const* hs, int32_t n, int32_t iter, double tol, int32_t viter, double vtol,
                                     int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline)
{
    TMVB_REQUIRE(tol >= 0 && vtol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");     // src/gpuCTPF.jl:679
    TMVB_REQUIRE(iter >= 0 && viter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative.");   // :680
    CtpfTrainOps ops{viter, vtol};
    return tmvb_train_group_loop("tmvb_ctpf_train", hs, n, iter, tol, checkelbo, elbo_traj, iters_done, elbo_baseline, ops);
}

extern "C" int tmvb_ctpf_train(tmvb_ctpf* h, int32_t iter, double tol, int32_t viter, double vtol, int32_t checkelbo,
                               double* elbo_traj, int32_t* iters_done, double* elbo_baseline)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_ctpf_train: handle is NULL");
    return tmvb_ctpf_train_group(&h, 1, iter, tol, viter, vtol, checkelbo, elbo_traj, iters_done, elbo_baseline);
}

int tmvb_ctpf_view_of(tmvb_ctpf* h, tmvb_ctpf_view* v)
{
    TMVB_REQUIRE(h && v, TMVB_EINVAL, "tmvb_ctpf_recommend: handle is NULL");
    v->ctx = h->ctx; v->corp = h->corp; v->K = h->K; v->M = h->M; v->U = h->U;
    v->gimel = h->d_gimel; v->zayin = h->d_zayin; v->he = h->d_he; v->rates = h->d_rates;
    return TMVB_OK;
}

// per-document sweep counts of the last E-step (document order of the corpus), for parity tests that compare the
// state of exactly those documents whose exit sweep agrees with the oracle's
extern "C" int tmvb_ctpf_doc_sweeps(tmvb_ctpf* h, uint8_t* out)
{
    TMVB_REQUIRE(h && (out || h->M == 0), TMVB_EINVAL, "tmvb_ctpf_doc_sweeps: NULL argument");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    if (h->M) TMVB_HIP(hipMemcpyAsync(out, h->d_sweeps, (size_t)h->M, hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_ctpf_sweep_hist(tmvb_ctpf* h, int64_t* hist, int32_t nbins)
{
    TMVB_REQUIRE(h && hist && nbins > 0, TMVB_EINVAL, "tmvb_ctpf_sweep_hist: bad argument");
    std::vector<uint8_t> sw((size_t)h->M);
    TMVB_HIP(hipSetDevice(h->ctx->device));
    if (h->M) TMVB_HIP(hipMemcpyAsync(sw.data(), h->d_sweeps, (size_t)h->M, hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    for (int b = 0; b < nbins; ++b) hist[b] = 0;
    for (uint8_t s : sw) hist[std::min<int>(s, nbins - 1)]++;
    return TMVB_OK;
}

extern "C" int tmvb_ctpf_last_estep_ms(tmvb_ctpf* h, float* ms)
{
    TMVB_REQUIRE(h && ms, TMVB_EINVAL, "tmvb_ctpf_last_estep_ms: NULL argument");
    TMVB_REQUIRE(h->timing, TMVB_EINVAL, "tmvb_ctpf_last_estep_ms: E-step timing is off (set TMVB_ESTEP_TIMING=1 before creating the model)");
    TMVB_REQUIRE(h->timed, TMVB_EINVAL, "tmvb_ctpf_last_estep_ms: no E-step has run");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    TMVB_HIP(hipEventSynchronize(h->ev1));
    TMVB_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return TMVB_OK;
}
