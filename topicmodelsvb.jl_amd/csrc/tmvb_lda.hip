// tmvb_lda.hip -- LDA variational-Bayes engine for gfx950 (MI355X).
//
// Path: the per-document coordinate ascent of src/LDA.jl:170-180 (update_phi! :150, update_gamma!
// :143, update_Elogtheta! :136, exit test :175, update_beta!(d) :129) fused per document, plus
// the corpus-wide M-step (update_beta! :121, update_alpha! :97) and update_elbo! (:83) on device.
// It replaces the seven OpenCL kernels of src/gpuLDA.jl:156-333 but follows the CPU path's
// semantics (per-document early exit), not the OpenCL path's global-median rule (:361).
//
// Data layout in HBM (fp32 state, int32 ids):
//   beta[2][V][KP] (gather layout, current / old ping-pong, KP = 4*odd >= K)
//   stats[K*V + K] (S = the reference's beta_temp, then Elogtheta_sum; the all-reduce payload)
//   gamma, Elogtheta, Elogtheta_old [M][K]   E [M][ES] (last-sweep exp(Elogtheta), 128-byte-aligned rows)
//   CSR doc_ptr i64[M+1], terms/counts i32[nnz]; doc_order i32[M] (processing order);
//   one inverted (term-major) index per document piece.
//
// Kernels (one 64-lane wave = one document; phi is never materialised):
//   lda_estep_reg_kernel<LPR,T>  documents of <= 64 T unique terms, any K <= 100: the N_d x KP tile lives in VGPRs
//                                as topic pairs; per sweep  s_n = K eps + sum_i B[n][i] e_i  (v_pk_fma_f32 with SGPR
//                                pairs), w_n = c_n / s_n, g_i = sum_n w_n B[n][i] (cross-lane reduce-scatter),
//                                gamma_i = eps + alpha_i + e_i g_i + eps sum w, Elogtheta = psi(gamma) - psi(sum gamma)
//   lda_estep_kernel<NSLOT,LPR>  any K <= 1024 / longer documents: the same arithmetic with the tile in LDS
//                                (LDS-DMA gather, up to 156 KiB per document)
//   termstats_* (tmvb_termstats.h)  update_beta!(model, d) as a gather over the inverted index, no atomics:
//                                S[:, j] += beta[:, j] .* sum_tokens w E[:, doc] + eps sum w, w recomputed in place
//   lda_alpha_kernel, beta_norm_kernel, lda_elbo_kernel   M-step and ELBO
// tmvb_lda_estep pipelines the statistics pass of document piece p under the document kernels of piece p + 1.
//
// Roofline: HBM/gather-bound by byte count, VALU-issue bound in practice.  Algorithmic
// bytes per outer iteration: nnz*(8 + 4K + 4K) + 12*M*K + 12*K*V + 4*(M+1).
#define TMVB_TS_LOGZ 1            // this translation unit instantiates the log-normaliser forms of the statistics pass (update_elbo!)
#include "tmvb_common_kernels.h"
#include "tmvb_train.h"
#include "tmvb_dirichlet.h"
#include "tmvb_regtile.h"
#include "tmvb_gridtile.h"

// e_q of the first TMVB_LDA_E_LDS topics reach the lanes through LDS (one ds_write, broadcast ds_read_b128) instead of
// v_readlane: moves phase 1's broadcast issue slots from the saturated VALU to the idle LDS pipe.  Must be a multiple of 32
// (the v_readlane blocks cover topics in blocks of 32 from the first one the LDS path does not) or 0 = all v_readlane.
// Round 2, steady state: 64 (all topics for K <= 60, the first 64 for K = 100) against 32: K = 50 +1 %, cold start +2 %,
// K = 100 +3 %, a 16 100-document shard +4 %.
#ifndef TMVB_LDA_E_LDS
#define TMVB_LDA_E_LDS 64
#endif

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>


struct LdaParams {
    int K, KP, LPR;          // topics, padded row stride (4*odd), 16-byte chunks per row (KP/4)
    unsigned lpr_magic;      // floor(2^32 / LPR) + 1  (exact f / LPR for f < 2^20)
    int64_t V;
    const int64_t* doc_ptr;
    const int32_t* terms;
    const int32_t* counts;
    const int32_t* doc_order;
    const float* alpha;
    const float* beta;       // [V][KP], pad columns zero
    float* wtok;             // [nnz] last-sweep c_n / s_n per token, stored in term-major (inverted index) order
    const int32_t* tok_inv;  // [nnz] term-major position of each CSR token
    float* E;                // [M][estride] last-sweep exp(Elogtheta_old) (estride = KP, pads zero, when K <= 256)
    int estride;
    float* gamma;
    float* elog;
    float* elog_old;
    uint8_t* sweeps;
    int viter;
    float vtol;
    int debug;               // TMVB_DEBUG_FLAGS (profiling experiments only): 1 = skip the statistics pass
    int store_w;             // 0 when the statistics pass recomputes w (K <= 64): no per-token stores at all
};


// ------------------------------------------------------------------------------ E-step kernel
// LPR_T: compile-time chunks-per-row (0 = use p.LPR); NSLOT = ceil(K/64).
template <int NSLOT, int LPR_T>
__global__ __launch_bounds__(64) void lda_estep_kernel(LdaParams p, int64_t first, int tile_rows)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int K = p.K;
    const int LPR = LPR_T ? LPR_T : p.LPR;
    const int KP = 4 * LPR;
    float* Bt = lds;                           // [tile_rows][KP]  beta columns of the document's terms
    float* e_l = Bt + (size_t)tile_rows * KP;  // [KP]             exp(Elogtheta), zero padded
    float* w_l = e_l + KP;                     // [tile_rows]      c_n / s_n
    float* c_l = w_l + tile_rows;              // [tile_rows]      counts as float
    int* t_l = (int*)(c_l + tile_rows);        // [tile_rows]      term ids

    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d];
    const int N = (int)(p.doc_ptr[d + 1] - off);
    const bool single = N <= tile_rows;

    float alpha[NSLOT], elog[NSLOT], elog_old[NSLOT], gam[NSLOT], e[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        int i = lane + 64 * s;
        alpha[s] = (i < K) ? p.alpha[i] : 0.0f;
        elog[s] = (i < K) ? p.elog[(int64_t)d * K + i] : 0.0f;
        elog_old[s] = elog[s];
        gam[s] = 0.0f;
        e[s] = 0.0f;
    }

    // Gather one chunk of the document into LDS.  The tile is a contiguous run of 16-byte chunks
    // (row stride = LPR chunks), so every LDS-DMA instruction (global_load_lds_dwordx4) moves 64
    // consecutive chunks = 1 KiB straight from L2/HBM into LDS with no VGPR round trip.
    auto load_chunk = [&](int c0, int rows) {
        for (int n = lane; n < rows; n += 64) {
            t_l[n] = p.terms[off + c0 + n];
            c_l[n] = (float)p.counts[off + c0 + n];
        }
        WAVE_LDS_FENCE();
        const int nch = rows * LPR;
#pragma unroll 4
        for (int f0 = 0; f0 < nch; f0 += 64) {
            const int f = f0 + lane;
            if (f < nch) {
                const int n = (LPR == 1) ? f : (int)__umulhi((unsigned)f, p.lpr_magic);
                const int c = f - n * LPR;
                const float* src = p.beta + ((int64_t)t_l[n] * KP + 4 * c);
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(Bt + (size_t)f0 * 4), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WAVE_LDS_FENCE();
    };

    // phase 1: lane = token.  w_n = c_n / (K eps + B[n,:] . e); returns this lane's sum of w
    auto phase1 = [&](int rows) -> float {
        float wl = 0.0f;
        const float4* er = (const float4*)e_l;
        for (int n = lane; n < rows; n += 64) {
            const float4* br = (const float4*)(Bt + (size_t)n * KP);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if (LPR_T) {
#pragma unroll
                for (int q = 0; q < LPR_T; ++q) {
                    float4 b = br[q], ev = er[q];
                    s0 = fmaf(b.x, ev.x, s0); s1 = fmaf(b.y, ev.y, s1);
                    s2 = fmaf(b.z, ev.z, s2); s3 = fmaf(b.w, ev.w, s3);
                }
            } else {
#pragma unroll 4
                for (int q = 0; q < LPR; ++q) {
                    float4 b = br[q], ev = er[q];
                    s0 = fmaf(b.x, ev.x, s0); s1 = fmaf(b.y, ev.y, s1);
                    s2 = fmaf(b.z, ev.z, s2); s3 = fmaf(b.w, ev.w, s3);
                }
            }
            float sn = ((s0 + s1) + (s2 + s3)) + (float)K * TMVB_EPS_F;
            float w = c_l[n] / sn;
            w_l[n] = w;
            wl += w;
        }
        WAVE_LDS_FENCE();
        return wl;
    };
    // documents that stream chunks keep the latest w in HBM every sweep (the last one survives)
    auto store_w = [&](int c0, int rows) {
        if (!p.store_w) return;
        for (int n = lane; n < rows; n += 64) p.wtok[p.tok_inv[off + c0 + n]] = w_l[n];
    };

    // phase 2: lane = 4*ql + r handles topic quad q = 16 s + ql for the tokens n = r (mod 4);
    // acc[s] += w_n * B[n][4q..4q+3]  (one ds_read_b128 per 4 fmas)
    const int r4 = lane & 3, ql = lane >> 2;
    auto phase2 = [&](int rows, float4* acc) {
        const int nfull = rows >> 2;
#pragma unroll 4
        for (int m = 0; m < nfull; ++m) {
            const int n = 4 * m + r4;
            const float w = w_l[n];
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                const int q = 16 * s + ql;
                if (q < LPR) {
                    const float4 b = *(const float4*)(Bt + (size_t)n * KP + 4 * q);
                    acc[s].x = fmaf(w, b.x, acc[s].x); acc[s].y = fmaf(w, b.y, acc[s].y);
                    acc[s].z = fmaf(w, b.z, acc[s].z); acc[s].w = fmaf(w, b.w, acc[s].w);
                }
            }
        }
        const int n = 4 * nfull + r4;
        if (n < rows) {
            const float w = w_l[n];
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                const int q = 16 * s + ql;
                if (q < LPR) {
                    const float4 b = *(const float4*)(Bt + (size_t)n * KP + 4 * q);
                    acc[s].x = fmaf(w, b.x, acc[s].x); acc[s].y = fmaf(w, b.y, acc[s].y);
                    acc[s].z = fmaf(w, b.z, acc[s].z); acc[s].w = fmaf(w, b.w, acc[s].w);
                }
            }
        }
    };

    int sweeps = 0;
    for (int v = 0; v < p.viter; ++v) {
        ++sweeps;
        // e = exp(Elogtheta)   (update_phi!, src/LDA.jl:152)
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int i = lane + 64 * s;
            e[s] = (i < K) ? expf(elog[s]) : 0.0f;
            if (i < KP) e_l[i] = e[s];
        }
        WAVE_LDS_FENCE();
        float4 acc[NSLOT];
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        float wl = 0.0f;
        for (int c0 = 0; c0 < N; c0 += tile_rows) {
            const int rows = min(tile_rows, N - c0);
            if (!(single && v > 0)) load_chunk(c0, rows);
            wl += phase1(rows);
            if (!single) store_w(c0, rows);
            phase2(rows, acc);
            if (!single) WAVE_LDS_FENCE();
        }
        const float wsum = wave_sum(wl);
        // update_gamma!  src/LDA.jl:145:  gamma = EPS + (alpha + phi*counts)
        float gl = 0.0f;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int i = lane + 64 * s;
            float4 a = dpp_add4<0xB1>(acc[s]);      // quad_perm [1,0,3,2]
            a = dpp_add4<0x4E>(a);                  // quad_perm [2,3,0,1]: all 4 lanes hold the quad's sums
            const float g = (r4 == 0) ? a.x : (r4 == 1) ? a.y : (r4 == 2) ? a.z : a.w;   // topic 4q + r = i
            gam[s] = TMVB_EPS_F + (alpha[s] + fmaf(e[s], g, TMVB_EPS_F * wsum));
            if (i < K) gl += gam[s];
        }
        const float gsum = wave_sum(gl);
        const float dgs = digamma_f(gsum);
        // update_Elogtheta!  src/LDA.jl:137-138 and the exit test :175
        float dl = 0.0f;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int i = lane + 64 * s;
            elog_old[s] = elog[s];
            if (i < K) {
                elog[s] = digamma_f(gam[s]) - dgs;
                float df = elog[s] - elog_old[s];
                dl = fmaf(df, df, dl);
            }
        }
        const float dist2 = wave_sum(dl);
        if (__builtin_amdgcn_sqrtf(dist2) < p.vtol) break;   // v_sqrt_f32 (1 ulp) on the wave-uniform sum
    }

    if (sweeps > 0) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int i = lane + 64 * s;
            if (i < K) {
                p.gamma[(int64_t)d * K + i] = gam[s];
                p.elog[(int64_t)d * K + i] = elog[s];
                p.elog_old[(int64_t)d * K + i] = elog_old[s];
            }
        }
        // update_beta!(model, d)  src/LDA.jl:131 is deferred to the gather-side statistics pass
        // (tmvb_termstats.h): the LAST sweep's phi .* counts' = w_n (beta[i,t_n] e_i + eps) is rebuilt
        // there from w (per token) and e = exp(Elogtheta_old) (per document), so only those are stored.
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int i = lane + 64 * s;
            if (i < p.estride) p.E[(int64_t)d * p.estride + i] = e[s];   // e is 0 beyond K
        }
        if (single) store_w(0, N);
    } else {
        // viter = 0: no responsibilities.  E = 0 makes the recomputing statistics pass produce eps-only columns (the
        // register-tile kernel does the same); the stored-weight pass reads w = 0.
        for (int i = lane; i < p.estride; i += 64) p.E[(int64_t)d * p.estride + i] = 0.0f;
        if (p.store_w) for (int n = lane; n < N; n += 64) p.wtok[p.tok_inv[off + n]] = 0.0f;
    }
    if (lane == 0) p.sweeps[d] = (uint8_t)min(sweeps, 255);
}

// ------------------------------------------------------------------------------ register-tile E-step
// For K <= 64 and documents of at most 64*T unique terms the whole N_d x KP topic tile lives in
// VGPRs (lane = token, register = topic): the register file (512 KiB per CU) is 3x the LDS, so the
// tile costs no LDS and the sweeps run without a single LDS wait.
//   phase 1  s_n = K eps + sum_q B[n][q] e_q     R = KP fmas per tile with e_q as an SGPR operand
//   phase 2  g_q = sum_n w_n B[n][q]             R products per tile, then a 6-stage reduce-scatter
//            over the 64 lanes (v_permlane32_swap, v_permlane16_swap, 4 DPP stages; ~2.2 R
//            instructions): afterwards lane L holds the total of topic pi(L)
// pi is the compile-time map kRegLaneMap<R> (tmvb_common_kernels.h); the host uploads topic_of_lane from it.
// W > 1: W waves (one workgroup) share a LONG document -- wave w holds the tokens lane + 64 (w + W t) in its own
// register tiles, the per-wave partial sums g_i (and sum w) are exchanged through a few hundred bytes of LDS once
// per sweep (one __syncthreads, double-buffered), and every wave then updates gamma / Elogtheta redundantly and
// bit-identically, so the exit test stays wave-uniform across the workgroup.  Same VALU-efficient arithmetic as the
// single-wave kernel for documents of up to 64 T W unique terms.
template <int LPR, int T, int W = 1>
__device__ __forceinline__ void lda_estep_reg_body(const LdaParams& p, const int d, const int64_t off, const int N,
                                                   const int* __restrict__ topic_of_lane)
{
    constexpr int R = 4 * LPR;
    const int lane = threadIdx.x & 63;
    const int wave = (W > 1) ? (int)(threadIdx.x >> 6) : 0;
    const int K = p.K;

    // token role: lane n owns tokens n + 64 t; the tile is held as topic PAIRS so that both phases run on
    // packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two fp32 lanes-ops per issue slot)
    v2f B2[T][R / 2];
    float c[T], w[T];
    int wpos[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        // branch-free: a lane past the document's last token reads term 0's row with count 0, so its weight
        // w = c / s is exactly 0 and nothing of the row reaches gamma (no zero-filled tile, no divergent tile load)
        const int n = lane + 64 * (wave + W * t);
        const bool in = n < N;
        const int term = in ? p.terms[off + n] : 0;
        c[t] = in ? (float)p.counts[off + n] : 0.0f;
        wpos[t] = in ? (p.store_w ? p.tok_inv[off + n] : 0) : -1;
        const float4* row = (const float4*)(p.beta + (int64_t)term * R);
#pragma unroll
        for (int q = 0; q < LPR; ++q) {
            const float4 v = row[q];
            B2[t][2 * q] = v2f{v.x, v.y}; B2[t][2 * q + 1] = v2f{v.z, v.w};
        }
        w[t] = 0.0f;
    }
    // topic role: lane L of result slot sl owns topic pi(sl, L) (duplicates are marked -1 by the host)
    constexpr int NS = (R + 63) / 64;
    constexpr int SPARE = kRegLaneMap<R>.spare;                    // a lane of the last result slot that owns no topic
    static_assert(SPARE >= 0, "register-tile kernel needs a spare lane for digamma(sum gamma)");
    int mytopic[NS];
    bool on[NS];
    float alpha[NS], elog[NS], elog_old[NS], gam[NS], e[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        mytopic[sl] = topic_of_lane[sl * 64 + lane];
        on[sl] = mytopic[sl] >= 0 && mytopic[sl] < K;
        alpha[sl] = on[sl] ? p.alpha[mytopic[sl]] : 0.0f;
        elog[sl] = on[sl] ? p.elog[(int64_t)d * K + mytopic[sl]] : 0.0f;
        elog_old[sl] = elog[sl]; gam[sl] = 0.0f; e[sl] = 0.0f;
    }

#if TMVB_LDA_E_LDS
    __shared__ __attribute__((aligned(16))) float e_lds_all[W][R];
    float* e_lds = e_lds_all[wave];
#endif
    __shared__ float xch[2][W > 1 ? W : 1][W > 1 ? R + 1 : 1];      // W > 1: per-wave partial (g | sum w), by sweep parity
    int sweeps = 0;
    for (int v = 0; v < p.viter; ++v) {
        ++sweeps;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) e[sl] = on[sl] ? fast_exp(elog[sl]) : 0.0f;     // update_phi!, src/LDA.jl:152
        // phase 1 in blocks of 32 topics: the e_q of a block live in SGPRs only while the block is consumed
        v2f sacc[T][2];
#pragma unroll
        for (int t = 0; t < T; ++t) { sacc[t][0] = v2f{0.f, 0.f}; sacc[t][1] = v2f{0.f, 0.f}; }
#if TMVB_LDA_E_LDS
        constexpr int ELDS = (TMVB_LDA_E_LDS < R) ? TMVB_LDA_E_LDS : R;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) if (mytopic[sl] >= 0 && mytopic[sl] < ELDS) e_lds[mytopic[sl]] = e[sl];
        if constexpr (W > 1) WAVE_PRIVATE_LDS_FENCE(); else WAVE_LDS_FENCE();
#pragma unroll
        for (int j = 0; j < ELDS / 4; ++j) {
            const float4 ev = ((const float4*)e_lds)[j];
            const v2f ea = v2f{ev.x, ev.y}, eb = v2f{ev.z, ev.w};
#pragma unroll
            for (int t = 0; t < T; ++t) {
                sacc[t][0] = __builtin_elementwise_fma(B2[t][2 * j], ea, sacc[t][0]);
                sacc[t][1] = __builtin_elementwise_fma(B2[t][2 * j + 1], eb, sacc[t][1]);
            }
        }
        if constexpr (ELDS <= 0) regtile_phase1_block<R, T, 0>(B2, e, sacc);
        if constexpr (R > 32 && ELDS <= 32) regtile_phase1_block<R, T, 32>(B2, e, sacc);
        if constexpr (R > 64 && ELDS <= 64) regtile_phase1_block<R, T, 64>(B2, e, sacc);
        if constexpr (R > 96 && ELDS <= 96) regtile_phase1_block<R, T, 96>(B2, e, sacc);
#else
        regtile_phase1_block<R, T, 0>(B2, e, sacc);
        if constexpr (R > 32) regtile_phase1_block<R, T, 32>(B2, e, sacc);
        if constexpr (R > 64) regtile_phase1_block<R, T, 64>(B2, e, sacc);
        if constexpr (R > 96) regtile_phase1_block<R, T, 96>(B2, e, sacc);
#endif
        float wl = 0.0f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const v2f s2 = sacc[t][0] + sacc[t][1];
            const float sn = (s2.x + s2.y) + (float)K * TMVB_EPS_F;
            w[t] = fast_div(c[t], sn);
            wl += w[t];
        }
        const float wsum = wave_sum(wl);
        float pr[NS];
        lane_reduce_scatter<R>([&](int q) {
            v2f a = B2[0][q] * v2f{w[0], w[0]};
#pragma unroll
            for (int t = 1; t < T; ++t) a = __builtin_elementwise_fma(B2[t][q], v2f{w[t], w[t]}, a);
            return a;
        }, pr, lane);
        float wtot = wsum;
        if constexpr (W > 1) {
            float (*xb)[R + 1] = xch[v & 1];
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) if (mytopic[sl] >= 0) xb[wave][mytopic[sl]] = pr[sl];
            if (lane == 0) xb[wave][R] = wsum;
            __syncthreads();
            wtot = 0.0f;
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) pr[sl] = 0.0f;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) {                    // fixed order: identical totals in every wave
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) pr[sl] += xb[ww][mytopic[sl] >= 0 ? mytopic[sl] : 0];
                wtot += xb[ww][R];
            }
        }
        float gl = 0.0f;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            gam[sl] = TMVB_EPS_F + (alpha[sl] + fmaf(e[sl], pr[sl], TMVB_EPS_F * wtot));   // update_gamma!, src/LDA.jl:145
            if (on[sl]) gl += gam[sl];
        }
        const float gsum = wave_sum(gl);
        // digamma(gamma_k) and digamma(sum gamma) in ONE evaluation: the spare lane takes the sum
        float dg[NS];
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
            dg[sl] = digamma_f((sl == NS - 1 && lane == (SPARE & 63)) ? gsum : gam[sl]);
        const float dgs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dg[NS - 1]), SPARE & 63));
        float dl = 0.0f;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            elog_old[sl] = elog[sl];                                  // update_Elogtheta!, :137-138
            if (on[sl]) {
                elog[sl] = dg[sl] - dgs;
                const float df = elog[sl] - elog_old[sl];
                dl = fmaf(df, df, dl);
            }
        }
        const float dist2 = wave_sum(dl);
        if (__builtin_amdgcn_sqrtf(dist2) < p.vtol) break;   // v_sqrt_f32 (1 ulp) on the wave-uniform sum                             // :175
    }
    if (sweeps > 0) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            if (wave != 0) break;                                // every wave holds the same state: wave 0 stores it
            if (on[sl]) {
                p.gamma[(int64_t)d * K + mytopic[sl]] = gam[sl];
                p.elog[(int64_t)d * K + mytopic[sl]] = elog[sl];
                p.elog_old[(int64_t)d * K + mytopic[sl]] = elog_old[sl];
            }
            if (mytopic[sl] >= 0 && mytopic[sl] < p.estride) p.E[(int64_t)d * p.estride + mytopic[sl]] = e[sl];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) if (wpos[t] >= 0 && p.store_w) p.wtok[wpos[t]] = w[t];
    } else {
        // viter = 0: no responsibilities; E = 0 makes the statistics pass produce eps-only columns
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
            if (wave == 0 && mytopic[sl] >= 0 && mytopic[sl] < p.estride) p.E[(int64_t)d * p.estride + mytopic[sl]] = 0.0f;
#pragma unroll
        for (int t = 0; t < T; ++t) if (wpos[t] >= 0 && p.store_w) p.wtok[wpos[t]] = 0.0f;
    }
    if (lane == 0 && wave == 0) p.sweeps[d] = (uint8_t)min(sweeps, 255);
}

template <int LPR, int T>
__global__ __launch_bounds__(64) void lda_estep_reg_kernel(LdaParams p, int64_t first,
                                                           const int* __restrict__ topic_of_lane)
{
    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d];
    lda_estep_reg_body<LPR, T>(p, d, off, (int)(p.doc_ptr[d + 1] - off), topic_of_lane);
}

// long documents: one workgroup of TMVB_LONG_WAVES waves per document (see lda_estep_reg_body, W > 1)
#define TMVB_LONG_WAVES 4
template <int LPR, int T>
__global__ __launch_bounds__(64 * TMVB_LONG_WAVES) void lda_estep_reg_long_kernel(LdaParams p, int64_t first,
                                                                                 const int* __restrict__ topic_of_lane)
{
    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d];
    lda_estep_reg_body<LPR, T, TMVB_LONG_WAVES>(p, d, off, (int)(p.doc_ptr[d + 1] - off), topic_of_lane);
}

// All register-tile documents of a SMALL corpus / shard in one launch: the tile count is read per document
// (wave-uniform) and the matching instantiation of the body is called.  Three kernel tails fewer than one launch
// per tile count; the price is the register allocation of the widest body for every wave, which only pays when
// the launch is latency- rather than occupancy-bound (tmvb_lda_estep uses it below ~2 M tokens).
template <int LPR, int TMAX>
__global__ __launch_bounds__(64) void lda_estep_reg_any_kernel(LdaParams p, int64_t first,
                                                               const int* __restrict__ topic_of_lane)
{
    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d];
    const int N = (int)(p.doc_ptr[d + 1] - off);
    const int tiles = __builtin_amdgcn_readfirstlane((N + 63) >> 6);
    if (TMAX >= 4 && tiles >= 4) lda_estep_reg_body<LPR, (TMAX >= 4 ? 4 : 1)>(p, d, off, N, topic_of_lane);
    else if (TMAX >= 3 && tiles == 3) lda_estep_reg_body<LPR, (TMAX >= 3 ? 3 : 1)>(p, d, off, N, topic_of_lane);
    else if (TMAX >= 2 && tiles == 2) lda_estep_reg_body<LPR, (TMAX >= 2 ? 2 : 1)>(p, d, off, N, topic_of_lane);
    else lda_estep_reg_body<LPR, 1>(p, d, off, N, topic_of_lane);
}

// ------------------------------------------------------------------------------ grid-tile E-step (tmvb_gridtile.h)
// One wave = one document of at most 32 NP unique terms; lane l = (token group a = l >> 2, topic class b = l & 3) holds the
// sub-tile {tokens 16 s + a, s < 2 NP} x {topics 4 j + b, j < LPR} as NP x LPR register pairs (pair = two token slots).
// Same arithmetic per sweep as lda_estep_reg_body (src/LDA.jl:172-175); what differs is where the sums run:
//   s_n   = K eps + sum_i B[n][i] e_i      LPR packed fmas per token pair + a 4-lane all-reduce (2 DPP adds per token slot)
//   g_i   = sum_n w_n B[n][i]              LPR packed fmas per token pair, one add per topic to fold the pair, then the
//                                          16-lane reduce-scatter of LPR + 1 values (sum_n w_n rides along as value LPR)
//   psi(sum gamma): sum_i gamma_i = K eps + sum alpha + C_d in exact arithmetic (sum_i phi_in = 1), a per-document constant --
//                   evaluated once per document, not once per sweep (the fp64 reference's own sum agrees with it to 1e-16).
// W > 1: W waves (one workgroup) share a LONG document of up to 32 NP W unique terms -- wave w holds the tokens
// 32 NP w + 16 s + a in its own register tile, the per-wave partial sums (g_i, sum w) meet in LDS once per sweep (one
// __syncthreads, double-buffered by sweep parity) and every wave then runs the tail redundantly and bit-identically, so the
// exit test stays uniform across the workgroup.
template <int LPR, int NP, int W = 1>
__device__ __forceinline__ void lda_estep_grid_body(const LdaParams& p, const int d, const int64_t off, const int N,
                                                    const int* __restrict__ topic_of_lane)
{
    constexpr int M = LPR + 1;                       // values of the reduce-scatter: LPR topics of the class + sum w
    constexpr int NS = (M + 15) / 16;
    const int lane = threadIdx.x & 63;
    const int wave = (W > 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int a = lane >> 2, b = lane & 3;
    const int K = p.K;
    __shared__ float xch[2][W > 1 ? W : 1][W > 1 ? NS * 64 : 1];      // W > 1: per-wave partial result slots, by sweep parity

    gv2f B[NP][LPR];
    gv2f c[NP];
    {
        const int64_t off0 = N > 0 ? off : 0;               // uniform base + 32-bit lane offsets
        grid_load_tile<LPR, NP>(B, c, p.beta, p.terms + off0, p.counts + off0, N, 32 * NP * wave, a, b);
    }
    // topic role: result slot r of this lane owns topic mytopic[r] (4 j + b for the primary owner of value j; -1 otherwise)
    int mytopic[NS];
    bool on[NS];
    float alpha_eps[NS], elog[NS], elog_old[NS], gam[NS], e[NS];
    float asum_l = 0.0f, csum_l = 0.0f;
#pragma unroll
    for (int r = 0; r < NS; ++r) {
        mytopic[r] = topic_of_lane[r * 64 + lane];
        on[r] = mytopic[r] >= 0 && mytopic[r] < K;
        const int tp = on[r] ? mytopic[r] : 0;                   // unconditional loads (no branch, no early wait), masked after
        const float al_ld = p.alpha[tp], el_ld = p.elog[(int64_t)d * K + tp];
        const float al = on[r] ? al_ld : 0.0f;
        alpha_eps[r] = al + TMVB_EPS_F;
        elog[r] = on[r] ? el_ld : 0.0f;
        elog_old[r] = elog[r]; gam[r] = 0.0f; e[r] = 0.0f;
        asum_l += al;
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) csum_l += (b == 0) ? c[q].x + c[q].y : 0.0f;       // each token once (class 0 of its group)
    // psi(sum_i gamma_i) with sum_i gamma_i = K eps + sum alpha + C_d
    float csum = wave_sum(csum_l);
    if constexpr (W > 1) {                                     // C_d over the waves of the document
        if (lane == 0) xch[1][wave][0] = csum;
        __syncthreads();
        csum = 0.0f;
#pragma unroll
        for (int ww = 0; ww < W; ++ww) csum += xch[1][ww][0];  // (sweep 0 writes buffer 0; buffer 1 is written again only in
    }                                                          //  sweep 1, behind sweep 0's barrier)
    const float gsum = (float)K * TMVB_EPS_F + wave_sum(asum_l) + csum;
    const float dgs = digamma_f(gsum);
    const float vtol2 = p.vtol * p.vtol;
    const float keps4 = 0.25f * (float)K * TMVB_EPS_F;                            // K eps, a quarter per lane of the quad
    constexpr int WS_A = kGridMap<M>.a_of[LPR], WS_R = kGridMap<M>.r_of[LPR];      // owner of sum_n w_n in class 0

    __shared__ __attribute__((aligned(16))) float e_lds_all[W][4][4 * ((LPR + 3) / 4)];   // e by class: [b][j], one copy per wave
    float (*e_lds)[4 * ((LPR + 3) / 4)] = e_lds_all[wave];
    int sweeps = 0;
    for (int v = 0; v < p.viter; ++v) {
        ++sweeps;
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            e[r] = on[r] ? sweep_exp(elog[r]) : 0.0f;                                 // update_phi!, src/LDA.jl:152
            if (mytopic[r] >= 0) e_lds[mytopic[r] & 3][mytopic[r] >> 2] = e[r];
        }
        if constexpr (W > 1) WAVE_PRIVATE_LDS_FENCE(); else WAVE_LDS_FENCE();
        // ---- phase 1: s_n over this lane's LPR topics, then over the quad (K eps rides in as the start value)
        gv2f w[NP];
        grid_phase1<LPR, NP>(B, e_lds[b], keps4, w);
#pragma unroll
        for (int q = 0; q < NP; ++q) w[q] = c[q] * gv2f{__builtin_amdgcn_rcpf(w[q].x), __builtin_amdgcn_rcpf(w[q].y)};
        // ---- phase 2: g_i over this lane's tokens, then over the 16 token groups
        float gv[M];
        grid_phase2<LPR, NP>(B, w, gv);
        {
            gv2f ws = w[0];
#pragma unroll
            for (int q = 1; q < NP; ++q) ws += w[q];
            gv[LPR] = ws.x + ws.y;
        }
        float pr[NS];
        grid_reduce_scatter<M>(gv, pr);
        if constexpr (W > 1) {
            float (*xb)[NS * 64] = xch[v & 1];
#pragma unroll
            for (int r = 0; r < NS; ++r) xb[wave][r * 64 + lane] = pr[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < NS; ++r) {
                float t = 0.0f;
#pragma unroll
                for (int ww = 0; ww < W; ++ww) t += xb[ww][r * 64 + lane];      // fixed order: identical totals in every wave
                pr[r] = t;
            }
        }
        const float wtot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr[WS_R]), 4 * WS_A));
        // ---- tail: one topic per lane
        float dl = 0.0f;
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            gam[r] = fmaf(wtot, TMVB_EPS_F, fmaf(e[r], pr[r], alpha_eps[r]));         // update_gamma!, src/LDA.jl:145
            const float dg = digamma_sweep_f(gam[r]);
            elog_old[r] = elog[r];                                                    // update_Elogtheta!, :137-138
            if (on[r]) {
                elog[r] = dg - dgs;
                const float df = elog[r] - elog_old[r];
                dl = fmaf(df, df, dl);
            }
        }
        const float dist2 = wave_sum(dl);
        if (dist2 < vtol2) break;                                                     // :175, norm < vtol on the squares
    }
    if (wave == 0) {                                                                  // every wave holds the same state
        if (sweeps > 0) {
#pragma unroll
            for (int r = 0; r < NS; ++r) {
                if (on[r]) {
                    p.gamma[(int64_t)d * K + mytopic[r]] = gam[r];
                    p.elog[(int64_t)d * K + mytopic[r]] = elog[r];
                    p.elog_old[(int64_t)d * K + mytopic[r]] = elog_old[r];
                }
                if (mytopic[r] >= 0 && mytopic[r] < p.estride) p.E[(int64_t)d * p.estride + mytopic[r]] = e[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < NS; ++r)
                if (mytopic[r] >= 0 && mytopic[r] < p.estride) p.E[(int64_t)d * p.estride + mytopic[r]] = 0.0f;
        }
        if (lane == 0) p.sweeps[d] = (uint8_t)min(sweeps, 255);
    }
}

template <int LPR, int NP>
__global__ __launch_bounds__(64) void lda_estep_grid_kernel(LdaParams p, int64_t first, const int* __restrict__ topic_of_lane)
{
    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d];
    lda_estep_grid_body<LPR, NP>(p, d, off, (int)(p.doc_ptr[d + 1] - off), topic_of_lane);
}

// (Round 4, measured and dropped: the same body compiled for one wave per SIMD more than the allocator takes by itself --
// __launch_bounds__(64, 5) for NP = 2: 96 VGPRs instead of 102, (64, 4) for NP = 3: 128 instead of 134, five spilled dwords, one reload
// per sweep.  The counters had said the kernels issue a vector instruction only 63 - 81 % of the time when alone; A/B on one box,
// alternating: 1192 / 1201 it/s against 1270 / 1280 without.  The scratch reload sits in the sweep's dependency chain.)
// (Round 4, measured and dropped too: D = 4 documents per wave with the NEXT document's loads in flight under the current one's sweeps -- a
// three-stage pipeline, doc_order three trips ahead, doc_ptr two, the ids and counts one, all held in vector registers until the trip's end so
// that the compiler does not wait for each to move it to a scalar register.  The twelve extra registers cost NP = 4 and 6 a wave per SIMD
// (210 / 280 VGPRs): E-step 0.856 ms against 0.751; restricted to NP <= 3 (168 VGPRs, occupancy unchanged) 0.763 / 0.769 against 0.750 / 0.751;
// NP = 2 alone 0.750.  Three or four resident waves per SIMD already cover the load chain of a starting document; a quarter as many waves, each
// four times as long, only lengthen the launch's tail.  A/B on one box, alternating, run r4x.)

// long documents: one workgroup of W waves per document (lda_estep_grid_body, W > 1)
template <int LPR, int NP, int W>
__global__ __launch_bounds__(64 * W) void lda_estep_grid_long_kernel(LdaParams p, int64_t first, const int* __restrict__ topic_of_lane)
{
    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d];
    lda_estep_grid_body<LPR, NP, W>(p, d, off, (int)(p.doc_ptr[d + 1] - off), topic_of_lane);
}

// ------------------------------------------------------------------------------ two-copy E-step for documents of <= 64 unique terms (round 4)
// A document of at most 64 terms (30 % of SYN-NSF) spends half of its grid-tile sweep on what does NOT shrink with the document: the fold and the
// 16-lane reduce-scatter behind phase 2 (~50 issue slots) and the one-topic-per-lane tail (~60) of a 200-slot sweep.  Here the N x KP tile is held
// TWICE, once per matrix-vector product, each copy in the layout that needs no cross-lane sum at all:
//   copy A  lane = token n, registers = the KP topics of its row        s_n = K eps + sum_i A[i] e_i         (update_phi!'s normaliser, src/LDA.jl:152-153)
//   copy B  lane = topic i, registers = the 64 tokens' values B[n]      g_i = sum_n w_n B[n]                 (update_gamma!, :145)
// e (one value per topic lane) reaches the token lanes and w = c ./ s (one value per token lane) reaches the topic lanes through 256 bytes of
// LDS each: one ds_write_b32 and broadcast ds_read_b128 (all lanes read the same address), so both products are plain packed fmas against
// wave-uniform operands -- 26 + 32 v_pk_fma_f32 at KP = 52 -- and the tail finds g_i already in the lane that owns topic i.  sum_n w_n (the eps term
// of gamma) rides in lane 63 of copy B as a row of ones.  ~130 issue slots per sweep against 200; 116 tile registers + ~40 (three waves per SIMD
// where the NP = 2 grid tile runs four).  Same arithmetic and exit rule as lda_estep_grid_body; the sums run in a different order (fp32: 1e-7).
// Requires KP <= 60 (one topic per lane, lane 63 free).
template <int LPR>
__global__ __launch_bounds__(64) void lda_estep_tt_kernel(LdaParams p, int64_t first)
{
    constexpr int KP = 4 * LPR;
    static_assert(KP <= 60, "lda_estep_tt_kernel: one topic per lane and a free lane 63");
    __shared__ __attribute__((aligned(16))) float e_l[64];
    __shared__ __attribute__((aligned(16))) float w_l[64];
    const int lane = threadIdx.x;
    const int K = p.K;
    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d];
    const int N = (int)(p.doc_ptr[d + 1] - off);                 // <= 64
    // ---- copy A: this lane's token (a lane past the last token reads token 0 and carries count 0 -> weight exactly 0)
    const bool tok_on = lane < N;
    const int64_t off0 = N > 0 ? off : 0;
    const int tm = p.terms[off0 + (tok_on ? lane : 0)];
    const float c = tok_on ? (float)p.counts[off0 + lane] : 0.0f;
    gv2f A[2 * LPR];
    {
        const float4* __restrict__ row = (const float4*)((const char*)p.beta + (uint32_t)tm * (uint32_t)(4 * KP));
#pragma unroll
        for (int q = 0; q < LPR; ++q) { const float4 v = row[q]; A[2 * q] = gv2f{v.x, v.y}; A[2 * q + 1] = gv2f{v.z, v.w}; }
    }
    // ---- copy B: this lane's topic over the 64 token slots (ids broadcast from the token lanes; rows are coalesced 4 KP-byte reads)
    gv2f B[32];
    {
        const uint32_t lo = 4u * (uint32_t)(lane < KP ? lane : 0);
        const char* __restrict__ tb = (const char*)p.beta;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const uint32_t i0 = (uint32_t)__builtin_amdgcn_readlane(tm, 2 * m), i1 = (uint32_t)__builtin_amdgcn_readlane(tm, 2 * m + 1);
            B[m] = gv2f{*(const float*)(tb + (i0 * (uint32_t)(4 * KP) + lo)), *(const float*)(tb + (i1 * (uint32_t)(4 * KP) + lo))};
        }
        if (lane >= KP) {
#pragma unroll
            for (int m = 0; m < 32; ++m) B[m] = (lane == 63) ? gv2f{1.0f, 1.0f} : gv2f{0.0f, 0.0f};      // lane 63: sum_n w_n
        }
    }
    // ---- topic role: lane i < K owns topic i
    const bool on = lane < K;
    const int tp = on ? lane : 0;
    const float al_ld = p.alpha[tp], el_ld = p.elog[(int64_t)d * K + tp];
    const float al = on ? al_ld : 0.0f;
    const float alpha_eps = al + TMVB_EPS_F;
    float elog = on ? el_ld : 0.0f, elog_old = elog, gam = 0.0f, e = 0.0f;
    // psi(sum_i gamma_i) with sum_i gamma_i = K eps + sum alpha + C_d (exact arithmetic: sum_i phi_in = 1)
    const float gsum = (float)K * TMVB_EPS_F + wave_sum(al) + wave_sum(c);
    const float dgs = digamma_f(gsum);
    const float vtol2 = p.vtol * p.vtol;
    const float keps = (float)K * TMVB_EPS_F;
    int sweeps = 0;
    for (int v = 0; v < p.viter; ++v) {
        ++sweeps;
        e = on ? sweep_exp(elog) : 0.0f;                                                  // update_phi!, src/LDA.jl:152
        e_l[lane] = e;
        WAVE_LDS_FENCE();
        // ---- phase 1 (lane = token): s_n = K eps + sum_i A[i] e_i
        gv2f s0 = gv2f{keps, 0.0f}, s1 = gv2f{0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < LPR; ++q) {
            const float4 ev = ((const float4*)e_l)[q];
            s0 = __builtin_elementwise_fma(A[2 * q], gv2f{ev.x, ev.y}, s0);
            s1 = __builtin_elementwise_fma(A[2 * q + 1], gv2f{ev.z, ev.w}, s1);
        }
        const gv2f st = s0 + s1;
        const float w = c * __builtin_amdgcn_rcpf(st.x + st.y);                           // w_n = c_n / s_n
        w_l[lane] = w;
        WAVE_LDS_FENCE();
        // ---- phase 2 (lane = topic): g_i = sum_n w_n B[n]
        gv2f g0 = gv2f{0.0f, 0.0f}, g1 = gv2f{0.0f, 0.0f};
#pragma unroll
        for (int m4 = 0; m4 < 16; ++m4) {
            const float4 wv = ((const float4*)w_l)[m4];
            g0 = __builtin_elementwise_fma(B[2 * m4], gv2f{wv.x, wv.y}, g0);
            g1 = __builtin_elementwise_fma(B[2 * m4 + 1], gv2f{wv.z, wv.w}, g1);
        }
        const gv2f gt = g0 + g1;
        const float g = gt.x + gt.y;
        const float wtot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g), 63));
        // ---- tail: one topic per lane
        gam = fmaf(wtot, TMVB_EPS_F, fmaf(e, g, alpha_eps));                              // update_gamma!, src/LDA.jl:145
        const float dg = digamma_sweep_f(gam);
        elog_old = elog;                                                                  // update_Elogtheta!, :137-138
        float dl = 0.0f;
        if (on) {
            elog = dg - dgs;
            const float df = elog - elog_old;
            dl = df * df;
        }
        if (wave_sum(dl) < vtol2) break;                                                  // :175, norm < vtol on the squares
    }
    if (sweeps > 0) {
        if (on) {
            p.gamma[(int64_t)d * K + lane] = gam;
            p.elog[(int64_t)d * K + lane] = elog;
            p.elog_old[(int64_t)d * K + lane] = elog_old;
        }
        if (lane < p.estride) p.E[(int64_t)d * p.estride + lane] = (lane < KP) ? e : 0.0f;
    } else if (lane < p.estride) {
        p.E[(int64_t)d * p.estride + lane] = 0.0f;
    }
    if (lane == 0) p.sweeps[d] = (uint8_t)min(sweeps, 255);
}

// ------------------------------------------------------------------------------ ELBO
// update_elbo!  src/LDA.jl:83-93 per document (terms :50-80 without the corpus-level constant of
// Elogptheta, added by lda_elbo_final_kernel).  One wave per document, lane = topic.
template <int NSLOT>
__global__ __launch_bounds__(64) void lda_elbo_kernel(int K, int KP, const int64_t* __restrict__ doc_ptr,
                                                      const int32_t* __restrict__ terms, const int32_t* __restrict__ counts,
                                                      const double* __restrict__ alpha_d, const float* __restrict__ beta,
                                                      const float* __restrict__ beta_old, const float* __restrict__ gamma,
                                                      const float* __restrict__ elog, const float* __restrict__ elog_old,
                                                      double* __restrict__ doc_val)
{
    const int lane = threadIdx.x;
    const int d = blockIdx.x;
    const int64_t off = doc_ptr[d];
    const int N = (int)(doc_ptr[d + 1] - off);
    float eo[NSLOT];
    double pc[NSLOT], acc = 0.0;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        int i = lane + 64 * s;
        eo[s] = (i < K) ? expf(elog_old[(int64_t)d * K + i]) : 0.0f;
        pc[s] = 0.0;
    }
    for (int n = 0; n < N; ++n) {
        const int t = terms[off + n];
        const float c = (float)counts[off + n];
        float x[NSLOT], xl = 0.0f;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int i = lane + 64 * s;
            x[s] = (i < K) ? fmaf(beta_old[(int64_t)t * KP + i], eo[s], TMVB_EPS_F) : 0.0f;   // :87
            xl += x[s];
        }
        const float inv = 1.0f / wave_sum(xl);                                               // :88
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int i = lane + 64 * s;
            if (i < K) {
                float ph = x[s] * inv;
                double cp = (double)(c * ph);
                pc[s] += cp;
                acc += cp * (double)logf(beta[(int64_t)t * KP + i] + TMVB_EPS_F);             // Elogpw :65
                if (ph > 0.0f) acc -= cp * (double)logf(ph);                                 // -Elogqz :78
            }
        }
    }
    double gl = 0.0;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        int i = lane + 64 * s;
        if (i < K) {
            double el = (double)elog[(int64_t)d * K + i];
            double g = (double)gamma[(int64_t)d * K + i];
            acc += (alpha_d[i] - 1.0) * el;                                                  // Elogptheta :51 (dot part)
            acc += pc[s] * el;                                                               // Elogpz :58
            if (K > 1) { double ps, lg; digamma_lgamma_d(g, ps, lg); acc += lg - (g - 1.0) * ps; }                          // -Elogqtheta :72 (utils.jl:172-176)
            gl += g;
        }
    }
    const double g0 = wave_sum_d(gl);
    double tot = wave_sum_d(acc);
    if (K > 1) { double ps0, lg0; digamma_lgamma_d(g0, ps0, lg0); tot += -lg0 + (g0 - (double)K) * ps0; }
    if (lane == 0) doc_val[d] = tot;
}

#ifndef TMVB_ELBO_SKIP_TAIL
#define TMVB_ELBO_SKIP_TAIL 0
#endif
// Token-parallel ELBO for K <= 100 (KP = 4 * LPR): lane = token, the beta_old and beta rows of the token in VGPRs,
// e_old_k = exp(Elogtheta_old_k) and Elogtheta_k as SGPR operands (v_readlane with an immediate lane: here lane = topic
// in natural order, the state is read straight from memory), every token term of update_elbo! (src/LDA.jl:58, :65, :78,
// phi rebuilt from beta_old / Elogtheta_old, :87-88) accumulated per lane in fp64 -- no per-token wave reduction, no
// cross-lane traffic until one final sum.  ~10 wave instructions per token instead of ~110 in lda_elbo_kernel.
// WITH_PW = false: E_q[log p(w)] comes from update_beta! (tmvb_lda::d_pw_partial); the beta rows are not read.
template <int LPR, bool WITH_PW>
__global__ __launch_bounds__(64) void lda_elbo_reg_kernel(int K, const int32_t* __restrict__ doc_order, const int64_t* __restrict__ doc_ptr,
                                                          const int32_t* __restrict__ terms, const int32_t* __restrict__ counts,
                                                          const double* __restrict__ alpha_d, const float* __restrict__ beta,
                                                          const float* __restrict__ beta_old, const float* __restrict__ gamma,
                                                          const float* __restrict__ elog, const float* __restrict__ elog_old,
                                                          double* __restrict__ doc_val)
{
    constexpr int R = 4 * LPR, NS = (R + 63) / 64;
    const int lane = threadIdx.x;
    const int d = doc_order[blockIdx.x];                        // longest documents first (doc_val is indexed by document)
    const int64_t off = doc_ptr[d];
    const int N = (int)(doc_ptr[d + 1] - off);
    float eo[NS], el[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = lane + 64 * s;
        eo[s] = (i < K) ? expf(elog_old[(int64_t)d * K + i]) : 0.0f;
        el[s] = (i < K) ? elog[(int64_t)d * K + i] : 0.0f;
    }
    double acc = 0.0;
    for (int n0 = 0; n0 < N; n0 += 64) {
        const int n = n0 + lane;
        const bool in = n < N;
        const int term = in ? terms[off + n] : 0;
        const float c = in ? (float)counts[off + n] : 0.0f;
        const float4* ro = (const float4*)(beta_old + (int64_t)term * R);
        const float4* rn = (const float4*)(beta + (int64_t)term * R);
        float bo[R], bn[WITH_PW ? R : 1];
#pragma unroll
        for (int q = 0; q < LPR; ++q) {
            const float4 a = ro[q];
            bo[4 * q] = a.x; bo[4 * q + 1] = a.y; bo[4 * q + 2] = a.z; bo[4 * q + 3] = a.w;
            if constexpr (WITH_PW) {
                const float4 b = rn[q];
                bn[4 * q] = b.x; bn[4 * q + 1] = b.y; bn[4 * q + 2] = b.z; bn[4 * q + 3] = b.w;
            }
        }
        // With x_k = beta_old_k e_old_k + eps, s = sum_k x_k and phi_k = x_k / s (:87-88), the token's terms are
        //   sum_k phi_k (Elogtheta_k + log(beta_k + eps) - log phi_k) = (1/s) sum_k x_k (Elogtheta_k + log((beta_k + eps) / x_k)) + log s
        // (sum_k phi_k = 1): two v_log_f32 per (token, topic) -- good to 1 ulp of log2, arguments are normal numbers
        // >= eps; libm logf is ~9x the instructions -- and four more VALU instructions.
        float s0 = 0.0f, s1 = 0.0f, t0 = 0.0f, t1 = 0.0f;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (k < R - 7 || k < K) {                                                // KP = 4 * odd >= K: at most 7 pad columns
                const float ek = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eo[k >> 6]), k & 63));
                const float lk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, el[k >> 6]), k & 63));
                const float x = fmaf(bo[k], ek, TMVB_EPS_F);
                const float dl = (WITH_PW ? __builtin_amdgcn_logf(bn[WITH_PW ? k : 0] + TMVB_EPS_F) : 0.0f) - __builtin_amdgcn_logf(x);
                const float u = fmaf(dl, 0.693147180559945f, lk);
                if (k & 1) { s1 += x; t1 = fmaf(x, u, t1); } else { s0 += x; t0 = fmaf(x, u, t0); }
            }
        }
        const float ssum = s0 + s1;
        const float t = fmaf(t0 + t1, 1.0f / ssum, 0.693147180559945f * __builtin_amdgcn_logf(ssum));
        acc += (double)c * (double)t;
    }
    // per-document topic terms, lane = topic
    double gl = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = lane + 64 * s;
        if (i < K) {
            const double g = (double)gamma[(int64_t)d * K + i];
            acc += (alpha_d[i] - 1.0) * (double)el[s];                               // Elogptheta :51 (dot part)
            if (K > 1 && !TMVB_ELBO_SKIP_TAIL) {                                     // -Elogqtheta :72 (utils.jl:172-176)
                // psi and lgamma of gamma from ONE evaluation (digamma_lgamma_d, <= 2e-15 against mpmath): the library's lgamma() +
                // the loop form of digamma_d were ~45 % of this kernel (round 4: 438 -> see DESIGN.md section 2.4)
                double ps, lg;
                digamma_lgamma_d(g, ps, lg);
                acc += lg - (g - 1.0) * ps;
            }
            gl += g;
        }
    }
    const double g0 = wave_sum_d(gl);
    double tot = wave_sum_d(acc);
    if (K > 1 && !TMVB_ELBO_SKIP_TAIL) {
        double ps0, lg0;
        digamma_lgamma_d(g0, ps0, lg0);
        tot += -lg0 + (g0 - (double)K) * ps0;
    }
    if (lane == 0) doc_val[d] = tot;
}

// (Round 4, measured and dropped, both against this kernel at 366 us per call on SYN-NSF K = 50: (i) a table form -- log(beta_old + eps) once per table
// entry and call plus Elogtheta_old per document instead of a v_log_f32 per (token, topic): 1 logarithm instead of 53 in the loop body, four VALU
// instructions per (token, topic) instead of eight -- took 1.58 ms per checked iteration against 1.07: the second gathered row per token costs more
// than the logarithms; (ii) the same sums on the E-step's 16 x 4 lane grid (16 rows per load instruction instead of 64, e_old / Elogtheta as LDS
// broadcast rows instead of 104 v_readlane per tile, packed arithmetic): 1.051 against 1.060 ms, i.e. nothing.  Neither the logarithms nor the
// gather's shape bound it; one wave per document with two dependent global loads in front of the rows and an fp64 tail behind them does.)


// Decomposed update_elbo! (round 5) -- the per-document half.  With x_nk = beta_old[k, v_n] e_old_k + eps, s_n = sum_k x_nk, phi_nk = x_nk / s_n
// (src/LDA.jl:87-88) and pc_k = sum_n c_n phi_nk = gamma_k - alpha_k (update_gamma!, :145, with the alpha the E-step read), the token terms of a document
//     sum_n c_n sum_k phi_nk (Elogtheta_k + log(beta_kv + eps) - log phi_nk)                                                       (:58, :65, :78)
// separate, up to eps inside log x_nk (phi_nk log(1 + eps / (beta_old e_old)) <= eps / s_n ~ 1e-25 per term), into
//     sum_k pc_k (Elogtheta_k - Elogtheta_old_k)                         this kernel, from the state the E-step left
//   + sum_n c_n log s_n                                                  the statistics pass has s_n in a register: one v_log_f32 + FMA per posting,
//                                                                        one partial per chunk (TermStatsParams::logz, log2 units), summed by beta_norm_kernel
//   + sum_{v,k} S_vk (log(beta_new + eps) - log(beta_old + eps))         update_beta! reads S and writes beta_new: beta_norm_kernel's partial
// so a checked iteration needs no second walk over the corpus (lda_elbo_reg_kernel: one gathered row and 2 K logarithms per token, 366 us on SYN-NSF
// K = 50 against a 758 us iteration).  What is left per document is the sum above and the Dirichlet entropy (:72), one psi / lgamma evaluation per
// (document, topic) in fp64; Elogptheta's dot product (:51) sum_d (alpha - 1) . Elogtheta_d = (alpha - 1) . Elogtheta_sum needs the NEW alpha and goes to
// lda_elbo_final_kernel.  Nothing here depends on the M-step, so the E-step enqueues this kernel itself, behind its document kernels on a stream of its own
// (aux[ELBO]), under the statistics pass and beside the side chain (column sums of Elogtheta, update_alpha!): in a row with them it made that chain the
// longest of a checked iteration (timeline: column sums 56 us + this kernel 70 us + update_alpha! 44 us against 135 us of statistics tail and M-step).
// alpha_e is a copy of alpha taken at the start of the E-step (update_alpha! may run while this kernel does).
// Layout: 64 documents per block, FOUR LANES PER DOCUMENT (topics q, q + 4, ...): the documents' rows of gamma / Elogtheta / Elogtheta_old are contiguous
// (64 K floats each), staged through LDS with coalesced loads; every lane evaluates ~K / 4 special functions, no cross-lane traffic but two quad steps for
// sum_k gamma_k (lda_elbo_reg_kernel's lane = topic tail: 2 evaluations per document-wave with 50 of 64 lanes in use).  One fp64 value per block, fixed order.
__global__ __launch_bounds__(256) void lda_copy_d_kernel(const double* __restrict__ src, double* __restrict__ dst, int n)
{
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void lda_elbo_doc_kernel(int K, int64_t M, const double* __restrict__ alpha_e,
                                                           const float* __restrict__ gamma, const float* __restrict__ elog,
                                                           const float* __restrict__ elog_old, double* __restrict__ block_val)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [3][64 K]
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int64_t d0 = (int64_t)blockIdx.x * 64;
    const int nd = (int)min((int64_t)64, M - d0);
    const int n = nd * K, n4 = n >> 2;
    float* sg = sm; float* se = sm + 64 * K; float* so = sm + 128 * K;
    {
        const float4* g4 = (const float4*)(gamma + d0 * K);       // 64 K floats per block: 16-byte aligned
        const float4* e4 = (const float4*)(elog + d0 * K);
        const float4* o4 = (const float4*)(elog_old + d0 * K);
        for (int i = tid; i < n4; i += 256) { ((float4*)sg)[i] = g4[i]; ((float4*)se)[i] = e4[i]; ((float4*)so)[i] = o4[i]; }
        for (int i = 4 * n4 + tid; i < n; i += 256) { sg[i] = gamma[d0 * K + i]; se[i] = elog[d0 * K + i]; so[i] = elog_old[d0 * K + i]; }
    }
    __syncthreads();
    const int dl = tid >> 2, q = tid & 3;
    double acc = 0.0, g0 = 0.0;
    if (dl < nd) {
        const float* rg = sg + dl * K; const float* re = se + dl * K; const float* ro = so + dl * K;
        for (int k = q; k < K; k += 4) {
            const double g = (double)rg[k];
            acc = fma(g - alpha_e[k], (double)re[k] - (double)ro[k], acc);             // Elogpz - Elogqz: the per-document part (see above)
            if (K > 1) { double ps, lg; digamma_lgamma_d(g, ps, lg); acc += lg - (g - 1.0) * ps; }   // -Elogqtheta :72 (utils.jl:172-176)
            g0 += g;
        }
    }
    g0 += __shfl_xor(g0, 1, 64);
    g0 += __shfl_xor(g0, 2, 64);                                  // the quad's lanes hold sum_k gamma_k
    if (K > 1 && dl < nd && q == 0) { double ps0, lg0; digamma_lgamma_d(g0, ps0, lg0); acc += -lg0 + (g0 - (double)K) * ps0; }
    acc = wave_sum_d(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) block_val[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

#define TMVB_REG_ANY_TILES 99   // bucket of register-tile documents with mixed tile counts (lda_estep_reg_any_kernel)
#define TMVB_GRID_ANY_NP 99     // bucket of grid-tile documents with mixed lengths (lda_estep_grid_any_kernel)
#define TMVB_GRID_TT_NP 1       // bucket of documents of <= 64 unique terms on the two-copy kernel (lda_estep_tt_kernel; KP <= 60)

// token pairs per lane of the widest grid-tile instantiation: 2 NP LPR tile registers + 2 LPR accumulators + ~40 must stay
// within the 256 architectural VGPRs.  KP <= 60: 6 pairs (documents of <= 192 unique terms per wave), KP <= 76: 4, KP <= 100: 3.
constexpr int lda_grid_np_max(int lpr) { return lpr <= 15 ? 6 : lpr <= 19 ? 4 : 3; }

// token-pair classes of the grid-tile kernel: a document runs the smallest instantiated NP (2, 3, 4, 6 up to the maximum) that holds it
static inline int lda_grid_np_class(int64_t n, int np_max)
{
    const int64_t np = (n + 31) / 32;
    const int c = np <= 2 ? 2 : np <= 3 ? 3 : np <= 4 ? 4 : 6;
    return c <= np_max ? c : np_max;
}

// All grid-tile documents of a SMALL corpus / shard in one launch (the pair count is read per document, wave-uniform)
template <int LPR>
__global__ __launch_bounds__(64) void lda_estep_grid_any_kernel(LdaParams p, int64_t first, const int* __restrict__ topic_of_lane)
{
    constexpr int NPM = lda_grid_np_max(LPR);
    const int d = p.doc_order[first + blockIdx.x];
    const int64_t off = p.doc_ptr[d];
    const int N = (int)(p.doc_ptr[d + 1] - off);
    const int np = __builtin_amdgcn_readfirstlane((N + 31) >> 5);
    if (NPM >= 6 && np > 4) lda_estep_grid_body<LPR, (NPM >= 6 ? 6 : 2)>(p, d, off, N, topic_of_lane);
    else if (NPM >= 4 && np == 4) lda_estep_grid_body<LPR, (NPM >= 4 ? 4 : 2)>(p, d, off, N, topic_of_lane);
    else if (np == 3) lda_estep_grid_body<LPR, 3>(p, d, off, N, topic_of_lane);
    else lda_estep_grid_body<LPR, 2>(p, d, off, N, topic_of_lane);
}

// (Round 4, measured and dropped: the one mixed launch of a small shard runs every document at the widest body's registers (NP = 6: 230 VGPRs,
// two waves per SIMD).  Splitting it by register need -- NP = 2, 3 on the chain at three waves per SIMD, NP = 4, 6 beside it on aux[0], what
// gained CTPF 7 % (tmvb_ctpf.hip) -- LOSES here: 16 100 documents 0.185 ms per iteration against 0.173, 32 200 documents 0.277 against 0.271
// (A/B on one box, alternating).  LDA's sweeps keep a SIMD busier than CTPF's one- or two-sweep documents do, and the second launch costs a
// cross-stream join in a 0.17 ms iteration.)
template <int LPR>
static void lda_launch_grid(int np, dim3 grid, dim3 block, hipStream_t st, const LdaParams& p, int64_t first, const int* tol)
{
    constexpr int NPM = lda_grid_np_max(LPR);
    if (block.x == 128) { hipLaunchKernelGGL((lda_estep_grid_long_kernel<LPR, NPM, 2>), grid, block, 0, st, p, first, tol); return; }
    if (block.x == 256) { hipLaunchKernelGGL((lda_estep_grid_long_kernel<LPR, NPM, 4>), grid, block, 0, st, p, first, tol); return; }
    if (np == TMVB_GRID_ANY_NP) hipLaunchKernelGGL((lda_estep_grid_any_kernel<LPR>), grid, block, 0, st, p, first, tol);
    else if (np == TMVB_GRID_TT_NP) {
        if constexpr (LPR <= 15) hipLaunchKernelGGL((lda_estep_tt_kernel<LPR>), grid, block, 0, st, p, first);
    }
    else if (np <= 2) hipLaunchKernelGGL((lda_estep_grid_kernel<LPR, 2>), grid, block, 0, st, p, first, tol);
    else if (np == 3) hipLaunchKernelGGL((lda_estep_grid_kernel<LPR, 3>), grid, block, 0, st, p, first, tol);
    else if (np == 4) hipLaunchKernelGGL((lda_estep_grid_kernel<LPR, (NPM >= 4 ? 4 : 3)>), grid, block, 0, st, p, first, tol);
    else hipLaunchKernelGGL((lda_estep_grid_kernel<LPR, (NPM >= 6 ? 6 : NPM)>), grid, block, 0, st, p, first, tol);
}

// register-tile launch for a bucket of `tiles`-tile documents (instantiates T = 1..TMAX only)
template <int LPR, int TMAX>
static void lda_launch_reg(int tiles, dim3 grid, dim3 block, hipStream_t st, const LdaParams& p, int64_t first, const int* tol)
{
    if (block.x > 64) {            // one workgroup of TMVB_LONG_WAVES waves per long document
        if (tiles <= 1) hipLaunchKernelGGL((lda_estep_reg_long_kernel<LPR, 1>), grid, block, 0, st, p, first, tol);
        else if (tiles == 2 || TMAX == 2) hipLaunchKernelGGL((lda_estep_reg_long_kernel<LPR, (TMAX >= 2 ? 2 : 1)>), grid, block, 0, st, p, first, tol);
        else if (tiles == 3 || TMAX == 3) hipLaunchKernelGGL((lda_estep_reg_long_kernel<LPR, (TMAX >= 3 ? 3 : 1)>), grid, block, 0, st, p, first, tol);
        else hipLaunchKernelGGL((lda_estep_reg_long_kernel<LPR, (TMAX >= 4 ? 4 : 1)>), grid, block, 0, st, p, first, tol);
        return;
    }
    if (tiles == TMVB_REG_ANY_TILES) hipLaunchKernelGGL((lda_estep_reg_any_kernel<LPR, TMAX>), grid, block, 0, st, p, first, tol);
    else if (tiles <= 1) hipLaunchKernelGGL((lda_estep_reg_kernel<LPR, 1>), grid, block, 0, st, p, first, tol);
    else if (tiles == 2 || TMAX == 2) hipLaunchKernelGGL((lda_estep_reg_kernel<LPR, (TMAX >= 2 ? 2 : 1)>), grid, block, 0, st, p, first, tol);
    else if (tiles == 3 || TMAX == 3) hipLaunchKernelGGL((lda_estep_reg_kernel<LPR, (TMAX >= 3 ? 3 : 1)>), grid, block, 0, st, p, first, tol);
    else hipLaunchKernelGGL((lda_estep_reg_kernel<LPR, (TMAX >= 4 ? 4 : 1)>), grid, block, 0, st, p, first, tol);
}

// ------------------------------------------------------------------------------ host side
struct tmvb_lda {
    tmvb_ctx* ctx = nullptr;
    tmvb_corpus* corp = nullptr;
    int K = 0, KP = 0, nslot = 1;
    int64_t M = 0, V = 0;
    int64_t M_total = 0;
    bool distributed = false;
    tmvb_comm* comm = nullptr;         // document-sharded train!: the all-reduce of the packed statistics (not owned)
    // device state
    double* d_alpha_d = nullptr;
    float* d_alpha_f = nullptr;
    float* d_beta[2] = {nullptr, nullptr};
    int cur = 0;                       // d_beta[cur] = beta, d_beta[cur^1] = beta_old
    float* d_stats = nullptr;          // S (K*V) | Elogtheta_sum (K)
    // round 5: the LAST statistics pass of the pipelined E-step writes its own buffer (and its own multi-chunk partials), so that it starts behind the last
    // document kernel instead of behind the passes before it (which accumulate in order on aux[0]); update_beta! merges (colsum_merge_partial_kernel)
    float* d_stats_b = nullptr; float* d_ts_partial_b = nullptr;
    bool stats_b_live = false;         // d_stats_b holds statistics that d_stats does not
    bool events_system_scope = false;  // a communicator was attached: the ordering events carry the system-scope fence (tmvb_lda_set_comm)
    bool own_stats = true;
    int lds_limit = -1;                // hipDeviceAttributeMaxSharedMemoryPerBlock of the context's device (read once)
    int elbo_doc_lds_set = 0;          // dynamic-LDS attribute already set on lda_elbo_doc_kernel
    float* d_gamma = nullptr;
    float* d_elog = nullptr;
    float* d_elog_old = nullptr;
    uint8_t* d_sweeps = nullptr;
    float* d_wtok = nullptr;           // [nnz]
    float* d_E = nullptr;              // [M][estride]
    int estride = 0;
    bool e_padded = false;             // E rows are zero padded to >= KP floats (float4 statistics kernels)
    float* d_ts_partial = nullptr;     // [n_slots][K+1]
    int* d_topic_of_lane = nullptr;    // register-tile kernel lane maps
    int* d_grid_topic_of_lane = nullptr;   // grid-tile kernel lane map (tmvb_gridtile.h)
    bool grid_path = false;            // KP <= 100 with recomputed statistics weights: documents of <= 32 grid_np_max terms use lda_estep_grid_kernel
    int grid_np_max = 0;               // lda_grid_np_max(KP / 4)
    bool reg_path = false;             // K <= 64 with a specialised LPR: short documents use lda_estep_reg_kernel
    int32_t* d_doc_order = nullptr;
    double* d_partial = nullptr;       // [TMVB_REDUCE_BLOCKS][K]
    double* d_rowsum = nullptr;        // [K]
    double* d_esum = nullptr;          // [K]
    double* d_doc_val = nullptr;       // [M]
    double* d_elbo = nullptr;          // [1]
    int* d_iters = nullptr;
    double elbo = 0.0;
    std::vector<tmvb_bucket> buckets;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // the length buckets of one E-step are independent: they are issued round-robin on a few
    // auxiliary streams (fork/join on the context's stream) so that their tails overlap
    static constexpr int NAUX = 4;
    hipStream_t aux[NAUX] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[NAUX] = {nullptr, nullptr, nullptr, nullptr};
    // pipelined E-step (large corpora, recomputed-weight statistics): the documents are cut into `pieces`
    // of about equal token count, each with its own inverted index; the HBM-bound statistics pass of piece
    // p runs on the context's stream while the VALU-bound document kernels of piece p+1 run on aux[0].
    std::vector<tmvb_inv_index> pieces;
    std::vector<hipEvent_t> ev_piece;
    // side chain: the Elogtheta column sums depend on the document kernels only, so they run on aux[SIDE]
    // under the statistics pass; update_alpha! (one wave of fp64 Newton steps) runs there under update_beta!.
    // The context's stream joins the side chain lazily (lda_join_side): before anything that reads or rewrites
    // alpha / Elogtheta_sum on it, and before a document-sharded caller all-reduces the statistics buffer;
    // the next E-step's document kernels wait for the side chain directly.
    static constexpr int SIDE = 1;     // aux[1]: the long documents' stream, idle by the time the side chain starts (round 3: was a stream of its own, aux[2])
    hipEvent_t ev_docs = nullptr, ev_side = nullptr, ev_mark = nullptr, ev_chain = nullptr;
    double* d_partial_side = nullptr;
    bool esum_fresh = false;           // d_esum / statistics tail hold the sums of the current Elogtheta
    bool esum_side = false;            // ... and they were produced on the side stream by the last E-step
    bool timing = false;               // TMVB_ESTEP_TIMING=1: record the events behind tmvb_lda_last_estep_ms
    bool mark_valid = false;           // ev_mark was recorded at the entry of the preceding update_beta call
    // E_q[log p(w)] = sum S .* log(beta_new + eps) falls out of update_beta! (the statistics ARE sum_n c_n phi_in), so the
    // ELBO pass right after an iteration needs neither the beta rows nor their logarithms.  A document-sharded rank
    // holds the GLOBAL S after the all-reduce: it contributes the share M / M_total of the global term to its local
    // ELBO, and the shares add up to one in the host's all-reduce of the local values
    double* d_pw_partial = nullptr;    // [2048] per-block partials of the last update_beta!
    int pw_blocks = 0;
    bool stats_fresh = false;          // S holds the statistics of an E-step that update_beta! has not consumed yet
    bool pw_valid = false;             // d_pw_partial belongs to the current (beta, beta_old, Elogtheta_old) state
    // decomposed update_elbo! (lda_elbo_doc_kernel): an iteration that WILL be checked (train!: known before its E-step; TMVB_LDA_ELBO_PARTS=2: every
    // iteration, =0: never) collects the chunks' log-normaliser sums in its statistics passes, keeps the alpha its E-step read, and has update_beta!
    // leave sum S (log beta_new - log beta_old); update_elbo! then needs one per-document kernel instead of a second walk over the corpus.
    bool want_parts = false;           // set for the coming iteration
    int parts_env = 1;                 // TMVB_LDA_ELBO_PARTS at tmvb_lda_create: 0 never, 1 the iterations train! will check, 2 every E-step
    double* d_logz = nullptr; size_t logz_cap = 0; int64_t n_logz = 0;
    double* d_lz_partial = nullptr;    // [2048] per-block sums of d_logz, by beta_norm_kernel
    static constexpr int ELBO = 2;     // aux[2]: lda_elbo_doc_kernel's stream
    double* d_alpha_e = nullptr;       // [K] alpha as the collecting E-step read it (copied at its start on the side stream: ev_acopy)
    hipEvent_t ev_acopy = nullptr, ev_elbo = nullptr;
    bool elbo_pending = false;         // ev_elbo marks lda_elbo_doc_kernel: whoever rewrites gamma / Elogtheta or reads its values waits for it (lda_join_side)
    int64_t n_elbo_blocks = 0;         // d_doc_val[0, n_elbo_blocks): lda_elbo_doc_kernel's values, enqueued by that E-step on its side stream
    bool logz_valid = false;           // d_logz[0, n_logz) and those values belong to the last E-step (every statistics pass of it)
    bool pw_diff = false;              // d_pw_partial holds the (log beta_new - log beta_old) form
    int elbo_form = 0;                 // the last update_elbo!: 1 decomposed, 0 token walk (tmvb_lda_elbo_form)
    bool force_walk = false;           // train!'s like-with-like evaluation at the switch of forms (tmvb_train.h)
    bool side_pending = false;         // ev_side marks side-stream work the context's stream has not waited for yet
    // tmvb_lda_estep_allreduce (document-sharded run): the LAST statistics pass is issued in vocabulary slices, and the slab of S
    // a slice completes is all-reduced on aux[AR] while the next slice's pass runs on the context's stream.  Cuts and order are
    // agreed over the communicator once (lda_ar_prepare) -- every rank issues the same collectives in the same order.
    static constexpr int AR = 3;
    tmvb_comm* ar_comm = nullptr;      // the communicator the plan below was agreed on (nullptr: none yet)
    int ar_slices = 0;                 // <= 1: one all-reduce of the whole buffer
    std::vector<int64_t> ar_cuts;      // slice s = term ids [ar_cuts[s], ar_cuts[s + 1])
    std::vector<int> ar_order;         // the slices in processing order
    tmvb_inv_index ar_index;           // pieces.empty(): the shard's index, slice-major (else pieces.back() is rebuilt in place)
    std::vector<int32_t> doc_piece;    // kept from lda_cut_pieces for that rebuild
    std::vector<hipEvent_t> ev_slice;
    hipEvent_t ev_comm = nullptr, ev_tail = nullptr;
    bool ar_live = false;              // set for the E-step inside tmvb_lda_estep_allreduce
    bool tail_early = false;           // ev_tail (on aux[AR]) marks the all-reduced Elogtheta_sum tail: update_alpha! starts from it
};

static bool lda_reg_lpr_supported(int lpr) { return lpr >= 1 && lpr <= 25 && (lpr & 1); }   // every KP = 4 * odd <= 100
// 64-token register tiles per document: the tile costs T * KP VGPRs of the 512 available per lane
// tiles * KP + KP / 4 + ~35 VGPRs must stay within the 256 architectural VGPRs (beyond that the tile spills to AGPRs;
// measured still ahead of the LDS-tile kernel at KP = 100, T = 3)
static int lda_reg_max_tiles(int lpr) { return lpr <= 13 ? 4 : (lpr <= 17 || lpr == 25) ? 3 : 2; }

static int lda_piece_count(const tmvb_lda* h);

static void lda_build_buckets(tmvb_lda* h, std::vector<int32_t>& order)
{
    const std::vector<int64_t>& len = h->corp->h_doc_len;
    order.resize(h->M);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return len[x] > len[y]; });
    h->buckets.clear();
    const int max_tiles = lda_reg_max_tiles(h->KP / 4);
    const int64_t reg_max = h->grid_path ? 32 * h->grid_np_max : h->reg_path ? 64 * max_tiles : -1;
    // documents longer than reg_max: LDS-tile kernel
    // long documents get up to 156 KiB of LDS (one workgroup per CU): a tile that holds the whole document is
    // gathered once per E-step, a streamed one once per sweep
    // Documents between 64 T and 64 T W unique terms keep the register-tile arithmetic with one workgroup of
    // W = TMVB_LONG_WAVES waves each (lda_estep_reg_long_kernel); only what is longer still goes through LDS.
    const bool long_reg = h->reg_path && getenv("TMVB_LDA_NO_LONG") == nullptr;
    const int64_t long_max = long_reg ? (int64_t)64 * max_tiles * TMVB_LONG_WAVES : reg_max;
    int64_t pos = tmvb_build_lds_buckets(len, order, h->M, h->KP, long_max, 3, h->buckets, TMVB_BIG_TILE_BYTES);
    // grid-tile kernel with 4 / 2 waves per document: up to 768 / 384 unique terms
    const int64_t grid_long_max = (h->grid_path && long_reg) ? (int64_t)32 * h->grid_np_max * 4 : 0;
    if (long_reg) {
        for (int T = max_tiles; T >= 1 && pos < h->M; --T) {   // (KP = 100, T = 3 spills to AGPRs and is still 2.8x the LDS kernel)
            const int64_t lo = std::max<int64_t>(std::max<int64_t>(64 * TMVB_LONG_WAVES * (int64_t)(T - 1), reg_max), grid_long_max);
            int64_t cnt = 0;
            while (pos + cnt < h->M && len[order[pos + cnt]] > lo) ++cnt;
            if (cnt) { tmvb_bucket b{pos, cnt, 0, T}; b.waves = TMVB_LONG_WAVES; h->buckets.push_back(b); }
            pos += cnt;
        }
        for (int Wv = 4; Wv >= 2 && grid_long_max > 0 && pos < h->M; Wv >>= 1) {
            const int64_t lo = (int64_t)32 * h->grid_np_max * (Wv / 2);
            int64_t cnt = 0;
            while (pos + cnt < h->M && len[order[pos + cnt]] > lo) ++cnt;
            if (cnt) { tmvb_bucket b{pos, cnt, 0, 1}; b.waves = Wv; b.grid_np = h->grid_np_max; h->buckets.push_back(b); }
            pos += cnt;
        }
    }
    // small corpora (one statistics pass): ONE launch for all register-tile documents, longest first
    // (KP <= 60 only: with two result slots per lane the widest body costs the short documents a wave per SIMD --
    //  measured at KP = 100, 16 k documents: 2.14 k it/s merged, 2.26 k separate)
    if (h->reg_path && h->KP <= 60 && pos < h->M && lda_piece_count(h) == 1 && getenv("TMVB_LDA_NO_MERGE") == nullptr) {
        tmvb_bucket b{pos, h->M - pos, 0, TMVB_REG_ANY_TILES};
        if (h->grid_path) b.grid_np = TMVB_GRID_ANY_NP;
        h->buckets.push_back(b);
        return;
    }
    if (h->grid_path) {
        // grid-tile buckets, longest first: one launch per instantiated pair count
        // (round 4) documents of <= 64 unique terms on the two-copy kernel (lda_estep_tt_kernel, KP <= 60): OFF by default -- measured A/B on
        // SYN-NSF K = 50, alternating on one box: 1256 / 1267 it/s with it against 1287 / 1288 without (its ~125 VALU instructions per sweep
        // against 200 are paid for with 29 dependent LDS broadcast reads per sweep and three waves per SIMD instead of four).  TMVB_LDA_TT=1
        // selects it (read per model: tests/test_lda_gpu.py runs it against the oracle).
        const char* tt_e = getenv("TMVB_LDA_TT");
        const bool tt = tt_e && atoi(tt_e) != 0 && h->KP <= 60 && h->estride <= 64;
        auto cls = [&](int64_t n) { return (tt && n <= 64) ? TMVB_GRID_TT_NP : lda_grid_np_class(n, h->grid_np_max); };
        while (pos < h->M) {
            const int np = cls(len[order[pos]]);
            int64_t cnt = 0;
            while (pos + cnt < h->M && cls(len[order[pos + cnt]]) == np) ++cnt;
            tmvb_bucket b{pos, cnt, 0, (np + 1) / 2};        // reg_tiles > 0 marks a register bucket (chain stream, piece cuts)
            b.grid_np = np;
            h->buckets.push_back(b);
            pos += cnt;
        }
        return;
    }
    // register-tile buckets: T = ceil(N / 64) tiles of 64 tokens
    for (int T = max_tiles; T >= 1 && pos < h->M; --T) {
        const int64_t lo = 64 * (int64_t)(T - 1);
        int64_t cnt = 0;
        while (pos + cnt < h->M && (len[order[pos + cnt]] > lo || T == 1)) ++cnt;
        if (cnt) h->buckets.push_back({pos, cnt, 0, T});
        pos += cnt;
    }
}

static int lda_piece_count(const tmvb_lda* h)
{
    if (!tmvb_termstats_recomputes(h->KP, h->e_padded) || !h->reg_path) return 1;
    if (const char* e = getenv("TMVB_LDA_PIECES")) return std::max(1, std::min(16, atoi(e)));
    // measured on SYN-NSF and its shards in the steady state (tools/run_pieces.sh, it/s): 10.9 M tokens 1 / 2 / 3 / 4 / 6 pieces
    // = 717 / 806 / 813 / 803 / ~790; 5.5 M tokens 1354 / 1413 / 1426; 2.7 M tokens 2416 / 2229; 1.4 M tokens 3926 / 3303.
    // Below ~4 M tokens the one-pass plan (critical chain on the context's stream, no cross-stream hops) wins.
    // K = 100 (LPR = 25; round 4, alternating in one call, SYN-NSF): 2 pieces 871 / 874 it/s, 3 pieces 850 / 852, 4 pieces 867 / 865, 5 pieces 829
    const int64_t nnz = h->corp->info.nnz;
    if (nnz < (int64_t)(1 << 22)) return 1;
    return h->KP / 4 >= 25 ? 2 : 3;
}

// Cut the register-tile buckets where the running token count crosses a multiple of nnz / P.  The LDS-tile
// (long-document) buckets go with the LAST piece: they are issued first, on their own stream, and have the
// whole E-step to finish.  Fills doc_piece[d] and replaces h->buckets.
static void lda_cut_pieces(tmvb_lda* h, const std::vector<int32_t>& order, int P, std::vector<int32_t>& doc_piece)
{
    const std::vector<int64_t>& len = h->corp->h_doc_len;
    const int64_t nnz = std::max<int64_t>(h->corp->info.nnz, 1);
    // cumulative token fractions at which a piece ends: the last pieces are smaller, because the statistics
    // pass of the last piece has no document kernels left to hide under
    std::vector<double> cum(P);
    {
        double tot = 0.0;
        for (int q = 0; q < P; ++q) { cum[q] = 1.0 + 0.5 * (double)(P - 1 - q) / (double)std::max(P - 1, 1); tot += cum[q]; }
        double run = 0.0;
        for (int q = 0; q < P; ++q) { run += cum[q] / tot; cum[q] = run; }
        if (const char* e = getenv("TMVB_LDA_PIECE_FRACS")) {          // tuning: "0.3,0.58,0.82"
            std::vector<double> f;
            for (const char* c = e; *c;) { char* end; f.push_back(strtod(c, &end)); c = (*end == ',') ? end + 1 : end; if (end == c && *c) break; }
            for (int q = 0; q + 1 < P && q < (int)f.size(); ++q) cum[q] = f[q];
        }
        cum[P - 1] = 2.0;
    }
    auto piece_of = [&](int64_t run) { const double x = (double)run / (double)nnz; int q = 0; while (q < P - 1 && x >= cum[q]) ++q; return q; };
    doc_piece.assign(h->M, 0);
    std::vector<tmvb_bucket> cut;
    int64_t run = 0;
    for (const tmvb_bucket& b : h->buckets) {
        if (b.reg_tiles == 0 || b.waves > 1) {               // LDS-tile and workgroup-per-document buckets: last piece, aux[1]
            tmvb_bucket c = b; c.piece = P - 1;
            for (int64_t q = b.first; q < b.first + b.count; ++q) doc_piece[order[q]] = P - 1;
            cut.push_back(c);
            continue;
        }
        int64_t start = b.first;
        int piece = piece_of(run);
        for (int64_t q = b.first; q < b.first + b.count; ++q) {
            const int pq = piece_of(run);
            if (pq != piece) {
                if (q > start) { tmvb_bucket c = b; c.first = start; c.count = q - start; c.piece = piece; cut.push_back(c); }
                start = q; piece = pq;
            }
            doc_piece[order[q]] = pq;
            run += len[order[q]];
        }
        if (b.first + b.count > start) { tmvb_bucket c = b; c.first = start; c.count = b.first + b.count - start; c.piece = piece; cut.push_back(c); }
    }
    h->buckets.swap(cut);
}

static int lda_join_side(tmvb_lda* h)
{
    if (h->side_pending) {
        TMVB_HIP(hipStreamWaitEvent(h->ctx->stream, h->ev_side, 0));
        h->side_pending = false;
    }
    if (h->elbo_pending) {                                  // lda_elbo_doc_kernel on aux[ELBO] reads gamma / Elogtheta / Elogtheta_old and writes d_doc_val
        TMVB_HIP(hipStreamWaitEvent(h->ctx->stream, h->ev_elbo, 0));
        h->elbo_pending = false;
    }
    return TMVB_OK;
}

// the statistics as ONE buffer again (for whoever hands d_stats out before update_beta!, which merges on its own way)
static int lda_merge_stats(tmvb_lda* h)
{
    if (!h->stats_b_live) return TMVB_OK;
    const int64_t n = (int64_t)h->K * h->V;
    const int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (n + 255) / 256));
    hipLaunchKernelGGL(stats_merge_kernel, dim3(nb), dim3(256), 0, h->ctx->stream, h->d_stats, h->d_stats_b, n);
    TMVB_HIP(hipGetLastError());
    h->stats_b_live = false;
    return TMVB_OK;
}

extern "C" int tmvb_lda_destroy(tmvb_lda* h)
{
    if (!h) return TMVB_OK;
    if (h->ctx) (void)hipSetDevice(h->ctx->device);
    for (int a = 0; a < tmvb_lda::NAUX; ++a) if (h->aux[a]) (void)hipStreamSynchronize(h->aux[a]);
    if (h->ctx) (void)hipStreamSynchronize(h->ctx->stream);
    (void)hipFree(h->d_alpha_d); (void)hipFree(h->d_alpha_f); (void)hipFree(h->d_beta[0]); (void)hipFree(h->d_beta[1]);
    (void)hipFree(h->d_stats_b); (void)hipFree(h->d_ts_partial_b);
    if (h->own_stats) (void)hipFree(h->d_stats);
    (void)hipFree(h->d_wtok); (void)hipFree(h->d_E); (void)hipFree(h->d_ts_partial);
    (void)hipFree(h->d_topic_of_lane); (void)hipFree(h->d_grid_topic_of_lane);
    (void)hipFree(h->d_gamma); (void)hipFree(h->d_elog); (void)hipFree(h->d_elog_old); (void)hipFree(h->d_sweeps);
    (void)hipFree(h->d_doc_order); (void)hipFree(h->d_partial); (void)hipFree(h->d_rowsum); (void)hipFree(h->d_esum);
    (void)hipFree(h->d_doc_val); (void)hipFree(h->d_elbo); (void)hipFree(h->d_iters);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_docs) (void)hipEventDestroy(h->ev_docs);
    if (h->ev_side) (void)hipEventDestroy(h->ev_side);
    if (h->ev_chain) (void)hipEventDestroy(h->ev_chain);
    if (h->ev_acopy) (void)hipEventDestroy(h->ev_acopy);
    if (h->ev_elbo) (void)hipEventDestroy(h->ev_elbo);
    if (h->ev_mark) (void)hipEventDestroy(h->ev_mark);
    (void)hipFree(h->d_partial_side); (void)hipFree(h->d_pw_partial); (void)hipFree(h->d_logz); (void)hipFree(h->d_lz_partial); (void)hipFree(h->d_alpha_e);
    for (tmvb_inv_index& ix : h->pieces) tmvb_free_inv_index(&ix);
    if (h->ar_index.built) tmvb_free_inv_index(&h->ar_index);
    for (hipEvent_t e : h->ev_slice) if (e) (void)hipEventDestroy(e);
    if (h->ev_comm) (void)hipEventDestroy(h->ev_comm);
    if (h->ev_tail) (void)hipEventDestroy(h->ev_tail);
    for (hipEvent_t e : h->ev_piece) if (e) (void)hipEventDestroy(e);
    for (int a = 0; a < tmvb_lda::NAUX; ++a) {
        if (h->ev_join[a]) (void)hipEventDestroy(h->ev_join[a]);
        tmvb_release_stream(h->aux[a]); h->aux[a] = nullptr;        // pooled streams stay (tmvb_pool_stream)
    }
    delete h;
    return TMVB_OK;
}

// host K x V (column-major) <-> device [V][KP] padded
static int upload_beta(tmvb_lda* h, float* dst, const double* src)
{
    const size_t K = h->K, KP = h->KP, V = h->V;
    std::vector<float> tmp(V * KP + 4, 0.0f);
    for (size_t j = 0; j < V; ++j)
        for (size_t i = 0; i < K; ++i) tmp[j * KP + i] = (float)src[j * K + i];
    TMVB_HIP(hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    return TMVB_OK;
}

static int download_beta(tmvb_lda* h, double* dst, const float* src)
{
    const size_t K = h->K, KP = h->KP, V = h->V;
    std::vector<float> tmp(V * KP);
    TMVB_HIP(hipMemcpyAsync(tmp.data(), src, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    for (size_t j = 0; j < V; ++j)
        for (size_t i = 0; i < K; ++i) dst[j * K + i] = (double)tmp[j * KP + i];
    return TMVB_OK;
}

extern "C" int tmvb_lda_create(tmvb_ctx* ctx, tmvb_corpus* corp, int32_t K, tmvb_lda** out)
{
    TMVB_REQUIRE(out != nullptr, TMVB_EINVAL, "tmvb_lda_create: out is NULL");
    *out = nullptr;
    TMVB_REQUIRE(ctx && corp, TMVB_EINVAL, "tmvb_lda_create: NULL context or corpus");
    TMVB_REQUIRE(K > 0, TMVB_EINVAL, "number of topics must be a positive integer.");   // src/gpuLDA.jl:47
    TMVB_REQUIRE(K <= 64 * TMVB_MAX_NSLOT, TMVB_EINVAL, "tmvb_lda_create: K=%d exceeds the supported maximum %d", K, 64 * TMVB_MAX_NSLOT);
    TMVB_HIP(hipSetDevice(ctx->device));
    tmvb_lda* h = new tmvb_lda();
    tmvb_create_guard<tmvb_lda, tmvb_lda_destroy> guard{h};      // every early return below destroys h
    h->ctx = ctx; h->corp = corp; h->K = K; h->KP = tmvb_kpad(K); h->nslot = (K + 63) / 64;
    h->M = corp->info.M; h->V = corp->info.V; h->M_total = h->M;
    h->e_padded = h->KP / 4 <= 64;
    // padded rows start on 128-byte lines when they span more than one (K = 50: 52 -> 64 floats)
    h->estride = h->e_padded ? (h->KP > 32 ? (h->KP + 31) / 32 * 32 : h->KP) : K;
    const size_t KV = (size_t)K * h->V, KM = (size_t)K * h->M;
    const size_t KPV = (size_t)h->KP * h->V + 4;      // padded gather layout (+ slack for 16-byte reads)
    int rc;
    if ((rc = dmalloc(&h->d_alpha_d, K)) || (rc = dmalloc(&h->d_alpha_f, K)) || (rc = dmalloc(&h->d_beta[0], KPV)) ||
        (rc = dmalloc(&h->d_beta[1], KPV)) || (rc = dmalloc(&h->d_stats, KV + K)) || (rc = dmalloc(&h->d_gamma, KM)) ||
        (rc = dmalloc(&h->d_elog, KM)) || (rc = dmalloc(&h->d_elog_old, KM)) || (rc = dmalloc(&h->d_sweeps, (size_t)h->M)) ||
        (rc = dmalloc(&h->d_doc_order, (size_t)h->M)) || (rc = dmalloc(&h->d_partial, (size_t)TMVB_REDUCE_BLOCKS * K)) ||
        (rc = dmalloc(&h->d_rowsum, K)) || (rc = dmalloc(&h->d_esum, K)) ||  (rc = dmalloc(&h->d_doc_val, (size_t)h->M)) ||
        (rc = dmalloc(&h->d_elbo, 1)) || (rc = dmalloc(&h->d_iters, 1)) || (rc = dmalloc(&h->d_wtok, tmvb_termstats_recomputes(h->KP, h->KP / 4 <= 64) ? (size_t)1 : (size_t)corp->info.nnz)) ||   // stored weights: K > 128 only
       
        (rc = dmalloc(&h->d_E, (size_t)((h->KP + 31) / 32 * 32) * h->M + 4))) {
        return rc;
    }
    h->reg_path = (K <= 128) && lda_reg_lpr_supported(h->KP / 4);
    if (h->reg_path) {
        std::vector<int> tol, lot;
        tmvb_reg_lane_maps(h->KP, tol, lot);
        if ((rc = dmalloc(&h->d_topic_of_lane, tol.size()))) return rc;
        TMVB_HIP(hipMemcpy(h->d_topic_of_lane, tol.data(), tol.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    // grid-tile kernel (tmvb_gridtile.h): KP <= 100 (LPR <= 25; up to 6 / 4 / 3 token pairs per lane for KP <= 60 / 76 / 100 within 256 VGPRs), statistics pass
    // recomputes the token weights (nothing per token is stored); TMVB_LDA_GRID=0 keeps the lane = token register tile
    h->grid_np_max = lda_grid_np_max(h->KP / 4);
    h->grid_path = h->reg_path && h->KP <= 100 && tmvb_termstats_recomputes(h->KP, h->e_padded) &&
                   (uint64_t)h->V * (uint64_t)h->KP * 4u < (1ull << 32) &&            // rows are addressed by 32-bit byte offsets
                   !(getenv("TMVB_LDA_GRID") && atoi(getenv("TMVB_LDA_GRID")) == 0);
    if (h->grid_path) {
        std::vector<int> tol;
        switch (h->KP / 4) {
#define LDA_GRID_MAP_CASE(LPRV) case LPRV: tmvb_grid_lane_map_fill<LPRV>(tol); break;
            LDA_GRID_MAP_CASE(1) LDA_GRID_MAP_CASE(3) LDA_GRID_MAP_CASE(5) LDA_GRID_MAP_CASE(7) LDA_GRID_MAP_CASE(9) LDA_GRID_MAP_CASE(11)
            LDA_GRID_MAP_CASE(13) LDA_GRID_MAP_CASE(15) LDA_GRID_MAP_CASE(17) LDA_GRID_MAP_CASE(19) LDA_GRID_MAP_CASE(21) LDA_GRID_MAP_CASE(23)
            LDA_GRID_MAP_CASE(25)
#undef LDA_GRID_MAP_CASE
            default: h->grid_path = false;
        }
        if (h->grid_path) {
            if ((rc = dmalloc(&h->d_grid_topic_of_lane, tol.size()))) return rc;
            TMVB_HIP(hipMemcpy(h->d_grid_topic_of_lane, tol.data(), tol.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    std::vector<int32_t> order;
    lda_build_buckets(h, order);
    {
        const int P = lda_piece_count(h);
        size_t slots = 0;
        if (P > 1) {
            std::vector<int32_t>& doc_piece = h->doc_piece;
            lda_cut_pieces(h, order, P, doc_piece);
            h->pieces.resize(P);
            h->ev_piece.assign(P, nullptr);
            // the pieces' inverted indices are independent host work (a counting sort over the corpus each, ~50 ms at SYN-NSF's size): one thread per piece
            // (tmvb_lda_create took 150-180 ms of a 265 ms gpuLDA(corp, K) call with them in a row, profiles/r6_train_end_to_end.txt; TMVB_CREATE_THREADS=0: in a row)
            static const bool create_threads = [] { const char* e = getenv("TMVB_CREATE_THREADS"); return !(e && atoi(e) == 0); }();
            std::vector<int> prc((size_t)P, TMVB_OK);
            std::vector<std::string> pmsg((size_t)P);
            auto build_piece = [&](int q) {
                (void)hipSetDevice(ctx->device);
                prc[(size_t)q] = tmvb_build_inv_index(ctx, h->M, h->V, corp->h_doc_ptr.data(), corp->h_terms.data(), corp->h_counts.data(),
                                                      &h->pieces[(size_t)q], doc_piece.data(), q);
                if (prc[(size_t)q]) pmsg[(size_t)q] = tmvb_last_error();          // (the error text is thread-local)
            };
            if (create_threads) {
                std::vector<std::thread> workers;
                for (int q = 1; q < P; ++q) workers.emplace_back(build_piece, q);
                build_piece(0);
                for (std::thread& t : workers) t.join();
            } else {
                for (int q = 0; q < P; ++q) build_piece(q);
            }
            for (int q = 0; q < P && !rc; ++q) {
                if (prc[(size_t)q]) { rc = prc[(size_t)q]; tmvb_set_error("%s", pmsg[(size_t)q].c_str()); break; }
                slots = std::max(slots, (size_t)h->pieces[q].n_slots);
                if (hipEventCreateWithFlags(&h->ev_piece[q], tmvb_event_flags()) != hipSuccess) rc = TMVB_EHIP;
            }
        } else {
            rc = tmvb_corpus_term_index(corp);
            slots = (size_t)corp->term_index.n_slots;
            h->ev_piece.assign(1, nullptr);
            if (!rc && hipEventCreateWithFlags(&h->ev_piece[0], tmvb_event_flags()) != hipSuccess) rc = TMVB_EHIP;
        }
        if (!rc) rc = dmalloc(&h->d_ts_partial, slots * (K + 1));
        if (rc) return rc;
    }
    if (h->M) TMVB_HIP(hipMemcpyAsync(h->d_doc_order, order.data(), (size_t)h->M * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_stats, 0, (KV + K) * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_sweeps, 0, std::max<size_t>((size_t)h->M, 1), ctx->stream));
    { const char* t = getenv("TMVB_ESTEP_TIMING"); h->timing = t && atoi(t) != 0; }
    TMVB_HIP(hipEventCreate(&h->ev0));
    TMVB_HIP(hipEventCreate(&h->ev1));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_fork, tmvb_event_flags()));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_docs, tmvb_event_flags()));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_side, tmvb_event_flags()));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_chain, tmvb_event_flags()));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_acopy, tmvb_event_flags()));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_elbo, tmvb_event_flags()));
    TMVB_HIP(hipEventCreateWithFlags(&h->ev_mark, hipEventDisableTiming /* may sit behind a collective: keeps the system-scope fence */));
    { const char* e = getenv("TMVB_LDA_ELBO_PARTS"); h->parts_env = e ? atoi(e) : 1; }
    if ((rc = dmalloc(&h->d_partial_side, (size_t)TMVB_REDUCE_BLOCKS * K)) || (rc = dmalloc(&h->d_pw_partial, 2048)) || (rc = dmalloc(&h->d_lz_partial, 2048)) || (rc = dmalloc(&h->d_alpha_e, (size_t)K))) return rc;
    for (int a = 0; a < tmvb_lda::NAUX; ++a) {
        h->aux[a] = tmvb_pool_stream(ctx->device, 1 + a);
        TMVB_REQUIRE(h->aux[a] != nullptr, TMVB_EHIP, "hipStreamCreate failed");
        TMVB_HIP(hipEventCreateWithFlags(&h->ev_join[a], tmvb_event_flags()));
    }
    TMVB_HIP(hipMemsetAsync(h->d_beta[0], 0, KPV * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_beta[1], 0, KPV * sizeof(float), ctx->stream));
    TMVB_HIP(hipMemsetAsync(h->d_E, 0, ((size_t)((h->KP + 31) / 32 * 32) * h->M + 4) * sizeof(float), ctx->stream));
    // constructor state, src/gpuLDA.jl:53-61 (beta: uniform, see tmvb.h)
    std::vector<double> alpha(K, 1.0), beta(KV, h->V ? 1.0 / (double)h->V : 0.0), gamma(KM, 1.0);
    const double e0 = -0.5772156649015329 - tmvb_digamma_host((double)K);   // -(eulergamma + digamma(K)), src/gpuLDA.jl:57
    std::vector<double> elog(KM, e0);
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    rc = tmvb_lda_set_state(h, alpha.data(), beta.data(), nullptr, gamma.data(), elog.data(), nullptr, nullptr);
    if (rc) return rc;
    guard.release();
    *out = h;
    return TMVB_OK;
}

extern "C" int tmvb_lda_set_state(tmvb_lda* h, const double* alpha, const double* beta, const double* beta_old,
                                  const double* gamma, const double* Elogtheta, const double* Elogtheta_old,
                                  const double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_set_state: handle is NULL");
    h->mark_valid = false; h->esum_fresh = false; h->esum_side = false; h->pw_valid = false; h->stats_fresh = false; h->logz_valid = false;
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    { int jrc = lda_join_side(h); if (jrc) return jrc; }
    const size_t K = h->K, KM = K * (size_t)h->M;
    int rc;
    if (alpha) {
        for (size_t i = 0; i < K; ++i)   // check_model: alpha finite and positive (src/modelutils.jl:262-264)
            TMVB_REQUIRE(std::isfinite(alpha[i]) && alpha[i] > 0.0, TMVB_ENONFINITE, "alpha must be finite and positive.");
        TMVB_HIP(hipMemcpyAsync(h->d_alpha_d, alpha, K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if ((rc = upload_f32(ctx, h->d_alpha_f, alpha, K))) return rc;
    }
    if (beta) {
        if ((rc = upload_beta(h, h->d_beta[h->cur], beta))) return rc;
        if (!beta_old && (rc = upload_beta(h, h->d_beta[h->cur ^ 1], beta))) return rc;
    }
    if (beta_old && (rc = upload_beta(h, h->d_beta[h->cur ^ 1], beta_old))) return rc;
    if (gamma && (rc = upload_f32(ctx, h->d_gamma, gamma, KM))) return rc;
    if (Elogtheta) {
        if ((rc = upload_f32(ctx, h->d_elog, Elogtheta, KM))) return rc;
        if (!Elogtheta_old && (rc = upload_f32(ctx, h->d_elog_old, Elogtheta, KM))) return rc;
    }
    if (Elogtheta_old && (rc = upload_f32(ctx, h->d_elog_old, Elogtheta_old, KM))) return rc;
    if (elbo) h->elbo = *elbo;
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_lda_get_state(tmvb_lda* h, double* alpha, double* beta, double* beta_old, double* gamma,
                                  double* Elogtheta, double* Elogtheta_old, double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_get_state: handle is NULL");
    h->mark_valid = false;
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    { int jrc = lda_join_side(h); if (jrc) return jrc; }
    const size_t K = h->K, KM = K * (size_t)h->M;
    int rc;
    if (alpha) {
        TMVB_HIP(hipMemcpyAsync(alpha, h->d_alpha_d, K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        TMVB_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (beta && (rc = download_beta(h, beta, h->d_beta[h->cur]))) return rc;
    if (beta_old && (rc = download_beta(h, beta_old, h->d_beta[h->cur ^ 1]))) return rc;
    if (gamma && (rc = download_f32(ctx, gamma, h->d_gamma, KM))) return rc;
    if (Elogtheta && (rc = download_f32(ctx, Elogtheta, h->d_elog, KM))) return rc;
    if (Elogtheta_old && (rc = download_f32(ctx, Elogtheta_old, h->d_elog_old, KM))) return rc;
    if (elbo) *elbo = h->elbo;
    return TMVB_OK;
}

static int lda_estep_impl(tmvb_lda* h, int32_t viter, double vtol);

// stream of the side chain (TMVB_LDA_SIDE_STREAM: experiment with the stream -> hardware-queue mapping)
static int lda_side_index()
{
    static const int ix = [] { const char* e = getenv("TMVB_LDA_SIDE_STREAM"); const int v = e ? atoi(e) : (int)tmvb_lda::SIDE; return (v >= 0 && v < (int)tmvb_lda::NAUX) ? v : (int)tmvb_lda::SIDE; }();
    return ix;
}

extern "C" int tmvb_lda_estep(tmvb_lda* h, int32_t viter, double vtol)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_estep: handle is NULL");
    TMVB_REQUIRE(viter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative.");     // src/gpuLDA.jl:350
    TMVB_REQUIRE(vtol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");      // src/gpuLDA.jl:349
    const int rc = lda_estep_impl(h, viter, vtol);
    if (rc != TMVB_OK) {
        // a launch or an event call failed half way: drain the forked streams and drop the half-updated bookkeeping so
        // that the handle is in a defined state (the statistics are not valid; the caller sees the error)
        const std::string msg = tmvb_last_error();
        for (int a = 0; a < tmvb_lda::NAUX; ++a) if (h->aux[a]) (void)hipStreamSynchronize(h->aux[a]);
        (void)hipStreamSynchronize(h->ctx->stream);
        (void)hipGetLastError();
        h->side_pending = false; h->esum_fresh = false; h->esum_side = false; h->stats_fresh = false; h->pw_valid = false; h->mark_valid = false; h->logz_valid = false; h->elbo_pending = false;
        if (h->d_stats_b) {                                 // the last pass's own buffer may hold a partial pass: the next E-step's merge must not fold it in
            (void)hipMemsetAsync(h->d_stats_b, 0, (size_t)h->K * h->V * sizeof(float), h->ctx->stream);
            (void)hipStreamSynchronize(h->ctx->stream);
            (void)hipGetLastError();
        }
        h->stats_b_live = false;
        tmvb_set_error("%s", msg.c_str());
    }
    return rc;
}

static int lda_estep_impl(tmvb_lda* h, int32_t viter, double vtol)
{
    tmvb_ctx* ctx = h->ctx;
    h->mark_valid = false; h->esum_fresh = false; h->esum_side = false; h->tail_early = false;
    TMVB_HIP(hipSetDevice(ctx->device));
    LdaParams p;
    p.K = h->K; p.KP = h->KP; p.LPR = h->KP / 4; p.lpr_magic = (unsigned)(0x100000000ull / (unsigned)p.LPR) + 1u; p.V = h->V;
    p.doc_ptr = h->corp->d_doc_ptr; p.terms = h->corp->d_terms; p.counts = h->corp->d_counts;
    p.doc_order = h->d_doc_order;
    p.alpha = h->d_alpha_f; p.beta = h->d_beta[h->cur]; p.wtok = h->d_wtok; p.tok_inv = h->pieces.empty() ? h->corp->term_index.d_inv : nullptr; p.E = h->d_E; p.estride = h->estride;
    p.gamma = h->d_gamma; p.elog = h->d_elog; p.elog_old = h->d_elog_old; p.sweeps = h->d_sweeps;
    p.viter = viter; p.vtol = (float)vtol;
    { const char* dbg = getenv("TMVB_DEBUG_FLAGS"); p.debug = dbg ? atoi(dbg) : 0; }
    p.store_w = tmvb_termstats_recomputes(h->KP, h->e_padded) ? 0 : 1;
    if (h->timing) TMVB_HIP(hipEventRecord(h->ev0, ctx->stream));
    // Stream plan: the register-tile buckets run back to back on aux[0] (piece after piece), the LDS-tile
    // buckets (long documents; every bucket when K has no register-tile instantiation) on aux[1] -- a single
    // long document is a latency-bound 0.1-0.2 ms and must not queue ahead of the short ones -- and the
    // statistics pass of each piece on the context's stream.  (More streams only alias the same hardware queues.)
    const int nb = (int)h->buckets.size();
    const int P = std::max<int>((int)h->pieces.size(), 1);      // 1: one statistics pass over the corpus' own index
    auto piece_index = [&](int q) -> const tmvb_inv_index& { return h->pieces.empty() ? h->corp->term_index : h->pieces[q]; };
    // One statistics pass (small corpora, one GPU's shard of a sharded run): the critical chain document kernels ->
    // statistics -> M-step stays on the context's stream, so that it pays kernel boundaries (~2 us) instead of
    // cross-stream event hops (~30 us each way, a quarter of a 0.3 ms iteration); only the latency-bound long
    // documents fork to aux[1].  Pipelined pieces need the two chains on two streams.
    // Pipelined pieces (round 3): the document chain ALSO stays on the context's stream; the statistics passes of the pieces
    // before the last run on aux[0] in the chain's shadow (where an event hop costs nothing), and the last pass -- the one
    // nothing hides -- follows the last document kernel on the context's stream, as do the M-step and the next E-step: no
    // cross-stream hop is left on the critical path of an iteration (there were two, ~20 us each, around the last pass).
    // TMVB_LDA_CHAIN_AUX=1 restores the round-2 plan (chain on aux[0], every pass on the context's stream).
    static const bool chain_aux = [] { const char* e = getenv("TMVB_LDA_CHAIN_AUX"); return e && atoi(e) != 0; }();
    hipStream_t chain_st = (h->reg_path && (P == 1 || !chain_aux)) ? ctx->stream : h->aux[0];
    const bool shadow_stats = P > 1 && chain_st == ctx->stream;       // passes 0 .. P-2 on aux[0]
    if (chain_st == ctx->stream) { int jrc = lda_join_side(h); if (jrc) return jrc; }   // update_alpha! of the last iteration
    if (h->elbo_pending) {                                  // (the chain on aux[0]: the document kernels there rewrite what lda_elbo_doc_kernel reads)
        TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_elbo, 0));
        h->elbo_pending = false;
    }
    TMVB_HIP(hipEventRecord(h->ev_fork, ctx->stream));
    for (int a = 0; a < 2; ++a) {
        // every stream that carries document kernels waits for the fork event, i.e. for the previous iteration's M-step on
        // the context's stream -- INCLUDING the chain stream aux[0] of the pipelined plan (a `continue` for "the chain stream"
        // once skipped it: the next iteration's first piece then started under the current M-step, reading beta while
        // beta_norm was still to run; caught in the kernel timeline, now covered by test_train_equals_stepwise_pipelined)
        if (a == 0 && chain_st == ctx->stream) continue;                                 // aux[0] is unused in the one-pass plan
        TMVB_HIP(hipStreamWaitEvent(h->aux[a], h->ev_fork, 0));
        if (h->side_pending) TMVB_HIP(hipStreamWaitEvent(h->aux[a], h->ev_side, 0));   // update_alpha! of the last iteration
    }
    TermStatsParams tp;
    tp.K = h->K; tp.tstride = h->KP; tp.ostride = h->K;
    tp.w = h->d_wtok; tp.E = h->d_E; tp.T = h->d_beta[h->cur]; tp.eps = TMVB_EPS_F; tp.base = 0.0f; tp.keps = (float)h->K * TMVB_EPS_F;
    tp.out = h->d_stats; tp.partial = h->d_ts_partial; tp.estride = h->e_padded ? h->estride : 0;
    // gather-side statistics of the documents whose kernels precede `after` on its stream:
    //   S[:, j] += beta[:, j] .* sum_tokens w E[:, doc] + eps sum w     (update_beta!(model, d))
    auto stats_pass = [&](const tmvb_inv_index& ix, hipStream_t on, const TermStatsParams& tpp) -> int {
        if ((p.debug & 1) || ix.n_chunks <= 0) return TMVB_OK;
        return tmvb_launch_termstats(ctx, h->nslot, h->KP, h->e_padded, ix, tpp, on);
    };
    // Decomposed update_elbo! (lda_elbo_doc_kernel): an iteration that will be checked has its statistics passes leave sum c log2 s per chunk
    const int parts_env = h->parts_env;                          // TMVB_LDA_ELBO_PARTS, read per model
    // lda_elbo_doc_kernel stages 3 x 64 rows of K floats in dynamic LDS (95 KB at K = 124; gfx950 has 160 KB per workgroup).  On a device whose limit is
    // below that (round-5 advice) the iteration does not collect, and update_elbo! takes the token walk instead of failing with TMVB_EHIP.
    const size_t elbo_doc_lds = (size_t)3 * 64 * (size_t)h->K * sizeof(float);
    if (h->lds_limit < 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 64 * 1024; }
        h->lds_limit = v;
    }
    const bool collect = (parts_env == 2 || (parts_env != 0 && h->want_parts)) && tmvb_termstats_recomputes(h->KP, h->e_padded) && !(p.debug & 1) &&
                         viter > 0 &&                               // (viter = 0: the document kernels leave E = 0, not the factor update_elbo! rebuilds phi from)
                         elbo_doc_lds <= (size_t)h->lds_limit;
    std::vector<int64_t> logz_off((size_t)P + 1, 0);
    auto stats_index = [&](int q) -> const tmvb_inv_index& {      // the index piece q's pass walks (tmvb_lda_estep_allreduce: its slice-major rebuild)
        return (h->ar_live && q == P - 1 && h->pieces.empty()) ? h->ar_index : piece_index(q);
    };
    if (collect) {
        for (int q = 0; q < P; ++q) logz_off[(size_t)q + 1] = logz_off[(size_t)q] + stats_index(q).n_chunks;
        const size_t need = (size_t)std::max<int64_t>(logz_off[(size_t)P], 1);
        if (need > h->logz_cap) {                                 // (first checked iteration of a plan; hipFree synchronises)
            (void)hipFree(h->d_logz); h->d_logz = nullptr; h->logz_cap = 0;
            int arc = dmalloc(&h->d_logz, need);
            if (arc) return arc;
            h->logz_cap = need;
        }
    }
    h->logz_valid = false; h->pw_diff = false;
    auto with_logz = [&](TermStatsParams t, int q) { t.logz = collect ? h->d_logz + logz_off[(size_t)q] : nullptr; return t; };
    // Round 5: the passes before the last accumulate in order in d_stats on aux[0]; the LAST pass, on the context's stream behind the last document
    // kernel, used to wait for them (round 4's timeline: last document kernel done at 589 us, pass 2 + its combine at 625 us, one cross-queue hop, last
    // pass 637 - 705 us -- 48 us of a 750 us iteration with the chain waiting).  It now writes a buffer of its own (and its own multi-chunk partials) and
    // starts at once; the join with aux[0] moves behind it, and update_beta!'s column-sum pass adds the two buffers on its way.  One context only (a
    // sharded handle all-reduces d_stats); TMVB_LDA_SPLIT_OUT=0: the round-4 order.
    static const bool split_out_env = [] { const char* e = getenv("TMVB_LDA_SPLIT_OUT"); return !(e && atoi(e) == 0); }();
    // (round-5 advice) ... and only into the library's OWN statistics buffer: a host that bound its tensor (tmvb_lda_bind_stats, dist.py) may read it
    // directly between estep and update_beta!, and must see the whole of S there, not S minus the last piece
    bool split_out = split_out_env && shadow_stats && h->own_stats && !h->distributed && h->comm == nullptr && !h->ar_live && !(p.debug & 1);
    if (split_out && !h->d_stats_b) {
        size_t slots = 1;
        for (const tmvb_inv_index& ix : h->pieces) slots = std::max(slots, (size_t)ix.n_slots);
        int arc = dmalloc(&h->d_stats_b, (size_t)h->K * h->V);
        if (!arc) arc = dmalloc(&h->d_ts_partial_b, slots * (size_t)(h->K + 1));
        if (arc) return arc;
        TMVB_HIP(hipMemsetAsync(h->d_stats_b, 0, (size_t)h->K * h->V * sizeof(float), ctx->stream));
    }
    if (split_out) { int mrc = lda_merge_stats(h); if (mrc) return mrc; }      // (an E-step behind an E-step without update_beta!: keep what is there)
    int piece_open = 0;                                 // pieces [0, piece_open) have their statistics pass issued
    auto close_pieces = [&](int upto) -> int {          // document kernels of pieces < upto are all issued
        for (; piece_open < upto; ++piece_open) {
            if (piece_open == P - 1) {                  // the last piece also holds the long documents (aux[1])
                TMVB_HIP(hipEventRecord(h->ev_chain, chain_st));          // the chain's own document kernels, before the join below
                TMVB_HIP(hipEventRecord(h->ev_join[1], h->aux[1]));
                TMVB_HIP(hipStreamWaitEvent(chain_st, h->ev_join[1], 0));
            }
            hipStream_t pass_st = ctx->stream;
            if (shadow_stats) {
                if (piece_open < P - 1) {                // in the chain's shadow: aux[0] waits for this piece's document kernels
                    pass_st = h->aux[0];
                    TMVB_HIP(hipEventRecord(h->ev_piece[piece_open], chain_st));
                    TMVB_HIP(hipStreamWaitEvent(pass_st, h->ev_piece[piece_open], 0));
                } else if (!split_out) {                 // the passes accumulate in order: the last one waits for aux[0]'s
                    TMVB_HIP(hipEventRecord(h->ev_join[0], h->aux[0]));
                    TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_join[0], 0));
                }
            } else {
                TMVB_HIP(hipEventRecord(h->ev_piece[piece_open], chain_st));
                if (chain_st != ctx->stream) TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_piece[piece_open], 0));
            }
            if (piece_open == P - 1 && h->ar_live) {     // tmvb_lda_estep_allreduce: slice by slice, an event behind each
                const tmvb_inv_index& ix = h->pieces.empty() ? h->ar_index : h->pieces.back();
                for (int k = 0; k < h->ar_slices; ++k) {
                    if (!(p.debug & 1)) { int rc = tmvb_launch_termstats(ctx, h->nslot, h->KP, h->e_padded, ix, with_logz(tp, piece_open), pass_st, h->ar_order[(size_t)k]); if (rc) return rc; }
                    TMVB_HIP(hipEventRecord(h->ev_slice[(size_t)k], pass_st));
                }
                continue;
            }
            if (split_out && piece_open == P - 1) {      // its own buffer, no wait; then the join with the passes before it
                TermStatsParams tpb = with_logz(tp, piece_open);
                tpb.out = h->d_stats_b; tpb.partial = h->d_ts_partial_b;
                int rc = stats_pass(piece_index(piece_open), pass_st, tpb);
                if (rc) return rc;
                h->stats_b_live = true;
                TMVB_HIP(hipEventRecord(h->ev_join[0], h->aux[0]));
                TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_join[0], 0));
                continue;
            }
            int rc = stats_pass(piece_index(piece_open), pass_st, with_logz(tp, piece_open));
            if (rc) return rc;
        }
        return TMVB_OK;
    };
    for (int bi = 0; bi < nb; ++bi) {
        const tmvb_bucket& b = h->buckets[bi];
        const bool chain = b.reg_tiles > 0 && b.waves == 1;   // single-wave register buckets: the piece chain on aux[0]
        hipStream_t st = chain ? chain_st : h->aux[1];
        if (!h->reg_path) st = h->aux[(bi & 1) ^ 1];          // LDS-tile buckets only: alternate the two streams
        if (chain) { int rc = close_pieces(b.piece); if (rc) return rc; }
        if (b.grid_np > 0) {
            const dim3 grid((unsigned)b.count), block(64 * (unsigned)b.waves);
            const int* tol = h->d_grid_topic_of_lane;
            switch (p.LPR) {
#define LDA_GRID_CASE(LPRV) case LPRV: lda_launch_grid<LPRV>(b.grid_np, grid, block, st, p, b.first, tol); break;
                LDA_GRID_CASE(1) LDA_GRID_CASE(3) LDA_GRID_CASE(5) LDA_GRID_CASE(7) LDA_GRID_CASE(9) LDA_GRID_CASE(11) LDA_GRID_CASE(13)
                LDA_GRID_CASE(15) LDA_GRID_CASE(17) LDA_GRID_CASE(19) LDA_GRID_CASE(21) LDA_GRID_CASE(23) LDA_GRID_CASE(25)
#undef LDA_GRID_CASE
                default: TMVB_REQUIRE(false, TMVB_EINVAL, "tmvb_lda_estep: no grid-tile kernel for KP=%d", h->KP);
            }
            TMVB_HIP(hipGetLastError());
            continue;
        }
        if (b.reg_tiles > 0) {
            const dim3 grid((unsigned)b.count), block(64 * (unsigned)b.waves);
            const int* tol = h->d_topic_of_lane;
            switch (p.LPR) {
#define LDA_REG_CASE(LPRV, TMAX) case LPRV: lda_launch_reg<LPRV, TMAX>(b.reg_tiles, grid, block, st, p, b.first, tol); break;
                LDA_REG_CASE(1, 4) LDA_REG_CASE(3, 4) LDA_REG_CASE(5, 4) LDA_REG_CASE(7, 4) LDA_REG_CASE(9, 4) LDA_REG_CASE(11, 4)
                LDA_REG_CASE(13, 4) LDA_REG_CASE(15, 3) LDA_REG_CASE(17, 3) LDA_REG_CASE(19, 2) LDA_REG_CASE(21, 2) LDA_REG_CASE(23, 2)
                LDA_REG_CASE(25, 3)
#undef LDA_REG_CASE
                default: TMVB_REQUIRE(false, TMVB_EINVAL, "tmvb_lda_estep: no register-tile kernel for KP=%d", h->KP);
            }
            TMVB_HIP(hipGetLastError());
            continue;
        }
        const size_t lds = tmvb_tile_bytes(b.tile_rows, h->KP);
        int rc = dispatch_nslot(h->nslot, [&](auto ns) -> int {
            constexpr int NS = decltype(ns)::value;
            const dim3 grid((unsigned)b.count), block(64);
            auto launch = [&](auto kern) -> int {
                if (lds > TMVB_MAX_TILE_BYTES)
                    TMVB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, grid, block, lds, st, p, b.first, b.tile_rows);
                return TMVB_OK;
            };
            if (NS == 1 && p.LPR == 13) return launch(lda_estep_kernel<1, 13>);
            if (NS == 1 && p.LPR == 3) return launch(lda_estep_kernel<1, 3>);
            if (NS == 1 && p.LPR == 5) return launch(lda_estep_kernel<1, 5>);
            if (NS == 2 && p.LPR == 25) return launch(lda_estep_kernel<2, 25>);
            return launch(lda_estep_kernel<NS, 0>);
        });
        if (rc) return rc;
        TMVB_HIP(hipGetLastError());
    }
    hipStream_t side = h->aux[lda_side_index()];
    {
        int rc = close_pieces(P);
        if (rc) return rc;
        // the side chain needs every document kernel: the chain's (ev_chain) and the long documents' (ev_join[1]) -- waited for
        // directly, not through ev_piece[P - 1], which the chain stream records only after ITS join with aux[1]: one cross-stream
        // hop (~20 us) less in front of update_alpha!, which on a small shard is what the next iteration waits for
        if (collect && h->M > 0) {
            // the alpha this E-step read, for lda_elbo_doc_kernel (update_alpha! may overtake that kernel): copied on the side stream, i.e. in front of this
            // iteration's update_alpha! in stream order and in front of the side chain's waits -- and enqueued HERE, behind every document-kernel launch in
            // host order (a checked iteration starts with the host behind the device: whatever it enqueues first delays the first document kernel)
            // (round-5 advice) the previous collecting E-step's lda_elbo_doc_kernel may still be reading d_alpha_e on aux[ELBO] when no host-synchronising
            // update_elbo! came in between (TMVB_LDA_ELBO_PARTS=2, or two E-steps in a row)
            if (h->elbo_pending) TMVB_HIP(hipStreamWaitEvent(side, h->ev_elbo, 0));
            hipLaunchKernelGGL(lda_copy_d_kernel, dim3(1), dim3(256), 0, side, (const double*)h->d_alpha_d, h->d_alpha_e, h->K);
            TMVB_HIP(hipGetLastError());
            TMVB_HIP(hipEventRecord(h->ev_acopy, side));
        }
        TMVB_HIP(hipStreamWaitEvent(side, h->ev_chain, 0));
        TMVB_HIP(hipStreamWaitEvent(side, h->ev_join[1], 0));
    }
    // Elogtheta_sum (update_alpha!'s input, src/LDA.jl:98) under the statistics pass
    {
        // (Round 4, measured and dropped: the row sums update_beta! normalises by are sum_v S[v][k] = sum_d (gamma_dk - alpha_k), so the column
        // sums of gamma -- one more job of this launch pair -- would take colsum(S) (17 + 5.5 us) off the tail of every iteration.  On the whole
        // corpus the tail is not what the next iteration waits for (1297.5 / 1302.4 it/s with, 1299.0 / 1274.0 without); on a 16 100-document
        // shard THIS side chain is -- document kernels -> column sums -> update_alpha! (42 us on one wave) -> next E-step -- and the extra job
        // lengthens it: period 186.8 us against 178.9 (timelines of run r4ag).  Stream priorities for the chain changed nothing either.)
        int rc = tmvb_colsum(ctx, h->nslot, h->K, h->d_elog, h->M, h->d_partial_side, h->d_esum, h->d_stats + (size_t)h->K * h->V, side);
        if (rc) return rc;
        h->n_elbo_blocks = 0;
        if (collect && h->M > 0) {        // update_elbo!'s per-document half, here: it needs the document kernels' output and the alpha they read, nothing else

            const unsigned nblk = (unsigned)((h->M + 63) / 64);
            const size_t lds = elbo_doc_lds;
            if (lds > 48 * 1024 && (int)lds > h->elbo_doc_lds_set) {       // once per handle and size, not on every collecting E-step
                TMVB_HIP(hipFuncSetAttribute((const void*)lda_elbo_doc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                h->elbo_doc_lds_set = (int)lds;
            }
            static const bool own_stream = [] { const char* e = getenv("TMVB_LDA_ELBO_STREAM"); return !(e && atoi(e) == 0); }();   // 0: in a row on the side stream (A/B)
            hipStream_t es = own_stream ? h->aux[tmvb_lda::ELBO] : side;
            // the kernel starts behind the column sums of Elogtheta (ev_acopy re-recorded behind them) rather than with the end of the document kernels: right
            // beside the last statistics pass it slows that pass more than it gains -- time to the plateau 1.262 / 1.247 / 1.239 s against 1.263 / 1.281 / 1.266 s,
            // alternating in one call (TMVB_LDA_ELBO_EARLY=1 selects the early start)
            static const bool early = [] { const char* e = getenv("TMVB_LDA_ELBO_EARLY"); return e && atoi(e) != 0; }();
            if (!early) TMVB_HIP(hipEventRecord(h->ev_acopy, side));
            TMVB_HIP(hipStreamWaitEvent(es, h->ev_chain, 0));     // every document kernel, as the side chain above
            TMVB_HIP(hipStreamWaitEvent(es, h->ev_join[1], 0));
            TMVB_HIP(hipStreamWaitEvent(es, h->ev_acopy, 0));
            hipLaunchKernelGGL(lda_elbo_doc_kernel, dim3(nblk), dim3(256), lds, es, h->K, h->M, (const double*)h->d_alpha_e, h->d_gamma, h->d_elog, h->d_elog_old, h->d_doc_val);
            TMVB_HIP(hipGetLastError());
            TMVB_HIP(hipEventRecord(h->ev_elbo, es));
            h->elbo_pending = true;
            h->n_elbo_blocks = nblk;
        }
        TMVB_HIP(hipEventRecord(h->ev_side, side));
        h->side_pending = true;
        h->esum_fresh = true; h->esum_side = true;
    }
    h->stats_fresh = !(p.debug & 1); h->pw_valid = false;
    h->logz_valid = collect; h->n_logz = collect ? logz_off[(size_t)P] : 0;
    if (h->timing) TMVB_HIP(hipEventRecord(h->ev1, ctx->stream));
    h->timed = true;
    return TMVB_OK;
}

static int colsum(tmvb_lda* h, const float* X, int64_t ncols, double* out_d, float* out_f)
{
    return tmvb_colsum(h->ctx, h->nslot, h->K, X, ncols, h->d_partial, out_d, out_f);
}

extern "C" int tmvb_lda_reduce_docs(tmvb_lda* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_reduce_docs: handle is NULL");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    h->mark_valid = false;
    if (h->esum_fresh) {                                  // enqueued by tmvb_lda_estep (side chain)
        h->esum_fresh = false;
        return h->distributed ? lda_join_side(h) : TMVB_OK;   // a sharded caller all-reduces the tail next, on this stream
    }
    h->esum_side = false;
    { int jrc = lda_join_side(h); if (jrc) return jrc; }
    return colsum(h, h->d_elog, h->M, h->d_esum, h->d_stats + (size_t)h->K * h->V);
}

// ---- the sharded E-step with its collective: see the fields of tmvb_lda.
// The plan.  Every rank needs the same slabs in the same order, so the postings per term of the WHOLE corpus are summed over the
// communicator once (fp64: exact).  Cuts: equal shares of  postings(v) / nnz + 1 / V  -- half a slice's weight is pass time, half is
// bytes on the wire.  Order: the two-machine flow shop (pass, then collective) has Johnson's rule -- slices with less pass than wire
// first, cheapest pass first; then the others, most wire first -- so a vocabulary sorted by frequency (few heavy ids in front, a long
// tail of light ones) sends its long tail first and hides it under the heavy ids' pass, and a shuffled one gets S alike slices.
static int lda_ar_prepare(tmvb_lda* h)
{
    if (h->ar_comm == h->comm) return TMVB_OK;
    tmvb_ctx* ctx = h->ctx;
// (default ONE slab: on the 16 100-document shard of an 8-GPU run every further slice costs ~20 us of launches and kernel tails -- fused
    // iteration 0.166 / 0.184 / 0.205 / 0.224 ms with 1 / 2 / 3 / 4 slabs at nranks = 1, tools/ar_slices_probe.py -- against a last pass of
    // ~50 us to hide wire under; what pays at that size is the early collective of the tail, below, which every setting has)
    const int want = [] { const char* e = getenv("TMVB_AR_SLICES"); return e ? atoi(e) : 1; }();      // read per plan (per communicator)
    const int64_t V = h->V;
    int S = (int)std::min<int64_t>(std::max(want, 1), std::max<int64_t>(V, 1));
    if (!(tmvb_termstats_recomputes(h->KP, h->e_padded) && h->reg_path)) S = 1;      // the stored-weight passes keep one collective
    std::vector<double> cnt((size_t)V, 0.0);
    for (int32_t t : h->corp->h_terms) cnt[(size_t)t] += 1.0;
    {   // the collective every rank takes part in, sliced or not (a rank must not decide alone)
        double* d = nullptr;
        int rc = dmalloc(&d, (size_t)V);
        if (rc) return rc;
        hipError_t e = hipMemcpyAsync(d, cnt.data(), (size_t)V * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) rc = tmvb_comm_allreduce(h->comm, d, V, TMVB_F64);
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(cnt.data(), d, (size_t)V * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d);
        if (rc) return rc;
        TMVB_HIP(e);
    }
    {   // cuts and order from the GLOBAL counts: a host function of its own (tmvb_allreduce_plan, tmvb_comm.hip), tested on the CPU
        std::vector<int64_t> cuts((size_t)S + 1);
        std::vector<int32_t> order((size_t)S);
        int32_t s_out = 1;
        int prc = tmvb_allreduce_plan(cnt.data(), V, S, cuts.data(), order.data(), &s_out);
        if (prc) return prc;
        S = s_out;
        h->ar_cuts.assign(cuts.begin(), cuts.begin() + S + 1);
        h->ar_order.assign(order.begin(), order.begin() + S);
    }
    if (S > 1) {
        // the last piece's index again, slice-major
        tmvb_inv_index fresh;
        const bool whole = h->pieces.empty();
        int rc = tmvb_build_inv_index(ctx, h->M, V, h->corp->h_doc_ptr.data(), h->corp->h_terms.data(), h->corp->h_counts.data(), &fresh,
                                      whole ? nullptr : h->doc_piece.data(), whole ? 0 : (int)h->pieces.size() - 1, &h->ar_cuts);
        if (rc) { tmvb_free_inv_index(&fresh); return rc; }
        tmvb_inv_index& slot = whole ? h->ar_index : h->pieces.back();
        const int64_t old_slots = whole ? h->corp->term_index.n_slots : slot.n_slots;
        if (slot.built) { TMVB_HIP(hipStreamSynchronize(ctx->stream)); tmvb_free_inv_index(&slot); }
        slot = fresh;
        TMVB_REQUIRE(slot.n_slots <= old_slots, TMVB_EINVAL, "tmvb_lda_estep_allreduce: the sliced index needs %lld partial-sum slots, %lld are allocated",
                     (long long)slot.n_slots, (long long)old_slots);
        while ((int)h->ev_slice.size() < S) {
            hipEvent_t ev = nullptr;
            TMVB_HIP(hipEventCreateWithFlags(&ev, h->events_system_scope ? (unsigned)hipEventDisableTiming : tmvb_event_flags()));
            h->ev_slice.push_back(ev);
        }
    }
    h->ar_slices = S;
    h->ar_comm = h->comm;
    return TMVB_OK;
}

extern "C" int tmvb_lda_estep_allreduce(tmvb_lda* h, int32_t viter, double vtol)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_estep_allreduce: handle is NULL");
    TMVB_REQUIRE(h->comm != nullptr && h->distributed, TMVB_EINVAL, "tmvb_lda_estep_allreduce: the handle has no communicator (tmvb_lda_set_comm)");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    int rc = lda_ar_prepare(h);
    if (rc) return rc;
    const int S = h->ar_slices;
    h->ar_live = S > 1;
    rc = tmvb_lda_estep(h, viter, vtol);
    h->ar_live = false;
    if (rc) return rc;
    const bool tail_on_side = h->esum_fresh && h->esum_side;             // the E-step's side chain made the tail (ev_side marks it)
    if ((rc = tmvb_lda_reduce_docs(h))) return rc;                       // ... or this call does, on the context's stream
    // The collectives, in the order every rank issues them:
    //   1. the Elogtheta_sum tail (K floats) on aux[AR], as soon as the side chain has it -- the statistics pass is still running; update_alpha!
    //      then starts from THAT event on the side stream instead of waiting for the whole buffer (a one-wave fp64 Newton kernel, ~45 us, that the
    //      next E-step waits for: in the three-call form it only starts after the last collective);
    //   2. slabs 0 .. S-2 on aux[AR], each behind its slice's event, under the following slices' passes;
    //   3. the last slab on the context's stream itself, behind aux[AR]'s collectives: no cross-stream hop (~25 us each way) is left on the
    //      critical path -- with the last slab on aux[AR] as well, four slabs cost 0.107 ms more than they could hide (tools/ar_slices_probe.py).
    const int64_t K = h->K, V = h->V;
    tmvb_ctx* ctx = h->ctx;
    hipStream_t cs = h->aux[tmvb_lda::AR];
    // the two events recorded BEHIND collectives keep the default system-scope fence: what they order was written with the help of other
    // devices (the ordering events of the single-device plans do without it, tmvb_event_flags)
    if (!h->ev_comm) TMVB_HIP(hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming));
    if (!h->ev_tail) TMVB_HIP(hipEventCreateWithFlags(&h->ev_tail, hipEventDisableTiming));
    if (tail_on_side) {
        TMVB_HIP(hipStreamWaitEvent(cs, h->ev_side, 0));                  // NOT the context's stream: the statistics pass is queued on it
    } else {
        TMVB_HIP(hipEventRecord(h->ev_tail, ctx->stream));
        TMVB_HIP(hipStreamWaitEvent(cs, h->ev_tail, 0));
    }
    auto fail = [&](int code) { (void)hipStreamSynchronize(cs); (void)hipStreamSynchronize(ctx->stream); return code; };
    if ((rc = tmvb_comm_allreduce_on(h->comm, h->d_stats + K * V, K, TMVB_F32, cs))) return fail(rc);
    TMVB_HIP(hipEventRecord(h->ev_tail, cs));
    h->tail_early = true;
    for (int k = 0; k + 1 < S; ++k) {
        const int sl = h->ar_order[(size_t)k];
        TMVB_HIP(hipStreamWaitEvent(cs, h->ev_slice[(size_t)k], 0));
        const int64_t count = (h->ar_cuts[(size_t)sl + 1] - h->ar_cuts[(size_t)sl]) * K;
        if ((rc = tmvb_comm_allreduce_on(h->comm, h->d_stats + h->ar_cuts[(size_t)sl] * K, count, TMVB_F32, cs))) return fail(rc);
    }
    TMVB_HIP(hipEventRecord(h->ev_comm, cs));
    TMVB_HIP(hipStreamWaitEvent(ctx->stream, h->ev_comm, 0));             // collectives of one communicator run in issue order anyway
    if (S <= 1) return tmvb_comm_allreduce(h->comm, h->d_stats, K * V, TMVB_F32);
    const int last = h->ar_order[(size_t)S - 1];
    return tmvb_comm_allreduce(h->comm, h->d_stats + h->ar_cuts[(size_t)last] * K, (h->ar_cuts[(size_t)last + 1] - h->ar_cuts[(size_t)last]) * K, TMVB_F32);
}

extern "C" int tmvb_lda_stats(tmvb_lda* h, void** dev_ptr, int64_t* n_f32)
{
    TMVB_REQUIRE(h && dev_ptr && n_f32, TMVB_EINVAL, "tmvb_lda_stats: NULL argument");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    { int jrc = lda_join_side(h); if (jrc) return jrc; }
    { int mrc = lda_merge_stats(h); if (mrc) return mrc; }
    *dev_ptr = h->d_stats;
    *n_f32 = (int64_t)h->K * h->V + h->K;
    return TMVB_OK;
}

extern "C" int tmvb_lda_bind_stats(tmvb_lda* h, void* dev_ptr, int64_t n_f32)
{
    TMVB_REQUIRE(h && dev_ptr, TMVB_EINVAL, "tmvb_lda_bind_stats: NULL argument");
    const int64_t need = (int64_t)h->K * h->V + h->K;
    TMVB_REQUIRE(n_f32 >= need, TMVB_ESHAPE, "tmvb_lda_bind_stats: buffer holds %lld floats, need %lld", (long long)n_f32, (long long)need);
    h->mark_valid = false;
    TMVB_HIP(hipSetDevice(h->ctx->device));
    { int jrc = lda_join_side(h); if (jrc) return jrc; }
    { int mrc = lda_merge_stats(h); if (mrc) return mrc; }
    TMVB_HIP(hipMemcpyAsync(dev_ptr, h->d_stats, (size_t)need * sizeof(float), hipMemcpyDeviceToDevice, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    if (h->own_stats) (void)hipFree(h->d_stats);
    h->d_stats = (float*)dev_ptr;
    h->own_stats = false;
    return TMVB_OK;
}

extern "C" int tmvb_lda_set_distributed(tmvb_lda* h, int64_t M_total, int32_t distributed)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_set_distributed: handle is NULL");
    TMVB_REQUIRE(M_total >= h->M, TMVB_ESHAPE, "tmvb_lda_set_distributed: M_total < local M");
    h->mark_valid = false;
    TMVB_HIP(hipSetDevice(h->ctx->device));
    { int jrc = lda_join_side(h); if (jrc) return jrc; }
    h->M_total = M_total;
    h->distributed = distributed != 0;
    return TMVB_OK;
}

extern "C" int tmvb_lda_update_beta(tmvb_lda* h)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_update_beta: handle is NULL");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    if (h->distributed || !h->esum_side) {                  // update_alpha! may start from here (it shares nothing with update_beta!)
        TMVB_HIP(hipEventRecord(h->ev_mark, ctx->stream));
        h->mark_valid = true;
    }
    int rc;
    if (h->stats_b_live) {                                  // the last statistics pass's own buffer, folded in by the column-sum pass itself
        const int nbc = tmvb_colsum_blocks(h->V);
        rc = dispatch_nslot(h->nslot, [&](auto ns) -> int {
            constexpr int NS = decltype(ns)::value;
            hipLaunchKernelGGL((colsum_merge_partial_kernel<NS>), dim3(nbc), dim3(256), 0, ctx->stream, h->d_stats, h->d_stats_b, h->V, h->K, h->d_partial);
            return TMVB_OK;
        });
        if (rc) return rc;
        TMVB_HIP(hipGetLastError());
        hipLaunchKernelGGL(colsum_final_kernel, dim3((h->K + 3) / 4), dim3(256), 0, ctx->stream, h->d_partial, nbc, h->K, h->d_rowsum, (float*)nullptr);
        TMVB_HIP(hipGetLastError());
        h->stats_b_live = false;
    } else {
        rc = colsum(h, h->d_stats, h->V, h->d_rowsum, nullptr);
        if (rc) return rc;
    }
    const int64_t total = (int64_t)h->KP * h->V;
    int nb = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (total + 255) / 256));
    const bool parts = h->logz_valid && h->stats_fresh;     // the coming update_elbo! is the decomposed one: leave sum S (log beta_new - log beta_old)
    hipLaunchKernelGGL(beta_norm_kernel, dim3(nb), dim3(256), (size_t)h->K * sizeof(double), ctx->stream,
                       h->d_stats, h->d_rowsum, h->d_beta[h->cur ^ 1], h->K, h->KP, h->V, h->d_pw_partial, TMVB_EPS_F,
                       parts ? (const float*)h->d_beta[h->cur] : (const float*)nullptr, -1.0f,
                       parts ? (const double*)h->d_logz : (const double*)nullptr, h->n_logz, h->d_lz_partial);
    TMVB_HIP(hipGetLastError());
    h->pw_blocks = nb;
    h->pw_diff = parts;
    h->pw_valid = h->stats_fresh;
    h->stats_fresh = false;
    h->cur ^= 1;   // beta_old <- beta, beta <- new   (src/LDA.jl:122-123)
    return TMVB_OK;
}

extern "C" int tmvb_lda_update_alpha(tmvb_lda* h, int32_t niter, double ntol)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_update_alpha: handle is NULL");
    TMVB_REQUIRE(niter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative.");
    TMVB_REQUIRE(ntol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const float* ef = h->distributed ? h->d_stats + (size_t)h->K * h->V : nullptr;
    hipStream_t side = h->aux[lda_side_index()];
    // One context: the Newton step needs nothing but Elogtheta_sum, which the E-step left on this very
    // stream, so it starts under the statistics pass.  Document-sharded: it needs the all-reduced sums, i.e.
    // the context's stream as of the preceding update_beta call (or as of now).
    if (h->tail_early) {                                    // tmvb_lda_estep_allreduce: the tail's own collective, long done
        TMVB_HIP(hipStreamWaitEvent(side, h->ev_tail, 0));
        h->tail_early = false;
    } else if (h->distributed || !h->esum_side) {
        if (!h->mark_valid) TMVB_HIP(hipEventRecord(h->ev_mark, ctx->stream));
        TMVB_HIP(hipStreamWaitEvent(side, h->ev_mark, 0));
    }
    h->mark_valid = false;
    int rc = dispatch_nslot(h->nslot, [&](auto ns) -> int {
        constexpr int NS = decltype(ns)::value;
        hipLaunchKernelGGL((lda_alpha_kernel<NS>), dim3(1), dim3(64), 0, side, h->K, (double)h->M_total, h->d_esum, ef,
                           h->d_alpha_d, h->d_alpha_f, niter, ntol, h->d_iters);
        return TMVB_OK;
    });
    if (rc) return rc;
    TMVB_HIP(hipGetLastError());
    TMVB_HIP(hipEventRecord(h->ev_side, side));
    h->side_pending = true;
    return TMVB_OK;
}

// update_elbo! enqueued on the context's stream: the value (this shard's share, for a sharded handle) is left in h->d_elbo, no host
// synchronisation -- the sharded train loop all-reduces it where it lies (tmvb_train.h)
static int lda_elbo_enqueue(tmvb_lda* h)
{
    h->mark_valid = false;
    tmvb_ctx* ctx = h->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    { int jrc = lda_join_side(h); if (jrc) return jrc; }
    bool use_pw = false;
    int64_t n_vals = 0;
    h->elbo_form = 0;
    // (no condition on this handle's own documents: every rank of a sharded run must take the same form -- train!'s evaluations are collectives -- and an
    //  empty shard simply has nothing per document)
    const bool parts = h->logz_valid && h->pw_valid && h->pw_diff && !h->force_walk;
    if (parts) {
        // the decomposed form: the per-document values were enqueued by the E-step itself (side stream: joined above), everything per token was left
        // behind by its statistics passes and by update_beta!
        use_pw = true; n_vals = h->n_elbo_blocks;
        h->elbo_form = 1;
    } else if (h->M > 0 && h->reg_path && getenv("TMVB_LDA_ELBO_LEGACY") == nullptr) {
        const dim3 grid((unsigned)h->M), block(64);
        use_pw = h->pw_valid && !h->pw_diff && getenv("TMVB_LDA_ELBO_NO_PW") == nullptr;
        switch (h->KP / 4) {
#define LDA_ELBO_CASE(LPRV) case LPRV: \
            if (use_pw) hipLaunchKernelGGL((lda_elbo_reg_kernel<LPRV, false>), grid, block, 0, ctx->stream, h->K, h->d_doc_order, h->corp->d_doc_ptr, \
                               h->corp->d_terms, h->corp->d_counts, h->d_alpha_d, h->d_beta[h->cur], h->d_beta[h->cur ^ 1], \
                               h->d_gamma, h->d_elog, h->d_elog_old, h->d_doc_val); \
            else hipLaunchKernelGGL((lda_elbo_reg_kernel<LPRV, true>), grid, block, 0, ctx->stream, h->K, h->d_doc_order, h->corp->d_doc_ptr, \
                               h->corp->d_terms, h->corp->d_counts, h->d_alpha_d, h->d_beta[h->cur], h->d_beta[h->cur ^ 1], \
                               h->d_gamma, h->d_elog, h->d_elog_old, h->d_doc_val); \
            break;
            LDA_ELBO_CASE(1) LDA_ELBO_CASE(3) LDA_ELBO_CASE(5) LDA_ELBO_CASE(7) LDA_ELBO_CASE(9) LDA_ELBO_CASE(11) LDA_ELBO_CASE(13)
            LDA_ELBO_CASE(15) LDA_ELBO_CASE(17) LDA_ELBO_CASE(19) LDA_ELBO_CASE(21) LDA_ELBO_CASE(23) LDA_ELBO_CASE(25)
#undef LDA_ELBO_CASE
            default: TMVB_REQUIRE(false, TMVB_EINVAL, "tmvb_lda_update_elbo: no token-parallel kernel for KP=%d", h->KP);
        }
        TMVB_HIP(hipGetLastError());
    } else if (h->M > 0) {
        int rc = dispatch_nslot(h->nslot, [&](auto ns) -> int {
            constexpr int NS = decltype(ns)::value;
            hipLaunchKernelGGL((lda_elbo_kernel<NS>), dim3((unsigned)h->M), dim3(64), 0, ctx->stream, h->K, h->KP, h->corp->d_doc_ptr,
                               h->corp->d_terms, h->corp->d_counts, h->d_alpha_d, h->d_beta[h->cur], h->d_beta[h->cur ^ 1],
                               h->d_gamma, h->d_elog, h->d_elog_old, h->d_doc_val);
            return TMVB_OK;
        });
        if (rc) return rc;
        TMVB_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(lda_elbo_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, h->d_doc_val, h->M, h->K, h->d_alpha_d, h->d_elbo,
                       use_pw ? h->d_pw_partial : (const double*)nullptr, h->pw_blocks,
                       h->distributed && h->M_total > 0 ? (double)h->M / (double)h->M_total : 1.0, n_vals,
                       parts ? h->d_esum : (const double*)nullptr, parts ? h->d_lz_partial : (const double*)nullptr);
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

extern "C" int tmvb_lda_update_elbo(tmvb_lda* h, double* elbo)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_update_elbo: handle is NULL");
    int rc = lda_elbo_enqueue(h);
    if (rc) return rc;
    tmvb_ctx* ctx = h->ctx;
    double v = 0.0;
    TMVB_HIP(hipMemcpyAsync(&v, h->d_elbo, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    h->elbo = v;
    if (elbo) *elbo = v;
    return TMVB_OK;
}

extern "C" int tmvb_lda_set_comm(tmvb_lda* h, tmvb_comm* comm, int64_t M_total)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_set_comm: handle is NULL");
    int rc = tmvb_lda_set_distributed(h, comm ? M_total : h->M, comm != nullptr);
    if (rc) return rc;
    h->comm = comm;
    h->ar_comm = nullptr;                               // the plan is agreed anew on every attach (a new communicator may live at an old address)
    if (comm != nullptr && !h->events_system_scope) {
        // Round-4 advice: the events that order this device's kernels BEFORE a collective (ev_side / ev_docs / ev_chain / ev_join / ev_piece on the way to
        // the statistics buffer, ev_slice behind a slab's pass) were created with tmvb_event_flags(), i.e. without the system-scope release a default
        // event performs -- right between the streams of ONE device, unvalidated when a peer reads the send buffer directly over xGMI.  A handle with
        // a communicator therefore gets them recreated as plain events (system-scope fence), once; single-device handles keep the cheap ones.
        TMVB_HIP(hipSetDevice(h->ctx->device));
        TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
        for (int a = 0; a < tmvb_lda::NAUX; ++a) if (h->aux[a]) TMVB_HIP(hipStreamSynchronize(h->aux[a]));
        auto remake = [](hipEvent_t& ev) -> hipError_t {
            if (!ev) return hipSuccess;
            (void)hipEventDestroy(ev); ev = nullptr;
            return hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        };
        TMVB_HIP(remake(h->ev_fork)); TMVB_HIP(remake(h->ev_docs)); TMVB_HIP(remake(h->ev_side)); TMVB_HIP(remake(h->ev_chain));
        for (int a = 0; a < tmvb_lda::NAUX; ++a) TMVB_HIP(remake(h->ev_join[a]));
        for (hipEvent_t& ev : h->ev_piece) TMVB_HIP(remake(ev));
        for (hipEvent_t& ev : h->ev_slice) TMVB_HIP(remake(ev));
        h->side_pending = false; h->mark_valid = false; h->elbo_pending = false;
        h->events_system_scope = true;
    }
    return TMVB_OK;
}

namespace {
struct LdaTrainOps {
    int niter, viter; double ntol, vtol;
    int estep(tmvb_lda* h) { return tmvb_lda_estep(h, viter, vtol); }                 // src/LDA.jl:170-180
    int reduce(tmvb_lda* h) { return tmvb_lda_reduce_docs(h); }                       // :98
    int estep_allreduce(tmvb_lda* h) { return tmvb_lda_estep_allreduce(h, viter, vtol); }   // one process per GPU: the three steps with the collective sliced
    int before_allreduce(tmvb_lda* h) { TMVB_HIP(hipSetDevice(h->ctx->device)); int rc = lda_join_side(h); return rc ? rc : lda_merge_stats(h); }
    float* stats(tmvb_lda* h) { return h->d_stats; }
    int64_t stats_len(tmvb_lda* h) { return (int64_t)h->K * h->V + h->K; }
    int mstep(tmvb_lda* h) { int rc = tmvb_lda_update_beta(h); return rc ? rc : tmvb_lda_update_alpha(h, niter, ntol); }   // :181-182
    int elbo_form(tmvb_lda* h) { return h->elbo_form; }
    void force_walk(tmvb_lda* h, bool on, bool doubled = true) { h->force_walk = on; if (!on && doubled) h->elbo_form = 1; }   // (switched off behind the one evaluation that doubled a decomposed one)
    void will_check(tmvb_lda* h, bool checked) { h->want_parts = checked; }           // the coming iteration ends in check_elbo!: collect update_elbo!'s parts on the way
    int elbo_local(tmvb_lda* h, double* s, double* once) { *once = 0.0; return tmvb_lda_update_elbo(h, s); }
    int elbo_enqueue(tmvb_lda* h, double* once) { *once = 0.0; TMVB_HIP(hipSetDevice(h->ctx->device)); return lda_elbo_enqueue(h); }   // -> elbo_dev(h), no sync
    double* elbo_dev(tmvb_lda* h) { return h->d_elbo; }
    tmvb_comm* comm(tmvb_lda* h) { return h->comm; }
    bool distributed(tmvb_lda* h) { return h->distributed; }
    tmvb_ctx* ctx(tmvb_lda* h) { return h->ctx; }
    int64_t nnz(tmvb_lda* h) { return h->corp->info.nnz; }
    void set_elbo(tmvb_lda* h, double v) { h->elbo = v; }
    double get_elbo(tmvb_lda* h) { return h->elbo; }
    int finish(tmvb_lda* h)
    {
        TMVB_HIP(hipSetDevice(h->ctx->device));
        int rc = lda_join_side(h);
        if (rc) return rc;
        TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
        return TMVB_OK;
    }
};
}  // namespace

extern "C" int tmvb_lda_train_group(tmvb_lda* const* hs, int32_t n, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter,
                                    double vtol, int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline)
{
    // src/gpuLDA.jl:349-351
    TMVB_REQUIRE(tol >= 0 && ntol >= 0 && vtol >= 0, TMVB_EINVAL, "tolerance parameters must be nonnegative.");
    TMVB_REQUIRE(iter >= 0 && niter >= 0 && viter >= 0, TMVB_EINVAL, "iteration parameters must be nonnegative.");
    LdaTrainOps ops{niter, viter, ntol, vtol};
    return tmvb_train_group_loop("tmvb_lda_train", hs, n, iter, tol, checkelbo, elbo_traj, iters_done, elbo_baseline, ops);
}

extern "C" int tmvb_lda_train(tmvb_lda* h, int32_t iter, double tol, int32_t niter, double ntol, int32_t viter,
                              double vtol, int32_t checkelbo, double* elbo_traj, int32_t* iters_done, double* elbo_baseline)
{
    TMVB_REQUIRE(h != nullptr, TMVB_EINVAL, "tmvb_lda_train: handle is NULL");
    return tmvb_lda_train_group(&h, 1, iter, tol, niter, ntol, viter, vtol, checkelbo, elbo_traj, iters_done, elbo_baseline);
}

extern "C" int tmvb_lda_elbo_form(tmvb_lda* h, int32_t* form)
{
    TMVB_REQUIRE(h && form, TMVB_EINVAL, "tmvb_lda_elbo_form: NULL argument");
    *form = h->elbo_form;
    return TMVB_OK;
}

// per-document sweep counts of the last E-step (document order of the corpus), for parity tests that compare the
// state of exactly those documents whose exit sweep agrees with the oracle's
extern "C" int tmvb_lda_doc_sweeps(tmvb_lda* h, uint8_t* out)
{
    TMVB_REQUIRE(h && (out || h->M == 0), TMVB_EINVAL, "tmvb_lda_doc_sweeps: NULL argument");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    if (h->M) TMVB_HIP(hipMemcpyAsync(out, h->d_sweeps, (size_t)h->M, hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_lda_sweep_hist(tmvb_lda* h, int64_t* hist, int32_t nbins)
{
    TMVB_REQUIRE(h && hist && nbins > 0, TMVB_EINVAL, "tmvb_lda_sweep_hist: bad argument");
    std::vector<uint8_t> sw((size_t)h->M);
    TMVB_HIP(hipSetDevice(h->ctx->device));
    if (h->M) TMVB_HIP(hipMemcpyAsync(sw.data(), h->d_sweeps, (size_t)h->M, hipMemcpyDeviceToHost, h->ctx->stream));
    TMVB_HIP(hipStreamSynchronize(h->ctx->stream));
    for (int b = 0; b < nbins; ++b) hist[b] = 0;
    for (uint8_t s : sw) hist[std::min<int>(s, nbins - 1)]++;
    return TMVB_OK;
}

extern "C" int tmvb_lda_estep_launches(tmvb_lda* h, int32_t* n)
{
    TMVB_REQUIRE(h && n, TMVB_EINVAL, "tmvb_lda_estep_launches: NULL argument");
    *n = (int32_t)h->buckets.size();
    return TMVB_OK;
}

extern "C" int tmvb_lda_last_estep_ms(tmvb_lda* h, float* ms)
{
    TMVB_REQUIRE(h && ms, TMVB_EINVAL, "tmvb_lda_last_estep_ms: NULL argument");
    TMVB_REQUIRE(h->timing, TMVB_EINVAL, "tmvb_lda_last_estep_ms: E-step timing is off (set TMVB_ESTEP_TIMING=1 before creating the model)");
    TMVB_REQUIRE(h->timed, TMVB_EINVAL, "tmvb_lda_last_estep_ms: no E-step has run");
    TMVB_HIP(hipSetDevice(h->ctx->device));
    TMVB_HIP(hipEventSynchronize(h->ev1));
    TMVB_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return TMVB_OK;
}
