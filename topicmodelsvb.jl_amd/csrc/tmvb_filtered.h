// tmvb_filtered.h -- what the two filtered models share (fLDA src/fLDA.jl, fCTM src/fCTM.jl): the log table
// L = log(beta + eps), the statistics pass that rebuilds the last sweep's phi from (tau_old, per-document vector, lse), the
// kappa / eta updates.  In fLDA the per-document vector is Elogtheta_old, in fCTM lambda_old: phi[i,n] =
// softmax_i(tau_n L[i,t_n] + x_i) has the same form in both (src/fLDA.jl:190, src/fCTM.jl:218).
#pragma once
#include "tmvb_common_kernels.h"

// ------------------------------------------------------------------------------ statistics
// update_beta!(model, d) (src/fLDA.jl:161) and update_kappa!(model, d) (:147) as ONE gather over the term-major
// inverted index: for every token n of term j
//     S[i, j] += tau_n c_n phi[i, n],   phi[i, n] = exp(tau_old_n L[j][i] + Elogtheta_old[i, doc_n] - lse_n)
//     kstat[j] += (1 - tau_n) c_n
// One wave per chunk of <= TMVB_CHUNK tokens of one term (lane = topic), partial slots for multi-chunk terms.
struct FldaStatsParams {
    int K, KP;
    const int32_t* tok_doc; const int32_t* tok_pos; const float* tok_val;
    const int32_t* chunk_id; const int32_t* chunk_begin; const int32_t* chunk_end; const int32_t* chunk_out;
    int n_chunks;
    const float* L; const float* elog_old; const float* tau; const float* tau_old; const float* lse;
    float* S;          // [V][K]
    float* kstat;      // [V]
    float* partial;    // [n_slots][K + 1]
};

template <int NS>
static __global__ __launch_bounds__(256) void flda_stats_kernel(FldaStatsParams p)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= p.n_chunks) return;
    const int K = p.K;
    const int j = p.chunk_id[c];
    const int b = p.chunk_begin[c], e = p.chunk_end[c];
    float Lj[NS], acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = lane + 64 * s;
        Lj[s] = (i < K) ? p.L[(int64_t)j * p.KP + i] : 0.0f;
        acc[s] = 0.0f;
    }
    float kl = 0.0f;
    for (int t0 = b; t0 < e; t0 += 64) {
        const int tok = t0 + lane;
        const bool valid = tok < e;
        const int dd = valid ? p.tok_doc[tok] : 0;
        const int pos = valid ? p.tok_pos[tok] : 0;
        const float cn = valid ? p.tok_val[tok] : 0.0f;
        const float tn = valid ? p.tau[pos] : 0.0f, tp = valid ? p.tau_old[pos] : 0.0f, ls = valid ? p.lse[pos] : INFINITY;
        const float vn = tn * cn;
        kl += cn - vn;                                                  // (1 - tau_n) c_n
        const int cnt = min(64, e - t0);
        // four postings per trip, their row loads unconditional (a lane past K re-reads element 0, a posting past the end is lane
        // data of an invalid token: document 0, weight 0, lse = inf -> contributes exactly 0): straight-line code, four gathers
        // in flight per wave instead of one
        constexpr int U = 4;
        for (int k0 = 0; k0 < cnt; k0 += U) {
            float ev[U][NS], vk[U], tk[U], lk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = min(k0 + u, 63);
                const int dk = __builtin_amdgcn_readlane(dd, k);
                vk[u] = (k0 + u < 64) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vn), k)) : 0.0f;
                tk[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tp), k));
                lk[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls), k));
                const float* erow = p.elog_old + (int64_t)dk * K;
#pragma unroll
                for (int s = 0; s < NS; ++s) { const int i = lane + 64 * s; ev[u][s] = erow[i < K ? i : 0]; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k0 + u >= cnt) break;                                   // uniform
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int i = lane + 64 * s;
                    if (i < K) acc[s] = fmaf(vk[u], __expf(fmaf(tk[u], Lj[s], ev[u][s]) - lk[u]), acc[s]);
                }
            }
        }
    }
    const float ksum = wave_sum(kl);
    const int slot = p.chunk_out[c];
    if (slot < 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = lane + 64 * s;
            if (i < K) p.S[(int64_t)j * K + i] += acc[s];
        }
        if (lane == 0) p.kstat[j] += ksum;
    } else {
        float* pr = p.partial + (int64_t)slot * (K + 1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = lane + 64 * s;
            if (i < K) pr[i] = acc[s];
        }
        if (lane == 0) pr[K] = ksum;
    }
}

// terms split over several chunks: partial slots summed in a fixed order
template <int NS>
static __global__ __launch_bounds__(64) void flda_stats_multi_kernel(FldaStatsParams p, const int32_t* __restrict__ multi_id,
                                                              const int32_t* __restrict__ multi_first,
                                                              const int32_t* __restrict__ multi_count, int n_multi)
{
    const int lane = threadIdx.x;
    const int m = blockIdx.x;
    if (m >= n_multi) return;
    const int K = p.K;
    const int j = multi_id[m], first = multi_first[m], cnt = multi_count[m];
    float acc[NS], ks = 0.0f;
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0f;
    // four partial rows in flight per trip (fixed order of summation; a row past the end is the last row again, not added)
    for (int c0 = 0; c0 < cnt; c0 += 4) {
        float v[4][NS], kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* pr = p.partial + (int64_t)(first + min(c0 + u, cnt - 1)) * (K + 1);
#pragma unroll
            for (int s = 0; s < NS; ++s) { const int i = lane + 64 * s; v[u][s] = pr[i < K ? i : K]; }
            kv[u] = pr[K];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + u >= cnt) break;
#pragma unroll
            for (int s = 0; s < NS; ++s) { const int i = lane + 64 * s; if (i < K) acc[s] += v[u][s]; }
            ks += kv[u];
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = lane + 64 * s;
        if (i < K) p.S[(int64_t)j * K + i] += acc[s];
    }
    if (lane == 0) p.kstat[j] += ks;
}

// ------------------------------------------------------------------------------ M-step
// L = log(beta + eps) in the padded layout (pads 0)
static __global__ __launch_bounds__(256) void flda_logbeta_kernel(const float* __restrict__ beta, float* __restrict__ L, int K, int KP, int64_t V)
{
    const int64_t total = V * KP;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(q % KP);
#ifdef TMVB_MUTANT_FLDA_NO_EPS
        // MUTANT (tests/test_mutants_gpu.py, never in a shipped build): the log table of the filtered models without the epsilon of @boink
        // (src/fLDA.jl:191, :184; src/fCTM.jl:219): a term whose beta column is zero then has phi = softmax(-inf - -inf) = NaN
        L[q] = (i < K) ? logf(beta[q]) : 0.0f;
#else
        L[q] = (i < K) ? logf(beta[q] + TMVB_EPS_F) : 0.0f;
#endif
    }
}

// update_kappa!(model) (src/fLDA.jl:138-142) and update_eta! (:122-124) in one workgroup:
//   kappa_old <- kappa; kappa <- kstat / sum(kstat); kstat <- 0;
//   eta = sum_d dot(tau_d, counts_d) / sum(C) = 1 - sum(kstat) / C_total     (sum_n (1 - tau_n) c_n = sum(kstat))
// do_eta = 0 leaves eta alone (the reference's order runs update_alpha! between the two, which touches neither).
static __global__ __launch_bounds__(1024) void flda_kappa_eta_kernel(float* __restrict__ kstat, float* __restrict__ kappa, float* __restrict__ kappa_old,
                                                              int64_t V, double C_total, double* __restrict__ eta, int do_kappa, int do_eta,
                                                              double* __restrict__ ksum_keep)
{
    __shared__ double red[1024];
    if (do_kappa) {
        double s = 0.0;
        for (int64_t j = threadIdx.x; j < V; j += 1024) s += (double)kstat[j];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        const double tot = red[0];
        for (int64_t j = threadIdx.x; j < V; j += 1024) {
            kappa_old[j] = kappa[j];
            kappa[j] = (float)((double)kstat[j] / tot);
            kstat[j] = 0.0f;
        }
        if (threadIdx.x == 0) *ksum_keep = tot;
    }
    if (do_eta && threadIdx.x == 0) eta[0] = 1.0 - *ksum_keep / C_total;
}


// launches the statistics pass over a term-major inverted index (NS = topic slots per lane)
static int tmvb_launch_filtered_stats(tmvb_ctx* ctx, int nslot, const tmvb_inv_index& ix, FldaStatsParams sp)
{
    if (ix.n_chunks <= 0) return TMVB_OK;
    sp.tok_doc = ix.d_doc; sp.tok_pos = ix.d_pos; sp.tok_val = ix.d_val;
    sp.chunk_id = ix.d_chunk_id; sp.chunk_begin = ix.d_chunk_begin; sp.chunk_end = ix.d_chunk_end; sp.chunk_out = ix.d_chunk_out;
    sp.n_chunks = (int)ix.n_chunks;
    const dim3 grid((unsigned)((ix.n_chunks + 3) / 4)), block(256);
    (void)dispatch_nslot(nslot, [&](auto ns) -> int {
        constexpr int NS = decltype(ns)::value;
        hipLaunchKernelGGL((flda_stats_kernel<NS>), grid, block, 0, ctx->stream, sp);
        if (ix.n_multi > 0)
            hipLaunchKernelGGL((flda_stats_multi_kernel<NS>), dim3((unsigned)ix.n_multi), dim3(64), 0, ctx->stream, sp, ix.d_multi_id, ix.d_multi_first, ix.d_multi_count, (int)ix.n_multi);
        return TMVB_OK;
    });
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}
