// tmvb_termstats.h -- gather-side sufficient statistics shared by the LDA / CTM / CTPF engines.
//
// The reference's CPU path scatters `X_temp[:, ids] += phi .* counts'` per document
// (src/LDA.jl:131, src/CTM.jl:124, src/CTPF.jl:261,:276); its OpenCL path gathers a materialised
// phi through an inverted index (src/gpuLDA.jl:156-177).  Here phi is never stored: for all three
// models the last-sweep responsibilities factor as
//       phi[i,n] * c_n = w_n * T[i, id_n] * E[i, doc_n]  (+ eps * w_n for LDA)
// with a per-token scalar w_n and a per-document K-vector E, so the statistics column of an id j is
//       out[i,j] = T[i,j] * sum_{tokens n of j} w_n E[i, doc_n]  + eps * sum w_n .
// One wave per chunk of <= TMVB_CHUNK tokens of one id (lane = topic): coalesced K-float reads of E
// rows, no atomics, fixed summation order => bitwise reproducible statistics.
#pragma once
#include "tmvb_internal.h"

struct TermStatsParams {
    int K;
    int tstride;                 // column stride of T (KP for the padded gather layout)
    int ostride;                 // column stride of out
    const int32_t* tok_doc;      // inverted index
    const int32_t* tok_pos;
    const int32_t* chunk_id;
    const int32_t* chunk_begin;
    const int32_t* chunk_end;
    const int32_t* chunk_out;
    int n_chunks;
    const float* w;              // [nnz] per-token weight, CSR order
    const float* E;              // [M][K] per-document factor
    const float* T;              // [n_ids][tstride] per-id factor
    float eps;                   // additive epsilon (LDA: EPSILON, others 0)
    float base;                  // value added to every written entry (CTPF priors a / e; else 0)
    float* out;                  // [n_ids][ostride]
    float* partial;              // [n_slots][K + 1]  raw (sum w E | sum w) of multi-chunk ids
};

template <int NSLOT>
__global__ __launch_bounds__(256) void termstats_chunk_kernel(TermStatsParams p)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= p.n_chunks) return;
    const int K = p.K;
    const int j = p.chunk_id[c];
    const int b = p.chunk_begin[c], e = p.chunk_end[c];
    float acc[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) acc[s] = 0.0f;
    float wl = 0.0f;
    for (int t0 = b; t0 < e; t0 += 64) {
        const int tok = t0 + lane;
        const bool valid = tok < e;
        const int dd = valid ? p.tok_doc[tok] : 0;
        const float wv = valid ? p.w[p.tok_pos[tok]] : 0.0f;
        wl += wv;
        const int cnt = min(64, e - t0);
#pragma unroll 8
        for (int k = 0; k < cnt; ++k) {
            const int dk = __builtin_amdgcn_readlane(dd, k);
            const float wk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv), k));
            const float* erow = p.E + (int64_t)dk * K;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                const int i = lane + 64 * s;
                if (i < K) acc[s] = fmaf(wk, erow[i], acc[s]);
            }
        }
    }
    const float wsum = wave_sum(wl);
    const int slot = p.chunk_out[c];
    if (slot < 0) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int i = lane + 64 * s;
            if (i < K) p.out[(int64_t)j * p.ostride + i] = p.base + fmaf(p.T[(int64_t)j * p.tstride + i], acc[s], p.eps * wsum);
        }
    } else {
        float* pr = p.partial + (int64_t)slot * (K + 1);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int i = lane + 64 * s;
            if (i < K) pr[i] = acc[s];
        }
        if (lane == 0) pr[K] = wsum;
    }
}

// ids whose tokens span several chunks: sum their partials in chunk order (deterministic)
template <int NSLOT>
__global__ __launch_bounds__(256) void termstats_multi_kernel(TermStatsParams p, const int32_t* __restrict__ multi_id,
                                                              const int32_t* __restrict__ multi_first,
                                                              const int32_t* __restrict__ multi_count, int n_multi)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= n_multi) return;
    const int K = p.K;
    const int j = multi_id[m], first = multi_first[m], cnt = multi_count[m];
    float acc[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) acc[s] = 0.0f;
    float wsum = 0.0f;
    for (int c = 0; c < cnt; ++c) {
        const float* pr = p.partial + (int64_t)(first + c) * (K + 1);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int i = lane + 64 * s;
            if (i < K) acc[s] += pr[i];
        }
        wsum += pr[K];
    }
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int i = lane + 64 * s;
        if (i < K) p.out[(int64_t)j * p.ostride + i] = p.base + fmaf(p.T[(int64_t)j * p.tstride + i], acc[s], p.eps * wsum);
    }
}
