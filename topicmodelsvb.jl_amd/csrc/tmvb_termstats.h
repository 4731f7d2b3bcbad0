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
// rows, no atomics, fixed summation order => bitwise reproducible statistics.  The kernels ACCUMULATE into
// `out` (the reference's `X_temp +=`): an id is written by exactly one wave per pass, passes are ordered on
// one stream, and the M-step resets the statistics (update_beta! / update_alef! / update_he!).
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
    const float* w;              // [nnz] per-token weight, id-major (inverted index) order (legacy kernels)
    const float* tok_val;        // [nnz] count / rating per token, id-major order (recompute kernel)
    float keps;                  // K * eps added to the recomputed normaliser (LDA), 0 otherwise
    const float* E;              // per-document factor: [M][K] (scalar kernel) or [M][estride] zero padded (float4 kernels)
    int estride = 0;             // row stride of the padded E (>= KP; 0 = KP).  A 128-byte-aligned stride makes every
                                 // gathered row touch the minimum number of cache lines (13% faster gather at K = 50)
    const float* T;              // [n_ids][tstride] per-id factor
    float eps;                   // additive epsilon (LDA: EPSILON, others 0)
    float base;                  // value added to every written entry (CTPF priors a / e; else 0)
    float* out;                  // [n_ids][ostride]
    float* partial;              // [n_slots][K + 1]  raw (sum w E | sum w) of multi-chunk ids
    double* logz = nullptr;      // [n_chunks] or NULL (LOGZ instantiations of the recompute kernels): per chunk sum_n val_n * log2(s_n), the token
                                 // normalisers' share of update_elbo! (src/LDA.jl:78, :87-88; see lda_elbo_doc_kernel in tmvb_lda.hip).  Accumulated in
                                 // fp64 from the fp32 products: in fp32 the chunk sums' rounding added up to +- 0.2 on an ELBO of -1e8 -- as much as the
                                 // increments check_elbo!'s stop rule (delta < tol = 1) looks at near the plateau
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
        const float wv = valid ? p.w[tok] : 0.0f;
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
            if (i < K) p.out[(int64_t)j * p.ostride + i] += p.base + fmaf(p.T[(int64_t)j * p.tstride + i], acc[s], p.eps * wsum);
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

// Vectorised variant for K <= 256 (LPR = KP/4 <= 64 sixteen-byte chunks per row): E is stored with
// the padded row stride KP, a lane owns one float4 column chunk c = lane % LPR of row slot
// rs = lane / LPR, so ONE global_load_dwordx4 instruction fetches RPI = 64 / LPR document rows
// (832 B per wave instruction at K = 50 instead of 200 B).  Row slots are combined at the end in a
// fixed order through LDS.
template <int LPR_T>
__global__ __launch_bounds__(256) void termstats_chunk4_kernel(TermStatsParams p, int LPR_rt)
{
    __shared__ int2 dw_l[4][64];           // per wave: (doc, w bits) of the 64 tokens in flight
    __shared__ float4 red_l[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wv;
    const int LPR = LPR_T ? LPR_T : LPR_rt;
    const int KP = 4 * LPR;
    const int ES = p.estride ? p.estride : KP;
    const int RPI = 64 / LPR;              // rows per instruction
    const int rs = lane / LPR, cc = lane - rs * LPR;
    const bool lane_on = rs < RPI;
    const bool active = c < p.n_chunks;
    const int j = active ? p.chunk_id[c] : 0;
    const int b = active ? p.chunk_begin[c] : 0, e = active ? p.chunk_end[c] : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float wl = 0.0f;
    for (int t0 = b; t0 < e; t0 += 64) {
        const int tok = t0 + lane;
        const bool valid = tok < e;
        const int dd = valid ? p.tok_doc[tok] : 0;
        const float wv_ = valid ? p.w[tok] : 0.0f;
        wl += wv_;
        dw_l[wv][lane] = make_int2(dd, __builtin_bit_cast(int, wv_));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int cnt = min(64, e - t0);
        if (lane_on) {
#pragma unroll 8
            for (int k = rs; k < cnt; k += RPI) {
                const int2 dw = dw_l[wv][k];
                const float4 ev = *(const float4*)(p.E + (int64_t)dw.x * ES + 4 * cc);
                const float wk = __builtin_bit_cast(float, dw.y);
                acc.x = fmaf(wk, ev.x, acc.x); acc.y = fmaf(wk, ev.y, acc.y);
                acc.z = fmaf(wk, ev.z, acc.z); acc.w = fmaf(wk, ev.w, acc.w);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const float wsum = wave_sum(wl);
    red_l[wv][lane] = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!active || lane >= LPR) return;
    float4 tot = red_l[wv][lane];
    for (int r = 1; r < RPI; ++r) {
        const float4 o = red_l[wv][lane + r * LPR];
        tot.x += o.x; tot.y += o.y; tot.z += o.z; tot.w += o.w;
    }
    const float tv[4] = {tot.x, tot.y, tot.z, tot.w};
    const int slot = p.chunk_out[c];
    const int K = p.K;
    if (slot < 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = 4 * lane + u;
            if (i < K) p.out[(int64_t)j * p.ostride + i] += p.base + fmaf(p.T[(int64_t)j * p.tstride + i], tv[u], p.eps * wsum);
        }
    } else {
        float* pr = p.partial + (int64_t)slot * (K + 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = 4 * lane + u;
            if (i < K) pr[i] = tv[u];
        }
        if (lane == 0) pr[K] = wsum;
    }
}

// PAD form of the recompute pass with CPL column chunks per lane (see termstats_recompute_body below for the algorithm and the PAD contract:
// rows of E zero-padded to 4 * LANES * CPL floats, 32-bit byte offsets, keps > 0 or the select on the row).
template <int LPR, int LANES, int CPL, bool LOGZ = false>
__device__ __forceinline__ void termstats_recompute_cpl_body(const TermStatsParams& p)
{
    constexpr int SLOTS = 64 / LANES;
#ifndef TMVB_TS_CPL_U
#define TMVB_TS_CPL_U ((32 / SLOTS) < 2 ? 2 : ((32 / SLOTS) > 4 ? 4 : (32 / SLOTS)))
#endif
    constexpr int U = TMVB_TS_CPL_U;                     // row-slot groups in flight per wave: U * SLOTS rows, U * CPL loads per lane
    static_assert(64 % (U * SLOTS) == 0, "the 64 staged tokens are whole groups");
    __shared__ int2 dw_l[4][64];
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wv;
    const int rs = lane / LANES, cc = lane % LANES;
    const bool active = c < p.n_chunks;
    const int j = active ? p.chunk_id[c] : 0;
    const int b = active ? p.chunk_begin[c] : 0, e = active ? p.chunk_end[c] : 0;
    v2f tlo[CPL], thi[CPL], alo[CPL], ahi[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        float4 tj = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active && cc + LANES * q < LPR) tj = *(const float4*)(p.T + (int64_t)j * p.tstride + 4 * (cc + LANES * q));
        tlo[q] = v2f{tj.x, tj.y}; thi[q] = v2f{tj.z, tj.w};
        alo[q] = v2f{0.f, 0.f}; ahi[q] = v2f{0.f, 0.f};
    }
    const char* __restrict__ Eb = (const char*)p.E;
    const uint32_t esb = (uint32_t)p.estride * 4u, lane_off = 16u * (uint32_t)cc;
    float wl = 0.0f;
    double ll = 0.0;
    for (int t0 = b; t0 < e; t0 += 64) {
        const int tok = t0 + lane;
        const bool valid = tok < e;
        dw_l[wv][lane] = make_int2(valid ? p.tok_doc[tok] : 0, __builtin_bit_cast(int, valid ? p.tok_val[tok] : 0.0f));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int cnt = min(64, e - t0);
        for (int k0 = 0; k0 < cnt; k0 += U * SLOTS) {
            int2 dw[U];
            float4 ev[U][CPL];
#pragma unroll
            for (int u = 0; u < U; ++u) dw[u] = dw_l[wv][k0 + u * SLOTS + rs];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t ro = __umul24((uint32_t)dw[u].x, esb) + lane_off;
#pragma unroll
                for (int q = 0; q < CPL; ++q) ev[u][q] = *(const float4*)(Eb + (ro + 16u * (uint32_t)(LANES * q)));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v2f d2 = tlo[0] * v2f{ev[u][0].x, ev[u][0].y};
                d2 = __builtin_elementwise_fma(thi[0], v2f{ev[u][0].z, ev[u][0].w}, d2);
#pragma unroll
                for (int q = 1; q < CPL; ++q) {
                    d2 = __builtin_elementwise_fma(tlo[q], v2f{ev[u][q].x, ev[u][q].y}, d2);
                    d2 = __builtin_elementwise_fma(thi[q], v2f{ev[u][q].z, ev[u][q].w}, d2);
                }
                float part = d2.x + d2.y;
                part += dpp_f<0xB1>(part);
                part += dpp_f<0x4E>(part);
                if (LANES >= 8) part += dpp_f<0x141>(part);
                if (LANES >= 16) part += dpp_f<0x140>(part);          // every lane of the slot holds s_n - keps
                float wz = __builtin_bit_cast(float, dw[u].y) * __builtin_amdgcn_rcpf(part + p.keps);
                wz = (k0 + u * SLOTS + rs < cnt) ? wz : 0.0f;
                if constexpr (LOGZ) ll += (k0 + u * SLOTS + rs < cnt) ? (double)(__builtin_bit_cast(float, dw[u].y) * __builtin_amdgcn_logf(part + p.keps)) : 0.0;
                const v2f w2 = v2f{wz, wz};
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    alo[q] = __builtin_elementwise_fma(w2, v2f{ev[u][q].x, ev[u][q].y}, alo[q]);
                    ahi[q] = __builtin_elementwise_fma(w2, v2f{ev[u][q].z, ev[u][q].w}, ahi[q]);
                }
                wl += wz;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // combine the row slots (lanes cc, cc + LANES, ...) in a fixed order: within a 16-lane row by DPP rotations, across rows by lane exchange
    float tot[CPL][4];
#pragma unroll
    for (int q = 0; q < CPL; ++q) { tot[q][0] = alo[q].x; tot[q][1] = alo[q].y; tot[q][2] = ahi[q].x; tot[q][3] = ahi[q].y; }
    float wsum = wl;
    if (LANES <= 4) {
#pragma unroll
        for (int q = 0; q < CPL; ++q)
#pragma unroll
            for (int v = 0; v < 4; ++v) tot[q][v] += dpp_f<0x124>(tot[q][v]);     // row_ror:4
        wsum += dpp_f<0x124>(wsum);
        if constexpr (LOGZ) ll += __shfl_xor(ll, 4, 64);          // (lanes cc, cc + 4, cc + 8, cc + 12 of a row, as the two rotations)
    }
    if (LANES <= 8) {
#pragma unroll
        for (int q = 0; q < CPL; ++q)
#pragma unroll
            for (int v = 0; v < 4; ++v) tot[q][v] += dpp_f<0x128>(tot[q][v]);     // row_ror:8
        wsum += dpp_f<0x128>(wsum);
        if constexpr (LOGZ) ll += __shfl_xor(ll, 8, 64);
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
#pragma unroll
        for (int q = 0; q < CPL; ++q)
#pragma unroll
            for (int v = 0; v < 4; ++v) tot[q][v] += __shfl_xor(tot[q][v], o, 64);
        wsum += __shfl_xor(wsum, o, 64);
        if constexpr (LOGZ) ll += __shfl_xor(ll, o, 64);
    }
    if (!active || lane >= LANES) return;
    if constexpr (LOGZ) { if (lane == 0) p.logz[c] = ll; }
    const int slot = p.chunk_out[c];
    const int K = p.K;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int ch = lane + LANES * q;
        if (ch >= LPR) continue;
        const float tjv[4] = {tlo[q].x, tlo[q].y, thi[q].x, thi[q].y};
        if (slot < 0) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = 4 * ch + v;
                if (i < K) p.out[(int64_t)j * p.ostride + i] += p.base + fmaf(tjv[v], tot[q][v], p.eps * wsum);
            }
        } else {
            float* pr = p.partial + (int64_t)slot * (K + 1);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = 4 * ch + v;
                if (i < K) pr[i] = tot[q][v];
            }
            if (ch == 0) pr[K] = wsum;
        }
    }
}

// Recompute variant for K <= 128 (LPR <= 32): the per-token weight w_n = val_n / s_n is NOT read from
// memory -- a per-token value written in document order and read in id order has one scattered side,
// and the scattered 4-byte stores cost 0.24 ms of the 0.9 ms document pass at NSF scale -- but
// recomputed here: s_n = keps + sum_i T[i,j] E[i,doc_n] needs only the id's T row (fixed per chunk, in
// registers) and the document's E row, which this kernel fetches anyway.  LANES = 16 or 32 lanes per row
// slot (LPR active), 64 / LANES row slots per wave instruction; the dot product is a 4-step DPP row reduction
// (+ one v_permlane16_swap step joining the two 16-lane rows of a 32-lane slot).
// Round 5 (CPL > 1, PAD form only): a lane owns CPL column chunks of its row slot (chunks cc, cc + LANES, ...), so a row slot needs only
// LANES = ceil(LPR / CPL) rounded to 4 / 8 / 16 lanes and one wave instruction covers 64 / LANES rows; the per-row VALU work (reduction steps,
// reciprocal, weight, bookkeeping) is shared by more rows with fewer DPP steps (K = 100, 8 lanes x 4 chunks: 8 rows per instruction instead of 2,
// no v_permlane16_swap; 3.1 instead of ~15 VALU instructions per row).  Measured (profiles/r5_lda_experiments.txt (7)): K = 100 +1.6 % (860.0 /
// 862.9 -> 875.2 / 875.3 VB it/s with CPL = 4, 870.7 / 874.7 with 2); K = 50 -1 % (CPL = 2) and -7 % (CPL = 4: twice the 128-byte lines per row and
// instruction) -- the pass is bound by what the L2 delivers to the vector L1s (12 - 15 TB/s of gathered rows), not by its instructions, so the
// default is CPL = 4 at KP = 100 and the round-4 form (CPL = 1) at KP = 52; TMVB_TS_CPL overrides.
template <int LPR_T, int LANES, bool PAD = false, int CPL = 1, bool LOGZ = false>
__device__ __forceinline__ void termstats_recompute_body(const TermStatsParams& p, int LPR_rt)
{
    static_assert(LANES == 4 || LANES == 8 || LANES == 16 || LANES == 32, "termstats_recompute_kernel: 4, 8, 16 or 32 lanes per row slot");
    static_assert(CPL == 1 || (PAD && LPR_T > 0 && LANES <= 16), "several chunks per lane: PAD form with a compile-time row length only");
    if constexpr (CPL > 1) { termstats_recompute_cpl_body<LPR_T, LANES, CPL, LOGZ>(p); return; }
    constexpr int SLOTS = 64 / LANES;
    __shared__ int2 dw_l[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wv;
    const int LPR = LPR_T ? LPR_T : LPR_rt;
    const int KP = 4 * LPR;
    const int ES = p.estride ? p.estride : KP;
    const int rs = lane / LANES, cc = lane % LANES;
    const bool lane_on = cc < LPR;
    const bool active = c < p.n_chunks;
    const int j = active ? p.chunk_id[c] : 0;
    const int b = active ? p.chunk_begin[c] : 0, e = active ? p.chunk_end[c] : 0;
    float4 tj = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active && lane_on) tj = *(const float4*)(p.T + (int64_t)j * p.tstride + 4 * cc);
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f tlo = v2f{tj.x, tj.y}, thi = v2f{tj.z, tj.w};
    v2f alo = v2f{0.f, 0.f}, ahi = v2f{0.f, 0.f};          // packed fp32 accumulators (v_pk_fma_f32)
    float wl = 0.0f;
    double ll = 0.0;
    for (int t0 = b; t0 < e; t0 += 64) {
        const int tok = t0 + lane;
        const bool valid = tok < e;
        dw_l[wv][lane] = make_int2(valid ? p.tok_doc[tok] : 0, __builtin_bit_cast(int, valid ? p.tok_val[tok] : 0.0f));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int cnt = min(64, e - t0);
        // U row gathers in flight per wave: the loads are UNCONDITIONAL (a row slot past the end reads document 0's row and gets
        // weight 0; a lane past the last column chunk reads the row's zero padding when the rows are padded to 16 bytes x LANES),
        // so the block is straight-line code and the compiler issues all U loads before the first wait.  With the loads under
        // `if (row_on && lane_on)` it kept ONE gather in flight per wave (s_waitcnt vmcnt(0) right behind every load).
        constexpr int U = 4;
        const bool pad_ok = ES >= 4 * LANES;            // uniform
        // Fast form (round 4) when every row is padded with zeros to 4 * LANES floats and E is addressable by 32-bit byte offsets
        // (PAD: decided by the launcher).  The pass is VALU-issue bound next to the document kernels, and of
        // the ~27 VALU instructions it spent per row slot 9 were bookkeeping: four v_cndmask zeroing the pad lanes' E values (their T
        // values are zero and the pad of E is zero: the products vanish by themselves), a v_cndmask + v_mad_i64_i32 + v_lshl_add_u64
        // for a 64-bit row address (now one v_mad_u32_u24: uniform base + 32-bit offset, as the document kernels' tile loads), and a
        // Newton step behind v_rcp_f32 (the document kernels' own w_n is c_n * v_rcp_f32(s_n)).  A slot past the chunk's end reads the
        // staged entry (document 0, value 0): weight exactly 0 as long as its s is not 0, which keps > 0 guarantees for LDA; the
        // select on the row stays for the models with keps = 0.
        if constexpr (PAD) {
            const char* __restrict__ Eb = (const char*)p.E;
            const uint32_t esb = (uint32_t)ES * 4u, lane_off = 16u * (uint32_t)cc;
            for (int k0 = 0; k0 < cnt; k0 += U * SLOTS) {
                int2 dw[U];
                float4 ev[U];
#pragma unroll
                for (int u = 0; u < U; ++u) dw[u] = dw_l[wv][k0 + u * SLOTS + rs];          // k0 + 3 SLOTS + rs <= 63
#pragma unroll
                for (int u = 0; u < U; ++u) ev[u] = *(const float4*)(Eb + (__umul24((uint32_t)dw[u].x, esb) + lane_off));
                // no early exit for a row-slot group past the end (its weights are 0): with a uniform `break` between the groups the
                // compiler sank every load next to its use -- ONE gather in flight per wave again, 8 % slower than the form it replaces
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const v2f elo = v2f{ev[u].x, ev[u].y}, ehi = v2f{ev[u].z, ev[u].w};
                    const v2f d2 = __builtin_elementwise_fma(tlo, elo, thi * ehi);
                    float part = d2.x + d2.y;
                    part += dpp_f<0xB1>(part);
                    part += dpp_f<0x4E>(part);
                    part += dpp_f<0x141>(part);
                    part += dpp_f<0x140>(part);                       // all 16 lanes of a row hold the row's sum
                    if (LANES == 32) {
                        float sib = part;
                        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(part), "+v"(sib));
                        part += sib;
                    }
                    float wz = __builtin_bit_cast(float, dw[u].y) * __builtin_amdgcn_rcpf(part + p.keps);
                    wz = (k0 + u * SLOTS + rs < cnt) ? wz : 0.0f;
                    if constexpr (LOGZ) ll += (k0 + u * SLOTS + rs < cnt) ? (double)(__builtin_bit_cast(float, dw[u].y) * __builtin_amdgcn_logf(part + p.keps)) : 0.0;
                    const v2f w2 = v2f{wz, wz};
                    alo = __builtin_elementwise_fma(w2, elo, alo);
                    ahi = __builtin_elementwise_fma(w2, ehi, ahi);
                    wl += wz;
                }
            }
        } else
        for (int k0 = 0; k0 < cnt; k0 += U * SLOTS) {
            int2 dw[U];
            bool ron[U];
            float4 ev[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = k0 + u * SLOTS + rs;
                ron[u] = kk < cnt;
                dw[u] = dw_l[wv][min(kk, 63)];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float* src = p.E + (int64_t)(ron[u] ? dw[u].x : 0) * ES + 4 * cc;
                if (pad_ok) ev[u] = *(const float4*)src;
                else { ev[u] = make_float4(0.f, 0.f, 0.f, 0.f); if (lane_on) ev[u] = *(const float4*)src; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k0 + u * SLOTS >= cnt) break;                 // uniform: whole row-slot group past the end
                const bool row_on = ron[u];
                const v2f elo = lane_on ? v2f{ev[u].x, ev[u].y} : v2f{0.f, 0.f}, ehi = lane_on ? v2f{ev[u].z, ev[u].w} : v2f{0.f, 0.f};
                const v2f d2 = __builtin_elementwise_fma(tlo, elo, thi * ehi);
                float part = d2.x + d2.y;
                part += dpp_f<0xB1>(part);
                part += dpp_f<0x4E>(part);
                part += dpp_f<0x141>(part);
                part += dpp_f<0x140>(part);                       // all 16 lanes of a row hold the row's sum
                if (LANES == 32) {                                // rows 2r and 2r+1 form one slot: add the sibling row
                    float sib = part;
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(part), "+v"(sib));
                    part += sib;                                  // every lane of the slot holds s_n - keps
                }
                const float wz = row_on ? fast_div(__builtin_bit_cast(float, dw[u].y), part + p.keps) : 0.0f;
                if constexpr (LOGZ) ll += row_on ? (double)(__builtin_bit_cast(float, dw[u].y) * __builtin_amdgcn_logf(part + p.keps)) : 0.0;
                const v2f w2 = v2f{wz, wz};
                alo = __builtin_elementwise_fma(w2, elo, alo);
                ahi = __builtin_elementwise_fma(w2, ehi, ahi);
                wl += wz;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    float4 acc = make_float4(alo.x, alo.y, ahi.x, ahi.y);
    // combine the row slots (lanes cc, cc + LANES, ...) in a fixed order
    float4 tot = acc;
    float wsum = wl;
#pragma unroll
    for (int o = LANES; o <= 32; o <<= 1) {
        tot.x += __shfl_xor(tot.x, o, 64); tot.y += __shfl_xor(tot.y, o, 64);
        tot.z += __shfl_xor(tot.z, o, 64); tot.w += __shfl_xor(tot.w, o, 64);
        wsum += __shfl_xor(wsum, o, 64);
        if constexpr (LOGZ) ll += __shfl_xor(ll, o, 64);
    }
    if (!active || lane >= LPR) return;
    if constexpr (LOGZ) { if (lane == 0) p.logz[c] = ll; }
    const float tv[4] = {tot.x, tot.y, tot.z, tot.w};
    const float tjv[4] = {tj.x, tj.y, tj.z, tj.w};
    const int slot = p.chunk_out[c];
    const int K = p.K;
    if (slot < 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = 4 * lane + u;
            if (i < K) p.out[(int64_t)j * p.ostride + i] += p.base + fmaf(tjv[u], tv[u], p.eps * wsum);
        }
    } else {
        float* pr = p.partial + (int64_t)slot * (K + 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = 4 * lane + u;
            if (i < K) pr[i] = tv[u];
        }
        if (lane == 0) pr[K] = wsum;
    }
}

template <int LPR_T, int LANES, bool PAD = false, int CPL = 1, bool LOGZ = false>
__global__ __launch_bounds__(256) void termstats_recompute_kernel(TermStatsParams p, int LPR_rt)
{
    termstats_recompute_body<LPR_T, LANES, PAD, CPL, LOGZ>(p, LPR_rt);
}
// Two independent passes (CTPF: the term index and the reader index) in ONE launch, blockIdx.y selects the pass: run on two streams
// they cost a ~19 us cross-stream join in a 0.3 ms iteration.
template <int LPR_T, int LANES, bool PAD = false, int CPL = 1, bool LOGZ = false>
__global__ __launch_bounds__(256) void termstats_recompute2_kernel(TermStatsParams p0, TermStatsParams p1, int LPR_rt)
{
    if (blockIdx.y == 0) termstats_recompute_body<LPR_T, LANES, PAD, CPL, LOGZ>(p0, LPR_rt);
    else termstats_recompute_body<LPR_T, LANES, PAD, CPL, LOGZ>(p1, LPR_rt);
}

// ids whose tokens span several chunks: one workgroup per id, its 4 waves sum interleaved partial
// slots, then wave 0 combines the 4 sums in a fixed order (deterministic)
template <int NSLOT>
__device__ __forceinline__ void termstats_multi_body(const TermStatsParams& p, const int32_t* __restrict__ multi_id,
                                                     const int32_t* __restrict__ multi_first,
                                                     const int32_t* __restrict__ multi_count, int n_multi, float (&red)[4][64 * NSLOT + 1])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = blockIdx.x;
    if (m >= n_multi) return;
    const int K = p.K;
    const int j = multi_id[m], first = multi_first[m], cnt = multi_count[m];
    float acc[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) acc[s] = 0.0f;
    float wsum = 0.0f;
    // four partial rows of this wave in flight per trip, loaded unconditionally (a row past the end: the last row again, not
    // added; a lane past K: element K): with the loads under `if (i < K)` in a loop of unknown length the wave waited for each
    // row before asking for the next.  Same order of summation.
    for (int c0 = wv; c0 < cnt; c0 += 16) {
        float v[4][NSLOT], wk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* pr = p.partial + (int64_t)(first + min(c0 + 4 * u, cnt - 1)) * (K + 1);
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) { const int i = lane + 64 * s; v[u][s] = pr[i < K ? i : K]; }
            wk[u] = pr[K];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + 4 * u >= cnt) break;                    // wave-uniform
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) { const int i = lane + 64 * s; if (i < K) acc[s] += v[u][s]; }
            wsum += wk[u];
        }
    }
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) red[wv][lane + 64 * s] = acc[s];
    if (lane == 0) red[wv][64 * NSLOT] = wsum;
    __syncthreads();
    if (wv != 0) return;
    const float ws = (red[0][64 * NSLOT] + red[1][64 * NSLOT]) + (red[2][64 * NSLOT] + red[3][64 * NSLOT]);
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int i = lane + 64 * s;
        if (i < K) {
            const float a = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
            p.out[(int64_t)j * p.ostride + i] += p.base + fmaf(p.T[(int64_t)j * p.tstride + i], a, p.eps * ws);
        }
    }
}

template <int NSLOT>
__global__ __launch_bounds__(256) void termstats_multi_kernel(TermStatsParams p, const int32_t* __restrict__ multi_id,
                                                              const int32_t* __restrict__ multi_first,
                                                              const int32_t* __restrict__ multi_count, int n_multi)
{
    __shared__ float red[4][64 * NSLOT + 1];
    termstats_multi_body<NSLOT>(p, multi_id, multi_first, multi_count, n_multi, red);
}
struct TermStatsMulti { const int32_t* id; const int32_t* first; const int32_t* count; int n; };
template <int NSLOT>
__global__ __launch_bounds__(256) void termstats_multi2_kernel(TermStatsParams p0, TermStatsMulti m0, TermStatsParams p1, TermStatsMulti m1)
{
    __shared__ float red[4][64 * NSLOT + 1];
    if (blockIdx.y == 0) termstats_multi_body<NSLOT>(p0, m0.id, m0.first, m0.count, m0.n, red);
    else termstats_multi_body<NSLOT>(p1, m1.id, m1.first, m1.count, m1.n, red);
}
