// tmvb_common_kernels.h -- device kernels and host helpers shared by the LDA / CTM / CTPF engines:
// deterministic column sums, the beta normalisation of update_beta!, wave helpers, small host
// upload/download utilities and the tile-size / bucket logic of the per-document kernels.
#pragma once
#include "tmvb_internal.h"
#include "tmvb_termstats.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>

#define TMVB_MAX_NSLOT 16                      // K <= 1024
#define TMVB_MAX_TILE_BYTES (64 * 1024)
#define TMVB_BIG_TILE_BYTES (156 * 1024)         // of the 160 KiB per CU: needs hipFuncAttributeMaxDynamicSharedMemorySize
#define TMVB_REDUCE_BLOCKS 1024                 // 4 waves each: the column sums are latency-bound, 256 blocks left a CU with 4 waves (54 us for the 26 MB of Elogtheta next to a statistics pass)

// one wave per workgroup: __syncthreads() lowers to an LDS fence (the s_barrier is elided)
#define WAVE_LDS_FENCE() __syncthreads()
// LDS fence for data private to ONE wave of a multi-wave workgroup (no s_barrier)
#define WAVE_PRIVATE_LDS_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

template <int CTRL>
__device__ __forceinline__ float4 dpp_add4(float4 v)
{
    v.x += dpp_f<CTRL>(v.x); v.y += dpp_f<CTRL>(v.y); v.z += dpp_f<CTRL>(v.z); v.w += dpp_f<CTRL>(v.w);
    return v;
}


// NOTE: hipcc 7.2 mis-selects __builtin_amdgcn_permlane{32,16}_swap when both results feed one add
// (it emits `v_add v, r0, r0`), so the swap is issued through inline asm.  hipcc inserts no hazard
// wait states inside asm, and the swap DOES need them around VALU producers/consumers of its operands:
// without the s_nop pair the T=1 register kernel summed garbage into topic 0 (found on hardware).
__device__ __forceinline__ void swap_add32(float& a, float b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a += b;                                   // lanes < 32: a.lo + a.hi, lanes >= 32: b.lo + b.hi
}
__device__ __forceinline__ void swap_add16(float& a, float b)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a += b;                                   // even rows: sums of a, odd rows: sums of b
}
typedef float v2f_t __attribute__((ext_vector_type(2)));

// two independent swaps under one pair of hazard nops.  PK: the two sums as one v_pk_add_f32 -- which costs a v_mov_b32 detour per swap
// pair (the tied asm operand of the pair's second half is not coalesced): worth it for R > 64 (K = 100: 451 vs 437 it/s), two
// v_add_f32 are better for R <= 64 (K = 50: 851 vs 837 it/s).
template <bool PK>
__device__ __forceinline__ void swap_add32_x2(float& a0, float b0, float& a1, float b1)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
    if constexpr (PK) { const v2f_t s = v2f_t{a0, a1} + v2f_t{b0, b1}; a0 = s.x; a1 = s.y; }
    else { a0 += b0; a1 += b1; }
}
template <bool PK>
__device__ __forceinline__ void swap_add16_x2(float& a0, float b0, float& a1, float b1)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
    if constexpr (PK) { const v2f_t s = v2f_t{a0, a1} + v2f_t{b0, b1}; a0 = s.x; a1 = s.y; }
    else { a0 += b0; a1 += b1; }
}
template <int CTRL>
__device__ __forceinline__ void dpp_stage(float& a, float b, bool hi)
{
    const float keep = hi ? b : a, give = hi ? a : b;
    a = keep + dpp_f<CTRL>(give);
}

// The products enter as a functor prod2(j) -> (value 2j, value 2j + 1) and are formed right before their stage-32
// swap; stages 32 and 16 are fused per output pair, so only R / 4 partial sums (not R products) are ever live
// next to the register tile.  Stage 32: value i meets value i + R / 2 (pair j meets pair j + R / 4).  Stage 16
// pairs whole register PAIRS (units) so that the adds stay packed: of the U = R / 4 units left by stage 32, unit u
// meets unit u + ceil(U / 2); for odd U the middle unit folds its own two halves.
template <int R, class F>
__device__ __forceinline__ void lane_reduce_scatter(F&& prod2, float (&res)[(R + 63) / 64], int lane)
{
    static_assert(R % 4 == 0, "lane_reduce_scatter: R must be a multiple of 4");
    constexpr int U = R / 4, H = (U + 1) / 2, m1 = R / 2, h1 = (m1 + 1) / 2;
    float p[h1 + 1];
    auto stage32 = [&](int j, float& x, float& y) {
        const v2f_t a = prod2(j), b = prod2(j + R / 4);
        x = a.x; y = a.y;
        swap_add32_x2<(R > 64)>(x, b.x, y, b.y);
    };
#pragma unroll
    for (int u = 0; u < U / 2; ++u) {
        float ax, ay, bx, by;
        stage32(u, ax, ay);
        stage32(u + H, bx, by);
        swap_add16_x2<(R > 64)>(ax, bx, ay, by);
        p[2 * u] = ax; p[2 * u + 1] = ay;
        if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);     // keep later products from being hoisted (register pressure)
    }
    if (U & 1) {
        float cx, cy;
        stage32(U / 2, cx, cy);
        swap_add16(cx, cy);
        p[U - 1] = cx;
    }
    constexpr int m2 = h1, h2 = (m2 + 1) / 2;
    const bool b8 = lane & 8, b4 = lane & 4, b2 = lane & 2, b1 = lane & 1;
#pragma unroll
    for (int i = 0; i < h2; ++i) {
        if (i + h2 < m2) dpp_stage<0x140>(p[i], p[i + h2], b8);      // row_mirror
        else p[i] += dpp_f<0x140>(p[i]);
    }
    constexpr int m3 = h2, h3 = (m3 + 1) / 2;
#pragma unroll
    for (int i = 0; i < h3; ++i) {
        if (i + h3 < m3) dpp_stage<0x141>(p[i], p[i + h3], b4);      // row_half_mirror
        else p[i] += dpp_f<0x141>(p[i]);
    }
    constexpr int m4 = h3, h4 = (m4 + 1) / 2;
#pragma unroll
    for (int i = 0; i < h4; ++i) {
        if (i + h4 < m4) dpp_stage<0x4E>(p[i], p[i + h4], b2);       // quad_perm [2,3,0,1]
        else p[i] += dpp_f<0x4E>(p[i]);
    }
    constexpr int m5 = h4, h5 = (m5 + 1) / 2;
#pragma unroll
    for (int i = 0; i < h5; ++i) {
        if (i + h5 < m5) dpp_stage<0xB1>(p[i], p[i + h5], b1);       // quad_perm [1,0,3,2]
        else p[i] += dpp_f<0xB1>(p[i]);
    }
    // p[0 .. ceil(R/64)) now hold the totals: lane L of result register s owns topic pi(s, L)
#pragma unroll
    for (int s = 0; s < (R + 63) / 64; ++s) res[s] = p[s];
}

// Compile-time replay of lane_reduce_scatter's pairing: lane_of_topic[q] = s * 64 + L of the primary owner
// of topic q, topic_of_lane[s * 64 + L] = the topic owned there (-1 for duplicates), spare = a slot of the
// last result register that owns no topic (-1 if none).  With the map a constant, the e_q broadcasts of the
// register-tile kernel are v_readlane with an IMMEDIATE lane (no selector SGPRs, no SGPR spills).
template <int R>
struct RegLaneMap {
    static constexpr int NS = (R + 63) / 64;
    int lane_of_topic[R];
    int topic_of_lane[NS * 64];
    int spare;
    constexpr RegLaneMap() : lane_of_topic{}, topic_of_lane{}, spare(-1)
    {
        int cur[R][64] = {}, nxt[R][64] = {};
        for (int q = 0; q < R; ++q) for (int l = 0; l < 64; ++l) cur[q][l] = q;
        int m = R;
        const int Ds[6] = {32, 16, 8, 4, 2, 1};
        for (int st = 0; st < 6; ++st) {
            const int D = Ds[st], h = (m + 1) / 2;
            if (st == 1) {                      // unit pairing of stage 16 (see lane_reduce_scatter)
                const int U = m / 2, H = (U + 1) / 2;
                for (int u = 0; u < U / 2; ++u)
                    for (int c = 0; c < 2; ++c)
                        for (int l = 0; l < 64; ++l) nxt[2 * u + c][l] = (l & D) ? cur[2 * (u + H) + c][l] : cur[2 * u + c][l];
                if (U & 1) for (int l = 0; l < 64; ++l) nxt[U - 1][l] = (l & D) ? cur[U][l] : cur[U - 1][l];
            } else {
                for (int i = 0; i < h; ++i)
                    for (int l = 0; l < 64; ++l) nxt[i][l] = (i + h < m && (l & D)) ? cur[i + h][l] : cur[i][l];
            }
            for (int i = 0; i < h; ++i) for (int l = 0; l < 64; ++l) cur[i][l] = nxt[i][l];
            m = h;
        }
        for (int q = 0; q < R; ++q) lane_of_topic[q] = -1;
        for (int i = 0; i < NS * 64; ++i) topic_of_lane[i] = -1;
        for (int sl = 0; sl < NS; ++sl)
            for (int l = 0; l < 64; ++l) {
                const int q = cur[sl][l];
                if (lane_of_topic[q] < 0) { lane_of_topic[q] = sl * 64 + l; topic_of_lane[sl * 64 + l] = q; }
            }
        for (int l = 63; l >= 0; --l) if (topic_of_lane[(NS - 1) * 64 + l] < 0) { spare = (NS - 1) * 64 + l; break; }
    }
};
template <int R> constexpr RegLaneMap<R> kRegLaneMap{};

// host copy of the compile-time map (the register-tile kernels exist for R = 4 * odd <= 100)
static inline int tmvb_reg_lane_maps(int R, std::vector<int>& topic_of_lane, std::vector<int>& lane_of_topic)
{
    auto copy = [&](const auto& m) {
        topic_of_lane.assign(m.topic_of_lane, m.topic_of_lane + sizeof(m.topic_of_lane) / sizeof(int));
        lane_of_topic.assign(m.lane_of_topic, m.lane_of_topic + sizeof(m.lane_of_topic) / sizeof(int));
        return (int)(sizeof(m.topic_of_lane) / sizeof(int) / 64);
    };
    switch (R) {
#define TMVB_LANE_MAP_CASE(RV) case RV: return copy(kRegLaneMap<RV>);
        TMVB_LANE_MAP_CASE(4) TMVB_LANE_MAP_CASE(12) TMVB_LANE_MAP_CASE(20) TMVB_LANE_MAP_CASE(28) TMVB_LANE_MAP_CASE(36)
        TMVB_LANE_MAP_CASE(44) TMVB_LANE_MAP_CASE(52) TMVB_LANE_MAP_CASE(60) TMVB_LANE_MAP_CASE(68) TMVB_LANE_MAP_CASE(76)
        TMVB_LANE_MAP_CASE(84) TMVB_LANE_MAP_CASE(92) TMVB_LANE_MAP_CASE(100)
#undef TMVB_LANE_MAP_CASE
        default: return 0;
    }
}


// ------------------------------------------------------------------------------ reductions
// partial[block][K] (double) = sum over a strided subset of columns of X (K x ncols, fp32)
template <int NSLOT>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int64_t ncols, int K,
                                                             double* __restrict__ partial)
{
    __shared__ double red[4][64 * NSLOT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wv, nw = (int64_t)gridDim.x * 4;
    double acc[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) acc[s] = 0.0;
#pragma unroll 8
    for (int64_t c = gw; c < ncols; c += nw) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int i = lane + 64 * s;
            if (i < K) acc[s] += (double)X[c * K + i];
        }
    }
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) red[wv][lane + 64 * s] = acc[s];
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256)
        partial[(int64_t)blockIdx.x * K + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// The same with a SECOND buffer folded in on the way: X += Xb, Xb = 0, and the column sums are those of the merged values (LDA, round 5: the last
// statistics pass of the pipelined E-step writes its own buffer so that it need not wait for the passes before it; update_beta! merges here, in the
// pass over the statistics it makes anyway).  Same summation order as colsum_partial_kernel.
template <int NSLOT>
__global__ __launch_bounds__(256) void colsum_merge_partial_kernel(float* __restrict__ X, float* __restrict__ Xb, int64_t ncols, int K,
                                                                   double* __restrict__ partial)
{
    __shared__ double red[4][64 * NSLOT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wv, nw = (int64_t)gridDim.x * 4;
    double acc[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) acc[s] = 0.0;
#pragma unroll 8
    for (int64_t c = gw; c < ncols; c += nw) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int i = lane + 64 * s;
            if (i < K) {
                const float v = X[c * K + i] + Xb[c * K + i];
                X[c * K + i] = v; Xb[c * K + i] = 0.0f;
                acc[s] += (double)v;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) red[wv][lane + 64 * s] = acc[s];
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256)
        partial[(int64_t)blockIdx.x * K + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
// X += Xb, Xb = 0 (the merge alone, for the callers that hand the statistics out before update_beta!)
static __global__ __launch_bounds__(256) void stats_merge_kernel(float* __restrict__ X, float* __restrict__ Xb, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) { X[q] += Xb[q]; Xb[q] = 0.0f; }
}

// out_d[i] = sum_b partial[b][i]: one wave per output, fixed reduction tree (deterministic); optional fp32 copy
static __global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ partial, int nblocks, int K,
                                                           double* __restrict__ out_d, float* __restrict__ out_f)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= K) return;
    double s = 0.0;
    for (int b0 = 0; b0 < nblocks; b0 += 64 * 16) {          // 16 independent loads in flight per lane, added in a fixed order
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int b = b0 + lane + 64 * u; v[u] = b < nblocks ? partial[(int64_t)b * K + i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    s = wave_sum_d(s);
    if (lane == 0) {
        if (out_d) out_d[i] = s;
        if (out_f) out_f[i] = (float)s;
    }
}

// update_beta!(model)  src/LDA.jl:121-125:  beta_new = S ./ rowsum(S);  S <- 0
// S is dense [V][K]; beta_new is the padded gather layout [V][KP] (pad columns zero).
// pw_partial (or NULL): per-block fp64 partial of sum_{i,j} S[j][i] * log(beta_new[j][i] + eps) = E_q[log p(w)] of the
// ELBO (src/LDA.jl:65 summed over the corpus: sum_n c_n phi_in = S), fixed summation order per block.
// beta_old (or NULL; the padded layout of the beta the E-step read): the partial becomes sum S * (log(beta_new + eps) - log(beta_old + eps)) -- the
// statistics' share of E_q[log p(w)] - E_q[log q(z)] in the decomposed update_elbo! (lda_elbo_doc_kernel, tmvb_lda.hip).
static __global__ __launch_bounds__(256) void beta_norm_kernel(float* __restrict__ S, const double* __restrict__ rowsum,
                                                            float* __restrict__ beta_new, int K, int KP, int64_t V,
                                                            double* __restrict__ pw_partial, float eps, const float* __restrict__ beta_old = nullptr,
                                                            float eps_old = -1.0f,       // the epsilon inside log(beta_old + .): < 0 = eps (LDA), 0 for CTM
                                                            const double* __restrict__ logz = nullptr, int64_t n_logz = 0, double* __restrict__ lz_partial = nullptr)
{
    if (eps_old < 0.0f) eps_old = eps;
    if (lz_partial) {                            // this block's slice of the statistics passes' per-chunk sums of c log2 s (TermStatsParams::logz), fixed order
        const int64_t per = (n_logz + gridDim.x - 1) / gridDim.x;
        const int64_t b = (int64_t)blockIdx.x * per, e = b + per < n_logz ? b + per : n_logz;
        double lz = 0.0;
        for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) lz += logz[i];
        lz = wave_sum_d(lz);
        __shared__ double lred[4];
        if ((threadIdx.x & 63) == 0) lred[threadIdx.x >> 6] = lz;
        __syncthreads();
        if (threadIdx.x == 0) lz_partial[blockIdx.x] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
    }
    extern __shared__ double rinv[];
    __shared__ double red[4];
    for (int i = threadIdx.x; i < K; i += blockDim.x) rinv[i] = 1.0 / rowsum[i];
    __syncthreads();
    const int64_t total = V * KP;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double pw = 0.0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const int64_t j = q / KP;
        const int i = (int)(q - j * KP);
        float o = 0.0f;
        if (i < K) {
            const float sv = S[j * K + i];
            o = (float)((double)sv * rinv[i]);
            // (S = 0 contributes 0: CTM's phi has no epsilon, so a zero beta_old entry has S = 0 and log 0 must not meet it)
            // (the decomposed form takes its two logarithms in fp64: S weighs them with up to 1e4 per entry, and fp32 logarithms left +- 0.2 of rounding
            //  on an ELBO whose increments the stop rule compares with 1)
            if (pw_partial) pw += beta_old ? (sv != 0.0f ? (double)sv * (log((double)(o + eps)) - log((double)(beta_old[q] + eps_old))) : 0.0)
                                           : (double)sv * (double)logf(o + eps);
            S[j * K + i] = 0.0f;
        }
        beta_new[q] = o;
    }
    if (pw_partial) {
        pw = wave_sum_d(pw);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pw;
        __syncthreads();
        if (threadIdx.x == 0) pw_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

__device__ __forceinline__ double wave_min_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}


template <typename F>
static int dispatch_nslot(int nslot, F&& f)
{
    switch (nslot) {
        case 1: return f(std::integral_constant<int, 1>());
        case 2: return f(std::integral_constant<int, 2>());
        case 3: case 4: return f(std::integral_constant<int, 4>());
        case 5: case 6: case 7: case 8: return f(std::integral_constant<int, 8>());
        default: return f(std::integral_constant<int, 16>());
    }
}

static size_t tmvb_tile_bytes(int rows, int KP) { return ((size_t)rows * KP + KP + 3 * (size_t)rows) * sizeof(float); }


template <typename T>
static int dmalloc(T** p, size_t n)
{
    *p = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    hipError_t e = hipMalloc((void**)p, bytes);
    if (e != hipSuccess) {
        tmvb_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return TMVB_ENOMEM;
    }
    return TMVB_OK;
}


static int upload_f32(tmvb_ctx* ctx, float* dst, const double* src, size_t n)
{
    std::vector<float> tmp(n);
    for (size_t q = 0; q < n; ++q) tmp[q] = (float)src[q];
    TMVB_HIP(hipMemcpyAsync(dst, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    return TMVB_OK;
}

static int download_f32(tmvb_ctx* ctx, double* dst, const float* src, size_t n)
{
    std::vector<float> tmp(n);
    TMVB_HIP(hipMemcpyAsync(tmp.data(), src, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t q = 0; q < n; ++q) dst[q] = (double)tmp[q];
    return TMVB_OK;
}



// blocks of a column sum: at least 128 columns per block (32 per wave) -- a function of the column count only, so the summation
// order is the same on every rank and in every run
static inline int tmvb_colsum_blocks(int64_t ncols) { return (int)std::min<int64_t>(TMVB_REDUCE_BLOCKS, std::max<int64_t>(1, (ncols + 127) / 128)); }

// deterministic column sums of a K x ncols fp32 matrix into out_d (fp64) and/or out_f (fp32)
static inline int tmvb_colsum(tmvb_ctx* ctx, int nslot, int K, const float* X, int64_t ncols, double* d_partial,
                              double* out_d, float* out_f, hipStream_t on_stream = nullptr)
{
    hipStream_t st = on_stream ? on_stream : ctx->stream;
    int nb = tmvb_colsum_blocks(ncols);
    int rc = dispatch_nslot(nslot, [&](auto ns) -> int {
        constexpr int NS = decltype(ns)::value;
        hipLaunchKernelGGL((colsum_partial_kernel<NS>), dim3(nb), dim3(256), 0, st, X, ncols, K, d_partial);
        return TMVB_OK;
    });
    if (rc) return rc;
    TMVB_HIP(hipGetLastError());
    hipLaunchKernelGGL(colsum_final_kernel, dim3((K + 3) / 4), dim3(256), 0, st, d_partial, nb, K, out_d, out_f);
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

// Two independent column sums in one pair of launches (blockIdx.y selects the job): the CTPF M-step is a chain of
// 5-8 us kernels, so halving the launches is what counts there.  K <= 64.
struct tmvb_colsum_job { const float* X; int64_t ncols; double* partial; double* out_d; float* out_f; };

static __global__ __launch_bounds__(256) void colsum2_partial_kernel(tmvb_colsum_job j0, tmvb_colsum_job j1, int K)
{
    __shared__ double red[4][64];
    const tmvb_colsum_job j = blockIdx.y ? j1 : j0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wv, nw = (int64_t)gridDim.x * 4;
    double acc = 0.0;
    if (lane < K) {
#pragma unroll 8
        for (int64_t c = gw; c < j.ncols; c += nw) acc += (double)j.X[c * K + lane];
    }
    red[wv][lane] = acc;
    __syncthreads();
    if ((int)threadIdx.x < K) j.partial[(int64_t)blockIdx.x * K + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

static __global__ __launch_bounds__(256) void colsum2_final_kernel(tmvb_colsum_job j0, tmvb_colsum_job j1, int nblocks, int K)
{
    const tmvb_colsum_job j = blockIdx.y ? j1 : j0;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= K) return;
    double s = 0.0;
    for (int b0 = 0; b0 < nblocks; b0 += 64 * 16) {          // 16 independent loads in flight per lane, added in a fixed order
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int b = b0 + lane + 64 * u; v[u] = b < nblocks ? j.partial[(int64_t)b * K + i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    s = wave_sum_d(s);
    if (lane == 0) {
        if (j.out_d) j.out_d[i] = s;
        if (j.out_f) j.out_f[i] = (float)s;
    }
}

// same partition of the columns over TMVB_REDUCE_BLOCKS blocks for both jobs (a job with fewer columns just has idle blocks),
// hence the same summation order as tmvb_colsum with that block count
static inline int tmvb_colsum2(tmvb_ctx* ctx, int K, tmvb_colsum_job j0, tmvb_colsum_job j1, hipStream_t on_stream = nullptr)
{
    TMVB_REQUIRE(K <= 64, TMVB_EINVAL, "tmvb_colsum2: K <= 64");
    hipStream_t st = on_stream ? on_stream : ctx->stream;
    const int64_t nmax = std::max(j0.ncols, j1.ncols);
    const int nb = tmvb_colsum_blocks(nmax);
    hipLaunchKernelGGL(colsum2_partial_kernel, dim3(nb, 2), dim3(256), 0, st, j0, j1, K);
    hipLaunchKernelGGL(colsum2_final_kernel, dim3((K + 3) / 4, 2), dim3(256), 0, st, j0, j1, nb, K);
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

// gather-side statistics pass over an inverted index (tmvb_termstats.h)
// true when the statistics pass recomputes the per-token weights (no wtok stores needed in the document kernels)
static inline bool tmvb_termstats_recomputes(int KP, bool e_padded) { return e_padded && KP / 4 <= 32; }
// column chunks per lane of the PAD form of the recompute pass (tmvb_termstats.h): default 1 at KP = 52 (the round-4 form), 4 at KP = 100;
// TMVB_TS_CPL = 1 | 2 | 4 overrides (A/B)
static inline int tmvb_termstats_cpl(int lpr)
{
    static const int env = [] { const char* e = getenv("TMVB_TS_CPL"); const int c = e ? atoi(e) : 0; return (c == 1 || c == 2 || c == 4) ? c : 0; }();
    return env ? env : (lpr == 25 ? 4 : 1);
}

// slice >= 0: only the ids of vocabulary slice `slice` of an index built with id cuts (tmvb_inv_index::slice_*)
static inline int tmvb_launch_termstats(tmvb_ctx* ctx, int nslot, int KP, bool e_padded, const tmvb_inv_index& ix,
                                        TermStatsParams tp, hipStream_t on_stream = nullptr, int slice = -1)
{
    if (ix.n_chunks <= 0) return TMVB_OK;
    hipStream_t st = on_stream ? on_stream : ctx->stream;
    int64_t c0 = 0, c1 = ix.n_chunks, m0 = 0, m1 = ix.n_multi;
    if (slice >= 0) {
        TMVB_REQUIRE((size_t)slice + 1 < ix.slice_chunk.size(), TMVB_EINVAL, "statistics pass: the index has no vocabulary slice %d", slice);
        c0 = ix.slice_chunk[(size_t)slice]; c1 = ix.slice_chunk[(size_t)slice + 1];
        m0 = ix.slice_multi[(size_t)slice]; m1 = ix.slice_multi[(size_t)slice + 1];
        if (c1 <= c0) return TMVB_OK;
    }
    const int64_t n_chunks = c1 - c0, n_multi = m1 - m0;
    tp.tok_doc = ix.d_doc; tp.tok_pos = ix.d_pos; tp.chunk_id = ix.d_chunk_id + c0; tp.chunk_begin = ix.d_chunk_begin + c0;
    tp.chunk_end = ix.d_chunk_end + c0; tp.chunk_out = ix.d_chunk_out + c0; tp.n_chunks = (int)n_chunks;
    tp.tok_val = ix.d_val;
    if (tp.logz) tp.logz += c0;                 // per chunk of the index
    int rc = dispatch_nslot(nslot, [&](auto ns) -> int {
        constexpr int NS = decltype(ns)::value;
        const dim3 grid((unsigned)((n_chunks + 3) / 4)), block(256);
        const int lpr = KP / 4;
        if (e_padded && lpr <= 32) {       // recompute w from (T row, E row, count): no per-token weights in memory
            // PAD form: rows zero-padded to 4 * LANES floats, 32-bit byte offsets into E, 24-bit document ids (tmvb_termstats.h)
            const int lanes = lpr <= 16 ? 16 : 32;
            const bool pad = tp.estride >= 4 * lanes && ix.n_docs < (1 << 24) && (uint64_t)ix.n_docs * (uint64_t)tp.estride * 4u < (1ull << 32) &&
                             !(getenv("TMVB_STATS_PAD") && atoi(getenv("TMVB_STATS_PAD")) == 0);
            const int cpl = tmvb_termstats_cpl(lpr);
            // TS(...): the instantiation, or -- tp.logz set, translation units that define TMVB_TS_LOGZ (LDA) -- its LOGZ form, which also leaves the
            // chunk's sum of val * log2(normaliser) for update_elbo!
#ifdef TMVB_TS_LOGZ
#define TS(LPRV, LANESV, PADV, CPLV) do { if (tp.logz) hipLaunchKernelGGL((termstats_recompute_kernel<LPRV, LANESV, PADV, CPLV, true>), grid, block, 0, st, tp, lpr); \
                                          else hipLaunchKernelGGL((termstats_recompute_kernel<LPRV, LANESV, PADV, CPLV, false>), grid, block, 0, st, tp, lpr); } while (0)
#else
#define TS(LPRV, LANESV, PADV, CPLV) do { TMVB_REQUIRE(tp.logz == nullptr, TMVB_EINVAL, "statistics pass: no log-normaliser form in this translation unit"); \
                                          hipLaunchKernelGGL((termstats_recompute_kernel<LPRV, LANESV, PADV, CPLV, false>), grid, block, 0, st, tp, lpr); } while (0)
#endif
            if (lpr == 13 && pad && cpl == 2) TS(13, 8, true, 2);
            else if (lpr == 13 && pad && cpl == 4) TS(13, 4, true, 4);
            else if (lpr == 25 && pad && cpl == 2) TS(25, 16, true, 2);
            else if (lpr == 25 && pad && cpl == 4) TS(25, 8, true, 4);
            else if (lpr == 13 && pad) TS(13, 16, true, 1);
            else if (lpr == 25 && pad) TS(25, 32, true, 1);
            else if (lpr == 13) TS(13, 16, false, 1);
            else if (lpr == 25) TS(25, 32, false, 1);
            else if (lpr <= 16) TS(0, 16, false, 1);
            else TS(0, 32, false, 1);
#undef TS
        } else if (e_padded) {
            if (lpr == 13) hipLaunchKernelGGL((termstats_chunk4_kernel<13>), grid, block, 0, st, tp, lpr);
            else if (lpr == 25) hipLaunchKernelGGL((termstats_chunk4_kernel<25>), grid, block, 0, st, tp, lpr);
            else hipLaunchKernelGGL((termstats_chunk4_kernel<0>), grid, block, 0, st, tp, lpr);
        } else {
            hipLaunchKernelGGL((termstats_chunk_kernel<NS>), grid, block, 0, st, tp);
        }
        if (n_multi > 0)
            hipLaunchKernelGGL((termstats_multi_kernel<NS>), dim3((unsigned)n_multi), dim3(256), 0, st, tp,
                               ix.d_multi_id + m0, ix.d_multi_first + m0, ix.d_multi_count + m0, (int)n_multi);
        return TMVB_OK;
    });
    if (rc) return rc;
    TMVB_HIP(hipGetLastError());
    return TMVB_OK;
}

// Two statistics passes over two inverted indices in one pair of launches (recompute variant only; returns TMVB_EINVAL otherwise so
// that the caller can fall back to two launches): see termstats_recompute2_kernel.
static inline int tmvb_launch_termstats2(tmvb_ctx* ctx, int nslot, int KP, const tmvb_inv_index& ix0, TermStatsParams tp0,
                                         const tmvb_inv_index& ix1, TermStatsParams tp1)
{
    const int lpr = KP / 4;
    if (!(lpr <= 32) || ix0.n_chunks <= 0 || ix1.n_chunks <= 0) return TMVB_EINVAL;
    hipStream_t st = ctx->stream;
    auto fill = [](TermStatsParams& tp, const tmvb_inv_index& ix) {
        tp.tok_doc = ix.d_doc; tp.tok_pos = ix.d_pos; tp.chunk_id = ix.d_chunk_id; tp.chunk_begin = ix.d_chunk_begin;
        tp.chunk_end = ix.d_chunk_end; tp.chunk_out = ix.d_chunk_out; tp.n_chunks = (int)ix.n_chunks; tp.tok_val = ix.d_val;
    };
    fill(tp0, ix0); fill(tp1, ix1);
    const dim3 grid((unsigned)((std::max(ix0.n_chunks, ix1.n_chunks) + 3) / 4), 2), block(256);
    const int lanes = lpr <= 16 ? 16 : 32;
    const int64_t nd = std::max(ix0.n_docs, ix1.n_docs);
    const bool pad = tp0.estride >= 4 * lanes && tp1.estride == tp0.estride && nd < (1 << 24) && (uint64_t)nd * (uint64_t)tp0.estride * 4u < (1ull << 32) &&
                     !(getenv("TMVB_STATS_PAD") && atoi(getenv("TMVB_STATS_PAD")) == 0);
    const int cpl = tmvb_termstats_cpl(lpr);
    TMVB_REQUIRE((tp0.logz == nullptr) == (tp1.logz == nullptr), TMVB_EINVAL, "statistics pass pair: log-normaliser sums for both indices or for neither");
#ifdef TMVB_TS_LOGZ
#define TS2(LPRV, LANESV, PADV, CPLV) do { if (tp0.logz) hipLaunchKernelGGL((termstats_recompute2_kernel<LPRV, LANESV, PADV, CPLV, true>), grid, block, 0, st, tp0, tp1, lpr); \
                                           else hipLaunchKernelGGL((termstats_recompute2_kernel<LPRV, LANESV, PADV, CPLV, false>), grid, block, 0, st, tp0, tp1, lpr); } while (0)
#else
#define TS2(LPRV, LANESV, PADV, CPLV) do { TMVB_REQUIRE(tp0.logz == nullptr, TMVB_EINVAL, "statistics pass pair: no log-normaliser form in this translation unit"); \
                                           hipLaunchKernelGGL((termstats_recompute2_kernel<LPRV, LANESV, PADV, CPLV, false>), grid, block, 0, st, tp0, tp1, lpr); } while (0)
#endif
    if (lpr == 13 && pad && cpl == 2) TS2(13, 8, true, 2);
    else if (lpr == 13 && pad && cpl == 4) TS2(13, 4, true, 4);
    else if (lpr == 13 && pad) TS2(13, 16, true, 1);
    else if (lpr == 13) TS2(13, 16, false, 1);
    else if (lpr == 25) TS2(25, 32, false, 1);
    else if (lpr <= 16) TS2(0, 16, false, 1);
    else TS2(0, 32, false, 1);
#undef TS2
    TMVB_HIP(hipGetLastError());
    const int64_t nm = std::max(ix0.n_multi, ix1.n_multi);
    if (nm > 0) {
        const TermStatsMulti m0{ix0.d_multi_id, ix0.d_multi_first, ix0.d_multi_count, (int)ix0.n_multi};
        const TermStatsMulti m1{ix1.d_multi_id, ix1.d_multi_first, ix1.d_multi_count, (int)ix1.n_multi};
        int rc = dispatch_nslot(nslot, [&](auto ns) -> int {
            constexpr int NS = decltype(ns)::value;
            hipLaunchKernelGGL((termstats_multi2_kernel<NS>), dim3((unsigned)nm, 2), dim3(256), 0, st, tp0, m0, tp1, m1);
            return TMVB_OK;
        });
        if (rc) return rc;
        TMVB_HIP(hipGetLastError());
    }
    return TMVB_OK;
}

// LDS-tile buckets for the documents of `order` (sorted by descending length) that are longer than
// `min_len_exclusive`; `extra_rows` = per-row side arrays (floats) next to the KP-float tile row.
// Returns the number of documents bucketed (a prefix of `order`).
static inline int64_t tmvb_build_lds_buckets(const std::vector<int64_t>& len, const std::vector<int32_t>& order, int64_t M,
                                             int KP, int64_t min_len_exclusive, int extra_rows,
                                             std::vector<tmvb_bucket>& buckets, size_t max_tile_bytes = TMVB_MAX_TILE_BYTES)
{
    (void)extra_rows;
    int64_t n_lds = 0;
    while (n_lds < M && len[order[n_lds]] > min_len_exclusive) ++n_lds;
    int64_t pos = 0;
    if (n_lds == 0) return 0;
    std::vector<int> tiles;
    for (int r = 32; r <= 8192; r += (r < 256 ? 32 : 128)) {
        if (tmvb_tile_bytes(r, KP) + (size_t)extra_rows * 0 + 2 * (size_t)KP * sizeof(float) > max_tile_bytes) break;
        tiles.push_back(r);
    }
    if (tiles.empty()) tiles.push_back(4);
    const size_t b0 = buckets.size();
    const int tmax = tiles.back();
    int64_t cnt = 0;
    while (pos + cnt < n_lds && len[order[pos + cnt]] > tmax) ++cnt;   // stream chunks through the largest tile
    if (cnt) buckets.push_back({pos, cnt, tmax, 0});
    pos += cnt;
    for (int b = (int)tiles.size() - 1; b >= 0 && pos < n_lds; --b) {
        const int64_t lo = (b > 0) ? tiles[b - 1] : -1;
        cnt = 0;
        while (pos + cnt < n_lds && len[order[pos + cnt]] > lo) ++cnt;
        if (cnt) buckets.push_back({pos, cnt, tiles[b], 0});
        pos += cnt;
    }
    // merge small buckets into their larger neighbour (a launch needs enough waves to matter)
    for (size_t b = b0 + 1; b < buckets.size();) {
        if (buckets[b].count < 512 && buckets[b - 1].tile_rows >= buckets[b].tile_rows &&
            len[order[buckets[b - 1].first + buckets[b - 1].count - 1]] <= buckets[b - 1].tile_rows) {
            buckets[b - 1].count += buckets[b].count;
            buckets.erase(buckets.begin() + b);
        } else {
            ++b;
        }
    }
    return pos;
}
