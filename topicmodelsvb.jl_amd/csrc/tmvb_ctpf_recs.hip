// tmvb_ctpf_recs.hip -- CTPF recommendation post-processing on the device.
//
// Reference: the tail of train!(model::CTPF) src/CTPF.jl:379-399 (same code in src/gpuCTPF.jl:711-731, on the host):
//     Eeta = he ./ vav ;  scores[d, :] = sum(Eeta .* (gimel[d] ./ dalet + zayin[d] ./ het), dims = 1)
//     urecs[u] = findall(ur)[reverse(sortperm(scores[ur, u]))]   ur = documents not in libs[u]
//     drecs[d] = findall(nr)[reverse(sortperm(scores[d, nr]))]   nr = users that are not readers of d
// i.e. scores = X * Y' with X (M x K) = E[theta] + E[epsilon], Y (U x K) = E[eta]; every user gets all
// unread documents ranked by descending score, every document all non-readers.  reverse(sortperm(.)) =
// descending score, equal scores in DESCENDING index order.
//
// Device plan: (1) X, Y in fp32 from the resident state; (2) one f32-MFMA pass (v_mfma_f32_32x32x2_f32) that
// computes every 32 x 32 tile twice -- X Y' and Y X' -- so that BOTH sort-key layouts are written with
// coalesced 128-byte stores: keyD[d][U-1-u] (segment = document) and keyU[u][M-1-d] (segment = user); the
// index order inside a segment is reversed so that a STABLE descending sort reproduces the reference's tie
// order; (3) the (document, reader) pairs are overwritten with -inf; (4) rocPRIM segmented radix sort
// (pairs, descending, stable) with a counting/transform iterator as the value input (no index array in HBM).
// The M x U matrix is written once per layout (HBM-bound: 2 x 4MU bytes) and sorted in place of the reference's
// M + U host sortperm calls on a 754 MB fp64 matrix.
#include "tmvb_internal.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// X[d][k] = gimel[d][k] / dalet[k] + zayin[d][k] / het[k]   (rows of K floats, zero padded to KS)
__global__ __launch_bounds__(256) void ctpf_expect_docs_kernel(const float* __restrict__ gimel, const float* __restrict__ zayin,
                                                               const double* __restrict__ rates, int K, int KS, int64_t M,
                                                               float* __restrict__ X)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= M * KS) return;
    const int64_t d = q / KS; const int k = (int)(q - d * KS);
    float v = 0.0f;
    if (k < K) v = (float)((double)gimel[d * K + k] / rates[2 * K + k] + (double)zayin[d * K + k] / rates[3 * K + k]);
    X[q] = v;
}

// Y[u][k] = he[u][k] / vav[k]
__global__ __launch_bounds__(256) void ctpf_expect_users_kernel(const float* __restrict__ he, const double* __restrict__ rates,
                                                                int K, int KS, int64_t U, float* __restrict__ Y)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= U * KS) return;
    const int64_t u = q / KS; const int k = (int)(q - u * KS);
    Y[q] = (k < K) ? (float)((double)he[u * K + k] / rates[K + k]) : 0.0f;
}

// One wave per 32 (documents) x 32 (users) tile.  A/B operand layout of v_mfma_f32_32x32x2_f32: lane l holds
// A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; C/D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5).  cDU = X Y' (rows = documents), cUD = Y X' (rows = users).
__global__ __launch_bounds__(64) void ctpf_scores_mfma_kernel(const float* __restrict__ X, const float* __restrict__ Y, int KS,
                                                              int64_t M, int64_t U, int tiles_u,
                                                              float* __restrict__ keyD, float* __restrict__ keyU,
                                                              float* __restrict__ scores /* column-major M x U, or NULL */)
{
    const int lane = threadIdx.x;
    const int64_t td = blockIdx.x / tiles_u, tu = blockIdx.x - td * tiles_u;
    const int64_t d0 = td * 32, u0 = tu * 32;
    const int i = lane & 31, kk = lane >> 5;
    const int64_t dr = min(d0 + i, M - 1), ur = min(u0 + i, U - 1);
    const float* xr = X + dr * KS + kk;
    const float* yr = Y + ur * KS + kk;
    f32x16 cDU = {0}, cUD = {0};
    for (int k = 0; k < KS; k += 2) {
        const float a = xr[k], b = yr[k];
        cDU = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, cDU, 0, 0, 0);
        cUD = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, cUD, 0, 0, 0);
    }
    const int col = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int64_t d = d0 + row, u = u0 + col;               // cDU: row = document, col = user
        if (keyD && d < M && u < U) keyD[d * U + (U - 1 - u)] = cDU[r];
        const int64_t u2 = u0 + row, d2 = d0 + col;             // cUD: row = user, col = document
        if (u2 < U && d2 < M) {
            if (keyU) keyU[u2 * M + (M - 1 - d2)] = cUD[r];
            if (scores) scores[u2 * M + d2] = cUD[r];
        }
    }
}

// (document, reader) pairs are not recommended: their keys sort last
__global__ __launch_bounds__(256) void ctpf_mask_read_kernel(const int64_t* __restrict__ rdr_ptr, const int32_t* __restrict__ readers,
                                                             int64_t M, int64_t U, float* __restrict__ keyD, float* __restrict__ keyU)
{
    const int64_t d = blockIdx.x;
    for (int64_t q = rdr_ptr[d] + threadIdx.x; q < rdr_ptr[d + 1]; q += blockDim.x) {
        const int64_t u = readers[q];
        keyD[d * U + (U - 1 - u)] = -INFINITY;
        keyU[u * M + (M - 1 - d)] = -INFINITY;
    }
}

// position p of a segment of length n holds index n - 1 - p
struct tmvb_rev_index {
    unsigned n;
    __host__ __device__ int operator()(unsigned q) const { return (int)(n - 1u - q % n); }
};
struct tmvb_seg_offset {
    unsigned n;
    __host__ __device__ unsigned operator()(unsigned s) const { return s * n; }
};

static int sort_segments(tmvb_ctx* ctx, float* keys, float* keys_tmp, int32_t* out, unsigned n_seg, unsigned seg_len)
{
    if (n_seg == 0 || seg_len == 0) return TMVB_OK;
    auto values = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), tmvb_rev_index{seg_len});
    auto offsets = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), tmvb_seg_offset{seg_len});
    size_t tmp_bytes = 0;
    const unsigned size = n_seg * seg_len;
    hipError_t e = rocprim::segmented_radix_sort_pairs_desc(nullptr, tmp_bytes, keys, keys_tmp, values, out, size, n_seg,
                                                            offsets, offsets + 1, 0, 32, ctx->stream);
    TMVB_REQUIRE(e == hipSuccess, TMVB_EHIP, "rocprim::segmented_radix_sort_pairs_desc (size query): %s", hipGetErrorString(e));
    void* tmp = nullptr;
    TMVB_HIP(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
    e = rocprim::segmented_radix_sort_pairs_desc(tmp, tmp_bytes, keys, keys_tmp, values, out, size, n_seg, offsets, offsets + 1,
                                                 0, 32, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    TMVB_REQUIRE(e == hipSuccess && e2 == hipSuccess, TMVB_EHIP, "rocprim::segmented_radix_sort_pairs_desc: %s",
                 hipGetErrorString(e != hipSuccess ? e : e2));
    return TMVB_OK;
}

extern "C" int tmvb_ctpf_recommend(tmvb_ctpf* h, double* scores, int32_t* drecs, int32_t* drec_count, int32_t* urecs,
                                   int32_t* urec_count, float* ms_scores, float* ms_rank)
{
    tmvb_ctpf_view v;
    int rc = tmvb_ctpf_view_of(h, &v);
    if (rc) return rc;
    tmvb_ctx* ctx = v.ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    const int64_t M = v.M, U = v.U;
    const int K = v.K, KS = (K + 1) / 2 * 2;
    TMVB_REQUIRE((drecs == nullptr) == (urecs == nullptr), TMVB_EINVAL, "tmvb_ctpf_recommend: drecs and urecs come together");
    TMVB_REQUIRE(M * U < (int64_t)4294967295ll, TMVB_ESHAPE, "tmvb_ctpf_recommend: M * U = %lld exceeds the 32-bit sort size", (long long)(M * U));
    if (M == 0 || U == 0) return TMVB_OK;
    const bool rank = drecs != nullptr;
    float *X = nullptr, *Y = nullptr, *keyD = nullptr, *keyU = nullptr, *ktmp = nullptr, *sc = nullptr;
    int32_t *rankD = nullptr, *rankU = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(X); (void)hipFree(Y); (void)hipFree(keyD); (void)hipFree(keyU); (void)hipFree(ktmp); (void)hipFree(sc);
        (void)hipFree(rankD); (void)hipFree(rankU);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (e2) (void)hipEventDestroy(e2);
    };
#define RECS_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { tmvb_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); cleanup(); return TMVB_EHIP; } } while (0)
    const size_t MU = (size_t)M * U;
    RECS_TRY(hipMalloc((void**)&X, (size_t)M * KS * sizeof(float)));
    RECS_TRY(hipMalloc((void**)&Y, (size_t)U * KS * sizeof(float)));
    if (rank) {
        RECS_TRY(hipMalloc((void**)&keyD, MU * sizeof(float)));
        RECS_TRY(hipMalloc((void**)&keyU, MU * sizeof(float)));
        RECS_TRY(hipMalloc((void**)&ktmp, MU * sizeof(float)));
        RECS_TRY(hipMalloc((void**)&rankD, MU * sizeof(int32_t)));
        RECS_TRY(hipMalloc((void**)&rankU, MU * sizeof(int32_t)));
    }
    if (scores) RECS_TRY(hipMalloc((void**)&sc, MU * sizeof(float)));
    RECS_TRY(hipEventCreate(&e0)); RECS_TRY(hipEventCreate(&e1)); RECS_TRY(hipEventCreate(&e2));
    RECS_TRY(hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(ctpf_expect_docs_kernel, dim3((unsigned)((M * KS + 255) / 256)), dim3(256), 0, ctx->stream, v.gimel, v.zayin,
                       v.rates, K, KS, M, X);
    hipLaunchKernelGGL(ctpf_expect_users_kernel, dim3((unsigned)((U * KS + 255) / 256)), dim3(256), 0, ctx->stream, v.he, v.rates, K,
                       KS, U, Y);
    const int tiles_u = (int)((U + 31) / 32);
    const int64_t tiles = ((M + 31) / 32) * tiles_u;
    hipLaunchKernelGGL(ctpf_scores_mfma_kernel, dim3((unsigned)tiles), dim3(64), 0, ctx->stream, X, Y, KS, M, U, tiles_u, keyD, keyU, sc);
    if (rank)
        hipLaunchKernelGGL(ctpf_mask_read_kernel, dim3((unsigned)M), dim3(256), 0, ctx->stream, v.corp->d_rdr_ptr, v.corp->d_readers, M, U,
                           keyD, keyU);
    RECS_TRY(hipGetLastError());
    RECS_TRY(hipEventRecord(e1, ctx->stream));
    if (rank) {
        // candidates per segment = segment length - distinct (document, reader) pairs masked above
        for (int64_t u = 0; u < U; ++u) urec_count[u] = (int32_t)M;
        std::vector<int32_t> rd;
        for (int64_t d = 0; d < M; ++d) {
            rd.assign(v.corp->h_readers.begin() + v.corp->h_rdr_ptr[d], v.corp->h_readers.begin() + v.corp->h_rdr_ptr[d + 1]);
            std::sort(rd.begin(), rd.end());
            rd.erase(std::unique(rd.begin(), rd.end()), rd.end());
            drec_count[d] = (int32_t)(U - (int64_t)rd.size());
            for (int32_t u : rd) urec_count[u]--;
        }
        if ((rc = sort_segments(ctx, keyD, ktmp, rankD, (unsigned)M, (unsigned)U))) { cleanup(); return rc; }
        if ((rc = sort_segments(ctx, keyU, ktmp, rankU, (unsigned)U, (unsigned)M))) { cleanup(); return rc; }
    }
    RECS_TRY(hipEventRecord(e2, ctx->stream));
    RECS_TRY(hipEventSynchronize(e2));
    if (ms_scores) RECS_TRY(hipEventElapsedTime(ms_scores, e0, e1));
    if (ms_rank) RECS_TRY(hipEventElapsedTime(ms_rank, e1, e2));
    if (scores) {
        std::vector<float> tmp(MU);
        RECS_TRY(hipMemcpyAsync(tmp.data(), sc, MU * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        RECS_TRY(hipStreamSynchronize(ctx->stream));
        for (size_t q = 0; q < MU; ++q) scores[q] = (double)tmp[q];
    }
    if (rank) {
        RECS_TRY(hipMemcpy(drecs, rankD, MU * sizeof(int32_t), hipMemcpyDeviceToHost));
        RECS_TRY(hipMemcpy(urecs, rankU, MU * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
#undef RECS_TRY
    cleanup();
    return TMVB_OK;
}
