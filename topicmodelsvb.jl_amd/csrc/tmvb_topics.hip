// tmvb_topics.hip -- model.topics behind train!, on the device.
//
// Every train! of the reference ends with
//     model.topics = [reverse(sortperm(vec(model.beta[i,:]))) for i in 1:model.K]          src/gpuLDA.jl:374 (CTM, CTPF, fLDA, fCTM alike)
// K stable ascending sorts of V values, read backwards.  On the host that was the largest single item of a train!(iter=150) call next to the
// loop itself (K = 50, V = 25 319: 65 ms of 209, profiles/r6_train_end_to_end.txt).  Here: one segmented radix sort of (value, column) pairs
// (rocPRIM through hipcub -- a library sort is what this is; stable, so ties keep the order sortperm gives them), written out reversed and 1-based.
#include "tmvb_internal.h"

#include <algorithm>
#include <hipcub/hipcub.hpp>

template <typename T>
static int dmalloc(T** p, size_t n)
{
    *p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    const hipError_t e = hipMalloc((void**)p, bytes);
    if (e != hipSuccess) { tmvb_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); return TMVB_ENOMEM; }
    return TMVB_OK;
}

// keys[i][j] = B[i + K j] (the model's column-major K x V field), vals[i][j] = j
static __global__ __launch_bounds__(256) void topics_keys_kernel(const double* __restrict__ B, int K, int64_t V, double* __restrict__ keys, int32_t* __restrict__ vals)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)K * V) return;
    const int64_t i = q / V, j = q - i * V;
    keys[q] = B[i + (int64_t)K * j];
    vals[q] = (int32_t)j;
}

// out[i][j] = 1 + (column of the j-th LARGEST value of row i) = the ascending order read backwards
static __global__ __launch_bounds__(256) void topics_reverse_kernel(const int32_t* __restrict__ sorted, int K, int64_t V, int32_t* __restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)K * V) return;
    const int64_t i = q / V, j = q - i * V;
    out[q] = sorted[i * V + (V - 1 - j)] + 1;
}

static __global__ void topics_offsets_kernel(int* __restrict__ off, int K, int64_t V)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= K) off[i] = (int)((int64_t)i * V);
}

extern "C" int tmvb_topic_order(tmvb_ctx* ctx, const double* beta, int32_t K, int64_t V, int32_t* topics)
{
    TMVB_REQUIRE(ctx != nullptr, TMVB_EINVAL, "tmvb_topic_order: ctx is NULL");
    TMVB_REQUIRE(K > 0 && V >= 0, TMVB_EINVAL, "tmvb_topic_order: K must be positive, V nonnegative");
    if (V == 0) return TMVB_OK;
    TMVB_REQUIRE(beta && topics, TMVB_EINVAL, "tmvb_topic_order: NULL argument");
    const int64_t n = (int64_t)K * V;
    TMVB_REQUIRE(n < (int64_t)INT32_MAX, TMVB_EINVAL, "tmvb_topic_order: K * V must fit int32");
    TMVB_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    double *d_in = nullptr, *d_keys = nullptr, *d_keys2 = nullptr;
    int32_t *d_vals = nullptr, *d_vals2 = nullptr;
    int* d_off = nullptr;
    void* d_tmp = nullptr;
    auto cleanup = [&] {
        (void)hipFree(d_in); (void)hipFree(d_keys); (void)hipFree(d_keys2); (void)hipFree(d_vals); (void)hipFree(d_vals2); (void)hipFree(d_off); (void)hipFree(d_tmp);
    };
    auto fail = [&](int rc) { cleanup(); return rc; };
    int rc;
    if ((rc = dmalloc(&d_in, (size_t)n)) || (rc = dmalloc(&d_keys, (size_t)n)) || (rc = dmalloc(&d_keys2, (size_t)n)) || (rc = dmalloc(&d_vals, (size_t)n)) ||
        (rc = dmalloc(&d_vals2, (size_t)n)) || (rc = dmalloc(&d_off, (size_t)K + 1)))
        return fail(rc);
#define TOPICS_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { tmvb_set_error("%s failed: %s", #call, hipGetErrorString(e_)); return fail(TMVB_EHIP); } } while (0)
    TOPICS_HIP(hipMemcpyAsync(d_in, beta, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(topics_keys_kernel, dim3(nb), dim3(256), 0, st, (const double*)d_in, (int)K, V, d_keys, d_vals);
    hipLaunchKernelGGL(topics_offsets_kernel, dim3((unsigned)((K + 256) / 256)), dim3(256), 0, st, d_off, (int)K, V);
    TOPICS_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    TOPICS_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, tmp_bytes, (const double*)d_keys, d_keys2, (const int32_t*)d_vals, d_vals2, (int)n, (int)K,
                                                           (const int*)d_off, (const int*)d_off + 1, 0, 64, st));
    TOPICS_HIP(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)));
    TOPICS_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(d_tmp, tmp_bytes, (const double*)d_keys, d_keys2, (const int32_t*)d_vals, d_vals2, (int)n, (int)K,
                                                           (const int*)d_off, (const int*)d_off + 1, 0, 64, st));
    hipLaunchKernelGGL(topics_reverse_kernel, dim3(nb), dim3(256), 0, st, (const int32_t*)d_vals2, (int)K, V, d_vals);
    TOPICS_HIP(hipGetLastError());
    TOPICS_HIP(hipMemcpyAsync(topics, d_vals, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TOPICS_HIP(hipStreamSynchronize(st));
#undef TOPICS_HIP
    cleanup();
    return TMVB_OK;
}
