// probe: cost of broadcasting lane j's value to the wave: v_readlane (SGPR) vs ds_bpermute (VGPR) vs LDS broadcast read
#include <hip/hip_runtime.h>
#include <cstdio>
#define R 52
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, const float* in, long long* cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) float row[64];
    const int lane = threadIdx.x;
    float H[R];
    for (int q = 0; q < R; ++q) H[q] = in[(blockIdx.x * 64 + lane) * R + q] * 1e-3f + (q == lane ? 1.0f : 0.0f);
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            float srow[R];
            if (MODE == 0) {
#pragma unroll
                for (int kk = j; kk < R; ++kk) srow[kk] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, H[kk]), j));
            } else if (MODE == 1) {
                int addr = 4 * j;
                asm volatile("" : "+v"(addr));
#pragma unroll
                for (int kk = j; kk < R; ++kk) srow[kk] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, H[kk])));
            } else {
                if (lane == j) {
#pragma unroll
                    for (int kk = (j & ~3); kk < R; kk += 4) *(float4*)(row + kk) = make_float4(H[kk], H[kk + 1], H[kk + 2], H[kk + 3]);
                }
                __syncthreads();
#pragma unroll
                for (int kk = (j & ~3); kk < R; kk += 4) { float4 v = *(const float4*)(row + kk); srow[kk] = v.x; srow[kk + 1] = v.y; srow[kk + 2] = v.z; srow[kk + 3] = v.w; }
                __syncthreads();
            }
            __builtin_amdgcn_sched_barrier(0);
            const float rp = 1.0f / srow[j];
            const float f = (lane == j) ? 0.0f : H[j] * rp;
#pragma unroll
            for (int kk = j + 1; kk < R; ++kk) H[kk] = fmaf(-f, srow[kk], H[kk]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = clock64();
    float s = 0; for (int q = 0; q < R; ++q) s += H[q];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* nm, int blocks)
{
    float *in, *out; long long* cyc;
    hipMalloc(&in, (size_t)blocks * 64 * R * 4); hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    hipMemset(in, 0, (size_t)blocks * 64 * R * 4);
    const int iters = 20;
    k<MODE><<<blocks, 64>>>(out, in, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<blocks, 64>>>(out, in, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-10s blocks=%6d  wall %.3f ms  -> %.2f ns per solve aggregate, wave0 cycles/solve %lld\n", nm, blocks, ms, ms * 1e6 / ((double)blocks * iters), c / iters);
}
int main()
{
    for (int blocks : {256 * 4, 256 * 8, 256 * 16}) {
        run<0>("readlane", blocks); run<1>("bpermute", blocks); run<2>("lds_b128", blocks);
    }
}
