// gather_probe.hip -- how fast can 208-byte rows be gathered at random from a table that does not fit in
// L2, and does padding the row stride to 256 B (two full 128-byte lines) help?
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/gp tools/probes/gather_probe.hip && /tmp/gp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ tab, const int* __restrict__ idx, int64_t n,
                                                      int stride, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, rs = lane >> 4, cc = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t per = 256;                       // tokens per wave, like one chunk
    const int64_t b = wave * per;
    if (b >= n) return;
    const int64_t e = b + per < n ? b + per : n;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int64_t t = b + rs; t < e; t += 4) {
        const int r = idx[t];
        if (cc < 13) {
            const float4 v = *(const float4*)(tab + (int64_t)r * stride + 4 * cc);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[wave] = acc.x;
}

int main()
{
    const int64_t n = 11300000;
    for (int rows : {4096, 8192, 16384, 32768, 65536, 128804}) {
        for (int stride : {64}) {
            std::vector<int> h(n);
            unsigned s = 12345u;
            for (int64_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)((s >> 8) % (unsigned)rows); }
            float* tab; int* idx; float* out;
            hipMalloc(&tab, (size_t)rows * stride * 4 + 64); hipMemset(tab, 0, (size_t)rows * stride * 4 + 64);
            hipMalloc(&idx, n * 4); hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
            hipMalloc(&out, (n / 256 + 4) * 4);
            const int nb = (int)((n + 1023) / 1024);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(gather_kernel, dim3(nb), dim3(256), 0, 0, tab, idx, n, stride, out);
            hipEventRecord(a);
            for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(gather_kernel, dim3(nb), dim3(256), 0, 0, tab, idx, n, stride, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
            printf("rows=%6d stride=%2d floats: %.3f ms  %.2f TB/s (208 B per row)\n", rows, stride, ms, n * 208.0 / ms / 1e9);
            hipFree(tab); hipFree(idx); hipFree(out);
        }
    }
    return 0;
}
