// grid_probe.hip -- checks tmvb_gridtile.h's 16-lane reduce-scatter (v_permlane32/16_swap + bank-masked DPP stages) against the
// compile-time ownership map GridMap<M>, for the value counts the LDA kernels instantiate (M = LPR + 1).
// Build: hipcc -O3 --offload-arch=gfx950 -I include -I topicmodelsvb.jl_amd/csrc tools/probes/grid_probe.hip -o tools/probes/grid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "tmvb_gridtile.h"

template <int M>
__global__ void probe(const float* in, float* out)
{
    const int lane = threadIdx.x;
    float v[M];
#pragma unroll
    for (int i = 0; i < M; ++i) v[i] = in[i * 64 + lane];
    float res[(M + 15) / 16];
    grid_reduce_scatter<M>(v, res);
#pragma unroll
    for (int r = 0; r < (M + 15) / 16; ++r) out[r * 64 + lane] = res[r];
}

template <int M>
static int run()
{
    constexpr int NS = (M + 15) / 16;
    std::vector<float> h(M * 64), o(NS * 64);
    for (int i = 0; i < M; ++i) for (int l = 0; l < 64; ++l) h[i * 64 + l] = (float)((i * 131 + l * 17) % 251) + 0.25f * (float)(l & 3);
    float *din, *dout;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, o.size() * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe<M>, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    const GridMap<M>& g = kGridMap<M>;
    for (int r = 0; r < NS; ++r)
        for (int l = 0; l < 64; ++l) {
            const int a = l >> 2, b = l & 3, i = g.idx[r][a];
            if (i < 0) continue;
            float want = 0.0f;
            for (int aa = 0; aa < 16; ++aa) want += h[i * 64 + 4 * aa + b];
            if (fabsf(want - o[r * 64 + l]) > 1e-3f * fabsf(want)) { if (bad < 5) printf("M=%d slot %d lane %d (a=%d b=%d) index %d: got %g want %g\n", M, r, l, a, b, i, o[r * 64 + l], want); ++bad; }
        }
    int owned = 0;
    for (int i = 0; i < M; ++i) owned += g.a_of[i] >= 0;
    printf("M=%2d: %s (%d mismatches, %d of %d indices owned)\n", M, bad ? "FAIL" : "ok", bad, owned, M);
    hipFree(din); hipFree(dout);
    return bad || owned != M;
}

int main()
{
    int bad = 0;
    bad += run<2>(); bad += run<4>(); bad += run<6>(); bad += run<8>(); bad += run<10>(); bad += run<12>(); bad += run<14>(); bad += run<16>();
    bad += run<18>(); bad += run<22>(); bad += run<26>(); bad += run<7>(); bad += run<13>();
    printf(bad ? "GRID PROBE FAILED\n" : "GRID PROBE OK\n");
    return bad ? 1 : 0;
}
