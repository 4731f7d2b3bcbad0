// matvec_probe.hip -- the inner product of the batched CTM Newton solve (lane = document): y = S p with the K x K matrix S
// shared by all lanes (scalar loads -> SGPR operands of v_pk_fma_f32) and the vectors p, y per lane in VGPRs.
// S is streamed through SGPRs in groups of G s_load_dwordx16 (inline asm: a plain C++ load is loop invariant, the compiler
// hoists all K*K values out of the solver loop and spills them to VGPR lanes); group g + 1 is in flight while group g is
// consumed (SMEM returns out of order, so every wait is lgkmcnt(0)).
// Prints shader cycles per mat-vec per wave at 1 and 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 matvec_probe.hip -o matvec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <utility>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int G> struct SGroup { v16f b[G]; };

template <int OFF>
__device__ __forceinline__ v16f sload16(const float* S)
{
    v16f v;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(S), "n"(OFF));
    return v;
}
template <int G>
__device__ __forceinline__ void swait(SGroup<G>& g)
{
    if constexpr (G == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g.b[0]));
    else if constexpr (G == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g.b[0]), "+s"(g.b[1]));
    else if constexpr (G == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g.b[0]), "+s"(g.b[1]), "+s"(g.b[2]));
}

template <int R, int G, int GI, int... K>
__device__ __forceinline__ void issue(SGroup<G>& g, const float* S, std::integer_sequence<int, K...>)
{
    constexpr int NB = (R * R + 15) / 16;
    ((g.b[K] = (GI * G + K < NB) ? sload16<(GI * G + K < NB ? (GI * G + K) * 64 : 0)>(S) : g.b[K]), ...);
}

template <int R, int G, int GI>
__device__ __forceinline__ void consume(const SGroup<G>& g, const float (&p)[R], v2f (&y)[R / 2])
{
#pragma unroll
    for (int k = 0; k < G; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int f = (GI * G + k) * 16 + 2 * e;
            if (f < R * R) {
                const int j = f / R, i = f % R;
                y[i / 2] = __builtin_elementwise_fma(v2f{g.b[k][2 * e], g.b[k][2 * e + 1]}, v2f{p[j], p[j]}, y[i / 2]);
            }
        }
    }
}

template <int R, int G, int GI, int NG>
__device__ __forceinline__ void pipeline(SGroup<G>& cur, SGroup<G>& nxt, const float* S, const float (&p)[R], v2f (&y)[R / 2])
{
    if constexpr (GI < NG) {
        if constexpr (GI + 1 < NG) issue<R, G, GI + 1>(nxt, S, std::make_integer_sequence<int, G>{});
        consume<R, G, GI>(cur, p, y);
        if constexpr (GI + 1 < NG) swait<G>(nxt);
        pipeline<R, G, GI + 1, NG>(nxt, cur, S, p, y);
    }
}

template <int R, int G>
__device__ __forceinline__ void matvec(const float* S, const float (&p)[R], v2f (&y)[R / 2])
{
    constexpr int NB = (R * R + 15) / 16, NG = (NB + G - 1) / G;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) y[i] = v2f{0.f, 0.f};
    SGroup<G> a, b;
    issue<R, G, 0>(a, S, std::make_integer_sequence<int, G>{});
    swait<G>(a);
    pipeline<R, G, 0, NG>(a, b, S, p, y);
}

template <int R, int G, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void probe(const float* S, float* out, long long* cyc, int iters)
{
    const int lane = threadIdx.x;
    float p[R];
#pragma unroll
    for (int i = 0; i < R; ++i) p[i] = 1.0f + 1e-3f * (lane + i);
    v2f y[R / 2];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        matvec<R, G>(S, p, y);
#pragma unroll
        for (int i = 0; i < R / 2; ++i) { p[2 * i] = fmaf(y[i].x, 1e-3f, p[2 * i] * 0.5f); p[2 * i + 1] = fmaf(y[i].y, 1e-3f, p[2 * i + 1] * 0.5f); }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) acc += p[i];
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

// instruction-cache probe: NC textual copies of the mat-vec (~13 KB of code each) executed back to back in one loop
template <int R, int NC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void icache_probe(const float* S, float* out, long long* cyc, int iters)
{
    const int lane = threadIdx.x;
    float p[R];
#pragma unroll
    for (int i = 0; i < R; ++i) p[i] = 1.0f + 1e-3f * (lane + i);
    v2f y[R / 2];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        auto step = [&](auto tag) {
            constexpr int T = decltype(tag)::value;
            if constexpr (T < NC) {
                matvec<R, 2>(S, p, y);
#pragma unroll
                for (int i = 0; i < R / 2; ++i) { p[2 * i] = fmaf(y[i].x, 1e-3f + 1e-6f * T, p[2 * i] * 0.5f); p[2 * i + 1] = fmaf(y[i].y, 1e-3f, p[2 * i + 1] * 0.5f); }
            }
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) acc += p[i];
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int R, int NC>
void run_icache(const float* S, float* out, long long* cyc)
{
    const int iters = 40, blocks = 1024;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((icache_probe<R, NC>), dim3(blocks), dim3(64), 0, 0, S, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    std::vector<long long> c(blocks);
    (void)hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += v;
    printf("icache: %d copies (~%d KB of straight-line code per loop trip): %.0f cycles per mat-vec\n", NC, NC * 13, s / blocks / iters / NC);
}

template <int R, int G, int WPE>
void run(const float* S, float* out, long long* cyc)
{
    const int iters = 200, blocks = 1024 * WPE;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<R, G, WPE>), dim3(blocks), dim3(64), 0, 0, S, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    std::vector<long long> c(blocks);
    (void)hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += v;
    printf("R=%d group=%d x16 waves/SIMD=%d: %.0f cycles per mat-vec per wave (%.2f per pk_fma, %d pk_fma), %.0f SIMD-cycles per 64-doc mat-vec\n", R, G, WPE,
           s / blocks / iters, s / blocks / iters / (R * R / 2), R * R / 2, s / blocks / iters / WPE);
}

int main()
{
    constexpr int R = 52;
    float* S; float* out; long long* cyc;
    (void)hipMalloc(&S, R * R * 4 + 256); (void)hipMalloc(&out, 8192 * 64 * 4); (void)hipMalloc(&cyc, 8192 * 8);
    std::vector<float> h(R * R + 64, 0.01f);
    (void)hipMemcpy(S, h.data(), R * R * 4 + 256, hipMemcpyHostToDevice);
    run<R, 1, 1>(S, out, cyc); run<R, 2, 1>(S, out, cyc); run<R, 3, 1>(S, out, cyc);
    run<R, 1, 2>(S, out, cyc); run<R, 2, 2>(S, out, cyc); run<R, 3, 2>(S, out, cyc);
    run_icache<R, 1>(S, out, cyc); run_icache<R, 2>(S, out, cyc); run_icache<R, 4>(S, out, cyc); run_icache<R, 6>(S, out, cyc);
    run_icache<R, 8>(S, out, cyc); run_icache<R, 12>(S, out, cyc);
    return 0;
}
