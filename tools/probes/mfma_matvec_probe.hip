// mfma_matvec_probe.hip -- row N2 of the review ("MFMA on the CTM Sigma^-1 (mu - lambda) contraction"), on the EXACT shape of the
// lane-per-document CTM kernel (tmvb_ctm_batch.h): Y = S P with S = invsigma, 52 x 52 fp32 (K = 50 padded to KP = 52), shared by all
// lanes; P, Y = one K-vector per LANE (lane = document, 64 documents per wave), in VGPRs, topic index = register index; one wave
// per SIMD (the kernel's occupancy: 512 registers per lane, 40 KB of LDS per wave).
//
// Three ways to run that 52 x 52 x 64 contraction, same box, same launch shape, correctness checked against the host:
//   V0  what the kernel does: S streamed through SGPRs (s_load_dwordx16, two groups in flight) as the scalar operand of
//       v_pk_fma_f32 -- 1352 packed fmas, no cross-lane traffic, no LDS, no extra registers.
//   V1  v_mfma_f32_4x4x1_16B_f32: 16 independent 4 x 4 outer products per instruction; block b = lanes 4b .. 4b+3 = documents
//       4b .. 4b+3, so P[k] of a document is the B operand AS IT LIES (the lane's own register k) and the four accumulators of row
//       group g ARE Y[4g .. 4g+3] of the lane's document: no shuffles, no padding (52 = 13 x 4) -- 676 MFMAs x 8 cycles = the same
//       5408 cycles as 1352 ideal packed fmas.  The catch is the A operand: lane l needs S[4g + (l & 3)][k], a DIFFERENT VGPR
//       value for every one of the 676 instructions (676 values cannot stay in registers), i.e. one LDS read (or one VALU move per
//       value from the SGPR stream) per 8-cycle MFMA.
//   V2  v_mfma_f32_32x32x2_f32: real tiles -- M = rows of S (52 -> two tiles of 32: 19 % of the rows are padding), N = 32 documents
//       (two halves), k pairs.  B operand (P[2q] for lanes 0..31 | P[2q+1] for lanes 32..63 of one document half) = ONE
//       v_permlane32_swap of the register pair (P[2q], P[2q+1]), which yields the operands of BOTH halves; the 32 x 32 results come
//       back to lane = document with one v_permlane32_swap per accumulator pair.  104 MFMAs x 64 cycles = 6656 cycles + 26 + 32
//       swaps; A operands from LDS (52 reads, 13 KB).
// V1 / V2 need 11 - 13 KB of LDS per wave for S in operand layout and (V2) 64 accumulator registers; the CTM kernel has neither to
// spare (its four waves per CU fill the 160 KB of LDS with lambda (fp64) and vsq; its register allocation is what five tail
// experiments died of).  The probe asks the prior question: would the contraction even be faster?
// Build: hipcc -O3 --offload-arch=gfx950 mfma_matvec_probe.hip -o mfma_matvec_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <utility>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int R = 52;

// ------------------------------------------------------------------------------------------------ V0 (as matvec_probe.hip, G = 2)
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
    static_assert(G == 2, "two groups in flight");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g.b[0]), "+s"(g.b[1]));
}
template <int G, int GI, int... K>
__device__ __forceinline__ void issue(SGroup<G>& g, const float* S, std::integer_sequence<int, K...>)
{
    constexpr int NB = (R * R + 15) / 16;
    ((g.b[K] = (GI * G + K < NB) ? sload16<(GI * G + K < NB ? (GI * G + K) * 64 : 0)>(S) : g.b[K]), ...);
}
template <int G, int GI>
__device__ __forceinline__ void consume(const SGroup<G>& g, const float (&p)[R], v2f (&y)[R / 2])
{
#pragma unroll
    for (int k = 0; k < G; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int f = (GI * G + k) * 16 + 2 * e;
            if (f < R * R) {
                const int j = f / R, i = f % R;                     // S stored column-major: element f = S[i][j] (S symmetric in the kernel)
                y[i / 2] = __builtin_elementwise_fma(v2f{g.b[k][2 * e], g.b[k][2 * e + 1]}, v2f{p[j], p[j]}, y[i / 2]);
            }
        }
}
template <int G, int GI, int NG>
__device__ __forceinline__ void pipeline(SGroup<G>& cur, SGroup<G>& nxt, const float* S, const float (&p)[R], v2f (&y)[R / 2])
{
    if constexpr (GI < NG) {
        if constexpr (GI + 1 < NG) issue<G, GI + 1>(nxt, S, std::make_integer_sequence<int, G>{});
        consume<G, GI>(cur, p, y);
        if constexpr (GI + 1 < NG) swait<G>(nxt);
        pipeline<G, GI + 1, NG>(nxt, cur, S, p, y);
    }
}
__device__ __forceinline__ void matvec_v0(const float* Scm, const float (&p)[R], float (&yo)[R])
{
    constexpr int G = 2, NB = (R * R + 15) / 16, NG = (NB + G - 1) / G;
    v2f y[R / 2];
#pragma unroll
    for (int i = 0; i < R / 2; ++i) y[i] = v2f{0.f, 0.f};
    SGroup<G> a, b;
    issue<G, 0>(a, Scm, std::make_integer_sequence<int, G>{});
    swait<G>(a);
    pipeline<G, 0, NG>(a, b, Scm, p, y);
#pragma unroll
    for (int i = 0; i < R / 2; ++i) { yo[2 * i] = y[i].x; yo[2 * i + 1] = y[i].y; }
}

// ------------------------------------------------------------------------------------------------ V1: 4x4x1, A from LDS
// LDS layout SA[g][r][k] (g < 13 row groups, r < 4, k < 52): lane l reads the four k-consecutive values S[4g + (l & 3)][k .. k+3]
// with one ds_read_b128 (4 distinct addresses per wave instruction: a broadcast)
template <bool SWAP_AB>
__device__ __forceinline__ void matvec_v1(const float* __restrict__ SA, const float (&p)[R], float (&yo)[R], const int lane)
{
    const v4f* base = (const v4f*)(SA + (lane & 3) * R);
    v4f acc[R / 4];                                    // thirteen independent accumulator chains (k outer, row group inner):
#pragma unroll                                         // a 4x4x1 MFMA that waits for its own previous result costs ~90 cycles
    for (int g = 0; g < R / 4; ++g) acc[g] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k4 = 0; k4 < R / 4; ++k4) {
        v4f a[R / 4];
#pragma unroll
        for (int g = 0; g < R / 4; ++g) a[g] = base[(g * 4 * R) / 4 + k4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int g = 0; g < R / 4; ++g) {
                if constexpr (SWAP_AB) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(p[4 * k4 + u], a[g][u], acc[g], 0, 0, 0);
                else acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g][u], p[4 * k4 + u], acc[g], 0, 0, 0);
            }
    }
#pragma unroll
    for (int g = 0; g < R / 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) yo[4 * g + i] = acc[g][i];
}

// ------------------------------------------------------------------------------------------------ V2: 32x32x2, A from LDS
// LDS layout SA2[t][q][l] (t < 2 row tiles, q < 26 k pairs, l < 64): S[32 t + (l & 31)][2 q + (l >> 5)], 0 for rows >= 52
__device__ __forceinline__ void swap32(float& x, float& y)       // x[32..63] <-> y[0..31]
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ void matvec_v2(const float* __restrict__ SA2, const float (&p)[R], float (&yo)[R], const int lane)
{
    v16f acc[2][2];                                               // [document half][row tile]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[h][t][i] = 0.f;
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
        float b0 = p[2 * q], b1 = p[2 * q + 1];
        swap32(b0, b1);            // b0 = [P[2q] docs 0..31 | P[2q+1] docs 0..31], b1 = the same for docs 32..63
        const float a0 = SA2[(0 * (R / 2) + q) * 64 + lane], a1 = SA2[(1 * (R / 2) + q) * 64 + lane];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    // D layout: lane l, register i: column n = l & 31, row m = (i & 3) + 8 (i >> 2) + 4 (l >> 5).  Swapping register i of the two
    // document halves leaves rows (i & 3) + 8 (i >> 2) of the lane's own document in the first and rows + 4 in the second.
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float x = acc[0][t][i], y = acc[1][t][i];
            swap32(x, y);
            const int m = 32 * t + (i & 3) + 8 * (i >> 2);
            if (m < R) yo[m] = x;
            if (m + 4 < R) yo[m + 4] = y;
        }
}

// ------------------------------------------------------------------------------------------------ harness
template <int V>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe(const float* Scm, const float* SAg, const float* SA2g, const float* p0, float* out, long long* cyc, int iters, int check)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    if (V == 1 || V == 3) for (int q = lane; q < R * R; q += 64) lds[q] = SAg[q];
    if (V == 2) for (int q = lane; q < 2 * (R / 2) * 64; q += 64) lds[q] = SA2g[q];
    __syncthreads();
    float p[R], y[R];
#pragma unroll
    for (int i = 0; i < R; ++i) p[i] = p0[(blockIdx.x * 64 + lane) % 4096 * R + i];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) matvec_v0(Scm, p, y);
        else if constexpr (V == 1) matvec_v1<false>(lds, p, y, lane);
        else if constexpr (V == 3) matvec_v1<true>(lds, p, y, lane);
        else matvec_v2(lds, p, y, lane);
        if (!check) {
#pragma unroll
            for (int i = 0; i < R; ++i) p[i] = fmaf(y[i], 1e-3f, p[i] * 0.5f);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (check) {
        if (blockIdx.x == 0)
#pragma unroll
            for (int i = 0; i < R; ++i) out[lane * R + i] = y[i];
    } else {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) acc += p[i];
        out[4096 * R + blockIdx.x * 64 + lane] = acc;
    }
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
double run(const char* name, const float* Scm, const float* SA, const float* SA2, const float* p0, float* out, long long* cyc,
           const std::vector<float>& S, const std::vector<float>& hp, int mfma, int lds_bytes)
{
    // correctness: one mat-vec of block 0 against the host
    hipLaunchKernelGGL((probe<V>), dim3(1), dim3(64), lds_bytes, 0, Scm, SA, SA2, p0, out, cyc, 1, 1);
    (void)hipDeviceSynchronize();
    std::vector<float> y(64 * R);
    (void)hipMemcpy(y.data(), out, y.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < R; ++i) {
            double ref = 0.0;
            for (int k = 0; k < R; ++k) ref += (double)S[i * R + k] * hp[l * R + k];
            worst = std::fmax(worst, std::fabs(y[l * R + i] - ref) / (std::fabs(ref) + 1e-3));
        }
    const int iters = 200, blocks = 1024;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<V>), dim3(blocks), dim3(64), lds_bytes, 0, Scm, SA, SA2, p0, out, cyc, iters, 0);
        (void)hipDeviceSynchronize();
    }
    std::vector<long long> c(blocks);
    (void)hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += v;
    const double per = s / blocks / iters;
    printf("%-58s %7.0f cycles per 52x52x64 mat-vec per wave (1 wave/SIMD, %d blocks)  max rel err vs host %.1e %s  [%d MFMA, %d B LDS]\n",
           name, per, blocks, worst, worst < 1e-4 ? "OK" : "WRONG", mfma, lds_bytes);
    return per;
}

int main()
{
    std::vector<float> S(R * R), Scm(R * R + 64, 0.f), SA(R * R), SA2(2 * (R / 2) * 64, 0.f), hp(4096 * R);
    unsigned s = 12345u;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : S) v = rnd();                                   // NOT symmetric: the check tells rows from columns
    for (auto& v : hp) v = 1.0f + rnd();
    for (int i = 0; i < R; ++i) for (int j = 0; j < R; ++j) Scm[j * R + i] = S[i * R + j];
    for (int g = 0; g < R / 4; ++g) for (int r = 0; r < 4; ++r) for (int k = 0; k < R; ++k) SA[(g * 4 + r) * R + k] = S[(4 * g + r) * R + k];
    for (int t = 0; t < 2; ++t) for (int q = 0; q < R / 2; ++q) for (int l = 0; l < 64; ++l) {
        const int m = 32 * t + (l & 31), k = 2 * q + (l >> 5);
        SA2[(t * (R / 2) + q) * 64 + l] = m < R ? S[m * R + k] : 0.f;
    }
    float *dS, *dSA, *dSA2, *dp, *out; long long* cyc;
    (void)hipMalloc(&dS, Scm.size() * 4); (void)hipMalloc(&dSA, SA.size() * 4); (void)hipMalloc(&dSA2, SA2.size() * 4);
    (void)hipMalloc(&dp, hp.size() * 4); (void)hipMalloc(&out, (4096 * R + 1024 * 64) * 4); (void)hipMalloc(&cyc, 1024 * 8);
    (void)hipMemcpy(dS, Scm.data(), Scm.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dSA, SA.data(), SA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dSA2, SA2.data(), SA2.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs; Y = S P, S 52 x 52 fp32 shared, P / Y one 52-vector per lane (lane = document), 2 x 52 x 52 x 64 = 346 112 flop per mat-vec\n",
           prop.gcnArchName, prop.multiProcessorCount);
    const double v0 = run<0>("V0 v_pk_fma_f32, S through SGPRs (the CTM kernel's form)", dS, dSA, dSA2, dp, out, cyc, S, hp, 0, 0);
    const double v1 = run<1>("V1 v_mfma_f32_4x4x1_16B_f32, A = S from LDS (b128), B = P as is", dS, dSA, dSA2, dp, out, cyc, S, hp, 676, R * R * 4);
    const double v3 = run<3>("V1' the same with the operands exchanged (A = P, B = S)", dS, dSA, dSA2, dp, out, cyc, S, hp, 676, R * R * 4);
    const double v2 = run<2>("V2 v_mfma_f32_32x32x2_f32, A from LDS, 26 + 32 permlane32 swaps", dS, dSA, dSA2, dp, out, cyc, S, hp, 104, 2 * (R / 2) * 64 * 4);
    printf("ideal: 1352 packed fmas x 4 cycles = 5408; 676 MFMA 4x4x1 x 8 = 5408; 104 MFMA 32x32x2 x 64 = 6656 (rows padded 52 -> 64)\n");
    printf("ratio to V0: V1 %.2f  V1' %.2f  V2 %.2f  (of V1 / V1' only the one marked OK computes S P; the other computes S' P)\n", v1 / v0, v3 / v0, v2 / v0);
    return 0;
}
