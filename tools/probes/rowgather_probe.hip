// rowgather_probe.hip -- the token phase of the lane-per-document CTM kernel in isolation: every lane of a wave needs "its" row of
// a [V][52]-float table (208 B, rows at stride 208 B, random row ids), one row per lane per token step, then
//   s = row . e ; w = 1 / s ; acc += w * row          (2 x 26 packed FMAs per step)
// One wave per SIMD (40 KB of LDS per wave, as in the kernel), 1024 waves = every SIMD of the chip busy with the same loop.
//   MODE 0  each lane loads its own row: 13 x 16 B per lane, 64 distinct rows per instruction, CH steps issued together (kernel today)
//   MODE 1  as 0 with the chunks at +0 / +128 / +192 B first (every line of the row touched before the second chunk of any line)
//   MODE 2  cooperative: chunk q = 64 i + lane of the step's 64 x 13 chunks -> row q / 13, chunk q % 13 (13 consecutive lanes read one
//           row: coalesced), S steps in flight in VGPRs, transposed through a 13 KB LDS buffer (ds_write_b128 linear, ds_read_b128 of
//           the lane's own row at stride 208 B)
//   MODE 3  as 2 with global_load_lds_dwordx4 (no VGPR round trip), NB LDS buffers of 13 KB
// Prints shader cycles per token step per wave and the implied gather rate.
// Build: hipcc -O3 --offload-arch=gfx950 rowgather_probe.hip -o rowgather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int R = 52, LPR = 13;

template <int I> struct ic { static constexpr int value = I; };
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(ic<I>{}); static_for<N, I + 1>(f); }
}

__device__ __forceinline__ void step_compute(const v4f (&row)[LPR], const v2f (&e2)[R / 2], v2f (&acc)[R / 2])
{
    v2f s0 = v2f{0.f, 0.f}, s1 = v2f{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < LPR; ++q) {
        s0 = __builtin_elementwise_fma(v2f{row[q].x, row[q].y}, e2[2 * q], s0);
        s1 = __builtin_elementwise_fma(v2f{row[q].z, row[q].w}, e2[2 * q + 1], s1);
    }
    const v2f ss = s0 + s1;
    const float w = 1.0f / (ss.x + ss.y);
    const v2f w2 = v2f{w, w};
#pragma unroll
    for (int q = 0; q < LPR; ++q) {
        acc[2 * q] = __builtin_elementwise_fma(w2, v2f{row[q].x, row[q].y}, acc[2 * q]);
        acc[2 * q + 1] = __builtin_elementwise_fma(w2, v2f{row[q].z, row[q].w}, acc[2 * q + 1]);
    }
}

__device__ __forceinline__ void exchange_ids(int* tl, int lane, const int (&rmap)[LPR], int t, int (&tt)[LPR])
{
    __builtin_amdgcn_wave_barrier();
    tl[lane] = t;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < LPR; ++i) tt[i] = tl[rmap[i]];
}
__device__ __forceinline__ void issue_coop(const float* __restrict__ beta, int* tl, int lane, const int (&rmap)[LPR], const int (&coff)[LPR], int t, v4f (&b)[LPR])
{
    int tt[LPR];
    exchange_ids(tl, lane, rmap, t, tt);
#pragma unroll
    for (int i = 0; i < LPR; ++i) b[i] = *(const v4f*)(beta + (size_t)tt[i] * R + coff[i]);
}

template <int MODE, int CH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(const float* __restrict__ beta, const int* __restrict__ ids,
                                                                                        int steps, float* __restrict__ out, long long* __restrict__ cyc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int* my = ids + (size_t)blockIdx.x * steps * 64;
    v2f e2[R / 2], acc[R / 2];
#pragma unroll
    for (int i = 0; i < R / 2; ++i) { e2[i] = v2f{1.0f + 0.01f * i, 1.0f - 0.01f * i}; acc[i] = v2f{0.f, 0.f}; }
    const long long t0 = __builtin_readcyclecounter();
    // every lane fetches the ids of "its" rows two rounds (2 CH steps) ahead: VMEM returns in order, so waiting for a load
    // that is younger than the rows in flight would drain them
    int tq[CH], tn[CH], t2[CH];
    auto load_ids = [&](int n0, int (&t)[CH]) {
#pragma unroll
        for (int u = 0; u < CH; ++u) t[u] = my[((n0 + u < steps) ? n0 + u : 0) * 64 + lane];
    };
    if constexpr (MODE == 0 || MODE == 1) {
        load_ids(0, tq);
        for (int n0 = 0; n0 < steps; n0 += CH) {
            v4f rows[CH][LPR];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const v4f* row = (const v4f*)(beta + (size_t)tq[u] * R);
                if constexpr (MODE == 1) {
                    rows[u][0] = row[0];
                    __builtin_amdgcn_sched_barrier(0);
                    rows[u][8] = row[8];
                    __builtin_amdgcn_sched_barrier(0);
                    rows[u][12] = row[12];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 1; q < 12; ++q) if (q != 8) rows[u][q] = row[q];
                } else {
#pragma unroll
                    for (int q = 0; q < LPR; ++q) rows[u][q] = row[q];
                }
            }
            load_ids(n0 + CH, tn);
#pragma unroll
            for (int u = 0; u < CH; ++u) step_compute(rows[u], e2, acc);
#pragma unroll
            for (int u = 0; u < CH; ++u) tq[u] = tn[u];
        }
    } else {
        // lane's role in cooperative instruction i: chunk q = 64 i + lane of the step's 64 x 13 chunks
        int rmap[LPR], coff[LPR];
#pragma unroll
        for (int i = 0; i < LPR; ++i) { const int q = 64 * i + lane; rmap[i] = q / LPR; coff[i] = (q % LPR) * 4; }
        int* tl = (int*)(lds + CH * R * 64);            // [64] id exchange (after the row buffers)
        if constexpr (MODE == 2) {
            v4f* l4 = (v4f*)lds;
            v4f buf[CH][LPR];
            load_ids(0, tq);
            static_for<CH>([&](auto tag) { constexpr int u = decltype(tag)::value; issue_coop(beta, tl, lane, rmap, coff, tq[u], buf[u]); });
            load_ids(CH, tn);
            for (int n0 = 0; n0 < steps; n0 += CH) {
                load_ids(n0 + 2 * CH, t2);
                static_for<CH>([&](auto tag) {
                    constexpr int u = decltype(tag)::value;
#pragma unroll
                    for (int i = 0; i < LPR; ++i) l4[64 * i + lane] = buf[u][i];
                    __builtin_amdgcn_wave_barrier();
                    v4f row[LPR];
#pragma unroll
                    for (int q = 0; q < LPR; ++q) row[q] = l4[lane * LPR + q];
                    issue_coop(beta, tl, lane, rmap, coff, tn[u], buf[u]);
                    step_compute(row, e2, acc);
                });
#pragma unroll
                for (int u = 0; u < CH; ++u) { tn[u] = t2[u]; }
            }
        } else {
            constexpr int NB = CH;                     // LDS buffers of 64 rows; NB - 1 steps in flight behind the one consumed
            auto issue = [&](int b, int t) __attribute__((always_inline)) {
                int tt[LPR];
                exchange_ids(tl, lane, rmap, t, tt);
#pragma unroll
                for (int i = 0; i < LPR; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(beta + (size_t)tt[i] * R + coff[i]),
                                                     (__attribute__((address_space(3))) void*)(lds + b * (R * 64) + i * 256), 16, 0, 0);
            };
            // ids through a rolling queue: step n's id is element n % (2 NB) of {tq, tn} loaded two rounds ahead
            load_ids(0, tq);
            load_ids(NB, tn);
            static_for<NB - 1>([&](auto tag) { constexpr int u = decltype(tag)::value; issue(u, tq[u]); });
            for (int n0 = 0; n0 < steps; n0 += NB) {
                load_ids(n0 + 2 * NB, t2);
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    // step n0 + u + NB - 1 goes into the buffer consumed in the previous slot
                    const int tnew = (u + NB - 1 < NB) ? tq[u + NB - 1] : tn[u - 1];
                    issue((u + NB - 1) % NB, tnew);
                    // DMA groups complete in order: everything but the newest NB - 1 steps' 13 loads each must be done.  (The id
                    // loads of t2 were issued before this round's DMAs; they are older and add nothing to the count.)
                    if constexpr (NB == 2) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
                    else if constexpr (NB == 3) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(39)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    v4f row[LPR];
                    const v4f* l4 = (const v4f*)(lds + u * (R * 64));
#pragma unroll
                    for (int q = 0; q < LPR; ++q) row[q] = l4[lane * LPR + q];
                    step_compute(row, e2, acc);
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) { tq[u] = tn[u]; tn[u] = t2[u]; }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) s += acc[i].x + acc[i].y;
    out[(size_t)blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int CH>
static void run(const char* name, const float* d_beta, const int* d_ids, int steps, int waves, float* d_out, long long* d_cyc, size_t lds_bytes)
{
    hipFuncSetAttribute((const void*)probe<MODE, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<MODE, CH>), dim3(waves), dim3(64), lds_bytes, 0, d_beta, d_ids, steps, d_out, d_cyc);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<long long> c(waves);
    hipMemcpy(c.data(), d_cyc, waves * sizeof(long long), hipMemcpyDeviceToHost);
    double tot = 0;
    for (auto v : c) tot += (double)v;
    const double per_step = tot / waves / steps;
    const double bytes = (double)waves * steps * 64 * R * 4;
    printf("%-44s %8.0f cycles / token step / wave   kernel %.3f ms   %.2f TB/s of rows\n", name, per_step, ms, bytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv)
{
    const int V = 25319, steps = 1200, waves = argc > 1 ? atoi(argv[1]) : 1024;
    std::vector<float> beta((size_t)V * R);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(0.01f, 1.0f);
    for (auto& x : beta) x = u(rng);
    std::vector<int> ids((size_t)waves * steps * 64);
    std::uniform_int_distribution<int> ui(0, V - 1);
    for (auto& x : ids) x = ui(rng);
    float *d_beta, *d_out; int* d_ids; long long* d_cyc;
    hipMalloc(&d_beta, beta.size() * 4 + 256); hipMalloc(&d_ids, ids.size() * 4); hipMalloc(&d_out, (size_t)waves * 64 * 4); hipMalloc(&d_cyc, waves * 8);
    hipMemcpy(d_beta, beta.data(), beta.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = 40 * 1024;
    printf("%d waves (one per SIMD on %d CUs), %d token steps, table %d x %d floats\n", waves, waves / 4, steps, V, R);
    run<0, 4>("0 per-lane rows, 4 steps issued together", d_beta, d_ids, steps, waves, d_out, d_cyc, lds);
    run<0, 2>("0 per-lane rows, 2 steps issued together", d_beta, d_ids, steps, waves, d_out, d_cyc, lds);
    run<1, 4>("1 per-lane rows, lines first, 4 steps", d_beta, d_ids, steps, waves, d_out, d_cyc, lds);
    run<2, 2>("2 cooperative via VGPR + LDS, 2 in flight", d_beta, d_ids, steps, waves, d_out, d_cyc, lds);
    run<2, 3>("2 cooperative via VGPR + LDS, 3 in flight", d_beta, d_ids, steps, waves, d_out, d_cyc, lds);
    run<2, 4>("2 cooperative via VGPR + LDS, 4 in flight", d_beta, d_ids, steps, waves, d_out, d_cyc, lds);
    run<2, 5>("2 cooperative via VGPR + LDS, 5 in flight", d_beta, d_ids, steps, waves, d_out, d_cyc, lds);
    run<2, 6>("2 cooperative via VGPR + LDS, 6 in flight", d_beta, d_ids, steps, waves, d_out, d_cyc, lds);
    run<3, 2>("3 cooperative LDS-DMA, 2 buffers", d_beta, d_ids, steps, waves, d_out, d_cyc, lds);
    run<3, 3>("3 cooperative LDS-DMA, 3 buffers", d_beta, d_ids, steps, waves, d_out, d_cyc, lds);
    return 0;
}
