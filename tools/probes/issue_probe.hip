// issue_probe.hip -- per-instruction issue cost on gfx950 for the instruction classes the CTM Newton solve is built
// from (round-2 design input): plain / packed fp32 FMA, v_readlane + dependent SGPR use, permlane32 swap, DPP move,
// ds_bpermute, fp64 FMA, f32 MFMA 32x32x2 / 16x16x4 (dependent and 4 independent accumulators).
// Build: hipcc -O3 --offload-arch=gfx950 issue_probe.hip -o issue_probe ; run: ./issue_probe
// Prints cycles per instruction per wave (s_memtime ticks = shader cycles) at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(64) void probe(float* out, long long* cyc, int iters)
{
    const int lane = threadIdx.x;
    float a = 1.0f + lane * 1e-3f, b = 0.999f, c = 0.5f, d = 0.25f;
    v2f pa = {a, b}, pb = {b, a}, pc = {c, d}, pd = {d, c};
    double da = a, db = b;
    f32x16 m0 = {0}, m1 = {0}, m2 = {0}, m3 = {0};
    f32x4 q0 = {0}, q1 = {0}, q2 = {0}, q3 = {0};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 64 independent-ish v_fma_f32 (4 chains)
            REP8(REP8(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c) : "v"(b));
                      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(d) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(pa.x) : "v"(b));))
        } else if (MODE == 1) {   // v_pk_fma_f32, 4 chains
            REP8(REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pa) : "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pc) : "v"(pb));
                      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pd) : "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pb) : "v"(pa));))
        } else if (MODE == 2) {   // v_readlane only (4 per group)
            int s0, s1, s2, s3;
            REP8(REP8(asm volatile("v_readlane_b32 %0, %4, 3\n\tv_readlane_b32 %1, %5, 7\n\tv_readlane_b32 %2, %6, 11\n\tv_readlane_b32 %3, %7, 13"
                                   : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a), "v"(b), "v"(c), "v"(d));))
            a += __builtin_bit_cast(float, s0 ^ s1 ^ s2 ^ s3) * 1e-30f;
        } else if (MODE == 3) {   // 2 readlanes + 1 pk_fma with the SGPR pair (the current Gauss-Jordan inner pattern), blocks of 8
            REP8(REP8(
                {
                    v2f s;
                    asm volatile("v_readlane_b32 %0, %2, 5\n\tv_readlane_b32 %1, %3, 9" : "=s"(s.x), "=s"(s.y) : "v"(a), "v"(b));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa) : "v"(pb), "s"(s));
                }
                {
                    v2f s;
                    asm volatile("v_readlane_b32 %0, %2, 6\n\tv_readlane_b32 %1, %3, 10" : "=s"(s.x), "=s"(s.y) : "v"(c), "v"(d));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pc) : "v"(pd), "s"(s));
                }))
        } else if (MODE == 4) {   // permlane32 swap pairs
            REP8(REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));))
        } else if (MODE == 5) {   // DPP mov row_mirror
            REP8(REP8(asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %3 row_mirror row_mask:0xf bank_mask:0xf"
                                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d));))
        } else if (MODE == 6) {   // ds_bpermute (2 per group, waited per 16)
            int ia = __builtin_bit_cast(int, a), ib = __builtin_bit_cast(int, b), addr = (lane ^ 5) * 4;
            REP8(REP8(asm volatile("ds_bpermute_b32 %0, %2, %0\n\tds_bpermute_b32 %1, %2, %1" : "+v"(ia), "+v"(ib) : "v"(addr));) asm volatile("s_waitcnt lgkmcnt(0)");)
            a = __builtin_bit_cast(float, ia); b = __builtin_bit_cast(float, ib);
        } else if (MODE == 7) {   // v_fma_f64, 2 chains
            REP8(REP8(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(da) : "v"(db)); asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(db) : "v"(da));))
        } else if (MODE == 8) {   // MFMA 32x32x2 f32, one accumulator (dependent)
            REP8(REP8(m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, m0, 0, 0, 0); m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, m0, 0, 0, 0);))
        } else if (MODE == 9) {   // MFMA 32x32x2 f32, 3 accumulators round robin
            REP8(REP8(m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, m1, 0, 0, 0);
                      m2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d, m2, 0, 0, 0);))
        } else if (MODE == 10) {  // MFMA 16x16x4 f32, 4 accumulators
            REP8(REP8(q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q0, 0, 0, 0); q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, d, q1, 0, 0, 0);
                      q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, d, q2, 0, 0, 0); q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, b, q3, 0, 0, 0);))
        } else if (MODE == 11) {  // MFMA 32x32x2 (3 accumulators) interleaved with 8 plain VALU per MFMA: does VALU hide under the MFMA pipe?
            REP8(REP8(m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, m0, 0, 0, 0);
                      REP8(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(pa.x) : "v"(pb.x));)
                      m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, m1, 0, 0, 0);
                      REP8(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(pa.y) : "v"(pb.x));)
                      m2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d, m2, 0, 0, 0);
                      REP8(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(pc.x) : "v"(pb.x));)))
        } else if (MODE == 12) {  // v_exp_f32 / v_rcp_f32 / v_log_f32 transcendental rate
            REP8(REP8(asm volatile("v_exp_f32 %0, %0" : "+v"(a)); asm volatile("v_rcp_f32 %0, %0" : "+v"(b)); asm volatile("v_log_f32 %0, %0" : "+v"(c));
                      asm volatile("v_exp_f32 %0, %0" : "+v"(d));))
        } else if (MODE == 13) {  // v_cndmask with SGPR mask
            unsigned long long msk = 0x00ff00ff00ff00ffull;
            REP8(REP8(asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(msk)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "s"(msk));))
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = a + b + c + d + pa.x + pa.y + pb.x + pb.y + pc.x + pc.y + pd.x + pd.y + (float)da + (float)db;
    for (int i = 0; i < 16; ++i) r += m0[i] + m1[i] + m2[i] + m3[i];
    for (int i = 0; i < 4; ++i) r += q0[i] + q1[i] + q2[i] + q3[i];
    out[blockIdx.x * 64 + lane] = r;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Mode { int id; const char* name; int per_iter; };

template <int MODE>
static void run(const Mode& m, float* d_out, long long* d_cyc, int ncu)
{
    const int iters = 50;
    for (int wps : {1, 2, 4}) {
        const int blocks = ncu * 4 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 2);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> c(blocks);
        hipMemcpy(c.data(), d_cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : c) avg += (double)v; avg /= blocks;
        const double n = (double)iters * m.per_iter;
        // wall-clock based: ns per instruction per SIMD (all waves of the SIMD together issue wps * n instructions)
        printf("%-46s waves/SIMD %d: %7.2f clk/instr/wave (s_memtime), %6.2f ns/instr/SIMD wall\n", m.name, wps, avg / n, ms * 1e6 / (n * wps));
    }
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, ncu, prop.clockRate);
    float* d_out; long long* d_cyc;
    hipMalloc(&d_out, (size_t)ncu * 16 * 64 * sizeof(float));
    hipMalloc(&d_cyc, (size_t)ncu * 16 * sizeof(long long));
    run<0>({0, "v_fma_f32 (4 chains)", 256}, d_out, d_cyc, ncu);
    run<1>({1, "v_pk_fma_f32 (4 chains)", 256}, d_out, d_cyc, ncu);
    run<2>({2, "v_readlane_b32", 256}, d_out, d_cyc, ncu);
    run<3>({3, "2 readlane + pk_fma(SGPR pair) [per group of 3]", 128}, d_out, d_cyc, ncu);
    run<4>({4, "v_permlane32_swap_b32", 128}, d_out, d_cyc, ncu);
    run<5>({5, "v_mov_b32_dpp row_mirror", 128}, d_out, d_cyc, ncu);
    run<6>({6, "ds_bpermute_b32", 128}, d_out, d_cyc, ncu);
    run<7>({7, "v_fma_f64 (2 chains)", 128}, d_out, d_cyc, ncu);
    run<8>({8, "v_mfma_f32_32x32x2_f32 dependent", 128}, d_out, d_cyc, ncu);
    run<9>({9, "v_mfma_f32_32x32x2_f32 3 accumulators", 192}, d_out, d_cyc, ncu);
    run<10>({10, "v_mfma_f32_16x16x4_f32 4 accumulators", 256}, d_out, d_cyc, ncu);
    run<11>({11, "mfma 32x32x2 + 8 v_fma each [per mfma]", 192}, d_out, d_cyc, ncu);
    run<12>({12, "v_exp/v_rcp/v_log/v_exp f32", 256}, d_out, d_cyc, ncu);
    run<13>({13, "v_cndmask_b32 SGPR mask", 128}, d_out, d_cyc, ncu);
    return 0;
}
