// probe: LDS-DMA staged row-per-lane load vs direct load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LPR>
__global__ __launch_bounds__(64) void k(const float* beta, const int* terms, int N, unsigned magic, float* out)
{
    constexpr int R = 4 * LPR;
    extern __shared__ __attribute__((aligned(16))) float stage[];
    int* t_l = (int*)(stage + 64 * R);
    const int lane = threadIdx.x;
    if (lane < N) t_l[lane] = terms[lane];
    __syncthreads();
    const int nch = N * LPR;
#pragma unroll
    for (int f0 = 0; f0 < 64 * LPR; f0 += 64) {
        const int f = f0 + lane;
        if (f < nch) {
            const int nn = (int)__umulhi((unsigned)f, magic);
            const int cc = f - nn * LPR;
            const float* src = beta + ((long)t_l[nn] * R + 4 * cc);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(stage + (size_t)f0 * 4), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane < N) for (int q = 0; q < R; ++q) out[lane * R + q] = stage[lane * R + q];
}
int main()
{
    constexpr int LPR = 13, R = 52, V = 1000, N = 50;
    std::vector<float> hb(V * R); for (int i = 0; i < V * R; ++i) hb[i] = (float)i;
    std::vector<int> ht(N); for (int i = 0; i < N; ++i) ht[i] = (i * 37 + 11) % V;
    float *db, *dout; int* dt;
    hipMalloc(&db, hb.size() * 4); hipMalloc(&dout, 64 * R * 4); hipMalloc(&dt, N * 4);
    hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dt, ht.data(), N * 4, hipMemcpyHostToDevice);
    unsigned magic = (unsigned)(0x100000000ull / LPR) + 1u;
    k<LPR><<<1, 64, (64 * R + 64) * 4>>>(db, dt, N, magic, dout);
    std::vector<float> ho(64 * R); hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int n = 0; n < N; ++n) for (int q = 0; q < R; ++q) if (ho[n * R + q] != hb[ht[n] * R + q]) { if (bad < 8) printf("n=%d q=%d got %.0f exp %.0f\n", n, q, ho[n * R + q], hb[ht[n] * R + q]); ++bad; }
    printf("bad=%d\n", bad);
}
