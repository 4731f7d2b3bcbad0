// Cost of a hand-rolled device-wide barrier on gfx950 (8 XCDs, one L2 each): a persistent kernel of G workgroups runs NB barriers in a row;
// time per barrier = kernel time / NB.  Forms:
//   flat   one counter, every workgroup's thread 0 adds 1 at agent scope; the last arriver resets it and bumps a generation word the others poll
//   tree   8 counters (workgroup index mod 8 ~ the XCD a workgroup lands on), the last arriver of each adds 1 to a top counter of 8
// Every barrier also publishes one word per workgroup and reads a neighbour's after it (checks that release / acquire works across XCDs).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/gridbar_probe tools/probes/gridbar_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr unsigned SPIN_LIMIT = 1u << 18;

__device__ __forceinline__ bool spin_until(unsigned* word, unsigned target)
{
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {   // (relaxed: the fence behind the barrier acquires once)
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) return false;
    }
    return true;
}

// bar[0] = count, bar[1] = generation, bar[2] = error, bar[16 + 16 * x] = sub-counter x (own cache lines)
// FENCE: 2 = every thread fences on both sides (agent scope: an L2 write-back before, an invalidate after -- the L2s of the 8 XCDs are not coherent),
//        1 = thread 0 alone does (one write-back and one invalidate per workgroup; the workgroup barrier orders the other waves), 0 = none (the
//        atomics' own cost; NOT correct for data)
template <bool TREE, int FENCE>
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned G, unsigned target)
{
    __shared__ unsigned ok_s;
    if (FENCE == 2) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE == 1) __threadfence();
        bool last = false;
        if (TREE) {
            const unsigned x = blockIdx.x & 7u;
            const unsigned members = (G >> 3) + ((G & 7u) > x ? 1u : 0u);
            unsigned* sub = bar + 16 + 16 * x;
            if (__hip_atomic_fetch_add(sub, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned groups = G < 8u ? G : 8u;
                last = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
            }
        } else {
            last = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == G - 1;
        }
        if (last) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&bar[1], target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (!spin_until(&bar[1], target)) {
            __hip_atomic_store(&bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ok_s = __hip_atomic_load(&bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
        if (FENCE == 1) __threadfence();
    }
    __syncthreads();
    if (FENCE == 2) __threadfence();
    return ok_s != 0u;                                  // false: some workgroup gave up waiting -- the caller leaves
}

// two barriers per step with distinct generations (the form the library would use)
template <bool TREE, int FENCE>
__global__ __launch_bounds__(1024) void probe2_kernel(unsigned* bar, unsigned* data, int nb, unsigned* bad)
{
    const unsigned G = gridDim.x;
    __shared__ unsigned gen0_s;
    if (threadIdx.x == 0) gen0_s = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned gen0 = gen0_s;
    unsigned wrong = 0;
    for (int b = 0; b < nb; ++b) {
        const unsigned stamp = gen0 + 2u * (unsigned)b + 1u;
        if (threadIdx.x == 0) data[blockIdx.x * 16] = stamp;
        if (!grid_barrier<TREE, FENCE>(bar, G, stamp)) break;
        if (threadIdx.x == 64) {
            const unsigned other = (blockIdx.x + 1u + (unsigned)b * 7u) % G;
            if (data[other * 16] != stamp) ++wrong;
        }
        if (!grid_barrier<TREE, FENCE>(bar, G, stamp + 1u)) break;
    }
    if (threadIdx.x == 64 && wrong) atomicAdd(bad, wrong);
}

int main()
{
    unsigned *bar, *data, *bad;
    CHECK(hipMalloc(&bar, 4096)); CHECK(hipMalloc(&data, 4096 * 64)); CHECK(hipMalloc(&bad, 4));
    CHECK(hipMemset(bar, 0, 4096)); CHECK(hipMemset(data, 0, 4096 * 64)); CHECK(hipMemset(bad, 0, 4));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int NB = 200;
    printf("# G  threads  form  fences  us_per_barrier  (kernel of %d steps, two barriers each)  wrong_reads  timeout\n", NB);
    for (int fence = 0; fence < 3; ++fence)
      for (int tree = 0; tree < 2; ++tree)
        for (unsigned T : {256u, 1024u})
          for (unsigned G : {8u, 32u, 64u, 128u, 256u, 512u}) {
            if (T == 1024u && G > 256u) continue;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, st));
#define LAUNCH(TR, FE) hipLaunchKernelGGL((probe2_kernel<TR, FE>), dim3(G), dim3(T), 0, st, bar, data, NB, bad)
                if (tree) { if (fence == 0) LAUNCH(true, 0); else if (fence == 1) LAUNCH(true, 1); else LAUNCH(true, 2); }
                else      { if (fence == 0) LAUNCH(false, 0); else if (fence == 1) LAUNCH(false, 1); else LAUNCH(false, 2); }
                CHECK(hipEventRecord(e1, st));
                CHECK(hipStreamSynchronize(st));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                unsigned hb[3], hbad; CHECK(hipMemcpy(hb, bar, 12, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
                if (rep == 2) printf("%5u  %4u  %s  %d  %8.2f  %u  %u\n", G, T, tree ? "tree" : "flat", fence, 1e3 * ms / (2.0 * NB), hbad, hb[2]);
                if (hb[2]) { CHECK(hipMemset(bar, 0, 4096)); }
                CHECK(hipMemset(bad, 0, 4));
            }
          }
    return 0;
}
