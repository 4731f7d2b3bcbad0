// probe: semantics of v_permlane32_swap / v_permlane16_swap on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
    int lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main()
{
    int* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[4] = {"swap32.0", "swap32.1", "swap16.0", "swap16.1"};
    for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; ++i) printf(" %d", h[64 * r + i]); printf("\n"); }
    return 0;
}
