#include "../../topicmodelsvb.jl_amd/csrc/tmvb_internal.h"
#include <cstdio>
XX
{
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    a = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ void swap_add16(float& a, float b)
{
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    a = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__global__ void k(float* out)
{
    int lane = threadIdx.x;
    float a = 1000 + lane, b = 2000 + lane;
    float x = a; swap_add32(x, b); out[lane] = x;
    float y = a; swap_add16(y, b); out[64 + lane] = y;
    float z = a; z += dpp_f<0x140>(z); out[128 + lane] = z;
    float u = a; u += dpp_f<0x141>(u); out[192 + lane] = u;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[4] = {"swap_add32", "swap_add16", "row_mirror", "half_mirror"};
    for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; ++i) printf(" %.0f", h[64 * r + i]); printf("\n"); }
}
