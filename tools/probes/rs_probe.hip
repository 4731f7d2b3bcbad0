// probe: lane_reduce_scatter correctness vs host replay
#include "../../topicmodelsvb.jl_amd/csrc/tmvb_lda.hip"
template <int R>
__global__ void k(float* out, const int* tol)
{
    int lane = threadIdx.x;
    float p[R];
    for (int q = 0; q < R; ++q) p[q] = (float)((q + 1) * 1000 + lane);
    float g = lane_reduce_scatter<R>(p, lane);
    out[lane] = g;
}
template <int R> void run()
{
    std::vector<int> tol, lot; tmvb_reg_lane_maps(R, tol, lot);
    float* d; hipMalloc(&d, 64 * 4); int* dt; hipMalloc(&dt, 64 * 4);
    k<R><<<1, 64>>>(d, dt);
    float h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (tol[l] < 0) continue;
        float expect = 64.0f * (tol[l] + 1) * 1000 + 2016.0f;
        if (h[l] != expect) { if (bad < 5) printf("R=%d lane %d topic %d got %.0f expect %.0f\n", R, l, tol[l], h[l], expect); ++bad; }
    }
    printf("R=%d bad=%d; lot:", R, bad); for (int q = 0; q < R; ++q) printf(" %d", lot[q]); printf("\n");
}
int main() { run<4>(); run<12>(); run<20>(); run<52>(); return 0; }
