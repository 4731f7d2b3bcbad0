// tmvb_gridtile.h -- the 2-D ("grid") register tile of the per-document sweeps (LDA: tmvb_lda.hip).
//
// A sweep of src/LDA.jl:172-174 is two matrix-vector products with the document's N x K tile B = beta[:, terms_d]':
//     s = B e      (update_phi!'s normaliser, one value per token)        w = c ./ s
//     g = B' w     (update_gamma!, one value per topic)
// With lane = token (tmvb_regtile.h) the first product is free of cross-lane traffic and the second is a 64-lane
// reduce-scatter of all KP topic sums: ~2 KP cross-lane instructions plus their hazard nops per sweep, 45 % of the issue slots
// of a kernel that is VALU-issue bound.  Here the 64 lanes form a 16 x 4 grid instead -- lane l = (a, b), a = l >> 2 one of
// 16 TOKEN groups, b = l & 3 one of 4 TOPIC classes -- and a lane holds the sub-tile {tokens n = 16 s + a} x {topics 4 j + b}:
// 2 NP token slots (as NP pairs, so that both products run on v_pk_fma_f32) times LPR = KP / 4 topics.  Per sweep:
//     phase 1   NP * LPR packed fmas (e_j broadcast inside the pair by op_sel), then an all-reduce over the 4 lanes of a quad
//               (2 DPP adds per token slot): every lane of the quad holds s_n of its tokens, w_n = c_n * rcp(s_n) in place;
//     phase 2   NP * LPR packed fmas with the pair (w_n, w_n') as is, LPR adds to fold the pairs, then a reduce-scatter over
//               the 16 lanes of a topic class: LPR + 1 values (the topics and sum_n w_n) -> one per lane in 4 stages
//               (v_permlane32_swap, v_permlane16_swap, two bank-masked DPP stages), ~2 instructions per output value;
//     tail      one topic per lane: gamma, digamma, Elogtheta, the exit test; e = exp(Elogtheta) goes back to the quads
//               through 256 bytes of LDS (one ds_write_b32, four broadcast ds_read_b128).
// Instruction count per sweep ~ 26 NP + 9 NP + ~100 (K = 50) against ~300 / ~480 for the lane = token kernel with one / two
// 64-token tiles, with the same or fewer registers: NP <= 6 (192 tokens) fits 256 VGPRs at KP = 52, and a document costs what
// its own length rounded up to 32 tokens costs, not what the next multiple of 64 costs.
#pragma once
#include "tmvb_common_kernels.h"

typedef float gv2f __attribute__((ext_vector_type(2)));

// ---- the reduce-scatter over the 16 token groups of a topic class --------------------------------------------------------
// M values per lane (index i = 0 .. M-1) are summed over the 16 lanes that share b; afterwards result slot r of lane (a, b)
// holds the total of index kGridMap<M>.idx[r][a] (M <= 16: one slot; M <= 32: two).  Stage order: lane bit 5 (a bit 3), bit 4,
// bit 3, bit 2.  At a stage with m values left, h = ceil(m / 2): the lane keeps value i (its bit clear) or value i + h (bit
// set) for i < m - h and adds the partner's copy of the same value; for odd m the middle value h - 1 is added on both sides
// (both lanes then own that index: the map marks one of them as the primary owner).
template <int M>
struct GridMap {
    static constexpr int NS = (M + 15) / 16;
    int idx[NS][16];          // index owned by result slot r of token group a
    bool primary[NS][16];     // first owner of that index (duplicates hold the same total)
    int a_of[M], r_of[M];     // primary owner of index i
    constexpr GridMap() : idx{}, primary{}, a_of{}, r_of{}
    {
        int cur[M][16] = {}, nxt[M][16] = {};
        for (int i = 0; i < M; ++i) for (int a = 0; a < 16; ++a) cur[i][a] = i;
        int m = M;
        const int Ds[4] = {8, 4, 2, 1};
        for (int st = 0; st < 4; ++st) {
            const int h = (m + 1) / 2;
            for (int i = 0; i < h; ++i)
                for (int a = 0; a < 16; ++a) nxt[i][a] = (i + h < m && (a & Ds[st])) ? cur[i + h][a] : cur[i][a];
            for (int i = 0; i < h; ++i) for (int a = 0; a < 16; ++a) cur[i][a] = nxt[i][a];
            m = h;
        }
        for (int i = 0; i < M; ++i) { a_of[i] = -1; r_of[i] = -1; }
        for (int r = 0; r < NS; ++r)
            for (int a = 0; a < 16; ++a) {
                const int i = (r < m) ? cur[r][a] : -1;
                idx[r][a] = i;
                primary[r][a] = false;
                if (i >= 0 && a_of[i] < 0) { a_of[i] = a; r_of[i] = r; primary[r][a] = true; }
            }
    }
};
template <int M> constexpr GridMap<M> kGridMap{};

// one swap stage (lane bit 5 or 4): pairs (i, i + h) exchange halves / rows and add; up to four swaps share ONE pair of
// hazard wait states (VALU write -> v_permlane*_swap read needs 2 states, and so does the swap's result -> VALU read;
// hipcc pads nothing inside an asm statement).  A value without a partner value (odd count) is swapped with a copy of itself.
template <bool B32, int NPAIR>
__device__ __forceinline__ void grid_swap_block(float* x, float* y)          // x[q] <-> y[q], q < NPAIR; x[q] += y[q]
{
    static_assert(NPAIR >= 1 && NPAIR <= 4, "grid_swap_block");
#define GRID_SW(OP)                                                                                                                  \
    if constexpr (NPAIR == 1) asm volatile("s_nop 1\n\t" OP " %0, %1\n\ts_nop 1" : "+v"(x[0]), "+v"(y[0]));                           \
    else if constexpr (NPAIR == 2) asm volatile("s_nop 1\n\t" OP " %0, %1\n\t" OP " %2, %3\n\ts_nop 1"                               \
                                                : "+v"(x[0]), "+v"(y[0]), "+v"(x[1]), "+v"(y[1]));                                   \
    else if constexpr (NPAIR == 3) asm volatile("s_nop 1\n\t" OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\ts_nop 1"             \
                                                : "+v"(x[0]), "+v"(y[0]), "+v"(x[1]), "+v"(y[1]), "+v"(x[2]), "+v"(y[2]));           \
    else asm volatile("s_nop 1\n\t" OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\ts_nop 1"                     \
                      : "+v"(x[0]), "+v"(y[0]), "+v"(x[1]), "+v"(y[1]), "+v"(x[2]), "+v"(y[2]), "+v"(x[3]), "+v"(y[3]))
    if constexpr (B32) { GRID_SW("v_permlane32_swap_b32"); } else { GRID_SW("v_permlane16_swap_b32"); }
#undef GRID_SW
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) x[q] += y[q];
}

// stage on the first m of the MF values of v (compile-time indices only: the array stays in registers)
template <int MF, int m, bool B32>
__device__ __forceinline__ void grid_swap_stage(float (&v)[MF])
{
    constexpr int h = (m + 1) / 2;                   // pairs (i, i + h), i < m - h; for odd m the value h - 1 pairs with a copy
    float x[h], y[h];
#pragma unroll
    for (int i = 0; i < h; ++i) { x[i] = v[i]; y[i] = (i + h < m) ? v[i + h] : v[i]; }
#pragma unroll
    for (int i0 = 0; i0 < h; i0 += 4) {
        constexpr int dummy = 0; (void)dummy;
        const int n = (h - i0 < 4) ? h - i0 : 4;
        if (n == 4) grid_swap_block<B32, 4>(x + i0, y + i0);
        else if (n == 3) grid_swap_block<B32, 3>(x + i0, y + i0);
        else if (n == 2) grid_swap_block<B32, 2>(x + i0, y + i0);
        else grid_swap_block<B32, 1>(x + i0, y + i0);
    }
#pragma unroll
    for (int i = 0; i < h; ++i) v[i] = x[i];
}

// one DPP stage inside a row of 16 lanes.  BIT8: partner = lane ^ 8 (row_ror:8, low side = banks 0,1); otherwise partner =
// lane ^ 4 (low side reads lane + 4 = row_shl:4 on banks 0 and 2, high side reads lane - 4 = row_shr:4 on banks 1 and 3).
// A bank-masked v_add_f32_dpp writes only the lanes of its banks, so the two adds of an output fill one register.
template <bool BIT8>
__device__ __forceinline__ float grid_dpp_pair(float lo, float hi)
{
    float out;
    if constexpr (BIT8)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                     "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc" : "=&v"(out) : "v"(lo), "v"(hi));
    else
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(out) : "v"(lo), "v"(hi));
    return out;
}
template <bool BIT8>
__device__ __forceinline__ float grid_dpp_self(float x)
{
    float out;
    if constexpr (BIT8)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=&v"(out) : "v"(x));
    else
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(out) : "v"(x));
    return out;
}

template <int MF, int m, bool BIT8>
__device__ __forceinline__ void grid_dpp_stage(float (&v)[MF])
{
    constexpr int h = (m + 1) / 2, P = m - h;
#pragma unroll
    for (int i = 0; i < P; ++i) v[i] = grid_dpp_pair<BIT8>(v[i], v[i + h]);
    if constexpr ((m & 1) != 0) v[h - 1] = grid_dpp_self<BIT8>(v[h - 1]);
}

// v[0 .. M) -> res[0 .. NS): see GridMap
template <int M>
__device__ __forceinline__ void grid_reduce_scatter(float (&v)[M], float (&res)[(M + 15) / 16])
{
    constexpr int m1 = (M + 1) / 2, m2 = (m1 + 1) / 2, m3 = (m2 + 1) / 2;
    grid_swap_stage<M, M, true>(v);
    grid_swap_stage<M, m1, false>(v);
    grid_dpp_stage<M, m2, true>(v);
    grid_dpp_stage<M, m3, false>(v);
#pragma unroll
    for (int r = 0; r < (M + 15) / 16; ++r) res[r] = v[r];
}

// ---- the tile and its two matrix-vector products ---------------------------------------------------------------------------
// Tile of the rows ids[nbase + 16 s + a], s < 2 NP, of a [.][4 LPR] fp32 table (uniform base + 32-bit byte offsets: the table
// must be smaller than 4 GiB), topics 4 j + b; vals -> c (as float).  Branch-free and in two waves of loads: every id / value of
// the lane first (a slot past the slice's last entry reads entry 0 and gets value 0 -> weight exactly 0), then every row.
// first wave of loads: the ids and (masked) values of the lane's 2 NP token slots
template <int NP>
__device__ __forceinline__ void grid_load_ids(int (&tm)[2 * NP], int (&cn)[2 * NP], const int* __restrict__ ids, const int* __restrict__ vals,
                                              const int N, const int nbase, const int a)
{
#pragma unroll
    for (int s = 0; s < 2 * NP; ++s) {
        const int n = nbase + 16 * s + a;
        const unsigned nc = n < N ? (unsigned)n : 0u;
        tm[s] = ids[nc];
        cn[s] = vals[nc];                                   // unconditional load, masked below
    }
#pragma unroll
    for (int s = 0; s < 2 * NP; ++s) cn[s] = (nbase + 16 * s + a < N) ? cn[s] : 0;
}
// second wave: the rows
template <int LPR, int NP>
__device__ __forceinline__ void grid_load_rows(gv2f (&B)[NP][LPR], gv2f (&c)[NP], const float* __restrict__ table, const int (&tm)[2 * NP],
                                               const int (&cn)[2 * NP], const int b)
{
    const char* __restrict__ tb = (const char*)table;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        c[q] = gv2f{(float)cn[2 * q], (float)cn[2 * q + 1]};
        const uint32_t o0 = (uint32_t)tm[2 * q] * (uint32_t)(16 * LPR) + 4u * (uint32_t)b;
        const uint32_t o1 = (uint32_t)tm[2 * q + 1] * (uint32_t)(16 * LPR) + 4u * (uint32_t)b;
#pragma unroll
        for (int j = 0; j < LPR; ++j)
            B[q][j] = gv2f{*(const float*)(tb + o0 + 16u * j), *(const float*)(tb + o1 + 16u * j)};
    }
}
template <int LPR, int NP>
__device__ __forceinline__ void grid_load_tile(gv2f (&B)[NP][LPR], gv2f (&c)[NP], const float* __restrict__ table,
                                               const int* __restrict__ ids, const int* __restrict__ vals, const int N,
                                               const int nbase, const int a, const int b)
{
    int tm[2 * NP], cn[2 * NP];
    grid_load_ids<NP>(tm, cn, ids, vals, N, nbase, a);
    grid_load_rows<LPR, NP>(B, c, table, tm, cn, b);
}

// s_n = 4 init4 + sum_i B[n][i] e_i for the lane's 2 NP tokens, complete in every lane of the quad: LPR packed fmas per token
// pair with e_j (the class's row in LDS, [j]) broadcast inside the pair by op_sel, then two DPP adds per token.
template <int LPR, int NP>
__device__ __forceinline__ void grid_phase1(const gv2f (&B)[NP][LPR], const float* __restrict__ e_row, const float init4, gv2f (&s)[NP])
{
    gv2f sacc[NP][2];
#pragma unroll
    for (int q = 0; q < NP; ++q) { sacc[q][0] = gv2f{init4, init4}; sacc[q][1] = gv2f{0.f, 0.f}; }
#pragma unroll
    for (int jq = 0; jq < (LPR + 3) / 4; ++jq) {
        const float4 ev = ((const float4*)e_row)[jq];
        const float ej[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 4 * jq + u;
            if (j < LPR) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    sacc[q][j & 1] = __builtin_elementwise_fma(B[q][j], gv2f{ej[u], ej[u]}, sacc[q][j & 1]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        gv2f s2 = sacc[q][0] + sacc[q][1];
        s2.x += dpp_f<0xB1>(s2.x); s2.y += dpp_f<0xB1>(s2.y);                     // quad_perm [1,0,3,2]
        s2.x += dpp_f<0x4E>(s2.x); s2.y += dpp_f<0x4E>(s2.y);                     // quad_perm [2,3,0,1]
        s[q] = s2;
    }
}

// gv[j] = sum over the lane's tokens of w_n B[n][j], j < LPR (the 16-lane reduce-scatter follows)
template <int LPR, int NP, int MV>
__device__ __forceinline__ void grid_phase2(const gv2f (&B)[NP][LPR], const gv2f (&w)[NP], float (&gv)[MV])
{
    static_assert(MV >= LPR, "grid_phase2");
    gv2f g2[LPR];
#pragma unroll
    for (int j = 0; j < LPR; ++j) g2[j] = B[0][j] * w[0];
#pragma unroll
    for (int q = 1; q < NP; ++q)
#pragma unroll
        for (int j = 0; j < LPR; ++j) g2[j] = __builtin_elementwise_fma(B[q][j], w[q], g2[j]);
#pragma unroll
    for (int j = 0; j < LPR; ++j) gv[j] = g2[j].x + g2[j].y;
}

// host copy of the map for LPR topics per class + EXTRA more values riding in the reduce-scatter (LDA: sum_n w_n, M = LPR + 1):
// topic_of_lane[r * 64 + l] = the topic lane l owns in result slot r (4 j + b for the primary owner of index j < LPR), -1 otherwise.
template <int LPR, int EXTRA = 1>
static void tmvb_grid_lane_map_fill(std::vector<int>& topic_of_lane)
{
    constexpr int M = LPR + EXTRA;
    const GridMap<M>& g = kGridMap<M>;
    topic_of_lane.assign((size_t)GridMap<M>::NS * 64, -1);
    for (int r = 0; r < GridMap<M>::NS; ++r)
        for (int l = 0; l < 64; ++l) {
            const int a = l >> 2, b = l & 3;
            if (g.primary[r][a] && g.idx[r][a] < LPR) topic_of_lane[(size_t)r * 64 + l] = 4 * g.idx[r][a] + b;
        }
}
