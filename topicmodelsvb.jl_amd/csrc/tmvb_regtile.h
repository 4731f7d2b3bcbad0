// tmvb_regtile.h -- device helpers of the register-tile E-step kernels (LDA: tmvb_lda.hip, CTPF: tmvb_ctpf.hip).
// The N_d x KP tile of a document lives in VGPRs as topic pairs (lane = token / reader, register = topic); the
// per-topic factor e_q reaches all lanes as SGPR pairs; the cross-lane sums run through lane_reduce_scatter
// (tmvb_common_kernels.h), whose topic -> lane map is the compile-time constant kRegLaneMap<R>.
#pragma once
#include "tmvb_common_kernels.h"

#include <utility>

typedef float v2f __attribute__((ext_vector_type(2)));

// e_q broadcasts of one block of <= 32 topics: v_readlane with an immediate lane (the lane map is a
// compile-time constant), results in SGPR pairs consumed directly by v_pk_fma_f32.
template <int R, int Q> struct LaneOfTopic { static constexpr int value = kRegLaneMap<R>.lane_of_topic[Q]; };

template <int R, int Q0, int NS, int... I>
__device__ __forceinline__ void regtile_bcast(float (&es)[sizeof...(I)], const float (&e)[NS], std::integer_sequence<int, I...>)
{
    ((es[I] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                  __builtin_bit_cast(int, LaneOfTopic<R, Q0 + I>::value < 64 ? e[0] : e[NS - 1]),
                  LaneOfTopic<R, Q0 + I>::value & 63))), ...);
}

// phase 1 for topics [Q0, Q0 + 32): sacc[t] += B[t][q] e_q, two topics per packed fma
template <int R, int T, int Q0>
__device__ __forceinline__ void regtile_phase1_block(const v2f (&B2)[T][R / 2], const float (&e)[(R + 63) / 64], v2f (&sacc)[T][2])
{
    constexpr int QB = (R - Q0 < 32) ? R - Q0 : 32;
    float es[QB];
    regtile_bcast<R, Q0>(es, e, std::make_integer_sequence<int, QB>{});
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int q = 0; q < QB; q += 2) {
            const v2f e2 = v2f{es[q], es[q + 1]};
            sacc[t][(q >> 1) & 1] = __builtin_elementwise_fma(B2[t][(Q0 + q) / 2], e2, sacc[t][(q >> 1) & 1]);
        }
    }
}

