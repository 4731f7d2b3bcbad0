// tmvb_ctm_batch.h -- CTM E-step with one LANE per document (64 documents per wave), KP <= 52.
//
// Why: the lambda Newton step (src/CTM.jl:129-142) solves (invsigma + C_d Diag(e^{...})) x = g per document.  With one
// wave per document (ctm_estep_kernel) every flop of the solve drags a cross-lane broadcast along (v_readlane), and the
// solve is K^3: 2416 VALU instructions per solve at K = 50, 97 % of the kernel (profiles/r2_ctm_k50_*).  invsigma is the
// SAME matrix for every document; only the diagonal differs.  Turn the batch sideways: lane = document, the K-vectors of a
// document live in that lane's registers, and invsigma streams through SGPRs (s_load_dwordx16) as the scalar operand of
// v_pk_fma_f32 -- a mat-vec invsigma * p for 64 documents is K^2 / 2 packed FMAs with NO cross-lane traffic
// (tools/probes/matvec_probe.hip: 7250 cycles per 64 documents at K = 50, 5.4 cycles per v_pk_fma_f32).  A direct
// factorisation per lane would need the K x K matrix per lane (no room), so the system is solved by Jacobi-preconditioned
// conjugate gradients: the Jacobi-scaled Newton matrices of this model have condition numbers of 2 - 16 (measured on
// SYN-NSF, K = 50, iterations 1 - 12), i.e. 8 - 12 iterations to a 1e-5 relative residual, each one mat-vec.  Newton is
// self-correcting (the next gradient is evaluated exactly, in fp64, at the new point), so a 1e-5 relative solve changes
// neither the iterates' limit nor -- beyond borderline cases that fp32 vs fp64 already flips -- the number of steps.
//
// Everything else of the per-document chain (update_phi!, update_logzeta!, update_vsq!, the exit tests) is the same
// arithmetic as ctm_estep_kernel, per lane instead of per wave; loops run until no lane of the wave needs another trip.
// One wave per SIMD (the 512-entry register file holds the lane's vectors: lambda in fp64, the five CG vectors, the
// gradient accumulators); vsq (fp64) and the CG solution live in LDS ([topic][lane], conflict free).
#pragma once

#include <utility>

typedef float cb_v2f __attribute__((ext_vector_type(2)));
typedef float cb_v16f __attribute__((ext_vector_type(16)));
typedef double cb_v8d __attribute__((ext_vector_type(8)));

struct CtmBatchTabs {
    const float* S = nullptr;        // [R * R] fp32 invsigma (pads 0), + 64 B readable slack
    const double* Sd = nullptr;      // [R * R] the same values as doubles, + 64 B slack
    const double* sdiag = nullptr;   // [R] diagonal of S as doubles (pads 0)
    const double* mud = nullptr;     // [R] mu as doubles (pads 0)
    float cg_tol2 = 1e-10f;          // squared relative residual at which a lane's CG stops
    int cg_maxit = 200;
    unsigned long long* cg_iters = nullptr;   // diagnostics: total CG iterations (wave trips x 64)
};

template <typename F, int... I>
__device__ __forceinline__ void tmvb_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant in the body
template <int N, typename F>
__device__ __forceinline__ void tmvb_static_for(F&& f) { tmvb_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ unsigned wave_sum_u(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// ---- SMEM streaming: a flat table goes through SGPRs in groups of two s_load_dwordx16; group g + 1 is in flight while
// group g is consumed.  Inline asm because a plain C++ load of a loop-invariant table is hoisted out of the solver loops
// and spilled to VGPR lanes (10 000 v_readlane / v_writelane in the probe), and because SMEM returns out of order: every
// wait is lgkmcnt(0), placed explicitly.
template <int OFF>
__device__ __forceinline__ cb_v16f cb_sload16(const void* base)
{
    cb_v16f v;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(base), "n"(OFF));
    return v;
}
struct CbGroup { cb_v16f b[2]; };
// No register operands on the wait: a tied ("+s") operand made the register allocator copy a group that was still in
// flight (s_mov of the destination SGPRs BEFORE the wait = garbage).  Every consumer of a group is a volatile asm FMA that
// follows the wait in program order, so volatile ordering alone is the dependence.  tools/check_smem_inflight.py scans the
// generated ISA for any read of an in-flight destination.
__device__ __forceinline__ void cb_wait() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0" ::: "memory"); }

template <int NB, int GI>
__device__ __forceinline__ void cb_issue(CbGroup& g, const void* base)
{
    if constexpr (2 * GI < NB) g.b[0] = cb_sload16<(2 * GI) * 64>(base);
    if constexpr (2 * GI + 1 < NB) g.b[1] = cb_sload16<(2 * GI + 1 < NB ? (2 * GI + 1) * 64 : 0)>(base);
}

// y[i] += S[j][i] p[j], two columns i per packed FMA.  The FMAs are volatile asm statements: in the full kernel (unlike in
// the probe) the DAG scheduler, under register pressure, moved every FMA of the mat-vec behind ALL its loads and spilled
// the 2704 loaded values to VGPR lanes; volatile asm keeps load group / FMAs / wait in program order.
template <int ODD>
__device__ __forceinline__ void cb_pk_fma_s(cb_v2f& y, const cb_v2f s, const cb_v2f p)
{
    if constexpr (ODD) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(y) : "s"(s), "v"(p));
    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(y) : "s"(s), "v"(p));
}
template <int R, int GI>
__device__ __forceinline__ void cb_consume_f32(const CbGroup& g, const cb_v2f (&p)[R / 2], cb_v2f (&y)[R / 2])
{
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int f = (2 * GI + k) * 16 + 2 * e;
            if (f < R * R) {
                const int j = f / R, i = f % R;
                if (j & 1) cb_pk_fma_s<1>(y[i / 2], cb_v2f{g.b[k][2 * e], g.b[k][2 * e + 1]}, p[j / 2]);
                else cb_pk_fma_s<0>(y[i / 2], cb_v2f{g.b[k][2 * e], g.b[k][2 * e + 1]}, p[j / 2]);
            }
        }
    }
}
template <int R, int GI, int NG>
__device__ __forceinline__ void cb_pipe_f32(CbGroup& cur, CbGroup& nxt, const float* S, const cb_v2f (&p)[R / 2], cb_v2f (&y)[R / 2])
{
    constexpr int NB = (R * R + 15) / 16;
    if constexpr (GI < NG) {
        if constexpr (GI + 1 < NG) cb_issue<NB, GI + 1>(nxt, S);
        cb_consume_f32<R, GI>(cur, p, y);
        if constexpr (GI + 1 < NG) cb_wait();
        cb_pipe_f32<R, GI + 1, NG>(nxt, cur, S, p, y);
    }
}
// y = S p  (S symmetric, [R][R] flat)
template <int R>
__device__ __forceinline__ void cb_matvec_f32(const float* S, const cb_v2f (&p)[R / 2], cb_v2f (&y)[R / 2])
{
    constexpr int NB = (R * R + 15) / 16, NG = (NB + 1) / 2;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) y[i] = cb_v2f{0.f, 0.f};
    CbGroup a, b;
    cb_issue<NB, 0>(a, S);
    cb_wait();
    cb_pipe_f32<R, 0, NG>(a, b, S, p, y);
}

// mv[i] += Sd[j][i] dm[j] in fp64 (the Newton gradient's invsigma (mu - lambda), src/CTM.jl:134)
template <int R, int GI>
__device__ __forceinline__ void cb_consume_f64(const CbGroup& g, const double (&dm)[R], double (&mv)[R])
{
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const cb_v8d blk = __builtin_bit_cast(cb_v8d, g.b[k]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int f = (2 * GI + k) * 8 + e;
            if (f < R * R) {
                const int j = f / R, i = f % R;
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(mv[i]) : "s"(blk[e]), "v"(dm[j]));
            }
        }
    }
}
template <int R, int GI, int NG>
__device__ __forceinline__ void cb_pipe_f64(CbGroup& cur, CbGroup& nxt, const double* Sd, const double (&dm)[R], double (&mv)[R])
{
    constexpr int NB = (R * R + 7) / 8;
    if constexpr (GI < NG) {
        if constexpr (GI + 1 < NG) cb_issue<NB, GI + 1>(nxt, Sd);
        cb_consume_f64<R, GI>(cur, dm, mv);
        if constexpr (GI + 1 < NG) cb_wait();
        cb_pipe_f64<R, GI + 1, NG>(nxt, cur, Sd, dm, mv);
    }
}
template <int R>
__device__ __forceinline__ void cb_matvec_f64(const double* Sd, const double (&dm)[R], double (&mv)[R])
{
    constexpr int NB = (R * R + 7) / 8, NG = (NB + 1) / 2;
#pragma unroll
    for (int i = 0; i < R; ++i) mv[i] = 0.0;
    CbGroup a, b;
    cb_issue<NB, 0>(a, Sd);
    cb_wait();
    cb_pipe_f64<R, 0, NG>(a, b, Sd, dm, mv);
}

// a uniform double table entry as an SGPR pair (one s_load per use: a C++ load would be hoisted and spilled)
template <int I>
__device__ __forceinline__ double cb_sdouble(const double* tab)
{
    double v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(tab), "n"(I * 8));
    return v;
}

template <int R>
__device__ __forceinline__ float cb_dot(const cb_v2f (&a)[R / 2], const cb_v2f (&b)[R / 2])
{
    cb_v2f s0 = a[0] * b[0], s1 = cb_v2f{0.f, 0.f};
#pragma unroll
    for (int i = 1; i < R / 2; ++i) {
        if (i & 1) s1 = __builtin_elementwise_fma(a[i], b[i], s1);
        else s0 = __builtin_elementwise_fma(a[i], b[i], s0);
    }
    const cb_v2f s = s0 + s1;
    return s.x + s.y;
}

// Jacobi-preconditioned CG for (S + Diag(D)) x = g, one system per lane.  x is left in x_l[(i / 2) * 64 + lane] (pairs).
// `live` lanes iterate until |r|^2 <= tol2 |g|^2; the wave stops when no lane is live.  Returns the wave's trip count.
template <int R>
__device__ __forceinline__ int cb_cg_solve(const CtmBatchTabs& tb, const cb_v2f (&D)[R / 2], const cb_v2f (&dinv)[R / 2],
                                           const cb_v2f (&g)[R / 2], bool live, cb_v2f* __restrict__ x_l, int lane)
{
    cb_v2f r[R / 2], pv[R / 2], y[R / 2];
#pragma unroll
    for (int i = 0; i < R / 2; ++i) { r[i] = g[i]; pv[i] = g[i] * dinv[i]; x_l[i * 64 + lane] = cb_v2f{0.f, 0.f}; }
    const float gg = cb_dot<R>(g, g);
    float rz = cb_dot<R>(r, pv);
    const float thr = tb.cg_tol2 * gg;
    live = live && gg > 0.0f;
    int trips = 0;
    while (trips < tb.cg_maxit && __any(live)) {
        ++trips;
        cb_matvec_f32<R>(tb.S, pv, y);
#pragma unroll
        for (int i = 0; i < R / 2; ++i) y[i] = __builtin_elementwise_fma(D[i], pv[i], y[i]);
        const float pHp = cb_dot<R>(pv, y);
        const float alpha = (live && pHp > 0.0f) ? rz / pHp : 0.0f;
        const cb_v2f a2 = cb_v2f{alpha, alpha}, na2 = cb_v2f{-alpha, -alpha};
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
            x_l[i * 64 + lane] = __builtin_elementwise_fma(a2, pv[i], x_l[i * 64 + lane]);
            r[i] = __builtin_elementwise_fma(na2, y[i], r[i]);
            y[i] = r[i] * dinv[i];                                   // z
        }
        const float rr = cb_dot<R>(r, r);
        const float rz_new = cb_dot<R>(r, y);
        if (rr <= thr) live = false;
        const float beta = (live && rz > 0.0f) ? rz_new / rz : 0.0f;
        rz = rz_new;
        const cb_v2f b2 = cb_v2f{beta, beta};
#pragma unroll
        for (int i = 0; i < R / 2; ++i) pv[i] = __builtin_elementwise_fma(b2, pv[i], y[i]);
    }
    return trips;
}

// ---- the kernel: wave w owns documents doc_order[64 w .. 64 w + 63] (sorted by length, so a wave's documents are alike)
template <int R>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void ctm_estep_batch_kernel(CtmParams p, CtmBatchTabs tb, int64_t M)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    double* vs_l = (double*)lds;                         // [R][64] vsq, fp64
    cb_v2f* x_l = (cb_v2f*)(lds + 2 * R * 64);           // [R / 2][64] CG solution pairs
    constexpr int LPR = R / 4;
    const int lane = threadIdx.x;
    const int K = p.K;
    const int64_t slot = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = slot < M;
    const int d = p.doc_order[valid ? slot : M - 1];
    const int64_t off = p.doc_ptr[d];
    const int N = valid ? (int)(p.doc_ptr[d + 1] - off) : 0;
    const int Nmax = wave_max_i(N);

    // C_d = sum of counts (src/CTM.jl:33)
    float cl = 0.0f;
    for (int n = 0; n < Nmax; ++n) cl += (n < N) ? (float)p.counts[off + n] : 0.0f;
    const double Cd = (double)cl;

    double lam[R];
    float dsum[R];                               // lambda - lambda_old of the current sweep (sum of its Newton steps)
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const bool on = i < K;
        lam[i] = on ? (double)p.lambda[(int64_t)d * K + i] : 0.0;
        vs_l[i * 64 + lane] = on ? (double)p.vsq[(int64_t)d * K + i] : 1.0;
        dsum[i] = 0.0f;
    }
    double lz = (double)p.logzeta[d];
    bool active = valid && p.viter > 0;
    int sweeps = 0;
    unsigned nsteps = 0, ncg = 0;

    for (int v = 0; v < p.viter; ++v) {
        if (!__any(active)) break;
        if (active) ++sweeps;
        // ---- update_phi!  src/CTM.jl:175-178 in linear space, (phi * counts)_i = e_i sum_n w_n beta[i, t_n]
        float phic[R];
        {
            float lmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < R; ++i) if (i < K) lmax = fmaxf(lmax, (float)lam[i]);
            cb_v2f e2[R / 2], acc[R / 2];
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
                e2[i] = cb_v2f{(2 * i < K) ? expf((float)lam[2 * i] - lmax) : 0.0f, (2 * i + 1 < K) ? expf((float)lam[2 * i + 1] - lmax) : 0.0f};
                acc[i] = cb_v2f{0.f, 0.f};
            }
            // E keeps the LAST executed sweep's factor e = exp(lambda_old - max) for the statistics pass
            if (active) {
#pragma unroll
                for (int i = 0; i < R / 2; ++i) *(cb_v2f*)(p.E + (int64_t)d * R + 2 * i) = e2[i];
            }
            float4 cur[LPR], nxt[LPR];
            int t0 = (0 < N) ? p.terms[off] : 0;
            float c_cur = (0 < N) ? (float)p.counts[off] : 0.0f;
            {
                const float4* row = (const float4*)(p.beta + (int64_t)t0 * R);
#pragma unroll
                for (int q = 0; q < LPR; ++q) cur[q] = row[q];
            }
            for (int n = 0; n < Nmax; ++n) {
                const bool more = n + 1 < N;
                const int t1 = more ? p.terms[off + n + 1] : 0;
                const float c_nxt = more ? (float)p.counts[off + n + 1] : 0.0f;
                const float4* row = (const float4*)(p.beta + (int64_t)t1 * R);
#pragma unroll
                for (int q = 0; q < LPR; ++q) nxt[q] = row[q];
                cb_v2f s0 = cb_v2f{0.f, 0.f}, s1 = cb_v2f{0.f, 0.f};
#pragma unroll
                for (int q = 0; q < LPR; ++q) {
                    s0 = __builtin_elementwise_fma(cb_v2f{cur[q].x, cur[q].y}, e2[2 * q], s0);
                    s1 = __builtin_elementwise_fma(cb_v2f{cur[q].z, cur[q].w}, e2[2 * q + 1], s1);
                }
                const cb_v2f ss = s0 + s1;
                // a lane past its document's end reads term 0 with count 0; beta[:, 0] may be all zero (a term the corpus never
                // uses: CTM's phi has no epsilon), so its weight is forced to 0 instead of 0 / 0
                const float w = (c_cur > 0.0f) ? c_cur / (ss.x + ss.y) : 0.0f;
                const cb_v2f w2 = cb_v2f{w, w};
#pragma unroll
                for (int q = 0; q < LPR; ++q) {
                    acc[2 * q] = __builtin_elementwise_fma(w2, cb_v2f{cur[q].x, cur[q].y}, acc[2 * q]);
                    acc[2 * q + 1] = __builtin_elementwise_fma(w2, cb_v2f{cur[q].z, cur[q].w}, acc[2 * q + 1]);
                }
#pragma unroll
                for (int q = 0; q < LPR; ++q) cur[q] = nxt[q];
                c_cur = c_nxt;
            }
#pragma unroll
            for (int i = 0; i < R / 2; ++i) { phic[2 * i] = e2[i].x * acc[i].x; phic[2 * i + 1] = e2[i].y * acc[i].y; }
        }
        // ---- update_logzeta!  src/CTM.jl:169-171
        {
            double m = -INFINITY;
#pragma unroll
            for (int i = 0; i < R; ++i) if (i < K) m = fmax(m, lam[i] + 0.5 * vs_l[i * 64 + lane]);
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) if (i < K) s += exp(lam[i] + 0.5 * vs_l[i * 64 + lane] - m);
            if (active) lz = m + log(s);
        }
        // ---- update_vsq!  src/CTM.jl:146-165 (one scalar Newton iteration per topic)
        if (!(p.debug & 1)) {
            auto vsq_topic = [&](auto tag) {
                constexpr int i = decltype(tag)::value;
                if (i >= K) return;
                const double isdiag = cb_sdouble<i>(tb.sdiag);
                double vs = vs_l[i * 64 + lane];
                bool act = active;
                for (int t = 0; t < p.niter; ++t) {
                    if (!__any(act)) break;
                    double rho = 1.0;
                    const double ex = exp(lam[i] + 0.5 * vs - lz);
                    const double grad = -0.5 * (isdiag + Cd * ex - 1.0 / vs);                  // :150
                    const double ihd = -1.0 / (0.25 * Cd * ex + 0.5 / (vs * vs));             // :151
                    const double pp = ihd * grad;
                    while (__any(act && vs - rho * pp <= 0.0)) rho = (vs - rho * pp <= 0.0) ? rho * 0.5 : rho;   // :154
                    if (act) vs -= rho * pp;
                    if (rho * fabs(grad) < p.ntol) act = false;                                // :159
                }
                if (active) vs += TMVB_EPS_D;                                                  // :164
                vs_l[i * 64 + lane] = vs;
            };
            tmvb_static_for<R>(vsq_topic);
        }
        // ---- update_lambda!  src/CTM.jl:129-142
        {
#pragma unroll
            for (int i = 0; i < R; ++i) if (active) dsum[i] = 0.0f;
            bool newt = active;
            for (int t = 0; t < p.niter; ++t) {
                if (!__any(newt)) break;
                if (newt) ++nsteps;
                cb_v2f g[R / 2], D[R / 2], dinv[R / 2];
                double gn2 = 0.0;
                {
                    double dm[R], mv[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) dm[i] = 0.0;
                    auto set_dm = [&](auto tag) { constexpr int i = decltype(tag)::value; dm[i] = cb_sdouble<i>(tb.mud) - lam[i]; };
                    tmvb_static_for<R>(set_dm);
                    cb_matvec_f64<R>(tb.Sd, dm, mv);
                    auto grad_i = [&](auto tag) {
                        constexpr int i = decltype(tag)::value;
                        const bool on = i < K;
                        const double ex = on ? exp(lam[i] + 0.5 * vs_l[i * 64 + lane] - lz) : 0.0;
                        const double gd = on ? (mv[i] + (double)phic[i] - Cd * ex) : 0.0;         // :134
                        gn2 = fma(gd, gd, gn2);
                        const float dval = on ? (float)(Cd * ex) : 1.0f;                        // pad rows: unit rows
                        const float hd = (float)cb_sdouble<i>(tb.sdiag) + dval;                  // -H_ii
                        if (i & 1) { g[i / 2].y = (float)gd; D[i / 2].y = dval; dinv[i / 2].y = 1.0f / hd; }
                        else { g[i / 2].x = (float)gd; D[i / 2].x = dval; dinv[i / 2].x = 1.0f / hd; }
                    };
                    tmvb_static_for<R>(grad_i);
                }
                ncg += (unsigned)cb_cg_solve<R>(tb, D, dinv, g, newt, x_l, lane);
                if (newt) {
#pragma unroll
                    for (int i = 0; i < R / 2; ++i) {
                        const cb_v2f x = x_l[i * 64 + lane];
                        lam[2 * i] += (double)x.x; lam[2 * i + 1] += (double)x.y;             // :136
                        dsum[2 * i] += x.x; dsum[2 * i + 1] += x.y;
                    }
                }
                if (sqrt(gn2) < p.ntol) newt = false;                                           // :138
            }
        }
        float dist2 = 0.0f;
#pragma unroll
        for (int i = 0; i < R; ++i) dist2 = fmaf(dsum[i], dsum[i], dist2);
        if (sqrtf(dist2) < (float)p.vtol) active = false;                                       // :200
    }

    if (valid) {
        if (sweeps > 0) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (i < K) {
                    p.lambda[(int64_t)d * K + i] = (float)lam[i];
                    p.lambda_old[(int64_t)d * K + i] = (float)(lam[i] - (double)dsum[i]);
                    p.vsq[(int64_t)d * K + i] = (float)vs_l[i * 64 + lane];
                }
            }
            p.logzeta[d] = (float)lz;
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) p.E[(int64_t)d * R + i] = 0.0f;                         // viter = 0: no responsibilities
        }
        p.sweeps[d] = (uint8_t)min(sweeps, 255);
    }
    const unsigned tot = wave_sum_u(valid ? nsteps : 0u);
    if (lane == 0) {
        if (p.newton_steps) atomicAdd(p.newton_steps, (unsigned long long)tot);
        if (tb.cg_iters) atomicAdd(tb.cg_iters, (unsigned long long)ncg);
    }
}
