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
// One wave per SIMD: the 512-entry register file holds the five CG vectors, the CG solution and the gradient accumulators (token
// phase: the rows in flight); lambda (fp64) and vsq (fp32) live in LDS ([topic][lane], conflict free), see the kernel.
#pragma once

#include <utility>

typedef float cb_v2f __attribute__((ext_vector_type(2)));
typedef float cb_v16f __attribute__((ext_vector_type(16)));
typedef float cb_v4f __attribute__((ext_vector_type(4)));

struct CtmBatchTabs {
    const float* S = nullptr;        // [R * R] fp32 invsigma (pads 0), + 64 B readable slack
    const float* sdiag = nullptr;    // [64] diagonal of S (pads 0)
    const float* muf = nullptr;      // [64] mu (pads 0)
    float cg_tol2 = 1e-8f;           // a lane's CG stops at |r|^2 <= max(cg_tol2 |g|^2, cg_abs2):  relative 1e-4 ...
    float cg_abs2 = 0.0f;            // ... or 5 % of the Newton exit threshold ntol, whichever is looser
    int cg_maxit = 200;
    unsigned long long* cg_iters = nullptr;   // diagnostics: [0] CG trips summed over waves, [1] Newton trips summed over waves, [2] waves,
                                              // [3..10] shader cycles per phase summed over waves: token, logzeta, vsq, gradient assembly, CG, gradient mat-vec,
                                              // lambda update, spare; [11] whole kernel
    unsigned* next_item = nullptr;            // work queue of the persistent launch: the next wave-of-documents to take (zeroed before the launch)
    int n_items = 0;                          // waves-of-documents in the queue (64 documents each)
    unsigned long long* wave_log = nullptr;   // PROF only (TMVB_CTM_WAVE_LOG): per wave [start, end] of the 100 MHz wall clock, HW_ID, longest document
    // four-waves-per-item kernel (tmvb_ctm_quad.h): per wave w the row block [H w, H w + H) of invsigma in column-pair order, its diagonal and mu (16 floats each)
    const float* Sq = nullptr; const float* sdq = nullptr; const float* muq = nullptr;
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

// ---- SMEM streaming of the K x K table through FIXED scalar registers.
// s[36:67] = group A (two s_load_dwordx16), s[68:99] = group B.  Group g + 1 is in flight while the packed FMAs of group g
// execute (SMEM returns out of order: every wait is lgkmcnt(0), placed by hand).  All of it is volatile inline asm, and every
// statement names its 16-register block as a PHYSICAL-register operand ("{s[36:51]}"): output of the load, read-write of the wait
// behind it, input of the FMAs.  What was tried before, and why it is like this:
//  * a plain C++ load of the loop-invariant table is hoisted out of the solver loops and spilled to VGPR lanes
//    (10 000 v_readlane / v_writelane in the probe);
//  * compiler-allocated destination registers ("=s" operands): in the full kernel the register allocator, short of SGPRs,
//    spilled or copied a group that was still IN FLIGHT (v_writelane / s_mov of garbage), in a different instantiation after
//    every unrelated edit;
//  * registers named only in the asm text and in clobber lists: nothing tells the compiler that they are live BETWEEN the
//    statements, and after a change of register pressure elsewhere it put short-lived temporaries (a spill reload, the zero
//    for the accumulators) into s36 - s39 between a load and the FMAs that read them.
// With the blocks as operands the compiler knows they are live from the load to the last FMA and has no reason to move a value
// that must sit in the same physical registers at both ends.  The loads and waits also clobber nothing else; a statement that
// defines SGPRs gets an s_nop behind it from the hazard recognizer, the FMAs (inputs only) do not.
// tools/check_smem_inflight.py still scans the generated ISA for any read of an in-flight destination and, between the
// CBFX_BEGIN / CBFX_END markers, for any compiler-generated write to s36 - s99; build() fails on a finding.
struct cb_fx_regs { cb_v16f a0, a1, b0, b1; };          // s[36:51], s[52:67], s[68:83], s[84:99]

template <int BUF, int OFF>
__device__ __forceinline__ void cb_fx_load(cb_fx_regs& g, const float* S)
{
    if constexpr (BUF == 0) asm volatile("s_load_dwordx16 %0, %1, %2" : "={s[36:51]}"(g.a0) : "s"(S), "n"(OFF));
    else if constexpr (BUF == 1) asm volatile("s_load_dwordx16 %0, %1, %2" : "={s[52:67]}"(g.a1) : "s"(S), "n"(OFF));
    else if constexpr (BUF == 2) asm volatile("s_load_dwordx16 %0, %1, %2" : "={s[68:83]}"(g.b0) : "s"(S), "n"(OFF));
    else asm volatile("s_load_dwordx16 %0, %1, %2" : "={s[84:99]}"(g.b1) : "s"(S), "n"(OFF));
}
// completion point of the group in buffers 2 GRP, 2 GRP + 1 (N of them loaded)
template <int GRP, int N>
__device__ __forceinline__ void cb_fx_wait(cb_fx_regs& g)
{
    if constexpr (GRP == 0 && N == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0" : "+{s[36:51]}"(g.a0), "+{s[52:67]}"(g.a1) : : "memory");
    else if constexpr (GRP == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0" : "+{s[36:51]}"(g.a0) : : "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0" : "+{s[68:83]}"(g.b0), "+{s[84:99]}"(g.b1) : : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0" : "+{s[68:83]}"(g.b0) : : "memory");
}
// y += {s[lo], s[hi]} * p.x (ODD = 0) or p.y (ODD = 1); PAIR = which of the 32 SGPR pairs of the two groups
#define CB_FX_CASE(P, LO, HI, BLK, FIELD)                                                                                                  \
    if constexpr (PAIR == P) {                                                                                                             \
        if constexpr (ODD) asm volatile("v_pk_fma_f32 %0, s[" #LO ":" #HI "], %1, %0 op_sel:[0,1,0]" : "+v"(y) : "v"(p), "{s[" BLK "]}"(g.FIELD)); \
        else asm volatile("v_pk_fma_f32 %0, s[" #LO ":" #HI "], %1, %0 op_sel_hi:[1,0,1]" : "+v"(y) : "v"(p), "{s[" BLK "]}"(g.FIELD));    \
    }
template <int PAIR, int ODD>
__device__ __forceinline__ void cb_fx_fma(const cb_fx_regs& g, cb_v2f& y, const cb_v2f p)
{
    CB_FX_CASE(0, 36, 37, "36:51", a0)
    CB_FX_CASE(1, 38, 39, "36:51", a0)
    CB_FX_CASE(2, 40, 41, "36:51", a0)
    CB_FX_CASE(3, 42, 43, "36:51", a0)
    CB_FX_CASE(4, 44, 45, "36:51", a0)
    CB_FX_CASE(5, 46, 47, "36:51", a0)
    CB_FX_CASE(6, 48, 49, "36:51", a0)
    CB_FX_CASE(7, 50, 51, "36:51", a0)
    CB_FX_CASE(8, 52, 53, "52:67", a1)
    CB_FX_CASE(9, 54, 55, "52:67", a1)
    CB_FX_CASE(10, 56, 57, "52:67", a1)
    CB_FX_CASE(11, 58, 59, "52:67", a1)
    CB_FX_CASE(12, 60, 61, "52:67", a1)
    CB_FX_CASE(13, 62, 63, "52:67", a1)
    CB_FX_CASE(14, 64, 65, "52:67", a1)
    CB_FX_CASE(15, 66, 67, "52:67", a1)
    CB_FX_CASE(16, 68, 69, "68:83", b0)
    CB_FX_CASE(17, 70, 71, "68:83", b0)
    CB_FX_CASE(18, 72, 73, "68:83", b0)
    CB_FX_CASE(19, 74, 75, "68:83", b0)
    CB_FX_CASE(20, 76, 77, "68:83", b0)
    CB_FX_CASE(21, 78, 79, "68:83", b0)
    CB_FX_CASE(22, 80, 81, "68:83", b0)
    CB_FX_CASE(23, 82, 83, "68:83", b0)
    CB_FX_CASE(24, 84, 85, "84:99", b1)
    CB_FX_CASE(25, 86, 87, "84:99", b1)
    CB_FX_CASE(26, 88, 89, "84:99", b1)
    CB_FX_CASE(27, 90, 91, "84:99", b1)
    CB_FX_CASE(28, 92, 93, "84:99", b1)
    CB_FX_CASE(29, 94, 95, "84:99", b1)
    CB_FX_CASE(30, 96, 97, "84:99", b1)
    CB_FX_CASE(31, 98, 99, "84:99", b1)
}
#undef CB_FX_CASE

// blocks of 16 floats of group GI that exist (the table has NB of them)
template <int NB, int GI> constexpr int cb_fx_nblk = (2 * GI + 1 < NB) ? 2 : ((2 * GI < NB) ? 1 : 0);

template <int NB, int GI>
__device__ __forceinline__ void cb_fx_issue(cb_fx_regs& g, const float* S)
{
    if constexpr (2 * GI < NB) cb_fx_load<(GI & 1) * 2, (2 * GI) * 64>(g, S);
    if constexpr (2 * GI + 1 < NB) cb_fx_load<(GI & 1) * 2 + 1, (2 * GI + 1 < NB ? (2 * GI + 1) * 64 : 0)>(g, S);
}
// y[i] += S[j][i] p[j] for the 32 table entries of group GI, two columns i per packed FMA
template <int R, int GI>
__device__ __forceinline__ void cb_fx_consume(const cb_fx_regs& g, const cb_v2f (&p)[R / 2], cb_v2f (&y)[R / 2])
{
    tmvb_static_for<16>([&](auto tag) {
        constexpr int q = decltype(tag)::value;                 // pair q of the group: block q / 8, element pair q % 8
        constexpr int f = (2 * GI + q / 8) * 16 + 2 * (q % 8);
        if constexpr (f < R * R) {
            constexpr int j = f / R, i = f % R;
            cb_fx_fma<(GI & 1) * 16 + q, (j & 1)>(g, y[i / 2], p[j / 2]);
        }
    });
}
template <int R, int GI, int NG>
__device__ __forceinline__ void cb_fx_pipe(cb_fx_regs& g, const float* S, const cb_v2f (&p)[R / 2], cb_v2f (&y)[R / 2])
{
    constexpr int NB = (R * R + 15) / 16;
    if constexpr (GI < NG) {
        if constexpr (GI + 1 < NG) cb_fx_issue<NB, GI + 1>(g, S);
        cb_fx_consume<R, GI>(g, p, y);
        if constexpr (GI + 1 < NG) cb_fx_wait<(GI + 1) & 1, cb_fx_nblk<NB, GI + 1>>(g);
        cb_fx_pipe<R, GI + 1, NG>(g, S, p, y);
    }
}
// y = S p  (S symmetric, [R][R] flat)
template <int R>
__device__ __forceinline__ void cb_matvec_f32(const float* S, const cb_v2f (&p)[R / 2], cb_v2f (&y)[R / 2])
{
    constexpr int NB = (R * R + 15) / 16, NG = (NB + 1) / 2;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) y[i] = cb_v2f{0.f, 0.f};
    cb_fx_regs g;
    asm volatile("; CBFX_BEGIN");
    cb_fx_issue<NB, 0>(g, S);
    cb_fx_wait<0, cb_fx_nblk<NB, 0>>(g);
    cb_fx_pipe<R, 0, NG>(g, S, p, y);
    asm volatile("; CBFX_END");
}

// exp(a) in fp64, ~1e-14 relative: n = rint(a / ln 2), degree-11 Taylor polynomial on |r| <= ln 2 / 2, ldexp.  19 instructions
// (the library exp is ~3x that, and the kernel has 3 x K inlined call sites).
__device__ __forceinline__ double cb_exp(double a)
{
    a = fmax(a, -745.0);
    const double n = __builtin_rint(a * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, a);
    r = fma(n, -1.90821492927058770002e-10, r);
    double q = 2.50521083854417187751e-08;                 // 1 / 11!
    q = fma(q, r, 2.75573192239858906526e-07);             // 1 / 10!
    q = fma(q, r, 2.75573192239858906526e-06);
    q = fma(q, r, 2.48015873015873015873e-05);
    q = fma(q, r, 1.98412698412698412698e-04);
    q = fma(q, r, 1.38888888888888888889e-03);
    q = fma(q, r, 8.33333333333333333333e-03);
    q = fma(q, r, 4.16666666666666666667e-02);
    q = fma(q, r, 1.66666666666666666667e-01);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return __builtin_ldexp(q, (int)n);
}
// the same on N independent arguments, stage by stage: N-fold instruction-level parallelism for the single resident wave
template <int N>
__device__ __forceinline__ void cb_exp_n(double (&a)[N])
{
    double n[N], q[N];
#pragma unroll
    for (int u = 0; u < N; ++u) { a[u] = fmax(a[u], -745.0); n[u] = __builtin_rint(a[u] * 1.4426950408889634074); }
#pragma unroll
    for (int u = 0; u < N; ++u) a[u] = fma(n[u], -6.93147180369123816490e-01, a[u]);
#pragma unroll
    for (int u = 0; u < N; ++u) { a[u] = fma(n[u], -1.90821492927058770002e-10, a[u]); q[u] = fma(2.50521083854417187751e-08, a[u], 2.75573192239858906526e-07); }
    const double c[9] = {2.75573192239858906526e-06, 2.48015873015873015873e-05, 1.98412698412698412698e-04, 1.38888888888888888889e-03,
                         8.33333333333333333333e-03, 4.16666666666666666667e-02, 1.66666666666666666667e-01, 0.5, 1.0};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int u = 0; u < N; ++u) q[u] = fma(q[u], a[u], c[k]);
    }
#pragma unroll
    for (int u = 0; u < N; ++u) a[u] = __builtin_ldexp(fma(q[u], a[u], 1.0), (int)n[u]);
}
template <int N>
__device__ __forceinline__ void cb_rcp_n(const double (&a)[N], double (&y)[N])
{
#pragma unroll
    for (int u = 0; u < N; ++u) y[u] = __builtin_amdgcn_rcp(a[u]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        double e[N];
#pragma unroll
        for (int u = 0; u < N; ++u) e[u] = fma(-a[u], y[u], 1.0);
#pragma unroll
        for (int u = 0; u < N; ++u) y[u] = fma(e[u], y[u], y[u]);
    }
}

// 1 / a in fp64 for a > 0: v_rcp_f64 seed and two Newton refinements
__device__ __forceinline__ double cb_rcp(double a)
{
    double y = __builtin_amdgcn_rcp(a);
    y = fma(fma(-a, y, 1.0), y, y);
    y = fma(fma(-a, y, 1.0), y, y);
    return y;
}

// Small uniform tables (mu, the diagonal of S) as SGPRs: load and wait in ONE asm statement.  Their consumers are ordinary
// compiler-generated instructions, which -- unlike the volatile asm FMAs of the mat-vecs -- the scheduler is free to move
// above a separate wait statement (it did: v_cvt_f64_f32 of an SGPR right behind its s_load).
template <int OFF>
__device__ __forceinline__ cb_v4f cb_sload4_sync(const float* tab)
{
    cb_v4f v;
    asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(tab), "n"(OFF * 4));
    return v;
}
template <int NBLK>
__device__ __forceinline__ void cb_sload64_sync(const float* tab, cb_v16f (&b)[4])
{
    if constexpr (NBLK == 1) asm volatile("s_load_dwordx16 %0, %1, 0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b[0]) : "s"(tab));
    else if constexpr (NBLK == 2)
        asm volatile("s_load_dwordx16 %0, %2, 0\n\ts_load_dwordx16 %1, %2, 64\n\ts_waitcnt lgkmcnt(0)" : "=&s"(b[0]), "=&s"(b[1]) : "s"(tab));
    else if constexpr (NBLK == 3)
        asm volatile("s_load_dwordx16 %0, %3, 0\n\ts_load_dwordx16 %1, %3, 64\n\ts_load_dwordx16 %2, %3, 128\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(b[0]), "=&s"(b[1]), "=&s"(b[2]) : "s"(tab));
    else
        asm volatile("s_load_dwordx16 %0, %4, 0\n\ts_load_dwordx16 %1, %4, 64\n\ts_load_dwordx16 %2, %4, 128\n\ts_load_dwordx16 %3, %4, 192\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(b[0]), "=&s"(b[1]), "=&s"(b[2]), "=&s"(b[3]) : "s"(tab));
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

// Jacobi-preconditioned CG for (S + Diag(D)) x = g, one system per lane; x (pairs) stays in registers.
// `live` lanes iterate until |r|^2 <= tol2 |g|^2; the wave stops when no lane is live.  Returns the wave's trip count.
template <int R, typename TB>
__device__ __forceinline__ int cb_cg_solve(const TB& tb, const cb_v2f (&D)[R / 2], const cb_v2f (&dinv)[R / 2],
                                           const cb_v2f (&g)[R / 2], bool live, cb_v2f (&x)[R / 2])
{
    cb_v2f r[R / 2], pv[R / 2], y[R / 2];
#pragma unroll
    for (int i = 0; i < R / 2; ++i) { r[i] = g[i]; pv[i] = g[i] * dinv[i]; x[i] = cb_v2f{0.f, 0.f}; }
    const float gg = cb_dot<R>(g, g);
    float rz = cb_dot<R>(r, pv);
    const float thr = fmaxf(tb.cg_tol2 * gg, tb.cg_abs2);
    live = live && gg > thr;
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
            x[i] = __builtin_elementwise_fma(a2, pv[i], x[i]);
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

// One kernel argument, so that field offsets in the kernarg segment are plain offsetof()s.
struct CtmBatchArgs { CtmParams p; CtmBatchTabs tb; int64_t M; };

// A pointer-sized kernel argument read from the kernarg segment AT THE POINT OF USE.  Referencing `a.p.lambda` the normal way
// makes the compiler load every pointer at kernel entry and keep it in SGPRs for the whole kernel; with ~20 pointers live,
// the two in-flight SMEM groups of the mat-vecs (64 SGPRs) no longer fit the 102-SGPR file and the register allocator
// spilled an IN-FLIGHT group to VGPR lanes (tools/check_smem_inflight.py caught it in the KP = 44 instantiation).
template <typename T, int OFF>
__device__ __forceinline__ T cb_karg()
{
    static_assert(sizeof(T) == 8, "pointer-sized fields only");
    T v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(OFF));
    return v;
}
#define CB_KARG(type, field) cb_karg<type, (int)offsetof(CtmBatchArgs, field)>()
template <int OFF>
__device__ __forceinline__ int cb_karg32()
{
    int v;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(OFF));
    return v;
}
#define CB_KARG32(field) cb_karg32<(int)offsetof(CtmBatchArgs, field)>()

// One token step of the cooperative row gather (see the token phase of the kernel): every lane publishes the row id of its own
// document, reads back the ids of the LPR rows it fetches chunks of, and issues its LPR 16-byte loads.
// ROWB: bytes between rows of the gathered table: R * 4, or 256 for the copy of a 52-float table that ctm_launch_batch makes per
// E-step (ctm_rowpad_kernel, 4 us) -- a 208-byte row at a 208-byte stride straddles two or three 128-byte lines (2.5 on average),
// at a 256-byte stride exactly two.  A/B in one process tree (tools/ctm_ab.py, three alternating rounds): 205.6 / 205.8 / 202.9 it/s
// without, 208.8 / 208.1 / 203.8 with (+1 %).  -DTMVB_CTM_ROWPAD=0 builds the unpadded gather.
#ifndef TMVB_CTM_PDOT
#define TMVB_CTM_PDOT 1
#endif
#ifndef TMVB_CTM_ROWPAD
#define TMVB_CTM_ROWPAD 1
#endif

template <int R, bool FILT> struct cb_rowb { static constexpr unsigned value = (TMVB_CTM_ROWPAD && !FILT && R == 52) ? 256u : (unsigned)(R * 4); };
template <int R, unsigned ROWB>
__device__ __forceinline__ void cb_token_issue(const __attribute__((address_space(1))) float* tab, int* tl, int lane, const int (&rmap)[R / 4],
                                               const unsigned (&cbyte)[R / 4], int t, cb_v4f (&b)[R / 4])
{
    // (LDS operations of one wave execute in program order, and the compiler keeps may-aliasing LDS accesses in order: no barrier)
    tl[lane] = t;
    int tt[R / 4];
#pragma unroll
    for (int i = 0; i < R / 4; ++i) tt[i] = tl[rmap[i]];
    // uniform base + unsigned 32-bit byte offset (24-bit multiply: row ids are < 2^24): the SGPR-base form of global_load, one
    // VGPR of address per load (64-bit per-lane addresses were hoisted out of the token loop as 26 registers, spilled to scratch
    // and reloaded behind s_waitcnt vmcnt(0))
    const __attribute__((address_space(1))) char* base = (const __attribute__((address_space(1))) char*)tab;
#pragma unroll
    for (int i = 0; i < R / 4; ++i) b[i] = *(const __attribute__((address_space(1))) cb_v4f*)(base + (__umul24((unsigned)tt[i], ROWB) + cbyte[i]));
}

// ---- the kernel: wave w owns documents doc_order[64 w .. 64 w + 63] (sorted by length, so a wave's documents are alike)
// PROF = true adds the per-phase cycle counters of tmvb_ctm_solver_stats (TMVB_CTM_PROF=1): twelve more live SGPRs, which is
// what the production instantiation cannot afford next to the two in-flight SMEM groups (the allocator then spills a group).
// FILT = true: the filtered CTM (src/fCTM.jl).  phi carries the per-token switch as an exponent,
// phi[i,n] = softmax_i(tau_n log(beta[i,t_n] + eps) + lambda_i) (:216-219), so the token phase gathers rows of L = log(beta + eps),
// runs the two-pass column softmax per lane with update_tau! (:208-213) fused in, and writes tau / tau_old / the log-sum-exp per
// token every sweep (the statistics pass rebuilds the last sweep's phi from them); the sweep order is phi, tau, logzeta, LAMBDA,
// VSQ (:236-241).
template <int R, bool PROF, bool FILT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void ctm_estep_batch_kernel(CtmBatchArgs a)
{
    // hot scalars through the normal path; every pointer through CB_KARG where it is used
    const int p_K = a.p.K, p_viter = a.p.viter, p_niter = a.p.niter, p_debug = a.p.debug;
    const double p_ntol = a.p.ntol, p_vtol = a.p.vtol;
    struct { int K, viter, niter, debug; double ntol, vtol; } p = {p_K, p_viter, p_niter, p_debug, p_ntol, p_vtol};
    struct { const float* S; const float* sdiag; const float* muf; float cg_tol2, cg_abs2; int cg_maxit; } tb = {a.tb.S, a.tb.sdiag, a.tb.muf, a.tb.cg_tol2, a.tb.cg_abs2, a.tb.cg_maxit};
    const int64_t M = a.M;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS per wave: lambda in fp64 [R][64], vsq in fp32 [R][64] (the same 13 KB stage the token rows in the token phase, vsq parked in
    // registers meanwhile), 64 row ids.  Lambda used to live in 104 registers per lane next to five CG vectors and the token rows:
    // past the 512-entry file, and what the allocator then spilled to scratch was reloaded behind s_waitcnt vmcnt(0) in every phase.
    double* lam_l = (double*)lds;                        // [R][64] lambda, fp64
    float* vsf_l = lds + 2 * R * 64;                     // [R][64] vsq, fp32 between the phases (every phase computes on it in fp64)
    // One base register + an immediate offset per access.  The bases are re-derived from an opaque copy of the lane id at the top
    // of every phase (CB_VIEW): addresses computed from `lane` itself are sweep-loop invariants, and the compiler hoisted one
    // address register PER ELEMENT out of the loop (~150 of them) and reloaded them from scratch before every LDS access.
    double* lamp = lam_l + threadIdx.x;
    float* vsp = vsf_l + threadIdx.x;
#define CB_VIEW() do { int ln_ = lane; asm volatile("" : "+v"(ln_)); lamp = lam_l + ln_; vsp = vsf_l + ln_; } while (0)
#define CB_LAM_PAD (-1.0e30)
#define LAM(i) lamp[(i) * 64]
#define VSF(i) vsp[(i) * 64]
#define VSQ(i) ((double)vsp[(i) * 64])
    constexpr int LPR = R / 4;
    const int lane = threadIdx.x;
    const int K = p.K;
    // Persistent launch: one workgroup per SIMD slot of the device, each takes the next wave-of-documents from a queue until none is
    // left.  (With one 64-document workgroup per wave-of-documents the hardware dispatcher placed wave 1024 + k a median 130 us after
    // the k-th slot had become free -- profiles/r3_ctm_wave_log.txt -- and the launch ended with its slowest slot.)
    for (;;) {
    int item;
    { unsigned* q = CB_KARG(unsigned*, tb.next_item); unsigned v = 0; if (lane == 0) v = atomicAdd(q, 1u); item = __builtin_amdgcn_readfirstlane((int)v); }
    if (item >= CB_KARG32(tb.n_items)) break;
    const int64_t slot = (int64_t)item * 64 + lane;
    const bool valid = slot < M;
    const int d = CB_KARG(const int32_t*, p.doc_order)[valid ? slot : M - 1];
    int64_t off;
    int N;
    { const int64_t* doc_ptr = CB_KARG(const int64_t*, p.doc_ptr); off = doc_ptr[d]; N = valid ? (int)(doc_ptr[d + 1] - off) : 0; }
    const int Nmax = wave_max_i(N);
    if constexpr (PROF) {
        unsigned long long* wl = CB_KARG(unsigned long long*, tb.wave_log);
        if (wl && lane == 0) wl[4 * (int64_t)item] = wall_clock64();
    }

    // C_d = sum of counts (src/CTM.jl:33)
    float cl = 0.0f;
    { const int32_t* counts = CB_KARG(const int32_t*, p.counts); for (int n = 0; n < Nmax; ++n) cl += (n < N) ? (float)counts[off + n] : 0.0f; }
    const double Cd = (double)cl;

    double lz;
    {
        const float* lam_in = CB_KARG(const float*, p.lambda);
        const float* vsq_in = CB_KARG(const float*, p.vsq);
        {
            // Round 5: KP - K <= 7, so only the last seven topics can be pads -- the K tests of this loop, of the lambda_old store / read-back and of the vsq groups
            // fold at compile time for the other 45 (`i < R - 7 || i < K`), and the pad rows load through a clamped index and a select instead of a branch.
            // 52 branches with spilled lane masks fewer per item and sweep: 207.6 -> 216.1 VB it/s (five alternating rounds, profiles/r5_ctm_token_experiments.txt (8)).
            // The same fold in the gradient assembly and in the final stores measured worse or equal (another register allocation) and is NOT applied there.
            const float* lrow = lam_in + (int64_t)d * K;
            const float* vrow = vsq_in + (int64_t)d * K;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (i < R - 7) { LAM(i) = (double)lrow[i]; VSF(i) = vrow[i]; }
                else {
                    const bool on = i < K;
                    const int ic = min(i, K - 1);
                    const float lv = lrow[ic], vv = vrow[ic];
                    LAM(i) = on ? (double)lv : CB_LAM_PAD;
                    VSF(i) = on ? vv : 1.0f;
                }
            }
        }
        lz = (double)CB_KARG(const float*, p.logzeta)[d];
    }
    bool active = valid && p.viter > 0;
    int sweeps = 0;
    unsigned nsteps = 0, ncg = 0, ntrip = 0;
    long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long t_start = PROF ? __builtin_readcyclecounter() : 0;
    long long t_mark = t_start;
    auto lap = [&](int which) {
        if constexpr (PROF) { const long long now = __builtin_readcyclecounter(); cyc[which] += now - t_mark; t_mark = now; }
    };

    for (int v = 0; v < p.viter; ++v) {
        if (!__any(active)) break;
        if (active) ++sweeps;
        // lambda_old of this sweep goes straight to its output array (and is read back for the exit test): 52 more live
        // registers per lane would push the kernel past the 512-entry file, and spills to scratch cost far more than these
        // two strided passes per sweep (measured: 35 000 cycles per Newton step for reloading lambda from scratch)
        CB_VIEW();
        if (active) {
            float* lam_old_out = CB_KARG(float*, p.lambda_old);
#pragma unroll
            for (int i = 0; i < R; ++i) if (i < R - 7 || i < K) lam_old_out[(int64_t)d * K + i] = (float)LAM(i);
        }
        // ---- update_phi!  src/CTM.jl:175-178 in linear space, (phi * counts)_i = e_i sum_n w_n beta[i, t_n]
        float phic[R];
        if constexpr (FILT) {
            const float* Ltab = CB_KARG(const float*, p.L);
            const float* kappa = CB_KARG(const float*, p.kappa);
            const int32_t* terms = CB_KARG(const int32_t*, p.terms);
            const int32_t* counts = CB_KARG(const int32_t*, p.counts);
            float* tau = CB_KARG(float*, p.tau);
            float* tau_old = CB_KARG(float*, p.tau_old);
            float* lse = CB_KARG(float*, p.lse);
            float* aold = CB_KARG(float*, p.aold);
            const float eta = a.p.eta;
            cb_v2f lf[R / 2], acc[R / 2];
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
                lf[i] = cb_v2f{(2 * i < K) ? (float)LAM(2 * i) : -INFINITY, (2 * i + 1 < K) ? (float)LAM(2 * i + 1) : -INFINITY};
                acc[i] = cb_v2f{0.f, 0.f};
            }
            // CH steps' rows (and kappa entries) in flight together, the ids / counts / tau of the next CH steps fetched alongside:
            // no step waits for the dependent chain ids -> addresses -> rows.  K exps per token (v_exp_f32 issues at quarter
            // rate) make this phase compute-heavier than CTM's.
            constexpr int CH = 2;
            int tq[CH], tn[CH];
            float cq[CH], cn[CH], uq[CH], un[CH];
            auto load_ids = [&](int n0, int (&t)[CH], float (&c)[CH], float (&u)[CH]) {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const bool in = n0 + k < N;
                    t[k] = in ? terms[off + n0 + k] : 0;
                    c[k] = in ? (float)counts[off + n0 + k] : 0.0f;
                    u[k] = in ? tau[off + n0 + k] : 0.5f;
                }
            };
            load_ids(0, tq, cq, uq);
            for (int n0 = 0; n0 < Nmax; n0 += CH) {
                float4 rows[CH][LPR];
                float kp[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const float4* row = (const float4*)(Ltab + (int64_t)tq[k] * R);
#pragma unroll
                    for (int q = 0; q < LPR; ++q) rows[k][q] = row[q];
                    kp[k] = kappa[tq[k]];
                }
                load_ids(n0 + CH, tn, cn, un);
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const cb_v2f tp2 = cb_v2f{uq[k], uq[k]};
                    cb_v2f x[R / 2];
                    float m = -INFINITY;
#pragma unroll
                    for (int q = 0; q < LPR; ++q) {
                        x[2 * q] = __builtin_elementwise_fma(tp2, cb_v2f{rows[k][q].x, rows[k][q].y}, lf[2 * q]);          // pads: -inf
                        x[2 * q + 1] = __builtin_elementwise_fma(tp2, cb_v2f{rows[k][q].z, rows[k][q].w}, lf[2 * q + 1]);
                        m = fmaxf(m, fmaxf(fmaxf(x[2 * q].x, x[2 * q].y), fmaxf(x[2 * q + 1].x, x[2 * q + 1].y)));
                    }
                    cb_v2f s2 = cb_v2f{0.f, 0.f}, a2 = cb_v2f{0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < LPR; ++q) {
                        x[2 * q] = cb_v2f{__expf(x[2 * q].x - m), __expf(x[2 * q].y - m)};                                 // un-normalised phi
                        x[2 * q + 1] = cb_v2f{__expf(x[2 * q + 1].x - m), __expf(x[2 * q + 1].y - m)};
                        s2 += x[2 * q] + x[2 * q + 1];
                        a2 = __builtin_elementwise_fma(x[2 * q], cb_v2f{rows[k][q].x, rows[k][q].y}, a2);                   // pads: 0 * L(= 0)
                        a2 = __builtin_elementwise_fma(x[2 * q + 1], cb_v2f{rows[k][q].z, rows[k][q].w}, a2);
                    }
                    const float sn = s2.x + s2.y, an = a2.x + a2.y;
                    const float prod = __expf(fminf(-(an / sn), 87.0f));                                                   // prod_i beta^-phi  (:212)
                    const float tnew = eta / (TMVB_EPS_F + (eta + (1.0f - eta) * (kp[k] * prod)));
                    const float w = (cq[k] > 0.0f) ? cq[k] / sn : 0.0f;
                    const cb_v2f w2 = cb_v2f{w, w};
#pragma unroll
                    for (int i = 0; i < R / 2; ++i) acc[i] = __builtin_elementwise_fma(w2, x[i], acc[i]);
                    if (active && n0 + k < N) {
                        tau_old[off + n0 + k] = uq[k]; tau[off + n0 + k] = tnew; lse[off + n0 + k] = m + __logf(sn);
                        if (aold) aold[off + n0 + k] = an / sn;                              // the decomposed update_elbo!'s per-token exponent
                    }
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) { tq[k] = tn[k]; cq[k] = cn[k]; uq[k] = un[k]; }
            }
#pragma unroll
            for (int i = 0; i < R / 2; ++i) { phic[2 * i] = acc[i].x; phic[2 * i + 1] = acc[i].y; }                // (phi * counts)_i
        } else
        {
            float lmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < R; ++i) lmax = fmaxf(lmax, (float)LAM(i));
            cb_v2f e2[R / 2], acc[R / 2];
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
                e2[i] = cb_v2f{expf((float)LAM(2 * i) - lmax), expf((float)LAM(2 * i + 1) - lmax)};          // pads: expf(-1e30) = 0
                acc[i] = cb_v2f{0.f, 0.f};
            }
            // E keeps the LAST executed sweep's factor e = exp(lambda_old - max) for the statistics pass
            if (active) {
                float* E = CB_KARG(float*, p.E);
#pragma unroll
                for (int i = 0; i < R / 2; ++i) *(cb_v2f*)(E + (int64_t)d * R + 2 * i) = e2[i];
            }
            typedef const __attribute__((address_space(1))) float* gfloat_p;          // global_load, not flat_load: a flat load ties up
            typedef const __attribute__((address_space(1))) int32_t* gint_p;          // lgkmcnt as well and every LDS wait would drain them
            gfloat_p beta = (gfloat_p)CB_KARG(const float*, p.beta);
            gint_p terms = (gint_p)CB_KARG(const int32_t*, p.terms);
            gint_p counts = (gint_p)CB_KARG(const int32_t*, p.counts);
            // Token step n = token n of each of the wave's 64 documents: lane l needs row terms[off_l + n] of beta (KP floats, 208 B at
            // K = 50).  Lane = document cannot keep 64 documents' tiles on chip (1.1 MB), so every sweep re-reads nnz * KP * 4 bytes
            // (22.7 GB per E-step on SYN-NSF, from L2 / Infinity Cache).  How the rows are fetched decides the phase
            // (tools/probes/rowgather_probe.hip, profiles/r2_rowgather_probe.txt): when every lane loads its own row, a
            // global_load_dwordx4 touches 64 different lines and the CU's vector L1 looks up one line per cycle -- 832 cycles of L1
            // per step per wave, shared by the four waves of the CU (TCP busy 87 % of the kernel, TCP_PENDING_STALL 58 %;
            // profiles/r2_ctm_k50_mem_pmc.txt).  Here the 64 x LPR 16-byte chunks of a step are loaded COOPERATIVELY: chunk
            // q = 64 i + lane of instruction i belongs to row q / LPR (13 consecutive lanes read one row: ~13 lines per instruction,
            // 5x fewer L1 look-ups), CH steps stay in flight in registers, and each step is transposed through the 13 KB LDS region of
            // the CG solution (idle in this phase): ds_write_b128 in chunk order, ds_read_b128 of the lane's own row (stride 208 B:
            // conflict free).  The row ids travel the other way through a 64-entry LDS table.  VMEM returns in order, so a lane's own
            // ids / counts are fetched two rounds ahead (waiting for a load younger than the rows in flight would drain them).
#ifndef TMVB_CTM_CH52
#define TMVB_CTM_CH52 2
#endif
            constexpr int CH = (R >= 52) ? TMVB_CTM_CH52 : 4;
            cb_v4f* xl4 = (cb_v4f*)vsf_l;                              // [64 rows][LPR chunks]: the vsq region, vsq parked in registers
            int* tl = (int*)(lds + 3 * R * 64);                        // [64] row ids of one step
            float vpark[R];
#pragma unroll
            for (int i = 0; i < R; ++i) vpark[i] = VSF(i);
            __builtin_amdgcn_wave_barrier();
            int rmap[LPR];
            unsigned cbyte[LPR];
#pragma unroll
            for (int i = 0; i < LPR; ++i) { const int q = 64 * i + lane; rmap[i] = q / LPR; cbyte[i] = (unsigned)(q % LPR) * 16u; }
            int tq[CH], tn[CH], t2[CH];
            float cq[CH], cn[CH], c2[CH];
            // branch-free, 32-bit offsets from the uniform array bases (the host checks nnz and KP * V against 2^30 elements); an index
            // past the document's end is clamped into it (an empty document reads its neighbour's first token or the upload slack)
            // (the document's offset is read again every sweep: kept live through the Newton phases it was spilled to scratch and its
            // reload at the top of the token loop -- VMEM returns in order -- drained the rows in flight every round)
            const unsigned off4 = (unsigned)CB_KARG(const int64_t*, p.doc_ptr)[d] * 4u;
            typedef const __attribute__((address_space(1))) char* gchar_p;
            auto load_ids = [&](int n0, int (&t)[CH], float (&c)[CH]) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const bool in = n0 + u < N;
                    const unsigned ix = off4 + 4u * (unsigned)max(min(n0 + u, N - 1), 0);
                    const int tv = *(gint_p)((gchar_p)terms + ix), cv = *(gint_p)((gchar_p)counts + ix);
                    t[u] = in ? tv : 0;
                    c[u] = in ? (float)cv : 0.0f;
                }
            };
            cb_v4f buf[CH][LPR];
            load_ids(0, tq, cq);
            tmvb_static_for<CH>([&](auto tag) {
                constexpr int u = decltype(tag)::value;
                cb_token_issue<R, cb_rowb<R, FILT>::value>(beta, tl, lane, rmap, cbyte, tq[u], buf[u]);
            });
            load_ids(CH, tn, cn);
            for (int n0 = 0; n0 < Nmax; n0 += CH) {
                load_ids(n0 + 2 * CH, t2, c2);
                tmvb_static_for<CH>([&](auto tag) {
                    constexpr int u = decltype(tag)::value;
#pragma unroll
                    for (int i = 0; i < LPR; ++i) xl4[64 * i + lane] = buf[u][i];
                    cb_v4f row[LPR];
#pragma unroll
                    for (int q = 0; q < LPR; ++q) row[q] = xl4[lane * LPR + q];
                    cb_token_issue<R, cb_rowb<R, FILT>::value>(beta, tl, lane, rmap, cbyte, tn[u], buf[u]);      // step n0 + CH + u (row 0 past the longest document)
                    cb_v2f s0 = cb_v2f{0.f, 0.f}, s1 = cb_v2f{0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < LPR; ++q) {
                        s0 = __builtin_elementwise_fma(cb_v2f{row[q].x, row[q].y}, e2[2 * q], s0);
                        s1 = __builtin_elementwise_fma(cb_v2f{row[q].z, row[q].w}, e2[2 * q + 1], s1);
                    }
                    const cb_v2f ss = s0 + s1;
                    // a lane past its document's end reads term 0 with count 0; beta[:, 0] may be all zero (a term the corpus
                    // never uses: CTM's phi has no epsilon), so its weight is forced to 0 instead of 0 / 0
                    const float w = (cq[u] > 0.0f) ? cq[u] / (ss.x + ss.y) : 0.0f;
                    const cb_v2f w2 = cb_v2f{w, w};
#pragma unroll
                    for (int q = 0; q < LPR; ++q) {
                        acc[2 * q] = __builtin_elementwise_fma(w2, cb_v2f{row[q].x, row[q].y}, acc[2 * q]);
                        acc[2 * q + 1] = __builtin_elementwise_fma(w2, cb_v2f{row[q].z, row[q].w}, acc[2 * q + 1]);
                    }
                });
#pragma unroll
                for (int u = 0; u < CH; ++u) { tq[u] = tn[u]; cq[u] = cn[u]; tn[u] = t2[u]; cn[u] = c2[u]; }
            }
            __builtin_amdgcn_wave_barrier();
            CB_VIEW();
#pragma unroll
            for (int i = 0; i < R; ++i) VSF(i) = vpark[i];
#pragma unroll
            for (int i = 0; i < R / 2; ++i) { phic[2 * i] = e2[i].x * acc[i].x; phic[2 * i + 1] = e2[i].y * acc[i].y; }
        }
        lap(0);
        // ---- update_logzeta!  src/CTM.jl:169-171
        {
            CB_VIEW();
            double m = -INFINITY;
#pragma unroll
            for (int i = 0; i < R; ++i) if (!FILT || i < K) m = fmax(m, LAM(i) + 0.5 * VSQ(i));     // (the filtered instantiation measured faster with the tests)
            double s = 0.0;
            auto lz_chunk = [&](auto tag) {
                constexpr int i0 = 4 * decltype(tag)::value;
                double a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = LAM(i0 + u) + 0.5 * VSQ(i0 + u) - m;
                cb_exp_n<4>(a);
#pragma unroll
                for (int u = 0; u < 4; ++u) if (!FILT || i0 + u < K) s += a[u];                 // pads add exp(-1e30 - m) = +0
            };
            tmvb_static_for<LPR>(lz_chunk);
            if (active) lz = m + log(s);
        }
        auto run_vsq = [&]() {
        // ---- update_vsq!  src/CTM.jl:146-165 (one scalar Newton iteration per topic)
        if (!(p.debug & 1)) {
            // four topics per loop: four independent fp64 dependency chains for the single resident wave
            auto vsq_group = [&](auto tag) {
                constexpr int i0 = 4 * decltype(tag)::value;
                if (i0 >= K) return;
                CB_VIEW();
                double vs[4], isd[4];
                bool act[4];
                const cb_v4f sd4 = cb_sload4_sync<i0>(tb.sdiag);
                double lm[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { vs[u] = VSQ(i0 + u); lm[u] = LAM(i0 + u); act[u] = active && ((i0 + u < R - 7) || (i0 + u < K)); }
#pragma unroll
                for (int u = 0; u < 4; ++u) isd[u] = (double)sd4[u];
                // (round 4, measured and dropped: the iterations far from the exit threshold in fp32 -- __expf, v_rcp_f32, a lane leaving the fp32
                //  loop without taking the step once its gradient is within 16 ntol, so that every exit test and the last steps stay in fp64.
                //  tools/ctm_ab.py, three alternating rounds: 199.8 / 202.7 / 203.2 it/s against 207.6 / 208.6 / 205.0 -- the extra loop's
                //  registers and divergence cost more than ~2 of ~4 fp64 iterations save.  CH = 1 / 3 / 4 token steps in flight instead of
                //  2 at KP = 52: 76 / 185 / 44 it/s against 207 -- scratch.)
                for (int t = 0; t < p.niter; ++t) {
                    if (!__any(act[0] || act[1] || act[2] || act[3])) break;
                    double ex[4], rv[4], den[4], ihd[4], grad[4], pp[4], rho[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) ex[u] = lm[u] + 0.5 * vs[u] - lz;
                    cb_exp_n<4>(ex);
                    cb_rcp_n<4>(vs, rv);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        grad[u] = -0.5 * (isd[u] + Cd * ex[u] - rv[u]);                        // :150
                        den[u] = 0.25 * Cd * ex[u] + 0.5 * rv[u] * rv[u];
                    }
                    cb_rcp_n<4>(den, ihd);                                                      // -1 / den = inverse Hessian, :151
                    bool shrink = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { pp[u] = -ihd[u] * grad[u]; rho[u] = 1.0; shrink = shrink || (act[u] && vs[u] - pp[u] <= 0.0); }
                    while (__any(shrink)) {                                                    // :154
                        shrink = false;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (act[u] && vs[u] - rho[u] * pp[u] <= 0.0) rho[u] *= 0.5;
                            shrink = shrink || (act[u] && vs[u] - rho[u] * pp[u] <= 0.0);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (act[u]) vs[u] -= rho[u] * pp[u];
                        if (rho[u] * fabs(grad[u]) < p.ntol) act[u] = false;                    // :159
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (active && i0 + u < K) vs[u] += TMVB_EPS_D;                              // :164
                    VSF(i0 + u) = (float)vs[u];
                }
            };
            tmvb_static_for<LPR>(vsq_group);
        }
        };
        auto run_lambda = [&]() {
        // ---- update_lambda!  src/CTM.jl:129-142
        {
            bool newt = active;
            for (int t = 0; t < p.niter; ++t) {
                if (!__any(newt)) break;
                ++ntrip;
                if (newt) ++nsteps;
                CB_VIEW();
                cb_v2f g[R / 2], D[R / 2], dinv[R / 2];
                double gn2 = 0.0;
                {
                    // invsigma (mu - lambda), :134, with the fp32 mat-vec of the CG: fp64 FMAs issue at half rate and a fp64 mat-vec
                    // (2704 of them + 2704 widenings of the table, or a second 21.6 KB table that evicts the first from the 16 KB
                    // scalar cache) measured 100 000+ cycles per Newton step against 9 000 for this one.  Rounding mu - lambda to
                    // fp32 and accumulating in fp32 perturbs each gradient component by ~eps sqrt(K) |S_ij dm_j| ~ 1e-6 - 1e-5,
                    // 40 - 400 times below the exit threshold ntol = 1 / K^2; the terms that cancel at the optimum (phi counts,
                    // C_d e^{...}) stay in fp64 below.
                    cb_v2f dmf[R / 2], mvf[R / 2];
                    {
                        cb_v16f mb[4];
                        cb_sload64_sync<(R + 15) / 16>(tb.muf, mb);
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const float v = (float)((double)mb[i / 16][i % 16] - LAM(i));
                            if (i & 1) dmf[i / 2].y = v; else dmf[i / 2].x = v;
                        }
                    }
                    cb_matvec_f32<R>(tb.S, dmf, mvf);
                    lap(5);
                    double mv[R];
#pragma unroll
                    for (int i = 0; i < R / 2; ++i) { mv[2 * i] = (double)mvf[i].x; mv[2 * i + 1] = (double)mvf[i].y; }
                    auto grad_chunk = [&](auto tag) {
                        constexpr int i0 = 4 * decltype(tag)::value;
                        double ex[4];
                        const cb_v4f sdg = cb_sload4_sync<i0>(tb.sdiag);
#pragma unroll
                        for (int u = 0; u < 4; ++u) ex[u] = LAM(i0 + u) + 0.5 * VSQ(i0 + u) - lz;
                        cb_exp_n<4>(ex);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int i = i0 + u;
                            const bool on = i < K;
                            const double gd = on ? (mv[i] + (double)phic[i] - Cd * ex[u]) : 0.0;     // :134
                            gn2 = fma(gd, gd, gn2);
                            const float dval = on ? (float)(Cd * ex[u]) : 1.0f;                     // pad rows: unit rows
                            const float hd = sdg[u] + dval;                                         // -H_ii
                            if (i & 1) { g[i / 2].y = (float)gd; D[i / 2].y = dval; dinv[i / 2].y = 1.0f / hd; }
                            else { g[i / 2].x = (float)gd; D[i / 2].x = dval; dinv[i / 2].x = 1.0f / hd; }
                        }
                    };
                    tmvb_static_for<LPR>(grad_chunk);
                }
                lap(3);
                cb_v2f x[R / 2];
                ncg += (unsigned)cb_cg_solve<R>(tb, D, dinv, g, newt, x);
                lap(4);
                CB_VIEW();
                if (newt) {
#pragma unroll
                    for (int i = 0; i < R / 2; ++i) { LAM(2 * i) += (double)x[i].x; LAM(2 * i + 1) += (double)x[i].y; }     // :136
                }
                if (sqrt(gn2) < p.ntol) newt = false;                                           // :138
                lap(6);
            }
        }
        };
        lap(1);
#ifdef TMVB_MUTANT_FCTM_VSQ_FIRST
        // MUTANT (tests/test_mutants_gpu.py, never in a shipped build): fCTM's sweep in CTM's order (update_vsq! in front of update_lambda!)
        run_vsq(); lap(2); run_lambda();
#else
        if constexpr (FILT) { run_lambda(); lap(2); run_vsq(); }                    // src/fCTM.jl:239-240
        else { run_vsq(); lap(2); run_lambda(); }                                   // src/CTM.jl:198-199
#endif
        CB_VIEW();
        if (active) {
            const float* lam_old_in = CB_KARG(const float*, p.lambda_old);
            float dist2 = 0.0f;
#if TMVB_CTM_PDOT
            // sum_i (phi counts)_i (lambda_i - lambda_old_i) of this sweep, for the decomposed update_elbo! (CtmParams::pdot): phic's last use was
            // the gradient of the Newton loop just behind; one more FMA per topic here and one store per document and sweep
            float pdot = 0.0f;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (i < R - 7 || i < K) { const float df = (float)(LAM(i) - (double)lam_old_in[(int64_t)d * K + i]); dist2 = fmaf(df, df, dist2); pdot = fmaf(phic[i], df, pdot); }
            }
            { float* pd_out = CB_KARG(float*, p.pdot); if (pd_out) pd_out[d] = pdot; }
#else
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (i < R - 7 || i < K) { const float df = (float)(LAM(i) - (double)lam_old_in[(int64_t)d * K + i]); dist2 = fmaf(df, df, dist2); }
            }
#endif
            if (sqrtf(dist2) < (float)p.vtol) active = false;                                   // :200
        }
    }

    CB_VIEW();
    if (valid) {
        if (sweeps > 0) {
            float* lam_out = CB_KARG(float*, p.lambda);
            float* vsq_out = CB_KARG(float*, p.vsq);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (i < K) {
                    lam_out[(int64_t)d * K + i] = (float)LAM(i);
                    vsq_out[(int64_t)d * K + i] = VSF(i);
                }
            }
            CB_KARG(float*, p.logzeta)[d] = (float)lz;
        } else if constexpr (FILT) {
            float* lse = CB_KARG(float*, p.lse);
            for (int n = 0; n < N; ++n) lse[off + n] = INFINITY;                                // viter = 0: phi = 0 in the statistics
        } else {
            float* E = CB_KARG(float*, p.E);
#pragma unroll
            for (int i = 0; i < R; ++i) E[(int64_t)d * R + i] = 0.0f;                           // viter = 0: no responsibilities
        }
        CB_KARG(uint8_t*, p.sweeps)[d] = (uint8_t)min(sweeps, 255);
        CB_KARG(uint16_t*, p.doc_newton)[d] = (uint16_t)min(nsteps, 65535u);       // next E-step's grouping key (ctm_reorder_kernel)
    }
    const unsigned tot = wave_sum_u(valid ? nsteps : 0u);
    if (lane == 0) {
        unsigned long long* newton_steps = CB_KARG(unsigned long long*, p.newton_steps);
        unsigned long long* diag = CB_KARG(unsigned long long*, tb.cg_iters);
        if (newton_steps) atomicAdd(newton_steps, (unsigned long long)tot);
        if (diag) {
            atomicAdd(diag, (unsigned long long)ncg);
            atomicAdd(diag + 1, (unsigned long long)ntrip);
            atomicAdd(diag + 2, 1ull);
            if constexpr (PROF) {
                for (int q = 0; q < 8; ++q) atomicAdd(diag + 3 + q, (unsigned long long)cyc[q]);
                atomicAdd(diag + 11, (unsigned long long)(__builtin_readcyclecounter() - t_start));
                unsigned long long* wl = CB_KARG(unsigned long long*, tb.wave_log);
                if (wl) {
                    unsigned hw;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                    wl[4 * (int64_t)item + 1] = wall_clock64();
                    wl[4 * (int64_t)item + 2] = hw;
                    wl[4 * (int64_t)item + 3] = (unsigned long long)Nmax | ((unsigned long long)ntrip << 32);
                }
            }
        }
    }
    }   // next item
}
#undef LAM
#undef CB_LAM_PAD
#undef VSQ
#undef VSF
#undef CB_VIEW
