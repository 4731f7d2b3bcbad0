// tmvb_ctm_quad.h -- round 6: the lane-per-document CTM E-step (tmvb_ctm_batch.h) re-cut so that FOUR WAVES share a wave-of-documents.
//
// Why (profiles/r5_ctm_token_experiments.txt, profiles/r5_kernel_resources.txt): the one-wave kernel owns a whole SIMD -- 512 registers, 372 B of
// scratch per lane, 40 KB of LDS -- because at lane = document every K-vector of the Newton solve costs 52 registers per lane and the solve needs
// seven of them.  Alone on its SIMD the wave exposes every latency it meets: the in-order vector-memory pipe of the token phase (42 % of the
// kernel), the fp64 chains of update_vsq!, the scalar-memory stream of the mat-vecs (VALU busy 41 %).  A second wave cannot be had by shrinking
// the kernel: the state of 64 documents IS what fills the SIMD.  So the wave-of-documents is cut the other way:
//   * a workgroup of four waves owns 64 documents; in the Newton phases (update_logzeta!, update_vsq!, update_lambda!, src/CTM.jl:129-171) lane =
//     document as before, but wave w holds only topics [H w, H w + H), H = KP / 4, of every K-vector: 13 registers per vector instead of 52.
//     Row block w of invsigma streams through the same fixed SGPRs into v_pk_fma_f32 -- pairs of COLUMNS j, j + 1 this time (H is odd), two
//     partial sums per row that are added at the end: 338 packed FMAs per mat-vec and wave instead of 1 352.  The full input vector comes
//     through the LDS ([topic][lane], conflict free), the scalar products as four partial sums per document that every wave adds in the same
//     order (bit-identical decisions in the four waves: all loops stay workgroup-uniform without a vote).  Two barriers per CG trip.
//   * in the token phase (update_phi!, src/CTM.jl:175-178) wave w walks the tokens of documents 16 w .. 16 w + 15 with FOUR LANES PER DOCUMENT:
//     lane q of a quad loads the 16-byte chunks q, q + 4, q + 8, q + 12 of the document's row (a quad reads 64 contiguous bytes per load), the
//     dot product closes with two quad_perm DPP adds, no LDS inside the loop.  exp(lambda - max) goes in and phi * counts comes out through
//     the LDS in the [topic][document] layout the Newton phases use.
//   * lambda (fp64) in registers; its fp32 image, phi * counts and vsq in LDS rows ([topic][document]); 256 registers, 88 B of scratch (none inside a
//     phase's loops), 78 KB of LDS per workgroup: two workgroups per CU = two waves per SIMD, the second wave is what hides the first one's latencies.
//     An item takes a quarter of the time: the launch ends with a shorter tail, and a 1/8 shard's E-step falls from 3.4 to 1.05 ms.
// Arithmetic: the same operations as tmvb_ctm_batch.h in another summation order (mat-vec, dot products); the CG stopping rule, the fp64 parts
// (gradient, logzeta, the vsq Newton iterations, lambda itself) and the exit tests are unchanged.  CTM only (fCTM keeps the one-wave kernel).
// Measured: 296 - 299 VB it/s on SYN-NSF K = 50 against 205 - 212 for the one-wave kernel (profiles/r6_ctm_experiments.txt: the six steps and what each was worth).
#pragma once

#ifndef TMVB_CTM_QWAVES
#define TMVB_CTM_QWAVES 2                 // workgroups per CU = waves per SIMD the kernel is compiled for (3: 168 registers, measured slower -- see DESIGN.md)
#endif
template <int R> struct cq_dim {
    static constexpr int H = R / 4;                      // topics per wave (R = 4 * odd)
    static constexpr int JP = R / 2;                     // column pairs
    static constexpr int NPAIR = JP * H;                 // packed FMAs per mat-vec and wave
    static constexpr int NB = (2 * NPAIR + 15) / 16;     // 16-float blocks of one wave's table
    static constexpr int NG = (NB + 1) / 2;              // groups of two blocks
    static constexpr int LPR = R / 4;                    // 16-byte chunks of a row
    static constexpr int NS = (LPR + 3) / 4;             // chunk slots per lane of a quad
    static constexpr unsigned ROWB = NS * 64u;           // bytes per row of the padded gather table (ctm_rowpad_generic_kernel)
    static constexpr int XSF = 2 * 3 * 4 * 64;           // floats of the scalar exchange: [parity][value][wave][lane]
    static constexpr int XSD = 2 * 1 * 4 * 64;           // doubles of the fp64 scalar exchange
    static constexpr size_t lds_bytes = (size_t)(5 * R * 64 + XSF) * 4 + (size_t)XSD * 8 + (5 * 64 + 4) * 4;
};

// y += {s[lo], s[hi]} * p component-wise; PAIR = which of the 32 SGPR pairs of the two groups (tmvb_ctm_batch.h: cb_fx_regs)
#define CQ_FX_CASE(P, LO, HI, BLK, FIELD)                                                                                   \
    if constexpr (PAIR == P) asm volatile("v_pk_fma_f32 %0, s[" #LO ":" #HI "], %1, %0" : "+v"(y) : "v"(p), "{s[" BLK "]}"(g.FIELD));
template <int PAIR>
__device__ __forceinline__ void cq_fx_fma(const cb_fx_regs& g, cb_v2f& y, const cb_v2f p)
{
    CQ_FX_CASE(0, 36, 37, "36:51", a0) CQ_FX_CASE(1, 38, 39, "36:51", a0) CQ_FX_CASE(2, 40, 41, "36:51", a0) CQ_FX_CASE(3, 42, 43, "36:51", a0)
    CQ_FX_CASE(4, 44, 45, "36:51", a0) CQ_FX_CASE(5, 46, 47, "36:51", a0) CQ_FX_CASE(6, 48, 49, "36:51", a0) CQ_FX_CASE(7, 50, 51, "36:51", a0)
    CQ_FX_CASE(8, 52, 53, "52:67", a1) CQ_FX_CASE(9, 54, 55, "52:67", a1) CQ_FX_CASE(10, 56, 57, "52:67", a1) CQ_FX_CASE(11, 58, 59, "52:67", a1)
    CQ_FX_CASE(12, 60, 61, "52:67", a1) CQ_FX_CASE(13, 62, 63, "52:67", a1) CQ_FX_CASE(14, 64, 65, "52:67", a1) CQ_FX_CASE(15, 66, 67, "52:67", a1)
    CQ_FX_CASE(16, 68, 69, "68:83", b0) CQ_FX_CASE(17, 70, 71, "68:83", b0) CQ_FX_CASE(18, 72, 73, "68:83", b0) CQ_FX_CASE(19, 74, 75, "68:83", b0)
    CQ_FX_CASE(20, 76, 77, "68:83", b0) CQ_FX_CASE(21, 78, 79, "68:83", b0) CQ_FX_CASE(22, 80, 81, "68:83", b0) CQ_FX_CASE(23, 82, 83, "68:83", b0)
    CQ_FX_CASE(24, 84, 85, "84:99", b1) CQ_FX_CASE(25, 86, 87, "84:99", b1) CQ_FX_CASE(26, 88, 89, "84:99", b1) CQ_FX_CASE(27, 90, 91, "84:99", b1)
    CQ_FX_CASE(28, 92, 93, "84:99", b1) CQ_FX_CASE(29, 94, 95, "84:99", b1) CQ_FX_CASE(30, 96, 97, "84:99", b1) CQ_FX_CASE(31, 98, 99, "84:99", b1)
}
#undef CQ_FX_CASE

// table of wave w: flat pairs t = jp * H + i -> {S[2 jp][H w + i], S[2 jp + 1][H w + i]}  (ctm_quad_tabs_kernel)
template <int R, int GI>
__device__ __forceinline__ void cq_fx_consume(const cb_fx_regs& g, const cb_v2f (&p2)[R / 2], cb_v2f (&y2)[R / 4])
{
    constexpr int H = cq_dim<R>::H;
    tmvb_static_for<16>([&](auto tag) {
        constexpr int q = decltype(tag)::value;
        constexpr int t = GI * 16 + q;
        if constexpr (t < cq_dim<R>::NPAIR) cq_fx_fma<(GI & 1) * 16 + q>(g, y2[t % H], p2[t / H]);
    });
}
template <int R, int GI>
__device__ __forceinline__ void cq_fx_pipe(cb_fx_regs& g, const float* S, const cb_v2f (&p2)[R / 2], cb_v2f (&y2)[R / 4])
{
    constexpr int NB = cq_dim<R>::NB, NG = cq_dim<R>::NG;
    if constexpr (GI < NG) {
        if constexpr (GI + 1 < NG) cb_fx_issue<NB, GI + 1>(g, S);
        cq_fx_consume<R, GI>(g, p2, y2);
        if constexpr (GI + 1 < NG) cb_fx_wait<(GI + 1) & 1, cb_fx_nblk<NB, GI + 1>>(g);
        cq_fx_pipe<R, GI + 1>(g, S, p2, y2);
    }
}
// y2[i].x + y2[i].y = sum_j S[j][H w + i] p[j]
template <int R>
__device__ __forceinline__ void cq_matvec(const float* S, const cb_v2f (&p2)[R / 2], cb_v2f (&y2)[R / 4])
{
    constexpr int NB = cq_dim<R>::NB;
#pragma unroll
    for (int i = 0; i < R / 4; ++i) y2[i] = cb_v2f{0.f, 0.f};
    cb_fx_regs g;
    asm volatile("; CBFX_BEGIN");
    cb_fx_issue<NB, 0>(g, S);
    cb_fx_wait<0, cb_fx_nblk<NB, 0>>(g);
    cq_fx_pipe<R, 0>(g, S, p2, y2);
    asm volatile("; CBFX_END");
}

template <int CTRL>
__device__ __forceinline__ float cq_quad_perm(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum / max over the four lanes of a quad, the same bits in all four (a + b == b + a)
__device__ __forceinline__ float cq_quad_sum(float v) { v += cq_quad_perm<0xB1>(v); v += cq_quad_perm<0x4E>(v); return v; }
__device__ __forceinline__ float cq_quad_max(float v) { v = fmaxf(v, cq_quad_perm<0xB1>(v)); v = fmaxf(v, cq_quad_perm<0x4E>(v)); return v; }

template <int NBLK>
__device__ __forceinline__ cb_v16f cq_sload16_sync(const float* tab)
{
    cb_v16f v;
    asm volatile("s_load_dwordx16 %0, %1, 0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(tab));
    return v;
}

// instantiations whose hand-scheduled token loop passes tools/check_vmem_inflight.py with this compiler (the others build the compiler-managed loop;
// -DTMVB_CTM_QASM=0 builds it everywhere)
#ifndef TMVB_CTM_QASM
#define TMVB_CTM_QASM 1
#endif
template <int R> struct cq_asm_loop { static constexpr bool value = TMVB_CTM_QASM != 0; };     // (round 6, with the id words in fixed registers: every instantiation passes)

// An opaque copy of a lane-dependent index.  LDS addresses computed from `lane` itself are invariants of the item / sweep / Newton / CG loops: the compiler
// hoists one address register per access out of all of them (the first build: ~40 of them computed in the kernel's prologue, spilled at once and reloaded
// from scratch -- 80 MB of it across the device, i.e. from HBM -- in front of every use).  A value that passes through a volatile asm inside the loop is not
// an invariant, and the one or two integer operations per address stay where the access is.
#ifdef TMVB_CQ_NOOPQ
__device__ __forceinline__ int cq_opq(int v) { return v; }
#else
__device__ __forceinline__ int cq_opq(int v) { asm volatile("" : "+v"(v)); return v; }
#endif
// fp64 constants as SGPR pairs defined at the point of use (for the same reason: 14 constants of cb_exp_n hoisted into VGPR pairs and spilled)
#ifdef TMVB_CQ_NOK
__device__ __forceinline__ double cq_k(double c) { return c; }
#else
__device__ __forceinline__ double cq_k(double c) { asm volatile("" : "+s"(c)); return c; }
#endif
template <int N>
__device__ __forceinline__ void cq_exp_n(double (&a)[N])
{
#ifdef TMVB_CQ_OLDEXP
    cb_exp_n<N>(a); return;
#endif
    double n[N], q[N];
    const double il2 = cq_k(1.4426950408889634074), l2h = cq_k(-6.93147180369123816490e-01), l2l = cq_k(-1.90821492927058770002e-10);
#pragma unroll
    for (int u = 0; u < N; ++u) { a[u] = fmax(a[u], -745.0); n[u] = __builtin_rint(a[u] * il2); }
#pragma unroll
    for (int u = 0; u < N; ++u) a[u] = fma(n[u], l2h, a[u]);
    const double c11 = cq_k(2.50521083854417187751e-08), c10 = cq_k(2.75573192239858906526e-07);
#pragma unroll
    for (int u = 0; u < N; ++u) { a[u] = fma(n[u], l2l, a[u]); q[u] = fma(c11, a[u], c10); }
    const double c[9] = {2.75573192239858906526e-06, 2.48015873015873015873e-05, 1.98412698412698412698e-04, 1.38888888888888888889e-03,
                         8.33333333333333333333e-03, 4.16666666666666666667e-02, 1.66666666666666666667e-01, 0.5, 1.0};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double ck = cq_k(c[k]);
#pragma unroll
        for (int u = 0; u < N; ++u) q[u] = fma(q[u], a[u], ck);
    }
#pragma unroll
    for (int u = 0; u < N; ++u) a[u] = __builtin_ldexp(fma(q[u], a[u], 1.0), (int)n[u]);
}

// ---- vector-memory loads of the token loop as inline asm (see the loop): the destination is a read-write operand, so the loop-carried value keeps its
// registers; the compiler inserts no waits for these loads -- cq_vmwait<N> is the hand-placed s_waitcnt vmcnt(N) and names the first register block that
// becomes valid behind it, cq_vmwait_def names the others
template <int OFF>
__device__ __forceinline__ void cq_gload16(cb_v4f& d, unsigned voff, const __attribute__((address_space(1))) char* sbase)
{
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
}
// the same for the LAST chunk slot of a row whose chunk count is not a multiple of four (KP = 52: 13 chunks, slot 3 holds chunk 12 in lane 0 of each quad and
// pad in the other three): only the lanes of `mask` issue their 16 bytes -- the others keep whatever their registers held, which only ever meets e = 0
#ifndef TMVB_CTM_QMASKED
#define TMVB_CTM_QMASKED 1
#endif
template <int OFF>
__device__ __forceinline__ void cq_gload16_masked(cb_v4f& d, unsigned voff, const __attribute__((address_space(1))) char* sbase, unsigned long long mask)
{
    unsigned long long sv;
    asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %5\n\tglobal_load_dwordx4 %0, %2, %3 offset:%4\n\ts_mov_b64 exec, %1"
                 : "+v"(d), "=&s"(sv) : "v"(voff), "s"(sbase), "n"(OFF), "s"(mask) : "scc");
}
// The id / count words live in FIXED physical registers from their load to the wait that lands them (WHICH = 0 .. 3 -> v244 .. v247): as plain "+v"
// operands the compiler twice chose to copy the (not yet landed) word out of its register, use the register as an address temporary and copy the stale
// word back in front of the wait -- run-to-run different results (tools/check_vmem_inflight.py reports it; build() refuses such a library).  With the
// register named in the constraint of the load, of the wait and of nothing else, the value has nowhere else to be.
template <int WHICH>
__device__ __forceinline__ void cq_gload4(int& d, unsigned voff, const __attribute__((address_space(1))) int32_t* sbase)
{
    if constexpr (WHICH == 0) asm volatile("global_load_dword %0, %1, %2" : "={v244}"(d) : "v"(voff), "s"(sbase));
    else if constexpr (WHICH == 1) asm volatile("global_load_dword %0, %1, %2" : "={v245}"(d) : "v"(voff), "s"(sbase));
    else if constexpr (WHICH == 2) asm volatile("global_load_dword %0, %1, %2" : "={v246}"(d) : "v"(voff), "s"(sbase));
    else asm volatile("global_load_dword %0, %1, %2" : "={v247}"(d) : "v"(voff), "s"(sbase));
}
template <int WHICH>
__device__ __forceinline__ void cq_vmwait_id(int& d)
{
    if constexpr (WHICH == 0) asm volatile("" : "+{v244}"(d));
    else if constexpr (WHICH == 1) asm volatile("" : "+{v245}"(d));
    else if constexpr (WHICH == 2) asm volatile("" : "+{v246}"(d));
    else asm volatile("" : "+{v247}"(d));
}
template <int N, typename T>
__device__ __forceinline__ void cq_vmwait(T& d) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "n"(N)); }
template <typename T>
__device__ __forceinline__ void cq_vmwait_def(T& d) { asm volatile("" : "+v"(d)); }

// ---- the kernel: workgroup = 4 waves, item = 64 documents of doc_order (the same queue as ctm_estep_batch_kernel)
// PROF (TMVB_CTM_QPROF=1): wave 0 of every workgroup adds its shader cycles per phase to tb.cg_iters[3..11] (the slots of tmvb_ctm_solver_stats:
// token incl. the wait for the slowest wave, logzeta, vsq, gradient assembly, CG, gradient mat-vec incl. its exchange, exit test; [11] the item)
template <int R, bool PROF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TMVB_CTM_QWAVES, TMVB_CTM_QWAVES))) void ctm_estep_quad_kernel(CtmBatchArgs a)
{
    using DM = cq_dim<R>;
    constexpr int H = DM::H, JP = DM::JP, NS = DM::NS, LPR = DM::LPR;
    const int K = a.p.K, p_viter = a.p.viter, p_niter = a.p.niter, p_debug = a.p.debug;
    const double p_ntol = a.p.ntol, p_vtol = a.p.vtol;
    const float cg_tol2 = a.tb.cg_tol2, cg_abs2 = a.tb.cg_abs2;
    const int cg_maxit = (a.p.debug & 8) ? 0 : a.tb.cg_maxit;              // TMVB_DEBUG_FLAGS & 8: no CG trips (timing experiments only)
    const int64_t M = a.M;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lam_l = lds;                                    // [R][64]  (float)lambda, for the token lanes
    float* xv0 = lds + R * 64;                             // [2][R][64] vector exchange (alternating: a buffer is rewritten two barriers after its last read)
    // (phi * counts) and vsq in [topic][document] rows of their own: written / read by the wave that owns the topic (vsq) or handed from the token lanes to it
    // (phi * counts); as 13 + 13 registers per lane across the Newton loop they -- and the doubles the compiler derived from them -- were what it spilled
    float* ph_l = lds + 3 * R * 64;                        // [R][64]
    float* vs_l = lds + 4 * R * 64;                        // [R][64]
    float* xsf = lds + 5 * R * 64;                         // scalar exchange, fp32
    double* xsd = (double*)(xsf + DM::XSF);                // scalar exchange, fp64
    int* dinfo = (int*)(xsd + DM::XSD);                    // [64] document, [64] first token, [64] tokens, [64] active, [64] C_d
    int* item_l = dinfo + 5 * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gt0 = H * w;                                 // first topic of this wave
    int vpar = 0, spar = 0;                                // parities of the exchanges (uniform, the same in all four waves)
    const float* Sq_w = a.tb.Sq + (size_t)w * (DM::NB * 16);
    const float* sdq_w = a.tb.sdq + 16 * w;
    const float* muq_w = a.tb.muq + 16 * w;

    // scalar exchanges: four partial sums per document, added in wave order by every wave (the same bits in all four)
    for (;;) {
        if (tid == 0) item_l[0] = (int)atomicAdd(CB_KARG(unsigned*, tb.next_item), 1u);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(item_l[0]);
        if (item >= CB_KARG32(tb.n_items)) break;
        const int64_t slot = (int64_t)item * 64 + lane;
        const bool valid = slot < M;
        const int d = CB_KARG(const int32_t*, p.doc_order)[valid ? slot : M - 1];
        int64_t off;
        int N;
        { const int64_t* doc_ptr = CB_KARG(const int64_t*, p.doc_ptr); off = doc_ptr[d]; N = valid ? (int)(doc_ptr[d + 1] - off) : 0; }
        int ln = cq_opq(lane);
        if (w == 0) { dinfo[ln] = d; dinfo[64 + ln] = (int)off; dinfo[128 + ln] = N; }
        __syncthreads();
        // token lanes: lane (ds, qd) = quarter qd of document 16 w + ds
        const int tj = 16 * w + (lane >> 2), qd = lane & 3;
        const int td = dinfo[tj], tN = dinfo[128 + tj];
        const unsigned toff4 = (unsigned)dinfo[64 + tj] * 4u;
        const int Nmax16 = (p_debug & 4) ? 0 : wave_max_i(tN);                 // TMVB_DEBUG_FLAGS & 4: no token walk (timing experiments only)
        {
            // C_d = sum of counts (src/CTM.jl:33): integers, exact in fp32 in any order
            typedef const __attribute__((address_space(1))) int32_t* gint_p;
            typedef const __attribute__((address_space(1))) char* gchar_p;
            gint_p counts = (gint_p)CB_KARG(const int32_t*, p.counts);
            float cl = 0.0f;
            for (int n = qd; n < Nmax16; n += 4) {
                const int cv = *(gint_p)((gchar_p)counts + (toff4 + 4u * (unsigned)max(min(n, tN - 1), 0)));
                cl += (n < tN) ? (float)cv : 0.0f;
            }
            cl = cq_quad_sum(cl);
            if (qd == 0) ((float*)dinfo)[256 + tj] = cl;
        }

        // ---- Newton lanes: own topics of document `lane`
        double lam[H];
#define VS(i) vs_l[(gt0 + (i)) * 64 + ln]
        double lz;
        {
            const float* lrow = CB_KARG(const float*, p.lambda) + (int64_t)d * K;
            const float* vrow = CB_KARG(const float*, p.vsq) + (int64_t)d * K;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const bool on = gt0 + i < K;
                const int ic = min(gt0 + i, K - 1);
                const float lv = lrow[ic], vv = vrow[ic];
                lam[i] = on ? (double)lv : -1.0e30;
                VS(i) = on ? vv : 1.0f;
                lam_l[(gt0 + i) * 64 + ln] = on ? lv : -1.0e30f;
            }
            lz = (double)CB_KARG(const float*, p.logzeta)[d];
        }
        __syncthreads();
        const double Cd = (double)((float*)dinfo)[256 + ln];
        bool active = valid && p_viter > 0;
        int sweeps = 0;
        unsigned nsteps = 0, ncg = 0, ntrip = 0;
        long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const long long t_start = PROF ? __builtin_readcyclecounter() : 0;
        long long t_mark = t_start;
        auto lap = [&](int which) {
            if constexpr (PROF) { const long long now = __builtin_readcyclecounter(); cyc[which] += now - t_mark; t_mark = now; }
        };
        // PROF: shader cycles wave 0 spends waiting in the workgroup barriers of the sweeps (slot 7 of the phase counters)
#define CQ_SYNC() do { if constexpr (PROF) { const long long tb_ = __builtin_readcyclecounter(); __syncthreads(); cyc[7] += __builtin_readcyclecounter() - tb_; } else __syncthreads(); } while (0)


        for (int v = 0; v < p_viter; ++v) {
            if (!__any(active)) break;
            if (active) ++sweeps;
            // (lambda_old of this sweep = the fp32 lambda the token lanes read from lam_l: it stays there until the exit test, which stores it -- and the factor
            //  e = exp(lambda_old - max) of the statistics pass -- once, for the document's LAST executed sweep, instead of every sweep)
            ln = cq_opq(lane);
            if (w == 0) dinfo[192 + ln] = active ? 1 : 0;
            CQ_SYNC();                                                         // (A) lam_l, active flags
            // ---- update_phi!  src/CTM.jl:175-178 in linear space, (phi * counts)_i = e_i sum_n w_n beta[i, t_n]
            {
                float* xw = ph_l;
                const int tjo = cq_opq(tj);
                const bool tact = dinfo[192 + tjo] != 0;
                cb_v4f ec[NS], acc[NS];
                float lmx = -INFINITY;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int c = qd + 4 * s;
                    const bool ok = (4 * s + 3 < LPR) || (c < LPR);
                    const int cc = ok ? c : 0;
                    ec[s] = cb_v4f{lam_l[(4 * cc) * 64 + tjo], lam_l[(4 * cc + 1) * 64 + tjo], lam_l[(4 * cc + 2) * 64 + tjo], lam_l[(4 * cc + 3) * 64 + tjo]};
                    if (ok) lmx = fmaxf(lmx, fmaxf(fmaxf(ec[s].x, ec[s].y), fmaxf(ec[s].z, ec[s].w)));
                    acc[s] = cb_v4f{0.f, 0.f, 0.f, 0.f};
                }
                lmx = cq_quad_max(lmx);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int c = qd + 4 * s;
                    const bool ok = (4 * s + 3 < LPR) || (c < LPR);
                    const cb_v4f e4 = cb_v4f{expf(ec[s].x - lmx), expf(ec[s].y - lmx), expf(ec[s].z - lmx), expf(ec[s].w - lmx)};     // pads: expf(-1e30) = 0
                    ec[s] = ok ? e4 : cb_v4f{0.f, 0.f, 0.f, 0.f};
                }
                typedef const __attribute__((address_space(1))) float* gfloat_p;
                typedef const __attribute__((address_space(1))) int32_t* gint_p;
                typedef const __attribute__((address_space(1))) char* gchar_p;
                gchar_p beta = (gchar_p)CB_KARG(const float*, p.beta);                // the padded copy: rows of 16 NS floats, pads 0
                gint_p terms = (gint_p)CB_KARG(const int32_t*, p.terms);
                gint_p counts = (gint_p)CB_KARG(const int32_t*, p.counts);
#ifndef TMVB_CTM_QCH
#define TMVB_CTM_QCH 4
#endif
#ifndef TMVB_CTM_QMASK
#define TMVB_CTM_QMASK 0
#endif
                constexpr int CH = 4;                                                 // token steps per round = lanes of a quad
                const unsigned qb = (unsigned)qd * 16u;
                // A round = four token steps; lane q of a quad fetches the id / count words of step q of a round (the steps read them through quad_perm
                // broadcasts).  EVERY vector-memory instruction of the loop is inline asm with hand-counted waits (cq_gload*, cq_vmwait): written as
                // C++ loads, the loop-carried row registers were copied at the loop end (v_mov of 48 registers behind s_waitcnt vmcnt(0): the whole
                // pipeline drained once per round, 275 cycles per gather instruction and wave).  Order of the stream (it returns in order):
                //   round k:  wait until <= 12 loads are out -> rows(k, 0) and ids(k + 1) are in;  rotate the id words;  ids(k + 2) [2 loads];
                //             step u = 0..3:  (u > 0: wait until <= 14 are out -> rows(k, u) are in;)  compute step u in place;  issue rows(k + 1, u)
                //             [4 loads] into the same registers behind their last use
                // (at KP = 52, four loads per row; younger than rows(k, u): rows(k, u+1..3) = 12 - 4u and, for u > 0, ids(k + 2) = 2 and rows(k + 1, 0..u-1) = 4u:
                // 12, 14, 14, 14 -- in general 3 NS and 3 NS + 2).
                // The registers of a row are asm operands of its loads, of the wait in front of its use and of nothing in between;
                // tools/check_smem_inflight.py --vmem scans the loop for anything else that touches them.
                auto id_off = [&](int n0) { return toff4 + 4u * (unsigned)max(min(n0 + qd, tN - 1), 0); };
                cb_v4f buf[CH][NS];
                int tq = 0, cq = 0, tn = 0, cn = 0, t2 = 0, c2 = 0;
                if constexpr (cq_asm_loop<R>::value) {
                    // lanes of a quad that hold a chunk of the row in the last slot: qd < LPR mod 4 (all four when LPR is a multiple of four: never masked then)
                    constexpr unsigned long long qmask = 0x1111111111111111ull * ((1u << ((LPR & 3) ? (LPR & 3) : 4)) - 1u);
                    // (the lanes a masked load never writes must hold finite values: they meet e = 0, and 0 * NaN is not 0 -- zeroed once, never written again)
                    if constexpr (TMVB_CTM_QMASKED && (LPR & 3)) {
#pragma unroll
                        for (int u = 0; u < CH; ++u) buf[u][NS - 1] = cb_v4f{0.f, 0.f, 0.f, 0.f};
                    }
                    asm volatile("; CQVM_BEGIN");
                    cq_gload4<2>(tn, id_off(0), terms); cq_gload4<3>(cn, id_off(0), counts);
                    asm volatile("s_waitcnt vmcnt(0)"); cq_vmwait_id<2>(tn); cq_vmwait_id<3>(cn);
                    cq_gload4<0>(t2, id_off(CH), terms); cq_gload4<1>(c2, id_off(CH), counts);
                    tmvb_static_for<CH>([&](auto tag) {
                        constexpr int u = decltype(tag)::value;
                        const int tb_ = __builtin_amdgcn_update_dpp(0, tn, u * 0x55, 0xF, 0xF, true);
                        const unsigned ab = __umul24((unsigned)((u < tN) ? tb_ : 0), DM::ROWB) + qb;      // row ids are < 2^24; row 0 past the document's end
                        tmvb_static_for<NS>([&](auto st) { constexpr int s = decltype(st)::value;
                            if constexpr (TMVB_CTM_QMASKED && 4 * s + 3 >= LPR) cq_gload16_masked<64 * s>(buf[u][s], ab, beta, qmask); else cq_gload16<64 * s>(buf[u][s], ab, beta); });
                    });
                    for (int n0 = 0; n0 < Nmax16; n0 += CH) {
                        tmvb_static_for<CH>([&](auto tag) {
                            constexpr int u = decltype(tag)::value;
                            cq_vmwait<(u == 0) ? 3 * NS : 3 * NS + 2>(buf[u][0]);
                            tmvb_static_for<NS>([&](auto st) { constexpr int s = decltype(st)::value; if constexpr (s > 0) cq_vmwait_def(buf[u][s]); });
                            if constexpr (u == 0) {
                                // behind the round's first wait the words of round k + 1 are in: rotate, then fetch round k + 2's into the freed pair
                                cq_vmwait_id<0>(t2); cq_vmwait_id<1>(c2);
                                tq = tn; cq = cn; tn = t2; cn = c2;
                                cq_gload4<0>(t2, id_off(n0 + 2 * CH), terms); cq_gload4<1>(c2, id_off(n0 + 2 * CH), counts);
                            }
                            cb_v2f s0 = cb_v2f{0.f, 0.f}, s1 = cb_v2f{0.f, 0.f};
    #pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                s0 = __builtin_elementwise_fma(cb_v2f{buf[u][s].x, buf[u][s].y}, cb_v2f{ec[s].x, ec[s].y}, s0);
                                s1 = __builtin_elementwise_fma(cb_v2f{buf[u][s].z, buf[u][s].w}, cb_v2f{ec[s].z, ec[s].w}, s1);
                            }
                            const cb_v2f ss = s0 + s1;
                            const float sn = cq_quad_sum(ss.x + ss.y);
                            // a lane past its document's end reads row 0 with count 0; beta[:, 0] may be all zero (CTM's phi has no epsilon)
                            const int cb_ = __builtin_amdgcn_update_dpp(0, cq, u * 0x55, 0xF, 0xF, true);
                            const float cf = (n0 + u < tN) ? (float)cb_ : 0.0f;
                            const float wt = (cf > 0.0f) ? cf * __builtin_amdgcn_rcpf(sn) : 0.0f;       // 1 ulp; the one-wave kernel divides
                            const cb_v2f w2 = cb_v2f{wt, wt};
    #pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                const cb_v2f lo = __builtin_elementwise_fma(w2, cb_v2f{buf[u][s].x, buf[u][s].y}, cb_v2f{acc[s].x, acc[s].y});
                                const cb_v2f hi = __builtin_elementwise_fma(w2, cb_v2f{buf[u][s].z, buf[u][s].w}, cb_v2f{acc[s].z, acc[s].w});
                                acc[s] = cb_v4f{lo.x, lo.y, hi.x, hi.y};
                            }
                            const int tb_ = __builtin_amdgcn_update_dpp(0, tn, u * 0x55, 0xF, 0xF, true);
                            const unsigned ab = __umul24((unsigned)((n0 + CH + u < tN) ? tb_ : 0), DM::ROWB) + qb;
                            __builtin_amdgcn_sched_barrier(0);
                            tmvb_static_for<NS>([&](auto st) { constexpr int s = decltype(st)::value;                                                     // step n0 + CH + u
                                if constexpr (TMVB_CTM_QMASKED && 4 * s + 3 >= LPR) cq_gload16_masked<64 * s>(buf[u][s], ab, beta, qmask); else cq_gload16<64 * s>(buf[u][s], ab, beta); });
                        });
                    }
                    cq_vmwait<0>(buf[0][0]);                                              // the rows fetched past the end land before their registers are reused
                    tmvb_static_for<CH>([&](auto tag) { constexpr int u = decltype(tag)::value;
                        tmvb_static_for<NS>([&](auto st) { constexpr int s = decltype(st)::value; cq_vmwait_def(buf[u][s]); }); });
                    cq_vmwait_id<0>(t2); cq_vmwait_id<1>(c2);
                    asm volatile("; CQVM_END");
                } else {
                    // the same loop with compiler-managed loads and waits (correct by construction; the instantiations whose hand-scheduled form
                    // does not pass tools/check_vmem_inflight.py build this one)
                    auto ld = [&](int n0, int& t, int& c) { const unsigned ix = id_off(n0); t = *(gint_p)((gchar_p)terms + ix); c = *(gint_p)((gchar_p)counts + ix); };
                    auto issue = [&](int t, bool in, cb_v4f (&b)[NS]) {
                        const unsigned ab = __umul24((unsigned)(in ? t : 0), DM::ROWB) + qb;
#pragma unroll
                        for (int s = 0; s < NS; ++s) b[s] = *(const __attribute__((address_space(1))) cb_v4f*)(beta + (ab + 64u * s));
                    };
                    ld(0, tq, cq);
                    tmvb_static_for<CH>([&](auto tag) { constexpr int u = decltype(tag)::value; issue(__builtin_amdgcn_update_dpp(0, tq, u * 0x55, 0xF, 0xF, true), u < tN, buf[u]); });
                    ld(CH, tn, cn);
                    for (int n0 = 0; n0 < Nmax16; n0 += CH) {
                        ld(n0 + 2 * CH, t2, c2);
                        tmvb_static_for<CH>([&](auto tag) {
                            constexpr int u = decltype(tag)::value;
                            cb_v2f s0 = cb_v2f{0.f, 0.f}, s1 = cb_v2f{0.f, 0.f};
#pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                s0 = __builtin_elementwise_fma(cb_v2f{buf[u][s].x, buf[u][s].y}, cb_v2f{ec[s].x, ec[s].y}, s0);
                                s1 = __builtin_elementwise_fma(cb_v2f{buf[u][s].z, buf[u][s].w}, cb_v2f{ec[s].z, ec[s].w}, s1);
                            }
                            const cb_v2f ss = s0 + s1;
                            const float sn = cq_quad_sum(ss.x + ss.y);
                            const int cb_ = __builtin_amdgcn_update_dpp(0, cq, u * 0x55, 0xF, 0xF, true);
                            const float cf = (n0 + u < tN) ? (float)cb_ : 0.0f;
                            const float wt = (cf > 0.0f) ? cf * __builtin_amdgcn_rcpf(sn) : 0.0f;
                            const cb_v2f w2 = cb_v2f{wt, wt};
#pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                const cb_v2f lo = __builtin_elementwise_fma(w2, cb_v2f{buf[u][s].x, buf[u][s].y}, cb_v2f{acc[s].x, acc[s].y});
                                const cb_v2f hi = __builtin_elementwise_fma(w2, cb_v2f{buf[u][s].z, buf[u][s].w}, cb_v2f{acc[s].z, acc[s].w});
                                acc[s] = cb_v4f{lo.x, lo.y, hi.x, hi.y};
                            }
                            const int tb_ = __builtin_amdgcn_update_dpp(0, tn, u * 0x55, 0xF, 0xF, true);
                            issue(tb_, n0 + CH + u < tN, buf[u]);
                        });
                        tq = tn; cq = cn; tn = t2; cn = c2;
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int c = qd + 4 * s;
                    const bool ok = (4 * s + 3 < LPR) || (c < LPR);
                    if (ok) {
                        xw[(4 * c) * 64 + tjo] = ec[s].x * acc[s].x; xw[(4 * c + 1) * 64 + tjo] = ec[s].y * acc[s].y;
                        xw[(4 * c + 2) * 64 + tjo] = ec[s].z * acc[s].z; xw[(4 * c + 3) * 64 + tjo] = ec[s].w * acc[s].w;
                    }
                }
                CQ_SYNC();                                                     // (B) phi * counts
            }
            lap(0);
            ln = cq_opq(lane);
#define PHIC(i) ph_l[(gt0 + (i)) * 64 + ln]
            // ---- update_logzeta!  src/CTM.jl:169-171
            {
                double mp = -INFINITY;
#pragma unroll
                for (int i = 0; i < H; ++i) mp = fmax(mp, lam[i] + 0.5 * (double)VS(i));
                double* b = xsd + spar * (4 * 64);
                b[w * 64 + ln] = mp;
                CQ_SYNC();
                const double m = fmax(fmax(b[ln], b[64 + ln]), fmax(b[128 + ln], b[192 + ln]));
                spar ^= 1;
                double sp = 0.0;
                tmvb_static_for<(H + 3) / 4>([&](auto tag) {
                    constexpr int i0 = 4 * decltype(tag)::value;
                    constexpr int NU = (H - i0 < 4) ? H - i0 : 4;
                    double ax[NU];
#pragma unroll
                    for (int u = 0; u < NU; ++u) ax[u] = lam[i0 + u] + 0.5 * (double)VS(i0 + u) - m;
                    cq_exp_n<NU>(ax);
#pragma unroll
                    for (int u = 0; u < NU; ++u) sp += ax[u];                         // pads add exp(-1e30 - m) = +0
                });
                double* b2 = xsd + spar * (4 * 64);
                b2[w * 64 + ln] = sp;
                CQ_SYNC();
                const double s = ((b2[ln] + b2[64 + ln]) + b2[128 + ln]) + b2[192 + ln];
                spar ^= 1;
                // log(s), 1 <= s <= KP: the fp32 logarithm and one Newton step of exp(y) = s in fp64 (error ~1e-14; the library's fp64 log brings six
                // polynomial constants that the compiler hoists into VGPR pairs in the kernel's prologue and spills)
#ifdef TMVB_CQ_OLDLOG
                if (active) lz = m + log(s);
#else
                {
                    const double y0 = (double)__logf((float)s);
                    double e1[1] = {-y0};
                    cq_exp_n<1>(e1);
                    if (active) lz = m + (y0 + (s * e1[0] - 1.0));
                }
#endif
            }
            lap(1);
            // ---- update_vsq!  src/CTM.jl:146-165 (one scalar Newton iteration per topic; four topics per loop)
            if (!(p_debug & 1)) {
                tmvb_static_for<(H + 3) / 4>([&](auto tag) {
                    constexpr int i0 = 4 * decltype(tag)::value;
                    constexpr int NU = (H - i0 < 4) ? H - i0 : 4;
                    if (gt0 + i0 >= K) return;
                    double vv[NU], isd[NU], lm[NU];
                    bool act[NU];
                    const cb_v4f sd4 = cb_sload4_sync<i0>(sdq_w);
#pragma unroll
                    for (int u = 0; u < NU; ++u) { vv[u] = (double)VS(i0 + u); lm[u] = lam[i0 + u]; act[u] = active && (gt0 + i0 + u < K); isd[u] = (double)sd4[u]; }
                    for (int t = 0; t < p_niter; ++t) {
                        bool anyact = false;
#pragma unroll
                        for (int u = 0; u < NU; ++u) anyact = anyact || act[u];
                        if (!__any(anyact)) break;
                        double ex[NU], rv[NU], den[NU], ihd[NU], grad[NU], pp[NU], rho[NU];
#pragma unroll
                        for (int u = 0; u < NU; ++u) ex[u] = lm[u] + 0.5 * vv[u] - lz;
                        cq_exp_n<NU>(ex);
                        cb_rcp_n<NU>(vv, rv);
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            grad[u] = -0.5 * (isd[u] + Cd * ex[u] - rv[u]);                        // :150
                            den[u] = 0.25 * Cd * ex[u] + 0.5 * rv[u] * rv[u];
                        }
                        cb_rcp_n<NU>(den, ihd);                                                     // -1 / den = inverse Hessian, :151
                        bool shrink = false;
#pragma unroll
                        for (int u = 0; u < NU; ++u) { pp[u] = -ihd[u] * grad[u]; rho[u] = 1.0; shrink = shrink || (act[u] && vv[u] - pp[u] <= 0.0); }
                        while (__any(shrink)) {                                                    // :154
                            shrink = false;
#pragma unroll
                            for (int u = 0; u < NU; ++u) {
                                if (act[u] && vv[u] - rho[u] * pp[u] <= 0.0) rho[u] *= 0.5;
                                shrink = shrink || (act[u] && vv[u] - rho[u] * pp[u] <= 0.0);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            if (act[u]) vv[u] -= rho[u] * pp[u];
                            if (rho[u] * fabs(grad[u]) < p_ntol) act[u] = false;                    // :159
                        }
                    }
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        if (active && gt0 + i0 + u < K) vv[u] += TMVB_EPS_D;                        // :164
                        VS(i0 + u) = (float)vv[u];
                    }
                });
            }
            lap(2);
            // ---- update_lambda!  src/CTM.jl:129-142
            {
                bool newt = active;
                for (int t = 0; t < p_niter; ++t) {
                    if (!__any(newt)) break;
                    ++ntrip;
                    if (newt) ++nsteps;
                    ln = cq_opq(lane);
                    cb_v2f p2[JP];
                    {
                        // invsigma (mu - lambda), :134, in fp32 (tmvb_ctm_batch.h: what that costs and why it is enough)
                        float* xw = xv0 + vpar * (R * 64);
                        const cb_v16f mb = cq_sload16_sync<1>(muq_w);
#pragma unroll
                        for (int i = 0; i < H; ++i) xw[(gt0 + i) * 64 + ln] = (float)((double)mb[i] - lam[i]);
                        CQ_SYNC();
                        vpar ^= 1;
#pragma unroll
                        for (int jp = 0; jp < JP; ++jp) p2[jp] = cb_v2f{xw[(2 * jp) * 64 + ln], xw[(2 * jp + 1) * 64 + ln]};
                    }
                    float g[H], D[H], dinv[H];
                    double gn2p = 0.0;
                    {
                        cb_v2f y2[H];
                        cq_matvec<R>(Sq_w, p2, y2);
                        lap(5);
                        tmvb_static_for<(H + 3) / 4>([&](auto tag) {
                            constexpr int i0 = 4 * decltype(tag)::value;
                            constexpr int NU = (H - i0 < 4) ? H - i0 : 4;
                            double ex[NU];
                            const cb_v4f sdg = cb_sload4_sync<i0>(sdq_w);
#pragma unroll
                            for (int u = 0; u < NU; ++u) ex[u] = lam[i0 + u] + 0.5 * (double)VS(i0 + u) - lz;
                            cq_exp_n<NU>(ex);
#pragma unroll
                            for (int u = 0; u < NU; ++u) {
                                const int i = i0 + u;
                                const bool on = gt0 + i < K;
                                const double mv = (double)(y2[i].x + y2[i].y);
                                const double gd = on ? (mv + (double)PHIC(i) - Cd * ex[u]) : 0.0;       // :134
                                gn2p = fma(gd, gd, gn2p);
                                const float dval = on ? (float)(Cd * ex[u]) : 1.0f;                     // pad rows: unit rows
                                g[i] = (float)gd; D[i] = dval; dinv[i] = 1.0f / (sdg[u] + dval);        // 1 / -H_ii
                            }
                        });
                    }
                    // Jacobi-preconditioned CG for (S + Diag(D)) x = g, one system per lane, rows split over the four waves
                    float r[H], pv[H], x[H];
                    float gg, rz, thr;
                    double gn2;
                    {
                        float ggp = 0.0f, rzp = 0.0f;
                        float* xw = xv0 + vpar * (R * 64);
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            r[i] = g[i]; pv[i] = g[i] * dinv[i]; x[i] = 0.0f;
                            ggp = fmaf(g[i], g[i], ggp); rzp = fmaf(r[i], pv[i], rzp);
                            xw[(gt0 + i) * 64 + ln] = pv[i];
                        }
                        float* bf = xsf + spar * (3 * 4 * 64);
                        double* bd = xsd + spar * (4 * 64);
                        bf[w * 64 + ln] = ggp; bf[(4 + w) * 64 + ln] = rzp; bd[w * 64 + ln] = gn2p;
                        CQ_SYNC();
                        vpar ^= 1; spar ^= 1;
                        gg = ((bf[ln] + bf[64 + ln]) + bf[128 + ln]) + bf[192 + ln];
                        rz = ((bf[256 + ln] + bf[320 + ln]) + bf[384 + ln]) + bf[448 + ln];
                        gn2 = ((bd[ln] + bd[64 + ln]) + bd[128 + ln]) + bd[192 + ln];
#pragma unroll
                        for (int jp = 0; jp < JP; ++jp) p2[jp] = cb_v2f{xw[(2 * jp) * 64 + ln], xw[(2 * jp + 1) * 64 + ln]};
                    }
                    lap(3);
                    thr = fmaxf(cg_tol2 * gg, cg_abs2);
                    bool live = newt && gg > thr;
                    int trips = 0;
                    while (trips < cg_maxit && __any(live)) {
                        ++trips;
                        ln = cq_opq(lane);
                        float y[H];
                        float pHp;
                        {
                            cb_v2f y2[H];
                            cq_matvec<R>(Sq_w, p2, y2);
                            float pp = 0.0f;
#pragma unroll
                            for (int i = 0; i < H; ++i) { y[i] = fmaf(D[i], pv[i], y2[i].x + y2[i].y); pp = fmaf(pv[i], y[i], pp); }
                            float* bf = xsf + spar * (3 * 4 * 64);
                            bf[w * 64 + ln] = pp;
                            CQ_SYNC();
                            spar ^= 1;
                            pHp = ((bf[ln] + bf[64 + ln]) + bf[128 + ln]) + bf[192 + ln];
                        }
                        const float alpha = (live && pHp > 0.0f) ? rz / pHp : 0.0f;
                        float rrp = 0.0f, rzp = 0.0f;
                        float* xw = xv0 + vpar * (R * 64);
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            x[i] = fmaf(alpha, pv[i], x[i]);
                            r[i] = fmaf(-alpha, y[i], r[i]);
                            y[i] = r[i] * dinv[i];                                   // z
                            rrp = fmaf(r[i], r[i], rrp); rzp = fmaf(r[i], y[i], rzp);
                            xw[(gt0 + i) * 64 + ln] = y[i];
                        }
                        float* bf = xsf + spar * (3 * 4 * 64);
                        bf[w * 64 + ln] = rrp; bf[(4 + w) * 64 + ln] = rzp;
                        CQ_SYNC();
                        vpar ^= 1; spar ^= 1;
                        const float rr = ((bf[ln] + bf[64 + ln]) + bf[128 + ln]) + bf[192 + ln];
                        const float rz_new = ((bf[256 + ln] + bf[320 + ln]) + bf[384 + ln]) + bf[448 + ln];
                        if (rr <= thr) live = false;
                        const float beta = (live && rz > 0.0f) ? rz_new / rz : 0.0f;
                        rz = rz_new;
                        const cb_v2f b2 = cb_v2f{beta, beta};
#pragma unroll
                        for (int i = 0; i < H; ++i) pv[i] = fmaf(beta, pv[i], y[i]);
#pragma unroll
                        for (int jp = 0; jp < JP; ++jp)
                            p2[jp] = __builtin_elementwise_fma(b2, p2[jp], cb_v2f{xw[(2 * jp) * 64 + ln], xw[(2 * jp + 1) * 64 + ln]});
                    }
                    ncg += (unsigned)trips;
                    lap(4);
                    if (newt) {
#pragma unroll
                        for (int i = 0; i < H; ++i) lam[i] += (double)x[i];                         // :136
                    }
                    if (sqrt(gn2) < p_ntol) newt = false;                                           // :138
                }
            }
            // ---- exit test (:200), the decomposed update_elbo!'s sum_i (phi counts)_i (lambda_i - lambda_old_i), and lambda for the next token phase
            {
                ln = cq_opq(lane);
                float d2p = 0.0f, pdp = 0.0f, lmp = -INFINITY;
                float lo[H];
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    const bool on = gt0 + i < K;
                    lo[i] = lam_l[(gt0 + i) * 64 + ln];                                             // lambda_old (pads: -1e30)
                    lmp = fmaxf(lmp, lo[i]);
                    const float df = on ? (float)(lam[i] - (double)lo[i]) : 0.0f;
                    d2p = fmaf(df, df, d2p); pdp = fmaf(PHIC(i), df, pdp);
                    lam_l[(gt0 + i) * 64 + ln] = on ? (float)lam[i] : -1.0e30f;
                }
                float* bf = xsf + spar * (3 * 4 * 64);
                bf[w * 64 + ln] = d2p; bf[(4 + w) * 64 + ln] = pdp; bf[(8 + w) * 64 + ln] = lmp;
                CQ_SYNC();
                spar ^= 1;
                const float dist2 = ((bf[ln] + bf[64 + ln]) + bf[128 + ln]) + bf[192 + ln];
                const float pdot = ((bf[256 + ln] + bf[320 + ln]) + bf[384 + ln]) + bf[448 + ln];
                const float lmx = fmaxf(fmaxf(bf[512 + ln], bf[576 + ln]), fmaxf(bf[640 + ln], bf[704 + ln]));
                const bool was = active;
                if (active) {
                    if (w == 0) { float* pd_out = CB_KARG(float*, p.pdot); if (pd_out) pd_out[d] = pdot; }
                    if (sqrtf(dist2) < (float)p_vtol) active = false;
                }
                // the document's last executed sweep: lambda_old and E = exp(lambda_old - max) (the factor of the last phi, for the statistics pass) to memory
                const bool fin = was && (!active || v == p_viter - 1);
                if (__any(fin)) {
                    if (fin) {
                        float* lam_old_out = CB_KARG(float*, p.lambda_old) + (int64_t)d * K;
                        float* E = CB_KARG(float*, p.E) + (int64_t)d * R;
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            if (gt0 + i < K) lam_old_out[gt0 + i] = lo[i];
                            E[gt0 + i] = expf(lo[i] - lmx);                                         // pads: expf(-1e30) = 0
                        }
                    }
                }
            }
            lap(6);
        }

        if (valid) {
            if (sweeps > 0) {
                float* lam_out = CB_KARG(float*, p.lambda) + (int64_t)d * K;
                float* vsq_out = CB_KARG(float*, p.vsq) + (int64_t)d * K;
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    if (gt0 + i < K) { lam_out[gt0 + i] = (float)lam[i]; vsq_out[gt0 + i] = VS(i); }
                }
                if (w == 0) CB_KARG(float*, p.logzeta)[d] = (float)lz;
            } else {
                float* E = CB_KARG(float*, p.E) + (int64_t)d * R;
#pragma unroll
                for (int i = 0; i < H; ++i) E[gt0 + i] = 0.0f;                                       // viter = 0: no responsibilities
            }
            if (w == 0) {
                CB_KARG(uint8_t*, p.sweeps)[d] = (uint8_t)min(sweeps, 255);
                CB_KARG(uint16_t*, p.doc_newton)[d] = (uint16_t)min(nsteps, 65535u);               // next E-step's grouping key (ctm_reorder_kernel)
            }
        }
        if (w == 0) {
            const unsigned tot = wave_sum_u(valid ? nsteps : 0u);
            if (lane == 0) {
                unsigned long long* newton_steps = CB_KARG(unsigned long long*, p.newton_steps);
                unsigned long long* diag = CB_KARG(unsigned long long*, tb.cg_iters);
                if (newton_steps) atomicAdd(newton_steps, (unsigned long long)tot);
                if (diag) {
                    atomicAdd(diag, (unsigned long long)ncg); atomicAdd(diag + 1, (unsigned long long)ntrip); atomicAdd(diag + 2, 1ull);
                    if constexpr (PROF) {
                        for (int q = 0; q < 8; ++q) atomicAdd(diag + 3 + q, (unsigned long long)cyc[q]);
                        atomicAdd(diag + 11, (unsigned long long)(__builtin_readcyclecounter() - t_start));
                    }
                }
            }
        }
    }   // next item
#undef CQ_SYNC
#undef VS
#undef PHIC
}
