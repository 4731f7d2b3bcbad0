// tmvb_internal.h -- shared host/device helpers of libtmvb_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cmath>
#include <string>
#include <vector>

#include "tmvb.h"

// ------------------------------------------------------------------------------------ errors
void tmvb_set_error(const char* fmt, ...);
// The library's HIP streams (tmvb_core.hip): slot 0 = a context's stream, slots 1 .. = the models' auxiliary streams.
// Default: every context / model creates its own and destroys them with it.  TMVB_STREAM_POOL=1: one set per device for the life of
// the process, shared by every context / model on that device (models alive at the same time then serialise).
// Why the switch exists (profiles/r3_stream_pool.txt): the runtime maps streams onto a few hardware queues, and how many queues are
// alive / which streams share one changes the iteration time by 6 - 10 % -- LDA K = 100 on SYN-NSF runs 835 it/s as the first model
// of a process (or next to a live one) and 750 as a model built after another one was closed; with GPU_MAX_HW_QUEUES=8 the FIRST
// model is the slow one (688), with 2 every model is (755).  The pool makes a sequence of models in a plain process all run at the
// first model's rate (833 / 832 / ...), but inside a process that also drives torch streams (bench.py) its effect depended on what
// had run before (LDA K = 100 712 ... 834, CTPF 1660 ... 5400 it/s), so it is not the default; bench.py measures each of its side
// configurations in a fresh process instead.  What fixed the sequences measured is in tmvb_core.hip (tmvb_env_defaults: eight hardware
// queues) and tmvb_lda.hip (the side chain on aux[1]); the pool stays as a switch.
hipStream_t tmvb_pool_stream(int device, int slot, bool high_priority = false);
// Flags of the library's ordering events (stream forks and joins on ONE device).  TMVB_EVENT_FLAGS: 0 = hipEventDisableTiming,
// 1 (default) = | hipEventDisableSystemFence, 2 = | hipEventReleaseToDevice (tmvb_core.hip)
unsigned tmvb_event_flags();
void tmvb_release_stream(hipStream_t st);            // no-op for pooled streams (TMVB_STREAM_POOL=0: destroys the caller's own stream)
bool tmvb_streams_pooled();

#define TMVB_HIP(call)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            tmvb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,    \
                           __LINE__);                                                          \
            return TMVB_EHIP;                                                                  \
        }                                                                                      \
    } while (0)

#define TMVB_REQUIRE(cond, code, ...)                                                          \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            tmvb_set_error(__VA_ARGS__);                                                       \
            return (code);                                                                     \
        }                                                                                      \
    } while (0)

// Destroys a half-built handle on every early return of its *_create (TMVB_HIP / TMVB_REQUIRE return from the middle of
// the function); release() on success.
template <class H, int (*Destroy)(H*)>
struct tmvb_create_guard {
    H* h;
    ~tmvb_create_guard() { if (h) (void)Destroy(h); }
    void release() { h = nullptr; }
};

// ------------------------------------------------------------------------------------ handles
struct tmvb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    size_t lds_per_cu = 160 * 1024;
};

// One launch bucket of the per-document kernels: documents [first, first+count) of doc_order,
// all with at most `tile_rows` rows (or more, for the last bucket, which streams chunks).
struct tmvb_bucket {
    int64_t first = 0, count = 0;
    int32_t tile_rows = 0;
    int32_t reg_tiles = 0;     // > 0: register-tile kernel with this many 64-token tiles (no LDS tile)
    int32_t piece = 0;         // pipelined E-step: which statistics pass consumes this bucket's documents
    int32_t waves = 1;         // > 1: register-tile kernel with one workgroup of `waves` waves per (long) document
    int32_t grid_np = 0;       // > 0: grid-tile kernel (tmvb_gridtile.h) with this many token PAIRS per lane (<= 32 grid_np tokens)
    int32_t grid_np2 = 0;      // CTPF: reader pairs per lane of the grid-tile kernel
};

// Inverted (id-major) index over a CSR token stream, cut into chunks of at most TMVB_CHUNK tokens.
// Same information as the reference's terms_sortperm / J_cumsum (src/modelutils.jl:372-381) and
// readers_sortperm / Y_cumsum (:443-472); used by the gather-side statistics kernels, which rebuild
// phi .* counts' from (w_token, E_doc) instead of reading a materialised phi buffer.
#define TMVB_CHUNK 256
#define TMVB_CHUNK_SMALL 256                 // chunk size of an index of fewer than TMVB_SMALL_INDEX_POSTINGS postings (round 6: measured)
#define TMVB_SMALL_INDEX_POSTINGS 3000000
#define TMVB_CLASS_DOCS 8192            // documents per class of the statistics pass (tmvb_build_inv_index): 2 MB of 256-byte rows
#define TMVB_CLASS_MIN_POSTINGS 1024    // ids with fewer postings are not cut at class boundaries
struct tmvb_inv_index {
    bool built = false;
    int64_t n_ids = 0, nnz = 0;
    int64_t n_docs = 0;                // documents of the corpus the index was built on (tok_doc values are < n_docs)
    int32_t* d_doc = nullptr;          // [nnz] document of each token, id-major order
    int32_t* d_pos = nullptr;          // [nnz] CSR position of each token, id-major order
    int32_t* d_inv = nullptr;          // [nnz] id-major position of each CSR token (inverse of d_pos)
    float* d_val = nullptr;            // [nnz] count / rating of each token, id-major order
    int64_t n_chunks = 0;
    int32_t* d_chunk_id = nullptr;     // [n_chunks] id (term / reader) the chunk belongs to
    int32_t* d_chunk_begin = nullptr;  // [n_chunks] token range in id-major order
    int32_t* d_chunk_end = nullptr;
    int32_t* d_chunk_out = nullptr;    // [n_chunks] -1: only chunk of its id (direct write); else partial slot
    int64_t n_multi = 0;               // ids split over several chunks
    int32_t* d_multi_id = nullptr;     // [n_multi]
    int32_t* d_multi_first = nullptr;  // [n_multi] first partial slot
    int32_t* d_multi_count = nullptr;  // [n_multi] number of partial slots
    int64_t n_slots = 0;
    // id_cuts given to the builder: the chunk order is slice-major -- slice s = ids [slice_id[s], slice_id[s + 1]) = chunks
    // [slice_chunk[s], slice_chunk[s + 1]) and multi-chunk ids [slice_multi[s], slice_multi[s + 1]); empty = one slice
    std::vector<int64_t> slice_id, slice_chunk, slice_multi;
};
// doc_piece != NULL restricts the index to the documents d with doc_piece[d] == piece (pipelined E-step)
int tmvb_build_inv_index(tmvb_ctx* ctx, int64_t M, int64_t n_ids, const int64_t* h_ptr, const int32_t* h_ids,
                         const int32_t* h_vals, tmvb_inv_index* out, const int32_t* doc_piece = nullptr, int piece = 0,
                         const std::vector<int64_t>* id_cuts = nullptr);
void tmvb_free_inv_index(tmvb_inv_index* ix);

struct tmvb_corpus {
    tmvb_ctx* ctx = nullptr;
    tmvb_corpus_info_t info{};
    // device CSR
    int64_t* d_doc_ptr = nullptr;
    int32_t* d_terms = nullptr;
    int32_t* d_counts = nullptr;
    int64_t* d_rdr_ptr = nullptr;
    int32_t* d_readers = nullptr;
    int32_t* d_ratings = nullptr;
    // processing order: documents sorted by descending length (longest first)
    int32_t* d_doc_order = nullptr;
    std::vector<int32_t> h_doc_order;
    std::vector<int64_t> h_doc_len;     // N_d
    std::vector<int64_t> h_rdr_len;     // R_d
    // host copies kept for building the inverted indices lazily
    std::vector<int64_t> h_doc_ptr, h_rdr_ptr;
    std::vector<int32_t> h_terms, h_readers, h_counts, h_ratings;
    tmvb_inv_index term_index, reader_index;
};
// read-only view of a CTPF handle's resident state for the recommendation pass (tmvb_ctpf_recs.hip)
struct tmvb_ctpf_view {
    tmvb_ctx* ctx; tmvb_corpus* corp;
    int K; int64_t M, U;
    const float* gimel; const float* zayin;     // [M][K]
    const float* he;                             // [U][K]
    const double* rates;                         // [8][K]: bet, vav, dalet, het, *_old
};
int tmvb_ctpf_view_of(tmvb_ctpf* h, tmvb_ctpf_view* v);
int tmvb_corpus_term_index(tmvb_corpus* c);
// n sum-all-reduces issued by one host thread inside one RCCL group (tmvb_comm.hip)
int tmvb_comm_allreduce_group(tmvb_comm* const* comms, void* const* dev_ptrs, const int64_t* counts, int n, int32_t dtype);
int tmvb_comm_allreduce_on(tmvb_comm* c, void* dev_ptr, int64_t count, int32_t dtype, hipStream_t on);
int tmvb_corpus_reader_index(tmvb_corpus* c);

// EPSILON of the reference (src/utils.jl:3) = 2^-99, exactly representable in fp32.
// MUTANTS (tests/test_mutants_gpu.py; never defined in a shipped build): deliberately wrong variants of three operators, one -D flag each, built as
// libtmvb_hip_<name>.so by tools/build_mutants.sh -- the negative controls of the parity suite: a named test must FAIL on each of them.
//   TMVB_MUTANT_LDA_NO_EPS          epsilon dropped from LDA's phi / gamma (src/LDA.jl:152, :145)
//   TMVB_MUTANT_CTPF_LOG_BET        `log bet` where xi needs `log vav` (the reference's own OpenCL bug, src/gpuCTPF.jl:624 vs src/CTPF.jl:336)
//   TMVB_MUTANT_CTM_SIGMA_NEW_MU    update_sigma! centred on the NEW mu (quirk Q2: the reference updates sigma first, src/CTM.jl:207-208)
//   TMVB_MUTANT_FLDA_NO_EPS         (round 6) the filtered models' log table without epsilon: log(beta) for log(beta + eps) (src/fLDA.jl:184, :191)
//   TMVB_MUTANT_FCTM_VSQ_FIRST      (round 6) fCTM's sweep in CTM's order: update_vsq! in front of update_lambda! (src/fCTM.jl:239-240 against src/CTM.jl:198-199)
#ifdef TMVB_MUTANT_LDA_NO_EPS
#define TMVB_EPS_F 0.0f
#else
#define TMVB_EPS_F 1.5777218104420236e-30f
#endif
#define TMVB_EPS_D 1.5777218104420236e-30

// host fp64 digamma (same published algorithm as the device version below)
static inline double tmvb_digamma_host(double x)
{
    double psi = 0.0;
    if (x < 7.0) {
        int n = 7 - (int)std::floor(x);
        for (int v = 1; v < n; ++v) psi -= 1.0 / (x + (double)v);
        psi -= 1.0 / x;
        x += (double)n;
    }
    double t = 1.0 / x;
    psi += std::log(x) - 0.5 * t;
    t *= t;
    double p = -0.4432598039215686;
    p = p * t + 0.08333333333333333;
    p = p * t + -0.021092796092796094;
    p = p * t + 0.007575757575757576;
    p = p * t + -0.004166666666666667;
    p = p * t + 0.003968253968253968;
    p = p * t + -0.008333333333333333;
    p = p * t + 0.08333333333333333;
    return psi - t * p;
}

static inline int tmvb_kpad(int K)
{
    // row stride of the LDS topic tile: smallest 4*odd >= K  (conflict-free ds_read_b128 by rows)
    int q = (K + 3) / 4;
    if ((q & 1) == 0) ++q;
    return 4 * q;
}

// ------------------------------------------------------------------------------------ device
#if defined(__HIPCC__)

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// sum over the 64 lanes of a wave, result uniform (held in an SGPR)
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);   // row_half_mirror
    v += dpp_f<0x140>(v);   // row_mirror  -> every lane of a 16-lane row holds the row sum
    // gfx9 wave-level DPP: rows 1 and 3 add lane 15 of the row before, then rows 2 and 3 add lane 31:
    // lane 63 holds the wave total ((r0 + r1) + (r2 + r3), a fixed order)
    // (inline asm: the builtin form is not folded into v_add_f32_dpp; s_nop = VALU-write -> DPP-read wait states)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0" : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)));
    r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)));
    r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
    return r;
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// a / b through v_rcp_f32 and one Newton step (4 VALU instructions; the IEEE sequence is 10).  The
// refined reciprocal is within 1 ulp; b must be finite, non-zero and normal (true at every call site:
// s_n >= K eps, digamma arguments >= eps).
__device__ __forceinline__ float fast_rcp(float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    return fmaf(fmaf(-b, r, 1.0f), r, r);
}
__device__ __forceinline__ float fast_div(float a, float b) { return a * fast_rcp(b); }

// exp(x) for the E-step (x = Elogtheta <= 0 in exact arithmetic): 2^n * 2^f with the product x*log2(e)
// carried in two floats, 8 VALU instructions (libm expf: 18); relative error < 2 ulp.  x is clamped
// at -150 (result 0 either way) so that -inf cannot turn into inf - inf.
__device__ __forceinline__ float fast_exp(float x)
{
    x = fmaxf(x, -150.0f);
    const float t = x * 1.44269502162933349609375f;
    const float n = __builtin_rintf(t);
    float f = fmaf(x, 1.44269502162933349609375f, -n);
    f = fmaf(x, 1.925963033500011e-08f, f);              // low part of log2(e)
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// digamma, x > 0, fp32.  Same algorithm as the reference's helper (src/utils.jl:21-53):
// upward recurrence then the asymptotic series; done branch-free with a fixed 6-step shift,
// and 4 series terms (the 5th is < 1e-10 at x >= 6).
__device__ __forceinline__ float digamma_f(float x)
{
    bool small = x < 6.0f;
    float y = small ? x + 6.0f : x;
    float rec = 0.0f;
    if (small) {
        // sum_{v=0}^{5} 1/(x+v) as two rational halves to save reciprocals
        float a0 = x, a1 = x + 1.0f, a2 = x + 2.0f, a3 = x + 3.0f, a4 = x + 4.0f, a5 = x + 5.0f;
        float p01 = a0 * a1, p23 = a2 * a3, p45 = a4 * a5;
        // 1/a0+1/a1 = (a0+a1)/p01 etc.
        float n01 = a0 + a1, n23 = a2 + a3, n45 = a4 + a5;
        rec = fast_div(n01, p01) + (fast_div(n23, p23) + fast_div(n45, p45));
    }
    float t = fast_rcp(y);
    float psi = logf(y) - 0.5f * t;
    float w = t * t;
    float p = -0.004166666666666667f;
    p = p * w + 0.003968253968253968f;
    p = p * w + -0.008333333333333333f;
    p = p * w + 0.08333333333333333f;
    psi -= w * p;
    return psi - rec;
}

// digamma for the sweep loops, x > 0, fp32, 19 VALU + 3 transcendental instructions (digamma_f: ~45): the recurrence is a fixed
// 3-step shift written as ONE rational, psi(x) = psi(y) - (3x^2 + 6x + 2) / (x (x+1) (x+2)), y = x + 3, merged with the series'
// 1/(2y); six series terms (the seventh is 1.7e-8 at y = 3); raw v_log_f32 (y >= 3 is never denormal).  Max error against
// fp64 over [1e-6, 1e5]: 8.8e-7 (absolute where |psi| <= 1, relative elsewhere); digamma_f: 4.9e-7.
__device__ __forceinline__ float digamma_sweep_f(float x)
{
    const float y = x + 3.0f, x1 = x + 1.0f, x2 = x + 2.0f;
    const float D = (x * x1) * x2;
    const float A = fmaf(fmaf(6.0f, x, 12.0f), x, 4.0f);            // 2 (3x^2 + 6x + 2)
    const float num = fmaf(y, A, D);                                 // (1/x + 1/(x+1) + 1/(x+2) + 1/(2y)) * 2 y D
    const float r = __builtin_amdgcn_rcpf(y * D);
    const float t2 = __builtin_amdgcn_rcpf(y * y);
    float p = -0.021092796092796094f;                                // B_12 / 12
    p = fmaf(p, t2, 0.007575757575757576f);
    p = fmaf(p, t2, -0.004166666666666667f);
    p = fmaf(p, t2, 0.003968253968253968f);
    p = fmaf(p, t2, -0.008333333333333333f);
    p = fmaf(p, t2, 0.08333333333333333f);
    float psi = fmaf(__builtin_amdgcn_logf(y), 0.6931471805599453f, -0.5f * num * r);
    return fmaf(-t2, p, psi);
}

// exp(x), x <= 0 finite: 2^t (v_exp_f32) with t = fl(x log2 e) and the rounding of t put back to first order,
// e = 2^t (1 + ln2 (x log2 e - t)); 5 VALU + 1 transcendental (fast_exp: 9 + 1), relative error 1.2e-7.
__device__ __forceinline__ float sweep_exp(float x)
{
    const float t = x * 1.44269502162933349609375f;
    float r = fmaf(x, 1.44269502162933349609375f, -t);
    r = fmaf(x, 1.925963033500011e-08f, r);
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.6931471805599453f, e);
}

// fp64 digamma / trigamma: the published SpecialFunctions.jl algorithm (recurrence to x>=7 / 8,
// then 8-term asymptotic series), used by the alpha Newton step and the ELBO kernels.
__device__ __forceinline__ double digamma_d(double x)
{
    double psi = 0.0;
    if (x < 7.0) {
        int n = 7 - (int)floor(x);
        for (int v = 1; v < n; ++v) psi -= 1.0 / (x + (double)v);
        psi -= 1.0 / x;
        x += (double)n;
    }
    double t = 1.0 / x;
    psi += log(x) - 0.5 * t;
    t *= t;
    double p = -0.4432598039215686;
    p = p * t + 0.08333333333333333;
    p = p * t + -0.021092796092796094;
    p = p * t + 0.007575757575757576;
    p = p * t + -0.004166666666666667;
    p = p * t + 0.003968253968253968;
    p = p * t + -0.008333333333333333;
    p = p * t + 0.08333333333333333;
    return psi - t * p;
}

__device__ __forceinline__ double trigamma_d(double x)
{
    double psi = 0.0;
    if (x < 8.0) {
        int n = 8 - (int)floor(x);
        psi += 1.0 / (x * x);
        for (int v = 1; v < n; ++v) {
            double y = x + (double)v;
            psi += 1.0 / (y * y);
        }
        x += (double)n;
    }
    double t = 1.0 / x, w = t * t;
    psi += t + 0.5 * w;
    double p = -7.092156862745098;
    p = p * w + 1.1666666666666667;
    p = p * w + -0.2531135531135531;
    p = p * w + 0.07575757575757576;
    p = p * w + -0.03333333333333333;
    p = p * w + 0.023809523809523808;
    p = p * w + -0.03333333333333333;
    p = p * w + 0.16666666666666666;
    return psi + t * w * p;
}

// fp64 reciprocal: v_rcp_f64 (about 27 bits) + two Newton steps; 1 ulp-class (not correctly rounded, which nothing here needs).
__device__ __forceinline__ double tmvb_rcp_d(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// digamma and trigamma of the same argument for the alpha Newton step, fp64, x > 0.  The published algorithm (recurrence to
// x >= 7 / 8, then the asymptotic series; Appendix A of SURVEY.md) with the recurrence written for the GPU: instead of a loop of
// 7 - floor(x) IEEE divisions per lane (the lanes of a wave run the longest loop: ~2 x 8 divisions of ~40 instructions each per
// call), a FIXED shift by 8 for x < 8 with both recurrence sums from ONE reciprocal: with P(x) = prod_{v=0}^{7} (x + v),
//   sum 1/(x+v) = P'/P,   sum 1/(x+v)^2 = (P'/P)^2 - P''/P
// (P, P', P'' by Horner on the expanded coefficients: 21 fmas).  The series then runs at y = x + 8 >= 8.  Agreement with the loop
// form: <= 4e-15 relative on (1e-6, 1e4) (host check in tests/test_oracle_special.py style; the alpha tolerance is 1e-4).
__device__ __forceinline__ void digamma_trigamma_d(double x, double& psi, double& psi1)
{
    const bool small = x < 8.0;
    double y = x, rec0 = 0.0, rec1 = 0.0;
    if (small) {
        // P(x) = x (x+1) ... (x+7) = x^8 + 28 x^7 + 322 x^6 + 1960 x^5 + 6769 x^4 + 13132 x^3 + 13068 x^2 + 5040 x
        double P = x + 28.0; P = fma(P, x, 322.0); P = fma(P, x, 1960.0); P = fma(P, x, 6769.0); P = fma(P, x, 13132.0);
        P = fma(P, x, 13068.0); P = fma(P, x, 5040.0); P *= x;
        double P1 = fma(8.0, x, 196.0); P1 = fma(P1, x, 1932.0); P1 = fma(P1, x, 9800.0); P1 = fma(P1, x, 27076.0);
        P1 = fma(P1, x, 39396.0); P1 = fma(P1, x, 26136.0); P1 = fma(P1, x, 5040.0);
        double P2 = fma(56.0, x, 1176.0); P2 = fma(P2, x, 9660.0); P2 = fma(P2, x, 39200.0); P2 = fma(P2, x, 81228.0);
        P2 = fma(P2, x, 78792.0); P2 = fma(P2, x, 26136.0);
        const double rP = tmvb_rcp_d(P);
        rec0 = P1 * rP;                                   // sum_{v=0}^{7} 1/(x+v)
        rec1 = fma(rec0, rec0, -P2 * rP);                 // sum_{v=0}^{7} 1/(x+v)^2
        y = x + 8.0;
    }
    const double t = tmvb_rcp_d(y), w = t * t;
    double p = -0.4432598039215686;
    p = fma(p, w, 0.08333333333333333);
    p = fma(p, w, -0.021092796092796094);
    p = fma(p, w, 0.007575757575757576);
    p = fma(p, w, -0.004166666666666667);
    p = fma(p, w, 0.003968253968253968);
    p = fma(p, w, -0.008333333333333333);
    p = fma(p, w, 0.08333333333333333);
    psi = (log(y) - 0.5 * t) - w * p - rec0;
    double q = -7.092156862745098;
    q = fma(q, w, 1.1666666666666667);
    q = fma(q, w, -0.2531135531135531);
    q = fma(q, w, 0.07575757575757576);
    q = fma(q, w, -0.03333333333333333);
    q = fma(q, w, 0.023809523809523808);
    q = fma(q, w, -0.03333333333333333);
    q = fma(q, w, 0.16666666666666666);
    psi1 = (t + 0.5 * w) + t * w * q + rec1;
}

// digamma and log-gamma of the same argument x > 0 in fp64 with ONE pass of the same fixed shift (no branch: the identities hold
// for every x > 0):  psi(x) = psi(x + 8) - P'/P,  lgamma(x) = lgamma(x + 8) - log P(x),  P(x) = x (x+1) ... (x+7); psi and
// lgamma at y = x + 8 >= 8 by their asymptotic series (psi as above; Stirling's series to the y^-13 term, next term 8e-16 at y = 8)
// sharing log y and 1 / y.  Two logarithms and two reciprocals per call where lgamma() + digamma_d() of the library cost several
// hundred instructions and a loop of divisions.  <= 2e-15 relative against mpmath on (1e-6, 1e5) (formula check on the host).
__device__ __forceinline__ void digamma_lgamma_d(double x, double& psi, double& lg)
{
    double P = x + 28.0; P = fma(P, x, 322.0); P = fma(P, x, 1960.0); P = fma(P, x, 6769.0); P = fma(P, x, 13132.0);
    P = fma(P, x, 13068.0); P = fma(P, x, 5040.0); P *= x;
    double P1 = fma(8.0, x, 196.0); P1 = fma(P1, x, 1932.0); P1 = fma(P1, x, 9800.0); P1 = fma(P1, x, 27076.0);
    P1 = fma(P1, x, 39396.0); P1 = fma(P1, x, 26136.0); P1 = fma(P1, x, 5040.0);
    const double rec0 = P1 * tmvb_rcp_d(P);               // sum_{v=0}^{7} 1/(x+v)
    const double y = x + 8.0, t = tmvb_rcp_d(y), w = t * t, ly = log(y);
    double p = -0.4432598039215686;
    p = fma(p, w, 0.08333333333333333);
    p = fma(p, w, -0.021092796092796094);
    p = fma(p, w, 0.007575757575757576);
    p = fma(p, w, -0.004166666666666667);
    p = fma(p, w, 0.003968253968253968);
    p = fma(p, w, -0.008333333333333333);
    p = fma(p, w, 0.08333333333333333);
    psi = (ly - 0.5 * t) - w * p - rec0;
    double s = 1.0 / 156.0;
    s = fma(s, w, -691.0 / 360360.0);
    s = fma(s, w, 1.0 / 1188.0);
    s = fma(s, w, -1.0 / 1680.0);
    s = fma(s, w, 1.0 / 1260.0);
    s = fma(s, w, -1.0 / 360.0);
    s = fma(s, w, 1.0 / 12.0);
    lg = ((y - 0.5) * ly - y + 0.91893853320467274178) + t * s - log(P);
}
// psi alone, same shift (one logarithm)
__device__ __forceinline__ double digamma_shift8_d(double x)
{
    double P = x + 28.0; P = fma(P, x, 322.0); P = fma(P, x, 1960.0); P = fma(P, x, 6769.0); P = fma(P, x, 13132.0);
    P = fma(P, x, 13068.0); P = fma(P, x, 5040.0); P *= x;
    double P1 = fma(8.0, x, 196.0); P1 = fma(P1, x, 1932.0); P1 = fma(P1, x, 9800.0); P1 = fma(P1, x, 27076.0);
    P1 = fma(P1, x, 39396.0); P1 = fma(P1, x, 26136.0); P1 = fma(P1, x, 5040.0);
    const double rec0 = P1 * tmvb_rcp_d(P);
    const double y = x + 8.0, t = tmvb_rcp_d(y), w = t * t;
    double p = -0.4432598039215686;
    p = fma(p, w, 0.08333333333333333);
    p = fma(p, w, -0.021092796092796094);
    p = fma(p, w, 0.007575757575757576);
    p = fma(p, w, -0.004166666666666667);
    p = fma(p, w, 0.003968253968253968);
    p = fma(p, w, -0.008333333333333333);
    p = fma(p, w, 0.08333333333333333);
    return (log(y) - 0.5 * t) - w * p - rec0;
}


#endif  // __HIPCC__
