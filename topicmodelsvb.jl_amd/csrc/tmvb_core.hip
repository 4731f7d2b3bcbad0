// tmvb_core.hip -- context, error reporting and corpus upload of libtmvb_hip.so.
//
// Replaces: cl.create_compute_context() (src/gpuLDA.jl:64) and the corpus half of update_buffer!
// (src/modelutils.jl:370-388, :438-472).  The reference uploads flat `terms/counts` + `N_cumsum`
// and an inverted index (`terms_sortperm`, `J_cumsum`); this engine needs no inverted index
// (phi is never materialised, statistics are scattered) but adds a length-sorted processing
// order so that the hardware dispatcher sees the longest documents first.
#include "tmvb_internal.h"

#include <cstdio>
#include <cstring>
#include <type_traits>
#include <utility>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

static thread_local std::string g_last_error;

void tmvb_set_error(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" int tmvb_abi_version(void) { return TMVB_ABI_VERSION; }
extern "C" const char* tmvb_last_error(void) { return g_last_error.c_str(); }

extern "C" int tmvb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// see tmvb_internal.h
bool tmvb_streams_pooled()
{
    static const bool on = [] { const char* e = getenv("TMVB_STREAM_POOL"); return e && atoi(e) != 0; }();   // opt-in, see tmvb_internal.h
    return on;
}

void tmvb_release_stream(hipStream_t st)
{
    if (st && !tmvb_streams_pooled()) (void)hipStreamDestroy(st);
}

unsigned tmvb_event_flags()
{
    static const unsigned f = [] {
        // default 1: the events order streams of ONE device, and every host read goes through a stream synchronisation or a copy, so the
        // system-scope fence of a default event buys nothing -- and costs: a 16 100-document LDA shard, whose iteration crosses streams four
        // times, runs 0.172 / 0.173 ms per iteration without it against 0.187 / 0.182 with (run r4ai, alternating); whole corpus and CTPF
        // inside the noise
        const char* e = getenv("TMVB_EVENT_FLAGS");
        const int v = e ? atoi(e) : 1;
        return (unsigned)hipEventDisableTiming | (v == 1 ? (unsigned)hipEventDisableSystemFence : v == 2 ? (unsigned)hipEventReleaseToDevice : 0u);
    }();
    return f;
}

hipStream_t tmvb_pool_stream(int device, int slot, bool high_priority)
{
    if (!tmvb_streams_pooled()) {                         // TMVB_STREAM_POOL=0: a stream of the caller's own (tmvb_release_stream destroys it)
        hipStream_t st = nullptr;
        hipError_t e;
        if (high_priority) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi); }
        else e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return st;
    }
    static std::mutex mu;
    static std::map<std::pair<int, int>, hipStream_t> pool;
    std::lock_guard<std::mutex> lk(mu);
    const std::pair<int, int> key{device, high_priority ? 1000 + slot : slot};
    auto it = pool.find(key);
    if (it != pool.end()) return it->second;
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(device);
    hipStream_t st = nullptr;
    hipError_t e;
    if (high_priority) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi);
    } else e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    (void)hipSetDevice(cur);
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    pool[key] = st;
    return st;
}

// Eight hardware queues (GPU_MAX_HW_QUEUES=8; the runtime's default is four) is the setting the stream plans were measured with: the
// models run two to four streams each, and with four queues a model built after others had been closed shared queues in ways that
// cost up to a third of its speed (one process, ctpf / lda100 / ctm / ctpf / lda100 / ...: LDA K = 100 531 it/s and CTM 185 against
// 838 and 202 in fresh processes; with eight every model runs at its fresh-process rate, profiles/r3_stream_pool.txt).  The runtime
// reads the variable when it initialises, and it is process-wide, so it is the LAUNCHER's to set, visibly and overridably -- python:
// _lib.py, bench.py before importing torch; Julia: TMVBHip.__init__; README -- and not this library's: loading a shared object must
// not reconfigure every other HIP user of the process (round-3 advice; the constructor that did so is gone).

extern "C" int tmvb_ctx_create(int32_t device_id, void* hip_stream, tmvb_ctx** out)
{
    TMVB_REQUIRE(out != nullptr, TMVB_EINVAL, "tmvb_ctx_create: out is NULL");
    *out = nullptr;
    int n = tmvb_device_count();
    TMVB_REQUIRE(n > 0, TMVB_ENODEVICE, "tmvb_ctx_create: no HIP device visible (the HIP engine has no CPU fallback)");
    TMVB_REQUIRE(device_id >= 0 && device_id < n, TMVB_EINVAL, "tmvb_ctx_create: device_id %d out of range [0,%d)", device_id, n);
    TMVB_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    TMVB_HIP(hipGetDeviceProperties(&prop, device_id));
    TMVB_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, TMVB_ENODEVICE,
                 "tmvb_ctx_create: device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    tmvb_ctx* c = new tmvb_ctx();
    c->device = device_id;
    c->num_cu = prop.multiProcessorCount;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
        c->own_stream = false;
    } else {
        c->stream = tmvb_pool_stream(device_id, 0);
        if (!c->stream) {
            delete c;
            tmvb_set_error("hipStreamCreate failed");
            return TMVB_EHIP;
        }
        c->own_stream = !tmvb_streams_pooled();           // pooled: lives as long as the process
    }
    *out = c;
    return TMVB_OK;
}

extern "C" int tmvb_ctx_destroy(tmvb_ctx* ctx)
{
    if (!ctx) return TMVB_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return TMVB_OK;
}

struct tmvb_event { tmvb_ctx* ctx; hipEvent_t ev; };

extern "C" int tmvb_event_create(tmvb_ctx* ctx, tmvb_event** out)
{
    TMVB_REQUIRE(ctx != nullptr && out != nullptr, TMVB_EINVAL, "tmvb_event_create: NULL argument");
    *out = nullptr;
    TMVB_HIP(hipSetDevice(ctx->device));
    hipEvent_t e = nullptr;
    TMVB_HIP(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));        // time stamps on, system-scope fence off
    *out = new tmvb_event{ctx, e};
    return TMVB_OK;
}

extern "C" int tmvb_event_record(tmvb_event* ev)
{
    TMVB_REQUIRE(ev != nullptr, TMVB_EINVAL, "tmvb_event_record: NULL argument");
    TMVB_HIP(hipSetDevice(ev->ctx->device));
    TMVB_HIP(hipEventRecord(ev->ev, ev->ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_event_elapsed_ms(tmvb_event* start, tmvb_event* stop, float* ms)
{
    TMVB_REQUIRE(start && stop && ms, TMVB_EINVAL, "tmvb_event_elapsed_ms: NULL argument");
    TMVB_HIP(hipSetDevice(stop->ctx->device));
    TMVB_HIP(hipEventSynchronize(stop->ev));
    TMVB_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return TMVB_OK;
}

extern "C" int tmvb_event_destroy(tmvb_event* ev)
{
    if (!ev) return TMVB_OK;
    (void)hipSetDevice(ev->ctx->device);
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return TMVB_OK;
}

extern "C" int tmvb_ctx_synchronize(tmvb_ctx* ctx)
{
    TMVB_REQUIRE(ctx != nullptr, TMVB_EINVAL, "tmvb_ctx_synchronize: ctx is NULL");
    TMVB_HIP(hipSetDevice(ctx->device));
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    return TMVB_OK;
}

template <typename T>
static int upload(tmvb_ctx* ctx, T** dptr, const T* h, size_t n)
{
    *dptr = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T) + 64;      // slack: kernels read token ids four at a time (tmvb_ctm_batch.h)
    hipError_t e = hipMalloc((void**)dptr, bytes);
    if (e != hipSuccess) {
        tmvb_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return TMVB_ENOMEM;
    }
    if (n) TMVB_HIP(hipMemcpyAsync(*dptr, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    // the slack (and the one element an EMPTY array still holds) is defined memory: the branch-free tile loads read entry 0 for an
    // empty document and use it as a row index before the weight masks it (tmvb_gridtile.h: grid_load_tile) -- id 0 is always valid
    TMVB_HIP(hipMemsetAsync((char*)*dptr + n * sizeof(T), 0, bytes - n * sizeof(T), ctx->stream));
    return TMVB_OK;
}

extern "C" int tmvb_corpus_create(tmvb_ctx* ctx, int64_t M, int64_t V, int64_t U,
                                  const int64_t* doc_ptr, const int32_t* terms, const int32_t* counts,
                                  const int64_t* rdr_ptr, const int32_t* readers, const int32_t* ratings,
                                  tmvb_corpus** out)
{
    TMVB_REQUIRE(out != nullptr, TMVB_EINVAL, "tmvb_corpus_create: out is NULL");
    *out = nullptr;
    TMVB_REQUIRE(ctx != nullptr, TMVB_EINVAL, "tmvb_corpus_create: ctx is NULL");
    TMVB_REQUIRE(M >= 0 && V >= 0 && U >= 0, TMVB_EINVAL, "tmvb_corpus_create: negative size");
    TMVB_REQUIRE(M < (int64_t)INT32_MAX && V < (int64_t)INT32_MAX && U < (int64_t)INT32_MAX, TMVB_EINVAL,
                 "tmvb_corpus_create: M, V, U must fit int32");
    TMVB_REQUIRE(doc_ptr != nullptr, TMVB_EINVAL, "tmvb_corpus_create: doc_ptr is NULL");
    TMVB_REQUIRE(doc_ptr[0] == 0, TMVB_ECORPUS, "doc_ptr[0] must be 0");
    for (int64_t d = 0; d < M; ++d)
        TMVB_REQUIRE(doc_ptr[d + 1] >= doc_ptr[d], TMVB_ECORPUS, "document %lld failed check: doc_ptr not monotone", (long long)(d + 1));
    int64_t nnz = doc_ptr[M];
    TMVB_REQUIRE(nnz == 0 || (terms && counts), TMVB_EINVAL, "tmvb_corpus_create: terms/counts are NULL");
    bool has_r = rdr_ptr != nullptr;
    int64_t nR = 0;
    if (has_r) {
        TMVB_REQUIRE(rdr_ptr[0] == 0, TMVB_ECORPUS, "rdr_ptr[0] must be 0");
        for (int64_t d = 0; d < M; ++d)
            TMVB_REQUIRE(rdr_ptr[d + 1] >= rdr_ptr[d], TMVB_ECORPUS, "document %lld failed check: rdr_ptr not monotone", (long long)(d + 1));
        nR = rdr_ptr[M];
        TMVB_REQUIRE(nR == 0 || (readers && ratings), TMVB_EINVAL, "tmvb_corpus_create: readers/ratings are NULL");
    }
    tmvb_corpus* c = new tmvb_corpus();
    tmvb_create_guard<tmvb_corpus, tmvb_corpus_destroy> guard{c};      // every early return below destroys c
    c->ctx = ctx;
    tmvb_corpus_info_t& I = c->info;
    I.M = M; I.V = V; I.U = U; I.nnz = nnz; I.nR = nR;
    c->h_doc_len.resize(M);
    c->h_rdr_len.assign(M, 0);
    // check_doc (src/Corpus.jl:41-50) + check_corp (src/Corpus.jl:111-122), 0-based ids
    std::vector<int32_t> seen_t(V, -1), seen_r(U, -1);
    for (int64_t d = 0; d < M; ++d) {
        int64_t a = doc_ptr[d], b = doc_ptr[d + 1];
        c->h_doc_len[d] = b - a;
        I.max_doc_len = std::max(I.max_doc_len, b - a);
        if (b == a) I.n_empty_docs++;
        bool dup = false;
        for (int64_t q = a; q < b; ++q) {
            if (terms[q] < 0 || terms[q] >= V) {
                tmvb_set_error("document %lld failed check: term id %d outside the vocabulary [0,%lld)", (long long)(d + 1), terms[q], (long long)V);
                return TMVB_ECORPUS;
            }
            if (counts[q] <= 0) {
                tmvb_set_error("document %lld failed check: all counts must be positive integers", (long long)(d + 1));
                return TMVB_ECORPUS;
            }
            I.sum_counts += counts[q];
            if (seen_t[terms[q]] == (int32_t)d) dup = true;
            seen_t[terms[q]] = (int32_t)d;
        }
        if (dup) I.n_docs_with_duplicate_terms++;
        if (has_r) {
            int64_t ra = rdr_ptr[d], rb = rdr_ptr[d + 1];
            c->h_rdr_len[d] = rb - ra;
            I.max_readers = std::max(I.max_readers, rb - ra);
            bool dupr = false;
            for (int64_t q = ra; q < rb; ++q) {
                if (readers[q] < 0 || readers[q] >= U) {
                    tmvb_set_error("document %lld failed check: reader id %d outside the users [0,%lld)", (long long)(d + 1), readers[q], (long long)U);
                    return TMVB_ECORPUS;
                }
                if (ratings[q] <= 0) {
                    tmvb_set_error("document %lld failed check: all ratings must be positive integers", (long long)(d + 1));
                    return TMVB_ECORPUS;
                }
                I.sum_ratings += ratings[q];
                if (seen_r[readers[q]] == (int32_t)d) dupr = true;
                seen_r[readers[q]] = (int32_t)d;
            }
            if (dupr) I.n_docs_with_duplicate_readers++;
        }
    }
    // processing order: longest document first (stable), so that the hardware dispatcher's
    // in-order workgroup issue behaves like longest-processing-time-first scheduling.
    c->h_doc_order.resize(M);
    std::iota(c->h_doc_order.begin(), c->h_doc_order.end(), 0);
    std::stable_sort(c->h_doc_order.begin(), c->h_doc_order.end(), [&](int32_t x, int32_t y) {
        int64_t lx = c->h_doc_len[x] + c->h_rdr_len[x], ly = c->h_doc_len[y] + c->h_rdr_len[y];
        return lx > ly;
    });
    (void)hipSetDevice(ctx->device);
    int rc;
    std::vector<int64_t> zero_ptr;
    if (!has_r) zero_ptr.assign(M + 1, 0);
    if ((rc = upload(ctx, &c->d_doc_ptr, doc_ptr, (size_t)M + 1)) != TMVB_OK ||
        (rc = upload(ctx, &c->d_terms, terms, (size_t)nnz)) != TMVB_OK ||
        (rc = upload(ctx, &c->d_counts, counts, (size_t)nnz)) != TMVB_OK ||
        (rc = upload(ctx, &c->d_rdr_ptr, has_r ? rdr_ptr : zero_ptr.data(), (size_t)M + 1)) != TMVB_OK ||
        (rc = upload(ctx, &c->d_readers, readers, (size_t)nR)) != TMVB_OK ||
        (rc = upload(ctx, &c->d_ratings, ratings, (size_t)nR)) != TMVB_OK ||
        (rc = upload(ctx, &c->d_doc_order, c->h_doc_order.data(), (size_t)M)) != TMVB_OK) {
        return rc;
    }
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    c->h_doc_ptr.assign(doc_ptr, doc_ptr + M + 1);
    c->h_terms.assign(terms, terms + nnz);
    c->h_counts.assign(counts, counts + nnz);
    if (has_r) {
        c->h_rdr_ptr.assign(rdr_ptr, rdr_ptr + M + 1);
        c->h_readers.assign(readers, readers + nR);
        c->h_ratings.assign(ratings, ratings + nR);
    } else {
        c->h_rdr_ptr.assign(M + 1, 0);
    }
    guard.release();
    *out = c;
    return TMVB_OK;
}

int tmvb_build_inv_index(tmvb_ctx* ctx, int64_t M, int64_t n_ids, const int64_t* h_ptr, const int32_t* h_ids,
                         const int32_t* h_vals, tmvb_inv_index* ix, const int32_t* doc_piece, int piece, const std::vector<int64_t>* id_cuts)
{
    const int64_t nnz_all = h_ptr[M];
    TMVB_REQUIRE(nnz_all < (int64_t)INT32_MAX, TMVB_EINVAL, "inverted index: token count must fit int32");
    auto in_piece = [&](int64_t d) { return doc_piece == nullptr || doc_piece[d] == piece; };
    // The postings sorted by id, documents in order within an id: a counting sort.  Large indices (round 6: the index build was most of the 200 - 300 ms a
    // gpuLDA / gpuCTM constructor takes on SYN-NSF) split the documents into T contiguous ranges of equal postings, one host thread each: per-range histograms,
    // one prefix over (id, range), then every range scatters its own documents from its own cursors -- the same positions the one-thread loop assigns
    // (TMVB_CREATE_THREADS=0, or fewer than 2^20 postings: one thread).
    static const int thread_env = [] { const char* e = getenv("TMVB_CREATE_THREADS"); return e ? atoi(e) : -1; }();
    int T = 1;
    if (thread_env != 0 && nnz_all >= (1 << 20)) {
        const unsigned hw = std::thread::hardware_concurrency();
        T = (int)std::min<unsigned>(thread_env > 0 ? (unsigned)thread_env : 8u, hw ? hw : 1u);
        T = std::max(1, std::min(T, 16));
    }
    std::vector<int64_t> dcut((size_t)T + 1, M);                       // document ranges [dcut[t], dcut[t + 1])
    dcut[0] = 0;
    for (int t = 1; t < T; ++t) {
        const int64_t target = nnz_all / T * t;
        dcut[(size_t)t] = std::lower_bound(h_ptr, h_ptr + M + 1, target) - h_ptr;
        dcut[(size_t)t] = std::min<int64_t>(std::max(dcut[(size_t)t], dcut[(size_t)t - 1]), M);
    }
    std::vector<std::vector<int32_t>> hist((size_t)T);
    auto count_range = [&](int t) {
        std::vector<int32_t>& H = hist[(size_t)t];
        H.assign((size_t)n_ids, 0);
        for (int64_t d = dcut[(size_t)t]; d < dcut[(size_t)t + 1]; ++d)
            if (in_piece(d))
                for (int64_t q = h_ptr[d]; q < h_ptr[d + 1]; ++q) H[(size_t)h_ids[q]]++;
    };
    auto run_ranges = [&](auto&& fn) {
        std::vector<std::thread> workers;
        for (int t = 1; t < T; ++t) workers.emplace_back(fn, t);
        fn(0);
        for (std::thread& w : workers) w.join();
    };
    run_ranges(count_range);
    std::vector<int64_t> cnt(n_ids + 1, 0);
    for (int64_t j = 0; j < n_ids; ++j) {
        int64_t c = 0;
        for (int t = 0; t < T; ++t) c += hist[(size_t)t][(size_t)j];
        cnt[j + 1] = cnt[j] + c;
    }
    const int64_t nnz = cnt[n_ids];
    ix->n_ids = n_ids; ix->nnz = nnz; ix->n_docs = M;
    std::vector<int32_t> doc(nnz), pos(nnz), inv(doc_piece ? 0 : nnz_all);
    std::vector<float> val(nnz);
    {
        // cursors of range t: behind the postings of the same id in the ranges before it
        std::vector<std::vector<int64_t>> cur((size_t)T);
        for (int t = 0; t < T; ++t) cur[(size_t)t].resize((size_t)n_ids);
        for (int64_t j = 0; j < n_ids; ++j) {
            int64_t w = cnt[j];
            for (int t = 0; t < T; ++t) { cur[(size_t)t][(size_t)j] = w; w += hist[(size_t)t][(size_t)j]; }
        }
        auto scatter_range = [&](int t) {
            std::vector<int64_t>& cu = cur[(size_t)t];
            for (int64_t d = dcut[(size_t)t]; d < dcut[(size_t)t + 1]; ++d) {
                if (!in_piece(d)) continue;
                for (int64_t q = h_ptr[d]; q < h_ptr[d + 1]; ++q) {
                    const int64_t w = cu[(size_t)h_ids[q]]++;
                    doc[w] = (int32_t)d; pos[w] = (int32_t)q; val[w] = (float)h_vals[q];
                    if (!doc_piece) inv[q] = (int32_t)w;
                }
            }
        };
        run_ranges(scatter_range);
    }
    // Chunks (one wave each, 4 per workgroup) and their ORDER.  The statistics pass gathers one row of per-document factors per
    // posting: 33 MB of rows on SYN-NSF against 4 MB of L2 per XCD, 20 % L2 hits (TCC_HIT / TCC_REQ), the rest from the Infinity
    // Cache at ~700 cycles.  Postings of an id are sorted by document, so a frequent id's list is cut at the boundaries of NC
    // document-id CLASSES (<= TMVB_CLASS_DOCS documents each: their rows fit an L2) and the chunks of class x go to workgroups
    // b = x (mod 8) -- workgroups are handed to the 8 XCDs round robin, so an XCD's L2 sees one class at a time (speed only: any
    // placement gives the same sums).  Rare ids (< TMVB_CLASS_MIN_POSTINGS postings) stay whole: cutting them would multiply the
    // waves and the partial-sum slots for a few postings each; they fill the classes' last workgroups and the tail of the launch.
    struct Chunk { int32_t id, b, e, out; };
    std::vector<int32_t> mid, mfirst, mcount;
    int64_t slots = 0;
    int64_t ndocs = 0;
    for (int64_t d = 0; d < M; ++d) if (in_piece(d)) ++ndocs;
    // postings per chunk (= per wave of the statistics pass): TMVB_CHUNK (256) for a corpus that fills the device with chunks of that size; a SMALL index (CTPF's
    // SYN-CITEU, an 8-GPU LDA shard: ~1.2 M postings = 4 800 chunks of 256 for 1 024 SIMDs that could hold 16 waves each) is cut finer, so that the gather-latency-bound
    // pass has more rows in flight (round 6; TMVB_CHUNK_SIZE overrides)
    static const int chunk_env = [] { const char* e = getenv("TMVB_CHUNK_SIZE"); return e ? std::max(16, atoi(e)) : 0; }();
    const int64_t n_post = nnz;                          // postings of this index (of this piece)
    const int64_t CHUNK = chunk_env ? chunk_env : (n_post < TMVB_SMALL_INDEX_POSTINGS ? TMVB_CHUNK_SMALL : TMVB_CHUNK);
    static const int class_env = [] { const char* e = getenv("TMVB_STATS_CLASSES"); return e ? atoi(e) : -1; }();
    static const int64_t class_docs = [] { const char* e = getenv("TMVB_CLASS_DOCS"); return e ? (int64_t)atoi(e) : (int64_t)TMVB_CLASS_DOCS; }();
    static const int64_t class_min = [] { const char* e = getenv("TMVB_CLASS_MIN_POSTINGS"); return e ? (int64_t)atoi(e) : (int64_t)TMVB_CLASS_MIN_POSTINGS; }();
    int NC = 8 * (int)((ndocs + 8 * class_docs - 1) / (8 * class_docs));
    if (class_env >= 0) NC = 8 * ((class_env + 7) / 8);               // classes are emitted in rounds of 8 (one per XCD): round up
    if (ndocs < 4 * class_docs && class_env < 0) NC = 0;               // everything fits a couple of L2s anyway
    // id_cuts: the same order built slice by slice of the id range, so that a slice is a contiguous run of chunks (and of multi-chunk ids)
    std::vector<int64_t> cuts;
    if (id_cuts) {
        cuts = *id_cuts;
        TMVB_REQUIRE(cuts.size() >= 2 && cuts.front() == 0 && cuts.back() == n_ids, TMVB_EINVAL, "inverted index: id cuts must run from 0 to the id count");
        for (size_t s = 0; s + 1 < cuts.size(); ++s) TMVB_REQUIRE(cuts[s] <= cuts[s + 1], TMVB_EINVAL, "inverted index: id cuts must ascend");
    } else {
        cuts = {0, n_ids};
    }
    auto doc_class = [&](int64_t d) { return (int)(d * NC / M); };
    std::vector<Chunk> order;
    size_t total_chunks = 0;
    std::vector<int64_t> slice_chunk{0}, slice_multi{0};
    for (size_t sl = 0; sl + 1 < cuts.size(); ++sl) {
        std::vector<std::vector<Chunk>> cls((size_t)std::max(NC, 1));
        std::vector<Chunk> rare;
        for (int64_t j = cuts[sl]; j < cuts[sl + 1]; ++j) {
            const int64_t a = cnt[j], b = cnt[j + 1];
            if (b == a) continue;
            // segments: the id's postings cut at class boundaries (one segment if the id is rare or classes are off)
            std::vector<std::pair<int64_t, int>> seg;                             // (begin, class)
            if (NC > 0 && b - a >= class_min) {
                int prev = -1;
                for (int64_t q = a; q < b; ++q) {
                    const int c = doc_class(doc[q]);
                    if (c != prev) { seg.emplace_back(q, c); prev = c; }
                }
            } else {
                seg.emplace_back(a, -1);
            }
            int64_t nch = 0;
            for (size_t g = 0; g < seg.size(); ++g) {
                const int64_t sa = seg[g].first, sb = g + 1 < seg.size() ? seg[g + 1].first : b;
                nch += (sb - sa + CHUNK - 1) / CHUNK;
            }
            if (nch > 1) { mid.push_back((int32_t)j); mfirst.push_back((int32_t)slots); mcount.push_back((int32_t)nch); }
            for (size_t g = 0; g < seg.size(); ++g) {
                const int64_t sa = seg[g].first, sb = g + 1 < seg.size() ? seg[g + 1].first : b;
                for (int64_t q = sa; q < sb; q += CHUNK) {
                    const Chunk ch{(int32_t)j, (int32_t)q, (int32_t)std::min<int64_t>(sb, q + CHUNK), nch > 1 ? (int32_t)(slots++) : -1};
                    if (seg[g].second >= 0) cls[(size_t)seg[g].second].push_back(ch); else rare.push_back(ch);
                }
            }
        }
        total_chunks += rare.size();
        for (const auto& L : cls) total_chunks += L.size();
        size_t rare_used = 0;
        if (NC > 0) {
            for (auto& L : cls)                                                   // whole workgroups per class: top up with rare chunks
                while (L.size() % 4 != 0 && rare_used < rare.size()) L.push_back(rare[rare_used++]);
            for (int r = 0; r < NC / 8; ++r) {
                size_t maxb = 0;
                for (int x = 0; x < 8; ++x) maxb = std::max(maxb, (cls[(size_t)(8 * r + x)].size() + 3) / 4);
                for (size_t k = 0; k < maxb; ++k)
                    for (int x = 0; x < 8; ++x) {
                        const auto& L = cls[(size_t)(8 * r + x)];
                        for (size_t u = 4 * k; u < std::min(L.size(), 4 * k + 4); ++u) order.push_back(L[u]);
                    }
            }
        }
        for (; rare_used < rare.size(); ++rare_used) order.push_back(rare[rare_used]);
        slice_chunk.push_back((int64_t)order.size()); slice_multi.push_back((int64_t)mid.size());
    }
    if (id_cuts) { ix->slice_id = cuts; ix->slice_chunk = slice_chunk; ix->slice_multi = slice_multi; }
    else { ix->slice_id.clear(); ix->slice_chunk.clear(); ix->slice_multi.clear(); }
    // every chunk holds postings (and multi-chunk ids a partial-sum slot): one left out of the launch order would drop them silently
    TMVB_REQUIRE(order.size() == total_chunks, TMVB_EINVAL, "statistics chunk order holds %zu of %zu chunks (NC = %d)", order.size(), total_chunks, NC);
    std::vector<int32_t> cid(order.size()), cb(order.size()), ce(order.size()), co(order.size());
    for (size_t q = 0; q < order.size(); ++q) { cid[q] = order[q].id; cb[q] = order[q].b; ce[q] = order[q].e; co[q] = order[q].out; }
    ix->n_chunks = (int64_t)cid.size(); ix->n_multi = (int64_t)mid.size(); ix->n_slots = slots;
    int rc;
    if ((rc = upload(ctx, &ix->d_doc, doc.data(), doc.size())) || (rc = upload(ctx, &ix->d_pos, pos.data(), pos.size())) ||
        (rc = upload(ctx, &ix->d_inv, inv.data(), inv.size())) || (rc = upload(ctx, &ix->d_val, val.data(), val.size())) ||
        (rc = upload(ctx, &ix->d_chunk_id, cid.data(), cid.size())) || (rc = upload(ctx, &ix->d_chunk_begin, cb.data(), cb.size())) ||
        (rc = upload(ctx, &ix->d_chunk_end, ce.data(), ce.size())) || (rc = upload(ctx, &ix->d_chunk_out, co.data(), co.size())) ||
        (rc = upload(ctx, &ix->d_multi_id, mid.data(), mid.size())) || (rc = upload(ctx, &ix->d_multi_first, mfirst.data(), mfirst.size())) ||
        (rc = upload(ctx, &ix->d_multi_count, mcount.data(), mcount.size())))
        return rc;
    TMVB_HIP(hipStreamSynchronize(ctx->stream));
    ix->built = true;
    return TMVB_OK;
}

void tmvb_free_inv_index(tmvb_inv_index* ix)
{
    (void)hipFree(ix->d_doc); (void)hipFree(ix->d_pos); (void)hipFree(ix->d_inv); (void)hipFree(ix->d_val); (void)hipFree(ix->d_chunk_id); (void)hipFree(ix->d_chunk_begin);
    (void)hipFree(ix->d_chunk_end); (void)hipFree(ix->d_chunk_out); (void)hipFree(ix->d_multi_id);
    (void)hipFree(ix->d_multi_first); (void)hipFree(ix->d_multi_count);
    *ix = tmvb_inv_index();
}

int tmvb_corpus_term_index(tmvb_corpus* c)
{
    if (c->term_index.built) return TMVB_OK;
    (void)hipSetDevice(c->ctx->device);
    return tmvb_build_inv_index(c->ctx, c->info.M, c->info.V, c->h_doc_ptr.data(), c->h_terms.data(), c->h_counts.data(), &c->term_index);
}

int tmvb_corpus_reader_index(tmvb_corpus* c)
{
    if (c->reader_index.built) return TMVB_OK;
    (void)hipSetDevice(c->ctx->device);
    return tmvb_build_inv_index(c->ctx, c->info.M, c->info.U, c->h_rdr_ptr.data(), c->h_readers.data(), c->h_ratings.data(), &c->reader_index);
}

extern "C" int tmvb_corpus_destroy(tmvb_corpus* c)
{
    if (!c) return TMVB_OK;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    (void)hipFree(c->d_doc_ptr); (void)hipFree(c->d_terms); (void)hipFree(c->d_counts);
    (void)hipFree(c->d_rdr_ptr); (void)hipFree(c->d_readers); (void)hipFree(c->d_ratings);
    (void)hipFree(c->d_doc_order);
    tmvb_free_inv_index(&c->term_index);
    tmvb_free_inv_index(&c->reader_index);
    delete c;
    return TMVB_OK;
}

extern "C" int tmvb_corpus_info(const tmvb_corpus* c, tmvb_corpus_info_t* out)
{
    TMVB_REQUIRE(c != nullptr && out != nullptr, TMVB_EINVAL, "tmvb_corpus_info: NULL argument");
    *out = c->info;
    return TMVB_OK;
}

// ------------------------------------------------------------------------------ special-function diagnostics
__global__ __launch_bounds__(256) void special_f32_kernel(int which, const float* __restrict__ x, float* __restrict__ y, int64_t n)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const float v = x[q];
    y[q] = which == 0 ? digamma_f(v) : which == 1 ? fast_exp(v) : fast_rcp(v);
}

extern "C" int tmvb_special_f32(tmvb_ctx* ctx, int32_t which, const float* x, float* y, int64_t n)
{
    TMVB_REQUIRE(ctx && x && y && n >= 0 && which >= 0 && which <= 2, TMVB_EINVAL, "tmvb_special_f32: bad argument");
    if (n == 0) return TMVB_OK;
    TMVB_HIP(hipSetDevice(ctx->device));
    float *dx = nullptr, *dy = nullptr;
    TMVB_HIP(hipMalloc((void**)&dx, (size_t)n * sizeof(float)));
    if (hipMalloc((void**)&dy, (size_t)n * sizeof(float)) != hipSuccess) { (void)hipFree(dx); tmvb_set_error("tmvb_special_f32: hipMalloc failed"); return TMVB_ENOMEM; }
    hipError_t e = hipMemcpyAsync(dx, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(special_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, which, dx, dy, n);
        e = hipMemcpyAsync(y, dy, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dx); (void)hipFree(dy);
    TMVB_REQUIRE(e == hipSuccess, TMVB_EHIP, "tmvb_special_f32: %s", hipGetErrorString(e));
    return TMVB_OK;
}

// ------------------------------------------------------------------------------ docfile ingest
// readcorp's document loop (src/Corpus.jl:285-299) as one streaming pass over the file: no per-document
// objects, integers parsed in place, output directly in the packed CSR the engine uploads.
static bool parse_int_line(const char* b, const char* e, char delim, std::vector<int64_t>& out)
{
    out.clear();
    const char* q = b;
    while (q < e && (*q == ' ' || *q == '\r')) ++q;
    if (q == e) return true;                                   // empty line = empty list
    for (;;) {
        while (q < e && *q == ' ') ++q;
        bool neg = false;
        if (q < e && (*q == '-' || *q == '+')) { neg = (*q == '-'); ++q; }
        if (q == e || *q < '0' || *q > '9') return false;
        int64_t v = 0;
        while (q < e && *q >= '0' && *q <= '9') { v = v * 10 + (*q - '0'); if (v > (int64_t)1 << 40) return false; ++q; }
        out.push_back(neg ? -v : v);
        while (q < e && (*q == ' ' || *q == '\r')) ++q;
        if (q == e) return true;
        if (*q != delim) return false;
        ++q;
    }
}

extern "C" void tmvb_docfile_free(tmvb_docfile_t* f)
{
    if (!f) return;
    free(f->doc_ptr); free(f->terms); free(f->counts); free(f->rdr_ptr); free(f->readers); free(f->ratings);
    memset(f, 0, sizeof(*f));
}

extern "C" int tmvb_docfile_read(const char* path, char delim, int32_t counts, int32_t readers, int32_t ratings, int32_t condense,
                                 tmvb_docfile_t* out)
{
    TMVB_REQUIRE(path && out, TMVB_EINVAL, "tmvb_docfile_read: NULL argument");
    memset(out, 0, sizeof(*out));
    if (ratings && !readers) ratings = 0;                       // "ratings require readers" (src/Corpus.jl:278)
    FILE* fp = fopen(path, "rb");
    TMVB_REQUIRE(fp != nullptr, TMVB_ECORPUS, "tmvb_docfile_read: cannot open %s", path);
    std::vector<char> buf;
    {
        fseek(fp, 0, SEEK_END);
        const long sz = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        buf.resize(sz > 0 ? (size_t)sz : 0);
        const size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), fp);
        fclose(fp);
        TMVB_REQUIRE(got == buf.size(), TMVB_ECORPUS, "tmvb_docfile_read: short read of %s", path);
    }
    const int block = 1 + (counts != 0) + (readers != 0) + (ratings != 0);
    std::vector<int64_t> doc_ptr{0}, rdr_ptr{0};
    std::vector<int32_t> terms, cnts, rdrs, rats;
    std::vector<int64_t> line[4];
    std::vector<std::pair<int64_t, int64_t>> tc;
    const char* q = buf.data();
    const char* end = q + buf.size();
    int64_t d = 0, lineno = 0, max_t = -1, max_u = -1;
    while (q < end) {
        ++d;
        const int64_t first_line = lineno + 1;
        bool ok = true;
        int have = 0;
        for (int l = 0; l < block && q < end; ++l, ++have) {
            const char* nl = (const char*)memchr(q, '\n', (size_t)(end - q));
            const char* le = nl ? nl : end;
            ok = parse_int_line(q, le, delim, line[l]) && ok;
            q = nl ? nl + 1 : end;
            ++lineno;
        }
        for (int l = have; l < block; ++l) line[l].clear();     // a truncated last block: missing lines are empty (zip semantics)
        const std::vector<int64_t>& T = line[0];
        int li = 1;
        const std::vector<int64_t>* C = counts ? &line[li++] : nullptr;
        const std::vector<int64_t>* R = readers ? &line[li++] : nullptr;
        const std::vector<int64_t>* G = ratings ? &line[li++] : nullptr;
        // check_doc (src/Corpus.jl:41-50): positive ids / counts / ratings, equal lengths
        ok = ok && (!C || C->size() == T.size()) && (!G || (R && G->size() == R->size()));
        for (int64_t v : T) ok = ok && v > 0 && v <= INT32_MAX;
        if (C) for (int64_t v : *C) ok = ok && v > 0 && v <= INT32_MAX;
        if (R) for (int64_t v : *R) ok = ok && v > 0 && v <= INT32_MAX;
        if (G) for (int64_t v : *G) ok = ok && v > 0 && v <= INT32_MAX;
        if (!ok) {
            tmvb_set_error("document %lld beginning on line %lld failed to load.", (long long)d, (long long)first_line);   // :295
            return TMVB_ECORPUS;
        }
        if (condense) {
            tc.clear();
            for (size_t n = 0; n < T.size(); ++n) tc.emplace_back(T[n] - 1, C ? (*C)[n] : 1);
            std::sort(tc.begin(), tc.end());
            for (size_t n = 0; n < tc.size();) {
                int64_t cs = 0; size_t m = n;
                while (m < tc.size() && tc[m].first == tc[n].first) cs += tc[m++].second;
                terms.push_back((int32_t)tc[n].first); cnts.push_back((int32_t)std::min<int64_t>(cs, INT32_MAX));
                n = m;
            }
        } else {
            for (size_t n = 0; n < T.size(); ++n) { terms.push_back((int32_t)(T[n] - 1)); cnts.push_back(C ? (int32_t)(*C)[n] : 1); }
        }
        for (int64_t v : T) max_t = std::max(max_t, v - 1);
        if (R) for (size_t n = 0; n < R->size(); ++n) {
            rdrs.push_back((int32_t)((*R)[n] - 1)); rats.push_back(G ? (int32_t)(*G)[n] : 1);
            max_u = std::max(max_u, (*R)[n] - 1);
        }
        doc_ptr.push_back((int64_t)terms.size());
        rdr_ptr.push_back((int64_t)rdrs.size());
    }
    auto give = [](auto& v, auto** dst) -> bool {
        typedef typename std::remove_reference<decltype(v[0])>::type E;
        *dst = (E*)malloc(std::max<size_t>(v.size(), 1) * sizeof(E));
        if (!*dst) return false;
        if (!v.empty()) memcpy(*dst, v.data(), v.size() * sizeof(E));
        return true;
    };
    out->M = (int64_t)doc_ptr.size() - 1; out->nnz = (int64_t)terms.size(); out->nR = (int64_t)rdrs.size();
    out->V_seen = max_t + 1; out->U_seen = max_u + 1;
    if (!give(doc_ptr, &out->doc_ptr) || !give(terms, &out->terms) || !give(cnts, &out->counts) || !give(rdr_ptr, &out->rdr_ptr) ||
        !give(rdrs, &out->readers) || !give(rats, &out->ratings)) {
        tmvb_docfile_free(out);
        tmvb_set_error("tmvb_docfile_read: out of host memory");
        return TMVB_ENOMEM;
    }
    return TMVB_OK;
}
