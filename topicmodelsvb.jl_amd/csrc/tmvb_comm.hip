// tmvb_comm.hip -- the collective of the document-sharded runs, behind the C ABI.
//
// The reference is single-device (one OpenCL queue, src/gpuLDA.jl:64); its only multi-batch precedent is the v0.6
// `newbeta +=` accumulation (v0.6/src/gpuLDA.jl:200-225).  Here documents shard across the GPUs of one node and the one
// exchange per outer iteration is a sum-all-reduce of the packed sufficient statistics.
// Backends: RCCL (bound with dlopen at the first communicator call; ncclCommInitRank with a host-broadcast unique id for one process per GPU, ncclCommInitAll for one host
// thread driving n GPUs) and a host-callback transport that stages the buffer through pinned memory (MPI / gloo hosts,
// single-GPU tests).  Host code only; no kernels.
#include "tmvb_internal.h"

#include <rccl/rccl.h>          // types and NCCL_VERSION_CODE only: the library itself is loaded on first use (below)

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>

struct tmvb_comm {
    tmvb_ctx* ctx = nullptr;
    int nranks = 1, rank = 0;
    int backend = 0;                      // 0 = RCCL, 1 = host callback
    ncclComm_t nccl = nullptr;
    tmvb_host_allreduce_fn fn = nullptr;
    void* user = nullptr;
    void* pinned = nullptr;               // host staging buffer (backend 1)
    size_t pinned_bytes = 0;
};

// RCCL is bound at the first communicator call, not at load time: single-GPU use of libtmvb_hip.so needs no RCCL at all, and a
// process may hold another RCCL already (PyTorch ships its own librccl with the same SONAME): the copy next to the HIP runtime
// in use is taken, privately (RTLD_LOCAL).  The header this file was compiled against and the runtime must agree in their
// major version (struct layouts, enum values); checked once.
namespace {
struct RcclApi {
    void* so = nullptr;
    int version = 0;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

void rccl_bind()
{
    RcclApi& a = g_rccl;
    // dlerror() clears the error it returns: read it ONCE, right after the failed dlopen it belongs to
    std::string why;
    auto open = [&](const char* path) {
        a.so = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!a.so) { const char* de = dlerror(); why = de ? de : (std::string(path) + " not found"); }
    };
    if (const char* e = getenv("TMVB_RCCL_LIB")) {
        // an explicit choice is used as given: if it does not load, say so instead of silently binding another copy
        open(e);
        if (!a.so) { a.error = "RCCL is not available: TMVB_RCCL_LIB=" + std::string(e) + ": " + why; return; }
    }
    // first choice: the RCCL that sits next to the HIP runtime this process really runs on (found through dladdr) -- an RCCL
    // built for another HIP release fails in ncclCommInitRank with "unhandled cuda error" (seen with PyTorch's bundled copy
    // under /opt/rocm's runtime); RTLD_LOCAL + dlsym on our own handle keep a second copy in the process from interfering
    if (!a.so) {
        Dl_info info;
        if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string path(info.dli_fname);
            const size_t slash = path.find_last_of('/');
            if (slash != std::string::npos) {
                path = path.substr(0, slash + 1) + "librccl.so.1";
                open(path.c_str());
            }
        }
    }
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};      // then the loader's search path (our rpath)
    for (const char* n : names) {
        if (a.so) break;
        open(n);
    }
    if (!a.so) { a.error = "RCCL is not available: " + (why.empty() ? std::string("librccl.so.1 not found") : why) + " (set TMVB_RCCL_LIB)"; return; }
#define TMVB_RCCL_SYM(field, sym)                                                         \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.so, #sym));                     \
    if (!a.field) { a.error = "librccl has no symbol " #sym; return; }
    TMVB_RCCL_SYM(GetVersion, ncclGetVersion) TMVB_RCCL_SYM(GetUniqueId, ncclGetUniqueId) TMVB_RCCL_SYM(CommInitRank, ncclCommInitRank)
    TMVB_RCCL_SYM(CommInitAll, ncclCommInitAll) TMVB_RCCL_SYM(CommDestroy, ncclCommDestroy) TMVB_RCCL_SYM(AllReduce, ncclAllReduce)
    TMVB_RCCL_SYM(GroupStart, ncclGroupStart) TMVB_RCCL_SYM(GroupEnd, ncclGroupEnd) TMVB_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef TMVB_RCCL_SYM
    if (a.GetVersion(&a.version) != ncclSuccess) { a.error = "ncclGetVersion failed"; return; }
    // version code: major * 10000 + minor * 100 + patch (NCCL >= 2.9)
    if (a.version / 10000 != NCCL_VERSION_CODE / 10000) {
        char buf[160];
        snprintf(buf, sizeof buf, "RCCL runtime version %d does not match the header libtmvb_hip.so was built against (%d): major versions differ",
                 a.version, (int)NCCL_VERSION_CODE);
        a.error = buf;
    }
}

// TMVB_OK when RCCL is bound and version-compatible, otherwise TMVB_ERCCL with the reason
int rccl_ready()
{
    std::call_once(g_rccl_once, rccl_bind);
    if (!g_rccl.error.empty()) { tmvb_set_error("%s", g_rccl.error.c_str()); return TMVB_ERCCL; }
    return TMVB_OK;
}
}  // namespace

#define TMVB_NCCL(call)                                                                         \
    do {                                                                                        \
        ncclResult_t r_ = (g_rccl.call);                                                        \
        if (r_ != ncclSuccess) {                                                                \
            tmvb_set_error("nccl%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__);    \
            return TMVB_ERCCL;                                                                  \
        }                                                                                       \
    } while (0)

extern "C" int tmvb_rccl_version(void)
{
    return rccl_ready() == TMVB_OK ? g_rccl.version : 0;
}

extern "C" int tmvb_comm_unique_id(void* id_out)
{
    TMVB_REQUIRE(id_out != nullptr, TMVB_EINVAL, "tmvb_comm_unique_id: id_out is NULL");
    static_assert(sizeof(ncclUniqueId) == TMVB_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId id;
    TMVB_NCCL(GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return TMVB_OK;
}

extern "C" int tmvb_comm_create_rccl(tmvb_ctx* ctx, const void* unique_id, int32_t nranks, int32_t rank, tmvb_comm** out)
{
    TMVB_REQUIRE(out != nullptr, TMVB_EINVAL, "tmvb_comm_create_rccl: out is NULL");
    *out = nullptr;
    TMVB_REQUIRE(ctx && unique_id, TMVB_EINVAL, "tmvb_comm_create_rccl: NULL context or unique id");
    TMVB_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, TMVB_EINVAL, "tmvb_comm_create_rccl: rank %d not in [0,%d)", rank, nranks);
    if (int rc = rccl_ready()) return rc;
    TMVB_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    TMVB_NCCL(CommInitRank(&c, nranks, id, rank));
    tmvb_comm* h = new tmvb_comm();
    h->ctx = ctx; h->nranks = nranks; h->rank = rank; h->backend = 0; h->nccl = c;
    *out = h;
    return TMVB_OK;
}

extern "C" int tmvb_comm_create_rccl_all(tmvb_ctx* const* ctxs, int32_t n, tmvb_comm** out)
{
    TMVB_REQUIRE(ctxs && out && n >= 1, TMVB_EINVAL, "tmvb_comm_create_rccl_all: bad argument");
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        TMVB_REQUIRE(ctxs[i] != nullptr, TMVB_EINVAL, "tmvb_comm_create_rccl_all: context %d is NULL", i);
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; ++j)
            TMVB_REQUIRE(devs[j] != devs[i], TMVB_EINVAL, "tmvb_comm_create_rccl_all: device %d appears twice (one rank per GPU)", devs[i]);
        out[i] = nullptr;
    }
    if (int rc = rccl_ready()) return rc;
    std::vector<ncclComm_t> comms(n, nullptr);
    TMVB_NCCL(CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        tmvb_comm* h = new tmvb_comm();
        h->ctx = ctxs[i]; h->nranks = n; h->rank = i; h->backend = 0; h->nccl = comms[i];
        out[i] = h;
    }
    return TMVB_OK;
}

extern "C" int tmvb_comm_create_host(tmvb_ctx* ctx, int32_t nranks, int32_t rank, tmvb_host_allreduce_fn fn, void* user, tmvb_comm** out)
{
    TMVB_REQUIRE(out != nullptr, TMVB_EINVAL, "tmvb_comm_create_host: out is NULL");
    *out = nullptr;
    TMVB_REQUIRE(ctx && fn, TMVB_EINVAL, "tmvb_comm_create_host: NULL context or callback");
    TMVB_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, TMVB_EINVAL, "tmvb_comm_create_host: rank %d not in [0,%d)", rank, nranks);
    tmvb_comm* h = new tmvb_comm();
    h->ctx = ctx; h->nranks = nranks; h->rank = rank; h->backend = 1; h->fn = fn; h->user = user;
    *out = h;
    return TMVB_OK;
}

extern "C" int tmvb_comm_destroy(tmvb_comm* c)
{
    if (!c) return TMVB_OK;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (c->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl);
    if (c->pinned) (void)hipHostFree(c->pinned);
    delete c;
    return TMVB_OK;
}

extern "C" int tmvb_comm_info(const tmvb_comm* c, int32_t* nranks, int32_t* rank, int32_t* backend)
{
    TMVB_REQUIRE(c != nullptr, TMVB_EINVAL, "tmvb_comm_info: comm is NULL");
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    if (backend) *backend = c->backend;
    return TMVB_OK;
}

static int comm_allreduce_one(tmvb_comm* c, void* dev_ptr, int64_t count, int32_t dtype, hipStream_t on = nullptr)
{
    TMVB_REQUIRE(c && dev_ptr, TMVB_EINVAL, "tmvb_comm_allreduce: NULL argument");
    TMVB_REQUIRE(count >= 0 && (dtype == TMVB_F32 || dtype == TMVB_F64), TMVB_EINVAL, "tmvb_comm_allreduce: bad count or dtype");
    if (count == 0) return TMVB_OK;
    tmvb_ctx* ctx = c->ctx;
    TMVB_HIP(hipSetDevice(ctx->device));
    hipStream_t st = on ? on : ctx->stream;
    if (c->backend == 0) {
        TMVB_NCCL(AllReduce(dev_ptr, dev_ptr, (size_t)count, dtype == TMVB_F32 ? ncclFloat32 : ncclFloat64, ncclSum, c->nccl, st));
        return TMVB_OK;
    }
    const size_t bytes = (size_t)count * (dtype == TMVB_F32 ? 4 : 8);
    if (bytes > c->pinned_bytes) {
        if (c->pinned) (void)hipHostFree(c->pinned);
        c->pinned = nullptr; c->pinned_bytes = 0;
        hipError_t e = hipHostMalloc(&c->pinned, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { tmvb_set_error("hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); return TMVB_ENOMEM; }
        c->pinned_bytes = bytes;
    }
    TMVB_HIP(hipMemcpyAsync(c->pinned, dev_ptr, bytes, hipMemcpyDeviceToHost, st));
    TMVB_HIP(hipStreamSynchronize(st));
    const int rc = c->fn(c->user, c->pinned, count, dtype);
    TMVB_REQUIRE(rc == 0, TMVB_ERCCL, "tmvb_comm_allreduce: the host all-reduce callback returned %d", rc);
    TMVB_HIP(hipMemcpyAsync(dev_ptr, c->pinned, bytes, hipMemcpyHostToDevice, st));
    TMVB_HIP(hipStreamSynchronize(st));              // the staging buffer is reused by the next call
    return TMVB_OK;
}

extern "C" int tmvb_comm_allreduce(tmvb_comm* c, void* dev_ptr, int64_t count, int32_t dtype)
{
    return comm_allreduce_one(c, dev_ptr, count, dtype);
}

// the same collective on a stream of the caller's (the context's device): the vocabulary slabs of tmvb_lda_estep_allreduce go through
// a side stream while the statistics pass still runs on the context's
int tmvb_comm_allreduce_on(tmvb_comm* c, void* dev_ptr, int64_t count, int32_t dtype, hipStream_t on)
{
    return comm_allreduce_one(c, dev_ptr, count, dtype, on);
}

// n collectives issued by ONE host thread (ncclCommInitAll communicators): they must sit inside one RCCL group,
// otherwise the first call blocks waiting for peers that this same thread has not launched yet.
int tmvb_comm_allreduce_group(tmvb_comm* const* comms, void* const* dev_ptrs, const int64_t* counts, int n, int32_t dtype)
{
    if (n == 1) return comm_allreduce_one(comms[0], dev_ptrs[0], counts[0], dtype);
    bool all_rccl = true;
    for (int i = 0; i < n; ++i) {
        TMVB_REQUIRE(comms[i] != nullptr, TMVB_EINVAL, "grouped all-reduce: handle %d has no communicator", i);
        all_rccl = all_rccl && comms[i]->backend == 0;
    }
    TMVB_REQUIRE(all_rccl, TMVB_EINVAL, "grouped all-reduce over several local handles needs RCCL communicators (tmvb_comm_create_rccl_all)");
    TMVB_NCCL(GroupStart());
    int rc = TMVB_OK;
    for (int i = 0; i < n && rc == TMVB_OK; ++i) rc = comm_allreduce_one(comms[i], dev_ptrs[i], counts[i], dtype);
    ncclResult_t r = g_rccl.GroupEnd();
    if (rc) return rc;
    if (r != ncclSuccess) { tmvb_set_error("ncclGroupEnd failed: %s", g_rccl.GetErrorString(r)); return TMVB_ERCCL; }
    return TMVB_OK;
}

// ---- the plan of a sliced all-reduce (tmvb_lda_estep_allreduce): pure host arithmetic on the GLOBAL postings per term, so that every
// rank derives the same slabs in the same order.  Cuts: equal shares of  postings(v) / nnz + 1 / V  (half a slice's weight is pass
// time, half is bytes on the wire).  Order: Johnson's rule for the two-machine flow shop (statistics pass, then collective) -- slices
// with less pass than wire first, cheapest pass first; then the others, most wire first.
extern "C" int tmvb_allreduce_plan(const double* counts, int64_t V, int32_t want, int64_t* cuts, int32_t* order, int32_t* slices)
{
    TMVB_REQUIRE(counts && cuts && order && slices && V >= 0 && want >= 1, TMVB_EINVAL, "tmvb_allreduce_plan: bad argument");
    int S = (int)std::min<int64_t>((int64_t)want, std::max<int64_t>(V, 1));
    double tot = 0.0;
    for (int64_t v = 0; v < V; ++v) { TMVB_REQUIRE(counts[v] >= 0.0, TMVB_EINVAL, "tmvb_allreduce_plan: negative count"); tot += counts[v]; }
    if (tot <= 0.0) S = 1;
    std::vector<int64_t> c(1, 0);
    if (S > 1) {
        double run = 0.0;
        for (int64_t v = 0; v < V; ++v) {
            run += 0.5 * counts[v] / tot + 0.5 / (double)V;
            while ((int)c.size() < S && run >= (double)c.size() / (double)S) c.push_back(v + 1);
        }
        while ((int)c.size() < S) c.push_back(V);
    }
    c.push_back(V);
    std::vector<int> ord;
    if (S > 1) {
        std::vector<double> pass((size_t)S, 0.0), wire((size_t)S, 0.0);
        for (int sl = 0; sl < S; ++sl) {
            for (int64_t v = c[(size_t)sl]; v < c[(size_t)sl + 1]; ++v) pass[(size_t)sl] += counts[v] / tot;
            wire[(size_t)sl] = (double)(c[(size_t)sl + 1] - c[(size_t)sl]) / (double)V;
        }
        std::vector<int> first, second;
        for (int sl = 0; sl < S; ++sl) (pass[(size_t)sl] < wire[(size_t)sl] ? first : second).push_back(sl);
        std::stable_sort(first.begin(), first.end(), [&](int a, int b) { return pass[(size_t)a] < pass[(size_t)b]; });
        std::stable_sort(second.begin(), second.end(), [&](int a, int b) { return wire[(size_t)a] > wire[(size_t)b]; });
        ord = first;
        ord.insert(ord.end(), second.begin(), second.end());
    } else {
        ord.assign(1, 0);
    }
    for (int sl = 0; sl <= S; ++sl) cuts[sl] = c[(size_t)sl];
    for (int sl = 0; sl < S; ++sl) order[sl] = ord[(size_t)sl];
    *slices = S;
    return TMVB_OK;
}
