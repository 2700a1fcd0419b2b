// RCCL from inside the library: the data-parallel step's two collectives as stream-ordered calls of the native launch
// loop (sgcn_step_run ops ALLREDUCE_AVG / ALLGATHER_I32) instead of Python calls between three program runs.
//
// Why: with torch.distributed the launching thread pays ~80 us per step for the collectives' bookkeeping (c10d call, work
// objects, three slice assignments to pack the history rows, one scatter call per rank) -- measured with a forced ONE-rank
// process group: 0.130 -> 0.211 ms per Reddit CVD+PP step, host-bound (profiles/r44_epoch_fixed_cost.jsonl).  An 8-GPU
// epoch is 38 such steps.  The reference has nothing here (single session, gcn/train.py:130); SURVEY.md 8e.
//
// The library is loaded at run time (dlopen: the process usually has torch's own librccl.so mapped already, and the same
// soname resolves to that copy), the communicator is the library's own: rank 0 draws the id (sgcn_coll_unique_id), the
// host side broadcasts its 128 bytes over the job's existing process group, every rank calls sgcn_coll_init.
#include "sgcn_host.h"
#include "../../include/sgcn.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace {

// The handful of RCCL declarations this file needs, restated (rccl.h is not included: the library is bound at run time).
// They are ABI constants of NCCL >= 2.10 -- ncclAvg = 4 arrived with 2.10, NCCL_UNIQUE_ID_BYTES = 128 and the data type
// codes are older -- and load() refuses a library whose ncclGetVersion reports less.
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
enum { kNcclSuccess = 0, kNcclInProgress = 7, kNcclInt32 = 2, kNcclFloat32 = 7, kNcclSum = 0, kNcclAvg = 4 };
constexpr int kMinNcclVersion = 21000;        // NCCL_VERSION(2, 10, 0) = 2 * 10000 + 10 * 100

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*CommAbort)(rcclComm_t) = nullptr;
    int (*CommGetAsyncError)(rcclComm_t, int*) = nullptr;
    rcclComm_t comm = nullptr;
    // The EXCHANGE communicator (round 6): the step's history all-gather runs on the library's exchange stream beside the
    // loss, the backward pass and the gradient all-reduce (csrc/sgcn_step.cpp).  One communicator used from two streams in
    // turn makes RCCL order the two streams itself (measured in round 5: 0.177 ms per step and erratic); a communicator of
    // its own per stream does not.
    rcclComm_t xcomm = nullptr;
    int world = 0, rank = -1, users = 0, version = 0;
};

Rccl& R() { static Rccl r; return r; }
std::mutex& mu() { static std::mutex m; return m; }

int load() {
    Rccl& r = R();
    if (r.lib) return SGCN_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {                           // the copy the process has mapped already, if any
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) return sgcn::fail(SGCN_ERR_INVALID, "coll: librccl.so not found (%s)", dlerror());
#define SGCN_SYM(field, name)                                                                               \
    *(void**)(&r.field) = dlsym(h, name);                                                                   \
    if (!r.field) { dlclose(h); return sgcn::fail(SGCN_ERR_INVALID, "coll: librccl.so lacks %s", name); }
    SGCN_SYM(GetUniqueId, "ncclGetUniqueId")
    SGCN_SYM(CommInitRank, "ncclCommInitRank")
    SGCN_SYM(CommDestroy, "ncclCommDestroy")
    SGCN_SYM(AllReduce, "ncclAllReduce")
    SGCN_SYM(AllGather, "ncclAllGather")
    SGCN_SYM(GetErrorString, "ncclGetErrorString")
    SGCN_SYM(GetVersion, "ncclGetVersion")
    SGCN_SYM(CommAbort, "ncclCommAbort")
    SGCN_SYM(CommGetAsyncError, "ncclCommGetAsyncError")
#undef SGCN_SYM
    int v = 0;
    if (r.GetVersion(&v) != kNcclSuccess || v < kMinNcclVersion) {
        dlclose(h);
        return sgcn::fail(SGCN_ERR_INVALID, "coll: librccl.so reports NCCL version code %d; ncclAvg needs >= %d (2.10)", v, kMinNcclVersion);
    }
    r.version = v;
    r.lib = h;
    return SGCN_OK;
}

int check(int rc, const char* what) {
    if (rc == kNcclSuccess) return SGCN_OK;
    return sgcn::fail(SGCN_ERR_HIP, "coll: %s: %s", what, R().GetErrorString ? R().GetErrorString(rc) : "?");
}

}  // namespace

extern "C" {

// Pure probe: can this process bind RCCL (library found, every symbol present, version >= 2.10)?  No unique id is drawn --
// ncclGetUniqueId starts a bootstrap listener thread and socket that live as long as the process -- so every
// rank may call this; only rank 0 calls sgcn_coll_unique_id.
int sgcn_coll_available(int32_t* version_code) {
    std::lock_guard<std::mutex> lk(mu());
    const int rc = load();
    if (version_code) *version_code = rc == SGCN_OK ? R().version : 0;
    return rc;
}

int sgcn_coll_unique_id(void* out128) {
    if (!out128) return sgcn::fail(SGCN_ERR_INVALID, "coll_unique_id: null buffer");
    std::lock_guard<std::mutex> lk(mu());
    const int rc = load();
    if (rc != SGCN_OK) return rc;
    rcclUniqueId id;
    const int e = check(R().GetUniqueId(&id), "ncclGetUniqueId");
    if (e != SGCN_OK) return e;
    std::memcpy(out128, &id, sizeof(id));
    return SGCN_OK;
}

int sgcn_coll_init(const void* id128, int32_t world, int32_t rank) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) return sgcn::fail(SGCN_ERR_INVALID, "coll_init: bad argument");
    std::lock_guard<std::mutex> lk(mu());
    const int rc = load();
    if (rc != SGCN_OK) return rc;
    Rccl& r = R();
    if (r.comm) return sgcn::fail(SGCN_ERR_INVALID, "coll_init: a communicator exists (sgcn_coll_destroy first)");
    rcclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    rcclComm_t c = nullptr;
    const int e = check(r.CommInitRank(&c, world, id, rank), "ncclCommInitRank");
    if (e != SGCN_OK) return e;
    r.comm = c; r.world = world; r.rank = rank; r.users = 1;
    return SGCN_OK;
}

int sgcn_coll_world(void) { return R().comm ? R().world : 0; }

// The second communicator of the same ranks, for collectives on the exchange stream (sgcn_coll_allgather_x_i32).  Its id
// is drawn by rank 0 like the first (a fresh sgcn_coll_unique_id) and every rank calls this behind sgcn_coll_init; it
// shares the first one's user count and is destroyed / aborted with it.  Optional: without it the exchange uses the first.
int sgcn_coll_init_exchange(const void* id128) {
    if (!id128) return sgcn::fail(SGCN_ERR_INVALID, "coll_init_exchange: bad argument");
    std::lock_guard<std::mutex> lk(mu());
    Rccl& r = R();
    if (!r.comm) return sgcn::fail(SGCN_ERR_INVALID, "coll_init_exchange: no communicator (sgcn_coll_init first)");
    if (r.xcomm) return sgcn::fail(SGCN_ERR_INVALID, "coll_init_exchange: an exchange communicator exists");
    rcclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    rcclComm_t c = nullptr;
    const int e = check(r.CommInitRank(&c, r.world, id, r.rank), "ncclCommInitRank (exchange)");
    if (e != SGCN_OK) return e;
    r.xcomm = c;
    return SGCN_OK;
}

int sgcn_coll_has_exchange(void) { return R().xcomm ? 1 : 0; }

// The communicator is process-global; a second user of it (another DataParallel object of the same job) retains it and
// the LAST sgcn_coll_destroy destroys it (the owner's shutdown used to destroy it under the others).
int sgcn_coll_retain(void) {
    std::lock_guard<std::mutex> lk(mu());
    Rccl& r = R();
    if (!r.comm) return sgcn::fail(SGCN_ERR_INVALID, "coll_retain: no communicator (sgcn_coll_init)");
    r.users++;
    return SGCN_OK;
}

int sgcn_coll_destroy(void) {
    std::lock_guard<std::mutex> lk(mu());
    Rccl& r = R();
    if (!r.comm) return SGCN_OK;
    if (--r.users > 0) return SGCN_OK;
    int e = SGCN_OK;
    if (r.xcomm) e = check(r.CommDestroy(r.xcomm), "ncclCommDestroy (exchange)");
    const int e2 = check(r.CommDestroy(r.comm), "ncclCommDestroy");
    r.comm = r.xcomm = nullptr; r.world = 0; r.rank = -1; r.users = 0;
    return e != SGCN_OK ? e : e2;
}

// A rank that fails ahead of a collective leaves its peers blocked inside RCCL for good (no watchdog on this
// communicator, unlike c10d's): the failing rank ABORTS the communicator -- the peers' pending and later collectives then
// return an error instead of hanging -- whatever the user count.  Called by the host side on any failing step.
int sgcn_coll_abort(void) {
    std::lock_guard<std::mutex> lk(mu());
    Rccl& r = R();
    if (!r.comm) return SGCN_OK;
    int e = SGCN_OK;
    if (r.xcomm) e = check(r.CommAbort(r.xcomm), "ncclCommAbort (exchange)");
    const int e2 = check(r.CommAbort(r.comm), "ncclCommAbort");
    r.comm = r.xcomm = nullptr; r.world = 0; r.rank = -1; r.users = 0;
    return e != SGCN_OK ? e : e2;
}

// 0 = the communicator is healthy; an asynchronous error a peer's abort or a network fault left behind otherwise.
int sgcn_coll_async_error(void) {
    Rccl& r = R();
    if (!r.comm) return sgcn::fail(SGCN_ERR_INVALID, "coll_async_error: no communicator (sgcn_coll_init)");
    int err = kNcclSuccess;
    const int e = check(r.CommGetAsyncError(r.comm, &err), "ncclCommGetAsyncError");
    if (e != SGCN_OK) return e;
    if (err != kNcclSuccess && err != kNcclInProgress) return check(err, "asynchronous error on the communicator");
    if (r.xcomm) {
        const int ex = check(r.CommGetAsyncError(r.xcomm, &err), "ncclCommGetAsyncError (exchange)");
        if (ex != SGCN_OK) return ex;
        if (err != kNcclSuccess && err != kNcclInProgress) return check(err, "asynchronous error on the exchange communicator");
    }
    return SGCN_OK;
}

int sgcn_coll_allreduce_avg_f32(float* dev_buf, int64_t n, void* stream) {
    Rccl& r = R();
    if (!r.comm) return sgcn::fail(SGCN_ERR_INVALID, "coll_allreduce: no communicator (sgcn_coll_init)");
    if (n < 0 || (n > 0 && !dev_buf)) return sgcn::fail(SGCN_ERR_INVALID, "coll_allreduce: bad argument");
    if (n == 0) return SGCN_OK;
    return check(r.AllReduce(dev_buf, dev_buf, (size_t)n, kNcclFloat32, kNcclAvg, r.comm, stream), "ncclAllReduce");
}

int sgcn_coll_allgather_i32(const int32_t* dev_send, int32_t* dev_recv, int64_t n, void* stream) {
    Rccl& r = R();
    if (!r.comm) return sgcn::fail(SGCN_ERR_INVALID, "coll_allgather: no communicator (sgcn_coll_init)");
    if (n < 0 || (n > 0 && (!dev_send || !dev_recv))) return sgcn::fail(SGCN_ERR_INVALID, "coll_allgather: bad argument");
    if (n == 0) return SGCN_OK;
    return check(r.AllGather(dev_send, dev_recv, (size_t)n, kNcclInt32, r.comm, stream), "ncclAllGather");
}

// The same all-gather on the EXCHANGE communicator (the first one when there is no second): for a stream other than the
// one the gradient all-reduce runs on.
int sgcn_coll_allgather_x_i32(const int32_t* dev_send, int32_t* dev_recv, int64_t n, void* stream) {
    Rccl& r = R();
    if (!r.comm) return sgcn::fail(SGCN_ERR_INVALID, "coll_allgather_x: no communicator (sgcn_coll_init)");
    if (n < 0 || (n > 0 && (!dev_send || !dev_recv))) return sgcn::fail(SGCN_ERR_INVALID, "coll_allgather_x: bad argument");
    if (n == 0) return SGCN_OK;
    return check(r.AllGather(dev_send, dev_recv, (size_t)n, kNcclInt32, r.xcomm ? r.xcomm : r.comm, stream), "ncclAllGather (exchange)");
}

}  // extern "C"
