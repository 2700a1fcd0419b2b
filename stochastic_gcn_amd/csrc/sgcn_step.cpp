// Native launch loop of one training / evaluation step (include/sgcn.h sgcn_step_run).
//
// The reference runs a step as ONE sess.run of a static TensorFlow graph (gcn/vrgcn.py:72-82,
// gcn/train.py:187-209).  The eager host path of this build issues the same step as ~17 C-ABI calls
// from Python, and measures host-bound: 0.30 ms of interpreter time per step around 0.26 ms of GPU
// work (profiles/host_bound_probe.py).  Here the step is a PROGRAM -- a flat list of C-ABI calls whose
// arguments are affine in a small table of per-minibatch slots (row counts, addresses inside the
// minibatch's staging buffer, dropout keys, the Adam step size) -- compiled once per model by
// stochastic_gcn_amd/step_program.py and executed by ONE foreign call per step.  Every op is one of the
// library's own entry points, called with exactly the arguments the eager path would pass, so the two
// paths are bit-identical (tests/test_step_program_gpu.py).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "sgcn_host.h"
#include "sgcn_fuse.h"
#include "sgcn_bwd.h"
#include "../../include/sgcn.h"

namespace sgcn {
// sgcn_gemm.hip: sgcn_dense_bwd_f32 with its weight-gradient side (dW GEMM, split-K and LayerNorm-parameter
// reductions) on the library's auxiliary stream, and the join that makes `stream` wait for that work
int dense_bwd_overlapped(int32_t n, int32_t N, int32_t K, const float* dy, int64_t lddy, const float* y, int64_t ldy,
                         const float* xhat, const float* rstd, const float* scale, int32_t relu, const float* x,
                         int64_t ldx, const float* W, int64_t ldw, float* dW, int64_t lddw, float* doffset,
                         float* dscale, float* dx, int64_t lddx, const sgcn_dropout_t* drop, float* g_tmp, float* ws,
                         const int32_t* gidx, void* stream);
int aux_join(void* stream);
void step_mode_override(int overlap, int fuse);     // sgcn_spmm.hip: this thread's view of the step_overlap / step_fuse knobs
int dense_bwd_chain(const DenseBwdArgs& up, const DenseBwdArgs& lo, void* stream);   // sgcn_gemm.hip
int aux_fork(void* stream, void** aux_stream);
int spin_launch(void* stream, int64_t usec);     // sgcn_rows.hip
int dw_group_begin();                    // sgcn_gemm.hip: record the weight-gradient GEMMs of the following DENSE_BWD ops ...
int dw_group_flush(void* stream, bool park_reduce);   // ... and issue them as one grouped launch + one reduction launch
int reduce_flush(void* stream);          // sgcn_dense.hip: reductions parked for an optimizer launch that did not come
void dw_group_abort();
void grad_store_mode(int on);            // sgcn_gemm.hip: parameter gradients are stored, not added to a zeroed buffer
void stats_defer(int on);                // sgcn_dense.hip: the loss kernel's statistics reduction rides in the optimizer's launch
int stats_flush(void* stream);
int adam_with_stats(float* theta, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                    float eps, void* stream);
bool scatter_park(float* H, int64_t ldh, const int32_t* idx, int32_t n, int32_t d, const float* src, int64_t lds);
int ce_impl(bool softmax, const float* logits, int64_t ldz, const float* labels, int64_t ldl, int32_t n, int32_t c,
            float* dlogits, int64_t lddz, float* pred, int64_t ldp, float* stats, float* rowstat, void* stream, bool overlap,
            const CeLastLayer* last);
void gemm_fwd_shape(int M, int N, int K, int* S, int* kgroups, int* kchunk);       // sgcn_gemm.hip
int dense_fwd_pair(int32_t M, int32_t N, int32_t K, const float* X, int64_t ldx, const float* X2, int64_t ldx2, int32_t split,
                   const float* W, int64_t ldw, const float* offset, const float* scale, float eps, int32_t relu, float* Y,
                   int64_t ldy, float* xhat, float* rstd, const sgcn_dropout_t* drop, float* ws, const int32_t* gidx,
                   const int32_t* gidx2, int32_t N2, const float* W2, int64_t ldw2, const float* offset2, const float* scale2,
                   float eps2, int32_t relu2, float* Y2, int64_t ldy2, float* xhat2, float* rstd2, const sgcn_dropout_t* drop2,
                   void* stream, int* fused);
}  // namespace sgcn

namespace {

inline float f32(int64_t v) {
    const uint32_t b = (uint32_t)(uint64_t)v;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
template <class T> inline T* ptr(int64_t v) { return reinterpret_cast<T*>((uintptr_t)v); }

struct Args {
    int64_t v[SGCN_STEP_MAX_ARGS];
    int n, pos;
    int64_t next() { return pos < n ? v[pos++] : 0; }
    template <class T> T* p() { return ptr<T>(next()); }
    int32_t i() { return (int32_t)next(); }
    float f() { return f32(next()); }
    // {on, key, keep, rows, width} -> sgcn_dropout_t (nullptr when off)
    const sgcn_dropout_t* drop(sgcn_dropout_t* st) {
        const int64_t on = next();
        st->key = (uint32_t)next(); st->keep = f(); st->rows = i(); st->width = i();
        return on ? st : nullptr;
    }
    // {on, seg, nseg, fix, nfix, nslots, ws, ws_elems} -> sgcn_plan_t
    const sgcn_plan_t* plan(sgcn_plan_t* st) {
        const int64_t on = next();
        st->dev_seg = p<const sgcn_seg_t>(); st->nseg = next();
        st->dev_fix = p<const sgcn_fix_t>(); st->nfix = next();
        st->nslots = next(); st->dev_ws = p<float>(); st->ws_elems = next();
        if (st->nfix == 0) st->dev_fix = nullptr;
        return on ? st : nullptr;
    }
};


// The argument records of the ops that sgcn_step_run also LOOKS AHEAD at (the same reader serves the op's own case and the
// peek of the op in front of it).  `drop()` is the layer's dropout site or nullptr.
struct DenseFwdOp {
    int32_t M, N, K, split, relu; float eps;
    const float *X, *X2, *W, *off, *sc; int64_t ldx, ldx2, ldw, ldy, ws_cap;
    float *Y, *xhat, *rstd, *ws; const int32_t *g1, *g2;
    sgcn_dropout_t dr; bool has_drop;
    void read(Args& a) {
        M = a.i(); N = a.i(); K = a.i();
        X = a.p<const float>(); ldx = a.next(); X2 = a.p<const float>(); ldx2 = a.next(); split = a.i();
        W = a.p<const float>(); ldw = a.next(); off = a.p<const float>(); sc = a.p<const float>();
        eps = a.f(); relu = a.i();
        Y = a.p<float>(); ldy = a.next(); xhat = a.p<float>(); rstd = a.p<float>();
        has_drop = a.drop(&dr) != nullptr;
        ws = a.p<float>(); ws_cap = a.next(); g1 = a.p<const int32_t>(); g2 = a.p<const int32_t>();
    }
    const sgcn_dropout_t* drop() const { return has_drop ? &dr : nullptr; }
    bool plain() const { return !off && !sc && !relu; }                    // no LayerNorm, no ReLU: an output layer
};
struct DenseBwdOp {
    int32_t n, N, K, relu;
    const float *dy, *y, *xhat, *rstd, *sc, *x, *W; int64_t lddy, ldy, ldx, ldw, lddw, lddx, ws_cap;
    float *dW, *doff, *dsc, *dx, *gtmp, *ws; const int32_t* gidx;
    sgcn_dropout_t dr; bool has_drop;
    void read(Args& a) {
        n = a.i(); N = a.i(); K = a.i();
        dy = a.p<const float>(); lddy = a.next(); y = a.p<const float>(); ldy = a.next();
        xhat = a.p<const float>(); rstd = a.p<const float>(); sc = a.p<const float>(); relu = a.i();
        x = a.p<const float>(); ldx = a.next(); W = a.p<const float>(); ldw = a.next();
        dW = a.p<float>(); lddw = a.next(); doff = a.p<float>(); dsc = a.p<float>();
        dx = a.p<float>(); lddx = a.next();
        has_drop = a.drop(&dr) != nullptr;
        gtmp = a.p<float>(); ws = a.p<float>(); ws_cap = a.next(); gidx = a.p<const int32_t>();
    }
    const sgcn_dropout_t* drop() const { return has_drop ? &dr : nullptr; }
};
struct CeOp {
    const float *z, *lab; int64_t ldz, ldl, lddz, ldp; int32_t n, c; float *dz, *pred, *stats, *rowstat;
    void read(Args& a) {
        z = a.p<const float>(); ldz = a.next(); lab = a.p<const float>(); ldl = a.next(); n = a.i(); c = a.i();
        dz = a.p<float>(); lddz = a.next(); pred = a.p<float>(); ldp = a.next(); stats = a.p<float>(); rowstat = a.p<float>();
    }
};
inline bool is_ce(int32_t op) { return op == SGCN_OP_SOFTMAX_CE || op == SGCN_OP_SIGMOID_CE; }

}  // namespace

extern "C" int sgcn_step_fill(const sgcn_step_fill_t* f, const int64_t* meta, int64_t meta_len, int64_t ip, int64_t fp,
                              int64_t seed, int64_t step, float lr, int64_t* slots, int64_t nslots) {
    if (!f || !meta || !slots || meta_len < 0 || nslots < 0 || f->n < 0 || f->n_cap < 0 || f->n_ws < 0 || f->n_keys < 0 ||
        f->n > nslots || f->lr_slot < 0 || f->lr_slot >= nslots || (f->n > 0 && (!f->idx || !f->mul || !f->base)) ||
        (f->n_cap > 0 && (!f->cap_idx || !f->cap_max)) || (f->n_ws > 0 && (!f->ws_idx || !f->ws_ld)) ||
        (f->n_keys > 0 && (!f->key_slot || !f->key_layer)))
        return sgcn::fail(SGCN_ERR_INVALID, "step_fill: bad argument");
    auto in_meta = [&](int64_t i) { return i >= 0 && i < meta_len; };
    for (int64_t j = 0; j < f->n_cap; j++) {
        if (!in_meta(f->cap_idx[j])) return sgcn::fail(SGCN_ERR_INVALID, "step_fill: capacity check outside the descriptor table");
        if (meta[f->cap_idx[j]] > f->cap_max[j]) return 1;
    }
    for (int64_t j = 0; j < f->n_ws; j++) {
        if (!in_meta(f->ws_idx[j])) return sgcn::fail(SGCN_ERR_INVALID, "step_fill: workspace check outside the descriptor table");
        if (meta[f->ws_idx[j]] * f->ws_ld[j] > f->ws_floats) return 1;
    }
    for (int64_t i = 0; i < f->n; i++)
        if (!in_meta(f->idx[i])) return sgcn::fail(SGCN_ERR_INVALID, "step_fill: slot %lld reads outside the descriptor table", (long long)i);
    for (int64_t j = 0; j < f->n_keys; j++)
        if (f->key_slot[j] < 0 || f->key_slot[j] >= nslots) return sgcn::fail(SGCN_ERR_INVALID, "step_fill: key slot out of range");
    for (int64_t i = 0; i < f->n; i++)
        slots[i] = meta[f->idx[i]] * f->mul[i] + (f->base[i] == 1 ? ip : f->base[i] == 2 ? fp : 0);
    auto fmix32 = [](uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; };
    for (int64_t j = 0; j < f->n_keys; j++) {
        const uint32_t site = fmix32((uint32_t)seed * 0x9E3779B1u + (uint32_t)f->key_layer[j] * 0x85EBCA77u + 0x27D4EB2Fu);
        slots[f->key_slot[j]] = (int64_t)fmix32(site + (uint32_t)step * 0xC2B2AE3Du);
    }
    uint32_t bits;
    memcpy(&bits, &lr, sizeof(bits));
    slots[f->lr_slot] = (int64_t)bits;
    return SGCN_OK;
}

extern "C" int sgcn_copy_h2d_async(void* dst, const void* src, int64_t bytes, void* stream) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return sgcn::fail(SGCN_ERR_INVALID, "copy_h2d_async: bad argument");
    if (bytes == 0) return SGCN_OK;
    const hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return sgcn::fail(SGCN_ERR_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e));
    return SGCN_OK;
}

// ---- the exchange stream (ABI v15) -----------------------------------------------------------------------------
// The data-parallel step's history exchange (HIST_PACK -> ALLGATHER_I32 -> HIST_APPLY, aux = 2) on a stream of its own:
// its payload -- the aggregator's input -- is final once the aggregator has been issued, and its first reader is the NEXT
// step's aggregator (the reference only orders the scatter behind the optimizer, gcn/models.py:186-194), so the three ops
// need not sit on the step's dependent chain behind the optimizer.  The stream waits for `stream` once per run (at the
// first exchange op: everything issued so far, the aggregator included), `stream` waits for it at the end of the run.
// Unlike the auxiliary stream it is never joined in between -- the gradient all-reduce must not wait for the all-gather --
// and its collective runs on its own communicator (sgcn_coll.cpp: xcomm).
namespace {
#define SGCN_XCHG_TRY(expr)                                                                                   \
    do {                                                                                                      \
        const hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return sgcn::fail(SGCN_ERR_HIP, "step_run (exchange stream): %s", hipGetErrorString(e_)); \
    } while (0)
struct XchgCtx {
    hipStream_t st = nullptr;
    hipEvent_t fork = nullptr, done = nullptr;
    bool pending = false, tried = false;
};
XchgCtx& xchg_ctx() { static XchgCtx c; return c; }

// A stream whose kernels really run BESIDE the step's.  HIP multiplexes a process's streams onto a few hardware queues
// (four per priority level by default; a new stream takes the least referenced), and two streams that share one run in
// order: the exchange then sits on the step's chain again, behind its event packets -- measured with a one-rank job:
// 143 - 146 us per step against 129.5 without any overlap, and the round-5 form's epochs alternating between 51 and 69 ms.
// (A stream of another PRIORITY has a queue of its own, but two priority levels active at once cost every kernel of the
// step ~50 us: 530 us per step.  hipExtStreamCreateWithCUMask gives a dedicated queue too, but a blocking stream, which
// the legacy default stream -- torch's -- synchronises with.)  So: create up to eight candidates and keep the first on
// which a tiny kernel completes while a 3 ms spin is still running on the step's stream; none -> no exchange stream, the
// ops run on the step's own (null).
int xchg_pick(void* stream, XchgCtx& c) {
    c.tried = true;
    if (getenv("SGCN_XCHG_NO_PROBE")) {            // (for the record: the first candidate, whatever queue it got)
        SGCN_XCHG_TRY(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
        return SGCN_OK;
    }
    hipStream_t cand[8] = {};
    hipEvent_t ev = nullptr;
    SGCN_XCHG_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int got = -1, n = 0;
    SGCN_XCHG_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (; n < 8 && got < 0; n++) {
        SGCN_XCHG_TRY(hipStreamCreateWithFlags(&cand[n], hipStreamNonBlocking));
        // the clock starts BEFORE the spin is launched: a host thread that is descheduled between the launches can only make
        // a good candidate look bad (rejected, the next one is tried), never a queue-sharing one look good
        const auto t0 = std::chrono::steady_clock::now();
        int rc = sgcn::spin_launch(stream, 3000);
        if (rc == SGCN_OK) rc = sgcn::spin_launch(cand[n], 1);
        if (rc != SGCN_OK) return rc;
        SGCN_XCHG_TRY(hipEventRecord(ev, cand[n]));
        while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(1500)) {
            if (hipEventQuery(ev) == hipSuccess) {
                if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(1500)) got = n;
                break;
            }
        }
        SGCN_XCHG_TRY(hipStreamSynchronize((hipStream_t)stream));
        SGCN_XCHG_TRY(hipStreamSynchronize(cand[n]));
    }
    for (int i = 0; i < n; i++)
        if (i != got) (void)hipStreamDestroy(cand[i]);
    (void)hipEventDestroy(ev);
    c.st = got >= 0 ? cand[got] : nullptr;
    if (getenv("SGCN_XCHG_DEBUG")) fprintf(stderr, "sgcn: exchange stream: candidate %d of %d runs beside the step's stream\n", got, n);
    return SGCN_OK;
}

int xchg_fork(void* stream, void** side) {
    XchgCtx& c = xchg_ctx();
    if (!c.tried) {
        const int rc = xchg_pick(stream, c);
        if (rc != SGCN_OK) return rc;
        // no system-scope fence at the events: both sides of either dependency are kernels of THIS device (a default event's
        // record writes the caches back for the host and other devices -- microseconds on eight L2s, on the step's chain)
        const unsigned flags = hipEventDisableTiming | (getenv("SGCN_XCHG_SYSFENCE") ? 0u : hipEventDisableSystemFence);
        SGCN_XCHG_TRY(hipEventCreateWithFlags(&c.fork, flags));
        SGCN_XCHG_TRY(hipEventCreateWithFlags(&c.done, flags));
    }
    if (!c.st) { *side = stream; return SGCN_OK; }      // no stream beside the step's: in place, no events
    if (!c.pending) {
        // MEASUREMENT ONLY (profiles/r63_exchange_chain_probe.jsonl: what each dependency costs): dropping one makes the
        // results undefined, so the knob needs its companion variable spelled out
        static const char* skip = getenv("SGCN_XCHG_DROPS_A_DEPENDENCY") ? getenv("SGCN_XCHG_SKIP") : nullptr;
        if (!(skip && strstr(skip, "fork"))) {
            SGCN_XCHG_TRY(hipEventRecord(c.fork, (hipStream_t)stream));
            SGCN_XCHG_TRY(hipStreamWaitEvent(c.st, c.fork, 0));
        }
        c.pending = true;
    }
    *side = (void*)c.st;
    return SGCN_OK;
}

int xchg_join(void* stream) {
    XchgCtx& c = xchg_ctx();
    if (!c.pending) return SGCN_OK;
    c.pending = false;
    static const char* skip = getenv("SGCN_XCHG_DROPS_A_DEPENDENCY") ? getenv("SGCN_XCHG_SKIP") : nullptr;
    if (skip && strstr(skip, "join")) return SGCN_OK;
    SGCN_XCHG_TRY(hipEventRecord(c.done, c.st));
    SGCN_XCHG_TRY(hipStreamWaitEvent((hipStream_t)stream, c.done, 0));
    return SGCN_OK;
}
}  // namespace

extern "C" int sgcn_step_run(const sgcn_step_op_t* ops, int32_t nops, const int64_t* slots, int32_t nslots,
                             void* stream) {
    if (nops < 0 || (nops > 0 && !ops) || nslots < 0 || (nslots > 0 && !slots))
        return sgcn::fail(SGCN_ERR_INVALID, "step_run: bad argument");
    // a run keeps state in the library's singletons (deferred weight-gradient jobs, gradient-store mode, parked reductions,
    // the auxiliary stream): runs of different models / threads of one process are serialised here
    static std::mutex run_mutex;
    std::lock_guard<std::mutex> run_lock(run_mutex);
    bool memset_on_aux = false;
    // the process-wide knobs are the defaults; a program's own MODE op overrides them for this run -- as a thread-local
    // override that the entry points this thread calls below read through tune_get: the process-wide values are never
    // written, so other threads (and other entry points) keep seeing them
    int64_t run_overlap = sgcn_tune_get("step_overlap"), run_fuse = sgcn_tune_get("step_fuse");
    bool own_mode = false;
    for (int32_t k = 0; k < nops; k++)
        if (ops[k].op == SGCN_OP_MODE && ops[k].nargs >= 1) {
            own_mode = true;
            run_overlap = ops[k].add[0] != 0;
            if (ops[k].nargs >= 2 && ops[k].add[1] >= 0) run_fuse = ops[k].add[1] & 127;
        }
    const bool overlap = run_overlap != 0;
    const int fuse = (int)run_fuse;
    struct ModeGuard {
        bool on;
        ~ModeGuard() { if (on) sgcn::step_mode_override(-1, -1); }
    } mode{own_mode};
    if (own_mode) sgcn::step_mode_override(overlap ? 1 : 0, fuse);
    bool grouped = false, store = false, l2 = false;
    int ce_at = -1, adam_at = -1;
    for (int32_t k = 0; k < nops; k++) {
        grouped = grouped || ops[k].op == SGCN_OP_DW_FLUSH;
        store = store || ops[k].op == SGCN_OP_GRAD_STORE;
        l2 = l2 || ops[k].op == SGCN_OP_L2_PENALTY;
        if (ops[k].op == SGCN_OP_SOFTMAX_CE || ops[k].op == SGCN_OP_SIGMOID_CE) ce_at = k;
        if (ops[k].op == SGCN_OP_ADAM) adam_at = k;
    }
    // the loss statistics ride in the optimizer's launch when this very run has one after the loss and nothing (the
    // weight-decay term) adds to the loss slot in between
    const bool park_stats = store && ce_at >= 0 && adam_at > ce_at && !l2;
    if (grouped) sgcn::dw_group_begin();
    sgcn::grad_store_mode(store ? 1 : 0);
    sgcn::stats_defer(park_stats ? 1 : 0);
    struct Guard {                       // an early return leaves no recorded job / mode behind
        bool on; void* st;
        ~Guard() { if (on) sgcn::dw_group_abort(); sgcn::grad_store_mode(0); sgcn::stats_defer(0); sgcn::reduce_flush(st); sgcn::stats_flush(st); }
    } guard{grouped, stream};
    auto eval_args = [&](const sgcn_step_op_t& o, Args& a) {
        a.n = o.nargs; a.pos = 0;
        for (int j = 0; j < o.nargs; j++) {
            const int32_t s = o.slot[j];
            if (s >= nslots) return false;
            a.v[j] = (s < 0 ? 0 : o.mul[j] * slots[s]) + o.add[j];
        }
        return true;
    };
    int32_t skip_until = 0, dx_done_at = -1;
    // an output layer waiting for its loss op -- and, when it is folded too, the dense layer in front of it
    struct Head { bool on = false, pre = false; int32_t at = -1, kg = 1, pS = 1, pkchunk = 0, pkg = 1; DenseFwdOp f, p; } head;
    auto peek = [&](int32_t j, Args& b) {          // the arguments of op j, if there is one and they are well-formed
        return j < nops && ops[j].nargs >= 0 && ops[j].nargs <= SGCN_STEP_MAX_ARGS && eval_args(ops[j], b);
    };
    for (int32_t k = 0; k < nops; k++) {
        const sgcn_step_op_t& op = ops[k];
        if (op.nargs < 0 || op.nargs > SGCN_STEP_MAX_ARGS)
            return sgcn::fail(SGCN_ERR_INVALID, "step_run: op %d has %d arguments", k, op.nargs);
        if (k < skip_until) continue;        // a history scatter that rode in the optimizer's launch
        Args a;
        if (!eval_args(ops[k], a)) return sgcn::fail(SGCN_ERR_INVALID, "step_run: op %d reads a slot beyond %d", k, nslots);
        int rc = SGCN_OK;
        sgcn_dropout_t dr;
        sgcn_plan_t pl;
        // weight-gradient work forked onto the auxiliary stream (DENSE_BWD) is joined before anything that
        // reads or writes gradients outside the backward chain
        if (op.op == SGCN_OP_ADAM || op.op == SGCN_OP_L2_PENALTY || op.op == SGCN_OP_SCATTER_ROWS ||
            op.op == SGCN_OP_MEMSET0 || op.op == SGCN_OP_VR_AGG_POST || (op.op == SGCN_OP_DENSE_BWD && memset_on_aux)) {
            rc = sgcn::aux_join(stream);
            if (rc != SGCN_OK) return rc;
            memset_on_aux = false;
        }
        void* side = stream;                 // where an AUX_* / *_PRE op runs
        if (overlap && (op.op == SGCN_OP_VR_AGG_PRE || op.op == SGCN_OP_AUX_SCATTER_ROWS || op.op == SGCN_OP_AUX_MEMSET0)) {
            rc = sgcn::aux_fork(stream, &side);
            if (rc != SGCN_OK) return rc;
        }
        switch (op.op) {
        case SGCN_OP_DENSE_FWD: {
            DenseFwdOp f;
            f.read(a);
            // The output layer (no LayerNorm, no ReLU, <= 64 classes, <= 128 inputs) directly in front of the loss: its
            // 5-MFLOP product becomes the head of the loss kernel's row pass (sgcn_dense.hip ce_head) instead of a launch
            head.on = false;
            Args b;
            if ((fuse & 1) && f.plain() && !f.X2 && !f.g1 && f.N <= 64 && f.K <= 128 && (int64_t)f.K * f.N * 4 <= 48 * 1024 &&
                peek(k + 1, b) && is_ce(ops[k + 1].op)) {
                CeOp ce;
                ce.read(b);
                int S = 0, kgq = 0;
                sgcn::gemm_fwd_shape(f.M, f.N, f.K, &S, &kgq, nullptr);
                if (S == 1 && kgq <= 2 && ce.z == f.Y && ce.ldz == f.ldy && ce.n == f.M && ce.c == f.N) {
                    head.on = true; head.pre = false; head.at = k + 1; head.kg = kgq; head.f = f;
                    break;                           // nothing launched: the loss op computes the logits
                }
            }
            // ... and the dense layer in front of THAT (<= 256 inputs, <= 128 outputs = the output layer's inputs, all rows):
            // its product, LayerNorm and ReLU become the pre-layer of the head -- three launches (MFMA tiles, split-K reduce,
            // loss) and the logits' and the hidden row's round trips through memory become one row pass (fuse bit 4)
            if ((fuse & 16) && (fuse & 1) && !f.X2 && !f.g1 && f.N <= 128 && f.K <= 256 && (f.K * f.N) % 4 == 0 &&
                peek(k + 1, b) && ops[k + 1].op == SGCN_OP_DENSE_FWD && k + 2 < nops && is_ce(ops[k + 2].op)) {
                DenseFwdOp q;
                q.read(b);
                Args c3;
                CeOp ce;
                int S = 0, kgq = 0, S0 = 0, kg0 = 0, kc0 = 0;
                sgcn::gemm_fwd_shape(q.M, q.N, q.K, &S, &kgq, nullptr);
                sgcn::gemm_fwd_shape(f.M, f.N, f.K, &S0, &kg0, &kc0);
                const bool ok = q.plain() && !q.X2 && !q.g1 && q.N <= 64 && q.K == f.N && q.M == f.M && q.X == f.Y && q.ldx == f.ldy &&
                                (int64_t)q.K * q.N * 4 <= 48 * 1024 && S == 1 && kgq <= 2 && S0 >= 1 && S0 <= 2 && kg0 <= 2 &&
                                f.ldw == f.N && q.ldw == q.N && peek(k + 2, c3);
                if (ok) {
                    ce.read(c3);
                    if (ce.z == q.Y && ce.ldz == q.ldy && ce.n == q.M && ce.c == q.N &&
                        (((int64_t)q.K * q.N + 255) / 256 * 256 + ((int64_t)f.K * f.N + 255) / 256 * 256) * 4 <= 160 * 1024) {
                        head.on = true; head.pre = true; head.at = k + 2; head.kg = kgq; head.f = q; head.p = f;
                        head.pS = S0; head.pkchunk = kc0; head.pkg = kg0;
                        skip_until = k + 2;          // neither layer is launched: the loss op computes both
                        break;
                    }
                }
            }
            // the eager wrapper (ops.dense_fwd): split-K scratch only where the library asks for it
            const int64_t need = f.N <= 128 ? sgcn_gemm_ws_floats(f.M, f.N, f.K) : 0;
            if (need > f.ws_cap) return sgcn::fail(SGCN_ERR_INVALID, "step_run: GEMM scratch %lld > %lld floats", (long long)need, (long long)f.ws_cap);
            // A layer that is cut over K (its epilogue is a row pass already) with a narrow dense layer on ALL of its output
            // rows right behind it: that layer rides in the row pass (sgcn_gemm.hip splitk_ln_dense_kernel)
            if ((fuse & 4) && need > 0 && peek(k + 1, b) && ops[k + 1].op == SGCN_OP_DENSE_FWD) {
                DenseFwdOp q;
                q.read(b);
                // (an output layer in front of the loss is better off as the loss kernel's head: one pass fewer)
                const bool is_head = (fuse & 1) && q.plain() && !q.X2 && q.N <= 64 && k + 2 < nops && is_ce(ops[k + 2].op);
                const bool whole = !is_head && q.X == f.Y && q.ldx == f.ldy && !q.g1 && !q.g2 && q.M == f.M && q.K == f.N &&
                                   (!q.X2 || (q.X2 == f.Y + (int64_t)q.split * f.ldy && q.ldx2 == f.ldy)) &&
                                   (!q.has_drop || q.dr.rows == q.split || !q.X2);
                if (whole) {
                    int fusedq = 0;
                    rc = sgcn::dense_fwd_pair(f.M, f.N, f.K, f.X, f.ldx, f.X2, f.ldx2, f.split, f.W, f.ldw, f.off, f.sc, f.eps, f.relu,
                                              f.Y, f.ldy, f.xhat, f.rstd, f.drop(), f.ws, f.g1, f.g2, q.N, q.W, q.ldw, q.off, q.sc, q.eps,
                                              q.relu, q.Y, q.ldy, q.xhat, q.rstd, q.drop(), stream, &fusedq);
                    if (rc != SGCN_OK) break;
                    if (fusedq) { skip_until = k + 2; break; }
                }
            }
            rc = sgcn_dense_fwd_f32(f.M, f.N, f.K, f.X, f.ldx, f.X2, f.ldx2, f.split, f.W, f.ldw, f.off, f.sc, f.eps, f.relu, f.Y, f.ldy,
                                    f.xhat, f.rstd, f.drop(), need ? f.ws : nullptr, f.g1, f.g2, stream);
            break;
        }
        case SGCN_OP_DENSE_BWD: {
            DenseBwdOp w;
            w.read(a);
            const bool norm = w.xhat != nullptr;
            const int64_t need = (norm ? (sgcn_ln_act_bwd_ws_floats(w.n, w.N) + 3) / 4 * 4 : 0) +
                                 std::max(sgcn_gemm_ws_floats(w.K, w.N, w.n), sgcn_gemm_ws_floats(w.n, w.K, w.N));
            if (need > w.ws_cap) return sgcn::fail(SGCN_ERR_INVALID, "step_run: backward scratch %lld > %lld floats", (long long)need, (long long)w.ws_cap);
            // The layer below follows at once and consumes nothing but this layer's dx (it is the first layer: no dx of its
            // own): its LayerNorm / ReLU backward pass rides behind this layer's row pass (sgcn_gemm.hip dense_bwd_chain)
            Args b;
            if ((fuse & 64) && w.dx && k != dx_done_at && peek(k + 1, b) && ops[k + 1].op == SGCN_OP_DENSE_BWD) {
                DenseBwdOp q;
                q.read(b);
                const bool qnorm = q.xhat != nullptr;
                const int64_t need2 = (qnorm ? (sgcn_ln_act_bwd_ws_floats(q.n, q.N) + 3) / 4 * 4 : 0) +
                                      std::max(sgcn_gemm_ws_floats(q.K, q.N, q.n), sgcn_gemm_ws_floats(q.n, q.K, q.N));
                if (q.dy == w.dx && q.lddy == w.lddx && q.n == w.n && q.N == w.K && !q.dx && need2 <= q.ws_cap) {
                    const sgcn::DenseBwdArgs up{w.n, w.N, w.K, w.dy, w.lddy, w.y, w.ldy, w.xhat, w.rstd, w.sc, w.relu, w.x, w.ldx, w.W, w.ldw,
                                                w.dW, w.lddw, w.doff, w.dsc, w.dx, w.lddx, w.drop(), w.gtmp, need ? w.ws : nullptr, w.gidx};
                    const sgcn::DenseBwdArgs lo{q.n, q.N, q.K, q.dy, q.lddy, q.y, q.ldy, q.xhat, q.rstd, q.sc, q.relu, q.x, q.ldx, q.W, q.ldw,
                                                q.dW, q.lddw, q.doff, q.dsc, nullptr, q.lddx, q.drop(), q.gtmp, need2 ? q.ws : nullptr, q.gidx};
                    rc = sgcn::dense_bwd_chain(up, lo, stream);
                    skip_until = k + 2;
                    break;
                }
            }
            rc = sgcn::dense_bwd_overlapped(w.n, w.N, w.K, w.dy, w.lddy, w.y, w.ldy, w.xhat, w.rstd, w.sc, w.relu, w.x, w.ldx, w.W, w.ldw,
                                            w.dW, w.lddw, w.doff, w.dsc, k == dx_done_at ? nullptr : w.dx, w.lddx, w.drop(), w.gtmp,
                                            need ? w.ws : nullptr, w.gidx, stream);
            break;
        }
        case SGCN_OP_DW_FLUSH:
            // the optimizer right behind the weight gradients: their reductions ride in its launch (fuse bit 5)
            rc = sgcn::dw_group_flush(stream, (fuse & 32) && k + 1 < nops && ops[k + 1].op == SGCN_OP_ADAM);
            break;
        case SGCN_OP_GRAD_STORE:         // (mode of the whole run: set before the loop)
        case SGCN_OP_MODE:
            break;
        // ---- the sparse-input first layer: slice of the feature CSR, LayerNorm / ReLU passes around sparse products, the
        // slice's transpose for the weight gradient (layers.Dense / AugmentedDropoutDense with sparse_inputs, call for call)
        case SGCN_OP_CSR_SLICE: {
            const int32_t n = a.i(); const int32_t* rows = a.p<const int32_t>();
            const float* av = a.p<const float>(); const int32_t* ac = a.p<const int32_t>(); const int32_t* arp = a.p<const int32_t>();
            int32_t* o_p = a.p<int32_t>(); float* o_d = a.p<float>(); int32_t* o_c = a.p<int32_t>(); int32_t* o_r = a.p<int32_t>();
            rc = sgcn_csr_slice_indptr_dev(n, rows, arp, o_p, stream);
            if (rc == SGCN_OK) rc = sgcn_csr_slice_f32(n, rows, av, ac, arp, o_p, o_d, o_c, o_r, stream);
            break;
        }
        case SGCN_OP_LN_ACT_FWD: {
            const float* x = a.p<const float>(); const int64_t ldx = a.next();
            const float* off = a.p<const float>(); const float* sc = a.p<const float>();
            const int32_t n = a.i(), d = a.i(); const float eps = a.f(); const int32_t relu = a.i();
            float* y = a.p<float>(); const int64_t ldy = a.next(); float* xhat = a.p<float>(); float* rstd = a.p<float>();
            rc = sgcn_ln_act_fwd_f32(x, ldx, off, sc, n, d, eps, relu, y, ldy, xhat, rstd, stream);
            break;
        }
        case SGCN_OP_LN_ACT_BWD: {
            const float* dy = a.p<const float>(); const int64_t lddy = a.next();
            const float* y = a.p<const float>(); const int64_t ldy = a.next();
            const float* xhat = a.p<const float>(); const float* rstd = a.p<const float>(); const float* sc = a.p<const float>();
            const int32_t n = a.i(), d = a.i(), relu = a.i();
            float* dx = a.p<float>(); const int64_t lddx = a.next(); float* doff = a.p<float>(); float* dsc = a.p<float>();
            float* ws = a.p<float>(); const int64_t ws_cap = a.next();
            if (sc && sgcn_ln_act_bwd_ws_floats(n, d) > ws_cap)
                return sgcn::fail(SGCN_ERR_INVALID, "step_run: LayerNorm-backward scratch too small at op %d", k);
            rc = sgcn_ln_act_bwd_f32(dy, lddy, y, ldy, xhat, rstd, sc, n, d, relu, dx, lddx, doff, dsc, ws, stream);
            break;
        }
        // ---- the --det_dropout stacks (ABI v16): the eager layers' calls, one op each ----------------------------------
        case SGCN_OP_GEMM: {
            const int32_t ta = a.i(), tb = a.i(), M = a.i(), N = a.i(), K = a.i();
            const float* A = a.p<const float>(); const int64_t lda = a.next();
            const float* B = a.p<const float>(); const int64_t ldb = a.next();
            float* Cm = a.p<float>(); const int64_t ldc = a.next(); const int32_t acc = a.i();
            float* ws = a.p<float>(); const int64_t ws_cap = a.next();
            const int64_t need = sgcn_gemm_ws_floats(M, N, K);           // (ops.gemm: scratch exactly when split-K pays)
            if (need > ws_cap) return sgcn::fail(SGCN_ERR_INVALID, "step_run: GEMM scratch too small at op %d", k);
            rc = (M > 0 && N > 0) ? sgcn_gemm_f32(ta, tb, M, N, K, A, lda, B, ldb, Cm, ldc, acc, need > 0 ? ws : nullptr,
                                                  nullptr, nullptr, stream) : SGCN_OK;
            break;
        }
        case SGCN_OP_DET_PRE: {
            const float* mu = a.p<const float>(); const float* var = a.p<const float>(); const int64_t n = a.next();
            const float keep = a.f(); float* out = a.p<float>();
            rc = sgcn_det_pre_f32(mu, var, n, keep, out, stream);
            break;
        }
        case SGCN_OP_DET_PRE_BWD: {
            const float* mu = a.p<const float>(); const float* g = a.p<const float>(); const int64_t n = a.next();
            const float keep = a.f(); float* d_mu = a.p<float>(); float* d_var = a.p<float>();
            rc = sgcn_det_pre_bwd_f32(mu, g, n, keep, d_mu, d_var, stream);
            break;
        }
        case SGCN_OP_SQUARE: {
            const float* x = a.p<const float>(); const int64_t n = a.next(); const float c = a.f(); float* y = a.p<float>();
            rc = sgcn_square_f32(x, n, c, y, stream);
            break;
        }
        case SGCN_OP_ADDMUL: {
            float* acc = a.p<float>(); const float* x = a.p<const float>(); const float* y = a.p<const float>();
            const int64_t n = a.next(); const float c = a.f();
            rc = sgcn_addmul_f32(acc, x, y, n, c, stream);
            break;
        }
        case SGCN_OP_DET_LNVAR_FWD: {
            const float* var1 = a.p<const float>(); const float* rstd = a.p<const float>(); const float* sc = a.p<const float>();
            const int32_t n = a.i(), d = a.i(); const float eps = a.f(); float* out = a.p<float>();
            rc = sgcn_det_lnvar_fwd_f32(var1, rstd, sc, n, d, eps, out, stream);
            break;
        }
        case SGCN_OP_DET_LNVAR_BWD: {
            const float* g = a.p<const float>(); const float* var1 = a.p<const float>(); const float* xhat = a.p<const float>();
            const float* rstd = a.p<const float>(); const float* sc = a.p<const float>();
            const int32_t n = a.i(), d = a.i(); const float eps = a.f();
            float* d_var1 = a.p<float>(); float* d_mu1 = a.p<float>(); float* dsc = a.p<float>(); float* tmp = a.p<float>();
            rc = sgcn_det_lnvar_bwd_f32(g, var1, xhat, rstd, sc, n, d, eps, d_var1, d_mu1, dsc, tmp, stream);
            break;
        }
        case SGCN_OP_DET_RELU_FWD: {
            const float* mu = a.p<const float>(); const float* var = a.p<const float>(); const int64_t n = a.next();
            float* mo = a.p<float>(); float* vo = a.p<float>();
            rc = sgcn_det_relu_fwd_f32(mu, var, n, mo, vo, stream);
            break;
        }
        case SGCN_OP_DET_RELU_BWD: {
            const float* mu = a.p<const float>(); const float* var = a.p<const float>();
            const float* gm = a.p<const float>(); const float* gv = a.p<const float>(); const int64_t n = a.next();
            float* d_mu = a.p<float>(); float* d_var = a.p<float>();
            rc = sgcn_det_relu_bwd_f32(mu, var, gm, gv, n, d_mu, d_var, stream);
            break;
        }
        case SGCN_OP_GAUSS: {
            const float* mu = a.p<const float>(); const float* var = a.p<const float>(); const int64_t n = a.next();
            const uint32_t key = (uint32_t)a.next(); float* x = a.p<float>();
            rc = sgcn_gauss_sample_f32(mu, var, n, key, x, stream);
            break;
        }
        case SGCN_OP_GAUSS_BWD: {
            const float* var = a.p<const float>(); const float* g = a.p<const float>(); const int64_t n = a.next();
            const uint32_t key = (uint32_t)a.next(); float* d_var = a.p<float>();
            rc = sgcn_gauss_sample_bwd_f32(var, g, n, key, d_var, stream);
            break;
        }
        case SGCN_OP_DET_AGG_PREP: {
            const float* mu = a.p<const float>(); const float* var = a.p<const float>();
            const float* Hm = a.p<const float>(); const float* Hv = a.p<const float>(); const int64_t ldh = a.next();
            const int32_t* ifield = a.p<const int32_t>(); const int32_t n0 = a.i(), d = a.i();
            float* dmu = a.p<float>(); float* ds2 = a.p<float>(); float* msig2 = a.p<float>(); float* ds = a.p<float>(); float* sbar = a.p<float>();
            rc = sgcn_det_agg_prep_f32(mu, var, Hm, Hv, ldh, ifield, n0, d, dmu, ds2, msig2, ds, sbar, stream);
            break;
        }
        case SGCN_OP_DET_AGG_PREP_BWD: {
            const float* var = a.p<const float>(); const float* ds = a.p<const float>(); const float* sbar = a.p<const float>();
            const float* g_ds2 = a.p<const float>(); const float* g_msig2 = a.p<const float>(); const int32_t n0 = a.i(), d = a.i();
            const float* add = a.p<const float>(); const int64_t ldadd = a.next(); const int32_t add_rows = a.i(); float* d_var = a.p<float>();
            rc = sgcn_det_agg_prep_bwd_f32(var, ds, sbar, g_ds2, g_msig2, n0, d, add, ldadd, add_rows, d_var, stream);
            break;
        }
        case SGCN_OP_RELU_EPS: {
            const float* raw = a.p<const float>(); const int64_t ldr = a.next(); const int32_t n = a.i(), d = a.i();
            const float eps = a.f(); float* y = a.p<float>(); const int64_t ldy = a.next();
            rc = sgcn_relu_eps_f32(raw, ldr, n, d, eps, y, ldy, stream);
            break;
        }
        case SGCN_OP_GATE: {
            const float* raw = a.p<const float>(); const int64_t ldr = a.next();
            const float* g = a.p<const float>(); const int64_t ldg = a.next(); const int32_t n = a.i(), d = a.i(); float* out = a.p<float>();
            rc = sgcn_gate_f32(raw, ldr, g, ldg, n, d, out, stream);
            break;
        }
        case SGCN_OP_CSR_TRANSPOSE: {
            const int32_t ncols = a.i(); const int64_t nnz = a.next();
            const int32_t* col = a.p<const int32_t>(); const int32_t* row = a.p<const int32_t>();
            int32_t* trp = a.p<int32_t>(); int32_t* trow = a.p<int32_t>(); int32_t* tsrc = a.p<int32_t>();
            int32_t* ws = a.p<int32_t>(); const int64_t ws_cap = a.next();
            if (sgcn_csr_transpose_ws_ints(ncols, nnz) > ws_cap)
                return sgcn::fail(SGCN_ERR_INVALID, "step_run: transpose scratch too small at op %d", k);
            rc = sgcn_csr_transpose_index(ncols, nnz, col, row, trp, trow, tsrc, ws, stream);
            break;
        }
        case SGCN_OP_GATHER_F32: {
            const float* src = a.p<const float>(); const int32_t* idx = a.p<const int32_t>(); const int64_t n = a.next();
            float* out = a.p<float>();
            rc = sgcn_gather_f32(src, idx, n, out, stream);
            break;
        }
        case SGCN_OP_VR_AGG: {
            const int32_t* arp = a.p<const int32_t>(); const int32_t* ac = a.p<const int32_t>(); const float* av = a.p<const float>();
            const int32_t* frp = a.p<const int32_t>(); const int32_t* fc = a.p<const int32_t>(); const float* fv = a.p<const float>();
            const int32_t n1 = a.i(), n0 = a.i(), nf = a.i(), d = a.i();
            const float* h = a.p<const float>(); const float* mu = a.p<const float>(); const int64_t ldx = a.next();
            const float* H = a.p<const float>(); const int64_t ldh = a.next();
            const int32_t* ifi = a.p<const int32_t>(); const int32_t* ffi = a.p<const int32_t>();
            const float* s = a.p<const float>();
            float* oh = a.p<float>(); float* om = a.p<float>(); const int64_t ldo = a.next();
            const int32_t cvd = a.i(), concat = a.i();
            const sgcn_plan_t* p = a.plan(&pl);
            rc = sgcn_vr_aggregate_f32(arp, ac, av, frp, fc, fv, n1, n0, nf, d, h, mu, ldx, H, ldh, ifi, ffi, s, oh, om, ldo,
                                       cvd, concat, p, stream);
            break;
        }
        case SGCN_OP_SPMM: {
            const int32_t* rp = a.p<const int32_t>(); const int32_t* c = a.p<const int32_t>(); const float* v = a.p<const float>();
            const int32_t M = a.i(), K = a.i(), d = a.i();
            const float* B = a.p<const float>(); const int64_t ldb = a.next();
            const int32_t* gidx = a.p<const int32_t>(); const float* rs = a.p<const float>(); const float* cs = a.p<const float>();
            float* C = a.p<float>(); const int64_t ldc = a.next(); const float beta = a.f();
            const sgcn_plan_t* p = a.plan(&pl);
            const float* add = a.p<const float>(); const int64_t ldadd = a.next(); const int32_t add_rows = a.i();
            rc = add ? sgcn_spmm_csr_add_f32(rp, c, v, M, K, d, B, ldb, gidx, rs, cs, C, ldc, beta, p, add, ldadd, add_rows, stream)
                     : sgcn_spmm_csr_f32(rp, c, v, M, K, d, B, ldb, gidx, rs, cs, C, ldc, beta, p, stream);
            break;
        }
        case SGCN_OP_SOFTMAX_CE:
        case SGCN_OP_SIGMOID_CE: {
            CeOp ce;
            ce.read(a);
            const bool hd = head.on && head.at == k;
            // The backward of the LAST dense layer follows at once (it has neither LayerNorm nor ReLU): its input gradient
            // dx = dlogits . W^T is a tail of the loss kernel's row pass (sgcn_dense.hip ce_dx_tail), that op then only
            // records its weight-gradient GEMM
            DenseBwdOp t{};
            bool tail = false;
            Args b;
            if ((fuse & 2) && ce.dz && ce.c <= 64 && peek(k + 1, b) && ops[k + 1].op == SGCN_OP_DENSE_BWD) {
                t.read(b);
                tail = t.n == ce.n && t.N == ce.c && t.dy == ce.dz && t.lddy == ce.lddz && !t.xhat && !t.sc && !t.relu && t.dx && t.W &&
                       t.K > 0 && (int64_t)t.K * ce.c * 4 <= 48 * 1024 &&
                       (!hd || (t.W == head.f.W && t.K == head.f.K && t.ldw == head.f.ldw));      // the head's own layer
                if (tail) dx_done_at = k + 1;
            }
            sgcn::CeLastLayer L;
            if (hd || tail) {
                L.W = hd ? head.f.W : t.W; L.ldw = hd ? head.f.ldw : t.ldw; L.K = hd ? head.f.K : t.K;
            }
            if (tail) { L.dx = t.dx; L.lddx = t.lddx; L.dx_drop = t.drop(); }
            if (hd) {
                L.hx = head.pre ? nullptr : head.f.X; L.ldhx = head.f.ldx; L.kg = head.kg; L.h_drop = head.f.drop();
                if (head.pre) {
                    const DenseFwdOp& p = head.p;
                    L.pre = true; L.px = p.X; L.ldpx = p.ldx; L.PK = p.K; L.PW = p.W; L.pS = head.pS; L.pkchunk = head.pkchunk;
                    L.pkg = head.pkg; L.p_drop = p.drop(); L.poff = p.off; L.psc = p.sc; L.peps = p.eps; L.prelu = p.relu;
                    L.pY = p.Y; L.ldpy = p.ldy; L.pxhat = p.xhat; L.prstd = p.rstd;
                }
            }
            // the loss / accuracy sums run beside the backward pass (joined before L2_PENALTY / ADAM) or ride in the optimizer's launch
            rc = sgcn::ce_impl(op.op == SGCN_OP_SOFTMAX_CE, ce.z, ce.ldz, ce.lab, ce.ldl, ce.n, ce.c, ce.dz, ce.lddz, ce.pred, ce.ldp,
                               ce.stats, ce.rowstat, stream, overlap, (hd || tail) ? &L : nullptr);
            head.on = false;
            break;
        }
        case SGCN_OP_ADAM: {
            float* th = a.p<float>(); const float* g = a.p<const float>(); float* m = a.p<float>(); float* v = a.p<float>();
            const int64_t n = a.next();
            const float lr = a.f(), b1 = a.f(), b2 = a.f(), eps = a.f();
            // history scatters that directly follow the optimizer (the reference orders them after it by a control dependency
            // only, gcn/models.py:186-194) touch nothing it touches: they ride in its launch as further workgroups
            int32_t nxt = k + 1;
            for (; nxt < nops && nxt <= k + 2 && ops[nxt].op == SGCN_OP_SCATTER_ROWS && ops[nxt].nargs >= 0 &&
                   ops[nxt].nargs <= SGCN_STEP_MAX_ARGS; nxt++) {
                Args b;
                if (!eval_args(ops[nxt], b)) break;
                float* H = b.p<float>(); const int64_t ldh = b.next();
                const int32_t* r = b.p<const int32_t>(); const int32_t rn = b.i(), rd = b.i();
                const float* src = b.p<const float>(); const int64_t lds = b.next();
                if (rn == 0 || rd == 0) continue;                  // nothing to write: counts as done
                if (!H || !r || !src || !sgcn::scatter_park(H, ldh, r, rn, rd, src, lds)) break;
            }
            skip_until = nxt;
            rc = sgcn::adam_with_stats(th, g, m, v, n, lr, b1, b2, eps, stream);
            break;
        }
        case SGCN_OP_ALLREDUCE_AVG: {
            // every gradient of the run is in the buffer: deferred weight-gradient work joined, parked reductions issued
            rc = sgcn::aux_join(stream);
            if (rc == SGCN_OK) rc = sgcn::reduce_flush(stream);
            if (rc != SGCN_OK) break;
            float* buf = a.p<float>(); const int64_t n = a.next();
            rc = sgcn_coll_allreduce_avg_f32(buf, n, stream);
            break;
        }
        case SGCN_OP_HIST_PACK: {
            const int32_t* ids = a.p<const int32_t>(); const int32_t n = a.i();
            const float* rows = a.p<const float>(); const int64_t ld = a.next();
            const int32_t d = a.i(), cap = a.i();
            int32_t* send = a.p<int32_t>();
            // (last argument != 0: on the auxiliary stream, forked here -- the exchange then runs beside the rest of the
            // step and is joined with the other auxiliary work in front of the gradient all-reduce / the optimizer)
            const int64_t where = a.next();
            if (where != 0) { rc = where == 2 ? xchg_fork(stream, &side) : sgcn::aux_fork(stream, &side); if (rc != SGCN_OK) break; }
            rc = sgcn_hist_pack_f32(ids, n, rows, ld, d, cap, send, side);
            break;
        }
        case SGCN_OP_ALLGATHER_I32: {
            const int32_t* send = a.p<const int32_t>(); int32_t* recv = a.p<int32_t>(); const int64_t n = a.next();
            const int64_t where = a.next();
            if (where != 0) { rc = where == 2 ? xchg_fork(stream, &side) : sgcn::aux_fork(stream, &side); if (rc != SGCN_OK) break; }
            rc = where == 2 ? sgcn_coll_allgather_x_i32(send, recv, n, side) : sgcn_coll_allgather_i32(send, recv, n, side);
            break;
        }
        case SGCN_OP_HIST_APPLY: {
            float* H = a.p<float>(); const int64_t ldh = a.next();
            const int32_t* recv = a.p<const int32_t>(); const int32_t world = a.i(), cap = a.i(), d = a.i();
            int32_t* owner = a.p<int32_t>();
            const int64_t where = a.next();
            if (where != 0) { rc = where == 2 ? xchg_fork(stream, &side) : sgcn::aux_fork(stream, &side); if (rc != SGCN_OK) break; }
            rc = sgcn_hist_apply_f32(H, ldh, recv, world, cap, d, owner, side);
            break;
        }
        case SGCN_OP_SCATTER_ROWS:
        case SGCN_OP_AUX_SCATTER_ROWS: {
            float* H = a.p<float>(); const int64_t ldh = a.next();
            const int32_t* r = a.p<const int32_t>(); const int32_t n = a.i(), d = a.i();
            const float* src = a.p<const float>(); const int64_t lds = a.next();
            rc = sgcn_scatter_rows_f32(H, ldh, r, n, d, src, lds, side);
            break;
        }
        case SGCN_OP_VR_AGG_PRE: {
            const int32_t* frp = a.p<const int32_t>(); const int32_t* fc = a.p<const int32_t>(); const float* fv = a.p<const float>();
            const int32_t n1 = a.i(), nf = a.i(), d = a.i();
            const float* H = a.p<const float>(); const int64_t ldh = a.next();
            const int32_t* ffi = a.p<const int32_t>();
            float* accP = a.p<float>();
            const sgcn_plan_t* p = a.plan(&pl);
            rc = sgcn_vr_aggregate_pre_f32(frp, fc, fv, n1, nf, d, H, ldh, ffi, accP, p, side);
            break;
        }
        case SGCN_OP_VR_AGG_POST: {
            const int32_t* arp = a.p<const int32_t>(); const int32_t* ac = a.p<const int32_t>(); const float* av = a.p<const float>();
            const int32_t n1 = a.i(), n0 = a.i(), d = a.i();
            const float* h = a.p<const float>(); const float* mu = a.p<const float>(); const int64_t ldx = a.next();
            const float* H = a.p<const float>(); const int64_t ldh = a.next();
            const int32_t* ifi = a.p<const int32_t>(); const float* s = a.p<const float>();
            float* oh = a.p<float>(); float* om = a.p<float>(); const int64_t ldo = a.next();
            const int32_t cvd = a.i(), concat = a.i();
            const float* accP = a.p<const float>();
            rc = sgcn_vr_aggregate_post_f32(arp, ac, av, n1, n0, d, h, mu, ldx, H, ldh, ifi, s, oh, om, ldo, cvd, concat, accP, stream);
            break;
        }
        case SGCN_OP_GATHER_ROWS: {
            const float* in = a.p<const float>(); const int64_t ldi = a.next();
            const int32_t* r = a.p<const int32_t>(); const int32_t n = a.i(), d = a.i();
            float* out = a.p<float>(); const int64_t ldo = a.next();
            rc = sgcn_gather_rows_f32(in, ldi, r, n, d, out, ldo, stream);
            break;
        }
        case SGCN_OP_DROPOUT: {
            const float* x = a.p<const float>(); const int64_t ldx = a.next();
            const int32_t n = a.i(), d = a.i();
            a.drop(&dr);
            float* out = a.p<float>(); const int64_t ldo = a.next();
            rc = sgcn_dropout_f32(x, ldx, n, d, &dr, out, ldo, stream);
            break;
        }
        case SGCN_OP_L2_PENALTY: {
            const float* th = a.p<const float>(); const int64_t lo = a.next(), hi = a.next();
            const float wd = a.f();
            float* g = a.p<float>(); float* loss = a.p<float>();
            rc = sgcn_l2_penalty_f32(th, lo, hi, wd, g, loss, stream);
            break;
        }
        case SGCN_OP_MEMSET0:
        case SGCN_OP_AUX_MEMSET0: {
            void* p = a.p<void>(); const int64_t bytes = a.next();
            if (bytes > 0 && hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)side) != hipSuccess)
                return sgcn::fail(SGCN_ERR_HIP, "step_run: hipMemsetAsync failed");
            if (op.op == SGCN_OP_AUX_MEMSET0 && side != stream) memset_on_aux = true;
            break;
        }
        case SGCN_OP_COPY2D: {
            void* dst = a.p<void>(); const int64_t ldd = a.next();
            const void* src = a.p<const void>(); const int64_t lds = a.next();
            const int64_t rows = a.next(), cols = a.next();          // floats
            if (rows > 0 && cols > 0 &&
                hipMemcpy2DAsync(dst, (size_t)ldd * 4, src, (size_t)lds * 4, (size_t)cols * 4, (size_t)rows,
                                 hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                return sgcn::fail(SGCN_ERR_HIP, "step_run: hipMemcpy2DAsync failed");
            break;
        }
        default:
            return sgcn::fail(SGCN_ERR_INVALID, "step_run: unknown opcode %d at op %d", op.op, k);
        }
        if (rc != SGCN_OK) { sgcn::aux_join(stream); xchg_join(stream); return rc; }       // the failing entry point has set the message
    }
    const int rj = sgcn::aux_join(stream);  // a run never returns with work pending on the auxiliary stream
    const int rx = xchg_join(stream);       // ... or on the exchange stream: `stream` waits for it (its next kernel is the next run's)
    return rj != SGCN_OK ? rj : rx;
}
