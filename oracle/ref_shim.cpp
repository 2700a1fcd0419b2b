// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// A C-ABI shim around the *real* reference C++ (compiled from the sources where they lie
// under /root/reference/gcn by oracle/Makefile; output goes to oracle/_ref/libsgcn_ref.so,
// which is git-ignored).  It lets tests/golden/make_golden.py and the CPU tests drive the
// reference's own `Scheduler` (gcn/scheduler.h:6-28), `Mult` (gcn/mult.h:8-27) and the row
// slicers (gcn/history.h:6-9) without Cython.  Nothing in this file restates reference
// logic; it only forwards calls and exposes the public vectors.
#include "scheduler.h"   // resolved with -I/root/reference/gcn
#include "mult.h"
#include "history.h"
#include <cstring>

extern "C" {

// ---- Scheduler ---------------------------------------------------------------------------
void* ref_sched_create(float* adj_w, int* adj_i, int* adj_p, int num_data, int num_edges,
                       int L, int cv, int is) {
    return new Scheduler(adj_w, adj_i, adj_p, num_data, num_edges, L, cv != 0, is != 0);
}
void ref_sched_destroy(void* h) { delete static_cast<Scheduler*>(h); }
void ref_sched_seed(void* h, int seed) { static_cast<Scheduler*>(h)->seed(seed); }
void ref_sched_start_batch(void* h, int n, int* data) {
    static_cast<Scheduler*>(h)->start_batch(n, data);
}
void ref_sched_expand(void* h, int degree) { static_cast<Scheduler*>(h)->expand(degree); }

// which: 0 field, 1 ffield, 2 edg_s, 3 edg_t, 4 fedg_s, 5 fedg_t, 6 adj_i (private CSR copy)
static const vector<int>& ivec(Scheduler* s, int which) {
    switch (which) {
        case 0: return s->field;
        case 1: return s->ffield;
        case 2: return s->edg_s;
        case 3: return s->edg_t;
        case 4: return s->fedg_s;
        case 5: return s->fedg_t;
        default: return s->adj_i;
    }
}
// which: 0 scales, 1 edg_w, 2 medg_w, 3 fedg_w, 4 adj_w (private CSR copy)
static const vector<float>& fvec(Scheduler* s, int which) {
    switch (which) {
        case 0: return s->scales;
        case 1: return s->edg_w;
        case 2: return s->medg_w;
        case 3: return s->fedg_w;
        default: return s->adj_w;
    }
}
int ref_sched_isize(void* h, int which) { return (int)ivec(static_cast<Scheduler*>(h), which).size(); }
int ref_sched_fsize(void* h, int which) { return (int)fvec(static_cast<Scheduler*>(h), which).size(); }
void ref_sched_icopy(void* h, int which, int* out) {
    const vector<int>& v = ivec(static_cast<Scheduler*>(h), which);
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(int));
}
void ref_sched_fcopy(void* h, int which, float* out) {
    const vector<float>& v = fvec(static_cast<Scheduler*>(h), which);
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(float));
}

// ---- Mult --------------------------------------------------------------------------------
void* ref_mult_create(const float* prob, int n) {
    try {
        return new Mult(std::vector<float>(prob, prob + n));
    } catch (...) {
        return nullptr;
    }
}
void ref_mult_destroy(void* h) { delete static_cast<Mult*>(h); }
int ref_mult_bit_size(void* h) { return (int)static_cast<Mult*>(h)->bit.size(); }
void ref_mult_bit_copy(void* h, float* out) {
    Mult* m = static_cast<Mult*>(h);
    memcpy(out, m->bit.data(), m->bit.size() * sizeof(float));
}
int ref_mult_query_u(void* h, float u) { return static_cast<Mult*>(h)->Query(u); }
int ref_mult_query(void* h) { return static_cast<Mult*>(h)->Query(); }

// ---- history slicers ---------------------------------------------------------------------
void ref_c_indptr(int N, int* r, int* a_i, int* o_i) { c_indptr(N, r, a_i, o_i); }
void ref_c_slice(int N, int* r, float* a_d, int* a_i, int* a_p, float* o_d, int* o_i, int* o_p) {
    c_slice(N, r, a_d, a_i, a_p, o_d, o_i, o_p);
}
void ref_c_dense_slice(int N, int C, int* r, float* i_data, float* o_data) {
    c_dense_slice(N, C, r, i_data, o_data);
}

}  // extern "C"
