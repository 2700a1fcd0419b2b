// Host neighbour sampler of libsgcn.so (stays on host by design: BASELINE.json north_star).
//
// Semantics: one call to expand(degree) grows the receptive field by one hop exactly as the
// reference `Scheduler::expand` does (gcn/scheduler.cpp:46-189; walk-through in SURVEY.md
// §3.3): per output row a partial Fisher-Yates draw of min(deg, degree) neighbours without
// replacement that permutes the sampler's PRIVATE copy of the CSR in place (state persists
// across batches, gcn/scheduler.cpp:144-145), receptive-field dedup in first-seen order,
// and -- in control-variate mode -- the full neighbour list of every output row in the
// post-permutation order.  Index output is bit-exact with the reference for the same seed
// and call sequence (tests/test_sampler.py, fixtures in tests/golden/).
//
// What is different from the reference (MI355X-first):
//   * the result is emitted as CSR (rowptr built on the fly; rows come out grouped and in
//     order, gcn/scheduler.cpp:126,153) next to the COO the reference exposes, because the
//     HIP kernels consume CSR row tiles;
//   * the transposed sampled adjacency (CSR of A^T) is produced here by a counting sort, so
//     the backward SpMM dX = A^T dY is a gather kernel too (no float atomics on device);
//   * explicit MT19937 / float draw (sgcn_host.h) instead of <random>.
#include "sgcn_host.h"
#include "../../include/sgcn.h"

#if defined(__x86_64__)
#include <immintrin.h>          // the AVX-512 placement loop; every other host takes the scalar loop (same numbering)
#endif
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace sgcn {

// sgcn_plan.cpp
void plan_build(const int32_t* rowptr, int32_t M, int32_t T, std::vector<int32_t>& seg,
                std::vector<int32_t>& fix, int64_t& nslots);

char* error_slot() {
    static thread_local char buf[512] = {0};
    return buf;
}

// ---- Fenwick multinomial (gcn/mult.h:8-27, gcn/mult.cpp:7-51) -------------------------------
class FenwickMultinomial {
public:
    // Throws nothing; `ok()` is false for an empty probability vector (mult.cpp:17-18).
    explicit FenwickMultinomial(const float* p, int n) : weight_(p, p + n), total_(0.f) {
        cap_ = 1;
        while (cap_ < n) cap_ <<= 1;          // smallest power of two >= n  (mult.cpp:9)
        if (n == 0) cap_ = 0;
        tree_.assign((size_t)cap_ + 1, 0.f);
        for (int i = 0; i < n; i++) add(i + 1, weight_[i]);
        last_ = n - 1;
    }
    bool ok() const { return !weight_.empty(); }
    const std::vector<float>& tree() const { return tree_; }

    // Largest prefix whose cumulative weight is <= u   (mult.cpp:38-51).
    int descend(float u) const {
        int pos = 0;
        for (int step = cap_; step > 0; step >>= 1) {
            int nxt = pos + step;
            if (nxt <= cap_ && !(tree_[nxt] > u)) {
                u -= tree_[nxt];
                pos = nxt;
            }
        }
        return pos;
    }
    // Draw one item and remove it   (mult.cpp:29-36).  The generator is the structure's
    // own default-seeded engine (mult.h:25-26), i.e. independent of the sampler seed.
    int draw() {
        float u = rng_.u01() * total_;
        int r = std::min(descend(u), last_);
        add(r + 1, -weight_[r]);
        weight_[r] = 0.f;
        return r;
    }

private:
    void add(int i, float v) {
        for (; i <= cap_; i += i & (-i)) tree_[i] += v;
        total_ += v;
    }
    std::vector<float> weight_, tree_;
    float total_;
    int cap_, last_;
    Mt19937 rng_;
};

// ---- neighbour sampler ------------------------------------------------------------------------
// Two halves.  The CORE is the reference's sequential state machine: the Mersenne twister, the private CSR it
// permutes, the first-seen table of the sampled field.  Its output per hop is a Hop whose full-neighbour lists
// still hold VERTEX ids.  The PACKER is a pure function of a Hop sequence: first-seen numbering of those lists
// (the reference's second table, scheduler.cpp:167-179), the transposed CSR, the launch plans, the packed
// layout.  One thread running both reproduces `Scheduler::expand` call by call; the prefetcher runs the core on
// one thread and packers on others (sgcn_prefetch, below) -- same bits, the core alone on the critical path.

// growable array WITHOUT value-initialisation (std::vector::resize zero-fills: 400 KB per Reddit batch)
template <class T> struct Buf {
    std::unique_ptr<T[]> p;
    size_t n = 0, cap = 0;
    T* data() { return p.get(); }
    const T* data() const { return p.get(); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void clear() { n = 0; }
    void reserve(size_t c) {
        if (c <= cap) return;
        const size_t nc = std::max(c, cap * 2 + 64);
        std::unique_ptr<T[]> q(new T[nc]);
        if (n) memcpy(q.get(), p.get(), n * sizeof(T));
        p.swap(q);
        cap = nc;
    }
    T* grow(size_t k) { reserve(n + k); T* r = p.get() + n; n += k; return r; }     // uninitialised tail
    void push_back(T v) { if (n == cap) reserve(n + 1); p[n++] = v; }
    void assign(const T* b, size_t k) { n = 0; reserve(k); if (k) memcpy(p.get(), b, k * sizeof(T)); n = k; }
    void fill(size_t k, T v) { n = 0; reserve(k); std::fill(p.get(), p.get() + k, v); n = k; }
    void shrink(size_t k) { n = k; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

struct Hop {
    Buf<int32_t> field;                      // the receptive field after the hop (= the layer's input rows)
    Buf<int32_t> edg_s, edg_t, edg_p;        // sampled edges: COO row, column (position in `field`), CSR offsets
    Buf<float> scales, edg_w, medg_w;
    Buf<int32_t> fedg_s, fedg_t, fedg_p;     // every neighbour of every output row; fedg_t: vertex ids until relabelled
    Buf<float> fedg_w;
    int rc = SGCN_OK;
    bool relabelled = false;
    char err[160] = {0};
    void clear() {
        field.clear(); edg_s.clear(); edg_t.clear(); edg_p.clear(); scales.clear(); edg_w.clear(); medg_w.clear();
        fedg_s.clear(); fedg_t.clear(); fedg_p.clear(); fedg_w.clear();
        rc = SGCN_OK; relabelled = false; err[0] = 0;
    }
};

// First-seen numbering of the full-neighbour lists (scheduler.cpp:167-179): vertex ids -> positions in `ffield`.
class Relabeller {
public:
    explicit Relabeller(int32_t n) : fslot_((size_t)n, -1) {}

    void run(Buf<int32_t>& t, Buf<int32_t>& ffield) {
        const size_t ne = t.size();
        ffield.clear();
        ffield.reserve(ne + 16);
        out_ = ffield.data();
        cnt_ = 0;
        if (avx512_) run_avx512(t.data(), ne);
        else for (size_t e = 0; e < ne; e++) t[e] = place(t[e]);
        ffield.shrink((size_t)cnt_);
        // entries are stamped (fbase_ + position): advancing the base un-marks this hop's field without touching
        // its ~43 k scattered table entries again
        fbase_ += cnt_;
        if (fbase_ > (1 << 30)) { std::fill(fslot_.begin(), fslot_.end(), -1); fbase_ = 0; }
    }

private:
    int32_t place(int32_t v) {
        int32_t& s = fslot_[(size_t)v];
        if (s < fbase_) {
            s = fbase_ + cnt_;
            out_[cnt_++] = v;
        }
        return s - fbase_;
    }
#if defined(__x86_64__)
    // 16 entries per step (AVX-512: gather the table entries, give the unseen vertices consecutive positions in
    // list order with an expand, scatter the new stamps, append the new vertices with a compress-store).  Same
    // first-seen numbering as the scalar loop: the entries of a vector are applied "at once", which differs from
    // one-by-one only if a vertex occurs twice among the unseen lanes -- detected with vpconflictd and handed to
    // the scalar loop.  (This numbering was 2/3 of the sampler's time: 50 k table lookups per Reddit batch.)
    __attribute__((target("avx512f,avx512cd")))
    void run_avx512(int32_t* t, size_t ne) {
        int32_t next = 0;
        const __m512i base = _mm512_set1_epi32(fbase_);
        const __m512i iota = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        const __m512i neg = _mm512_sub_epi32(_mm512_setzero_si512(), _mm512_add_epi32(iota, _mm512_set1_epi32(1)));
        for (size_t k = 0; k < ne; k += 16) {
            const __mmask16 m = ne - k >= 16 ? (__mmask16)0xffff : (__mmask16)((1u << (ne - k)) - 1u);
            const __m512i v = _mm512_maskz_loadu_epi32(m, t + k);
            __m512i sl = _mm512_mask_i32gather_epi32(_mm512_set1_epi32(-1), m, v, fslot_.data(), 4);
            const __mmask16 fresh = _mm512_mask_cmplt_epi32_mask(m, sl, base);
            if (fresh) {
                const __m512i conf = _mm512_conflict_epi32(_mm512_mask_mov_epi32(neg, fresh, v));
                if (_mm512_mask_test_epi32_mask(fresh, conf, conf)) {      // a repeated unseen vertex: one by one
                    cnt_ = next;
                    for (size_t q = k; q < std::min(ne, k + 16); q++) t[q] = place(t[q]);
                    next = cnt_;
                    continue;
                }
                const __m512i ids = _mm512_maskz_expand_epi32(fresh, _mm512_add_epi32(iota, _mm512_set1_epi32(next)));
                const __m512i stamped = _mm512_add_epi32(ids, base);
                _mm512_mask_i32scatter_epi32(fslot_.data(), fresh, v, stamped, 4);
                _mm512_mask_compressstoreu_epi32(out_ + next, fresh, v);
                next += __builtin_popcount((unsigned)fresh);
                sl = _mm512_mask_mov_epi32(sl, fresh, stamped);
            }
            _mm512_mask_storeu_epi32(t + k, m, _mm512_sub_epi32(sl, base));
        }
        cnt_ = next;
    }
    const bool avx512_ = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512cd") && !getenv("SGCN_NO_AVX512");
#else
    void run_avx512(int32_t* t, size_t ne) { for (size_t e = 0; e < ne; e++) t[e] = place(t[e]); }   // never taken
    const bool avx512_ = false;
#endif
    std::vector<int32_t> fslot_;
    int32_t fbase_ = 0, cnt_ = 0;          // stamp base: an entry >= fbase_ is a position in this hop's ffield
    int32_t* out_ = nullptr;
};

// ---- packed minibatch (one call = PyScheduler.batch, gcn/_scheduler.pyx:55-127) ----------
// Lays every array the device step needs into two growable staging vectors (int32 / fp32), each sub-array
// 16-byte aligned: fields, ffields, scales, labels[fields[-1]], and per layer the CSR of adj, adj^T and fadj
// with their row plans.  `meta` receives (offset, length) descriptors, layer 0 = input-most layer (the reversal
// of gcn/_scheduler.pyx:121-126).  No Python objects are touched: the call runs without the GIL.
class Packer {
public:
    enum { kCsrDesc = 11 };   // nrows ncols nnz rowptr col val seg nseg fix nfix nslots
    explicit Packer(int32_t n) : relabel_(n) {}
    // (... + L x (off,len) of the medg weights in TRANSPOSED order, appended behind the CSR descriptors: ABI v16)
    static int64_t meta_len(int32_t L) { return 4 + 2 * (L + 1) + 2 * L + 2 * L + 3 + 2 * L + 3 * L * kCsrDesc + 2 * L; }

    void relabel(Hop& h) {
        if (h.relabelled) return;
        relabel_.run(h.fedg_t, ffield);
        h.relabelled = true;
    }
    // CSR of A^T (A = the hop's sampled adjacency, n_out x |field|): stable counting sort by column, so within a
    // transposed row entries keep ascending output-row order (deterministic).
    void transpose(const Hop& h) {
        const size_t n_in = h.field.size(), ne = h.edg_t.size();
        tedg_p.fill(n_in + 1, 0);
        for (size_t e = 0; e < ne; e++) tedg_p[(size_t)h.edg_t[e] + 1]++;
        for (size_t c = 0; c < n_in; c++) tedg_p[c + 1] += tedg_p[c];
        tedg_t.clear(); tedg_t.grow(ne);
        tedg_w.clear(); tedg_w.grow(ne);
        // the det-dropout aggregator's third matrix (the adjacency's pattern with the medg weights, gcn/scheduler.cpp:164)
        // needs its transpose on the way back: the same permutation applied to the medg weights (control-variate hops only)
        const bool with_medg = h.medg_w.size() == ne;
        tmedg_w.clear();
        if (with_medg) tmedg_w.grow(ne);
        cur_.assign(tedg_p.data(), n_in);
        for (size_t e = 0; e < ne; e++) {
            const int32_t q = cur_[(size_t)h.edg_t[e]]++;
            tedg_t[(size_t)q] = h.edg_s[e];
            tedg_w[(size_t)q] = h.edg_w[e];
            if (with_medg) tmedg_w[(size_t)q] = h.medg_w[e];
        }
    }

    // hops[l] = the l-th expansion (hops[0] grew the batch itself).  Relabels the hops in place.
    int pack(int32_t n, const int32_t* ids, Hop* hops, int32_t L, bool cv, const float* labels, int32_t n_classes,
             int32_t plan_T, int64_t* meta, int64_t meta_cap, int64_t* n_i32, int64_t* n_f32) {
        const int64_t need = meta_len(L);
        if (meta_cap < need) return fail(SGCN_ERR_INVALID, "batch_packed: meta too small (%lld < %lld)",
                                         (long long)meta_cap, (long long)need);
        pi_.clear(); pf_.clear();
        std::fill(meta, meta + need, 0);
        meta[0] = L; meta[1] = cv ? 1 : 0; meta[2] = n_classes;
        int64_t* fields_d = meta + 4;                    // (L+1) x (off,len)
        int64_t* scales_d = fields_d + 2 * (L + 1);      // L x (off,len)
        int64_t* ffields_d = scales_d + 2 * L;           // L x (off,len)
        int64_t* labels_d = ffields_d + 2 * L;           // off, rows, cols
        int64_t* medg_d = labels_d + 3;                  // L x (off,len)
        int64_t* csr_d = medg_d + 2 * L;                 // L x 3 x kCsrDesc  (adj, adjT, fadj)
        int64_t* tmedg_d = csr_d + 3 * (int64_t)L * kCsrDesc;   // L x (off,len): medg in the order of adjT's nonzeros
        put_i(ids, (size_t)n, fields_d + 2 * L);         // fields[L] = the batch itself
        int rc = SGCN_OK;
        for (int32_t l = 0; l < L; l++) {
            Hop& h = hops[l];
            const int32_t slot = L - 1 - l;              // position after the reversal
            if (h.rc != SGCN_OK) rc = fail(h.rc, "%s", h.err);
            const int32_t n1 = (int32_t)h.edg_p.size() - 1, n0 = (int32_t)h.field.size();
            put_i(h.field.data(), h.field.size(), fields_d + 2 * slot);
            put_f(h.scales.data(), h.scales.size(), scales_d + 2 * slot);
            put_f(h.medg_w.data(), h.medg_w.size(), medg_d + 2 * slot);
            transpose(h);
            put_csr(csr_d + (3 * slot + 0) * kCsrDesc, n1, n0, h.edg_p, h.edg_t, h.edg_w, plan_T);
            put_csr(csr_d + (3 * slot + 1) * kCsrDesc, n0, n1, tedg_p, tedg_t, tedg_w, plan_T);
            put_f(tmedg_w.data(), tmedg_w.size(), tmedg_d + 2 * slot);
            if (cv) {
                relabel(h);
                put_i(ffield.data(), ffield.size(), ffields_d + 2 * slot);
                // the full-neighbour matrix is consumed by the aggregator, one WORKGROUP per plan segment (sgcn_agg.hip):
                // 128 nonzeros = one round of the workgroup's 8 lane groups x 16 loads in flight
                put_csr(csr_d + (3 * slot + 2) * kCsrDesc, n1, (int32_t)ffield.size(), h.fedg_p, h.fedg_t, h.fedg_w,
                        std::max(plan_T, (int32_t)SGCN_AGG_PLAN_T));
            }
        }
        if (labels && n_classes > 0) {                   // labels[fields[-1]]  (_scheduler.pyx:138)
            labels_d[0] = (int64_t)pf_.size(); labels_d[1] = n; labels_d[2] = n_classes;
            float* dst = pf_.grow((size_t)n * (size_t)n_classes);
            for (int32_t i = 0; i < n; i++)
                memcpy(dst + (size_t)i * n_classes, labels + (int64_t)ids[i] * n_classes, sizeof(float) * (size_t)n_classes);
            pad_f();
        }
        *n_i32 = (int64_t)pi_.size();
        *n_f32 = (int64_t)pf_.size();
        return rc;
    }
    void copy_out(int32_t* di, float* df) const {
        if (di && !pi_.empty()) memcpy(di, pi_.data(), pi_.size() * sizeof(int32_t));
        if (df && !pf_.empty()) memcpy(df, pf_.data(), pf_.size() * sizeof(float));
    }

    Buf<int32_t> ffield, tedg_p, tedg_t;     // of the hop last relabelled / transposed (the view API reads them)
    Buf<float> tedg_w, tmedg_w;

private:
    void pad_i() { while (pi_.size() & 3) pi_.push_back(0); }
    void pad_f() { while (pf_.size() & 3) pf_.push_back(0.f); }
    void put_i(const int32_t* v, size_t k, int64_t* d) {
        d[0] = (int64_t)pi_.size(); d[1] = (int64_t)k;
        if (k) memcpy(pi_.grow(k), v, k * sizeof(int32_t));
        pad_i();
    }
    void put_f(const float* v, size_t k, int64_t* d) {
        d[0] = (int64_t)pf_.size(); d[1] = (int64_t)k;
        if (k) memcpy(pf_.grow(k), v, k * sizeof(float));
        pad_f();
    }
    void put_csr(int64_t* d, int32_t nrows, int32_t ncols, const Buf<int32_t>& rowptr, const Buf<int32_t>& col,
                 const Buf<float>& val, int32_t plan_T) {
        int64_t tmp[2];
        d[0] = nrows; d[1] = ncols; d[2] = (int64_t)col.size();
        put_i(rowptr.data(), rowptr.size(), tmp); d[3] = tmp[0];
        put_i(col.data(), col.size(), tmp); d[4] = tmp[0];
        put_f(val.data(), val.size(), tmp); d[5] = tmp[0];
        int64_t nslots = 0;
        plan_build(rowptr.data(), nrows, plan_T, seg_, fix_, nslots);
        put_i(seg_.data(), seg_.size(), tmp); d[6] = tmp[0]; d[7] = (int64_t)seg_.size() / 4;
        put_i(fix_.data(), fix_.size(), tmp); d[8] = tmp[0]; d[9] = (int64_t)fix_.size() / 3;
        d[10] = nslots;
    }
    Relabeller relabel_;
    Buf<int32_t> pi_, cur_;
    Buf<float> pf_;
    std::vector<int32_t> seg_, fix_;
};

class NeighbourSampler {
public:
    NeighbourSampler(const float* w, const int32_t* idx, const int32_t* ptr, int32_t n,
                     int32_t nnz, bool cv, bool is)
        : n_(n), cv_(cv), is_(is), nbr_(idx, idx + nnz), wgt_(w, w + nnz), ptr_(ptr, ptr + n),
          slot_(n, -1), importance_(n, 1.0f), packer_(n) {
        ptr_.push_back(nnz);  // the reference trusts only the first n offsets (scheduler.cpp:16,20)
        if (is_) {
            // column-wise squared weight mass on top of 1e-6   (scheduler.cpp:18,22-25)
            std::fill(importance_.begin(), importance_.end(), 1e-6f);
            for (int32_t r = 0; r < n; r++)
                for (int32_t p = ptr[r]; p < ptr[r + 1]; p++) importance_[idx[p]] += w[p] * w[p];
        }
    }
    int32_t num_vertices() const { return n_; }
    bool cv() const { return cv_; }

    void seed(int32_t s) { rng_.reseed((uint32_t)s); }

    void start_batch(int32_t n, const int32_t* ids) { field_.assign(ids, (size_t)n); }

    // ---- the core: one hop of the receptive field into `h` (full-neighbour lists as vertex ids) ----------------
    int expand_core(int32_t degree, Hop& h, bool want_coo) {
        const size_t n_out = field_.size();
        h.clear();
        // output rows are the first |field| entries of the next field   (scheduler.cpp:50-52)
        h.field.assign(field_.data(), n_out);
        for (size_t i = 0; i < n_out; i++) slot_[(size_t)field_[i]] = (int32_t)i;
        h.edg_p.push_back(0);
        h.fedg_p.push_back(0);
        h.rc = is_ ? expand_importance(degree, h) : expand_uniform(degree, n_out, h, want_coo);
        if (h.rc != SGCN_OK) snprintf(h.err, sizeof(h.err), "%s", error_slot());
        for (size_t i = 0; i < h.field.size(); i++) slot_[(size_t)h.field[i]] = -1;
        field_.assign(h.field.data(), h.field.size());
        return h.rc;
    }
    // the L hops of a minibatch (gcn/_scheduler.pyx:60-66)
    int sample_batch(int32_t n, const int32_t* ids, int32_t L, const int32_t* degrees, std::vector<Hop>& hops) {
        if ((int32_t)hops.size() < L) hops.resize((size_t)L);
        start_batch(n, ids);
        int rc = SGCN_OK;
        for (int32_t l = 0; l < L; l++) {
            const int r = expand_core(degrees[L - l - 1], hops[(size_t)l], false);    // the packed layout is CSR only
            if (r != SGCN_OK) rc = r;
        }
        return rc;
    }

    // ---- the reference's call-by-call interface: expand, then read the arrays ----------------------------------
    int expand(int32_t degree) {
        const int rc = expand_core(degree, cur_, true);
        if (cv_) packer_.relabel(cur_);
        else packer_.ffield.clear();
        transpose_ready_ = false;
        return rc;
    }
    void build_transpose() {
        if (transpose_ready_) return;
        packer_.transpose(cur_);
        transpose_ready_ = true;
    }
    bool ivec(int which, const int32_t** p, int64_t* len) {
        const Buf<int32_t>* b = nullptr;
        switch (which) {
            case SGCN_SCHED_FIELD: *p = field_.data(); *len = (int64_t)field_.size(); return true;
            case SGCN_SCHED_FFIELD: b = &packer_.ffield; break;
            case SGCN_SCHED_EDG_S: b = &cur_.edg_s; break;
            case SGCN_SCHED_EDG_T: b = &cur_.edg_t; break;
            case SGCN_SCHED_FEDG_S: b = &cur_.fedg_s; break;
            case SGCN_SCHED_FEDG_T: b = &cur_.fedg_t; break;
            case SGCN_SCHED_EDG_P: b = &cur_.edg_p; break;
            case SGCN_SCHED_FEDG_P: b = &cur_.fedg_p; break;
            case SGCN_SCHED_ADJ_I: *p = nbr_.data(); *len = (int64_t)nbr_.size(); return true;
            case SGCN_SCHED_TEDG_P: build_transpose(); b = &packer_.tedg_p; break;
            case SGCN_SCHED_TEDG_T: build_transpose(); b = &packer_.tedg_t; break;
            default: return false;
        }
        *p = b->data(); *len = (int64_t)b->size();
        return true;
    }
    bool fvec(int which, const float** p, int64_t* len) {
        const Buf<float>* b = nullptr;
        switch (which) {
            case SGCN_SCHED_SCALES: b = &cur_.scales; break;
            case SGCN_SCHED_EDG_W: b = &cur_.edg_w; break;
            case SGCN_SCHED_MEDG_W: b = &cur_.medg_w; break;
            case SGCN_SCHED_FEDG_W: b = &cur_.fedg_w; break;
            case SGCN_SCHED_ADJ_W: *p = wgt_.data(); *len = (int64_t)wgt_.size(); return true;
            case SGCN_SCHED_TEDG_W: build_transpose(); b = &packer_.tedg_w; break;
            default: return false;
        }
        *p = b->data(); *len = (int64_t)b->size();
        return true;
    }

    // one thread: core, then the own packer
    int pack_batch(int32_t n, const int32_t* ids, int32_t L, const int32_t* degrees,
                   const float* labels, int32_t n_classes, int32_t plan_T, int64_t* meta,
                   int64_t meta_cap, int64_t* n_i32, int64_t* n_f32) {
        if (meta_cap < Packer::meta_len(L))
            return fail(SGCN_ERR_INVALID, "batch_packed: meta too small (%lld < %lld)", (long long)meta_cap,
                        (long long)Packer::meta_len(L));
        sample_batch(n, ids, L, degrees, hops_);
        return packer_.pack(n, ids, hops_.data(), L, cv_, labels, n_classes, plan_T, meta, meta_cap, n_i32, n_f32);
    }
    static int64_t meta_len(int32_t L) { return Packer::meta_len(L); }
    void packed_copy(int32_t* di, float* df) const { packer_.copy_out(di, df); }

private:
    // position of vertex v in the growing next field (appending it on first sight)
    int32_t place(int32_t v, Hop& h) {
        int32_t& s = slot_[(size_t)v];
        if (s < 0) {
            s = (int32_t)h.field.size();
            h.field.push_back(v);
        }
        return s;
    }

    // Uniform sampling w/o replacement, optional control-variate extras (scheduler.cpp:125-180)
    int expand_uniform(int32_t degree, size_t n_out, Hop& h, bool want_coo) {
        // The batch's rows are scattered over a CSR of hundreds of MB (shuffled train ids): every row
        // starts with two DRAM misses (neighbour ids, weights).  The ids are known up front, so the rows a
        // few iterations ahead are prefetched -- the row pointer first (it is needed to form the address),
        // then the lines of both arrays (rows are too short to train the hardware prefetcher).
        constexpr size_t kAheadPtr = 16, kAhead = 8;
        constexpr int kLines = 16;
        for (size_t i = 0; i < n_out; i++) {
            if (i + kAheadPtr < n_out) __builtin_prefetch(ptr_.data() + field_[i + kAheadPtr], 0, 1);
            if (i + kAhead < n_out) {
                const int32_t pv = ptr_[(size_t)field_[i + kAhead]];
                const int32_t pdeg = ptr_[(size_t)field_[i + kAhead] + 1] - pv;
                const char* c0 = reinterpret_cast<const char*>(nbr_.data() + pv);
                const char* w0 = reinterpret_cast<const char*>(wgt_.data() + pv);
                const int lines = std::min(kLines, (pdeg * 4 + 63) / 64);
                for (int q = 0; q < lines; q++) { __builtin_prefetch(c0 + 64 * q, 1, 1); __builtin_prefetch(w0 + 64 * q, 1, 1); }
            }
            const int32_t v = field_[i];
            int32_t* cols = nbr_.data() + ptr_[(size_t)v];
            float* vals = wgt_.data() + ptr_[(size_t)v];
            const int32_t deg = ptr_[(size_t)v + 1] - ptr_[(size_t)v];
            const int32_t take = std::min(deg, degree);
            // amplification deg/take in fp32; isolated rows amplify by 1  (scheduler.cpp:132-133)
            const float amp = deg == 0 ? 1.0f : (float)deg / (float)take;
            // 1/sqrt(amp): fp32 sqrt, fp64 reciprocal, stored fp32        (scheduler.cpp:134)
            h.scales.push_back((float)(1.0 / (double)std::sqrt(amp)));

            for (int32_t k = 0; k < take; k++) {
                // pick a position in [k, deg) and move it to the front segment; the product
                // and the sum are separate fp32 roundings (no FMA: built with -ffp-contract=off)
                const float span = (float)(deg - k) * rng_.u01();
                const float posf = (float)k + span;
                const int32_t j = std::min((int32_t)posf, deg - 1);
                std::swap(cols[k], cols[j]);
                std::swap(vals[k], vals[j]);
                const float w = vals[k] * amp;
                h.edg_s.push_back((int32_t)i);
                h.edg_t.push_back(place(cols[k], h));
                h.edg_w.push_back(w);
                if (cv_) h.medg_w.push_back(vals[k] * w);
            }
            h.edg_p.push_back((int32_t)h.edg_t.size());

            if (cv_) {
                // every neighbour, in the row's current (post-swap) order   (scheduler.cpp:167-179): the vertex
                // ids now, their first-seen positions when a packer relabels the hop
                if (deg) {
                    memcpy(h.fedg_t.grow((size_t)deg), cols, (size_t)deg * sizeof(int32_t));
                    memcpy(h.fedg_w.grow((size_t)deg), vals, (size_t)deg * sizeof(float));
                    if (want_coo) std::fill_n(h.fedg_s.grow((size_t)deg), (size_t)deg, (int32_t)i);
                }
                h.fedg_p.push_back((int32_t)h.fedg_t.size());
            }
        }
        if (!cv_) h.fedg_p.fill(n_out + 1, 0);
        return SGCN_OK;
    }

    // Importance sampling of the joint neighbourhood (scheduler.cpp:63-123)
    int expand_importance(int32_t degree, Hop& h) {
        const size_t n_out = field_.size();
        std::vector<int32_t> cand;
        std::vector<float> prob;
        std::vector<char> seen((size_t)n_, 0);
        std::vector<int32_t> hits((size_t)n_, 0);
        float mass = 0.f;
        for (size_t i = 0; i < n_out; i++) {
            const int32_t v = field_[i];
            for (int32_t p = ptr_[(size_t)v]; p < ptr_[(size_t)v + 1]; p++) {
                const int32_t t = nbr_[(size_t)p];
                if (!seen[(size_t)t]) {
                    seen[(size_t)t] = 1;
                    cand.push_back(t);
                    mass += importance_[(size_t)t];
                    prob.push_back(importance_[(size_t)t]);
                }
            }
        }
        if (prob.empty()) {
            h.edg_p.fill(n_out + 1, 0);
            h.fedg_p.fill(n_out + 1, 0);
            return fail(SGCN_ERR_EMPTY_PROB, "Prob is empty");
        }
        FenwickMultinomial mult(prob.data(), (int)prob.size());
        const int32_t draws = (int32_t)std::min(n_out * (size_t)degree, cand.size());
        for (int32_t k = 0; k < draws; k++) {
            const int32_t t = cand[(size_t)mult.draw()];
            hits[(size_t)t]++;
            place(t, h);
        }
        int rc = SGCN_OK;
        for (size_t i = 0; i < n_out; i++) {
            const int32_t v = field_[i];
            for (int32_t p = ptr_[(size_t)v]; p < ptr_[(size_t)v + 1]; p++) {
                const int32_t t = nbr_[(size_t)p];
                if (!hits[(size_t)t]) continue;
                // ((hits*w)*mass) / (importance*draws), all fp32, left to right (scheduler.cpp:107-108)
                const float num = ((float)hits[(size_t)t] * wgt_[(size_t)p]) * mass;
                const float den = importance_[(size_t)t] * (float)draws;
                const float w = num / den;
                h.edg_s.push_back((int32_t)i);
                h.edg_t.push_back(slot_[(size_t)t]);
                h.edg_w.push_back(w);
                if (std::isnan(w)) rc = fail(SGCN_ERR_NAN, "nan");
            }
            h.edg_p.push_back((int32_t)h.edg_t.size());
        }
        h.fedg_p.fill(n_out + 1, 0);
        return rc;
    }

    int32_t n_;
    bool cv_, is_, transpose_ready_ = false;
    std::vector<int32_t> nbr_;   // private, permuted in place
    std::vector<float> wgt_;
    std::vector<int32_t> ptr_;
    std::vector<int32_t> slot_;
    std::vector<float> importance_;
    Buf<int32_t> field_;         // the current receptive field (input of the next hop)
    Hop cur_;                    // the call-by-call interface's hop
    std::vector<Hop> hops_;      // pack_batch's hops
    Packer packer_;
    Mt19937 rng_;
};

}  // namespace sgcn

// ---- C ABI --------------------------------------------------------------------------------------
struct sgcn_sched { sgcn::NeighbourSampler impl; };

// The CPUs of the NUMA node the calling thread runs on (Linux sysfs).  The sampler walks a private
// ~200 MB CSR copy that the creating thread first-touched; a producer thread scheduled on the other
// socket of a two-socket host measured 0.40 ms per batch instead of 0.28.
static bool numa_cpus_of_caller(cpu_set_t* set) {
    if (getenv("SGCN_NO_AFFINITY")) return false;
    const int cpu = sched_getcpu();
    if (cpu < 0) return false;
    for (int node = 0; node < 64; node++) {
        char path[96];
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        FILE* f = fopen(path, "r");
        if (!f) { if (node == 0) return false; break; }
        char buf[4096];
        const bool ok = fgets(buf, sizeof(buf), f) != nullptr;
        fclose(f);
        if (!ok) continue;
        CPU_ZERO(set);
        bool mine = false;
        for (char* p = buf; *p;) {                      // "0-63,128-191"
            char* e;
            const long lo = strtol(p, &e, 10);
            if (e == p) break;
            long hi = lo;
            if (*e == '-') { p = e + 1; hi = strtol(p, &e, 10); }
            for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) { CPU_SET((int)c, set); if (c == cpu) mine = true; }
            p = (*e == ',') ? e + 1 : e;
            if (*e != ',' ) break;
        }
        if (mine) return true;
    }
    return false;
}

// ---- native prefetch threads ------------------------------------------------------------------
// The sampler of an epoch as C++ threads: the epoch's id slices are walked in order (so the sample
// sequence is the synchronous loop's, bit for bit), each minibatch is packed and copied into the
// next free pinned staging slot.  No Python runs on these threads: a Python producer thread has to
// re-take the interpreter lock after every foreign call, and against a launching thread that
// releases and re-takes it every ~20 us that costs ~0.25 ms per batch (measured: 0.5 ms per batch
// next to 0.26 ms alone) -- enough to make the sampler the bottleneck of the epoch.
//
// Three shapes:
//   * one sampler, no packers: one thread samples and packs (rounds 1-2);
//   * one sampler + P packers (round 3, the default): the sampler's CORE (random draws, the in-place permutation,
//     the rows' neighbour lists) runs on one thread and hands each batch's hops to one of P packer threads (numbering
//     of the full-neighbour field, transposes, plans, layout, copy into the slot).  The core is the only sequential
//     part and less than half of the work; the output is the single thread's, bit for bit;
//   * N samplers: N independent streams (NOT the reference's sample sequence: the non-parity fast mode).
struct sgcn_prefetch {
    struct Ready { int32_t batch = 0, slot = 0; int64_t n_i = 0, n_f = 0; std::vector<int64_t> meta; std::unique_ptr<std::vector<int32_t>> spill; };
    struct Raw { int32_t batch = 0; std::vector<sgcn::Hop> hops; };
    std::vector<sgcn_sched*> ss;                   // one sampler per producer thread
    std::vector<int32_t> ids; std::vector<int64_t> off;
    int32_t L = 0, n_classes = 0, plan_T = 0, lag = 2;
    std::vector<int32_t> degrees; const float* labels = nullptr;
    std::vector<void*> slot_words; std::vector<int64_t> slot_caps;
    int64_t meta_len = 0;
    std::mutex mu; std::condition_variable cv_free, cv_ready, cv_raw;
    std::deque<int32_t> free_slots;
    std::vector<Ready> ready; std::vector<char> is_ready;       // indexed by batch
    int32_t next_batch = 0;                                      // the batch the consumer takes next
    std::vector<std::unique_ptr<std::vector<int32_t>>> spills;   // batches that outgrew their slot
    bool stop = false; int32_t running = 0; int error = 0; std::string error_msg;
    std::unique_ptr<std::atomic<char>[]> ready_flag;             // is_ready mirrored for the consumer's short spin before it blocks
    std::atomic<int32_t> taken_raw{0};
    std::atomic<int32_t> raw_count{0};                           // hops handed over by the core so far (same, for the packers)
    double t_wait = 0, t_pack = 0, t_copy = 0, t_sample = 0;     // producer seconds: waiting for a slot / packing / copying / the core
    std::vector<std::thread> ths;
    // pipelined shape
    std::vector<std::unique_ptr<sgcn::Packer>> packers;
    std::vector<std::unique_ptr<Raw>> raw_pool;
    std::deque<Raw*> raw_free, raw_ready;
    bool core_done = false;

    using clk = std::chrono::steady_clock;
    // A blocked thread wakes 50-100 us after its condition variable is signalled (futex + scheduler), which is a whole
    // batch of this pipeline: whoever is about to block polls the thing it waits for for a few tens of microseconds first.
    template <class F> static void spin_for(F&& ready, double us) {
        const auto t0 = clk::now();
        while (!ready()) {
#if defined(__x86_64__)
            for (int i = 0; i < 32; i++) _mm_pause();
#endif
            if (std::chrono::duration<double, std::micro>(clk::now() - t0).count() > us) return;
        }
    }
    static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

    // A batch may take a slot only inside the window of batches the consumer will ask for next: with `window` = slots
    // the consumer never holds, the batches in the window can all be staged at once, so the one the consumer is waiting
    // for can never be starved of a slot by later ones.
    int32_t window() const { return std::max<int32_t>(1, (int32_t)slot_words.size() - lag - 1); }
    bool take_slot(int32_t b, int32_t* slot) {
        std::unique_lock<std::mutex> lk(mu);
        const int32_t w = window();
        cv_free.wait(lk, [&] { return stop || error || (b < next_batch + w && !free_slots.empty()); });
        if (stop || error) return false;
        *slot = free_slots.front(); free_slots.pop_front();
        return true;
    }
    int32_t* dest(Ready& r, int64_t ni, int64_t nf) {
        if (ni + nf <= slot_caps[(size_t)r.slot]) return static_cast<int32_t*>(slot_words[(size_t)r.slot]);
        r.spill.reset(new std::vector<int32_t>((size_t)(ni + nf)));   // rare: the batch outgrew the slot -> heap buffer
        return r.spill->data();
    }
    void publish(Ready&& r, double tw, double tp, double tc) {
        const int32_t b = r.batch;
        {
            std::lock_guard<std::mutex> lk(mu);
            ready[(size_t)b] = std::move(r);
            is_ready[(size_t)b] = 1;
            t_wait += tw; t_pack += tp; t_copy += tc;
        }
        ready_flag[(size_t)b].store(1, std::memory_order_release);
        cv_ready.notify_all();
    }
    void set_error(int rc) {
        std::lock_guard<std::mutex> lk(mu);
        if (!error) { error = rc; error_msg = sgcn::error_slot(); }
    }
    void producer_exit() {
        std::lock_guard<std::mutex> lk(mu);
        running--;
        cv_ready.notify_all();
        cv_free.notify_all();
        cv_raw.notify_all();
    }

    // Thread k builds batches k, k + N, k + 2N, ... with its own sampler (core and packer on the same thread).
    void run(int32_t k) {
        const int32_t nb = (int32_t)off.size() - 1, N = (int32_t)ss.size();
        for (int32_t b = k; b < nb; b += N) {
            const auto c0 = clk::now();
            Ready r; r.batch = b;
            if (!take_slot(b, &r.slot)) break;
            const auto c1 = clk::now();
            r.meta.assign((size_t)meta_len, 0);
            const int rc = ss[(size_t)k]->impl.pack_batch((int32_t)(off[b + 1] - off[b]), ids.data() + off[b], L,
                                                          degrees.data(), labels, n_classes, plan_T, r.meta.data(),
                                                          meta_len, &r.n_i, &r.n_f);
            if (rc != SGCN_OK) { set_error(rc); break; }
            const int64_t ni = std::max<int64_t>(r.n_i, 1), nf = std::max<int64_t>(r.n_f, 1);
            int32_t* dst = dest(r, ni, nf);
            const auto c2 = clk::now();
            ss[(size_t)k]->impl.packed_copy(dst, reinterpret_cast<float*>(dst + ni));
            const auto c3 = clk::now();
            publish(std::move(r), secs(c0, c1), secs(c1, c2), secs(c2, c3));
        }
        producer_exit();
    }

    // The core thread of the pipelined shape: batches in order, each into a Raw from the pool.
    void run_core() {
        const int32_t nb = (int32_t)off.size() - 1;
        for (int32_t b = 0; b < nb; b++) {
            Raw* raw;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_raw.wait(lk, [&] { return stop || error || !raw_free.empty(); });
                if (stop || error) break;
                raw = raw_free.front(); raw_free.pop_front();
            }
            const auto c0 = clk::now();
            raw->batch = b;
            (void)ss[0]->impl.sample_batch((int32_t)(off[b + 1] - off[b]), ids.data() + off[b], L, degrees.data(), raw->hops);
            const auto c1 = clk::now();
            {
                std::lock_guard<std::mutex> lk(mu);
                raw_ready.push_back(raw);
                t_sample += secs(c0, c1);
            }
            raw_count.fetch_add(1, std::memory_order_release);
            cv_raw.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        core_done = true;
        cv_raw.notify_all();
    }
    void run_packer(int32_t j) {
        sgcn::Packer& pk = *packers[(size_t)j];
        const bool cv = ss[0]->impl.cv();
        for (;;) {
            Raw* raw;
            {
                const int32_t seen = taken_raw.load(std::memory_order_relaxed);
                spin_for([&] { return raw_count.load(std::memory_order_acquire) > seen; }, 40.0);
                std::unique_lock<std::mutex> lk(mu);
                cv_raw.wait(lk, [&] { return stop || error || !raw_ready.empty() || core_done; });
                if (stop || error || raw_ready.empty()) break;
                raw = raw_ready.front(); raw_ready.pop_front();       // batches arrive in order: the oldest first
                taken_raw.fetch_add(1, std::memory_order_relaxed);
            }
            const int32_t b = raw->batch;
            const auto c0 = clk::now();
            Ready r; r.batch = b;
            if (!take_slot(b, &r.slot)) break;
            const auto c1 = clk::now();
            r.meta.assign((size_t)meta_len, 0);
            const int32_t n = (int32_t)(off[b + 1] - off[b]);
            const int rc = pk.pack(n, ids.data() + off[b], raw->hops.data(), L, cv, labels, n_classes, plan_T, r.meta.data(),
                                   meta_len, &r.n_i, &r.n_f);
            {
                std::lock_guard<std::mutex> lk(mu);
                raw_free.push_back(raw);
            }
            cv_raw.notify_all();
            if (rc != SGCN_OK) { set_error(rc); break; }
            const int64_t ni = std::max<int64_t>(r.n_i, 1), nf = std::max<int64_t>(r.n_f, 1);
            int32_t* dst = dest(r, ni, nf);
            const auto c2 = clk::now();
            pk.copy_out(dst, reinterpret_cast<float*>(dst + ni));
            const auto c3 = clk::now();
            publish(std::move(r), secs(c0, c1), secs(c1, c2), secs(c2, c3));
        }
        producer_exit();
    }
};

struct sgcn_mult { sgcn::FenwickMultinomial impl; };

extern "C" {

const char* sgcn_last_error(void) { return sgcn::error_slot(); }
int sgcn_abi_version(void) { return 16; }

int sgcn_sched_create(const float* w, const int32_t* idx, const int32_t* ptr, int32_t num_data,
                      int32_t num_edges, int32_t L, int32_t cv, int32_t is, sgcn_sched_t** out) {
    (void)L;
    if (!out || num_data < 0 || num_edges < 0 || (num_edges > 0 && (!w || !idx)) ||
        (num_data > 0 && !ptr))
        return sgcn::fail(SGCN_ERR_INVALID, "sgcn_sched_create: bad argument");
    sgcn_sched* s = new (std::nothrow)
        sgcn_sched{sgcn::NeighbourSampler(w, idx, ptr, num_data, num_edges, cv != 0, is != 0)};
    if (!s) return sgcn::fail(SGCN_ERR_INVALID, "sgcn_sched_create: out of memory");
    *out = s;
    return SGCN_OK;
}
void sgcn_sched_destroy(sgcn_sched_t* s) { delete s; }
int sgcn_sched_seed(sgcn_sched_t* s, int32_t seed) {
    if (!s) return sgcn::fail(SGCN_ERR_INVALID, "null sampler");
    s->impl.seed(seed);
    return SGCN_OK;
}
int sgcn_sched_start_batch(sgcn_sched_t* s, int32_t n, const int32_t* ids) {
    if (!s || n < 0 || (n > 0 && !ids)) return sgcn::fail(SGCN_ERR_INVALID, "start_batch: bad argument");
    s->impl.start_batch(n, ids);
    return SGCN_OK;
}
int sgcn_sched_expand(sgcn_sched_t* s, int32_t degree) {
    if (!s) return sgcn::fail(SGCN_ERR_INVALID, "null sampler");
    return s->impl.expand(degree);
}
int sgcn_sched_view_i32(sgcn_sched_t* s, int32_t which, const int32_t** ptr, int64_t* len) {
    if (!s || !ptr || !len || !s->impl.ivec(which, ptr, len)) return sgcn::fail(SGCN_ERR_INVALID, "view_i32: bad selector %d", which);
    return SGCN_OK;
}
int sgcn_sched_view_f32(sgcn_sched_t* s, int32_t which, const float** ptr, int64_t* len) {
    if (!s || !ptr || !len || !s->impl.fvec(which, ptr, len)) return sgcn::fail(SGCN_ERR_INVALID, "view_f32: bad selector %d", which);
    return SGCN_OK;
}

int sgcn_sched_batch_packed(sgcn_sched_t* s, int32_t n, const int32_t* ids, int32_t L,
                            const int32_t* degrees, const float* labels, int32_t n_classes,
                            int32_t plan_T, int64_t* meta, int64_t meta_cap, int64_t* n_i32,
                            int64_t* n_f32) {
    if (!s || n < 0 || (n > 0 && !ids) || L < 0 || (L > 0 && !degrees) || !meta || !n_i32 || !n_f32)
        return sgcn::fail(SGCN_ERR_INVALID, "batch_packed: bad argument");
    return s->impl.pack_batch(n, ids, L, degrees, labels, n_classes, plan_T, meta, meta_cap, n_i32, n_f32);
}
int sgcn_sched_batch_packed_into(sgcn_sched_t* s, int32_t n, const int32_t* ids, int32_t L,
                                 const int32_t* degrees, const float* labels, int32_t n_classes,
                                 int32_t plan_T, int64_t* meta, int64_t meta_cap, void* words,
                                 int64_t cap_words, int64_t* n_i32, int64_t* n_f32) {
    const int rc = sgcn_sched_batch_packed(s, n, ids, L, degrees, labels, n_classes, plan_T, meta, meta_cap,
                                           n_i32, n_f32);
    if (rc != SGCN_OK) return rc;
    const int64_t ni = *n_i32 > 0 ? *n_i32 : 1, nf = *n_f32 > 0 ? *n_f32 : 1;
    if (!words || ni + nf > cap_words) return 1;          // too small: grow, then sgcn_sched_packed_copy
    s->impl.packed_copy(static_cast<int32_t*>(words), reinterpret_cast<float*>(static_cast<int32_t*>(words) + ni));
    return SGCN_OK;
}
int64_t sgcn_sched_packed_meta_len(int32_t L) { return sgcn::NeighbourSampler::meta_len(L); }

int sgcn_prefetch_start(sgcn_sched_t* const* samplers, int32_t n_samplers, int32_t n_batches,
                        const int32_t* ids, const int64_t* offsets, int32_t L, const int32_t* degrees,
                        const float* labels, int32_t n_classes, int32_t plan_T, int32_t n_slots,
                        void* const* slot_words, const int64_t* slot_caps, int32_t lag, int32_t n_packers,
                        sgcn_prefetch_t** out) {
    if (!samplers || n_samplers < 1 || n_packers < 0 || (n_packers > 0 && n_samplers != 1) || !out || n_batches < 0 || !offsets || (n_batches > 0 && !ids) || L < 0 ||
        (L > 0 && !degrees) || n_slots < 1 || !slot_words || !slot_caps || lag < 0 || lag + 1 >= n_slots)
        return sgcn::fail(SGCN_ERR_INVALID, "prefetch_start: bad argument");
    for (int32_t i = 0; i < n_samplers; i++)
        if (!samplers[i]) return sgcn::fail(SGCN_ERR_INVALID, "prefetch_start: null sampler");
    try {
        std::unique_ptr<sgcn_prefetch> p(new sgcn_prefetch);
        p->ss.assign(samplers, samplers + n_samplers);
        p->off.assign(offsets, offsets + n_batches + 1);
        p->ids.assign(ids, ids + (n_batches ? offsets[n_batches] : 0));
        p->L = L; p->n_classes = n_classes; p->plan_T = plan_T; p->labels = labels; p->lag = lag;
        p->degrees.assign(degrees, degrees + L);
        p->meta_len = sgcn::NeighbourSampler::meta_len(L);
        p->ready.resize((size_t)n_batches);
        p->is_ready.assign((size_t)n_batches, 0);
        p->ready_flag.reset(new std::atomic<char>[(size_t)std::max(n_batches, 1)]);
        for (int32_t i = 0; i < n_batches; i++) p->ready_flag[(size_t)i].store(0, std::memory_order_relaxed);
        for (int32_t i = 0; i < n_slots; i++) {
            p->slot_words.push_back(slot_words[i]); p->slot_caps.push_back(slot_caps[i]);
            p->free_slots.push_back(i);
        }
        sgcn_prefetch* raw = p.get();
        cpu_set_t node_cpus;
        const bool pin = numa_cpus_of_caller(&node_cpus);      // keep the producers next to the CSR copy
        auto spawn = [&](std::function<void()> body) {
            p->ths.emplace_back([body, pin, node_cpus] {
                if (pin) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &node_cpus);
                body();
            });
        };
        if (n_packers == 0) {
            p->running = n_samplers;
            for (int32_t k = 0; k < n_samplers; k++) spawn([raw, k] { raw->run(k); });
        } else {
            const int32_t nv = samplers[0]->impl.num_vertices();
            for (int32_t j = 0; j < n_packers; j++) p->packers.emplace_back(new sgcn::Packer(nv));
            for (int32_t j = 0; j < 2 * n_packers + 1; j++) {       // hops in flight: one per packer + what the core runs ahead
                p->raw_pool.emplace_back(new sgcn_prefetch::Raw);
                p->raw_free.push_back(p->raw_pool.back().get());
            }
            p->running = n_packers;
            spawn([raw] { raw->run_core(); });
            for (int32_t j = 0; j < n_packers; j++) spawn([raw, j] { raw->run_packer(j); });
        }
        *out = p.release();
    } catch (const std::exception& e) {
        return sgcn::fail(SGCN_ERR_INVALID, "prefetch_start: %s", e.what());
    }
    return SGCN_OK;
}

/* Blocks until the next batch (in order) is ready.  Returns 0 and fills slot / sizes / meta; 1 when
 * the epoch is exhausted; < 0 on a sampler error.  *spill != NULL: the batch outgrew its slot and
 * lives in that heap buffer (n_i + n_f words) until the slot index is released. */
int sgcn_prefetch_next(sgcn_prefetch_t* p, int32_t* slot, int64_t* meta, int64_t* n_i, int64_t* n_f,
                       const void** spill) {
    if (!p || !slot || !meta || !n_i || !n_f || !spill) return sgcn::fail(SGCN_ERR_INVALID, "prefetch_next: bad argument");
    sgcn_prefetch::Ready r;
    {
        const int32_t b0 = p->next_batch;              // only this (the consumer) thread writes it
        if (b0 < (int32_t)p->ready.size())
            sgcn_prefetch::spin_for([&] { return p->ready_flag[(size_t)b0].load(std::memory_order_acquire) != 0; }, 40.0);
        std::unique_lock<std::mutex> lk(p->mu);
        const int32_t b = p->next_batch;
        if (b >= (int32_t)p->ready.size()) return 1;
        p->cv_ready.wait(lk, [&] { return p->is_ready[(size_t)b] || p->error || p->running == 0; });
        if (!p->is_ready[(size_t)b]) {
            if (p->error) return sgcn::fail(p->error, "%s", p->error_msg.c_str());
            return sgcn::fail(SGCN_ERR_INVALID, "prefetch_next: producers exited before batch %d", (int)b);
        }
        r = std::move(p->ready[(size_t)b]);
        p->next_batch = b + 1;
    }
    p->cv_free.notify_all();           // the window moved
    *slot = r.slot; *n_i = r.n_i; *n_f = r.n_f;
    memcpy(meta, r.meta.data(), sizeof(int64_t) * (size_t)p->meta_len);
    *spill = nullptr;
    if (r.spill) {
        std::lock_guard<std::mutex> lk(p->mu);
        if ((size_t)r.slot >= p->spills.size()) p->spills.resize((size_t)r.slot + 1);
        *spill = r.spill->data();
        p->spills[(size_t)r.slot] = std::move(r.spill);
    }
    return SGCN_OK;
}

/* The consumer is done with the slot (its H2D copy has completed): the producer may overwrite it. */
int sgcn_prefetch_release(sgcn_prefetch_t* p, int32_t slot) {
    if (!p || slot < 0 || (size_t)slot >= p->slot_words.size()) return sgcn::fail(SGCN_ERR_INVALID, "prefetch_release: bad slot");
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if ((size_t)slot < p->spills.size()) p->spills[(size_t)slot].reset();
        p->free_slots.push_back(slot);
    }
    p->cv_free.notify_all();
    return SGCN_OK;
}

/* Producer-side seconds so far: out[0] waiting for a free slot, out[1] sampling + packing (pipelined shape: packing,
 * summed over the packers), out[2] copying into the staging slots, out[3] the core thread's sampling (pipelined shape). */
int sgcn_prefetch_stats(sgcn_prefetch_t* p, double* out) {
    if (!p || !out) return sgcn::fail(SGCN_ERR_INVALID, "prefetch_stats: bad argument");
    std::lock_guard<std::mutex> lk(p->mu);
    out[0] = p->t_wait; out[1] = p->t_pack; out[2] = p->t_copy; out[3] = p->t_sample;
    return SGCN_OK;
}

void sgcn_prefetch_stop(sgcn_prefetch_t* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_free.notify_all();
    p->cv_raw.notify_all();
    for (auto& t : p->ths)
        if (t.joinable()) t.join();
    delete p;
}
int sgcn_sched_packed_copy(sgcn_sched_t* s, int32_t* dst_i32, float* dst_f32) {
    if (!s) return sgcn::fail(SGCN_ERR_INVALID, "null sampler");
    s->impl.packed_copy(dst_i32, dst_f32);
    return SGCN_OK;
}

int sgcn_mult_create(const float* prob, int32_t n, sgcn_mult_t** out) {
    if (!out || n < 0) return sgcn::fail(SGCN_ERR_INVALID, "sgcn_mult_create: bad argument");
    if (n == 0) return sgcn::fail(SGCN_ERR_EMPTY_PROB, "Prob is empty");
    *out = new sgcn_mult{sgcn::FenwickMultinomial(prob, n)};
    return SGCN_OK;
}
void sgcn_mult_destroy(sgcn_mult_t* m) { delete m; }
int sgcn_mult_tree(sgcn_mult_t* m, const float** bit, int64_t* len) {
    if (!m || !bit || !len) return sgcn::fail(SGCN_ERR_INVALID, "mult_tree: bad argument");
    *bit = m->impl.tree().data();
    *len = (int64_t)m->impl.tree().size();
    return SGCN_OK;
}
int sgcn_mult_query_u(sgcn_mult_t* m, float u, int32_t* result) {
    if (!m || !result) return sgcn::fail(SGCN_ERR_INVALID, "mult_query_u: bad argument");
    *result = m->impl.descend(u);
    return SGCN_OK;
}
int sgcn_mult_query(sgcn_mult_t* m, int32_t* result) {
    if (!m || !result) return sgcn::fail(SGCN_ERR_INVALID, "mult_query: bad argument");
    *result = m->impl.draw();
    return SGCN_OK;
}

int sgcn_csr_slice_indptr(int32_t n, const int32_t* r, const int32_t* a_p, int32_t* o_p) {
    if (n < 0 || (n > 0 && (!r || !a_p)) || !o_p)
        return sgcn::fail(SGCN_ERR_INVALID, "csr_slice_indptr: bad argument");
    int32_t run = 0;
    for (int32_t i = 0; i < n; i++) {
        o_p[i] = run;
        run += a_p[r[i] + 1] - a_p[r[i]];
    }
    o_p[n] = run;
    return SGCN_OK;
}

}  // extern "C"
