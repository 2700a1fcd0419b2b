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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
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
class NeighbourSampler {
public:
    NeighbourSampler(const float* w, const int32_t* idx, const int32_t* ptr, int32_t n,
                     int32_t nnz, bool cv, bool is)
        : n_(n), cv_(cv), is_(is), nbr_(idx, idx + nnz), wgt_(w, w + nnz), ptr_(ptr, ptr + n),
          slot_(n, -1), fslot_(n, -1), importance_(n, 1.0f) {
        ptr_.push_back(nnz);  // the reference trusts only the first n offsets (scheduler.cpp:16,20)
        if (is_) {
            // column-wise squared weight mass on top of 1e-6   (scheduler.cpp:18,22-25)
            std::fill(importance_.begin(), importance_.end(), 1e-6f);
            for (int32_t r = 0; r < n; r++)
                for (int32_t p = ptr[r]; p < ptr[r + 1]; p++) importance_[idx[p]] += w[p] * w[p];
        }
    }

    void seed(int32_t s) { rng_.reseed((uint32_t)s); }

    void start_batch(int32_t n, const int32_t* ids) { field_.assign(ids, ids + n); }

    int expand(int32_t degree) {
        const size_t n_out = field_.size();
        clear_outputs();
        // output rows are the first |field| entries of the next field   (scheduler.cpp:50-52)
        next_ = field_;
        for (size_t i = 0; i < next_.size(); i++) slot_[next_[i]] = (int32_t)i;
        edg_p_.push_back(0);
        fedg_p_.push_back(0);

        int rc = is_ ? expand_importance(degree) : expand_uniform(degree, n_out);

        field_.swap(next_);
        for (int32_t v : field_) slot_[v] = -1;
        // fslot_ entries are stamped (fbase_ + position): advancing the base un-marks this hop's
        // full-neighbour field without touching its ~43 k scattered table entries again
        fbase_ += (int32_t)ffield_.size();
        if (fbase_ > (1 << 30)) { std::fill(fslot_.begin(), fslot_.end(), -1); fbase_ = 0; }
        transpose_ready_ = false;
        return rc;
    }

    // CSR of A^T (A = last sampled adjacency, n_out x |field|): stable counting sort by column,
    // so within a transposed row entries keep ascending output-row order (deterministic).
    void build_transpose() {
        if (transpose_ready_) return;
        const size_t n_in = field_.size(), ne = edg_t_.size();
        tedg_p_.assign(n_in + 1, 0);
        for (size_t e = 0; e < ne; e++) tedg_p_[edg_t_[e] + 1]++;
        for (size_t c = 0; c < n_in; c++) tedg_p_[c + 1] += tedg_p_[c];
        tedg_t_.resize(ne);
        tedg_w_.resize(ne);
        std::vector<int32_t> cur(tedg_p_.begin(), tedg_p_.end() - 1);
        for (size_t e = 0; e < ne; e++) {
            int32_t q = cur[edg_t_[e]]++;
            tedg_t_[q] = edg_s_[e];
            tedg_w_[q] = edg_w_[e];
        }
        transpose_ready_ = true;
    }

    const std::vector<int32_t>* ivec(int which) {
        switch (which) {
            case SGCN_SCHED_FIELD: return &field_;
            case SGCN_SCHED_FFIELD: return &ffield_;
            case SGCN_SCHED_EDG_S: return &edg_s_;
            case SGCN_SCHED_EDG_T: return &edg_t_;
            case SGCN_SCHED_FEDG_S: return &fedg_s_;
            case SGCN_SCHED_FEDG_T: return &fedg_t_;
            case SGCN_SCHED_EDG_P: return &edg_p_;
            case SGCN_SCHED_FEDG_P: return &fedg_p_;
            case SGCN_SCHED_ADJ_I: return &nbr_;
            case SGCN_SCHED_TEDG_P: build_transpose(); return &tedg_p_;
            case SGCN_SCHED_TEDG_T: build_transpose(); return &tedg_t_;
            default: return nullptr;
        }
    }
    const std::vector<float>* fvec(int which) {
        switch (which) {
            case SGCN_SCHED_SCALES: return &scales_;
            case SGCN_SCHED_EDG_W: return &edg_w_;
            case SGCN_SCHED_MEDG_W: return &medg_w_;
            case SGCN_SCHED_FEDG_W: return &fedg_w_;
            case SGCN_SCHED_ADJ_W: return &wgt_;
            case SGCN_SCHED_TEDG_W: build_transpose(); return &tedg_w_;
            default: return nullptr;
        }
    }

    // ---- packed minibatch (one call = PyScheduler.batch, gcn/_scheduler.pyx:55-127) ----------
    // Runs the L expansions and lays every array the device step needs into two growable
    // staging vectors (int32 / fp32), each sub-array 16-byte aligned: fields, ffields, scales,
    // labels[fields[-1]], and per layer the CSR of adj, adj^T and fadj with their row plans.
    // `meta` receives (offset, length) descriptors, layer 0 = input-most layer (the reversal of
    // gcn/_scheduler.pyx:121-126).  No Python objects are touched: the call runs without the GIL.
    enum { kCsrDesc = 11 };   // nrows ncols nnz rowptr col val seg nseg fix nfix nslots
    int pack_batch(int32_t n, const int32_t* ids, int32_t L, const int32_t* degrees,
                   const float* labels, int32_t n_classes, int32_t plan_T, int64_t* meta,
                   int64_t meta_cap, int64_t* n_i32, int64_t* n_f32) {
        const int64_t need = meta_len(L);
        if (meta_cap < need) return fail(SGCN_ERR_INVALID, "batch_packed: meta too small (%lld < %lld)",
                                         (long long)meta_cap, (long long)need);
        pi_.clear(); pf_.clear();
        std::fill(meta, meta + need, 0);
        struct CooOff { bool& f; bool old; ~CooOff() { f = old; } } coo_off{want_coo_, want_coo_};
        want_coo_ = false;                               // the packed layout is CSR only
        meta[0] = L; meta[1] = cv_ ? 1 : 0; meta[2] = n_classes;
        start_batch(n, ids);
        int64_t* fields_d = meta + 4;                    // (L+1) x (off,len)
        int64_t* scales_d = fields_d + 2 * (L + 1);      // L x (off,len)
        int64_t* ffields_d = scales_d + 2 * L;           // L x (off,len)
        int64_t* labels_d = ffields_d + 2 * L;           // off, rows, cols
        int64_t* medg_d = labels_d + 3;                  // L x (off,len)
        int64_t* csr_d = medg_d + 2 * L;                 // L x 3 x kCsrDesc  (adj, adjT, fadj)
        put_i(field_, fields_d + 2 * L);                 // fields[L] = the batch itself
        int rc = SGCN_OK;
        for (int32_t l = 0; l < L; l++) {
            const int32_t slot = L - 1 - l;              // position after the reversal
            const int r = expand(degrees[L - l - 1]);    // gcn/_scheduler.pyx:66
            if (r != SGCN_OK) rc = r;
            const int32_t n1 = (int32_t)edg_p_.size() - 1, n0 = (int32_t)field_.size();
            put_i(field_, fields_d + 2 * slot);
            put_f(scales_, scales_d + 2 * slot);
            put_f(medg_w_, medg_d + 2 * slot);
            build_transpose();
            put_csr(csr_d + (3 * slot + 0) * kCsrDesc, n1, n0, edg_p_, edg_t_, edg_w_, plan_T);
            put_csr(csr_d + (3 * slot + 1) * kCsrDesc, n0, n1, tedg_p_, tedg_t_, tedg_w_, plan_T);
            if (cv_) {
                put_i(ffield_, ffields_d + 2 * slot);
                put_csr(csr_d + (3 * slot + 2) * kCsrDesc, n1, (int32_t)ffield_.size(), fedg_p_,
                        fedg_t_, fedg_w_, plan_T);
            }
        }
        if (labels && n_classes > 0) {                   // labels[fields[-1]]  (_scheduler.pyx:138)
            labels_d[0] = (int64_t)pf_.size(); labels_d[1] = n; labels_d[2] = n_classes;
            for (int32_t i = 0; i < n; i++) {
                const float* src = labels + (int64_t)ids[i] * n_classes;
                pf_.insert(pf_.end(), src, src + n_classes);
            }
            pad_f();
        }
        *n_i32 = (int64_t)pi_.size();
        *n_f32 = (int64_t)pf_.size();
        return rc;
    }
    static int64_t meta_len(int32_t L) { return 4 + 2 * (L + 1) + 2 * L + 2 * L + 3 + 2 * L + 3 * L * kCsrDesc; }
    void packed_copy(int32_t* di, float* df) const {
        if (di && !pi_.empty()) memcpy(di, pi_.data(), pi_.size() * sizeof(int32_t));
        if (df && !pf_.empty()) memcpy(df, pf_.data(), pf_.size() * sizeof(float));
    }

private:
    void clear_outputs() {
        ffield_.clear(); scales_.clear();
        edg_s_.clear(); edg_t_.clear(); edg_w_.clear(); medg_w_.clear(); edg_p_.clear();
        fedg_s_.clear(); fedg_t_.clear(); fedg_w_.clear(); fedg_p_.clear();
    }
    // position of vertex v in the growing next field (appending it on first sight)
    int32_t place(int32_t v) {
        int32_t& s = slot_[v];
        if (s < 0) {
            s = (int32_t)next_.size();
            next_.push_back(v);
        }
        return s;
    }
    int32_t fplace(int32_t v) {
        int32_t& s = fslot_[v];
        if (s < fbase_) {
            s = fbase_ + (int32_t)ffield_.size();
            ffield_.push_back(v);
        }
        return s - fbase_;
    }

#if defined(__x86_64__)
    // fplace() over a whole neighbour list, 16 entries per step (AVX-512: gather the table entries, give the
    // unseen vertices consecutive positions in list order with an expand, scatter the new stamps, append the
    // new vertices with a compress-store).  Same first-seen numbering as the scalar loop: the entries of a
    // vector are applied "at once", which differs from one-by-one only if a vertex occurs twice among the
    // unseen lanes -- detected with vpconflictd and handed to the scalar loop.  This loop was 2/3 of the
    // sampler's time (50 k table lookups per Reddit batch).
    __attribute__((target("avx512f,avx512cd")))
    void fplace_row_avx512(const int32_t* cols, int32_t deg, int32_t* ft) {
        const size_t old = ffield_.size();
        ffield_.resize(old + (size_t)deg);                 // room for the worst case; trimmed below
        int32_t* fnew = ffield_.data();
        int32_t next = (int32_t)old;
        const __m512i base = _mm512_set1_epi32(fbase_);
        const __m512i iota = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        for (int32_t k = 0; k < deg; k += 16) {
            const __mmask16 m = deg - k >= 16 ? (__mmask16)0xffff : (__mmask16)((1u << (deg - k)) - 1u);
            const __m512i v = _mm512_maskz_loadu_epi32(m, cols + k);
            __m512i sl = _mm512_mask_i32gather_epi32(_mm512_set1_epi32(-1), m, v, fslot_.data(), 4);
            const __mmask16 fresh = _mm512_mask_cmplt_epi32_mask(m, sl, base);
            if (fresh) {
                const __m512i conf = _mm512_conflict_epi32(_mm512_mask_mov_epi32(_mm512_sub_epi32(_mm512_setzero_si512(), _mm512_add_epi32(iota, _mm512_set1_epi32(1))), fresh, v));
                if (_mm512_mask_test_epi32_mask(fresh, conf, conf)) {      // a repeated unseen vertex: one by one
                    ffield_.resize((size_t)next);
                    for (int32_t q = k; q < std::min(deg, k + 16); q++) ft[q] = fplace(cols[q]);
                    next = (int32_t)ffield_.size();
                    ffield_.resize(old + (size_t)deg);
                    fnew = ffield_.data();
                    continue;
                }
                const __m512i ids = _mm512_maskz_expand_epi32(fresh, _mm512_add_epi32(iota, _mm512_set1_epi32(next)));
                const __m512i stamped = _mm512_add_epi32(ids, base);
                _mm512_mask_i32scatter_epi32(fslot_.data(), fresh, v, stamped, 4);
                _mm512_mask_compressstoreu_epi32(fnew + next, fresh, v);
                next += __builtin_popcount((unsigned)fresh);
                sl = _mm512_mask_mov_epi32(sl, fresh, stamped);
            }
            _mm512_mask_storeu_epi32(ft + k, m, _mm512_sub_epi32(sl, base));
        }
        ffield_.resize((size_t)next);
    }
#else
    void fplace_row_avx512(const int32_t* cols, int32_t deg, int32_t* ft) {          // never taken: avx512_ is false
        for (int32_t k = 0; k < deg; k++) ft[k] = fplace(cols[k]);
    }
#endif

    // Uniform sampling w/o replacement, optional control-variate extras (scheduler.cpp:125-180)
    int expand_uniform(int32_t degree, size_t n_out) {
        // The batch's rows are scattered over a CSR of hundreds of MB (shuffled train ids): every row
        // starts with two DRAM misses (neighbour ids, weights).  The ids are known up front, so the rows a
        // few iterations ahead are prefetched -- the row pointer first (it is needed to form the address),
        // then the first lines of both arrays; the hardware prefetcher follows the rest of a row.
        constexpr size_t kAheadPtr = 16, kAhead = 8;
        for (size_t i = 0; i < n_out; i++) {
            if (i + kAheadPtr < n_out) __builtin_prefetch(ptr_.data() + field_[i + kAheadPtr], 0, 1);
            if (i + kAhead < n_out) {
                const int32_t pv = ptr_[field_[i + kAhead]];
                const int32_t pdeg = ptr_[field_[i + kAhead] + 1] - pv;
                const char* c0 = reinterpret_cast<const char*>(nbr_.data() + pv);
                const char* w0 = reinterpret_cast<const char*>(wgt_.data() + pv);
                const int lines = std::min(4, (pdeg * 4 + 63) / 64);
                for (int q = 0; q < lines; q++) { __builtin_prefetch(c0 + 64 * q, 1, 1); __builtin_prefetch(w0 + 64 * q, 1, 1); }
            }
            const int32_t v = field_[i];
            int32_t* cols = nbr_.data() + ptr_[v];
            float* vals = wgt_.data() + ptr_[v];
            const int32_t deg = ptr_[v + 1] - ptr_[v];
            const int32_t take = std::min(deg, degree);
            // amplification deg/take in fp32; isolated rows amplify by 1  (scheduler.cpp:132-133)
            const float amp = deg == 0 ? 1.0f : (float)deg / (float)take;
            // 1/sqrt(amp): fp32 sqrt, fp64 reciprocal, stored fp32        (scheduler.cpp:134)
            scales_.push_back((float)(1.0 / (double)std::sqrt(amp)));

            for (int32_t k = 0; k < take; k++) {
                // pick a position in [k, deg) and move it to the front segment; the product
                // and the sum are separate fp32 roundings (no FMA: built with -ffp-contract=off)
                const float span = (float)(deg - k) * rng_.u01();
                const float posf = (float)k + span;
                const int32_t j = std::min((int32_t)posf, deg - 1);
                std::swap(cols[k], cols[j]);
                std::swap(vals[k], vals[j]);
                const float w = vals[k] * amp;
                edg_s_.push_back((int32_t)i);
                edg_t_.push_back(place(cols[k]));
                edg_w_.push_back(w);
                if (cv_) medg_w_.push_back(vals[k] * w);
            }
            edg_p_.push_back((int32_t)edg_t_.size());

            if (cv_) {
                // every neighbour, in the row's current (post-swap) order   (scheduler.cpp:167-179);
                // bulk-grown and filled through raw pointers: this loop is half of the sampler's time
                const size_t base = fedg_t_.size();
                fedg_t_.resize(base + (size_t)deg);
                fedg_w_.resize(base + (size_t)deg);
                int32_t* ft = fedg_t_.data() + base;
                if (avx512_) fplace_row_avx512(cols, deg, ft);
                else for (int32_t k = 0; k < deg; k++) ft[k] = fplace(cols[k]);
                if (deg) memcpy(fedg_w_.data() + base, vals, (size_t)deg * sizeof(float));
                if (want_coo_) fedg_s_.insert(fedg_s_.end(), (size_t)deg, (int32_t)i);
                fedg_p_.push_back((int32_t)fedg_t_.size());
            }
        }
        if (!cv_) fedg_p_.assign(n_out + 1, 0);
        return SGCN_OK;
    }

    // Importance sampling of the joint neighbourhood (scheduler.cpp:63-123)
    int expand_importance(int32_t degree) {
        const size_t n_out = field_.size();
        std::vector<int32_t> cand;
        std::vector<float> prob;
        std::vector<char> seen((size_t)n_, 0);
        std::vector<int32_t> hits((size_t)n_, 0);
        float mass = 0.f;
        for (int32_t v : field_)
            for (int32_t p = ptr_[v]; p < ptr_[v + 1]; p++) {
                const int32_t t = nbr_[p];
                if (!seen[t]) {
                    seen[t] = 1;
                    cand.push_back(t);
                    mass += importance_[t];
                    prob.push_back(importance_[t]);
                }
            }
        if (prob.empty()) {
            edg_p_.assign(n_out + 1, 0);
            fedg_p_.assign(n_out + 1, 0);
            return fail(SGCN_ERR_EMPTY_PROB, "Prob is empty");
        }
        FenwickMultinomial mult(prob.data(), (int)prob.size());
        const int32_t draws = (int32_t)std::min(n_out * (size_t)degree, cand.size());
        for (int32_t k = 0; k < draws; k++) {
            const int32_t t = cand[mult.draw()];
            hits[t]++;
            place(t);
        }
        int rc = SGCN_OK;
        for (size_t i = 0; i < n_out; i++) {
            const int32_t v = field_[i];
            for (int32_t p = ptr_[v]; p < ptr_[v + 1]; p++) {
                const int32_t t = nbr_[p];
                if (!hits[t]) continue;
                // ((hits*w)*mass) / (importance*draws), all fp32, left to right (scheduler.cpp:107-108)
                const float num = ((float)hits[t] * wgt_[p]) * mass;
                const float den = importance_[t] * (float)draws;
                const float w = num / den;
                edg_s_.push_back((int32_t)i);
                edg_t_.push_back(slot_[t]);
                edg_w_.push_back(w);
                if (std::isnan(w)) rc = fail(SGCN_ERR_NAN, "nan");
            }
            edg_p_.push_back((int32_t)edg_t_.size());
        }
        fedg_p_.assign(n_out + 1, 0);
        return rc;
    }

    void pad_i() { while (pi_.size() & 3) pi_.push_back(0); }
    void pad_f() { while (pf_.size() & 3) pf_.push_back(0.f); }
    void put_i(const std::vector<int32_t>& v, int64_t* d) {
        d[0] = (int64_t)pi_.size(); d[1] = (int64_t)v.size();
        pi_.insert(pi_.end(), v.begin(), v.end());
        pad_i();
    }
    void put_f(const std::vector<float>& v, int64_t* d) {
        d[0] = (int64_t)pf_.size(); d[1] = (int64_t)v.size();
        pf_.insert(pf_.end(), v.begin(), v.end());
        pad_f();
    }
    void put_csr(int64_t* d, int32_t nrows, int32_t ncols, const std::vector<int32_t>& rowptr,
                 const std::vector<int32_t>& col, const std::vector<float>& val, int32_t plan_T) {
        int64_t tmp[2];
        d[0] = nrows; d[1] = ncols; d[2] = (int64_t)col.size();
        put_i(rowptr, tmp); d[3] = tmp[0];
        put_i(col, tmp); d[4] = tmp[0];
        put_f(val, tmp); d[5] = tmp[0];
        int64_t nslots = 0;
        plan_build(rowptr.data(), nrows, plan_T, seg_, fix_, nslots);
        put_i(seg_, tmp); d[6] = tmp[0]; d[7] = (int64_t)seg_.size() / 4;
        put_i(fix_, tmp); d[8] = tmp[0]; d[9] = (int64_t)fix_.size() / 3;
        d[10] = nslots;
    }

    std::vector<int32_t> pi_, seg_, fix_;
    std::vector<float> pf_;
    int32_t n_;
    bool cv_, is_, transpose_ready_ = false;
    bool want_coo_ = true;     // the COO source-row array of the full edges (only the view API needs it)
    std::vector<int32_t> nbr_;   // private, permuted in place
    std::vector<float> wgt_;
    std::vector<int32_t> ptr_;
    std::vector<int32_t> slot_, fslot_;
    int32_t fbase_ = 0;                    // stamp base of fslot_: an entry >= fbase_ is a position in this hop's ffield_
#if defined(__x86_64__)
    const bool avx512_ = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512cd") && !getenv("SGCN_NO_AVX512");
#else
    const bool avx512_ = false;
#endif
    std::vector<float> importance_;
    std::vector<int32_t> field_, next_, ffield_;
    std::vector<float> scales_;
    std::vector<int32_t> edg_s_, edg_t_, edg_p_, fedg_s_, fedg_t_, fedg_p_;
    std::vector<float> edg_w_, medg_w_, fedg_w_;
    std::vector<int32_t> tedg_p_, tedg_t_;
    std::vector<float> tedg_w_;
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

// ---- native prefetch thread --------------------------------------------------------------------
// The sampler of an epoch as a C++ thread: it walks the epoch's id slices in order (so the sample
// sequence is the synchronous loop's, bit for bit), packs each minibatch and copies it into the
// next free pinned staging slot.  No Python runs on this thread: a Python producer thread has to
// re-take the interpreter lock after every foreign call, and against a launching thread that
// releases and re-takes it every ~20 us that costs ~0.25 ms per batch (measured: 0.5 ms per batch
// next to 0.26 ms alone) -- enough to make the sampler the bottleneck of the epoch.
struct sgcn_prefetch {
    struct Ready { int32_t batch = 0, slot = 0; int64_t n_i = 0, n_f = 0; std::vector<int64_t> meta; std::unique_ptr<std::vector<int32_t>> spill; };
    std::vector<sgcn_sched*> ss;                   // one sampler per producer thread
    std::vector<int32_t> ids; std::vector<int64_t> off;
    int32_t L = 0, n_classes = 0, plan_T = 0, lag = 2;
    std::vector<int32_t> degrees; const float* labels = nullptr;
    std::vector<void*> slot_words; std::vector<int64_t> slot_caps;
    int64_t meta_len = 0;
    std::mutex mu; std::condition_variable cv_free, cv_ready;
    std::deque<int32_t> free_slots;
    std::vector<Ready> ready; std::vector<char> is_ready;       // indexed by batch
    int32_t next_batch = 0;                                      // the batch the consumer takes next
    std::vector<std::unique_ptr<std::vector<int32_t>>> spills;   // batches that outgrew their slot
    bool stop = false; int32_t running = 0; int error = 0; std::string error_msg;
    double t_wait = 0, t_pack = 0, t_copy = 0;       // producer seconds: waiting for a slot / packing / copying
    std::vector<std::thread> ths;

    // Thread k builds batches k, k + N, k + 2N, ...  A batch may take a slot only inside the window
    // of batches the consumer will ask for next: with `window` = slots the consumer never holds,
    // the batches in the window can all be staged at once, so the one the consumer is waiting for
    // can never be starved of a slot by later ones.
    void run(int32_t k) {
        const int32_t nb = (int32_t)off.size() - 1, N = (int32_t)ss.size();
        const int32_t window = std::max<int32_t>(1, (int32_t)slot_words.size() - lag - 1);
        using clk = std::chrono::steady_clock;
        for (int32_t b = k; b < nb; b += N) {
            int32_t slot;
            const auto c0 = clk::now();
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_free.wait(lk, [&] { return stop || error || (b < next_batch + window && !free_slots.empty()); });
                if (stop || error) break;
                slot = free_slots.front(); free_slots.pop_front();
            }
            const auto c1 = clk::now();
            Ready r; r.batch = b; r.slot = slot; r.meta.assign((size_t)meta_len, 0);
            const int rc = ss[(size_t)k]->impl.pack_batch((int32_t)(off[b + 1] - off[b]), ids.data() + off[b], L,
                                                          degrees.data(), labels, n_classes, plan_T, r.meta.data(),
                                                          meta_len, &r.n_i, &r.n_f);
            if (rc != SGCN_OK) {
                std::lock_guard<std::mutex> lk(mu);
                error = rc; error_msg = sgcn::error_slot();
                break;
            }
            const int64_t ni = std::max<int64_t>(r.n_i, 1), nf = std::max<int64_t>(r.n_f, 1);
            int32_t* dst;
            if (ni + nf <= slot_caps[(size_t)slot]) dst = static_cast<int32_t*>(slot_words[(size_t)slot]);
            else {                       // rare: the batch outgrew the slot -> heap buffer, slot stays unused
                r.spill.reset(new std::vector<int32_t>((size_t)(ni + nf)));
                dst = r.spill->data();
            }
            const auto c2 = clk::now();
            ss[(size_t)k]->impl.packed_copy(dst, reinterpret_cast<float*>(dst + ni));
            const auto c3 = clk::now();
            {
                std::lock_guard<std::mutex> lk(mu);
                ready[(size_t)b] = std::move(r);
                is_ready[(size_t)b] = 1;
                t_wait += std::chrono::duration<double>(c1 - c0).count();
                t_pack += std::chrono::duration<double>(c2 - c1).count();
                t_copy += std::chrono::duration<double>(c3 - c2).count();
            }
            cv_ready.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        running--;
        cv_ready.notify_all();
        cv_free.notify_all();
    }
};

struct sgcn_mult { sgcn::FenwickMultinomial impl; };

extern "C" {

const char* sgcn_last_error(void) { return sgcn::error_slot(); }
int sgcn_abi_version(void) { return 6; }

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
    const std::vector<int32_t>* v = s ? s->impl.ivec(which) : nullptr;
    if (!v || !ptr || !len) return sgcn::fail(SGCN_ERR_INVALID, "view_i32: bad selector %d", which);
    *ptr = v->data();
    *len = (int64_t)v->size();
    return SGCN_OK;
}
int sgcn_sched_view_f32(sgcn_sched_t* s, int32_t which, const float** ptr, int64_t* len) {
    const std::vector<float>* v = s ? s->impl.fvec(which) : nullptr;
    if (!v || !ptr || !len) return sgcn::fail(SGCN_ERR_INVALID, "view_f32: bad selector %d", which);
    *ptr = v->data();
    *len = (int64_t)v->size();
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
                        void* const* slot_words, const int64_t* slot_caps, int32_t lag,
                        sgcn_prefetch_t** out) {
    if (!samplers || n_samplers < 1 || !out || n_batches < 0 || !offsets || (n_batches > 0 && !ids) || L < 0 ||
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
        for (int32_t i = 0; i < n_slots; i++) {
            p->slot_words.push_back(slot_words[i]); p->slot_caps.push_back(slot_caps[i]);
            p->free_slots.push_back(i);
        }
        sgcn_prefetch* raw = p.get();
        p->running = n_samplers;
        cpu_set_t node_cpus;
        const bool pin = numa_cpus_of_caller(&node_cpus);      // keep the producers next to the CSR copy
        for (int32_t k = 0; k < n_samplers; k++)
            p->ths.emplace_back([raw, k, pin, node_cpus] {
                if (pin) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &node_cpus);
                raw->run(k);
            });
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

/* Producer-side seconds so far: out[0] waiting for a free slot, out[1] sampling + packing,
 * out[2] copying into the staging slots. */
int sgcn_prefetch_stats(sgcn_prefetch_t* p, double* out) {
    if (!p || !out) return sgcn::fail(SGCN_ERR_INVALID, "prefetch_stats: bad argument");
    std::lock_guard<std::mutex> lk(p->mu);
    out[0] = p->t_wait; out[1] = p->t_pack; out[2] = p->t_copy;
    return SGCN_OK;
}

void sgcn_prefetch_stop(sgcn_prefetch_t* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_free.notify_all();
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
