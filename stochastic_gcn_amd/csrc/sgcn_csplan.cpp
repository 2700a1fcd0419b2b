// Host-side column-sweep plan (include/sgcn.h, sgcn_csplan_t): virtual rows -> degree-sorted
// tiles -> per-tile column-sorted merged nonzero lists.
//
// Round 6: the plan is built in ONE pass on all the cores the process may use (sgcn_csplan_build).  The reference
// runs the product this plan serves once per matrix (gcn/utils.py:321-322), so the plan's build time is user-visible
// time: 2.6 s single-threaded for S-Reddit (two passes: count, then fill) became one parallel pass.  Only the
// longest-processing-time dealing of the virtual rows is serial (a heap over the bins); a tile's entries are the
// stable MERGE of its rows' column-sorted runs (the CSR's rows are sorted -- checked, and sorted into a private copy
// if they are not), tiles are claimed by the threads a few at a time and written to per-thread arenas, and the export
// copies them out in tile order.  The plan does not depend on the thread count (tests/test_csplan.py).
#include "sgcn_host.h"
#include "../../include/sgcn.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <new>
#include <numeric>
#include <queue>
#include <thread>
#include <vector>

#include <sched.h>

namespace {

struct VRow { int32_t row, piece, npieces, nnz; };

// ---- threads ------------------------------------------------------------------------------------------------
// The cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container with 16 of
// a box's 256 cores must not start 256 threads), by SGCN_PLAN_THREADS and by 64.
int plan_threads(int32_t requested) {
    if (requested > 0) return std::min<int32_t>(requested, 256);
    if (const char* e = getenv("SGCN_PLAN_THREADS")) {
        const int v = atoi(e);
        if (v > 0) return std::min(v, 256);
    }
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long long period = 0;
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long long quota = atoll(q);
            if (quota > 0) n = (int)std::min<long long>(n, std::max<long long>(1, quota / period));
        }
        fclose(f);
    }
    return std::max(1, std::min(n, 64));
}

// fn(tid) on nthr threads (the caller is thread 0); an exception in any of them (std::bad_alloc) becomes `false`
template <class F>
bool run_threads(int nthr, F fn) {
    std::atomic<int> bad{0};
    auto guarded = [&](int tid) {
        try { fn(tid); } catch (...) { bad.store(1); }
    };
    if (nthr <= 1) { guarded(0); return !bad.load(); }
    std::vector<std::thread> pool;
    pool.reserve((size_t)nthr - 1);
    try {
        for (int t = 1; t < nthr; t++) pool.emplace_back(guarded, t);
    } catch (...) { bad.store(1); }
    guarded(0);
    for (auto& th : pool) th.join();
    return !bad.load();
}

// items [0, n) claimed `chunk` at a time: fn(begin, end, tid)
template <class F>
bool parallel_chunks(int64_t n, int64_t chunk, int nthr, F fn) {
    if (n <= 0) return true;
    chunk = std::max<int64_t>(1, chunk);
    nthr = (int)std::max<int64_t>(1, std::min<int64_t>(nthr, (n + chunk - 1) / chunk));
    std::atomic<int64_t> next{0};
    return run_threads(nthr, [&](int tid) {
        for (;;) {
            const int64_t b = next.fetch_add(chunk);
            if (b >= n) break;
            fn(b, std::min(n, b + chunk), tid);
        }
    });
}

// ---- layout: virtual rows dealt to bins ---------------------------------------------------------------------
// vertices bucketed by group label (stable: ascending vertex id inside a group); no labels = one group
void bucket_rows(const int32_t* row_group, int32_t M, std::vector<int32_t>& order, std::vector<int64_t>& gptr) {
    order.resize((size_t)M);
    if (!row_group) {
        std::iota(order.begin(), order.end(), 0);
        gptr = {0, (int64_t)M};
        return;
    }
    int32_t ng = 0;
    for (int32_t r = 0; r < M; r++) ng = std::max(ng, row_group[r] + 1);
    gptr.assign((size_t)ng + 1, 0);
    for (int32_t r = 0; r < M; r++) gptr[(size_t)row_group[r] + 1]++;
    for (int32_t g = 0; g < ng; g++) gptr[(size_t)g + 1] += gptr[g];
    std::vector<int64_t> fill(gptr.begin(), gptr.end() - 1);
    for (int32_t r = 0; r < M; r++) order[(size_t)fill[row_group[r]]++] = r;
}

// default split threshold: 8 x the mean degree -- half the load of a 16-row bin, so that no single (virtual) row
// dominates its bin -- within [64, 1024].  (Round 5, S-Reddit at d = 602, sustained: T = 400 3.13 - 3.17 ms, 800 3.12, 1,600
// 3.10 - 3.15, 2,000 3.40: fewer split rows mean fewer workspace slots and a shorter fix-up launch until a row outweighs
// its bin; profiles/r45_headline_T_probe.jsonl.)
inline int32_t default_t(const int32_t* rowptr, int32_t M) {
    const int64_t avg = M > 0 ? ((int64_t)rowptr[M] - rowptr[0]) / M : 0;
    return (int32_t)std::min<int64_t>(1024, std::max<int64_t>(64, 8 * avg));
}

// virtual rows: a row with n <= T nonzeros is one; a longer row becomes ceil(n/T) strided pieces.  Appended to v
// sorted by weight (heaviest first) for the LPT dealing; ties keep row order (deterministic) -- a counting sort, the
// weights being bounded by the longest unsplit row.
void make_vrows(const int32_t* rowptr, const int32_t* rows, int64_t nrows, int32_t T, std::vector<VRow>& v) {
    std::vector<VRow> tmp;
    tmp.reserve((size_t)nrows + 64);
    int32_t wmax = 0;
    for (int64_t ri = 0; ri < nrows; ri++) {
        const int32_t r = rows[ri];
        const int32_t n = rowptr[r + 1] - rowptr[r];
        if (n <= T) { tmp.push_back({r, 0, 1, n}); wmax = std::max(wmax, n); continue; }
        const int32_t c = (n + T - 1) / T;
        for (int32_t q = 0; q < c; q++) {
            const int32_t w = (n - q + c - 1) / c;
            tmp.push_back({r, q, c, w});
            wmax = std::max(wmax, w);
        }
    }
    std::vector<int64_t> start((size_t)wmax + 2, 0);
    for (const VRow& x : tmp) start[(size_t)(wmax - x.nnz) + 1]++;
    for (size_t i = 1; i < start.size(); i++) start[i] += start[i - 1];
    const size_t base = v.size();
    v.resize(base + tmp.size());
    for (const VRow& x : tmp) v[base + (size_t)start[(size_t)(wmax - x.nnz)]++] = x;
}

// Longest-processing-time dealing: the next heaviest virtual row goes to the lightest bin that
// still has a free slot, so every bin carries (nearly) the same number of nonzeros -- a paced
// sweep lasts as long as its heaviest tile.  assign[slot] = vbase + index into v (or -1).
void deal(const VRow* v, int64_t nv, int64_t vbase, int32_t R, int64_t nbins, int64_t* assign) {
    std::fill(assign, assign + nbins * R, (int64_t)-1);
    std::vector<int32_t> fill((size_t)nbins, 0);
    typedef std::pair<int64_t, int64_t> WT;          // (weight, bin); min-heap, ties -> low bin id
    std::priority_queue<WT, std::vector<WT>, std::greater<WT>> heap;
    for (int64_t t = 0; t < nbins; t++) heap.push({0, t});
    for (int64_t i = 0; i < nv; i++) {
        WT top = heap.top();
        heap.pop();
        const int64_t t = top.second;
        assign[t * R + fill[t]++] = vbase + i;
        if (fill[t] < R) heap.push({top.first + v[i].nnz, t});
    }
}

struct Layout {
    int32_t R = 16;                   // virtual rows per bin
    int32_t NG = 1;                   // bins per tile (lane groups per wavefront)
    std::vector<VRow> v;
    std::vector<int64_t> assign;      // [ntiles * NG * R] -> index into v, or -1
    int64_t ntiles = 0;
};

// one lane group: tiles never straddle row groups, groups in label order
void layout_g1(const int32_t* rowptr, int32_t M, int32_t R, int32_t T, const int32_t* row_group, Layout& L) {
    std::vector<int32_t> order;
    std::vector<int64_t> gptr;
    bucket_rows(row_group, M, order, gptr);
    L.R = R; L.NG = 1; L.ntiles = 0;
    std::vector<int64_t> first_tile, vbase;
    for (size_t g = 0; g + 1 < gptr.size(); g++) {
        vbase.push_back((int64_t)L.v.size());
        make_vrows(rowptr, order.data() + gptr[g], gptr[g + 1] - gptr[g], T, L.v);
        first_tile.push_back(L.ntiles);
        L.ntiles += ((int64_t)L.v.size() - vbase.back() + R - 1) / R;
    }
    L.assign.resize((size_t)L.ntiles * R);
    for (size_t g = 0; g + 1 < gptr.size(); g++) {
        const int64_t nv = (g + 2 < gptr.size() ? vbase[g + 1] : (int64_t)L.v.size()) - vbase[g];
        const int64_t nt = (nv + R - 1) / R;
        deal(L.v.data() + vbase[g], nv, vbase[g], R, nt, L.assign.data() + first_tile[g] * R);
    }
}

// NG = 2 / 4 lane groups: ungrouped, the tile count rounded up to whole launches of `round_tiles` resident waves
constexpr int kG2R = 16;
void layout_gn(const int32_t* rowptr, int32_t M, int32_t T, int32_t round_tiles, int NG, Layout& L) {
    std::vector<int32_t> rows((size_t)M);
    std::iota(rows.begin(), rows.end(), 0);
    make_vrows(rowptr, rows.data(), M, T, L.v);
    const int64_t nv = (int64_t)L.v.size();
    int64_t nt = (nv + NG * kG2R - 1) / (NG * kG2R);
    if (round_tiles > 0 && nt > round_tiles / 2)                       // whole launches (a small matrix just gets enough tiles)
        nt = (nt + round_tiles - 1) / round_tiles * round_tiles;
    L.R = kG2R; L.NG = NG; L.ntiles = nt;
    L.assign.resize((size_t)nt * NG * kG2R);
    deal(L.v.data(), nv, 0, kG2R, nt * NG, L.assign.data());
}

// ---- the matrix: rows column-sorted (or a sorted private copy) ------------------------------------------------
struct Csr {
    const int32_t* rowptr = nullptr;
    const int32_t* col = nullptr;
    const float* val = nullptr;               // nullable (count only): values read as 0
    std::vector<int32_t> own_col;
    std::vector<float> own_val;
};

// 0 = fine, 1 = a column outside [0, 2^bits), 2 = out of memory
int prepare_csr(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int bits, int nthr, Csr& A) {
    A.rowptr = rowptr; A.col = col; A.val = val;
    std::atomic<int> unsorted{0}, bad{0};
    const int64_t lim = (int64_t)1 << bits;
    if (!parallel_chunks(M, 2048, nthr, [&](int64_t r0, int64_t r1, int) {
            int u = 0, b = 0;
            for (int64_t r = r0; r < r1; r++)
                for (int32_t p = rowptr[r]; p < rowptr[r + 1]; p++) {
                    b |= (col[p] < 0) | ((int64_t)col[p] >= lim);
                    if (p > rowptr[r]) u |= col[p] < col[p - 1];
                }
            if (u) unsorted.store(1);
            if (b) bad.store(1);
        })) return 2;
    if (bad.load()) return 1;
    if (!unsorted.load()) return 0;
    // a CSR whose rows are not column-sorted: a stably sorted private copy (what the per-tile merge assumes)
    const int64_t nnz = (int64_t)rowptr[M] - rowptr[0];
    const int64_t p0 = rowptr[0];
    try {
        A.own_col.resize((size_t)nnz);
        if (val) A.own_val.resize((size_t)nnz);
    } catch (...) { return 2; }
    if (!parallel_chunks(M, 512, nthr, [&](int64_t r0, int64_t r1, int) {
            std::vector<std::pair<int32_t, float>> buf;
            for (int64_t r = r0; r < r1; r++) {
                buf.clear();
                for (int32_t p = rowptr[r]; p < rowptr[r + 1]; p++) buf.push_back({col[p], val ? val[p] : 0.f});
                std::stable_sort(buf.begin(), buf.end(),
                                 [](const std::pair<int32_t, float>& a, const std::pair<int32_t, float>& b) { return a.first < b.first; });
                for (size_t i = 0; i < buf.size(); i++) {
                    A.own_col[(size_t)(rowptr[r] - p0) + i] = buf[i].first;
                    if (val) A.own_val[(size_t)(rowptr[r] - p0) + i] = buf[i].second;
                }
            }
        })) return 2;
    A.col = A.own_col.data() - p0;
    if (val) A.val = A.own_val.data() - p0;
    return 0;
}

// ---- a bin's entries: the stable merge of its rows' runs ------------------------------------------------------
struct Ent { uint32_t col, pos; float val; uint32_t lr; };       // pos: the column's sweep position (the column id, or its warp table entry)
struct Scratch { std::vector<Ent> a, b; std::vector<uint32_t> runs, runs2; };

// The column-sorted entries of one bin (ties: lower local row first, then the row's own order -- what a stable sort
// by column of the rows laid end to end gives).  Strided pieces of a split row span the whole sweep.
const std::vector<Ent>& gather_bin(const Layout& L, const Csr& A, int64_t bin, const uint32_t* warp, int32_t wshift, Scratch& S) {
    S.a.clear();
    S.runs.clear();
    S.runs.push_back(0);
    for (int32_t k = 0; k < L.R; k++) {
        const int64_t vi = L.assign[(size_t)bin * L.R + k];
        if (vi < 0) continue;
        const VRow& vr = L.v[(size_t)vi];
        const int32_t b = A.rowptr[vr.row], e = A.rowptr[vr.row + 1];
        for (int32_t p = b + vr.piece; p < e; p += vr.npieces) {
            const uint32_t c = (uint32_t)A.col[p];
            S.a.push_back({c, warp ? warp[c >> wshift] : c, A.val ? A.val[p] : 0.f, (uint32_t)k});
        }
        if (S.a.size() > S.runs.back()) S.runs.push_back((uint32_t)S.a.size());
    }
    std::vector<Ent>* src = &S.a;
    std::vector<Ent>* dst = &S.b;
    std::vector<uint32_t>* rs = &S.runs;
    std::vector<uint32_t>* rd = &S.runs2;
    while (rs->size() > 2) {
        dst->resize(src->size());
        rd->clear();
        rd->push_back(0);
        const size_t nr = rs->size() - 1;
        for (size_t i = 0; i < nr; i += 2) {
            const uint32_t lo = (*rs)[i], mid = (*rs)[i + 1], hi = i + 1 < nr ? (*rs)[i + 2] : mid;
            const Ent* x = src->data() + lo;
            const Ent* xe = src->data() + mid;
            const Ent* y = xe;
            const Ent* ye = src->data() + hi;
            Ent* o = dst->data() + lo;
            while (x < xe && y < ye) *o++ = (y->col < x->col) ? *y++ : *x++;
            while (x < xe) *o++ = *x++;
            while (y < ye) *o++ = *y++;
            rd->push_back(hi);
        }
        std::swap(src, dst);
        std::swap(rs, rd);
    }
    return *src;
}

// ---- two / four lane groups per wavefront (sgcn_csplan_t.G == 2 / 4) --------------------------------------------
// A tile is NG bins of up to 16 virtual rows; with two groups lanes 0-31 hold the accumulators of bin 0, lanes 32-63
// those of bin 1, each lane one float4 of a 128-column slab.  One dwordx4 load instruction then gathers the 512-byte
// slab pieces of two DIFFERENT B rows (the microbenchmark's "x4 on 2 rows": 28.4 TB/s from L2, against 17.7
// for 512-byte pieces fetched as dwordx2), and a wave holds 32 rows of a 128-column slab instead of 16 rows of
// a 304-column one: twice the rows per register byte of slab width, i.e. half the passes of B through every
// XCD (fabric bytes ~ 4 K M d / rows-per-XCD).
// A step applies one entry of each bin.  Entry NG*step + g is bin g's; a PAD entry (value bits 0x80000000, i.e.
// -0.0f -- real -0.0f values are stored as +0.0f) is masked off by the kernel and never touches an
// accumulator.  Pads do two jobs: they fill the shorter bin, and they keep the bins' column positions
// ALIGNED -- the k-th smallest column of two random 1,400-entry bins differs by thousands of columns, and a
// wave whose halves gather from places that far apart needs an L2 window the XCD does not have (measured
// without alignment: 2.6 fetches per B row and pass instead of 1.0).  The schedule walks the sorted lists
// and lets a bin advance only while it is at most `align` positions ahead of the slowest.
// The tile count is rounded up to whole launches of `round_tiles` resident waves so that every launch is full.
constexpr uint32_t kPadBits = 0x80000000u;

// Aligned NG-bin schedule; emit(g, entry-or-null, window).  Returns the number of steps.  A bin applies its next
// entry in a step only while that entry is at most `align` sweep positions (columns; with a warp table: the clock's work
// coordinates, see sgcn_csplan_t.dev_warp) ahead of the slowest bin that still has
// entries (the others get a pad): the bins of a wave then gather from one L2 window.  The step count is padded to
// whole chunks of 64 entries (64 / NG steps): the pipelined kernels run without tail code.
template <class Emit>
int64_t gn_schedule(const std::vector<Ent>* const* e, int NG, int32_t align, Emit emit) {
    size_t pos[4] = {0, 0, 0, 0};
    int64_t step = 0;
    uint32_t window = 0;
    for (;;) {
        int64_t lo = INT64_MAX;
        uint32_t lo_col = 0;
        for (int g = 0; g < NG; g++)
            if (pos[g] < e[g]->size() && (int64_t)(*e[g])[pos[g]].pos < lo) { lo = (*e[g])[pos[g]].pos; lo_col = (*e[g])[pos[g]].col; }
        if (lo == INT64_MAX) break;
        window = lo_col;                      // (a pad gathers from the COLUMN the slowest bin is at)
        for (int g = 0; g < NG; g++) {
            const bool has = pos[g] < e[g]->size();
            const bool take = has && (align <= 0 || (int64_t)(*e[g])[pos[g]].pos <= lo + align);
            emit(g, take ? &(*e[g])[pos[g]] : nullptr, window);
            pos[g] += take;
        }
        step++;
    }
    const int64_t per_chunk = 64 / NG;
    while (step % per_chunk != 0) {
        for (int g = 0; g < NG; g++) emit(g, nullptr, window);
        step++;
    }
    return step;
}

inline uint64_t pack_entry(uint32_t word, float v) {
    uint32_t bits;
    memcpy(&bits, &v, 4);
    return (uint64_t)word | ((uint64_t)bits << 32);
}

}  // namespace

// The plan as the builder holds it between sgcn_csplan_build and sgcn_csbuild_export.
struct sgcn_csbuild {
    int32_t G = 1, R = 16, T = 0;
    int64_t ntiles = 0, nentries = 0, nfix = 0, nslots = 0;
    bool count_only = false;
    std::vector<int64_t> tile_ptr;                    // [ntiles + 1]
    std::vector<int32_t> tile_rows, tile_slots;       // [ntiles * G * R]
    std::vector<sgcn_fix_t> fix;
    std::unique_ptr<uint64_t[]> flat;                 // G == 1: entries at their final offsets (value bits << 32 | colrow word)
    std::vector<std::vector<uint64_t>> arena;         // G >= 2: per building thread, tiles in the order it claimed them
    std::vector<int32_t> tile_tid;
    std::vector<int64_t> tile_off;
    int nthreads = 1;
};

namespace {

int build_plan(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t G, int32_t R, int32_t T,
               int32_t round_tiles, int32_t align, const int32_t* row_group, const uint32_t* warp, int32_t wshift,
               int32_t nthreads, bool count_only, sgcn_csbuild* B) {
    const int nthr = plan_threads(nthreads);
    B->G = G; B->R = G == 1 ? R : kG2R; B->nthreads = nthr; B->count_only = count_only;
    if (T <= 0) T = default_t(rowptr, M);
    B->T = T;
    if (row_group)
        for (int32_t r = 0; r < M; r++)
            if (row_group[r] < 0) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: negative group label at row %d", r);
    // slots: consecutive per split row, in row order (the fix-up adds them in this order)
    std::vector<int32_t> first_slot((size_t)M, -1);
    {
        int64_t slot = 0;
        for (int32_t r = 0; r < M; r++) {
            const int64_t n = (int64_t)rowptr[r + 1] - rowptr[r];
            if (n < 0) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: rowptr not monotone at %d", r);
            if (n <= T) continue;
            const int32_t c = (int32_t)((n + T - 1) / T);
            first_slot[r] = (int32_t)slot;
            B->fix.push_back(sgcn_fix_t{r, (int32_t)slot, c});
            slot += c;
            if (slot > INT32_MAX) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: too many workspace slots");
        }
        B->nfix = (int64_t)B->fix.size();
        B->nslots = slot;
    }
    const int bits = (G == 1 && R > 16) ? 27 : 28;
    Csr A;
    if (M > 0) {
        const int rc = prepare_csr(rowptr, col, val, M, bits, nthr, A);
        if (rc == 1) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: a column does not fit %d bits", bits);
        if (rc == 2) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: out of host memory");
    }
    Layout L;
    if (G == 1) layout_g1(rowptr, M, R, T, row_group, L);
    else layout_gn(rowptr, M, T, round_tiles, G, L);
    const int64_t nt = L.ntiles;
    const int32_t RR = L.R * L.NG;                                   // row slots per tile
    B->ntiles = nt;
    B->tile_ptr.assign((size_t)nt + 1, 0);
    B->tile_rows.resize((size_t)nt * RR);
    B->tile_slots.resize((size_t)nt * RR);
    auto fill_rows = [&](int64_t t) {
        for (int32_t k = 0; k < RR; k++) {
            const int64_t s = t * RR + k;
            const int64_t vi = L.assign[(size_t)s];
            if (vi < 0) { B->tile_rows[(size_t)s] = -1; B->tile_slots[(size_t)s] = -1; continue; }
            const VRow& vr = L.v[(size_t)vi];
            B->tile_rows[(size_t)s] = vr.row;
            B->tile_slots[(size_t)s] = vr.npieces > 1 ? first_slot[vr.row] + vr.piece : -1;
        }
    };
    const uint32_t shift = (uint32_t)bits;
    if (G == 1) {
        // no pads: a tile's entry count is the sum of its virtual rows' weights -- offsets first, then the tiles in parallel
        for (int64_t t = 0; t < nt; t++) {
            int64_t n = 0;
            for (int32_t k = 0; k < RR; k++) {
                const int64_t vi = L.assign[(size_t)(t * RR + k)];
                if (vi >= 0) n += L.v[(size_t)vi].nnz;
            }
            B->tile_ptr[(size_t)t + 1] = B->tile_ptr[(size_t)t] + n;
        }
        B->nentries = B->tile_ptr[(size_t)nt];
        if (count_only) { for (int64_t t = 0; t < nt; t++) fill_rows(t); return SGCN_OK; }
        B->flat.reset(new (std::nothrow) uint64_t[(size_t)std::max<int64_t>(B->nentries, 1)]);
        if (!B->flat) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: out of host memory");
        uint64_t* flat = B->flat.get();
        const bool ok = parallel_chunks(nt, 8, nthr, [&](int64_t t0, int64_t t1, int) {
            Scratch S;
            for (int64_t t = t0; t < t1; t++) {
                fill_rows(t);
                const std::vector<Ent>& e = gather_bin(L, A, t, nullptr, 0, S);
                uint64_t* o = flat + B->tile_ptr[(size_t)t];
                for (const Ent& en : e) *o++ = pack_entry(en.col | (en.lr << shift), en.val);
            }
        });
        if (!ok) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: out of host memory");
        return SGCN_OK;
    }
    // G = 2 / 4: the schedule decides a tile's length (pads) -- per-thread arenas, offsets afterwards
    B->arena.resize((size_t)nthr);
    B->tile_tid.assign((size_t)nt, 0);
    B->tile_off.assign((size_t)nt, 0);
    std::vector<int64_t> tile_len((size_t)nt, 0);
    const int64_t nnz = M > 0 ? (int64_t)rowptr[M] - rowptr[0] : 0;
    const int NG = G;
    const bool ok = parallel_chunks(nt, 8, nthr, [&](int64_t t0, int64_t t1, int tid) {
        Scratch S[4];
        const std::vector<Ent>* e[4] = {nullptr, nullptr, nullptr, nullptr};
        std::vector<uint64_t>& ar = B->arena[(size_t)tid];
        if (!count_only && ar.capacity() == 0) ar.reserve((size_t)(nnz / nthr + nnz / (4 * nthr) + 4096));
        for (int64_t t = t0; t < t1; t++) {
            fill_rows(t);
            for (int g = 0; g < NG; g++) e[g] = &gather_bin(L, A, NG * t + g, warp, wshift, S[g]);
            B->tile_tid[(size_t)t] = tid;
            B->tile_off[(size_t)t] = (int64_t)ar.size();
            int64_t steps;
            if (count_only) {
                steps = gn_schedule(e, NG, align, [](int, const Ent*, uint32_t) {});
            } else {
                // a pad gathers from the column the wave's slowest bin is at: inside the L2 window, never applied
                steps = gn_schedule(e, NG, align, [&](int, const Ent* en, uint32_t window) {
                    if (en) {
                        float v = en->val;
                        uint32_t bts;
                        memcpy(&bts, &v, 4);
                        if (bts == kPadBits) v = 0.0f;          // a real -0.0f: stored as +0.0f (the pad marker is -0.0f)
                        ar.push_back(pack_entry(en->col | (en->lr << 28), v));
                    } else {
                        ar.push_back((uint64_t)window | ((uint64_t)kPadBits << 32));
                    }
                });
            }
            tile_len[(size_t)t] = steps * NG;
        }
    });
    if (!ok) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: out of host memory");
    for (int64_t t = 0; t < nt; t++) B->tile_ptr[(size_t)t + 1] = B->tile_ptr[(size_t)t] + tile_len[(size_t)t];
    B->nentries = B->tile_ptr[(size_t)nt];
    return SGCN_OK;
}

int export_plan(const sgcn_csbuild* B, int64_t* tile_ptr, int32_t* colrow, float* valout, int32_t* tile_rows,
                int32_t* tile_slots, sgcn_fix_t* fix) {
    if (B->count_only) return sgcn::fail(SGCN_ERR_INVALID, "csbuild_export: a count-only build holds no entries");
    if (B->ntiles > 0 && (!tile_ptr || !tile_rows || !tile_slots)) return sgcn::fail(SGCN_ERR_INVALID, "csbuild_export: bad argument");
    if (B->nentries > 0 && (!colrow || !valout)) return sgcn::fail(SGCN_ERR_INVALID, "csbuild_export: bad argument");
    if (B->nfix > 0 && !fix) return sgcn::fail(SGCN_ERR_INVALID, "csbuild_export: split rows but no fix array");
    if (tile_ptr) memcpy(tile_ptr, B->tile_ptr.data(), B->tile_ptr.size() * sizeof(int64_t));
    if (B->tile_rows.size()) {
        memcpy(tile_rows, B->tile_rows.data(), B->tile_rows.size() * sizeof(int32_t));
        memcpy(tile_slots, B->tile_slots.data(), B->tile_slots.size() * sizeof(int32_t));
    }
    if (B->nfix) memcpy(fix, B->fix.data(), B->fix.size() * sizeof(sgcn_fix_t));
    auto split = [&](const uint64_t* src, int64_t n, int64_t at) {
        for (int64_t i = 0; i < n; i++) {
            colrow[at + i] = (int32_t)(uint32_t)src[i];
            const uint32_t bits = (uint32_t)(src[i] >> 32);
            memcpy(valout + at + i, &bits, 4);
        }
    };
    if (B->G == 1) {
        parallel_chunks(B->nentries, 1 << 18, B->nthreads, [&](int64_t b, int64_t e, int) { split(B->flat.get() + b, e - b, b); });
        return SGCN_OK;
    }
    parallel_chunks(B->ntiles, 8, B->nthreads, [&](int64_t t0, int64_t t1, int) {
        for (int64_t t = t0; t < t1; t++)
            split(B->arena[(size_t)B->tile_tid[(size_t)t]].data() + B->tile_off[(size_t)t],
                  B->tile_ptr[(size_t)t + 1] - B->tile_ptr[(size_t)t], B->tile_ptr[(size_t)t]);
    });
    return SGCN_OK;
}

}  // namespace

extern "C" {

int sgcn_csplan_build(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t ngroups, int32_t R,
                      int32_t T, int32_t round_tiles, int32_t align, const int32_t* row_group, const uint32_t* host_warp,
                      int32_t warp_shift, int32_t nthreads, sgcn_csbuild_t** out) {
    if (!out) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: bad argument");
    *out = nullptr;
    if (M < 0 || (M > 0 && (!rowptr || !col || !val))) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: bad argument");
    if (ngroups != 1 && ngroups != 2 && ngroups != 4) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: one, two or four lane groups per wavefront");
    if (ngroups == 1 && (R < 1 || R > 32)) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: 1 <= R <= 32");
    if (ngroups != 1 && row_group) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: plans with lane groups are ungrouped");
    if (host_warp && (warp_shift < 0 || warp_shift > 27)) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: bad warp_shift");
    sgcn_csbuild* B = new (std::nothrow) sgcn_csbuild();
    if (!B) return sgcn::fail(SGCN_ERR_INVALID, "csplan_build: out of host memory");
    int rc;
    try {
        rc = build_plan(rowptr, col, val, M, ngroups, R, T, round_tiles, align, row_group, host_warp, warp_shift, nthreads, false, B);
    } catch (...) {
        rc = sgcn::fail(SGCN_ERR_INVALID, "csplan_build: out of host memory");
    }
    if (rc != SGCN_OK) { delete B; return rc; }
    *out = B;
    return SGCN_OK;
}

int sgcn_csbuild_sizes(const sgcn_csbuild_t* b, int64_t* ntiles, int64_t* nentries, int64_t* nfix, int64_t* nslots,
                       int32_t* T_used, int32_t* threads_used) {
    if (!b) return sgcn::fail(SGCN_ERR_INVALID, "csbuild_sizes: bad argument");
    if (ntiles) *ntiles = b->ntiles;
    if (nentries) *nentries = b->nentries;
    if (nfix) *nfix = b->nfix;
    if (nslots) *nslots = b->nslots;
    if (T_used) *T_used = b->T;
    if (threads_used) *threads_used = b->nthreads;
    return SGCN_OK;
}

int sgcn_csbuild_export(const sgcn_csbuild_t* b, int64_t* tile_ptr, int32_t* colrow, float* valout, int32_t* tile_rows,
                        int32_t* tile_slots, sgcn_fix_t* fix) {
    if (!b) return sgcn::fail(SGCN_ERR_INVALID, "csbuild_export: bad argument");
    return export_plan(b, tile_ptr, colrow, valout, tile_rows, tile_slots, fix);
}

void sgcn_csbuild_free(sgcn_csbuild_t* b) { delete b; }

// ---- the two-call forms (count, then fill into the caller's arrays): thin wrappers of the builder ----------------
int sgcn_csplan_count(const int32_t* rowptr, int32_t M, int32_t R, int32_t T, const int32_t* row_group,
                      int64_t* ntiles, int64_t* nfix, int64_t* nslots) {
    if (M < 0 || (M > 0 && !rowptr) || R < 1 || R > 32 || !ntiles || !nfix || !nslots)
        return sgcn::fail(SGCN_ERR_INVALID, "csplan_count: bad argument");
    if (T <= 0) T = default_t(rowptr, M);
    if (row_group)
        for (int32_t r = 0; r < M; r++)
            if (row_group[r] < 0) return sgcn::fail(SGCN_ERR_INVALID, "csplan_count: negative group label at row %d", r);
    std::vector<int32_t> order;
    std::vector<int64_t> gptr;
    bucket_rows(row_group, M, order, gptr);
    int64_t nt = 0, f = 0, s = 0;
    for (size_t g = 0; g + 1 < gptr.size(); g++) {
        int64_t nv = 0;
        for (int64_t i = gptr[g]; i < gptr[g + 1]; i++) {
            const int32_t r = order[(size_t)i];
            const int64_t n = (int64_t)rowptr[r + 1] - rowptr[r];
            if (n < 0) return sgcn::fail(SGCN_ERR_INVALID, "csplan_count: rowptr not monotone at %d", r);
            if (n <= T) nv += 1;
            else { const int64_t c = (n + T - 1) / T; nv += c; s += c; f += 1; }
        }
        nt += (nv + R - 1) / R;        // tiles never straddle groups
    }
    *ntiles = nt;
    *nfix = f;
    *nslots = s;
    return SGCN_OK;
}

int sgcn_csplan_fill(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M,
                     int32_t R, int32_t T, const int32_t* row_group, int64_t* tile_ptr, int32_t* colrow,
                     float* valout, int32_t* tile_rows, int32_t* tile_slots, sgcn_fix_t* fix) {
    if (M < 0 || R < 1 || R > 32 || (M > 0 && (!rowptr || !tile_ptr || !tile_rows || !tile_slots)))
        return sgcn::fail(SGCN_ERR_INVALID, "csplan_fill: bad argument");
    sgcn_csbuild_t* b = nullptr;
    int rc = sgcn_csplan_build(rowptr, col, val, M, 1, R, T, 0, 0, row_group, nullptr, 0, 0, &b);
    if (rc != SGCN_OK) return rc;
    rc = export_plan(b, tile_ptr, colrow, valout, tile_rows, tile_slots, fix);
    delete b;
    return rc;
}

int sgcn_csplang_count(const int32_t* rowptr, const int32_t* col, int32_t M, int32_t T, int32_t round_tiles,
                       int32_t align, int32_t ngroups, const uint32_t* host_warp, int32_t warp_shift,
                       int64_t* ntiles, int64_t* nentries, int64_t* nfix, int64_t* nslots) {
    if (ngroups != 2 && ngroups != 4) return sgcn::fail(SGCN_ERR_INVALID, "csplang_count: two or four lane groups per wavefront");
    if (host_warp && (warp_shift < 0 || warp_shift > 27)) return sgcn::fail(SGCN_ERR_INVALID, "csplang_count: bad warp_shift");
    if (M < 0 || (M > 0 && (!rowptr || !col)) || !ntiles || !nentries || !nfix || !nslots)
        return sgcn::fail(SGCN_ERR_INVALID, "csplang_count: bad argument");
    sgcn_csbuild B;
    int rc;
    try {
        rc = build_plan(rowptr, col, nullptr, M, ngroups, kG2R, T, round_tiles, align, nullptr, host_warp, warp_shift, 0, true, &B);
    } catch (...) {
        rc = sgcn::fail(SGCN_ERR_INVALID, "csplang_count: out of host memory");
    }
    if (rc != SGCN_OK) return rc;
    *ntiles = B.ntiles; *nentries = B.nentries; *nfix = B.nfix; *nslots = B.nslots;
    return SGCN_OK;
}

int sgcn_csplang_fill(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t T,
                      int32_t round_tiles, int32_t align, int32_t ngroups, const uint32_t* host_warp, int32_t warp_shift,
                      int64_t* tile_ptr, int32_t* colrow,
                      float* valout, int32_t* tile_rows, int32_t* tile_slots, sgcn_fix_t* fix) {
    if (ngroups != 2 && ngroups != 4) return sgcn::fail(SGCN_ERR_INVALID, "csplang_fill: two or four lane groups per wavefront");
    if (M > 0 && (!tile_ptr || !tile_rows || !tile_slots)) return sgcn::fail(SGCN_ERR_INVALID, "csplang_fill: bad argument");
    sgcn_csbuild_t* b = nullptr;
    int rc = sgcn_csplan_build(rowptr, col, val, M, ngroups, kG2R, T, round_tiles, align, nullptr, host_warp, warp_shift, 0, &b);
    if (rc != SGCN_OK) return rc;
    rc = export_plan(b, tile_ptr, colrow, valout, tile_rows, tile_slots, fix);
    delete b;
    return rc;
}

// ---- the sweep clock's warp table (sgcn_csplan_t.dev_warp) on the host, in parallel --------------------------------
// table[b] = share of the nonzeros in columns < (b << shift), scaled to [0, K); *nbuckets = 0 when `mode` is auto (0) and
// no column's share of the work in front of it is off its share of the ids by more than auto_dev (the linear clock holds).
// The arithmetic is float64 in the order ops.ColumnSweepCSR.make_warp had it in numpy (rounds 5): the same table, bit for bit.
int sgcn_cs_warp_table(const int32_t* col, int64_t nnz, int32_t K, int32_t max_buckets, int32_t mode, double auto_dev,
                       int32_t nthreads, uint32_t* table, int32_t* nbuckets, int32_t* shift_out) {
    if (!nbuckets || !shift_out || K < 0 || nnz < 0 || (nnz > 0 && !col) || max_buckets < 1)
        return sgcn::fail(SGCN_ERR_INVALID, "cs_warp_table: bad argument");
    *nbuckets = 0;
    *shift_out = 0;
    if (K == 0 || nnz == 0) return SGCN_OK;
    int32_t shift = 0;
    while ((((int64_t)K - 1) >> shift) + 1 > max_buckets) shift++;
    const int64_t nb = (((int64_t)K - 1) >> shift) + 1;
    const int nthr = (int)std::max<int64_t>(1, std::min<int64_t>(plan_threads(nthreads), nnz / (1 << 16) + 1));
    std::vector<std::vector<int64_t>> part((size_t)nthr);
    std::atomic<int> bad{0};
    const bool ok = run_threads(nthr, [&](int tid) {
        std::vector<int64_t>& h = part[(size_t)tid];
        h.assign((size_t)nb, 0);
        const int64_t b = nnz * tid / nthr, e = nnz * (tid + 1) / nthr;
        int bb = 0;
        for (int64_t p = b; p < e; p++) {
            const int64_t c = col[p];
            if (c < 0 || c >= K) { bb = 1; continue; }
            h[(size_t)(c >> shift)]++;
        }
        if (bb) bad.store(1);
    });
    if (!ok) return sgcn::fail(SGCN_ERR_INVALID, "cs_warp_table: out of host memory");
    if (bad.load()) return sgcn::fail(SGCN_ERR_INVALID, "cs_warp_table: a column outside [0, K)");
    std::vector<int64_t>& hist = part[0];
    for (int t = 1; t < nthr; t++)
        for (int64_t b = 0; b < nb; b++) hist[(size_t)b] += part[(size_t)t][(size_t)b];
    const double total = (double)nnz;
    double worst = 0.0;
    int64_t before = 0;
    std::vector<double> share((size_t)nb);
    for (int64_t b = 0; b < nb; b++) {
        share[(size_t)b] = (double)before / total;
        const double ids = ((double)b * (double)((int64_t)1 << shift)) / (double)K;
        worst = std::max(worst, std::fabs(share[(size_t)b] - ids));
        before += hist[(size_t)b];
    }
    if (mode == 0 && worst <= auto_dev) return SGCN_OK;
    if (!table) return sgcn::fail(SGCN_ERR_INVALID, "cs_warp_table: no table array");
    for (int64_t b = 0; b < nb; b++) {
        const double v = std::floor(share[(size_t)b] * (double)K);
        table[b] = (uint32_t)std::min<double>(v, (double)(K - 1));
    }
    *nbuckets = (int32_t)nb;
    *shift_out = shift;
    return SGCN_OK;
}

// ---- CSR transpose on the host, in parallel ---------------------------------------------------------------------
// The plan of A^T (the backward product, gcn/layers.py:31-37's autodiff) needs A^T as a CSR on the host.  A stable
// counting sort by column: row blocks of equal nonzeros count their columns, a prefix over (column, block) gives every
// block its write position per column, the blocks scatter.  Within a column the rows ascend (and duplicates keep their
// order) -- SciPy's csr -> csc pass (sparsetools csr_tocsc), which the plans were built from until round 5, bit for bit.
int sgcn_csr_transpose_host(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t K,
                            int32_t nthreads, int32_t* t_rowptr, int32_t* t_col, float* t_val) {
    if (M < 0 || K < 0 || (M > 0 && !rowptr) || !t_rowptr)
        return sgcn::fail(SGCN_ERR_INVALID, "csr_transpose_host: bad argument");
    const int64_t p0 = M > 0 ? rowptr[0] : 0;
    const int64_t nnz = M > 0 ? (int64_t)rowptr[M] - p0 : 0;
    if (nnz < 0 || (nnz > 0 && (!col || !t_col || (val && !t_val)))) return sgcn::fail(SGCN_ERR_INVALID, "csr_transpose_host: bad argument");
    // (K x blocks) int32 counters: at most 1 GiB
    int nthr = plan_threads(nthreads);
    while (nthr > 1 && (int64_t)nthr * (K + 1) * 4 > ((int64_t)1 << 30)) nthr--;
    nthr = (int)std::max<int64_t>(1, std::min<int64_t>(nthr, nnz / (1 << 16) + 1));
    std::vector<int32_t> rb((size_t)nthr + 1, M);
    rb[0] = 0;
    for (int t = 1; t < nthr; t++)
        rb[(size_t)t] = (int32_t)(std::lower_bound(rowptr, rowptr + M + 1, (int32_t)(p0 + nnz * t / nthr)) - rowptr);
    for (int t = 1; t <= nthr; t++) rb[(size_t)t] = std::max(rb[(size_t)t], rb[(size_t)t - 1]);
    rb[(size_t)nthr] = M;
    std::vector<std::vector<int32_t>> cnt((size_t)nthr);
    std::atomic<int> bad{0};
    bool ok = run_threads(nthr, [&](int tid) {
        std::vector<int32_t>& c = cnt[(size_t)tid];
        c.assign((size_t)K + 1, 0);
        int bb = 0;
        for (int64_t p = rowptr[rb[(size_t)tid]]; p < rowptr[rb[(size_t)tid + 1]]; p++) {
            const int32_t cc = col[p];
            if (cc < 0 || cc >= K) { bb = 1; continue; }
            c[(size_t)cc]++;
        }
        if (bb) bad.store(1);
    });
    if (!ok) return sgcn::fail(SGCN_ERR_INVALID, "csr_transpose_host: out of host memory");
    if (bad.load()) return sgcn::fail(SGCN_ERR_INVALID, "csr_transpose_host: a column outside [0, K)");
    // per-column totals (parallel over column ranges), exclusive scan (serial: K adds), then each block's start per column
    parallel_chunks(K, 1 << 14, nthr, [&](int64_t c0, int64_t c1, int) {
        for (int64_t c = c0; c < c1; c++) {
            int32_t s = 0;
            for (int t = 0; t < nthr; t++) s += cnt[(size_t)t][(size_t)c];
            t_rowptr[c + 1] = s;
        }
    });
    t_rowptr[0] = 0;
    for (int64_t c = 0; c < K; c++) t_rowptr[c + 1] += t_rowptr[c];
    parallel_chunks(K, 1 << 14, nthr, [&](int64_t c0, int64_t c1, int) {
        for (int64_t c = c0; c < c1; c++) {
            int32_t run = t_rowptr[c];
            for (int t = 0; t < nthr; t++) {
                const int32_t n = cnt[(size_t)t][(size_t)c];
                cnt[(size_t)t][(size_t)c] = run;
                run += n;
            }
        }
    });
    run_threads(nthr, [&](int tid) {
        std::vector<int32_t>& c = cnt[(size_t)tid];
        for (int32_t r = rb[(size_t)tid]; r < rb[(size_t)tid + 1]; r++)
            for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) {
                const int32_t dst = c[(size_t)col[p]]++;
                t_col[dst] = r;
                if (val) t_val[dst] = val[p];
            }
    });
    return SGCN_OK;
}

int32_t sgcn_host_threads(void) { return plan_threads(0); }

// Graph-only locality labelling: asynchronous label propagation (Raghavan et al. 2007) on the
// symmetrised pattern of a square CSR.  Every vertex starts in its own community and, visited in a
// seeded random order, adopts the label most of its neighbours carry (ties: the smallest label --
// deterministic); a few sweeps find the dense blocks of a graph that has them and collapse a graph
// that has none (uniform S-Reddit) into one label, which makes the reordering a no-op there.
// Output: comm[v] in [0, ncomm), communities numbered by decreasing size (ties: smallest member).
int sgcn_reorder_lp(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t max_iters, uint32_t seed,
                    int32_t min_size, int32_t* comm, int32_t* ncomm) {
    if (n < 0 || (n > 0 && (!rowptr || !comm)) || !ncomm)
        return sgcn::fail(SGCN_ERR_INVALID, "reorder_lp: bad argument");
    if (max_iters <= 0) max_iters = 12;
    std::vector<int32_t> label((size_t)n), visit((size_t)n);
    std::iota(label.begin(), label.end(), 0);
    std::iota(visit.begin(), visit.end(), 0);
    sgcn::Mt19937 gen(seed);
    for (int32_t i = n - 1; i > 0; i--) std::swap(visit[i], visit[gen.next() % (uint32_t)(i + 1)]);
    std::vector<int32_t> cnt((size_t)n, 0), touched;
    for (int32_t it = 0; it < max_iters; it++) {
        int64_t changed = 0;
        for (int32_t vi = 0; vi < n; vi++) {
            const int32_t u = visit[vi];
            const int32_t b = rowptr[u], e = rowptr[u + 1];
            if (e == b) continue;
            touched.clear();
            for (int32_t p = b; p < e; p++) {
                const int32_t c = col[p];
                if (c < 0 || c >= n) return sgcn::fail(SGCN_ERR_INVALID, "reorder_lp: column %d out of range", c);
                const int32_t l = label[c];
                if (cnt[l]++ == 0) touched.push_back(l);
            }
            int32_t best = label[u], bestc = 0;
            for (int32_t l : touched) {
                const int32_t c = cnt[l];
                if (c > bestc || (c == bestc && l < best)) { best = l; bestc = c; }
                cnt[l] = 0;
            }
            if (best != label[u]) { label[u] = best; changed++; }
        }
        if (changed * 1000 < (int64_t)n) break;        // < 0.1 % of the vertices moved
    }
    // sizes, then number the communities by decreasing size; communities below min_size share the last id
    std::vector<int64_t> size((size_t)n, 0);
    for (int32_t u = 0; u < n; u++) size[label[u]]++;
    std::vector<int32_t> ids;
    for (int32_t l = 0; l < n; l++) if (size[l] > 0) ids.push_back(l);
    std::stable_sort(ids.begin(), ids.end(), [&](int32_t a, int32_t b) { return size[a] > size[b]; });
    std::vector<int32_t> newid((size_t)n, -1);
    int32_t next = 0;
    bool misc = false;
    for (int32_t l : ids) {
        if (size[l] >= min_size) newid[l] = next++;
        else misc = true;
    }
    for (int32_t l : ids) if (newid[l] < 0) newid[l] = next;
    for (int32_t u = 0; u < n; u++) comm[u] = newid[label[u]];
    *ncomm = next + (misc ? 1 : 0);
    return SGCN_OK;
}

}  // extern "C"
