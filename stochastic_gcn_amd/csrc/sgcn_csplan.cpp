// Host-side column-sweep plan (include/sgcn.h, sgcn_csplan_t): virtual rows -> degree-sorted
// tiles -> per-tile column-sorted merged nonzero lists.
#include "sgcn_host.h"
#include "../../include/sgcn.h"

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <queue>
#include <functional>
#include <vector>

namespace {

struct VRow { int32_t row, piece, npieces, nnz; };

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

// virtual rows: a row with n <= T nonzeros is one; a longer row becomes ceil(n/T) strided pieces
void make_vrows(const int32_t* rowptr, const int32_t* rows, int64_t nrows, int32_t T, std::vector<VRow>& v) {
    for (int64_t ri = 0; ri < nrows; ri++) {
        const int32_t r = rows[ri];
        const int32_t n = rowptr[r + 1] - rowptr[r];
        if (n <= T) { v.push_back({r, 0, 1, n}); continue; }
        const int32_t c = (n + T - 1) / T;
        for (int32_t q = 0; q < c; q++) v.push_back({r, q, c, (n - q + c - 1) / c});
    }
    // sorted by weight for the LPT dealing in fill(); ties keep row order (deterministic)
    std::stable_sort(v.begin(), v.end(), [](const VRow& a, const VRow& b) { return a.nnz > b.nnz; });
}

// Longest-processing-time dealing: the next heaviest virtual row goes to the lightest tile that
// still has a free slot, so every tile carries (nearly) the same number of nonzeros -- a paced
// sweep lasts as long as its heaviest tile.  Returns slot -> index into v (or -1).
std::vector<int64_t> deal(const std::vector<VRow>& v, int32_t R, int64_t nt) {
    std::vector<int64_t> assign((size_t)nt * R, -1);
    std::vector<int32_t> fill((size_t)nt, 0);
    typedef std::pair<int64_t, int64_t> WT;          // (weight, tile); min-heap, ties -> low tile id
    std::priority_queue<WT, std::vector<WT>, std::greater<WT>> heap;
    for (int64_t t = 0; t < nt; t++) heap.push({0, t});
    for (size_t i = 0; i < v.size(); i++) {
        WT top = heap.top();
        heap.pop();
        const int64_t t = top.second;
        assign[(size_t)t * R + fill[t]++] = (int64_t)i;
        if (fill[t] < R) heap.push({top.first + v[i].nnz, t});
    }
    return assign;
}

}  // namespace

extern "C" {

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
    if (T <= 0) T = default_t(rowptr, M);
    // slots: consecutive per split row, in row order (the fix-up adds them in this order)
    std::vector<int32_t> first_slot((size_t)M, -1);
    int32_t slot = 0;
    int64_t f = 0;
    for (int32_t r = 0; r < M; r++) {
        const int32_t n = rowptr[r + 1] - rowptr[r];
        if (n <= T) continue;
        const int32_t c = (n + T - 1) / T;
        first_slot[r] = slot;
        if (!fix) return sgcn::fail(SGCN_ERR_INVALID, "csplan_fill: split rows but no fix array");
        fix[f++] = sgcn_fix_t{r, slot, c};
        slot += c;
    }
    std::vector<int32_t> order;
    std::vector<int64_t> gptr;
    bucket_rows(row_group, M, order, gptr);
    struct Ent { int32_t col, lr; float val; };
    std::vector<Ent> ents;
    std::vector<std::pair<int32_t, float>> rowbuf;
    std::vector<VRow> v;
    int64_t out = 0, tbase = 0;
    // groups in label order, each dealt into its own tiles: with a locality-preserving labelling
    // (sgcn_reorder_lp) the tiles that are resident on an XCD together draw on the same B rows
    for (size_t g = 0; g + 1 < gptr.size(); g++) {
        v.clear();
        make_vrows(rowptr, order.data() + gptr[g], gptr[g + 1] - gptr[g], T, v);
        const int64_t nt = ((int64_t)v.size() + R - 1) / R;
        const std::vector<int64_t> assign = deal(v, R, nt);
        for (int64_t tl = 0; tl < nt; tl++) {
            const int64_t t = tbase + tl;
            tile_ptr[t] = out;
            ents.clear();
            for (int32_t k = 0; k < R; k++) {
                const int64_t vi = assign[(size_t)tl * R + k];
                if (vi < 0) { tile_rows[t * R + k] = -1; tile_slots[t * R + k] = -1; continue; }
                const VRow& vr = v[vi];
                tile_rows[t * R + k] = vr.row;
                tile_slots[t * R + k] = vr.npieces > 1 ? first_slot[vr.row] + vr.piece : -1;
                const int32_t b = rowptr[vr.row], e = rowptr[vr.row + 1];
                if (vr.npieces == 1) {
                    for (int32_t p = b; p < e; p++) ents.push_back({col[p], k, val[p]});
                } else {
                    // strided pieces of the column-sorted row: each piece spans the whole sweep
                    rowbuf.clear();
                    for (int32_t p = b; p < e; p++) rowbuf.push_back({col[p], val[p]});
                    std::stable_sort(rowbuf.begin(), rowbuf.end(),
                                     [](const std::pair<int32_t, float>& a, const std::pair<int32_t, float>& c2) { return a.first < c2.first; });
                    for (int32_t i = vr.piece; i < e - b; i += vr.npieces)
                        ents.push_back({rowbuf[i].first, k, rowbuf[i].second});
                }
            }
            std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.col < b.col; });
            for (const Ent& en : ents) {
                const int shift = R <= 16 ? 28 : 27;
                if (en.col < 0 || en.col >= (1 << shift))
                    return sgcn::fail(SGCN_ERR_INVALID, "csplan_fill: column %d does not fit %d bits", en.col, shift);
                colrow[out] = (int32_t)((uint32_t)en.col | ((uint32_t)en.lr << shift));
                valout[out] = en.val;
                out++;
            }
        }
        tbase += nt;
    }
    tile_ptr[tbase] = out;
    return SGCN_OK;
}

}  // extern "C"

// ---- two lane groups per wavefront (sgcn_csplan_t.G == 2) ---------------------------------------------
// A tile is TWO bins of up to 16 virtual rows; lanes 0-31 hold the accumulators of bin 0, lanes 32-63 those of
// bin 1, each lane one float4 of a 128-column slab.  One dwordx4 load instruction then gathers the 512-byte
// slab pieces of two DIFFERENT B rows (the microbenchmark's "x4 on 2 rows": 28.4 TB/s from L2, against 17.7
// for 512-byte pieces fetched as dwordx2), and a wave holds 32 rows of a 128-column slab instead of 16 rows of
// a 304-column one: twice the rows per register byte of slab width, i.e. half the passes of B through every
// XCD (fabric bytes ~ 4 K M d / rows-per-XCD).
// A step applies one entry of each bin.  Entry 2*step + g is bin g's; a PAD entry (value bits 0x80000000, i.e.
// -0.0f -- real -0.0f values are stored as +0.0f) is masked off by the kernel and never touches an
// accumulator.  Pads do two jobs: they fill the shorter bin, and they keep the two bins' column positions
// ALIGNED -- the k-th smallest column of two random 1,400-entry bins differs by thousands of columns, and a
// wave whose halves gather from places that far apart needs an L2 window the XCD does not have (measured
// without alignment: 2.6 fetches per B row and pass instead of 1.0).  The schedule walks both sorted lists
// and lets a bin advance only while it is at most `align` columns ahead of the other.
// The tile count is rounded up to whole launches of `round_tiles` resident waves so that every launch is full.
namespace {
constexpr int kG2R = 16;
constexpr uint32_t kPadBits = 0x80000000u;

struct G2Ent { int32_t col, lr; float val; int64_t pos; };   // pos: the column's sweep position (the column id, or its warp table entry)

struct G2Layout {
    std::vector<VRow> v;
    std::vector<int64_t> assign;      // [nbins * 16] -> index into v, or -1
    int64_t ntiles = 0;
};

void g2_layout(const int32_t* rowptr, int32_t M, int32_t T, int32_t round_tiles, G2Layout& L, int NG = 2) {
    std::vector<int32_t> rows((size_t)M);
    std::iota(rows.begin(), rows.end(), 0);
    make_vrows(rowptr, rows.data(), M, T, L.v);
    const int64_t nv = (int64_t)L.v.size();
    int64_t nt = (nv + NG * kG2R - 1) / (NG * kG2R);
    if (round_tiles > 0 && nt > round_tiles / 2)                       // whole launches (a small matrix just gets enough tiles)
        nt = (nt + round_tiles - 1) / round_tiles * round_tiles;
    L.ntiles = nt;
    L.assign = deal(L.v, kG2R, nt * NG);
}

// the column-sorted entries of one bin
void g2_bin(const G2Layout& L, const int32_t* rowptr, const int32_t* col, const float* val, int64_t bin,
            std::vector<G2Ent>& ents, std::vector<std::pair<int32_t, float>>& rowbuf, const uint32_t* warp, int32_t wshift) {
    ents.clear();
    auto posof = [&](int32_t c) -> int64_t { return warp ? (int64_t)warp[c >> wshift] : (int64_t)c; };
    for (int32_t k = 0; k < kG2R; k++) {
        const int64_t vi = L.assign[(size_t)bin * kG2R + k];
        if (vi < 0) continue;
        const VRow& vr = L.v[vi];
        const int32_t b = rowptr[vr.row], e = rowptr[vr.row + 1];
        if (vr.npieces == 1) {
            for (int32_t p = b; p < e; p++) ents.push_back({col[p], k, val ? val[p] : 0.f, posof(col[p])});
        } else {
            rowbuf.clear();
            for (int32_t p = b; p < e; p++) rowbuf.push_back({col[p], val ? val[p] : 0.f});
            std::stable_sort(rowbuf.begin(), rowbuf.end(),
                             [](const std::pair<int32_t, float>& a, const std::pair<int32_t, float>& c2) { return a.first < c2.first; });
            for (int32_t i = vr.piece; i < e - b; i += vr.npieces) ents.push_back({rowbuf[i].first, k, rowbuf[i].second, posof(rowbuf[i].first)});
        }
    }
    std::stable_sort(ents.begin(), ents.end(), [](const G2Ent& a, const G2Ent& b) { return a.col < b.col; });
}

// Aligned NG-bin schedule; emit(step, g, entry-or-null).  Returns the number of steps.  A bin applies its next
// entry in a step only while that entry is at most `align` sweep positions (columns; with a warp table: the clock's work
// coordinates, see sgcn_csplan_t.dev_warp) ahead of the slowest bin that still has
// entries (the others get a pad): the bins of a wave then gather from one L2 window.  The step count is padded to
// whole chunks of 64 entries (64 / NG steps): the pipelined kernels run without tail code.
template <class Emit>
int64_t gn_schedule(const std::vector<G2Ent>* e, int NG, int32_t align, Emit emit) {
    size_t pos[4] = {0, 0, 0, 0};
    int64_t step = 0, window = 0;
    for (;;) {
        int64_t lo = INT64_MAX, lo_col = 0;
        for (int g = 0; g < NG; g++)
            if (pos[g] < e[g].size() && e[g][pos[g]].pos < lo) { lo = e[g][pos[g]].pos; lo_col = e[g][pos[g]].col; }
        if (lo == INT64_MAX) break;
        window = lo_col;                      // (a pad gathers from the COLUMN the slowest bin is at)
        for (int g = 0; g < NG; g++) {
            const bool has = pos[g] < e[g].size();
            const bool take = has && (align <= 0 || e[g][pos[g]].pos <= lo + align);
            emit(step, g, take ? &e[g][pos[g]] : nullptr, window);
            pos[g] += take;
        }
        step++;
    }
    const int64_t per_chunk = 64 / NG;
    while (step % per_chunk != 0) {
        for (int g = 0; g < NG; g++) emit(step, g, nullptr, window);
        step++;
    }
    return step;
}

int gn_count(const int32_t* rowptr, const int32_t* col, int32_t M, int32_t T, int32_t round_tiles, int32_t align, int NG,
             const uint32_t* warp, int32_t wshift, int64_t* ntiles, int64_t* nentries, int64_t* nfix, int64_t* nslots) {
    if (M < 0 || (M > 0 && (!rowptr || !col)) || !ntiles || !nentries || !nfix || !nslots || (NG != 2 && NG != 4))
        return sgcn::fail(SGCN_ERR_INVALID, "csplang_count: bad argument");
    if (T <= 0) T = default_t(rowptr, M);
    int64_t f = 0, sl = 0;
    for (int32_t r = 0; r < M; r++) {
        const int64_t n = (int64_t)rowptr[r + 1] - rowptr[r];
        if (n < 0) return sgcn::fail(SGCN_ERR_INVALID, "csplang_count: rowptr not monotone at %d", r);
        if (n > T) { sl += (n + T - 1) / T; f += 1; }
    }
    G2Layout L;
    g2_layout(rowptr, M, T, round_tiles, L, NG);
    std::vector<G2Ent> e[4];
    std::vector<std::pair<int32_t, float>> rowbuf;
    int64_t entries = 0;
    for (int64_t t = 0; t < L.ntiles; t++) {
        for (int g = 0; g < NG; g++) g2_bin(L, rowptr, col, nullptr, NG * t + g, e[g], rowbuf, warp, wshift);
        entries += NG * gn_schedule(e, NG, align, [](int64_t, int, const G2Ent*, int64_t) {});
    }
    *ntiles = L.ntiles; *nentries = entries; *nfix = f; *nslots = sl;
    return SGCN_OK;
}

int gn_fill(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t T, int32_t round_tiles,
            int32_t align, int NG, const uint32_t* warp, int32_t wshift, int64_t* tile_ptr, int32_t* colrow, float* valout,
            int32_t* tile_rows, int32_t* tile_slots, sgcn_fix_t* fix) {
    if (M < 0 || (M > 0 && (!rowptr || !col || !val || !tile_ptr || !tile_rows || !tile_slots)) || (NG != 2 && NG != 4))
        return sgcn::fail(SGCN_ERR_INVALID, "csplang_fill: bad argument");
    if (T <= 0) T = default_t(rowptr, M);
    std::vector<int32_t> first_slot((size_t)M, -1);
    int32_t slot = 0;
    int64_t f = 0;
    for (int32_t r = 0; r < M; r++) {
        const int32_t n = rowptr[r + 1] - rowptr[r];
        if (n <= T) continue;
        const int32_t c = (n + T - 1) / T;
        first_slot[r] = slot;
        if (!fix) return sgcn::fail(SGCN_ERR_INVALID, "csplang_fill: split rows but no fix array");
        fix[f++] = sgcn_fix_t{r, slot, c};
        slot += c;
    }
    G2Layout L;
    g2_layout(rowptr, M, T, round_tiles, L, NG);
    std::vector<G2Ent> e[4];
    std::vector<std::pair<int32_t, float>> rowbuf;
    int64_t out = 0;
    int bad = 0;
    for (int64_t t = 0; t < L.ntiles; t++) {
        tile_ptr[t] = out;
        for (int g = 0; g < NG; g++) {
            for (int32_t k = 0; k < kG2R; k++) {
                const int64_t slot_idx = (t * NG + g) * kG2R + k;
                const int64_t vi = L.assign[(size_t)slot_idx];
                if (vi < 0) { tile_rows[slot_idx] = -1; tile_slots[slot_idx] = -1; continue; }
                const VRow& vr = L.v[vi];
                tile_rows[slot_idx] = vr.row;
                tile_slots[slot_idx] = vr.npieces > 1 ? first_slot[vr.row] + vr.piece : -1;
            }
            g2_bin(L, rowptr, col, val, NG * t + g, e[g], rowbuf, warp, wshift);
        }
        // a pad gathers from the column the wave's slowest bin is at: inside the L2 window, never applied
        gn_schedule(e, NG, align, [&](int64_t, int, const G2Ent* en, int64_t window) {
            uint32_t word;
            float v;
            if (en) {
                if (en->col < 0 || en->col >= (1 << 28)) bad = 1;
                word = (uint32_t)en->col | ((uint32_t)en->lr << 28);
                v = en->val;
                uint32_t bits;
                memcpy(&bits, &v, 4);
                if (bits == kPadBits) v = 0.0f;          // a real -0.0f: stored as +0.0f (the pad marker is -0.0f)
            } else {
                word = (uint32_t)window;
                const uint32_t bits = kPadBits;
                memcpy(&v, &bits, 4);
            }
            colrow[out] = (int32_t)word;
            valout[out] = v;
            out++;
        });
    }
    if (bad) return sgcn::fail(SGCN_ERR_INVALID, "csplang_fill: a column does not fit 28 bits");
    tile_ptr[L.ntiles] = out;
    return SGCN_OK;
}
}  // namespace

extern "C" {

int sgcn_csplang_count(const int32_t* rowptr, const int32_t* col, int32_t M, int32_t T, int32_t round_tiles,
                       int32_t align, int32_t ngroups, const uint32_t* host_warp, int32_t warp_shift,
                       int64_t* ntiles, int64_t* nentries, int64_t* nfix, int64_t* nslots) {
    if (ngroups != 2 && ngroups != 4) return sgcn::fail(SGCN_ERR_INVALID, "csplang_count: two or four lane groups per wavefront");
    if (host_warp && (warp_shift < 0 || warp_shift > 27)) return sgcn::fail(SGCN_ERR_INVALID, "csplang_count: bad warp_shift");
    return gn_count(rowptr, col, M, T, round_tiles, align, ngroups, host_warp, warp_shift, ntiles, nentries, nfix, nslots);
}

int sgcn_csplang_fill(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t T,
                      int32_t round_tiles, int32_t align, int32_t ngroups, const uint32_t* host_warp, int32_t warp_shift,
                      int64_t* tile_ptr, int32_t* colrow,
                      float* valout, int32_t* tile_rows, int32_t* tile_slots, sgcn_fix_t* fix) {
    if (ngroups != 2 && ngroups != 4) return sgcn::fail(SGCN_ERR_INVALID, "csplang_fill: two or four lane groups per wavefront");
    if (host_warp && (warp_shift < 0 || warp_shift > 27)) return sgcn::fail(SGCN_ERR_INVALID, "csplang_fill: bad warp_shift");
    return gn_fill(rowptr, col, val, M, T, round_tiles, align, ngroups, host_warp, warp_shift, tile_ptr, colrow, valout, tile_rows,
                   tile_slots, fix);
}

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
