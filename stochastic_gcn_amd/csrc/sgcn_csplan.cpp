// Host-side column-sweep plan (include/sgcn.h, sgcn_csplan_t): virtual rows -> degree-sorted
// tiles -> per-tile column-sorted merged nonzero lists.
#include "sgcn_host.h"
#include "../../include/sgcn.h"

#include <algorithm>
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

// default split threshold: a few times the mean degree, so that no single (virtual) row
// dominates a 16-row tile, within [64, 512]
inline int32_t default_t(const int32_t* rowptr, int32_t M) {
    const int64_t avg = M > 0 ? ((int64_t)rowptr[M] - rowptr[0]) / M : 0;
    return (int32_t)std::min<int64_t>(512, std::max<int64_t>(64, 4 * avg));
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
