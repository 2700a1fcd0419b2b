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

// default split threshold: a few times the mean degree, so that no single (virtual) row
// dominates a 16-row tile, within [64, 512]
inline int32_t default_t(const int32_t* rowptr, int32_t M) {
    const int64_t avg = M > 0 ? ((int64_t)rowptr[M] - rowptr[0]) / M : 0;
    return (int32_t)std::min<int64_t>(512, std::max<int64_t>(64, 4 * avg));
}

// virtual rows: a row with n <= T nonzeros is one; a longer row becomes ceil(n/T) strided pieces
void make_vrows(const int32_t* rowptr, int32_t M, int32_t T, std::vector<VRow>& v) {
    for (int32_t r = 0; r < M; r++) {
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

int sgcn_csplan_count(const int32_t* rowptr, int32_t M, int32_t R, int32_t T, int64_t* ntiles,
                      int64_t* nfix, int64_t* nslots) {
    if (M < 0 || (M > 0 && !rowptr) || R < 1 || R > 32 || !ntiles || !nfix || !nslots)
        return sgcn::fail(SGCN_ERR_INVALID, "csplan_count: bad argument");
    if (T <= 0) T = default_t(rowptr, M);
    int64_t nv = 0, f = 0, s = 0;
    for (int32_t r = 0; r < M; r++) {
        const int64_t n = (int64_t)rowptr[r + 1] - rowptr[r];
        if (n < 0) return sgcn::fail(SGCN_ERR_INVALID, "csplan_count: rowptr not monotone at %d", r);
        if (n <= T) nv += 1;
        else { const int64_t c = (n + T - 1) / T; nv += c; s += c; f += 1; }
    }
    *ntiles = (nv + R - 1) / R;
    *nfix = f;
    *nslots = s;
    return SGCN_OK;
}

int sgcn_csplan_fill(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M,
                     int32_t R, int32_t T, int64_t* tile_ptr, int32_t* colrow, float* valout,
                     int32_t* tile_rows, int32_t* tile_slots, sgcn_fix_t* fix) {
    if (M < 0 || R < 1 || R > 32 || (M > 0 && (!rowptr || !tile_ptr || !tile_rows || !tile_slots)))
        return sgcn::fail(SGCN_ERR_INVALID, "csplan_fill: bad argument");
    if (T <= 0) T = default_t(rowptr, M);
    std::vector<VRow> v;
    make_vrows(rowptr, M, T, v);
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
    const int64_t nt = ((int64_t)v.size() + R - 1) / R;
    const std::vector<int64_t> assign = deal(v, R, nt);
    struct Ent { int32_t col, lr; float val; };
    std::vector<Ent> ents;
    std::vector<std::pair<int32_t, float>> rowbuf;
    int64_t out = 0;
    for (int64_t t = 0; t < nt; t++) {
        tile_ptr[t] = out;
        ents.clear();
        for (int32_t k = 0; k < R; k++) {
            const int64_t vi = assign[(size_t)t * R + k];
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
    tile_ptr[nt] = out;
    return SGCN_OK;
}

}  // extern "C"
