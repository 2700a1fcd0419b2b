// Host-side column-sweep plan (include/sgcn.h, sgcn_csplan_t): virtual rows -> degree-sorted
// tiles -> per-tile column-sorted merged nonzero lists.
#include "sgcn_host.h"
#include "../../include/sgcn.h"

#include <algorithm>
#include <numeric>
#include <vector>

namespace {

struct VRow { int32_t row, piece, npieces, nnz; };

constexpr int32_t kDefaultT = 512;

// virtual rows: a row with n <= T nonzeros is one; a longer row becomes ceil(n/T) strided pieces
void make_vrows(const int32_t* rowptr, int32_t M, int32_t T, std::vector<VRow>& v) {
    for (int32_t r = 0; r < M; r++) {
        const int32_t n = rowptr[r + 1] - rowptr[r];
        if (n <= T) { v.push_back({r, 0, 1, n}); continue; }
        const int32_t c = (n + T - 1) / T;
        for (int32_t q = 0; q < c; q++) v.push_back({r, q, c, (n - q + c - 1) / c});
    }
    // sorted by weight; fill() deals them out boustrophedon so that every tile carries the same
    // number of nonzeros (a paced sweep takes as long as its heaviest tile); ties keep row order
    std::stable_sort(v.begin(), v.end(), [](const VRow& a, const VRow& b) { return a.nnz > b.nnz; });
}

// virtual row k of tile t: pass k of a snake over the weight-sorted list
inline int64_t snake(int64_t t, int32_t k, int64_t nt) { return (int64_t)k * nt + ((k & 1) ? nt - 1 - t : t); }

}  // namespace

extern "C" {

int sgcn_csplan_count(const int32_t* rowptr, int32_t M, int32_t R, int32_t T, int64_t* ntiles,
                      int64_t* nfix, int64_t* nslots) {
    if (M < 0 || (M > 0 && !rowptr) || R < 1 || R > 32 || !ntiles || !nfix || !nslots)
        return sgcn::fail(SGCN_ERR_INVALID, "csplan_count: bad argument");
    if (T <= 0) T = kDefaultT;
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
    if (T <= 0) T = kDefaultT;
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
    struct Ent { int32_t col, lr; float val; };
    std::vector<Ent> ents;
    std::vector<std::pair<int32_t, float>> rowbuf;
    int64_t out = 0;
    for (int64_t t = 0; t < nt; t++) {
        tile_ptr[t] = out;
        ents.clear();
        for (int32_t k = 0; k < R; k++) {
            const int64_t vi = snake(t, k, nt);
            if (vi >= (int64_t)v.size()) { tile_rows[t * R + k] = -1; tile_slots[t * R + k] = -1; continue; }
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
