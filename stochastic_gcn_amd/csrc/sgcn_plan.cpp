// Host-side work plan for power-law CSR rows (include/sgcn.h, "work plan").
// Pure arithmetic on the row pointer: rows with more than T nonzeros are cut into
// ceil(nnz/T) segments that accumulate into consecutive workspace slots; all other rows
// are one direct segment.  Segments keep row order, so a wavefront's neighbouring groups
// touch neighbouring rows of C.
#include "sgcn_host.h"
#include "../../include/sgcn.h"
#include <vector>

namespace {
constexpr int32_t kDefaultT = 256;
inline int32_t pick_t(int32_t T) { return T > 0 ? T : kDefaultT; }
}  // namespace

namespace sgcn {
// vector-based variant used by the sampler's packed batch (same rules as sgcn_plan_fill)
void plan_build(const int32_t* rowptr, int32_t M, int32_t T, std::vector<int32_t>& seg,
                std::vector<int32_t>& fix, int64_t& nslots) {
    T = pick_t(T);
    seg.clear(); fix.clear();
    int32_t slot = 0;
    for (int32_t r = 0; r < M; r++) {
        const int32_t b = rowptr[r], e = rowptr[r + 1];
        if (e - b <= T) { seg.insert(seg.end(), {r, b, e, -1}); continue; }
        const int32_t first = slot;
        const int32_t c = (int32_t)(((int64_t)e - b + T - 1) / T);
        for (int32_t q = 0; q < c; q++) {
            const int32_t qb = b + (int32_t)(((int64_t)(e - b) * q) / c);
            const int32_t qe = b + (int32_t)(((int64_t)(e - b) * (q + 1)) / c);
            seg.insert(seg.end(), {r, qb, qe, slot++});
        }
        fix.insert(fix.end(), {r, first, c});
    }
    nslots = slot;
}
}  // namespace sgcn

extern "C" {

int sgcn_plan_count(const int32_t* rowptr, int32_t M, int32_t T, int64_t* nseg, int64_t* nfix,
                    int64_t* nslots) {
    if (M < 0 || (M > 0 && !rowptr) || !nseg || !nfix || !nslots)
        return sgcn::fail(SGCN_ERR_INVALID, "plan_count: bad argument");
    T = pick_t(T);
    int64_t s = 0, f = 0, w = 0;
    for (int32_t r = 0; r < M; r++) {
        const int64_t n = (int64_t)rowptr[r + 1] - rowptr[r];
        if (n < 0) return sgcn::fail(SGCN_ERR_INVALID, "plan_count: rowptr not monotone at %d", r);
        if (n <= T) s += 1;
        else { const int64_t c = (n + T - 1) / T; s += c; w += c; f += 1; }
    }
    *nseg = s; *nfix = f; *nslots = w;
    return SGCN_OK;
}

int sgcn_plan_fill(const int32_t* rowptr, int32_t M, int32_t T, sgcn_seg_t* seg, sgcn_fix_t* fix) {
    if (M < 0 || (M > 0 && (!rowptr || !seg)))
        return sgcn::fail(SGCN_ERR_INVALID, "plan_fill: bad argument");
    T = pick_t(T);
    int64_t s = 0, f = 0;
    int32_t slot = 0;
    for (int32_t r = 0; r < M; r++) {
        const int32_t b = rowptr[r], e = rowptr[r + 1];
        if (e - b <= T) { seg[s++] = sgcn_seg_t{r, b, e, -1}; continue; }
        if (!fix) return sgcn::fail(SGCN_ERR_INVALID, "plan_fill: split rows but no fix array");
        const int32_t first = slot;
        // equal-sized pieces (not T,T,..,rest) so no wavefront gets a tiny tail piece
        const int32_t c = (int32_t)(((int64_t)e - b + T - 1) / T);
        for (int32_t q = 0; q < c; q++) {
            const int32_t qb = b + (int32_t)(((int64_t)(e - b) * q) / c);
            const int32_t qe = b + (int32_t)(((int64_t)(e - b) * (q + 1)) / c);
            seg[s++] = sgcn_seg_t{r, qb, qe, slot++};
        }
        fix[f++] = sgcn_fix_t{r, first, c};
    }
    return SGCN_OK;
}

}  // extern "C"
