// Fused control-variate aggregator forward for gfx950.
//
// Replaces VRAggregator._call (gcn/layers.py:298-319 cvd branch, :350-362 plain-CV branch):
// three tf.sparse_tensor_dense_matmul + two tf.gather per layer become ONE pass per output
// row in which the history rows Hbar[ffield[c]] / Hbar[ifield[c]] are read straight from the
// resident N x d history (mu_large / mu_small are never materialised), the sampled adjacency
// is walked once for both streams, and the (h_self | h_nbr) concat of normalization !=
// 'gcn' is written in place.
//
//   cvd : mu_nbr = A (mu - Hbar[ifield]) + P Hbar[ffield]
//         h_nbr  = (A (h - mu)) * s + mu_nbr
//   cv  : out    = A x - A Hbar[ifield] + P Hbar[ffield]
//
// The three sums are kept in separate fp32 accumulators and combined in the reference's
// order.  P rows are full neighbour lists (power-law): they follow the same host plan /
// workspace-slot / ordered fix-up scheme as sgcn_spmm.hip.  One G-lane group per row, one
// 16-byte vector per lane (G = 32 for d = 128, 8 for d = 32), feature slabs of G vectors
// for wider rows.
#include "sgcn_dev.h"

namespace sgcn {

int group_lanes(int nvec);  // sgcn_spmm.hip

struct AggArgs {
    const int32_t* a_rowptr; const int32_t* a_col; const float* a_val;
    const int32_t* f_rowptr; const int32_t* f_col; const float* f_val;
    const sgcn_seg_t* seg; int64_t nseg; int64_t nsegblk;
    const float* h; const float* mu; int64_t ldx;
    const float* H; int64_t ldh;
    const int32_t* ifield; const int32_t* ffield; const float* s;
    float* out_h; float* out_mu; int64_t ldo;
    int32_t d, nvec, cvd, off, concat;
    float* ws; int64_t ldw;
    // two-phase form (sgcn_vr_aggregate_pre/post_f32): the P-sum of every row goes to / comes from
    // accP[row * ldw ...] instead of living in registers across the two halves of the fused pass
    float* accP_out; const float* accP_in;
};

template <int VW>
__device__ __forceinline__ void store_masked(float* p, typename Vec<VW>::type v, int left) {
    if (left >= VW) vstore<VW>(p, v);
    else vstore_head<VW>(p, v, left);
}

// Sampled-adjacency part of one (row, vector): acc1 = A (mu - Hbar[ifield]) | A x,  acc2 = A (h - mu) | A Hbar[ifield]
template <int G, int VW>
__device__ __forceinline__ void agg_apart(const AggArgs& a, int row, int vi, int lig, bool act,
                                          typename Vec<VW>::type& acc1, typename Vec<VW>::type& acc2) {
    typedef typename Vec<VW>::type VT;
    acc1 = vzero<VW>(); acc2 = vzero<VW>();
    const int start = uniform_i<G>(a.a_rowptr[row]), end = uniform_i<G>(a.a_rowptr[row + 1]);
    const int64_t voff = (int64_t)vi * VW;
    for (int p0 = start; p0 < end; p0 += G) {
        const int n = min(G, end - p0);
        int mycol = 0, myhist = 0;
        float myval = 0.f;
        if (lig < n) {
            mycol = a.a_col[p0 + lig];
            myval = a.a_val[p0 + lig];
            myhist = a.ifield[mycol];
        }
        for (int j = 0; j < n; j++) {
            const int c = bcast_i<G>(mycol, j);
            const int hr = bcast_i<G>(myhist, j);
            const float v = bcast_f<G>(myval, j);
            if (!act) continue;
            const VT hb = vload<VW>(a.H + (int64_t)hr * a.ldh + voff);
            const VT xv = vload<VW>(a.h + (int64_t)c * a.ldx + voff);
            if (a.cvd) {
                const VT mv = vload<VW>(a.mu + (int64_t)c * a.ldx + voff);
                acc1 += v * (mv - hb);   // A (mu - Hbar[ifield])
                acc2 += v * (xv - mv);   // A (h - mu)
            } else {
                acc1 += v * xv;          // A x
                acc2 += v * hb;          // A Hbar[ifield]
            }
        }
    }
}

// The three sums combined in the reference's order, and the (self | neighbour) concat written in place.
template <int VW>
__device__ __forceinline__ void agg_epilogue(const AggArgs& a, int row, int vi, bool act, typename Vec<VW>::type accP,
                                             typename Vec<VW>::type acc1, typename Vec<VW>::type acc2) {
    typedef typename Vec<VW>::type VT;
    if (!act) return;
    const int64_t voff = (int64_t)vi * VW;
    const int left = a.d - vi * VW;
    float* oh = a.out_h + (int64_t)row * a.ldo;
    if (a.cvd) {
        float* om = a.out_mu + (int64_t)row * a.ldo;
        const VT mu_nbr = acc1 + accP;
        const VT h_nbr = acc2 * a.s[row] + mu_nbr;
        store_masked<VW>(oh + a.off + voff, h_nbr, left);
        store_masked<VW>(om + a.off + voff, mu_nbr, left);
        if (a.concat) {
            store_masked<VW>(oh + voff, vload<VW>(a.h + (int64_t)row * a.ldx + voff), left);
            store_masked<VW>(om + voff, vload<VW>(a.mu + (int64_t)row * a.ldx + voff), left);
        }
    } else {
        const VT out = (acc1 - acc2) + accP;
        store_masked<VW>(oh + a.off + voff, out, left);
        if (a.concat) store_masked<VW>(oh + voff, vload<VW>(a.h + (int64_t)row * a.ldx + voff), left);
    }
}

// Sampled-adjacency part + epilogue for one (row, vector) given the finished P-sum.
template <int G, int VW>
__device__ __forceinline__ void agg_finish(const AggArgs& a, int row, int vi, int lig, bool act,
                                           typename Vec<VW>::type accP) {
    typename Vec<VW>::type acc1, acc2;
    agg_apart<G, VW>(a, row, vi, lig, act, acc1, acc2);
    agg_epilogue<VW>(a, row, vi, act, accP, acc1, acc2);
}

// One WORKGROUP per plan segment (= one output row unless the row is longer than the plan's T): the workgroup's
// NG = 256 / G lane groups take consecutive chunks of the segment's nonzeros, every lane keeps up to U row pieces of
// the history in flight, and the NG partial sums are added in group order through LDS (fixed order: deterministic).
// A Reddit CVD batch is 512 rows of ~100 history rows (512 B each, scattered over a 119 MB history): with one lane
// group per 64-nonzero segment (rounds 1-2) 2 MB were in flight and the pass took 15 us + 6 us for the fix-up
// kernel every row then needed; this form has the whole 26 MB requested at once and needs the fix-up only for the rows
// longer than T = 128 (one round of 8 groups x U = 16; a longer segment would make the launch wait for its extra rounds:
// T = 1,024 measured 29-34 us, set by the one hub row half of the batches contain).
template <int G, int VW, int U>
__global__ __launch_bounds__(kBlock) void agg_row_kernel(AggArgs a) {
    typedef typename Vec<VW>::type VT;
    constexpr int NG = kBlock / G, NP = NG - 1;       // NP groups share the P-sum, the last one walks the sampled adjacency
    __shared__ float part[NG + 1][G * VW];
    const int lig = threadIdx.x & (G - 1), gq = threadIdx.x / G;
    const int slab = (int)(blockIdx.x / a.nseg);
    const int64_t s = blockIdx.x % a.nseg;
    int row, start, end, slot;
    if (a.seg) {
        const sgcn_seg_t sg = a.seg[s];
        row = sg.row; start = sg.start; end = sg.end; slot = sg.slot;
    } else {
        row = (int)s; start = a.f_rowptr[s]; end = a.f_rowptr[s + 1]; slot = -1;
    }
    row = uniform_i<G>(row); start = uniform_i<G>(start);
    end = uniform_i<G>(end); slot = uniform_i<G>(slot);
    const int vi = slab * G + lig;
    const bool act = vi < a.nvec;
    const bool finish_here = slot < 0 && !a.accP_out;              // an unsplit row of the fused pass
    const float* Hl = a.H + (int64_t)vi * VW;

    if (gq < NP) {
        const int chunk = (end - start + NP - 1) / NP;
        const int gs = start + gq * chunk, ge = min(end, gs + chunk);
        VT accP = vzero<VW>();
        for (int p0 = gs; p0 < ge; p0 += G) {
            const int n = min(G, ge - p0);
            int myrow = 0;
            float myval = 0.f;
            if (lig < n) {
                myrow = a.ffield[a.f_col[p0 + lig]];
                myval = a.f_val[p0 + lig];
            }
            for (int j = 0; j < n; j += U) {
                VT b[U];
                float v[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const bool in = j + u < n;
                    const int c = bcast_i<G>(myrow, in ? j + u : 0);
                    v[u] = in ? bcast_f<G>(myval, j + u) : 0.f;
                    b[u] = (act && in) ? vload<VW>(Hl + (int64_t)c * a.ldh) : vzero<VW>();
                }
#pragma unroll
                for (int u = 0; u < U; u++) accP += v[u] * b[u];
            }
        }
        vstore<VW>(&part[gq][lig * VW], accP);
    } else if (finish_here) {
        // beside the P-sum instead of after it: its index chain (rowptr -> column -> ifield -> rows) is as long as the P-sum's
        VT acc1, acc2;
        agg_apart<G, VW>(a, row, vi, lig, act, acc1, acc2);
        vstore<VW>(&part[NP][lig * VW], acc1);
        vstore<VW>(&part[NG][lig * VW], acc2);
    }
    __syncthreads();
    if (gq != 0) return;
    VT accP = vload<VW>(&part[0][lig * VW]);
#pragma unroll
    for (int q = 1; q < NP; q++) accP += vload<VW>(&part[q][lig * VW]);
    if (slot >= 0) { if (act) vstore<VW>(a.ws + (int64_t)slot * a.ldw + (int64_t)vi * VW, accP); }
    else if (a.accP_out) { if (act) vstore<VW>(a.accP_out + (int64_t)row * a.ldw + (int64_t)vi * VW, accP); }
    else agg_epilogue<VW>(a, row, vi, act, accP, vload<VW>(&part[NP][lig * VW]), vload<VW>(&part[NG][lig * VW]));
}

template <int G, int VW>
__global__ __launch_bounds__(kBlock) void agg_fix_kernel(AggArgs a, const sgcn_fix_t* fix, int64_t nfix) {
    typedef typename Vec<VW>::type VT;
    constexpr int GPB = kBlock / G;
    const int lig = threadIdx.x & (G - 1);
    const int64_t nfblk = (nfix + GPB - 1) / GPB;
    const int slab = (int)(blockIdx.x / nfblk);
    const int64_t f = (blockIdx.x % nfblk) * GPB + threadIdx.x / G;
    if (f >= nfix) return;
    const sgcn_fix_t fx = fix[f];
    const int vi = slab * G + lig;
    const bool act = vi < a.nvec;
    VT accP = vzero<VW>();
    if (act) {
        const float* w = a.ws + (int64_t)fx.first_slot * a.ldw + (int64_t)vi * VW;
        for (int q = 0; q < fx.nslots; q++) accP += vload<VW>(w + (int64_t)q * a.ldw);
    }
    if (a.accP_out) { if (act) vstore<VW>(a.accP_out + (int64_t)fx.row * a.ldw + (int64_t)vi * VW, accP); }
    else agg_finish<G, VW>(a, uniform_i<G>(fx.row), vi, lig, act, accP);
}

// second phase of the two-phase form: one group per output row, P-sum read back from accP_in
template <int G, int VW>
__global__ __launch_bounds__(kBlock) void agg_post_kernel(AggArgs a, int32_t n1) {
    typedef typename Vec<VW>::type VT;
    constexpr int GPB = kBlock / G;
    const int lig = threadIdx.x & (G - 1);
    const int64_t nrblk = ((int64_t)n1 + GPB - 1) / GPB;
    const int slab = (int)(blockIdx.x / nrblk);
    const int64_t row = (blockIdx.x % nrblk) * GPB + threadIdx.x / G;
    if (row >= n1) return;
    const int vi = slab * G + lig;
    const bool act = vi < a.nvec;
    const VT accP = act ? vload<VW>(a.accP_in + row * a.ldw + (int64_t)vi * VW) : vzero<VW>();
    agg_finish<G, VW>(a, uniform_i<G>((int)row), vi, lig, act, accP);
}

template <int VW>
static int launch_agg_post(int G, const AggArgs& a, int32_t n1, hipStream_t st) {
    const int nslab = (a.nvec + G - 1) / G;
#define SGCN_AGGP_CASE(GG)                                                                               \
    case GG: {                                                                                           \
        const int64_t nrblk = ((int64_t)n1 + (kBlock / GG) - 1) / (kBlock / GG);                          \
        hipLaunchKernelGGL((agg_post_kernel<GG, VW>), dim3((unsigned)(nrblk * nslab)), dim3(kBlock), 0, st, a, n1); \
        break;                                                                                           \
    }
    switch (G) {
        SGCN_AGGP_CASE(8)
        SGCN_AGGP_CASE(16)
        SGCN_AGGP_CASE(32)
        SGCN_AGGP_CASE(64)
        default: return fail(SGCN_ERR_INVALID, "vr_aggregate_post: bad group %d", G);
    }
#undef SGCN_AGGP_CASE
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

template <int VW>
static int launch_agg(int G, const AggArgs& a, const sgcn_plan_t* plan, hipStream_t st) {
    const int nslab = (a.nvec + G - 1) / G;
    const int64_t nblocks = a.nseg * nslab;                      // one workgroup per (segment, feature slab)
    SGCN_REQUIRE(nblocks < (1ll << 31), "vr_aggregate: grid too large");
#define SGCN_AGG_CASE(GG)                                                                          \
    case GG: {                                                                                     \
        hipLaunchKernelGGL((agg_row_kernel<GG, VW, 16>), dim3((unsigned)nblocks), dim3(kBlock), 0,  \
                           st, a);                                                                 \
        if (plan && plan->nfix > 0) {                                                              \
            const int64_t nfblk = (plan->nfix + (kBlock / GG) - 1) / (kBlock / GG);               \
            hipLaunchKernelGGL((agg_fix_kernel<GG, VW>), dim3((unsigned)(nfblk * nslab)),          \
                               dim3(kBlock), 0, st, a, plan->dev_fix, plan->nfix);                 \
        }                                                                                          \
        break;                                                                                     \
    }
    switch (G) {
        SGCN_AGG_CASE(8)
        SGCN_AGG_CASE(16)
        SGCN_AGG_CASE(32)
        SGCN_AGG_CASE(64)
        default: return fail(SGCN_ERR_INVALID, "vr_aggregate: bad group %d", G);
    }
#undef SGCN_AGG_CASE
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

}  // namespace sgcn

using namespace sgcn;

extern "C" int sgcn_vr_aggregate_f32(const int32_t* a_rowptr, const int32_t* a_col,
                                     const float* a_val, const int32_t* f_rowptr,
                                     const int32_t* f_col, const float* f_val, int32_t n1,
                                     int32_t n0, int32_t nf, int32_t d, const float* h,
                                     const float* mu, int64_t ldx, const float* Hbar, int64_t ldh,
                                     const int32_t* ifield, const int32_t* ffield, const float* s,
                                     float* out_h, float* out_mu, int64_t ldo, int32_t cvd,
                                     int32_t concat_self, const sgcn_plan_t* f_plan, void* stream) {
    SGCN_REQUIRE(n1 >= 0 && n0 >= 0 && nf >= 0 && d >= 0, "vr_aggregate: negative size");
    if (n1 == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(a_rowptr && f_rowptr && h && Hbar && ifield && out_h, "vr_aggregate: null operand");
    SGCN_REQUIRE(nf == 0 || ffield, "vr_aggregate: null ffield");
    SGCN_REQUIRE(!cvd || (mu && s && out_mu), "vr_aggregate: cvd needs mu, s, out_mu");
    SGCN_REQUIRE(n1 <= n0 || !concat_self, "vr_aggregate: concat_self needs n1 <= n0");
    const int64_t width = concat_self ? 2 * (int64_t)d : d;
    SGCN_REQUIRE(ldx >= d && ldh >= d && ldo >= width, "vr_aggregate: leading dimension too small");

    AggArgs a{};
    a.a_rowptr = a_rowptr; a.a_col = a_col; a.a_val = a_val;
    a.f_rowptr = f_rowptr; a.f_col = f_col; a.f_val = f_val;
    a.h = h; a.mu = mu; a.ldx = ldx; a.H = Hbar; a.ldh = ldh;
    a.ifield = ifield; a.ffield = ffield; a.s = s;
    a.out_h = out_h; a.out_mu = out_mu; a.ldo = ldo;
    a.d = d; a.cvd = cvd; a.concat = concat_self; a.off = concat_self ? d : 0;
    a.nseg = n1;
    if (f_plan) {
        SGCN_REQUIRE(f_plan->dev_seg && f_plan->nseg >= n1, "vr_aggregate: malformed plan");
        a.seg = f_plan->dev_seg; a.nseg = f_plan->nseg;
        a.ws = f_plan->dev_ws; a.ldw = ((int64_t)d + 3) / 4 * 4;
        if (f_plan->nfix > 0) {
            SGCN_REQUIRE(f_plan->dev_fix && f_plan->dev_ws, "vr_aggregate: plan needs dev_fix/dev_ws");
            SGCN_REQUIRE(f_plan->ws_elems >= f_plan->nslots * a.ldw, "vr_aggregate: workspace too small");
        }
    }
    // the output offset `off` = d must keep vector alignment too
    int vw = pick_vw(d, {h, mu, Hbar, out_h, out_mu, f_plan ? f_plan->dev_ws : nullptr}, {ldx, ldh, ldo});
    if (concat_self) while (vw > 1 && d % vw != 0) vw >>= 1;
    a.nvec = (d + vw - 1) / vw;
    const int G = group_lanes(a.nvec);
    a.nsegblk = (a.nseg + (kBlock / G) - 1) / (kBlock / G);
    hipStream_t st = (hipStream_t)stream;
    if (vw == 4) return launch_agg<4>(G, a, f_plan, st);
    if (vw == 2) return launch_agg<2>(G, a, f_plan, st);
    return launch_agg<1>(G, a, f_plan, st);
}

// ---- the same aggregate in two phases -------------------------------------------------------------
// P . Hbar[ffield] -- the dominant gather of the step (Reddit CVD+PP: 50.8 k history rows of 512 B) --
// depends on nothing the step computes: only on the history and the minibatch.  _pre computes it for
// every output row into accP (n1 x ldw, ldw = 4*ceil(d/4)), so that it can run BESIDE the dense layers
// that produce h / mu (the step program issues it on the auxiliary stream); _post is the rest of the
// fused pass.  Per element the two phases perform exactly the fused kernel's operations in the fused
// kernel's order -- accP only takes a round trip through memory -- so pre + post == sgcn_vr_aggregate_f32
// bit for bit (tests/test_kernels_gpu.py).
extern "C" int sgcn_vr_aggregate_pre_f32(const int32_t* f_rowptr, const int32_t* f_col, const float* f_val,
                                         int32_t n1, int32_t nf, int32_t d, const float* Hbar, int64_t ldh,
                                         const int32_t* ffield, float* accP, const sgcn_plan_t* f_plan,
                                         void* stream) {
    SGCN_REQUIRE(n1 >= 0 && nf >= 0 && d >= 0, "vr_aggregate_pre: negative size");
    if (n1 == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(f_rowptr && Hbar && accP && ldh >= d, "vr_aggregate_pre: bad operand");
    SGCN_REQUIRE(nf == 0 || ffield, "vr_aggregate_pre: null ffield");
    AggArgs a{};
    a.f_rowptr = f_rowptr; a.f_col = f_col; a.f_val = f_val;
    a.H = Hbar; a.ldh = ldh; a.ffield = ffield; a.d = d;
    a.ldw = ((int64_t)d + 3) / 4 * 4;
    a.accP_out = accP;
    a.nseg = n1;
    if (f_plan) {
        SGCN_REQUIRE(f_plan->dev_seg && f_plan->nseg >= n1, "vr_aggregate_pre: malformed plan");
        a.seg = f_plan->dev_seg; a.nseg = f_plan->nseg; a.ws = f_plan->dev_ws;
        if (f_plan->nfix > 0) {
            SGCN_REQUIRE(f_plan->dev_fix && f_plan->dev_ws, "vr_aggregate_pre: plan needs dev_fix/dev_ws");
            SGCN_REQUIRE(f_plan->ws_elems >= f_plan->nslots * a.ldw, "vr_aggregate_pre: workspace too small");
        }
    }
    const int vw = pick_vw(d, {Hbar, accP, f_plan ? f_plan->dev_ws : nullptr}, {ldh, a.ldw});
    a.nvec = (d + vw - 1) / vw;
    const int G = group_lanes(a.nvec);
    a.nsegblk = (a.nseg + (kBlock / G) - 1) / (kBlock / G);
    hipStream_t st = (hipStream_t)stream;
    if (vw == 4) return launch_agg<4>(G, a, f_plan, st);
    if (vw == 2) return launch_agg<2>(G, a, f_plan, st);
    return launch_agg<1>(G, a, f_plan, st);
}

extern "C" int sgcn_vr_aggregate_post_f32(const int32_t* a_rowptr, const int32_t* a_col, const float* a_val,
                                          int32_t n1, int32_t n0, int32_t d, const float* h, const float* mu,
                                          int64_t ldx, const float* Hbar, int64_t ldh, const int32_t* ifield,
                                          const float* s, float* out_h, float* out_mu, int64_t ldo, int32_t cvd,
                                          int32_t concat_self, const float* accP, void* stream) {
    SGCN_REQUIRE(n1 >= 0 && n0 >= 0 && d >= 0, "vr_aggregate_post: negative size");
    if (n1 == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(a_rowptr && h && Hbar && ifield && out_h && accP, "vr_aggregate_post: null operand");
    SGCN_REQUIRE(!cvd || (mu && s && out_mu), "vr_aggregate_post: cvd needs mu, s, out_mu");
    SGCN_REQUIRE(n1 <= n0 || !concat_self, "vr_aggregate_post: concat_self needs n1 <= n0");
    const int64_t width = concat_self ? 2 * (int64_t)d : d;
    SGCN_REQUIRE(ldx >= d && ldh >= d && ldo >= width, "vr_aggregate_post: leading dimension too small");
    AggArgs a{};
    a.a_rowptr = a_rowptr; a.a_col = a_col; a.a_val = a_val;
    a.h = h; a.mu = mu; a.ldx = ldx; a.H = Hbar; a.ldh = ldh; a.ifield = ifield; a.s = s;
    a.out_h = out_h; a.out_mu = out_mu; a.ldo = ldo;
    a.d = d; a.cvd = cvd; a.concat = concat_self; a.off = concat_self ? d : 0;
    a.ldw = ((int64_t)d + 3) / 4 * 4;
    a.accP_in = accP;
    int vw = pick_vw(d, {h, mu, Hbar, out_h, out_mu, accP}, {ldx, ldh, ldo, a.ldw});
    if (concat_self) while (vw > 1 && d % vw != 0) vw >>= 1;
    a.nvec = (d + vw - 1) / vw;
    const int G = group_lanes(a.nvec);
    hipStream_t st = (hipStream_t)stream;
    if (vw == 4) return launch_agg_post<4>(G, a, n1, st);
    if (vw == 2) return launch_agg_post<2>(G, a, n1, st);
    return launch_agg_post<1>(G, a, n1, st);
}
