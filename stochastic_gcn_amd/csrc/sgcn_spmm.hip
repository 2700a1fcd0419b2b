// CSR x dense SpMM for gfx950 (MI355X):  C = rscale (.) (A (cscale (.) B[g])) + beta C.
//
// Replaces tf.sparse_tensor_dense_matmul at gcn/layers.py:31-37 (K1/K3/K4/K5/K11 in
// SURVEY.md §2.1), fused with the tf.gather of history rows (K2+K7, gcn/layers.py:304-308)
// and, called on the transposed CSR, its autodiff backward dB = A^T dC (K6).
//
// Shape of the kernel (HBM/cache-bound gather; no MFMA on purpose -- the sparse operand has
// no dense tiles to feed a matrix core):
//   * a GROUP of G lanes owns one row segment (G = 8/16/32/64 chosen from the row width so
//     that a group reads whole 16-byte vectors of a B row; G = 64 = one wavefront for wide
//     rows, several rows per wavefront for d <= 128);
//   * the segment's (col, val) pairs are read G at a time, coalesced, one pair per lane, and
//     broadcast with v_readlane (G = 64: the B-row base lives in SGPRs, the lane offset in a
//     VGPR) or ds_bpermute (G < 64);
//   * every lane keeps NV float4 accumulators: lane l owns vectors l, l+G, l+2G.. of the slab,
//     so a wavefront's loads of one B row are contiguous 1 KiB pieces (coalesced HBM/L2
//     reads of the dense operand);
//   * U nonzeros are in flight per group before the first FMA (memory-level parallelism:
//     U * NV dwordx4 loads per lane outstanding);
//   * wide rows can be cut into feature slabs (slab-major block order so one B slab stays
//     resident in the 256 MiB Infinity Cache while every row visits it);
//   * power-law rows are cut into <= T-nonzero segments by a host plan (include/sgcn.h);
//     split rows go through workspace slots and an ordered fix-up pass (deterministic).
#include "sgcn_dev.h"
#include <cstring>

namespace sgcn {

namespace {
int g_tune_nv = 0;
int g_tune_unroll = 0;
int g_tune_slabmajor = 1;
int g_tune_cs_round = 0;
int g_tune_cs_unroll = 0;
int g_tune_cs_pace = 0;
int g_tune_cs_slack = 0;
int g_tune_cs_noextra = 0;
int g_tune_step_overlap = 1;
int g_tune_step_fuse = 127;           // bit 0: the output layer's forward as the loss kernel's head, bit 1: its input gradient as the
                                     // tail, bit 2: the dense layer behind a split-K layer in that layer's reduce pass, bit 3: a layer's
                                     // input gradient in its LayerNorm / ReLU backward pass, bit 4: the dense layer in front of the output
                                     // layer as the pre-layer of the loss kernel's head, bit 5: the weight gradients' reductions in the
                                     // optimizer's launch, bit 6: the first layer's LayerNorm backward behind the second layer's row pass
int g_tune_cs_g2_wide = 0;
int g_tune_cs_nowarp = 0;            // experiments: ignore a plan's warp table (the sweep clock linear in the column id)
int g_tune_cs_last_pct = 80;          // (round 4: 80 % holds the lock-step on the 90-column pass of d = 602: 3.16 vs 3.22 ms at 90, 3.43 at 70)
int g_tune_gemm_min_steps = 0;
int g_tune_lds_wave_bias = 100;    // LDS plan: entries of a tile's waves 0-3 per 100 of its waves 4-7 (100: even; made moot by the s_setprio around the update chain)
int g_tune_lds_mix = 1;            // LDS plan: columns a tile uses once or twice dealt into the chunks among the reused ones (all-staged plans)
int g_tune_lds_dbg = 0;             // experiments on the LDS sweep: bit 0 no ring fills after the first, bit 1 no arithmetic
// a step program's own MODE (sgcn_step_run): seen by the entry points that very thread calls during the run, by nobody else
thread_local int tl_step_overlap = -1, tl_step_fuse = -1;
}  // namespace

void step_mode_override(int overlap, int fuse) { tl_step_overlap = overlap; tl_step_fuse = fuse; }

int tune_get(const char* key) {
    if (!strcmp(key, "spmm_nv")) return g_tune_nv;
    if (!strcmp(key, "spmm_unroll")) return g_tune_unroll;
    if (!strcmp(key, "spmm_slabmajor")) return g_tune_slabmajor;
    if (!strcmp(key, "cs_round")) return g_tune_cs_round;
    if (!strcmp(key, "cs_unroll")) return g_tune_cs_unroll;
    if (!strcmp(key, "cs_pace")) return g_tune_cs_pace;
    if (!strcmp(key, "cs_slack")) return g_tune_cs_slack;
    if (!strcmp(key, "cs_noextra")) return g_tune_cs_noextra;
    if (!strcmp(key, "step_overlap")) return tl_step_overlap >= 0 ? tl_step_overlap : g_tune_step_overlap;
    if (!strcmp(key, "step_fuse")) return tl_step_fuse >= 0 ? tl_step_fuse : g_tune_step_fuse;
    if (!strcmp(key, "cs_g2_wide")) return g_tune_cs_g2_wide;
    if (!strcmp(key, "cs_nowarp")) return g_tune_cs_nowarp;
    if (!strcmp(key, "cs_last_pct")) return g_tune_cs_last_pct;
    if (!strcmp(key, "gemm_min_steps")) return g_tune_gemm_min_steps;
    if (!strcmp(key, "lds_wave_bias")) return g_tune_lds_wave_bias;
    if (!strcmp(key, "lds_mix")) return g_tune_lds_mix;
    if (!strcmp(key, "lds_dbg")) return g_tune_lds_dbg;
    return -1;
}

struct SpmmArgs {
    const int32_t* rowptr;
    const int32_t* col;
    const float* val;
    const sgcn_seg_t* seg;   // nullable: implicit one segment per row
    int64_t nseg;
    int64_t nsegblk;         // blocks per slab
    const float* B;
    int64_t ldb;
    const int32_t* gidx;
    const float* rscale;
    const float* cscale;
    float* C;
    int64_t ldc;
    float beta;
    int32_t d;
    int32_t nvec;            // ceil(d / VW)
    float* ws;
    int64_t ldw;
    int32_t slabmajor;
    const float* add;        // optional addend: C[row] += add[row] for row < add_rows
    int64_t ldadd;
    int32_t add_rows;
};

template <int VW>
__device__ __forceinline__ void epilogue_store(const SpmmArgs& a, int row, int vi,
                                               typename Vec<VW>::type acc, float rs) {
    float* out = a.C + (int64_t)row * a.ldc + (int64_t)vi * VW;
    typename Vec<VW>::type r = acc * rs;
    const int left = a.d - vi * VW;   // > 0 by construction
    if (a.beta != 0.f) {
        if (left >= VW) r += a.beta * vload<VW>(out);
        else {
#pragma unroll
            for (int e = 0; e < VW; e++)
                if (e < left) {
                    if constexpr (VW == 1) r += a.beta * out[0]; else r[e] += a.beta * out[e];
                }
        }
    }
    if (a.add && row < a.add_rows) {     // e.g. the self term of a concat aggregator's backward
        const float* ad = a.add + (int64_t)row * a.ldadd + (int64_t)vi * VW;
        if (left >= VW) r += vload<VW>(ad);
        else {
#pragma unroll
            for (int e = 0; e < VW; e++)
                if (e < left) {
                    if constexpr (VW == 1) r += ad[0]; else r[e] += ad[e];
                }
        }
    }
    if (left >= VW) vstore<VW>(out, r);
    else vstore_head<VW>(out, r, left);
}

template <int G, int NV, int VW, int U>
__global__ __launch_bounds__(kBlock) void spmm_seg_kernel(SpmmArgs a) {
    typedef typename Vec<VW>::type VT;
    constexpr int GPB = kBlock / G;
    const int lig = threadIdx.x & (G - 1);
    const int gib = threadIdx.x / G;

    int slab;
    int64_t sblk;
    if (a.slabmajor) { slab = (int)(blockIdx.x / a.nsegblk); sblk = blockIdx.x % a.nsegblk; }
    else { const int nslab = (int)(gridDim.x / a.nsegblk); slab = blockIdx.x % nslab; sblk = blockIdx.x / nslab; }
    const int64_t s = sblk * GPB + gib;
    if (s >= a.nseg) return;

    int row, start, end, slot;
    if (a.seg) {
        const sgcn_seg_t sg = a.seg[s];
        row = sg.row; start = sg.start; end = sg.end; slot = sg.slot;
    } else {
        row = (int)s; start = a.rowptr[s]; end = a.rowptr[s + 1]; slot = -1;
    }
    row = uniform_i<G>(row); start = uniform_i<G>(start);
    end = uniform_i<G>(end); slot = uniform_i<G>(slot);

    const int vbase = slab * (G * NV) + lig;
    bool act[NV];
    VT acc[NV];
    // Lanes past the row width re-read the last valid vector instead of branching around the
    // load (their accumulators are never stored): the gather loop stays branch-free.
    uint32_t loff[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int vi = vbase + k * G;
        act[k] = vi < a.nvec;
        acc[k] = vzero<VW>();
        loff[k] = (uint32_t)min(vi, a.nvec - 1) * (uint32_t)(VW * sizeof(float));
    }
    const char* Bb = reinterpret_cast<const char*>(a.B);
    const int64_t ldb_bytes = a.ldb * (int64_t)sizeof(float);

    for (int p0 = start; p0 < end; p0 += G) {
        const int n = min(G, end - p0);
        int mycol = 0;
        float myval = 0.f;
        if (lig < n) {
            mycol = a.col[p0 + lig];
            myval = a.val[p0 + lig];
            if (a.cscale) myval *= a.cscale[mycol];
            if (a.gidx) mycol = a.gidx[mycol];
        }
        int j = 0;
        for (; j + U <= n; j += U) {
            VT b[U][NV];
            float v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int c = bcast_i<G>(mycol, j + u);
                v[u] = bcast_f<G>(myval, j + u);
                const char* src = Bb + (int64_t)c * ldb_bytes;   // G == 64: scalar base
#pragma unroll
                for (int k = 0; k < NV; k++)
                    b[u][k] = *reinterpret_cast<const VT*>(src + loff[k]);
            }
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int k = 0; k < NV; k++) acc[k] += v[u] * b[u][k];
        }
        for (; j < n; j++) {
            const int c = bcast_i<G>(mycol, j);
            const float v = bcast_f<G>(myval, j);
            const char* src = Bb + (int64_t)c * ldb_bytes;
#pragma unroll
            for (int k = 0; k < NV; k++) acc[k] += v * *reinterpret_cast<const VT*>(src + loff[k]);
        }
    }

    if (slot < 0) {
        const float rs = a.rscale ? a.rscale[row] : 1.0f;
#pragma unroll
        for (int k = 0; k < NV; k++)
            if (act[k]) epilogue_store<VW>(a, row, vbase + k * G, acc[k], rs);
    } else {
        float* w = a.ws + (int64_t)slot * a.ldw + (int64_t)vbase * VW;
#pragma unroll
        for (int k = 0; k < NV; k++)
            if (act[k]) vstore<VW>(w + (int64_t)k * G * VW, acc[k]);
    }
}

// Ordered sum of the workspace slots of every split row, then the same epilogue.
template <int VW>
__global__ __launch_bounds__(kBlock) void spmm_fix_kernel(SpmmArgs a, const sgcn_fix_t* fix,
                                                          int64_t nfix) {
    typedef typename Vec<VW>::type VT;
    const int nvblk = (a.nvec + kBlock - 1) / kBlock;
    const int64_t f = blockIdx.x / nvblk;
    if (f >= nfix) return;
    const sgcn_fix_t fx = fix[f];
    const int vi = (int)(blockIdx.x % nvblk) * kBlock + threadIdx.x;
    if (vi >= a.nvec) return;
    const float* w = a.ws + (int64_t)fx.first_slot * a.ldw + (int64_t)vi * VW;
    VT acc = vzero<VW>();
    for (int q = 0; q < fx.nslots; q++) acc += vload<VW>(w + (int64_t)q * a.ldw);
    const float rs = a.rscale ? a.rscale[fx.row] : 1.0f;
    epilogue_store<VW>(a, fx.row, vi, acc, rs);
}

template <int G, int NV, int VW, int U>
static void launch_seg(const SpmmArgs& a, int64_t nblocks, hipStream_t st) {
    hipLaunchKernelGGL((spmm_seg_kernel<G, NV, VW, U>), dim3((unsigned)nblocks), dim3(kBlock), 0, st, a);
}

template <int VW, int U>
static bool dispatch_gnv(int G, int NV, const SpmmArgs& a, int64_t nblocks, hipStream_t st) {
    switch (G) {
        case 8: launch_seg<8, 1, VW, U>(a, nblocks, st); return true;
        case 16: launch_seg<16, 1, VW, U>(a, nblocks, st); return true;
        case 32: launch_seg<32, 1, VW, U>(a, nblocks, st); return true;
        case 64:
            switch (NV) {
                case 1: launch_seg<64, 1, VW, U>(a, nblocks, st); return true;
                case 2: launch_seg<64, 2, VW, U>(a, nblocks, st); return true;
                case 3: launch_seg<64, 3, VW, U>(a, nblocks, st); return true;
                case 4: launch_seg<64, 4, VW, U>(a, nblocks, st); return true;
            }
    }
    return false;
}

template <int VW>
static bool dispatch_u(int U, int G, int NV, const SpmmArgs& a, int64_t nblocks, hipStream_t st) {
    switch (U) {
        case 2: return dispatch_gnv<VW, 2>(G, NV, a, nblocks, st);
        case 4: return dispatch_gnv<VW, 4>(G, NV, a, nblocks, st);
        case 8: return dispatch_gnv<VW, 8>(G, NV, a, nblocks, st);
    }
    return false;
}

// Geometry shared with the aggregator kernels: lanes per row group for `nvec` vectors.
int group_lanes(int nvec) {
    if (nvec <= 8) return 8;
    if (nvec <= 16) return 16;
    if (nvec <= 32) return 32;
    return 64;
}

}  // namespace sgcn

using namespace sgcn;

extern "C" int64_t sgcn_tune_get(const char* key) { return key ? tune_get(key) : -1; }

extern "C" int sgcn_tune(const char* key, int64_t value) {
    if (!key) return fail(SGCN_ERR_INVALID, "sgcn_tune: null key");
    if (!strcmp(key, "spmm_nv")) { SGCN_REQUIRE(value >= 0 && value <= 4, "spmm_nv in 0..4"); g_tune_nv = (int)value; return SGCN_OK; }
    if (!strcmp(key, "spmm_unroll")) {
        SGCN_REQUIRE(value == 0 || value == 2 || value == 4 || value == 8, "spmm_unroll in {0,2,4,8}");
        g_tune_unroll = (int)value; return SGCN_OK;
    }
    if (!strcmp(key, "spmm_slabmajor")) { g_tune_slabmajor = value != 0; return SGCN_OK; }
    if (!strcmp(key, "cs_pace")) { SGCN_REQUIRE(value >= 0, "cs_pace >= 0"); g_tune_cs_pace = (int)value; return SGCN_OK; }
    if (!strcmp(key, "cs_slack")) { SGCN_REQUIRE(value >= 0, "cs_slack >= 0"); g_tune_cs_slack = (int)value; return SGCN_OK; }
    if (!strcmp(key, "cs_noextra")) { g_tune_cs_noextra = value != 0; return SGCN_OK; }
    if (!strcmp(key, "step_overlap")) { g_tune_step_overlap = value != 0; return SGCN_OK; }
    if (!strcmp(key, "step_fuse")) { SGCN_REQUIRE(value >= 0 && value <= 127, "step_fuse in 0..127"); g_tune_step_fuse = (int)value; return SGCN_OK; }
    if (!strcmp(key, "cs_g2_wide")) { g_tune_cs_g2_wide = value != 0; return SGCN_OK; }
    if (!strcmp(key, "cs_nowarp")) { g_tune_cs_nowarp = value != 0; return SGCN_OK; }
    if (!strcmp(key, "gemm_min_steps")) { g_tune_gemm_min_steps = (int)value; return SGCN_OK; }
    if (!strcmp(key, "lds_wave_bias")) { SGCN_REQUIRE(value >= 50 && value <= 300, "lds_wave_bias in 50..300 (per cent)"); g_tune_lds_wave_bias = (int)value; return SGCN_OK; }
    if (!strcmp(key, "lds_mix")) { g_tune_lds_mix = value != 0; return SGCN_OK; }
    if (!strcmp(key, "lds_dbg")) { g_tune_lds_dbg = (int)value; return SGCN_OK; }
    if (!strcmp(key, "cs_last_pct")) { SGCN_REQUIRE(value >= 0 && value <= 100, "cs_last_pct in [0, 100]"); g_tune_cs_last_pct = (int)value; return SGCN_OK; }
    if (!strcmp(key, "cs_round")) { SGCN_REQUIRE(value >= 0, "cs_round >= 0"); g_tune_cs_round = (int)value; return SGCN_OK; }
    if (!strcmp(key, "cs_unroll")) {
        SGCN_REQUIRE(value == 0 || value == 4 || value == 8, "cs_unroll in {0,4,8}");
        g_tune_cs_unroll = (int)value; return SGCN_OK;
    }
    return fail(SGCN_ERR_INVALID, "sgcn_tune: unknown key '%s'", key);
}

extern "C" int sgcn_spmm_csr_add_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                     int32_t M, int32_t K, int32_t d, const float* B, int64_t ldb,
                                     const int32_t* gidx, const float* rscale, const float* cscale,
                                     float* C, int64_t ldc, float beta, const sgcn_plan_t* plan,
                                     const float* add, int64_t ldadd, int32_t add_rows, void* stream);

extern "C" int sgcn_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                 int32_t M, int32_t K, int32_t d, const float* B, int64_t ldb,
                                 const int32_t* gidx, const float* rscale, const float* cscale,
                                 float* C, int64_t ldc, float beta, const sgcn_plan_t* plan,
                                 void* stream) {
    return sgcn_spmm_csr_add_f32(rowptr, col, val, M, K, d, B, ldb, gidx, rscale, cscale, C, ldc, beta, plan,
                                 nullptr, 0, 0, stream);
}

extern "C" int sgcn_spmm_csr_add_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                     int32_t M, int32_t K, int32_t d, const float* B, int64_t ldb,
                                     const int32_t* gidx, const float* rscale, const float* cscale,
                                     float* C, int64_t ldc, float beta, const sgcn_plan_t* plan,
                                     const float* add, int64_t ldadd, int32_t add_rows, void* stream) {
    SGCN_REQUIRE(M >= 0 && K >= 0 && d >= 0, "spmm: negative size");
    if (M == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(rowptr && C && (B || K == 0), "spmm: null operand");
    SGCN_REQUIRE(ldb >= d && ldc >= d, "spmm: leading dimension smaller than d");
    hipStream_t st = (hipStream_t)stream;

    SpmmArgs a{};
    a.rowptr = rowptr; a.col = col; a.val = val;
    a.B = B; a.ldb = ldb; a.gidx = gidx; a.rscale = rscale; a.cscale = cscale;
    a.C = C; a.ldc = ldc; a.beta = beta; a.d = d;
    a.nseg = M;
    a.slabmajor = g_tune_slabmajor;
    if (add && add_rows > 0) {
        SGCN_REQUIRE(ldadd >= d && add_rows <= M, "spmm: bad addend");
        a.add = add; a.ldadd = ldadd; a.add_rows = add_rows;
    }
    if (plan) {
        SGCN_REQUIRE(plan->dev_seg && plan->nseg >= M, "spmm: malformed plan");
        a.seg = plan->dev_seg; a.nseg = plan->nseg;
        a.ws = plan->dev_ws; a.ldw = ((int64_t)d + 3) / 4 * 4;
        if (plan->nfix > 0) {
            SGCN_REQUIRE(plan->dev_fix && plan->dev_ws, "spmm: plan with split rows needs dev_fix/dev_ws");
            SGCN_REQUIRE(plan->ws_elems >= plan->nslots * a.ldw,
                         "spmm: workspace too small (%lld < %lld floats)", (long long)plan->ws_elems,
                         (long long)(plan->nslots * a.ldw));
        }
    }
    const int vw = pick_vw(d, {B, C, plan ? plan->dev_ws : nullptr, a.add}, {ldb, ldc, a.add ? a.ldadd : ldc});
    a.nvec = (d + vw - 1) / vw;
    const int G = group_lanes(a.nvec);
    int NV = 1;
    if (G == 64) {
        const int per_wave = (a.nvec + 63) / 64;          // vectors per lane to cover the row
        const int nslab0 = (per_wave + 3) / 4;            // slabs at the NV<=4 cap
        NV = (per_wave + nslab0 - 1) / nslab0;
        if (g_tune_nv > 0) NV = g_tune_nv;
    }
    const int nslab = (a.nvec + G * NV - 1) / (G * NV);
    const int U = g_tune_unroll > 0 ? g_tune_unroll : (NV >= 3 ? 4 : 8);
    a.nsegblk = (a.nseg + (kBlock / G) - 1) / (kBlock / G);
    const int64_t nblocks = a.nsegblk * nslab;
    SGCN_REQUIRE(nblocks < (1ll << 31), "spmm: grid too large");

    bool ok = false;
    if (vw == 4) ok = dispatch_u<4>(U, G, NV, a, nblocks, st);
    else if (vw == 2) ok = dispatch_u<2>(U, G, NV, a, nblocks, st);
    else ok = dispatch_u<1>(U, G, NV, a, nblocks, st);
    SGCN_REQUIRE(ok, "spmm: no kernel for G=%d NV=%d U=%d", G, NV, U);
    SGCN_HIP_TRY(hipGetLastError());

    if (plan && plan->nfix > 0) {
        const int64_t nfblk = (int64_t)((a.nvec + kBlock - 1) / kBlock) * plan->nfix;
        SGCN_REQUIRE(nfblk < (1ll << 31), "spmm: too many split rows");
        dim3 grid((unsigned)nfblk);
        if (vw == 4) hipLaunchKernelGGL((spmm_fix_kernel<4>), grid, dim3(kBlock), 0, st, a, plan->dev_fix, plan->nfix);
        else if (vw == 2) hipLaunchKernelGGL((spmm_fix_kernel<2>), grid, dim3(kBlock), 0, st, a, plan->dev_fix, plan->nfix);
        else hipLaunchKernelGGL((spmm_fix_kernel<1>), grid, dim3(kBlock), 0, st, a, plan->dev_fix, plan->nfix);
        SGCN_HIP_TRY(hipGetLastError());
    }
    return SGCN_OK;
}
