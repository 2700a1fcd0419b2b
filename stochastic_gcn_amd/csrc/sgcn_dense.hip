// Fused row-wise kernels of the "downstream dense" part of the step,
// gfx950.  The GEMMs themselves stay on rocBLAS (matrix cores); what is fused here is the chain
// of small elementwise / row-reduction ops around them, because at minibatch sizes
// (~1,000 x 128 fp32) the step is launch-latency-bound, not bandwidth-bound:
//   ln_act_fwd   MyLayerNorm2 + ReLU            gcn/layers.py:95-97,134-137,404-411
//   ln_act_bwd   their backward (+ column sums for d(offset), d(scale))
//   softmax_ce   softmax-CE mean loss, accuracy, softmax "pred", dlogits   gcn/models.py:68-94,198-202
//   adam         tf.train.AdamOptimizer update on the flat parameter buffer   gcn/models.py:50-51
// One wavefront per row; lanes stride the row (coalesced 256-byte pieces); wave reductions by
// DPP shuffles (__shfl_xor over 64 lanes).
#include <algorithm>

#include "sgcn_dev.h"
#include "sgcn_fuse.h"

namespace sgcn {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// y = act(LN(x) * scale + offset); keeps xhat and rstd for the backward.  norm == 0: y = act(x).
__global__ __launch_bounds__(kBlock) void ln_act_fwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ offset,
    const float* __restrict__ scale, int32_t n, int32_t d, float eps, int32_t norm, int32_t relu,
    float* __restrict__ y, int64_t ldy, float* __restrict__ xhat, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (row >= n) return;
    const float* xr = x + row * ldx;
    float* yr = y + row * ldy;
    if (!norm) {
        for (int c = lane; c < d; c += kWave) { const float v = xr[c]; yr[c] = relu ? fmaxf(v, 0.f) : v; }
        return;
    }
    float s = 0.f;
    for (int c = lane; c < d; c += kWave) s += xr[c];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int c = lane; c < d; c += kWave) { const float t = xr[c] - mean; q += t * t; }
    const float r = rsqrtf(wave_sum(q) / (float)d + eps);
    if (lane == 0) rstd[row] = r;
    float* hr = xhat + row * (int64_t)d;
    for (int c = lane; c < d; c += kWave) {
        const float h = (xr[c] - mean) * r;
        hr[c] = h;
        const float v = h * scale[c] + offset[c];
        yr[c] = relu ? fmaxf(v, 0.f) : v;
    }
}

// dx = LN-backward(dy masked by y > 0); per-block column partials of d(offset), d(scale).
//
// Optional tail (LnBwdDx, d <= 128, K <= 256): the layer's INPUT gradient dxo[row][j] = sum_k g[row][k] W[j][k] (* the dropout
// mask of the layer's input) in the same row pass -- the wave keeps the row of g it has just produced in registers (two
// columns per lane), the K x d weight matrix is staged in LDS once per workgroup, transposed with an odd pitch so that both
// the staging writes and the per-lane reads are conflict-free, and lane j accumulates in exactly the order of the 32 x 128
// MFMA launch it replaces (32-wide K-steps alternating between its K-groups, the groups' sums added in group order): the
// same bits, one launch and one round trip of g through memory fewer (9.7 us of the step's chain, twice per step).
struct LnBwdDx { const float* W; int32_t K, kg; DropArgs drop; float* dxo; int64_t lddxo;
                 // dma (d = 32 | 64 | 128): the K x d matrix goes from global memory straight into the LDS
                 // (global_load_lds_dwordx4: no staging registers, no transposing stores), row-major as it lies, its 16-byte
                 // chunks XOR-swizzled within a row by the row index so that the lanes' 128-bit reads of a column block --
                 // lane j owns output row j -- fall into different banks
                 int32_t dma; };
// ... and, behind that tail, the LayerNorm / ReLU backward of the layer BELOW (its dy is the dx just produced, K <= 128 wide):
// g = LN-backward(dx masked by y > 0) written to `g`, per-workgroup parameter partials to `partial` -- the row pass of the
// next ln_act_bwd_kernel launch, without the launch.  y == nullptr: off.
struct LnBwdNext { const float* y; int64_t ldy; const float* xhat; const float* rstd; const float* scale; int32_t norm, relu;
                   float* g; float* partial; };

// The LayerNorm backward's arithmetic, ONE expression tree for its two users (a layer's own row pass and the pass chained
// behind the layer above): contraction is off and the fused multiply-adds are spelled out, so that the compiler cannot
// fuse the two sites differently (it did: reusing g * scale in one and fusing it into the subtraction in the other --
// one ulp, which the bit-identity of the chained and the unchained step would not survive).
__device__ __forceinline__ void ln_bwd_stats(float g, float h, float sc, float& doff, float& dsc, float& s1, float& s2) {
#pragma clang fp contract(off)
    doff += g;                        // d(offset) column sum
    dsc = fmaf(g, h, dsc);            // d(scale)
    const float t2 = g * sc;
    s1 += t2;
    s2 = fmaf(t2, h, s2);
}
__device__ __forceinline__ float ln_bwd_out(float g, float h, float sc, float m1, float m2, float r) {
#pragma clang fp contract(off)
    return r * fmaf(-h, m2, fmaf(g, sc, -m1));
}

constexpr int kBwdRowsPerWave = 1;     // one row per wave: 4x the workgroups, a quarter of the dependent chain (the step is GPU-latency-bound)
__global__ __launch_bounds__(kBlock) void ln_act_bwd_kernel(
    const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y, int64_t ldy,
    const float* __restrict__ xhat, const float* __restrict__ rstd, const float* __restrict__ scale,
    int32_t n, int32_t d, int32_t norm, int32_t relu, float* __restrict__ dx, int64_t lddx,
    float* __restrict__ partial /* [gridDim.x][2][d] */, LnBwdDx t, LnBwdNext nx) {
    extern __shared__ float lds[];      // [4 waves][2][d] (LayerNorm), the tail's weights [d][K + 1], [4 waves][2][K] (the layer below)
    const int lane = threadIdx.x & 63, wave = threadIdx.x / kWave;
    const int64_t row0 = ((int64_t)blockIdx.x * (kBlock / kWave) + wave) * kBwdRowsPerWave;
    float* my = lds + (size_t)wave * 2 * d;
    float* wt = lds + (norm ? (size_t)8 * d : 0);
    const int kp = t.K + 1;
    const int wregion = t.dma ? (t.K * d + 255) / 256 * 256 : d * kp;      // floats
    if (t.W && t.dma) {
        // LDS slot s (16 bytes) = row j = s / P, chunk s % P of the row  <-  chunk (s % P) ^ (j % P) of W's row j   (P = d / 4)
        const int P = d >> 2, sh = __ffs(P) - 1, total4 = t.K * P;
        for (int base4 = wave * kWave; base4 < total4; base4 += kBlock) {
            const int sl = min(base4 + lane, total4 - 1);                // (past the end: into the region's padding)
            const int j = sl >> sh, src4 = (j << sh) + ((sl & (P - 1)) ^ (j & (P - 1)));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t.W + (int64_t)src4 * 4),
                                             (__attribute__((address_space(3))) void*)(wt + (int64_t)base4 * 4), 16, 0, 0);
        }
    } else if (t.W) {                   // stage W^T: element i = j * d + k of the contiguous K x d matrix -> wt[k][j]
        // every round of loads is a trip past the L2 (~2 us), so a thread requests its whole share -- up to 32 float4 for
        // the 256 x 128 matrix -- before it stores any (d % 4 == 0 and a 16-byte aligned matrix: checked by the host)
        const int total4 = t.K * d / 4;
        const float inv_d = 1.0f / (float)d;
        constexpr int kU = 32;
        float4 v[kU];
        const float4* W4 = reinterpret_cast<const float4*>(t.W);
#pragma unroll
        for (int u = 0; u < kU; u++) { const int i4 = threadIdx.x + u * kBlock; if (i4 < total4) v[u] = W4[i4]; }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int i4 = threadIdx.x + u * kBlock;
            if (i4 < total4) {
                const int i = i4 * 4;
                const int j = (int)(((float)i + 0.5f) * inv_d), k = i - j * d;       // exact: i < 2^15, d <= 128
                float* w = wt + k * kp + j;
                w[0] = v[u].x; w[kp] = v[u].y; w[2 * kp] = v[u].z; w[3 * kp] = v[u].w;
            }
        }
    }
    if (norm) for (int c = lane; c < 2 * d; c += kWave) my[c] = 0.f;
    float gk[2] = {0.f, 0.f};           // the row of g this wave produced (tail: d <= 128)
    bool have = false;
    int64_t myrow = 0;
    for (int k = 0; k < kBwdRowsPerWave; k++) {
        const int64_t row = row0 + k;
        if (row >= n) break;
        have = true; myrow = row;
        const float* gr = dy + row * lddy;
        const float* yr = y + row * ldy;
        float* dr = dx + row * lddx;
        if (!norm) {
            for (int c = lane; c < d; c += kWave) {
                const float g = (relu && !(yr[c] > 0.f)) ? 0.f : gr[c];
                dr[c] = g;
                if (c < 2 * kWave) gk[c / kWave] = g;
            }
            continue;
        }
        const float* hr = xhat + row * (int64_t)d;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < d; c += kWave) {
            const float g = (relu && !(yr[c] > 0.f)) ? 0.f : gr[c];
            ln_bwd_stats(g, hr[c], scale[c], my[c], my[d + c], s1, s2);      // lane-private columns: no race
        }
        const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d, r = rstd[row];
        for (int c = lane; c < d; c += kWave) {
            const float g = (relu && !(yr[c] > 0.f)) ? 0.f : gr[c];
            const float o = ln_bwd_out(g, hr[c], scale[c], m1, m2, r);
            dr[c] = o;
            if (c < 2 * kWave) gk[c / kWave] = o;
        }
    }
    if (!norm && !t.W) return;
    if (t.dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (norm) {
        float* out = partial + (size_t)blockIdx.x * 2 * d;
        for (int c = threadIdx.x; c < 2 * d; c += kBlock)
            out[c] = (lds[c] + lds[2 * d + c]) + (lds[4 * d + c] + lds[6 * d + c]);
    }
    if (!t.W || (!have && !nx.y)) return;
    const int d2 = t.K;                                  // (nx) the layer below is d2 wide
    float* my2 = lds + (norm ? (size_t)8 * d : 0) + (size_t)wregion + (size_t)wave * 2 * d2;
    if (nx.y && nx.norm) for (int c = lane; c < 2 * d2; c += kWave) my2[c] = 0.f;
    float dyl[2] = {0.f, 0.f};
    if (have) {
    // ---- the tail: four output columns per lane
    float acc[4][2];
#pragma unroll
    for (int e = 0; e < 4; e++) { acc[e][0] = 0.f; acc[e][1] = 0.f; }
    int jj[4];
#pragma unroll
    for (int e = 0; e < 4; e++) jj[e] = lane + e * kWave < t.K ? lane + e * kWave : 0;
    int rowf[4], sw[4];                 // (dma) the lane's four rows: first float of the row, its swizzle
    const int Pm = (d >> 2) - 1;
#pragma unroll
    for (int e = 0; e < 4; e++) { rowf[e] = jj[e] * d; sw[e] = jj[e] & Pm; }
    for (int s0 = 0; s0 * 32 < d; s0++) {
        const int gi = t.kg > 1 ? (s0 & 1) : 0;
        const float xs = s0 < 2 ? gk[0] : gk[1];
        const int kb = s0 * 32, lb = kb & (kWave - 1);
        const int kn = min(32, d - kb);
        const float* wk = wt + kb * kp;
        float p[4];
#pragma unroll
        for (int e = 0; e < 4; e++) p[e] = acc[e][gi];
        if (t.dma) {                    // d % 32 == 0: whole K-steps, eight 16-byte chunks each, k ascending as below
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int c = (kb >> 2) + q;
                float4 w[4];
#pragma unroll
                for (int e = 0; e < 4; e++) w[e] = *reinterpret_cast<const float4*>(wt + rowf[e] + ((c ^ sw[e]) << 2));
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), lb + q * 4 + u));
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        p[e] = fmaf(xv, u == 0 ? w[e].x : u == 1 ? w[e].y : u == 2 ? w[e].z : w[e].w, p[e]);
                }
            }
        } else if (kn == 32) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float w[8][4];
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int e = 0; e < 4; e++) w[u][e] = wk[(q * 8 + u) * kp + jj[e]];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), lb + q * 8 + u));
#pragma unroll
                    for (int e = 0; e < 4; e++) p[e] = fmaf(xv, w[u][e], p[e]);
                }
            }
        } else {
            for (int kk = 0; kk < kn; kk++) {
                const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), lb + kk));
#pragma unroll
                for (int e = 0; e < 4; e++) p[e] = fmaf(xv, wk[kk * kp + jj[e]], p[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e][gi] = p[e];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int j = lane + e * kWave;
        if (j < t.K) {
            float v = t.kg > 1 ? acc[e][0] + acc[e][1] : acc[e][0];
            if (t.drop.on) v *= drop_factor(t.drop, (int)myrow, j);
            t.dxo[myrow * t.lddxo + j] = v;
            if (e < 2) dyl[e] = v;
        }
    }
    }
    if (!nx.y) return;
    // ---- the layer below: ln_act_bwd_kernel's row pass on dy = dyl (two columns per lane, in its order of additions)
    if (have) {
        const float* yr = nx.y + myrow * nx.ldy;
        float* gr = nx.g + myrow * (int64_t)d2;
        if (!nx.norm) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int c = lane + e * kWave;
                if (c < d2) gr[c] = (nx.relu && !(yr[c] > 0.f)) ? 0.f : dyl[e];
            }
        } else {
            const float* hr = nx.xhat + myrow * (int64_t)d2;
            float s1 = 0.f, s2 = 0.f, gg[2] = {0.f, 0.f}, hh[2] = {0.f, 0.f}, sc[2] = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int c = lane + e * kWave;
                if (c < d2) {
                    gg[e] = (nx.relu && !(yr[c] > 0.f)) ? 0.f : dyl[e];
                    hh[e] = hr[c]; sc[e] = nx.scale[c];
                    ln_bwd_stats(gg[e], hh[e], sc[e], my2[c], my2[d2 + c], s1, s2);
                }
            }
            const float m1 = wave_sum(s1) / (float)d2, m2 = wave_sum(s2) / (float)d2, r = nx.rstd[myrow];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int c = lane + e * kWave;
                if (c < d2) gr[c] = ln_bwd_out(gg[e], hh[e], sc[e], m1, m2, r);
            }
        }
    }
    if (!nx.norm) return;
    __syncthreads();
    const float* l2 = lds + (norm ? (size_t)8 * d : 0) + (size_t)wregion;
    float* out2 = nx.partial + (size_t)blockIdx.x * 2 * d2;
    for (int c = threadIdx.x; c < 2 * d2; c += kBlock)
        out2[c] = (l2[c] + l2[2 * d2 + c]) + (l2[4 * d2 + c] + l2[6 * d2 + c]);
}

// doffset[c] += sum_b partial[b][0][c]; dscale[c] += sum_b partial[b][1][c]   (fixed order)
__global__ void ln_param_reduce_kernel(const float* __restrict__ partial, int32_t nblk, int32_t d,
                                       float* __restrict__ doffset, float* __restrict__ dscale) {
    ln_param_reduce_cols32(partial, nblk, d, doffset, dscale, (int)blockIdx.x);
}

// The input gradient of the LAST dense layer, dx = dlogits . W^T (no LayerNorm, no ReLU behind the logits), as a tail of
// the loss kernel's row pass: the wave that has just produced a row's dlogits (one per lane, c <= 64) multiplies them into
// the K x c weight matrix -- 2 x 41 fused multiply-adds per lane on a 512 x 41 batch -- instead of a 16-workgroup MFMA launch
// of its own (7.8 us of the step's chain for 5 MFLOP).  k ascending from zero in one fmaf chain: the same bits as the GEMM
// (sgcn_gemm.hip: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain), dropout mask of the layer's input included.
struct CeDx { const float* W; int64_t ldw; int32_t K, wflat; float* dx; int64_t lddx; DropArgs drop;
              // ... and the last layer's FORWARD as a head of the same pass: logits = dropout(x) . W for x = hx[row][0..K),
              // K <= 128, with the K-step / K-group order of additions of the launch it replaces (kg groups of alternating
              // 32-wide K-steps, partial sums added in group order: sgcn_gemm.hip gemm_body)
              const float* hx; int64_t ldhx; int32_t kg; DropArgs hdrop; float* zout; int64_t ldzo;
              // ... and the dense layer in front of it as a PRE-layer of the head (sgcn_fuse.h CeLastLayer): the head's
              // input row is computed here -- dropout(px)[row][0..PK) . PW, LayerNorm, ReLU -- and stays in registers
              const float* px; int64_t ldpx; int32_t PK, pS, pkchunk, pkg; const float* PW; DropArgs pdrop;
              const float* poff; const float* psc; float peps; int32_t prelu, pepi;
              float* pY; int64_t ldpy; float* pxhat; float* prstd; };

// W is staged in LDS by the whole workgroup first (coalesced; lane j then reads its row W[j][0..c) with stride c floats --
// odd for the class counts that occur, so conflict-free -- instead of 64 different cache lines per load instruction)
// LDS layout of the loss kernel's weights: the output layer's [K][c] matrix in a region rounded up to whole 1 KB load
// instructions, the pre-layer's [PK][K] matrix behind it
__host__ __device__ inline int ce_w_region(int K, int c) { return (K * c + 255) / 256 * 256; }

__device__ __forceinline__ void ce_dx_stage(const CeDx& t, int c, float* wl) {
    // The matrices were written by the previous step's optimizer: every round of loads is a trip past the L2 (~2 us), so
    // ALL of them are requested before anything waits -- as direct global -> LDS loads (global_load_lds_dwordx4: 64 lanes x
    // 16 bytes land contiguously at a wave-uniform LDS address), which costs no registers however large the matrix is.
    // A lane past the end of a matrix re-reads its last 16 bytes into the region's padding.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto copy = [&](const float* src, int floats, float* dst) {
        const int total4 = floats / 4;
        for (int base4 = wave * kWave; base4 < total4; base4 += (kBlock / kWave) * kWave) {
            const int i4 = min(base4 + lane, total4 - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)i4 * 4),
                                             (__attribute__((address_space(3))) void*)(dst + (int64_t)base4 * 4), 16, 0, 0);
        }
    };
    const int total = t.K * c;
    if (t.wflat) copy(t.W, total, wl);          // (host: contiguous, 16-byte aligned, a multiple of four floats)
    else                                    // (a pitched or unaligned matrix: element by element -- not on the step's path)
        for (int i = threadIdx.x; i < total; i += kBlock) wl[i] = t.W[(int64_t)(i / c) * t.ldw + (i % c)];
    if (t.PW) copy(t.PW, t.PK * t.K, wl + ce_w_region(t.K, c));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// The pre-layer for one row: returns the head's input (columns lane and lane + 64 of act(LN(dropout(px) . PW))) with the
// additions in the order of the launch(es) it replaces -- per K slice z (split-K), K-groups of alternating 32-wide steps,
// the groups' sums added in group order, the slices' sums added in z order starting from zero -- and stores y / xhat / rstd.
__device__ __forceinline__ void ce_pre_layer(const CeDx& t, const float* pw, int64_t row, int lane, float xin[4], float out[2]) {
    if (t.pdrop.on) {
#pragma unroll
        for (int e = 0; e < 4; e++) xin[e] *= drop_factor(t.pdrop, (int)row, lane + e * kWave);
    }
    const int N = t.K;                      // the pre-layer's outputs = the output layer's inputs
    const int c0 = lane < N ? lane : 0, c1 = lane + kWave < N ? lane + kWave : 0;
    float v0 = 0.f, v1 = 0.f;
    for (int z = 0; z < t.pS; z++) {
        const int kbeg = z * t.pkchunk, kend = min(t.PK, kbeg + t.pkchunk);
        float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
        for (int kb = kbeg, sl = 0; kb < kend; kb += 32, sl++) {
            const int gi = t.pkg > 1 ? (sl & 1) : 0;
            const int blk = kb >> 6, lb = kb & (kWave - 1);
            const float xs = blk == 0 ? xin[0] : (blk == 1 ? xin[1] : (blk == 2 ? xin[2] : xin[3]));
            const float* wk = pw + kb * N;
            float p0 = a0[gi], p1 = a1[gi];
            if (kb + 32 <= kend) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float w0[8], w1[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { w0[u] = wk[(q * 8 + u) * N + c0]; w1[u] = wk[(q * 8 + u) * N + c1]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), lb + q * 8 + u));
                        p0 = fmaf(xv, w0[u], p0); p1 = fmaf(xv, w1[u], p1);
                    }
                }
            } else {
                for (int kk = 0; kb + kk < kend; kk++) {
                    const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), lb + kk));
                    p0 = fmaf(xv, wk[kk * N + c0], p0); p1 = fmaf(xv, wk[kk * N + c1], p1);
                }
            }
            a0[gi] = p0; a1[gi] = p1;
        }
        const float s0 = t.pkg > 1 ? a0[0] + a0[1] : a0[0], s1 = t.pkg > 1 ? a1[0] + a1[1] : a1[0];
        if (t.pS > 1) { v0 += s0; v1 += s1; } else { v0 = s0; v1 = s1; }
    }
    const float v[2] = {v0, v1};
    float* yr = t.pY + row * t.ldpy;
    if (t.pepi == 2) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 2; e++) if (lane + e * kWave < N) s += v[e];
        const float mean = wave_sum(s) / (float)N;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 2; e++) if (lane + e * kWave < N) { const float d = v[e] - mean; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / (float)N + t.peps);
        if (lane == 0) t.prstd[row] = rs;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = lane + e * kWave;
            out[e] = 0.f;
            if (c < N) {
                const float h = (v[e] - mean) * rs;
                t.pxhat[row * N + c] = h;
                const float y = h * t.psc[c] + t.poff[c];
                out[e] = t.prelu ? fmaxf(y, 0.f) : y;
                yr[c] = out[e];
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = lane + e * kWave;
            out[e] = 0.f;
            if (c < N) { out[e] = (t.pepi == 1 && t.prelu) ? fmaxf(v[e], 0.f) : v[e]; yr[c] = out[e]; }
        }
    }
}
// lane cc < c returns logit cc of `row`
__device__ __forceinline__ float ce_head(const CeDx& t, const float* wl, int64_t row, int lane, int c, float x0, float x1) {
    if (t.hdrop.on) { x0 *= drop_factor(t.hdrop, (int)row, lane); x1 *= drop_factor(t.hdrop, (int)row, lane + kWave); }
    const int cc = lane < c ? lane : 0;
    float acc[2] = {0.f, 0.f};
    for (int s0 = 0; s0 * 32 < t.K; s0++) {
        const int g = t.kg > 1 ? (s0 & 1) : 0;
        float a = acc[g];
        const float xs = s0 < 2 ? x0 : x1;          // a 32-wide K-step never straddles the two halves of the row
        const int kb = s0 * 32, lb = kb & (kWave - 1);
        const float* wk = wl + kb * c + cc;
        if (kb + 32 <= t.K) {                         // a whole K-step: straight-line
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float w[8];
#pragma unroll
                for (int u = 0; u < 8; u++) w[u] = wk[(q * 8 + u) * c];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    a = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), lb + q * 8 + u)), w[u], a);
            }
        } else {                                      // the ragged last one
            for (int kk = 0; kb + kk < t.K; kk++)
                a = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), lb + kk)), wk[kk * c], a);
        }
        acc[g] = a;
    }
    const float z = t.kg > 1 ? acc[0] + acc[1] : acc[0];
    if (lane < c && t.zout) t.zout[row * t.ldzo + lane] = z;
    return z;
}

__device__ __forceinline__ void ce_dx_tail(const CeDx& t, const float* wl, int64_t row, int lane, int c, float mydz) {
    // k is wave-uniform: the dlogit comes through v_readlane (a scalar operand), not the LDS crossbar; the weights of eight
    // k are read before they are used so that the LDS latency is paid once per eight, and two output columns run side by side
    auto dzk = [&](int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mydz), k)); };
    for (int j0 = 0; j0 < t.K; j0 += 2 * kWave) {
        const int ja = j0 + lane, jb = j0 + kWave + lane;
        const bool oka = ja < t.K, okb = jb < t.K;
        const float* wa = wl + (oka ? ja : 0) * c;
        const float* wb = wl + (okb ? jb : 0) * c;
        float acca = 0.f, accb = 0.f;
        int k = 0;
        for (; k + 8 <= c; k += 8) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { a[u] = wa[k + u]; b[u] = wb[k + u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const float d = dzk(k + u); acca = fmaf(d, a[u], acca); accb = fmaf(d, b[u], accb); }
        }
        for (; k < c; k++) { const float d = dzk(k); acca = fmaf(d, wa[k], acca); accb = fmaf(d, wb[k], accb); }
        if (t.drop.on) { acca *= drop_factor(t.drop, (int)row, oka ? ja : 0); accb *= drop_factor(t.drop, (int)row, okb ? jb : 0); }
        if (oka) t.dx[row * t.lddx + ja] = acca;
        if (okb) t.dx[row * t.lddx + jb] = accb;
    }
}

// One wavefront per row, as many workgroups as rows need (a single-workgroup version spent
// 277 us on 512 x 41 logits: ~40 dependent cross-lane shuffles per row, 128 rows per wave).
// Per-row CE and hit flags go to `rowstat`; softmax_stats_kernel adds them in a fixed order
// (deterministic loss / accuracy).
__global__ __launch_bounds__(kBlock) void softmax_ce_kernel(
    const float* __restrict__ z, int64_t ldz, const float* __restrict__ lab, int64_t ldl, int32_t n,
    int32_t c, float* __restrict__ dz, int64_t lddz, float* __restrict__ pred, int64_t ldp,
    float* __restrict__ rowstat /* [2][n], [3][n] with pred */, CeDx tail) {
    extern __shared__ __attribute__((aligned(1024))) float ce_lds[];
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    const bool head = tail.hx != nullptr || tail.PW != nullptr;            // c <= 64: logit k lives in lane k
    // everything the row pass reads from memory is requested before the weight matrix is staged (one round trip, not four)
    const bool live = row < n;
    const float* lr = lab + (live ? row : 0) * ldl;
    const float lab0 = (live && lane < c) ? lr[lane] : 0.f;
    float hx0 = 0.f, hx1 = 0.f, pin[4] = {0.f, 0.f, 0.f, 0.f};
    if (head && live && !tail.PW) {
        const float* xr = tail.hx + row * tail.ldhx;
        hx0 = lane < tail.K ? xr[lane] : 0.f; hx1 = lane + kWave < tail.K ? xr[lane + kWave] : 0.f;
    }
    if (tail.PW && live) {
        const float* xr = tail.px + row * tail.ldpx;
#pragma unroll
        for (int e = 0; e < 4; e++) pin[e] = lane + e * kWave < tail.PK ? xr[lane + e * kWave] : 0.f;
    }
    if (tail.K > 0) ce_dx_stage(tail, c, ce_lds);
    if (!live) return;
    const float inv_n = 1.0f / (float)n;
    if (tail.PW) {
        float o[2];
        ce_pre_layer(tail, ce_lds + ce_w_region(tail.K, c), row, lane, pin, o);
        hx0 = o[0]; hx1 = o[1];
    }
    const float myz = head ? ce_head(tail, ce_lds, row, lane, c, hx0, hx1) : 0.f;
    const float* zr = z + row * ldz;
    auto zv = [&](int k) { return head ? myz : zr[k]; };
    auto lv = [&](int k) { return k < kWave ? lab0 : lr[k]; };     // k == lane in the first trip of every loop below
    float m = -INFINITY, lm = -INFINITY;
    int am = 0, alm = 0;
    for (int k = lane; k < c; k += kWave) {
        if (zv(k) > m) { m = zv(k); am = k; }
        if (lv(k) > lm) { lm = lv(k); alm = k; }
    }
    // wave arg-max with lowest-index tie break (np.argmax / tf.argmax semantics)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64); const int oa = __shfl_xor(am, o, 64);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
        const float ol = __shfl_xor(lm, o, 64); const int ola = __shfl_xor(alm, o, 64);
        if (ol > lm || (ol == lm && ola < alm)) { lm = ol; alm = ola; }
    }
    float se = 0.f, sl = 0.f;
    for (int k = lane; k < c; k += kWave) { se += __expf(zv(k) - m); sl += lv(k); }
    se = wave_sum(se); sl = wave_sum(sl);
    const float lse = m + __logf(se);
    float l = 0.f, mydz = 0.f, pm = -1.f;
    int apm = 0;
    for (int k = lane; k < c; k += kWave) {
        const float logp = zv(k) - lse, p = __expf(logp);
        l -= lv(k) * logp;
        mydz = (p * sl - lv(k)) * inv_n;
        if (dz) dz[row * lddz + k] = mydz;
        if (pred) { pred[row * ldp + k] = p; if (p > pm) { pm = p; apm = k; } }
    }
    l = wave_sum(l);
    if (lane == 0) { rowstat[row] = l; rowstat[n + row] = (am == alm) ? 1.f : 0.f; }
    if (pred) {
        // evaluation: the row's classes for the F1 scores (gcn/utils.py:521-529 takes np.argmax of the PREDICTION, i.e. of
        // the rounded probabilities, and of the labels): third plane of rowstat, prediction + 4096 * label (exact in fp32)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float op = __shfl_xor(pm, o, 64); const int oa = __shfl_xor(apm, o, 64);
            if (op > pm || (op == pm && oa < apm)) { pm = op; apm = oa; }
        }
        if (lane == 0) rowstat[2 * (int64_t)n + row] = (float)(apm + 4096 * alm);
    }
    if (tail.dx) ce_dx_tail(tail, ce_lds, row, lane, c, mydz);
}

// stats = {sum_i CE_i, #correct, mean CE, accuracy}: one workgroup, fixed summation order
__global__ __launch_bounds__(kBlock) void softmax_stats_kernel(const float* __restrict__ rowstat,
                                                               int32_t n, float* __restrict__ stats) {
    __shared__ float red[2][kBlock];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += kBlock) { a += rowstat[i]; b += rowstat[n + i]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] = red[0][0]; stats[1] = red[1][0];
        stats[2] = red[0][0] / (float)n; stats[3] = red[1][0] / (float)n;
    }
}

// Multitask loss (ppi): mean over all n*c elements of sigmoid cross-entropy with logits,
// max(z,0) - z*y + log1p(exp(-|z|)) (tf.nn.sigmoid_cross_entropy_with_logits, gcn/models.py:77-79);
// accuracy = mean((z > 0) == (y > 0.5)) (gcn/models.py:86-90); pred = sigmoid(z) (:198-200);
// dlogits = (sigmoid(z) - y) / (n*c).  One wave per row; per-row sums -> sigmoid_stats_kernel.
__global__ __launch_bounds__(kBlock) void sigmoid_ce_kernel(
    const float* __restrict__ z, int64_t ldz, const float* __restrict__ lab, int64_t ldl, int32_t n,
    int32_t c, float* __restrict__ dz, int64_t lddz, float* __restrict__ pred, int64_t ldp,
    float* __restrict__ rowstat /* [2][n] */, CeDx tail) {
    extern __shared__ __attribute__((aligned(1024))) float ce_lds[];
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    const bool head = tail.hx != nullptr || tail.PW != nullptr;
    const bool live = row < n;
    const float lab0 = (live && lane < c) ? lab[row * ldl + lane] : 0.f;      // requested before the weights are staged
    float hx0 = 0.f, hx1 = 0.f, pin[4] = {0.f, 0.f, 0.f, 0.f};
    if (head && live && !tail.PW) {
        const float* xr = tail.hx + row * tail.ldhx;
        hx0 = lane < tail.K ? xr[lane] : 0.f; hx1 = lane + kWave < tail.K ? xr[lane + kWave] : 0.f;
    }
    if (tail.PW && live) {
        const float* xr = tail.px + row * tail.ldpx;
#pragma unroll
        for (int e = 0; e < 4; e++) pin[e] = lane + e * kWave < tail.PK ? xr[lane + e * kWave] : 0.f;
    }
    if (tail.K > 0) ce_dx_stage(tail, c, ce_lds);
    if (!live) return;
    const float inv = 1.0f / ((float)n * (float)c);
    if (tail.PW) {
        float o[2];
        ce_pre_layer(tail, ce_lds + ce_w_region(tail.K, c), row, lane, pin, o);
        hx0 = o[0]; hx1 = o[1];
    }
    const float myz = head ? ce_head(tail, ce_lds, row, lane, c, hx0, hx1) : 0.f;
    float l = 0.f, hit = 0.f, mydz = 0.f;
    for (int k = lane; k < c; k += kWave) {
        const float x = head ? myz : z[row * ldz + k], y = k < kWave ? lab0 : lab[row * ldl + k];
        const float e = __expf(-fabsf(x));
        l += fmaxf(x, 0.f) - x * y + log1pf(e);
        const float p = x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
        hit += ((x > 0.f) == (y > 0.5f)) ? 1.f : 0.f;
        mydz = (p - y) * inv;
        if (dz) dz[row * lddz + k] = mydz;
        if (pred) pred[row * ldp + k] = p;
    }
    l = wave_sum(l); hit = wave_sum(hit);
    if (lane == 0) { rowstat[row] = l; rowstat[n + row] = hit; }
    if (tail.dx) ce_dx_tail(tail, ce_lds, row, lane, c, mydz);
}

// stats = {sum CE, #correct elements, mean CE, accuracy} over n*c elements, fixed summation order
__global__ __launch_bounds__(kBlock) void sigmoid_stats_kernel(const float* __restrict__ rowstat, int32_t n,
                                                               int32_t c, float* __restrict__ stats) {
    __shared__ float red[2][kBlock];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += kBlock) { a += rowstat[i]; b += rowstat[n + i]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float cnt = (float)n * (float)c;
        stats[0] = red[0][0]; stats[1] = red[1][0];
        stats[2] = red[0][0] / cnt; stats[3] = red[1][0] / cnt;
    }
}

// Weight decay of the first parametrised layer (gcn/models.py:68-75: loss += wd * l2_loss(var),
// l2_loss = sum(v^2)/2): ONE workgroup over the flat-buffer range [lo, hi) adds wd*theta to the
// gradient and 0.5*wd*sum(theta^2) to the loss slot, partials combined in a fixed order.
__global__ __launch_bounds__(kBlock) void l2_penalty_kernel(const float* __restrict__ theta, int64_t lo, int64_t hi,
                                                            float wd, float* __restrict__ grad,
                                                            float* __restrict__ loss) {
    __shared__ float red[kBlock];
    float a = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
        const float t = theta[i];
        a += t * t;
        if (grad) grad[i] += wd * t;
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) loss[0] += 0.5f * wd * red[0];
}

__global__ __launch_bounds__(kBlock) void adam_kernel(AdamArgs A) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * kBlock) adam_one(A, i, A.grad[i]);
}

// Adam with the step's loss / accuracy statistics riding along: workgroups [0, gridDim.x - 1) are adam_kernel, the LAST one
// is softmax_stats_kernel / sigmoid_stats_kernel (same code, same summation order -> the same bits).  Nothing in a step
// reads the statistics before the optimizer has run, so they need neither a launch nor a stream of their own.
// ... and the history scatter (tf.scatter_update after the optimizer, gcn/models.py:160-166): rows of the step's activations
// into the resident history, independent of the weights -- up to two of them as further workgroups of the same launch.
struct TailScatter {
    float* H[2]; int64_t ldh[2]; const int32_t* idx[2]; int32_t n[2], d[2]; const float* src[2]; int64_t lds[2];
    int32_t first[3];            // workgroups [first[j], first[j + 1]) after the Adam + statistics workgroups copy job j's rows
    int32_t jobs;
};
constexpr int kTailRowsPerBlock = kBlock / 32;      // 32 lanes x float4 per 128 columns of a row

// ... and, when the step's grouped weight-gradient launch sits right in front of the optimizer, ITS reductions (split-K
// partial tiles, LayerNorm parameter partials: reduce_multi_body) as the first workgroups of the launch, each applying the
// update to the element it has just reduced; the Adam workgroups then walk only the GAPS the reductions do not cover.
struct TailGaps { int64_t start[24]; int64_t len[24]; int32_t n; int64_t total; };   // total < 0: no reductions, everything is a gap

__global__ __launch_bounds__(kBlock) void adam_stats_kernel(AdamArgs A, const float* __restrict__ rowstat, int32_t rows, int32_t c,
                                                            int32_t softmax, float* __restrict__ stats, int32_t nb, TailScatter sc,
                                                            ReduceMulti R, int32_t rblocks, TailGaps G) {
    if ((int)blockIdx.x < rblocks) { reduce_multi_body(R, (int)blockIdx.x, &A); return; }
    const int bx = (int)blockIdx.x - rblocks;
    if (bx < nb) {
        if (G.total < 0) {
            for (int64_t i = (int64_t)bx * kBlock + threadIdx.x; i < A.n; i += (int64_t)nb * kBlock) adam_one(A, i, A.grad[i]);
        } else {
            for (int64_t t = (int64_t)bx * kBlock + threadIdx.x; t < G.total; t += (int64_t)nb * kBlock) {
                int64_t r = t;
                int q = 0;
                while (q + 1 < G.n && r >= G.len[q]) { r -= G.len[q]; q++; }
                const int64_t i = G.start[q] + r;
                adam_one(A, i, A.grad[i]);
            }
        }
        return;
    }
    if (bx > nb) {                                   // history rows: one 32-lane group per row, float4 per lane and 128 columns
        const int b = bx - nb - 1;
        const int j = (sc.jobs > 1 && b >= sc.first[1]) ? 1 : 0;
        const int64_t i = (int64_t)(b - sc.first[j]) * kTailRowsPerBlock + threadIdx.x / 32;
        if (i >= sc.n[j]) return;
        const int64_t ri = sc.idx[j][i];
        if (ri < 0) return;                          // negative row id: padding slot, nothing to write
        const float* src = sc.src[j] + i * sc.lds[j];
        float* dst = sc.H[j] + ri * sc.ldh[j];
        for (int cq = (threadIdx.x & 31) * 4; cq < sc.d[j]; cq += 128)
            *reinterpret_cast<float4*>(dst + cq) = *reinterpret_cast<const float4*>(src + cq);
        return;
    }
    if (rows <= 0) return;                           // no statistics parked
    __shared__ float red[2][kBlock];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < rows; i += kBlock) { a += rowstat[i]; b += rowstat[rows + i]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float cnt = softmax ? (float)rows : (float)rows * (float)c;
        stats[0] = red[0][0]; stats[1] = red[1][0];
        stats[2] = red[0][0] / cnt; stats[3] = red[1][0] / cnt;
    }
}

// out = x * mask / keep with the hash mask of sgcn_dropout_t (unfused form and its own backward)
__global__ __launch_bounds__(kBlock) void dropout_kernel(const float* __restrict__ x, int64_t ldx,
                                                         int32_t n, int32_t d, DropArgs a,
                                                         float* __restrict__ out, int64_t ldo) {
    const int64_t total = (int64_t)n * d;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int row = (int)(i / d), col = (int)(i % d);
        out[row * ldo + col] = x[row * ldx + col] * drop_factor(a, row, col);
    }
}

}  // namespace sgcn

using namespace sgcn;

extern "C" int sgcn_ln_act_fwd_f32(const float* x, int64_t ldx, const float* offset,
                                   const float* scale, int32_t n, int32_t d, float eps,
                                   int32_t relu, float* y, int64_t ldy, float* xhat, float* rstd,
                                   void* stream) {
    SGCN_REQUIRE(n >= 0 && d >= 0, "ln_act_fwd: negative size");
    if (n == 0 || d == 0) return SGCN_OK;
    const int norm = (offset && scale) ? 1 : 0;
    SGCN_REQUIRE(x && y && (!norm || (xhat && rstd)), "ln_act_fwd: null operand");
    const unsigned blocks = (unsigned)((n + 3) / 4);
    hipLaunchKernelGGL(ln_act_fwd_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, x, ldx,
                       offset, scale, n, d, eps, norm, relu, y, ldy, xhat, rstd);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

extern "C" int64_t sgcn_ln_act_bwd_ws_floats(int32_t n, int32_t d) {
    const int64_t blocks = ((int64_t)n + 4 * kBwdRowsPerWave - 1) / (4 * kBwdRowsPerWave);
    return blocks * 2 * d;
}

int sgcn::ln_act_bwd_launch(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* xhat,
                            const float* rstd, const float* scale, int32_t n, int32_t d, int32_t relu,
                            float* dx, int64_t lddx, float* doffset, float* dscale, float* ws,
                            bool reduce_params, int32_t* nblk, hipStream_t st, const float* tail_W, int32_t tail_K,
                            int32_t tail_kg, const sgcn_dropout_t* tail_drop, float* tail_dx, int64_t tail_lddx,
                            const float* nx_y, int64_t nx_ldy, const float* nx_xhat, const float* nx_rstd, const float* nx_scale,
                            int32_t nx_relu, float* nx_g, float* nx_partial) {
    SGCN_REQUIRE(n >= 0 && d >= 0, "ln_act_bwd: negative size");
    if (nblk) *nblk = 0;
    if (n == 0 || d == 0) return SGCN_OK;
    const int norm = scale ? 1 : 0;
    SGCN_REQUIRE(dy && y && dx && (!norm || (xhat && rstd && doffset && dscale && ws)),
                 "ln_act_bwd: null operand");
    SGCN_REQUIRE(!norm || (size_t)d * 8 * sizeof(float) <= 64 * 1024, "ln_act_bwd: d too large for LDS");
    LnBwdDx t{};
    LnBwdNext nx{};
    size_t lds = norm ? (size_t)d * 8 * sizeof(float) : 0;
    if (tail_W) {
        SGCN_REQUIRE(tail_dx && d <= 2 * kWave && d % 4 == 0 && aligned16(tail_W) && tail_K > 0 && tail_K <= 4 * kWave &&
                     tail_lddx >= tail_K && tail_kg >= 1 && tail_kg <= 2, "ln_act_bwd: bad input-gradient tail");
        t.W = tail_W; t.K = tail_K; t.kg = tail_kg; t.drop = drop_args(tail_drop); t.dxo = tail_dx; t.lddxo = tail_lddx;
        SGCN_REQUIRE(!t.drop.on || t.drop.width == tail_K, "ln_act_bwd: dropout width must be the layer's input width");
        t.dma = (d == 32 || d == 64 || d == 128) ? 1 : 0;
        lds += (t.dma ? ((size_t)tail_K * d + 255) / 256 * 256 : (size_t)d * (tail_K + 1)) * sizeof(float);
        if (nx_y) {
            SGCN_REQUIRE(tail_K <= 2 * kWave && nx_g && nx_ldy >= tail_K && (!nx_scale || (nx_xhat && nx_rstd && nx_partial)),
                         "ln_act_bwd: bad chained layer");
            nx.y = nx_y; nx.ldy = nx_ldy; nx.xhat = nx_xhat; nx.rstd = nx_rstd; nx.scale = nx_scale; nx.norm = nx_scale ? 1 : 0;
            nx.relu = nx_relu; nx.g = nx_g; nx.partial = nx_partial;
            if (nx.norm) lds += (size_t)8 * tail_K * sizeof(float);
        }
        SGCN_REQUIRE(lds <= 160 * 1024, "ln_act_bwd: weights too large for LDS");
        static size_t raised = 0;          // > 64 KB of dynamic LDS needs the attribute
        if (lds > 64 * 1024 && lds > raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_act_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised = 160 * 1024;
        }
    }
    const int rows_per_block = 4 * kBwdRowsPerWave;
    const unsigned blocks = (unsigned)((n + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(ln_act_bwd_kernel, dim3(blocks), dim3(kBlock), lds, st, dy, lddy, y, ldy, xhat, rstd, scale, n, d, norm,
                       relu, dx, lddx, ws, t, nx);
    if (nblk) *nblk = norm ? (int32_t)blocks : 0;
    if (norm && reduce_params)
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * d + kLnRedCols - 1) / kLnRedCols), dim3(256), 0, st, ws,
                           (int32_t)blocks, d, doffset, dscale);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

extern "C" int sgcn_ln_act_bwd_f32(const float* dy, int64_t lddy, const float* y, int64_t ldy,
                                   const float* xhat, const float* rstd, const float* scale,
                                   int32_t n, int32_t d, int32_t relu, float* dx, int64_t lddx,
                                   float* doffset, float* dscale, float* ws, void* stream) {
    return sgcn::ln_act_bwd_launch(dy, lddy, y, ldy, xhat, rstd, scale, n, d, relu, dx, lddx, doffset, dscale, ws,
                                   true, nullptr, (hipStream_t)stream);
}

namespace sgcn {
int softmax_stats_launch(const float* rowstat, int32_t n, float* stats, void* stream) {
    hipLaunchKernelGGL(softmax_stats_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, rowstat, n, stats);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}
int aux_fork(void* stream, void** aux_stream);          // sgcn_gemm.hip

// The statistics reduction of the step's loss kernel, parked until the optimizer's launch (stats_defer(1) .. the next
// sgcn::adam_with_stats or stats_flush): {softmax?, rowstat, rows, c, stats}
namespace {
struct PendingStats { int on = 0, armed = 0, softmax = 0; const float* rowstat = nullptr; int32_t n = 0, c = 0; float* stats = nullptr; };
PendingStats& pending_stats() { static PendingStats p; return p; }
}  // namespace
void stats_defer(int on) { pending_stats().on = on; }
// whatever is parked runs now, on `stream` (a program without an optimizer step after its loss; the end of a run)
int stats_flush(void* stream) {
    PendingStats& p = pending_stats();
    if (!p.armed) return SGCN_OK;
    p.armed = 0;
    if (p.softmax) hipLaunchKernelGGL(softmax_stats_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, p.rowstat, p.n, p.stats);
    else hipLaunchKernelGGL(sigmoid_stats_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, p.rowstat, p.n, p.c, p.stats);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}
// history scatters parked for the optimizer's launch (sgcn_step_run: SCATTER_ROWS ops that directly follow ADAM)
namespace { TailScatter& pending_scatter() { static TailScatter t{}; return t; } }
// true when the job can ride (whole float4 columns, 16-byte aligned rows); false: the caller launches it itself
bool scatter_park(float* H, int64_t ldh, const int32_t* idx, int32_t n, int32_t d, const float* src, int64_t lds) {
    TailScatter& t = pending_scatter();
    if (t.jobs >= 2 || n <= 0 || d <= 0 || d % 4 || ldh % 4 || lds % 4 || !aligned16(H) || !aligned16(src) || !idx) return false;
    const int j = t.jobs++;
    t.H[j] = H; t.ldh[j] = ldh; t.idx[j] = idx; t.n[j] = n; t.d[j] = d; t.src[j] = src; t.lds[j] = lds;
    if (j == 0) t.first[0] = 0;
    t.first[j + 1] = t.first[j] + (n + kTailRowsPerBlock - 1) / kTailRowsPerBlock;
    return true;
}
__global__ void reduce_multi_kernel(ReduceMulti R) { reduce_multi_body(R, (int)blockIdx.x, nullptr); }

// reductions parked for the optimizer's launch (sgcn_gemm.hip dw_group_flush, when sgcn_step_run sees ADAM right behind it)
namespace { struct PendingReduce { bool on = false; ReduceMulti R; int blocks = 0; }; PendingReduce& pending_reduce() { static PendingReduce p; return p; } }
bool reduce_parked() { return pending_reduce().on; }       // (its operands point into the aux ring: sgcn_gemm.hip aux_join)
bool reduce_park(const ReduceMulti& R, int blocks) {
    PendingReduce& p = pending_reduce();
    if (p.on) return false;
    p.on = true; p.R = R; p.blocks = blocks;
    return true;
}
// the parked reductions as their own launch (no optimizer followed after all, or their outputs are not a clean cover)
static int reduce_unpark(void* stream) {
    PendingReduce& p = pending_reduce();
    if (!p.on) return SGCN_OK;
    p.on = false;
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)p.blocks), dim3(256), 0, (hipStream_t)stream, p.R);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}
int reduce_flush(void* stream) { return reduce_unpark(stream); }
// the gaps of [0, n) that the parked reductions' outputs leave (false: they are not disjoint contiguous ranges inside it)
static bool reduce_gaps(const ReduceMulti& R, const float* grad, int64_t n, TailGaps& G) {
    struct Rg { int64_t lo, hi; } r[3 * kMaxGroup];
    int nr = 0;
    for (int k = 0; k < R.n; k++) {
        if (R.j[k].pending) {
            if (R.j[k].ldc != R.j[k].N) return false;
            r[nr++] = Rg{R.j[k].C - grad, R.j[k].C - grad + (int64_t)R.j[k].M * R.j[k].N};
        }
        if (R.nblk[k] > 0) {
            r[nr++] = Rg{R.doffset[k] - grad, R.doffset[k] - grad + R.d[k]};
            r[nr++] = Rg{R.dscale[k] - grad, R.dscale[k] - grad + R.d[k]};
        }
    }
    std::sort(r, r + nr, [](const Rg& a, const Rg& b) { return a.lo < b.lo; });
    G.n = 0; G.total = 0;
    int64_t at = 0;
    for (int k = 0; k <= nr; k++) {
        const int64_t lo = k < nr ? r[k].lo : n;
        if (lo < at || (k < nr && r[k].hi > n)) return false;          // overlapping, or outside the buffer
        if (lo > at) {
            if (G.n >= 24) return false;
            G.start[G.n] = at; G.len[G.n] = lo - at; G.total += lo - at; G.n++;
        }
        if (k < nr) at = r[k].hi;
    }
    if (G.n == 0) { G.start[0] = 0; G.len[0] = 0; G.n = 1; }
    return true;
}
int adam_with_stats(float* theta, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                    float eps, void* stream) {
    PendingStats& p = pending_stats();
    TailScatter sc = pending_scatter();
    pending_scatter().jobs = 0;
    PendingReduce& pr = pending_reduce();
    TailGaps G{};
    G.total = -1;
    bool red = false;
    if (pr.on && n > 0 && theta && grad && m && v && reduce_gaps(pr.R, grad, n, G)) { red = true; pr.on = false; }
    else { const int rc = reduce_unpark(stream); if (rc != SGCN_OK) return rc; G.total = -1; }
    if ((!p.armed && sc.jobs == 0 && !red) || n <= 0) {
        const int rc = stats_flush(stream);
        if (rc != SGCN_OK) return rc;
        for (int j = 0; j < sc.jobs; j++) {
            const int r2 = sgcn_scatter_rows_f32(sc.H[j], sc.ldh[j], sc.idx[j], sc.n[j], sc.d[j], sc.src[j], sc.lds[j], stream);
            if (r2 != SGCN_OK) return r2;
        }
        return n > 0 ? sgcn_adam_f32(theta, grad, m, v, n, lr_t, beta1, beta2, eps, stream) : SGCN_OK;
    }
    const bool st = p.armed != 0;
    p.armed = 0;
    SGCN_REQUIRE(theta && grad && m && v, "adam: null operand");
    const int64_t walk = red ? G.total : n;
    const unsigned blocks = (unsigned)std::min<int64_t>((walk + kBlock - 1) / kBlock, 2048);
    const unsigned extra = sc.jobs ? (unsigned)sc.first[sc.jobs] : 0u;
    const unsigned rblocks = red ? (unsigned)pr.blocks : 0u;
    hipLaunchKernelGGL(adam_stats_kernel, dim3(rblocks + blocks + 1 + extra), dim3(kBlock), 0, (hipStream_t)stream,
                       AdamArgs{theta, grad, m, v, n, lr_t, beta1, beta2, eps}, st ? p.rowstat : nullptr, st ? p.n : 0, p.c, p.softmax,
                       p.stats, (int32_t)blocks, sc, red ? pr.R : ReduceMulti{}, (int32_t)rblocks, G);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}
// The loss kernels with the statistics reduction (loss / accuracy sums: nothing in the step depends on them
// before the optimizer's join) on the auxiliary stream when `overlap`: one kernel less on the step's chain.
// `last` (sgcn_fuse.h): what of the network around the loss the kernel's row pass takes over -- nullptr / empty: nothing.
int ce_impl(bool softmax, const float* logits, int64_t ldz, const float* labels, int64_t ldl, int32_t n, int32_t c,
            float* dlogits, int64_t lddz, float* pred, int64_t ldp, float* stats, float* rowstat, void* stream,
            bool overlap, const CeLastLayer* last) {
    CeDx tail{};
    if (last && last->W && last->K > 0 && (last->dx || last->hx || last->pre)) {
        const CeLastLayer& L = *last;
        SGCN_REQUIRE(c <= kWave && L.ldw >= c && (int64_t)L.K * c * 4 <= 48 * 1024, "ce: bad fused last layer");
        tail.W = L.W; tail.ldw = L.ldw; tail.K = L.K;
        tail.wflat = (L.ldw == c && ((int64_t)L.K * c) % 4 == 0 && aligned16(L.W)) ? 1 : 0;
        if (L.dx) {
            SGCN_REQUIRE(L.lddx >= L.K, "ce: bad dx pitch");
            tail.dx = L.dx; tail.lddx = L.lddx; tail.drop = drop_args(L.dx_drop);
            SGCN_REQUIRE(!tail.drop.on || tail.drop.width == L.K, "ce: dropout width must be the layer's input width");
        }
        if (L.hx || L.pre) {
            SGCN_REQUIRE(L.K <= 2 * kWave && L.kg >= 1 && L.kg <= 2 && (L.pre || L.ldhx >= L.K), "ce: bad fused head");
            tail.hx = L.hx; tail.ldhx = L.ldhx; tail.kg = L.kg; tail.hdrop = drop_args(L.h_drop);
            tail.zout = const_cast<float*>(logits); tail.ldzo = ldz;
            SGCN_REQUIRE(!tail.hdrop.on || tail.hdrop.width == L.K, "ce: dropout width must be the layer's input width");
        }
        if (L.pre) {
            SGCN_REQUIRE(L.px && L.PW && L.pY && L.PK > 0 && L.PK <= 4 * kWave && L.ldpx >= L.PK && L.ldpy >= L.K && (L.PK * L.K) % 4 == 0 &&
                         aligned16(L.PW) && L.pS >= 1 && L.pkg >= 1 && L.pkg <= 2 && L.pkchunk % 32 == 0 && L.pkchunk > 0 &&
                         (int64_t)L.PK * L.K <= 32 * 4 * kBlock, "ce: bad fused pre-layer");
            const int norm = (L.poff && L.psc) ? 1 : 0;
            SGCN_REQUIRE(!norm || (L.pxhat && L.prstd), "ce: the pre-layer's LayerNorm needs xhat / rstd");
            tail.px = L.px; tail.ldpx = L.ldpx; tail.PK = L.PK; tail.pS = L.pS; tail.pkchunk = L.pkchunk; tail.pkg = L.pkg;
            tail.PW = L.PW; tail.pdrop = drop_args(L.p_drop); tail.poff = L.poff; tail.psc = L.psc; tail.peps = L.peps;
            tail.prelu = L.prelu; tail.pepi = norm ? 2 : (L.prelu ? 1 : 0);
            tail.pY = L.pY; tail.ldpy = L.ldpy; tail.pxhat = L.pxhat; tail.prstd = L.prstd;
            SGCN_REQUIRE(!tail.pdrop.on || tail.pdrop.width == L.PK, "ce: dropout width must be the pre-layer's input width");
        }
    }
    hipStream_t st = (hipStream_t)stream;
    size_t lds = tail.K > 0 ? (size_t)ce_w_region(tail.K, c) * sizeof(float) : 0;
    if (tail.PW) {
        lds += ((size_t)tail.PK * tail.K + 255) / 256 * 256 * sizeof(float);
        SGCN_REQUIRE(lds <= 160 * 1024, "ce: the two weight matrices do not fit the LDS");
        static bool raised = false;        // > 64 KB of dynamic LDS needs the attribute, once per kernel
        if (lds > 64 * 1024 && !raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&softmax_ce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sigmoid_ce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised = true;
        }
    }
    if (softmax)
        hipLaunchKernelGGL(softmax_ce_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), lds, st, logits, ldz, labels, ldl,
                           n, c, dlogits, lddz, pred, ldp, rowstat, tail);
    else
        hipLaunchKernelGGL(sigmoid_ce_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), lds, st, logits, ldz, labels, ldl,
                           n, c, dlogits, lddz, pred, ldp, rowstat, tail);
    if (pending_stats().on) {            // parked for the optimizer's launch
        PendingStats& p = pending_stats();
        p.armed = 1; p.softmax = softmax ? 1 : 0; p.rowstat = rowstat; p.n = n; p.c = c; p.stats = stats;
        SGCN_HIP_TRY(hipGetLastError());
        return SGCN_OK;
    }
    hipStream_t ss = st;
    if (overlap) {
        void* side = nullptr;
        const int rc = aux_fork(stream, &side);
        if (rc != SGCN_OK) return rc;
        ss = (hipStream_t)side;
    }
    if (softmax) hipLaunchKernelGGL(softmax_stats_kernel, dim3(1), dim3(kBlock), 0, ss, rowstat, n, stats);
    else hipLaunchKernelGGL(sigmoid_stats_kernel, dim3(1), dim3(kBlock), 0, ss, rowstat, n, c, stats);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}
}  // namespace sgcn

extern "C" int sgcn_softmax_ce_f32(const float* logits, int64_t ldz, const float* labels,
                                   int64_t ldl, int32_t n, int32_t c, float* dlogits, int64_t lddz,
                                   float* pred, int64_t ldp, float* stats, float* rowstat,
                                   void* stream) {
    // the class plane of rowstat packs prediction + 4096 * label into one exact fp32
    SGCN_REQUIRE(!(pred && rowstat) || c <= 4096, "softmax_ce: the class plane of rowstat holds at most 4096 classes (c = %d)", c);
    return sgcn::ce_impl(true, logits, ldz, labels, ldl, n, c, dlogits, lddz, pred, ldp, stats, rowstat, stream, false, nullptr);
}

extern "C" int sgcn_sigmoid_ce_f32(const float* logits, int64_t ldz, const float* labels, int64_t ldl, int32_t n,
                                   int32_t c, float* dlogits, int64_t lddz, float* pred, int64_t ldp,
                                   float* stats, float* rowstat, void* stream) {
    return sgcn::ce_impl(false, logits, ldz, labels, ldl, n, c, dlogits, lddz, pred, ldp, stats, rowstat, stream, false, nullptr);
}

extern "C" int sgcn_l2_penalty_f32(const float* theta, int64_t lo, int64_t hi, float wd, float* grad,
                                   float* loss, void* stream) {
    SGCN_REQUIRE(lo >= 0 && hi >= lo, "l2_penalty: bad range");
    if (hi == lo || wd == 0.f || (!grad && !loss)) return SGCN_OK;
    SGCN_REQUIRE(theta, "l2_penalty: null operand");
    hipLaunchKernelGGL(l2_penalty_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, theta, lo, hi, wd, grad, loss);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

extern "C" int sgcn_adam_f32(float* theta, const float* grad, float* m, float* v, int64_t n,
                             float lr_t, float beta1, float beta2, float eps, void* stream) {
    SGCN_REQUIRE(n >= 0, "adam: negative size");
    if (n == 0) return SGCN_OK;
    SGCN_REQUIRE(theta && grad && m && v, "adam: null operand");
    const unsigned blocks = (unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 2048);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream,
                       AdamArgs{theta, grad, m, v, n, lr_t, beta1, beta2, eps});
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

extern "C" int sgcn_dropout_f32(const float* x, int64_t ldx, int32_t n, int32_t d,
                                const sgcn_dropout_t* drop, float* out, int64_t ldo, void* stream) {
    SGCN_REQUIRE(n >= 0 && d >= 0, "dropout: negative size");
    if (n == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(x && out && drop && ldx >= d && ldo >= d, "dropout: bad operand");
    SGCN_REQUIRE(drop->keep > 0.f && drop->width >= d, "dropout: keep must be > 0 and width >= d");
    DropArgs a = drop_args(drop);
    if (!a.on) { a.on = 1; a.rows = 0; a.scale = 1.f; }      // keep >= 1: plain copy
    const int64_t total = (int64_t)n * d;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, 4096);
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, x, ldx, n, d, a,
                       out, ldo);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}
