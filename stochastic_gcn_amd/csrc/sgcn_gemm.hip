// fp32 MFMA GEMM for the dense weight layers of the step, gfx950.
//
//   C[M x N] = op(A)[M x K] . op(B)[K x N] (+ C)        op = identity or transpose
//   optional fused epilogue  Y = act(LN(C) * scale + offset)   (MyLayerNorm2 + ReLU,
//                                                                gcn/layers.py:95-97,404-411)
//
// These GEMMs are small (Reddit step: 2042 x 1204 x 128 and smaller) and sit between the sparse
// kernels of a launch-latency-bound step, so the point is one cheap launch per layer with the
// LayerNorm fused, not peak FLOPs: exact-fp32 v_mfma_f32_32x32x2_f32 (the f32 matrix-core op of
// CDNA4: bitwise a k-ordered fmaf chain), a 32 x 128 block tile (4 wavefronts x one 32 x 32
// accumulator), K stepped by 32 through LDS (A tile padded to 33 floats/row so the per-lane
// column reads of the A fragment are bank-conflict free), epilogue through LDS so a wavefront
// owns whole rows for the LayerNorm statistics.
#include "sgcn_dev.h"
#include "sgcn_bwd.h"

namespace sgcn {

typedef float f16acc __attribute__((ext_vector_type(16)));

constexpr int kTM = 32, kTN = 128, kTK = 32;

struct GemmArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int32_t M, N, K;
    int32_t accumulate;              // C += instead of C =
    int32_t kchunk;                  // split-K: blockIdx.z covers k in [z*kchunk, (z+1)*kchunk); the
    float* ws;                       // partial tiles go to ws[z][M][N] and splitk_reduce_kernel adds them
    // fused LN/act epilogue (N <= 128 only): Y = act(LN(C)*scale + offset)
    const float* offset; const float* scale; float eps; int32_t relu;
    float* xhat; float* rstd;        // [M x N], [M]   kept for the backward when LN is on
    int32_t epi;                     // 0 plain, 1 act only, 2 LN + act
    const float* A2; int64_t lda2; int32_t a_split;   // rows >= a_split of A come from A2 (not TA)
    DropArgs drop_a;                 // dropout on the stored A matrix, applied while loading
    DropArgs drop_c;                 // dropout on the output (plain epilogue only)
    int32_t vec_a, vec_b;            // operand rows 16-byte aligned: 4-float runs load as one float4
    const int32_t* a_gidx;           // row indirection of the stored A (and of A2): row r reads A[gidx[r]]
    const int32_t* a_gidx2;          //   -- the minibatch's feature rows are gathered by the GEMM itself
};

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The K loop is software-pipelined: the global loads of K-step s+1 are issued into registers
// before the MFMAs of step s, so their latency hides behind the matrix work and the two
// barriers (these GEMMs have 4..64 workgroups: nothing else would hide it).
//
// KG "K-groups": a workgroup is KG x 4 wavefronts; group kg takes the K-steps s = kg (mod KG)
// with its own LDS tiles, and the groups' 32 x 128 partial tiles are added in group order through
// LDS before the epilogue.  A forward GEMM of the step has M/32 = 16..64 workgroups and up to 38
// dependent K-steps of ~1,000 MFMA cycles each: KG = 4 cuts that serial chain to a quarter and
// still fits the fused LayerNorm epilogue (which needs whole rows in one workgroup, so split-K
// across workgroups is not an option there).
constexpr int kAsFloats = kTM * (kTK + 1), kBsFloats = kTK * (kTN + 4);
constexpr int kGroupFloats = kAsFloats + kBsFloats;

// (bx, by, bz): the tile (row block, column block) and the K slice -- the workgroup id of a plain launch, decoded from a
// linear workgroup index by the grouped launch below
template <bool TA, bool TB, int KG>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int bz) {
    extern __shared__ float smem[];
    const int kg = threadIdx.x / kBlock;                       // K-group of this thread
    const int tid = threadIdx.x % kBlock, lane = tid & 63, wave = tid >> 6;
    float (*As)[kTK + 1] = reinterpret_cast<float (*)[kTK + 1]>(smem + kg * kGroupFloats);            // [i][kk]
    float (*Bs)[kTN + 4] = reinterpret_cast<float (*)[kTN + 4]>(smem + kg * kGroupFloats + kAsFloats); // [kk][j]
    const int m0 = bx * kTM, n0 = by * kTN;
    const int kbeg = bz * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    f16acc acc = {};
    float ra[4], rb[16];
    // the LayerNorm parameters of the fused epilogue: requested now, used ~10 us later (a load issued inside the
    // epilogue's row loop costs every row a trip to the L2: 4 us of the 12.6 this kernel took on the step's 2,036 x 128 layer)
    float ep_scale[2] = {1.f, 1.f}, ep_offset[2] = {0.f, 0.f};
    if (g.epi == 2) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = (threadIdx.x & 63) + e * kWave;
            if (c < g.N) { ep_scale[e] = g.scale[c]; ep_offset[e] = g.offset[c]; }
        }
    }

    // four consecutive floats rowp[first .. first + 4) of a row whose start `rowp` is always addressable (the caller clamps the
    // row): elements at or past `limit`, and everything when !rowok, read as zero.  Straight-line -- the address is clamped
    // and the result selected -- so that the five loads of a K-step are all in flight before any of them is waited for.
    // (Rounds 1-2 guarded each load with a branch; the compiler put a full wait behind every one of them: a trip to the L2
    // per LOAD, five per K-step.  `vec`: the host verified 16-byte aligned rows and a run length that is a multiple of 4,
    // so a float4 is either all in range or all out.)
    auto ld4 = [&](const float* rowp, bool vec, bool rowok, int first, int limit, float* out) {
        if (vec) {
            const bool ok = rowok && first + 3 < limit;
            const float4 v = *reinterpret_cast<const float4*>(rowp + (ok ? first : 0));
            out[0] = ok ? v.x : 0.f; out[1] = ok ? v.y : 0.f; out[2] = ok ? v.z : 0.f; out[3] = ok ? v.w : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool ok = rowok && first + e < limit;
                const float v = rowp[ok ? first + e : 0];
                out[e] = ok ? v : 0.f;
            }
        }
    };
    // (!TA) the thread's A row is the same in every K-step: its address -- an index load away when the rows are gathered --
    // is formed once, not once per step in front of the load that needs it
    const float* arow = nullptr;
    if (!TA) {
        const int row = min(m0 + (tid >> 3), g.M - 1);
        if (g.A2 && row >= g.a_split) {
            const int r2 = row - g.a_split;
            arow = g.A2 + (int64_t)(g.a_gidx2 ? g.a_gidx2[r2] : r2) * g.lda2;
        } else {
            arow = g.A + (int64_t)(g.a_gidx ? g.a_gidx[row] : row) * g.lda;
        }
    }
    auto fetch = [&](int k0) {
        if (!TA) {          // A is [M x K]: thread reads 4 consecutive k of one row
            const int i = tid >> 3, kq = (tid & 7) * 4, row = m0 + i;
            ld4(arow, g.vec_a, row < g.M, k0 + kq, kend, ra);
            if (g.drop_a.on) {
#pragma unroll
                for (int e = 0; e < 4; e++) ra[e] *= drop_factor(g.drop_a, row, k0 + kq + e);   // stored A = x[row][k]
            }
        } else {            // A is [K x M]: thread reads 4 consecutive m of one k
            const int kk = tid >> 3, iq = (tid & 7) * 4, k = k0 + kk, kc = min(k, kend - 1);
            const int64_t ak = g.a_gidx ? g.a_gidx[kc] : kc;                      // stored row k of x
            ld4(g.A + ak * g.lda, g.vec_a, k < kend, m0 + iq, g.M, ra);
            if (g.drop_a.on) {
#pragma unroll
                for (int e = 0; e < 4; e++) ra[e] *= drop_factor(g.drop_a, k, m0 + iq + e);     // stored A = x[k][row]
            }
        }
        if (!TB) {          // B is [K x N]: 4 consecutive columns of one k
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int kk = (tid >> 5) + 8 * r, jq = (tid & 31) * 4, k = k0 + kk;
                ld4(g.B + (int64_t)min(k, kend - 1) * g.ldb, g.vec_b, k < kend, n0 + jq, g.N, rb + r * 4);
            }
        } else {            // B is [N x K]: 4 consecutive k of one column
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = (tid >> 3) + 32 * r, kq = (tid & 7) * 4, col = n0 + j;
                ld4(g.B + (int64_t)min(col, g.N - 1) * g.ldb, g.vec_b, col < g.N, k0 + kq, kend, rb + r * 4);
            }
        }
    };
    auto stage = [&]() {
        if (!TA) {
            const int i = tid >> 3, kq = (tid & 7) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) As[i][kq + e] = ra[e];
        } else {
            const int kk = tid >> 3, iq = (tid & 7) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) As[iq + e][kk] = ra[e];
        }
        if (!TB) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int kk = (tid >> 5) + 8 * r, jq = (tid & 31) * 4;
                *reinterpret_cast<float4*>(&Bs[kk][jq]) = make_float4(rb[r * 4], rb[r * 4 + 1], rb[r * 4 + 2], rb[r * 4 + 3]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = (tid >> 3) + 32 * r, kq = (tid & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; e++) Bs[kq + e][j] = rb[r * 4 + e];
            }
        }
    };

    // every group runs the same number of iterations (uniform barriers); a step past kend loads zeros
    const int iters = ((kend - kbeg + kTK - 1) / kTK + KG - 1) / KG;
    if (iters > 0) fetch(kbeg + kg * kTK);
    for (int it = 0; it < iters; it++) {
        const int k0 = kbeg + (it * KG + kg) * kTK;
        stage();
        __syncthreads();
        if (it + 1 < iters) fetch(k0 + KG * kTK);      // in flight during the MFMAs below
        // ---- 16 x (32x32x2) MFMAs: lane l feeds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]
        const int fi = lane & 31, fk = lane >> 5;
#pragma unroll
        for (int kk2 = 0; kk2 < kTK / 2; kk2++) {
            const float a = As[fi][kk2 * 2 + fk];
            const float b = Bs[kk2 * 2 + fk][wave * 32 + fi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int cj = wave * 32 + (lane & 31);
    if (KG > 1) {        // partial tiles of groups 1.. -> their own Bs region -> group 0 adds them in order
        if (kg > 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) Bs[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][cj] = acc[r];
        }
        __syncthreads();
        if (kg == 0) {
            for (int o = 1; o < KG; o++) {
                float (*P)[kTN + 4] = reinterpret_cast<float (*)[kTN + 4]>(smem + o * kGroupFloats + kAsFloats);
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] += P[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][cj];
            }
        }
    }
    if (g.epi == 0) {
        if (kg != 0) return;
        float* base = g.ws ? g.ws + (int64_t)bz * g.M * g.N : g.C;
        const int64_t ld = g.ws ? g.N : g.ldc;
        const bool add = !g.ws && g.accumulate;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = n0 + cj;
            if (row < g.M && col < g.N) {
                float* p = base + (int64_t)row * ld + col;
                float v = acc[r];
                if (g.drop_c.on) v *= drop_factor(g.drop_c, row, col);
                *p = add ? *p + v : v;
            }
        }
        return;
    }
    // ---- fused epilogue: tile -> LDS, then one wavefront per row (needs the whole row: N <= 128)
    float (*Cs)[kTN + 4] = reinterpret_cast<float (*)[kTN + 4]>(smem + kAsFloats);   // group 0's Bs: 32 x 132 floats
    if (KG > 1) __syncthreads();          // group 0 has finished reading the other groups' tiles
    if (kg == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) Cs[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][cj] = acc[r];
    }
    __syncthreads();
    for (int rr = kg * (kBlock / kWave) + wave; rr < kTM; rr += KG * (kBlock / kWave)) {
        const int row = m0 + rr;
        if (row >= g.M) break;
        float* yr = g.C + (int64_t)row * g.ldc;
        if (g.epi == 1) {
            for (int c = lane; c < g.N; c += kWave) { const float v = Cs[rr][c]; yr[c] = g.relu ? fmaxf(v, 0.f) : v; }
            continue;
        }
        float s = 0.f;
        for (int c = lane; c < g.N; c += kWave) s += Cs[rr][c];
        const float mean = wsum(s) / (float)g.N;
        float q = 0.f;
        for (int c = lane; c < g.N; c += kWave) { const float t = Cs[rr][c] - mean; q += t * t; }
        const float rs = rsqrtf(wsum(q) / (float)g.N + g.eps);
        if (lane == 0) g.rstd[row] = rs;
        float* hr = g.xhat + (int64_t)row * g.N;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = lane + e * kWave;
            if (c < g.N) {
                const float h = (Cs[rr][c] - mean) * rs;
                hr[c] = h;
                const float v = h * ep_scale[e] + ep_offset[e];
                yr[c] = g.relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
}

template <bool TA, bool TB, int KG>
__global__ __launch_bounds__(kBlock * KG) void gemm_kernel(GemmArgs g) {
    gemm_body<TA, TB, KG>(g, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// Several independent plain GEMMs (epi == 0) of the same operand layout in ONE launch: workgroups [first[j], first[j + 1])
// run job j exactly as its own launch would (same tiles, same K slices, same order of additions) -- the weight-gradient
// GEMMs of a training step, which depend on nothing but their layer's g and are needed only by the optimizer.
struct GemmGroup {
    GemmArgs g[kMaxGroup];
    int32_t first[kMaxGroup + 1];
    int32_t gx[kMaxGroup], gy[kMaxGroup];
    int32_t n;
};

template <bool TA, bool TB, int KG>
__global__ __launch_bounds__(kBlock * KG) void gemm_group_kernel(GemmGroup G) {
    const int b = (int)blockIdx.x;
    int j = 0;
#pragma unroll
    for (int q = 1; q < kMaxGroup; q++) j += (q < G.n && b >= G.first[q]) ? 1 : 0;
    const int l = b - G.first[j];
    const int gx = G.gx[j], gy = G.gy[j];
    gemm_body<TA, TB, KG>(G.g[j], l % gx, (l / gx) % gy, l / (gx * gy));
}

// C (+)= sum_z ws[z]   in z order (deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int32_t S, int32_t M, int32_t N,
                                     float* __restrict__ C, int64_t ldc, int32_t accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t mn = (int64_t)M * N;
    if (i >= mn) return;
    float s = 0.f;
    for (int z = 0; z < S; z++) s += ws[(int64_t)z * mn + i];
    float* p = C + (i / N) * ldc + (i % N);
    *p = accumulate ? *p + s : s;
}

// The two small reductions of a dense layer's backward in ONE launch: the split-K partial tiles of
// dW (workgroups [0, gemm_blocks)) and the LayerNorm parameter-gradient partials (the rest).
__global__ void dense_bwd_reduce_kernel(ReduceJob j, int32_t gemm_blocks, const float* __restrict__ ln_partial,
                                        int32_t nblk, int32_t d, float* __restrict__ doffset,
                                        float* __restrict__ dscale, int32_t ln_accumulate) {
    if ((int)blockIdx.x < gemm_blocks) {
        const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const int64_t mn = (int64_t)j.M * j.N;
        if (i >= mn) return;
        float s = 0.f;
        for (int z = 0; z < j.S; z++) s += j.ws[(int64_t)z * mn + i];
        float* p = j.C + (i / j.N) * j.ldc + (i % j.N);
        *p = j.accumulate ? *p + s : s;
    } else {
        ln_param_reduce_cols32(ln_partial, nblk, d, doffset, dscale, (int)blockIdx.x - gemm_blocks, ln_accumulate != 0);
    }
}

// The reductions of SEVERAL dense layers' backward in one launch (the step program's deferred weight-gradient group):
// workgroups [gfirst[j], gfirst[j + 1]) add job j's split-K partial tiles, workgroups [lfirst[j], lfirst[j + 1]) its
// LayerNorm parameter partials -- each exactly as that layer's own dense_bwd_reduce_kernel launch would (sgcn_dev.h
// reduce_multi_body; the optimizer's launch runs the same body with the update applied on the spot, sgcn_dense.hip).
__global__ void dense_bwd_reduce_multi_kernel(ReduceMulti R) { reduce_multi_body(R, (int)blockIdx.x, nullptr); }

// Split-K with the dense layer's epilogue: Y = act(LN(sum_z ws[z]) * scale + offset), one wavefront
// per output row (the forward GEMM of the first layer is 2,042 x 128 x 1,204: 64 tiles whose
// MFMA work alone is ~16 us on 64 CUs -- cut 4-ways over K it runs on all 256).
__global__ __launch_bounds__(kBlock) void splitk_ln_act_kernel(const float* __restrict__ ws, int32_t S,
                                                               GemmArgs g) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (row >= g.M) return;
    const int64_t mn = (int64_t)g.M * g.N;
    float v[2];                                   // N <= 128: two columns per lane
    float sc[2] = {1.f, 1.f}, of[2] = {0.f, 0.f};   // requested with the partial tiles, not after the row statistics
    if (g.epi == 2) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = lane + e * kWave;
            if (c < g.N) { sc[e] = g.scale[c]; of[e] = g.offset[c]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int c = lane + e * kWave;
        float s = 0.f;
        if (c < g.N)
            for (int z = 0; z < S; z++) s += ws[(int64_t)z * mn + row * g.N + c];
        v[e] = s;
    }
    float* yr = g.C + row * g.ldc;
    if (g.epi == 2) {
        const float mean = wsum(v[0] + v[1]) / (float)g.N;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 2; e++) if (lane + e * kWave < g.N) { const float t = v[e] - mean; q += t * t; }
        const float rs = rsqrtf(wsum(q) / (float)g.N + g.eps);
        if (lane == 0) g.rstd[row] = rs;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = lane + e * kWave;
            if (c < g.N) {
                const float h = (v[e] - mean) * rs;
                g.xhat[row * g.N + c] = h;
                const float y = h * sc[e] + of[e];
                yr[c] = g.relu ? fmaxf(y, 0.f) : y;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = lane + e * kWave;
            if (c < g.N) yr[c] = (g.epi == 1 && g.relu) ? fmaxf(v[e], 0.f) : v[e];
        }
    }
}


// ---- a split-K layer's reduce + epilogue AND the dense layer that follows it, as one row pass ----------------------------
// The step's first layer (2,036 x 1,204 -> 128) is cut over K to fill the chip, so its LayerNorm + ReLU already run in a
// one-wavefront-per-row kernel (splitk_ln_act_kernel above).  The layer behind it is 2,036 x 128 -> 128: an MFMA launch of
// its own took 13.3 us of the step's chain, of which the arithmetic is under one.  Here the wave that has just finished a
// row of layer 1 keeps it in registers (two columns per lane), applies layer 2's dropout mask, and multiplies it into
// layer 2's weight matrix, staged in LDS once per workgroup -- x_k through v_readlane, two output columns per lane -- in
// exactly the launch's order of additions (32-wide K-steps alternating between its K-groups, the groups' partial sums
// added in group order), then layer 2's own LayerNorm / ReLU epilogue with gemm_body's arithmetic.  Same bits as the two
// launches + the reduce it replaces (tests/test_step_program_gpu.py); profiles/rowmlp_probe.hip is the stand-alone form.
struct RowDense {
    const float* W; int64_t ldw; int32_t N, kg;
    const float* offset; const float* scale; float eps; int32_t relu, epi;
    DropArgs drop;
    float* Y; int64_t ldy; float* xhat; float* rstd;
};

__global__ __launch_bounds__(kBlock) void splitk_ln_dense_kernel(const float* __restrict__ ws, int32_t S, GemmArgs g, RowDense d) {
    extern __shared__ __attribute__((aligned(1024))) float wl[];                 // layer 2's weights, [K2 = g.N][d.N]
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    const bool live = row < g.M;
    const int K2 = g.N, total = K2 * d.N;
    // everything the pass reads from memory is requested before anything is used; layer 2's weights go straight from global
    // memory to the LDS (global_load_lds_dwordx4: no registers, all of them in flight at once; a lane past the end re-reads the
    // last 16 bytes into the region's padding -- the host rounds the allocation up to whole 1 KB instructions)
    {
        const int total4 = total / 4, wave = threadIdx.x >> 6;
        for (int base4 = wave * kWave; base4 < total4; base4 += (kBlock / kWave) * kWave) {
            const int i4 = min(base4 + lane, total4 - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(d.W + (int64_t)i4 * 4),
                                             (__attribute__((address_space(3))) void*)(wl + (int64_t)base4 * 4), 16, 0, 0);
        }
    }
    float sc1[2] = {1.f, 1.f}, of1[2] = {0.f, 0.f}, sc2[2] = {1.f, 1.f}, of2[2] = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int c = lane + e * kWave;
        if (g.epi == 2 && c < g.N) { sc1[e] = g.scale[c]; of1[e] = g.offset[c]; }
        if (d.epi == 2 && c < d.N) { sc2[e] = d.scale[c]; of2[e] = d.offset[c]; }
    }
    const int64_t mn = (int64_t)g.M * g.N;
    float v[2] = {0.f, 0.f};
    if (live) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = lane + e * kWave;
            float s = 0.f;
            if (c < g.N)
                for (int z = 0; z < S; z++) s += ws[(int64_t)z * mn + row * g.N + c];
            v[e] = s;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!live) return;
    // ---- layer 1's epilogue (splitk_ln_act_kernel)
    float x[2];
    float* yr = g.C + row * g.ldc;
    if (g.epi == 2) {
        const float mean = wsum(v[0] + v[1]) / (float)g.N;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 2; e++) if (lane + e * kWave < g.N) { const float t = v[e] - mean; q += t * t; }
        const float rs = rsqrtf(wsum(q) / (float)g.N + g.eps);
        if (lane == 0) g.rstd[row] = rs;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = lane + e * kWave;
            x[e] = 0.f;
            if (c < g.N) {
                const float h = (v[e] - mean) * rs;
                g.xhat[row * g.N + c] = h;
                const float y = h * sc1[e] + of1[e];
                x[e] = g.relu ? fmaxf(y, 0.f) : y;
                yr[c] = x[e];
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = lane + e * kWave;
            x[e] = 0.f;
            if (c < g.N) { x[e] = (g.epi == 1 && g.relu) ? fmaxf(v[e], 0.f) : v[e]; yr[c] = x[e]; }
        }
    }
    // ---- layer 2: dropout on the operand, then the product in the MFMA launch's order
    if (d.drop.on) { x[0] *= drop_factor(d.drop, (int)row, lane); x[1] *= drop_factor(d.drop, (int)row, lane + kWave); }
    const int c0 = lane < d.N ? lane : 0, c1 = lane + kWave < d.N ? lane + kWave : 0;
    float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
    for (int s0 = 0; s0 * 32 < K2; s0++) {
        const int gi = d.kg > 1 ? (s0 & 1) : 0;
        const float xs = s0 < 2 ? x[0] : x[1];    // a 32-wide K-step never straddles the two halves of the row
        const int kb = s0 * 32, lb = kb & (kWave - 1);
        const float* wk = wl + kb * d.N;
        float p0 = a0[gi], p1 = a1[gi];
        if (kb + 32 <= K2) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float w0[8], w1[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { w0[u] = wk[(q * 8 + u) * d.N + c0]; w1[u] = wk[(q * 8 + u) * d.N + c1]; }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), lb + q * 8 + u));
                    p0 = fmaf(xv, w0[u], p0); p1 = fmaf(xv, w1[u], p1);
                }
            }
        } else {
            for (int kk = 0; kb + kk < K2; kk++) {
                const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), lb + kk));
                p0 = fmaf(xv, wk[kk * d.N + c0], p0); p1 = fmaf(xv, wk[kk * d.N + c1], p1);
            }
        }
        a0[gi] = p0; a1[gi] = p1;
    }
    const float z[2] = {d.kg > 1 ? a0[0] + a0[1] : a0[0], d.kg > 1 ? a1[0] + a1[1] : a1[0]};
    // ---- layer 2's epilogue (gemm_body)
    float* y2 = d.Y + row * d.ldy;
    if (d.epi != 2) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int c = lane + e * kWave;
            if (c < d.N) y2[c] = (d.epi == 1 && d.relu) ? fmaxf(z[e], 0.f) : z[e];
        }
        return;
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 2; e++) if (lane + e * kWave < d.N) s += z[e];
    const float mean = wsum(s) / (float)d.N;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 2; e++) if (lane + e * kWave < d.N) { const float t = z[e] - mean; q += t * t; }
    const float rs = rsqrtf(wsum(q) / (float)d.N + d.eps);
    if (lane == 0) d.rstd[row] = rs;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int c = lane + e * kWave;
        if (c < d.N) {
            const float h = (z[e] - mean) * rs;
            d.xhat[row * d.N + c] = h;
            const float y = h * sc2[e] + of2[e];
            y2[c] = d.relu ? fmaxf(y, 0.f) : y;
        }
    }
}

}  // namespace sgcn

using namespace sgcn;

// Split-K factor: a weight-gradient GEMM (dW = X^T g) has a tiny output (128 x 128: FOUR tiles)
// and a long K (the ~1,000 rows of the minibatch), so the K range is cut across blockIdx.z until
// the grid has ~256 workgroups, every slice keeping at least three K-steps (so the K = 128 GEMMs of the step
// are never split: the reduction is a second launch and must buy more than it costs; knob gemm_min_steps --
// measured on the step's compute queue, one box: 3 -> 18 kernels, 150 us of kernel time; 5 -> 17 kernels, 154 us).
static int split_factor(int M, int N, int K) {
    const int tiles = ((M + kTM - 1) / kTM) * ((N + kTN - 1) / kTN);
    int s = 256 / std::max(tiles, 1);
    const int min_steps = tune_get("gemm_min_steps") > 0 ? tune_get("gemm_min_steps") : 3;
    s = std::min(s, K / (min_steps * kTK));
    return std::max(s, 1);
}

template <bool TA, bool TB, int KG>
static void launch_one(const GemmArgs& g, dim3 grid, hipStream_t st) {
    const size_t lds = (size_t)KG * kGroupFloats * sizeof(float);
    static bool raised = false;          // > 64 KB of dynamic LDS needs the attribute once per kernel
    if (lds > 64 * 1024 && !raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<TA, TB, KG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = true;
    }
    hipLaunchKernelGGL((gemm_kernel<TA, TB, KG>), grid, dim3(kBlock * KG), lds, st, g);
}

template <bool TA, bool TB>
static void launch_kg(const GemmArgs& g, dim3 grid, int kgroups, hipStream_t st) {
    if (kgroups >= 4) launch_one<TA, TB, 4>(g, grid, st);
    else if (kgroups == 2) launch_one<TA, TB, 2>(g, grid, st);
    else launch_one<TA, TB, 1>(g, grid, st);
}

// what launch_gemm decides before it launches: alignment flags, the K slicing (written into g), grid and K-groups
struct GemmPlan { dim3 grid; int kgroups, S, epi; };

static GemmPlan prepare_gemm(GemmArgs& g, float* ws, int ta = 0, int tb = 0) {
    auto al = [](const void* p, int64_t ld) { return p && ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); };
    // (a float4 must be all in range or all out: the run it lies in is a multiple of 4 long -- gemm_body ld4)
    g.vec_a = al(g.A, g.lda) && (!g.A2 || al(g.A2, g.lda2)) && (ta ? g.M : g.K) % 4 == 0;
    g.vec_b = al(g.B, g.ldb) && (tb ? g.K : g.N) % 4 == 0;
    int S = ws ? split_factor(g.M, g.N, g.K) : 1;
    g.kchunk = ((g.K + S - 1) / S + kTK - 1) / kTK * kTK;
    S = g.K > 0 ? (g.K + g.kchunk - 1) / g.kchunk : 1;
    if (g.K == 0) g.kchunk = kTK;
    g.ws = S > 1 ? ws : nullptr;
    GemmPlan p;
    p.epi = g.epi;
    p.S = S;
    if (S > 1) g.epi = 0;                 // partial tiles are plain; the epilogue moves to the reduce
    p.grid = dim3((unsigned)((g.M + kTM - 1) / kTM), (unsigned)((g.N + kTN - 1) / kTN), (unsigned)S);
    // K-groups inside the workgroup: worth it while the grid leaves most CUs idle and the
    // per-workgroup K chain is long
    const int steps = (std::min(g.kchunk, g.K) + kTK - 1) / kTK;
    const int blocks = (int)(p.grid.x * p.grid.y * p.grid.z);
    p.kgroups = (blocks <= 128 && steps >= 8) ? 4 : ((blocks <= 256 && steps >= 4) ? 2 : 1);
    return p;
}

// K slicing and K-groups a forward dense layer of this shape would get (sgcn_dense.hip's fused output head reproduces
// the launch's order of additions, so it asks)
namespace sgcn {
void gemm_fwd_shape(int M, int N, int K, int* S, int* kgroups, int* kchunk) {
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K;
    float dummy;
    const GemmPlan p = prepare_gemm(g, sgcn_gemm_ws_floats(M, N, K) > 0 ? &dummy : nullptr);
    *S = p.S; *kgroups = p.kgroups;
    if (kchunk) *kchunk = g.kchunk;
}
}  // namespace sgcn

static void launch_prepared(const GemmArgs& g, const GemmPlan& p, int ta, int tb, hipStream_t st) {
    if (!ta && !tb) launch_kg<false, false>(g, p.grid, p.kgroups, st);
    else if (ta && !tb) launch_kg<true, false>(g, p.grid, p.kgroups, st);
    else if (!ta && tb) launch_kg<false, true>(g, p.grid, p.kgroups, st);
    else launch_kg<true, true>(g, p.grid, p.kgroups, st);
}

static int launch_gemm(GemmArgs g, int ta, int tb, float* ws, hipStream_t st, ReduceJob* defer = nullptr) {
    if (defer) defer->pending = 0;
    const GemmPlan p = prepare_gemm(g, ws, ta, tb);
    const int S = p.S, epi = p.epi;
    launch_prepared(g, p, ta, tb, st);
    if (S > 1 && epi != 0) {
        g.epi = epi;
        const unsigned rb = (unsigned)((g.M + (kBlock / kWave) - 1) / (kBlock / kWave));
        hipLaunchKernelGGL(splitk_ln_act_kernel, dim3(rb), dim3(kBlock), 0, st, g.ws, S, g);
    } else if (S > 1 && defer) {           // the caller folds this reduction into its own launch
        *defer = ReduceJob{g.ws, S, g.M, g.N, g.C, g.ldc, g.accumulate, 1};
    } else if (S > 1) {
        const int64_t mn = (int64_t)g.M * g.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, st,
                           g.ws, S, g.M, g.N, g.C, g.ldc, g.accumulate);
    }
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

extern "C" int64_t sgcn_gemm_ws_floats(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int S = split_factor(M, N, K);
    return S > 1 ? (int64_t)S * M * N : 0;
}

extern "C" int sgcn_gemm_f32(int32_t trans_a, int32_t trans_b, int32_t M, int32_t N, int32_t K,
                             const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                             int64_t ldc, int32_t accumulate, float* ws,
                             const sgcn_dropout_t* drop_a, const sgcn_dropout_t* drop_c, void* stream) {
    SGCN_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative size");
    if (M == 0 || N == 0) return SGCN_OK;
    SGCN_REQUIRE(A && B && C, "gemm: null operand");
    GemmArgs g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.accumulate = accumulate; g.epi = 0;
    g.drop_a = drop_args(drop_a);
    g.drop_c = drop_args(drop_c);
    SGCN_REQUIRE(!g.drop_a.on || g.drop_a.width == (trans_a ? M : K), "gemm: drop_a width must be the stored A's row length");
    SGCN_REQUIRE(!g.drop_c.on || g.drop_c.width == N, "gemm: drop_c width must be N");
    if (g.drop_c.on) ws = nullptr;           // the output mask is applied in the GEMM's own epilogue
    return launch_gemm(g, trans_a, trans_b, ws, (hipStream_t)stream);
}


// Layer 1 (cut over K: partial tiles) and the dense layer on its output as GEMM + ONE row pass.  *fused = 0: the shapes do
// not allow it, nothing was launched.
namespace sgcn {
int dense_fwd_pair(int32_t M, int32_t N, int32_t K, const float* X, int64_t ldx, const float* X2, int64_t ldx2, int32_t split,
                   const float* W, int64_t ldw, const float* offset, const float* scale, float eps, int32_t relu, float* Y,
                   int64_t ldy, float* xhat, float* rstd, const sgcn_dropout_t* drop, float* ws, const int32_t* gidx,
                   const int32_t* gidx2, int32_t N2, const float* W2, int64_t ldw2, const float* offset2, const float* scale2,
                   float eps2, int32_t relu2, float* Y2, int64_t ldy2, float* xhat2, float* rstd2, const sgcn_dropout_t* drop2,
                   void* stream, int* fused) {
    *fused = 0;
    if (M <= 0 || N <= 0 || N > kTN || N2 <= 0 || N2 > kTN || !ws || !W2 || !Y2 || ldw2 != N2 || ldy != N) return SGCN_OK;
    const int norm = (offset && scale) ? 1 : 0, norm2 = (offset2 && scale2) ? 1 : 0;
    if ((norm && !(xhat && rstd)) || (norm2 && !(xhat2 && rstd2))) return SGCN_OK;
    GemmArgs g{};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy;
    g.M = M; g.N = N; g.K = K; g.offset = offset; g.scale = scale; g.eps = eps; g.relu = relu;
    g.xhat = xhat; g.rstd = rstd; g.epi = norm ? 2 : (relu ? 1 : 0);
    g.A2 = X2; g.lda2 = ldx2; g.a_split = split;
    g.a_gidx = gidx; g.a_gidx2 = gidx2;
    g.drop_a = drop_args(drop);
    const GemmPlan p = prepare_gemm(g, ws);
    int S2 = 0, kg2 = 0;
    gemm_fwd_shape(M, N2, N, &S2, &kg2, nullptr);
    if (p.S <= 1 || S2 != 1 || kg2 > 2 || (size_t)N * N2 * sizeof(float) > 64 * 1024 || (N * N2) % 4 || !aligned16(W2)) return SGCN_OK;
    SGCN_REQUIRE(!g.drop_a.on || g.drop_a.width == K, "dense_fwd: dropout width must be K");
    RowDense d{};
    d.W = W2; d.ldw = ldw2; d.N = N2; d.kg = kg2; d.offset = offset2; d.scale = scale2; d.eps = eps2; d.relu = relu2;
    d.epi = norm2 ? 2 : (relu2 ? 1 : 0);
    d.drop = drop_args(drop2);
    SGCN_REQUIRE(!d.drop.on || d.drop.width == N, "dense_fwd: the second layer's dropout width must be its input width");
    d.Y = Y2; d.ldy = ldy2; d.xhat = xhat2; d.rstd = rstd2;
    hipStream_t st = (hipStream_t)stream;
    launch_prepared(g, p, 0, 0, st);
    g.epi = p.epi;
    const unsigned rb = (unsigned)((g.M + (kBlock / kWave) - 1) / (kBlock / kWave));
    hipLaunchKernelGGL(splitk_ln_dense_kernel, dim3(rb), dim3(kBlock), ((size_t)N * N2 + 255) / 256 * 256 * sizeof(float), st, g.ws, p.S, g, d);
    SGCN_HIP_TRY(hipGetLastError());
    *fused = 1;
    return SGCN_OK;
}
}  // namespace sgcn

extern "C" int sgcn_dense_fwd_f32(int32_t M, int32_t N, int32_t K, const float* X, int64_t ldx,
                                  const float* X2, int64_t ldx2, int32_t split,
                                  const float* W, int64_t ldw, const float* offset,
                                  const float* scale, float eps, int32_t relu, float* Y, int64_t ldy,
                                  float* xhat, float* rstd, const sgcn_dropout_t* drop, float* ws,
                                  const int32_t* gidx, const int32_t* gidx2, void* stream) {
    SGCN_REQUIRE(M >= 0 && N >= 0 && K >= 0, "dense_fwd: negative size");
    if (M == 0 || N == 0) return SGCN_OK;
    SGCN_REQUIRE(X && W && Y, "dense_fwd: null operand");
    const int norm = (offset && scale) ? 1 : 0;
    SGCN_REQUIRE(!norm || (xhat && rstd), "dense_fwd: LayerNorm needs xhat / rstd");
    SGCN_REQUIRE(N <= kTN || (!norm && !relu), "dense_fwd: fused epilogue needs N <= 128");
    GemmArgs g{};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy;
    g.M = M; g.N = N; g.K = K; g.offset = offset; g.scale = scale; g.eps = eps; g.relu = relu;
    g.xhat = xhat; g.rstd = rstd; g.epi = norm ? 2 : (relu ? 1 : 0);
    SGCN_REQUIRE(!X2 || (split >= 0 && split <= M), "dense_fwd: bad split");
    g.A2 = X2; g.lda2 = ldx2; g.a_split = split;
    g.a_gidx = gidx; g.a_gidx2 = gidx2;
    g.drop_a = drop_args(drop);
    SGCN_REQUIRE(!g.drop_a.on || g.drop_a.width == K, "dense_fwd: dropout width must be K");
    return launch_gemm(g, 0, 0, N <= kTN ? ws : nullptr, (hipStream_t)stream);
}


// The whole backward of one dense layer in ONE call (three Python round trips are a quarter of
// the launch-bound training step):  g = LN/ReLU-backward(dy)  ->  dW += dropout(x)^T . g  ->
// dx = (g . W^T) * mask.   Same kernels, same order, same results as the three separate entries.
// ---- the weight-gradient side of a dense layer's backward on a second stream -----------------------
// dW = x^T g (+ its split-K reduction, + the LayerNorm parameter reduction) and dx = g W^T only share
// their input g: in the step program (csrc/sgcn_step.cpp) the dW side runs on a library-owned auxiliary
// stream, forked after the LayerNorm / ReLU backward that produces g, while the main stream goes on with
// dx and the layers below.  The join is one event wait before the optimizer touches the gradients.
// Nothing about the arithmetic changes -- same kernels, same operand order -- so results stay
// bit-identical to the serial path; only buffers that the aux work reads must stay untouched until the
// join: the caller's activations (the step program's arena gives every intermediate its own storage)
// and the reduction scratch, which therefore comes from a ring owned by the aux context.
namespace sgcn {
namespace {
struct AuxCtx {
    hipStream_t st = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    float* ring = nullptr;
    int64_t cap = 0, off = 0, want = 0;
    bool pending = false;
};
AuxCtx& aux_ctx() { static AuxCtx c; return c; }
int g_dw_recorded = 0;          // weight-gradient jobs recorded for the group launch: their ring regions are live
int g_grad_store = 0;           // step program in gradient-STORE mode: every parameter gradient is written exactly once per
                                // step, so the weight-gradient GEMMs and the LayerNorm-parameter reductions store instead of
                                // adding to a zeroed buffer (the step needs no memset, and no join on it)

int aux_init(AuxCtx& c) {
    if (c.st) return SGCN_OK;
    SGCN_HIP_TRY(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
    SGCN_HIP_TRY(hipEventCreateWithFlags(&c.fork, hipEventDisableTiming));
    SGCN_HIP_TRY(hipEventCreateWithFlags(&c.join, hipEventDisableTiming));
    return SGCN_OK;
}
}  // namespace

// make the auxiliary stream wait for everything `stream` has been given so far, and hand it out: work the
// caller then issues on it runs beside `stream` until the next aux_join
int aux_fork(void* stream, void** aux_stream) {
    AuxCtx& c = aux_ctx();
    const int rc = aux_init(c);
    if (rc != SGCN_OK) return rc;
    SGCN_HIP_TRY(hipEventRecord(c.fork, (hipStream_t)stream));
    SGCN_HIP_TRY(hipStreamWaitEvent(c.st, c.fork, 0));
    c.pending = true;
    *aux_stream = (void*)c.st;
    return SGCN_OK;
}

bool reduce_parked();           // sgcn_dense.hip: a reduction parked for the optimizer's launch still reads the ring
int aux_join(void* stream) {
    AuxCtx& c = aux_ctx();
    if (c.pending) {
        SGCN_HIP_TRY(hipEventRecord(c.join, c.st));
        SGCN_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, c.join, 0));
        c.pending = false;
    }
    // recorded jobs still own their regions of the ring, and so does a reduction parked for the optimizer (its split-K
    // partials and LayerNorm partials live there until adam_with_stats launches it): no reset, and above all no
    // free-and-regrow, until both are gone -- the join at the end of sgcn_step_run does it
    if (g_dw_recorded > 0 || reduce_parked()) return SGCN_OK;
    c.off = 0;
    if (c.want > c.cap) {                 // a call did not fit: grow for the next step (nothing is in flight on
        SGCN_HIP_TRY(hipStreamSynchronize(c.st));          // the aux stream once it is idle)
        if (c.ring) SGCN_HIP_TRY(hipFree(c.ring));
        c.ring = nullptr;
        c.cap = 0;
        const int64_t cap = c.want + c.want / 2;
        SGCN_HIP_TRY(hipMalloc((void**)&c.ring, (size_t)cap * sizeof(float)));
        c.cap = cap;
    }
    c.want = 0;
    return SGCN_OK;
}

struct BwdScratch { float* ws_ln; float* ws_dw; float* ws_gemm; hipStream_t st_dw; bool deferred; };

// ---- the step's weight-gradient GEMMs as ONE launch ------------------------------------------------------
// A layer's dW = x^T . g (+ its split-K reduction and LayerNorm-parameter reduction) depends on that layer's g only
// and is read by the optimizer only.  Run layer by layer beside the input-gradient chain (above) it costs the launching
// host four HIP calls per layer (event record + wait, GEMM, reduce) -- sixteen of the ~45 calls of a Reddit step, on a
// step whose period is set by the host on a slow box (BENCH_r02: 0.277 ms per step against 0.21 here).  Between
// dw_group_begin() and dw_group_flush() the layers' weight-gradient jobs are RECORDED instead (operands, K slicing
// and grids exactly as their own launches would have them; reduction scratch from the aux ring, one region per
// layer) and the flush issues all of them as one gemm_group_kernel + one dense_bwd_reduce_multi_kernel on the
// step's stream: two launches, no events, bit-identical gradients.
namespace {
struct DwJob { GemmArgs q; GemmPlan p; ReduceJob rj; const float* ws_ln; int32_t nblk, N; float* doffset; float* dscale; };
struct DwGroupState { DwJob jobs[kMaxGroup]; int n = 0; bool active = false; };
DwGroupState& dw_state() { static DwGroupState s; return s; }
}  // namespace

void grad_store_mode(int on) { g_grad_store = on; }

int dw_group_begin() {
    DwGroupState& d = dw_state();
    d.n = 0;
    d.active = true;
    return SGCN_OK;
}

template <int KG>
static void launch_group(const GemmGroup& G, unsigned blocks, hipStream_t st) {
    const size_t lds = (size_t)KG * kGroupFloats * sizeof(float);
    static bool raised = false;
    if (lds > 64 * 1024 && !raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_group_kernel<true, false, KG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = true;
    }
    hipLaunchKernelGGL((gemm_group_kernel<true, false, KG>), dim3(blocks), dim3(kBlock * KG), lds, st, G);
}

void dw_group_abort() {
    DwGroupState& d = dw_state();
    d.active = false;
    d.n = 0;
    g_dw_recorded = 0;
}

bool reduce_park(const ReduceMulti& R, int blocks);       // sgcn_dense.hip: the reductions ride in the optimizer's launch
int dw_group_flush(void* stream, bool park_reduce) {
    DwGroupState& d = dw_state();
    d.active = false;
    const int n = d.n;
    d.n = 0;
    g_dw_recorded = 0;
    if (n == 0) return SGCN_OK;
    hipStream_t st = (hipStream_t)stream;
    bool uniform = true;
    for (int k = 1; k < n; k++) uniform = uniform && d.jobs[k].p.kgroups == d.jobs[0].p.kgroups;
    if (uniform && n > 1) {
        GemmGroup G{};
        G.n = n;
        int first = 0;
        for (int k = 0; k < n; k++) {
            G.g[k] = d.jobs[k].q;
            G.first[k] = first;
            G.gx[k] = (int32_t)d.jobs[k].p.grid.x; G.gy[k] = (int32_t)d.jobs[k].p.grid.y;
            first += (int)(d.jobs[k].p.grid.x * d.jobs[k].p.grid.y * d.jobs[k].p.grid.z);
        }
        for (int k = n; k <= kMaxGroup; k++) G.first[k] = first;
        const int kg = d.jobs[0].p.kgroups;
        if (kg >= 4) launch_group<4>(G, (unsigned)first, st);
        else if (kg == 2) launch_group<2>(G, (unsigned)first, st);
        else launch_group<1>(G, (unsigned)first, st);
    } else {
        for (int k = 0; k < n; k++) launch_prepared(d.jobs[k].q, d.jobs[k].p, 1, 0, st);
    }
    ReduceMulti R{};
    R.n = n;
    R.ln_accumulate = g_grad_store ? 0 : 1;
    int b = 0;
    for (int k = 0; k < n; k++) {
        R.j[k] = d.jobs[k].rj;
        R.gfirst[k] = b;
        if (d.jobs[k].rj.pending) b += (int)(((int64_t)d.jobs[k].rj.M * d.jobs[k].rj.N + 255) / 256);
    }
    for (int k = n; k <= kMaxGroup; k++) R.gfirst[k] = b;
    for (int k = 0; k < n; k++) {
        R.ln_partial[k] = d.jobs[k].ws_ln; R.nblk[k] = d.jobs[k].nblk; R.d[k] = d.jobs[k].N;
        R.doffset[k] = d.jobs[k].doffset; R.dscale[k] = d.jobs[k].dscale;
        R.lfirst[k] = b;
        if (d.jobs[k].nblk > 0) b += (2 * d.jobs[k].N + kLnRedCols - 1) / kLnRedCols;
    }
    for (int k = n; k <= kMaxGroup; k++) R.lfirst[k] = b;
    if (b > 0 && !(park_reduce && reduce_park(R, b)))
        hipLaunchKernelGGL(dense_bwd_reduce_multi_kernel, dim3((unsigned)b), dim3(256), 0, st, R);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

// where a layer's reduction scratch lives and on which stream its weight-gradient side runs (overlapped: the aux
// ring and the aux stream -- only when there is an input-gradient GEMM to run beside)
static int bwd_scratch(const DenseBwdArgs& a, hipStream_t st, bool overlap, BwdScratch& s) {
    const int64_t ln_floats = a.scale ? (sgcn_ln_act_bwd_ws_floats(a.n, a.N) + 3) / 4 * 4 : 0;
    s.ws_gemm = a.ws ? a.ws + ln_floats : nullptr;
    s.ws_ln = a.ws;
    s.ws_dw = s.ws_gemm;
    s.st_dw = st;
    s.deferred = false;
    if (dw_state().active && dw_state().n < kMaxGroup) {
        // recorded for the group launch: the layer's reduction scratch must outlive this call -- its own region of the
        // aux ring (a ring that is still too small grows at the end of the step; until then the layer runs at once)
        AuxCtx& c = aux_ctx();
        const int rc0 = aux_init(c);
        if (rc0 != SGCN_OK) return rc0;
        const int64_t need = ln_floats + (a.ws ? (sgcn_gemm_ws_floats(a.K, a.N, a.n) + 3) / 4 * 4 : 0);
        c.want += need;
        if (c.off + need <= c.cap) {
            if (a.ws) { s.ws_ln = c.ring + c.off; s.ws_dw = c.ring + c.off + ln_floats; }
            c.off += need;
            s.deferred = true;
        }
        return SGCN_OK;
    }
    if (overlap && a.dx) {
        AuxCtx& c = aux_ctx();
        const int rc0 = aux_init(c);
        if (rc0 != SGCN_OK) return rc0;
        const int64_t need = ln_floats + (a.ws ? (sgcn_gemm_ws_floats(a.K, a.N, a.n) + 3) / 4 * 4 : 0);
        c.want += need;
        if (c.off + need <= c.cap) {
            if (a.ws) { s.ws_ln = c.ring + c.off; s.ws_dw = c.ring + c.off + ln_floats; }
            c.off += need;
            s.st_dw = c.st;
        }
    }
    return SGCN_OK;
}

// One dense layer's backward.
// `lower` / `slower`: the layer BELOW (its dy is this layer's dx), whose LayerNorm / ReLU backward pass rides behind this
// layer's input-gradient tail when both fit the row pass (*lower_done = true: only its weight-gradient job is left to do)
// Gradient-STORE mode (SGCN_OP_GRAD_STORE): a layer must WRITE its parameter gradients every step.  A degenerate call (an
// empty minibatch level) has no GEMM to do that, so its gradient ranges are zeroed here -- otherwise the optimizer would apply
// the previous step's values.
static int store_mode_zero(const DenseBwdArgs& a, void* stream) {
    if (!g_grad_store || a.N <= 0 || a.K <= 0) return SGCN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (a.dW) SGCN_HIP_TRY(hipMemset2DAsync(a.dW, (size_t)a.lddw * sizeof(float), 0, (size_t)a.N * sizeof(float), (size_t)a.K, st));
    if (a.doffset) SGCN_HIP_TRY(hipMemsetAsync(a.doffset, 0, (size_t)a.N * sizeof(float), st));
    if (a.dscale) SGCN_HIP_TRY(hipMemsetAsync(a.dscale, 0, (size_t)a.N * sizeof(float), st));
    return SGCN_OK;
}

static int dense_bwd_run(const DenseBwdArgs& a, const BwdScratch& s, void* stream, const DenseBwdArgs* lower = nullptr,
                         const BwdScratch* slower = nullptr, bool* lower_done = nullptr, int32_t* lower_nblk = nullptr) {
    SGCN_REQUIRE(a.n >= 0 && a.N >= 0 && a.K >= 0, "dense_bwd: negative size");
    if (a.n == 0 || a.N == 0 || a.K == 0) return store_mode_zero(a, stream);
    SGCN_REQUIRE(a.dy && a.x && a.W && a.dW, "dense_bwd: null operand");
    hipStream_t st = (hipStream_t)stream;
    const float* g = a.dy;
    int64_t ldg = a.lddy;
    int32_t nblk = 0;
    bool dx_in_row_pass = false;
    int dx_kg = 1;
    if (lower_done && !lower && *lower_done) {            // this layer's row pass has run behind the layer above's: g is in g_tmp
        nblk = lower_nblk ? *lower_nblk : 0;
        g = a.g_tmp; ldg = a.N;
    } else if (a.scale || a.relu) {
        SGCN_REQUIRE(a.g_tmp && a.y, "dense_bwd: LayerNorm / ReLU backward needs y and an n x N scratch");
        SGCN_REQUIRE(!a.scale || a.ws, "dense_bwd: LayerNorm backward needs the workspace");
        // the input gradient dx = g . W^T in the same row pass where the launch it replaces would have run as ONE chain per
        // output (no split over K) in two K-groups at most: sgcn_dense.hip LnBwdDx; knob step_fuse bit 3
        if (a.dx && (tune_get("step_fuse") & 8) && a.N <= 128 && a.N % 4 == 0 && a.K <= 256 && a.ldw == a.N && aligned16(a.W) &&
            (size_t)(8 * a.N + a.N * (a.K + 1)) * sizeof(float) <= 160 * 1024) {
            int S = 0, kgq = 0;
            const sgcn_dropout_t* dr = a.drop;
            if (dr && dr->keep >= 1.0f) dr = nullptr;
            if (dr) { S = 1; GemmArgs q{}; q.M = a.n; q.N = a.K; q.K = a.N; kgq = prepare_gemm(q, nullptr).kgroups; }   // a masked output is never split
            else gemm_fwd_shape(a.n, a.K, a.N, &S, &kgq, nullptr);
            if (S == 1 && kgq <= 2) dx_in_row_pass = true, dx_kg = kgq;
        }
        const bool chain = dx_in_row_pass && lower && slower && lower_done && a.K <= 128 && lower->dy == a.dx && lower->lddy == a.lddx &&
                           lower->n == a.n && lower->N == a.K && (lower->scale || lower->relu) && lower->g_tmp && lower->y &&
                           (!lower->scale || (lower->xhat && lower->rstd && slower->ws_ln)) &&
                           (size_t)(8 * a.N + a.N * (a.K + 1) + 8 * a.K) * sizeof(float) <= 160 * 1024;
        const int rc = ln_act_bwd_launch(a.dy, a.lddy, a.y, a.ldy, a.xhat, a.rstd, a.scale, a.n, a.N, a.relu, a.g_tmp, a.N,
                                         a.doffset, a.dscale, s.ws_ln, /*reduce_params=*/false, &nblk, st,
                                         dx_in_row_pass ? a.W : nullptr, a.K, dx_kg, a.drop, a.dx, a.lddx,
                                         chain ? lower->y : nullptr, chain ? lower->ldy : 0, chain ? lower->xhat : nullptr,
                                         chain ? lower->rstd : nullptr, chain ? lower->scale : nullptr, chain ? lower->relu : 0,
                                         chain ? lower->g_tmp : nullptr, chain ? slower->ws_ln : nullptr);
        if (rc != SGCN_OK) return rc;
        if (chain) {
            *lower_done = true;
            if (lower_nblk) *lower_nblk = (lower->scale && a.scale) ? nblk : (lower->scale ? (a.n + 3) / 4 : 0);
        }
        g = a.g_tmp; ldg = a.N;
    }
    if (s.st_dw != st) {                  // fork: the aux stream sees everything the main stream did so far
        AuxCtx& c = aux_ctx();
        SGCN_HIP_TRY(hipEventRecord(c.fork, st));
        SGCN_HIP_TRY(hipStreamWaitEvent(c.st, c.fork, 0));
        c.pending = true;
    }
    // dW[K x N] += x^T[K x n] . g[n x N]      (x stored [n x K]: trans_a); its split-K reduction and the
    // LayerNorm parameter reduction share one launch
    GemmArgs q{};
    q.A = a.x; q.lda = a.ldx; q.B = g; q.ldb = ldg; q.C = a.dW; q.ldc = a.lddw;
    q.M = a.K; q.N = a.N; q.K = a.n; q.accumulate = g_grad_store ? 0 : 1; q.epi = 0;
    q.a_gidx = a.gidx;                     // x rows gathered on the fly (x = features, gidx = the field)
    q.drop_a = drop_args(a.drop);
    SGCN_REQUIRE(!q.drop_a.on || q.drop_a.width == a.K, "dense_bwd: dropout width must be K");
    ReduceJob job{};
    int rc = SGCN_OK;
    if (s.deferred) {
        DwGroupState& dg = dw_state();
        DwJob& J = dg.jobs[dg.n++];
        g_dw_recorded = dg.n;
        J.q = q;
        J.p = prepare_gemm(J.q, s.ws_dw, 1, 0);
        J.rj = ReduceJob{J.q.ws, J.p.S, J.q.M, J.q.N, J.q.C, J.q.ldc, J.q.accumulate, J.p.S > 1 ? 1 : 0};
        J.ws_ln = s.ws_ln; J.nblk = nblk; J.N = a.N; J.doffset = a.doffset; J.dscale = a.dscale;
    } else {
        rc = launch_gemm(q, 1, 0, s.ws_dw, s.st_dw, &job);
        if (rc != SGCN_OK) return rc;
    }
    if (!s.deferred && (job.pending || nblk > 0)) {
        const int gb = job.pending ? (int)(((int64_t)job.M * job.N + 255) / 256) : 0;
        const int lb = nblk > 0 ? (2 * a.N + kLnRedCols - 1) / kLnRedCols : 0;
        hipLaunchKernelGGL(dense_bwd_reduce_kernel, dim3((unsigned)(gb + lb)), dim3(256), 0, s.st_dw, job, gb, s.ws_ln, nblk, a.N,
                           a.doffset, a.dscale, g_grad_store ? 0 : 1);
        SGCN_HIP_TRY(hipGetLastError());
    }
    if (!a.dx || dx_in_row_pass) return SGCN_OK;
    // dx[n x K] = g[n x N] . W^T              (W stored [K x N]: trans_b)
    return sgcn_gemm_f32(0, 1, a.n, a.K, a.N, g, ldg, a.W, a.ldw, a.dx, a.lddx, 0, s.ws_gemm, nullptr, a.drop, stream);
}

static int dense_bwd_impl(int32_t n, int32_t N, int32_t K, const float* dy, int64_t lddy,
                          const float* y, int64_t ldy, const float* xhat, const float* rstd,
                          const float* scale, int32_t relu, const float* x, int64_t ldx,
                          const float* W, int64_t ldw, float* dW, int64_t lddw, float* doffset,
                          float* dscale, float* dx, int64_t lddx, const sgcn_dropout_t* drop,
                          float* g_tmp, float* ws, const int32_t* gidx, void* stream, bool overlap) {
    const DenseBwdArgs a{n, N, K, dy, lddy, y, ldy, xhat, rstd, scale, relu, x, ldx, W, ldw, dW, lddw, doffset, dscale,
                         dx, lddx, drop, g_tmp, ws, gidx};
    if (n == 0 || N == 0 || K == 0)
        return n >= 0 && N >= 0 && K >= 0 ? store_mode_zero(a, stream) : fail(SGCN_ERR_INVALID, "dense_bwd: negative size");
    BwdScratch s{};
    const int rc = bwd_scratch(a, (hipStream_t)stream, overlap, s);
    if (rc != SGCN_OK) return rc;
    return dense_bwd_run(a, s, stream);
}

// Two consecutive layers' backward (upper, then the layer below it) with the lower one's LayerNorm / ReLU backward pass
// behind the upper one's row pass when the shapes allow it (knob step_fuse bit 6); otherwise exactly the two calls.
int dense_bwd_chain(const DenseBwdArgs& up, const DenseBwdArgs& lo, void* stream) {
    const bool overlap = tune_get("step_overlap") != 0;
    if (up.n == 0 || up.N == 0 || up.K == 0 || lo.n == 0 || lo.N == 0 || lo.K == 0) {
        // degenerate (the caller checked sizes are >= 0): no GEMM for either layer; in gradient-STORE mode both still write
        const int rz = store_mode_zero(up, stream);
        return rz != SGCN_OK ? rz : store_mode_zero(lo, stream);
    }
    BwdScratch su{}, sl{};
    int rc = bwd_scratch(up, (hipStream_t)stream, overlap, su);
    if (rc != SGCN_OK) return rc;
    rc = bwd_scratch(lo, (hipStream_t)stream, overlap, sl);
    if (rc != SGCN_OK) return rc;
    bool lower_done = false;
    int32_t lower_nblk = 0;
    // (both layers' scratch must be their own regions of the ring: the shared per-call buffer of the undeferred form would
    // hold two layers' LayerNorm partials at once)
    const bool try_chain = (tune_get("step_fuse") & 64) && !lo.dx && su.deferred && sl.deferred;
    rc = dense_bwd_run(up, su, stream, try_chain ? &lo : nullptr, try_chain ? &sl : nullptr, &lower_done, &lower_nblk);
    if (rc != SGCN_OK) return rc;
    return dense_bwd_run(lo, sl, stream, nullptr, nullptr, &lower_done, &lower_nblk);
}

int dense_bwd_overlapped(int32_t n, int32_t N, int32_t K, const float* dy, int64_t lddy, const float* y, int64_t ldy,
                         const float* xhat, const float* rstd, const float* scale, int32_t relu, const float* x,
                         int64_t ldx, const float* W, int64_t ldw, float* dW, int64_t lddw, float* doffset,
                         float* dscale, float* dx, int64_t lddx, const sgcn_dropout_t* drop, float* g_tmp, float* ws,
                         const int32_t* gidx, void* stream) {
    return dense_bwd_impl(n, N, K, dy, lddy, y, ldy, xhat, rstd, scale, relu, x, ldx, W, ldw, dW, lddw, doffset, dscale,
                          dx, lddx, drop, g_tmp, ws, gidx, stream, tune_get("step_overlap") != 0);
}
}  // namespace sgcn

extern "C" int sgcn_dense_bwd_f32(int32_t n, int32_t N, int32_t K, const float* dy, int64_t lddy,
                                  const float* y, int64_t ldy, const float* xhat, const float* rstd,
                                  const float* scale, int32_t relu, const float* x, int64_t ldx,
                                  const float* W, int64_t ldw, float* dW, int64_t lddw, float* doffset,
                                  float* dscale, float* dx, int64_t lddx, const sgcn_dropout_t* drop,
                                  float* g_tmp, float* ws, const int32_t* gidx, void* stream) {
    return sgcn::dense_bwd_impl(n, N, K, dy, lddy, y, ldy, xhat, rstd, scale, relu, x, ldx, W, ldw, dW, lddw, doffset,
                                dscale, dx, lddx, drop, g_tmp, ws, gidx, stream, false);
}
