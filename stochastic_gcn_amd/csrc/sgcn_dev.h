// Device-side helpers shared by the HIP translation units of libsgcn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "sgcn_host.h"
#include "../../include/sgcn.h"

#define SGCN_HIP_TRY(expr)                                                              \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess)                                                          \
            return sgcn::fail(SGCN_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__));   \
    } while (0)

#define SGCN_REQUIRE(cond, ...)                                       \
    do {                                                              \
        if (!(cond)) return sgcn::fail(SGCN_ERR_INVALID, __VA_ARGS__); \
    } while (0)

namespace sgcn {

constexpr int kWave = 64;      // gfx950 wavefront
constexpr int kBlock = 256;    // 4 wavefronts per workgroup, one per SIMD

template <int VW> struct Vec;
template <> struct Vec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<1> { typedef float type; };

template <int VW>
__device__ __forceinline__ typename Vec<VW>::type vzero() {
    typename Vec<VW>::type z = {};
    return z;
}
template <int VW>
__device__ __forceinline__ typename Vec<VW>::type vload(const float* p) {
    return *reinterpret_cast<const typename Vec<VW>::type*>(p);
}
template <int VW>
__device__ __forceinline__ void vstore(float* p, typename Vec<VW>::type v) {
    *reinterpret_cast<typename Vec<VW>::type*>(p) = v;
}
template <int VW>
__device__ __forceinline__ float velem(const typename Vec<VW>::type& v, int e) {
    if constexpr (VW == 1) return v; else return v[e];
}
// store the first `cnt` (< VW) elements only: the ragged tail when d % VW != 0
template <int VW>
__device__ __forceinline__ void vstore_head(float* p, typename Vec<VW>::type v, int cnt) {
#pragma unroll
    for (int e = 0; e < VW; e++)
        if (e < cnt) p[e] = velem<VW>(v, e);
}

// Broadcast lane `j` of a G-lane group.  G == 64: the source lane is wave-uniform, so
// v_readlane_b32 puts the value in an SGPR and the gathered row base becomes a scalar
// address (global_load ... s[base:base+1]).  G < 64: ds_bpermute inside the group.
template <int G>
__device__ __forceinline__ int bcast_i(int x, int j) {
    if constexpr (G == kWave) return __builtin_amdgcn_readlane(x, j);
    else return __shfl(x, j, G);
}
template <int G>
__device__ __forceinline__ float bcast_f(float x, int j) {
    if constexpr (G == kWave) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), j));
    else return __shfl(x, j, G);
}
template <int G>
__device__ __forceinline__ int uniform_i(int x) {
    if constexpr (G == kWave) return __builtin_amdgcn_readfirstlane(x);
    else return x;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

// Pick the widest vector the operands allow.  A ragged last vector may over-READ up to
// VW-1 floats inside the row pitch (never used) and is stored element-wise.
inline int pick_vw(int d, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds) {
    auto ok = [&](int vw) {
        const int64_t span = ((int64_t)d + vw - 1) / vw * vw;
        for (int64_t ld : lds)
            if (ld % vw != 0 || ld < span) return false;
        for (const void* p : ptrs) {
            if (!p) continue;
            if (vw == 4 && !aligned16(p)) return false;
            if (vw == 2 && !aligned8(p)) return false;
        }
        return true;
    };
    if (ok(4)) return 4;
    if (ok(2)) return 2;
    return 1;
}

int tune_get(const char* key);  // sgcn_spmm.hip
// the calling thread's override of the step_overlap / step_fuse knobs (-1: none); sgcn_step_run sets it for one run
void step_mode_override(int overlap, int fuse);

// ---- counter-based dropout (include/sgcn.h sgcn_dropout_t) ---------------------------------------
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {      // murmur3 finaliser
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

struct DropArgs {            // device-side form of sgcn_dropout_t
    uint32_t key, thr;       // keep iff fmix32(idx * 0x9E3779B1 + key) < thr
    float scale;             // 1 / keep
    int32_t rows, width;
    int32_t on;
};

inline DropArgs drop_args(const sgcn_dropout_t* d) {
    DropArgs a{};
    if (!d || d->keep >= 1.0f) return a;
    a.on = 1; a.key = d->key; a.rows = d->rows < 0 ? 0x7fffffff : d->rows; a.width = d->width;
    const double t = (double)d->keep * 4294967296.0;
    a.thr = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    a.scale = 1.0f / d->keep;
    return a;
}

__device__ __forceinline__ float drop_factor(const DropArgs& a, int row, int col) {
    if (row >= a.rows) return 1.0f;
    const uint32_t idx = (uint32_t)row * (uint32_t)a.width + (uint32_t)col;
    return fmix32(idx * 0x9E3779B1u + a.key) < a.thr ? a.scale : 0.0f;
}

// ---- LayerNorm parameter gradients: 32 columns of the [d(offset) | d(scale)] vector from the per-workgroup
// partials of ln_act_bwd_kernel.  One 256-thread workgroup per 32 columns: 8 thread rows each add a
// contiguous eighth of the partials (two independent chains keep loads in flight), then one thread per
// column adds the eight sums in order -- a fixed summation order whatever the schedule.  (One thread per
// column walking all partials took 13-19 us once ln_act_bwd_kernel ran one row per wavefront.)  Shared by
// the standalone reduce and the dense-layer backward's combined reduce (sgcn_gemm.hip).
constexpr int kLnRedCols = 32;

// One element of tf.train.AdamOptimizer's update (the ONE copy of this arithmetic: the stand-alone kernel, the optimizer
// launch of the step program and the reductions that feed it all go through here, so that they agree bit for bit).
struct AdamArgs { float* theta; const float* grad; float* m; float* v; int64_t n; float lr_t, b1, b2, eps; };
__device__ __forceinline__ void adam_one(const AdamArgs& A, int64_t i, float g) {
    const float mi = A.b1 * A.m[i] + (1.f - A.b1) * g;
    const float vi = A.b2 * A.v[i] + (1.f - A.b2) * g * g;
    A.m[i] = mi; A.v[i] = vi;
    A.theta[i] -= A.lr_t * mi / (sqrtf(vi) + A.eps);
}

// `adam` != nullptr: the gradient element just reduced is consumed by the optimizer on the spot (it lives in A->grad)
__device__ __forceinline__ void ln_param_reduce_cols32(const float* __restrict__ partial, int nblk, int d,
                                                       float* __restrict__ doffset, float* __restrict__ dscale,
                                                       int cblock, bool accumulate = true, const AdamArgs* adam = nullptr) {
    __shared__ float red[8][kLnRedCols];
    const int lane = threadIdx.x & (kLnRedCols - 1), part = threadIdx.x >> 5;      // 256 threads: 8 x 32
    const int c = cblock * kLnRedCols + lane;
    const int chunk = (nblk + 7) / 8;
    const int b0 = part * chunk, b1 = min(nblk, b0 + chunk);
    float s0 = 0.f, s1 = 0.f;
    if (c < 2 * d) {
        int b = b0;
        for (; b + 2 <= b1; b += 2) {
            s0 += partial[(size_t)(b + 0) * 2 * d + c];
            s1 += partial[(size_t)(b + 1) * 2 * d + c];
        }
        if (b < b1) s0 += partial[(size_t)b * 2 * d + c];
    }
    red[part][lane] = s0 + s1;
    __syncthreads();
    if (part == 0 && c < 2 * d) {
        float s = red[0][lane];
#pragma unroll
        for (int q = 1; q < 8; q++) s += red[q][lane];
        float* p = c < d ? doffset + c : dscale + (c - d);
        const float g = accumulate ? *p + s : s;
        *p = g;
        if (adam) adam_one(*adam, p - adam->grad, g);
    }
}

// The reductions behind the step's grouped weight-gradient launch (sgcn_gemm.hip): workgroups [gfirst[j], gfirst[j + 1]) add
// job j's split-K partial tiles, workgroups [lfirst[j], lfirst[j + 1]) its LayerNorm parameter partials.
constexpr int kMaxGroup = 6;
struct ReduceJob {                // a split-K reduction the caller wants to launch itself
    const float* ws; int32_t S, M, N; float* C; int64_t ldc; int32_t accumulate; int32_t pending;
};
struct ReduceMulti {
    ReduceJob j[kMaxGroup];
    const float* ln_partial[kMaxGroup];
    float* doffset[kMaxGroup];
    float* dscale[kMaxGroup];
    int32_t nblk[kMaxGroup], d[kMaxGroup];
    int32_t gfirst[kMaxGroup + 1], lfirst[kMaxGroup + 1];
    int32_t n, ln_accumulate;
};
__device__ __forceinline__ void reduce_multi_body(const ReduceMulti& R, int b, const AdamArgs* adam) {
    if (b < R.gfirst[R.n]) {
        int q = 0;
#pragma unroll
        for (int t = 1; t < kMaxGroup; t++) q += (t < R.n && b >= R.gfirst[t]) ? 1 : 0;
        const ReduceJob& j = R.j[q];
        const int64_t i = (int64_t)(b - R.gfirst[q]) * blockDim.x + threadIdx.x;
        const int64_t mn = (int64_t)j.M * j.N;
        if (i >= mn) return;
        float s = 0.f;
        for (int z = 0; z < j.S; z++) s += j.ws[(int64_t)z * mn + i];
        float* p = j.C + (i / j.N) * j.ldc + (i % j.N);
        const float g = j.accumulate ? *p + s : s;
        *p = g;
        if (adam) adam_one(*adam, p - adam->grad, g);
    } else {
        int q = 0;
#pragma unroll
        for (int t = 1; t < kMaxGroup; t++) q += (t < R.n && b >= R.lfirst[t]) ? 1 : 0;
        ln_param_reduce_cols32(R.ln_partial[q], R.nblk[q], R.d[q], R.doffset[q], R.dscale[q], b - R.lfirst[q], R.ln_accumulate != 0, adam);
    }
}

// sgcn_dense.hip: LN/ReLU backward; with reduce_params == false the parameter-gradient partials stay
// in ws ([*nblk][2][d]) for the caller to reduce.
int ln_act_bwd_launch(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* xhat,
                      const float* rstd, const float* scale, int32_t n, int32_t d, int32_t relu, float* dx,
                      int64_t lddx, float* doffset, float* dscale, float* ws, bool reduce_params, int32_t* nblk,
                      hipStream_t st, const float* tail_W = nullptr, int32_t tail_K = 0, int32_t tail_kg = 1,
                      const sgcn_dropout_t* tail_drop = nullptr, float* tail_dx = nullptr, int64_t tail_lddx = 0,
                      const float* nx_y = nullptr, int64_t nx_ldy = 0, const float* nx_xhat = nullptr, const float* nx_rstd = nullptr,
                      const float* nx_scale = nullptr, int32_t nx_relu = 0, float* nx_g = nullptr, float* nx_partial = nullptr);

}  // namespace sgcn
