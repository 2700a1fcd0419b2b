// Deterministic dropout ("det-dropout", --det_dropout): the moment-propagation variant of the reference -- every
// activation is a pair (mean, variance) under the dropout noise, pushed through the layers analytically instead of
// sampled (gcn/layers.py:141-202 DetDropoutFC, :236-248 / :320-349 the aggregators' tuple branches, :425-428 the
// Gaussian re-sampling in front of the last dense layers).  The matrix products of the variant are the
// library's own GEMM / SpMM kernels; this file holds its element-wise and row-wise pieces, forward and backward, as the
// reference's formulas state them (autodiff of those formulas for the backward).  Launch-latency-bound at minibatch
// size like the other dense-side kernels (sgcn_dense.hip): plain grid-stride kernels, fixed summation orders.
#include "sgcn_dev.h"

#include <cmath>

namespace sgcn {
namespace {

constexpr float kInvSqrt2Pi = 0.3989422804014327f, kInvSqrt2 = 0.7071067811865476f;

__device__ __forceinline__ float npdf(float x) { return kInvSqrt2Pi * __expf(-0.5f * x * x); }
__device__ __forceinline__ float ncdf(float x) { return 0.5f * erfcf(-x * kInvSqrt2); }       // tf Normal.cdf

// ---- dropout moments (gcn/layers.py:168-176): var' = (var + mu^2) / p - mu^2 = var / p + (1/p - 1) mu^2 (var absent: 0) ----
__global__ __launch_bounds__(kBlock) void det_pre_kernel(const float* __restrict__ mu, const float* __restrict__ var, int64_t n,
                                                         float inv_p, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float m = mu[i];
        out[i] = (var ? var[i] * inv_p : 0.f) + (inv_p - 1.f) * m * m;
    }
}
__global__ __launch_bounds__(kBlock) void det_pre_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ g, int64_t n,
                                                             float inv_p, float* __restrict__ d_mu, float* __restrict__ d_var) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float gi = g[i];
        d_mu[i] += gi * 2.f * (inv_p - 1.f) * mu[i];
        if (d_var) d_var[i] = gi * inv_p;
    }
}

// y = c * x^2 ;  acc += c * a * b
__global__ __launch_bounds__(kBlock) void square_kernel(const float* __restrict__ x, int64_t n, float c, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) y[i] = c * x[i] * x[i];
}
__global__ __launch_bounds__(kBlock) void addmul_kernel(float* __restrict__ acc, const float* __restrict__ a, const float* __restrict__ b,
                                                        int64_t n, float c) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) acc[i] += c * a[i] * b[i];
}

// ---- LayerNorm, variance stream (gcn/layers.py:184-188): var2 = var1 * scale^2 / variance(mu1 row) -----------------------
// the row's variance comes back out of the mean stream's rstd = rsqrt(variance + eps)
__global__ __launch_bounds__(kBlock) void det_lnvar_fwd_kernel(const float* __restrict__ var1, const float* __restrict__ rstd,
                                                               const float* __restrict__ scale, int32_t n, int32_t d, float eps,
                                                               float* __restrict__ var2) {
    const int64_t total = (int64_t)n * d;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int r = (int)(i / d), c = (int)(i % d);
        const float rs = rstd[r], iv = 1.0f / (1.0f / (rs * rs) - eps), s = scale[c];
        var2[i] = var1[i] * (s * s) * iv;
    }
}
// one wavefront per row: d_var1 = g s^2 / V;  d_mu1 += dV * 2 (mu1 - mean) / d  with  dV = -sum_c g var1 s^2 / V^2  and
// (mu1 - mean) = xhat / rstd;  tmp[r][c] = g var1 2 s / V (its column sums are d(scale), added by colsum_add_kernel)
__global__ __launch_bounds__(kBlock) void det_lnvar_bwd_kernel(const float* __restrict__ g, const float* __restrict__ var1,
                                                               const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                               const float* __restrict__ scale, int32_t n, int32_t d, float eps,
                                                               float* __restrict__ d_var1, float* __restrict__ d_mu1,
                                                               float* __restrict__ tmp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (row >= n) return;
    const float rs = rstd[row], iv = 1.0f / (1.0f / (rs * rs) - eps);
    float acc = 0.f;
    for (int c = lane; c < d; c += kWave) {
        const int64_t i = row * d + c;
        const float s = scale[c], gi = g[i], v1 = var1[i];
        d_var1[i] = gi * s * s * iv;
        tmp[i] = gi * v1 * 2.f * s * iv;
        acc += gi * v1 * s * s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    const float dV = -acc * iv * iv;
    const float k = dV * 2.f / ((float)d * rs);
    for (int c = lane; c < d; c += kWave) d_mu1[row * d + c] += k * xhat[row * d + c];
}
// out[c] += sum_r x[r][c], rows in order (one thread per column)
__global__ __launch_bounds__(kBlock) void colsum_add_kernel(const float* __restrict__ x, int32_t n, int32_t d, float* __restrict__ out) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= d) return;
    float s = 0.f;
    for (int r = 0; r < n; r++) s += x[(int64_t)r * d + c];
    out[c] += s;
}

// ---- ReLU by moment matching (gcn/layers.py:190-202) ---------------------------------------------------------------------
struct ReluM { float sigma, alpha, phi, Phi, Z, r, m, mo, q, t; };
__device__ __forceinline__ ReluM relu_moments(float mu, float v) {
    ReluM x;
    x.sigma = sqrtf(v);
    x.alpha = -mu / x.sigma;
    x.phi = npdf(x.alpha);
    x.Phi = ncdf(x.alpha);
    x.Z = ncdf(-x.alpha) + 1e-10f;
    x.r = x.phi / x.Z;
    x.m = mu + x.sigma * x.r;
    x.mo = x.Z * x.m;
    x.q = 1.f + x.alpha * x.r - x.r * x.r;
    x.t = v * x.q;
    return x;
}
__global__ __launch_bounds__(kBlock) void det_relu_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ var, int64_t n,
                                                              float* __restrict__ mu_out, float* __restrict__ var_out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const ReluM x = relu_moments(mu[i], var[i]);
        const float vr = fmaxf(x.t, 0.f) + 1e-10f;
        mu_out[i] = x.mo;
        var_out[i] = x.Z * vr + x.Z * x.Phi * x.mo * x.mo;
    }
}
__global__ __launch_bounds__(kBlock) void det_relu_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                              const float* __restrict__ g_mo, const float* __restrict__ g_vo, int64_t n,
                                                              float* __restrict__ d_mu, float* __restrict__ d_var) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float m_ = mu[i], v = var[i];
        const ReluM x = relu_moments(m_, v);
        const float s = x.sigma, a = x.alpha, phi = x.phi, r = x.r;
        const float a_mu = -1.f / s, a_s = -a / s;                      // d alpha / d mu, d alpha / d sigma
        const float dr = -a * r + r * r;                               // d r / d alpha
        const float Z_mu = phi / s, Z_s = a * phi / s;                 // Z = cdf(-alpha) + eps: dZ/dalpha = -phi
        const float m_mu = 1.f - dr, m_s = r - a * dr;
        const float mo_mu = Z_mu * x.m + x.Z * m_mu, mo_s = Z_s * x.m + x.Z * m_s;
        const float dq = r + a * dr - 2.f * r * dr;
        const float t_mu = -s * dq, t_s = 2.f * s * x.q - s * a * dq;  // t = sigma^2 q
        const float gate = x.t > 0.f ? 1.f : 0.f;
        const float vr = fmaxf(x.t, 0.f) + 1e-10f;
        const float P_mu = phi * a_mu, P_s = phi * a_s;                // d Phi
        const float mo2 = x.mo * x.mo;
        const float vo_mu = Z_mu * vr + x.Z * gate * t_mu + (Z_mu * x.Phi + x.Z * P_mu) * mo2 + 2.f * x.Z * x.Phi * x.mo * mo_mu;
        const float vo_s = Z_s * vr + x.Z * gate * t_s + (Z_s * x.Phi + x.Z * P_s) * mo2 + 2.f * x.Z * x.Phi * x.mo * mo_s;
        const float gm = g_mo[i], gv = g_vo[i];
        d_mu[i] = gm * mo_mu + gv * vo_mu;
        d_var[i] = (gm * mo_s + gv * vo_s) / (2.f * s);
    }
}

// ---- Gaussian re-sampling in front of a Dropout that receives (mu, var) (gcn/layers.py:425-428) ------------------------
// x = mu + eps * sqrt(var + 1e-10), eps ~ N(0, 1): Box-Muller on two counter-based hashes of the element index, so that
// (like the dropout masks, sgcn_dropout_t) the noise is a pure function of (key, index) the oracle can replay
__host__ __device__ __forceinline__ float gauss_of(uint32_t idx, uint32_t key) {
    const uint32_t h1 = fmix32(idx * 0x9E3779B1u + key), h2 = fmix32((idx * 0x85EBCA6Bu + 0x165667B1u) ^ key);
    const float u1 = ((float)(h1 >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = ((float)(h2 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}
__global__ __launch_bounds__(kBlock) void gauss_sample_kernel(const float* __restrict__ mu, const float* __restrict__ var, int64_t n,
                                                              uint32_t key, float* __restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        x[i] = mu[i] + gauss_of((uint32_t)i, key) * sqrtf(var[i] + 1e-10f);
}
__global__ __launch_bounds__(kBlock) void gauss_sample_bwd_kernel(const float* __restrict__ var, const float* __restrict__ g, int64_t n,
                                                                  uint32_t key, float* __restrict__ d_var) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        d_var[i] = g[i] * gauss_of((uint32_t)i, key) * 0.5f / sqrtf(var[i] + 1e-10f);
}

// ---- control-variate aggregator on (mu, var) (gcn/layers.py:320-349): the element-wise operands of its five SpMMs --------
//   delta_mu = mu - Hm[if]   ds = sqrt(var) - sqrt(Hv[if])   ds2 = ds^2   msig2 = 2 ds sqrt(Hv[if])
__global__ __launch_bounds__(kBlock) void det_agg_prep_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                              const float* __restrict__ Hm, const float* __restrict__ Hv, int64_t ldh,
                                                              const int32_t* __restrict__ ifield, int32_t n0, int32_t d,
                                                              float* __restrict__ delta_mu, float* __restrict__ ds2,
                                                              float* __restrict__ msig2, float* __restrict__ ds, float* __restrict__ sbar) {
    const int64_t total = (int64_t)n0 * d;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int r = (int)(i / d), c = (int)(i % d);
        const int64_t h = (int64_t)ifield[r] * ldh + c;
        const float sb = sqrtf(Hv[h]), dsi = sqrtf(var[i]) - sb;
        delta_mu[i] = mu[i] - Hm[h];
        ds[i] = dsi; sbar[i] = sb;
        ds2[i] = dsi * dsi;
        msig2[i] = 2.f * dsi * sb;
    }
}
//   d_var = (2 ds g_ds2 + 2 sbar g_msig2) / (2 sqrt(var))
__global__ __launch_bounds__(kBlock) void det_agg_prep_bwd_kernel(const float* __restrict__ var, const float* __restrict__ ds,
                                                                  const float* __restrict__ sbar, const float* __restrict__ g_ds2,
                                                                  const float* __restrict__ g_msig2, int32_t n0, int32_t d,
                                                                  const float* __restrict__ add, int64_t ldadd, int32_t add_rows,
                                                                  float* __restrict__ d_var) {
    const int64_t n = (int64_t)n0 * d;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        float v = (2.f * ds[i] * g_ds2[i] + 2.f * sbar[i] * g_msig2[i]) / (2.f * sqrtf(var[i]));
        const int64_t r = i / d;
        if (r < add_rows) v += add[r * ldadd + (i - r * d)];        // the self half of a concat aggregator's gradient
        d_var[i] = v;
    }
}
// y = relu(x) + eps (in place on a strided block);   g <- (raw > 0) ? g : 0
__global__ __launch_bounds__(kBlock) void relu_eps_kernel(const float* __restrict__ raw, int64_t ldr, int32_t n, int32_t d, float eps,
                                                          float* __restrict__ y, int64_t ldy) {
    const int64_t total = (int64_t)n * d;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / d, c = i % d;
        y[r * ldy + c] = fmaxf(raw[r * ldr + c], 0.f) + eps;
    }
}
__global__ __launch_bounds__(kBlock) void gate_kernel(const float* __restrict__ raw, int64_t ldr, const float* __restrict__ g, int64_t ldg,
                                                      int32_t n, int32_t d, float* __restrict__ out) {
    const int64_t total = (int64_t)n * d;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / d, c = i % d;
        out[i] = raw[r * ldr + c] > 0.f ? g[r * ldg + c] : 0.f;
    }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 4096); }

}  // namespace
}  // namespace sgcn

using namespace sgcn;

#define SGCN_DET_LAUNCH(kernel, n, ...)                                                                    \
    do {                                                                                                   \
        if ((n) > 0) hipLaunchKernelGGL(kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, __VA_ARGS__); \
        SGCN_HIP_TRY(hipGetLastError());                                                                   \
        return SGCN_OK;                                                                                    \
    } while (0)

extern "C" int sgcn_det_pre_f32(const float* mu, const float* var, int64_t n, float keep, float* var_out, void* stream) {
    SGCN_REQUIRE(n >= 0 && keep > 0.f && (n == 0 || (mu && var_out)), "det_pre: bad operand");
    SGCN_DET_LAUNCH(det_pre_kernel, n, mu, var, n, 1.0f / keep, var_out);
}
extern "C" int sgcn_det_pre_bwd_f32(const float* mu, const float* d_var_out, int64_t n, float keep, float* d_mu, float* d_var,
                                    void* stream) {
    SGCN_REQUIRE(n >= 0 && keep > 0.f && (n == 0 || (mu && d_var_out && d_mu)), "det_pre_bwd: bad operand");
    SGCN_DET_LAUNCH(det_pre_bwd_kernel, n, mu, d_var_out, n, 1.0f / keep, d_mu, d_var);
}
extern "C" int sgcn_square_f32(const float* x, int64_t n, float c, float* y, void* stream) {
    SGCN_REQUIRE(n >= 0 && (n == 0 || (x && y)), "square: bad operand");
    SGCN_DET_LAUNCH(square_kernel, n, x, n, c, y);
}
extern "C" int sgcn_addmul_f32(float* acc, const float* a, const float* b, int64_t n, float c, void* stream) {
    SGCN_REQUIRE(n >= 0 && (n == 0 || (acc && a && b)), "addmul: bad operand");
    SGCN_DET_LAUNCH(addmul_kernel, n, acc, a, b, n, c);
}
extern "C" int sgcn_det_lnvar_fwd_f32(const float* var1, const float* rstd, const float* scale, int32_t n, int32_t d, float eps,
                                      float* var2, void* stream) {
    SGCN_REQUIRE(n >= 0 && d >= 0 && (n == 0 || d == 0 || (var1 && rstd && scale && var2)), "det_lnvar_fwd: bad operand");
    SGCN_DET_LAUNCH(det_lnvar_fwd_kernel, (int64_t)n * d, var1, rstd, scale, n, d, eps, var2);
}
extern "C" int sgcn_det_lnvar_bwd_f32(const float* d_var2, const float* var1, const float* xhat, const float* rstd, const float* scale,
                                      int32_t n, int32_t d, float eps, float* d_var1, float* d_mu1, float* dscale, float* tmp,
                                      void* stream) {
    SGCN_REQUIRE(n >= 0 && d >= 0, "det_lnvar_bwd: negative size");
    if (n == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(d_var2 && var1 && xhat && rstd && scale && d_var1 && d_mu1 && dscale && tmp, "det_lnvar_bwd: null operand");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(det_lnvar_bwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), 0, st, d_var2, var1, xhat, rstd, scale, n, d,
                       eps, d_var1, d_mu1, tmp);
    hipLaunchKernelGGL(colsum_add_kernel, dim3((unsigned)((d + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, tmp, n, d, dscale);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}
extern "C" int sgcn_det_relu_fwd_f32(const float* mu, const float* var, int64_t n, float* mu_out, float* var_out, void* stream) {
    SGCN_REQUIRE(n >= 0 && (n == 0 || (mu && var && mu_out && var_out)), "det_relu_fwd: bad operand");
    SGCN_DET_LAUNCH(det_relu_fwd_kernel, n, mu, var, n, mu_out, var_out);
}
extern "C" int sgcn_det_relu_bwd_f32(const float* mu, const float* var, const float* d_mu_out, const float* d_var_out, int64_t n,
                                     float* d_mu, float* d_var, void* stream) {
    SGCN_REQUIRE(n >= 0 && (n == 0 || (mu && var && d_mu_out && d_var_out && d_mu && d_var)), "det_relu_bwd: bad operand");
    SGCN_DET_LAUNCH(det_relu_bwd_kernel, n, mu, var, d_mu_out, d_var_out, n, d_mu, d_var);
}
extern "C" int sgcn_gauss_sample_f32(const float* mu, const float* var, int64_t n, uint32_t key, float* x, void* stream) {
    SGCN_REQUIRE(n >= 0 && n < (1ll << 32) && (n == 0 || (mu && var && x)), "gauss_sample: bad operand");
    SGCN_DET_LAUNCH(gauss_sample_kernel, n, mu, var, n, key, x);
}
extern "C" int sgcn_gauss_sample_bwd_f32(const float* var, const float* g, int64_t n, uint32_t key, float* d_var, void* stream) {
    SGCN_REQUIRE(n >= 0 && n < (1ll << 32) && (n == 0 || (var && g && d_var)), "gauss_sample_bwd: bad operand");
    SGCN_DET_LAUNCH(gauss_sample_bwd_kernel, n, var, g, n, key, d_var);
}
extern "C" int sgcn_det_agg_prep_f32(const float* mu, const float* var, const float* Hm, const float* Hv, int64_t ldh,
                                     const int32_t* ifield, int32_t n0, int32_t d, float* delta_mu, float* ds2, float* msig2,
                                     float* ds, float* sbar, void* stream) {
    SGCN_REQUIRE(n0 >= 0 && d >= 0 && ldh >= d, "det_agg_prep: bad size");
    if (n0 == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(mu && var && Hm && Hv && ifield && delta_mu && ds2 && msig2 && ds && sbar, "det_agg_prep: null operand");
    SGCN_DET_LAUNCH(det_agg_prep_kernel, (int64_t)n0 * d, mu, var, Hm, Hv, ldh, ifield, n0, d, delta_mu, ds2, msig2, ds, sbar);
}
extern "C" int sgcn_det_agg_prep_bwd_f32(const float* var, const float* ds, const float* sbar, const float* g_ds2, const float* g_msig2,
                                         int32_t n0, int32_t d, const float* add, int64_t ldadd, int32_t add_rows, float* d_var,
                                         void* stream) {
    SGCN_REQUIRE(n0 >= 0 && d >= 0 && add_rows >= 0 && add_rows <= n0, "det_agg_prep_bwd: bad size");
    if (n0 == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(var && ds && sbar && g_ds2 && g_msig2 && d_var && (add_rows == 0 || (add && ldadd >= d)), "det_agg_prep_bwd: bad operand");
    SGCN_DET_LAUNCH(det_agg_prep_bwd_kernel, (int64_t)n0 * d, var, ds, sbar, g_ds2, g_msig2, n0, d, add, ldadd, add_rows, d_var);
}
extern "C" int sgcn_relu_eps_f32(const float* raw, int64_t ldr, int32_t n, int32_t d, float eps, float* y, int64_t ldy, void* stream) {
    SGCN_REQUIRE(n >= 0 && d >= 0 && (n == 0 || d == 0 || (raw && y && ldr >= d && ldy >= d)), "relu_eps: bad operand");
    SGCN_DET_LAUNCH(relu_eps_kernel, (int64_t)n * d, raw, ldr, n, d, eps, y, ldy);
}
extern "C" int sgcn_gate_f32(const float* raw, int64_t ldr, const float* g, int64_t ldg, int32_t n, int32_t d, float* out, void* stream) {
    SGCN_REQUIRE(n >= 0 && d >= 0 && (n == 0 || d == 0 || (raw && g && out && ldr >= d && ldg >= d)), "gate: bad operand");
    SGCN_DET_LAUNCH(gate_kernel, (int64_t)n * d, raw, ldr, g, ldg, n, d, out);
}
