// Row-kernel dense layer probe (round 3): y = act(LN(x . W)) for the step's small layers as ONE WAVEFRONT PER ROW with the
// weight matrix staged in LDS -- the form the output layer already takes inside the loss kernel (sgcn_dense.hip ce_head) --
// against the 32 x 128 MFMA tile launch (profiles/gemm_probe.py: 11.9 us for 2,036 x 128 x 128 with LayerNorm, 9.1 us for
// 512 x 128 x 41).  Launches are issued back to back on one stream, each reading the previous one's output.
//
//   hipcc -O3 --offload-arch=gfx950 profiles/rowmlp_probe.hip -o /tmp/rowmlp_probe && /tmp/rowmlp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kBlock = 256, kWave = 64;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// K = 128 inputs (two per lane), N outputs (<= 128: two per lane), W [K][N] row-major, staged in LDS with pitch N
template <int ROWS_PER_WAVE>
__global__ __launch_bounds__(kBlock) void rowmlp_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                        const float* __restrict__ scale, const float* __restrict__ offset,
                                                        int n, int K, int N, int norm, float* __restrict__ y) {
    extern __shared__ float wl[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = K * N;
    // everything requested before anything is used
    float v[64];
#pragma unroll
    for (int u = 0; u < 64; u++) {
        const int i = threadIdx.x + u * kBlock;
        v[u] = i < total ? W[i] : 0.f;
    }
    const float sc0 = lane < N ? scale[lane] : 0.f, sc1 = lane + 64 < N ? scale[lane + 64] : 0.f;
    const float of0 = lane < N ? offset[lane] : 0.f, of1 = lane + 64 < N ? offset[lane + 64] : 0.f;
    const long row0 = ((long)blockIdx.x * (kBlock / kWave) + wave) * ROWS_PER_WAVE;
    float x0[ROWS_PER_WAVE], x1[ROWS_PER_WAVE];
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; r++) {
        const long row = row0 + r;
        x0[r] = (row < n && lane < K) ? x[row * K + lane] : 0.f;
        x1[r] = (row < n && lane + 64 < K) ? x[row * K + lane + 64] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 64; u++) {
        const int i = threadIdx.x + u * kBlock;
        if (i < total) wl[i] = v[u];
    }
    __syncthreads();
    const int c0 = lane < N ? lane : 0, c1 = lane + 64 < N ? lane + 64 : 0;
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; r++) {
        const long row = row0 + r;
        if (row >= n) break;
        // the MFMA launch's order: 32-wide K-steps alternating between two K-groups, partial sums added at the end
        float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
        for (int s0 = 0; s0 * 32 < K; s0++) {
            const int g = s0 & 1;
            const float xs = s0 < 2 ? x0[r] : x1[r];
            const float* wk = wl + s0 * 32 * N;
            float p0 = a0[g], p1 = a1[g];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float w0[8], w1[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { w0[u] = wk[(q * 8 + u) * N + c0]; w1[u] = wk[(q * 8 + u) * N + c1]; }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), ((s0 * 32) & 63) + q * 8 + u));
                    p0 = fmaf(xv, w0[u], p0); p1 = fmaf(xv, w1[u], p1);
                }
            }
            a0[g] = p0; a1[g] = p1;
        }
        float z0 = a0[0] + a0[1], z1 = a1[0] + a1[1];
        if (norm) {
            const float s = (lane < N ? z0 : 0.f) + (lane + 64 < N ? z1 : 0.f);
            const float mean = wsum(s) / (float)N;
            const float d0 = lane < N ? z0 - mean : 0.f, d1 = lane + 64 < N ? z1 - mean : 0.f;
            const float rs = rsqrtf(wsum(d0 * d0 + d1 * d1) / (float)N + 1e-9f);
            z0 = fmaxf(d0 * rs * sc0 + of0, 0.f); z1 = fmaxf(d1 * rs * sc1 + of1, 0.f);
        }
        if (lane < N) y[row * N + lane] = z0;
        if (lane + 64 < N) y[row * N + lane + 64] = z1;
    }
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Shape { int n, K, N, norm; const char* name; } shapes[] = {
        {2036, 128, 128, 1, "dense1 2036 x 128 -> 128 + LN + ReLU"}, {1018, 128, 128, 1, "1018 x 128 -> 128 + LN + ReLU"},
        {512, 128, 128, 1, "512 x 128 -> 128 + LN + ReLU"}, {512, 128, 41, 0, "dense3 512 x 128 -> 41"}};
    for (const Shape& s : shapes) {
        float *x, *y, *W, *sc, *of;
        CHECK(hipMalloc(&x, (size_t)s.n * 128 * 4)); CHECK(hipMalloc(&y, (size_t)s.n * 128 * 4));
        CHECK(hipMalloc(&W, 128 * 128 * 4)); CHECK(hipMalloc(&sc, 512)); CHECK(hipMalloc(&of, 512));
        std::vector<float> h((size_t)s.n * 128, 0.01f), hw(128 * 128, 0.02f), hs(128, 1.f), ho(128, 0.1f);
        CHECK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(y, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(sc, hs.data(), 512, hipMemcpyHostToDevice)); CHECK(hipMemcpy(of, ho.data(), 512, hipMemcpyHostToDevice));
        const size_t lds = (size_t)s.K * s.N * 4;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowmlp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowmlp_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        for (int rpw = 1; rpw <= 2; rpw++) {
            const int rows_per_block = 4 * rpw;
            const unsigned blocks = (unsigned)((s.n + rows_per_block - 1) / rows_per_block);
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipEventRecord(e0, st));
                for (int it = 0; it < 100; it++) {
                    // K == N == 128 shapes ping-pong x <-> y so that every launch depends on the previous one
                    const float* in = (it & 1) && s.N == 128 ? y : x;
                    float* out = (it & 1) && s.N == 128 ? x : y;
                    if (rpw == 1) hipLaunchKernelGGL(rowmlp_kernel<1>, dim3(blocks), dim3(kBlock), lds, st, in, W, sc, of, s.n, s.K, s.N, s.norm, out);
                    else hipLaunchKernelGGL(rowmlp_kernel<2>, dim3(blocks), dim3(kBlock), lds, st, in, W, sc, of, s.n, s.K, s.N, s.norm, out);
                }
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best = ms < best ? ms : best;
            }
            printf("{\"shape\": \"%s\", \"rows_per_wave\": %d, \"workgroups\": %u, \"us_per_launch\": %.2f}\n", s.name, rpw, blocks, best * 10.f);
        }
        CHECK(hipFree(x)); CHECK(hipFree(y)); CHECK(hipFree(W)); CHECK(hipFree(sc)); CHECK(hipFree(of));
    }
    return 0;
}
