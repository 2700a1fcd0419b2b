// Pure-gather microbenchmark for gfx950: what rate can wavefronts pull whole B rows (the SpMM's
// dense operand: 1,216-byte slab pieces of 2,432-byte rows) from the L2 / the fabric into VGPRs
// when NOTHING else happens -- no FMA, no accumulators, no index decoding beyond a readlane?
// This is the ceiling any gather-formulated SpMM on a locality-free graph sits under
// (DESIGN.md 3.1b), measured instead of quoted.
//
//   regimes   hit   : row ids drawn from a set that stays L2-resident in every XCD
//             miss  : row ids uniform over the whole operand (561 MB: beyond L2 and Infinity Cache)
//   variants  bytes per gathered row piece (1024 = one dwordx4 per lane, 1216 = + one dword per
//             lane, 1280 = + dwordx4 on 16 lanes...), loads in flight per wave (U), waves per SIMD,
//             cache policy of the load (plain / nt / sc1 / sc0 sc1)
//
// Build + run: profiles/gather_ceiling.py (hipcc --offload-arch=gfx950).  Prints one JSON per line.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum { PLAIN = 0, NT = 1, SC1 = 2, SC0SC1 = 3 };

template <int POL>
__device__ __forceinline__ f4 ld16(const char* base, uint32_t off) {
    f4 r;
    if constexpr (POL == PLAIN) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
    else if constexpr (POL == NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r) : "v"(off), "s"(base) : "memory");
    else if constexpr (POL == SC1) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(r) : "v"(off), "s"(base) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}
template <int POL>
__device__ __forceinline__ float ld4(const char* base, uint32_t off) {
    float r;
    if constexpr (POL == PLAIN) asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
    else if constexpr (POL == NT) asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(r) : "v"(off), "s"(base) : "memory");
    else if constexpr (POL == SC1) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(r) : "v"(off), "s"(base) : "memory");
    else asm volatile("global_load_dword %0, %1, %2 sc0 sc1" : "=v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}
template <int POL>
__device__ __forceinline__ f2 ld8(const char* base, uint32_t off) {
    f2 r;
    if constexpr (POL == PLAIN) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
    else if constexpr (POL == NT) asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(r) : "v"(off), "s"(base) : "memory");
    else if constexpr (POL == SC1) asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(r) : "v"(off), "s"(base) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, %2 sc0 sc1" : "=v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}

// per-lane 64-bit address form (sub-wave groups gather DIFFERENT rows in one instruction)
__device__ __forceinline__ f4 ld16v(const char* addr) {
    f4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(addr) : "memory");
    return r;
}

// SHAPE: 0 = 1024 B (dwordx4), 1 = 1216 B (dwordx4 + dword on 48 lanes... all 64 lanes load, 48 matter),
//        2 = 512 B (dwordx2), 3 = 256 B (dword), 4 = 2048 B (2 x dwordx4)
//        5 = 2 rows x 512 B per dwordx4 instruction (half-waves gather different rows)
//        6 = 4 rows x 256 B per dwordx4 instruction (quarter-waves)
template <int U, int POL, int SHAPE>
__global__ __launch_bounds__(256) void gather_kernel(const char* __restrict__ B, int64_t ldb_bytes,
                                                     const int32_t* __restrict__ idx, int n_per_wave,
                                                     float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int32_t* my = idx + wave * n_per_wave;
    const uint32_t off16 = lane * 16u, off8 = lane * 8u, off4 = lane * 4u;
    for (int k = 0; k < n_per_wave; k += 64) {
        const int myidx = my[k + lane];
        if constexpr (SHAPE == 5 || SHAPE == 6) {
            constexpr int G = SHAPE == 5 ? 2 : 4;          // rows per instruction
            constexpr int LPG = 64 / G;                    // lanes per row
            const int grp = lane / LPG;
            const uint32_t goff = (lane % LPG) * 16u;
#pragma unroll 1
            for (int j = 0; j < 64; j += U * G) {
                f4 a[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    // lane group g takes row id of lane j + u*G + g (a bpermute in the real kernel;
                    // here: readlanes + selects, the scalar-friendly form)
                    int r = __builtin_amdgcn_readlane(myidx, j + u * G);
#pragma unroll
                    for (int g = 1; g < G; g++) {
                        const int rg = __builtin_amdgcn_readlane(myidx, j + u * G + g);
                        r = grp == g ? rg : r;
                    }
                    a[u] = ld16v(B + (int64_t)r * ldb_bytes + goff);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < U; u++) asm volatile("" :: "v"(a[u]));
            }
            continue;
        }
#pragma unroll 1
        for (int j = 0; j < 64; j += U) {
            f4 a[U], b2[U];
            float e[U];
            f2 h[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int r = __builtin_amdgcn_readlane(myidx, j + u);
                const char* base = B + (int64_t)r * ldb_bytes;
                if constexpr (SHAPE == 0 || SHAPE == 1 || SHAPE == 4) a[u] = ld16<POL>(base, off16);
                if constexpr (SHAPE == 1) e[u] = ld4<POL>(base, 1024u + (lane < 48 ? off4 : 0u));
                if constexpr (SHAPE == 4) b2[u] = ld16<POL>(base, 1024u + off16);
                if constexpr (SHAPE == 2) h[u] = ld8<POL>(base, off8);
                if constexpr (SHAPE == 3) e[u] = ld4<POL>(base, off4);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; u++) {
                if constexpr (SHAPE == 0 || SHAPE == 1 || SHAPE == 4) asm volatile("" :: "v"(a[u]));
                if constexpr (SHAPE == 1 || SHAPE == 3) asm volatile("" :: "v"(e[u]));
                if constexpr (SHAPE == 4) asm volatile("" :: "v"(b2[u]));
                if constexpr (SHAPE == 2) asm volatile("" :: "v"(h[u]));
            }
        }
    }
    if (sink && lane == 9999) sink[0] = 1.f;
}

// Do L2 hits flow past outstanding misses?  Two wave populations in ONE launch: waves with an even
// index in their workgroup gather nA rows each from the L2-resident set, odd waves nB rows each from
// the whole operand (all-miss).  Run (nA, 0), (0, nB) and (nA, nB): if hits and misses overlap, the
// mixed launch takes ~max of the two; if they serialise in the CU's vector-memory path, ~the sum.
template <int U>
__global__ __launch_bounds__(256) void mixed_kernel(const char* __restrict__ B, int64_t ldb_bytes,
                                                    const int32_t* __restrict__ idxA, int nA,
                                                    const int32_t* __restrict__ idxB, int nB, int stride,
                                                    int by_cu) {
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wib;
    // by_cu: the role is a property of the CU (HW_REG_HW_ID bits 11:8 = cu_id), so a CU's vector
    // memory path only ever sees one kind of traffic
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const bool roleB = by_cu ? ((hwid >> 8) & 1) : (wib & 1);
    const int32_t* my = (roleB ? idxB : idxA) + wave * stride;
    const int n = roleB ? nB : nA;
    const uint32_t off16 = lane * 16u, off4 = lane * 4u;
    for (int k = 0; k < n; k += 64) {
        const int myidx = my[k + lane];
#pragma unroll 1
        for (int j = 0; j < 64; j += U) {
            f4 a[U];
            float e[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int r = __builtin_amdgcn_readlane(myidx, j + u);
                const char* base = B + (int64_t)r * ldb_bytes;
                a[u] = ld16<PLAIN>(base, off16);
                e[u] = ld4<PLAIN>(base, 1024u + (lane < 48 ? off4 : 0u));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; u++) { asm volatile("" :: "v"(a[u])); asm volatile("" :: "v"(e[u])); }
        }
    }
}

template <int U>
static float run_mixed(const char* B, int64_t ldb, const int32_t* ia, int nA, const int32_t* ib, int nB, int stride, int blocks, int by_cu = 0) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((mixed_kernel<U>), dim3(blocks), dim3(256), 0, 0, B, ldb, ia, nA, ib, nB, stride, by_cu);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    return best;
}

struct Cfg { const char* name; int U, pol, shape; };

template <int U, int POL, int SHAPE>
static float run(const char* B, int64_t ldb, const int32_t* idx, int npw, int blocks, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather_kernel<U, POL, SHAPE>), dim3(blocks), dim3(256), 0, 0, B, ldb, idx, npw, (float*)nullptr);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((gather_kernel<U, POL, SHAPE>), dim3(blocks), dim3(256), 0, 0, B, ldb, idx, npw, (float*)nullptr);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

static const int kShapeBytes[7] = {1024, 1216, 512, 256, 2048, 512, 256};
static const char* kShapeName[7] = {"x4", "x4+x1", "x2", "x1", "2*x4", "x4 on 2 rows", "x4 on 4 rows"};
static const char* kPolName[4] = {"plain", "nt", "sc1", "sc0sc1"};

int main(int argc, char** argv) {
    const int64_t K = 232965, pitch = 608 * 4;            // S-Reddit operand: 232,965 rows of 2,432 B
    const int npw = argc > 1 ? atoi(argv[1]) : 4096;       // rows gathered per wave
    char* B; CHECK(hipMalloc(&B, K * pitch));
    CHECK(hipMemset(B, 0, K * pitch));
    const int max_waves = 256 * 4 * 8;
    std::vector<int32_t> h((size_t)max_waves * npw);
    int32_t *idx_hit, *idx_miss, *idx_win, *idx_mall;
    CHECK(hipMalloc(&idx_hit, h.size() * 4)); CHECK(hipMalloc(&idx_miss, h.size() * 4)); CHECK(hipMalloc(&idx_win, h.size() * 4));
    CHECK(hipMalloc(&idx_mall, h.size() * 4));

    std::mt19937 gen(1);
    // mall: uniform over 49,152 rows = 120 MB -- far beyond the L2s, well inside the 256 MiB Infinity Cache
    for (auto& x : h) x = (int32_t)(gen() % 49152);
    CHECK(hipMemcpy(idx_mall, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // hit: 1,024 rows at a stride that spreads them over all channels (2.5 MB of lines per XCD)
    for (auto& x : h) x = (int32_t)((gen() % 1024) * 227 % K);
    CHECK(hipMemcpy(idx_hit, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (auto& x : h) x = (int32_t)(gen() % K);
    CHECK(hipMemcpy(idx_miss, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // window: every wave walks the row space front to back (sorted ids), like the column sweep:
    // waves that run at the same speed share an L2-sized window of B
    for (int64_t w = 0; w < max_waves; w++) {
        std::vector<int32_t> v(npw);
        for (auto& x : v) x = (int32_t)(gen() % K);
        std::sort(v.begin(), v.end());
        std::copy(v.begin(), v.end(), h.begin() + w * npw);
    }
    CHECK(hipMemcpy(idx_win, h.data(), h.size() * 4, hipMemcpyHostToDevice));

    struct Regime { const char* name; const int32_t* idx; } regimes[4] = {{"hit", idx_hit}, {"miss", idx_miss}, {"window", idx_win}, {"mall", idx_mall}};
#define RUN(U, POL, SHAPE, WPS)                                                                          \
    for (auto& rg : regimes) {                                                                            \
        const int blocks = 256 * WPS;                                                                     \
        const float ms = run<U, POL, SHAPE>(B, pitch, rg.idx, npw, blocks, 3);                            \
        const double bytes = (double)blocks * 4 * npw * kShapeBytes[SHAPE];                               \
        printf("{\"regime\": \"%s\", \"shape\": \"%s\", \"bytes_per_row\": %d, \"U\": %d, \"policy\": \"%s\", \"waves_per_simd\": %d, "   \
               "\"ms\": %.4f, \"TBps\": %.3f, \"B_per_clk_per_CU\": %.2f}\n", rg.name, kShapeName[SHAPE], kShapeBytes[SHAPE], U,   \
               kPolName[POL], WPS, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256.0 / 2.4e9);              \
        fflush(stdout);                                                                                   \
    }
    // the shipped kernel's shape: 1,216 B per nonzero, U = 4, 4 waves per SIMD
    RUN(4, PLAIN, 1, 4)
    RUN(4, NT, 1, 4)
    RUN(4, SC1, 1, 4)
    RUN(4, SC0SC1, 1, 4)
    RUN(8, PLAIN, 1, 4)
    RUN(8, NT, 1, 4)
    RUN(4, PLAIN, 1, 8)
    RUN(8, PLAIN, 1, 8)
    RUN(16, PLAIN, 1, 2)
    // one dwordx4 per lane only (1,024 B): is the extra dword plane what costs?
    RUN(4, PLAIN, 0, 4)
    RUN(8, PLAIN, 0, 4)
    RUN(8, NT, 0, 4)
    RUN(8, PLAIN, 0, 8)
    RUN(16, PLAIN, 0, 4)
    // 2 x dwordx4 per lane (2,048 B of the 2,432 B row in one visit)
    RUN(4, PLAIN, 4, 4)
    RUN(8, PLAIN, 4, 4)
    RUN(4, NT, 4, 4)
    // narrower pieces (more rows per register budget): 512 B and 256 B per nonzero
    RUN(8, PLAIN, 2, 4)
    RUN(16, PLAIN, 2, 8)
    RUN(16, PLAIN, 3, 8)
    // sub-wave groups: one dwordx4 instruction gathers 2 x 512 B / 4 x 256 B pieces of different rows
    RUN(4, PLAIN, 5, 4)
    RUN(8, PLAIN, 5, 4)
    RUN(8, PLAIN, 5, 8)
    RUN(4, PLAIN, 6, 4)
    RUN(8, PLAIN, 6, 4)
    RUN(8, PLAIN, 6, 8)
    // hit-under-miss: 2 hit waves + 2 miss waves per SIMD-group (4 waves per workgroup, 4 workgroups per CU)
    {
        const int blocks = 256 * 4;
        const int nA = 4096, nB = 1024;
        const float ta = run_mixed<8>(B, pitch, idx_hit, nA, idx_miss, 0, npw, blocks);
        const float tb = run_mixed<8>(B, pitch, idx_hit, 0, idx_miss, nB, npw, blocks);
        const float tm = run_mixed<8>(B, pitch, idx_hit, nA, idx_miss, nB, npw, blocks);
        const double ba = (double)blocks * 2 * nA * 1216, bb = (double)blocks * 2 * nB * 1216;
        printf("{\"regime\": \"mixed\", \"shape\": \"x4+x1\", \"bytes_per_row\": 1216, \"U\": 8, \"policy\": \"plain\", "
               "\"waves_per_simd\": 4, \"ms_hit_only\": %.4f, \"ms_miss_only\": %.4f, \"ms\": %.4f, "
               "\"TBps_hit_only\": %.3f, \"TBps_miss_only\": %.3f, \"TBps\": %.3f, \"B_per_clk_per_CU\": %.2f}\n",
               ta, tb, tm, ba / ta / 1e9, bb / tb / 1e9, (ba + bb) / tm / 1e9, (ba + bb) / (tm * 1e-3) / 256.0 / 2.4e9);
    }
    {   // the same with the role tied to the CU: half the CUs only hit, the other half only miss
        const int blocks = 256 * 4;
        const int nA = 4096, nB = 1024;
        const float ta = run_mixed<8>(B, pitch, idx_hit, nA, idx_miss, 0, npw, blocks, 1);
        const float tb = run_mixed<8>(B, pitch, idx_hit, 0, idx_miss, nB, npw, blocks, 1);
        const float tm = run_mixed<8>(B, pitch, idx_hit, nA, idx_miss, nB, npw, blocks, 1);
        const double ba = (double)blocks * 2 * nA * 1216, bb = (double)blocks * 2 * nB * 1216;   // ~half the waves per role
        printf("{\"regime\": \"mixed_by_cu\", \"shape\": \"x4+x1\", \"bytes_per_row\": 1216, \"U\": 8, \"policy\": \"plain\", "
               "\"waves_per_simd\": 4, \"ms_hit_only\": %.4f, \"ms_miss_only\": %.4f, \"ms\": %.4f, "
               "\"TBps_hit_only\": %.3f, \"TBps_miss_only\": %.3f, \"TBps\": %.3f, \"B_per_clk_per_CU\": %.2f}\n",
               ta, tb, tm, ba / ta / 1e9, bb / tb / 1e9, (ba + bb) / tm / 1e9, (ba + bb) / (tm * 1e-3) / 256.0 / 2.4e9);
    }
    return 0;
}
