// "LDS-staged, reuse r" gather regime (round 3): the ceiling of a column sweep that shares B
// pieces between the rows of ONE CU through the LDS instead of fetching a piece from the L2 once per nonzero.
//
// Shape under test (the only one whose accumulators fit: DESIGN.md 3.1c): one workgroup of 16 wavefronts per CU,
// all 160 KB of LDS as a double-buffered ring of 256-byte pieces (a 64-column slab of one B row); per chunk every
// wavefront (a) gathers its share of the NEXT chunk's pieces from the L2 straight into the ring
// (global_load_lds_dwordx4: four 256-byte pieces per instruction, the widest form there is for pieces this narrow),
// (b) reads r x its share of the CURRENT chunk's pieces back into VGPRs, two pieces per ds_read_b64 (one per
// half-wave, the lane layout of a half-wave FMA update), then (c) waits for its gathers and meets the others at a
// barrier.  NO FMA, no VGPR indexing, no plan decoding: whatever a real kernel adds comes on top.
//
//   regimes  hit    : row ids from a set that stays L2-resident            miss : uniform over the 561 MB operand
//            sweep  : chunk c draws from the c-th window of the operand, the same window on every CU (what a
//                     clock-locked column sweep gives: the first CU to touch a piece misses, the others hit)
//   r        pieces delivered to registers per piece gathered (S-Reddit, 1,792 rows per CU: 1.43; 4,096 rows: 1.75)
//
// Output: one JSON line per (regime, r): gathered TB/s (L2 -> LDS), delivered TB/s (LDS -> VGPR), and the time one
// S-Reddit SpMM would need at that delivered rate (55.8 GB of pieces).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kHalfRing = 65536;            // bytes per half ring (two of them: 128 KB of the 160 KB)
constexpr int kWaves = 16;

typedef float f2v __attribute__((ext_vector_type(2)));

// PIECE: bytes per piece -- 256 (64-column slab, lanes hold float2, ds_read_b64) or 128 (32-column slab, the VERDICT's
//        4,096 rows x 32 floats per CU: lanes hold one float, ds_read_b32; 8 pieces per gather instruction).
// RD: LDS read instructions per wave and chunk (2 pieces each): r = RD * 2 * 16 / kSlots
template <int RD, int PIECE>
__global__ __launch_bounds__(1024) void staged(const char* __restrict__ B, int64_t ldb_bytes, const int32_t* __restrict__ rows,
                                               const uint16_t* __restrict__ reads, int n_chunks, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(1024))) char ring[];
    constexpr int kPiece = PIECE, kSlots = kHalfRing / PIECE, kPerInstr = 1024 / PIECE, kLanesPerPiece = 64 / kPerInstr;
    constexpr int kGatherPerWave = kSlots / kWaves / kPerInstr;       // global_load_lds instructions per wave and chunk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t* myrows = rows + ((int64_t)blockIdx.x * n_chunks) * kSlots;
    const uint16_t* myreads = reads + ((int64_t)blockIdx.x * kWaves + wave) * 64;
    f2v acc = {0.f, 0.f};
    // my slot pairs for the read phase (the same pattern every chunk: what is measured is the rate, not the data)
    uint32_t raddr[RD];
#pragma unroll
    for (int k = 0; k < RD; k++) raddr[k] = (uint32_t)(myreads[(2 * k + (lane >> 5)) & 63] % kSlots) * kPiece + (lane & 31) * (PIECE / 32);
    auto gather = [&](int c) {
        char* half = ring + (c & 1) * kSlots * kPiece;
#pragma unroll
        for (int g = 0; g < kGatherPerWave; g++) {
            const int slot = (wave * kGatherPerWave + g) * kPerInstr;         // first slot of this instruction
            const int32_t row = myrows[(int64_t)c * kSlots + slot + lane / kLanesPerPiece];
            const char* src = B + (int64_t)row * ldb_bytes + (lane % kLanesPerPiece) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(half + slot * kPiece), 16, 0, 0);
        }
    };
    gather(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < n_chunks; c++) {
        if (c + 1 < n_chunks) gather(c + 1);
        const uint32_t base = (c & 1) * kSlots * kPiece;
        f2v x[RD];
#pragma unroll
        for (int k = 0; k < RD; k++) {
            if constexpr (PIECE == 256) asm volatile("ds_read_b64 %0, %1" : "=v"(x[k]) : "v"(raddr[k] + base) : "memory");
            else { x[k].y = 0.f; asm volatile("ds_read_b32 %0, %1" : "=v"(x[k].x) : "v"(raddr[k] + base) : "memory"); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < RD; k++) acc += x[k];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (acc.x + acc.y == 1.2345f) sink[0] = acc.x;
}

template <int RD, int PIECE>
static void run(const char* regime, const char* dB, int64_t ldb, const int32_t* d_rows, const uint16_t* d_reads, int n_chunks,
                float* sink, hipEvent_t e0, hipEvent_t e1) {
    const size_t lds = 2 * kHalfRing;
    constexpr int kPiece = PIECE, kSlots = kHalfRing / PIECE;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&staged<RD, PIECE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((staged<RD, PIECE>), dim3(256), dim3(1024), lds, 0, dB, ldb, d_rows, d_reads, n_chunks, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((staged<RD, PIECE>), dim3(256), dim3(1024), lds, 0, dB, ldb, d_rows, d_reads, n_chunks, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    const double gathered = 256.0 * n_chunks * kSlots * kPiece;
    const double delivered = 256.0 * n_chunks * kWaves * RD * 2.0 * kPiece;
    const double r = delivered / gathered;
    printf("{\"regime\": \"lds_staged_%s\", \"piece_bytes\": %d, \"reuse_r\": %.3f, \"ms\": %.4f, \"gathered_TBps\": %.2f, \"delivered_TBps\": %.2f, "
           "\"spmm_ms_at_this_rate\": %.3f, \"chunks\": %d}\n", regime, kPiece, r, best, gathered / best * 1e-9, delivered / best * 1e-9,
           55.8e9 / (delivered / best * 1e-9) * 1e-9 * 1e-3 * 1e3, n_chunks);
}

int main() {
    const int64_t N = 232965, ldb = 2432;                 // S-Reddit operand: 608 floats per row
    const int n_chunks = 480;                             // 480 x 256 = 123 k distinct columns: one 64-column slab of a 1,792-row tile
    constexpr int kSlotsMax = 512;
    std::vector<float> hB((size_t)N * (ldb / 4), 1.f);
    char* dB;
    CHECK(hipMalloc(&dB, (size_t)N * ldb));
    CHECK(hipMemcpy(dB, hB.data(), (size_t)N * ldb, hipMemcpyHostToDevice));
    std::mt19937 rng(7);
    std::vector<uint16_t> hreads((size_t)256 * kWaves * 64);
    for (auto& x : hreads) x = (uint16_t)(rng() % 1024);
    uint16_t* d_reads;
    CHECK(hipMalloc(&d_reads, hreads.size() * 2));
    CHECK(hipMemcpy(d_reads, hreads.data(), hreads.size() * 2, hipMemcpyHostToDevice));
    std::vector<int32_t> hrows((size_t)256 * n_chunks * kSlotsMax);
    int32_t* d_rows;
    CHECK(hipMalloc(&d_rows, hrows.size() * 4));
    float* sink;
    CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int piece : {256, 128})
    for (int regime = 0; regime < 3; regime++) {
        const int kSlots = kHalfRing / piece;
        const char* name = regime == 0 ? "hit" : regime == 1 ? "sweep" : "miss";
        for (int cu = 0; cu < 256; cu++)
            for (int c = 0; c < n_chunks; c++) {
                int32_t* r = &hrows[((size_t)cu * n_chunks + c) * kSlots];
                if (regime == 0) for (int k = 0; k < kSlots; k++) r[k] = (int32_t)(rng() % 3000);
                else if (regime == 2) for (int k = 0; k < kSlots; k++) r[k] = (int32_t)(rng() % N);
                else {
                    // the c-th window of the operand, ascending inside the chunk (a tile's distinct columns in sweep order)
                    const int64_t lo = N * c / n_chunks, hi = N * (c + 1) / n_chunks;
                    for (int k = 0; k < kSlots; k++) r[k] = (int32_t)(lo + rng() % (hi - lo));
                    std::sort(r, r + kSlots);
                }
            }
        CHECK(hipMemcpy(d_rows, hrows.data(), hrows.size() * 4, hipMemcpyHostToDevice));
        if (piece == 256) {
            run<8, 256>(name, dB, ldb, d_rows, d_reads, n_chunks, sink, e0, e1);       // r = 1
            run<12, 256>(name, dB, ldb, d_rows, d_reads, n_chunks, sink, e0, e1);      // r = 1.5
            run<16, 256>(name, dB, ldb, d_rows, d_reads, n_chunks, sink, e0, e1);      // r = 2
            run<32, 256>(name, dB, ldb, d_rows, d_reads, n_chunks, sink, e0, e1);      // r = 4
        } else {
            run<16, 128>(name, dB, ldb, d_rows, d_reads, n_chunks, sink, e0, e1);      // r = 1
            run<28, 128>(name, dB, ldb, d_rows, d_reads, n_chunks, sink, e0, e1);      // r = 1.75
            run<32, 128>(name, dB, ldb, d_rows, d_reads, n_chunks, sink, e0, e1);      // r = 2
            run<64, 128>(name, dB, ldb, d_rows, d_reads, n_chunks, sink, e0, e1);      // r = 4
        }
    }
    return 0;
}
