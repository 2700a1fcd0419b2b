// Occupancy probe for the LDS sweep's chunk statement (round 5): the shipped kernel keeps 192 accumulator registers per
// wave, i.e. TWO waves per SIMD, and its chunk statements -- per nonzero v_readlane + v_and_or + ds_read_b64 +
// s_set_gpr_idx_idx + v_pk_add -- run at ~6.3 CU-clocks per nonzero where the busiest pipeline alone would need ~3.4
// (DESIGN.md 3.3).  Is that the price of two waves per SIMD?  This probe issues the statement's GROUP structure
// (8 reads of the next group, wait, 8 indexed updates of this one) on dummy LDS data with 1 / 2 / 3 / 4 waves per SIMD on
// every CU (accumulators v[64:127]: occupancy set by the workgroup size alone), in four forms:
//   single    the shipped single-word group                     pair     the shipped pair-word group (one read, two updates)
//   addtid    readlane + s_and m0 + 2 x ds_read_addtid_b32 + idx + pk_add (no VALU address)
//   noread    the updates alone (no LDS reads): the VALU / scalar side of a group
//   wide_b128 1 KB pieces, ds_read_b128, two packed adds per nonzero (a nonzero covers 256 columns); *_reads_only the read side alone
//   *_words_in_sgprs   the entry words already in scalar registers (no v_readlane): what SMEM-delivered entries would cost
// Output: one JSON line per (form, waves per SIMD): CU-clocks per nonzero (2.4 GHz nominal).
// Build + run: hipcc -O3 --offload-arch=gfx950 profiles/lds_stmt_probe.hip -o /tmp/lds_stmt_probe && /tmp/lds_stmt_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { SINGLE, PAIR, ADDTID, NOREAD, WIDE, WIDEONLY, B64ONLY, SSINGLE, SPAIR, NFORM };
static const char* kNames[NFORM] = {"single", "pair", "addtid", "noread", "wide_b128", "b128_reads_only", "b64_reads_only", "single_words_in_sgprs", "pair_words_in_sgprs"};
static const int kNnzPerGroup[NFORM] = {8, 16, 8, 8, 8, 8, 8, 8, 16};   // (wide: a nonzero covers 256 columns, not 128)

#define RD1(Q, K, LANE) "v_readlane_b32 s[36+8*" #Q "+" #K "], %[ew], " #LANE "\n\t"
#define AD1(Q, K, T) "v_and_or_b32 v[28+" #T "], s[36+8*" #Q "+" #K "], %[mask], %[lane]\n\t"
#define DS1(Q, K, T) "ds_read_b64 v[32+16*" #Q "+2*" #K ":32+16*" #Q "+2*" #K "+1], v[28+" #T "]\n\t"
#define READ8(Q, L0, L1, L2, L3, L4, L5, L6, L7)                                                   \
    RD1(Q, 0, L0) RD1(Q, 1, L1) RD1(Q, 2, L2) RD1(Q, 3, L3)                                        \
    AD1(Q, 0, 0) AD1(Q, 1, 1) AD1(Q, 2, 2) AD1(Q, 3, 3) DS1(Q, 0, 0) DS1(Q, 1, 1) DS1(Q, 2, 2) DS1(Q, 3, 3) \
    RD1(Q, 4, L4) RD1(Q, 5, L5) RD1(Q, 6, L6) RD1(Q, 7, L7)                                        \
    AD1(Q, 4, 0) AD1(Q, 5, 1) AD1(Q, 6, 2) AD1(Q, 7, 3) DS1(Q, 4, 0) DS1(Q, 5, 1) DS1(Q, 6, 2) DS1(Q, 7, 3)
// words already in scalar registers (what entries delivered by s_load_dwordx8 would look like): no v_readlane
#define READ8S(Q, L0, L1, L2, L3, L4, L5, L6, L7)                                                  \
    AD1(Q, 0, 0) AD1(Q, 1, 1) AD1(Q, 2, 2) AD1(Q, 3, 3) DS1(Q, 0, 0) DS1(Q, 1, 1) DS1(Q, 2, 2) DS1(Q, 3, 3) \
    AD1(Q, 4, 0) AD1(Q, 5, 1) AD1(Q, 6, 2) AD1(Q, 7, 3) DS1(Q, 4, 0) DS1(Q, 5, 1) DS1(Q, 6, 2) DS1(Q, 7, 3)
// addtid: m0 = piece address (the word's upper bits), two dword reads: columns [0, 64) and [64, 128)
#define AT1(Q, K)                                                                                   \
    "s_and_b32 m0, s[36+8*" #Q "+" #K "], %[smask]\n\t"                                              \
    "ds_read_addtid_b32 v[32+16*" #Q "+2*" #K "]\n\t"                                                \
    "ds_read_addtid_b32 v[32+16*" #Q "+2*" #K "+1] offset:256\n\t"
#define READ8T(Q, L0, L1, L2, L3, L4, L5, L6, L7)                                                  \
    RD1(Q, 0, L0) RD1(Q, 1, L1) RD1(Q, 2, L2) RD1(Q, 3, L3) AT1(Q, 0) AT1(Q, 1) AT1(Q, 2) AT1(Q, 3)  \
    RD1(Q, 4, L4) RD1(Q, 5, L5) RD1(Q, 6, L6) RD1(Q, 7, L7) AT1(Q, 4) AT1(Q, 5) AT1(Q, 6) AT1(Q, 7)
#define OP1(Q, K) "v_pk_add_f32 v[64:65], v[64:65], v[32+16*" #Q "+2*" #K ":32+16*" #Q "+2*" #K "+1]\n\t"
#define APPLY8(Q, WAIT)                                                                             \
    "s_waitcnt lgkmcnt(" #WAIT ")\n\t"                                                              \
    "s_setprio 2\n\t"                                                                               \
    "s_set_gpr_idx_on s[36+8*" #Q "+0], 0x9\n\t" OP1(Q, 0)                                           \
    "s_set_gpr_idx_idx s[36+8*" #Q "+1]\n\t" OP1(Q, 1) "s_set_gpr_idx_idx s[36+8*" #Q "+2]\n\t" OP1(Q, 2)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+3]\n\t" OP1(Q, 3) "s_set_gpr_idx_idx s[36+8*" #Q "+4]\n\t" OP1(Q, 4)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+5]\n\t" OP1(Q, 5) "s_set_gpr_idx_idx s[36+8*" #Q "+6]\n\t" OP1(Q, 6)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+7]\n\t" OP1(Q, 7)                                               \
    "s_set_gpr_idx_off\n\t"                                                                         \
    "s_setprio 0\n\t"
#define PAIR1(Q, K) OP1(Q, K) "s_lshr_b32 s[36+8*" #Q "+" #K "], s[36+8*" #Q "+" #K "], 24\n\t"       \
    "s_set_gpr_idx_idx s[36+8*" #Q "+" #K "]\n\t" OP1(Q, K)
#define APPLY8P(Q, WAIT)                                                                            \
    "s_waitcnt lgkmcnt(" #WAIT ")\n\t"                                                              \
    "s_setprio 2\n\t"                                                                               \
    "s_set_gpr_idx_on s[36+8*" #Q "+0], 0x9\n\t" PAIR1(Q, 0)                                         \
    "s_set_gpr_idx_idx s[36+8*" #Q "+1]\n\t" PAIR1(Q, 1) "s_set_gpr_idx_idx s[36+8*" #Q "+2]\n\t" PAIR1(Q, 2)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+3]\n\t" PAIR1(Q, 3) "s_set_gpr_idx_idx s[36+8*" #Q "+4]\n\t" PAIR1(Q, 4)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+5]\n\t" PAIR1(Q, 5) "s_set_gpr_idx_idx s[36+8*" #Q "+6]\n\t" PAIR1(Q, 6)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+7]\n\t" PAIR1(Q, 7)                                             \
    "s_set_gpr_idx_off\n\t"                                                                         \
    "s_setprio 0\n\t"
#define PAIR1S(Q, K) OP1(Q, K) "s_lshr_b32 s52, s[36+8*" #Q "+" #K "], 24\n\t" "s_set_gpr_idx_idx s52\n\t" OP1(Q, K)
#define APPLY8PS(Q, WAIT)                                                                           \
    "s_waitcnt lgkmcnt(" #WAIT ")\n\t"                                                              \
    "s_setprio 2\n\t"                                                                               \
    "s_set_gpr_idx_on s[36+8*" #Q "+0], 0x9\n\t" PAIR1S(Q, 0)                                        \
    "s_set_gpr_idx_idx s[36+8*" #Q "+1]\n\t" PAIR1S(Q, 1) "s_set_gpr_idx_idx s[36+8*" #Q "+2]\n\t" PAIR1S(Q, 2)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+3]\n\t" PAIR1S(Q, 3) "s_set_gpr_idx_idx s[36+8*" #Q "+4]\n\t" PAIR1S(Q, 4)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+5]\n\t" PAIR1S(Q, 5) "s_set_gpr_idx_idx s[36+8*" #Q "+6]\n\t" PAIR1S(Q, 6)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+7]\n\t" PAIR1S(Q, 7)                                            \
    "s_set_gpr_idx_off\n\t"                                                                         \
    "s_setprio 0\n\t"
#define CLOBBER_V "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
#define CLOBBER_S "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51"

// wide: 1 KB pieces (256 columns), ds_read_b128, a row's float4 at v[64 + 4 r : 67 + 4 r]: two packed adds per nonzero.
// piece registers v[128:191] (8 x 4 x two parities) -- for the measurement only
#define ADW(Q, K, T) "v_and_or_b32 v[28+" #T "], s[36+8*" #Q "+" #K "], %[mask], %[lane]\n\t"
#define DSW(Q, K, T) "ds_read_b128 v[128+32*" #Q "+4*" #K ":128+32*" #Q "+4*" #K "+3], v[28+" #T "]\n\t"
#define READ8W(Q, L0, L1, L2, L3, L4, L5, L6, L7)                                                  \
    RD1(Q, 0, L0) RD1(Q, 1, L1) RD1(Q, 2, L2) RD1(Q, 3, L3)                                        \
    ADW(Q, 0, 0) ADW(Q, 1, 1) ADW(Q, 2, 2) ADW(Q, 3, 3) DSW(Q, 0, 0) DSW(Q, 1, 1) DSW(Q, 2, 2) DSW(Q, 3, 3) \
    RD1(Q, 4, L4) RD1(Q, 5, L5) RD1(Q, 6, L6) RD1(Q, 7, L7)                                        \
    ADW(Q, 4, 0) ADW(Q, 5, 1) ADW(Q, 6, 2) ADW(Q, 7, 3) DSW(Q, 4, 0) DSW(Q, 5, 1) DSW(Q, 6, 2) DSW(Q, 7, 3)
#define OPW(Q, K) "v_pk_add_f32 v[64:65], v[64:65], v[128+32*" #Q "+4*" #K ":128+32*" #Q "+4*" #K "+1]\n\t"         \
                  "v_pk_add_f32 v[66:67], v[66:67], v[128+32*" #Q "+4*" #K "+2:128+32*" #Q "+4*" #K "+3]\n\t"
#define APPLY8W(Q, WAIT)                                                                            \
    "s_waitcnt lgkmcnt(" #WAIT ")\n\t"                                                              \
    "s_setprio 2\n\t"                                                                               \
    "s_set_gpr_idx_on s[36+8*" #Q "+0], 0x9\n\t" OPW(Q, 0)                                           \
    "s_set_gpr_idx_idx s[36+8*" #Q "+1]\n\t" OPW(Q, 1) "s_set_gpr_idx_idx s[36+8*" #Q "+2]\n\t" OPW(Q, 2)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+3]\n\t" OPW(Q, 3) "s_set_gpr_idx_idx s[36+8*" #Q "+4]\n\t" OPW(Q, 4)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+5]\n\t" OPW(Q, 5) "s_set_gpr_idx_idx s[36+8*" #Q "+6]\n\t" OPW(Q, 6)  \
    "s_set_gpr_idx_idx s[36+8*" #Q "+7]\n\t" OPW(Q, 7)                                               \
    "s_set_gpr_idx_off\n\t"                                                                         \
    "s_setprio 0\n\t"
#define APPLY8X(Q, WAIT) "s_waitcnt lgkmcnt(" #WAIT ")\n\t"
#define CLOBBER_W "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191"

typedef float acc32_t __attribute__((ext_vector_type(32)));

// 8 groups per block (64 entries of one register), as in the kernel
#define BLOCK(RD, AP)                                                                                \
    RD(1, 8, 9, 10, 11, 12, 13, 14, 15) AP(0, 8) RD(0, 16, 17, 18, 19, 20, 21, 22, 23) AP(1, 8)        \
    RD(1, 24, 25, 26, 27, 28, 29, 30, 31) AP(0, 8) RD(0, 32, 33, 34, 35, 36, 37, 38, 39) AP(1, 8)      \
    RD(1, 40, 41, 42, 43, 44, 45, 46, 47) AP(0, 8) RD(0, 48, 49, 50, 51, 52, 53, 54, 55) AP(1, 8)      \
    RD(1, 56, 57, 58, 59, 60, 61, 62, 63) AP(0, 8) RD(0, 0, 1, 2, 3, 4, 5, 6, 7) AP(1, 8)
#define NORD(Q, a, b, c, d, e, f, g, h)
#define APPLY8N(Q, WAIT) APPLY8(Q, 15)
#define APPLY8T(Q, WAIT) APPLY8(Q, 15)          // (16 dword reads per group in flight: the counter saturates at 15)

template <int F>
__global__ __launch_bounds__(1024) void probe(int reps, float* sink, const uint32_t* words) {
    extern __shared__ char smem[];
    for (int k = threadIdx.x; k < 32768; k += blockDim.x) reinterpret_cast<float*>(smem)[k] = 1.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t ew = words[(threadIdx.x & 1023)];          // (slot * 512) | row * 2 (| second row * 2 << 24)
    const uint32_t mask = ~511u, lane_off = lane * 8u, smask = 0x1fe00u;
    acc32_t a0 = {}, a1 = {};
    for (int r = 0; r < reps; r++) {
        if constexpr (F == SINGLE) {
            asm volatile(READ8(0, 0, 1, 2, 3, 4, 5, 6, 7) BLOCK(READ8, APPLY8) "s_waitcnt lgkmcnt(0)"
                         : "+{v[64:95]}"(a0), "+{v[96:127]}"(a1) : [ew] "v"(ew), [mask] "v"(mask), [lane] "v"(lane_off)
                         : CLOBBER_V, CLOBBER_S, "scc", "m0", "memory");
        } else if constexpr (F == PAIR) {
            asm volatile(READ8(0, 0, 1, 2, 3, 4, 5, 6, 7) BLOCK(READ8, APPLY8P) "s_waitcnt lgkmcnt(0)"
                         : "+{v[64:95]}"(a0), "+{v[96:127]}"(a1) : [ew] "v"(ew), [mask] "v"(mask), [lane] "v"(lane_off)
                         : CLOBBER_V, CLOBBER_S, "scc", "m0", "memory");
        } else if constexpr (F == ADDTID) {
            asm volatile(READ8T(0, 0, 1, 2, 3, 4, 5, 6, 7) BLOCK(READ8T, APPLY8T) "s_waitcnt lgkmcnt(0)"
                         : "+{v[64:95]}"(a0), "+{v[96:127]}"(a1) : [ew] "v"(ew), [smask] "s"(smask)
                         : CLOBBER_V, CLOBBER_S, "scc", "m0", "memory");
        } else if constexpr (F == WIDE || F == WIDEONLY) {
            const uint32_t maskw = ~1023u, lane_w = lane * 16u;
            if constexpr (F == WIDE)
                asm volatile(READ8W(0, 0, 1, 2, 3, 4, 5, 6, 7) BLOCK(READ8W, APPLY8W) "s_waitcnt lgkmcnt(0)"
                             : "+{v[64:95]}"(a0), "+{v[96:127]}"(a1) : [ew] "v"(ew), [mask] "v"(maskw), [lane] "v"(lane_w)
                             : CLOBBER_V, CLOBBER_W, CLOBBER_S, "scc", "m0", "memory");
            else
                asm volatile(READ8W(0, 0, 1, 2, 3, 4, 5, 6, 7) BLOCK(READ8W, APPLY8X) "s_waitcnt lgkmcnt(0)"
                             : "+{v[64:95]}"(a0), "+{v[96:127]}"(a1) : [ew] "v"(ew), [mask] "v"(maskw), [lane] "v"(lane_w)
                             : CLOBBER_V, CLOBBER_W, CLOBBER_S, "scc", "m0", "memory");
        } else if constexpr (F == SSINGLE || F == SPAIR) {
            // the 16 words of the two group parities come down ONCE (outside the timed pattern's inner structure)
            if constexpr (F == SSINGLE)
                asm volatile(RD1(0, 0, 0) RD1(0, 1, 1) RD1(0, 2, 2) RD1(0, 3, 3) RD1(0, 4, 4) RD1(0, 5, 5) RD1(0, 6, 6) RD1(0, 7, 7)
                             RD1(1, 0, 8) RD1(1, 1, 9) RD1(1, 2, 10) RD1(1, 3, 11) RD1(1, 4, 12) RD1(1, 5, 13) RD1(1, 6, 14) RD1(1, 7, 15)
                             READ8S(0, 0, 1, 2, 3, 4, 5, 6, 7) BLOCK(READ8S, APPLY8) "s_waitcnt lgkmcnt(0)"
                             : "+{v[64:95]}"(a0), "+{v[96:127]}"(a1) : [ew] "v"(ew), [mask] "v"(mask), [lane] "v"(lane_off)
                             : CLOBBER_V, CLOBBER_S, "s52", "scc", "m0", "memory");
            else
                asm volatile(RD1(0, 0, 0) RD1(0, 1, 1) RD1(0, 2, 2) RD1(0, 3, 3) RD1(0, 4, 4) RD1(0, 5, 5) RD1(0, 6, 6) RD1(0, 7, 7)
                             RD1(1, 0, 8) RD1(1, 1, 9) RD1(1, 2, 10) RD1(1, 3, 11) RD1(1, 4, 12) RD1(1, 5, 13) RD1(1, 6, 14) RD1(1, 7, 15)
                             READ8S(0, 0, 1, 2, 3, 4, 5, 6, 7) BLOCK(READ8S, APPLY8PS) "s_waitcnt lgkmcnt(0)"
                             : "+{v[64:95]}"(a0), "+{v[96:127]}"(a1) : [ew] "v"(ew), [mask] "v"(mask), [lane] "v"(lane_off)
                             : CLOBBER_V, CLOBBER_S, "s52", "scc", "m0", "memory");
        } else if constexpr (F == B64ONLY) {
            asm volatile(READ8(0, 0, 1, 2, 3, 4, 5, 6, 7) BLOCK(READ8, APPLY8X) "s_waitcnt lgkmcnt(0)"
                         : "+{v[64:95]}"(a0), "+{v[96:127]}"(a1) : [ew] "v"(ew), [mask] "v"(mask), [lane] "v"(lane_off)
                         : CLOBBER_V, CLOBBER_S, "scc", "m0", "memory");
        } else {
            asm volatile(READ8(0, 0, 1, 2, 3, 4, 5, 6, 7) "s_waitcnt lgkmcnt(0)\n\t" RD1(1, 0, 8) RD1(1, 1, 9) RD1(1, 2, 10) RD1(1, 3, 11)
                         RD1(1, 4, 12) RD1(1, 5, 13) RD1(1, 6, 14) RD1(1, 7, 15)
                         BLOCK(NORD, APPLY8N)
                         : "+{v[64:95]}"(a0), "+{v[96:127]}"(a1) : [ew] "v"(ew), [mask] "v"(mask), [lane] "v"(lane_off)
                         : CLOBBER_V, CLOBBER_S, "scc", "m0", "memory");
        }
    }
    float t = 0.f;
    for (int k = 0; k < 32; k++) t += a0[k] + a1[k];
    if (t == 123.456f) sink[threadIdx.x] = t;
}

template <int F>
static void run(int wps, float* sink, const uint32_t* words, hipEvent_t e0, hipEvent_t e1) {
    const int reps = 2000, nblk = 256;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipLaunchKernelGGL(probe<F>, dim3(nblk), dim3(256 * wps), 131072, 0, 10, sink, words);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<F>, dim3(nblk), dim3(256 * wps), 131072, 0, reps, sink, words);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double nnz_per_cu = (double)reps * 8 * kNnzPerGroup[F] * 4 * wps;       // 8 groups per block, 4 SIMDs x wps waves
    const double ns = ms * 1e6 / nnz_per_cu;
    printf("{\"form\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"cu_ns_per_nnz\": %.3f, \"cu_clocks_per_nnz\": %.2f, "
           "\"simd_clocks_per_nnz_per_wave\": %.1f}\n", kNames[F], wps, ms, ns, ns * 2.4, ns * 2.4 * 4 * wps);
}

int main() {
    float* sink;
    uint32_t* words;
    CHECK(hipMalloc(&sink, 4096 * sizeof(float)));
    std::vector<uint32_t> h(1024);
    uint32_t s = 12345;
    for (int k = 0; k < 1024; k++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t slot = (s >> 8) % 200, r0 = (s >> 20) % 32, r1 = (s >> 25) % 32;
        h[k] = (slot * 512u) | (r0 * 2u) | ((r1 * 2u) << 24);
    }
    CHECK(hipMalloc(&words, h.size() * 4));
    CHECK(hipMemcpy(words, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 3, 4}) {
        run<SINGLE>(wps, sink, words, e0, e1);
        run<PAIR>(wps, sink, words, e0, e1);
        run<ADDTID>(wps, sink, words, e0, e1);
        run<NOREAD>(wps, sink, words, e0, e1);
        if (wps <= 2) run<WIDE>(wps, sink, words, e0, e1);          // (192 registers: two waves per SIMD at most)
        if (wps <= 2) run<WIDEONLY>(wps, sink, words, e0, e1);
        run<B64ONLY>(wps, sink, words, e0, e1);
        run<SSINGLE>(wps, sink, words, e0, e1);
        run<SPAIR>(wps, sink, words, e0, e1);
    }
    return 0;
}
