// Instruction-issue probe for gfx950 (round 3): what does ONE step of an indexed, partially-occupied FMA
// update cost on the instruction side?  The column-sweep kernels select the accumulator row with the gfx9
// VGPR-indexing mode and the lane group with an execution mask; the four-lane-group form (cs_spmm16g4p) is
// instruction-bound.  This probe measures the pieces in isolation, with 1 / 2 / 4
// wavefronts per SIMD on every CU:
//
//   fma_full      v_fma_f32, all 64 lanes                      fma_half   the same under a 32-lane exec mask
//   fma_quarter   the same under a 16-lane exec mask           fmac_dpp   v_fmac_f32_dpp row_mask:0x3 (no exec write)
//   idx_idx       s_set_gpr_idx_idx alone                      exec_wr    s_mov_b64 exec alone
//   onoff         s_set_gpr_idx_on + 4 v_fma + s_set_gpr_idx_off            (the shipped G = 1 / 2 update)
//   exec_onoff    s_mov exec + on + 4 v_fma + off                            (the shipped G = 4 update, per group)
//   idx_dpp       s_set_gpr_idx_idx + 4 v_fmac_dpp row_mask (inside ONE on/off) (the lean G = 4 candidate)
//   pk_fma        v_pk_fma_f32 (two columns per lane)          pk_idx     s_mov exec + s_set_gpr_idx_idx + 2 v_pk_fma (lean G = 4, per group)
//   readlane      v_readlane_b32                              bperm      ds_bpermute_b32
//   lds_b64       ds_read_b64, half-wave broadcast address     lds_b64r   ds_read_b64, per-lane random 256-B pieces
//
// and checks the SEMANTICS the lean update relies on: v_fmac_f32_dpp inside the indexing mode (is the
// accumulator read relocated together with the destination?  does row_mask leave the other rows alone?).
//
// Output: one JSON line per (probe, waves per SIMD): ns and clocks (2.4 GHz nominal) per instruction GROUP per SIMD.
// Build + run: hipcc -O3 --offload-arch=gfx950 profiles/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { FMA_FULL, FMA_HALF, FMA_QUARTER, FMAC_DPP, IDX_IDX, EXEC_WR, ONOFF, EXEC_ONOFF, IDX_DPP, READLANE, BPERM, LDS_B64, LDS_B64R, PK_FMA, PK_IDX, NPROBE };
static const char* kNames[NPROBE] = {"fma_full", "fma_half", "fma_quarter", "fmac_dpp", "idx_idx", "exec_wr", "onoff",
                                     "exec_onoff", "idx_dpp", "readlane", "bperm", "lds_b64", "lds_b64r", "pk_fma", "pk_idx"};
// instructions per repetition of the body below (what "per group" means in the output)
static const int kGroup[NPROBE] = {1, 1, 1, 1, 1, 1, 6, 7, 5, 1, 1, 1, 1, 1, 4};

#define R8(x) x x x x x x x x
#define R32(x) R8(x) R8(x) R8(x) R8(x)

template <int P>
__global__ __launch_bounds__(1024) void probe(int reps, float* sink, const int* idxs) {
    extern __shared__ char smem[];
    float acc0 = threadIdx.x, acc1 = 1.f, acc2 = 2.f, acc3 = 3.f, b = 1.0001f, v = 0.5f;
    const int lane = threadIdx.x & 63;
    int s_i = idxs[0];          // 0 at run time (the compiler does not know)
    uint32_t lo = 0xffffffffu, zero = 0;
    uint32_t addr = (P == LDS_B64) ? (lane >> 5) * 8 : ((uint32_t)idxs[1 + (threadIdx.x & 1023)] * 256u + (lane & 31) * 8u);
    float d0 = 0.f, d1 = 0.f;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v p0 = {acc0, 1.f}, p1 = {2.f, 3.f}, p2 = {4.f, 5.f}, p3 = {6.f, 7.f}, pv = {0.5f, 0.25f}, pb = {1.0001f, 0.999f};
    if (P == LDS_B64 || P == LDS_B64R)
        for (int k = threadIdx.x; k < 32768; k += blockDim.x) reinterpret_cast<float*>(smem)[k] = 1.f;
    __syncthreads();
    for (int r = 0; r < reps; r++) {
        if constexpr (P == FMA_FULL) {
            asm volatile(R32("v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "s"(v), "v"(b));
        } else if constexpr (P == FMA_HALF || P == FMA_QUARTER) {
            asm volatile("s_mov_b64 s[20:21], exec\n\t"
                         "s_mov_b32 exec_lo, %6\n\t"
                         "s_mov_b32 exec_hi, 0\n\t"
                         R32("v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3\n\t")
                         "s_mov_b64 exec, s[20:21]"
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "s"(v), "v"(b), "s"(P == FMA_HALF ? 0xffffffffu : 0xffffu)
                         : "s20", "s21");
        } else if constexpr (P == FMAC_DPP) {
            asm volatile(R32("v_fmac_f32_dpp %0, %4, %5 quad_perm:[0,1,2,3] row_mask:0x3 bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %1, %4, %5 quad_perm:[0,1,2,3] row_mask:0x3 bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %2, %4, %5 quad_perm:[0,1,2,3] row_mask:0xc bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %3, %4, %5 quad_perm:[0,1,2,3] row_mask:0xc bank_mask:0xf\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(v), "v"(b));
        } else if constexpr (P == IDX_IDX) {
            asm volatile("s_set_gpr_idx_on %0, 0x8\n\t"
                         R32("s_set_gpr_idx_idx %0\n\ts_set_gpr_idx_idx %0\n\ts_set_gpr_idx_idx %0\n\ts_set_gpr_idx_idx %0\n\t")
                         "s_set_gpr_idx_off" : : "s"(s_i));
        } else if constexpr (P == EXEC_WR) {
            asm volatile("s_mov_b64 s[20:21], exec\n\t"
                         R32("s_mov_b64 exec, s[20:21]\n\ts_mov_b64 exec, s[20:21]\n\ts_mov_b64 exec, s[20:21]\n\ts_mov_b64 exec, s[20:21]\n\t")
                         : : : "s20", "s21");
        } else if constexpr (P == ONOFF) {
            asm volatile(R32("s_set_gpr_idx_on %6, 0xc\n\t"
                             "v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3\n\t"
                             "s_set_gpr_idx_off\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "s"(v), "v"(b), "s"(s_i));
        } else if constexpr (P == EXEC_ONOFF) {
            asm volatile("s_mov_b64 s[20:21], exec\n\t"
                         R32("s_mov_b64 exec, s[20:21]\n\t"
                             "s_set_gpr_idx_on %6, 0xc\n\t"
                             "v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3\n\t"
                             "s_set_gpr_idx_off\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "s"(v), "v"(b), "s"(s_i) : "s20", "s21");
        } else if constexpr (P == IDX_DPP) {
            asm volatile("s_set_gpr_idx_on %6, 0x8\n\t"
                         R32("s_set_gpr_idx_idx %6\n\t"
                             "v_fmac_f32_dpp %0, %4, %5 quad_perm:[0,1,2,3] row_mask:0x1 bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %1, %4, %5 quad_perm:[0,1,2,3] row_mask:0x1 bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %2, %4, %5 quad_perm:[0,1,2,3] row_mask:0x1 bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %3, %4, %5 quad_perm:[0,1,2,3] row_mask:0x1 bank_mask:0xf\n\t")
                         "s_set_gpr_idx_off"
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(v), "v"(b), "s"(s_i));
        } else if constexpr (P == PK_FMA) {
            asm volatile(R32("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1]\n\t"
                             "v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[0,1,1]\n\t")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pv), "v"(pb));
        } else if constexpr (P == PK_IDX) {
            asm volatile("s_mov_b64 s[20:21], exec\n\t"
                         "s_set_gpr_idx_on %6, 0xc\n\t"
                         R32("s_mov_b64 exec, s[20:21]\n\t"
                             "s_set_gpr_idx_idx %6\n\t"
                             "v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1]\n\t")
                         "s_set_gpr_idx_off"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pv), "v"(pb), "s"(s_i) : "s20", "s21");
        } else if constexpr (P == READLANE) {
            asm volatile(R32("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %0, 35\n\tv_readlane_b32 s22, %0, 7\n\tv_readlane_b32 s23, %0, 39\n\t")
                         : : "v"(acc0) : "s20", "s21", "s22", "s23");
        } else if constexpr (P == BPERM) {
            asm volatile(R32("ds_bpermute_b32 %0, %2, %1\n\tds_bpermute_b32 %0, %2, %1 offset:4\n\tds_bpermute_b32 %0, %2, %1 offset:8\n\tds_bpermute_b32 %0, %2, %1 offset:12\n\t")
                         "s_waitcnt lgkmcnt(0)" : "+v"(d0) : "v"(acc1), "v"(zero));
        } else if constexpr (P == LDS_B64 || P == LDS_B64R) {
            float2 x0, x1, x2, x3;
            asm volatile(R8("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:16\n\tds_read_b64 %2, %4 offset:32\n\tds_read_b64 %3, %4 offset:48\n\t"
                            "ds_read_b64 %0, %4 offset:64\n\tds_read_b64 %1, %4 offset:80\n\tds_read_b64 %2, %4 offset:96\n\tds_read_b64 %3, %4 offset:112\n\t"
                            "ds_read_b64 %0, %4 offset:128\n\tds_read_b64 %1, %4 offset:144\n\tds_read_b64 %2, %4 offset:160\n\tds_read_b64 %3, %4 offset:176\n\t"
                            "ds_read_b64 %0, %4 offset:192\n\tds_read_b64 %1, %4 offset:208\n\tds_read_b64 %2, %4 offset:224\n\tds_read_b64 %3, %4 offset:240\n\t")
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(addr) : "memory");
            d0 += x0.x + x1.x + x2.x + x3.x;
        }
    }
    if (acc0 + acc1 + acc2 + acc3 + d0 + d1 + (float)lo + p0.x + p0.y + p1.x + p2.y + p3.x == 1.2345f) sink[0] = acc0;
}

// ---- semantics of v_fmac_f32_dpp under the VGPR-indexing mode -------------------------------------------
// 8 consecutive registers v[40:47] hold 100..107 in every lane; under idx = 3:
//   mode 0x8 (DST only)      "v_fmac_f32_dpp v40, vval, vb row_mask:0x3"
//   mode 0xc (SRC2 | DST)    the same
// if the accumulator read follows the destination, v43 becomes 103 + val * b in lanes 0-31 and stays 103 in
// lanes 32-63, every other register is unchanged.
__global__ void semantics(float* out, int mode_c, const int* idxs) {
    typedef float v8 __attribute__((ext_vector_type(8)));
    v8 r = {100.f, 101.f, 102.f, 103.f, 104.f, 105.f, 106.f, 107.f};
    float val = 2.f, b = 10.f + (float)(threadIdx.x & 63);
    int idx = idxs[2048];          // 3
    if (mode_c)
        asm volatile("s_set_gpr_idx_on %3, 0xc\n\t"
                     "v_fmac_f32_dpp v40, %1, %2 quad_perm:[0,1,2,3] row_mask:0x3 bank_mask:0xf\n\t"
                     "s_set_gpr_idx_off" : "+{v[40:47]}"(r) : "v"(val), "v"(b), "s"(idx));
    else
        asm volatile("s_set_gpr_idx_on %3, 0x8\n\t"
                     "v_fmac_f32_dpp v40, %1, %2 quad_perm:[0,1,2,3] row_mask:0x3 bank_mask:0xf\n\t"
                     "s_set_gpr_idx_off" : "+{v[40:47]}"(r) : "v"(val), "v"(b), "s"(idx));
    for (int k = 0; k < 8; k++) out[(mode_c * 64 + threadIdx.x) * 8 + k] = r[k];
}

// v_pk_fma_f32 with a broadcast scalar value (op_sel_hi:[0,1,1]) under the indexing mode (SRC2 | DST), idx = 2, lanes 16-31 only:
// the pair (v42, v43) becomes (102 + val * b, 103 + val * (b + 0.5)) in lanes 16-31, nothing else moves.
__global__ void semantics_pk(float* out, const int* idxs) {
    typedef float v8 __attribute__((ext_vector_type(8)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    v8 r = {100.f, 101.f, 102.f, 103.f, 104.f, 105.f, 106.f, 107.f};
    f2v val = {2.f, 777.f};
    f2v b = {10.f + (float)(threadIdx.x & 63), 10.5f + (float)(threadIdx.x & 63)};
    int idx = idxs[2049];          // 2
    asm volatile("s_mov_b64 s[20:21], exec\n\t"
                 "s_mov_b32 exec_lo, 0xffff0000\n\t"
                 "s_mov_b32 exec_hi, 0\n\t"
                 "s_set_gpr_idx_on %3, 0xc\n\t"
                 "v_pk_fma_f32 v[40:41], %1, %2, v[40:41] op_sel_hi:[0,1,1]\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "s_mov_b64 exec, s[20:21]" : "+{v[40:47]}"(r) : "v"(val), "v"(b), "s"(idx) : "s20", "s21");
    for (int k = 0; k < 8; k++) out[threadIdx.x * 8 + k] = r[k];
}

template <int P>
static void run(int wps, float* sink, const int* idxs, hipEvent_t e0, hipEvent_t e1) {
    const int reps = 400;
    const int threads = 256 * wps;
    const size_t lds = 128 * 1024;           // one workgroup per CU
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(probe<P>, dim3(256), dim3(threads), lds, 0, 20, sink, idxs);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<P>, dim3(256), dim3(threads), lds, 0, reps, sink, idxs);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_rep = (P == ONOFF || P == EXEC_ONOFF || P == IDX_DPP || P == PK_IDX) ? 32.0 : 128.0;
    const double groups = (double)reps * per_rep * wps;         // groups issued per SIMD
    const double ns = ms * 1e6 / groups;
    printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"instr_per_group\": %d, \"ns_per_group_per_simd\": %.3f, \"clk_per_group_per_simd\": %.2f, "
           "\"clk_per_instr_per_simd\": %.2f, \"ms\": %.3f}\n", kNames[P], wps, kGroup[P], ns, ns * 2.4, ns * 2.4 / kGroup[P], ms);
}

int main() {
    float* sink;
    int* idxs;
    CHECK(hipMalloc(&sink, 64 * 2 * 8 * sizeof(float)));
    std::vector<int> h(4096, 0);
    uint32_t s = 12345;
    for (int k = 1; k <= 1024; k++) { s = s * 1664525u + 1013904223u; h[k] = (s >> 8) % 500; }
    h[2048] = 3;
    h[2049] = 2;
    CHECK(hipMalloc(&idxs, h.size() * sizeof(int)));
    CHECK(hipMemcpy(idxs, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    for (int m = 0; m < 2; m++) hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, sink, m, idxs);
    CHECK(hipDeviceSynchronize());
    std::vector<float> o(64 * 2 * 8);
    CHECK(hipMemcpy(o.data(), sink, o.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int m = 0; m < 2; m++) {
        int ok = 1;
        for (int l = 0; l < 64; l++)
            for (int k = 0; k < 8; k++) {
                float want = 100.f + k;
                if (k == 3 && l < 32) want = 103.f + 2.f * (10.f + l);
                if (o[(m * 64 + l) * 8 + k] != want) ok = 0;
            }
        printf("{\"semantics\": \"v_fmac_f32_dpp row_mask:0x3 under s_set_gpr_idx_on idx, %s\", \"relocates_accumulator_and_masks_rows\": %s, "
               "\"lane0\": [%.0f, %.0f, %.0f, %.0f, %.0f], \"lane40\": [%.0f, %.0f, %.0f, %.0f, %.0f]}\n", m ? "0xc" : "0x8", ok ? "true" : "false",
               o[(m * 64) * 8 + 0], o[(m * 64) * 8 + 1], o[(m * 64) * 8 + 2], o[(m * 64) * 8 + 3], o[(m * 64) * 8 + 4],
               o[(m * 64 + 40) * 8 + 0], o[(m * 64 + 40) * 8 + 1], o[(m * 64 + 40) * 8 + 2], o[(m * 64 + 40) * 8 + 3], o[(m * 64 + 40) * 8 + 4]);
    }
    hipLaunchKernelGGL(semantics_pk, dim3(1), dim3(64), 0, 0, sink, idxs);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(o.data(), sink, 64 * 8 * sizeof(float), hipMemcpyDeviceToHost));
    {
        int ok = 1;
        for (int l = 0; l < 64; l++)
            for (int k = 0; k < 8; k++) {
                float want = 100.f + k;
                if (l >= 16 && l < 32 && k == 2) want = 102.f + 2.f * (10.f + l);
                if (l >= 16 && l < 32 && k == 3) want = 103.f + 2.f * (10.5f + l);
                if (o[l * 8 + k] != want) ok = 0;
            }
        printf("{\"semantics\": \"v_pk_fma_f32 op_sel_hi:[0,1,1] under s_set_gpr_idx_on 2, 0xc and a 16-lane exec mask\", \"as_expected\": %s, "
               "\"lane20\": [%.1f, %.1f, %.1f, %.1f, %.1f], \"lane3\": [%.1f, %.1f, %.1f, %.1f]}\n", ok ? "true" : "false",
               o[20 * 8 + 0], o[20 * 8 + 1], o[20 * 8 + 2], o[20 * 8 + 3], o[20 * 8 + 4], o[3 * 8 + 0], o[3 * 8 + 1], o[3 * 8 + 2], o[3 * 8 + 3]);
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4}) {
        run<FMA_FULL>(wps, sink, idxs, e0, e1);
        run<FMA_HALF>(wps, sink, idxs, e0, e1);
        run<FMA_QUARTER>(wps, sink, idxs, e0, e1);
        run<FMAC_DPP>(wps, sink, idxs, e0, e1);
        run<IDX_IDX>(wps, sink, idxs, e0, e1);
        run<EXEC_WR>(wps, sink, idxs, e0, e1);
        run<ONOFF>(wps, sink, idxs, e0, e1);
        run<EXEC_ONOFF>(wps, sink, idxs, e0, e1);
        run<IDX_DPP>(wps, sink, idxs, e0, e1);
        run<READLANE>(wps, sink, idxs, e0, e1);
        run<BPERM>(wps, sink, idxs, e0, e1);
        run<LDS_B64>(wps, sink, idxs, e0, e1);
        run<LDS_B64R>(wps, sink, idxs, e0, e1);
        run<PK_FMA>(wps, sink, idxs, e0, e1);
        run<PK_IDX>(wps, sink, idxs, e0, e1);
    }
    return 0;
}
