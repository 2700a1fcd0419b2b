// Column-sweep SpMM for static graphs on gfx950 (plan: include/sgcn.h sgcn_csplan_t).
//
// Why: the row-gather kernel (sgcn_spmm.hip) is bound by XCD<->fabric bandwidth -- every nonzero
// re-fetches its 2.4 KB B row because the 4 MiB L2 of an XCD holds 0.7 % of B (measured:
// 55.6 GB fetched for 1.3 GB of algorithmic bytes, L2 hit 3 %).  Here one wavefront keeps the
// accumulators of a TILE of 16 (virtual) rows x 64 float4 in 64 VGPRs and walks the tile's
// nonzeros in COLUMN order; all wavefronts of a launch start together and advance through the
// column space at the same average rate, so the B rows a wave needs were just fetched into its
// XCD's L2 by a neighbour.  The leading wave pays the miss, the followers hit -- misses also
// slow leaders down, which keeps the pack together.  B is fetched ~once per XCD per launch
// instead of once per nonzero.
//
// The local row id travels in the top 4 bits of the column word and selects the accumulator
// through the gfx9 VGPR-indexing mode (s_set_gpr_idx_on lr; v_fma v<plane>...; s_set_gpr_idx_off):
// the accumulators are PINNED to fixed registers, so a nonzero costs one indexed FMA group -- no
// branches, no LDS, no register copies.  The sweep is clock-paced (s_memrealtime) so that all
// waves of an XCD stay inside one L2-sized window of B.
// Split rows / ordered fix-up exactly as in sgcn_spmm.hip (deterministic, no atomics).
//
// Where its time goes (profiles/gather_ceiling.*): L2 hits and fabric misses do
// not overlap in the vector memory path -- T ~ miss_bytes / 7.2 TB/s + hit_bytes / 30 TB/s -- and
// the kernel sits on that line; the levers left are fewer fabric bytes (rows resident per XCD) and
// graphs with locality (grouped plans, xcd_map).
//
// Kernels in this file (what the product dispatches; the forms measured and dropped on the way -- plain and unpacked
// two-group kernels, four lane groups per wave (unpacked: round 2; packed, one round of resident tiles: commit af99d18,
// 3.75 ms and a no-arithmetic floor of 3.21 ms, profiles/r29_g4k_probe.jsonl), the compiler-indexed generic form with
// 32-row tiles -- live in git history, their measurements in profiles/HISTORY.md 3.1b and DESIGN.md 3.2):
//   cs_spmm16_kernel<U, EXTRA>     one 16-row tile per wavefront, pinned accumulators, up to 320 columns per pass
//   cs_spmm16g2k_kernel<U, WIDE>   two 16-row bins per wavefront on 128-column passes, software-pipelined, packed FMAs:
//                                  what ColumnSweepCSR.choose_g selects for most widths (bench default at d = 602)
//   cs_fix_kernel<VW>              ordered sum of a split row's workspace slots
// Every asm statement that enters the VGPR-indexing mode (s_set_gpr_idx_*: the index lives in M0) or points M0 at an
// LDS-DMA destination lists "m0" as a clobber, so that LLVM's M0-initialisation merging never carries a value across
// it.  M0 is a reserved (non-allocatable) register, which makes clang warn about the entry; the entry is intended.
#pragma clang diagnostic ignored "-Winline-asm"
#include "sgcn_dev.h"

namespace sgcn {

typedef Vec<4>::type f4;

struct CsArgs {
    const int64_t* tile_ptr;
    const uint32_t* colrow;
    const float* val;
    const int32_t* tile_rows;
    const int32_t* tile_slots;
    int64_t tile_base, tile_end;
    const float* B; int64_t ldb;
    const int32_t* gidx; const float* rscale; const float* cscale;
    float* C; int64_t ldc; float beta;
    int32_t d, nvec, slab;
    int32_t slab_floats;   // columns per pass of the pinned kernel: 64 float4 (+ up to 64 extra floats)
    float* ws; int64_t ldw;
    float cols_per_tick;   // pacing: columns the sweep may advance per 100 MHz tick (0 = unpaced)
    float slack_cols;      // how far ahead of the clock a wave may run
    int32_t xcd_map;       // consecutive tiles of the launch on the same XCD (grouped plans)
    const uint32_t* warp;  // sweep position of column bucket (col >> warp_shift), in [0, K): the clock runs in WORK
    int32_t warp_shift;    // coordinates (plan->dev_warp; null: positions are the column ids themselves)
};

// The sweep clock in work coordinates.  A clock that is linear in the COLUMN ID holds the lock-step only if the nonzeros are
// spread evenly over the column ids (S-Reddit: hubs carry random ids).  Where they are not -- R-MAT: a third of the nonzeros
// sit in the first sixteenth of the ids -- every wave falls behind such a clock in the dense ranges and free-runs, i.e. the
// sweep is unpaced exactly where the sharing is to be had.  With a warp table the position of a column is the share of the
// matrix's nonzeros that lie in front of its bucket (scaled to [0, K), so pace and slack keep their units): one scalar load
// per batch of gathers, issued a batch ahead like the clock read, from a table of at most 64 KiB (scalar cache / L2).
typedef const __attribute__((address_space(4))) uint32_t* cs_warp_ptr;
__device__ __forceinline__ cs_warp_ptr cs_warp_table(const CsArgs& a) {
    return (cs_warp_ptr)(uintptr_t)a.warp;
}

// Workgroup -> first tile of the launch.  The dispatcher deals workgroups round-robin over the 8
// XCDs (workgroup b runs on XCD b % 8); with xcd_map the launch's tile range is cut into 8
// contiguous pieces, one per XCD, so tiles that are adjacent in the plan -- same row group, same B
// rows -- share an L2.  Pure placement: any dispatch order gives the same result.
__device__ __forceinline__ int64_t cs_first_tile(const CsArgs& a) {
    const unsigned b = blockIdx.x, nb = gridDim.x;
    if (!a.xcd_map || nb < 8) return a.tile_base + (int64_t)b * (kBlock / kWave);
    const unsigned x = b & 7u, q = nb >> 3, r = nb & 7u;
    const unsigned start = x * q + (x < r ? x : r);            // workgroups of XCDs 0..x-1
    return a.tile_base + (int64_t)(start + (b >> 3)) * (kBlock / kWave);
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

#define SGCN_CS_ROWS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// R = 16 specialisation with the accumulators PINNED to fixed VGPRs and updated by ONE indexed FMA
// group per nonzero:
//     s_set_gpr_idx_on lr, SRC2|DST ; v_fma_f32 v<plane>, val, b, v<plane> x 4 (or 5) ; s_set_gpr_idx_off
// The generic kernel below lets hipcc lower `acc[e][lr]`, which costs an on/off pair per extract
// AND per insert (16 scalar + 12 vector instructions per nonzero instead of 2 + 4).
//
// EXTRA: a sweep costs the same whatever the slab width (measured: 1.45 ms per pass at d = 256,
// 512, 602 or 768 on S-Reddit), so a 602-wide row must not take three 256-float passes.  With
// EXTRA every lane carries one more fp32 column next to its float4 -- a slab is 64 float4 + up to
// 64 floats = up to 320 columns -- and d = 602 (pitch 608) is covered by TWO passes of 304.
// Planes: x,y,z,w (+ e) of 16 registers each at v[64:127] (v[48:127] with EXTRA).
template <int U, bool EXTRA, bool WARP = false>
__global__ __launch_bounds__(kBlock) void cs_spmm16_kernel(CsArgs a) {
    typedef Vec<4>::type VT;
    constexpr int kShift = 28;
    constexpr uint32_t kColMask = (1u << kShift) - 1u;
    const int lane = threadIdx.x & 63;
    const int64_t tile = cs_first_tile(a) + threadIdx.x / kWave;
    if (tile >= a.tile_end) return;
    const int fbase = a.slab * a.slab_floats;                    // first column of this pass
    const int f4 = fbase + lane * 4;                             // my float4
    const bool act4 = lane * 4 < min(a.slab_floats, 256) && f4 < a.d;
    const int fe = fbase + 256 + lane;                           // my extra column
    const bool acte = EXTRA && lane < a.slab_floats - 256 && fe < a.d;
    // inactive lanes re-read a valid address of the row instead of branching around the load
    const uint32_t off4 = (uint32_t)(act4 ? f4 : fbase) * 4u;
    const uint32_t offe = (uint32_t)(acte ? fe : fbase) * 4u;
    const char* Bb = reinterpret_cast<const char*>(a.B);
    const int64_t ldb_bytes = a.ldb * 4;

    typedef float accv_t __attribute__((ext_vector_type(16)));
    accv_t ax = {}, ay = {}, az = {}, aw = {}, ae = {};

    // Clock-paced sweep: every wave of a launch starts within ~1 us and holds its column position
    // to `elapsed * cols_per_tick` on the chip-wide constant 100 MHz counter (s_memrealtime), so
    // all waves of an XCD gather from the same L2-sized window of B at the same time without
    // exchanging a single message (counting nonzeros is not enough: a tile's position after k
    // nonzeros jitters by ~N/(2 sqrt(nnz_tile)) columns, more than the window).
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t tnow = t0;
    // WARP: the clock in work coordinates -- a column's position is its warp table entry, looked up a batch ahead (the
    // first batch of a block of 64 nonzeros waits for its own: one exposed scalar load per ~20 us of gathers)
    const cs_warp_ptr wtab = cs_warp_table(a);
    const uint32_t wshift = (uint32_t)a.warp_shift;
    float pnext = 0.f;
    const int64_t start = a.tile_ptr[tile], end = a.tile_ptr[tile + 1];
    for (int64_t p0 = start; p0 < end; p0 += kWave) {
        const int n = (int)min((int64_t)kWave, end - p0);
        uint32_t mycr = 0;
        float myv = 0.f;
        if (lane < n) {
            mycr = a.colrow[p0 + lane];
            myv = a.val[p0 + lane];
            uint32_t c = mycr & kColMask;
            if (a.cscale) myv *= a.cscale[c];
            if (a.gidx) { c = (uint32_t)a.gidx[c]; mycr = (mycr & ~kColMask) | c; }
        }
        auto apply = [&](uint32_t cr, float v, VT b, float be) {
            const int lr = (int)(cr >> kShift);
            if constexpr (EXTRA) {
                asm volatile("s_set_gpr_idx_on %5, 0xc\n\t"
                             "v_fma_f32 v48, %6, %7, v48\n\t"
                             "v_fma_f32 v64, %6, %8, v64\n\t"
                             "v_fma_f32 v80, %6, %9, v80\n\t"
                             "v_fma_f32 v96, %6, %10, v96\n\t"
                             "v_fma_f32 v112, %6, %11, v112\n\t"
                             "s_set_gpr_idx_off"
                             : "+{v[48:63]}"(ax), "+{v[64:79]}"(ay), "+{v[80:95]}"(az), "+{v[96:111]}"(aw),
                               "+{v[112:127]}"(ae)
                             : "s"(lr), "s"(v), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(be)
                             : "m0");
            } else {
                asm volatile("s_set_gpr_idx_on %4, 0xc\n\t"
                             "v_fma_f32 v64, %5, %6, v64\n\t"
                             "v_fma_f32 v80, %5, %7, v80\n\t"
                             "v_fma_f32 v96, %5, %8, v96\n\t"
                             "v_fma_f32 v112, %5, %9, v112\n\t"
                             "s_set_gpr_idx_off"
                             : "+{v[64:79]}"(ax), "+{v[80:95]}"(ay), "+{v[96:111]}"(az), "+{v[112:127]}"(aw)
                             : "s"(lr), "s"(v), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w)
                             : "m0");
            }
        };
        const int nb = n / U;
        for (int k = 0; k < nb; k++) {
            const int jj = k * U;
            // The clock read is a long-latency scalar memory op: issued right after a batch's
            // gathers, consumed before the NEXT batch (the stale reading only adds look-ahead).
            if (a.cols_per_tick > 0.f) {
                float mycol;
                if constexpr (WARP) {
                    if (k == 0) pnext = (float)wtab[((uint32_t)__builtin_amdgcn_readlane((int)mycr, 0) & kColMask) >> wshift];
                    mycol = pnext;
                } else {
                    mycol = (float)((uint32_t)__builtin_amdgcn_readlane((int)mycr, jj) & kColMask);
                }
                float allowed = (float)(tnow - t0) * a.cols_per_tick + a.slack_cols;
                for (int spin = 0; spin < 4096 && mycol > allowed; spin++) {   // bounded: never hangs
                    __builtin_amdgcn_s_sleep(8);
                    allowed = (float)(__builtin_amdgcn_s_memrealtime() - t0) * a.cols_per_tick + a.slack_cols;
                }
            }
            VT bb[U];
            float be[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t cr = (uint32_t)__builtin_amdgcn_readlane((int)mycr, jj + u);
                const char* src = Bb + (int64_t)(cr & kColMask) * ldb_bytes;
                bb[u] = *reinterpret_cast<const VT*>(src + off4);
                be[u] = EXTRA ? *reinterpret_cast<const float*>(src + offe) : 0.f;
            }
            if (a.cols_per_tick > 0.f) tnow = __builtin_amdgcn_s_memrealtime();
            if constexpr (WARP) {
                if (a.cols_per_tick > 0.f && k + 1 < nb)
                    pnext = (float)wtab[((uint32_t)__builtin_amdgcn_readlane((int)mycr, jj + U) & kColMask) >> wshift];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t cr = (uint32_t)__builtin_amdgcn_readlane((int)mycr, jj + u);
                const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myv), jj + u));
                apply(cr, v, bb[u], be[u]);
            }
        }
        for (int j = nb * U; j < n; j++) {
            const uint32_t cr = (uint32_t)__builtin_amdgcn_readlane((int)mycr, j);
            const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myv), j));
            const char* src = Bb + (int64_t)(cr & kColMask) * ldb_bytes;
            const VT b = *reinterpret_cast<const VT*>(src + off4);
            const float e1 = EXTRA ? *reinterpret_cast<const float*>(src + offe) : 0.f;
            apply(cr, v, b, e1);
        }
    }

    const int32_t* rows = a.tile_rows + tile * 16;
    const int32_t* slots = a.tile_slots + tile * 16;
    const int left = a.d - f4;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = rows[r];
        if (row < 0) continue;
        const VT accv = {ax[r], ay[r], az[r], aw[r]};
        const float acce = ae[r];
        const int slot = slots[r];
        if (slot >= 0) {
            float* w = a.ws + (int64_t)slot * a.ldw;
            if (act4) vstore<4>(w + f4, accv);
            if (acte) w[fe] = acce;
        } else {
            float* out = a.C + (int64_t)row * a.ldc;
            const float rs = a.rscale ? a.rscale[row] : 1.0f;
            if (act4) {
                VT res = accv * rs;
                if (a.beta != 0.f) {
                    if (left >= 4) res += a.beta * vload<4>(out + f4);
                    else for (int e = 0; e < left; e++) res[e] += a.beta * out[f4 + e];
                }
                if (left >= 4) vstore<4>(out + f4, res); else vstore_head<4>(out + f4, res, left);
            }
            if (acte) out[fe] = acce * rs + (a.beta != 0.f ? a.beta * out[fe] : 0.f);
        }
    }
}


// ---- two lane groups per wavefront (plan->G == 2): the default for most operand widths ------------------------------
// A tile is TWO bins of 16 virtual rows (sgcn_csplang_*): lanes 0-31 hold the accumulators of bin 0, lanes 32-63 those of
// bin 1, each lane one float4 of a 128-column slab; ONE dwordx4 load per step gathers the 512-byte pieces of two
// different B rows ("x4 on 2 rows" in profiles/gather_ceiling.json: 28 TB/s from L2) and a wave holds 32 rows instead of
// 16 -- half the passes of B through every XCD per register byte.  The plan keeps the two bins' column positions within
// `align` columns of each other (pads), so a wave gathers from ONE L2 window (profiles/l2_sweep_sim.py).
//
// History of the step (profiles/HISTORY.md 3.1b; the earlier forms live in git history): plain form
// 4.4 ms per S-Reddit SpMM; software-pipelined, 8 half-wave v_fma + 15 scalar + 2 ds_bpermute per step: 3.5 ms; THIS
// form, 3.2 ms.  profiles/issue_probe.hip says what a step costs on the instruction side of gfx950: a v_fma_f32 occupies
// its SIMD for 4.4 clocks WHATEVER the execution mask (a half-wave update is not cheaper than a full one), a
// v_pk_fma_f32 for 5.0 (two columns per lane), a scalar instruction 1.05 clocks of the CU's one scalar unit, a
// ds_bpermute 6 clocks of the CU's LDS pipeline (three times a ds_read_b64).  So a row's (x, y) and (z, w) live in
// adjacent, even-aligned register pairs and a step is 4 packed FMAs + 2 v_readlane, 8 scalar instructions and ONE
// ds_bpermute:
//  * the gathers of batch k+1 are in flight while batch k is applied (two register buffers of U float4), across chunk
//    boundaries; the next chunk's entries are fetched a whole chunk ahead;
//  * a lane gets ITS bin's column of a step by ds_bpermute, the row offset is one v_mad_u32_u24, the load uses the
//    scalar-base + 32-bit-offset form; the two values of a step come down by v_readlane into fixed scalar pairs;
//  * the 64 accumulator offsets of a chunk are packed into 16 scalars once per chunk (2 DPP ORs + 16 v_readlane) and
//    read as bytes; the chunk's pad mask is ONE v_cmp, tested per step by s_bitcmp + s_cselect_b64 on exec;
//  * the clock comparison is integer scalar arithmetic.
// Requires every tile's entry count to be a multiple of 64 (the plan pads to that).  WIDE: K >= 2^24 or B beyond 4 GiB
// -- 64-bit row offsets.
template <int U, bool WIDE, bool WARP = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void cs_spmm16g2k_kernel(CsArgs a) {
    typedef Vec<4>::type VT;
    constexpr int kShift = 28;
    constexpr uint32_t kColMask = (1u << kShift) - 1u;
    constexpr int kSteps = kWave / 2;                 // steps per chunk of 64 entries
    constexpr int kLanesPerStep = 2;
    constexpr int kBatches = kSteps / U;
    static_assert(kBatches % 2 == 0, "the two buffers alternate evenly over a chunk");
    const int lane = threadIdx.x & 63;
    const bool hi = lane >= 32;
    const int li = lane & 31;
    const int64_t tile = cs_first_tile(a) + threadIdx.x / kWave;
    if (tile >= a.tile_end) return;
    const int fbase = a.slab * 128;
    const int f4 = fbase + li * 4;
    const bool act = f4 < a.d;
    const uint32_t off4 = (uint32_t)(act ? f4 : fbase) * 4u;
    const char* Bb = reinterpret_cast<const char*>(a.B);
    const uint32_t ldb32 = (uint32_t)(a.ldb * 4);
    const int sel0 = hi ? 4 : 0;                      // ds_bpermute byte address of entry (2 j + bin) is sel0 + 8 j

    // row r of a bin: (x, y) = axy[2 r], axy[2 r + 1] and (z, w) = azw[2 r], azw[2 r + 1] -- adjacent, even-aligned register
    // pairs, so that ONE v_pk_fma_f32 updates two columns
    typedef float accv_t __attribute__((ext_vector_type(32)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    accv_t axy = {}, azw = {};
    const uint64_t lo64 = 0x00000000ffffffffull, hi64 = 0xffffffff00000000ull;

    const uint32_t t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    uint32_t tnow = t0;
    // columns per tick in 16.16 fixed point (a launch lasts < 2^16 ticks of 10 ns; K / ticks < 2^15)
    const uint32_t cpt16 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.cols_per_tick * 65536.0f));
    const uint32_t slack = (uint32_t)a.slack_cols;
    const int64_t start = a.tile_ptr[tile], end = a.tile_ptr[tile + 1];

    // one chunk of plan entries: lane l holds entry l (even lanes bin 0, odd lanes bin 1 of step l / 2)
    auto entries = [&](int64_t p, uint32_t& cr, float& v) {
        cr = 0;
        v = __int_as_float((int)0x80000000);                       // beyond the tile: pads on column 0
        if (p < end) {
            cr = a.colrow[p + lane];
            v = a.val[p + lane];
            uint32_t c = cr & kColMask;
            if (a.cscale && __float_as_int(v) != (int)0x80000000) {
                v *= a.cscale[c];
                if (__float_as_int(v) == (int)0x80000000) v = 0.f;
            }
            if (a.gidx) { c = (uint32_t)a.gidx[c]; cr = (cr & ~kColMask) | c; }
        }
    };
    // per-chunk scalars: the register offset (2 x local row id) of every entry's accumulator pair, 4 lanes x 8 bits per
    // word -- s_set_gpr_idx_on / _idx read the low byte of their operand, so an id costs at most one shift -- and the pad mask
    struct Meta { uint32_t lr[16]; uint32_t pad_lo, pad_hi; };
    auto meta = [&](uint32_t cr, float v) -> Meta {
        Meta m;
        int x = (int)(((cr >> kShift) << 1) << (8 * (lane & 3)));
        x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1
        x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);      // row_shr:2 -> lane 4g+3 holds group g
#pragma unroll
        for (int g = 0; g < 16; g++) m.lr[g] = (uint32_t)__builtin_amdgcn_readlane(x, 4 * g + 3);
        const uint64_t pm = __ballot(__float_as_int(v) == (int)0x80000000);
        m.pad_lo = (uint32_t)pm; m.pad_hi = (uint32_t)(pm >> 32);
        return m;
    };
    auto pace = [&](uint32_t crs, int j) {
        if (cpt16 != 0) {
            const uint32_t mycol = (uint32_t)__builtin_amdgcn_readlane((int)crs, 2 * j) & kColMask;
            uint32_t allowed = (uint32_t)(((uint64_t)(tnow - t0) * cpt16) >> 16) + slack;
            for (int spin = 0; spin < 4096 && mycol > allowed; spin++) {
                __builtin_amdgcn_s_sleep(8);
                allowed = (uint32_t)(((uint64_t)((uint32_t)__builtin_amdgcn_s_memrealtime() - t0) * cpt16) >> 16) + slack;
            }
        }
    };
    // WARP: the same wait on a POSITION that was looked up a batch earlier (wpos: the scalar load of the table entry of
    // step j's first column; its latency lies under the batch that is applied in between)
    const cs_warp_ptr wtab = cs_warp_table(a);
    const uint32_t wshift = (uint32_t)a.warp_shift;
    auto wpos = [&](uint32_t crs, int j) -> uint32_t {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)crs, kLanesPerStep * j) & kColMask;
        return wtab[c >> wshift];
    };
    auto pace_at = [&](uint32_t pos) {
        if (cpt16 != 0) {
            uint32_t allowed = (uint32_t)(((uint64_t)(tnow - t0) * cpt16) >> 16) + slack;
            for (int spin = 0; spin < 4096 && pos > allowed; spin++) {
                __builtin_amdgcn_s_sleep(8);
                allowed = (uint32_t)(((uint64_t)((uint32_t)__builtin_amdgcn_s_memrealtime() - t0) * cpt16) >> 16) + slack;
            }
        }
    };
    auto gather = [&](uint32_t crs, int j) -> VT {
        const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(sel0 + 8 * j, (int)crs);   // my bin's column word
        if constexpr (WIDE) {
            return *reinterpret_cast<const VT*>(Bb + (uint64_t)(c & kColMask) * ldb32 + off4);
        } else {                                       // u24 multiply: the row id above bit 24 is ignored by the instruction
            const uint32_t off = __umul24(c, ldb32) + off4;
            return *reinterpret_cast<const VT*>(Bb + off);
        }
    };
    // step j of the chunk (compile-time j).  The two values come down by v_readlane into FIXED scalar pairs (the packed
    // FMA takes a 64-bit scalar operand and, with op_sel_hi:[0,1,1], uses its low word for both columns); a pad's
    // execution mask is empty (s_bitcmp0 on the chunk's pad mask + ONE s_cselect_b64); the indexing mode is switched on
    // once per step and re-pointed for the second bin.  6 VALU (2 v_readlane + 4 v_pk_fma_f32) and 8 scalar instructions
    // per step, and one ds_bpermute (the column word).
    auto fma2 = [&](const Meta& m, float vs, auto jc, VT b) {
        constexpr int j = decltype(jc)::value;
        const int l0 = (int)(m.lr[j >> 1] >> ((j & 1) * 16));
        const int l1 = (int)(m.lr[j >> 1] >> ((j & 1) * 16 + 8));
        const uint32_t pw = j < 16 ? m.pad_lo : m.pad_hi;
        const f2_t bxy = {b.x, b.y}, bzw = {b.z, b.w};
        accv_t& rxy = axy;               // (named here: operands that appear only inside a dependent asm statement are not captured)
        accv_t& rzw = azw;
        const uint64_t mlo = lo64, mhi = hi64;
        asm volatile("v_readlane_b32 s20, %[vs], %[e0]\n\t"
                     "v_readlane_b32 s22, %[vs], %[e1]\n\t"
                     "s_bitcmp0_b32 %[pw], %[b0]\n\t"
                     "s_cselect_b64 exec, %[lo], 0\n\t"
                     "s_set_gpr_idx_on %[l0], 0xc\n\t"
                     "v_pk_fma_f32 v[64:65], s[20:21], %[bxy], v[64:65] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 v[96:97], s[20:21], %[bzw], v[96:97] op_sel_hi:[0,1,1]\n\t"
                     "s_bitcmp0_b32 %[pw], %[b1]\n\t"
                     "s_cselect_b64 exec, %[hi], 0\n\t"
                     "s_set_gpr_idx_idx %[l1]\n\t"
                     "v_pk_fma_f32 v[64:65], s[22:23], %[bxy], v[64:65] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 v[96:97], s[22:23], %[bzw], v[96:97] op_sel_hi:[0,1,1]\n\t"
                     "s_set_gpr_idx_off\n\t"
                     "s_mov_b64 exec, -1"
                     : "+{v[64:95]}"(rxy), "+{v[96:127]}"(rzw)
                     : [vs] "v"(vs), [e0] "i"(2 * j), [e1] "i"(2 * j + 1), [pw] "s"(pw), [b0] "i"((2 * j) & 31),
                       [b1] "i"(((2 * j) & 31) + 1), [lo] "s"(mlo), [hi] "s"(mhi), [l0] "s"(l0), [l1] "s"(l1),
                       [bxy] "v"(bxy), [bzw] "v"(bzw)
                     : "s20", "s21", "s22", "s23", "scc", "m0");
    };

    uint32_t ccr, ncr;
    float cv, nv;
    entries(start, ccr, cv);
    entries(start + kWave, ncr, nv);
    Meta cm = meta(ccr, cv);
    VT buf[2][U];
    uint32_t pnext = 0;                             // WARP: position of the next batch's first column
    if constexpr (WARP) { if (cpt16 != 0) pace_at(wpos(ccr, 0)); } else pace(ccr, 0);
#pragma unroll
    for (int u = 0; u < U; u++) buf[0][u] = gather(ccr, u);
    if (cpt16 != 0) tnow = (uint32_t)__builtin_amdgcn_s_memrealtime();
    if constexpr (WARP) { if (cpt16 != 0) pnext = wpos(ccr, U); }
    for (int64_t p0 = start; p0 < end; p0 += kWave) {
        static_for<kBatches>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            // batch k+1 goes in flight (the first batch of the NEXT chunk after this chunk's last) ...
            if constexpr (k + 1 < kBatches) {
                if constexpr (WARP) pace_at(pnext); else pace(ccr, (k + 1) * U);
#pragma unroll
                for (int u = 0; u < U; u++) buf[(k + 1) & 1][u] = gather(ccr, (k + 1) * U + u);
            } else {
                if constexpr (WARP) pace_at(pnext); else pace(ncr, 0);
#pragma unroll
                for (int u = 0; u < U; u++) buf[(k + 1) & 1][u] = gather(ncr, u);
            }
            if (cpt16 != 0) tnow = (uint32_t)__builtin_amdgcn_s_memrealtime();
            if constexpr (WARP) {                  // batch k + 2's position: asked for now, waited on in the next iteration
                if (cpt16 != 0) {
                    if constexpr (k + 2 < kBatches) pnext = wpos(ccr, (k + 2) * U);
                    else pnext = wpos(ncr, (k + 2 - kBatches) * U);
                }
            }
            // ... while batch k is applied
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                fma2(cm, cv, std::integral_constant<int, k * U + u>{}, buf[k & 1][u]);
            });
        });
        ccr = ncr; cv = nv;
        cm = meta(ccr, cv);
        entries(p0 + 2 * kWave, ncr, nv);
    }

    const int32_t* rows = a.tile_rows + tile * 32 + (hi ? 16 : 0);
    const int32_t* slots = a.tile_slots + tile * 32 + (hi ? 16 : 0);
    const int left = a.d - f4;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = rows[r];
        const VT accv = {axy[2 * r], axy[2 * r + 1], azw[2 * r], azw[2 * r + 1]};
        if (row < 0 || !act) continue;
        const int slot = slots[r];
        if (slot >= 0) {
            vstore<4>(a.ws + (int64_t)slot * a.ldw + f4, accv);
        } else {
            float* out = a.C + (int64_t)row * a.ldc;
            const float rs = a.rscale ? a.rscale[row] : 1.0f;
            VT res = accv * rs;
            if (a.beta != 0.f) {
                if (left >= 4) res += a.beta * vload<4>(out + f4);
                else for (int e = 0; e < left; e++) res[e] += a.beta * out[f4 + e];
            }
            if (left >= 4) vstore<4>(out + f4, res); else vstore_head<4>(out + f4, res, left);
        }
    }
}

// ---- four lane groups per wavefront (plan->G == 4): every row of the graph resident in ONE round of tiles ---------------
// A tile is FOUR bins of 16 virtual rows; lanes 16 g .. 16 g + 15 hold bin g, each lane one float4 of a 64-column slab, and
// one dwordx4 load per step gathers the 256-byte pieces of four different B rows ("x4 on 4 rows" in
// profiles/gather_ceiling.json: 25 TB/s from L2 against 28.9 for two rows).  4,096 resident waves then hold 262,144 rows:
// S-Reddit's 232,965 in ONE round, so every XCD fetches B once per product instead of twice -- the fabric bytes are what the
// two-group kernel has left on the table.  The price is on the instruction side: a step is 8 packed FMAs
// under four quarter masks + 4 v_readlane and 14 scalar instructions for the same 1 KiB of gathers, and ceil(d / 64)
// passes.  Same structure as cs_spmm16g2k_kernel otherwise (entry l of a chunk: step l / 4, bin l % 4; the accumulator
// offsets of a step are the four bytes of ONE scalar).  The unpacked round-2 form of this kernel was instruction-bound at
// 4.4 ms (profiles/HISTORY.md 3.1b).
template <int U, bool WIDE, bool WARP = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void cs_spmm16g4k_kernel(CsArgs a) {
    typedef Vec<4>::type VT;
    constexpr int kShift = 28;
    constexpr uint32_t kColMask = (1u << kShift) - 1u;
    constexpr int kSteps = kWave / 4;                 // steps per chunk of 64 entries
    constexpr int kLanesPerStep = 4;
    constexpr int kBatches = kSteps / U;
    static_assert(kBatches % 2 == 0, "the two buffers alternate evenly over a chunk");
    const int lane = threadIdx.x & 63;
    const int bin = lane >> 4;
    const int li = lane & 15;
    const int64_t tile = cs_first_tile(a) + threadIdx.x / kWave;
    if (tile >= a.tile_end) return;
    const int fbase = a.slab * 64;
    const int f4 = fbase + li * 4;
    const bool act = f4 < a.d;
    const uint32_t off4 = (uint32_t)(act ? f4 : fbase) * 4u;
    const char* Bb = reinterpret_cast<const char*>(a.B);
    const uint32_t ldb32 = (uint32_t)(a.ldb * 4);
    const int sel0 = bin * 4;                         // ds_bpermute byte address of entry (4 j + bin) is sel0 + 16 j

    typedef float accv_t __attribute__((ext_vector_type(32)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    accv_t axy = {}, azw = {};
    const uint64_t q0 = 0x000000000000ffffull, q1 = 0x00000000ffff0000ull, q2 = 0x0000ffff00000000ull, q3 = 0xffff000000000000ull;

    const uint32_t t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    uint32_t tnow = t0;
    const uint32_t cpt16 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.cols_per_tick * 65536.0f));
    const uint32_t slack = (uint32_t)a.slack_cols;
    const int64_t start = a.tile_ptr[tile], end = a.tile_ptr[tile + 1];

    auto entries = [&](int64_t p, uint32_t& cr, float& v) {
        cr = 0;
        v = __int_as_float((int)0x80000000);                       // beyond the tile: pads on column 0
        if (p < end) {
            cr = a.colrow[p + lane];
            v = a.val[p + lane];
            uint32_t c = cr & kColMask;
            if (a.cscale && __float_as_int(v) != (int)0x80000000) {
                v *= a.cscale[c];
                if (__float_as_int(v) == (int)0x80000000) v = 0.f;
            }
            if (a.gidx) { c = (uint32_t)a.gidx[c]; cr = (cr & ~kColMask) | c; }
        }
    };
    // per-chunk scalars: word j = the register offsets (2 x local row id) of step j's four entries, one byte each
    struct Meta { uint32_t lr[16]; uint32_t pad_lo, pad_hi; };
    auto meta = [&](uint32_t cr, float v) -> Meta {
        Meta m;
        int x = (int)(((cr >> kShift) << 1) << (8 * (lane & 3)));
        x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1
        x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);      // row_shr:2 -> lane 4j+3 holds step j
#pragma unroll
        for (int g = 0; g < 16; g++) m.lr[g] = (uint32_t)__builtin_amdgcn_readlane(x, 4 * g + 3);
        const uint64_t pm = __ballot(__float_as_int(v) == (int)0x80000000);
        m.pad_lo = (uint32_t)pm; m.pad_hi = (uint32_t)(pm >> 32);
        return m;
    };
    auto pace = [&](uint32_t crs, int j) {
        if (cpt16 != 0) {
            const uint32_t mycol = (uint32_t)__builtin_amdgcn_readlane((int)crs, 4 * j) & kColMask;
            uint32_t allowed = (uint32_t)(((uint64_t)(tnow - t0) * cpt16) >> 16) + slack;
            for (int spin = 0; spin < 4096 && mycol > allowed; spin++) {
                __builtin_amdgcn_s_sleep(8);
                allowed = (uint32_t)(((uint64_t)((uint32_t)__builtin_amdgcn_s_memrealtime() - t0) * cpt16) >> 16) + slack;
            }
        }
    };
    // WARP: the same wait on a POSITION that was looked up a batch earlier (wpos: the scalar load of the table entry of
    // step j's first column; its latency lies under the batch that is applied in between)
    const cs_warp_ptr wtab = cs_warp_table(a);
    const uint32_t wshift = (uint32_t)a.warp_shift;
    auto wpos = [&](uint32_t crs, int j) -> uint32_t {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)crs, kLanesPerStep * j) & kColMask;
        return wtab[c >> wshift];
    };
    auto pace_at = [&](uint32_t pos) {
        if (cpt16 != 0) {
            uint32_t allowed = (uint32_t)(((uint64_t)(tnow - t0) * cpt16) >> 16) + slack;
            for (int spin = 0; spin < 4096 && pos > allowed; spin++) {
                __builtin_amdgcn_s_sleep(8);
                allowed = (uint32_t)(((uint64_t)((uint32_t)__builtin_amdgcn_s_memrealtime() - t0) * cpt16) >> 16) + slack;
            }
        }
    };
    auto gather = [&](uint32_t crs, int j) -> VT {
        const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(sel0 + 16 * j, (int)crs);   // my bin's column word
        if constexpr (WIDE) {
            return *reinterpret_cast<const VT*>(Bb + (uint64_t)(c & kColMask) * ldb32 + off4);
        } else {
            const uint32_t off = __umul24(c, ldb32) + off4;
            return *reinterpret_cast<const VT*>(Bb + off);
        }
    };
    // step j: 4 v_readlane (the values, into fixed scalar pairs), then per bin: pad test -> quarter mask, the bin's
    // accumulator offset (byte g of the step's word: one s_lshr between bins), two packed FMAs
    auto fma4 = [&](const Meta& m, float vs, auto jc, VT b) {
        constexpr int j = decltype(jc)::value;
        const uint32_t lw = m.lr[j];
        const uint32_t pw = j < 8 ? m.pad_lo : m.pad_hi;
        const f2_t bxy = {b.x, b.y}, bzw = {b.z, b.w};
        accv_t& rxy = axy;
        accv_t& rzw = azw;
        const uint64_t m0 = q0, m1 = q1, m2 = q2, m3 = q3;
        asm volatile("v_readlane_b32 s20, %[vs], %[e0]\n\t"
                     "v_readlane_b32 s22, %[vs], %[e0]+1\n\t"
                     "v_readlane_b32 s24, %[vs], %[e0]+2\n\t"
                     "v_readlane_b32 s26, %[vs], %[e0]+3\n\t"
                     "s_bitcmp0_b32 %[pw], %[b0]\n\t"
                     "s_cselect_b64 exec, %[m0], 0\n\t"
                     "s_set_gpr_idx_on %[lw], 0xc\n\t"
                     "v_pk_fma_f32 v[64:65], s[20:21], %[bxy], v[64:65] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 v[96:97], s[20:21], %[bzw], v[96:97] op_sel_hi:[0,1,1]\n\t"
                     "s_lshr_b32 s28, %[lw], 8\n\t"
                     "s_bitcmp0_b32 %[pw], %[b0]+1\n\t"
                     "s_cselect_b64 exec, %[m1], 0\n\t"
                     "s_set_gpr_idx_idx s28\n\t"
                     "v_pk_fma_f32 v[64:65], s[22:23], %[bxy], v[64:65] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 v[96:97], s[22:23], %[bzw], v[96:97] op_sel_hi:[0,1,1]\n\t"
                     "s_lshr_b32 s28, %[lw], 16\n\t"
                     "s_bitcmp0_b32 %[pw], %[b0]+2\n\t"
                     "s_cselect_b64 exec, %[m2], 0\n\t"
                     "s_set_gpr_idx_idx s28\n\t"
                     "v_pk_fma_f32 v[64:65], s[24:25], %[bxy], v[64:65] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 v[96:97], s[24:25], %[bzw], v[96:97] op_sel_hi:[0,1,1]\n\t"
                     "s_lshr_b32 s28, %[lw], 24\n\t"
                     "s_bitcmp0_b32 %[pw], %[b0]+3\n\t"
                     "s_cselect_b64 exec, %[m3], 0\n\t"
                     "s_set_gpr_idx_idx s28\n\t"
                     "v_pk_fma_f32 v[64:65], s[26:27], %[bxy], v[64:65] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 v[96:97], s[26:27], %[bzw], v[96:97] op_sel_hi:[0,1,1]\n\t"
                     "s_set_gpr_idx_off\n\t"
                     "s_mov_b64 exec, -1"
                     : "+{v[64:95]}"(rxy), "+{v[96:127]}"(rzw)
                     : [vs] "v"(vs), [e0] "i"(4 * j), [pw] "s"(pw), [b0] "i"((4 * j) & 31),
                       [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3), [lw] "s"(lw),
                       [bxy] "v"(bxy), [bzw] "v"(bzw)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "scc", "m0");
    };

    uint32_t ccr, ncr;
    float cv, nv;
    entries(start, ccr, cv);
    entries(start + kWave, ncr, nv);
    Meta cm = meta(ccr, cv);
    VT buf[2][U];
    uint32_t pnext = 0;                             // WARP: position of the next batch's first column
    if constexpr (WARP) { if (cpt16 != 0) pace_at(wpos(ccr, 0)); } else pace(ccr, 0);
#pragma unroll
    for (int u = 0; u < U; u++) buf[0][u] = gather(ccr, u);
    if (cpt16 != 0) tnow = (uint32_t)__builtin_amdgcn_s_memrealtime();
    if constexpr (WARP) { if (cpt16 != 0) pnext = wpos(ccr, U); }
    for (int64_t p0 = start; p0 < end; p0 += kWave) {
        static_for<kBatches>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k + 1 < kBatches) {
                if constexpr (WARP) pace_at(pnext); else pace(ccr, (k + 1) * U);
#pragma unroll
                for (int u = 0; u < U; u++) buf[(k + 1) & 1][u] = gather(ccr, (k + 1) * U + u);
            } else {
                if constexpr (WARP) pace_at(pnext); else pace(ncr, 0);
#pragma unroll
                for (int u = 0; u < U; u++) buf[(k + 1) & 1][u] = gather(ncr, u);
            }
            if (cpt16 != 0) tnow = (uint32_t)__builtin_amdgcn_s_memrealtime();
            if constexpr (WARP) {                  // batch k + 2's position: asked for now, waited on in the next iteration
                if (cpt16 != 0) {
                    if constexpr (k + 2 < kBatches) pnext = wpos(ccr, (k + 2) * U);
                    else pnext = wpos(ncr, (k + 2 - kBatches) * U);
                }
            }
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                fma4(cm, cv, std::integral_constant<int, k * U + u>{}, buf[k & 1][u]);
            });
        });
        ccr = ncr; cv = nv;
        cm = meta(ccr, cv);
        entries(p0 + 2 * kWave, ncr, nv);
    }

    const int32_t* rows = a.tile_rows + tile * 64 + bin * 16;
    const int32_t* slots = a.tile_slots + tile * 64 + bin * 16;
    const int left = a.d - f4;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = rows[r];
        const VT accv = {axy[2 * r], axy[2 * r + 1], azw[2 * r], azw[2 * r + 1]};
        if (row < 0 || !act) continue;
        const int slot = slots[r];
        if (slot >= 0) {
            vstore<4>(a.ws + (int64_t)slot * a.ldw + f4, accv);
        } else {
            float* out = a.C + (int64_t)row * a.ldc;
            const float rs = a.rscale ? a.rscale[row] : 1.0f;
            VT res = accv * rs;
            if (a.beta != 0.f) {
                if (left >= 4) res += a.beta * vload<4>(out + f4);
                else for (int e = 0; e < left; e++) res[e] += a.beta * out[f4 + e];
            }
            if (left >= 4) vstore<4>(out + f4, res); else vstore_head<4>(out + f4, res, left);
        }
    }
}

// fix-up of split rows: ordered slot sum + epilogue (float4 path only).  An ITEM = 64 float4 columns of one split row.
// A row of up to four slots -- every row of a column-range plan (round 6: each row of a small block has one piece per
// range), most rows of any plan -- is summed by ONE wave in slot order; a workgroup takes four items, a wave each: no LDS,
// no barrier, a quarter of the workgroups (an eighth of S-Reddit, 29 k rows of two or three slots: the launch was one
// workgroup of four waves per item, two of them idle).  When one of a workgroup's four rows has more slots (a hub row of
// an R-MAT block has hundreds: one thread walking them all was 0.35 ms of a 4.3 ms product) the workgroup does its items
// one after the other, the four waves taking consecutive quarters of the row's slots, each summing its quarter in slot
// order and wave 0 adding the four partial sums in wave order.  Either way a fixed association -- for up to four slots
// both forms ARE the plain slot order, bit for bit -- so reruns stay bit-identical.
template <int VW>
__device__ __forceinline__ void cs_fix_store(const CsArgs& a, const sgcn_fix_t& fx, int vi,
                                             typename Vec<VW>::type acc) {
    typedef typename Vec<VW>::type VT;
    float* out = a.C + (int64_t)fx.row * a.ldc + (int64_t)vi * VW;
    VT res = acc * (a.rscale ? a.rscale[fx.row] : 1.0f);
    const int left = a.d - vi * VW;
    if (a.beta != 0.f) {
        if (left >= VW) res += a.beta * vload<VW>(out);
        else for (int e = 0; e < left; e++) { if constexpr (VW == 1) res += a.beta * out[0]; else res[e] += a.beta * out[e]; }
    }
    if (left >= VW) vstore<VW>(out, res); else vstore_head<VW>(out, res, left);
}

template <int VW>
__global__ __launch_bounds__(kBlock) void cs_fix_kernel(CsArgs a, const sgcn_fix_t* fix, int64_t nfix) {
    typedef typename Vec<VW>::type VT;
    constexpr int NWV = kBlock / kWave;
    __shared__ VT part[NWV - 1][kWave];
    const int nvblk = (a.nvec + kWave - 1) / kWave;
    const int64_t nitems = nfix * nvblk, item0 = (int64_t)blockIdx.x * NWV;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const int64_t mine = item0 + w;
    sgcn_fix_t fx{0, 0, 0};
    if (mine < nitems) fx = fix[mine / nvblk];
    if (__syncthreads_and(fx.nslots <= NWV)) {
        if (mine >= nitems) return;
        const int vi = (int)(mine % nvblk) * kWave + lane;
        if (vi >= a.nvec) return;
        const float* wp = a.ws + (int64_t)fx.first_slot * a.ldw + (int64_t)vi * VW;
        VT acc = vzero<VW>();
        for (int q = 0; q < fx.nslots; q++) acc += vload<VW>(wp + (int64_t)q * a.ldw);
        cs_fix_store<VW>(a, fx, vi, acc);
        return;
    }
    for (int k = 0; k < NWV; k++) {
        const int64_t item = item0 + k;
        if (item >= nitems) break;                        // (uniform over the workgroup)
        const sgcn_fix_t fk = fix[item / nvblk];
        const int vi = (int)(item % nvblk) * kWave + lane;
        const bool live = vi < a.nvec;
        const int q0 = (int)((int64_t)fk.nslots * w / NWV), q1 = (int)((int64_t)fk.nslots * (w + 1) / NWV);
        VT acc = vzero<VW>();
        if (live) {
            const float* wp = a.ws + (int64_t)fk.first_slot * a.ldw + (int64_t)vi * VW;
#pragma unroll 4
            for (int q = q0; q < q1; q++) acc += vload<VW>(wp + (int64_t)q * a.ldw);
        }
        if (w > 0) part[w - 1][lane] = acc;
        __syncthreads();
        if (w == 0 && live) {
#pragma unroll
            for (int j = 0; j < NWV - 1; j++) acc += part[j][lane];
            cs_fix_store<VW>(a, fk, vi, acc);
        }
        __syncthreads();                                  // `part` is written again by the next item
    }
}

}  // namespace sgcn

using namespace sgcn;

namespace {
// What sgcn_spmm_cs_f32 dispatches for (plan, d) under the current knobs: passes over the feature
// dimension, columns per pass, the fifth plane, gathers in flight -- and the kernel's name.
struct CsVariant { int nslab, slab_floats, U; bool extra; const char* name; };

CsVariant cs_variant(const sgcn_csplan_t* plan, int d) {
    CsVariant v{};
    if (plan->G == 4) {             // four lane groups per wave: 64-column passes, every row resident in one round
        v.nslab = ((d + 3) / 4 * 4 + 63) / 64;
        v.slab_floats = 64;
        v.extra = false;
        v.U = 4;
        v.name = "sgcn::cs_spmm16g4k_kernel<4, false>";
        return v;
    }
    if (plan->G == 2) {             // two lane groups per wave: 128-column passes, one dwordx4 per step
        v.nslab = ((d + 3) / 4 * 4 + 127) / 128;
        v.slab_floats = 128;
        v.extra = false;
        v.U = 4;
        v.name = "sgcn::cs_spmm16g2k_kernel<4, false>";     // <4, true> when B needs 64-bit row offsets
        return v;
    }
    const int nvec = (d + 3) / 4;
    v.U = tune_get("cs_unroll") > 0 ? tune_get("cs_unroll") : 8;
    v.nslab = (nvec + kWave - 1) / kWave;
    v.slab_floats = 256;
    // a pass costs the same whatever its width: cover d in ceil(d / 320) passes of
    // (64 float4 + extra floats) instead of ceil(d / 256) passes of 64 float4
    const int dp = (d + 3) / 4 * 4;
    const int np = (dp + 319) / 320;
    if (np < v.nslab && tune_get("cs_noextra") <= 0) {
        v.nslab = np;
        v.slab_floats = ((dp + np - 1) / np + 3) / 4 * 4;
        v.extra = v.slab_floats > 256;
    }
    // EXTRA at U = 8 needs 142 VGPRs (3 waves/SIMD, breaks the 4096-tile residency): U = 4
    if (v.extra) v.U = tune_get("cs_unroll") == 8 ? 8 : 4;
    else v.U = v.U == 4 ? 4 : 8;
    v.name = v.extra ? (v.U == 8 ? "sgcn::cs_spmm16_kernel<8, true>" : "sgcn::cs_spmm16_kernel<4, true>")
                     : (v.U == 8 ? "sgcn::cs_spmm16_kernel<8, false>" : "sgcn::cs_spmm16_kernel<4, false>");
    return v;
}
}  // namespace

extern "C" int sgcn_spmm_cs_variant(const sgcn_csplan_t* plan, int32_t d, char* buf, int32_t buflen) {
    SGCN_REQUIRE(plan && buf && buflen > 0 && d > 0, "spmm_cs_variant: bad argument");
    SGCN_REQUIRE(plan->R == 16 && (plan->G == 1 || plan->G == 2 || plan->G == 4 || plan->G == 0), "spmm_cs_variant: 16-row bins; one, two or four lane groups");
    const CsVariant v = cs_variant(plan, d);
    int64_t round = plan->round_tiles > 0 ? plan->round_tiles : (tune_get("cs_round") > 0 ? tune_get("cs_round") : 4096);
    snprintf(buf, (size_t)buflen, "%s x %d launches (%d passes of %d columns x %lld rounds of %lld tiles)", v.name,
             (int)(v.nslab * ((plan->ntiles + round - 1) / round)), v.nslab, v.slab_floats,
             (long long)((plan->ntiles + round - 1) / round), (long long)round);
    return SGCN_OK;
}

extern "C" int sgcn_spmm_cs_f32(const sgcn_csplan_t* plan, int32_t M, int32_t K, int32_t d,
                                const float* B, int64_t ldb, const int32_t* gidx,
                                const float* rscale, const float* cscale, float* C, int64_t ldc,
                                float beta, void* stream) {
    SGCN_REQUIRE(plan && M >= 0 && K >= 0 && d >= 0, "spmm_cs: bad argument");
    if (M == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(plan->R == 16 && (plan->G == 0 || plan->G == 1 || plan->G == 2 || plan->G == 4),
                 "spmm_cs: the plan must have 16-row bins and one, two or four lane groups per wavefront");
    SGCN_REQUIRE(plan->dev_tile_ptr && plan->dev_tile_rows && plan->dev_tile_slots && B && C,
                 "spmm_cs: null operand");
    SGCN_REQUIRE(K < (1 << 28), "spmm_cs: K too large for the packed column word");
    const int VW = 4;               // float4 per lane (16-byte aligned rows)
    SGCN_REQUIRE(pick_vw(d, {B, C, plan->dev_ws}, {ldb, ldc}) >= VW,
                 "spmm_cs: rows must be %d-byte aligned (pitch multiple of %d floats covering d)", VW * 4, VW);
    hipStream_t st = (hipStream_t)stream;
    CsArgs a{};
    a.tile_ptr = plan->dev_tile_ptr; a.colrow = reinterpret_cast<const uint32_t*>(plan->dev_colrow);
    a.val = plan->dev_val; a.tile_rows = plan->dev_tile_rows; a.tile_slots = plan->dev_tile_slots;
    a.B = B; a.ldb = ldb; a.gidx = gidx; a.rscale = rscale; a.cscale = cscale;
    a.C = C; a.ldc = ldc; a.beta = beta; a.d = d; a.nvec = (d + VW - 1) / VW;
    a.ws = plan->dev_ws; a.ldw = ((int64_t)d + 3) / 4 * 4;
    a.xcd_map = plan->xcd_map;
    a.warp = tune_get("cs_nowarp") <= 0 && !gidx ? plan->dev_warp : nullptr;
    a.warp_shift = plan->warp_shift;
    SGCN_REQUIRE(!a.warp || (plan->warp_shift >= 0 && plan->warp_shift < 28), "spmm_cs: bad warp_shift");
    SGCN_REQUIRE((plan->G != 2 && plan->G != 4) || ldb * 4 < (1ll << 32), "spmm_cs: row pitch of B must fit 32 bits");
    if (plan->nfix > 0) {
        SGCN_REQUIRE(plan->dev_fix && plan->dev_ws && plan->ws_elems >= plan->nslots * a.ldw,
                     "spmm_cs: workspace missing or too small");
    }
    int64_t round = plan->round_tiles;
    if (round <= 0) {
        const int tuned = tune_get("cs_round");
        round = tuned > 0 ? tuned : 4096;       // 256 CUs x 4 SIMDs x 4 waves of 64 acc VGPRs
    }
    // pace: plan value wins (set by the autotuner); 0 -> global knob; < 0 -> unpaced
    const int pace_ns_per_nnz = plan->pace_ns_per_nnz != 0 ? (plan->pace_ns_per_nnz < 0 ? 0 : plan->pace_ns_per_nnz)
                                                           : tune_get("cs_pace");
    const int slack = tune_get("cs_slack") > 0 ? tune_get("cs_slack") : 512;
    const CsVariant var = cs_variant(plan, d);
    const int nslab = var.nslab, U = var.U;
    const bool extra = var.extra;
    a.slab_floats = var.slab_floats;
    for (int slab = 0; slab < nslab; slab++) {
        a.slab = slab;
        for (int64_t t0 = 0; t0 < plan->ntiles; t0 += round) {
            a.tile_base = t0;
            a.tile_end = std::min(plan->ntiles, t0 + round);
            a.cols_per_tick = 0.f;
            a.slack_cols = (float)slack;
            if (pace_ns_per_nnz > 0 && plan->host_tile_nnz_hint) {
                double launch_ns = (double)plan->host_tile_nnz_hint[t0 / round] * pace_ns_per_nnz;
                // A last pass that covers at most 3/4 of a slab gathers fewer cache lines per step and holds the
                // lock-step on a faster clock (d = 602, G = 2, sustained: 3.16 ms per product at 80 %, 3.22 at 90, 3.28 at 100, 3.43 at 70:
                // profiles/r27_headline_knobs.jsonl).
                if (nslab > 1 && slab == nslab - 1 && tune_get("cs_last_pct") > 0 &&
                    4 * (((d + 3) / 4 * 4) - slab * var.slab_floats) <= 3 * var.slab_floats)
                    launch_ns *= tune_get("cs_last_pct") / 100.0;
                a.cols_per_tick = (float)((double)K / (launch_ns / 10.0));   // 100 MHz: 10 ns per tick
            }
            const unsigned blocks = (unsigned)((a.tile_end - t0 + 3) / 4);
#define SGCN_CS16(UU, EE)                                                                                              \
            do {                                                                                                        \
                if (a.warp != nullptr && a.cols_per_tick > 0.f)                                                         \
                    hipLaunchKernelGGL((cs_spmm16_kernel<UU, EE, true>), dim3(blocks), dim3(kBlock), 0, st, a);         \
                else hipLaunchKernelGGL((cs_spmm16_kernel<UU, EE, false>), dim3(blocks), dim3(kBlock), 0, st, a);       \
            } while (0)
            // the clock in work coordinates (plan->dev_warp): only a paced launch looks positions up
            const bool warp = a.warp != nullptr && a.cols_per_tick > 0.f;
#define SGCN_CSG(KERNEL, WIDE_)                                                                                         \
            do {                                                                                                        \
                if (warp) hipLaunchKernelGGL((KERNEL<4, WIDE_, true>), dim3(blocks), dim3(kBlock), 0, st, a);          \
                else hipLaunchKernelGGL((KERNEL<4, WIDE_, false>), dim3(blocks), dim3(kBlock), 0, st, a);              \
            } while (0)
            if (plan->G == 4) {
                const bool wide = K >= (1 << 24) || ldb * 4 >= (1 << 24) || (int64_t)K * ldb * 4 >= (1ll << 32) ||
                                  tune_get("cs_g2_wide") > 0;
                if (wide) SGCN_CSG(cs_spmm16g4k_kernel, true); else SGCN_CSG(cs_spmm16g4k_kernel, false);
            } else if (plan->G == 2) {
                const bool wide = K >= (1 << 24) || ldb * 4 >= (1 << 24) || (int64_t)K * ldb * 4 >= (1ll << 32) ||
                                  tune_get("cs_g2_wide") > 0;
                if (wide) SGCN_CSG(cs_spmm16g2k_kernel, true); else SGCN_CSG(cs_spmm16g2k_kernel, false);
            } else {
                if (extra) { if (U == 8) SGCN_CS16(8, true); else SGCN_CS16(4, true); }
                else { if (U == 4) SGCN_CS16(4, false); else SGCN_CS16(8, false); }
            }
#undef SGCN_CS16
#undef SGCN_CSG
        }
    }
    SGCN_HIP_TRY(hipGetLastError());
    if (plan->nfix > 0) {
        // an item = 64 float4 columns of a split row; a workgroup takes four (cs_fix_kernel)
        const int64_t nfblk = ((int64_t)((a.nvec + kWave - 1) / kWave) * plan->nfix + kBlock / kWave - 1) / (kBlock / kWave);
        SGCN_REQUIRE(nfblk < (1ll << 31), "spmm_cs: too many split rows");
        hipLaunchKernelGGL(cs_fix_kernel<4>, dim3((unsigned)nfblk), dim3(kBlock), 0, st, a, plan->dev_fix, plan->nfix);
        SGCN_HIP_TRY(hipGetLastError());
    }
    return SGCN_OK;
}
