// LDS-staged column sweep for static graphs WITH locality (plan: include/sgcn.h sgcn_ldsplan_t).
//
// Why: the column-sweep kernels of sgcn_spmm_cs.hip move one d-float row of B from an L2 to the VGPRs PER NONZERO
// (nnz * d * 4 = 55.8 GB per S-Reddit product; the L2s deliver <= 30 TB/s of such gathers: >= 1.8 ms whatever the hit
// rate).  The only level below the L2 that can absorb that traffic is the LDS, and it can only if the rows that live on
// ONE compute unit share columns.  On a graph with communities they do: a tile of 768 rows of one community references
// every column of that community ~7 times.  So here
//   * one workgroup (8 wavefronts, 2 per SIMD at 256 VGPRs) owns a TILE of 8 x RW virtual rows; a wave keeps the
//     RW x (64 lanes x VW floats) accumulators of its rows in 192 pinned VGPRs (v[64:255]);
//   * the tile's nonzeros are sorted by (community-ordered) column and cut into CHUNKS of at most S distinct columns;
//     a chunk's S pieces of B (one 64*VW-float slab of a B row each) are staged ONCE into one half of an LDS ring by
//     global_load_lds_dwordx4 (no staging registers) while the previous chunk is consumed from the other half;
//   * a nonzero is then: ds_read (a whole-wave, conflict-free read of the staged piece), s_set_gpr_idx_idx (selects the
//     row's accumulators: the gfx9 VGPR-indexing mode, as in sgcn_spmm_cs.hip) and VW/2 v_pk_fma_f32.  Its plan entry
//     {value, LDS address of the piece | register offset of the row} arrives through the SCALAR cache (s_load), so the
//     vector memory path carries nothing but the ring fills and the LDS path nothing but pieces.
// Columns a tile references fewer than `min_reuse` times are not worth a ring slot; the plan leaves those nonzeros to
// a residual CSR that the ordinary column sweep adds afterwards (ops.LdsSweepCSR).
//
// Determinism: an accumulator receives its nonzeros in plan order; split rows meet in the ordered fix-up
// (cs_fix-style); no atomics.  Same contract as sgcn_spmm_cs_f32 minus cscale.
#include "sgcn_dev.h"

namespace sgcn {

struct LdsArgs {
    const int32_t* tile_chunk_ptr;      // [ntiles + 1]
    const int32_t* chunk_cols;          // [nchunks * S] column (B row) of every ring slot, padded with a valid column
    const int64_t* ent_ptr;             // [nchunks * NW + 1], entries in (tile, wave, chunk) order
    const uint64_t* entries;            // {value bits, word}: word = LDS byte address of the piece | register offset
    const int32_t* tile_rows;           // [ntiles * NW * RW]
    const int32_t* tile_slots;
    const float* B; int64_t ldb;
    const int32_t* gidx; const float* rscale;
    float* C; int64_t ldc; float beta;
    int32_t d, nslab, ntiles;
    float* ws; int64_t ldw;
};

typedef uint32_t ent16_t __attribute__((ext_vector_type(16)));     // 8 plan entries {value, word} in 16 SGPRs
typedef float acc32_t __attribute__((ext_vector_type(32)));

// ---- the inner loop: ONE asm statement per (wave, chunk) ---------------------------------------------------------------
// Registers (fixed: the statement is the only code that runs while loads are in flight, so nothing the compiler does --
// spills, copies -- can observe a register a load has not written yet):
//   E0 E1 E2  s[36:51] s[52:67] s[68:83]   three groups of 8 plan entries, rotating: applied / read for / being loaded.
//                                           Operands ("+s"): they carry the read-ahead from one chunk to the next.
//   V0 V1     v[32:47] v[48:63]            pieces of the group being applied / of the next group (clobbers)
//   v[28:31]                               LDS addresses (clobbers)
//   a0 .. a5  v[64:255]                    the accumulators: row r of the wave at v[64 + 2 r : 65 + 2 r] ("+v")
// A GROUP (8 nonzeros) of the steady state, phase P = group index mod 6 (the entry buffers rotate with period 3, the
// piece buffers with period 2):
//     s_waitcnt lgkmcnt(0)        this group's pieces and the next group's entries have landed -- both were requested
//                                 before the previous group's eight FMAs
//     8 x (v_and_or_b32, ds_read_b64)   next group's pieces: address = (word & ~511) | lane * 8
//     s_load_dwordx16             the entries of the group after next, into the buffer whose FMAs were issued a group ago
//     8 x (s_set_gpr_idx_idx, v_pk_fma_f32)   the row's accumulators are selected by the word's low byte (the gfx9
//                                 VGPR-indexing mode); the value is the low word of the entry's scalar pair (op_sel_hi)
// i.e. per nonzero 2 VALU + ~2 scalar instructions and one LDS read; no vector-memory instruction at all.
// The statement ends with everything landed (s_waitcnt lgkmcnt(0)); the read-ahead of the chunk's last group ran into the
// next chunk's ring half, so the next statement starts by reading its first group's pieces again (after the barrier).
#define SGCN_LDS_READ8(EN, VN)                                                 \
    "v_and_or_b32 v28, s[" #EN "+1], %[mask], %[lane]\n\t"                      \
    "v_and_or_b32 v29, s[" #EN "+3], %[mask], %[lane]\n\t"                      \
    "v_and_or_b32 v30, s[" #EN "+5], %[mask], %[lane]\n\t"                      \
    "v_and_or_b32 v31, s[" #EN "+7], %[mask], %[lane]\n\t"                      \
    "ds_read_b64 v[" #VN "+0:" #VN "+1], v28\n\t"                               \
    "ds_read_b64 v[" #VN "+2:" #VN "+3], v29\n\t"                               \
    "ds_read_b64 v[" #VN "+4:" #VN "+5], v30\n\t"                               \
    "ds_read_b64 v[" #VN "+6:" #VN "+7], v31\n\t"                               \
    "v_and_or_b32 v28, s[" #EN "+9], %[mask], %[lane]\n\t"                      \
    "v_and_or_b32 v29, s[" #EN "+11], %[mask], %[lane]\n\t"                     \
    "v_and_or_b32 v30, s[" #EN "+13], %[mask], %[lane]\n\t"                     \
    "v_and_or_b32 v31, s[" #EN "+15], %[mask], %[lane]\n\t"                     \
    "ds_read_b64 v[" #VN "+8:" #VN "+9], v28\n\t"                               \
    "ds_read_b64 v[" #VN "+10:" #VN "+11], v29\n\t"                             \
    "ds_read_b64 v[" #VN "+12:" #VN "+13], v30\n\t"                             \
    "ds_read_b64 v[" #VN "+14:" #VN "+15], v31\n\t"
#define SGCN_LDS_FMA1(EC, VC, K, OP)                                           \
    #OP " s[" #EC "+2*" #K "+1]" SGCN_LDS_IDXMODE_##OP "\n\t"                    \
    "v_pk_fma_f32 v[64:65], s[" #EC "+2*" #K ":" #EC "+2*" #K "+1], v[" #VC "+2*" #K ":" #VC "+2*" #K "+1], v[64:65] op_sel_hi:[0,1,1]\n\t"
#define SGCN_LDS_IDXMODE_s_set_gpr_idx_on ", 0xc"
#define SGCN_LDS_IDXMODE_s_set_gpr_idx_idx ""
#define SGCN_LDS_GROUP(P, EC, EN, EL, VC, VN, PNEXT)                           \
    "Lg" #P "_%=:\n\t"                                                          \
    "s_waitcnt lgkmcnt(0)\n\t"                                                  \
    SGCN_LDS_READ8(EN, VN)                                                      \
    "s_load_dwordx16 s[" #EL ":" #EL "+15], %[base], %[off]\n\t"                \
    SGCN_LDS_FMA1(EC, VC, 0, s_set_gpr_idx_on)                                  \
    SGCN_LDS_FMA1(EC, VC, 1, s_set_gpr_idx_idx)                                 \
    SGCN_LDS_FMA1(EC, VC, 2, s_set_gpr_idx_idx)                                 \
    SGCN_LDS_FMA1(EC, VC, 3, s_set_gpr_idx_idx)                                 \
    SGCN_LDS_FMA1(EC, VC, 4, s_set_gpr_idx_idx)                                 \
    SGCN_LDS_FMA1(EC, VC, 5, s_set_gpr_idx_idx)                                 \
    SGCN_LDS_FMA1(EC, VC, 6, s_set_gpr_idx_idx)                                 \
    SGCN_LDS_FMA1(EC, VC, 7, s_set_gpr_idx_idx)                                 \
    "s_set_gpr_idx_off\n\t"                                                     \
    "s_add_u32 %[off], %[off], 64\n\t"                                          \
    "s_sub_u32 %[n], %[n], 1\n\t"                                               \
    "s_cmp_eq_u32 %[n], 0\n\t"                                                  \
    "s_cbranch_scc0 Lg" #PNEXT "_%=\n\t"                                        \
    "s_mov_b32 %[ph], " #PNEXT "\n\t"                                           \
    "s_branch Lx_%=\n\t"
#define SGCN_LDS_ENTRY(P, EC, VC)                                              \
    "Le" #P "_%=:\n\t"                                                          \
    SGCN_LDS_READ8(EC, VC)                                                      \
    "s_branch Lg" #P "_%=\n\t"

// n groups (> 0) of this wave's entry stream, starting in phase `ph` (both updated); `off` = byte offset (from `base`) of
// the entries two groups ahead of the first one (what the next s_load fetches; updated).
__device__ __forceinline__ void lds_chunk(ent16_t& E0, ent16_t& E1, ent16_t& E2, acc32_t& a0, acc32_t& a1, acc32_t& a2,
                                          acc32_t& a3, acc32_t& a4, acc32_t& a5, const uint64_t* base, uint32_t& off,
                                          uint32_t& n, uint32_t& ph, uint32_t mask, uint32_t lane_off) {
    asm volatile("s_cmp_eq_u32 %[ph], 0\n\t"
                 "s_cbranch_scc1 Le0_%=\n\t"
                 "s_cmp_eq_u32 %[ph], 1\n\t"
                 "s_cbranch_scc1 Le1_%=\n\t"
                 "s_cmp_eq_u32 %[ph], 2\n\t"
                 "s_cbranch_scc1 Le2_%=\n\t"
                 "s_cmp_eq_u32 %[ph], 3\n\t"
                 "s_cbranch_scc1 Le3_%=\n\t"
                 "s_cmp_eq_u32 %[ph], 4\n\t"
                 "s_cbranch_scc1 Le4_%=\n\t"
                 "s_branch Le5_%=\n\t"
                 SGCN_LDS_ENTRY(0, 36, 32)
                 SGCN_LDS_ENTRY(1, 52, 48)
                 SGCN_LDS_ENTRY(2, 68, 32)
                 SGCN_LDS_ENTRY(3, 36, 48)
                 SGCN_LDS_ENTRY(4, 52, 32)
                 SGCN_LDS_ENTRY(5, 68, 48)
                 //             P  EC  EN  EL  VC  VN  next
                 SGCN_LDS_GROUP(0, 36, 52, 68, 32, 48, 1)
                 SGCN_LDS_GROUP(1, 52, 68, 36, 48, 32, 2)
                 SGCN_LDS_GROUP(2, 68, 36, 52, 32, 48, 3)
                 SGCN_LDS_GROUP(3, 36, 52, 68, 48, 32, 4)
                 SGCN_LDS_GROUP(4, 52, 68, 36, 32, 48, 5)
                 SGCN_LDS_GROUP(5, 68, 36, 52, 48, 32, 0)
                 "Lx_%=:\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "+{s[36:51]}"(E0), "+{s[52:67]}"(E1), "+{s[68:83]}"(E2),
                   "+{v[64:95]}"(a0), "+{v[96:127]}"(a1), "+{v[128:159]}"(a2), "+{v[160:191]}"(a3),
                   "+{v[192:223]}"(a4), "+{v[224:255]}"(a5), [off] "+s"(off), [n] "+s"(n), [ph] "+s"(ph)
                 : [base] "s"(base), [mask] "v"(mask), [lane] "v"(lane_off)
                 : "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",
                   "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57",
                   "v58", "v59", "v60", "v61", "v62", "v63", "scc", "memory");
}

// 128-column slabs: a lane holds a float2 of every row of its wave; S ring slots (512-byte pieces) per half.
template <int S>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void lds_spmm_kernel(LdsArgs a) {
    constexpr int VW = 2, NW = 8, GE = 8;               // floats per lane, waves per tile, entries per group
    constexpr int RW = 192 / VW;                        // rows per wave: 192 accumulator registers
    constexpr int PIECE = 64 * VW * 4;                  // bytes of one staged piece
    constexpr int HALF = S * PIECE;
    constexpr int LPP = PIECE / 16;                     // lanes that fetch one piece (dwordx4 each)
    constexpr int PPI = 64 / LPP;                       // pieces per fill instruction
    constexpr int FPW = S / PPI / NW;                   // fill instructions per wave and chunk
    static_assert(S % (PPI * NW) == 0, "ring slots must divide evenly over the waves' fill instructions");
    typedef float VT __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(1024))) char ring[];       // two halves + one zero piece (the pads' operand)

    // workgroup b runs on XCD b % 8 (round-robin dispatch): every XCD gets a contiguous eighth of the tiles, ordered
    // (slab, tile) -- the 32 workgroups an XCD runs at a time are neighbouring tiles of one slab, i.e. rows of the same
    // few communities fetching the same pieces through one L2.  Pure placement.
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int lo = (int)((int64_t)a.ntiles * x / 8), hi = (int)((int64_t)a.ntiles * (x + 1) / 8);
    const int ntx = hi - lo;
    if (q >= ntx * a.nslab) return;
    const int slab = q / ntx, tile = lo + q % ntx;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fbase = slab * 64 * VW;
    const int f = fbase + lane * VW;                                     // my first column
    const bool act = f < a.d;
    const uint32_t lane_off = (uint32_t)lane * (VW * 4);
    const uint32_t mask = ~(uint32_t)(PIECE - 1);

    ent16_t E0 = {}, E1 = {}, E2 = {};
    acc32_t a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {};

    // the pads' operand: PIECE bytes of zeros behind the ring
    if (threadIdx.x < PIECE / 4) reinterpret_cast<float*>(ring + 2 * HALF)[threadIdx.x] = 0.f;

    const int c0 = __builtin_amdgcn_readfirstlane(a.tile_chunk_ptr[tile]);
    const int c1 = __builtin_amdgcn_readfirstlane(a.tile_chunk_ptr[tile + 1]);
    const char* Bb = reinterpret_cast<const char*>(a.B);
    const int64_t ldb_bytes = a.ldb * 4;
    // my share of a fill instruction: piece `lane / LPP` of the instruction, 16 bytes at `(lane % LPP) * 16`; a lane
    // past the row's pitch (last slab) re-reads the slab's first bytes instead of running off the row
    const int lp = lane / LPP;
    int64_t boff = (int64_t)fbase * 4 + (lane % LPP) * 16;
    if (boff + 16 > ldb_bytes) boff = (int64_t)fbase * 4;

    int32_t cols[FPW];
    auto load_cols = [&](int c) {
#pragma unroll
        for (int g = 0; g < FPW; g++) {
            int32_t col = a.chunk_cols[(int64_t)c * S + (wave * FPW + g) * PPI + lp];
            if (a.gidx) col = a.gidx[col];
            cols[g] = col;
        }
    };
    auto fill = [&](int half) {
#pragma unroll
        for (int g = 0; g < FPW; g++) {
            const char* src = Bb + (int64_t)cols[g] * ldb_bytes + boff;
            char* dst = ring + half * HALF + (wave * FPW + g) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // this wave's entries: contiguous over the tile's chunks ((tile, wave, chunk) order); ent_ptr through the scalar cache
    const int nc = c1 - c0;
    const __attribute__((address_space(4))) int64_t* eptr =
        (const __attribute__((address_space(4))) int64_t*)(a.ent_ptr) + ((int64_t)c0 * NW + (int64_t)wave * nc);
    if (nc > 0) {
        int64_t ecur = eptr[0];
        const uint64_t* ep = a.entries + ecur;             // (entry offsets from a.entries stay below 4 GiB: checked on the host)
        load_cols(c0);
        fill(0);
        if (nc > 1) load_cols(c0 + 1);
        // groups 0 and 1 of the wave's stream -> E0, E1 (the array ends in two pad groups: reading ahead is safe)
        asm volatile("s_load_dwordx16 s[36:51], %[ptr], 0x0\n\t"
                     "s_load_dwordx16 s[52:67], %[ptr], 0x40\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "+{s[36:51]}"(E0), "+{s[52:67]}"(E1) : [ptr] "s"(ep) : "memory");
        uint32_t eoff = (uint32_t)((ecur + 2 * GE) * 8);    // byte offset of the group the next s_load fetches
        uint32_t ph = 0;
        for (int k = 0; k < nc; k++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                // chunk k's pieces have landed; everybody is past chunk k - 1
            if (k + 1 < nc) {
                fill((k + 1) & 1);                          // the next chunk's pieces go in flight ...
                if (k + 2 < nc) load_cols(c0 + k + 2);      // ... and the column ids of the one after
            }
            const int64_t e1 = eptr[k + 1];
            uint32_t n = (uint32_t)((e1 - ecur) / GE);
            ecur = e1;
            if (n) lds_chunk(E0, E1, E2, a0, a1, a2, a3, a4, a5, a.entries, eoff, n, ph, mask, lane_off);
        }
    }

    // epilogue: row r of this wave sits at register offset r * VW
    const int32_t* rows = a.tile_rows + ((int64_t)tile * NW + wave) * RW;
    const int32_t* slots = a.tile_slots + ((int64_t)tile * NW + wave) * RW;
    const int left_cols = a.d - f;
    for (int r = 0; r < RW; r++) {
        const int row = rows[r];
        float x0, x1;
        asm volatile("s_set_gpr_idx_on %[i], 0x1\n\t"
                     "v_mov_b32 %[x0], v64\n\t"
                     "v_mov_b32 %[x1], v65\n\t"
                     "s_set_gpr_idx_off"
                     : [x0] "=&v"(x0), [x1] "=&v"(x1)
                     : [i] "s"(r * VW), "{v[64:95]}"(a0), "{v[96:127]}"(a1), "{v[128:159]}"(a2),
                       "{v[160:191]}"(a3), "{v[192:223]}"(a4), "{v[224:255]}"(a5));
        const VT accv = {x0, x1};
        if (row < 0 || !act) continue;
        const int slot = slots[r];
        if (slot >= 0) {
            float* w = a.ws + (int64_t)slot * a.ldw + f;
            if (left_cols >= VW) vstore<VW>(w, accv); else vstore_head<VW>(w, accv, left_cols);
        } else {
            float* out = a.C + (int64_t)row * a.ldc + f;
            const float rs = a.rscale ? a.rscale[row] : 1.0f;
            VT res = accv * rs;
            if (a.beta != 0.f) {
                if (left_cols >= VW) res += a.beta * vload<VW>(out);
                else for (int e = 0; e < left_cols; e++) res[e] += a.beta * out[e];
            }
            if (left_cols >= VW) vstore<VW>(out, res); else vstore_head<VW>(out, res, left_cols);
        }
    }
}

// fix-up of split rows: ordered slot sum + epilogue (sgcn_spmm_cs.hip has the same kernel for its plans)
__global__ __launch_bounds__(kBlock) void lds_fix_kernel(LdsArgs a, const sgcn_fix_t* fix, int64_t nfix) {
    typedef Vec<4>::type VT;
    const int nvec = (a.d + 3) / 4;
    const int nvblk = (nvec + kBlock - 1) / kBlock;
    const int64_t fi = blockIdx.x / nvblk;
    if (fi >= nfix) return;
    const sgcn_fix_t fx = fix[fi];
    const int vi = (int)(blockIdx.x % nvblk) * kBlock + threadIdx.x;
    if (vi >= nvec) return;
    const float* w = a.ws + (int64_t)fx.first_slot * a.ldw + (int64_t)vi * 4;
    VT acc = vzero<4>();
    for (int q = 0; q < fx.nslots; q++) acc += vload<4>(w + (int64_t)q * a.ldw);
    float* out = a.C + (int64_t)fx.row * a.ldc + (int64_t)vi * 4;
    VT res = acc * (a.rscale ? a.rscale[fx.row] : 1.0f);
    const int left = a.d - vi * 4;
    if (a.beta != 0.f) {
        if (left >= 4) res += a.beta * vload<4>(out);
        else for (int e = 0; e < left; e++) res[e] += a.beta * out[e];
    }
    if (left >= 4) vstore<4>(out, res); else vstore_head<4>(out, res, left);
}

}  // namespace sgcn

using namespace sgcn;

extern "C" int sgcn_spmm_lds_f32(const sgcn_ldsplan_t* plan, int32_t M, int32_t K, int32_t d,
                                 const float* B, int64_t ldb, const int32_t* gidx, const float* rscale,
                                 float* C, int64_t ldc, float beta, void* stream) {
    SGCN_REQUIRE(plan && M >= 0 && K >= 0 && d >= 0, "spmm_lds: bad argument");
    if (M == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(plan->NW == 8 && plan->VW == 2 && plan->RW == 96 && plan->U == 8 && plan->S == 128,
                 "spmm_lds: the plan must be built for 8 waves x 96 rows x float2, groups of 8 entries, 128 ring slots");
    SGCN_REQUIRE(plan->dev_tile_chunk_ptr && plan->dev_chunk_cols && plan->dev_ent_ptr && plan->dev_entries &&
                 plan->dev_tile_rows && plan->dev_tile_slots && B && C, "spmm_lds: null operand");
    SGCN_REQUIRE(pick_vw(d, {B, C, plan->dev_ws}, {ldb, ldc}) >= 4,
                 "spmm_lds: rows must be 16-byte aligned (pitch a multiple of 4 floats covering d)");
    hipStream_t st = (hipStream_t)stream;
    LdsArgs a{};
    a.tile_chunk_ptr = plan->dev_tile_chunk_ptr; a.chunk_cols = plan->dev_chunk_cols;
    a.ent_ptr = plan->dev_ent_ptr; a.entries = reinterpret_cast<const uint64_t*>(plan->dev_entries);
    a.tile_rows = plan->dev_tile_rows; a.tile_slots = plan->dev_tile_slots;
    a.B = B; a.ldb = ldb; a.gidx = gidx; a.rscale = rscale; a.C = C; a.ldc = ldc; a.beta = beta;
    a.d = d; a.ntiles = (int32_t)plan->ntiles;
    a.ws = plan->dev_ws; a.ldw = ((int64_t)d + 3) / 4 * 4;
    if (plan->nfix > 0)
        SGCN_REQUIRE(plan->dev_fix && plan->dev_ws && plan->ws_elems >= plan->nslots * a.ldw,
                     "spmm_lds: workspace missing or too small");
    const int slabw = 64 * plan->VW;
    a.nslab = (d + slabw - 1) / slabw;
    const int64_t per_xcd = (plan->ntiles + 7) / 8;
    const int64_t blocks = 8 * per_xcd * a.nslab;
    SGCN_REQUIRE(blocks < (1ll << 31), "spmm_lds: too many work items");
    constexpr int lds = 2 * 128 * 512 + 512;
    static bool once = false;                   // (an attribute of the function, not of a launch)
    if (!once) {
        SGCN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_spmm_kernel<128>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once = true;
    }
    hipLaunchKernelGGL((lds_spmm_kernel<128>), dim3((unsigned)blocks), dim3(512), lds, st, a);
    SGCN_HIP_TRY(hipGetLastError());
    if (plan->nfix > 0) {
        const int nvec = (d + 3) / 4;
        const int64_t nfblk = (int64_t)((nvec + kBlock - 1) / kBlock) * plan->nfix;
        SGCN_REQUIRE(nfblk < (1ll << 31), "spmm_lds: too many split rows");
        hipLaunchKernelGGL(lds_fix_kernel, dim3((unsigned)nfblk), dim3(kBlock), 0, st, a, plan->dev_fix, plan->nfix);
        SGCN_HIP_TRY(hipGetLastError());
    }
    return SGCN_OK;
}
