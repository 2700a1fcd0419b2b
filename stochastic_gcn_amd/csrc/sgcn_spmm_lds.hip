// LDS-staged column sweep for static graphs WITH locality (plan: include/sgcn.h sgcn_ldsplan_t).
//
// Why: the column-sweep kernels of sgcn_spmm_cs.hip move one d-float row of B from an L2 to the VGPRs PER NONZERO
// (nnz * d * 4 = 55.8 GB per S-Reddit product; the L2s deliver <= 30 TB/s of such gathers: >= 1.8 ms whatever the hit
// rate).  The only level below the L2 that can absorb that traffic is the LDS, and it can only if the rows that live on
// ONE compute unit share columns.  On a graph with communities they do: a tile of 768 rows of one community references
// every column of that community ~7 times.  So here
//   * one workgroup (8 wavefronts, 2 per SIMD at 256 VGPRs) owns a TILE of 8 x 96 virtual rows; a wave keeps the
//     96 x (64 lanes x float2) accumulators of its rows in 192 pinned VGPRs (v[64:255]);
//   * the tile's nonzeros are sorted by (community-ordered) column and cut into CHUNKS of at most 128 distinct columns;
//     a chunk's pieces of B (the 128-column slab of a B row: 512 bytes) are staged ONCE into one half of an LDS ring by
//     global_load_lds_dwordx4 (no staging registers) while the previous chunk is consumed from the other half;
//   * a nonzero is then: ds_read_b64 (a whole-wave, conflict-free read of the staged piece), s_set_gpr_idx_idx (selects
//     the row's accumulators: the gfx9 VGPR-indexing mode, as in sgcn_spmm_cs.hip) and ONE packed add / FMA;
//   * its plan entry -- a 32-bit word = LDS address of the piece | register offset of the row -- was loaded a chunk
//     ahead with the wave's other entries (64 per VGPR) and comes down by v_readlane.
// Values: when every planned nonzero of a row has the same value (any row-normalised adjacency: the reference's
// D^-1 A, gcn/utils.py:299-309) the plan FOLDS it into the row scale ("unit" plans): an entry is 4 bytes and a nonzero
// is a packed ADD.  Otherwise the values travel beside the words and a nonzero is a packed FMA.
// Columns a tile references fewer than `min_reuse` times are not worth a ring slot; the plan leaves those nonzeros to
// a residual CSR that the ordinary column sweep adds afterwards (ops.LdsSweepCSR).
//
// Determinism: an accumulator receives its nonzeros in plan order; split rows meet in the ordered fix-up; no atomics.
// Contract of sgcn_spmm_cs_f32 minus cscale and gidx.
// Every asm statement that enters the VGPR-indexing mode (s_set_gpr_idx_*: the index lives in M0) or points M0 at an
// LDS-DMA destination lists "m0" as a clobber, so that LLVM's M0-initialisation merging never carries a value across
// it.  M0 is a reserved (non-allocatable) register, which makes clang warn about the entry; the entry is intended.
#pragma clang diagnostic ignored "-Winline-asm"
#include "sgcn_dev.h"

namespace sgcn {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, class F>
__device__ __forceinline__ void lds_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lds_static_for<N, I + 1>(f);
    }
}

struct LdsArgs {
    const int32_t* tile_chunk_ptr;      // [ntiles + 1]
    const int32_t* chunk_hdr;           // [nchunks * NW * 16] per (chunk, wave): its S / NW column ids, entry groups, entry index
    const uint32_t* words;              // LDS byte address of the piece | register offset of the row
    const float* vals;                  // parallel to words; unused by unit plans
    const float* row_fold;              // unit plans: the value of the row's planned nonzeros (else null)
    const int32_t* tile_rows;           // [ntiles * NW * RW]
    const int32_t* tile_slots;
    const float* B; int64_t ldb;
    const float* rscale;
    float* C; int64_t ldc; float beta;
    int32_t d, nslab, ntiles;
    int32_t xcd_ptr[9];                 // tiles [xcd_ptr[x], xcd_ptr[x + 1]) run on XCD x
    float* ws; int64_t ldw;
    int32_t wide;                       // B spans 4 GiB or more: 64-bit row offsets in the fills
    unsigned long long* prof;           // experiments (sgcn_lds_profile_buffer): per workgroup and wave, cycles spent in
                                        // {prologue, fill issue, chunk statements, fill wait, barrier, epilogue, all} + chunk count
    int32_t dbg;                        // experiments (lds_dbg knob): bit 0 no piece fills, bit 1 no arithmetic, bit 2 no stores
};

typedef float acc32_t __attribute__((ext_vector_type(32)));

// ---- the inner loop: ONE asm statement per (wave, chunk) ---------------------------------------------------------------
// Everything that is in flight lives and dies inside the statement, on fixed registers (the compiler never sees -- and
// never copies -- a register a load has not written yet):
//   v[64:255]   the accumulators ("+v" operands): row r of the wave at v[64 + 2 r : 65 + 2 r]
//   v[32:47] / v[48:63]   the pieces of the even / odd group of 8 nonzeros
//   v[28:31]    LDS addresses
//   v[24:27]    the wave's plan entries of the current / next BLOCK of 64, one per lane: unit plans words in v26 (even
//               blocks) / v27 (odd); general plans {word, value} in v[24:25] / v[26:27]
//   s[36:43] / s[44:51]   the even / odd group's words; s[52:67] / s[68:83] their values, one per even-aligned pair
// The entries themselves were staged into the LDS (the wave's 1 KB of the entry ring) by the same global_load_lds
// burst that fetches a chunk's pieces, two chunks ahead; a block's 64 entries come out with one ds_read a block ahead and
// down to scalars by v_readlane with immediate lanes -- so the statement is the blocks unrolled, left at the first group
// boundary where the count runs out.  A GROUP of 8 nonzeros:
//     8 x (v_readlane [x 2], v_and_or_b32, ds_read_b64)   the NEXT group: address = (word & ~511) | lane * 8
//     s_waitcnt lgkmcnt(8)                                this group's pieces have landed (LDS returns in order)
//     s_set_gpr_idx_on, 8 x (s_set_gpr_idx_idx, v_pk_add_f32 | v_pk_fma_f32), s_set_gpr_idx_off
// i.e. per nonzero 3 (unit) or 4 VALU instructions, ~1.5 scalar, one LDS read, no vector memory.
#define SGCN_LDS_RD1_U(Q, K, EW, EX, LANE)                                                                    \
    "v_readlane_b32 s[36+8*" #Q "+" #K "], " #EW ", " #LANE "\n\t"
#define SGCN_LDS_RD1_V(Q, K, EW, EX, LANE)                                                                    \
    "v_readlane_b32 s[36+8*" #Q "+" #K "], " #EW ", " #LANE "\n\t"                                            \
    "v_readlane_b32 s[52+16*" #Q "+2*" #K "], " #EX ", " #LANE "\n\t"
#define SGCN_LDS_AD1(Q, K, T)                                                                                 \
    "v_and_or_b32 v[28+" #T "], s[36+8*" #Q "+" #K "], %[mask], %[lane]\n\t"
#define SGCN_LDS_DS1(Q, K, T)                                                                                 \
    "ds_read_b64 v[32+16*" #Q "+2*" #K ":32+16*" #Q "+2*" #K "+1], v[28+" #T "]\n\t"
// the reads of one group: entries at lanes L0 .. L0 + 7 of entry register EW (values: EX) -> parity Q
#define SGCN_LDS_READ8(M, Q, EW, EX, L0, L1, L2, L3, L4, L5, L6, L7)                                           \
    SGCN_LDS_RD1_##M(Q, 0, EW, EX, L0) SGCN_LDS_RD1_##M(Q, 1, EW, EX, L1)                                     \
    SGCN_LDS_RD1_##M(Q, 2, EW, EX, L2) SGCN_LDS_RD1_##M(Q, 3, EW, EX, L3)                                     \
    SGCN_LDS_AD1(Q, 0, 0) SGCN_LDS_AD1(Q, 1, 1) SGCN_LDS_AD1(Q, 2, 2) SGCN_LDS_AD1(Q, 3, 3)                   \
    SGCN_LDS_DS1(Q, 0, 0) SGCN_LDS_DS1(Q, 1, 1) SGCN_LDS_DS1(Q, 2, 2) SGCN_LDS_DS1(Q, 3, 3)                   \
    SGCN_LDS_RD1_##M(Q, 4, EW, EX, L4) SGCN_LDS_RD1_##M(Q, 5, EW, EX, L5)                                     \
    SGCN_LDS_RD1_##M(Q, 6, EW, EX, L6) SGCN_LDS_RD1_##M(Q, 7, EW, EX, L7)                                     \
    SGCN_LDS_AD1(Q, 4, 0) SGCN_LDS_AD1(Q, 5, 1) SGCN_LDS_AD1(Q, 6, 2) SGCN_LDS_AD1(Q, 7, 3)                   \
    SGCN_LDS_DS1(Q, 4, 0) SGCN_LDS_DS1(Q, 5, 1) SGCN_LDS_DS1(Q, 6, 2) SGCN_LDS_DS1(Q, 7, 3)
// unit plans: acc += piece (source 0 and destination indexed: mode 0x9); general: acc += value * piece (source 2 and
// destination indexed: mode 0xc; the value is the low word of an even-aligned scalar pair: op_sel_hi [0, 1, 1])
#define SGCN_LDS_MODE_U "0x9"
#define SGCN_LDS_MODE_V "0xc"
#define SGCN_LDS_OP1_U(Q, K)                                                                                  \
    "v_pk_add_f32 v[64:65], v[64:65], v[32+16*" #Q "+2*" #K ":32+16*" #Q "+2*" #K "+1]\n\t"
#define SGCN_LDS_OP1_V(Q, K)                                                                                  \
    "v_pk_fma_f32 v[64:65], s[52+16*" #Q "+2*" #K ":52+16*" #Q "+2*" #K "+1], v[32+16*" #Q "+2*" #K ":32+16*" #Q "+2*" #K "+1], v[64:65] op_sel_hi:[0,1,1]\n\t"
#define SGCN_LDS_APPLY8(M, Q, WAIT)                                                                           \
    "s_waitcnt lgkmcnt(" #WAIT ")\n\t"                                                                        \
    "s_setprio 2\n\t"                                                                                         \
    "s_set_gpr_idx_on s[36+8*" #Q "+0], " SGCN_LDS_MODE_##M "\n\t" SGCN_LDS_OP1_##M(Q, 0)                      \
    "s_set_gpr_idx_idx s[36+8*" #Q "+1]\n\t" SGCN_LDS_OP1_##M(Q, 1)                                            \
    "s_set_gpr_idx_idx s[36+8*" #Q "+2]\n\t" SGCN_LDS_OP1_##M(Q, 2)                                            \
    "s_set_gpr_idx_idx s[36+8*" #Q "+3]\n\t" SGCN_LDS_OP1_##M(Q, 3)                                            \
    "s_set_gpr_idx_idx s[36+8*" #Q "+4]\n\t" SGCN_LDS_OP1_##M(Q, 4)                                            \
    "s_set_gpr_idx_idx s[36+8*" #Q "+5]\n\t" SGCN_LDS_OP1_##M(Q, 5)                                            \
    "s_set_gpr_idx_idx s[36+8*" #Q "+6]\n\t" SGCN_LDS_OP1_##M(Q, 6)                                            \
    "s_set_gpr_idx_idx s[36+8*" #Q "+7]\n\t" SGCN_LDS_OP1_##M(Q, 7)                                            \
    "s_set_gpr_idx_off\n\t"                                                                                   \
    "s_setprio 0\n\t"                                                                                         \
    "s_sub_u32 %[n], %[n], 1\n\t"                                                                             \
    "s_cmp_eq_u32 %[n], 0\n\t"                                                                                \
    "s_cbranch_scc1 Lx_%=\n\t"
// Unit plans, the first 16 groups of a chunk: a group is 8 PAIR words while %[np] > 0 -- two rows of this wave on the same
// staged piece: row offsets in bits 0-7 and 24-31 of the word, ONE LDS read, two updates -- and 8 single words after that
// (the plan puts a chunk's pairs first).  On S-Reddit-SBM a wave's 138 entries of a chunk sit on 84 different pieces: pairs
// take the LDS reads of a chunk from 138 to 97 per wave, and the chunk statement is bound by the LDS bandwidth.
#define SGCN_LDS_PAIR1(Q, K)                                                                                  \
    SGCN_LDS_OP1_U(Q, K)                                                                                      \
    "s_lshr_b32 s[36+8*" #Q "+" #K "], s[36+8*" #Q "+" #K "], 24\n\t"                                          \
    "s_set_gpr_idx_idx s[36+8*" #Q "+" #K "]\n\t"                                                              \
    SGCN_LDS_OP1_U(Q, K)
#define SGCN_LDS_UBODY(Q)                                                                                     \
    "s_setprio 2\n\t"                                                                                         \
    "s_cmp_eq_u32 %[np], 0\n\t"                                                                               \
    "s_cbranch_scc1 1f\n\t"                                                                                   \
    "s_set_gpr_idx_on s[36+8*" #Q "+0], 0x9\n\t" SGCN_LDS_PAIR1(Q, 0)                                          \
    "s_set_gpr_idx_idx s[36+8*" #Q "+1]\n\t" SGCN_LDS_PAIR1(Q, 1)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+2]\n\t" SGCN_LDS_PAIR1(Q, 2)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+3]\n\t" SGCN_LDS_PAIR1(Q, 3)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+4]\n\t" SGCN_LDS_PAIR1(Q, 4)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+5]\n\t" SGCN_LDS_PAIR1(Q, 5)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+6]\n\t" SGCN_LDS_PAIR1(Q, 6)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+7]\n\t" SGCN_LDS_PAIR1(Q, 7)                                              \
    "s_set_gpr_idx_off\n\t"                                                                                   \
    "s_sub_u32 %[np], %[np], 1\n\t"                                                                           \
    "s_branch 2f\n\t"                                                                                         \
    "1:\n\t"                                                                                                  \
    "s_set_gpr_idx_on s[36+8*" #Q "+0], 0x9\n\t" SGCN_LDS_OP1_U(Q, 0)                                          \
    "s_set_gpr_idx_idx s[36+8*" #Q "+1]\n\t" SGCN_LDS_OP1_U(Q, 1)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+2]\n\t" SGCN_LDS_OP1_U(Q, 2)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+3]\n\t" SGCN_LDS_OP1_U(Q, 3)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+4]\n\t" SGCN_LDS_OP1_U(Q, 4)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+5]\n\t" SGCN_LDS_OP1_U(Q, 5)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+6]\n\t" SGCN_LDS_OP1_U(Q, 6)                                              \
    "s_set_gpr_idx_idx s[36+8*" #Q "+7]\n\t" SGCN_LDS_OP1_U(Q, 7)                                              \
    "s_set_gpr_idx_off\n\t"                                                                                   \
    "2:\n\t"                                                                                                  \
    "s_setprio 0\n\t"
#define SGCN_LDS_APPLY8P(Q, WAIT)                                                                             \
    "s_waitcnt lgkmcnt(" #WAIT ")\n\t"                                                                        \
    SGCN_LDS_UBODY(Q)                                                                                         \
    "s_sub_u32 %[n], %[n], 1\n\t"                                                                             \
    "s_cmp_eq_u32 %[n], 0\n\t"                                                                                \
    "s_cbranch_scc1 Lx_%=\n\t"
#define SGCN_LDS_BLOCKP(EW, NW, ENEXT)                                                                        \
    ENEXT                                                                                                     \
    SGCN_LDS_READ8(U, 1, EW, EW, 8, 9, 10, 11, 12, 13, 14, 15)         SGCN_LDS_APPLY8P(0, 8)                  \
    SGCN_LDS_READ8(U, 0, EW, EW, 16, 17, 18, 19, 20, 21, 22, 23)       SGCN_LDS_APPLY8P(1, 8)                  \
    SGCN_LDS_READ8(U, 1, EW, EW, 24, 25, 26, 27, 28, 29, 30, 31)       SGCN_LDS_APPLY8P(0, 8)                  \
    SGCN_LDS_READ8(U, 0, EW, EW, 32, 33, 34, 35, 36, 37, 38, 39)       SGCN_LDS_APPLY8P(1, 8)                  \
    SGCN_LDS_READ8(U, 1, EW, EW, 40, 41, 42, 43, 44, 45, 46, 47)       SGCN_LDS_APPLY8P(0, 8)                  \
    SGCN_LDS_READ8(U, 0, EW, EW, 48, 49, 50, 51, 52, 53, 54, 55)       SGCN_LDS_APPLY8P(1, 8)                  \
    SGCN_LDS_READ8(U, 1, EW, EW, 56, 57, 58, 59, 60, 61, 62, 63)       SGCN_LDS_APPLY8P(0, 8)                  \
    SGCN_LDS_READ8(U, 0, NW, NW, 0, 1, 2, 3, 4, 5, 6, 7)               SGCN_LDS_APPLY8P(1, 8)
// a block = 64 entries in EW (/ EX); ENEXT fetches the next block's entries into NW (/ NX) first -- it is the OLDEST LDS
// operation in flight when the first group waits (lgkmcnt(8) then covers it: LDS returns in order), so the block's last
// group can read ahead from NW's first lanes
#define SGCN_LDS_BLOCK(M, EW, EX, NW, NX, ENEXT)                                                              \
    ENEXT                                                                                                     \
    SGCN_LDS_READ8(M, 1, EW, EX, 8, 9, 10, 11, 12, 13, 14, 15)         SGCN_LDS_APPLY8(M, 0, 8)                \
    SGCN_LDS_READ8(M, 0, EW, EX, 16, 17, 18, 19, 20, 21, 22, 23)       SGCN_LDS_APPLY8(M, 1, 8)                \
    SGCN_LDS_READ8(M, 1, EW, EX, 24, 25, 26, 27, 28, 29, 30, 31)       SGCN_LDS_APPLY8(M, 0, 8)                \
    SGCN_LDS_READ8(M, 0, EW, EX, 32, 33, 34, 35, 36, 37, 38, 39)       SGCN_LDS_APPLY8(M, 1, 8)                \
    SGCN_LDS_READ8(M, 1, EW, EX, 40, 41, 42, 43, 44, 45, 46, 47)       SGCN_LDS_APPLY8(M, 0, 8)                \
    SGCN_LDS_READ8(M, 0, EW, EX, 48, 49, 50, 51, 52, 53, 54, 55)       SGCN_LDS_APPLY8(M, 1, 8)                \
    SGCN_LDS_READ8(M, 1, EW, EX, 56, 57, 58, 59, 60, 61, 62, 63)       SGCN_LDS_APPLY8(M, 0, 8)                \
    SGCN_LDS_READ8(M, 0, NW, NX, 0, 1, 2, 3, 4, 5, 6, 7)               SGCN_LDS_APPLY8(M, 1, 8)
// The NEXT chunk's piece requests ride inside the statement (two-part ring, 128 slots: 8 per wave): request J goes out
// behind group J of the first block -- outside the indexing mode, M0 being both its index register and the LDS-DMA's
// destination -- so that the vector L1 sees them spread over the chunk instead of as a burst in front of it (the burst was
// 18 % of the sweep: profiles/lds_phase_probe.py).  Source address: lanes 0-31 the piece of column word A, lanes 32-63 that
// of B (operand r<B> = the DIFFERENCE of the two row offsets, masked by hsel = ~0 in the upper half), + the lane's offset in the row.  A wave with fewer than 8 groups issues the rest on its
// way out (labels Lf<J>).
#define SGCN_LDS_FILL1(J, A, B)                                                                               \
    "v_and_b32 v28, %[r" #B "], %[hsel]\n\t"                                                                   \
    "v_add3_u32 v28, v28, %[r" #A "], %[boff]\n\t"                                                             \
    "s_add_u32 m0, %[m0b], " #J "*1024\n\t"                                                                    \
    "s_nop 0\n\t"                                                                                              \
    "global_load_lds_dwordx4 v28, %[bb]\n\t"
#define SGCN_LDS_APPLY8F(M, Q, WAIT, J, A, B, JN)                                                             \
    "s_waitcnt lgkmcnt(" #WAIT ")\n\t"                                                                        \
    SGCN_LDS_UBODY(Q)                                                                                         \
    SGCN_LDS_FILL1(J, A, B)                                                                                   \
    "s_sub_u32 %[n], %[n], 1\n\t"                                                                             \
    "s_cmp_eq_u32 %[n], 0\n\t"                                                                                \
    "s_cbranch_scc1 Lf" #JN "_%=\n\t"
#define SGCN_LDS_BLOCKF(M, EW, EX, NW, NX, ENEXT)                                                             \
    ENEXT                                                                                                     \
    SGCN_LDS_READ8(M, 1, EW, EX, 8, 9, 10, 11, 12, 13, 14, 15)         SGCN_LDS_APPLY8F(M, 0, 8, 0, 0, 1, 1)   \
    SGCN_LDS_READ8(M, 0, EW, EX, 16, 17, 18, 19, 20, 21, 22, 23)       SGCN_LDS_APPLY8F(M, 1, 8, 1, 2, 3, 2)   \
    SGCN_LDS_READ8(M, 1, EW, EX, 24, 25, 26, 27, 28, 29, 30, 31)       SGCN_LDS_APPLY8F(M, 0, 8, 2, 4, 5, 3)   \
    SGCN_LDS_READ8(M, 0, EW, EX, 32, 33, 34, 35, 36, 37, 38, 39)       SGCN_LDS_APPLY8F(M, 1, 8, 3, 6, 7, 4)   \
    SGCN_LDS_READ8(M, 1, EW, EX, 40, 41, 42, 43, 44, 45, 46, 47)       SGCN_LDS_APPLY8F(M, 0, 8, 4, 8, 9, 5)   \
    SGCN_LDS_READ8(M, 0, EW, EX, 48, 49, 50, 51, 52, 53, 54, 55)       SGCN_LDS_APPLY8F(M, 1, 8, 5, 10, 11, 6) \
    SGCN_LDS_READ8(M, 1, EW, EX, 56, 57, 58, 59, 60, 61, 62, 63)       SGCN_LDS_APPLY8F(M, 0, 8, 6, 12, 13, 7) \
    SGCN_LDS_READ8(M, 0, NW, NX, 0, 1, 2, 3, 4, 5, 6, 7)               SGCN_LDS_APPLY8F(M, 1, 8, 7, 14, 15, 8)
// the requests a short wave still owes, entered at Lf<J>; behind the last block (which jumps over them)
#define SGCN_LDS_FILL_TAIL                                                                                    \
    "s_branch Lx_%=\n\t"                                                                                      \
    "Lf1_%=:\n\t" SGCN_LDS_FILL1(1, 2, 3) "Lf2_%=:\n\t" SGCN_LDS_FILL1(2, 4, 5) "Lf3_%=:\n\t" SGCN_LDS_FILL1(3, 6, 7)    \
    "Lf4_%=:\n\t" SGCN_LDS_FILL1(4, 8, 9) "Lf5_%=:\n\t" SGCN_LDS_FILL1(5, 10, 11) "Lf6_%=:\n\t" SGCN_LDS_FILL1(6, 12, 13) \
    "Lf7_%=:\n\t" SGCN_LDS_FILL1(7, 14, 15) "Lf8_%=:\n\t"
#define SGCN_LDS_CLOBBER_V "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37",   \
        "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", \
        "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
#define SGCN_LDS_CLOBBER_SW "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49",  \
        "s50", "s51"
#define SGCN_LDS_CLOBBER_SV "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65",  \
        "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", \
        "s83"
#define SGCN_LDS_ACC "+{v[64:95]}"(a0), "+{v[96:127]}"(a1), "+{v[128:159]}"(a2), "+{v[160:191]}"(a3),          \
                     "+{v[192:223]}"(a4), "+{v[224:255]}"(a5)

// n (> 0) groups of 8 entries; `ea`: this lane's LDS address of its entry of the wave's first block (the wave's 1 KB of
// the entry ring: unit plans 4 blocks of 64 words; general plans 2 blocks of words, then their values at + 512)
template <bool UNIT>
__device__ __forceinline__ void lds_chunk(acc32_t& a0, acc32_t& a1, acc32_t& a2, acc32_t& a3, acc32_t& a4, acc32_t& a5,
                                          uint32_t n, uint32_t np, uint32_t ea, uint32_t mask, uint32_t lane_off) {
    if constexpr (UNIT) {
        asm volatile("ds_read_b32 v26, %[ea]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     SGCN_LDS_READ8(U, 0, v26, v26, 0, 1, 2, 3, 4, 5, 6, 7)
                     SGCN_LDS_BLOCKP(v26, v27, "ds_read_b32 v27, %[ea] offset:256\n\t")
                     SGCN_LDS_BLOCKP(v27, v26, "ds_read_b32 v26, %[ea] offset:512\n\t")
                     SGCN_LDS_BLOCK(U, v26, v26, v27, v27, "ds_read_b32 v27, %[ea] offset:768\n\t")
                     SGCN_LDS_BLOCK(U, v27, v27, v27, v27, "ds_read_b32 v26, %[ea]\n\t")
                     "Lx_%=:\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : SGCN_LDS_ACC, [n] "+s"(n), [np] "+s"(np)
                     : [ea] "v"(ea), [mask] "v"(mask), [lane] "v"(lane_off)
                     : SGCN_LDS_CLOBBER_V, SGCN_LDS_CLOBBER_SW, "scc", "m0", "memory");
    } else {
        asm volatile("ds_read2_b32 v[24:25], %[ea] offset1:128\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     SGCN_LDS_READ8(V, 0, v24, v25, 0, 1, 2, 3, 4, 5, 6, 7)
                     SGCN_LDS_BLOCK(V, v24, v25, v26, v27, "ds_read2_b32 v[26:27], %[ea] offset0:64 offset1:192\n\t")
                     SGCN_LDS_BLOCK(V, v26, v27, v26, v27, "ds_read2_b32 v[24:25], %[ea] offset1:128\n\t")
                     "Lx_%=:\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : SGCN_LDS_ACC, [n] "+s"(n)
                     : [ea] "v"(ea), [mask] "v"(mask), [lane] "v"(lane_off)
                     : SGCN_LDS_CLOBBER_V, SGCN_LDS_CLOBBER_SW, SGCN_LDS_CLOBBER_SV, "scc", "m0", "memory");
    }
}

// The same chunk with the next chunk's 8 piece requests inside (unit plans, two-part ring): r[2 J] = byte offset of the B row
// of the first piece of request J, r[2 J + 1] = the second piece's minus that, hsel = ~0 in lanes 32-63, boff = the lane's offset in a row, m0b = LDS address of the wave's first
// destination slot, bb = B.
struct LdsFill { uint32_t r[16]; uint32_t boff, hsel, m0b; const char* bb; };
__device__ __forceinline__ void lds_chunk_fill_u(acc32_t& a0, acc32_t& a1, acc32_t& a2, acc32_t& a3, acc32_t& a4, acc32_t& a5,
                                                 uint32_t n, uint32_t np, uint32_t ea, uint32_t mask, uint32_t lane_off, const LdsFill& F) {
    asm volatile("ds_read_b32 v26, %[ea]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 SGCN_LDS_READ8(U, 0, v26, v26, 0, 1, 2, 3, 4, 5, 6, 7)
                 SGCN_LDS_BLOCKF(U, v26, v26, v27, v27, "ds_read_b32 v27, %[ea] offset:256\n\t")
                 SGCN_LDS_BLOCKP(v27, v26, "ds_read_b32 v26, %[ea] offset:512\n\t")
                 SGCN_LDS_BLOCK(U, v26, v26, v27, v27, "ds_read_b32 v27, %[ea] offset:768\n\t")
                 SGCN_LDS_BLOCK(U, v27, v27, v27, v27, "ds_read_b32 v26, %[ea]\n\t")
                 SGCN_LDS_FILL_TAIL
                 "Lx_%=:\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : SGCN_LDS_ACC, [n] "+s"(n), [np] "+s"(np)
                 : [ea] "v"(ea), [mask] "v"(mask), [lane] "v"(lane_off), [boff] "v"(F.boff), [hsel] "v"(F.hsel), [m0b] "s"(F.m0b), [bb] "s"(F.bb),
                   [r0] "s"(F.r[0]), [r1] "s"(F.r[1]), [r2] "s"(F.r[2]), [r3] "s"(F.r[3]), [r4] "s"(F.r[4]), [r5] "s"(F.r[5]),
                   [r6] "s"(F.r[6]), [r7] "s"(F.r[7]), [r8] "s"(F.r[8]), [r9] "s"(F.r[9]), [r10] "s"(F.r[10]), [r11] "s"(F.r[11]),
                   [r12] "s"(F.r[12]), [r13] "s"(F.r[13]), [r14] "s"(F.r[14]), [r15] "s"(F.r[15])
                 : SGCN_LDS_CLOBBER_V, SGCN_LDS_CLOBBER_SW, "scc", "m0", "memory");
}

// 128-column slabs: a lane holds a float2 of every row of its wave.  The piece ring has NPART parts of S slots
// (512-byte pieces): chunk k is consumed from part k % NPART while the next NPART - 1 chunks are in flight into the
// others.  Three parts of 80 slots give a fill two chunk times to land; two parts of 128 slots give it one, but
// a third fewer chunks -- and a chunk costs ~1800 cycles of barrier, request issue and pipeline start whatever its size.
// The waves' entries travel the same way, 1 KB per wave and chunk, into the entry ring behind the pieces; their
// headers (128 bytes) into the header ring behind that.
template <int S, int NPART, bool UNIT, bool INL = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void lds_spmm_kernel(LdsArgs a) {
    constexpr int VW = 2, NW = 8;                       // floats per lane, waves per tile
    constexpr int RW = 192 / VW;                        // rows per wave: 192 accumulator registers
    constexpr int PIECE = 64 * VW * 4;                  // bytes of one staged piece
    constexpr int PART = S * PIECE;
    constexpr int LPP = PIECE / 16;                     // lanes that fetch one piece (dwordx4 each): 32
    constexpr int PPI = 64 / LPP;                       // pieces per fill instruction: 2
    constexpr int FPW = S / PPI / NW;                   // fill instructions per wave and chunk
    static_assert(PPI == 2 && S % (PPI * NW) == 0 && 2 * FPW <= 16, "ring slots: a multiple of 16, at most 128");
    static_assert(NPART == 2 || NPART == 3, "two or three ring parts");
    constexpr int LA = NPART - 1;                       // chunks requested ahead
    constexpr int ERING = NPART * PART + PIECE;         // the entry ring: NPART x NW x 1 KB behind the pieces and the zero piece
    constexpr int HRING = ERING + NPART * NW * 1024;    // the header ring: NPART x NW x 128 bytes
    typedef float VT __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(1024))) char ring[];

    // workgroup b runs on XCD b % 8 (round-robin dispatch): every XCD gets a contiguous range of the tiles (the plan cuts
    // the ranges to equal estimated time), ordered (slab, tile) -- the 32 workgroups an XCD runs at a time are neighbouring
    // tiles of one slab, i.e. rows of the same few communities fetching the same pieces through one L2.  Pure placement.
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int lo = a.xcd_ptr[xcd], hi = a.xcd_ptr[xcd + 1];
    const int ntx = hi - lo;
    if (q >= ntx * a.nslab) return;
    const int slab = q / ntx, tile = lo + q % ntx;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fbase = slab * 64 * VW;
    const int f = fbase + lane * VW;                                     // my first column
    const bool act = f < a.d;
    const uint32_t lane_off = (uint32_t)lane * (VW * 4);
    const uint32_t mask = 0x0003fe00u;                                   // the LDS address bits of an entry word (pieces of 512 bytes in a ring of < 256 KB)

    acc32_t a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {};

    // the pads' operand: PIECE bytes of zeros behind the piece ring
    if (threadIdx.x < PIECE / 4) reinterpret_cast<float*>(ring + NPART * PART)[threadIdx.x] = 0.f;

    const int c0 = __builtin_amdgcn_readfirstlane(a.tile_chunk_ptr[tile]);
    const int c1 = __builtin_amdgcn_readfirstlane(a.tile_chunk_ptr[tile + 1]);
    const char* Bb = reinterpret_cast<const char*>(a.B);
    const int64_t ldb_bytes = a.ldb * 4;
    // my share of a fill instruction: lanes 0-31 fetch its first piece, lanes 32-63 its second, 16 bytes each; a lane
    // past the row's pitch (last slab) re-reads the slab's first bytes instead of running off the row
    const bool second = lane >= LPP;
    uint32_t boff = (uint32_t)fbase * 4 + (lane % LPP) * 16;
    if ((int64_t)boff + 16 > ldb_bytes) boff = (uint32_t)fbase * 4;

    // A (chunk, wave) header -- the column ids of the wave's ring slots (16 words), its entry count in groups, the index
    // of its first entry: 32 words -- travels like everything else: global_load_lds into the wave's slot of the header
    // ring a chunk before it is needed, then ONE asm statement reads it (ds_read + wait + v_readlane) into scalars.  No
    // scalar memory instruction in the loop (the chunk statement opens with s_waitcnt lgkmcnt(0), which would wait for an
    // outstanding s_load too: measured with the ids on the scalar path, 0.7 us of exposed latency per chunk) and no
    // compiler-visible load either (the compiler's own s_waitcnt for one would be vmcnt(0): the fills just issued).
    struct Hdr { int32_t col[16]; uint32_t n, np; int64_t e; };
    auto hdr_fetch = [&](int c, int slot) {                 // chunk c's header -> my slot of the header ring
        if (lane < 32)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.chunk_hdr + ((int64_t)c * NW + wave) * 32 + lane),
                                             (__attribute__((address_space(3))) void*)(ring + HRING + (slot * NW + wave) * 128), 4, 0, 0);
    };
    auto hdr_get = [&](int slot) -> Hdr {                   // (after the s_waitcnt vmcnt that covers its fetch)
        Hdr h;
        uint32_t elo, ehi;
        const uint32_t addr = (uint32_t)(HRING + (slot * NW + wave) * 128) + (lane & 31) * 4;
        asm volatile("ds_read_b32 v28, %[addr]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_readlane_b32 %[c0], v28, 0\n\t"
                     "v_readlane_b32 %[c1], v28, 1\n\t"
                     "v_readlane_b32 %[c2], v28, 2\n\t"
                     "v_readlane_b32 %[c3], v28, 3\n\t"
                     "v_readlane_b32 %[c4], v28, 4\n\t"
                     "v_readlane_b32 %[c5], v28, 5\n\t"
                     "v_readlane_b32 %[c6], v28, 6\n\t"
                     "v_readlane_b32 %[c7], v28, 7\n\t"
                     "v_readlane_b32 %[c8], v28, 8\n\t"
                     "v_readlane_b32 %[c9], v28, 9\n\t"
                     "v_readlane_b32 %[c10], v28, 10\n\t"
                     "v_readlane_b32 %[c11], v28, 11\n\t"
                     "v_readlane_b32 %[c12], v28, 12\n\t"
                     "v_readlane_b32 %[c13], v28, 13\n\t"
                     "v_readlane_b32 %[c14], v28, 14\n\t"
                     "v_readlane_b32 %[c15], v28, 15\n\t"
                     "v_readlane_b32 %[n], v28, 16\n\t"
                     "v_readlane_b32 %[elo], v28, 17\n\t"
                     "v_readlane_b32 %[ehi], v28, 18\n\t"
                     "v_readlane_b32 %[np], v28, 19"
                     : [c0] "=s"(h.col[0]), [c1] "=s"(h.col[1]), [c2] "=s"(h.col[2]), [c3] "=s"(h.col[3]), [c4] "=s"(h.col[4]),
                       [c5] "=s"(h.col[5]), [c6] "=s"(h.col[6]), [c7] "=s"(h.col[7]), [c8] "=s"(h.col[8]), [c9] "=s"(h.col[9]),
                       [c10] "=s"(h.col[10]), [c11] "=s"(h.col[11]), [c12] "=s"(h.col[12]), [c13] "=s"(h.col[13]),
                       [c14] "=s"(h.col[14]), [c15] "=s"(h.col[15]), [n] "=s"(h.n), [elo] "=s"(elo), [ehi] "=s"(ehi), [np] "=s"(h.np)
                     : [addr] "v"(addr)
                     : "v28", "memory");
        h.e = (int64_t)(((uint64_t)ehi << 32) | elo);
        return h;
    };
    // chunk -> ring part p: its pieces (FPW instructions) and my entries of it (one instruction: 1 KB from the wave's
    // stream position -- unit plans 256 words; general plans 128 words by lanes 0-31 and their 128 values by lanes 32-63;
    // what lies behind the wave's count is fetched and never applied)
    auto fill = [&](int p, const Hdr& h, bool pieces = true) {
#pragma unroll
        for (int g = 0; g < FPW; g++) {
            if ((a.dbg & 1) || !pieces) break;              // (experiment: entries and headers only; or: the chunk statement issues them)
            // (row offsets as scalars, picked per half-wave: a select between the two column ids themselves is turned into
            // a per-lane vector index by the compiler -- a dozen compares per fill)
            const char* src;
            if (a.wide) {
                const int64_t r0 = (int64_t)h.col[2 * g] * ldb_bytes, r1 = (int64_t)h.col[2 * g + 1] * ldb_bytes;
                src = Bb + (second ? r1 : r0) + boff;
            } else {                                        // scalar base + 32-bit lane offset: 3 VALU per fill instead of 8
                const uint32_t r0 = (uint32_t)h.col[2 * g] * (uint32_t)ldb_bytes, r1 = (uint32_t)h.col[2 * g + 1] * (uint32_t)ldb_bytes;
                src = Bb + (size_t)((second ? r1 : r0) + boff);
            }
            char* dst = ring + p * PART + (wave * FPW + g) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        const char* esrc;
        if (UNIT) esrc = reinterpret_cast<const char*>(a.words + h.e) + lane * 16;
        else esrc = (second ? reinterpret_cast<const char*>(a.vals + h.e) : reinterpret_cast<const char*>(a.words + h.e)) + (lane % LPP) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)esrc,
                                         (__attribute__((address_space(3))) void*)(ring + ERING + (p * NW + wave) * 1024), 16, 0, 0);
    };
    constexpr int kPerChunk = FPW + 1;                      // the wave's requests of one chunk: pieces + entries

    const int nc = c1 - c0;
    // (experiments: the wave's cycle counts by phase; a.prof is null in production and the stamps fold away)
    unsigned long long t_fill = 0, t_comp = 0, t_wait = 0, t_bar = 0, t_all0 = 0, t_pro = 0, t_mark = 0;
    auto stamp = [&]() -> unsigned long long { return a.prof ? __builtin_readcyclecounter() : 0ull; };
    t_all0 = stamp();
    if (nc > 0) {
        uint32_t n0 = 0, n1 = 0, n2 = 0, p0 = 0, p1 = 0, p2 = 0;   // groups (and pair groups among them) of chunks k, k + 1, k + 2
        // headers of the first LA + 1 chunks, then the first LA chunks themselves
#pragma unroll
        for (int j = 0; j <= LA; j++)
            if (j < nc) hdr_fetch(c0 + j, j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            const Hdr h = hdr_get(0);
            fill(0, h);
            n0 = h.n; p0 = h.np;
        }
        if (LA == 2 && nc > 1) {
            const Hdr h = hdr_get(1);
            fill(1, h);
            n1 = h.n; p1 = h.np;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_pro = stamp() - t_all0;
        int part = 0;                                       // k % NPART: the ring part of chunk k
        for (int k = 0; k < nc; k++) {
            // chunk k has landed for everybody (each wave waited for its own fills below); everybody is past chunk k - 1.
            // A bare s_barrier: __syncthreads() would add a fence, i.e. wait for the fills just issued as well.
            const bool more = k + LA < nc;
            const int pn = part >= 1 ? part - 1 : NPART - 1;    // (k + LA) % NPART: the part chunk k - 1 left
            Hdr h;
            if (more) h = hdr_get(pn);                      // chunk k + LA's header came in during the last iteration (it is
            t_mark = stamp();                               // mine alone: read ahead of the barrier, its latency in the barrier's)
            asm volatile("s_barrier" ::: "memory");
            { const unsigned long long t = stamp(); t_bar += t - t_mark; t_mark = t; }
            // INL (unit plans on the two-part ring, B within 4 GB): the piece requests of chunk k + 1 go out from INSIDE chunk
            // k's statement -- the ONLY chunk statement of this instantiation (two different statements on the pinned
            // accumulators make the compiler copy them: 352 spills).  The last chunk of a tile requests row 0 eight times
            // into the part nobody reads any more; a wave without entries in this chunk issues the requests the old way.
            static_assert(!INL || (UNIT && NPART == 2 && FPW == 8), "inline requests: unit plans, two parts of 128 slots");
            const bool inl = INL && n0;
            if (more) {
                if (LA == 2) { n2 = h.n; p2 = h.np; } else { n1 = h.n; p1 = h.np; }
                if (k + LA + 1 < nc) hdr_fetch(c0 + k + LA + 1, part);  // the next header goes first (older than the fills: see the wait)
                fill(pn, h, !inl);                          // chunk k + LA goes in flight into the part chunk k - 1 left
            }
            { const unsigned long long t = stamp(); t_fill += t - t_mark; t_mark = t; }
            if constexpr (INL) {
                if (n0) {
                    LdsFill F;
#pragma unroll
                    for (int j = 0; j < 16; j += 2) {          // (even: the first piece's row offset; odd: second - first)
                        F.r[j] = more ? (uint32_t)h.col[j] * (uint32_t)ldb_bytes : 0u;
                        F.r[j + 1] = more ? (uint32_t)h.col[j + 1] * (uint32_t)ldb_bytes - F.r[j] : 0u;
                    }
                    F.boff = boff; F.hsel = second ? ~0u : 0u;
                    F.m0b = (uint32_t)(pn * PART + wave * FPW * 1024); F.bb = Bb;
                    lds_chunk_fill_u(a0, a1, a2, a3, a4, a5, n0, p0, (uint32_t)(ERING + (part * NW + wave) * 1024) + lane * 4, mask, lane_off, F);
                }
            } else {
                if (n0 && !(a.dbg & 2))
                    lds_chunk<UNIT>(a0, a1, a2, a3, a4, a5, n0, p0, (uint32_t)(ERING + (part * NW + wave) * 1024) + lane * 4, mask, lane_off);
            }
            { const unsigned long long t = stamp(); t_comp += t - t_mark; t_mark = t; }
            n0 = n1; n1 = n2; p0 = p1; p1 = p2;
            part = part == NPART - 1 ? 0 : part + 1;
            // chunk k + 1 has to be here before the barrier.  Three parts: it was requested an iteration ago -- everything
            // older than THIS iteration's fills (the header fetch was issued ahead of them: covered too).  Two parts: it is
            // this iteration's request.
            if (LA == 2 && more && !(a.dbg & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPerChunk) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            { const unsigned long long t = stamp(); t_wait += t - t_mark; }
        }
    }
    const unsigned long long t_epi0 = stamp();

    // epilogue: row r of this wave sits at register offset r * VW.  The wave's row ids / workspace slots come in with two
    // vector loads (lane r: row r) and down by v_readlane -- a scalar load per row would expose its latency 96 times.
    const int32_t* rows = a.tile_rows + ((int64_t)tile * NW + wave) * RW;
    const int32_t* slots = a.tile_slots + ((int64_t)tile * NW + wave) * RW;
    const int rows_lo = rows[lane], rows_hi = lane < RW - 64 ? rows[64 + lane] : -1;
    const int slots_lo = slots[lane], slots_hi = lane < RW - 64 ? slots[64 + lane] : -1;
    // the rows' factors the same way (unit plans: the folded value; the caller's row scale): lane r holds row r's
    float fold_lo = 1.0f, fold_hi = 1.0f, rs_lo = 1.0f, rs_hi = 1.0f;
    if (UNIT) {
        if (rows_lo >= 0) fold_lo = a.row_fold[rows_lo];
        if (rows_hi >= 0) fold_hi = a.row_fold[rows_hi];
    }
    if (a.rscale) {
        if (rows_lo >= 0) rs_lo = a.rscale[rows_lo];
        if (rows_hi >= 0) rs_hi = a.rscale[rows_hi];
    }
    auto lane_f = [](float lo, float hi, int r) -> float {
        return __int_as_float(r < 64 ? __builtin_amdgcn_readlane(__float_as_int(lo), r) : __builtin_amdgcn_readlane(__float_as_int(hi), r - 64));
    };
    const int left_cols = a.d - f;
    // The common case -- whole float2 columns (d even), nothing to add to (beta = 0) -- with every register of the wave's
    // rows named at compile time: per row two v_readlane (row id, factor), a scalar address, one or two packed multiplies
    // and ONE store with a scalar base.  (The general loop below costs ~530 cycles per row, 12 % of the sweep:
    // profiles/lds_phase_probe.py; this one ~10x less.)
    if (a.beta == 0.f && (a.d & 1) == 0 && !(a.dbg & 4)) {
        const uint64_t valid_lo = __ballot(rows_lo >= 0), valid_hi = __ballot(rows_hi >= 0);
        const uint64_t split_lo = __ballot(slots_lo >= 0), split_hi = __ballot(slots_hi >= 0);
        int64_t ldc_bytes = a.ldc * 4, ldw_bytes = a.ldw * 4;
        char* cbase = reinterpret_cast<char*>(a.C);
        char* wsbase = reinterpret_cast<char*>(a.ws);
        const uint32_t foff = (uint32_t)f * 4u;
        const bool has_rs = a.rscale != nullptr;
        // the bases as opaque scalars (otherwise re-read from the kernel arguments for every row: a scalar load and its wait
        // per row), and the rows' factors consumed once here: the wait for their loads then sits in front of the loop and
        // not, as an s_waitcnt vmcnt(0) that also waits for the previous row's STORE, inside every row
        asm volatile("" : "+s"(cbase), "+s"(wsbase), "+s"(ldc_bytes), "+s"(ldw_bytes));
        asm volatile("" ::"v"(fold_lo), "v"(fold_hi), "v"(rs_lo), "v"(rs_hi), "v"(rows_lo), "v"(rows_hi), "v"(slots_lo), "v"(slots_hi));
        if (act) {
            lds_static_for<RW>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                constexpr int k = r / 16, e = (r % 16) * VW;
                const bool valid = r < 64 ? ((valid_lo >> (r & 63)) & 1) : ((valid_hi >> ((r - 64) & 63)) & 1);
                if (!valid) return;
                const acc32_t& ak = k == 0 ? a0 : k == 1 ? a1 : k == 2 ? a2 : k == 3 ? a3 : k == 4 ? a4 : a5;
                VT res = {ak[e], ak[e + 1]};
                const bool split = r < 64 ? ((split_lo >> (r & 63)) & 1) : ((split_hi >> ((r - 64) & 63)) & 1);
                if (split) {                                    // a split row's piece: raw sum, scaled by the fix-up
                    const int slot = r < 64 ? __builtin_amdgcn_readlane(slots_lo, r & 63) : __builtin_amdgcn_readlane(slots_hi, (r - 64) & 63);
                    char* wsb = wsbase + (int64_t)slot * ldw_bytes;
                    *(__attribute__((address_space(1))) VT*)(uintptr_t)(wsb + foff) = res;      // (a GLOBAL store: the opaque base lost its address space)
                    return;
                }
                const int row = r < 64 ? __builtin_amdgcn_readlane(rows_lo, r & 63) : __builtin_amdgcn_readlane(rows_hi, (r - 64) & 63);
                if (UNIT) res = res * lane_f(fold_lo, fold_hi, r);
                if (has_rs) res = res * lane_f(rs_lo, rs_hi, r);
                char* ob = cbase + (int64_t)row * ldc_bytes;
                *(__attribute__((address_space(1))) VT*)(uintptr_t)(ob + foff) = res;
            });
        }
    } else
    for (int r = 0; r < RW; r++) {
        const int row = r < 64 ? __builtin_amdgcn_readlane(rows_lo, r) : __builtin_amdgcn_readlane(rows_hi, r - 64);
        float x0, x1;
        asm volatile("s_set_gpr_idx_on %[i], 0x1\n\t"
                     "v_mov_b32 %[x0], v64\n\t"
                     "v_mov_b32 %[x1], v65\n\t"
                     "s_set_gpr_idx_off"
                     : [x0] "=&v"(x0), [x1] "=&v"(x1)
                     : [i] "s"(r * VW), "{v[64:95]}"(a0), "{v[96:127]}"(a1), "{v[128:159]}"(a2),
                       "{v[160:191]}"(a3), "{v[192:223]}"(a4), "{v[224:255]}"(a5)
                     : "m0");
        const VT accv = {x0, x1};
        if (row < 0 || !act || (a.dbg & 4)) continue;
        const int slot = r < 64 ? __builtin_amdgcn_readlane(slots_lo, r) : __builtin_amdgcn_readlane(slots_hi, r - 64);
        if (slot >= 0) {                                    // a split row's piece: raw sum, scaled by the fix-up
            float* wsp = a.ws + (int64_t)slot * a.ldw + f;
            if (left_cols >= VW) vstore<VW>(wsp, accv); else vstore_head<VW>(wsp, accv, left_cols);
        } else {
            float* out = a.C + (int64_t)row * a.ldc + f;
            VT res = accv;
            if (UNIT) res = res * lane_f(fold_lo, fold_hi, r);
            if (a.rscale) res = res * lane_f(rs_lo, rs_hi, r);
            if (a.beta != 0.f) {
                if (left_cols >= VW) res += a.beta * vload<VW>(out);
                else for (int e = 0; e < left_cols; e++) res[e] += a.beta * out[e];
            }
            if (left_cols >= VW) vstore<VW>(out, res); else vstore_head<VW>(out, res, left_cols);
        }
    }
    if (a.prof && lane == 0) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        unsigned long long* o = a.prof + ((size_t)blockIdx.x * NW + wave) * 8;
        o[0] = t_pro; o[1] = t_fill; o[2] = t_comp; o[3] = t_wait; o[4] = t_bar; o[5] = t_end - t_epi0; o[6] = t_end - t_all0;
        o[7] = (unsigned long long)nc;
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
    VT res = acc;
    if (a.row_fold) res = res * a.row_fold[fx.row];
    if (a.rscale) res = res * a.rscale[fx.row];
    const int left = a.d - vi * 4;
    if (a.beta != 0.f) {
        if (left >= 4) res += a.beta * vload<4>(out);
        else for (int e = 0; e < left; e++) res[e] += a.beta * out[e];
    }
    if (left >= 4) vstore<4>(out, res); else vstore_head<4>(out, res, left);
}

}  // namespace sgcn

using namespace sgcn;

namespace { unsigned long long* g_lds_prof = nullptr; }
// experiments: a device buffer of 8 x 8 uint64 per workgroup of the next launches (NULL: off) -- profiles/lds_phase_probe.py
extern "C" int sgcn_lds_profile_buffer(void* dev_buf) { g_lds_prof = (unsigned long long*)dev_buf; return SGCN_OK; }

extern "C" int sgcn_spmm_lds_f32(const sgcn_ldsplan_t* plan, int32_t M, int32_t K, int32_t d,
                                 const float* B, int64_t ldb, const float* rscale,
                                 float* C, int64_t ldc, float beta, void* stream) {
    SGCN_REQUIRE(plan && M >= 0 && K >= 0 && d >= 0, "spmm_lds: bad argument");
    if (M == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(plan->NW == 8 && plan->VW == 2 && plan->RW == 96 && plan->U == 8 &&
                 ((plan->S == 80 && plan->nparts == 3) || (plan->S == 128 && plan->nparts == 2)),
                 "spmm_lds: the plan must be built for 8 waves x 96 rows x float2, groups of 8 entries, a ring of 3 x 80 or 2 x 128 slots");
    // (a plan without a single staged column -- an empty matrix, or everything in the residual -- has no chunk headers: its
    // tiles still write their rows, rscale (.) 0 + beta C)
    SGCN_REQUIRE(plan->dev_tile_chunk_ptr && (plan->dev_chunk_hdr || plan->nchunks == 0) && plan->dev_words &&
                 plan->dev_tile_rows && plan->dev_tile_slots && B && C, "spmm_lds: null operand");
    SGCN_REQUIRE(plan->unit ? plan->dev_row_fold != nullptr : plan->dev_vals != nullptr,
                 "spmm_lds: a unit plan needs its row values, a general plan its entry values");
    SGCN_REQUIRE(pick_vw(d, {B, C, plan->dev_ws}, {ldb, ldc}) >= 4,
                 "spmm_lds: rows must be 16-byte aligned (pitch a multiple of 4 floats covering d)");
    hipStream_t st = (hipStream_t)stream;
    LdsArgs a{};
    a.tile_chunk_ptr = plan->dev_tile_chunk_ptr; a.chunk_hdr = plan->dev_chunk_hdr;
    a.words = plan->dev_words; a.vals = plan->dev_vals;
    a.row_fold = plan->unit ? plan->dev_row_fold : nullptr;
    a.tile_rows = plan->dev_tile_rows; a.tile_slots = plan->dev_tile_slots;
    a.B = B; a.ldb = ldb; a.rscale = rscale; a.C = C; a.ldc = ldc; a.beta = beta;
    a.d = d; a.ntiles = (int32_t)plan->ntiles;
    a.ws = plan->dev_ws; a.ldw = ((int64_t)d + 3) / 4 * 4;
    if (plan->nfix > 0)
        SGCN_REQUIRE(plan->dev_fix && plan->dev_ws && plan->ws_elems >= plan->nslots * a.ldw,
                     "spmm_lds: workspace missing or too small");
    a.nslab = (d + 127) / 128;
    a.wide = ((int64_t)K * ldb * 4 + 4096 >= (1ll << 32)) ? 1 : 0;
    a.dbg = tune_get("lds_dbg");
    a.prof = g_lds_prof;
    int64_t per_xcd = 0;
    SGCN_REQUIRE(plan->xcd_tile_ptr[0] == 0 && plan->xcd_tile_ptr[8] == plan->ntiles, "spmm_lds: malformed XCD tile ranges");
    for (int x = 0; x < 8; x++) {
        SGCN_REQUIRE(plan->xcd_tile_ptr[x + 1] >= plan->xcd_tile_ptr[x], "spmm_lds: malformed XCD tile ranges");
        per_xcd = std::max<int64_t>(per_xcd, plan->xcd_tile_ptr[x + 1] - plan->xcd_tile_ptr[x]);
        a.xcd_ptr[x] = plan->xcd_tile_ptr[x];
    }
    a.xcd_ptr[8] = plan->xcd_tile_ptr[8];
    const int64_t blocks = 8 * per_xcd * a.nslab;
    SGCN_REQUIRE(blocks < (1ll << 31), "spmm_lds: too many work items");
    const int lds = plan->nparts * (plan->S * 512 + 8 * 1024 + 8 * 128) + 512;       // pieces + entries + headers, zero piece
    static bool once = false;                   // (an attribute of the function, not of a launch)
    if (!once) {
        constexpr int kMax = 160 * 1024;
        SGCN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_spmm_kernel<80, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax));
        SGCN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_spmm_kernel<80, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax));
        SGCN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_spmm_kernel<128, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax));
        SGCN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_spmm_kernel<128, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax));
        SGCN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_spmm_kernel<128, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax));
        once = true;
    }
    const dim3 grid((unsigned)blocks), block(512);
    if (plan->S == 80) {
        if (plan->unit) hipLaunchKernelGGL((lds_spmm_kernel<80, 3, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((lds_spmm_kernel<80, 3, false>), grid, block, lds, st, a);
    } else {
        // unit plans with B inside 4 GB: the variant whose chunk statement carries the next chunk's piece requests
        // (tune knob lds_dbg != 0: the plain variant, which the experiments' switches act on)
        if (plan->unit && !a.wide && !a.dbg) hipLaunchKernelGGL((lds_spmm_kernel<128, 2, true, true>), grid, block, lds, st, a);
        else if (plan->unit) hipLaunchKernelGGL((lds_spmm_kernel<128, 2, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((lds_spmm_kernel<128, 2, false>), grid, block, lds, st, a);
    }
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
