/* include/sgcn.h -- C-ABI of libsgcn.so: the MI355X-native (gfx950) drop-in for the training
 * hot path of thu-ml/stochastic_gcn (sampled sparse-adjacency x dense-feature SpMM fwd/bwd,
 * control-variate history gather/scatter, host neighbour sampler feeding a device CSR).
 *
 * The reference binds its native code through Cython (gcn/_scheduler.pyx, gcn/_history.pyx)
 * and runs its sparse arithmetic as TensorFlow ops (gcn/layers.py:31-37,304-311;
 * gcn/models.py:165).  This header is what a binding for the same seams targets instead:
 * plain pointers and sizes, no torch / numpy types.  Each entry cites the reference
 * interface it replaces as `file:line` relative to the reference tree.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; sgcn_last_error() returns a
 *     thread-local message.  Nothing throws across the boundary (the reference aborts on a
 *     C++ throw inside expand(): gcn/_scheduler.pyx:14-16 has no `except +`).
 *   - "dev" pointers are HBM (device) addresses, "host" pointers are CPU addresses.
 *   - values fp32, indices int32 (gcn/_scheduler.pyx:70-85, gcn/_history.pyx:27,38);
 *     leading dimensions (ld*) are in elements and 64-bit.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Kernels never
 *     allocate; all buffers are caller-owned.
 */
#ifndef SGCN_H
#define SGCN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGCN_OK 0
#define SGCN_ERR_INVALID (-1)
#define SGCN_ERR_HIP (-2)
#define SGCN_ERR_EMPTY_PROB (-3) /* gcn/mult.cpp:17-18 "Prob is empty" */
#define SGCN_ERR_NAN (-4)        /* gcn/scheduler.cpp:114-115 "nan" */

const char* sgcn_last_error(void);
/* ABI version of this header (bumped on any change of a signature or of a buffer contract): 16.
 *   v6  retired the kernels measured slower (two dense layers per launch, loss / LayerNorm backward in GEMM epilogues)
 *   v7  sampler core + packer threads (sgcn_prefetch_start: lag, n_packers), SGCN_AGG_PLAN_T
 *   v8  sgcn_step_fill, sgcn_copy_h2d_async (the launching thread's per-step work as foreign calls)
 *   v9  sgcn_softmax_ce_f32: with a prediction output, rowstat has a third plane (the rows' class indices)
 *   v10 sgcn_ldsplan_* / sgcn_spmm_lds_f32: the LDS-staged column sweep for graphs with locality
 *   v11 sgcn_csr_slice_indptr_dev; step ops MODE and CSR_SLICE .. GATHER_F32 (sparse-input stacks as step programs)
 *   v12 sgcn_csplan_t: dev_warp / warp_shift (the column sweep's clock in work coordinates); sgcn_csplang_*: host_warp
 *   v13 sgcn_coll_* (own RCCL communicator), sgcn_hist_pack / _apply, step ops ALLREDUCE_AVG .. HIST_APPLY
 *   v14 sgcn_csplan_build / sgcn_csbuild_* (the column-sweep plan in one parallel pass), sgcn_cs_warp_table,
 *       sgcn_csr_transpose_host, sgcn_host_threads; sgcn_coll_available / _retain / _abort / _async_error,
 *       sgcn_coll_destroy counts users
 *   v15 sgcn_coll_init_exchange / _has_exchange / sgcn_coll_allgather_x_i32 (a second communicator for the history exchange);
 *       step ops HIST_PACK .. HIST_APPLY: aux = 2 = the library's exchange stream, joined at the end of the run
 *   v16 packed minibatch: + the medg weights in the order of adj^T's nonzeros (descriptors behind the CSR table); step ops
 *       GEMM .. GATE (the --det_dropout stacks as step programs) */
int sgcn_abi_version(void);

/* ======================================================================================
 * Device kernels (HIP, gfx950)
 * ====================================================================================== */

/* ---- work plan for power-law rows -------------------------------------------------------
 * A gather SpMM is latency-bound per wavefront, so a 31k-nonzero hub row must not sit on one
 * wavefront.  A plan cuts every CSR row into segments of at most T nonzeros.  Rows that fit
 * in one segment are computed and stored directly; longer rows write per-segment partial
 * sums to a caller-owned workspace slot and a fix-up pass adds the slots IN ORDER (results
 * are deterministic, no float atomics).  The plan is pure host arithmetic on the row
 * pointer (the sampler has it on host anyway; a static graph builds it once). */
typedef struct { int32_t row, start, end, slot; } sgcn_seg_t;   /* slot < 0: direct row   */
typedef struct { int32_t row, first_slot, nslots; } sgcn_fix_t; /* one per split row      */
typedef struct {
    const sgcn_seg_t* dev_seg; int64_t nseg;
    const sgcn_fix_t* dev_fix; int64_t nfix;
    int64_t nslots;            /* total workspace slots used by split rows               */
    float* dev_ws;             /* [nslots x ldw] fp32, ldw = 4*ceil(d/4)                  */
    int64_t ws_elems;          /* capacity of dev_ws in floats (checked against nslots)   */
} sgcn_plan_t;
/* Count, then fill (host arrays sized by the counts).  T <= 0 selects the default (256). */
int sgcn_plan_count(const int32_t* host_rowptr, int32_t M, int32_t T,
                    int64_t* nseg, int64_t* nfix, int64_t* nslots);
int sgcn_plan_fill(const int32_t* host_rowptr, int32_t M, int32_t T,
                   sgcn_seg_t* host_seg, sgcn_fix_t* host_fix);

/* C[M x d] = rscale (.) ( A[M x K, CSR] * (cscale (.) B[g]) ) + beta * C
 *   B row used for column c is  B + (gidx ? gidx[c] : c) * ldb;  rscale[M], cscale[K]
 *   and gidx[K] are nullable; plan nullable (then one wavefront group per whole row).
 * Replaces  dot(x, y, sparse=True) = tf.sparse_tensor_dense_matmul   gcn/layers.py:31-37
 *           (K1 gcn/layers.py:253; K3/K4 :310-311; K5 :353-355; K11 gcn/utils.py:169-170,
 *           321-322), fused with tf.gather(history, field) gcn/layers.py:304-305 when gidx
 *           is given (K2+K7), and its TF-autodiff backward  dB = A^T dC  (K6) when called
 *           with the transposed CSR. */
int sgcn_spmm_csr_f32(const int32_t* dev_rowptr, const int32_t* dev_col, const float* dev_val,
                      int32_t M, int32_t K, int32_t d,
                      const float* dev_B, int64_t ldb, const int32_t* dev_gidx,
                      const float* dev_rscale, const float* dev_cscale,
                      float* dev_C, int64_t ldc, float beta,
                      const sgcn_plan_t* plan, void* stream);
/* The same with an addend in the epilogue: C[row, :] += add[row, :] for row < add_rows -- the
 * backward of a concat aggregator, dX = A^T (s (.) g_nbr) + [g_self ; 0], in one launch
 * (gcn/layers.py:311-319 autodiff). */
int sgcn_spmm_csr_add_f32(const int32_t* dev_rowptr, const int32_t* dev_col, const float* dev_val,
                          int32_t M, int32_t K, int32_t d, const float* dev_B, int64_t ldb,
                          const int32_t* dev_gidx, const float* dev_rscale, const float* dev_cscale,
                          float* dev_C, int64_t ldc, float beta, const sgcn_plan_t* plan,
                          const float* dev_add, int64_t ldadd, int32_t add_rows, void* stream);

/* ---- column-sweep plan for a STATIC graph (full-graph / PP products, K11) --------------------
 * A row-gather SpMM re-fetches a B row for every nonzero (measured on S-Reddit: 55.6 GB of
 * fabric traffic for 1.3 GB of algorithmic bytes, L2 hit rate 3 %, profiles/r01).  The column
 * sweep makes the gathers hit L2 instead: R virtual rows form a TILE whose R x (64 float4)
 * accumulators live in one wavefront's registers, the tile's nonzeros are merged and sorted by
 * COLUMN, and all wavefronts of a launch walk the column space front to back together, so at
 * any moment the chip touches one L2-sized window of B.  Rows longer than T are split into
 * virtual rows (strided in column order, so each spans the whole sweep) that meet again in the
 * ordered workspace fix-up (same deterministic scheme as sgcn_plan_t).  Built once on the host.
 * colrow[p] = col | (local_row << 28) for R = 16 (K < 2^28), << 27 for R = 32 (K < 2^27). */
typedef struct {
    int32_t R;                      /* virtual rows per tile: 16 (float4/lane) or 32 (float2) */
    int64_t ntiles;
    const int64_t* dev_tile_ptr;    /* [ntiles+1] offsets into colrow/val                   */
    const int32_t* dev_colrow;      /* [nnz]                                                */
    const float* dev_val;           /* [nnz]                                                */
    const int32_t* dev_tile_rows;   /* [ntiles*R] output row of each virtual row, -1 = pad  */
    const int32_t* dev_tile_slots;  /* [ntiles*R] workspace slot, -1 = store to C directly  */
    const sgcn_fix_t* dev_fix; int64_t nfix; int64_t nslots;
    float* dev_ws; int64_t ws_elems;
    int64_t round_tiles;            /* tiles per launch (0: derive from the device)         */
    const int64_t* host_tile_nnz_hint; /* HOST array: nonzeros of the heaviest tile of every
                                          launch, for clock pacing; nullable = unpaced          */
    int32_t pace_ns_per_nnz;        /* sweep clock: a launch lasts (heaviest tile nnz) x this many
                                       ns; found by timing a few values once per plan+width.
                                       0 = library default knob, < 0 = unpaced                  */
    int32_t G;                      /* lane groups per wavefront: 0 / 1 = one 16-row tile per wave on 304-column
                                       slabs; 2 = two 16-row bins per wave (lanes 0-31 / 32-63) on 128-column
                                       slabs, entries interleaved with pads (sgcn_csplang_*); 4 = four 16-row bins
                                       per wave (16 lanes each) on 64-column slabs: 64 rows per wave, so a 233 k-row
                                       graph is resident in ONE round -- the sweep of SPARSE matrices (the LDS plan's
                                       residual), instruction-bound on a full graph                               */
    int32_t xcd_map;                /* != 0: consecutive tiles of a launch go to the SAME XCD (the
                                       dispatcher deals workgroups round-robin over the 8 XCDs):
                                       with a grouped plan the tiles that share B rows share an L2.
                                       Placement only -- results do not depend on it.            */
    const uint32_t* dev_warp;       /* nullable, [((K - 1) >> warp_shift) + 1]: the sweep clock in WORK coordinates --
                                       entry b = the share of the matrix's nonzeros in columns < (b << warp_shift),
                                       scaled to [0, K).  A clock linear in the column id loses the lock-step where the
                                       nonzeros are not spread evenly over the ids (R-MAT); with the table a wave holds
                                       the POSITION of its column to the clock (G = 2 / 4 kernels; pace and slack keep
                                       their units).  Pacing only -- results do not depend on it.                     */
    int32_t warp_shift;
} sgcn_csplan_t;
/* row_group (nullable, [M], labels >= 0): tiles are formed inside groups, groups in label order
 * (locality-preserving reordering of a graph that has communities, e.g. from sgcn_reorder_lp);
 * NULL = one group = tiles balanced over all rows (the uniform-graph default). */
int sgcn_csplan_count(const int32_t* host_rowptr, int32_t M, int32_t R, int32_t T,
                      const int32_t* host_row_group, int64_t* ntiles, int64_t* nfix, int64_t* nslots);
int sgcn_csplan_fill(const int32_t* host_rowptr, const int32_t* host_col, const float* host_val,
                     int32_t M, int32_t R, int32_t T, const int32_t* host_row_group,
                     int64_t* host_tile_ptr, int32_t* host_colrow, float* host_valout,
                     int32_t* host_tile_rows, int32_t* host_tile_slots, sgcn_fix_t* host_fix);
/* Plan with `ngroups` = 2 or 4 lane groups per wavefront (sgcn_csplan_t.G, 16-row bins): tile_rows / tile_slots are
 * [ntiles * 16 * ngroups] (bin 0's 16 rows, then bin 1's, ...), colrow / val hold `nentries` interleaved entries (entry
 * ngroups*step + g is bin g's); pad entries carry the value bits 0x80000000 (-0.0f; real -0.0f values are stored as +0.0f) and are masked
 * off by the kernel.  align > 0: a bin advances only while at most `align` columns ahead of the other (keeps the two
 * halves of a wave inside one L2 window); every tile's entry count is padded to a multiple of 64 (the pipelined kernel
 * has no tail code); the tile count is rounded up to whole launches of round_tiles waves (0: no rounding).
 * align > 0 with four groups: every bin stays within `align` columns of the slowest.
 * host_warp (nullable, with warp_shift: the HOST copy of sgcn_csplan_t.dev_warp): `align` then counts sweep POSITIONS
 * (warp[col >> warp_shift]) instead of column ids -- the bins are aligned in the coordinates the clock runs in. */
int sgcn_csplang_count(const int32_t* host_rowptr, const int32_t* host_col, int32_t M, int32_t T,
                       int32_t round_tiles, int32_t align, int32_t ngroups, const uint32_t* host_warp, int32_t warp_shift,
                       int64_t* ntiles, int64_t* nentries, int64_t* nfix, int64_t* nslots);
int sgcn_csplang_fill(const int32_t* host_rowptr, const int32_t* host_col, const float* host_val, int32_t M,
                      int32_t T, int32_t round_tiles, int32_t align, int32_t ngroups, const uint32_t* host_warp,
                      int32_t warp_shift, int64_t* host_tile_ptr,
                      int32_t* host_colrow, float* host_valout, int32_t* host_tile_rows, int32_t* host_tile_slots,
                      sgcn_fix_t* host_fix);
/* The plan in ONE pass on `nthreads` host threads (0: every core the process may use -- affinity mask, cgroup quota,
 * SGCN_PLAN_THREADS --, at most 64).  The reference runs the product this plan serves once per matrix
 * (gcn/utils.py:321-322), so the plan's build time is part of the product's user-visible time.  ngroups = 1: the
 * sgcn_csplan_count / _fill plan (R rows per tile, optional row groups); 2 / 4: the sgcn_csplang_* plan (R ignored,
 * host_row_group must be NULL).  The plan does not depend on the thread count.  The builder owns the plan until
 * sgcn_csbuild_free; sgcn_csbuild_export copies it (in parallel) into arrays sized by sgcn_csbuild_sizes:
 * tile_ptr [ntiles + 1], colrow / val [nentries], tile_rows / tile_slots [ntiles * R * ngroups], fix [nfix]. */
typedef struct sgcn_csbuild sgcn_csbuild_t;
int sgcn_csplan_build(const int32_t* host_rowptr, const int32_t* host_col, const float* host_val, int32_t M,
                      int32_t ngroups, int32_t R, int32_t T, int32_t round_tiles, int32_t align,
                      const int32_t* host_row_group, const uint32_t* host_warp, int32_t warp_shift,
                      int32_t nthreads, sgcn_csbuild_t** out);
int sgcn_csbuild_sizes(const sgcn_csbuild_t* b, int64_t* ntiles, int64_t* nentries, int64_t* nfix, int64_t* nslots,
                       int32_t* T_used, int32_t* threads_used);
int sgcn_csbuild_export(const sgcn_csbuild_t* b, int64_t* host_tile_ptr, int32_t* host_colrow, float* host_valout,
                        int32_t* host_tile_rows, int32_t* host_tile_slots, sgcn_fix_t* host_fix);
void sgcn_csbuild_free(sgcn_csbuild_t* b);
/* sgcn_csplan_t.dev_warp's HOST copy from the matrix's column array (parallel histogram): table[b] = share of the
 * nonzeros in columns < (b << *shift), scaled to [0, K), for the smallest shift that gives <= max_buckets entries
 * (table must hold max_buckets).  mode 0 ("auto"): *nbuckets = 0 (no table: the clock stays linear in the column id)
 * when no bucket's share of the work in front of it is off its share of the ids by more than auto_dev; mode 1: always. */
int sgcn_cs_warp_table(const int32_t* host_col, int64_t nnz, int32_t K, int32_t max_buckets, int32_t mode,
                       double auto_dev, int32_t nthreads, uint32_t* host_table, int32_t* nbuckets, int32_t* shift);
/* A^T of an M x K CSR as a CSR on the host (stable parallel counting sort by column: rows ascend inside a column,
 * as in SciPy's csr -> csc pass): the backward product's plan is built from it (the autodiff of gcn/layers.py:31-37).
 * t_rowptr [K + 1], t_col / t_val [nnz]; host_val / host_t_val nullable together (pattern only). */
int sgcn_csr_transpose_host(const int32_t* host_rowptr, const int32_t* host_col, const float* host_val, int32_t M,
                            int32_t K, int32_t nthreads, int32_t* host_t_rowptr, int32_t* host_t_col, float* host_t_val);
/* The number of host threads the plan builders use by default (see sgcn_csplan_build). */
int32_t sgcn_host_threads(void);
/* Community labels of a square CSR pattern by seeded asynchronous label propagation (host, graph
 * only; new -- the reference has no reordering).  comm[n] in [0, *ncomm), numbered by decreasing
 * size; communities smaller than min_size share the last label.  max_iters <= 0: 12 sweeps. */
int sgcn_reorder_lp(const int32_t* host_rowptr, const int32_t* host_col, int32_t n, int32_t max_iters,
                    uint32_t seed, int32_t min_size, int32_t* host_comm, int32_t* ncomm);
/* Same contract as sgcn_spmm_csr_f32 (C = rscale (.) (A (cscale (.) B[g])) + beta C). */
int sgcn_spmm_cs_f32(const sgcn_csplan_t* plan, int32_t M, int32_t K, int32_t d,
                     const float* dev_B, int64_t ldb, const int32_t* dev_gidx,
                     const float* dev_rscale, const float* dev_cscale,
                     float* dev_C, int64_t ldc, float beta, void* stream);

/* The kernel variant and launch geometry sgcn_spmm_cs_f32 dispatches for (plan, d) under the
 * current knobs, as text (bench.py reports it as roofline.kernel). */
int sgcn_spmm_cs_variant(const sgcn_csplan_t* plan, int32_t d, char* buf, int32_t buflen);

/* ---- LDS-staged column sweep: static graphs WITH locality (sgcn_spmm_lds.hip) ----------------
 * The column sweep above moves one B row from an L2 to the registers per nonzero; on a graph
 * whose rows share columns inside a compute unit's tile (communities) that traffic can live in
 * the LDS instead.  A TILE is NW x RW virtual rows of ONE workgroup (wave w keeps the
 * accumulators of its RW rows in registers); the tile's nonzeros are sorted by sweep position
 * of their column and cut into CHUNKS of at most S distinct columns (and at most 256 -- general
 * plans 128 -- entries per wave); a chunk's pieces of B are staged once into an LDS ring and every nonzero of the chunk
 * reads its piece from there.  Nonzeros whose column the tile references fewer than `min_reuse`
 * times are left out of the plan and returned as a residual CSR (same M x K), to be added by
 * sgcn_spmm_cs_f32 / _csr_f32.
 *   words[i] = LDS byte address of the piece ((chunk index mod nparts) * S * P + slot * P, P = 256 * VW
 *              bytes; the zero piece at nparts * S * P for pads)  |  register offset of the row (local row * VW)
 *   vals[i]  = the nonzero's value (general plans).  UNIT plans: every nonzero of a row has the same
 *              value (row-normalised adjacency); it is kept once per row in row_fold[M] (1 for empty
 *              rows) and applied with the row scale -- no vals.
 *   entries are ordered (tile, wave, chunk); a wave's share of a chunk is padded to a multiple of
 *   U entries (pads: value 0 on the zero piece); 256 pad entries end the arrays (the kernel loads
 *   256 slots from the start of a wave's share, whatever its length).
 *   ent_ptr[(first chunk of the tile) * NW + w * (chunks of the tile) + k] = first entry of wave w in
 *   the tile's k-th chunk (one more element ends the array).
 *   chunk_hdr[(chunk * NW + w) * 32 ..]: what wave w needs to REQUEST the chunk, in one 128-byte read: the
 *   S / NW column ids of its ring slots (words 0-15), its entry count in groups of U (16), the 64-bit index
 *   of its first entry (17 low, 18 high).  (chunk_cols / ent_ptr hold the same facts chunk-wise.)
 * New -- the reference has no such kernel; the op is gcn/layers.py:31-37 (dot(x, y, sparse=True)). */
typedef struct {
    int32_t VW, NW, RW, S, U;       /* floats per lane (2: 128-column slabs), waves per tile (8), rows per wave
                                       (192 / VW = 96), slots per ring part (80 | 128), entries per group (8)    */
    int32_t nparts;                 /* ring parts: 3 (S = 80) or 2 (S = 128)                              */
    int32_t unit;                   /* != 0: values folded into dev_row_fold, dev_vals unused             */
    int32_t xcd_tile_ptr[9];        /* tiles [p[x], p[x + 1]) run on XCD x: contiguous ranges of equal estimated time    */
    int64_t ntiles, nchunks, nent;
    const int32_t* dev_tile_chunk_ptr; /* [ntiles + 1]                                            */
    const int32_t* dev_chunk_cols;  /* [nchunks * S] column of every slot (padded with a valid one) */
    const int32_t* dev_chunk_hdr;   /* [nchunks * NW * 32]                                         */
    const int64_t* dev_ent_ptr;     /* [nchunks * NW + 1]                                          */
    const uint32_t* dev_words;      /* [nent + 256]                                                */
    const float* dev_vals;          /* [nent + 256] (general plans)                                */
    const float* dev_row_fold;      /* [M] (unit plans)                                            */
    const int32_t* dev_tile_rows;   /* [ntiles * NW * RW] output row of each virtual row, -1 = pad  */
    const int32_t* dev_tile_slots;  /* [ntiles * NW * RW] workspace slot, -1 = store to C directly  */
    const sgcn_fix_t* dev_fix; int64_t nfix; int64_t nslots;
    float* dev_ws; int64_t ws_elems;
} sgcn_ldsplan_t;
/* Host-side builder (a handle: the sizes are known only once the plan exists).
 *   col_pos   nullable [K]: sweep position of every column (a permutation; community by community);
 *   row_group nullable [M]: tiles are formed inside groups, groups in label order;
 *   T         rows longer than T become strided virtual rows (<= 0: 2048);
 *   min_reuse a column is staged for a tile only if the tile references it at least this often (>= 1);
 *   mode      0: a unit plan when the values allow it, 1: always a general plan;
 *   ring_slots 128 (two ring parts; <= 0 selects it) or 80 (three parts: a fill has two chunk times to land).
 * sizes[19] = {ntiles, nchunks, nent, nfix, nslots, residual nnz, staged pieces (sum over chunks of distinct
 * columns), unit, xcd_tile_ptr[0..8], S, nparts}. */
typedef struct sgcn_ldsplan_host sgcn_ldsplan_host_t;
int sgcn_ldsplan_create(const int32_t* host_rowptr, const int32_t* host_col, const float* host_val,
                        int32_t M, int32_t K, const int32_t* host_col_pos, const int32_t* host_row_group,
                        int32_t VW, int32_t T, int32_t min_reuse, int32_t mode, int32_t ring_slots,
                        sgcn_ldsplan_host_t** out);
int sgcn_ldsplan_sizes(const sgcn_ldsplan_host_t* h, int64_t* sizes19);
int sgcn_ldsplan_export(const sgcn_ldsplan_host_t* h, int32_t* tile_chunk_ptr, int32_t* chunk_cols,
                        int32_t* chunk_hdr, int64_t* ent_ptr, uint32_t* words, float* vals, float* row_fold,
                        int32_t* tile_rows, int32_t* tile_slots, sgcn_fix_t* fix, int32_t* res_rowptr,
                        int32_t* res_col, float* res_val);
void sgcn_ldsplan_destroy(sgcn_ldsplan_host_t* h);
/* C[M x d] = rscale (.) (A_local * B) + beta * C   (A_local = the planned nonzeros; the residual is the caller's) */
int sgcn_spmm_lds_f32(const sgcn_ldsplan_t* plan, int32_t M, int32_t K, int32_t d,
                      const float* dev_B, int64_t ldb, const float* dev_rscale,
                      float* dev_C, int64_t ldc, float beta, void* stream);

/* Experiments: per-wave cycle counts by phase of the next sgcn_spmm_lds_f32 launches into dev_buf (8 x 8 uint64 per
 * workgroup; NULL switches it off).  profiles/lds_phase_probe.py reads it. */
int sgcn_lds_profile_buffer(void* dev_buf);

/* Runtime tuning knobs for experiments (bench.py --tune key=value); unknown key -> error.
 *   spmm_nv / spmm_unroll / spmm_slabmajor : row-gather kernel geometry (0 = auto)
 *   cs_round (tiles per launch), cs_unroll (4|8), cs_pace (ns per nonzero of the heaviest tile,
 *   0 = unpaced), cs_slack (columns), cs_noextra (no fifth fp32 accumulator plane), cs_g2_wide (G = 2 plans: force the
 *   64-bit row offsets), cs_last_pct (default 80: clock of a last pass that covers <= 3/4 of a slab, in percent of the
 *   plan's pace; 0 = off) : column-sweep kernels
 *   step_overlap (library default 1; the step program sets it from its flags, 0 unless --agg_overlap or a non-lean mode):
 *   in sgcn_step_run the AUX_* / VR_AGG_PRE ops (and, without --group_dw / --lean_sync, weight-gradient GEMMs and loss
 *   statistics) go to an auxiliary stream
 *   step_fuse (default 127): small dense products folded into the row pass next to them -- bit 0 the output layer's forward
 *   product into the loss kernel (the op in front of the loss), bit 1 its input gradient (the op behind it), bit 2 a narrow
 *   dense layer into the split-K reduce pass of the layer in front of it (all three in sgcn_step_run), bit 3 a dense layer's
 *   input gradient into its LayerNorm / ReLU backward pass (sgcn_dense_bwd_f32), bit 4 (with bit 0) the dense layer in front
 *   of the output layer as the pre-layer of the loss kernel's head, bit 5 the grouped weight-gradient launch's reductions in
 *   the optimizer's launch (when ADAM directly follows DW_FLUSH), bit 6 (with bit 3) the LayerNorm backward pass of a layer
 *   without an input gradient behind the row pass of the layer above it; 0: one launch per piece, the same bits
 *   gemm_min_steps: K-steps a split-K slice keeps at least (default 3) */
int sgcn_tune(const char* key, int64_t value);
int64_t sgcn_tune_get(const char* key);   /* current value, -1 for an unknown key */

/* Fused control-variate aggregator forward                    gcn/layers.py:298-319 (cvd)
 *                                                             gcn/layers.py:350-362 (cv)
 *   A  = sampled adjacency  [n1 x n0] CSR (a_rowptr,a_col,a_val)    placeholder 'adj'
 *   P  = full-neighbour adj [n1 x nf] CSR (f_rowptr,f_col,f_val)    placeholder 'fadj'
 *   Hbar [N x d] history (ldh), ifield[n0], ffield[nf], s[n1] = 'scales'
 * cvd != 0 (inputs h, mu [n0 x d]):
 *   mu_nbr = A (mu - Hbar[ifield]) + P Hbar[ffield]
 *   h_nbr  = (A (h - mu)) (.) s[:,None] + mu_nbr
 *   out_h[:, off:off+d] = h_nbr ; out_mu[:, off:off+d] = mu_nbr
 * cvd == 0 (single stream x = h, mu ignored, out_mu ignored):
 *   out_h[:, off:off+d] = A x - A Hbar[ifield] + P Hbar[ffield]
 * concat_self != 0 (normalization != 'gcn'): off = d and out[:, 0:d] = h[:n1] (resp. mu[:n1]);
 * else off = 0. */
int sgcn_vr_aggregate_f32(const int32_t* dev_a_rowptr, const int32_t* dev_a_col,
                          const float* dev_a_val, const int32_t* dev_f_rowptr,
                          const int32_t* dev_f_col, const float* dev_f_val,
                          int32_t n1, int32_t n0, int32_t nf, int32_t d,
                          const float* dev_h, const float* dev_mu, int64_t ldx,
                          const float* dev_Hbar, int64_t ldh,
                          const int32_t* dev_ifield, const int32_t* dev_ffield,
                          const float* dev_s,
                          float* dev_out_h, float* dev_out_mu, int64_t ldo,
                          int32_t cvd, int32_t concat_self,
                          const sgcn_plan_t* f_plan /* plan over f_rowptr, nullable */,
                          void* stream);

/* The same aggregate in two phases that are bit-identical to the fused call: _pre computes the
 * history-only part  accP[n1 x ldw] = P . Hbar[ffield]  (ldw = 4*ceil(d/4); it depends on nothing the
 * step computes, so the step program runs it beside the dense layers), _post the rest. */
int sgcn_vr_aggregate_pre_f32(const int32_t* dev_f_rowptr, const int32_t* dev_f_col, const float* dev_f_val,
                              int32_t n1, int32_t nf, int32_t d, const float* dev_Hbar, int64_t ldh,
                              const int32_t* dev_ffield, float* dev_accP, const sgcn_plan_t* f_plan,
                              void* stream);
int sgcn_vr_aggregate_post_f32(const int32_t* dev_a_rowptr, const int32_t* dev_a_col, const float* dev_a_val,
                               int32_t n1, int32_t n0, int32_t d, const float* dev_h, const float* dev_mu,
                               int64_t ldx, const float* dev_Hbar, int64_t ldh, const int32_t* dev_ifield,
                               const float* dev_s, float* dev_out_h, float* dev_out_mu, int64_t ldo,
                               int32_t cvd, int32_t concat_self, const float* dev_accP, void* stream);

/* out[i, 0:d] = in[r[i], 0:d]                  replaces history.dense_slice / c_dense_slice
 *                                              gcn/_history.pyx:53-62, gcn/history.cpp:74-88
 *                                              and tf.gather gcn/layers.py:304-305 */
int sgcn_gather_rows_f32(const float* dev_in, int64_t ldi, const int32_t* dev_r, int32_t n,
                         int32_t d, float* dev_out, int64_t ldo, void* stream);

/* H[r[i], 0:d] = src[i, 0:d]   (r unique)       replaces tf.scatter_update gcn/models.py:165
 * r[i] < 0 skips row i (padding of the fixed-capacity multi-GPU history exchange). */
int sgcn_scatter_rows_f32(float* dev_H, int64_t ldh, const int32_t* dev_r, int32_t n,
                          int32_t d, const float* dev_src, int64_t lds, void* stream);

/* ---- multi-GPU (the reference is single-process, gcn/train.py:130) ---------------------------------------
 * The library's own RCCL communicator (librccl.so by dlopen), so that the data-parallel step's collectives are stream-
 * ordered calls of sgcn_step_run (SGCN_OP_ALLREDUCE_AVG / _ALLGATHER_I32) rather than host-language calls between program
 * runs.  Rank 0 draws the id, the host side carries its 128 bytes to every rank (any channel: the job's process group),
 * every rank calls _init on its device.  One communicator per process, shared by reference count: _init = 1 user,
 * _retain adds one, _destroy drops one and destroys the communicator with the last.
 * sgcn_coll_available: a pure probe (library found, symbols bound, ncclGetVersion >= 2.10 -- ncclAvg); every rank calls it
 * before rank 0 draws the id (ncclGetUniqueId starts a listener that lives as long as the process).
 * sgcn_coll_abort: a rank that fails ahead of a collective aborts the communicator so that its peers' collectives return
 * an error instead of blocking for good (there is no watchdog on this communicator); sgcn_coll_async_error polls for
 * such an error (0 = healthy). */
int sgcn_coll_available(int32_t* nccl_version_code);
int sgcn_coll_unique_id(void* host_out128);
int sgcn_coll_init(const void* host_id128, int32_t world, int32_t rank);
int sgcn_coll_world(void);                      /* ranks of the communicator, 0 = none */
int sgcn_coll_retain(void);
int sgcn_coll_destroy(void);
int sgcn_coll_abort(void);
int sgcn_coll_async_error(void);
int sgcn_coll_allreduce_avg_f32(float* dev_buf, int64_t n, void* stream);                      /* in place, mean over ranks */
int sgcn_coll_allgather_i32(const int32_t* dev_send, int32_t* dev_recv, int64_t n, void* stream); /* recv = world x n */
/* A second communicator of the same ranks (ABI v15) for collectives issued on ANOTHER stream than the gradient all-reduce's:
 * one communicator used from two streams in turn makes RCCL order the streams itself.  Rank 0 draws a fresh id
 * (sgcn_coll_unique_id), every rank calls _init_exchange behind sgcn_coll_init; it lives and dies with the first
 * (_destroy / _abort).  sgcn_coll_allgather_x_i32 uses it (the first communicator when there is none). */
int sgcn_coll_init_exchange(const void* host_id128);
int sgcn_coll_has_exchange(void);               /* 1 = the exchange communicator exists */
int sgcn_coll_allgather_x_i32(const int32_t* dev_send, int32_t* dev_recv, int64_t n, void* stream);
/* History exchange (policy H-a): send = [cap ids | cap x d row bits], ids[n..cap) = -1;  apply = every rank's block of the
 * gathered buffer (world x cap x (d + 1) words) scattered into H in rank order (sgcn_scatter_rows_f32 per rank: ids < 0
 * skipped; a vertex two ranks updated keeps the higher rank's row on every replica).  dev_owner (nullable): one int32 per
 * history row, ZERO on entry and on return -- with it, more than two ranks are applied in two launches instead of `world`
 * (per vertex the highest (rank, slot) claims it by an integer atomicMax, then the owners copy): the same result. */
int sgcn_hist_pack_f32(const int32_t* dev_ids, int32_t n, const float* dev_rows, int64_t ld, int32_t d, int32_t cap,
                       int32_t* dev_send, void* stream);
int sgcn_hist_apply_f32(float* dev_H, int64_t ldh, const int32_t* dev_recv, int32_t world, int32_t cap, int32_t d,
                        int32_t* dev_owner, void* stream);

/* CSR row slice -> CSR of the n selected rows.          replaces history.slice / c_indptr +
 *   phase 1 (host): o_p[0..n] prefix over deg(r[i])      c_slice  gcn/_history.pyx:25-51,
 *   phase 2 (device): copy values + column ids           gcn/history.cpp:50-72
 * The reference returns COO [nnz,2]; rows are grouped in order so (o_p, o_col) is the same
 * matrix in CSR; o_row (nullable) additionally receives the COO row ids. */
int sgcn_csr_slice_indptr(int32_t n, const int32_t* host_r, const int32_t* host_a_p,
                          int32_t* host_o_p);
/* the same prefix pass on the device (one workgroup), for the compiled step: o_p[n + 1] from device row ids */
int sgcn_csr_slice_indptr_dev(int32_t n, const int32_t* dev_r, const int32_t* dev_a_p, int32_t* dev_o_p, void* stream);
int sgcn_csr_slice_f32(int32_t n, const int32_t* dev_r, const float* dev_a_d,
                       const int32_t* dev_a_i, const int32_t* dev_a_p,
                       const int32_t* dev_o_p, float* dev_o_d, int32_t* dev_o_col,
                       int32_t* dev_o_row, void* stream);
/* Transpose index of a row-sliced sparse block (stable device counting sort): for the n x ncols CSR
 * with column ids col[nnz] and COO row ids coo_row[nnz] (sgcn_csr_slice_f32's o_row), t_rowptr[ncols+1],
 * t_row[nnz] (row of every transposed entry, ascending inside a column) and t_src[nnz] (its position in
 * the source arrays, so values -- e.g. after sparse dropout -- follow with sgcn_gather_f32).  ws:
 * sgcn_csr_transpose_ws_ints(ncols, nnz) int32.  New: TF derives dW = X^T g inside its autodiff of
 * dot(x, W, sparse=True), gcn/layers.py:125,401-402. */
int64_t sgcn_csr_transpose_ws_ints(int32_t ncols, int64_t nnz);
int sgcn_csr_transpose_index(int32_t ncols, int64_t nnz, const int32_t* dev_col, const int32_t* dev_coo_row,
                             int32_t* dev_t_rowptr, int32_t* dev_t_row, int32_t* dev_t_src, int32_t* dev_ws,
                             void* stream);
/* out[r, :] = s[r] * x[r, :]  (16-byte aligned rows).  For products whose matrix carries one value per COLUMN -- the transpose
 * of a row-normalised adjacency, the backward of gcn/layers.py:31-37 on D^-1 A: M . B = pattern(M) . (s (.) B), which lets
 * the LDS sweep use its unit plan (ops.LdsSweepCSR). */
int sgcn_scale_rows_f32(const float* dev_x, int64_t ldx, const float* dev_s, int32_t n, int32_t d, float* dev_out, int64_t ldo,
                        void* stream);
/* out[i] = src[idx[i]] */
int sgcn_gather_f32(const float* dev_src, const int32_t* dev_idx, int64_t n, float* dev_out, void* stream);

/* ---- fused row-wise kernels of the dense part of the step --------
 * y = act(LN(x) * scale + offset), eps as MyLayerNorm2 (1e-9)     gcn/layers.py:95-97,134-137
 *   offset/scale NULL -> no normalisation (y = act(x)); relu != 0 -> ReLU.
 *   xhat [n x d] and rstd [n] are kept for the backward. */
int sgcn_ln_act_fwd_f32(const float* dev_x, int64_t ldx, const float* dev_offset,
                        const float* dev_scale, int32_t n, int32_t d, float eps, int32_t relu,
                        float* dev_y, int64_t ldy, float* dev_xhat, float* dev_rstd, void* stream);
/* Backward of the above: dx, and doffset[d] += / dscale[d] += column sums (two-stage, fixed
 * order).  ws: sgcn_ln_act_bwd_ws_floats(n, d) floats of scratch.  scale NULL -> no norm. */
int64_t sgcn_ln_act_bwd_ws_floats(int32_t n, int32_t d);
int sgcn_ln_act_bwd_f32(const float* dev_dy, int64_t lddy, const float* dev_y, int64_t ldy,
                        const float* dev_xhat, const float* dev_rstd, const float* dev_scale,
                        int32_t n, int32_t d, int32_t relu, float* dev_dx, int64_t lddx,
                        float* dev_doffset, float* dev_dscale, float* dev_ws, void* stream);
/* Dropout with a counter-based hash instead of a stateful generator (tf.nn.dropout gcn/layers.py:396,
 * 425-433: keep with probability `keep`, scale by 1/keep).  Element (row, col) of an [n x width]
 * activation is kept iff  fmix32((row * width + col) * 0x9E3779B1 + key) < keep * 2^32  (fmix32 = the
 * murmur3 finaliser); `key` is derived on the host from (seed, layer index, step).  Being a pure
 * function of the element index, the mask is never stored: the forward GEMM applies it while loading
 * its operand, the weight-gradient GEMM recomputes it, the input-gradient GEMM applies it in its
 * epilogue -- and the CPU oracle replays exactly the same masks.  rows: only rows < rows are dropped
 * (the clean CVD stream stacked below the dropout stream is not); rows < 0 = all rows. */
typedef struct {
    uint32_t key;
    float keep;
    int32_t rows;
    int32_t width;   /* columns of the activation the mask is defined on */
} sgcn_dropout_t;
/* out[i, 0:d] = x[i, 0:d] * mask / keep   (the unfused form; also its own backward) */
int sgcn_dropout_f32(const float* dev_x, int64_t ldx, int32_t n, int32_t d, const sgcn_dropout_t* drop,
                     float* dev_out, int64_t ldo, void* stream);

/* fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32, exact fp32) for the dense weight layers:
 *   C[M x N] = op(A) . op(B) (+ C when accumulate != 0);  trans_a: A is stored [K x M];
 *   trans_b: B is stored [N x K].        replaces tf.matmul gcn/layers.py:36 and its autodiff
 * dev_ws (nullable): sgcn_gemm_ws_floats(M, N, K) floats of scratch enable deterministic split-K --
 * a weight-gradient GEMM has a 128 x 128 output and K ~ 1,000, i.e. four output tiles; its K
 * range is cut across workgroups and the partial tiles are summed in a fixed order.
 * drop_a (nullable): the STORED A matrix is a dropout input -- its elements are masked and scaled
 *   while they are loaded (forward: A = x; weight gradient, trans_a: A = x stored [n x fin]);
 * drop_c (nullable): mask and scale the OUTPUT (input gradient dx = (g . W^T) * mask / keep). */
int64_t sgcn_gemm_ws_floats(int32_t M, int32_t N, int32_t K);
int sgcn_gemm_f32(int32_t trans_a, int32_t trans_b, int32_t M, int32_t N, int32_t K,
                  const float* dev_A, int64_t lda, const float* dev_B, int64_t ldb, float* dev_C,
                  int64_t ldc, int32_t accumulate, float* dev_ws, const sgcn_dropout_t* drop_a,
                  const sgcn_dropout_t* drop_c, void* stream);
/* One launch per dense layer: Y = act(LN(X . W) * scale + offset)   (N <= 128 when LN / ReLU is
 * requested; offset/scale NULL -> no LayerNorm).   gcn/layers.py:120-138, :396-411
 * dev_X2 (nullable): rows >= split of the operand come from X2 (row - split): the CVD layer runs its
 * dropout stream and its clean stream, which share W and the LayerNorm parameters, as ONE stacked
 * GEMM without materialising [dropout(x) ; mu].  drop (nullable): dropout on the operand rows. */
int sgcn_dense_fwd_f32(int32_t M, int32_t N, int32_t K, const float* dev_X, int64_t ldx,
                       const float* dev_X2, int64_t ldx2, int32_t split,
                       const float* dev_W, int64_t ldw, const float* dev_offset,
                       const float* dev_scale, float eps, int32_t relu, float* dev_Y, int64_t ldy,
                       float* dev_xhat, float* dev_rstd, const sgcn_dropout_t* drop,
                       float* dev_ws /* nullable: sgcn_gemm_ws_floats(M, N, K) floats -> split-K, the
                                        epilogue then runs in the reduction */,
                       const int32_t* dev_gidx /* nullable: operand row r is X[gidx[r]] -- the gather of
                                                  the minibatch's feature rows (history.dense_slice,
                                                  gcn/vrgcn.py:43-45) happens inside the GEMM */,
                       const int32_t* dev_gidx2 /* the same for X2 */, void* stream);
/* The backward of one dense layer in one call: g = LN/ReLU-backward(dy) (skipped when scale == NULL
 * and relu == 0), dW[K x N] += dropout(x)^T . g, dx[n x K] = (g . W^T) * mask (dx nullable).
 * g_tmp: n * N floats; ws: sgcn_ln_act_bwd_ws_floats(n, N) (rounded up to 4) + max(
 * sgcn_gemm_ws_floats(K, N, n), sgcn_gemm_ws_floats(n, K, N)) floats.       autodiff of gcn/layers.py:120-138,396-411 */
int sgcn_dense_bwd_f32(int32_t n, int32_t N, int32_t K, const float* dev_dy, int64_t lddy,
                       const float* dev_y, int64_t ldy, const float* dev_xhat, const float* dev_rstd,
                       const float* dev_scale, int32_t relu, const float* dev_x, int64_t ldx,
                       const float* dev_W, int64_t ldw, float* dev_dW, int64_t lddw,
                       float* dev_doffset, float* dev_dscale, float* dev_dx, int64_t lddx,
                       const sgcn_dropout_t* drop, float* dev_g_tmp, float* dev_ws,
                       const int32_t* dev_gidx /* nullable: layer input row r is x[gidx[r]] */, void* stream);
/* Softmax cross-entropy over n rows: stats[4] = {sum_i CE_i, #rows whose arg-max matches the label
 * arg-max, mean CE (the loss), accuracy}; dlogits (nullable) = (softmax * sum(labels) - labels) / n;
 * pred (nullable) = softmax; rowstat: 2*n floats of scratch (per-row CE and hit flag, summed
 * in a fixed order) -- 3*n when pred is given (ABI v9): rowstat[2n + i] = argmax(pred[i]) + 4096 * argmax(labels[i]),
 * the two class indices gcn/utils.py:521-529 takes per row for the F1 scores (first maximum, as np.argmax; exact in
 * fp32 for c <= 4096), so that an evaluation sweep brings 4 bytes per row to the host instead of 2 c floats.
 *                                                              gcn/models.py:68-94,198-202 */
int sgcn_softmax_ce_f32(const float* dev_logits, int64_t ldz, const float* dev_labels, int64_t ldl,
                        int32_t n, int32_t c, float* dev_dlogits, int64_t lddz, float* dev_pred,
                        int64_t ldp, float* dev_stats, float* dev_rowstat, void* stream);
/* Multitask (ppi) loss: mean sigmoid cross-entropy with logits over all n*c elements, element
 * accuracy, pred = sigmoid(z), dlogits = (pred - y)/(n*c).  stats as sgcn_softmax_ce_f32, rowstat: 2*n floats.
 * Replaces tf.nn.sigmoid_cross_entropy_with_logits + reduce_mean   gcn/models.py:77-79,86-90,198-200 */
int sgcn_sigmoid_ce_f32(const float* dev_logits, int64_t ldz, const float* dev_labels, int64_t ldl,
                        int32_t n, int32_t c, float* dev_dlogits, int64_t lddz, float* dev_pred, int64_t ldp,
                        float* dev_stats, float* dev_rowstat, void* stream);
/* Weight decay on the flat-buffer range [lo, hi) (the vars of the first parametrised layer):
 * grad[i] += wd * theta[i] (grad nullable) and loss[0] += 0.5 * wd * sum theta[i]^2 (loss nullable),
 * deterministic.  Replaces FLAGS.weight_decay * tf.nn.l2_loss(var) and its gradient  gcn/models.py:68-75 */
int sgcn_l2_penalty_f32(const float* dev_theta, int64_t lo, int64_t hi, float wd, float* dev_grad,
                        float* dev_loss, void* stream);
/* tf.train.AdamOptimizer step on flat buffers: m,v updated in place,
 * theta -= lr_t * m / (sqrt(v) + eps)  with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the
 * caller.                                                       gcn/models.py:50-51 */
int sgcn_adam_f32(float* dev_theta, const float* dev_grad, float* dev_m, float* dev_v, int64_t n,
                  float lr_t, float beta1, float beta2, float eps, void* stream);

/* ---- deterministic dropout (--det_dropout; gcn/layers.py:141-202, 236-248, 320-349, 425-428): the element-wise and
 * row-wise pieces of the moment-propagation variant, forward and backward (autodiff of the reference's formulas).  The
 * variant's matrix products are sgcn_gemm_f32 / sgcn_spmm_csr_f32.  All arrays dense with pitch d unless a pitch is given. */
/* var_out = var / keep + (1 / keep - 1) mu^2 (var NULL: 0)                                   layers.py:168-176 */
int sgcn_det_pre_f32(const float* dev_mu, const float* dev_var, int64_t n, float keep, float* dev_var_out, void* stream);
/* d_mu += g 2 (1 / keep - 1) mu;  d_var = g / keep (d_var nullable) */
int sgcn_det_pre_bwd_f32(const float* dev_mu, const float* dev_g, int64_t n, float keep, float* dev_d_mu, float* dev_d_var,
                         void* stream);
int sgcn_square_f32(const float* dev_x, int64_t n, float c, float* dev_y, void* stream);                 /* y = c x^2 */
int sgcn_addmul_f32(float* dev_acc, const float* dev_a, const float* dev_b, int64_t n, float c, void* stream); /* acc += c a b */
/* var2 = var1 scale^2 / variance, variance of the mean stream's row recovered from its rstd = rsqrt(variance + eps)  :184-188 */
int sgcn_det_lnvar_fwd_f32(const float* dev_var1, const float* dev_rstd, const float* dev_scale, int32_t n, int32_t d, float eps,
                           float* dev_var2, void* stream);
/* d_var1 = g scale^2 / V;  d_mu1 += dV 2 (mu1 - mean) / d;  dscale += column sums of g var1 2 scale / V.  tmp: n * d floats */
int sgcn_det_lnvar_bwd_f32(const float* dev_g, const float* dev_var1, const float* dev_xhat, const float* dev_rstd,
                           const float* dev_scale, int32_t n, int32_t d, float eps, float* dev_d_var1, float* dev_d_mu1,
                           float* dev_dscale, float* dev_tmp, void* stream);
/* ReLU by moment matching of a Gaussian (mu, var) -> (mean, variance) of max(x, 0)                        :190-202 */
int sgcn_det_relu_fwd_f32(const float* dev_mu, const float* dev_var, int64_t n, float* dev_mu_out, float* dev_var_out, void* stream);
int sgcn_det_relu_bwd_f32(const float* dev_mu, const float* dev_var, const float* dev_g_mu, const float* dev_g_var, int64_t n,
                          float* dev_d_mu, float* dev_d_var, void* stream);
/* x = mu + eps sqrt(var + 1e-10), eps ~ N(0, 1) a pure function of (key, element index): Box-Muller on two fmix32 hashes
 * (tf.random_normal in the reference, :427: a stateful generator nothing outside TensorFlow can replay).  d_mu = g. */
int sgcn_gauss_sample_f32(const float* dev_mu, const float* dev_var, int64_t n, uint32_t key, float* dev_x, void* stream);
int sgcn_gauss_sample_bwd_f32(const float* dev_var, const float* dev_g, int64_t n, uint32_t key, float* dev_d_var, void* stream);
/* operands of the control-variate aggregator on (mu, var), :320-349: delta_mu = mu - Hm[if], ds = sqrt(var) - sqrt(Hv[if]),
 * ds2 = ds^2, msig2 = 2 ds sqrt(Hv[if]); sbar = sqrt(Hv[if]) */
int sgcn_det_agg_prep_f32(const float* dev_mu, const float* dev_var, const float* dev_Hm, const float* dev_Hv, int64_t ldh,
                          const int32_t* dev_ifield, int32_t n0, int32_t d, float* dev_delta_mu, float* dev_ds2, float* dev_msig2,
                          float* dev_ds, float* dev_sbar, void* stream);
/* d_var = (2 ds g_ds2 + 2 sbar g_msig2) / (2 sqrt(var)) (+ add[r] for rows r < add_rows: the self half of a concat aggregator) */
int sgcn_det_agg_prep_bwd_f32(const float* dev_var, const float* dev_ds, const float* dev_sbar, const float* dev_g_ds2,
                              const float* dev_g_msig2, int32_t n0, int32_t d, const float* dev_add, int64_t ldadd,
                              int32_t add_rows, float* dev_d_var, void* stream);
/* y = relu(raw) + eps;   out = raw > 0 ? g : 0 */
int sgcn_relu_eps_f32(const float* dev_raw, int64_t ldr, int32_t n, int32_t d, float eps, float* dev_y, int64_t ldy, void* stream);
int sgcn_gate_f32(const float* dev_raw, int64_t ldr, const float* dev_g, int64_t ldg, int32_t n, int32_t d, float* dev_out, void* stream);

/* ======================================================================================
 * Host neighbour sampler (stays on host: BASELINE.json north_star)
 *   replaces class Scheduler gcn/scheduler.h:6-28, gcn/scheduler.cpp:11-189
 *   (driven by PyScheduler.batch gcn/_scheduler.pyx:55-127).  Index output is bit-exact
 *   with the reference for the same seed and call sequence; the RNG is an explicit
 *   MT19937 + the libstdc++ float draw (no dependence on the box's <random>).
 * ====================================================================================== */
typedef struct sgcn_sched sgcn_sched_t;

/* Copies the caller's CSR (gcn/scheduler.cpp:14-16); host_adj_p has num_data+1 entries. */
int sgcn_sched_create(const float* host_adj_w, const int32_t* host_adj_i,
                      const int32_t* host_adj_p, int32_t num_data, int32_t num_edges,
                      int32_t L, int32_t cv, int32_t is, sgcn_sched_t** out);
void sgcn_sched_destroy(sgcn_sched_t* s);
int sgcn_sched_seed(sgcn_sched_t* s, int32_t seed);                   /* scheduler.cpp:37-39 */
int sgcn_sched_start_batch(sgcn_sched_t* s, int32_t n, const int32_t* host_ids); /* :41-44 */
int sgcn_sched_expand(sgcn_sched_t* s, int32_t degree);               /* :46-189 */

/* Borrowed views of the last expand() result, valid until the next call on `s`. */
enum {
    SGCN_SCHED_FIELD = 0,   /* field    (input field of this layer)   int32 */
    SGCN_SCHED_FFIELD = 1,  /* ffield                                  int32 */
    SGCN_SCHED_EDG_S = 2,   /* COO row of adj  (non-decreasing)        int32 */
    SGCN_SCHED_EDG_T = 3,   /* COO col of adj                          int32 */
    SGCN_SCHED_FEDG_S = 4,  /* COO row of fadj                         int32 */
    SGCN_SCHED_FEDG_T = 5,  /* COO col of fadj                         int32 */
    SGCN_SCHED_EDG_P = 6,   /* CSR rowptr of adj  [n1+1]   (new: device CSR feed) */
    SGCN_SCHED_FEDG_P = 7,  /* CSR rowptr of fadj [n1+1] */
    SGCN_SCHED_ADJ_I = 8,   /* private (permuted) CSR column copy: statefulness probe */
    SGCN_SCHED_TEDG_P = 9,  /* CSR rowptr of adj^T [n0+1]  (for the backward SpMM, K6) */
    SGCN_SCHED_TEDG_T = 10  /* column ids of adj^T (= output-row ids i) */
};
enum {
    SGCN_SCHED_SCALES = 0,  /* 1/sqrt(deg/sampled) per output row      fp32 */
    SGCN_SCHED_EDG_W = 1,
    SGCN_SCHED_MEDG_W = 2,
    SGCN_SCHED_FEDG_W = 3,
    SGCN_SCHED_ADJ_W = 4,
    SGCN_SCHED_TEDG_W = 5   /* values of adj^T */
};
int sgcn_sched_view_i32(sgcn_sched_t* s, int32_t which, const int32_t** ptr, int64_t* len);
int sgcn_sched_view_f32(sgcn_sched_t* s, int32_t which, const float** ptr, int64_t* len);

/* One call = PyScheduler.batch (gcn/_scheduler.pyx:55-127): start_batch + L x expand with
 * degrees[L-l-1] (:66), the list reversal (:121-126) and labels[fields[-1]] (:138), plus what
 * the device step needs on top: CSR of adj / adj^T / fadj and their row plans.  Everything is
 * laid into two internal staging arrays (int32 / fp32, every sub-array 16-byte aligned) that
 * sgcn_sched_packed_copy() copies into caller-owned (pinned) buffers of *n_i32 / *n_f32
 * elements; `meta` (int64, length sgcn_sched_packed_meta_len(L)) receives the descriptors:
 *   [0]=L [1]=cv [2]=n_classes [3]=0
 *   fields  (L+1) x {off,len}      scales  L x {off,len}       ffields L x {off,len}
 *   labels  {off,rows,cols}        medg_w  L x {off,len}
 *   csr     L x {adj, adj^T, fadj} x {nrows,ncols,nnz,rowptr,col,val,seg,nseg,fix,nfix,nslots}
 *   tmedg_w L x {off,len}          (ABI v16: medg_w permuted like adj^T's values; control-variate hops only)
 * (offsets into the int32 buffer for index arrays, into the fp32 buffer for values; layer 0 =
 * input-most).  Touches no Python state, so a prefetch thread runs it outside the GIL. */
int sgcn_sched_batch_packed(sgcn_sched_t* s, int32_t n, const int32_t* host_ids, int32_t L,
                            const int32_t* host_degrees, const float* host_labels,
                            int32_t n_classes, int32_t plan_T, int64_t* meta, int64_t meta_cap,
                            int64_t* n_i32, int64_t* n_f32);
/* sgcn_sched_batch_packed + sgcn_sched_packed_copy in one call when the caller's staging buffer
 * (cap_words 4-byte words, laid out [max(n_i32,1) ints | max(n_f32,1) floats]) is large enough:
 * returns 0 and the buffer is filled; returns 1 (not an error) when it is too small -- grow it
 * and call sgcn_sched_packed_copy.  One foreign call per minibatch keeps the producer thread's
 * interpreter-lock traffic away from the launching thread. */
int sgcn_sched_batch_packed_into(sgcn_sched_t* s, int32_t n, const int32_t* host_ids, int32_t L,
                                 const int32_t* host_degrees, const float* host_labels,
                                 int32_t n_classes, int32_t plan_T, int64_t* host_meta,
                                 int64_t meta_cap, void* host_words, int64_t cap_words,
                                 int64_t* n_i32, int64_t* n_f32);
int64_t sgcn_sched_packed_meta_len(int32_t L);
/* the packed batch's fadj plan uses segments of max(plan_T, SGCN_AGG_PLAN_T) nonzeros: sgcn_vr_aggregate_* give a whole
 * workgroup to a segment (8 lane groups x 16 history rows in flight per lane = one round for 128 nonzeros) (ABI v7) */
#define SGCN_AGG_PLAN_T 128

/* ---- native prefetch thread: the sampler of a whole epoch off the interpreter ------------------
 * sgcn_prefetch_start copies the epoch's id slices (batch b = host_ids[offsets[b] : offsets[b+1]]) and
 * starts the producer threads.  With ONE sampler the batches are sampled 0, 1, ... in order -- the same
 * sample sequence as calling sgcn_sched_batch_packed in a loop -- either by one thread that also packs them
 * (n_packers = 0) or by a core thread (random draws, the CSR permutation, the rows' neighbour lists) feeding
 * n_packers > 0 packer threads (numbering of the full-neighbour field, transposes, plans, layout, the copy into
 * the slot): the same bits with only the core on the critical path (ABI v7).  With N samplers (the NON-PARITY
 * fast mode: independent RNG streams; n_packers must be 0) thread k packs batches k, k + N, ... and the consumer
 * still receives them in batch order.  Each batch goes into a free staging slot (host_slot_words[i]: cap
 * 4-byte words, pinned by the caller, laid out [max(n_i32,1) ints | max(n_f32,1) floats]).  The
 * sampler handle must not be used by anyone else until sgcn_prefetch_stop.
 * sgcn_prefetch_next blocks (holding no interpreter lock) for the next batch: 0 = filled *slot, sizes,
 * meta[sgcn_sched_packed_meta_len(L)] (*spill != NULL: the batch outgrew the slot and lives in that
 * heap buffer instead); 1 = epoch exhausted; < 0 = sampler error.  sgcn_prefetch_release hands a slot
 * back once the consumer's copy out of it has completed (`lag` = how many handed-out slots the
 * consumer keeps besides the current one; it bounds the producers' look-ahead).  Replaces the synchronous
 * `feed_dict = sch.minibatch(batch_size)` of gcn/train.py:189-191. */
typedef struct sgcn_prefetch sgcn_prefetch_t;
int sgcn_prefetch_start(sgcn_sched_t* const* samplers, int32_t n_samplers, int32_t n_batches,
                        const int32_t* host_ids, const int64_t* host_offsets, int32_t L,
                        const int32_t* host_degrees, const float* host_labels, int32_t n_classes,
                        int32_t plan_T, int32_t n_slots, void* const* host_slot_words,
                        const int64_t* host_slot_caps, int32_t lag, int32_t n_packers, sgcn_prefetch_t** out);
int sgcn_prefetch_next(sgcn_prefetch_t* p, int32_t* slot, int64_t* host_meta, int64_t* n_i32,
                       int64_t* n_f32, const void** spill);
int sgcn_prefetch_release(sgcn_prefetch_t* p, int32_t slot);
/* producer-side seconds so far: [waiting for a free slot, sampling + packing (with packers: packing, summed over them),
 * copying to slots, the core thread's sampling (with packers; else 0)] */
int sgcn_prefetch_stats(sgcn_prefetch_t* p, double* out4);
void sgcn_prefetch_stop(sgcn_prefetch_t* p);
int sgcn_sched_packed_copy(sgcn_sched_t* s, int32_t* dst_i32, float* dst_f32);

/* Fenwick-tree multinomial sampler without replacement (IS mode only)
 *   replaces struct Mult gcn/mult.h:8-27, gcn/mult.cpp:7-51 */
typedef struct sgcn_mult sgcn_mult_t;
int sgcn_mult_create(const float* host_prob, int32_t n, sgcn_mult_t** out);
void sgcn_mult_destroy(sgcn_mult_t* m);
int sgcn_mult_tree(sgcn_mult_t* m, const float** bit, int64_t* len); /* bit[0..N] */
int sgcn_mult_query_u(sgcn_mult_t* m, float u, int32_t* result);     /* mult.cpp:38-51 */
int sgcn_mult_query(sgcn_mult_t* m, int32_t* result);                /* mult.cpp:29-36 */


/* ---- one step as ONE call: the native launch loop (csrc/sgcn_step.cpp) --------------------------
 * The reference runs a training step as one sess.run of a static graph (gcn/vrgcn.py:72-82,
 * gcn/train.py:187-209).  A step PROGRAM is a flat list of calls to this library's own entry points;
 * argument j of an op evaluates to  mul[j] * slots[slot[j]] + add[j]  (slot[j] < 0: the constant
 * add[j]), pointers and integers as int64, floats as their bit pattern in the low 32 bits, the
 * sgcn_dropout_t / sgcn_plan_t arguments flattened to {on, fields...}.  The slot table is what changes
 * from minibatch to minibatch (row counts, addresses inside the staging buffer, dropout keys, the Adam
 * step size); the program is compiled once per model (stochastic_gcn_amd/step_program.py).  Ops are
 * executed in order on `stream` with exactly the arguments the eager path passes: same results, bit
 * for bit.  Argument order of every op = the parameter order of the entry point it names. */
#define SGCN_STEP_MAX_ARGS 48
enum {
    SGCN_OP_DENSE_FWD = 1,    /* sgcn_dense_fwd_f32 (ws, ws_capacity: split-K scratch used iff the library asks) */
    SGCN_OP_DENSE_BWD = 2,    /* sgcn_dense_bwd_f32 (g_tmp, ws, ws_capacity) */
    SGCN_OP_VR_AGG = 3,       /* sgcn_vr_aggregate_f32 */
    SGCN_OP_SPMM = 4,         /* sgcn_spmm_csr_f32, or sgcn_spmm_csr_add_f32 when add != NULL */
    SGCN_OP_SOFTMAX_CE = 5,   /* sgcn_softmax_ce_f32 */
    SGCN_OP_ADAM = 6,         /* sgcn_adam_f32 */
    SGCN_OP_SCATTER_ROWS = 7, /* sgcn_scatter_rows_f32 */
    SGCN_OP_MEMSET0 = 8,      /* hipMemsetAsync(ptr, 0, bytes) */
    SGCN_OP_DROPOUT = 9,      /* sgcn_dropout_f32 */
    SGCN_OP_L2_PENALTY = 10,  /* sgcn_l2_penalty_f32 */
    SGCN_OP_GATHER_ROWS = 11, /* sgcn_gather_rows_f32 */
    SGCN_OP_COPY2D = 12,      /* hipMemcpy2DAsync(dst, ldd, src, lds, rows, cols) in floats */
    SGCN_OP_SIGMOID_CE = 13,  /* sgcn_sigmoid_ce_f32 */
    SGCN_OP_VR_AGG_PRE = 14,  /* sgcn_vr_aggregate_pre_f32, issued on the auxiliary stream (forked from `stream`) */
    SGCN_OP_VR_AGG_POST = 15, /* `stream` waits for the auxiliary stream, then sgcn_vr_aggregate_post_f32 */
    SGCN_OP_AUX_SCATTER_ROWS = 16, /* sgcn_scatter_rows_f32 on the auxiliary stream (forked from `stream`) */
    SGCN_OP_AUX_MEMSET0 = 17, /* hipMemsetAsync on the auxiliary stream; joined before the first DENSE_BWD */
    /* 18, 19, 20: retired with ABI v6 (two dense layers as one launch; the loss / the lower layer's LayerNorm backward in
     * a GEMM epilogue -- measured slower than the separate launches in round 2: profiles/HISTORY.md 3.6) */
    SGCN_OP_DW_FLUSH = 21,      /* no arguments.  A program that contains this op runs in DEFERRED weight-gradient mode: every
                                 * DENSE_BWD before it only records its dW GEMM (+ split-K and LayerNorm-parameter
                                 * reductions); this op issues all of them as ONE grouped GEMM launch + ONE reduction launch
                                 * on `stream` (bit-identical to the layer-by-layer launches, 2 HIP calls instead of 4 per
                                 * layer) */
    SGCN_OP_MODE = 23,          /* (overlap, fuse), constants, anywhere in the program: this RUN uses the auxiliary stream iff
                                 * overlap != 0 and, when fuse >= 0, the fusion bits `fuse` -- instead of the process-wide
                                 * knobs step_overlap / step_fuse (which stay the defaults of programs without this op) */
    /* the sparse-input first layer (gcn/layers.py:125,401-402 with sparse_inputs), ABI v11: */
    SGCN_OP_CSR_SLICE = 24,     /* sgcn_csr_slice_indptr_dev + sgcn_csr_slice_f32: (n, rows, a_val, a_col, a_rowptr, o_p, o_d, o_col, o_row) */
    SGCN_OP_LN_ACT_FWD = 25,    /* sgcn_ln_act_fwd_f32 */
    SGCN_OP_LN_ACT_BWD = 26,    /* sgcn_ln_act_bwd_f32 (ws, ws_capacity in floats) */
    SGCN_OP_CSR_TRANSPOSE = 27, /* sgcn_csr_transpose_index (ws, ws_capacity in int32) */
    SGCN_OP_GATHER_F32 = 28,    /* sgcn_gather_f32 */
    /* the data-parallel step's exchanges (ABI v13): */
    SGCN_OP_ALLREDUCE_AVG = 29, /* sgcn_coll_allreduce_avg_f32 (buf, n) -- behind every gradient write of the run (joins the
                                 * auxiliary stream, flushes parked reductions) */
    SGCN_OP_HIST_PACK = 30,     /* sgcn_hist_pack_f32 (ids, n, rows, ld, d, cap, send, aux) */
    SGCN_OP_ALLGATHER_I32 = 31, /* sgcn_coll_allgather_i32 (send, recv, n, aux) */
    SGCN_OP_HIST_APPLY = 32,    /* sgcn_hist_apply_f32 (H, ldh, recv, world, cap, d, owner, aux); aux = 1 (30 - 32): on the auxiliary
                                 * stream, forked from `stream` at the op.  aux = 2 (ABI v15): on the library's EXCHANGE stream,
                                 * which waits for `stream` once, at the run's first such op -- an exchange issued right behind the
                                 * aggregator that read the history runs beside the loss, the backward pass, the gradient
                                 * all-reduce and the optimizer; the all-gather goes to the exchange communicator
                                 * (sgcn_coll_allgather_x_i32); `stream` waits for the exchange at the END of the run (the first
                                 * reader of the history is the next run's aggregator, gcn/models.py:186-194) */
    /* the --det_dropout stacks (gcn/layers.py:141-202, 236-248, 320-349, 425-428) as step programs (ABI v16): the entry
     * points of the same names with their arguments in order (element counts n = rows x pitch; floats as bit patterns) */
    SGCN_OP_GEMM = 33,          /* sgcn_gemm_f32 (ta, tb, M, N, K, A, lda, B, ldb, C, ldc, accumulate, ws, ws_capacity): split-K scratch
                                 * is passed exactly when sgcn_gemm_ws_floats(M, N, K) > 0, as the host path does */
    SGCN_OP_DET_PRE = 34,       /* sgcn_det_pre_f32 (mu, var, n, keep, var_out) */
    SGCN_OP_DET_PRE_BWD = 35,   /* sgcn_det_pre_bwd_f32 (mu, g, n, keep, d_mu, d_var) */
    SGCN_OP_SQUARE = 36,        /* sgcn_square_f32 (x, n, c, y) */
    SGCN_OP_ADDMUL = 37,        /* sgcn_addmul_f32 (acc, a, b, n, c) */
    SGCN_OP_DET_LNVAR_FWD = 38, /* sgcn_det_lnvar_fwd_f32 (var1, rstd, scale, n, d, eps, var2) */
    SGCN_OP_DET_LNVAR_BWD = 39, /* sgcn_det_lnvar_bwd_f32 (g, var1, xhat, rstd, scale, n, d, eps, d_var1, d_mu1, dscale, tmp) */
    SGCN_OP_DET_RELU_FWD = 40,  /* sgcn_det_relu_fwd_f32 (mu, var, n, mu_out, var_out) */
    SGCN_OP_DET_RELU_BWD = 41,  /* sgcn_det_relu_bwd_f32 (mu, var, g_mu, g_var, n, d_mu, d_var) */
    SGCN_OP_GAUSS = 42,         /* sgcn_gauss_sample_f32 (mu, var, n, key, x) */
    SGCN_OP_GAUSS_BWD = 43,     /* sgcn_gauss_sample_bwd_f32 (var, g, n, key, d_var) */
    SGCN_OP_DET_AGG_PREP = 44,  /* sgcn_det_agg_prep_f32 (mu, var, Hm, Hv, ldh, ifield, n0, d, delta_mu, ds2, msig2, ds, sbar) */
    SGCN_OP_DET_AGG_PREP_BWD = 45, /* sgcn_det_agg_prep_bwd_f32 (var, ds, sbar, g_ds2, g_msig2, n0, d, add, ldadd, add_rows, d_var) */
    SGCN_OP_RELU_EPS = 46,      /* sgcn_relu_eps_f32 (raw, ldr, n, d, eps, y, ldy) */
    SGCN_OP_GATE = 47,          /* sgcn_gate_f32 (raw, ldr, g, ldg, n, d, out) */
    SGCN_OP_GRAD_STORE = 22     /* no arguments, anywhere in the program: the run is in gradient-STORE mode -- every DENSE_BWD
                                 * writes its dW / doffset / dscale instead of adding to them, so the program zeroes nothing
                                 * (it must write every parameter gradient exactly once per step); and the statistics
                                 * reduction of SOFTMAX_CE / SIGMOID_CE runs in the launch of an ADAM op later in the SAME run
                                 * when no L2_PENALTY adds to the loss in between (otherwise at the end of the run) */
};
typedef struct {
    int32_t op, nargs;
    int64_t mul[SGCN_STEP_MAX_ARGS];
    int32_t slot[SGCN_STEP_MAX_ARGS];
    int64_t add[SGCN_STEP_MAX_ARGS];
} sgcn_step_op_t;
int sgcn_step_run(const sgcn_step_op_t* host_ops, int32_t nops, const int64_t* host_slots, int32_t nslots,
                  void* stream);

/* The slot table of one minibatch, filled in C (ABI v8: as NumPy expressions on the launching thread this was ~25 us of a
 * 135 us step, on a thread that has ~115 us of work per step):
 *   host_slots[i] = host_meta[idx[i]] * mul[i] + (base[i] == 1 ? ip : base[i] == 2 ? fp : 0)           i < n
 * (host_meta: the descriptor table of sgcn_sched_batch_packed; ip / fp: device addresses of the minibatch's int32 / fp32
 * sections), behind the program's capacity checks -- host_meta[cap_idx[j]] <= cap_max[j] for j < n_cap (row counts against
 * the arena the program was built for), host_meta[ws_idx[j]] * ws_ld[j] <= ws_floats for j < n_ws (plan workspaces) -- then the
 * step's dropout keys host_slots[key_slot[j]] = fmix32(fmix32(seed * 0x9E3779B1 + key_layer[j] * 0x85EBCA77 + 0x27D4EB2F)
 * + step * 0xC2B2AE3D) (sgcn_dropout_t.key of site key_layer[j] at this step) and host_slots[lr_slot] = the bits of `lr`
 * (the Adam step size of this step).  Returns SGCN_OK; 1 when the minibatch does not fit the program (host_slots
 * untouched: the caller runs this minibatch op by op); SGCN_ERR_INVALID on an index outside host_meta / host_slots. */
typedef struct {
    int64_t n;      const int64_t* idx;      const int64_t* mul;       const int64_t* base;
    int64_t n_cap;  const int64_t* cap_idx;  const int64_t* cap_max;
    int64_t n_ws;   const int64_t* ws_idx;   const int64_t* ws_ld;     int64_t ws_floats;
    int64_t n_keys; const int64_t* key_slot; const int64_t* key_layer; int64_t lr_slot;
} sgcn_step_fill_t;
int sgcn_step_fill(const sgcn_step_fill_t* fill, const int64_t* host_meta, int64_t meta_len, int64_t ip, int64_t fp,
                   int64_t seed, int64_t step, float lr, int64_t* host_slots, int64_t nslots);

/* hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, stream): the staging copy of a packed minibatch
 * (stochastic_gcn_amd/models.py Model.stage) as one foreign call instead of a stream context + tensor copy. */
int sgcn_copy_h2d_async(void* dev_dst, const void* host_src, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGCN_H */
