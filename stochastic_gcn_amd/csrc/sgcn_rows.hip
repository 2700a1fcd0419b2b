// Row gather / scatter / CSR row-slice on device (gfx950).
//   gather  : out[i,:] = in[r[i],:]      replaces history.dense_slice (gcn/_history.pyx:53-62,
//             gcn/history.cpp:74-88: single-thread scalar host loop, then a host->device feed
//             every step) and tf.gather(history, field) (gcn/layers.py:304-305)
//   scatter : H[r[i],:] = src[i,:]       replaces tf.scatter_update (gcn/models.py:160-166);
//             r[i] < 0 skips row i (fixed-capacity multi-GPU history exchange pads with -1)
//   slice   : CSR rows r -> CSR          replaces history.slice (gcn/_history.pyx:25-51,
//             gcn/history.cpp:50-72)
// All three are pure HBM copies: one G-lane group per row moving 16-byte vectors, so a
// wavefront's accesses to a row are contiguous; the feature matrix and the history stay
// resident in HBM (288 GB) instead of being gathered on the host.
#include "sgcn_dev.h"

namespace sgcn {

int group_lanes(int nvec);  // sgcn_spmm.hip

template <int VW, bool SCATTER>
__global__ __launch_bounds__(kBlock) void rows_kernel(const float* __restrict__ in, int64_t ldi,
                                                      const int32_t* __restrict__ r, int32_t n,
                                                      int32_t d, int32_t nvec, int32_t G,
                                                      float* __restrict__ out, int64_t ldo) {
    const int gpb = kBlock / G;
    const int lig = threadIdx.x % G;
    const int64_t i = (int64_t)blockIdx.x * gpb + threadIdx.x / G;
    if (i >= n) return;
    const int64_t ri = r[i];
    if (SCATTER && ri < 0) return;          // negative row id: padding slot, nothing to write
    const float* src = in + (SCATTER ? i : ri) * ldi;
    float* dst = out + (SCATTER ? ri : i) * ldo;
    for (int vi = lig; vi < nvec; vi += G) {
        const int left = d - vi * VW;
        if (left >= VW) vstore<VW>(dst + (int64_t)vi * VW, vload<VW>(src + (int64_t)vi * VW));
        else
            for (int e = 0; e < left; e++) dst[(int64_t)vi * VW + e] = src[(int64_t)vi * VW + e];
    }
}

template <bool SCATTER>
static int launch_rows(const float* in, int64_t ldi, const int32_t* r, int32_t n, int32_t d,
                       float* out, int64_t ldo, hipStream_t st) {
    // a ragged last vector is copied element-wise, so only pitch/alignment matter here
    int vw = 1;
    if (ldi % 4 == 0 && ldo % 4 == 0 && aligned16(in) && aligned16(out)) vw = 4;
    else if (ldi % 2 == 0 && ldo % 2 == 0 && aligned8(in) && aligned8(out)) vw = 2;
    const int nvec = (d + vw - 1) / vw;
    const int G = group_lanes(nvec);
    const int64_t blocks = ((int64_t)n + (kBlock / G) - 1) / (kBlock / G);
    dim3 grid((unsigned)blocks), block(kBlock);
    if (vw == 4) hipLaunchKernelGGL((rows_kernel<4, SCATTER>), grid, block, 0, st, in, ldi, r, n, d, nvec, G, out, ldo);
    else if (vw == 2) hipLaunchKernelGGL((rows_kernel<2, SCATTER>), grid, block, 0, st, in, ldi, r, n, d, nvec, G, out, ldo);
    else hipLaunchKernelGGL((rows_kernel<1, SCATTER>), grid, block, 0, st, in, ldi, r, n, d, nvec, G, out, ldo);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

// One wavefront per output row: copies values and column ids, optionally writes COO row ids.
__global__ __launch_bounds__(kBlock) void csr_slice_kernel(int32_t n, const int32_t* __restrict__ r,
                                                           const float* __restrict__ a_d,
                                                           const int32_t* __restrict__ a_i,
                                                           const int32_t* __restrict__ a_p,
                                                           const int32_t* __restrict__ o_p,
                                                           float* __restrict__ o_d,
                                                           int32_t* __restrict__ o_col,
                                                           int32_t* __restrict__ o_row) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (i >= n) return;
    const int32_t src = a_p[r[i]];
    const int32_t dst = o_p[i];
    const int32_t len = o_p[i + 1] - dst;
    for (int32_t j = lane; j < len; j += kWave) {
        o_d[dst + j] = a_d[src + j];
        o_col[dst + j] = a_i[src + j];
        if (o_row) o_row[dst + j] = (int32_t)i;
    }
}

// ---- stable CSR -> transposed-CSR index (device counting sort, deterministic) -------------------
// The sparse first layer's weight gradient is dW = X[f0]^T g (autodiff of gcn/layers.py:125,401-402
// with sparse_inputs): X[f0] is the minibatch's row slice, so its transpose is needed once per
// minibatch.  Nonzeros are cut into chunks of kTChunk in storage (row-major) order:
//   t_hist   per-chunk column histogram H[b][c]                       (integer LDS atomics)
//   t_scan   column totals -> exclusive scan = t_rowptr; H[b][c] <- first output slot of chunk b in column c
//   t_place  one lane per chunk walks its nonzeros IN ORDER: slot = H[b][c]++   -> rows ascending inside
//            every column, whatever the schedule (the float sums that consume the transpose are ordered)
constexpr int kTChunk = 512;
constexpr int kTLdsCols = 16384;     // counters of a chunk live in LDS up to this many columns

__global__ __launch_bounds__(kBlock) void t_hist_kernel(const int32_t* __restrict__ col, int64_t nnz, int32_t ncols,
                                                        int32_t* __restrict__ H) {
    int32_t* h = H + (int64_t)blockIdx.x * ncols;
    for (int c = threadIdx.x; c < ncols; c += kBlock) h[c] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * kTChunk;
    for (int64_t p = p0 + threadIdx.x; p < min(nnz, p0 + kTChunk); p += kBlock) atomicAdd(&h[col[p]], 1);
}

__global__ __launch_bounds__(kBlock) void t_scan_kernel(int32_t* __restrict__ H, int32_t nchunks, int32_t ncols,
                                                        int32_t* __restrict__ t_rowptr) {
    __shared__ int32_t part[kBlock];
    // column totals into t_rowptr[c + 1]
    for (int c = threadIdx.x; c < ncols; c += kBlock) {
        int32_t s = 0;
        for (int b = 0; b < nchunks; b++) s += H[(int64_t)b * ncols + c];
        t_rowptr[c + 1] = s;
    }
    __syncthreads();
    // exclusive scan over the columns: each thread owns a contiguous span
    const int span = (ncols + kBlock - 1) / kBlock;
    const int lo = min(ncols, (int)threadIdx.x * span), hi = min(ncols, lo + span);
    int32_t s = 0;
    for (int c = lo; c < hi; c++) s += t_rowptr[c + 1];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t run = 0;
        for (int t = 0; t < kBlock; t++) { const int32_t v = part[t]; part[t] = run; run += v; }
        t_rowptr[0] = 0;
    }
    __syncthreads();
    int32_t run = part[threadIdx.x];
    for (int c = lo; c < hi; c++) { run += t_rowptr[c + 1]; t_rowptr[c + 1] = run; }
    __syncthreads();
    // first slot of every (chunk, column)
    for (int c = threadIdx.x; c < ncols; c += kBlock) {
        int32_t r = t_rowptr[c];
        for (int b = 0; b < nchunks; b++) {
            const int64_t k = (int64_t)b * ncols + c;
            const int32_t v = H[k];
            H[k] = r;
            r += v;
        }
    }
}

template <bool LDS>
__global__ __launch_bounds__(kWave) void t_place_kernel(const int32_t* __restrict__ col, const int32_t* __restrict__ coo_row,
                                                        int64_t nnz, int32_t ncols, int32_t* __restrict__ H,
                                                        int32_t* __restrict__ t_row, int32_t* __restrict__ t_src) {
    extern __shared__ int32_t lds_cnt[];
    int32_t* h = H + (int64_t)blockIdx.x * ncols;
    if constexpr (LDS) {
        for (int c = threadIdx.x; c < ncols; c += kWave) lds_cnt[c] = h[c];
        __syncthreads();
        h = lds_cnt;
    }
    if (threadIdx.x != 0) return;
    const int64_t p0 = (int64_t)blockIdx.x * kTChunk, p1 = min(nnz, p0 + kTChunk);
    for (int64_t p = p0; p < p1; p++) {
        const int32_t slot = h[col[p]]++;
        t_row[slot] = coo_row[p];
        t_src[slot] = (int32_t)p;
    }
}

__global__ __launch_bounds__(kBlock) void gather_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                            int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = src[idx[i]];
}


// One wavefront that does nothing for `ticks` of the 100 MHz counter: the step program's exchange stream (sgcn_step.cpp)
// uses it once, to find out whether a candidate stream shares the step's hardware queue (HIP multiplexes streams onto a few).
__global__ void spin_kernel(uint64_t ticks) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int spin_launch(void* stream, int64_t usec) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(kWave), 0, (hipStream_t)stream, (uint64_t)usec * 100u);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}
}  // namespace sgcn

using namespace sgcn;

extern "C" int64_t sgcn_csr_transpose_ws_ints(int32_t ncols, int64_t nnz) {
    if (ncols <= 0 || nnz <= 0) return 0;
    return ((nnz + kTChunk - 1) / kTChunk) * (int64_t)ncols;
}

extern "C" int sgcn_csr_transpose_index(int32_t ncols, int64_t nnz, const int32_t* col, const int32_t* coo_row,
                                        int32_t* t_rowptr, int32_t* t_row, int32_t* t_src, int32_t* ws, void* stream) {
    SGCN_REQUIRE(ncols >= 0 && nnz >= 0 && nnz < (1ll << 31), "csr_transpose_index: bad size");
    SGCN_REQUIRE(t_rowptr || ncols == 0, "csr_transpose_index: null t_rowptr");
    hipStream_t st = (hipStream_t)stream;
    if (nnz == 0) {
        if (ncols > 0) SGCN_HIP_TRY(hipMemsetAsync(t_rowptr, 0, (size_t)(ncols + 1) * 4, st));
        return SGCN_OK;
    }
    SGCN_REQUIRE(col && coo_row && t_row && t_src && ws, "csr_transpose_index: null operand");
    const int64_t nchunks = (nnz + kTChunk - 1) / kTChunk;
    hipLaunchKernelGGL(t_hist_kernel, dim3((unsigned)nchunks), dim3(kBlock), 0, st, col, nnz, ncols, ws);
    hipLaunchKernelGGL(t_scan_kernel, dim3(1), dim3(kBlock), 0, st, ws, (int32_t)nchunks, ncols, t_rowptr);
    if (ncols <= kTLdsCols)
        hipLaunchKernelGGL(t_place_kernel<true>, dim3((unsigned)nchunks), dim3(kWave), (size_t)ncols * 4, st, col, coo_row,
                           nnz, ncols, ws, t_row, t_src);
    else
        hipLaunchKernelGGL(t_place_kernel<false>, dim3((unsigned)nchunks), dim3(kWave), 0, st, col, coo_row, nnz, ncols,
                           ws, t_row, t_src);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

// out[r, :] = s[r] * x[r, :]: the operand of a product whose matrix carries one value per COLUMN (the transpose of a
// row-normalised adjacency, i.e. the backward of a mean aggregation): M . B = pattern(M) . (s (.) B).  float4 per lane.
__global__ __launch_bounds__(kBlock) void scale_rows_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ s,
                                                            int32_t n, int32_t nvec, int32_t d, float* __restrict__ out, int64_t ldo) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)n * nvec) return;
    const int32_t r = (int32_t)(i / nvec), v = (int32_t)(i % nvec);
    const float f = s[r];
    const float* src = x + (int64_t)r * ldx + 4 * v;
    float* dst = out + (int64_t)r * ldo + 4 * v;
    if (4 * v + 4 <= d) {
        float4 t = *reinterpret_cast<const float4*>(src);
        t.x *= f; t.y *= f; t.z *= f; t.w *= f;
        *reinterpret_cast<float4*>(dst) = t;
    } else {
        for (int e = 0; 4 * v + e < d; e++) dst[e] = f * src[e];
    }
}

extern "C" int sgcn_scale_rows_f32(const float* x, int64_t ldx, const float* s, int32_t n, int32_t d, float* out, int64_t ldo,
                                   void* stream) {
    SGCN_REQUIRE(n >= 0 && d >= 0, "scale_rows: negative size");
    if (n == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(x && s && out && ldx >= d && ldo >= d, "scale_rows: bad operand");
    SGCN_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0,
                 "scale_rows: rows must be 16-byte aligned");
    const int32_t nvec = (d + 3) / 4;
    const int64_t blocks = ((int64_t)n * nvec + kBlock - 1) / kBlock;
    SGCN_REQUIRE(blocks < (1ll << 31), "scale_rows: too large");
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, ldx, s, n, nvec, d, out, ldo);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

extern "C" int sgcn_gather_f32(const float* src, const int32_t* idx, int64_t n, float* out, void* stream) {
    SGCN_REQUIRE(n >= 0, "gather: negative size");
    if (n == 0) return SGCN_OK;
    SGCN_REQUIRE(src && idx && out, "gather: null operand");
    const unsigned blocks = (unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 4096);
    hipLaunchKernelGGL(gather_f32_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, src, idx, n, out);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

extern "C" int sgcn_gather_rows_f32(const float* in, int64_t ldi, const int32_t* r, int32_t n,
                                    int32_t d, float* out, int64_t ldo, void* stream) {
    SGCN_REQUIRE(n >= 0 && d >= 0, "gather_rows: negative size");
    if (n == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(in && r && out && ldi >= d && ldo >= d, "gather_rows: bad operand");
    return launch_rows<false>(in, ldi, r, n, d, out, ldo, (hipStream_t)stream);
}

extern "C" int sgcn_scatter_rows_f32(float* H, int64_t ldh, const int32_t* r, int32_t n,
                                     int32_t d, const float* src, int64_t lds, void* stream) {
    SGCN_REQUIRE(n >= 0 && d >= 0, "scatter_rows: negative size");
    if (n == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(H && r && src && ldh >= d && lds >= d, "scatter_rows: bad operand");
    return launch_rows<true>(src, lds, r, n, d, H, ldh, (hipStream_t)stream);
}

// ---- the multi-GPU history exchange (policy H-a) as two launches around one all-gather ---------------------
// pack : send = [cap ids | cap x d row bits] of this rank's update -- ids[0..n) and their rows, ids[n..cap) = -1
// apply: every rank's block of the gathered buffer scattered into the local replica, in RANK ORDER (a vertex that two
//        ranks updated in one step keeps the higher rank's row on every replica: W scatter launches, stream-ordered)
__global__ __launch_bounds__(kBlock) void hist_pack_kernel(const int32_t* __restrict__ ids, int32_t n, const float* __restrict__ rows,
                                                           int64_t ld, int32_t d, int32_t cap, int32_t* __restrict__ send) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (i >= cap) return;
    if (i >= n) {
        if (lane == 0) send[i] = -1;
        return;
    }
    if (lane == 0) send[i] = ids[i];
    const float* src = rows + i * ld;
    float* dst = reinterpret_cast<float*>(send + cap) + i * d;
    for (int c = lane; c < d; c += kWave) dst[c] = src[c];
}

extern "C" int sgcn_hist_pack_f32(const int32_t* ids, int32_t n, const float* rows, int64_t ld, int32_t d, int32_t cap,
                                  int32_t* send, void* stream) {
    SGCN_REQUIRE(n >= 0 && d >= 0 && cap >= n, "hist_pack: %d rows do not fit the capacity %d", n, cap);
    if (cap == 0) return SGCN_OK;
    SGCN_REQUIRE(send && (n == 0 || (ids && rows && ld >= d)), "hist_pack: bad operand");
    const int64_t blocks = ((int64_t)cap + (kBlock / kWave) - 1) / (kBlock / kWave);
    hipLaunchKernelGGL(hist_pack_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, ids, n, rows, ld, d, cap, send);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

// The same result in TWO launches whatever the number of ranks (eight scatter launches are ~25 us of an 8-GPU step's dependent
// chain): `owner` (one word per history row, zero between calls) first takes, per vertex, the highest (rank, slot) that updates it
// -- an integer atomicMax, order-independent --, then every (rank, slot) that owns its vertex copies its row and clears the word.
__global__ __launch_bounds__(kBlock) void hist_claim_kernel(const int32_t* __restrict__ recv, int32_t world, int32_t cap, int64_t per,
                                                            int32_t* __restrict__ owner) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= (int64_t)world * cap) return;
    const int32_t r = (int32_t)(k / cap), i = (int32_t)(k % cap);
    const int32_t id = recv[r * per + i];
    if (id >= 0) atomicMax(&owner[id], (int32_t)k + 1);
}

__global__ __launch_bounds__(kBlock) void hist_write_kernel(float* __restrict__ H, int64_t ldh, const int32_t* __restrict__ recv,
                                                            int32_t world, int32_t cap, int32_t d, int64_t per,
                                                            int32_t* __restrict__ owner) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t k = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (k >= (int64_t)world * cap) return;
    const int32_t r = (int32_t)(k / cap), i = (int32_t)(k % cap);
    const int32_t id = recv[r * per + i];
    if (id < 0 || owner[id] != (int32_t)k + 1) return;
    const float* src = reinterpret_cast<const float*>(recv + r * per + cap) + (int64_t)i * d;
    float* dst = H + (int64_t)id * ldh;
    for (int c = lane; c < d; c += kWave) dst[c] = src[c];
    if (lane == 0) owner[id] = 0;                     // (only the owner of a vertex touches its word in this launch)
}

extern "C" int sgcn_hist_apply_f32(float* H, int64_t ldh, const int32_t* recv, int32_t world, int32_t cap, int32_t d,
                                   int32_t* owner, void* stream) {
    SGCN_REQUIRE(world >= 1 && cap >= 0 && d >= 0, "hist_apply: bad size");
    if (cap == 0 || d == 0) return SGCN_OK;
    SGCN_REQUIRE(H && recv && ldh >= d, "hist_apply: bad operand");
    const int64_t per = (int64_t)cap * (d + 1);
    if (owner && world > 2) {
        const int64_t n = (int64_t)world * cap;
        hipLaunchKernelGGL(hist_claim_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                           recv, world, cap, per, owner);
        hipLaunchKernelGGL(hist_write_kernel, dim3((unsigned)((n + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock), 0,
                           (hipStream_t)stream, H, ldh, recv, world, cap, d, per, owner);
        SGCN_HIP_TRY(hipGetLastError());
        return SGCN_OK;
    }
    for (int32_t r = 0; r < world; r++) {
        const int32_t* blk = recv + r * per;
        const int rc = launch_rows<true>(reinterpret_cast<const float*>(blk + cap), d, blk, cap, d, H, ldh, (hipStream_t)stream);
        if (rc != SGCN_OK) return rc;
    }
    return SGCN_OK;
}

// o_p of a row slice on the device (what sgcn_csr_slice_indptr computes on the host, gcn/history.cpp:50-58): ONE workgroup,
// every thread a contiguous run of rows, run totals scanned through LDS.  For the compiled step (SGCN_OP_CSR_SLICE): the
// minibatch's input vertices are on the device already and no host pass / copy sits in front of the step.
__global__ __launch_bounds__(kBlock) void csr_slice_indptr_kernel(int32_t n, const int32_t* __restrict__ r,
                                                                  const int32_t* __restrict__ a_p, int32_t* __restrict__ o_p) {
    __shared__ int32_t part[kBlock];
    const int t = threadIdx.x;
    const int32_t per = (n + kBlock - 1) / kBlock;
    const int32_t b = min(n, t * per), e = min(n, b + per);
    int32_t s = 0;
    for (int32_t i = b; i < e; i++) { const int32_t row = r[i]; s += a_p[row + 1] - a_p[row]; }
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int32_t acc = 0;
        for (int k = 0; k < kBlock; k++) { const int32_t v = part[k]; part[k] = acc; acc += v; }
        o_p[n] = acc;
    }
    __syncthreads();
    int32_t acc = part[t];
    for (int32_t i = b; i < e; i++) { o_p[i] = acc; const int32_t row = r[i]; acc += a_p[row + 1] - a_p[row]; }
}

extern "C" int sgcn_csr_slice_indptr_dev(int32_t n, const int32_t* r, const int32_t* a_p, int32_t* o_p, void* stream) {
    SGCN_REQUIRE(n >= 0, "csr_slice_indptr_dev: negative size");
    SGCN_REQUIRE(o_p && (n == 0 || (r && a_p)), "csr_slice_indptr_dev: null operand");
    hipLaunchKernelGGL(csr_slice_indptr_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, n, r, a_p, o_p);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}

extern "C" int sgcn_csr_slice_f32(int32_t n, const int32_t* r, const float* a_d,
                                  const int32_t* a_i, const int32_t* a_p, const int32_t* o_p,
                                  float* o_d, int32_t* o_col, int32_t* o_row, void* stream) {
    SGCN_REQUIRE(n >= 0, "csr_slice: negative size");
    if (n == 0) return SGCN_OK;
    SGCN_REQUIRE(r && a_p && o_p, "csr_slice: null index operand");
    const int64_t blocks = ((int64_t)n + (kBlock / kWave) - 1) / (kBlock / kWave);
    hipLaunchKernelGGL(csr_slice_kernel, dim3((unsigned)blocks), dim3(kBlock), 0,
                       (hipStream_t)stream, n, r, a_d, a_i, a_p, o_p, o_d, o_col, o_row);
    SGCN_HIP_TRY(hipGetLastError());
    return SGCN_OK;
}
