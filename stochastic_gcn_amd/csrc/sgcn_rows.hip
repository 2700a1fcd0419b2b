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

}  // namespace sgcn

using namespace sgcn;

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
