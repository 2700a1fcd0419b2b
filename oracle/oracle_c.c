/* oracle/oracle_c.c -- TEST INFRASTRUCTURE ONLY (parity checker + bench.py cpu_baseline leg).
 *
 * Plain-C CPU restatement of the sparse kernels on the hot path of thu-ml/stochastic_gcn.
 * The reference executes them as TensorFlow-1 ops (third-party, un-vendored, un-pinned:
 * `tensorflow-gpu` in /root/reference/setup.py:12-16), so this file restates the *op
 * definitions* at the reference's call sites.  PARITY UNPINNED at the TF boundary: the
 * reference holds no test/golden vector for these ops; the restatement is
 * cross-checked against scipy.sparse (the library the reference itself uses for the PP
 * product, gcn/utils.py:169-170,321-322) in tests/test_oracle.py.
 *
 * Loop order follows the authors' own (commented-out) CPU kernel for exactly this product,
 * gcn/history.cpp:10-48: row-parallel, `o[k] += v * history[idx[j]*dims + k]`, fp32
 * accumulation in storage order of the row's nonzeros.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <stdint.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* thread count of the row-parallel loops (bench.py times the baseline on all cores and on one) */
void oracle_set_threads(int32_t n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int32_t oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Touch a freshly allocated buffer page by page from all threads (static round-robin): with the
 * threads spread over the sockets (OMP_PROC_BIND=spread) first-touch placement interleaves the pages
 * over the NUMA nodes -- what a gather-bound SpMM wants of its dense operand (bench.py cpu_baseline). */
void oracle_first_touch_interleaved(float* p, int64_t n) {
    const int64_t page = 4096 / (int64_t)sizeof(float);
    const int64_t npages = (n + page - 1) / page;
#pragma omp parallel for schedule(static, 1)
    for (int64_t g = 0; g < npages; g++) {
        const int64_t lo = g * page, hi = lo + page < n ? lo + page : n;
        for (int64_t i = lo; i < hi; i++) p[i] = 0.0f;
    }
}

/* C[M x d] = A[M x K, CSR] * B[K x d]           (gcn/layers.py:31-37 dot(sparse=True) ->
 * tf.sparse_tensor_dense_matmul; K1/K11 in SURVEY §2.1).  beta==0 overwrites C. */
void oracle_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                         int32_t M, int32_t d, const float* B, int64_t ldb,
                         float* C, int64_t ldc, float beta) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int32_t i = 0; i < M; i++) {
        float* o = C + (int64_t)i * ldc;
        if (beta == 0.0f) memset(o, 0, (size_t)d * sizeof(float));
        else for (int32_t k = 0; k < d; k++) o[k] *= beta;
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++) {
            const float v = val[p];
            const float* c = B + (int64_t)col[p] * ldb;
            for (int32_t k = 0; k < d; k++) o[k] += v * c[k];
        }
    }
}

/* C = A * H[gidx]    (gcn/layers.py:305,308: tf.gather(history, ffield) then
 * dot(fadj, mu_large); K2+K7).  gidx maps A's column space into rows of H. */
void oracle_spmm_csr_gather_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                int32_t M, int32_t d, const float* H, int64_t ldh,
                                const int32_t* gidx, float* C, int64_t ldc) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int32_t i = 0; i < M; i++) {
        float* o = C + (int64_t)i * ldc;
        memset(o, 0, (size_t)d * sizeof(float));
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++) {
            const float v = val[p];
            const float* c = H + (int64_t)gidx[col[p]] * ldh;
            for (int32_t k = 0; k < d; k++) o[k] += v * c[k];
        }
    }
}

/* out[i,:] = in[r[i],:]   (gcn/history.cpp:74-88 c_dense_slice; tf.gather gcn/layers.py:304) */
void oracle_gather_rows_f32(const float* in, int64_t ldi, const int32_t* r, int32_t n,
                            int32_t d, float* out, int64_t ldo) {
    for (int32_t i = 0; i < n; i++)
        memcpy(out + (int64_t)i * ldo, in + (int64_t)r[i] * ldi, (size_t)d * sizeof(float));
}

/* H[r[i],:] = src[i,:]    (gcn/models.py:160-166 tf.scatter_update; r unique) */
void oracle_scatter_rows_f32(float* H, int64_t ldh, const int32_t* r, int32_t n, int32_t d,
                             const float* src, int64_t lds) {
    for (int32_t i = 0; i < n; i++)
        memcpy(H + (int64_t)r[i] * ldh, src + (int64_t)i * lds, (size_t)d * sizeof(float));
}
