// Internal: the dense layer's backward as a value (sgcn_gemm.hip).
#pragma once
#include <cstdint>

#include "../../include/sgcn.h"

namespace sgcn {
struct DenseBwdArgs {           // the arguments of sgcn_dense_bwd_f32
    int32_t n, N, K;
    const float* dy; int64_t lddy;
    const float* y; int64_t ldy;
    const float* xhat; const float* rstd; const float* scale; int32_t relu;
    const float* x; int64_t ldx;
    const float* W; int64_t ldw;
    float* dW; int64_t lddw;
    float* doffset; float* dscale;
    float* dx; int64_t lddx;
    const sgcn_dropout_t* drop;
    float* g_tmp; float* ws;
    const int32_t* gidx;
};
}  // namespace sgcn
