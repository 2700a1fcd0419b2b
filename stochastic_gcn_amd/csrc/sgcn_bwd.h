// Internal: the dense layer's backward as a value (shared by sgcn_gemm.hip and the step interpreter).
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
// sgcn_gemm.hip: two consecutive layers' backward where the upper layer's dx is the lower layer's dy -- the lower
// layer's LayerNorm / ReLU backward runs in the epilogue of the upper layer's input-gradient GEMM
int dense_bwd_pair(const DenseBwdArgs& upper, const DenseBwdArgs& lower, void* stream, bool overlap);
}  // namespace sgcn
