// What sgcn_step_run hands to the loss kernel when it folds the layers around the loss into its row pass (host-side PODs:
// shared by sgcn_step.cpp, a host translation unit, and sgcn_dense.hip).  All pointers are device pointers.
#pragma once
#include <cstdint>
#include "../../include/sgcn.h"

namespace sgcn {

// the output layer (no LayerNorm, no ReLU, <= 64 classes): W [K x c]
struct CeLastLayer {
    const float* W = nullptr; int64_t ldw = 0; int32_t K = 0;
    // its input gradient dx = dlogits . W^T (* dx_drop) as the tail of the row pass (dx == nullptr: off)
    float* dx = nullptr; int64_t lddx = 0; const sgcn_dropout_t* dx_drop = nullptr;
    // its forward product as the head (hx == nullptr and !pre: off): logits = dropout(hx) . W, kg = K-groups of the launch replaced
    const float* hx = nullptr; int64_t ldhx = 0; int32_t kg = 1; const sgcn_dropout_t* h_drop = nullptr;
    // the dense layer IN FRONT of the output layer (pre: the head's input never leaves the registers):
    //   y = act(LN(dropout(px)[n x PK] . PW[PK x K])), written to pY / pxhat / prstd for the backward pass;
    //   pS / pkchunk / pkg = the K slices, slice length and K-groups of the launch (+ split-K reduce) it replaces
    bool pre = false;
    const float* px = nullptr; int64_t ldpx = 0; int32_t PK = 0; const float* PW = nullptr;
    int32_t pS = 1, pkchunk = 0, pkg = 1; const sgcn_dropout_t* p_drop = nullptr;
    const float* poff = nullptr; const float* psc = nullptr; float peps = 0.f; int32_t prelu = 0;
    float* pY = nullptr; int64_t ldpy = 0; float* pxhat = nullptr; float* prstd = nullptr;
};

}  // namespace sgcn
