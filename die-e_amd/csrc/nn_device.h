// nn_device.h -- the tail of the network shared by the API path (k_softmax_value) and the search (k_expand):
// softmax over the 1352 policy logits and the value head's FC + tanh (nnet.rs:83-85, 95-98).  Both callers run these
// functions on one wave per board, so the search reads the SAME bits the API returns (the oracle-based parity tests
// rely on that) without a softmax launch and a [boards][1352] policy round trip per evaluation.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "wave_ops.h"

namespace diee {

// M = max logit, inv = 1 / sum exp(logit - M) of one board's 1352 logits (22 per lane, butterfly reductions)
// in two halves so that a caller can put the row's loads in flight long before it needs the constants
__device__ __forceinline__ void softmax_load(const float* __restrict__ lr, int lane, float (&v)[22]) {
#pragma unroll
    for (int j = 0; j < 22; ++j) {
        const int a = lane + 64 * j;
        v[j] = a < 1352 ? lr[a] : -INFINITY;
    }
}
__device__ __forceinline__ void softmax_reduce(const float (&v)[22], int lane, float& M, float& inv) {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 22; ++j) mx = fmaxf(mx, v[j]);
    mx = wave_allmax_f32(mx);                               // a maximum: any pairing order gives the same bits
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 22; ++j) sum += lane + 64 * j < 1352 ? expf(v[j] - mx) : 0.0f;
    sum = wave_butterfly_sum(sum);                          // the xor butterfly's pairing order, off the LDS pipe (wave_ops.h)
    M = mx; inv = 1.0f / sum;
}
__device__ __forceinline__ void softmax_consts(const float* __restrict__ lr, int lane, float& M, float& inv) {
    float v[22];
    softmax_load(lr, lane, v);
    softmax_reduce(v, lane, M, inv);
}
__device__ __forceinline__ float softmax_prob(float logit, float M, float inv) { return expf(logit - M) * inv; }

// value head: Linear(72 -> 1) + tanh over the value features hv[p*3 + c]; every lane returns the value.
// In two halves (loads / arithmetic) so that the search can request the operands early; both callers run the same code.
struct ValueHeadIn { float h0, w0, h1, w1, bias; };
__device__ __forceinline__ ValueHeadIn value_head_load(const float* __restrict__ hv_row, const float* __restrict__ wv, int lane) {
    ValueHeadIn q;
    q.h0 = hv_row[lane]; q.w0 = wv[lane];
    q.h1 = lane < 8 ? hv_row[64 + lane] : 0.0f; q.w1 = lane < 8 ? wv[64 + lane] : 0.0f;
    q.bias = wv[72];
    return q;
}
__device__ __forceinline__ float value_head_eval(const ValueHeadIn& q, int lane) {
    float dot = 0.0f;
    dot += q.h0 * q.w0;
    if (lane < 8) dot += q.h1 * q.w1;
    dot = wave_butterfly_sum(dot);
    return tanhf(dot + q.bias);
}
__device__ __forceinline__ float value_head(const float* __restrict__ hv_row, const float* __restrict__ wv, int lane) {
    return value_head_eval(value_head_load(hv_row, wv, lane), lane);
}

}  // namespace diee
