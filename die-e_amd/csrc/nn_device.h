// nn_device.h -- the tail of the network shared by the API path (k_softmax_value) and the search (k_expand):
// softmax over the 1352 policy logits and the value head's FC + tanh (nnet.rs:83-85, 95-98).  Both callers run these
// functions on one wave per board, so the search reads the SAME bits the API returns (the oracle-based parity tests
// rely on that) without a softmax launch and a [boards][1352] policy round trip per evaluation.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace diee {

// M = max logit, inv = 1 / sum exp(logit - M) of one board's 1352 logits (22 per lane, butterfly reductions)
__device__ __forceinline__ void softmax_consts(const float* __restrict__ lr, int lane, float& M, float& inv) {
    float v[22];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 22; ++j) {
        const int a = lane + 64 * j;
        v[j] = a < 1352 ? lr[a] : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 22; ++j) sum += lane + 64 * j < 1352 ? expf(v[j] - mx) : 0.0f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    M = mx; inv = 1.0f / sum;
}
__device__ __forceinline__ float softmax_prob(float logit, float M, float inv) { return expf(logit - M) * inv; }

// value head: Linear(72 -> 1) + tanh over the value features hv[p*3 + c]; every lane returns the value
__device__ __forceinline__ float value_head(const float* __restrict__ hv_row, const float* __restrict__ wv, int lane) {
    float dot = 0.0f;
    for (int k = lane; k < 72; k += 64) dot += hv_row[k] * wv[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d);
    return tanhf(dot + wv[72]);
}

}  // namespace diee
