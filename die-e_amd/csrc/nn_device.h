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

// ---- policy FC 768 -> 1352 (nnet.rs:80-85): one wave per 32 games x 32 outputs, operands straight from L2 (the layer is
// ~0.2 % of the network's FLOPs).  A device function (a tile per wave); k_policy_fc is its launch.
typedef __attribute__((ext_vector_type(8))) __bf16 fc_bf16x8;
typedef __attribute__((ext_vector_type(16))) float fc_f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t fc_u32x4;
__device__ __forceinline__ void policy_fc_tile(const uint16_t* __restrict__ hp,    // [G][768] bf16, k' = p*32+c
                                               const fc_u32x4* __restrict__ wpack, // [43][48][64] x 16 B
                                               const float* __restrict__ bias,     // [1376]
                                               float* __restrict__ logits,         // [G][1352]
                                               int G, int g0, int nslice, int lane) {
    int row = g0 + (lane & 31);
    const bool rok = row < G;
    if (!rok) row = G - 1;
    const uint16_t* ap = hp + (size_t)row * 768 + (lane >> 5) * 8;
    const fc_u32x4* wp = wpack + (size_t)nslice * 48 * 64 + lane;
    fc_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // the layer is latency-bound (48 dependent-free k-steps, operands straight from L2): request 24 k-steps of
    // both operands up front, then issue their MFMAs, twice
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        fc_bf16x8 av[24];
        fc_u32x4 bvq[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            av[i] = *(const fc_bf16x8*)(ap + (half * 24 + i) * 16);
            bvq[i] = wp[(half * 24 + i) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);          // keep all 48 loads ahead of the first MFMA
#pragma unroll
        for (int i = 0; i < 24; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], __builtin_bit_cast(fc_bf16x8, bvq[i]), acc, 0, 0, 0);
    }
    const int n = nslice * 32 + (lane & 31);
    if (n >= 1352) return;
    const float bv = bias[n];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int g = g0 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        if (g < G) logits[(size_t)g * 1352 + n] = acc[i] + bv;
    }
}

}  // namespace diee
