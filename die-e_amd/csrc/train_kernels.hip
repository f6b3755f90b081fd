// train_kernels.hip -- memory-bound kernels of the learn loop's training step in the NHWC token layout
// x[row = board*24 + point][256] bf16: BatchNorm in train mode fused with the residual add and the ReLU
// (ResBlock::forward_t, src/alphazero/nnet.rs:24-34, with `train = true`: batch statistics over batch x 4 x 6), its
// backward, and a column sum (the convolution's bias gradient).  PyTorch's own batch-norm kernels take 33 us per pass on
// this shape (a [6144, 256] bf16 matrix, 3 MB: profiles/r02_train_step_engine_kernel_stats*.csv); these take one
// 3 MB sweep each.  Reductions are two-stage (per-stripe partials, then every workgroup folds the partials in a fixed
// order): deterministic, no atomics.
#include <hip/hip_runtime.h>
#include <atomic>
#include <vector>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>

#include "launch.h"

namespace diee {

typedef __attribute__((ext_vector_type(4))) uint32_t tu32x4;

__device__ __forceinline__ float tbf2f(uint32_t b16) { return __uint_as_float(b16 << 16); }
__device__ __forceinline__ uint16_t tf2bf(float x) { const __bf16 b = (__bf16)x; return __builtin_bit_cast(uint16_t, b); }
__device__ __forceinline__ void unpack8(const tu32x4 v, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = tbf2f(v[j] & 0xffffu); f[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u); }
}
__device__ __forceinline__ tu32x4 pack8(const float (&f)[8]) {
    tu32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (uint32_t)tf2bf(f[2 * j]) | ((uint32_t)tf2bf(f[2 * j + 1]) << 16);
    return o;
}

constexpr int kStripe = 64;             // rows per workgroup: 96 workgroups at 256 boards
// thread t of a 256-thread workgroup: channels 8*(t & 31) .. +8, rows (t >> 5) + 8*k of the stripe

// partial[stripe][q][256]: q = 0 sum of a, 1 sum of a*b  (b == nullptr: a*a), optionally with a masked by mask > 0 and
// b normalised as (b - mean) * invstd
template <bool BWD>
__global__ __launch_bounds__(256) void k_col_partials(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                      const uint16_t* __restrict__ mask, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, float* __restrict__ partial, int M) {
    __shared__ float red[2][8][256];
    const int t = threadIdx.x, c8 = t & 31, r0 = t >> 5;
    const int row_lo = blockIdx.x * kStripe;
    float s0[8], s1[8], mu[8], is[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; mu[j] = 0.f; is[j] = 1.f; }
    if (BWD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { mu[j] = mean[c8 * 8 + j]; is[j] = invstd[c8 * 8 + j]; }
    }
    for (int r = row_lo + r0; r < row_lo + kStripe && r < M; r += 8) {
        float av[8], bv[8];
        unpack8(*(const tu32x4*)(a + (size_t)r * 256 + c8 * 8), av);
        if (BWD) {
            float mv[8];
            unpack8(*(const tu32x4*)(mask + (size_t)r * 256 + c8 * 8), mv);
            unpack8(*(const tu32x4*)(b + (size_t)r * 256 + c8 * 8), bv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dz = mv[j] > 0.f ? av[j] : 0.f;
                s0[j] += dz; s1[j] += dz * ((bv[j] - mu[j]) * is[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0[j] += av[j]; s1[j] += av[j] * av[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][r0][c8 * 8 + j] = s0[j]; red[1][r0][c8 * 8 + j] = s1[j]; }
    __syncthreads();
    float x0 = 0.f, x1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { x0 += red[0][k][t]; x1 += red[1][k][t]; }
    partial[((size_t)blockIdx.x * 2 + 0) * 256 + t] = x0;
    partial[((size_t)blockIdx.x * 2 + 1) * 256 + t] = x1;
}

// fold the stripes' partials: ONE workgroup of 1024 threads = 4 parts x 256 channels; a part folds its quarter of the
// stripes with two interleaved accumulators, the parts are combined through LDS in a fixed order (deterministic).
// Every thread returns the totals of channel (threadIdx.x & 255).
__device__ __forceinline__ void fold_partials(const float* __restrict__ partial, int stripes, float& q0, float& q1) {
    __shared__ float acc[2][4][256];
    const int c = threadIdx.x & 255, part = threadIdx.x >> 8;
    const int per = (stripes + 3) / 4, s0 = part * per, s1 = s0 + per < stripes ? s0 + per : stripes;
    float a[2] = {0.f, 0.f}, b[2] = {0.f, 0.f};
    int s = s0;
    for (; s + 2 <= s1; s += 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            a[k] += partial[((size_t)(s + k) * 2 + 0) * 256 + c];
            b[k] += partial[((size_t)(s + k) * 2 + 1) * 256 + c];
        }
    }
    if (s < s1) { a[0] += partial[((size_t)s * 2 + 0) * 256 + c]; b[0] += partial[((size_t)s * 2 + 1) * 256 + c]; }
    acc[0][part][c] = a[0] + a[1]; acc[1][part][c] = b[0] + b[1];
    __syncthreads();
    q0 = (acc[0][0][c] + acc[0][1][c]) + (acc[0][2][c] + acc[0][3][c]);
    q1 = (acc[1][0][c] + acc[1][1][c]) + (acc[1][2][c] + acc[1][3][c]);
}

// forward finalisation: mean / invstd, running statistics (momentum, unbiased variance, like torch.nn.BatchNorm2d.train()),
// and the affine map of the apply pass: coef[0][c] = gamma * invstd, coef[1][c] = beta - mean * gamma * invstd
__global__ __launch_bounds__(1024) void k_bn_stats(const float* __restrict__ partial, int stripes, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ save_mean,
                                                  float* __restrict__ save_invstd, float* __restrict__ run_mean,
                                                  float* __restrict__ run_var, float momentum, float eps, float* __restrict__ coef, int M) {
    const int t = threadIdx.x & 255;
    float s, ss;
    fold_partials(partial, stripes, s, ss);
    if (threadIdx.x >= 256) return;
    const float mean = s / (float)M;
    float var = ss / (float)M - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float is = 1.0f / sqrtf(var + eps);
    coef[t] = gamma[t] * is; coef[256 + t] = beta[t] - mean * gamma[t] * is;
    save_mean[t] = mean; save_invstd[t] = is;
    if (run_mean) {
        const float unb = M > 1 ? var * (float)M / (float)(M - 1) : var;
        run_mean[t] = (1.f - momentum) * run_mean[t] + momentum * mean;
        run_var[t] = (1.f - momentum) * run_var[t] + momentum * unb;
    }
}

// y = relu(x * coef0 + coef1 [+ res])
__global__ __launch_bounds__(256) void k_bn_relu_fwd(const uint16_t* __restrict__ x, const uint16_t* __restrict__ res,
                                                     const float* __restrict__ coef, uint16_t* __restrict__ y, int M) {
    const int t = threadIdx.x, c8 = t & 31, r0 = t >> 5;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = coef[c8 * 8 + j]; sh[j] = coef[256 + c8 * 8 + j]; }
    const int row_lo = blockIdx.x * kStripe;
    for (int r = row_lo + r0; r < row_lo + kStripe && r < M; r += 8) {
        float v[8], rv[8];
        unpack8(*(const tu32x4*)(x + (size_t)r * 256 + c8 * 8), v);
        if (res) unpack8(*(const tu32x4*)(res + (size_t)r * 256 + c8 * 8), rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = v[j] * sc[j] + sh[j];
            if (res) z += rv[j];
            v[j] = z > 0.f ? z : 0.f;
        }
        *(tu32x4*)(y + (size_t)r * 256 + c8 * 8) = pack8(v);
    }
}

// backward finalisation: dgamma = sum dz * xhat, dbeta = sum dz, and the coefficients of the apply pass:
// coef[0] = gamma * invstd, [1] = mean(dz), [2] = mean(dz * xhat), [3] = mean, [4] = invstd
__global__ __launch_bounds__(1024) void k_bn_bwd_stats(const float* __restrict__ partial, int stripes, const float* __restrict__ gamma,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef, int M) {
    const int t = threadIdx.x & 255;
    float sdz, sdzx;
    fold_partials(partial, stripes, sdz, sdzx);
    if (threadIdx.x >= 256) return;
    dgamma[t] = sdzx; dbeta[t] = sdz;
    coef[t] = gamma[t] * invstd[t]; coef[256 + t] = sdz / (float)M; coef[512 + t] = sdzx / (float)M;
    coef[768 + t] = mean[t]; coef[1024 + t] = invstd[t];
}

// dz = dy * (y > 0); dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)); dres = dz
__global__ __launch_bounds__(256) void k_bn_relu_bwd(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ y,
                                                     const uint16_t* __restrict__ x, const float* __restrict__ coef,
                                                     uint16_t* __restrict__ dx, uint16_t* __restrict__ dres, int M) {
    const int t = threadIdx.x, c8 = t & 31, r0 = t >> 5;
    float k0[8], k1[8], k2[8], mu[8], is[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;
        k0[j] = coef[c]; k1[j] = coef[256 + c]; k2[j] = coef[512 + c]; mu[j] = coef[768 + c]; is[j] = coef[1024 + c];
    }
    const int row_lo = blockIdx.x * kStripe;
    for (int r = row_lo + r0; r < row_lo + kStripe && r < M; r += 8) {
        float g[8], yv[8], xv[8], dz[8];
        unpack8(*(const tu32x4*)(dy + (size_t)r * 256 + c8 * 8), g);
        unpack8(*(const tu32x4*)(y + (size_t)r * 256 + c8 * 8), yv);
        unpack8(*(const tu32x4*)(x + (size_t)r * 256 + c8 * 8), xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dz[j] = yv[j] > 0.f ? g[j] : 0.f;
            const float xh = (xv[j] - mu[j]) * is[j];
            g[j] = k0[j] * (dz[j] - k1[j] - xh * k2[j]);
        }
        *(tu32x4*)(dx + (size_t)r * 256 + c8 * 8) = pack8(g);
        if (dres) *(tu32x4*)(dres + (size_t)r * 256 + c8 * 8) = pack8(dz);
    }
}

// out[c] = sum over rows of a[row][c] (fp32), from the partials of k_col_partials<false>
__global__ __launch_bounds__(1024) void k_colsum_final(const float* __restrict__ partial, int stripes, float* __restrict__ out) {
    float s, ss;
    fold_partials(partial, stripes, s, ss);
    if (threadIdx.x < 256) out[threadIdx.x] = s;
}

// ---- single-launch BatchNorm passes -------------------------------------------------------------------------------
// The three launches above move 3 MB each and run 5-8 us apiece, almost all of it launch latency.  When every stripe's
// workgroup can be resident at once (stripes <= resident workgroups: the launcher checks, 96 at batch 256) one launch does the
// whole pass: a workgroup of 1024 threads keeps its 64-row stripe in registers, publishes its partial sums with
// device-coherent stores, meets the other workgroups on a counter, folds ALL partials itself in the fixed order (every
// workgroup computes the same statistics, bit for bit), and applies them to the registers it still holds.
// The meeting is bounded: a workgroup that waits too long stops waiting and poisons its statistics with NaN, so a starved
// launch is loud in the loss instead of hanging the device.
#ifndef DIEE_BN_ABLATE
#define DIEE_BN_ABLATE 0      // dev builds: 1 = fold one stripe per part only, 2 = no meeting (both compute garbage)
#endif
constexpr int kBnSpinLimit = 1 << 18;               // x s_sleep(2) ~ 128 cycles: tens of milliseconds
#define DIEE_AGENT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void st_coh(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, DIEE_AGENT); }
__device__ __forceinline__ float ld_coh(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, DIEE_AGENT); }

// bar[0] arrivals (back to 0 when the last one arrives), bar[1] generation.  All threads call; returns false on a timeout.
__device__ __forceinline__ bool grid_meet(uint32_t* bar, unsigned n) {
    __shared__ int ok_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's coherent stores are acknowledged
    __syncthreads();
#if DIEE_BN_ABLATE == 2
    return true;
#endif
    if (threadIdx.x == 0) {
        int ok = 1;
        const uint32_t gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, DIEE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the generation is read before this arrival counts
        if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, DIEE_AGENT) == n - 1) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, DIEE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, DIEE_AGENT);
        } else {
            int spins = 0;
            while (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, DIEE_AGENT) == gen) {
                if (++spins > kBnSpinLimit) { ok = 0; __hip_atomic_fetch_or(bar + 3, 1u, __ATOMIC_RELAXED, DIEE_AGENT); break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

// sums over the stripe's rows of s0 / s1 (8 channels per thread, thread t: channels 8*(t & 31).., row slot t >> 5 of 32):
// the two half-waves meet by a cross-lane add, the 16 waves through LDS; threads 0..255 return the totals of channel t
__device__ __forceinline__ void stripe_reduce(float (&s0)[8], float (&s1)[8], float (*red)[16][256], float& x0, float& x1) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] += __shfl_xor(s0[j], 32); s1[j] += __shfl_xor(s1[j], 32); }
    __syncthreads();                                            // red may still be read by the previous reduction
    if (lane < 32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[0][wave][lane * 8 + j] = s0[j]; red[1][wave][lane * 8 + j] = s1[j]; }
    }
    __syncthreads();
    x0 = 0.f; x1 = 0.f;
    if (t < 256) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { x0 += red[0][k][t]; x1 += red[1][k][t]; }
    }
}

// fold_partials over device-coherent loads, for the launches whose partials were written by this very launch
__device__ __forceinline__ void fold_partials_coh(const float* partial, int stripes, int nq, float& q0, float& q1) {
    __shared__ float acc[2][4][256];
    const int c = threadIdx.x & 255, part = threadIdx.x >> 8;
    const int per = (stripes + 3) / 4, s0 = part * per;
    int s1 = s0 + per < stripes ? s0 + per : stripes;
    // 16 stripes' loads in flight at a time (a coherent load is a round trip past the L2), added in stripe order
    float a[2] = {0.f, 0.f}, b[2] = {0.f, 0.f};
#if DIEE_BN_ABLATE == 1
    if (s1 > s0 + 1) s1 = s0 + 1;
#endif
    for (int s = s0; s < s1; s += 16) {
        float va[16], vb[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int ss = s + k < s1 ? s + k : s1 - 1;
            va[k] = ld_coh(partial + ((size_t)ss * nq + 0) * 256 + c);
            vb[k] = nq == 2 ? ld_coh(partial + ((size_t)ss * nq + 1) * 256 + c) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (s + k < s1) { a[k & 1] += va[k]; b[k & 1] += vb[k]; }
    }
    __syncthreads();
    acc[0][part][c] = a[0] + a[1]; acc[1][part][c] = b[0] + b[1];
    __syncthreads();
    q0 = (acc[0][0][c] + acc[0][1][c]) + (acc[0][2][c] + acc[0][3][c]);
    q1 = (acc[1][0][c] + acc[1][1][c]) + (acc[1][2][c] + acc[1][3][c]);
}

__global__ __launch_bounds__(1024) void k_bn_fwd_coop(const uint16_t* __restrict__ x, const uint16_t* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* partial, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                     float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                     float eps, uint16_t* __restrict__ y, int M, uint32_t* bar) {
    __shared__ float red[2][16][256];
    __shared__ float coef[2][256];
    const int t = threadIdx.x, c8 = t & 31, rs = t >> 5, row_lo = blockIdx.x * kStripe, S = gridDim.x;
    tu32x4 xv[2], rv[2];
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = row_lo + rs + 32 * k;
        xv[k] = tu32x4{0u, 0u, 0u, 0u}; rv[k] = tu32x4{0u, 0u, 0u, 0u};
        if (r < M) {
            xv[k] = *(const tu32x4*)(x + (size_t)r * 256 + c8 * 8);
            if (res) rv[k] = *(const tu32x4*)(res + (size_t)r * 256 + c8 * 8);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float v[8];
        unpack8(xv[k], v);                                      // rows past M hold zeros: they add nothing
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0[j] += v[j]; s1[j] += v[j] * v[j]; }
    }
    float p0, p1;
    stripe_reduce(s0, s1, red, p0, p1);
    if (t < 256) { st_coh(partial + ((size_t)blockIdx.x * 2 + 0) * 256 + t, p0); st_coh(partial + ((size_t)blockIdx.x * 2 + 1) * 256 + t, p1); }
    const bool ok = grid_meet(bar, S);
    float sum, sq;
    fold_partials_coh(partial, S, 2, sum, sq);
    if (t < 256) {
        float mean = sum / (float)M;
        float var = sq / (float)M - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float is = 1.0f / sqrtf(var + eps);
        if (!ok) mean = __builtin_nanf("");
        coef[0][t] = gamma[t] * is; coef[1][t] = beta[t] - mean * gamma[t] * is;
        if (blockIdx.x == 0) {
            save_mean[t] = mean; save_invstd[t] = is;
            if (run_mean) {
                const float unb = M > 1 ? var * (float)M / (float)(M - 1) : var;
                run_mean[t] = (1.f - momentum) * run_mean[t] + momentum * mean;
                run_var[t] = (1.f - momentum) * run_var[t] + momentum * unb;
            }
        }
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = coef[0][c8 * 8 + j]; sh[j] = coef[1][c8 * 8 + j]; }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = row_lo + rs + 32 * k;
        if (r >= M) continue;
        float v[8], rr[8];
        unpack8(xv[k], v); unpack8(rv[k], rr);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = v[j] * sc[j] + sh[j];
            if (res) z += rr[j];
            v[j] = z > 0.f ? z : 0.f;
        }
        *(tu32x4*)(y + (size_t)r * 256 + c8 * 8) = pack8(v);
    }
}

// backward in one launch; with dx_colsum also the column sums of the (bf16-rounded) dx it writes -- the bias gradient of
// the convolution in front -- from a second round of partials that only workgroup 0 waits for (bar[2] counts them)
__global__ __launch_bounds__(1024) void k_bn_bwd_coop(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ y,
                                                     const uint16_t* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     float* partial, float* partial2, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, uint16_t* __restrict__ dx,
                                                     uint16_t* __restrict__ dres, float* __restrict__ dx_colsum, int M, uint32_t* bar) {
    __shared__ float red[2][16][256];
    __shared__ float coef[3][256];
    const int t = threadIdx.x, c8 = t & 31, rs = t >> 5, row_lo = blockIdx.x * kStripe, S = gridDim.x;
    tu32x4 gv[2], yv[2], xv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = row_lo + rs + 32 * k;
        gv[k] = tu32x4{0u, 0u, 0u, 0u}; yv[k] = gv[k]; xv[k] = gv[k];
        if (r < M) {
            gv[k] = *(const tu32x4*)(dy + (size_t)r * 256 + c8 * 8);
            yv[k] = *(const tu32x4*)(y + (size_t)r * 256 + c8 * 8);
            xv[k] = *(const tu32x4*)(x + (size_t)r * 256 + c8 * 8);
        }
    }
    float mu[8], is[8], s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = mean[c8 * 8 + j]; is[j] = invstd[c8 * 8 + j]; s0[j] = 0.f; s1[j] = 0.f; }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float g[8], yy[8], xx[8];
        unpack8(gv[k], g); unpack8(yv[k], yy); unpack8(xv[k], xx);      // rows past M: y = 0 masks them out
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dz = yy[j] > 0.f ? g[j] : 0.f;
            s0[j] += dz; s1[j] += dz * ((xx[j] - mu[j]) * is[j]);
        }
    }
    float p0, p1;
    stripe_reduce(s0, s1, red, p0, p1);
    if (t < 256) { st_coh(partial + ((size_t)blockIdx.x * 2 + 0) * 256 + t, p0); st_coh(partial + ((size_t)blockIdx.x * 2 + 1) * 256 + t, p1); }
    const bool ok = grid_meet(bar, S);
    float sdz, sdzx;
    fold_partials_coh(partial, S, 2, sdz, sdzx);
    if (t < 256) {
        if (!ok) sdz = __builtin_nanf("");
        if (blockIdx.x == 0) { dgamma[t] = sdzx; dbeta[t] = sdz; }
        coef[0][t] = gamma[t] * invstd[t]; coef[1][t] = sdz / (float)M; coef[2][t] = sdzx / (float)M;
    }
    __syncthreads();
    float k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { k0[j] = coef[0][c8 * 8 + j]; k1[j] = coef[1][c8 * 8 + j]; k2[j] = coef[2][c8 * 8 + j]; s0[j] = 0.f; s1[j] = 0.f; }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = row_lo + rs + 32 * k;
        if (r >= M) continue;
        float g[8], yy[8], xx[8], dz[8];
        unpack8(gv[k], g); unpack8(yv[k], yy); unpack8(xv[k], xx);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dz[j] = yy[j] > 0.f ? g[j] : 0.f;
            const float xh = (xx[j] - mu[j]) * is[j];
            g[j] = k0[j] * (dz[j] - k1[j] - xh * k2[j]);
        }
        const tu32x4 o = pack8(g);
        *(tu32x4*)(dx + (size_t)r * 256 + c8 * 8) = o;
        if (dres) *(tu32x4*)(dres + (size_t)r * 256 + c8 * 8) = pack8(dz);
        if (dx_colsum) {
            float q[8];
            unpack8(o, q);                                      // the sum of what was written, like a pass over dx would see it
#pragma unroll
            for (int j = 0; j < 8; ++j) s0[j] += q[j];
        }
    }
    if (!dx_colsum) return;
    stripe_reduce(s0, s1, red, p0, p1);
    if (t < 256) st_coh(partial2 + (size_t)blockIdx.x * 256 + t, p0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok2_s;
    if (t == 0) {
        __hip_atomic_fetch_add(bar + 2, 1u, __ATOMIC_RELAXED, DIEE_AGENT);
        int ok2 = 1;
        if (blockIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(bar + 2, __ATOMIC_RELAXED, DIEE_AGENT) != (uint32_t)S) {
                if (++spins > kBnSpinLimit) { ok2 = 0; __hip_atomic_fetch_or(bar + 3, 2u, __ATOMIC_RELAXED, DIEE_AGENT); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            __hip_atomic_store(bar + 2, 0u, __ATOMIC_RELAXED, DIEE_AGENT);
        }
        ok2_s = ok2;
    }
    if (blockIdx.x != 0) return;
    __syncthreads();
    float cs, unused;
    fold_partials_coh(partial2, S, 1, cs, unused);
    if (t < 256) dx_colsum[t] = ok2_s ? cs : __builtin_nanf("");
}

// Barrier words of the one-launch passes: one set of 16 words (64 B) PER (device, stream) -- two passes running at once on
// different streams of one device would otherwise mix their arrivals on shared words and release each other early on
// partials that are not written yet.  Word 3 of a set collects its timeouts (bit 0 forward, bit 1 backward): the host reads
// and clears them with bn_coop_poll_timeouts().  `resident`: how many 1024-thread workgroups the device holds at once.
namespace {
constexpr int kBnSets = 16, kBnSetWords = 32;         // 16 streams per device, one 128-byte line each
struct BnDev { uint32_t* pool = nullptr; hipStream_t owner[kBnSets]; int n = 0; int resident = 0; bool failed = false; };
BnDev g_bn_dev[16];
std::mutex g_bn_mu;                                   // two host threads may take their first training step together
std::atomic<int> g_bn_coop_on{-1};                    // -1: not decided yet (DIEE_BN_COOP), 0 off, 1 on
}
static uint32_t* bn_sync_words(hipStream_t st, int& resident) {
    std::lock_guard<std::mutex> lock(g_bn_mu);
    int dev = 0;
    resident = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    BnDev& D = g_bn_dev[dev];
    if (D.failed) return nullptr;
    if (!D.pool) {
        // the whole pool is allocated and zeroed at the first pass on the device (an eager step: the training loop warms up
        // before it captures), so that handing a set to a new stream later -- possibly inside a capture -- is pure host work
        uint32_t* p = nullptr;
        if (hipMalloc((void**)&p, sizeof(uint32_t) * kBnSets * kBnSetWords) != hipSuccess) { (void)hipGetLastError(); D.failed = true; return nullptr; }
        if (hipMemset(p, 0, sizeof(uint32_t) * kBnSets * kBnSetWords) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError(); (void)hipFree(p); D.failed = true; return nullptr;
        }
        int per_cu_f = 0, per_cu_b = 0, cus = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_f, (const void*)k_bn_fwd_coop, 1024, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, (const void*)k_bn_bwd_coop, 1024, 0);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        D.resident = (per_cu_f > 0 && per_cu_b > 0) ? cus : 0;     // one workgroup per CU is all the launcher counts on
        D.pool = p;
    }
    resident = D.resident;
    for (int i = 0; i < D.n; ++i) if (D.owner[i] == st) return D.pool + i * kBnSetWords;
    if (D.n >= kBnSets) return nullptr;                 // a caller cycling through streams: the three-launch passes take it
    D.owner[D.n] = st;
    return D.pool + (D.n++) * kBnSetWords;
}
static bool bn_coop_enabled() {
    int v = g_bn_coop_on.load();
    if (v < 0) {
        const char* e = getenv("DIEE_BN_COOP");
        v = !(e && e[0] == '0') ? 1 : 0;
        g_bn_coop_on.store(v);
    }
    return v != 0;
}
// runtime switch (diee_train_set_bn_coop): the one-launch passes count on having the device to themselves -- callers that
// share it (RCCL kernels of a DDP step, a second rank or process on the GPU) turn them off
void bn_coop_set(int on) { g_bn_coop_on.store(on ? 1 : 0); }
int bn_coop_get() { return bn_coop_enabled() ? 1 : 0; }
// timeouts of the one-launch passes on the current device since the last clear (bit 0 forward, bit 1 backward); blocks
// until the device is idle.  -1: the query itself failed.
int bn_coop_poll_timeouts(int clear) {
    std::lock_guard<std::mutex> lock(g_bn_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
    BnDev& D = g_bn_dev[dev];
    if (!D.pool) return 0;
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return -1; }
    uint32_t w[kBnSets * kBnSetWords];
    if (hipMemcpy(w, D.pool, sizeof w, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
    uint32_t all = 0;
    bool dirty = false;
    for (int i = 0; i < D.n; ++i) {
        const uint32_t* q = w + i * kBnSetWords;
        all |= q[3];
        dirty = dirty || q[3] || q[0] || q[2];          // a starved meeting leaves arrivals behind: re-arm the sets
    }
    if (clear && dirty) {
        if (hipMemset(D.pool, 0, sizeof w) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return -1; }
    }
    return (int)all;
}

int train_stripes(int M) { return (M + kStripe - 1) / kStripe; }

void launch_colsum(hipStream_t st, const uint16_t* a, float* partial, float* out, int M);
void launch_bn_relu_fwd(hipStream_t st, const uint16_t* x, const uint16_t* res, const float* gamma, const float* beta, float* partial,
                        float* save_mean, float* save_invstd, float* run_mean, float* run_var, float momentum, float eps,
                        uint16_t* y, int M) {
    const int S = train_stripes(M);
    float* coef = partial + (size_t)S * 768;
    int resident = 0;
    uint32_t* bar = bn_coop_enabled() ? bn_sync_words(st, resident) : nullptr;
    if (bar && S <= resident) {
        hipLaunchKernelGGL(k_bn_fwd_coop, dim3(S), dim3(1024), 0, st, x, res, gamma, beta, partial, save_mean, save_invstd, run_mean,
                           run_var, momentum, eps, y, M, bar);
        return;
    }
    hipLaunchKernelGGL((k_col_partials<false>), dim3(S), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, partial, M);
    hipLaunchKernelGGL(k_bn_stats, dim3(1), dim3(1024), 0, st, partial, S, gamma, beta, save_mean, save_invstd, run_mean, run_var,
                       momentum, eps, coef, M);
    hipLaunchKernelGGL(k_bn_relu_fwd, dim3(S), dim3(256), 0, st, x, res, coef, y, M);
}
void launch_bn_relu_bwd(hipStream_t st, const uint16_t* dy, const uint16_t* y, const uint16_t* x, const float* gamma,
                        const float* mean, const float* invstd, float* partial, float* dgamma, float* dbeta, uint16_t* dx,
                        uint16_t* dres, float* dx_colsum, int M) {
    const int S = train_stripes(M);
    float* coef = partial + (size_t)S * 768;
    int resident = 0;
    uint32_t* bar = bn_coop_enabled() ? bn_sync_words(st, resident) : nullptr;
    if (bar && S <= resident) {
        hipLaunchKernelGGL(k_bn_bwd_coop, dim3(S), dim3(1024), 0, st, dy, y, x, gamma, mean, invstd, partial, partial + (size_t)S * 512,
                           dgamma, dbeta, dx, dres, dx_colsum, M, bar);
        return;
    }
    hipLaunchKernelGGL((k_col_partials<true>), dim3(S), dim3(256), 0, st, dy, x, y, mean, invstd, partial, M);
    hipLaunchKernelGGL(k_bn_bwd_stats, dim3(1), dim3(1024), 0, st, partial, S, gamma, mean, invstd, dgamma, dbeta, coef, M);
    hipLaunchKernelGGL(k_bn_relu_bwd, dim3(S), dim3(256), 0, st, dy, y, x, coef, dx, dres, M);
    if (dx_colsum) launch_colsum(st, dx, partial, dx_colsum, M);
}
void launch_colsum(hipStream_t st, const uint16_t* a, float* partial, float* out, int M) {
    const int S = train_stripes(M);
    hipLaunchKernelGGL((k_col_partials<false>), dim3(S), dim3(256), 0, st, a, nullptr, nullptr, nullptr, nullptr, partial, M);
    hipLaunchKernelGGL(k_colsum_final, dim3(1), dim3(1024), 0, st, partial, S, out);
}


// ---- weight gradient of the 3x3 tower convolution -----------------------------------------------------------------
//   dW[n][c][t] = sum over rows r of x[r + shift_t][c] * dy[r][n]      (pairs whose x position leaves the board drop out)
// One contraction over the batch*24 token rows per tap: [256 x rows] x [rows x 256], nine times.  Both operands are
// contracted over their ROW index, i.e. MFMA wants them transposed -- gfx950's ds_read_b64_tr_b16 does that on the way
// out of LDS (a 16-lane group reads a block of 4 rows x 16 columns and every lane gets one column of it), so the token
// matrices are staged as they lie in memory ([row][channel], coalesced 16-byte copies) and a tap is nothing but a row
// offset of the x image; the positions a tap would take off the board are zeroed in the A operand with per-lane masks
// (period: 3 k-steps of 16 rows = 2 boards).  A workgroup owns a 64 x 64 block of (c, n) for all nine taps and a 1/16
// share of the rows (slices of 16 boards, staged one after the other); its 8 waves split the block's four 32 x 32 tiles
// and the taps (5 + 4).  fp32 partial blocks go to scratch and a second kernel folds the 16 shares in a fixed order
// (deterministic) into the OIHW gradient.  PyTorch's route (im2col + hipBLASLt [2304 x 6144] x [6144 x 256]) takes
// 9 + 37 us per layer at batch 256 and rounds the result to bf16.
typedef __attribute__((ext_vector_type(8))) __bf16 wbf16x8;
typedef __attribute__((ext_vector_type(16))) float wf32x16;
typedef short wv4s __attribute__((ext_vector_type(4)));
constexpr int kWgBoards = 16, kWgRows = kWgBoards * 24, kWgHalo = 8, kWgRS = 192;
#ifndef DIEE_WG_SPLIT
#define DIEE_WG_SPLIT 16
#endif
constexpr int kWgSplit = DIEE_WG_SPLIT;
#ifndef DIEE_WG_ABLATE
#define DIEE_WG_ABLATE 0
#endif
constexpr int kWgXBytes = (kWgRows + 2 * kWgHalo) * kWgRS, kWgYBytes = kWgRows * kWgRS;

// 8 consecutive rows (k) x 32 columns of an LDS image [row][64 columns], as the 32x32x16 MFMA operand of lane `lane`:
// element j of the result = image[row0 + 8*(lane>>5) + j][col0 + (lane & 31)]
__device__ __forceinline__ void tr_operand(const char* img, int row0, int col0, int lane, uint2& lo, uint2& hi) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const int off = (row0 + 8 * (lane >> 5) + q) * kWgRS + (col0 + 16 * ((lane >> 4) & 1) + 4 * p) * 2;
    const wv4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wv4s __attribute__((address_space(3)))*)(img + off));
    const wv4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wv4s __attribute__((address_space(3)))*)(img + off + 4 * kWgRS));
    lo = __builtin_bit_cast(uint2, a); hi = __builtin_bit_cast(uint2, b);
}

// 64-bit mask (4 x 16 bits) of the positions 4j .. 4j+3 whose neighbour at tap t is on the board: a compile-time table
// (computing the 30 masks of a lane with integer divisions cost ~800 instructions per wave, 2.3 us of the kernel)
constexpr unsigned long long tap_mask_bits(int t, int j) {
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    unsigned long long m = 0;
    for (int q = 0; q < 4; ++q) {
        const int pos = 4 * j + q, y = pos / 6, x = pos % 6;
        if (y + dy >= 0 && y + dy < 4 && x + dx >= 0 && x + dx < 6) m |= 0xFFFFull << (16 * q);
    }
    return m;
}
struct TapMaskTable {
    unsigned long long m[9][6];
    constexpr TapMaskTable() : m{} {
        for (int t = 0; t < 9; ++t)
            for (int j = 0; j < 6; ++j) m[t][j] = tap_mask_bits(t, j);
    }
};
__constant__ TapMaskTable kTapMasks{};
__device__ __forceinline__ uint2 tap_mask(int t, int j) {
    const unsigned long long m = kTapMasks.m[t][j];
    return make_uint2((uint32_t)m, (uint32_t)(m >> 32));
}

// the k-loop of one staged slice for a wave with NT taps (5 or 4): straight-line code per 3 k-steps, all transposed reads
// of the body issued ahead of its MFMAs
template <int NT>
__device__ __forceinline__ void wgrad_slice(const char* xs, const char* ys, int ct, int nt, int lane, const int (&shift)[5],
                                            const uint2 (&mk)[3][5][2], wf32x16 (&acc)[5]) {
    for (int ks3 = 0; ks3 < kWgRows / 16; ks3 += 3) {
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3) {
            const int row0 = (ks3 + k3) * 16;
            uint2 blo, bhi;
            tr_operand(ys, row0, nt * 32, lane, blo, bhi);
            const wbf16x8 b = __builtin_bit_cast(wbf16x8, make_uint4(blo.x, blo.y, bhi.x, bhi.y));
            uint2 alo[NT], ahi[NT];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) tr_operand(xs, kWgHalo + row0 + shift[ti], ct * 32, lane, alo[ti], ahi[ti]);
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                alo[ti].x &= mk[k3][ti][0].x; alo[ti].y &= mk[k3][ti][0].y; ahi[ti].x &= mk[k3][ti][1].x; ahi[ti].y &= mk[k3][ti][1].y;
                const wbf16x8 a = __builtin_bit_cast(wbf16x8, make_uint4(alo[ti].x, alo[ti].y, ahi[ti].x, ahi[ti].y));
                acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ti], 0, 0, 0);
            }
        }
    }
}

__global__ __launch_bounds__(512) void k_wgrad3x3(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                  float* __restrict__ partial /*[kWgSplit][9][256][256]*/, int M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;                       // rows -kWgHalo .. kWgRows + kWgHalo of the slice, 64 channels of this block
    char* ys = smem + kWgXBytes;           // rows 0 .. kWgRows, 64 output channels of this block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = blockIdx.x & 3, nb = blockIdx.x >> 2, sg = blockIdx.y;
    const int ct = wave & 1, nt = (wave >> 1) & 1, t0 = (wave >> 2) * 5, ntaps = (wave >> 2) ? 4 : 5;
    // per-lane masks: this lane's two 4-row chunks of a k-step sit at positions 4*j0, 4*j0 + 4 with j0 = (4*ks + 2*h) mod 6
    uint2 mk[3][5][2];
    const int h = lane >> 5;
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3)
#pragma unroll
        for (int ti = 0; ti < 5; ++ti) {
            const int t = t0 + ti < 9 ? t0 + ti : 8;
            const int j0 = (4 * k3 + 2 * h) % 6;
            mk[k3][ti][0] = tap_mask(t, j0); mk[k3][ti][1] = tap_mask(t, (j0 + 1) % 6);
        }
    int shift[5];
#pragma unroll
    for (int ti = 0; ti < 5; ++ti) { const int t = t0 + ti < 9 ? t0 + ti : 8; shift[ti] = 6 * (t / 3 - 1) + (t % 3 - 1); }
    wf32x16 acc[5];
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ti][i] = 0.0f;

    const int n_slices = (M + kWgRows - 1) / kWgRows;
    for (int sl = sg; sl < n_slices; sl += kWgSplit) {
        const int row_base = sl * kWgRows;
        __syncthreads();                                        // the previous slice's reads are done
        // all global loads of the slice in flight before the first LDS store (7 + 6 per thread)
        constexpr int kXv = ((kWgRows + 2 * kWgHalo) * 8 + 511) / 512, kYv = kWgRows * 8 / 512;
        tu32x4 vx[kXv], vy[kYv];
#pragma unroll
        for (int k = 0; k < kXv; ++k) {
            const int i = tid + k * 512, r = i >> 3, ch = i & 7, gr = row_base + r - kWgHalo;
            vx[k] = tu32x4{0u, 0u, 0u, 0u};
            if (DIEE_WG_ABLATE != 3 && r < kWgRows + 2 * kWgHalo && gr >= 0 && gr < M) vx[k] = *(const tu32x4*)(x + (size_t)gr * 256 + cb * 64 + ch * 8);
        }
#pragma unroll
        for (int k = 0; k < kYv; ++k) {
            const int i = tid + k * 512, r = i >> 3, ch = i & 7, gr = row_base + r;
            vy[k] = tu32x4{0u, 0u, 0u, 0u};
            if (DIEE_WG_ABLATE != 3 && gr < M) vy[k] = *(const tu32x4*)(dy + (size_t)gr * 256 + nb * 64 + ch * 8);
        }
#pragma unroll
        for (int k = 0; k < kXv; ++k) {
            const int i = tid + k * 512, r = i >> 3, ch = i & 7;
            if (r < kWgRows + 2 * kWgHalo) *(tu32x4*)(xs + r * kWgRS + ch * 16) = vx[k];
        }
#pragma unroll
        for (int k = 0; k < kYv; ++k) {
            const int i = tid + k * 512, r = i >> 3, ch = i & 7;
            *(tu32x4*)(ys + r * kWgRS + ch * 16) = vy[k];
        }
        __syncthreads();
#if DIEE_WG_ABLATE != 2
        if (ntaps == 5) wgrad_slice<5>(xs, ys, ct, nt, lane, shift, mk, acc);
        else wgrad_slice<4>(xs, ys, ct, nt, lane, shift, mk, acc);
#endif
    }
    // C/D layout of 32x32: column (n) = lane & 31, rows (c) = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
#if DIEE_WG_ABLATE == 1
    if (M != 12345) return;
#endif
#pragma unroll
    for (int ti = 0; ti < 5; ++ti) {
        if (ti >= ntaps) break;
        float* pt = partial + (((size_t)sg * 9 + (t0 + ti)) * 256 + cb * 64 + ct * 32) * 256 + nb * 64 + nt * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) pt[(size_t)((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * 256] = acc[ti][i];
    }
}

// dW[n][c][t] (OIHW, fp32) = the row shares of partial[share][t][c][n] added in share order
// a block folds 32 n x 8 c: coalesced 128-byte reads of the shares, an LDS transpose, then 288-byte contiguous runs of OIHW
__global__ __launch_bounds__(256) void k_wgrad_fold(const float* __restrict__ partial, float* __restrict__ dw) {
    __shared__ float tile[32 * 72];
    const int tid = threadIdx.x, nl = tid & 31, cl = tid >> 5, n0 = blockIdx.x * 32, c0 = blockIdx.y * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < kWgSplit; ++g) a += partial[(((size_t)g * 9 + t) * 256 + c0 + cl) * 256 + n0 + nl];
        tile[nl * 72 + cl * 9 + t] = a;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int i = tid + k * 256, nn = i / 72, off = i - nn * 72;
        dw[((size_t)(n0 + nn) * 256 + c0) * 9 + off] = tile[i];
    }
}

size_t wgrad_scratch_floats() { return (size_t)kWgSplit * 9 * 256 * 256; }
void launch_wgrad3x3(hipStream_t st, const uint16_t* x, const uint16_t* dy, float* partial, float* dw, int boards) {
    static bool attr_set_dev[16] = {};                       // (idempotent: a race sets the same attribute twice)
    int attr_dev = 0;
    (void)hipGetDevice(&attr_dev);
    bool& attr_set = attr_set_dev[attr_dev & 15];             // per device: a ctx on another GPU of this process sets it there too
    constexpr int lds = kWgXBytes + kWgYBytes;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_wgrad3x3, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
    hipLaunchKernelGGL(k_wgrad3x3, dim3(16, kWgSplit), dim3(512), lds, st, x, dy, partial, boards * 24);
    hipLaunchKernelGGL(k_wgrad_fold, dim3(8, 32), dim3(256), 0, st, partial, dw);
}

}  // namespace diee
