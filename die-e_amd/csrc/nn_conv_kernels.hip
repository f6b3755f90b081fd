// nn_conv_kernels.hip -- policy/value ResNet inference (reference: src/alphazero/nnet.rs:24-34,57-107,
// 120-155, eval-mode BatchNorm folded) as hand-written bf16 MFMA kernels for gfx950.
//
// conv3x3 (the hot kernel): implicit GEMM  out[M=G*24][N] = sum over 9 taps of shift_t(act)[M][C] x W_t[C][N].
//   * one workgroup = kGames whole boards (192 rows) x 128 output channels, 4 waves, one per SIMD;
//   * the activation tile (all C_IN channels of the 8 boards) is staged ONCE into LDS (padded rows:
//     conflict-free ds_read_b128) and stays stationary: the 9 taps re-read it at row offsets, board
//     borders are redirected to a zero row (per-lane addresses precomputed, no masking in the loop);
//   * the weights are pre-packed on the host in MFMA B-fragment order and streamed straight from
//     L2 into registers (1 KiB coalesced per wave-instruction), prefetched one channel-step ahead;
//     no barrier inside the K loop;
//   * v_mfma_f32_32x32x16_bf16, 6 M-fragments x 1 N-fragment per wave: 96 accumulator registers.
#include "nn_common.h"

namespace diee {

// NN input: as_tensor planes (backgammon_logic.rs:198-252) as bf16 NHWC rows [g*24+p][16] (6 real
// channels, 10 zero): small integers, exact in bf16.
__global__ void k_planes_bf16(const BgState* __restrict__ states, uint32_t n, uint16_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (state, point)
    if (i >= n * 24u) return;
    const BgState s = states[i / 24u];
    const int p = (int)(i % 24u);
    uint32_t w[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t lo = 2 * c < 6 ? f2bf(bg_plane_dev(s, 2 * c, p)) : 0u;
        const uint32_t hi = 2 * c + 1 < 6 ? f2bf(bg_plane_dev(s, 2 * c + 1, p)) : 0u;
        w[c] = lo | (hi << 16);
    }
    u32x4* o = (u32x4*)(out + (size_t)i * 16);
    o[0] = u32x4{w[0], w[1], w[2], w[3]};
    o[1] = u32x4{w[4], w[5], w[6], w[7]};
}

// MODE 0: out = relu(conv + bias)          (init block, ResBlock conv1; nnet.rs:26-28, 64-67)
// MODE 1: out = relu(conv + bias + res)    (ResBlock conv2 + skip;      nnet.rs:29-33)
// MODE 2: heads: channels 0..31 -> policy features bf16 [g][p*32+c], 32..34 -> value features f32
//         [g][p*3+c], both after ReLU       (nnet.rs:75-79, 87-91)
// GT = boards per workgroup (rows = 24*GT, padded to MF fragments of 32), NW = waves (32 channels each).
template <int C_IN, int MODE, int GT, int NW>
__global__ __launch_bounds__(64 * NW) void k_conv3x3(const uint16_t* __restrict__ act,      // [M][C_IN] bf16
                                                     const u32x4* __restrict__ wpack,      // [N/32][KSTEPS][64] x 16 B
                                                     const float* __restrict__ bias,       // [N]
                                                     const uint16_t* __restrict__ res,     // [M][N] bf16 (MODE 1)
                                                     uint16_t* __restrict__ out,           // [M][N] bf16 / policy feats
                                                     float* __restrict__ out_v,            // value feats (MODE 2)
                                                     int M, int N) {
    constexpr int ROWS = GT * 24;
    constexpr int MF = (ROWS + 31) / 32;
    constexpr int NT = 64 * NW;
    constexpr int RS = C_IN * 2 + 16;              // LDS row stride (bytes): +16 B pad => conflict-free b128
    constexpr int CPR = C_IN * 2 / 16;             // 16-B chunks per row
    constexpr int CSTEPS = C_IN / 16;              // channel steps of 16
    constexpr int KSTEPS = CSTEPS * 9;
    constexpr int UNR = CSTEPS >= 2 ? 2 : 1;       // channel steps per loop body (18 / 9 MFMA k-steps)
    constexpr int NF = 1, PD = 1;                  // one 32-channel N-fragment per wave; LDS reads one k-step ahead
    constexpr int NC = NW * NF * 32;               // output channels per workgroup
    constexpr int ORS = NC * 4 + 16;               // epilogue tile row stride (bytes)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS;
    const int nslice = (blockIdx.y * NW + wave) * NF;   // NF x 32 output channels per wave

    // weights: issue the first 9 fragment loads before touching the activation tile
    const u32x4* wp = wpack + (size_t)nslice * KSTEPS * 64 + lane;
    u32x4 bq[9][NF];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < NF; ++q) bq[t][q] = wp[((size_t)q * KSTEPS + t) * 64];

    // ---- stage the activation tile (whole boards, all input channels) ----
    for (int i = tid; i < ROWS * CPR; i += NT) {
        const int r = i / CPR, ch = i % CPR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (C_IN == 16) {
            // init block: `act` is the BgState array; build the as_tensor planes (backgammon_logic.rs:198-252)
            // on the fly as bf16 (6 real channels, small integers: exact)
            if (row0 + r < M && ch == 0) {
                const BgState st = *((const BgState*)act + (row0 + r) / 24);
                const int p = (row0 + r) % 24;
                uint32_t w[3];
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    w[c] = (uint32_t)f2bf(bg_plane_dev(st, 2 * c, p)) | ((uint32_t)f2bf(bg_plane_dev(st, 2 * c + 1, p)) << 16);
                v = u32x4{w[0], w[1], w[2], 0u};
            }
        } else if (row0 + r < M) {
            v = *(const u32x4*)(act + (size_t)(row0 + r) * C_IN + ch * 8);
        }
        *(u32x4*)(smem + r * RS + ch * 16) = v;
    }
    for (int i = tid; i < CPR + 3; i += NT) *(u32x4*)(smem + ROWS * RS + i * 16) = u32x4{0u, 0u, 0u, 0u};   // zero row
    // per-lane LDS byte addresses of the A fragments: [tap][M-fragment]
    int base[9][MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int R = 32 * f + (lane & 31);
        const int p = R % 24, y = p / 6, x = p % 6;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool ok = R < ROWS && (unsigned)(y + dy) < 4u && (unsigned)(x + dx) < 6u;
            const int src = ok ? R + 6 * dy + dx : ROWS;
            base[t][f] = src * RS + (lane >> 5) * 16;
        }
    }
    __syncthreads();

    f32x16 acc[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int q = 0; q < NF; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][q][i] = 0.0f;

    // software pipeline: A fragments of k-step s+PD are read from LDS while the MFMAs of k-step s
    // issue (a distance of 2 k-steps measured no faster); the weight fragment of k-step s+9 is requested
    // from L2 at k-step s.
    constexpr int NB = PD + 1;                     // A-fragment ring
    static_assert(CSTEPS / UNR == 1 || (9 * UNR) % NB == 0, "ring index must be static");
    bf16x8 a[NB][MF];
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int f = 0; f < MF; ++f) a[d][f] = *(const bf16x8*)(smem + base[d % 9][f] + (d / 9) * 32);
    for (int it = 0; it < CSTEPS / UNR; ++it) {
#pragma unroll
        for (int u = 0; u < 9 * UNR; ++u) {
            const int t = u % 9, cur = u % NB, nxt = (u + PD) % NB;
            const int un = u + PD;                                 // k-step to prefetch (inside / after this body)
            const int tn = un % 9;
            const int csn = it * UNR + un / 9;                     // may run past CSTEPS at the very end: reads padding, unused
#pragma unroll
            for (int f = 0; f < MF; ++f) a[nxt][f] = *(const bf16x8*)(smem + base[tn][f] + csn * 32);
            bf16x8 b[NF];
#pragma unroll
            for (int q = 0; q < NF; ++q) b[q] = __builtin_bit_cast(bf16x8, bq[t][q]);
            {
                const int cs_pf = it * UNR + u / 9 + 1;            // same tap, next channel step
#pragma unroll
                for (int q = 0; q < NF; ++q)
                    bq[t][q] = wp[((size_t)q * KSTEPS + (cs_pf < CSTEPS ? cs_pf : CSTEPS - 1) * 9 + t) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int q = 0; q < NF; ++q)
                    acc[f][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][f], b[q], acc[f][q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: accumulators -> LDS tile [ROWS][NW*32] f32 -> 16-byte coalesced global stores ----
    __syncthreads();                               // every wave is done reading the activation tile
    {
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            const float bv = bias[(nslice + q) * 32 + (lane & 31)];
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int i = 0; i < 16; ++i) {     // C/D layout: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
                    const int r = 32 * f + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    if (r < ROWS) *(float*)(smem + r * ORS + ((wave * NF + q) * 32 + (lane & 31)) * 4) = acc[f][q][i] + bv;
                }
        }
    }
    __syncthreads();
    constexpr int CPRW = NC / 8;                   // 8-channel chunks per row
    constexpr int CHUNKS = ROWS * CPRW;
    const int nbase = blockIdx.y * NC;
    for (int i = tid; i < CHUNKS; i += NT) {
        const int r = i / CPRW, c8 = i % CPRW;
        const int gr = row0 + r;
        if (gr >= M) continue;
        const float4 lo = *(const float4*)(smem + r * ORS + c8 * 32);
        const float4 hi = *(const float4*)(smem + r * ORS + c8 * 32 + 16);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (MODE == 1) {
            const u32x4 rv = *(const u32x4*)(res + (size_t)gr * N + nbase + c8 * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] += __uint_as_float(rv[j] << 16);
                v[2 * j + 1] += __uint_as_float(rv[j] & 0xffff0000u);
            }
        }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // MODE 3 (training: the raw convolution, BatchNorm follows in train mode) keeps the sign
            const float x0 = (MODE == 3 || v[2 * j] > 0.0f) ? v[2 * j] : 0.0f, x1 = (MODE == 3 || v[2 * j + 1] > 0.0f) ? v[2 * j + 1] : 0.0f;
            o[j] = (uint32_t)f2bf(x0) | ((uint32_t)f2bf(x1) << 16);
            v[2 * j] = x0; v[2 * j + 1] = x1;
        }
        if (MODE == 2) {
            const int g = gr / 24, p = gr % 24, n0 = nbase + c8 * 8;
            if (n0 < 32) *(u32x4*)(out + (size_t)g * 768 + p * 32 + n0) = o;
            else if (n0 == 32) { float* ov = out_v + (size_t)g * 72 + p * 3; ov[0] = v[0]; ov[1] = v[1]; ov[2] = v[2]; }
        } else {
            *(u32x4*)(out + (size_t)gr * N + nbase + c8 * 8) = o;
        }
    }
}

// Small-batch variant of the tower conv: when few games are alive the layer is latency-bound, so the
// K dimension (16 channel steps) is split over the 4 waves of a workgroup (split-K inside the
// workgroup, partial tiles reduced through LDS) and a workgroup owns only GT boards x 32 channels:
// 8x more workgroups than the large-batch geometry, 36 instead of 144 dependent k-steps per wave,
// and each wave requests its whole 36 KiB weight stream up front.
template <int MODE, int GT, int NSPLIT = 4>
__global__ __launch_bounds__(64 * NSPLIT) void k_conv3x3_sk(const uint16_t* __restrict__ act,       // [M][256] bf16
                                                    const u32x4* __restrict__ wpack,       // [N/32][144][64] x 16 B
                                                    const float* __restrict__ bias,
                                                    const uint16_t* __restrict__ res,
                                                    uint16_t* __restrict__ out, float* __restrict__ out_v,
                                                    int M, int N) {
    constexpr int C_IN = 256, ROWS = GT * 24, MF = (ROWS + 31) / 32, RS = C_IN * 2 + 16, CPR = 32;
    constexpr int NT = 64 * NSPLIT;
    constexpr int KS = 144 / NSPLIT;                // k-steps per wave: (16 / NSPLIT) channel steps x 9 taps
    constexpr int PF = GT <= 2 ? KS : 18;           // weight fragments in flight per wave
    constexpr int PRS = 32 * 4 + 16;                // partial-tile row stride (bytes)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* part = smem;                              // [4 waves][MF*32 rows][32] f32, aliases the activation tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS;
    const int nslice = blockIdx.y;

    const u32x4* wp = wpack + ((size_t)nslice * 144 + wave * KS) * 64 + lane;
    u32x4 bq[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) bq[i] = wp[i * 64];

    for (int i = tid; i < ROWS * CPR; i += NT) {
        const int r = i / CPR, ch = i % CPR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + r < M) v = *(const u32x4*)(act + (size_t)(row0 + r) * C_IN + ch * 8);
        *(u32x4*)(smem + r * RS + ch * 16) = v;
    }
    for (int i = tid; i < CPR + 3; i += NT) *(u32x4*)(smem + ROWS * RS + i * 16) = u32x4{0u, 0u, 0u, 0u};
    int base[9][MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int R = 32 * f + (lane & 31);
        const int p = R % 24, y = p / 6, x = p % 6;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool ok = R < ROWS && (unsigned)(y + dy) < 4u && (unsigned)(x + dx) < 6u;
            const int src = ok ? R + 6 * dy + dx : ROWS;
            base[t][f] = src * RS + (lane >> 5) * 16 + wave * (16 / NSPLIT) * 32;     // this wave's share of the channels
        }
    }
    __syncthreads();

    f32x16 acc[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[f][i] = 0.0f;
    bf16x8 a[2][MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) a[0][f] = *(const bf16x8*)(smem + base[0][f]);
#pragma unroll
    for (int u = 0; u < KS; ++u) {
        const int t = u % 9, cur = u & 1, nxt = cur ^ 1, un = u + 1;
#pragma unroll
        for (int f = 0; f < MF; ++f) a[nxt][f] = *(const bf16x8*)(smem + base[un % 9][f] + (un / 9) * 32);
        const bf16x8 b = __builtin_bit_cast(bf16x8, bq[u % PF]);
        if (u + PF < KS) bq[u % PF] = wp[(u + PF) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < MF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][f], b, acc[f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        (void)t;
    }

    // partial tiles -> LDS, then all 256 threads reduce the 4 partials and run the epilogue
    __syncthreads();                                // every wave is done reading the activation tile
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = 32 * f + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            *(float*)(part + ((wave * MF * 32 + r) * PRS) + (lane & 31) * 4) = acc[f][i];
        }
    __syncthreads();
    for (int i = tid; i < ROWS * 4; i += NT) {
        const int r = i >> 2, c8 = i & 3;
        const int gr = row0 + r;
        if (gr >= M) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bias[nslice * 32 + c8 * 8 + j];
#pragma unroll
        for (int w = 0; w < NSPLIT; ++w) {
            const float4 lo = *(const float4*)(part + (w * MF * 32 + r) * PRS + c8 * 32);
            const float4 hi = *(const float4*)(part + (w * MF * 32 + r) * PRS + c8 * 32 + 16);
            v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
        }
        if (MODE == 1) {
            const u32x4 rv = *(const u32x4*)(res + (size_t)gr * N + nslice * 32 + c8 * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] += __uint_as_float(rv[j] << 16);
                v[2 * j + 1] += __uint_as_float(rv[j] & 0xffff0000u);
            }
        }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = (MODE == 3 || v[2 * j] > 0.0f) ? v[2 * j] : 0.0f, x1 = (MODE == 3 || v[2 * j + 1] > 0.0f) ? v[2 * j + 1] : 0.0f;
            o[j] = (uint32_t)f2bf(x0) | ((uint32_t)f2bf(x1) << 16);
            v[2 * j] = x0; v[2 * j + 1] = x1;
        }
        if (MODE == 2) {                            // heads: slice 0 = policy features, slice 1 = value features
            const int g = gr / 24, p = gr % 24;
            if (nslice == 0) *(u32x4*)(out + (size_t)g * 768 + p * 32 + c8 * 8) = o;
            else if (c8 == 0) { float* ov = out_v + (size_t)g * 72 + p * 3; ov[0] = v[0]; ov[1] = v[1]; ov[2] = v[2]; }
        } else {
            *(u32x4*)(out + (size_t)gr * N + nslice * 32 + c8 * 8) = o;
        }
    }
}

// policy FC 768 -> 1352 (nnet.rs:80-85) as a launch of its own: policy_fc_tile (nn_device.h), one wave per 32 games x 32 outputs
__global__ __launch_bounds__(64) void k_policy_fc(const uint16_t* __restrict__ hp,    // [G][768] bf16, k' = p*32+c
                                                  const u32x4* __restrict__ wpack,   // [43][48][64] x 16 B
                                                  const float* __restrict__ bias,    // [1376]
                                                  float* __restrict__ logits,        // [G][1352]
                                                  int G, const uint32_t* __restrict__ n_rows /* non-null: the rows of a compacted batch */) {
    const int g0 = blockIdx.x * 32;
    if (n_rows) { G = (int)*n_rows; if (g0 >= G) return; }
    policy_fc_tile(hp, (const fc_u32x4*)wpack, bias, logits, G, g0, blockIdx.y, threadIdx.x);
}

__global__ __launch_bounds__(64) void k_softmax_value(const float* __restrict__ logits, const float* __restrict__ hv,
                                                      const float* __restrict__ wv /* [72] + bias */,
                                                      float* __restrict__ policy, float* __restrict__ value, int G) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    const float* lr = logits + (size_t)g * 1352;
    float M, inv;
    softmax_consts(lr, lane, M, inv);
    for (int a = lane; a < 1352; a += 64) policy[(size_t)g * 1352 + a] = softmax_prob(lr[a], M, inv);
    const float v = value_head(hv + (size_t)g * 72, wv, lane);
    if (lane == 0) value[g] = v;
}

// ---- host launchers -----------------------------------------------------------------------------
template <int C_IN, int GT, int NC>
static constexpr int conv_lds_bytes() {
    constexpr int rows = GT * 24;
    constexpr int a = (rows + 1) * (C_IN * 2 + 16) + 16 * 34 + 128;    // tile + zero row + over-read slack
    constexpr int o = rows * (NC * 4 + 16);
    return a > o ? a : o;
}

template <int C_IN, int MODE, int GT, int NW>
static void conv_launch(hipStream_t st, const uint16_t* act, const void* wpack, const float* bias, const uint16_t* res,
                        uint16_t* out, float* out_v, int G, int N) {
    static bool attr_set_dev[16] = {};
    int attr_dev = 0;
    (void)hipGetDevice(&attr_dev);
    bool& attr_set = attr_set_dev[attr_dev & 15];             // per device: a ctx on another GPU of this process sets it there too
    constexpr int lds = conv_lds_bytes<C_IN, GT, NW * 32>();
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_conv3x3<C_IN, MODE, GT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const dim3 grid((G + GT - 1) / GT, N / (32 * NW)), block(64 * NW);
    hipLaunchKernelGGL((k_conv3x3<C_IN, MODE, GT, NW>), grid, block, lds, st, act, (const u32x4*)wpack, bias, res, out,
                       out_v, G * 24, N);
}

template <int MODE, int GT, int NSPLIT = 4>
static void conv_sk_launch(hipStream_t st, const uint16_t* act, const void* wpack, const float* bias, const uint16_t* res,
                           uint16_t* out, int G, int N, float* out_v = nullptr) {
    static bool attr_set_dev[16] = {};
    int attr_dev = 0;
    (void)hipGetDevice(&attr_dev);
    bool& attr_set = attr_set_dev[attr_dev & 15];             // per device: a ctx on another GPU of this process sets it there too
    constexpr int rows = GT * 24, mf = (rows + 31) / 32;
    constexpr int lds_a = (rows + 1) * 528 + 16 * 34 + 64, lds_p = NSPLIT * mf * 32 * (32 * 4 + 16);
    constexpr int lds = lds_a > lds_p ? lds_a : lds_p;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_conv3x3_sk<MODE, GT, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_conv3x3_sk<MODE, GT, NSPLIT>), dim3((G + GT - 1) / GT, N / 32), dim3(64 * NSPLIT), lds, st, act,
                       (const u32x4*)wpack, bias, res, out, out_v, G * 24, N);
}

unsigned long long* g_tower_dbg = nullptr;     // diagnostic builds: per-workgroup clock stamps
void nn_set_tower_dbg(unsigned long long* p) { g_tower_dbg = p; }

void nn_setup_kernels() {}

void launch_planes_bf16(hipStream_t st, const void* states, uint32_t n, uint16_t* out) {
    if (!n) return;
    hipLaunchKernelGGL(k_planes_bf16, dim3((n * 24 + 255) / 256), dim3(256), 0, st, (const BgState*)states, n, out);
}

#ifdef DIEE_DEV_BUILD
static int g_conv_variant = -1;     // development override (diee_dev_conv_bench): <= 0 auto, else a fixed geometry id
void nn_set_conv_variant(int v) { g_conv_variant = v; }
#else
void nn_set_conv_variant(int) {}
#endif

// geometry by batch size: keep ~>=256 workgroups in flight while boards per workgroup (weight reuse) stay high
static int pick_variant(int G) {
#ifdef DIEE_DEV_BUILD
    if (g_conv_variant > 0) return g_conv_variant;
#endif
    // measured (scripts/conv_sweep.py, MI355X, us per launch): large batches are throughput-bound and want
    // 4 boards x 128 channels per workgroup (up to 3 workgroups per CU); small batches are latency-bound
    // and want the split-K geometry (8x more workgroups, 4x shorter dependent chains)
    if (G > 320) return 2;          // 4 boards x 128 channels, 4 waves (only reached when the fused tower is disabled)
    if (G > 80) return 6;           // split-K over 4 waves, 4 boards x 32 channels
    if (G > 72) return 5;           // split-K over 4 waves, 2 boards x 32 channels
    return 17;                      // split-K over 8 waves, 2 boards x 32 channels
}

template <int MODE>
static void conv256_dispatch(hipStream_t st, const uint16_t* act, const void* wpack, const float* bias,
                             const uint16_t* res, uint16_t* out, float* out_v, int G, int N) {
    switch (pick_variant(G)) {
        case 2: conv_launch<256, MODE, 4, 4>(st, act, wpack, bias, res, out, out_v, G, N); break;
        case 5: conv_sk_launch<MODE, 2>(st, act, wpack, bias, res, out, G, N); break;
        case 6: conv_sk_launch<MODE, 4>(st, act, wpack, bias, res, out, G, N); break;
#ifdef DIEE_DEV_BUILD               // geometries only diee_dev_conv_bench asks for
        case 1: conv_launch<256, MODE, 8, 4>(st, act, wpack, bias, res, out, out_v, G, N); break;
        case 3: conv_launch<256, MODE, 2, 2>(st, act, wpack, bias, res, out, out_v, G, N); break;
        case 4: conv_launch<256, MODE, 2, 1>(st, act, wpack, bias, res, out, out_v, G, N); break;
        case 18: conv_sk_launch<MODE, 4, 8>(st, act, wpack, bias, res, out, G, N); break;
        case 7: conv_sk_launch<MODE, 8>(st, act, wpack, bias, res, out, G, N); break;
#endif
        default: conv_sk_launch<MODE, 2, 8>(st, act, wpack, bias, res, out, G, N); break;   // 17: split-K over 8 waves
    }
}

// mode: 0 relu(conv+b), 1 relu(conv+b+res), 2 heads; c_in 16 (init block, 6 real channels) or 256
void launch_conv3x3(hipStream_t st, int c_in, int mode, const uint16_t* act, const void* wpack, const float* bias,
                    const uint16_t* res, uint16_t* out, float* out_v, int G, int N) {
    if (G <= 0) return;
    if (c_in == 16) {
        if (G > 512) conv_launch<16, 0, 8, 4>(st, act, wpack, bias, res, out, out_v, G, N);
        else conv_launch<16, 0, 2, 2>(st, act, wpack, bias, res, out, out_v, G, N);
    }
    else if (mode == 0) conv256_dispatch<0>(st, act, wpack, bias, res, out, out_v, G, N);
    else if (mode == 1) conv256_dispatch<1>(st, act, wpack, bias, res, out, out_v, G, N);
    else if (mode == 3) conv256_dispatch<3>(st, act, wpack, bias, res, out, out_v, G, N);   // raw conv + bias (training)
    else if (G > 96) conv_sk_launch<2, 4>(st, act, wpack, bias, res, out, G, N, out_v);  // heads: N = 64 (35 real), split-K
    else if (G > 72) conv_sk_launch<2, 2>(st, act, wpack, bias, res, out, G, N, out_v);
    else conv_sk_launch<2, 2, 8>(st, act, wpack, bias, res, out, G, N, out_v);           // K over 8 waves, like the tower layers of this size
                                                                                          // (and like the cluster tower's own head convs: same bits)
}

// ---- training-step helpers (die-e_amd/train_ops.py): the tower convolutions of the learn loop's training step run on the
// inference conv kernel (MODE 3 = raw conv + bias) for the forward pass and, with the weights transposed and flipped,
// for the input gradient; the weight gradient is col^T x dY with col = im2col of the saved input.
// fp32 OIHW [256][256][3][3] -> bf16 B fragments [n/32][cs*9 + tap][lane][8] of k_conv3x3 (pack_conv on the host);
// transpose: the fragments of W'[c][n][2-ky][2-kx] (the convolution that maps dY to dX)
__global__ void k_pack_conv_w(const float* __restrict__ w, uint16_t* __restrict__ out, int transpose) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;           // one thread per fragment element
    if (i >= 8 * 144 * 64 * 8) return;
    const int j = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) % 144, s = i / (144 * 512);
    const int n = s * 32 + (lane & 31), c = (ks / 9) * 16 + 8 * (lane >> 5) + j, t = ks % 9;
    const float v = transpose ? w[((size_t)c * 256 + n) * 9 + (8 - t)] : w[((size_t)n * 256 + c) * 9 + t];
    out[i] = f2bf(v);
}
// the same packing for every tower convolution of the net in one launch, both layouts (forward, transposed), with the
// gathers staged through LDS: block (s*16 + cs, layout, layer) reads 32 x 16 x 9 weights in contiguous runs and writes
// the 9 fragments [cs*9 .. cs*9 + 8] of output tile s as one 9216-byte run
struct PackPtrs { const float* w[kMaxPackLayers]; };
__global__ __launch_bounds__(256) void k_pack_conv_w_multi(PackPtrs ptrs, uint16_t* __restrict__ out) {
    __shared__ float tile[32 * 145];
    const int tid = threadIdx.x, s = blockIdx.x >> 4, cs = blockIdx.x & 15, transpose = blockIdx.y;
    const float* __restrict__ w = ptrs.w[blockIdx.z];
    for (int i = tid; i < 32 * 144; i += 256) {
        if (!transpose) {
            const int nl = i / 144, r = i - nl * 144;                // run of 16 c x 9 taps of output channel s*32 + nl
            tile[nl * 145 + r] = w[((size_t)(s * 32 + nl) * 256 + cs * 16) * 9 + r];
        } else {
            const int cl = i / 288, r = i - cl * 288, nl = r / 9, t = 8 - (r - nl * 9);
            tile[nl * 145 + cl * 9 + t] = w[((size_t)(cs * 16 + cl) * 256 + s * 32) * 9 + r];
        }
    }
    __syncthreads();
    uint16_t* o = out + ((size_t)blockIdx.z * 2 + transpose) * (8 * 144 * 64 * 8) + (size_t)(s * 144 + cs * 9) * 512;
    for (int i = tid; i < 9 * 64; i += 256) {
        const int t = i >> 6, lane = i & 63;
        const float* src = tile + (lane & 31) * 145 + 8 * (lane >> 5) * 9 + t;
        u32x4 v;
        v.x = f2bf(src[0]) | ((uint32_t)f2bf(src[9]) << 16);   v.y = f2bf(src[18]) | ((uint32_t)f2bf(src[27]) << 16);
        v.z = f2bf(src[36]) | ((uint32_t)f2bf(src[45]) << 16); v.w = f2bf(src[54]) | ((uint32_t)f2bf(src[63]) << 16);
        *(u32x4*)(o + (size_t)i * 8) = v;
    }
}
// col[row][t*256 + c] = x[row + 6*dy + dx][c] inside the board, 0 outside (16 bytes per thread)
__global__ void k_im2col3x3(const uint16_t* __restrict__ x, uint16_t* __restrict__ col, int M) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * 9 * 32) return;
    const int ch = (int)(i & 31), t = (int)((i >> 5) % 9), row = (int)(i / (9 * 32));
    const int p = row % 24, y = p / 6, xx = p % 6, dy = t / 3 - 1, dx = t % 3 - 1;
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned)(y + dy) < 4u && (unsigned)(xx + dx) < 6u) v = *(const u32x4*)(x + (size_t)(row + 6 * dy + dx) * 256 + ch * 8);
    *(u32x4*)(col + (size_t)row * 2304 + t * 256 + ch * 8) = v;
}
void launch_pack_conv_w(hipStream_t st, const float* w, uint16_t* out, int transpose) {
    hipLaunchKernelGGL(k_pack_conv_w, dim3(8 * 144 * 64 * 8 / 256), dim3(256), 0, st, w, out, transpose);
}
void launch_pack_conv_w_multi(hipStream_t st, const float* const* w, int n, uint16_t* out) {
    PackPtrs p{};
    for (int i = 0; i < n; ++i) p.w[i] = w[i];
    hipLaunchKernelGGL(k_pack_conv_w_multi, dim3(128, 2, n), dim3(256), 0, st, p, out);
}
void launch_im2col3x3(hipStream_t st, const uint16_t* x, uint16_t* col, int boards) {
    const size_t n = (size_t)boards * 24 * 9 * 32;
    hipLaunchKernelGGL(k_im2col3x3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, col, boards * 24);
}

void launch_policy_fc(hipStream_t st, const uint16_t* hp, const void* wpack, const float* bias, float* logits, int G,
                      const uint32_t* n_rows) {
    if (G <= 0) return;
    hipLaunchKernelGGL(k_policy_fc, dim3((G + 31) / 32, 43), dim3(64), 0, st, hp, (const u32x4*)wpack, bias, logits, G, n_rows);
}

void launch_softmax_value(hipStream_t st, const float* logits, const float* hv, const float* wv, float* policy,
                          float* value, int G) {
    if (G <= 0) return;
    hipLaunchKernelGGL(k_softmax_value, dim3(G), dim3(64), 0, st, logits, hv, wv, policy, value, G);
}

}  // namespace diee
