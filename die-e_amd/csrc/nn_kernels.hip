// nn_kernels.hip -- policy/value ResNet inference (reference: src/alphazero/nnet.rs:24-34,57-107,
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
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdint.h>
#include <cstdio>
#include <type_traits>

#include "bg_device.h"
#include "launch.h"
#include "mcts_device.h"
#include "nn_device.h"

#ifndef DIEE_TOWER_SCHED
#define DIEE_TOWER_SCHED 1        // 1 = sched_group_barrier interleave of each k-step's loads between its MFMAs (0: loads issued as a block)
#endif
#ifndef DIEE_TOWER_BORDER
#define DIEE_TOWER_BORDER 1      // 1 = the 4-board fused tower skips (tap, fragment) pairs that are all zero padding
#endif
#ifndef DIEE_TOWER_STAGGER
#define DIEE_TOWER_STAGGER 0      // s_sleep argument (x 64 cycles) for waves 4..7 at the start of every fused-tower layer (0: none)
#endif
#ifndef DIEE_TOWER_PRIO
#define DIEE_TOWER_PRIO 1         // 1 = waves 4..7 of the 8-wave fused tower run at s_setprio 1
#endif
#ifndef DIEE_CL_POLL_SLEEP
#define DIEE_CL_POLL_SLEEP 2      // s_sleep argument (x 64 cycles) between two polls of the cluster tower's input tile (6: same, 12 / 24: slower)
#endif
#ifndef DIEE_CL_PD
#define DIEE_CL_PD 0              // cluster tower: LDS prefetch distance in k-steps (0 = by geometry)
#endif
#ifndef DIEE_CL_LATE
#define DIEE_CL_LATE 6            // cluster tower: how many of a layer's 18 next-layer weight fragments per wave are requested AFTER the MFMA loop
                                  // (in the shadow of the partial-tile reduction) instead of inside it; 0 = all inside (rounds 1-2)
#endif
#ifndef DIEE_PAIR_STORE_AUX
#define DIEE_PAIR_STORE_AUX 0     // pair tower hand-off stores: 0 = plain (the line stays in the XCD's L2, where the other member's sc1 loads find it:
                                  // 327 ... 407 us), 16 = sc1 (write-through, placement-independent: 340 ... 414 us).  As in the cluster tower a pair that
                                  // does NOT share an XCD never sees plain data: its polls time out, the engine reports it and falls back (tags per
                                  // 8 bytes: nothing stale is ever taken)
#endif
#ifndef DIEE_PAIR_PF
#define DIEE_PAIR_PF 6            // pair tower: weight k-steps in flight per wave and column fragment
#endif
#ifndef DIEE_PAIR_AHEAD
#define DIEE_PAIR_AHEAD 1         // pair tower: the other member's half requested ahead of its use (see pair_layer)
#endif
#ifndef DIEE_PAIR_ABLATE
#define DIEE_PAIR_ABLATE 0        // timing builds (wrong results): 1 = members do not wait for each other (one unchecked read), 2 = no exchange at all
#endif
#ifndef DIEE_PAIR_RES
#define DIEE_PAIR_RES 0           // timing builds (wrong results): the pair tower's K loop without its LDS reads / weight loads (see pair_layer)
#endif
#ifndef DIEE_PAIR_POLL_SLEEP
#define DIEE_PAIR_POLL_SLEEP 4    // pair tower: s_sleep argument (x 64 cycles) between two polls of a member's sentinel chunk
#endif
#ifndef DIEE_REM_SPLIT
#define DIEE_REM_SPLIT 512        // a compacted batch's remainder of at most this many boards runs on the pair tower (k_tower16p; 416 with
                                  // the 2-board geometry of rounds 1-2), above on the 4-board geometry <4,8,6>
#endif
#ifndef DIEE_CL_PRIVATE
#define DIEE_CL_PRIVATE 1         // cluster tower: every wave stages its own channel columns of the activation tile (no barrier between staging and the MFMA loop)
#endif
#ifndef DIEE_CL_LATE_OUT
#define DIEE_CL_LATE_OUT DIEE_CL_LATE
#endif
#ifndef DIEE_CL_LATE_SLEEP
#define DIEE_CL_LATE_SLEEP 0      // cluster tower: s_sleep argument (x 64 cycles) of the waves without output chunks in front of their late weight requests
#endif
#ifndef DIEE_CL_ABLATE
#define DIEE_CL_ABLATE 0          // timing experiments on the cluster tower: 1 = no MFMA loop, 2 = no partial-tile exchange, 3 = no weight loads
#endif
#ifndef DIEE_CL_SPLIT4_SMALL
#define DIEE_CL_SPLIT4_SMALL 0
#endif
#ifndef DIEE_PAIR_BIAS_EARLY
#define DIEE_PAIR_BIAS_EARLY 1
#endif
#ifndef DIEE_TOWER_BIAS_EARLY
#define DIEE_TOWER_BIAS_EARLY 1
#endif
#ifndef DIEE_TOWER_EPI_OVERLAP
#define DIEE_TOWER_EPI_OVERLAP 0
#endif
#ifndef DIEE_PAIR_UNROLL
#define DIEE_PAIR_UNROLL 1         // the pair tower's k loop unrolled in full (round 4: 320 ... 390 us against 333 ... 406 at 300 ... 512 boards, profiles/r04h_pair_unroll_ab.txt)
#endif
#ifndef DIEE_TOWER_UNROLL4
#define DIEE_TOWER_UNROLL4 1
#endif
#ifndef DIEE_TOWER_DUPW
#define DIEE_TOWER_DUPW 0        // timing build (results unchanged): every wave of the fused tower also requests the weight fragments of wave ^ 4 and waits for them like
                                 // for its own -- the weight traffic of a 2 row-group x 4 column-group split of the workgroup (with -DDIEE_TOWER_ABLATE=7: its LDS traffic too)
#endif
#ifndef DIEE_TOWER_ABLATE
#define DIEE_TOWER_ABLATE 0      // diagnostic builds only: 1 = no main loop, 2 = no epilogue, 3 = in-kernel clock stamps, 4 = cluster tower re-reads two layers' weights, 6 ... 9 = fused tower without its LDS reads / weight loads, 12 = with half its weight loads (tower_layer16)
#endif

namespace diee {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;


__device__ __forceinline__ uint16_t f2bf(float x) {
    const __bf16 b = (__bf16)x;            // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }

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

// Small-batch tower in ONE launch ("cluster tower").  k_conv3x3_sk's geometry -- GT boards x 32 channels per
// workgroup, K split over 8 waves -- is kept for all 38 layers; the 8 workgroups that own the 8 channel slices of
// one board group form a cluster: a layer's output goes to global memory (the X / H ping-pong of the per-layer path)
// and the cluster meets on one counter per board group before the next layer's tile is staged.  What a launch
// boundary costs the per-layer path (~2 us of gap, cold weight loads, tile staging behind them) shrinks to one
// release / acquire pair: the next layer's 18 weight fragments are requested before the wait and arrive during it,
// and the residual slice never leaves registers.  Arithmetic per output element is exactly k_conv3x3_sk<., GT, 8>'s
// (same MFMA, same split, same reduction order): results are bit-identical to the per-layer path.
//
// Placement: workgroups are dispatched round-robin over the 8 XCDs, so the cluster of group g is given the linear
// ids {g%8 + 8*(8*(g/8) + slice)}: all 8 on one XCD, sharing its L2.  Correctness does not depend on that (the
// handshake is an agent-scope release / acquire), only the latency does.  Every workgroup of the grid must be
// resident at once (the launcher checks the occupancy); a wait is bounded and reports through `err` instead of hanging.
constexpr int kTowerLayerStride = 8 * 144 * 64;          // u32x4 per layer (1.18 MB)
constexpr int kClusterSpinLimit = 1 << 18;
#ifndef DIEE_CL_PRS
#define DIEE_CL_PRS 144
#endif
// partial-tile row stride (bytes); 160 (half-waves of a C-layout store on disjoint bank halves) measured no faster
constexpr int kClusterPartStride = DIEE_CL_PRS;

// device-coherent 16-byte accesses for data other workgroups exchange inside a launch: relaxed agent-scope atomics
// (global_load/store_dwordx2 sc1) reach the coherent level themselves, so the handshake needs no L2-wide
// write-back / invalidate (an agent-scope fence costs ~0.1 us per wave and serialises per XCD: measured 19 us per layer)
// (16-byte forms: buffer_load/store_dwordx4 with aux 16 = sc1; the ready tags below are per 8-byte half, so nothing
// depends on a 16-byte access being performed as one)
typedef __attribute__((ext_vector_type(4))) unsigned int rb_u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coherent_rsrc(uint16_t* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 ld_coherent16(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const rb_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    return u32x4{v[0], v[1], v[2], v[3]};
}
// Producer stores are PLAIN: a cluster lives on one XCD (see the blockIdx -> (xcd, slice, group) mapping in k_tower_cl), the
// vector L1 is write-through, so a plain store lands in that XCD's L2 and STAYS there, where the consumers' sc1 (L1-bypassing,
// L2-served) loads find it; an sc1 store also writes through to memory and drops the line, and the same-XCD reader then pays the
// cross-XCD rate (guide: 104-122 vs 66-73 GB/s per block, +0.1-0.3 us per hand-off).  Measured: cluster forward 122.2 ->
// 119.3 us at 8 boards, 350 -> 340 us at 256, +0.6 % games/s (same box).  A cluster that did NOT sit on one XCD would never
// see the data: its polls time out and the engine falls back (loud), it cannot read a half-written tile (tags per 8 bytes).
#ifndef DIEE_CL_STORE_AUX
#define DIEE_CL_STORE_AUX 0       // 16 = sc1 (write-through past the XCD's L2, round 1); 0 = plain (the line stays in the XCD's L2)
#endif
__device__ __forceinline__ void st_coherent16(__amdgpu_buffer_rsrc_t r, int byte_off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(rb_u32x4{v[0], v[1], v[2], v[3]}, r, byte_off, 0, DIEE_CL_STORE_AUX);
}

// Ready flag carried by the data: activations are post-ReLU bf16, so their sign bits are free.  The output of layer w
// is written with the sign bit of the first element of every 8-byte word set to tag_of(w); consecutive writes into the
// same buffer (w, w+2) carry opposite tags, so a consumer that polls its tile with 8-byte coherent loads knows word by
// word when the new layer has landed: no store acknowledgement, counter update or counter poll on the critical path.
// (w = 0 writes plain data over unknown leftovers, so that one hand-over uses the counter; w = 37 is the tower output.)
__device__ __forceinline__ uint32_t tag_of(int w) { return (uint32_t)(((w >> 1) ^ w) & 1) << 15; }

// The rest of the network inside the cluster launch (round 3).  Below 257 boards an evaluation used to be four launches --
// cluster tower, head convs, policy FC, k_expand -- and every launch boundary is 1.2-1.5 us plus a cold start on a chain
// that is pure latency (a handful of boards on a 256-CU chip).  With `whead` set the launch goes on after layer 37:
//   "layer 38" = the two head convolutions (nnet.rs:76-78, 88-90), 64 output columns = two 32-column slices, on the
//       workgroups of slices 0 (policy, 32 channels) and 1 (value, 3 channels) with the SAME loop body as a tower layer
//       (k_conv3x3_sk<2, GT, NSPLIT>'s arithmetic: per-layer path = same bits); the tower output reaches them as a
//       tagged hand-off like any layer (tag in bit 31 of a word, see tag38: bit 15 must end the launch clear);
//   the policy features (post-ReLU bf16: sign bits free) go to channels 0..31 of the H rows of the group, tagged
//       tag_of(38) -- H held layer 36 (the other polarity), so the data is its own ready flag once more and the hand-off
//       to all eight workgroups of the cluster costs one tagged poll; the value features go to hv (nobody in here reads them);
//   policy FC 768 -> 1352 (nnet.rs:80-85): the cluster's eight workgroups share the 43 output slices, one wave per
//       slice, k_policy_fc's MFMA sequence per slice (same bits), the slice's first 24 weight fragments requested before
//       the poll; logits [board][1352] f32.
struct ClusterHeads {
    const u32x4* whead;     // [2][144][64] x 16 B: head convs as two 32-column slices (wconv[39]); null = stop after the tower
    const float* bhead;     // [64]
    const u32x4* wfc;       // [43][48][64] x 16 B
    const float* bfc;       // [1376]
    float* hv;              // [G][72]
    float* logits;          // [G][1352]
};
constexpr int kGrowLdsPerWave = (sizeof(WaveScratch) + 255) / 256 * 256;      // LDS of one growth wave
constexpr int kFcRowStride = 1536 + 16;       // LDS stride of a board's 768 policy features (bank-conflict-free ds_read_b128 over boards)

template <int GT, int NSPLIT>
__global__ __launch_bounds__(64 * NSPLIT) void k_tower_cl(uint16_t* X,               // [M][256] bf16: init block output in, tower output out
                                                          uint16_t* H,               // [M][256] bf16 scratch (conv1 outputs)
                                                          const u32x4* __restrict__ wt,      // [38][8][144][64] x 16 B
                                                          const float* __restrict__ bias,    // [38][256]
                                                          int M, int n_groups,
                                                          uint32_t* sync,            // [n_groups] counters, 128 B apart, zero between launches
                                                          uint32_t* err, unsigned long long* dbg,
                                                          const BgState* __restrict__ states,    // non-null: the init block runs in here
                                                          const u32x4* __restrict__ winit,   // [8][9][64] x 16 B (k_conv3x3<16,...>'s fragments)
                                                          const float* __restrict__ binit,   // [256]
                                                          ClusterHeads hd,                   // whead non-null: head convs + policy FC in here
                                                          GrowReq gr, int tower_blocks,      // tower_blocks > 0: the blocks behind them grow the tree
                                                          int nx,                            // XCDs that host clusters (8; fewer: option cl_pack, see the launcher)
                                                          const uint32_t* __restrict__ n_rows_dev,   // non-null: the boards to evaluate are counted on the device
                                                          uint32_t* __restrict__ rows_log) {         // (<= the host's M / 24; 0: nothing to do) and noted here
    constexpr int ROWS = GT * 24, MF = (ROWS + 31) / 32, RS = 528, CPR = 32, NT = 64 * NSPLIT;
    constexpr int KS = 144 / NSPLIT;                // k-steps per wave: (16 / NSPLIT) channel steps x 9 taps
    constexpr int PF = 18;                          // weight fragments in flight per wave
    constexpr int LATE = KS == PF ? DIEE_CL_LATE : 0;   // of them, requested after the MFMA loop (see there); K split 4 ways: the ring covers half a layer, all inside
    constexpr int LATE_OUT = KS == PF ? DIEE_CL_LATE_OUT : 0;     // the same for the waves that reduce and store (their late requests sit behind their stores)
    constexpr int PRS = kClusterPartStride;
    constexpr int TILE = ((ROWS + 1) * RS + 16 * 35 + 15) / 16 * 16;
    constexpr int PART = NSPLIT * MF * 32 * PRS;
    constexpr bool ALIAS = TILE + PART > 160 * 1024;        // the partial tiles must reuse the activation tile's LDS
    constexpr int CH = (ROWS * 4 + NT - 1) / NT;    // output chunks (row, 8 channels) per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* part = ALIAS ? smem : smem + TILE;        // [NSPLIT waves][MF*32 rows][32] f32

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tower_blocks > 0 && (int)blockIdx.x >= tower_blocks) {
        // ---- growth blocks (the search's request, launch.h): one wave per slot creates the children of the leaf the selection chose
        // -- legal plays, codes, states, headers: everything of an expansion that does not wait for this very evaluation -- while the
        // cluster workgroups evaluate it; k_expand<true> commits them with their priors afterwards.  These blocks share nothing with
        // the tower's (the launcher keeps the whole grid resident: nobody waits for a block that cannot start).
        const uint32_t slot = ((uint32_t)blockIdx.x - (uint32_t)tower_blocks) * NSPLIT + (uint32_t)wave;
        grow_slot<false>(gr.T, gr.S, gr.G, gr.n, gr.it, slot, *(WaveScratch*)(smem + (size_t)wave * kGrowLdsPerWave));
        return;
    }
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    if (xcd >= nx) {
        // packed clusters (nx < 8): the workgroups dispatched to the XCDs that host no cluster are the growth blocks
        if (gr.n > 0) {
            const uint32_t slot = (uint32_t)(j * (8 - nx) + (xcd - nx)) * NSPLIT + (uint32_t)wave;
            grow_slot<false>(gr.T, gr.S, gr.G, gr.n, gr.it, slot, *(WaveScratch*)(smem + (size_t)wave * kGrowLdsPerWave));
        }
        return;
    }
    const int nslice = j & 7, grp = xcd + nx * (j >> 3);            // a whole cluster on one XCD (measured 5-8 % faster than
                                                                    // slice s of every group on XCD s, which would stream 1/8 of the weights per XCD)
    // (the first layer's weight fragments are requested before the row count is looked at: its round trip hides behind them)
    const u32x4* wp = wt + ((size_t)nslice * 144 + wave * KS) * 64 + lane;
    u32x4 bq[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) bq[i] = wp[i * 64];
    if (n_rows_dev) {
        // the tail of a batch (search_types.h, Tail): k_tail planned this launch's rows on the device; the clusters beyond them
        // return at once -- a cluster takes part in a launch as a whole or not at all, so the tags and counters it leaves behind
        // are those of its last complete launch
        const int nr = (int)*n_rows_dev < M / 24 ? (int)*n_rows_dev : M / 24;
        if (rows_log && blockIdx.x == 0 && tid == 0) *rows_log = (uint32_t)nr;
        M = nr * 24; n_groups = (nr + GT - 1) / GT;
    }
    if (grp >= n_groups) return;
    const int row0 = grp * ROWS;
    uint32_t* cnt = sync + grp * 32;


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
            base[t][f] = src * RS + (lane >> 5) * 16 + wave * (16 / NSPLIT) * 32;     // this wave's channel steps
        }
    }
    const bool has_out = wave * 64 < ROWS * 4;      // this wave reduces and stores output chunks (wave-uniform)
    // this thread's output chunks (row, 8 channels) and their residual: the block input, kept in registers
    const __amdgpu_buffer_rsrc_t rX = coherent_rsrc(X, M * 512), rH = coherent_rsrc(H, M * 512);
    u32x4 resreg[CH];
    if (states) {
        // ---- init block in here (nnet.rs:64-67: conv 6 -> 256 + BN + ReLU), for ALL 256 channels of this cluster's boards:
        // every workgroup of the cluster repeats it (331 k MAC per board) instead of waiting for a launch of its own and
        // a hand-over.  Same fragments, same MFMA sequence as k_conv3x3<16, 0, ...>: the tile gets the same bits.
        static_assert(!ALIAS || (TILE + (ROWS + 1) * 32 <= PART), "room for the input planes behind the activation tile");
        char* pt = smem + TILE;                     // [ROWS + 1][16 channels] bf16 planes (6 real), 32-byte rows
        for (int r = tid; r < ROWS + 1; r += NT) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r < ROWS && row0 + r < M) {
                const BgState st = states[(row0 + r) / 24];
                const int p = (row0 + r) % 24;
                uint32_t w[3];
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    w[c] = (uint32_t)f2bf(bg_plane_dev(st, 2 * c, p)) | ((uint32_t)f2bf(bg_plane_dev(st, 2 * c + 1, p)) << 16);
                v = u32x4{w[0], w[1], w[2], 0u};
            }
            *(u32x4*)(pt + r * 32) = v;
            *(u32x4*)(pt + r * 32 + 16) = u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        constexpr int NTW = 8 / NSPLIT;             // 32-channel N-tiles per wave
#pragma unroll
        for (int q = 0; q < NTW; ++q) {
            const int nt = wave * NTW + q;
            f32x16 acc[MF];
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[f][i] = 0.0f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const bf16x8 bw = __builtin_bit_cast(bf16x8, winit[((size_t)nt * 9 + t) * 64 + lane]);
                const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
                for (int f = 0; f < MF; ++f) {
                    const int R = 32 * f + (lane & 31);
                    const int p = R % 24, y = p / 6, x = p % 6;
                    const bool ok = R < ROWS && (unsigned)(y + dy) < 4u && (unsigned)(x + dx) < 6u;
                    const bf16x8 av = *(const bf16x8*)(pt + (ok ? R + 6 * dy + dx : ROWS) * 32 + (lane >> 5) * 16);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bw, acc[f], 0, 0, 0);
                }
            }
            const float bv = binit[nt * 32 + (lane & 31)];
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = 32 * f + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    if (r >= ROWS) continue;
                    float v = acc[f][i] + bv;
                    v = (v > 0.0f && row0 + r < M) ? v : 0.0f;
                    *(uint16_t*)(smem + r * RS + (nt * 32 + (lane & 31)) * 2) = f2bf(v);
                }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; ++c) {              // the first block's residual: this workgroup's slice of the init output
            const int i = tid + c * NT;
            resreg[c] = u32x4{0u, 0u, 0u, 0u};
            if (i < ROWS * 4) resreg[c] = *(const u32x4*)(smem + (i >> 2) * RS + (nslice * 32 + (i & 3) * 8) * 2);
        }
    } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = tid + c * NT, gr = row0 + (i >> 2);
            resreg[c] = u32x4{0u, 0u, 0u, 0u};
            if (i < ROWS * 4 && gr < M) resreg[c] = ld_coherent16(rX, (gr * 256 + nslice * 32 + (i & 3) * 8) * 2);
        }
    }
    // a wait timed out, now or in an earlier launch (reported through err): stop waiting, finish the launch
    bool dead = (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u) != 0u;

#if DIEE_TOWER_ABLATE == 3
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;    // per-phase shader-clock sums over layers 2..36 (thread 0)
#define CL_STAMP(i) do { const unsigned long long tn = __builtin_readcyclecounter(); if (l >= 2 && l <= 36) ph[i] += tn - tprev; tprev = tn; } while (0);
#else
#define CL_STAMP(i) do {} while (0);
#endif
    const bool heads = hd.whead != nullptr;
    const int n_layers = heads ? (nslice < 2 ? 39 : 38) : 38;       // "layer 38": the head convs, on the workgroups of slices 0 and 1
    // tag of layer 37's output when the heads consume it in here: bit 31 of the tagged words (the second element's sign
    // bit) instead of bit 15 -- X must END the launch with bit 15 clear in every word, because the next launch's layer 1
    // announces itself through bit 15 = 1 over whatever this one left; bit 31 is set by nobody else, layer 35's data
    // (in place before layer 37's) has it clear, and every layer's store rewrites the whole word
    constexpr uint32_t tag38 = 0x80000000u;
    for (int l = 0; l < n_layers; ++l) {
        const __amdgpu_buffer_rsrc_t in = (l & 1) ? rH : rX, out = (l & 1) ? rX : rH;
        CL_STAMP(5)                                 // end-of-layer barrier
        float4 bias_lo[CH], bias_hi[CH];            // requested ahead of the epilogue
        const float* bl = l < 38 ? bias + l * 256 + nslice * 32 : hd.bhead + nslice * 32;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = (tid + c * NT) < ROWS * 4 ? tid + c * NT : 0;
            bias_lo[c] = *(const float4*)(bl + (i & 3) * 8);
            bias_hi[c] = *(const float4*)(bl + (i & 3) * 8 + 4);
        }
        if (l == 1) {
            // ---- first hand-over (H holds unknown leftovers): meet on the group's counter ----
            if (tid == 0 && !dead) {
                int spins = 0;
                while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 8u) < 0) {
                    if (++spins > kClusterSpinLimit) { atomicOr(err, 4u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();                        // the tile loads below are device-coherent themselves
        }
        // ---- stage the activation tile; from layer 2 on the data is its own ready flag (see tag_of) ----
        // DIEE_CL_PRIVATE: every wave stages exactly the channel columns its own K share reads (wave w <-> input channels
        // [w * 256 / NSPLIT, ...): with K split 8 ways those are the 32 channels ONE producer workgroup wrote), so nothing
        // a wave reads in the MFMA loop was written by another wave: no workgroup barrier between staging and the loop, and a
        // wave starts as soon as ITS producer's slice has landed instead of when the slowest of the eight has.
#if DIEE_CL_PRIVATE
        constexpr int CW = CPR / NSPLIT;                        // 16-byte chunks of a row inside this wave's columns (4 or 8)
        constexpr int RPI = 64 / CW;                            // rows per wave-instruction (16 or 8)
        const int crow = lane / CW, ccol = wave * CW + lane % CW;
        if (l > 0 || !states) {
            constexpr int NCH = (ROWS + RPI - 1) / RPI;         // chunks per lane
            constexpr int BATCH = NCH > 8 ? 8 : NCH;            // requested back to back before the first is stored
            const uint32_t tmask = l == 38 ? tag38 : 0x8000u;                 // where the producer's tag sits
            const uint32_t want = l == 38 ? tag38 : l >= 2 ? tag_of(l - 1) : 0u;
#pragma unroll
            for (int k0 = 0; k0 < NCH; k0 += BATCH) {
                u32x4 v[BATCH];
                for (int spins = 0;; ++spins) {
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) {
                        const int r = crow + (k0 + k) * RPI;
                        const int gr = row0 + ((k0 + k < NCH && r < ROWS) ? r : 0);      // ragged: re-read row 0, never stored
                        v[k] = ld_coherent16(in, (gr < M ? gr : M - 1) * 512 + ccol * 16);
                    }
                    uint32_t bad = 0u;
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) bad |= (v[k][0] ^ want) | (v[k][2] ^ want);
                    if ((bad & tmask) == 0u || l < 2 || dead) break;
                    if (spins > kClusterSpinLimit / 16) { atomicOr(err, 4u); dead = true; break; }
                    __builtin_amdgcn_s_sleep(DIEE_CL_POLL_SLEEP);
                }
                if (k0 == 0) CL_STAMP(0)            // first batch of the tile polled in
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int r = crow + (k0 + k) * RPI;
                    if (row0 + r >= M) v[k] = u32x4{0u, 0u, 0u, 0u};
                    v[k][0] &= ~tmask; v[k][2] &= ~tmask;
                    if (k0 + k < NCH && r < ROWS) *(u32x4*)(smem + r * RS + ccol * 16) = v[k];
                }
            }
        }
        // zero row, this wave's columns (the partial tiles may alias it: every layer)
        if (lane < CW) *(u32x4*)(smem + ROWS * RS + (wave * CW + lane) * 16) = u32x4{0u, 0u, 0u, 0u};
        if (wave == NSPLIT - 1 && lane >= 61) *(u32x4*)(smem + ROWS * RS + (CPR + lane - 61) * 16) = u32x4{0u, 0u, 0u, 0u};
        __builtin_amdgcn_wave_barrier();            // (LDS operations of one wave execute in order: its reads below see these writes)
#else
        if (l > 0 || !states) {
            constexpr int NCH = (ROWS * CPR + NT - 1) / NT;     // 16-byte chunks per thread
            constexpr int BATCH = NCH > 8 ? 8 : NCH;            // requested back to back before the first is stored
            const uint32_t tmask = l == 38 ? tag38 : 0x8000u;                 // where the producer's tag sits
            const uint32_t want = l == 38 ? tag38 : l >= 2 ? tag_of(l - 1) : 0u;
#pragma unroll
            for (int k0 = 0; k0 < NCH; k0 += BATCH) {
                u32x4 v[BATCH];
                for (int spins = 0;; ++spins) {
                    // rows past the batch re-read its last row: no branch, no wait in between
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) {
                        int gr = row0 + (tid + (k0 + k) * NT) / CPR;
                        if ((ROWS * CPR % NT != 0 || NCH % BATCH != 0) && tid + (k0 + k) * NT >= ROWS * CPR) gr = row0;      // ragged: re-read, never stored
                        v[k] = ld_coherent16(in, (gr < M ? gr : M - 1) * 512 + (tid % CPR) * 16);
                    }
                    uint32_t bad = 0u;
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) bad |= (v[k][0] ^ want) | (v[k][2] ^ want);
                    if ((bad & tmask) == 0u || l < 2 || dead) break;
                    if (spins > kClusterSpinLimit / 16) { atomicOr(err, 4u); dead = true; break; }
                    __builtin_amdgcn_s_sleep(DIEE_CL_POLL_SLEEP);
                }
                if (k0 == 0) CL_STAMP(0)            // first batch of the tile polled in
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int i = tid + (k0 + k) * NT;
                    if (row0 + i / CPR >= M) v[k] = u32x4{0u, 0u, 0u, 0u};
                    v[k][0] &= ~tmask; v[k][2] &= ~tmask;
                    if ((ROWS * CPR % NT == 0 && NCH % BATCH == 0) || i < ROWS * CPR)
                        *(u32x4*)(smem + (i / CPR) * RS + (tid % CPR) * 16) = v[k];
                }
            }
        }
        if (tid < CPR + 3) *(u32x4*)(smem + ROWS * RS + tid * 16) = u32x4{0u, 0u, 0u, 0u};      // zero row
        __syncthreads();
#endif
        CL_STAMP(1)                                 // tile staged (barrier)

        f32x16 acc[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][i] = 0.0f;
        // A fragments are read PD k-steps ahead of their MFMAs.  Distances of 2 .. 8 (-DDIEE_CL_PD=n) measured no faster
        // than 1 at any geometry: the loop is bound by the weight stream (timing builds -DDIEE_CL_ABLATE=1 / 2 / 3: at one
        // board per cluster the loop costs 1.35 us per layer, 0.8 us of it vanish without the weight loads; the partial
        // exchange costs 0.13 - 0.84 us), not by LDS latency.
        constexpr int PD = DIEE_CL_PD > 0 ? DIEE_CL_PD : 1, NB = PD + 1;
        bf16x8 a[NB][MF];
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int f = 0; f < MF; ++f) a[d][f] = *(const bf16x8*)(smem + base[d % 9][f] + (d / 9) * 32);
        // next layer's fragments (the head convs' after layer 37 where this workgroup runs them; last layer: reloads its own, unused)
        const u32x4* whp = heads ? hd.whead + ((size_t)(nslice < 2 ? nslice : 0) * 144 + wave * KS) * 64 + lane : wp;
#if DIEE_TOWER_ABLATE == 4
        const u32x4* wn = wp + (size_t)(l & 1) * kTowerLayerStride;                  // timing experiment: weights stay L2-resident (wrong results)
#else
        const u32x4* wn = l < 37 ? wp + (size_t)(l + 1) * kTowerLayerStride : (heads && nslice < 2) ? whp : wp + (size_t)37 * kTowerLayerStride;
#endif
        const u32x4* wc = l < 38 ? wp + (size_t)l * kTowerLayerStride : whp;         // this layer's
        // the ring runs ahead into the next layer.  A CU takes in weights at ~64 B/clk: a layer's 147 KB need ~2300 cycles of
        // that pipe, the MFMA loop lasts ~1400 -- with every request inside the loop the waves queue at the pipe and the
        // loop stretches to the stream's length (in-kernel stamps: loop + wait for the slowest wave 3100 cycles of a 6100-cycle
        // layer).  The last LATE fragments per wave (needed last in the next loop) are requested after the loop instead,
        // while two of the eight waves reduce the partial tiles and store: that part of the layer uses no memory pipe.
        // (The count is a compile-time constant of the loop body: two instances, picked by the wave's role.)
        auto mfma_loop = [&](auto late_c) {
            constexpr int late_k = decltype(late_c)::value;
#pragma unroll
            for (int u = 0; u < (DIEE_CL_ABLATE == 1 ? 0 : KS); ++u) {
                const int un = u + PD;
                if (un < KS) {
#pragma unroll
                    for (int f = 0; f < MF; ++f) a[un % NB][f] = *(const bf16x8*)(smem + base[un % 9][f] + (un / 9) * 32);
                }
                const bf16x8 b = __builtin_bit_cast(bf16x8, bq[u % PF]);
                if (DIEE_CL_ABLATE != 3 && (KS != PF || u < KS - late_k)) bq[u % PF] = u + PF < KS ? wc[(u + PF) * 64] : wn[(u + PF - KS) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < MF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u % NB][f], b, acc[f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (LATE == LATE_OUT || !has_out) mfma_loop(std::integral_constant<int, LATE>{});
        else mfma_loop(std::integral_constant<int, LATE_OUT>{});

        if (ALIAS) __syncthreads();                 // every wave is done reading the activation tile
        CL_STAMP(2)                                 // MFMA loop + barrier
#pragma unroll
        for (int f = 0; f < (DIEE_CL_ABLATE == 2 ? 0 : MF); ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = 32 * f + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                *(float*)(part + ((wave * MF * 32 + r) * PRS) + (lane & 31) * 4) = acc[f][i];
            }
        if (DIEE_CL_ABLATE == 2) { float sink = 0.0f; for (int f = 0; f < MF; ++f) for (int i = 0; i < 16; ++i) sink += acc[f][i]; if (sink == 12345.678f) part[0] = 1; }
        __syncthreads();
        CL_STAMP(3)                                 // partial tiles written (barrier)
        if (LATE > 0 && !has_out && DIEE_CL_ABLATE != 3) {
            if (DIEE_CL_LATE_SLEEP) __builtin_amdgcn_s_sleep(DIEE_CL_LATE_SLEEP);      // (measured: every delay here costs, 114.6 -> 117.5 / 121.3 / 123.1 us at 4 / 8 / 12)
#pragma unroll
            for (int u = KS - LATE; u < KS; ++u) bq[u % PF] = wn[(u + PF - KS) * 64];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = tid + c * NT, er = i >> 2, ec8 = i & 3, egr = row0 + er;
            if (i >= ROWS * 4 || egr >= M) continue;
            float v[8] = {bias_lo[c].x, bias_lo[c].y, bias_lo[c].z, bias_lo[c].w, bias_hi[c].x, bias_hi[c].y, bias_hi[c].z, bias_hi[c].w};
#pragma unroll
            for (int w = 0; w < (DIEE_CL_ABLATE == 2 ? 0 : NSPLIT); ++w) {
                const float4 lo = *(const float4*)(part + (w * MF * 32 + er) * PRS + ec8 * 32);
                const float4 hi = *(const float4*)(part + (w * MF * 32 + er) * PRS + ec8 * 32 + 16);
                v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
            }
            if (l & 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[2 * k] += __uint_as_float(resreg[c][k] << 16);
                    v[2 * k + 1] += __uint_as_float(resreg[c][k] & 0xffff0000u);
                }
            }
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x0 = v[2 * k] > 0.0f ? v[2 * k] : 0.0f, x1 = v[2 * k + 1] > 0.0f ? v[2 * k + 1] : 0.0f;
                o[k] = (uint32_t)f2bf(x0) | ((uint32_t)f2bf(x1) << 16);
            }
            if (l & 1) resreg[c] = o;               // block output = next block's input
            if (l == 38) {
                // heads (MODE 2 of k_conv3x3_sk): slice 0 = policy features, handed to the whole cluster through channels
                // 0..31 of the H rows; slice 1 = value features (3 channels) -> hv, f32
                if (nslice == 0) {
                    o[0] |= tag_of(38); o[2] |= tag_of(38);
                    st_coherent16(rH, (egr * 256 + ec8 * 8) * 2, o);
                } else if (ec8 == 0) {
                    float* ov = hd.hv + (size_t)(egr / 24) * 72 + (egr % 24) * 3;
                    ov[0] = v[0] > 0.0f ? v[0] : 0.0f; ov[1] = v[1] > 0.0f ? v[1] : 0.0f; ov[2] = v[2] > 0.0f ? v[2] : 0.0f;
                }
                continue;
            }
            // the tower output leaves untagged when nothing in here reads it, else tagged in bit 31 (see tag38)
            const uint32_t tg = l < 37 ? tag_of(l) : heads ? tag38 : 0u;
            o[0] |= tg; o[2] |= tg;
            st_coherent16(out, (egr * 256 + nslice * 32 + ec8 * 8) * 2, o);
        }
        if (LATE_OUT > 0 && has_out && DIEE_CL_ABLATE != 3) {  // behind this wave's stores: they are what the other workgroups wait for
#pragma unroll
            for (int u = KS - LATE_OUT; u < KS; ++u) bq[u % PF] = wn[(u + PF - KS) * 64];
        }
        CL_STAMP(4)                                 // reduce + store issued
        if (l == 0 || l == 37) {
            // first hand-over: signal through the counter once this workgroup's stores are acknowledged;
            // the second arrival round (end of launch) re-arms the counter for the next launch
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const uint32_t old = atomicAdd(cnt, 1u);
                if (old == 15u) atomicExch(cnt, 0u);
            }
        } else {
            // partial tiles are consumed before the next tile is staged over them; without aliasing the barrier still
            // pays: it keeps the waves that have no output chunk from polling the next tile (and loading the memory
            // system) while the others reduce and store
            __syncthreads();
        }
    }
    if (heads) {
        // ---- policy FC 768 -> 1352 over the cluster's boards: slices nslice, nslice + 8, ... of the 43, one wave each ----
        char* hpt = smem;                               // [GT][kFcRowStride] policy features (the activation tile is done with)
        const int s_first = nslice + 8 * wave;
        // the whole first slice (48 fragments, 48 KB per wave) is requested before the poll: six of the eight workgroups wait
        // out the head convolutions here anyway, and the tower's ring registers are dead by now
        u32x4 fb[48];
        {
            const u32x4* wf = hd.wfc + (size_t)(s_first < 43 ? s_first : 0) * 48 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 48; ++i) fb[i] = wf[i * 64];
        }
        constexpr int NHC = (ROWS * 4 + NT - 1) / NT;   // 16-byte chunks of the features per thread (a row = 32 channels = 4 chunks)
#pragma unroll
        for (int c = 0; c < NHC; ++c) {
            const int i = tid + c * NT, row = (i < ROWS * 4 ? i : 0) >> 2, ch = i & 3, gr = row0 + row;
            u32x4 v;
            for (int spins = 0;; ++spins) {
                v = ld_coherent16(rH, (gr < M ? gr : M - 1) * 512 + ch * 16);
                if ((((v[0] ^ tag_of(38)) | (v[2] ^ tag_of(38))) & 0x8000u) == 0u || dead) break;
                if (spins > kClusterSpinLimit / 16) { atomicOr(err, 4u); dead = true; break; }
                __builtin_amdgcn_s_sleep(DIEE_CL_POLL_SLEEP);
            }
            v[0] &= ~0x8000u; v[2] &= ~0x8000u;
            if (gr >= M) v = u32x4{0u, 0u, 0u, 0u};
            if (i < ROWS * 4) *(u32x4*)(hpt + (row / 24) * kFcRowStride + (row % 24) * 64 + ch * 16) = v;
        }
        __syncthreads();
        for (int sl = s_first; sl < 43; sl += 8 * NSPLIT) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            const u32x4* wfs = hd.wfc + (size_t)sl * 48 * 64 + lane;
            const char* ap = hpt + ((lane & 31) < GT ? (lane & 31) : 0) * kFcRowStride + (lane >> 5) * 16;     // rows past the cluster's boards: computed, never stored
            if (sl != s_first) {                        // (K split 4 ways: a wave's second slice)
#pragma unroll
                for (int i = 0; i < 48; ++i) fb[i] = wfs[i * 64];
            }
            // k_policy_fc's sequence: one accumulator, k-steps in order (same bits as the stand-alone FC)
#pragma unroll
            for (int i = 0; i < 48; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(ap + i * 32), __builtin_bit_cast(bf16x8, fb[i]), acc, 0, 0, 0);
            const int n = sl * 32 + (lane & 31);
            if (n < 1352) {
                const float bv = hd.bfc[n];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    const int g = grp * GT + r;
                    if (r < GT && g * 24 < M) hd.logits[(size_t)g * 1352 + n] = acc[i] + bv;
                }
            }
        }
    }
#if DIEE_TOWER_ABLATE == 3
    if (dbg && tid == 0)
        for (int i = 0; i < 6; ++i) dbg[(size_t)blockIdx.x * 8 + i] = ph[i];
#endif
}

// Large-batch variant: the whole 38-layer tower in ONE launch.  A workgroup owns 4 boards and all 256
// channels (4 waves x 2 N-fragments x 3 M-fragments), so a layer's output tile is exactly the next
// layer's input tile: activations ping-pong between two LDS tiles and never leave the CU, the
// residual is read from LDS, and the per-layer launch gap, tile staging and global epilogue (6.4 us
// of a 27 us layer) disappear.  Weights stream L2 -> registers as in k_conv3x3, the ring of 9 x 2
// fragments runs ahead across layer boundaries.  One __syncthreads() per layer.

// One tower layer inside the fused kernel.  GT boards per workgroup (MF M-fragments), NF N-fragments
// per wave, PF = weight fragments in flight per wave and N-fragment (k-steps ahead; 9 or 18).
template <bool RES, int GT, int NF, int PF>
__device__ __forceinline__ void tower_layer(char* tin, char* tout, const u32x4* wp, const u32x4* wp_next,
                                            const float* __restrict__ bias, const int (&base)[9][(GT * 24 + 31) / 32],
                                            u32x4 (&bq)[PF][NF], int lane, int wave) {
    constexpr int ROWS = GT * 24, MF = (ROWS + 31) / 32;
    f32x16 acc[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int q = 0; q < NF; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][q][i] = 0.0f;
    bf16x8 a[2][MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) a[0][f] = *(const bf16x8*)(tin + base[0][f]);
    for (int it = 0; it < 8; ++it) {
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int t = u % 9, cur = u & 1, nxt = cur ^ 1, un = u + 1;
            const int csn = it * 2 + un / 9;                  // 16 on the very last step: reads padding, unused
#pragma unroll
            for (int f = 0; f < MF; ++f) a[nxt][f] = *(const bf16x8*)(tin + base[un % 9][f] + csn * 32);
            bf16x8 b[NF];
#pragma unroll
            for (int q = 0; q < NF; ++q) b[q] = __builtin_bit_cast(bf16x8, bq[u % PF][q]);
            {
                const int cs_pf = it * 2 + u / 9 + PF / 9;     // same tap, PF/9 channel steps ahead (next layer at the end)
                const u32x4* src = cs_pf < 16 ? wp + (size_t)(cs_pf * 9 + t) * 64 : wp_next + (size_t)((cs_pf - 16) * 9 + t) * 64;
#pragma unroll
                for (int q = 0; q < NF; ++q) bq[u % PF][q] = src[(size_t)q * 144 * 64];
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
    // epilogue straight into the other LDS tile (bf16 [row][channel], same padded layout)
#pragma unroll
    for (int q = 0; q < NF; ++q) {
        const int n = (wave * NF + q) * 32 + (lane & 31);
        const float bv = bias[n];
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = 32 * f + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (ROWS % 32 != 0 && r >= ROWS) continue;
                const int off = r * 528 + n * 2;
                float v = acc[f][q][i] + bv;
                if (RES) v += bf2f(*(const uint16_t*)(tout + off));      // y = relu(conv2(h) + x), in place over x
                v = v > 0.0f ? v : 0.0f;
                *(uint16_t*)(tout + off) = f2bf(v);
            }
    }
    __syncthreads();
}

// GT boards x all 256 channels per workgroup: 256 / (32 * NF) waves.
template <int GT, int NF, int PF>
__global__ __launch_bounds__(64 * (8 / NF)) void k_tower(const uint16_t* __restrict__ x_in,   // [M][256] bf16 (init block output)
                                                         const u32x4* __restrict__ wt,      // [38][8][144][64] x 16 B
                                                         const float* __restrict__ bias,    // [38][256]
                                                         uint16_t* __restrict__ x_out, int M) {
    constexpr int ROWS = GT * 24, MF = (ROWS + 31) / 32, RS = 528, NT = 64 * (8 / NF);
    constexpr int TILE = ((ROWS + 1) * RS + 16 * 34 + 128 + 15) / 16 * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tx = smem;
    char* th = smem + TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS;

    const u32x4* wp0 = wt + (size_t)(wave * NF) * 144 * 64 + lane;
    u32x4 bq[PF][NF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
        for (int q = 0; q < NF; ++q) bq[i][q] = wp0[((size_t)q * 144 + i) * 64];

    for (int i = tid; i < ROWS * 32; i += NT) {
        const int r = i >> 5, ch = i & 31;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + r < M) v = *(const u32x4*)(x_in + (size_t)(row0 + r) * 256 + ch * 8);
        *(u32x4*)(tx + r * RS + ch * 16) = v;
    }
    for (int i = tid; i < 2 * 36; i += NT) {                  // zero rows of both tiles (+ over-read slack)
        char* tl = i < 36 ? tx : th;
        *(u32x4*)(tl + ROWS * RS + (i % 36) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    int base[9][MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int R = 32 * f + (lane & 31);
        const int p = R % 24, y = p / 6, x = p % 6;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool ok = R < ROWS && (unsigned)(y + dy) < 4u && (unsigned)(x + dx) < 6u;
            base[t][f] = (ok ? R + 6 * dy + dx : ROWS) * RS + (lane >> 5) * 16;
        }
    }
    __syncthreads();

    for (int blk = 0; blk < 19; ++blk) {
        const u32x4* w1 = wp0 + (size_t)(2 * blk) * kTowerLayerStride;
        const u32x4* w2 = w1 + kTowerLayerStride;
        const u32x4* w3 = blk < 18 ? w2 + kTowerLayerStride : w2;       // after the last layer: harmless re-read
        tower_layer<false, GT, NF, PF>(tx, th, w1, w2, bias + (2 * blk) * 256, base, bq, lane, wave);
        tower_layer<true, GT, NF, PF>(th, tx, w2, w3, bias + (2 * blk + 1) * 256, base, bq, lane, wave);
    }
    for (int i = tid; i < ROWS * 32; i += NT) {
        const int r = i >> 5, ch = i & 31;
        if (row0 + r < M) *(u32x4*)(x_out + (size_t)(row0 + r) * 256 + ch * 8) = *(const u32x4*)(tx + r * RS + ch * 16);
    }
}

// ---- the same fused tower on v_mfma_f32_16x16x32_bf16 -------------------------------------------------
// 16-row fragments fit boards exactly (24 rows: 2 boards = 3 fragments, no padding rows) and the chip
// holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS give-back item 7).  Weights are packed a
// second time as 16-column B fragments: [layer][n/16][k-step = cstep32*9 + tap][lane][8].
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int kTower16LayerStride = 16 * 72 * 64;        // u32x4 per layer (1.18 MB)

// Border-aware row order (4 boards per workgroup): a 16-row fragment holds the SAME four board positions of the four
// boards, and the six fragments are the board's left column, right column, top and bottom edge (without corners) and
// its two interior rows.  For a tap that points off the board the whole fragment is zero padding -- left column x
// dx = -1, right column x dx = +1, top edge x dy = -1, bottom edge x dy = +1: 12 of the 54 (tap, fragment) pairs --
// so its LDS read and MFMAs are not issued at all: 22 % fewer MFMAs for bit-identical results (the skipped products
// are exact zeros).  The LDS tile itself keeps the [board*24 + position] layout; only the lane -> row map changes.
__device__ __forceinline__ constexpr int border_pos(int f, int i) {      // position (6*y + x) number i of fragment f
    return f == 0 ? 6 * i : f == 1 ? 6 * i + 5 : f == 2 ? 1 + i : f == 3 ? 19 + i : f == 4 ? 7 + i : 13 + i;
}
__device__ __forceinline__ constexpr bool border_skip(bool sp, int t, int f) {
    return sp && ((f == 0 && t % 3 == 0) || (f == 1 && t % 3 == 2) || (f == 2 && t / 3 == 0) || (f == 3 && t / 3 == 2));
}
__device__ __forceinline__ constexpr int border_live(bool sp, int t, int mf) {   // fragments with work at tap t
    int n = 0;
    for (int f = 0; f < mf; ++f) n += border_skip(sp, t, f) ? 0 : 1;
    return n;
}
// LDS row of lane-column n (0..15) of fragment f
template <bool SP>
__device__ __forceinline__ int tower_row(int f, int n) {
    if (!SP) return 16 * f + n;
    const int i = n & 3;
    const int pos = f == 0 ? 6 * i : f == 1 ? 6 * i + 5 : f == 2 ? 1 + i : f == 3 ? 19 + i : f == 4 ? 7 + i : 13 + i;
    return (n >> 2) * 24 + pos;
}

template <bool RES, int GT, int NW, int PF, bool SP = false>
__device__ __forceinline__ void tower_layer16(char* tin, char* tout, const u32x4* wp, const u32x4* wp_next,
                                              const float* __restrict__ bias, const uint32_t (&basep)[9][((GT * 24 + 15) / 16 + 1) / 2],
                                              u32x4 (&bq)[PF][16 / NW], int lane, int wave) {
    constexpr int ROWS = GT * 24, MF = (ROWS + 15) / 16, NFR = 16 / NW;
#ifdef DIEE_TOWER_HALFN
    constexpr int NQ = NW == 4 ? NFR / 2 : NFR;       // timing experiment: half the output channels per workgroup (wrong results)
#else
    constexpr int NQ = NFR;
#endif
    f32x4 acc[MF][NFR];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int q = 0; q < NFR; ++q) acc[f][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    bf16x8 a[2][MF];
    // One wave per SIMD: the layer's bias is requested HERE, a whole k loop ahead of the epilogue that adds it -- requested there, its L2
    // round trip (and an s_waitcnt vmcnt(0) that also drains the next layer's weight ring) sat exposed at the end of every layer, with no
    // second wave on the SIMD to fill it.  16 registers; the 8-wave geometries (256 VGPRs, already spilling) keep the late load.
    constexpr bool kBiasEarly = DIEE_TOWER_BIAS_EARLY && NW == 4;
    float4 bvq[NFR];
    if (kBiasEarly) {
#pragma unroll
        for (int q = 0; q < NFR; ++q) bvq[q] = *(const float4*)(bias + (wave * NFR + q) * 16 + (lane >> 4) * 4);
    }
#if DIEE_TOWER_DUPW
    u32x4 bq2[PF][NFR];
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
        for (int q = 0; q < NFR; ++q) bq2[i][q] = u32x4{0u, 0u, 0u, 0u};
#endif
#if DIEE_TOWER_STAGGER
    // the two waves of a SIMD run the same program and leave the layer barrier together: a short delay for the upper
    // half puts one wave's load phase beside its partner's MFMA phase (MI355X_MICROARCH.md, "Two waves per SIMD", item 9)
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_sleep(DIEE_TOWER_STAGGER);
#endif
    // per-lane LDS addresses of the A fragments are < 64 KiB: two per register (keeps the 4-board geometry out of scratch)
    auto baddr = [&](int t, int f) -> int { return (f & 1) ? (int)(basep[t][f >> 1] >> 16) : (int)(basep[t][f >> 1] & 0xffffu); };
#pragma unroll
    for (int f = 0; f < MF; ++f)
        if (!border_skip(SP, 0, f)) a[0][f] = *(const bf16x8*)(tin + baddr(0, f));
    // One wave per SIMD (NW == 4): the 96 accumulators live in AGPRs and the register allocator permutes them across the back edge of this
    // loop -- 132 v_accvgpr_read / _write / _mov per 18 k-steps (~10 % of the loop's issue slots, in front of the first MFMA of every
    // trip); unrolled in full there is no back edge to permute across (DIEE_TOWER_UNROLL4, round 4).  The 8-wave geometries keep the loop.
    constexpr int kUnrollIt = (DIEE_TOWER_UNROLL4 && NW == 4) ? 4 : 1;
#pragma unroll kUnrollIt
    for (int it = 0; it < (DIEE_TOWER_ABLATE == 1 ? 0 : 4); ++it) {
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int cur = u & 1, nxt = cur ^ 1, un = u + 1;
            const int csn = it * 2 + un / 9;                  // 8 on the very last step: reads padding, unused
            // timing builds (wrong results): 6 / 9 = no LDS reads of the A fragments in the loop, 7 = every second k-step's only, 8 / 9 = no weight loads
            constexpr bool kNoLds = DIEE_TOWER_ABLATE == 6 || DIEE_TOWER_ABLATE == 9, kNoW = DIEE_TOWER_ABLATE == 8 || DIEE_TOWER_ABLATE == 9;
            const bool lds_step = !kNoLds && !(DIEE_TOWER_ABLATE == 7 && (u & 1));
#pragma unroll
            for (int f = 0; f < MF; ++f)
                if (lds_step && !border_skip(SP, un % 9, f)) a[nxt][f] = *(const bf16x8*)(tin + baddr(un % 9, f) + csn * 64);
            bf16x8 b[NFR];
#pragma unroll
            for (int q = 0; q < NFR; ++q) b[q] = __builtin_bit_cast(bf16x8, bq[u % PF][q]);
            if (!kNoW && !(DIEE_TOWER_ABLATE == 12 && (u & 1))) {      // 12: timing build, weight fragments requested on every second k-step only = the
                                                                       // weight traffic per board of an 8-boards-per-workgroup geometry (wrong results)
                const int sp = it * 18 + u + PF;               // k-step to prefetch (of the next layer past 72)
                const u32x4* src = sp < 72 ? wp + (size_t)sp * 64 : wp_next + (size_t)(sp - 72) * 64;
#pragma unroll
                for (int q = 0; q < NQ; ++q) bq[u % PF][q] = src[(size_t)q * 72 * 64];
#if DIEE_TOWER_DUPW
                // the partner wave's fragments, in a ring of their own: consumed (waited for) where this wave's own are, PF k-steps later
                const ptrdiff_t partner = (ptrdiff_t)(((wave ^ 4) - wave) * NFR) * 72 * 64;
#pragma unroll
                for (int q = 0; q < NQ; ++q) { asm volatile("" :: "v"(bq2[u % PF][q])); bq2[u % PF][q] = src[partner + (ptrdiff_t)q * 72 * 64]; }
#endif
            }
#if DIEE_TOWER_SCHED == 0
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                if (border_skip(SP, u % 9, f)) continue;           // this fragment x tap is all padding
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    acc[f][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[q], a[cur][f], acc[f][q], 0, 0, 0);   // D = W^T x act^T
            }
#if DIEE_TOWER_SCHED == 0
            __builtin_amdgcn_sched_barrier(0);
#else
            // (DIEE_TOWER_EPI_OVERLAP, one wave per SIMD only -- the loop is unrolled, `it` is a constant: the LAST k-step of a layer is left to the
            // scheduler, without the pattern and the fence below, so that the epilogue's residual reads and VALU of the fragments that are
            // finished may move up between the MFMAs of the fragments that are not: nothing else fills the SIMD while one wave converts 24 tiles)
            if (!(DIEE_TOWER_EPI_OVERLAP && kUnrollIt == 4 && it == 3 && u == 17))
            // interleave this k-step's loads between its MFMAs instead of issuing them as a block in front
            {
                constexpr int dummy = 0; (void)dummy;
                const int n_mfma = border_live(SP, u % 9, MF) * NFR, n_lds = lds_step ? border_live(SP, un % 9, MF) : 0;
#pragma unroll
                for (int i = 0; i < MF * NFR; ++i) {
                    if (i >= n_mfma) break;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
                    if (i < n_lds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // 1 LDS read
                    else if (!kNoW && i < n_lds + NFR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
                }
            }
            if (!(DIEE_TOWER_EPI_OVERLAP && kUnrollIt == 4 && it == 3 && u == 17)) __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }
    // epilogue.  The operands are swapped (weights as the MFMA's A, activations as its B), so in the 16x16 C/D
    // layout (col = lane&15, row = (lane>>4)*4 + i) a lane holds FOUR CONSECUTIVE CHANNELS of one board position:
    // one 8-byte LDS write (and residual read) per tile instead of four 2-byte ones.
    auto epilogue_tile = [&](int f, int q, const float4 bv) {
        const int n0 = (wave * NFR + q) * 16 + (lane >> 4) * 4;
        const int r = tower_row<SP>(f, lane & 15);
        if (ROWS % 16 != 0 && r >= ROWS) return;
        const int off = r * 528 + n0 * 2;
        float v0 = acc[f][q][0] + bv.x, v1 = acc[f][q][1] + bv.y, v2 = acc[f][q][2] + bv.z, v3 = acc[f][q][3] + bv.w;
        if (RES) {                                            // y = relu(conv2(h) + x), in place over x
            const uint2 rv = *(const uint2*)(tout + off);
            v0 += __uint_as_float(rv.x << 16); v1 += __uint_as_float(rv.x & 0xffff0000u);
            v2 += __uint_as_float(rv.y << 16); v3 += __uint_as_float(rv.y & 0xffff0000u);
        }
        v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; v2 = v2 > 0.0f ? v2 : 0.0f; v3 = v3 > 0.0f ? v3 : 0.0f;
        uint2 o;
        o.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
        o.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
        *(uint2*)(tout + off) = o;
    };
    if (kBiasEarly) {
        // fragment-major, the order the last k-step finishes the tiles in (see above)
#pragma unroll
        for (int f = 0; f < (DIEE_TOWER_ABLATE == 2 ? 0 : MF); ++f)
#pragma unroll
            for (int q = 0; q < NFR; ++q) epilogue_tile(f, q, bvq[q]);
    } else {
#pragma unroll
        for (int q = 0; q < (DIEE_TOWER_ABLATE == 2 ? 0 : NFR); ++q) {
            const float4 bv = *(const float4*)(bias + (wave * NFR + q) * 16 + (lane >> 4) * 4);
#pragma unroll
            for (int f = 0; f < MF; ++f) epilogue_tile(f, q, bv);
        }
    }
#if DIEE_TOWER_ABLATE != 5      // 5: timing experiment, what the one barrier per layer costs (wrong results)
    __syncthreads();
#endif
}

// Which boards a launch of the fused tower evaluates when the batch is COMPACTED on the device (the search skips the
// slots whose selected leaf was terminal: their network row would be computed and never read).  row_slot[row] = slot of
// the row-th slot that needs an evaluation, *n_rows = how many there are; both are written by k_row_map right before, so
// the host does not know n_rows and launches up to three towers whose workgroups decide for themselves:
//   mode 1  the whole passes of the chip: rows [0, main)      (4 boards per workgroup, full rounds of 256 workgroups)
//   mode 2  the remainder [main, n_rows) if it has more than kRemSplit boards   (4 boards per workgroup)
//   mode 3  the remainder if it has at most kRemSplit boards                    (2 boards per workgroup)
// main = n_rows rounded down to a multiple of kFullChip (or n_rows itself if the rest would fill > 928 boards of a pass),
// capped by what the host launched for mode 1.  All three are the same arithmetic per output element (the 16x16x32
// fused family), so WHICH launch evaluates a row never shows in its result.
constexpr int kFullChip = 1024, kRemSplit = DIEE_REM_SPLIT, kFullRest = 928;
constexpr int kFourWaveMin = 640;      // boards above which the 4-wave fused geometry beats the 8-wave one (NetWeights::tower_table says the same)
struct RowMap {
    const uint32_t* row_slot;   // null: rows are slots (no compaction)
    const uint32_t* n_rows;
    int mode;                   // 0 plain, 1 / 2 / 3 see above
    int main_cap;               // boards the mode-1 launch of this evaluation can take (0: there is none)
};

// BAND: no effect on the code -- a second instantiation of the same geometry for another band of live games, so that each band is a
// row of its own in a rocprofv3 kernel summary (bench.py's per-launch figure for the full-chip band must agree with ONE such row)
template <int GT, int NW, int PF, int BAND = 0>
__global__ __launch_bounds__(64 * NW) void k_tower16(const uint16_t* __restrict__ x_in, const u32x4* __restrict__ wt,
                                                    const float* __restrict__ bias, uint16_t* __restrict__ x_out, int M,
                                                    RowMap rm,
                                                    unsigned long long* dbg /* clock stamps, diagnostic builds only */,
                                                    const BgState* __restrict__ states,   // non-null: the init block runs in here
                                                    const u32x4* __restrict__ winit,      // [16][9][64] x 16 B (pack_init16)
                                                    const float* __restrict__ binit,
                                                    const u32x4* __restrict__ whead,      // non-null: the head convs run in here
                                                    const float* __restrict__ bhead,      // [64] (policy 0..31, value 32..34)
                                                    uint16_t* __restrict__ hp,            // [G][768] bf16, k' = p*32 + c
                                                    float* __restrict__ hv) {             // [G][72]  f32,  k' = p*3 + c
    constexpr int ROWS = GT * 24, MF = (ROWS + 15) / 16, RS = 528, NT = 64 * NW, NFR = 16 / NW;
    constexpr int TILE = ((ROWS + 1) * RS + 16 * 34 + 128 + 15) / 16 * 16;
    constexpr bool SP = GT == 4 && DIEE_TOWER_BORDER != 0;       // border-aware row order (see border_skip)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tx = smem;
    char* th = smem + TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int board0 = blockIdx.x * GT;
    if (rm.mode != 0) {                                          // compacted batch: this launch's share of the rows
        const int nr = (int)*rm.n_rows;
        const int tail = nr % kFullChip;
        int main_b = tail > kFullRest ? nr : nr - tail;
        main_b = main_b < rm.main_cap ? main_b : rm.main_cap;
        const int rest = nr - main_b;
        int lo, hi;
        if (rm.mode == 1) { lo = 0; hi = main_b; }
        else if (rm.mode == 2) { lo = main_b; hi = rest > kRemSplit ? nr : main_b; }
        else { lo = main_b; hi = rest <= kRemSplit ? nr : main_b; }
        board0 += lo;
        if (board0 >= hi) return;
        M = hi * 24;
    }
    const int row0 = board0 * 24;

    const u32x4* wp0 = wt + (size_t)(wave * NFR) * 72 * 64 + lane;
    u32x4 bq[PF][NFR];
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
        for (int q = 0; q < NFR; ++q) bq[i][q] = wp0[((size_t)q * 72 + i) * 64];

    if (states) {
        // input planes (backgammon_logic.rs:198-252) -> th, 64 bytes (32 channels, 6 real) per row; the init block below
        for (int i = tid; i < ROWS * 4; i += NT) {
            const int r = i >> 2, ch = i & 3;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ch == 0 && row0 + r < M) {
                const int brd = (row0 + r) / 24;
                const BgState st = states[rm.row_slot ? (int)rm.row_slot[brd] : brd];
                const int p = (row0 + r) % 24;
                uint32_t w[3];
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    w[c] = (uint32_t)f2bf(bg_plane_dev(st, 2 * c, p)) | ((uint32_t)f2bf(bg_plane_dev(st, 2 * c + 1, p)) << 16);
                v = u32x4{w[0], w[1], w[2], 0u};
            }
            *(u32x4*)(th + r * RS + ch * 16) = v;
        }
    } else {
        for (int i = tid; i < ROWS * 32; i += NT) {
            const int r = i >> 5, ch = i & 31;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row0 + r < M) v = *(const u32x4*)(x_in + (size_t)(row0 + r) * 256 + ch * 8);
            *(u32x4*)(tx + r * RS + ch * 16) = v;
        }
    }
    for (int i = tid; i < 2 * 36; i += NT) {
        char* tl = i < 36 ? tx : th;
        *(u32x4*)(tl + ROWS * RS + (i % 36) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    uint32_t basep[9][(MF + 1) / 2];
    auto fill_basep = [&](uint32_t (&bp)[9][(MF + 1) / 2], int ln) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int h = 0; h < (MF + 1) / 2; ++h) bp[t][h] = 0;
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int R = tower_row<SP>(f, ln & 15);
            const int p = R % 24, y = p / 6, x = p % 6;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const bool ok = R < ROWS && (unsigned)(y + dy) < 4u && (unsigned)(x + dx) < 6u;
                const uint32_t ad = (uint32_t)((ok ? R + 6 * dy + dx : ROWS) * RS + (ln >> 4) * 16);
                bp[t][f >> 1] |= (f & 1) ? ad << 16 : ad;
            }
        }
    };
    fill_basep(basep, lane);
    __syncthreads();
    if (states) {
        // ---- init block: conv 6 -> 256 + BN + ReLU (nnet.rs:64-67), th -> tx, one 32-channel k-step per tap; in the
        // border-aware order the all-padding (tap, fragment) pairs are skipped here too ----
        auto baddr = [&](int t, int f) -> int { return (f & 1) ? (int)(basep[t][f >> 1] >> 16) : (int)(basep[t][f >> 1] & 0xffffu); };
        f32x4 acc[MF][NFR];
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int q = 0; q < NFR; ++q) acc[f][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            bf16x8 b[NFR];
#pragma unroll
            for (int q = 0; q < NFR; ++q) b[q] = __builtin_bit_cast(bf16x8, winit[((size_t)(wave * NFR + q) * 9 + t) * 64 + lane]);
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                if (border_skip(SP, t, f)) continue;
                const bf16x8 av = *(const bf16x8*)(th + baddr(t, f));
#pragma unroll
                for (int q = 0; q < NFR; ++q) acc[f][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[q], av, acc[f][q], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < NFR; ++q) {
            const int n0 = (wave * NFR + q) * 16 + (lane >> 4) * 4;
            const float4 bv = *(const float4*)(binit + n0);
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const int r = tower_row<SP>(f, lane & 15);
                if (ROWS % 16 != 0 && r >= ROWS) continue;
                float v0 = acc[f][q][0] + bv.x, v1 = acc[f][q][1] + bv.y, v2 = acc[f][q][2] + bv.z, v3 = acc[f][q][3] + bv.w;
                const bool live = row0 + r < M;
                v0 = v0 > 0.0f && live ? v0 : 0.0f; v1 = v1 > 0.0f && live ? v1 : 0.0f;
                v2 = v2 > 0.0f && live ? v2 : 0.0f; v3 = v3 > 0.0f && live ? v3 : 0.0f;
                uint2 o;
                o.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
                o.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
                *(uint2*)(tx + r * RS + n0 * 2) = o;
            }
        }
        __syncthreads();
    }
    unsigned long long t0 = 0, r0 = 0;
    if (DIEE_TOWER_ABLATE == 3) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

#if DIEE_TOWER_PRIO
    // the second-dispatched half of an 8-wave workgroup loses every issue arbitration against its older SIMD partner
    // (MI355X_MICROARCH.md, "Two waves per SIMD", item 4): one static priority for that half, no per-segment flips
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    for (int blk = 0; blk < 19; ++blk) {
        const u32x4* w1 = wp0 + (size_t)(2 * blk) * kTower16LayerStride;
        const u32x4* w2 = w1 + kTower16LayerStride;
        const u32x4* w3 = blk < 18 ? w2 + kTower16LayerStride : w2;
        tower_layer16<false, GT, NW, PF, SP>(tx, th, w1, w2, bias + (2 * blk) * 256, basep, bq, lane, wave);
        tower_layer16<true, GT, NW, PF, SP>(th, tx, w2, w3, bias + (2 * blk + 1) * 256, basep, bq, lane, wave);
    }
    if (DIEE_TOWER_ABLATE == 3 && dbg && tid == 0) {      // in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
        dbg[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        dbg[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    if (whead) {
        // Everything the head convs need per lane is derived again from an opaque copy of the thread id: left alone the
        // compiler keeps the init block's unpacked tile addresses (and friends) alive across the 38 layers -- in scratch:
        // 36 dwords per lane stored before the tower and reloaded here, the 18 MB of WRITE_SIZE per launch that round 1
        // read as partial-line stores.
        int htid = tid;
        asm volatile("" : "+v"(htid));
        const int lane = htid & 63, wave = __builtin_amdgcn_readfirstlane(htid >> 6);
        uint32_t basep[9][(MF + 1) / 2];
        fill_basep(basep, lane);
        // ---- head convs in here (nnet.rs:76-78, 88-90): policy 32 + value 3 channels = three 16-column fragments, each
        // over the whole K; with 8 waves the fragments of a column go to two waves (every other row fragment each).
        // The tower output never leaves the CU: no x_out store, no head-conv launch. ----
        constexpr int HW = NW >= 6 ? 2 : 1;
        if (wave < 3 * HW) {
            auto baddr = [&](int t, int f) -> int { return (f & 1) ? (int)(basep[t][f >> 1] >> 16) : (int)(basep[t][f >> 1] & 0xffffu); };
            const int nt = wave % 3, mh = wave / 3;
            const u32x4* wh = whead + (size_t)nt * 72 * 64 + lane;
            constexpr int MFH = (MF + HW - 1) / HW;
            f32x4 acc[MFH];
#pragma unroll
            for (int j = 0; j < MFH; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            u32x4 ring[18];
#pragma unroll
            for (int i = 0; i < 18; ++i) ring[i] = wh[(size_t)i * 64];
            // fragment of slot j: f = mh + HW * j (border_skip needs a constant f, so both parities are spelled out)
            auto skipj = [&](int t, int j) -> bool {
                if (HW == 1) return border_skip(SP, t, j);
                return mh == 0 ? (2 * j >= MF || border_skip(SP, t, 2 * j)) : (2 * j + 1 >= MF || border_skip(SP, t, 2 * j + 1));
            };
            auto addrj = [&](int t, int j) -> int {
                if (HW == 1) return baddr(t, j);
                return mh == 0 ? baddr(t, 2 * j < MF ? 2 * j : 0) : baddr(t, 2 * j + 1 < MF ? 2 * j + 1 : 0);
            };
            bf16x8 ah[2][MFH];
#pragma unroll
            for (int j = 0; j < MFH; ++j) ah[0][j] = *(const bf16x8*)(tx + addrj(0, j));
            for (int it = 0; it < 4; ++it) {
#pragma unroll
                for (int u = 0; u < 18; ++u) {
                    const int t = u % 9, sp = it * 18 + u + 18, cur = u & 1, nxt = cur ^ 1, un = u + 1;
                    const int csn = it * 2 + un / 9;                  // 8 on the very last step: reads padding, unused
#pragma unroll
                    for (int j = 0; j < MFH; ++j) ah[nxt][j] = *(const bf16x8*)(tx + addrj(un % 9, j) + csn * 64);
                    const bf16x8 b = __builtin_bit_cast(bf16x8, ring[u]);
                    ring[u] = wh[(size_t)(sp < 72 ? sp : 71) * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < MFH; ++j)
                        if (!skipj(t, j)) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, ah[cur][j], acc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // The head features are staged in the idle tile `th` -- policy [row][32] bf16 (64 B rows), value [row][3] f32 --
            // and leave the CU below as whole 16-byte-per-lane lines: a lane holds 4 channels of one position, stored
            // straight to hp / hv that was an 8-byte partial-line write per lane (rocprofv3 WRITE_SIZE 20.3 MB per launch
            // at 1024 boards for 1.87 MB of features, round 1).
            const int n0 = nt * 16 + (lane >> 4) * 4;
            const float4 bv = *(const float4*)(bhead + n0);
#pragma unroll
            for (int j = 0; j < MFH; ++j) {
                const int f = mh + HW * j;
                if (f >= MF) continue;
                const int r = tower_row<SP>(f, lane & 15);
                if (ROWS % 16 != 0 && r >= ROWS) continue;
                float v0 = acc[j][0] + bv.x, v1 = acc[j][1] + bv.y, v2 = acc[j][2] + bv.z, v3 = acc[j][3] + bv.w;
                v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; v2 = v2 > 0.0f ? v2 : 0.0f; v3 = v3 > 0.0f ? v3 : 0.0f;
                if (n0 < 32) {
                    uint2 o;
                    o.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
                    o.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
                    *(uint2*)(th + r * 64 + n0 * 2) = o;
                } else if (n0 == 32) {
                    float* ov = (float*)(th + ROWS * 64) + r * 3;
                    ov[0] = v0; ov[1] = v1; ov[2] = v2;
                }
            }
        }
        __syncthreads();
        // hp [g][p*32 + c] and hv [g][p*3 + c] of this workgroup's boards are contiguous in HBM: 16 bytes per lane
        for (int i = tid; i < ROWS * 4; i += NT)
            if (row0 + (i >> 2) < M) *(u32x4*)(hp + (size_t)row0 * 32 + i * 8) = *(const u32x4*)(th + i * 16);
        for (int i = tid; i < ROWS * 3 / 4; i += NT)
            if (row0 + (i * 4) / 3 < M) *(u32x4*)(hv + (size_t)row0 * 3 + i * 4) = *(const u32x4*)(th + ROWS * 64 + i * 16);
        return;
    }
    for (int i = tid; i < ROWS * 32; i += NT) {
        const int r = i >> 5, ch = i & 31;
        if (row0 + r < M) *(u32x4*)(x_out + (size_t)(row0 + r) * 256 + ch * 8) = *(const u32x4*)(tx + r * RS + ch * 16);
    }
}

// ---- the fused tower on TWO workgroups per board group ("pair tower", 257 ... 512 boards) ---------------------------------
// Between 257 and 512 boards the chip is half empty under the 4-board fused tower (65 ... 128 workgroups on 256 CUs, each
// taking its 420 ... 500 us whatever the batch), and the 2-board geometry that fills it streams all 44.8 MB of weights through
// every CU for half the rows (L2-bound, 405 us).  Here the 4 boards of a group go to a PAIR of workgroups: both hold the whole
// activation tile in LDS, each computes 128 of the 256 output channels of every layer (4 waves x two 16-column fragments,
// border-aware rows: k_tower16<4, ...>'s arithmetic per output element, the K order included, so results are bit-identical to
// the other fused geometries) and streams HALF the weights; after a layer each member hands its 96 x 128 outputs to the other
// through global memory: 8-byte granules whose first element's sign bit is the ready tag (activations are post-ReLU), written
// plain (both members sit on one XCD under round-robin dispatch and meet in its L2; see DIEE_PAIR_STORE_AUX) and polled with
// L1-bypassing sc1 loads: no fence, no counter (MI355X guide, data-tagged granules).  The K loop runs the input channels in order, 0 ... 127 then 128 ... 255: member 0 owns the first half, so it
// computes on what it wrote itself while the other half arrives; member 1 needs member 0's half first and runs one hand-off
// behind, for the whole tower, not per layer.
// Tags: layer l's output goes to exchange buffer l & 1; layers 0 and 37 tag bit 31 of a word, the others bit 15 with the
// cluster tower's alternating tag_of(l) -- every region ends a launch as (bit 15 clear) whatever was launched before, so a
// reader can always tell this launch's data from leftovers (the reasoning of tag38 in k_tower_cl).
constexpr int kPairSpinLimit = 1 << 16;
constexpr int kPairMaxGroups = 128;
constexpr size_t kPairHalfBytes = 96 * 128 * 2;                   // one member's output of one layer: 96 rows x 128 channels bf16
__device__ __forceinline__ void st_coherent8(__amdgpu_buffer_rsrc_t r, int byte_off, uint2 v) {
    typedef __attribute__((ext_vector_type(2))) unsigned int rb_u32x2;
    __builtin_amdgcn_raw_buffer_store_b64(rb_u32x2{v.x, v.y}, r, byte_off, 0, 16);       // sc1: write-through
}

// fetch_peer(): the other member's half of this layer's input -> tin (blocks until it is there).  DIEE_PAIR_AHEAD: issue_peer()
// requests it an eighth of a layer ahead of its use in member 0 (the reply travels while the MFMAs run; earlier, the bytes are
// not there yet), issue_next() requests member 1's NEXT input as soon as it has published (member 0 finished that layer one
// hand-off ago); fetch_peer() then finds its chunks in registers and only re-reads what had not landed.
template <bool RES, int GT, int PF, class Fetch, class Issue, class IssueNext>
__device__ __forceinline__ void pair_layer(char* tin, char* tout, const u32x4* wp, const u32x4* wp_next, const float* __restrict__ bias,
                                           const uint32_t (&basep)[9][(GT * 24 / 16 + 1) / 2], u32x4 (&bq)[PF][2], int lane, int wave, int half,
                                           __amdgpu_buffer_rsrc_t ex_out, uint32_t tag15, uint32_t tag31, bool publish, Fetch&& fetch_peer,
                                           Issue&& issue_peer, IssueNext&& issue_next) {
    constexpr int ROWS = GT * 24, MF = ROWS / 16, NQ = 2;
    constexpr bool SP = GT == 4;                                  // border-aware fragment order (4 boards: see border_skip)
    constexpr int kPairUnrollIt = DIEE_PAIR_UNROLL ? 4 : 1;       // (round 4 experiment: the k loop unrolled in full, as in the 4-wave k_tower16)
    float4 bvq[NQ];                                               // the layer's bias, requested a k loop ahead of the epilogue (one wave per SIMD: see tower_layer16)
    if (DIEE_PAIR_BIAS_EARLY) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) bvq[q] = *(const float4*)(bias + half * 128 + (wave * NQ + q) * 16 + (lane >> 4) * 4);
    }
    f32x4 acc[MF][NQ];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[f][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    bf16x8 a[2][MF];
    auto baddr = [&](int t, int f) -> int { return (f & 1) ? (int)(basep[t][f >> 1] >> 16) : (int)(basep[t][f >> 1] & 0xffffu); };
    if (half == 1) fetch_peer();                                  // member 1: input channels 0 ... 127 are the other member's
#pragma unroll
    for (int f = 0; f < MF; ++f)
        if (!border_skip(SP, 0, f)) a[0][f] = *(const bf16x8*)(tin + baddr(0, f));
#pragma unroll kPairUnrollIt
    for (int it = 0; it < 4; ++it) {
        if (it == 2 && half == 0) {                               // member 0: the second half of K is the other member's
            fetch_peer();
#pragma unroll
            for (int f = 0; f < MF; ++f)                          // (the first fragments of this half were read ahead, before they had landed)
                if (!border_skip(SP, 0, f)) a[0][f] = *(const bf16x8*)(tin + baddr(0, f) + 4 * 64);
        }
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int cur = u & 1, nxt = cur ^ 1, un = u + 1;
            const int csn = it * 2 + un / 9;                      // 8 on the very last step: reads padding, unused
            if (DIEE_PAIR_AHEAD && u == 9 && it == 1 && half == 0) issue_peer();
            // timing builds (wrong results): DIEE_PAIR_RES 1 = A fragments read on every second k-step only, 2 = never, 3 = no weight loads, 4 = neither
            constexpr bool kNoW = DIEE_PAIR_RES == 3 || DIEE_PAIR_RES == 4;
            const bool lds_step = !(DIEE_PAIR_RES == 2 || DIEE_PAIR_RES == 4) && !(DIEE_PAIR_RES == 1 && (u & 1));
#pragma unroll
            for (int f = 0; f < MF; ++f)
                if (lds_step && !border_skip(SP, un % 9, f)) a[nxt][f] = *(const bf16x8*)(tin + baddr(un % 9, f) + csn * 64);
            bf16x8 b[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) b[q] = __builtin_bit_cast(bf16x8, bq[u % PF][q]);
            if (!kNoW) {
                const int sp = it * 18 + u + PF;                  // k-step to prefetch (of the next layer past 72)
                const u32x4* src = sp < 72 ? wp + (size_t)sp * 64 : wp_next + (size_t)(sp - 72) * 64;
#pragma unroll
                for (int q = 0; q < NQ; ++q) bq[u % PF][q] = src[(size_t)q * 72 * 64];
            }
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                if (border_skip(SP, u % 9, f)) continue;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    acc[f][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[q], a[cur][f], acc[f][q], 0, 0, 0);     // D = W^T x act^T
            }
            {
                const int n_mfma = border_live(SP, u % 9, MF) * NQ, n_lds = lds_step ? border_live(SP, un % 9, MF) : 0;
#pragma unroll
                for (int i = 0; i < MF * NQ; ++i) {
                    if (i >= n_mfma) break;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < n_lds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    else if (!kNoW && i < n_lds + NQ) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // epilogue: own 128 channels -> the other LDS tile and, tagged, the exchange buffer
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int nl = (wave * NQ + q) * 16 + (lane >> 4) * 4;    // channel inside this member's half
        const int n0 = half * 128 + nl;
        const float4 bv = DIEE_PAIR_BIAS_EARLY ? bvq[q] : *(const float4*)(bias + n0);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int r = tower_row<SP>(f, lane & 15);
            const int off = r * 528 + n0 * 2;
            float v0 = acc[f][q][0] + bv.x, v1 = acc[f][q][1] + bv.y, v2 = acc[f][q][2] + bv.z, v3 = acc[f][q][3] + bv.w;
            if (RES) {
                const uint2 rv = *(const uint2*)(tout + off);
                v0 += __uint_as_float(rv.x << 16); v1 += __uint_as_float(rv.x & 0xffff0000u);
                v2 += __uint_as_float(rv.y << 16); v3 += __uint_as_float(rv.y & 0xffff0000u);
            }
            v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; v2 = v2 > 0.0f ? v2 : 0.0f; v3 = v3 > 0.0f ? v3 : 0.0f;
            uint2 o;
            o.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
            o.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
            *(uint2*)(tout + off) = o;
        }
    }
    __syncthreads();
    // hand the 96 x 128 outputs over: whole 256-byte rows out of the LDS tile as 16-byte write-through stores (a lane's own
    // results are 8-byte pieces 32 bytes apart: 3072 eight-byte fabric writes per layer took 2.7x the time per byte), the tag
    // in the first element of both 8-byte words.  The next layer reads the same columns meanwhile: no barrier behind this.
    if (publish && DIEE_PAIR_ABLATE != 2) {
        const int tid = wave * 64 + lane;
#pragma unroll
        for (int k = 0; k < ROWS * 16 / 256; ++k) {
            const int i = tid + k * 256, r = i >> 4, c16 = i & 15;
            u32x4 v = *(const u32x4*)(tout + r * 528 + half * 256 + c16 * 16);
            v[0] |= tag15 | tag31; v[2] |= tag15 | tag31;
            __builtin_amdgcn_raw_buffer_store_b128(rb_u32x4{v[0], v[1], v[2], v[3]}, ex_out, i * 16, 0, DIEE_PAIR_STORE_AUX);
        }
    }
    if (DIEE_PAIR_AHEAD && half == 1) issue_next();
}

template <int GT, int PF>
__global__ __launch_bounds__(256) void k_tower16p(const u32x4* __restrict__ wt, const float* __restrict__ bias, int M, RowMap rm,
                                                  const BgState* __restrict__ states, const u32x4* __restrict__ winit, const float* __restrict__ binit,
                                                  const u32x4* __restrict__ whead, const float* __restrict__ bhead,
                                                  uint16_t* __restrict__ hp, float* __restrict__ hv,
                                                  uint16_t* ex /* [2][kPairMaxGroups][2][96][128] bf16 */, uint32_t* err) {
    constexpr int ROWS = GT * 24, MF = ROWS / 16, RS = 528, NT = 256;
    constexpr bool SP = GT == 4;
    constexpr int TILE = ((ROWS + 1) * RS + 16 * 34 + 128 + 15) / 16 * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tx = smem;
    char* th = smem + TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // both members of a pair on one XCD under round-robin dispatch.  With the default plain hand-off stores (DIEE_PAIR_STORE_AUX=0)
    // this is a LIVENESS condition, not just speed: a plain store stays in the producer XCD's L2, where only a same-XCD reader's
    // L1-bypassing load finds it; tower_pair_device_ok() keeps the pair tower off devices where blockIdx & 7 is not the XCD
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3, half = j & 1, grp = xcd + 8 * (j >> 1);
    int board0 = grp * GT;
    if (rm.mode != 0) {                                          // compacted batch: mode 3, the remainder of at most kRemSplit boards
        const int nr = (int)*rm.n_rows;
        const int tail = nr % kFullChip;
        int main_b = tail > kFullRest ? nr : nr - tail;
        main_b = main_b < rm.main_cap ? main_b : rm.main_cap;
        const int rest = nr - main_b;
        board0 += main_b;
        const int hi = rest <= kRemSplit ? nr : main_b;
        if (board0 >= hi) return;
        M = hi * 24;
    } else if (board0 * 24 >= M) return;
    const int row0 = board0 * 24;
    bool dead = (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u) != 0u;

    const int cf0 = half * 8 + wave * 2;                         // this wave's first 16-column fragment
    const u32x4* wp0 = wt + (size_t)cf0 * 72 * 64 + lane;
    u32x4 bq[PF][2];
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) bq[i][q] = wp0[((size_t)q * 72 + i) * 64];

    // input planes -> th (64 bytes per row), then the init block for ALL 256 channels on both members (331 k MAC per board)
    for (int i = tid; i < ROWS * 4; i += NT) {
        const int r = i >> 2, ch = i & 3;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ch == 0 && row0 + r < M) {
            const int brd = (row0 + r) / 24;
            const BgState st = states[rm.row_slot ? (int)rm.row_slot[brd] : brd];
            const int p = (row0 + r) % 24;
            uint32_t w[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                w[c] = (uint32_t)f2bf(bg_plane_dev(st, 2 * c, p)) | ((uint32_t)f2bf(bg_plane_dev(st, 2 * c + 1, p)) << 16);
            v = u32x4{w[0], w[1], w[2], 0u};
        }
        *(u32x4*)(th + r * RS + ch * 16) = v;
    }
    for (int i = tid; i < 2 * 36; i += NT) {
        char* tl = i < 36 ? tx : th;
        *(u32x4*)(tl + ROWS * RS + (i % 36) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    uint32_t basep[9][(MF + 1) / 2];
    auto fill_basep = [&](uint32_t (&bp)[9][(MF + 1) / 2], int ln) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int h = 0; h < (MF + 1) / 2; ++h) bp[t][h] = 0;
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int R = tower_row<SP>(f, ln & 15);
            const int p = R % 24, y = p / 6, x = p % 6;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const bool ok = (unsigned)(y + dy) < 4u && (unsigned)(x + dx) < 6u;
                const uint32_t ad = (uint32_t)((ok ? R + 6 * dy + dx : ROWS) * RS + (ln >> 4) * 16);
                bp[t][f >> 1] |= (f & 1) ? ad << 16 : ad;
            }
        }
    };
    fill_basep(basep, lane);
    __syncthreads();
    {
        auto baddr = [&](int t, int f) -> int { return (f & 1) ? (int)(basep[t][f >> 1] >> 16) : (int)(basep[t][f >> 1] & 0xffffu); };
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {                          // 16 column fragments = 4 per wave, two at a time (k_tower16's order per fragment)
            f32x4 acc[MF][2];
#pragma unroll
            for (int f = 0; f < MF; ++f) { acc[f][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[f][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                bf16x8 b[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) b[q] = __builtin_bit_cast(bf16x8, winit[((size_t)(wave * 4 + qq * 2 + q) * 9 + t) * 64 + lane]);
#pragma unroll
                for (int f = 0; f < MF; ++f) {
                    if (border_skip(SP, t, f)) continue;
                    const bf16x8 av = *(const bf16x8*)(th + baddr(t, f));
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[f][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[q], av, acc[f][q], 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int n0 = (wave * 4 + qq * 2 + q) * 16 + (lane >> 4) * 4;
                const float4 bv = *(const float4*)(binit + n0);
#pragma unroll
                for (int f = 0; f < MF; ++f) {
                    const int r = tower_row<SP>(f, lane & 15);
                    float v0 = acc[f][q][0] + bv.x, v1 = acc[f][q][1] + bv.y, v2 = acc[f][q][2] + bv.z, v3 = acc[f][q][3] + bv.w;
                    const bool live = row0 + r < M;
                    v0 = v0 > 0.0f && live ? v0 : 0.0f; v1 = v1 > 0.0f && live ? v1 : 0.0f;
                    v2 = v2 > 0.0f && live ? v2 : 0.0f; v3 = v3 > 0.0f && live ? v3 : 0.0f;
                    uint2 o;
                    o.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
                    o.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
                    *(uint2*)(tx + r * RS + n0 * 2) = o;
                }
            }
        }
        __syncthreads();
    }
    // exchange buffers of this pair
    const __amdgpu_buffer_rsrc_t rex = coherent_rsrc(ex, (int)(2 * kPairMaxGroups * 2 * kPairHalfBytes));
    auto ex_off = [&](int parity, int member) -> int { return (int)((((size_t)parity * kPairMaxGroups + grp) * 2 + member) * kPairHalfBytes); };
    // the other member's output of layer w -> its columns of tile `tl`: issue() requests this thread's six 16-byte chunks,
    // finish() checks every granule's tag (chunks requested too early hold the previous layer's bytes: they are waited for on a
    // sentinel and read again), stages the chunks and meets the workgroup
    constexpr int NCH = ROWS * 16 / NT;                           // 96 rows x 256 B / 256 threads = 6
    u32x4 pre[NCH];
    bool have = false;
    auto issue = [&](int w) {
        if (DIEE_PAIR_ABLATE == 2) return;
        const int src = ex_off(w & 1, half ^ 1);
#pragma unroll
        for (int k = 0; k < NCH; ++k) pre[k] = ld_coherent16(rex, src + (tid + k * NT) * 16);
        have = true;
    };
    auto finish = [&](char* tl, int w) {
        if (DIEE_PAIR_ABLATE == 2) return;
        const int src = ex_off(w & 1, half ^ 1);
        const bool t31 = w == 0 || w == 37;
        const uint32_t mask = t31 ? 0x80000000u : 0x8000u, want = t31 ? 0x80000000u : tag_of(w);
        for (int spins = 0;; ++spins) {
            if (have) {
                uint32_t bad = 0u;
#pragma unroll
                for (int k = 0; k < NCH; ++k) bad |= (pre[k][0] ^ want) | (pre[k][2] ^ want);
                if ((bad & mask) == 0u || dead || DIEE_PAIR_ABLATE) break;
            }
            // not there yet: wait on ONE chunk (256 pollers x 6 chunks per poll from every workgroup of the chip is memory traffic
            // the weight streams pay for), then take the whole share again
            for (; !dead && DIEE_PAIR_ABLATE == 0; ++spins) {
                const u32x4 p = ld_coherent16(rex, src + (tid + (NCH - 1) * NT) * 16);
                if ((((p[0] ^ want) | (p[2] ^ want)) & mask) == 0u) break;
                if (spins > kPairSpinLimit) { atomicOr(err, 4u); dead = true; break; }
                __builtin_amdgcn_s_sleep(DIEE_PAIR_POLL_SLEEP);
            }
#pragma unroll
            for (int k = 0; k < NCH; ++k) pre[k] = ld_coherent16(rex, src + (tid + k * NT) * 16);
            have = true;
            if (spins > kPairSpinLimit) { atomicOr(err, 4u); dead = true; }
        }
        have = false;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int i = tid + k * NT, r = i >> 4, c16 = i & 15;
            pre[k][0] &= ~0x80008000u; pre[k][2] &= ~0x80008000u;
            *(u32x4*)(tl + r * RS + (half ^ 1) * 256 + c16 * 16) = pre[k];
        }
        __syncthreads();
    };

    for (int blk = 0; blk < 19; ++blk) {
        const u32x4* w1 = wp0 + (size_t)(2 * blk) * kTower16LayerStride;
        const u32x4* w2 = w1 + kTower16LayerStride;
        const u32x4* w3 = blk < 18 ? w2 + kTower16LayerStride : w2;
        const int l1 = 2 * blk, l2 = 2 * blk + 1;
        // layer l1: tx -> th; its input's other half is the other member's output of layer l1 - 1 (none for the first layer:
        // both members computed the whole init block)
        {
            const __amdgpu_buffer_rsrc_t out = coherent_rsrc((uint16_t*)((char*)ex + ex_off(l1 & 1, half)), (int)kPairHalfBytes);
            pair_layer<false, GT, PF>(tx, th, w1, w2, bias + l1 * 256, basep, bq, lane, wave, half, out, l1 == 0 ? 0u : tag_of(l1),
                                  l1 == 0 ? 0x80000000u : 0u, true, [&] { if (l1 > 0) finish(tx, l1 - 1); }, [&] { if (l1 > 0) issue(l1 - 1); },
                                  [&] { issue(l1); });
        }
        {
            const __amdgpu_buffer_rsrc_t out = coherent_rsrc((uint16_t*)((char*)ex + ex_off(l2 & 1, half)), (int)kPairHalfBytes);
            pair_layer<true, GT, PF>(th, tx, w2, w3, bias + l2 * 256, basep, bq, lane, wave, half, out, l2 == 37 ? 0u : tag_of(l2),
                                 l2 == 37 ? 0x80000000u : 0u, !(l2 == 37 && half == 0), [&] { finish(th, l2 - 1); }, [&] { issue(l2 - 1); },
                                 [&] { if (l2 < 37) issue(l2); });
        }
    }
    if (half == 1) return;                                        // the head convs run on member 0
    finish(tx, 37);
    {
        // ---- head convs (nnet.rs:76-78, 88-90): three 16-column fragments over the whole K, waves 0 .. 2 (k_tower16's head section
        // at four waves: same arithmetic); everything per lane derived again from an opaque copy of the thread id (see there)
        int htid = tid;
        asm volatile("" : "+v"(htid));
        const int lane = htid & 63, wave = __builtin_amdgcn_readfirstlane(htid >> 6);
        uint32_t basep[9][(MF + 1) / 2];
        fill_basep(basep, lane);
        if (wave < 3) {
            auto baddr = [&](int t, int f) -> int { return (f & 1) ? (int)(basep[t][f >> 1] >> 16) : (int)(basep[t][f >> 1] & 0xffffu); };
            const u32x4* wh = whead + (size_t)wave * 72 * 64 + lane;
            f32x4 acc[MF];
#pragma unroll
            for (int jf = 0; jf < MF; ++jf) acc[jf] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            u32x4 ring[18];
#pragma unroll
            for (int i = 0; i < 18; ++i) ring[i] = wh[(size_t)i * 64];
            bf16x8 ah[2][MF];
#pragma unroll
            for (int jf = 0; jf < MF; ++jf) ah[0][jf] = *(const bf16x8*)(tx + baddr(0, jf));
            for (int it = 0; it < 4; ++it) {
#pragma unroll
                for (int u = 0; u < 18; ++u) {
                    const int t = u % 9, sp = it * 18 + u + 18, cur = u & 1, nxt = cur ^ 1, un = u + 1;
                    const int csn = it * 2 + un / 9;
#pragma unroll
                    for (int jf = 0; jf < MF; ++jf) ah[nxt][jf] = *(const bf16x8*)(tx + baddr(un % 9, jf) + csn * 64);
                    const bf16x8 b = __builtin_bit_cast(bf16x8, ring[u]);
                    ring[u] = wh[(size_t)(sp < 72 ? sp : 71) * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int jf = 0; jf < MF; ++jf)
                        if (!border_skip(SP, t, jf)) acc[jf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, ah[cur][jf], acc[jf], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int n0 = wave * 16 + (lane >> 4) * 4;
            const float4 bv = *(const float4*)(bhead + n0);
#pragma unroll
            for (int jf = 0; jf < MF; ++jf) {
                const int r = tower_row<SP>(jf, lane & 15);
                float v0 = acc[jf][0] + bv.x, v1 = acc[jf][1] + bv.y, v2 = acc[jf][2] + bv.z, v3 = acc[jf][3] + bv.w;
                v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; v2 = v2 > 0.0f ? v2 : 0.0f; v3 = v3 > 0.0f ? v3 : 0.0f;
                if (n0 < 32) {
                    uint2 o;
                    o.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
                    o.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
                    *(uint2*)(th + r * 64 + n0 * 2) = o;
                } else if (n0 == 32) {
                    float* ov = (float*)(th + ROWS * 64) + r * 3;
                    ov[0] = v0; ov[1] = v1; ov[2] = v2;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < ROWS * 4; i += NT)
            if (row0 + (i >> 2) < M) *(u32x4*)(hp + (size_t)row0 * 32 + i * 8) = *(const u32x4*)(th + i * 16);
        for (int i = tid; i < ROWS * 3 / 4; i += NT)
            if (row0 + (i * 4) / 3 < M) *(u32x4*)(hv + (size_t)row0 * 3 + i * 4) = *(const u32x4*)(th + ROWS * 64 + i * 16);
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
    static bool attr_set = false;
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
    static bool attr_set = false;
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

static unsigned long long* g_tower_dbg = nullptr;     // diagnostic builds: per-workgroup clock stamps
void nn_set_tower_dbg(unsigned long long* p) { g_tower_dbg = p; }

// cluster tower (small batches): returns false when the grid could not be resident at once (the caller then
// runs the per-layer path)

template <int GT, int NSPLIT>
static bool tower_cl_launch(hipStream_t st, int device, uint16_t* X, uint16_t* H, const void* wt, const float* bias, int G,
                            uint32_t* sync, uint32_t* err, const void* states, const void* winit, const float* binit, const ClusterHeads& hd,
                            const GrowReq* grow, bool* grown, bool pack, const uint32_t* n_rows_dev, uint32_t* rows_log) {
    constexpr int ROWS = GT * 24, MF = (ROWS + 31) / 32;
    constexpr int lds_a = ((ROWS + 1) * 528 + 16 * 35 + 15) / 16 * 16, lds_p = NSPLIT * MF * 32 * kClusterPartStride;
    constexpr int lds_tower = lds_a + lds_p > 160 * 1024 ? (lds_a > lds_p ? lds_a : lds_p) : lds_a + lds_p;
    constexpr int lds_grow = NSPLIT * kGrowLdsPerWave;
    constexpr int lds_max = lds_tower > lds_grow ? lds_tower : lds_grow;
    static int capacity_of[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};     // per device
    int& capacity = capacity_of[device & 15];
    if (capacity < 0) {
        (void)hipFuncSetAttribute((const void*)k_tower_cl<GT, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        // one workgroup per CU is always admitted (the occupancy API is not asked: with 160 KB of dynamic LDS
        // it answers 0 under some runtimes), and no geometry here needs more than one per CU
        int cus = 0;
        capacity = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess ? cus : 0;
    }
    const int groups = (G + GT - 1) / GT;
    // Few clusters share few XCDs (option cl_pack = 0: one XCD per cluster, round 2's layout): up to 8 clusters on TWO XCDs, up to 16 on four
    // -- at most 4 clusters = 32 workgroups per XCD, one per CU --, so that every XCD that streams the 44.8 MB of weights through its L2
    // does it for up to four clusters (FETCH_SIZE per launch at 4 boards: 45 MB against 181 MB; at 16: 181 against 362) and the
    // workgroups dispatched to the cluster-free XCDs are the growth blocks.  Measured per search iteration: 105.8 vs 108.4 us at 4 boards,
    // 106.4 vs 111.3 at 8, 108.1 vs 110.6 at 16 (one XCD for 4 boards: 108.1; four XCDs for 8: 107.8; profiles/r03s_cl_pack_*).
    int nx = (!pack || n_rows_dev) ? 8 : groups <= 8 ? 2 : groups <= 16 ? 4 : 8;      // (rows counted on the device: any number of the G clusters may run)
    if (64 * ((groups + nx - 1) / nx) > capacity) nx = 8;  // (a device with fewer CUs than the packed grid dispatches: one XCD per cluster)
    const int grid = 64 * ((groups + nx - 1) / nx);
    if (grid > capacity || groups > kClusterMaxGroups) {
        static bool told = false;
        if (!told) fprintf(stderr, "[diee] cluster tower <%d>: %d boards need %d resident workgroups, the device holds %d: using per-layer kernels\n", GT, G, grid, capacity);
        told = true;
        return false;
    }
    // growth blocks ride along when the whole grid still fits the chip (one workgroup per CU): NSPLIT slots per block
    int extra = 0;
    bool ride = false;                                      // packed: the workgroups of the cluster-free XCDs grow (no extra blocks)
    if (grown) *grown = false;
    if (grow && grow->n > 0) {
        const int want = ((int)grow->n + NSPLIT - 1) / NSPLIT;
        if (nx < 8) ride = want <= (grid / 8) * (8 - nx);
        else if (grid + want <= capacity) extra = want;
        if (grown) *grown = ride || extra > 0;
    }
    const GrowReq none{};
    hipLaunchKernelGGL((k_tower_cl<GT, NSPLIT>), dim3(grid + extra), dim3(64 * NSPLIT), (extra || ride) ? lds_max : lds_tower, st, X, H, (const u32x4*)wt, bias, G * 24, groups,
                       sync, err, g_tower_dbg, (const BgState*)states, (const u32x4*)winit, binit, hd, (extra || ride) ? *grow : none, extra ? grid : 0, nx,
                       n_rows_dev, rows_log);
    return true;
}
// whead != nullptr: the launch also runs the head convs and the policy FC (hv / logits are written; X holds no output then)
bool launch_tower_cluster(hipStream_t st, int device, int boards_per_group, uint16_t* X, uint16_t* H, const void* wt, const float* bias,
                          int G, uint32_t* sync, uint32_t* err, const void* states, const void* winit, const float* binit,
                          const void* whead, const float* bhead, const void* wfc, const float* bfc, float* hv, float* logits,
                          const GrowReq* grow, bool* grown, bool pack, const uint32_t* n_rows_dev, uint32_t* rows_log) {
    const ClusterHeads hd{(const u32x4*)whead, bhead, (const u32x4*)wfc, bfc, hv, logits};
    if (grown) *grown = false;
    switch (boards_per_group) {
#if DIEE_CL_SPLIT4_SMALL      // timing experiment (another summation order than the per-layer reference): K split over 4 waves, one per SIMD, at 1 / 2 boards per cluster
        case 1: return tower_cl_launch<1, 4>(st, device, X, H, wt, bias, G, sync, err, states, winit, binit, hd, grow, grown, pack, n_rows_dev, rows_log);
        case 2: return tower_cl_launch<2, 4>(st, device, X, H, wt, bias, G, sync, err, states, winit, binit, hd, grow, grown, pack, n_rows_dev, rows_log);
#else
        case 1: return tower_cl_launch<1, 8>(st, device, X, H, wt, bias, G, sync, err, states, winit, binit, hd, grow, grown, pack, n_rows_dev, rows_log);
        case 2: return tower_cl_launch<2, 8>(st, device, X, H, wt, bias, G, sync, err, states, winit, binit, hd, grow, grown, pack, n_rows_dev, rows_log);
#endif
        case 4: return tower_cl_launch<4, 8>(st, device, X, H, wt, bias, G, sync, err, states, winit, binit, hd, grow, grown, pack, n_rows_dev, rows_log);
        case 8: return tower_cl_launch<8, 4>(st, device, X, H, wt, bias, G, sync, err, states, winit, binit, hd, grow, grown, pack, n_rows_dev, rows_log);     // K split over 4 waves (one per SIMD)
        default: return false;
    }
}

// the whole tower in one launch; x_in/x_out may alias.  geometry 0: 4 boards x (4 waves x 2 N-fragments),
// 1: 2 boards x (8 waves x 1 N-fragment, 18 weight fragments in flight) for mid-size batches
template <int GT, int NF, int PF>
static void tower_launch(hipStream_t st, const uint16_t* x_in, const void* wt, const float* bias, uint16_t* x_out, int G) {
    static bool attr_set = false;
    constexpr int tile = ((GT * 24 + 1) * 528 + 16 * 34 + 128 + 15) / 16 * 16;
    constexpr int lds = 2 * tile;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_tower<GT, NF, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_tower<GT, NF, PF>), dim3((G + GT - 1) / GT), dim3(64 * (8 / NF)), lds, st, x_in,
                       (const u32x4*)wt, bias, x_out, G * 24);
}

template <int GT, int NW, int PF, int BAND = 0>
static void tower16_launch(hipStream_t st, const uint16_t* x_in, const void* wt, const float* bias, uint16_t* x_out, int G,
                           const void* states = nullptr, const void* winit16 = nullptr, const float* binit = nullptr,
                           const void* whead16 = nullptr, const float* bhead = nullptr, uint16_t* hp = nullptr, float* hv = nullptr,
                           const RowMap rm = RowMap{nullptr, nullptr, 0, 0}) {
    static bool attr_set = false;
    constexpr int tile = ((GT * 24 + 1) * 528 + 16 * 34 + 128 + 15) / 16 * 16;
    constexpr int lds = 2 * tile;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_tower16<GT, NW, PF, BAND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_tower16<GT, NW, PF, BAND>), dim3((G + GT - 1) / GT), dim3(64 * NW), lds, st, x_in,
                       (const u32x4*)wt, bias, x_out, G * 24, rm, g_tower_dbg, (const BgState*)states, (const u32x4*)winit16, binit,
                       (const u32x4*)whead16, bhead, hp, hv);
}

// pair tower: 4 boards per pair of workgroups; the grid must be resident at once (one workgroup per CU, at most 256)
template <int GT, int PF>
static void tower16p_launch(hipStream_t st, const void* wt16, const float* bias, int G, const void* states, const void* winit16,
                            const float* binit, const void* whead16, const float* bhead, uint16_t* hp, float* hv, uint16_t* ex, uint32_t* err,
                            const RowMap rm = RowMap{nullptr, nullptr, 0, 0}) {
    static bool attr_set = false;
    constexpr int tile = ((GT * 24 + 1) * 528 + 16 * 34 + 128 + 15) / 16 * 16;
    constexpr int lds = 2 * tile;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_tower16p<GT, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int groups = (G + GT - 1) / GT;
    const int grid = 16 * ((groups + 7) / 8);                     // blockIdx -> (xcd, member, group): whole octets of pairs
    hipLaunchKernelGGL((k_tower16p<GT, PF>), dim3(grid), dim3(256), lds, st, (const u32x4*)wt16, bias, G * 24, rm, (const BgState*)states,
                       (const u32x4*)winit16, binit, (const u32x4*)whead16, bhead, hp, hv, ex, err);
}
bool launch_tower_pair(hipStream_t st, int boards_per_pair, const void* wt16, const float* bias, int G, const void* states, const void* winit16,
                       const float* binit, const void* whead16, const float* bhead, uint16_t* hp, float* hv, uint16_t* ex, uint32_t* err) {
    if (G > boards_per_pair * kPairMaxGroups) return false;
    if (boards_per_pair == 4) tower16p_launch<4, DIEE_PAIR_PF>(st, wt16, bias, G, states, winit16, binit, whead16, bhead, hp, hv, ex, err);
    else if (boards_per_pair == 2) tower16p_launch<2, DIEE_PAIR_PF>(st, wt16, bias, G, states, winit16, binit, whead16, bhead, hp, hv, ex, err);
    else return false;
    return true;
}
bool tower_pair_device_ok(int device) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return false;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) return false;
    // gfx950 in SPX mode: 256 CUs = 8 XCDs x 32; CPX / DPX partitions report 32 / 128 CUs (one / four XCDs)
    return cus == 256 && strncmp(pr.gcnArchName, "gfx950", 6) == 0;
}
size_t tower_pair_exchange_bytes() { return 2 * (size_t)kPairMaxGroups * 2 * kPairHalfBytes; }
int tower_pair_max_boards(int boards_per_pair) { return boards_per_pair * kPairMaxGroups; }

// The fused tower over a batch compacted on the device (see RowMap): up to three launches, each workgroup decides from
// *n_rows whether it has work.  n_upper = the host's upper bound of n_rows (the number of live slots).
void launch_tower_compact(hipStream_t st, const void* wt16, const float* bias, int n_upper, const void* states,
                          const void* winit16, const float* binit, const void* whead16, const float* bhead, uint16_t* hp, float* hv,
                          const uint32_t* row_slot, const uint32_t* n_rows, uint16_t* pair_ex, uint32_t* err) {
    const int tail = n_upper % kFullChip;
    const int main_cap = n_upper < kFullChip ? 0 : (tail > kFullRest ? n_upper : n_upper - tail);
    if (main_cap > 0)
        tower16_launch<4, 4, 3>(st, nullptr, wt16, bias, nullptr, main_cap, states, winit16, binit, whead16, bhead, hp, hv,
                                RowMap{row_slot, n_rows, 1, main_cap});
    // the remainder launches go out whatever n_upper is: n_rows may fall short of it by any amount
    const int rest_max = n_upper < kFullChip ? n_upper : kFullChip - 1;
    // (one wave per SIMD wins where most CUs are busy -- 525 vs 540 us at 768 boards, 489 vs 472 at 520: profiles/r04h_four_wave_probe.txt --;
    // the host picks by its upper bound of the rows, the same bits either way)
    if (rest_max > kFourWaveMin)
        tower16_launch<4, 4, 3, 1>(st, nullptr, wt16, bias, nullptr, rest_max, states, winit16, binit, whead16, bhead, hp, hv,
                                RowMap{row_slot, n_rows, 2, main_cap});
    else if (rest_max > kRemSplit)
        tower16_launch<4, 8, 6>(st, nullptr, wt16, bias, nullptr, rest_max, states, winit16, binit, whead16, bhead, hp, hv,
                                RowMap{row_slot, n_rows, 2, main_cap});
    if (pair_ex)      // the remainder of at most kRemSplit boards: the pair tower
        tower16p_launch<4, DIEE_PAIR_PF>(st, wt16, bias, rest_max < kRemSplit ? rest_max : kRemSplit, states, winit16, binit, whead16, bhead, hp, hv,
                           pair_ex, err, RowMap{row_slot, n_rows, 3, main_cap});
    else
        tower16_launch<2, 8, 9>(st, nullptr, wt16, bias, nullptr, rest_max < kRemSplit ? rest_max : kRemSplit, states, winit16,
                                binit, whead16, bhead, hp, hv, RowMap{row_slot, n_rows, 3, main_cap});
}
// geometry 0/1: 32x32x16 MFMA (wt = 32-column fragments); 2..9, 14: 16x16x32 MFMA (wt16 = 16-column fragments):
// 3 = 2 boards x 8 waves, 4/5 = 3/4 boards x 4 waves, 7/8 = 3/4 boards x 8 waves (the dispatch table uses 5 and 14 = 5 again, since round 4)
void launch_tower(hipStream_t st, int geometry, const uint16_t* x_in, const void* wt, const void* wt16, const float* bias,
                  uint16_t* x_out, int G, const void* states, const void* winit16, const float* binit,
                  const void* whead16, const float* bhead, uint16_t* hp, float* hv) {
    switch (geometry) {
        case 0: tower_launch<4, 2, 9>(st, x_in, wt, bias, x_out, G); break;
        case 1: tower_launch<2, 1, 18>(st, x_in, wt, bias, x_out, G); break;
        case 2: tower16_launch<4, 4, 6>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        case 5: tower16_launch<4, 4, 3>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        case 3: tower16_launch<2, 8, 9>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        case 6: tower16_launch<4, 8, 6>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;     // 4 boards, 8 waves (2 per SIMD)
        case 7: tower16_launch<3, 8, 6>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        case 8: tower16_launch<4, 8, 3>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        case 9: tower16_launch<3, 8, 3>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        // (10 / 11 are the pair tower, nn_host.cpp.)  14: geometry 5 again, instantiated for the band below one pass of the chip (BAND)
        case 14: tower16_launch<4, 4, 3, 1>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        default: tower16_launch<3, 4, 6>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
    }
}
// (kept beside the switch above: the probe's names are the instantiations it launches, as rocprofv3 prints them)
const char* tower_geometry_name(int geometry) {
    switch (geometry) {
        case 0: return "k_tower<4, 2, 9>";
        case 1: return "k_tower<2, 1, 18>";
        case 2: return "k_tower16<4, 4, 6, 0>";
        case 5: return "k_tower16<4, 4, 3, 0>";
        case 3: return "k_tower16<2, 8, 9, 0>";
        case 6: return "k_tower16<4, 8, 6, 0>";
        case 7: return "k_tower16<3, 8, 6, 0>";
        case 8: return "k_tower16<4, 8, 3, 0>";
        case 9: return "k_tower16<3, 8, 3, 0>";
        case 14: return "k_tower16<4, 4, 3, 1>";
        case -1: return "k_tower16 (compacted: <4, 4, 3, 0> / <4, 4, 3, 1> / <4, 8, 6, 0> / k_tower16p<4, 6> by the device-side row count)";
        default: return "k_tower16<3, 4, 6, 0>";
    }
}
// geometries 2..9 can run the init block themselves (states != nullptr); 0 / 1 (32x32x16) need it launched in front
bool tower_geometry_has_init(int geometry) { return geometry >= 2; }
bool tower_geometry_is_full_chip(int geometry) { return geometry == 5 || geometry == 8; }    // 4 boards per workgroup, the instantiation of the full-chip band

void nn_setup_kernels() {}

void launch_planes_bf16(hipStream_t st, const void* states, uint32_t n, uint16_t* out) {
    if (!n) return;
    hipLaunchKernelGGL(k_planes_bf16, dim3((n * 24 + 255) / 256), dim3(256), 0, st, (const BgState*)states, n, out);
}

static int g_conv_variant = -1;     // development override (diee_dev_conv_bench): <= 0 auto, else a fixed geometry id
void nn_set_conv_variant(int v) { g_conv_variant = v; }

// geometry by batch size: keep ~>=256 workgroups in flight while boards per workgroup (weight reuse) stay high
static int pick_variant(int G) {
    if (g_conv_variant > 0) return g_conv_variant;
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
        case 1: conv_launch<256, MODE, 8, 4>(st, act, wpack, bias, res, out, out_v, G, N); break;
        case 2: conv_launch<256, MODE, 4, 4>(st, act, wpack, bias, res, out, out_v, G, N); break;
        case 3: conv_launch<256, MODE, 2, 2>(st, act, wpack, bias, res, out, out_v, G, N); break;
        case 4: conv_launch<256, MODE, 2, 1>(st, act, wpack, bias, res, out, out_v, G, N); break;
        case 5: conv_sk_launch<MODE, 2>(st, act, wpack, bias, res, out, G, N); break;
        case 6: conv_sk_launch<MODE, 4>(st, act, wpack, bias, res, out, G, N); break;
        case 17: conv_sk_launch<MODE, 2, 8>(st, act, wpack, bias, res, out, G, N); break;   // split-K over 8 waves
        case 18: conv_sk_launch<MODE, 4, 8>(st, act, wpack, bias, res, out, G, N); break;
        default: conv_sk_launch<MODE, 8>(st, act, wpack, bias, res, out, G, N); break;
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
