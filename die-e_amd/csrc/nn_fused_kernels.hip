// nn_fused_kernels.hip -- the fused tower: the 38 3x3 convolutions of the ResNet (src/alphazero/nnet.rs:24-34,69-73) in ONE launch, the
// activations of a workgroup's 2 / 4 boards ping-ponging between two LDS tiles; init block in front, head convolutions behind.
#include "nn_common.h"

namespace diee {

#ifdef DIEE_DEV_BUILD   // the fused tower on 32x32x16 MFMAs: superseded by k_tower16 (round 2), bit-identical to the per-layer kernels
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
#endif  // DIEE_DEV_BUILD

template <bool RES, int GT, int NW, int PF, bool SP = false>
__device__ __forceinline__ void tower_layer16(char* tin, char* tout, const u32x4* wp, const u32x4* wp_next,
                                              const float* __restrict__ bias, const uint32_t (&basep)[9][((GT * 24 + 15) / 16 + 1) / 2],
                                              u32x4 (&bq)[PF][16 / NW], int lane, int wave) {
    constexpr int ROWS = GT * 24, MF = (ROWS + 15) / 16, NFR = 16 / NW;
    constexpr int NQ = NFR;
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
            }
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                if (border_skip(SP, u % 9, f)) continue;           // this fragment x tap is all padding
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    acc[f][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[q], a[cur][f], acc[f][q], 0, 0, 0);   // D = W^T x act^T
            }
            // interleave this k-step's loads between its MFMAs instead of issuing them as a block in front
            {
                const int n_mfma = border_live(SP, u % 9, MF) * NFR, n_lds = lds_step ? border_live(SP, un % 9, MF) : 0;
#pragma unroll
                for (int i = 0; i < MF * NFR; ++i) {
                    if (i >= n_mfma) break;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
                    if (i < n_lds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // 1 LDS read
                    else if (!kNoW && i < n_lds + NFR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
                }
            }
            __builtin_amdgcn_sched_barrier(0);
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
        else if (rm.mode == 4) { lo = 0; hi = nr > kRemSplit ? nr : 0; }        // a free-running round of up to one pass: every row, unless the pair tower's launch takes them (mode 3)
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

// ---- host launchers ----
#ifdef DIEE_DEV_BUILD
// the whole tower in one launch; x_in/x_out may alias.  geometry 0: 4 boards x (4 waves x 2 N-fragments),
// 1: 2 boards x (8 waves x 1 N-fragment, 18 weight fragments in flight) for mid-size batches
template <int GT, int NF, int PF>
static void tower_launch(hipStream_t st, const uint16_t* x_in, const void* wt, const float* bias, uint16_t* x_out, int G) {
    static bool attr_set_dev[16] = {};
    int attr_dev = 0;
    (void)hipGetDevice(&attr_dev);
    bool& attr_set = attr_set_dev[attr_dev & 15];             // per device: a ctx on another GPU of this process sets it there too
    constexpr int tile = ((GT * 24 + 1) * 528 + 16 * 34 + 128 + 15) / 16 * 16;
    constexpr int lds = 2 * tile;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_tower<GT, NF, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_tower<GT, NF, PF>), dim3((G + GT - 1) / GT), dim3(64 * (8 / NF)), lds, st, x_in,
                       (const u32x4*)wt, bias, x_out, G * 24);
}
#endif

template <int GT, int NW, int PF, int BAND = 0>
static void tower16_launch(hipStream_t st, const uint16_t* x_in, const void* wt, const float* bias, uint16_t* x_out, int G,
                           const void* states = nullptr, const void* winit16 = nullptr, const float* binit = nullptr,
                           const void* whead16 = nullptr, const float* bhead = nullptr, uint16_t* hp = nullptr, float* hv = nullptr,
                           const RowMap rm = RowMap{nullptr, nullptr, 0, 0}) {
    static bool attr_set_dev[16] = {};
    int attr_dev = 0;
    (void)hipGetDevice(&attr_dev);
    bool& attr_set = attr_set_dev[attr_dev & 15];             // per device: a ctx on another GPU of this process sets it there too
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
        launch_tower_pair_rows(st, wt16, bias, rest_max < kRemSplit ? rest_max : kRemSplit, states, winit16, binit, whead16, bhead, hp, hv,
                               pair_ex, err, RowMap{row_slot, n_rows, 3, main_cap});
    else
        tower16_launch<2, 8, 9>(st, nullptr, wt16, bias, nullptr, rest_max < kRemSplit ? rest_max : kRemSplit, states, winit16,
                                binit, whead16, bhead, hp, hv, RowMap{row_slot, n_rows, 3, main_cap});
}
// A round of the free-running search (search_types.h, Free): up to n_upper <= 1024 rows gathered by index, their number on the device.  TWO launches,
// one of which has work: the one-pass geometry when there are more than kRemSplit rows (the part-filled instantiation is the same code: no third launch
// whose workgroups would only look at the count and leave, ~5 us a round), else the pair tower.
void launch_tower_free(hipStream_t st, const void* wt16, const float* bias, int n_upper, const void* states, const void* winit16, const float* binit,
                       const void* whead16, const float* bhead, uint16_t* hp, float* hv, const uint32_t* row_idx, const uint32_t* n_rows, uint16_t* pair_ex, uint32_t* err) {
    if (n_upper > kRemSplit)
        tower16_launch<4, 4, 3>(st, nullptr, wt16, bias, nullptr, n_upper < kFullChip ? n_upper : kFullChip, states, winit16, binit, whead16, bhead, hp, hv,
                                RowMap{row_idx, n_rows, 4, 0});
    const int rest_max = n_upper < kRemSplit ? n_upper : kRemSplit;
    if (pair_ex)
        launch_tower_pair_rows(st, wt16, bias, rest_max, states, winit16, binit, whead16, bhead, hp, hv, pair_ex, err, RowMap{row_idx, n_rows, 3, 0});
    else
        tower16_launch<2, 8, 9>(st, nullptr, wt16, bias, nullptr, rest_max, states, winit16, binit, whead16, bhead, hp, hv, RowMap{row_idx, n_rows, 3, 0});
}
// The geometries of the fused tower (wt16 = 16-column fragments, 16x16x32 MFMA).  PRODUCT build -- what the dispatch tables and their
// fallbacks launch: 5 = 4 boards x 4 waves (one per SIMD, k loop unrolled) for one pass of the chip, 14 = the same code instantiated
// again for 641 ... 928 boards, 6 = 4 boards x 8 waves (513 ... 640), 3 = 2 boards x 8 waves (the pair tower's fallback, the remainder of
// a compacted batch without it, DIEE_FLAG_INVARIANT_NN below 257 boards).  -DDIEE_DEV_BUILD adds 0 / 1 (k_tower on 32x32x16, wt =
// 32-column fragments), 2, 7, 8, 9 and the 3-board default that rounds 1-3 measured against them.
bool tower_geometry_supported(int geometry) {
#ifdef DIEE_DEV_BUILD
    return geometry >= 0;
#else
    return geometry == 3 || geometry == 5 || geometry == 6 || geometry == 14;
#endif
}
void launch_tower(hipStream_t st, int geometry, const uint16_t* x_in, const void* wt, const void* wt16, const float* bias,
                  uint16_t* x_out, int G, const void* states, const void* winit16, const float* binit,
                  const void* whead16, const float* bhead, uint16_t* hp, float* hv) {
    switch (geometry) {
        case 5: tower16_launch<4, 4, 3>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        case 14: tower16_launch<4, 4, 3, 1>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;   // geometry 5 again, instantiated for the band below one pass of the chip (BAND)
        case 6: tower16_launch<4, 8, 6>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;     // 4 boards, 8 waves (2 per SIMD)
#ifdef DIEE_DEV_BUILD
        case 0: tower_launch<4, 2, 9>(st, x_in, wt, bias, x_out, G); break;
        case 1: tower_launch<2, 1, 18>(st, x_in, wt, bias, x_out, G); break;
        case 2: tower16_launch<4, 4, 6>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        case 7: tower16_launch<3, 8, 6>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        case 8: tower16_launch<4, 8, 3>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        case 9: tower16_launch<3, 8, 3>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
        case 4: tower16_launch<3, 4, 6>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
#endif
        // (10 / 11 are the pair tower, nn_host.cpp.)  Anything else -- Engine::set_option refuses tables that name it -- runs geometry 3.
        default: (void)wt; tower16_launch<2, 8, 9>(st, x_in, wt16, bias, x_out, G, states, winit16, binit, whead16, bhead, hp, hv); break;
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
        case 4: return "k_tower16<3, 4, 6, 0>";
        default: return "k_tower16<2, 8, 9, 0>";
    }
}
// geometries 2..9 can run the init block themselves (states != nullptr); 0 / 1 (32x32x16) need it launched in front
bool tower_geometry_has_init(int geometry) { return geometry >= 2; }     // (0 / 1, development build: the init block is launched in front)
bool tower_geometry_is_full_chip(int geometry) { return geometry == 5 || geometry == 8; }    // 4 boards per workgroup, the instantiation of the full-chip band

}  // namespace diee
