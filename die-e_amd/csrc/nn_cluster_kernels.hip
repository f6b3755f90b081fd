// nn_cluster_kernels.hip -- the cluster tower: init block, 38 tower layers, head convolutions and policy FC of a batch of at most 256 boards
// in ONE launch (reference: src/alphazero/nnet.rs:24-34,57-107,120-133, eval-mode BatchNorm folded).
#include "nn_common.h"

namespace diee {

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
constexpr int kClusterSpinLimit = 1 << 18;
// partial-tile row stride (bytes); 160 (half-waves of a C-layout store on disjoint bank halves) measured no faster
constexpr int kClusterPartStride = DIEE_CL_PRS;

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
        // Every wave stages exactly the channel columns its own K share reads (wave w <-> input channels
        // [w * 256 / NSPLIT, ...): with K split 8 ways those are the 32 channels ONE producer workgroup wrote), so nothing
        // a wave reads in the MFMA loop was written by another wave: no workgroup barrier between staging and the loop, and a
        // wave starts as soon as ITS producer's slice has landed instead of when the slowest of the eight has.
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

// ---- host launchers ----
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
        case 1: return tower_cl_launch<1, 8>(st, device, X, H, wt, bias, G, sync, err, states, winit, binit, hd, grow, grown, pack, n_rows_dev, rows_log);
        case 2: return tower_cl_launch<2, 8>(st, device, X, H, wt, bias, G, sync, err, states, winit, binit, hd, grow, grown, pack, n_rows_dev, rows_log);
        case 4: return tower_cl_launch<4, 8>(st, device, X, H, wt, bias, G, sync, err, states, winit, binit, hd, grow, grown, pack, n_rows_dev, rows_log);
        case 8: return tower_cl_launch<8, 4>(st, device, X, H, wt, bias, G, sync, err, states, winit, binit, hd, grow, grown, pack, n_rows_dev, rows_log);     // K split over 4 waves (one per SIMD)
        default: return false;
    }
}

}  // namespace diee
