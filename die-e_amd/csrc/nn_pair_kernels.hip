// nn_pair_kernels.hip -- the pair tower: the fused tower of nn_fused_kernels.hip on TWO workgroups per board group (129 ... 512 boards).
#include "nn_common.h"

namespace diee {

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

// ---- host launchers ----
// pair tower: 4 boards per pair of workgroups; the grid must be resident at once (one workgroup per CU, at most 256)
template <int GT, int PF>
static void tower16p_launch(hipStream_t st, const void* wt16, const float* bias, int G, const void* states, const void* winit16,
                            const float* binit, const void* whead16, const float* bhead, uint16_t* hp, float* hv, uint16_t* ex, uint32_t* err,
                            const RowMap rm = RowMap{nullptr, nullptr, 0, 0}) {
    static bool attr_set_dev[16] = {};
    int attr_dev = 0;
    (void)hipGetDevice(&attr_dev);
    bool& attr_set = attr_set_dev[attr_dev & 15];             // per device: a ctx on another GPU of this process sets it there too
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
void launch_tower_pair_rows(hipStream_t st, const void* wt16, const float* bias, int G, const void* states, const void* winit16, const float* binit,
                            const void* whead16, const float* bhead, uint16_t* hp, float* hv, uint16_t* ex, uint32_t* err, const RowMap& rm) {
    tower16p_launch<4, DIEE_PAIR_PF>(st, wt16, bias, G, states, winit16, binit, whead16, bhead, hp, hv, ex, err, rm);
}
void launch_tower_pair_counted(hipStream_t st, const void* wt16, const float* bias, int G, const void* states, const void* winit16, const float* binit,
                               const void* whead16, const float* bhead, uint16_t* hp, float* hv, uint16_t* ex, uint32_t* err, const uint32_t* n_rows) {
    // (RowMap mode 3 without a main launch and without a row map: rows [0, *n_rows) are the boards, G <= kRemSplit bounds them)
    tower16p_launch<4, DIEE_PAIR_PF>(st, wt16, bias, G, states, winit16, binit, whead16, bhead, hp, hv, ex, err, RowMap{nullptr, n_rows, 3, 0});
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

}  // namespace diee
