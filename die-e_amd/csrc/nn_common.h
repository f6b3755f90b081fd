// nn_common.h -- what the network's translation units share: the compile-time switches, the MFMA operand types, bf16 conversion,
// the device-coherent accesses and ready tags of the in-launch hand-overs (cluster tower, pair tower), the border-aware fragment
// order of the 4-board fused geometries, and the row map of a compacted evaluation.
//   nn_conv_kernels.hip     per-layer convolutions (init block, heads, fallback tower layers, training), policy FC, softmax + value
//   nn_cluster_kernels.hip  k_tower_cl: the whole network for <= 256 boards in one launch, 8-workgroup clusters
//   nn_fused_kernels.hip    k_tower16: the 38-layer tower in LDS, one workgroup per 2 / 4 boards (> 512 boards)
//   nn_pair_kernels.hip     k_tower16p: the fused tower on pairs of workgroups (129 ... 512 boards)
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdint.h>
#include <cstdio>
#include <type_traits>

#include "bg_device.h"
#include "launch.h"
#include "mcts_device.h"
#include "nn_device.h"

// ---- compile-time switches ---------------------------------------------------------------------------------------------------------
// The PRODUCT build (die-e_amd/build.py, libdiee.so) fixes every switch at the value below and compiles only the kernel
// instantiations the dispatch tables (nn_host.h) and their documented fallbacks can reach.  -DDIEE_DEV_BUILD (scripts/: timing builds,
// A/B libraries) lets the switches be overridden on the command line and adds the superseded / experimental kernels: the 32x32x16
// fused tower k_tower, the k_tower16 geometries no table uses, the per-layer conv variants behind diee_dev_conv_bench.
#if !defined(DIEE_DEV_BUILD) && (defined(DIEE_CL_ABLATE) || defined(DIEE_CL_LATE) || defined(DIEE_CL_LATE_OUT) || defined(DIEE_CL_LATE_SLEEP) || defined(DIEE_CL_PD) || defined(DIEE_CL_POLL_SLEEP) || defined(DIEE_CL_PRS) || defined(DIEE_CL_STORE_AUX) || defined(DIEE_PAIR_ABLATE) || defined(DIEE_PAIR_AHEAD) || defined(DIEE_PAIR_BIAS_EARLY) || defined(DIEE_PAIR_PF) || defined(DIEE_PAIR_POLL_SLEEP) || defined(DIEE_PAIR_RES) || defined(DIEE_PAIR_STORE_AUX) || defined(DIEE_PAIR_UNROLL) || defined(DIEE_REM_SPLIT) || defined(DIEE_TOWER_ABLATE) || defined(DIEE_TOWER_BIAS_EARLY) || defined(DIEE_TOWER_BORDER) || defined(DIEE_TOWER_PRIO) || defined(DIEE_TOWER_UNROLL4))
#error "timing / ablation switches (-DDIEE_...) need -DDIEE_DEV_BUILD: the product library is built with the defaults of nn_common.h"
#endif
#ifndef DIEE_TOWER_BORDER
#define DIEE_TOWER_BORDER 1      // 1 = the 4-board fused tower skips (tap, fragment) pairs that are all zero padding
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
#ifndef DIEE_CL_LATE_OUT
#define DIEE_CL_LATE_OUT DIEE_CL_LATE
#endif
#ifndef DIEE_CL_LATE_SLEEP
#define DIEE_CL_LATE_SLEEP 0      // cluster tower: s_sleep argument (x 64 cycles) of the waves without output chunks in front of their late weight requests
#endif
#ifndef DIEE_CL_ABLATE
#define DIEE_CL_ABLATE 0          // timing experiments on the cluster tower: 1 = no MFMA loop, 2 = no partial-tile exchange, 3 = no weight loads
#endif
#ifndef DIEE_PAIR_BIAS_EARLY
#define DIEE_PAIR_BIAS_EARLY 1
#endif
#ifndef DIEE_TOWER_BIAS_EARLY
#define DIEE_TOWER_BIAS_EARLY 1
#endif
#ifndef DIEE_PAIR_UNROLL
#define DIEE_PAIR_UNROLL 1         // the pair tower's k loop unrolled in full (round 4: 320 ... 390 us against 333 ... 406 at 300 ... 512 boards, profiles/r04h_pair_unroll_ab.txt)
#endif
#ifndef DIEE_TOWER_UNROLL4
#define DIEE_TOWER_UNROLL4 1
#endif
#ifndef DIEE_TOWER_ABLATE
#define DIEE_TOWER_ABLATE 0      // diagnostic builds only: 1 = no main loop, 2 = no epilogue, 3 = in-kernel clock stamps, 4 = cluster tower re-reads two layers' weights, 6 ... 9 = fused tower without its LDS reads / weight loads, 12 = with half its weight loads (tower_layer16)
#endif
#ifndef DIEE_CL_PRS
#define DIEE_CL_PRS 144           // cluster tower: partial-tile row stride in bytes (160, half-waves on disjoint bank halves, measured no faster)
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

constexpr int kTowerLayerStride = 8 * 144 * 64;          // u32x4 per layer (1.18 MB)

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

// diagnostic builds (-DDIEE_TOWER_ABLATE=3): per-workgroup clock stamps of the tower kernels (nn_conv_kernels.hip owns the pointer)
extern unsigned long long* g_tower_dbg;

// the pair tower over the remainder rows of a compacted evaluation (nn_pair_kernels.hip; called by launch_tower_compact)
void launch_tower_pair_rows(hipStream_t st, const void* wt16, const float* bias, int G, const void* states, const void* winit16, const float* binit,
                            const void* whead16, const float* bhead, uint16_t* hp, float* hv, uint16_t* ex, uint32_t* err, const RowMap& rm);

}  // namespace diee
