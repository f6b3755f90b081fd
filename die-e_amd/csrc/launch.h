// launch.h -- host-side launch wrappers, one group per kernel translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "search_types.h"

namespace diee {

// rules_kernels.hip
void launch_legal_moves(hipStream_t st, const void* states, uint32_t n, uint32_t* plays, uint32_t cap,
                        uint32_t* counts, uint32_t* overflow);
void launch_encode(hipStream_t st, const void* states, const uint32_t* plays, uint32_t n, uint32_t* codes);
void launch_decode(hipStream_t st, const void* states, const uint32_t* codes, uint32_t n, uint32_t* plays);
void launch_apply(hipStream_t st, void* states, const uint32_t* plays, const uint8_t* dice, uint32_t n);
void launch_planes(hipStream_t st, const void* states, uint32_t n, float* out);
void launch_wave_selftest(hipStream_t st, uint32_t* mismatches, uint32_t salt);
void launch_probe_f32(hipStream_t st, const float* a, const float* b, uint32_t n, float* sq, float* dv, float* pw);
void launch_probe_dice(hipStream_t st, uint64_t seed, const uint32_t* ctr, uint32_t n, uint8_t* dice, double* uni);

// The network-independent half of expansion `it` (grow_slot, mcts_device.h) for the n slots of a search, as a request a network launch
// may take along: the cluster tower runs it on extra workgroups of its own launch while the chip has CUs to spare.
struct GrowReq { Tree T; Slots S; Segs G; uint32_t n, it; };

// nn_conv_kernels.hip / nn_cluster_kernels.hip / nn_fused_kernels.hip / nn_pair_kernels.hip (shared device code: nn_common.h)
void nn_setup_kernels();
void nn_set_conv_variant(int v);   // 0 = pick by batch size, 1..4 = fixed geometry (development)
void launch_planes_bf16(hipStream_t st, const void* states, uint32_t n, uint16_t* out);
void launch_conv3x3(hipStream_t st, int c_in, int mode, const uint16_t* act, const void* wpack, const float* bias,
                    const uint16_t* res, uint16_t* out, float* out_v, int G, int N);
void launch_tower(hipStream_t st, int geometry, const uint16_t* x_in, const void* wt, const void* wt16, const float* bias,
                  uint16_t* x_out, int G, const void* states = nullptr, const void* winit16 = nullptr, const float* binit = nullptr,
                  const void* whead16 = nullptr, const float* bhead = nullptr, uint16_t* hp = nullptr, float* hv = nullptr);
bool tower_geometry_supported(int geometry);       // compiled into this build (the product build holds 3, 5, 6, 14; -DDIEE_DEV_BUILD all)
const char* tower_geometry_name(int geometry);    // "k_tower16<4, 4, 3, 0>" ...: the instantiation launch_tower sends for a geometry
bool tower_geometry_is_full_chip(int geometry);   // the geometry (and instantiation) the dispatch uses for one pass of the chip
bool tower_geometry_has_init(int geometry);   // the fused geometry can run the init block (states != nullptr) and the
                                              // head convs (whead16 != nullptr: hp / hv are written, x_out is not) itself
void nn_set_tower_dbg(unsigned long long* p);
// cluster tower: 38 layers in one launch for small batches; `sync` = kClusterMaxGroups counters 128 B apart (zeroed),
// `err` gets bit 2 set if a cluster wait timed out.  false = not launched (grid would not be co-resident).
constexpr int kClusterMaxGroups = 64;
// `states` non-null: the init block runs inside the launch (winit / binit = its fragments and bias); X then holds no input
// `whead` non-null: the head convs (two 32-column slices, wconv[39]) and the policy FC run inside the launch too: hv / logits are written
bool launch_tower_cluster(hipStream_t st, int device, int boards_per_group, uint16_t* X, uint16_t* H, const void* wt, const float* bias,
                          int G, uint32_t* sync, uint32_t* err, const void* states, const void* winit, const float* binit,
                          const void* whead = nullptr, const float* bhead = nullptr, const void* wfc = nullptr, const float* bfc = nullptr,
                          float* hv = nullptr, float* logits = nullptr, const GrowReq* grow = nullptr, bool* grown = nullptr, bool pack = true,
                          const uint32_t* n_rows_dev = nullptr, uint32_t* rows_log = nullptr);
                          // n_rows_dev: the rows to evaluate are counted on the device (<= G; 0: the launch returns at once); rows_log: where to note them
                          // grow: also grow the tree on extra workgroups if the whole grid stays resident (*grown tells)
// train_kernels.hip (token layout [M][256] bf16; `partial` = train_stripes(M) * 768 + 1280 floats of scratch)
int train_stripes(int M);
void launch_bn_relu_fwd(hipStream_t st, const uint16_t* x, const uint16_t* res, const float* gamma, const float* beta, float* partial,
                        float* save_mean, float* save_invstd, float* run_mean, float* run_var, float momentum, float eps,
                        uint16_t* y, int M);
void launch_bn_relu_bwd(hipStream_t st, const uint16_t* dy, const uint16_t* y, const uint16_t* x, const float* gamma,
                        const float* mean, const float* invstd, float* partial, float* dgamma, float* dbeta, uint16_t* dx,
                        uint16_t* dres, float* dx_colsum /*[256] or nullptr*/, int M);
void bn_coop_set(int on);
int bn_coop_get();
int bn_coop_poll_timeouts(int clear);     // bit 0 forward, bit 1 backward pass timed out since the last clear; -1: query failed
size_t wgrad_scratch_floats();
void launch_wgrad3x3(hipStream_t st, const uint16_t* x, const uint16_t* dy, float* partial, float* dw, int boards);
void launch_colsum(hipStream_t st, const uint16_t* a, float* partial, float* out, int M);
void launch_pack_conv_w(hipStream_t st, const float* w_oihw, uint16_t* wpack, int transpose);
constexpr int kMaxPackLayers = 64;
// w: host array of n (<= kMaxPackLayers) device pointers; wpack [n][2 = forward, transposed][589 824] bf16
void launch_pack_conv_w_multi(hipStream_t st, const float* const* w, int n, uint16_t* wpack);
void launch_im2col3x3(hipStream_t st, const uint16_t* x, uint16_t* col, int boards);
void launch_policy_fc(hipStream_t st, const uint16_t* hp, const void* wpack, const float* bias, float* logits, int G,
                      const uint32_t* n_rows = nullptr);   // n_rows non-null: device-side row count of a compacted batch (<= G)
// fused tower over a batch compacted on the device: row_slot[row] = slot to evaluate, *n_rows rows (<= n_upper); see RowMap
void launch_tower_compact(hipStream_t st, const void* wt16, const float* bias, int n_upper, const void* states,
                          const void* winit16, const float* binit, const void* whead16, const float* bhead, uint16_t* hp, float* hv,
                          const uint32_t* row_slot, const uint32_t* n_rows, uint16_t* pair_ex = nullptr, uint32_t* err = nullptr);
// the same for a round of the free-running search (<= 1024 rows gathered from the tree arena by index): two launches, one with work
void launch_tower_free(hipStream_t st, const void* wt16, const float* bias, int n_upper, const void* states, const void* winit16, const float* binit,
                       const void* whead16, const float* bhead, uint16_t* hp, float* hv, const uint32_t* row_idx, const uint32_t* n_rows,
                       uint16_t* pair_ex = nullptr, uint32_t* err = nullptr);
// pair tower (k_tower16p): the fused tower with 4 boards per PAIR of workgroups, 257 ... 512 boards; `ex` = tower_pair_exchange_bytes()
// of zeroed device memory, `err` gets bit 2 set if a hand-over timed out.  false = too many boards.
// the 4-board pair tower over the first *n_rows (counted on the device, <= G) of G dense rows: a tail launch at 129 ... 256 live games
void launch_tower_pair_counted(hipStream_t st, const void* wt16, const float* bias, int G, const void* states, const void* winit16, const float* binit,
                               const void* whead16, const float* bhead, uint16_t* hp, float* hv, uint16_t* ex, uint32_t* err, const uint32_t* n_rows);
bool launch_tower_pair(hipStream_t st, int boards_per_pair /* 4 or 2 */, const void* wt16, const float* bias, int G, const void* states, const void* winit16,
                       const float* binit, const void* whead16, const float* bhead, uint16_t* hp, float* hv, uint16_t* ex, uint32_t* err);
bool tower_pair_device_ok(int device);     // 8 XCDs x >= 32 CUs: the pair's co-placement (blockIdx & 7) and co-residency (<= 256 workgroups) hold
size_t tower_pair_exchange_bytes();
int tower_pair_max_boards(int boards_per_pair);
void launch_softmax_value(hipStream_t st, const float* logits, const float* hv, const float* wv, float* policy,
                          float* value, int G);

// mcts_kernels.hip
void launch_init_roots(hipStream_t st, const Tree& T, const Slots& S, uint32_t n);
struct ExpandVariant { bool two = true, two_c = true; };    // options expand2 / expand2c
void launch_expand(hipStream_t st, const Tree& T, const Slots& S, const Segs& G, uint32_t n, uint32_t it, const SearchParams& P,
                   uint32_t next_it, float c, bool pre_grown = false, ExpandVariant v = ExpandVariant{});   // next_it: iteration to select for afterwards, kNoNextIteration = none;
                                                                         // pre_grown: the tower launch's growth workgroups created the children already (GrowReq)
// the tail of a batch (search_types.h, Tail): launch number q of a move-step's search -- takes in the rows of tower launch q - 1, runs
// iterations while every live game's selected leaf has its evaluation, plans the rows of tower launch q
void launch_tail(hipStream_t st, const Tree& T, const Slots& S, const Segs& G, uint32_t n, const SearchParams& P, float c, const Tail& L, uint32_t q);
// the free-running search (search_types.h, Free): round q -- k_free takes in the rows of tower launch q - 1, runs every game's own iterations,
// lists its wishes; k_free_pack grants the rows of tower launch q
void launch_free(hipStream_t st, const Tree& T, const Slots& S, const Segs& G, uint32_t n, const SearchParams& P, float c, const Free& F, uint32_t q);
uint32_t free_lds_nodes_for(uint32_t n, uint32_t cus);      // nodes of a game's tree k_free stages in LDS when n games share `cus` CUs
void launch_reduce_counters(hipStream_t st, const Slots& S, const Segs& G, unsigned long long* step_log = nullptr, uint32_t step = 0);
// rows of the next network evaluation: the slots with skip[slot] == 0, in slot order (row_slot / slot_row / *n_rows; the
// count also goes to rows_log[log_idx])
void launch_row_map(hipStream_t st, const uint8_t* skip, uint32_t n, uint32_t* row_slot, uint32_t* slot_row, uint32_t* n_rows,
                    uint32_t* rows_log, uint32_t log_idx);
void launch_root_probs(hipStream_t st, const Tree& T, uint32_t n, float* probs, uint32_t* nch, float* root_visits);
void launch_init_games(hipStream_t st, const Games& Gm, const Segs& G, uint32_t n);
void launch_gather_roots(hipStream_t st, const Games& Gm, const Slots& S, const Segs& G, uint32_t n_live);
void launch_play_move(hipStream_t st, const Tree& T, const Games& Gm, const Segs& G, uint32_t n_live, uint32_t step, const PlayParams& P);
void launch_compact_live(hipStream_t st, const Games& Gm, uint32_t n_live, uint32_t n_segs, uint32_t* n_live_out);
// output delivery of a move-step: the step's flushes in the reference's order (summary: DeliverSummary), then ranges of their rows
void launch_deliver_scan(hipStream_t st, const Games& Gm, const Segs& G, uint32_t n_live, uint32_t step, DeliverEvent* ev, uint32_t* summary);
void launch_deliver_copy(hipStream_t st, const Games& Gm, const Segs& G, const DeliverEvent* ev, uint32_t n_ev, uint32_t r0, uint32_t r1,
                         const DeliverOut& out);
constexpr uint32_t kRootIteration = 0xFFFFFFFFu;
constexpr uint32_t kNoNextIteration = 0xFFFFFFFEu;

}  // namespace diee
