// nn_host.h -- device-resident network state shared between nn_host.cpp and the search driver.
#pragma once
#include <vector>

#include "engine.h"

namespace diee {

struct GrowReq;

struct NetWeights {
    DevBuf<uint16_t> wconv[40];     // packed bf16 B fragments: 0 init, 39 heads (1..38 live in wtower)
    DevBuf<float> bconv[40];        // folded bias (1..38 live in btower)
    DevBuf<uint16_t> wtower;        // [38][8][144][64][8] bf16: the tower layers, contiguous (fused tower kernel)
    DevBuf<uint16_t> wtower16;      // [38][16][72][64][8] bf16: the same weights as 16-column fragments (16x16x32 MFMA)
    DevBuf<float> btower;           // [38][256]
    DevBuf<uint16_t> winit16, whead16;   // init block [16][9][64][8] and head convs [4][72][64][8] as 16-column fragments
    uint16_t* wl(int layer) { return (layer >= 1 && layer <= 38) ? wtower.p + (size_t)(layer - 1) * 8 * 144 * 64 * 8 : wconv[layer].p; }
    float* bl(int layer) { return (layer >= 1 && layer <= 38) ? btower.p + (size_t)(layer - 1) * 256 : bconv[layer].p; }
    // fused-tower dispatch: the first entry with G > min_games wins; batches below every entry run per-layer kernels.
    // Measured on MI355X (scripts/fwd_sweep*.py, scripts/tower_clock.py): 16x16x32 MFMA, 8 waves per workgroup (two per
    // SIMD: one wave's loads overlap the other's MFMAs: 79-81 % MFMA issue efficiency vs 62-67 % with one wave per SIMD),
    // 4 boards per workgroup above 416 boards (border-aware fragment order: 22 % of the MFMAs are padding and not issued;
    // 3 weight k-steps in flight above 928 boards, 6 below: within 1 % of each other), 2 boards above 256 (below that: the cluster tower).  option tower_table = "min:geom,min:geom" overrides ("none" disables).
    struct TowerRule { int min_games, geometry; };
    // Round 4: ONE wave per SIMD with four column fragments (geometry 5 = k_tower16<4,4,3>; 14 = the same code instantiated again for
    // the band below one pass of the chip) replaces the 8-wave geometries 8 / 6: half the A-fragment LDS reads at the same weight
    // traffic; its k loop is unrolled in full -- with the loop the accumulators (in AGPRs) were permuted across the back edge, 132
    // v_accvgpr moves per 18 k-steps, which is what "62-67 %" above measured: 603 vs 634 us at 1024 boards, 514 vs 531 at 768.
    // Below ~640 boards (fewer than 160 of 256 CUs busy: no power limit to give back to) the 8-wave geometry 6 is still the faster one.
    static std::vector<TowerRule> default_tower_table() { return {{928, 5}, {640, 14}, {512, 6}, {256, 10}, {40, 11}}; }   // 10 / 11 = the pair tower (k_tower16p) with 4 boards per
                                    // pair (257 ... 512 boards) / 2 boards per pair (41 ... 256).  Round 6: the fused family starts at 41 boards, not 129 -- a plain
                                    // evaluation of 41 ... 128 boards costs ~258 us on the pair tower against 110 ... 172 on the cluster tower, but it happens once per
                                    // move-step, and it makes the search's launches there 512-row launches of the fused family (the free-running search): 18 ... 33 %
                                    // faster searches at 48 ... 128 live games (profiles/r06j_*, r06l_*)
    static std::vector<TowerRule> default_tower_table_no_pair() { return {{928, 5}, {640, 14}, {416, 6}, {256, 3}}; }   // option tower_pair = 0 / shared_gpu = 1
    std::vector<TowerRule> tower_table = default_tower_table();      // option "tower_table" (Engine::apply_options)
    DevBuf<uint16_t> pair_ex;       // its exchange buffers (zeroed once)
    bool pair_tower = true;         // option "tower_pair" = 0: the 2-board geometry instead (rounds 1-2)
    bool starved = false;           // an in-launch hand-over timed out in this ctx: cluster and pair tower stay off
    bool told_no_pair = false;
    // what the last network evaluation launched for its tower (development probe diee_dev_last_dispatch: the tolerance tests name
    // the kernel they ran): family 0 per-layer kernels, 1 fused tower k_tower16 (geometry = launch_tower's), 2 pair tower k_tower16p
    // (geometry = boards per pair), 3 cluster tower k_tower_cl (geometry = boards per cluster), 4 fused tower on 32x32x16 (k_tower)
    struct Launched { int family, geometry, boards; };
    std::vector<Launched> last_dispatch;
    bool cl_pack = true;            // option "cl_pack": few clusters share few XCDs (launch_tower_cluster)
    bool trace_dispatch = false;    // option "trace_dispatch"
    bool invariant = false;         // DIEE_FLAG_INVARIANT_NN / diee_set_invariant_nn: every batch size on the fused 16x16x32 tower
    int tower_geometry_for(int G) const {
        for (const auto& r : tower_table) if (G > r.min_games) return r.geometry;
        return invariant ? 3 : -1;  // 3 = k_tower16<2 boards, 8 waves>: same arithmetic per output element as the 4-board geometries
    }
    // cluster tower (k_tower_cl): batches of at most max_games boards run the 38 layers in one launch, boards_per_group
    // boards per 8-workgroup cluster (1, 2, 4: K split over 8 waves; 8: over 4 waves); tried in order, a batch no rule
    // takes (or whose grid would not be co-resident) runs per-layer kernels.  Measured (scripts/cluster_check.py), forward
    // of 16 / 64 / 128 / 200 / 256 boards: 121 / 149 / 207 / 318 / 346 us against 250 / 253 / 317 / 564 / 426 before.  option tower_cl = "max:boards,max:boards" overrides ("none" disables).
    struct ClusterRule { int max_games, boards_per_group; };
    static std::vector<ClusterRule> default_cluster_table() { return {{32, 1}, {64, 2}, {128, 4}, {256, 8}}; }
    std::vector<ClusterRule> cluster_table = default_cluster_table();      // option "tower_cl"
    DevBuf<uint32_t> cl_sync;       // [kClusterMaxGroups] counters, 128 B apart
    bool cluster_used = false;      // a cluster launch went out since nn_cluster_used() was asked last
    int full_chip_boards = 1024;    // one pass of the chip through the 4-board fused tower (256 CUs x 4 boards); batches above
                                    // it run whole multiples in one launch and the remainder in a launch of its own (0: never split)
    bool fused_heads = true;        // the fused tower runs the head convs itself (its output tile never leaves the CU)
    bool cluster_init = true;       // the cluster tower runs the init block itself (every workgroup, for its cluster's boards)
    // the search's growth request for the evaluation about to be launched (launch.h GrowReq; null: none) and whether a launch took it
    const struct GrowReq* grow_req = nullptr;
    bool grow_done = false;
    bool cluster_heads = true;      // ... and the head convs and the policy FC (an evaluation below 257 boards = cluster launch + k_expand)
    DevBuf<uint16_t> wfc;           // policy FC fragments
    DevBuf<float> bfc, wv;          // policy FC bias [1376]; value FC weights [72] + bias
    bool loaded = false;
    // activations (bf16 NHWC rows [g*24+p][C])
    DevBuf<uint16_t> x16, actX, actH, hp;
    DevBuf<float> hv, logits;
    int cap_games = 0;
    // sampled HIP-event timing of the tower conv kernel
    struct Pending { hipEvent_t a, b; double flops; int launches; int kind; int rows_seq; int boards; int dem_seq = -1; int step = -1; };   // kind 0 per-layer, 1 fused tower (> 256 boards), 3 fused tower as one <4,8,3> launch, 2 small batch (<= 256 boards: cluster tower, two-board pair tower)
                                    // tower; rows_seq >= 0: flops is per row, the row count of that (compacted) launch sits in rows_log[rows_seq]
    bool compact = true;            // search iterations above compact_above live games evaluate only the slots that need it (k_row_map)
    int compact_above = 256;        // (below, the batch is latency-bound and runs whole on the cluster tower)
    DevBuf<uint32_t> rows_log;      // [kRowsLog] rows evaluated by the compacted forward number (forward_count mod kRowsLog)
    int cur_step = -1;              // the move-step whose search is being enqueued (search_host.cpp; -1: none): a sampled launch remembers it
    DevBuf<uint32_t> dem_log;       // [kRowsLog] ... of which DEMANDED by the search (tail / free-running launches: the others are speculative)
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_events;
    int sample_every = 17;
    uint64_t forward_count = 0;
    double conv_seconds = 0, conv_flops = 0, tower_seconds = 0, tower_flops = 0, cluster_seconds = 0, cluster_flops = 0;
    uint64_t conv_launches = 0, tower_launches = 0, cluster_launches = 0;
    double full_seconds = 0, full_flops = 0;      // the subset of the fused-tower samples that were ONE k_tower16<4,8,3> launch
    uint64_t full_launches = 0;
    double band_flops_demanded[DIEE_BANDS] = {};                         // band_flops without the rows evaluated on speculation
    double band_seconds[DIEE_BANDS] = {}, band_flops[DIEE_BANDS] = {};   // every sample again, by the boards its launch(es) were dispatched for
    uint64_t band_launches[DIEE_BANDS] = {};
    hipEvent_t get_event() {
        if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
        hipEvent_t e; HIPCHK(hipEventCreate(&e)); return e;
    }
    ~NetWeights() {
        for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        for (auto e : free_events) (void)hipEventDestroy(e);
    }
};

constexpr uint32_t kRowsLog = 1u << 20;
// device-side description of a compacted batch (owned by the search): skip[slot] != 0 = no evaluation needed;
// k_row_map fills row_slot / slot_row / n_rows right before the network launches
struct NnRows { const uint8_t* skip; uint32_t* row_slot; uint32_t* slot_row; uint32_t* n_rows; };
void nn_reserve(Engine& e, int G);
// policy_dev == nullptr: stop after the policy FC and the head convs; the caller (the search) finishes the softmax and the
// value head itself from nn_heads() with the functions of nn_device.h (same bits, one launch less per evaluation)
bool nn_forward(Engine& e, const void* states_dev, int G, float* policy_dev, float* value_dev, const NnRows* rows = nullptr);
struct NetHeads { const float* logits; const float* hv; const float* wv; };
NetHeads nn_heads(Engine& e, int G);     // valid until a larger batch is reserved
// development probes (include/diee_dev.h): kernel of a (family, geometry), and the dispatch of plain evaluations by board count
const char* nn_kernel_name(int family, int geometry);
struct DispatchBand { int boards_min, boards_max, family, geometry; };
std::vector<DispatchBand> nn_dispatch_bands(Engine& e, int upto);
bool nn_tail_available(Engine& e, int G_upper, int n);
bool nn_forward_tail(Engine& e, const void* states_dev, int G_upper, const uint32_t* n_rows_dev, float* hv_out, float* logits_out, int boards_band,
                     const uint32_t* n_dem_dev = nullptr);
bool nn_free_available(Engine& e, int n);
void nn_forward_free(Engine& e, const void* arena_states, const uint32_t* rows_idx, const uint32_t* n_rows_dev, int rows_upper, float* hv_out, float* logits_out, int boards_band,
                     const uint32_t* n_dem_dev = nullptr);
bool nn_cluster_used(Engine& e);
void nn_disable_cluster(Engine& e);
void nn_reset_cluster(Engine& e);
// step_log (host copy, may be empty): per move-step { expansions, rows evaluated }: a sampled launch counts towards band_flops_demanded with the share of its
// move-step's rows that the search used
void nn_harvest(Engine& e, diee_stats* stats, const std::vector<unsigned long long>* step_log = nullptr);
void nn_reset_timing(Engine& e);
void nn_conv_bench(Engine& e, int G, int variant, int reps, float* us_mode0, float* us_mode1, float* us_forward);

}  // namespace diee
