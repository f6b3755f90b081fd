// engine.h -- host-side engine behind the C ABI: owns the HIP stream, HBM buffers and drives the
// kernels.  No CPU compute path exists: every method launches HIP kernels on `device`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/diee.h"

namespace diee {

struct EngineError : public std::runtime_error {
    int code;
    EngineError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            throw ::diee::EngineError(DIEE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void ensure(size_t n) {
        if (n <= cap) return;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        HIPCHK(hipMalloc((void**)&p, n * sizeof(T)));
        cap = n;
    }
    size_t bytes() const { return cap * sizeof(T); }
};

struct NetWeights;   // nn_host.cpp
struct SearchBufs;   // search_host.cpp

// Per-ctx switches (diee_set_option, include/diee.h).  Defaults here; the environment (DIEE_<KEY IN CAPITALS>, e.g. DIEE_TOWER_CL for
// "tower_cl") is read ONCE, when the ctx is created, as a development override of these defaults; nothing reads it afterwards.
struct Options {
    // network dispatch (applied to NetWeights by Engine::apply_options)
    std::string tower_table = "default";    // "min:geometry,..." | "none" | "default": fused / pair tower by live games (nn_host.h)
    std::string tower_cl = "default";       // "max:boards,..."   | "none" | "default": cluster tower (<= 256 boards)
    int tower_pair = 1;                     // pair tower (257 ... 512 and 129 ... 256 boards); 0: single-workgroup geometries
    int fused_heads = 1, cluster_heads = 1, cluster_init = 1;   // head convs / init block inside the tower launches
    int compact = 1;                        // above 256 live games evaluate only the slots that need it
    int cl_pack = 1;                        // few clusters share few XCDs
    int shared_gpu = 0;                     // another PROCESS uses this GPU: no kernel of this ctx waits for a co-resident workgroup
                                            // (cluster and pair tower off; the caller's training step: diee_train_set_bn_coop(0))
    // search
    int cl_grow = 1, expand2 = 1, expand2c = 1;
    int spec_eval = 1;                      // the tail of a batch (<= spec_max_games live games): iterations in one launch while the leaves' evaluations are at
                                            // hand, speculative rows in the launches in between (search_types.h, Tail); 0: one launch per iteration
    uint32_t spec_rollout_steps = 24;       // virtual descents per game and launch that look for those rows (0: demanded rows only)
    uint32_t spec_fused_from = 129;         // live games from which a tail launch is one of the fused family (where tower_table sends the plain evaluations of so many boards)
    uint32_t spec_fused_games = 256;        // 129 ... this many live games search in tail mode too, on 512-row launches of the fused family (<= 256; 0: off) -- where the
                                            // free-running search does not take them (free_eval = 0, or free_min_games above): round 6 measured it 8 ... 15 % faster there
    uint32_t spec_child_rows = 16;          // children of a demanded leaf that are evaluated in the same tail launch as the leaf at most (created ahead; 0: off)
    uint32_t spec_extra_rows = 2;           // candidates a game may find beyond its share of a full tail launch: they take the rows other games left free
    uint32_t spec_max_games = 96;           // live games up to which a move-step's search runs in tail mode (<= 128 = kTailMaxSlots; beyond 64 a launch has fewer spare rows than games)
    uint32_t spec_rows64_from = 5, spec_rows128_from = 10;      // live games from which a tail launch carries 64 / 128 rows instead of 32
    int free_eval = 1;                      // the free-running search (search_types.h, Free) at free_min_games ... free_max_games live games; 0: one launch per iteration there
    uint32_t free_min_games = 17, free_max_games = 800;      // (17 ... 40 live games on 64 / 128-row launches of the cluster family, from 41 on the fused family's
                                                             // 512 / 1024 rows; k_tail keeps <= 16; above ~820 the launch per iteration wins (profiles/r06r_*); <= 1024: k_free_pack runs one thread per game)
    uint32_t free_rows1024_from = 200;      // live games from which a launch of it is one pass of the chip (1024 rows) instead of the pair tower's 512 (profiles/r06e_*)
    uint32_t free_rollout_steps = 24, free_cand_max = 12;    // virtual descents / candidates per game and round at most (the candidates follow the spare rows: free_view)
    uint32_t free_cand_x4 = 8;              // candidates per game and round = 1 + this / 4 x the spare rows per game (8: twice the rows a game can hope for)
    uint32_t free_lag_boost = 4, free_lag_step = 4;          // the packer serves a game free_lag_step iterations behind the leader one rank earlier, up to free_lag_boost ranks (<= 16; 0: off)
    uint32_t free_iter_cap = 4;             // iterations a game runs in one of its launches at most
    uint32_t free_ring = 128;               // launches whose rows stay in its ring (a row that aged out is evaluated again: same bits)
    uint32_t free_lds_nodes = 3072;         // cap of the tree nodes k_free stages in LDS per game (tests lower it to reach the in-place path)
    uint32_t spec_ring_mb = 8192;           // HBM the tail's ring of evaluated rows may take (MiB): (iterations + 1) launches x rows x 5.7 KB; a search whose ring
                                            // would be larger runs one launch per iteration instead (iterations = 1600 x 512 rows: 4.7 GB)
    uint32_t path_cap = 64, nodes_per_expansion = 128;
    // output delivery
    uint32_t deliver_stage_rows = 16384, deliver_rows_per_game = 128;
    // development traces on stderr
    int trace_steps = 0, trace_dispatch = 0;
    int test_tail_skip = 0;                 // tests: 1 + the meeting word of the next tail search at which one game's workgroup never arrives (k_tail's device-side time-out)
    int test_starve_at = 0;                 // tests: the k-th hand-over check of this ctx reports a starved hand-over (0: never)
};

class Engine {
public:
    char err[512];
    explicit Engine(int device);
    ~Engine();

    // pure game functions
    void legal_moves(const diee_bg_state* s, uint32_t n, int8_t* plays, uint32_t cap, uint32_t* counts);
    void rules_bench(const diee_bg_state* s, uint32_t n, int reps, float* us_legal_moves, float* mean_plays);
    void encode(const diee_bg_state* s, const int8_t* plays, uint32_t n, uint32_t* codes);
    void decode(const diee_bg_state* s, const uint32_t* codes, uint32_t n, int8_t* plays);
    void apply(diee_bg_state* s, const int8_t* plays, const uint8_t* dice, uint32_t n);
    void planes(const diee_bg_state* s, uint32_t n, float* out);
    void probe_f32(const float* a, const float* b, uint32_t n, float* sq, float* dv, float* pw);
    void probe_dice(uint64_t seed, const uint32_t* ctr, uint32_t n, uint8_t* dice, double* uni);

    // network
    void load_weights(const float* blob, size_t n);
    void nn_forward_host(const diee_bg_state* states, uint32_t n, float* policy, float* value);

    // search / self-play
    void mcts_batch(const diee_bg_state* roots, uint32_t n, const diee_mcts_cfg* cfg, uint64_t seed, uint32_t step,
                    const uint32_t* game_ids, const uint32_t* rounds, uint32_t flags, float* visit_probs,
                    uint32_t* n_children, float* root_visits, diee_stats* stats);
    void self_play(uint32_t n_games, uint32_t first_game_id, const diee_mcts_cfg* cfg, float temperature,
                   uint64_t seed, uint32_t flags, uint32_t max_steps, diee_fragments* out, diee_stats* stats);
    void self_play_multi(const diee_batch* batches, uint32_t n_batches, const diee_mcts_cfg* cfg, float temperature,
                         uint32_t flags, uint32_t max_steps, diee_fragments* outs, diee_stats* stats);

    int device;
    hipStream_t stream = nullptr;

    template <class T>
    void h2d(T* dst, const T* src, size_t n) {
        if (n) HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, stream));
    }
    template <class T>
    void d2h(T* dst, const T* src, size_t n) {
        if (n) HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, stream));
    }
    void sync() { HIPCHK(hipStreamSynchronize(stream)); }
    void check_overflow();

    Options opt;
    int starve_checks = 0;
    void set_option(const std::string& key, const std::string& value);    // throws DIEE_ERR_ARG on an unknown key / malformed value
    std::string get_option(const std::string& key) const;
    void apply_options();                                                  // opt -> NetWeights (after load_weights, after set_option)

    DevBuf<uint8_t> tmp_a, tmp_b, tmp_c, tmp_d, tmp_e;
    DevBuf<uint32_t> flags_dev;      // [0] capacity-overflow flag
    NetWeights* net = nullptr;
    SearchBufs* search = nullptr;
};

}  // namespace diee
