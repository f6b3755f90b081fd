// diee.hpp -- C++ host-side mirror of die-e's interface for the self-play hot path, over the C ABI of diee.h.
//
// The reference's host is Rust (no toolchain in this image); this header is what its `impl` blocks for the path become
// in a compiled host language: same names, argument meaning and error behaviour (a Rust panic = a C++ exception here),
// header-only, no dependency beyond libdiee.so.  INTEGRATION.md shows the equivalent Rust `extern "C"` binding.
//
//   LearnableGame for Backgammon (src/base.rs:8-51, src/backgammon/backgammon_logic.rs)  -> Engine::get_valid_moves / encode / decode / as_tensor
//   ResNet::forward_t (src/alphazero/nnet.rs:120-133)                                     -> Engine::forward_t
//   alpha_mcts_parallel + get_prob_tensor_parallel (src/mcts/alpha_mcts.rs:91, utils.rs:42) -> Engine::alpha_mcts_parallel
//   AlphaZero::self_play_parallel (src/alphazero/alpha_parallel.rs:101-231)                -> Engine::self_play_parallel
//   the self_play_iterations loop of learn_parallel (alpha_parallel.rs:49-62)              -> Engine::self_play_iterations
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "diee.h"

namespace diee_host {

struct Error : std::runtime_error {
    diee_status status;
    Error(diee_status s, const std::string& m) : std::runtime_error(m), status(s) {}
};

using Backgammon = diee_bg_state;          // Backgammon{board, roll, player, is_second_play}
using MctsConfig = diee_mcts_cfg;          // MctsConfig, src/lib.rs:33-40
struct Play { int8_t mv[4]; };             // Actions = up to two (from, to); unused slots DIEE_NO_MOVE

struct MemoryFragment {                    // src/alphazero/alphazero.rs:68-73
    int8_t outcome;
    std::vector<float> ps;                 // [1352]
    std::vector<float> state;              // [6*4*6]
};

struct SearchResult {                      // what callers read from the NodeStore after alpha_mcts_parallel
    std::vector<float> probs;              // [n][1352] root child visits / row sum (NaN row: root without children)
    std::vector<uint32_t> n_children;      // [n]
    diee_stats stats;
};

class Engine {
public:
    explicit Engine(int device = 0) {
        const diee_status st = diee_create(device, DIEE_GAME_BACKGAMMON, &ctx_);
        if (st != DIEE_OK) throw Error(st, "diee_create failed: no HIP device (the HIP path is the only path)");
    }
    ~Engine() { if (ctx_) diee_destroy(ctx_); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    // ResNet::new with tch's default initialisers (nnet.rs:57-107), seeded
    static std::vector<float> random_weights(uint64_t seed) {
        std::vector<float> blob(diee_weights_count(DIEE_GAME_BACKGAMMON));
        const diee_status st = diee_random_weights(DIEE_GAME_BACKGAMMON, seed, blob.data(), blob.size());
        if (st != DIEE_OK) throw Error(st, "diee_random_weights");
        return blob;
    }
    void load_weights(const std::vector<float>& blob) { chk(diee_load_weights(ctx_, blob.data(), blob.size())); }

    // get_valid_moves, backgammon_logic.rs:403-414 (panics on an unrolled die there, throws here)
    std::vector<std::vector<Play>> get_valid_moves(const std::vector<Backgammon>& states, uint32_t cap = 256) {
        for (const auto& s : states)
            if (s.roll[0] == 0 && s.roll[1] == 0) throw Error(DIEE_ERR_ARG, "die has not been rolled");
        std::vector<int8_t> plays((size_t)states.size() * cap * 4);
        std::vector<uint32_t> counts(states.size());
        chk(diee_bg_legal_moves(ctx_, states.data(), (uint32_t)states.size(), plays.data(), cap, counts.data()));
        std::vector<std::vector<Play>> out(states.size());
        for (size_t i = 0; i < states.size(); ++i) {
            if (counts[i] > cap) throw Error(DIEE_ERR_CAPACITY, "more plays than `cap`");
            for (uint32_t j = 0; j < counts[i]; ++j) {
                Play p;
                for (int k = 0; k < 4; ++k) p.mv[k] = plays[((size_t)i * cap + j) * 4 + k];
                out[i].push_back(p);
            }
        }
        return out;
    }
    uint32_t encode(const Backgammon& s, const Play& p) { uint32_t c = 0; chk(diee_bg_encode(ctx_, &s, p.mv, 1, &c)); return c; }
    Play decode(const Backgammon& s, uint32_t code) { Play p; chk(diee_bg_decode(ctx_, &s, &code, 1, p.mv)); return p; }
    std::vector<float> as_tensor(const Backgammon& s) { std::vector<float> t(DIEE_BG_PLANES); chk(diee_bg_planes(ctx_, &s, 1, t.data())); return t; }

    // forward_t(.., train = false): softmax policy [n][1352], tanh value [n]
    void forward_t(const std::vector<Backgammon>& states, std::vector<float>& policy, std::vector<float>& value) {
        policy.resize(states.size() * DIEE_BG_ACTIONS); value.resize(states.size());
        chk(diee_nn_forward(ctx_, states.data(), (uint32_t)states.size(), policy.data(), value.data()));
    }

    // alpha_mcts_parallel(&mut store, &states, &net, &cfg) + get_prob_tensor_parallel(&roots, &store)
    SearchResult alpha_mcts_parallel(const std::vector<Backgammon>& states, const MctsConfig& cfg, uint64_t seed, uint32_t mcts_run = 0,
                                     bool ref_quirks = true) {
        SearchResult r;
        r.probs.resize(states.size() * DIEE_BG_ACTIONS); r.n_children.resize(states.size());
        chk(diee_mcts_batch(ctx_, states.data(), (uint32_t)states.size(), &cfg, seed, mcts_run, nullptr, nullptr,
                            ref_quirks ? DIEE_FLAG_REF_QUIRKS : 0u, r.probs.data(), r.n_children.data(), nullptr, &r.stats));
        return r;
    }

    // self_play_parallel: num_self_play_batches games to completion -> Vec<MemoryFragment>
    std::vector<MemoryFragment> self_play_parallel(uint32_t num_self_play_batches, const MctsConfig& cfg, float temperature, uint64_t seed,
                                                   diee_stats* stats = nullptr, bool ref_quirks = true) {
        diee_fragments fr;
        chk(diee_self_play(ctx_, num_self_play_batches, 0, &cfg, temperature, seed, ref_quirks ? DIEE_FLAG_REF_QUIRKS : 0u, 0, &fr, stats));
        return take(fr);
    }

    // the `for sp_i in 0..self_play_iterations` loop of learn_parallel, its calls played side by side
    std::vector<std::vector<MemoryFragment>> self_play_iterations(uint32_t self_play_iterations, uint32_t num_self_play_batches,
                                                                  const MctsConfig& cfg, float temperature, uint64_t seed,
                                                                  std::vector<diee_stats>* stats = nullptr) {
        std::vector<diee_batch> bt(self_play_iterations);
        for (uint32_t i = 0; i < self_play_iterations; ++i) bt[i] = diee_batch{num_self_play_batches, 0u, seed + i};
        std::vector<diee_fragments> fr(self_play_iterations);
        std::vector<diee_stats> st(self_play_iterations);
        chk(diee_self_play_multi(ctx_, bt.data(), self_play_iterations, &cfg, temperature, DIEE_FLAG_REF_QUIRKS, 0, fr.data(), st.data()));
        std::vector<std::vector<MemoryFragment>> out;
        for (auto& f : fr) out.push_back(take(f));
        if (stats) *stats = st;
        return out;
    }

    diee_ctx* raw() { return ctx_; }

private:
    diee_ctx* ctx_ = nullptr;
    void chk(diee_status st) { if (st != DIEE_OK) throw Error(st, diee_last_error(ctx_)); }
    static std::vector<MemoryFragment> take(diee_fragments& fr) {
        std::vector<MemoryFragment> out(fr.n);
        for (uint32_t i = 0; i < fr.n; ++i) {
            out[i].outcome = fr.outcome[i];
            out[i].ps.assign(fr.ps + (size_t)i * DIEE_BG_ACTIONS, fr.ps + (size_t)(i + 1) * DIEE_BG_ACTIONS);
            out[i].state.assign(fr.state + (size_t)i * DIEE_BG_PLANES, fr.state + (size_t)(i + 1) * DIEE_BG_PLANES);
        }
        diee_free_fragments(&fr);
        return out;
    }
};

}  // namespace diee_host
