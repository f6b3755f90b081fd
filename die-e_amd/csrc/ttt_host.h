// ttt_host.h -- tic-tac-toe behind the same C ABI (BASELINE.json configs[0]: "Tic-Tac-Toe, 1 self-play game,
// iterations=50, random-init net, CPU reference path (plumbing, no GPU)").
//
// The reference drives tic-tac-toe through the very functions it drives backgammon through (learn_parallel /
// self_play_parallel / alpha_mcts_parallel are generic over LearnableGame, src/main.rs:119-124), on the CPU: a 3x3
// board, 9 actions, a 64-filter 4-block ResNet of 2.85 M MAC per evaluation.  That is host work by definition of the
// config -- there is nothing for 256 CUs to do -- so this path is plain C++ on the host: rules (src/tictactoe/mod.rs:
// 28-100), the fp32 ResNet (src/alphazero/nnet.rs:24-34,57-133 with N_FILTERS = 64, N_RES_BLOCKS = 4, mod.rs:20-24),
// the batched search (src/mcts/alpha_mcts.rs:91-202) and the self-play driver (src/alphazero/alpha_parallel.rs:101-231).
// It is reached only through diee_create(.., DIEE_GAME_TTT, ..); the backgammon path has no host fallback.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/diee.h"

namespace diee {
namespace ttt {

constexpr int A = 9, PLANES = 27, F = 64, BLOCKS = 4, CIN = 3, HW = 9, PH = 32, VH = 3;

// LearnableGame for TicTacToe, src/tictactoe/mod.rs
int valid_moves(const diee_ttt_state& s, uint8_t out[9]);        // :36-44
void apply_move(diee_ttt_state& s, uint8_t a);                    // :46-49
void skip_turn(diee_ttt_state& s);                                // :51-53
bool check_winner(const diee_ttt_state& s, int& winner);          // :60-81 (winner 0 = draw)
void planes(const diee_ttt_state& s, float out[27]);              // :83-94

size_t weights_count();
void random_weights(uint64_t seed, float* blob);

class Engine {
public:
    void load_weights(const float* blob, size_t n);
    bool loaded() const { return !w_.empty(); }
    // ResNet::forward_t (nnet.rs:120-133), eval mode: softmax policy [n][9], tanh value [n]
    void forward(const diee_ttt_state* s, uint32_t n, float* policy, float* value) const;
    void mcts_batch(const diee_ttt_state* roots, uint32_t n, const diee_mcts_cfg& cfg, uint64_t seed, uint32_t step,
                    const uint32_t* game_ids, const uint32_t* rounds, uint32_t flags, float* visit_probs, uint32_t* n_children,
                    float* root_visits, diee_stats* stats) const;
    void self_play(uint32_t n_games, uint32_t first_game_id, const diee_mcts_cfg& cfg, float temperature, uint64_t seed,
                   uint32_t flags, uint32_t max_steps, diee_fragments* out, diee_stats* stats) const;

private:
    struct Conv { std::vector<float> w, b; int cout = 0, cin = 0; };      // BatchNorm folded (eval mode, eps 1e-5)
    Conv init_, c1_[BLOCKS], c2_[BLOCKS], pconv_, vconv_;
    std::vector<float> pfc_w_, pfc_b_, vfc_w_, vfc_b_;
    std::vector<float> w_;                                                  // the blob as loaded (non-empty = loaded)
};

}  // namespace ttt
}  // namespace diee
