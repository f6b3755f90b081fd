// A compiled host driving the engine through the C ABI (include/diee.hpp over include/diee.h), the way die-e's Rust
// host would through its extern "C" block: new net -> self_play_parallel -> the learn loop's
// self_play_iterations side by side.  Build: g++ -std=c++17 -Iinclude examples/self_play.cpp -Ldie-e_amd -ldiee -o self_play
//   ./self_play [games=8] [iterations=8] [self_play_iterations=2]
#include <cstdio>
#include <cstdlib>

#include "diee.hpp"

int main(int argc, char** argv) {
    const uint32_t games = argc > 1 ? (uint32_t)atoi(argv[1]) : 8, iters = argc > 2 ? (uint32_t)atoi(argv[2]) : 8;
    const uint32_t sp_iters = argc > 3 ? (uint32_t)atoi(argv[3]) : 2;
    try {
        diee_host::Engine eng(0);
        eng.load_weights(diee_host::Engine::random_weights(0));
        const diee_host::MctsConfig cfg{iters, 2.0f, 400, 0.3f, 0.25f};          // config-example.toml:11-15
        diee_host::Backgammon start{};                                            // Backgammon::new, backgammon_logic.rs:80-94
        const int8_t pts[24] = {2, 0, 0, 0, 0, -5, 0, -3, 0, 0, 0, 5, -5, 0, 0, 0, 3, 0, 5, 0, 0, 0, 0, -2};
        for (int i = 0; i < 24; ++i) start.pts[i] = pts[i];
        start.roll[0] = 2; start.roll[1] = 1; start.player = -1;
        const auto plays = eng.get_valid_moves({start});
        std::printf("start position, roll (2,1): %zu legal plays, first code %u\n", plays[0].size(), eng.encode(start, plays[0][0]));
        const auto search = eng.alpha_mcts_parallel({start}, cfg, 7);
        std::printf("alpha_mcts_parallel: %u root children, %llu expansions\n", search.n_children[0], (unsigned long long)search.stats.expansions);
        diee_stats st;
        const auto memory = eng.self_play_parallel(games, cfg, 1.25f, 0xD1EE0001ull, &st);
        std::printf("self_play_parallel: %llu games, %zu fragments, %llu plies, %.3f s\n", (unsigned long long)st.games, memory.size(),
                    (unsigned long long)st.plies, st.seconds);
        std::vector<diee_stats> sts;
        const auto batches = eng.self_play_iterations(sp_iters, games, cfg, 1.25f, 0xD1EE0001ull, &sts);
        size_t total = 0;
        for (const auto& b : batches) total += b.size();
        std::printf("self_play_iterations: %u batches side by side, %zu fragments; batch 0 equals the single call: %s\n", sp_iters, total,
                    (batches[0].size() == memory.size() && batches[0][0].ps == memory[0].ps && batches[0].back().outcome == memory.back().outcome) ? "yes" : "NO");
        return batches[0].size() == memory.size() ? 0 : 2;
    } catch (const diee_host::Error& e) {
        std::fprintf(stderr, "diee error %d: %s\n", (int)e.status, e.what());
        return 1;
    }
}
