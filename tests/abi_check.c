/* Compiled as C11 by tests/test_abi_cpu.py (gcc -std=c11 -Iinclude): include/diee.h must be a valid C header, and the
 * struct layouts a Rust #[repr(C)] binding relies on are asserted at compile time and printed for the
 * comparison with the ctypes mirror in die-e_amd/__init__.py. */
#include <stddef.h>
#include <stdio.h>
#include "diee.h"
#include "diee_dev.h"

_Static_assert(sizeof(diee_bg_state) == 32, "diee_bg_state is 32 bytes");
_Static_assert(offsetof(diee_bg_state, bar) == 24 && offsetof(diee_bg_state, off) == 26 && offsetof(diee_bg_state, roll) == 28 &&
               offsetof(diee_bg_state, player) == 30 && offsetof(diee_bg_state, second) == 31, "diee_bg_state fields");
_Static_assert(sizeof(diee_mcts_cfg) == 20, "diee_mcts_cfg is 5 x 4 bytes");
_Static_assert(offsetof(diee_mcts_cfg, c) == 4 && offsetof(diee_mcts_cfg, round_limit) == 8 &&
               offsetof(diee_mcts_cfg, dir_alpha) == 12 && offsetof(diee_mcts_cfg, dir_eps) == 16, "diee_mcts_cfg fields");
_Static_assert(sizeof(diee_batch) == 16 && offsetof(diee_batch, first_game_id) == 4 && offsetof(diee_batch, seed) == 8, "diee_batch");
_Static_assert(sizeof(diee_stats) == (29 + 27 + 3 + 9) * 8, "diee_stats is 68 x 8 bytes");
_Static_assert(sizeof(diee_fragments) == 8 + 4 * sizeof(void*), "diee_fragments");

#define F(T, f) printf("  \"%s.%s\": %zu,\n", #T, #f, offsetof(T, f))
int main(void) {
    printf("{\n");
    F(diee_stats, games); F(diee_stats, plies); F(diee_stats, move_steps); F(diee_stats, nn_evals); F(diee_stats, expansions);
    F(diee_stats, children); F(diee_stats, terminal_hits); F(diee_stats, depth_sum); F(diee_stats, selections);
    F(diee_stats, illegal_decodes); F(diee_stats, max_children); F(diee_stats, fragments); F(diee_stats, seconds);
    F(diee_stats, nn_seconds); F(diee_stats, conv_seconds); F(diee_stats, conv_launches); F(diee_stats, conv_flops);
    F(diee_stats, tower_seconds); F(diee_stats, tower_launches); F(diee_stats, tower_flops); F(diee_stats, cluster_seconds);
    F(diee_stats, cluster_launches); F(diee_stats, cluster_flops); F(diee_stats, nn_rows);
    F(diee_stats, full_seconds); F(diee_stats, full_launches); F(diee_stats, full_flops);
    F(diee_stats, deliver_seconds); F(diee_stats, deliver_bytes);
    F(diee_stats, band_seconds); F(diee_stats, band_launches); F(diee_stats, band_flops);
    F(diee_stats, tail_iterations); F(diee_stats, tail_launches); F(diee_stats, tail_spec_rows); F(diee_stats, band_flops_demanded);
    F(diee_fragments, n); F(diee_fragments, outcome); F(diee_fragments, ps); F(diee_fragments, state); F(diee_fragments, game);
    F(diee_batch, n_games); F(diee_batch, first_game_id); F(diee_batch, seed);
    printf("  \"sizeof.diee_stats\": %zu, \"sizeof.diee_fragments\": %zu, \"sizeof.diee_bg_state\": %zu,\n", sizeof(diee_stats),
           sizeof(diee_fragments), sizeof(diee_bg_state));
    printf("  \"sizeof.diee_mcts_cfg\": %zu, \"sizeof.diee_batch\": %zu\n}\n", sizeof(diee_mcts_cfg), sizeof(diee_batch));
    return 0;
}
