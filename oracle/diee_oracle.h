/*
 * diee_oracle.h -- CPU ORACLE for the die-e batched self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (alibasaran/die-e, Rust) used as the parity checker for the HIP
 * path in die-e_amd/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path never does.
 *
 * Parity pin: the Rust reference cannot be built in this image (no cargo /
 * rustc / libtorch, no network), so the restatement is pinned by the
 * reference's own test vectors transcribed as data in tests/golden/
 * (tests/backgammon_test.rs, tests/encoding_test.rs, tests/tictactoe_test.rs,
 * tests/mcts_test.rs).  Network outputs and everything downstream of a random
 * draw are "parity unpinned" in the reference itself (unseeded thread_rng, no
 * test constructs a ResNet) -- see DESIGN.md.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef DIEE_ORACLE_H
#define DIEE_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- backgammon state (reference: src/backgammon/backgammon_logic.rs:10,53-60) ---- */
typedef struct {
    int8_t  pts[24];   /* Board.0 : <0 player -1, >0 player +1                */
    uint8_t bar[2];    /* Board.1 : checkers on the bar (p-1, p+1)           */
    uint8_t off[2];    /* Board.2 : checkers collected   (p-1, p+1)          */
    uint8_t roll[2];   /* roll as rolled (unsorted)                          */
    int8_t  player;    /* -1 moves first                                     */
    uint8_t second;    /* is_second_play                                     */
} or_bg_state;         /* 32 bytes */

#define OR_BG_ACTIONS 1352
#define OR_BG_PLANES  144          /* 6 x 4 x 6 */
#define OR_NO_MOVE   (-2)          /* sentinel in a play's unused (from,to) slots */
#define OR_MAX_PLAYS  2048         /* upper bound on DFS sequences we store   */

/* a play = up to 2 checker moves: mv = {f1,t1,f2,t2}; unused slots = OR_NO_MOVE */
typedef struct { int8_t mv[4]; } or_play;

static inline int or_play_len(const or_play* p) {
    return p->mv[0] == OR_NO_MOVE ? 0 : (p->mv[2] == OR_NO_MOVE ? 1 : 2);
}

/* flattened action tree (pre-order) for the get_normal_moves / get_entry_moves tests */
typedef struct { int8_t depth, from, to; } or_tree_node;

/* ---- rules -------------------------------------------------------------- */
void or_bg_new(or_bg_state* s);                                         /* :80-94  */
void or_bg_next_state(or_bg_state* board, const int8_t (*moves)[2], int n, int player); /* :467-517 */
int  or_bg_is_collectible(const or_bg_state* board, int player);        /* :638-659 */
int  or_bg_check_winner(const or_bg_state* board, int* winner);         /* :106-108,527-534 */
int  or_bg_normal_moves(const uint8_t* dice, int nd, const or_bg_state* b, int player,
                        or_tree_node* out, int cap);                    /* :555-636 */
int  or_bg_entry_moves(const uint8_t* dice, int nd, const or_bg_state* b, int player,
                       or_tree_node* out, int cap);                     /* :662-703 */
int  or_bg_action_trees(const uint8_t* dice, int nd, const or_bg_state* b, int player,
                        or_tree_node* out, int cap);                    /* :544-552 */
/* DFS sequences of a flattened tree list (sequence = up to 4 moves, as the generic reference code allows) */
typedef struct { int8_t n; int8_t mv[4][2]; } or_seq;
int  or_bg_extract_sequences(const or_tree_node* tree, int n, or_seq* out, int cap); /* :722-750 */
int  or_bg_remove_duplicate_states(const or_bg_state* initial, const or_seq* seqs, int n,
                                   int player, or_seq* out);            /* :753-774 */
int  or_bg_valid_moves(const or_bg_state* s, or_play* out, int cap);    /* :403-414 */
int  or_bg_valid_moves_seq(const or_bg_state* s, or_seq* out, int cap, int* n_before_dedup);
void or_bg_apply_move(or_bg_state* s, const or_play* p, uint8_t d0, uint8_t d1); /* :176-186 */
void or_bg_skip_turn(or_bg_state* s, uint8_t d0, uint8_t d1);           /* :192-196 */
uint32_t or_bg_encode(const or_bg_state* s, const or_play* p);          /* :262-359 */
void or_bg_decode(const or_bg_state* s, uint32_t code, or_play* out);   /* :361-401 */
void or_bg_planes(const or_bg_state* s, float* out144);                 /* :198-252 */

/* ---- tic-tac-toe (reference: src/tictactoe/mod.rs) ----------------------- */
typedef struct { int8_t board[9]; int8_t player; uint8_t pad[22]; } or_ttt_state; /* 32 bytes */
void or_ttt_new(or_ttt_state* s);
int  or_ttt_valid_moves(const or_ttt_state* s, uint8_t* out);
void or_ttt_apply_move(or_ttt_state* s, uint8_t a);
int  or_ttt_check_winner(const or_ttt_state* s, int* winner);
void or_ttt_planes(const or_ttt_state* s, float* out27);

/* ---- deterministic helpers shared (by algorithm, not by code) with the HIP path */
void     or_philox4x32(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]);
void     or_dice(uint64_t seed, uint32_t game, uint32_t round, uint32_t tag, uint32_t ord,
                 uint8_t* d0, uint8_t* d1);
double   or_uniform01(uint64_t seed, uint32_t game, uint32_t round, uint32_t tag, uint32_t ord);
float    or_det_powf(float x, float y);
void     or_dirichlet(uint64_t seed, uint32_t step, float alpha, int n, float* out);

/* RNG tags (purpose field of the Philox counter) */
#define OR_TAG_INIT_ROLL 0xFFFFFFFFu
#define OR_TAG_MOVE_ROLL 0xFFFFFFFEu
#define OR_TAG_SAMPLE    0xFFFFFFFDu
#define OR_TAG_DIRICHLET 0xFFFFFFFCu

/* ---- generic game vtable (reference: src/base.rs:8-51 trait LearnableGame) */
typedef struct { uint8_t b[32]; } or_state;     /* opaque 32-byte game state    */
typedef struct or_game {
    int  id;                 /* 0 ttt, 1 backgammon */
    int  n_actions;          /* ACTION_SPACE_SIZE   */
    int  n_planes;           /* C*H*W floats        */
    int  deterministic;      /* IS_DETERMINISTIC    */
    void (*new_state)(or_state*);
    int  (*valid_moves)(const or_state*, or_play* out, int cap);
    void (*apply_move)(or_state*, const or_play*, uint8_t d0, uint8_t d1);
    void (*skip_turn)(or_state*, uint8_t d0, uint8_t d1);
    int  (*get_player)(const or_state*);
    int  (*check_winner)(const or_state*, int* winner);
    uint32_t (*encode)(const or_state*, const or_play*);
    void (*decode)(const or_state*, uint32_t, or_play*);
    void (*planes)(const or_state*, float*);
    void (*set_roll)(or_state*, uint8_t d0, uint8_t d1);
} or_game;
const or_game* or_game_by_id(int id);

/* ---- MCTS (reference: src/mcts/ alpha_mcts.rs, node.rs, utils.rs, noise.rs, simple_mcts.rs) ---- */
typedef struct {
    uint32_t iterations;      /* lib.rs:34 */
    float    c;               /* exploration_const */
    uint32_t round_limit;     /* simulate_round_limit */
    float    dir_alpha, dir_eps;
} or_mcts_cfg;

typedef struct {
    or_state state;
    int32_t  parent;          /* -1 = None */
    int32_t  first_child;     /* children are contiguous (node.rs:157-174 pushes them in order) */
    int32_t  n_children;
    float    visits, value, policy;
    int32_t  action;          /* encode(action_taken) w.r.t. the parent state, -1 for roots */
    uint8_t  drained;         /* expandable_moves has been drained by alpha_expand_tensor    */
} or_node;

typedef struct { or_node* nodes; int n, cap; } or_store;

/* evaluator: forward_t on n states -> softmax policy [n][A], tanh value [n]  (nnet.rs:120-133) */
typedef void (*or_eval_fn)(void* ctx, const or_state* states, int n, float* policy, float* value);

typedef struct {
    uint64_t nn_evals;        /* states pushed through the evaluator (incl. stale slots)       */
    uint64_t expansions;      /* non-terminal leaves expanded + root expansions                */
    uint64_t children;        /* nodes created below roots                                     */
    uint64_t terminal_hits;   /* selections that ended on a terminal leaf                      */
    uint64_t depth_sum;       /* sum of leaf depths over selections                            */
    uint64_t selections;
    uint64_t code_collisions; /* two plays of one node with the same action code               */
    uint64_t illegal_decodes; /* alpha_parallel.rs:204 assertion failures (counted, not fatal) */
    uint64_t max_children;
} or_stats;

void or_store_init(or_store* st);
void or_store_free(or_store* st);
float or_alpha_ucb(const or_store* st, int idx, float c);                 /* node.rs:98-112 */
int  or_select_leaf(const or_store* st, int root, float c, int* depth);   /* alpha_mcts.rs:14-33 */
void or_backpropagate(or_store* st, int idx, float v);                    /* simple_mcts.rs:96-103 */
/* alpha_mcts_parallel, alpha_mcts.rs:91-202.  game_ids/rounds key the child dice. */
void or_alpha_mcts_parallel(const or_game* g, or_store* st, const or_state* states, int n,
                            const or_mcts_cfg* cfg, or_eval_fn eval, void* ectx,
                            uint64_t seed, uint32_t step, const uint32_t* game_ids,
                            const uint32_t* rounds, int ref_quirks, or_stats* stats);
/* the same search as a resumable machine (one network evaluation per step), so that several batches can share
 * one evaluator call: begin -> { next: EVAL -> evaluate or_mcts_batch() -> feed | IDLE | DONE } -> end */
typedef struct or_mcts_run or_mcts_run;
enum { OR_MCTS_DONE = 0, OR_MCTS_EVAL = 1, OR_MCTS_IDLE = 2 };
or_mcts_run* or_mcts_begin(const or_game* g, or_store* st, const or_state* states, int n, const or_mcts_cfg* cfg,
                           uint64_t seed, uint32_t step, const uint32_t* game_ids, const uint32_t* rounds,
                           int ref_quirks, or_stats* stats);
int  or_mcts_next(or_mcts_run* r);
const or_state* or_mcts_batch(const or_mcts_run* r);
int  or_mcts_rows(const or_mcts_run* r);
void or_mcts_feed(or_mcts_run* r, float* policy, const float* value);
/* read-only views for trace tools, valid inside the evaluator callback: the run begun last (single-threaded use), its
 * selected_nodes_idxs, which of them were selected in this iteration, 0 = roots pending / 1 + it, and its node store */
or_mcts_run* or_mcts_last(void);
const int* or_mcts_sel(const or_mcts_run* r);
const uint8_t* or_mcts_fresh(const or_mcts_run* r);
int  or_mcts_phase(const or_mcts_run* r);
const or_store* or_mcts_store(const or_mcts_run* r);
void or_mcts_end(or_mcts_run* r);
/* get_prob_tensor_parallel, utils.rs:42-58: probs[n][A] */
void or_get_prob_tensor_parallel(const or_game* g, const or_store* st, int n, float* probs);

/* ---- self-play driver (reference: src/alphazero/alpha_parallel.rs:101-231) */
typedef struct {
    int32_t  n;               /* fragments */
    int8_t*  outcome;         /* [n]       */
    float*   ps;              /* [n][A]    */
    float*   state;           /* [n][planes] */
    uint32_t* game;           /* [n] originating game (extra, for tests) */
} or_fragments;

int  or_self_play_parallel(const or_game* g, uint32_t n_games, uint32_t first_game_id,
                           const or_mcts_cfg* cfg, float temperature, uint64_t seed,
                           or_eval_fn eval, void* ectx, int ref_quirks, uint32_t max_steps,
                           or_fragments* out, or_stats* stats,
                           uint32_t* plies /* [n_games] or NULL */, int8_t* winners /* or NULL */);
/* K self_play_parallel calls in lockstep, one merged evaluator call per search phase (see diee_oracle.c) */
int  or_self_play_multi(const or_game* g, uint32_t n_batches, const uint32_t* n_games, const uint32_t* first_game_ids,
                        const uint64_t* seeds, const or_mcts_cfg* cfg, float temperature, or_eval_fn eval, void* ectx,
                        int ref_quirks, uint32_t max_steps, or_fragments* outs, or_stats* stats, uint32_t* steps);
void or_free_fragments(or_fragments* f);

/* a cheap deterministic evaluator for CPU-only tests (NOT a network) */
void or_hash_eval(void* ctx /* const or_game* */, const or_state* states, int n, float* policy, float* value);

/* batch helpers for the parity tests */
int  or_random_walk_states(uint64_t seed, uint32_t n_games, uint32_t max_plies, or_bg_state* out, int cap);
void or_bg_valid_moves_batch(const or_bg_state* s, int n, int8_t* plays, int cap, uint32_t* counts);
void or_bg_encode_batch(const or_bg_state* s, const int8_t* plays, int n, uint32_t* codes);
void or_bg_decode_batch(const or_bg_state* s, const uint32_t* codes, int n, int8_t* plays);
void or_bg_apply_batch(or_bg_state* s, const int8_t* plays, const uint8_t* dice, int n);
void or_bg_planes_batch(const or_bg_state* s, int n, float* out);

#ifdef __cplusplus
}
#endif
#endif
