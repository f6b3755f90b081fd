/*
 * diee.h -- C ABI of the MI355X-native die-e self-play engine (libdiee.so, gfx950 / CDNA4).
 *
 * Drop-in boundary for ONE hot path of alibasaran/die-e: batched AlphaZero-style self-play for
 * backgammon (step dynamics + MCTS + policy/value ResNet).  The reference has no FFI of its own;
 * the seam is the Rust generic function pair
 *
 *     alpha_mcts_parallel(&mut NodeStore<T>, &[T], &ResNet, &MctsConfig, Option<ProgressBar>)
 *                                                             src/mcts/alpha_mcts.rs:91
 *     get_prob_tensor_parallel(&[&Node<T>], &NodeStore<T>) -> Tensor[N,1352]
 *                                                             src/mcts/utils.rs:42
 *
 * called from AlphaZero::self_play_parallel (src/alphazero/alpha_parallel.rs:101-231, at :146-166)
 * and get_actions_for_player (src/versus.rs:279-283).  Each entry point below names the reference
 * interface it replaces.  INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add.
 *
 * Conventions: plain pointers and sizes; all pointers are HOST pointers unless a name ends in
 * `_dev`; the caller owns inputs and pre-sized outputs; the engine owns `diee_fragments` until
 * diee_free_fragments.  Errors are status codes (the reference panics); diee_last_error() gives
 * text.  One diee_ctx per GPU; calls on a ctx must be serialised by the caller; contexts are
 * independent (that is the multi-GPU story: one process per GPU, games block-partitioned).
 * Every compute entry point runs hand-written HIP kernels on the ctx's device; there is no CPU
 * fallback -- without a GPU diee_create fails with DIEE_ERR_HIP.
 */
#ifndef DIEE_H
#define DIEE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DIEE_BG_ACTIONS 1352u   /* Backgammon::ACTION_SPACE_SIZE, backgammon_logic.rs:74 */
#define DIEE_BG_PLANES  144u    /* 6 x 4 x 6,  backgammon_logic.rs:75-76,198-252         */
#define DIEE_NO_MOVE    (-2)    /* filler for the unused (from,to) slots of a play         */
#define DIEE_GAME_TTT        0  /* tic-tac-toe: BASELINE configs[0], host path (see below)  */
#define DIEE_GAME_BACKGAMMON 1
#define DIEE_TTT_ACTIONS 9u     /* TicTacToe::ACTION_SPACE_SIZE, src/tictactoe/mod.rs:20     */
#define DIEE_TTT_PLANES  27u    /* 3 x 3 x 3, mod.rs:83-94                                   */
#define DIEE_BANDS 9u           /* live-game bands of diee_stats.band_*: <= 16, 32, 64, 128, 256, 512, 928, 1024, above */

typedef enum {
    DIEE_OK = 0,
    DIEE_ERR_ARG = 1,          /* bad argument                                            */
    DIEE_ERR_HIP = 2,          /* HIP runtime error / no device / a starved in-launch hand-over (two PROCESSES
                                  sharing one GPU: diee_set_option(ctx, "shared_gpu", "1"))          */
    DIEE_ERR_NO_WEIGHTS = 3,   /* diee_load_weights has not been called                   */
    DIEE_ERR_CAPACITY = 4,     /* tree arena / sequence buffer overflow (never silent)    */
    DIEE_ERR_UNSUPPORTED = 5
} diee_status;

/* Backgammon{board,roll,player,is_second_play}, backgammon_logic.rs:10,53-60 (id dropped: it is
 * only a HashMap key in the arena, versus.rs:178,220).  32 bytes. */
typedef struct {
    int8_t  pts[24];           /* Board.0: <0 player -1 ("Player 1"), >0 player +1        */
    uint8_t bar[2];            /* Board.1: (p-1, p+1) checkers on the bar                 */
    uint8_t off[2];            /* Board.2: (p-1, p+1) checkers collected                  */
    uint8_t roll[2];           /* as rolled                                               */
    int8_t  player;
    uint8_t second;            /* is_second_play                                          */
} diee_bg_state;

/* TicTacToe{player, board}, src/tictactoe/mod.rs:5-12 (id dropped), padded to the 32-byte state record the
 * batched entry points take.  BASELINE.json configs[0] names this game as "CPU reference path (plumbing, no GPU)":
 * a ctx created with DIEE_GAME_TTT runs rules, the 64-filter 4-block ResNet (mod.rs:20-24), the search and the
 * self-play driver in C++ on the HOST (die-e_amd/csrc/ttt_host.cpp) and needs no GPU; its states are diee_ttt_state
 * records (cast the pointer where a prototype says diee_bg_state), policies / ps rows have 9 entries, planes 27.
 * Served for such a ctx: diee_load_weights, diee_nn_forward, diee_mcts_batch, diee_self_play (+ lifetime / error calls);
 * everything else answers DIEE_ERR_UNSUPPORTED.  The backgammon ctx has no host path. */
typedef struct {
    int8_t  board[9];          /* 0 | 1 | 2 / 3 | 4 | 5 / 6 | 7 | 8; -1 / 0 / +1            */
    int8_t  player;            /* -1 moves first, mod.rs:28-30                              */
    uint8_t pad[22];
} diee_ttt_state;

/* MctsConfig, src/lib.rs:33-40 (keys: iterations, exploration_const, simulate_round_limit,
 * dirichlet_alpha, dirichlet_epsilon; config-example.toml:11-15) */
typedef struct {
    uint32_t iterations;
    float    c;
    uint32_t round_limit;
    float    dir_alpha, dir_eps;
} diee_mcts_cfg;

/* flags */
#define DIEE_FLAG_REF_QUIRKS 1u /* reproduce SURVEY Appendix A Q14 (stale selected_nodes_idxs
                                   slots, alpha_mcts.rs:142,157-166,175-200) and Q18 (double
                                   flush, alpha_parallel.rs:172-180,215-223)                */
#define DIEE_FLAG_INVARIANT_NN 2u /* evaluate every batch size with the fused 16x16x32 tower (one arithmetic
                                   for all sizes): the network output of a state is then a pure function of
                                   the state, whatever shares its launch.  Slower below 257 boards (the
                                   default picks a latency-optimised split-K kernel there).        */

typedef struct {
    uint64_t games;            /* games retired (winner or round limit)                    */
    uint64_t plies;            /* rounds played, skipped turns included                    */
    uint64_t move_steps;       /* passes of the while loop, alpha_parallel.rs:129          */
    uint64_t nn_evals;         /* batch rows pushed through the ResNet (stale rows too)    */
    uint64_t expansions;       /* first-time node expansions, roots included               */
    uint64_t children;         /* nodes created below roots                                */
    uint64_t terminal_hits;    /* selections ending on a terminal leaf                     */
    uint64_t depth_sum;        /* sum of leaf depths over selections                       */
    uint64_t selections;
    uint64_t illegal_decodes;  /* alpha_parallel.rs:204 self-check failures (counted)      */
    uint64_t max_children;
    uint64_t fragments;
    double   seconds;          /* wall clock of the call, inputs resident, outputs delivered */
    double   nn_seconds;       /* HIP-event time of the ResNet kernels                     */
    double   conv_seconds;     /* HIP-event time of sampled per-layer tower conv launches   */
    uint64_t conv_launches;    /* sampled launches of k_conv3x3 / k_conv3x3_sk (tower)      */
    double   conv_flops;       /* algorithmic FLOPs those launches performed               */
    double   tower_seconds;    /* HIP-event time of sampled fused-tower launches (k_tower)  */
    uint64_t tower_launches;
    double   tower_flops;
    double   cluster_seconds;  /* HIP-event time of sampled small-batch launches (<= 256 boards: k_tower_cl, k_tower16p<2>) */
    uint64_t cluster_launches;
    double   cluster_flops;
    uint64_t nn_rows;          /* rows the ResNet really evaluated: above 256 live games the rows of slots whose selected leaf
                                  was terminal (stale in the reference, never read) are skipped; nn_evals keeps the reference's count */
    double   full_seconds;     /* the subset of the tower_* samples that were exactly ONE k_tower16<4,8,3> launch (929 ... 1024   */
    uint64_t full_launches;    /* boards, or whole multiples of 1024): comparable one to one with that kernel's row in a          */
    double   full_flops;       /* rocprofv3 --kernel-trace --stats summary                                                        */
    double   deliver_seconds;  /* host wall clock the call spent on output delivery: enqueueing each move-step's gather + copies    */
                               /* on the copy stream, plus the wait for the last of them after the last move-step (batch 0 only)    */
    uint64_t deliver_bytes;    /* bytes copied into the diee_fragments arrays, all batches of the call (batch 0 only)               */
    /* the same sampled network evaluations (every kernel family) binned by the number of live games the evaluation was dispatched
     * for -- upper bounds DIEE_BAND_BOUNDS, the last band is everything above one pass of the chip: where a batch's time goes as
     * it shrinks.  seconds = HIP-event time of the tower part of the evaluation, flops = its 38 layers' algorithmic FLOPs */
    double   band_seconds[9];
    uint64_t band_launches[9]; /* sampled evaluations                                                                              */
    double   band_flops[9];
    /* the tail of a batch (<= spec_max_games = 96 and 129 ... spec_fused_games = 256 live games; option spec_eval): search iterations run by the looping tree kernel, the network launches
     * they needed (one per iteration without it), and the rows those launches evaluated on speculation (all batches of the call,
     * reported with batch 0) */
    uint64_t tail_iterations;  /* (round 6: the iterations, launches and speculative rows of the free-running search at 17 ... 800 live games count here too) */
    uint64_t tail_launches;
    uint64_t tail_spec_rows;
    /* band_flops without the rows evaluated in vain: every row of a plain or compacted evaluation; of a tail / free-running launch -- whose rows
     * are evaluated ahead of the search -- the share its move-step's search went on to use (expansions / rows evaluated of that move-step) */
    double   band_flops_demanded[9];
} diee_stats;

/* Vec<MemoryFragment>, src/alphazero/alphazero.rs:68-73: host arrays owned by the engine (page-locked: every move-step's
 * records are copied out of HBM while the following move-steps search; `seconds` includes the last copy).  Rows are in the
 * reference's order: by the move-step that removed the game, then by game, a round-limit flush before a win flush. */
typedef struct {
    uint32_t  n;
    int8_t*   outcome;         /* [n]        +1 / -1 / 0                                   */
    float*    ps;              /* [n][1352]  (visits/sum)^(1/T), not renormalised (Q17)    */
    float*    state;           /* [n][144]   as_tensor planes, c*24 + point                */
    uint32_t* game;            /* [n]        originating game id (extra)                   */
} diee_fragments;

typedef struct diee_ctx diee_ctx;

/* ---- lifetime ---------------------------------------------------------------------------- */
diee_status diee_create(int device, int game_id, diee_ctx** out);
void        diee_destroy(diee_ctx*);
const char* diee_last_error(const diee_ctx*);      /* valid until the next call on the ctx */
const char* diee_version(void);

/* the PCI address ("0000:c1:00.0") of the GPU this library's HIP runtime calls `device` -- a host whose own framework carries a second
 * HIP runtime (PyTorch does) compares it with what that runtime reports for the ordinal it is about to use, once, before diee_create:
 * both runtimes enumerate the same visible devices, and a mismatch would put the engine and the host's tensors on different GPUs.
 * No ctx needed; initialises this library's runtime.  DIEE_ERR_HIP: no such device. */
diee_status diee_device_pci_bus_id(int device, char* out /*[cap], >= 16*/, size_t cap);

/* ---- options of a ctx ---------------------------------------------------------------------
 * What a host may legitimately have to tell the engine, per ctx, as key / value strings (values: non-negative decimal integers, or
 * for the two tables "a:b,c:d" | "none" | "default").  The environment is NOT consulted by any call below: it is read once, inside
 * diee_create, as a development override of the defaults (DIEE_<KEY IN CAPITALS>); a host sets what it needs through this call.
 *   shared_gpu            0 | 1   another PROCESS computes on this GPU (several ranks per GPU): the kernels whose workgroups wait
 *                                 for each other inside a launch (cluster tower <= 40 boards, pair tower 41 ... 512, the looping tree kernel
 *                                 of a batch's last 16 games: one workgroup per game, each alone on its compute unit -- also refused on a
 *                                 device with fewer compute units than games) are not used; the caller's training step does the same with
 *                                 diee_train_set_bn_coop(0).  Default 0.
 *   tower_pair            0 | 1   the pair tower alone
 *   tower_cl              "max_boards:boards_per_cluster,..."   the cluster tower's table ("none": per-layer kernels below 257 boards)
 *   tower_table           "min_boards:geometry,..."             the fused tower's table (development)
 *   compact               0 | 1   above 256 live games evaluate only the slots whose leaf needs it (default 1; 0 = every row, like the reference)
 *   spec_eval             0 | 1   speculative leaf evaluation in the free rows of the launches of a batch's tail (default 1; same results)
 *   spec_max_games        live games (all batches of the call) up to which a move-step's search runs that way (default 96, at most 128) -- where the
 *                                 free-running search below does not take the move-step first (by default it takes 17 ... 800 live games)
 *   spec_rows64_from, spec_rows128_from   live games from which a tail launch carries 64 / 128 rows instead of 32 (defaults 5 / 10)
 *   spec_extra_rows       candidates a game may find beyond its share of a tail launch whose rows are scarce: they take what other games leave free (default 2)
 *   spec_child_rows       children of a leaf that waits for its evaluation that are evaluated in the same tail launch at most (default 16; 0: off)
 *   spec_fused_games      129 ... this many live games search that way too, on 512-row launches of the fused kernel family (default 256 = at most; 0: off)
 *   free_eval             0 | 1   the free-running search at free_min_games ... free_max_games live games (defaults 17 ... 800; at most 1024): every
 *                                 game keeps its own iteration counter, a launch of 512 / 1024 rows (free_rows1024_from, default 200 games; 128 rows of the cluster family below 41 games) carries the
 *                                 leaves the games wait for and the nodes their virtual descents predict; no kernel of it waits for a co-resident
 *                                 workgroup.  Default 1; same results.  free_rollout_steps / free_cand_max (12 / 6): virtual descents / candidates per
 *                                 game and round; free_iter_cap (4): iterations a game runs per launch at most (lifted for the last games of a search); free_lag_boost / free_lag_step (4 / 4): the games behind the leader are granted their rows up to so many ranks earlier, one per so many iterations of lag; free_cand_x4 (8): candidates per game and round = 1 + this / 4 x the spare rows per game; free_ring (128): launches whose rows are kept; free_lds_nodes: cap of the tree nodes staged in LDS (tests)
 *   spec_ring_mb          MiB of HBM the ring of evaluated rows of such a search may take ((iterations + 1) launches x rows x 5.7 KB; default 8192):
 *                                 a search whose ring would be larger runs one launch per iteration instead
 *   pinned_pool_mb        MiB of page-locked output blocks the PROCESS keeps for reuse after diee_free_fragments (default 8192;
 *                                 with several ranks per host: what each may retain)
 *   deliver_stage_rows, deliver_rows_per_game, nodes_per_expansion, path_cap        buffer sizes (tests)
 *   cl_pack, cl_grow, expand2, expand2c, spec_rollout_steps (virtual descents per game and launch, default 24), spec_fused_from, fused_heads, cluster_heads, cluster_init, trace_steps,
 *   trace_dispatch, test_starve_at, test_tail_skip                                                  development / test switches
 * Unknown key or malformed value: DIEE_ERR_ARG.  Not for a tic-tac-toe ctx (DIEE_ERR_UNSUPPORTED). */
diee_status diee_set_option(diee_ctx*, const char* key, const char* value);
diee_status diee_get_option(diee_ctx*, const char* key, char* value /*[cap]*/, size_t cap);

/* ---- network weights ---------------------------------------------------------------------
 * Replaces ResNet::new / VarStore::load (src/alphazero/nnet.rs:57-118, alphazero.rs:81-100).
 * The blob is fp32 in the creation order of nnet.rs:62-97:
 *   init:   conv.w[256][6][3][3] conv.b[256]  bn.gamma bn.beta bn.mean bn.var [256 each]
 *   19 x:   conv1.w[256][256][3][3] conv1.b  conv2.w conv2.b  bn1.{g,b,m,v}  bn2.{g,b,m,v}
 *   policy: conv.w[32][256][3][3] conv.b[32] bn.{g,b,m,v}[32] fc.w[1352][768] fc.b[1352]
 *   value:  conv.w[3][256][3][3]  conv.b[3]  bn.{g,b,m,v}[3]  fc.w[1][72]     fc.b[1]
 * BatchNorm is folded (eval mode, eps 1e-5) and weights are packed to bf16 MFMA fragments. */
size_t      diee_weights_count(int game_id);       /* same layout for DIEE_GAME_TTT with 64 filters, 4 blocks, 3 input planes, 3x3 */
/* tch-default random init (nnet.rs: kaiming-uniform conv/linear weights, conv bias 0, linear bias
 * U(+-1/sqrt(fan_in)), BN gamma U(0,1), beta 0, mean 0, var 1) from a counter-based generator */
diee_status diee_random_weights(int game_id, uint64_t seed, float* blob, size_t n);
diee_status diee_load_weights(diee_ctx*, const float* blob, size_t n);

/* batch-size independent network arithmetic for every later call on this ctx (what DIEE_FLAG_INVARIANT_NN
 * selects for one call): see the flag.  Needs loaded weights. */
diee_status diee_set_invariant_nn(diee_ctx*, int on);

/* ---- ResNet::forward_t, nnet.rs:120-133 (eval mode): softmax policy [n][1352], tanh value [n] */
diee_status diee_nn_forward(diee_ctx*, const diee_bg_state* states, uint32_t n,
                            float* policy, float* value);

/* ---- level 1: alpha_mcts_parallel (alpha_mcts.rs:91) + get_prob_tensor_parallel (utils.rs:42)
 * roots[i] must have dice rolled.  game_ids/rounds (may be NULL: i / 0) key the child dice.
 * visit_probs[n][1352]: root child visits / row sum (rows of roots without children are NaN,
 * like the reference's 0/0); n_children[n]; child_visits (optional, [n][cap] in child order). */
diee_status diee_mcts_batch(diee_ctx*, const diee_bg_state* roots, uint32_t n,
                            const diee_mcts_cfg* cfg, uint64_t seed, uint32_t step,
                            const uint32_t* game_ids, const uint32_t* rounds, uint32_t flags,
                            float* visit_probs, uint32_t* n_children,
                            float* root_visits /* [n] or NULL */, diee_stats* stats);

/* ---- level 2: AlphaZero::self_play_parallel, alpha_parallel.rs:101-231
 * n_games = num_self_play_batches; first_game_id offsets the RNG keys (rank * n_games when games
 * are sharded over GPUs); max_steps = 0 plays to completion. */
diee_status diee_self_play(diee_ctx*, uint32_t n_games, uint32_t first_game_id,
                           const diee_mcts_cfg* cfg, float temperature, uint64_t seed,
                           uint32_t flags, uint32_t max_steps,
                           diee_fragments* out /* may be NULL: keep results in HBM only */,
                           diee_stats* stats);
void        diee_free_fragments(diee_fragments*);

/* ---- level 2, pipelined: n_batches calls of self_play_parallel played side by side on one GPU.
 * learn_parallel issues self_play_iterations such calls back to back with the same network
 * (alpha_parallel.rs:49-62); a single batch ends in a long tail of move-steps with a handful of live
 * games, during which the GPU idles.  Here every batch starts at move-step 0 and all of them share each
 * network launch; each keeps its own seed, game ids, Dirichlet stream, `node_selected` flags and slot-0
 * bookkeeping (Q14), so batch b produces what diee_self_play(batches[b]) produces -- byte for byte when the
 * ResNet kernel is batch-size independent (DIEE_FLAG_INVARIANT_NN), otherwise up to which bf16 tower
 * kernel evaluated a row (the dispatch goes by the number of live games; all are within the NN tolerance).
 * outs / stats: [n_batches] (either may be NULL); stats[b].seconds is the wall clock of the whole call,
 * the sampled tower timings are reported with batch 0. */
typedef struct {
    uint32_t n_games;          /* num_self_play_batches of this call                        */
    uint32_t first_game_id;    /* offsets the RNG keys, as in diee_self_play                 */
    uint64_t seed;
} diee_batch;
diee_status diee_self_play_multi(diee_ctx*, const diee_batch* batches, uint32_t n_batches,
                                 const diee_mcts_cfg* cfg, float temperature, uint32_t flags,
                                 uint32_t max_steps, diee_fragments* outs, diee_stats* stats);

/* ---- training-step kernels (no ctx: plain device pointers of the caller's framework, launched on `stream`, a
 * hipStream_t; NULL = the default stream).  AlphaZero::train (alphazero.rs:202-261) is tch autograd in the reference;
 * here the 38 tower convolutions of the step run on the engine's MFMA conv kernel in the NHWC token layout
 * x[board*24 + point][256] bf16 (die-e_amd/train_ops.py wraps these in a torch.autograd.Function):
 *   forward   y = conv3x3(x, W) + b                   diee_train_conv3x3(x, pack(W, 0), b, y)
 *   dgrad     dx = conv3x3(dy, W'), W' = W transposed and flipped   diee_train_conv3x3(dy, pack(W, 1), NULL, dx)
 *   wgrad     dW[t*256 + c][n] = col^T x dy, col = diee_train_im2col3x3(x)   (the GEMM is the framework's)  */
diee_status diee_train_pack_conv3x3(const float* w_oihw /*[256][256][3][3]*/, void* wpack /*589 824 bf16*/, int transpose, void* stream);
/* every tower convolution of a step in one launch: w = host array of n (<= 64) device pointers to OIHW fp32 weights;
 * wpack[n][2][589 824] bf16 receives the forward (0) and the transposed (1) packing of each */
diee_status diee_train_pack_conv3x3_multi(const float* const* w_oihw, int n, void* wpack, void* stream);
diee_status diee_train_conv3x3(const void* x_bf16, const void* wpack, const float* bias /*[256] or NULL*/, void* y_bf16,
                               int boards, void* stream);
diee_status diee_train_im2col3x3(const void* x_bf16, void* col_bf16 /*[boards*24][2304]*/, int boards, void* stream);
/* BatchNorm2d in training mode over the rows (= batch x 4 x 6) fused with the residual add and the ReLU of
 * ResBlock::forward_t (nnet.rs:24-34): y = relu(gamma * (x - mean) / sqrt(var + eps) + beta [+ res]); updates the running
 * statistics (momentum, unbiased variance) when given.  scratch: diee_train_scratch_floats(rows) floats.  Deterministic.
 * Each pass is ONE launch while rows / 64 workgroups can be resident together (they meet on a device counter, one set of
 * counters per stream), three launches above that, with DIEE_BN_COOP=0 or after diee_train_set_bn_coop(0).
 * dx_colsum, when given, receives the column sums of dx: the bias gradient of the convolution that produced x. */
size_t      diee_train_scratch_floats(int rows);
diee_status diee_train_bn_relu_fwd(const void* x_bf16, const void* res_bf16 /*or NULL*/, const float* gamma, const float* beta,
                                   float* running_mean /*or NULL*/, float* running_var, float momentum, float eps,
                                   float* save_mean /*[256]*/, float* save_invstd /*[256]*/, void* y_bf16, int rows,
                                   float* scratch, void* stream);
/* The one-launch passes count on having the device to themselves (their workgroups wait for each other inside the launch;
 * every stream has its own barrier words, the waits are bounded and a starved pass writes NaN statistics instead of hanging).
 * diee_train_set_bn_coop(0) selects the three-launch passes for the rest of the process (callers whose step shares the GPU:
 * RCCL kernels of a data-parallel step, a second rank or process; DIEE_BN_COOP=0 does the same from the environment).
 * diee_train_bn_coop_timeouts(clear): waits for the device, then returns the timeouts since the last clear (bit 0 forward,
 * bit 1 backward; -1 = the query failed) and, with clear != 0, re-arms the barrier words -- the caller then repeats the step
 * (die-e_amd/alphazero.py restores its pre-epoch snapshot and re-runs the epoch on the three-launch passes). */
diee_status diee_train_set_bn_coop(int on);
int         diee_train_bn_coop_timeouts(int clear);
/* its backward: dx (to the convolution), dres (= dy masked by the ReLU, to the skip connection; may be NULL), dgamma, dbeta */
diee_status diee_train_bn_relu_bwd(const void* dy_bf16, const void* y_bf16, const void* x_bf16, const float* gamma,
                                   const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                                   void* dx_bf16, void* dres_bf16 /*or NULL*/, float* dx_colsum /*[256] or NULL*/, int rows,
                                   float* scratch, void* stream);
/* weight gradient of the tower convolution, hand-written: dW[n][c][ky][kx] (fp32, OIHW) = sum over rows of
 * x[row + shift(ky,kx)][c] * dy[row][n]; scratch: diee_train_wgrad_scratch_floats() floats.  Deterministic. */
size_t      diee_train_wgrad_scratch_floats(void);
diee_status diee_train_wgrad3x3(const void* x_bf16, const void* dy_bf16, float* dw_oihw /*[256][256][3][3]*/, int boards,
                                float* scratch, void* stream);
/* out[c] = sum over rows of a[row][c] (the convolution's bias gradient) */
diee_status diee_train_colsum(const void* a_bf16, float* out /*[256]*/, int rows, float* scratch, void* stream);

/* ---- pure game functions, batched on the GPU (parity tests; LearnableGame trait, base.rs:8-51)
 * A play is int8 {f1,t1,f2,t2}; unused slots DIEE_NO_MOVE. */
/* get_valid_moves, backgammon_logic.rs:403-414: plays[n][cap][4], counts[n] (count may exceed cap) */
diee_status diee_bg_legal_moves(diee_ctx*, const diee_bg_state* s, uint32_t n,
                                int8_t* plays, uint32_t cap, uint32_t* counts);
/* encode / decode, backgammon_logic.rs:262-359 / :361-401 */
diee_status diee_bg_encode(diee_ctx*, const diee_bg_state* s, const int8_t* plays /*[n][4]*/,
                           uint32_t n, uint32_t* codes);
diee_status diee_bg_decode(diee_ctx*, const diee_bg_state* s, const uint32_t* codes, uint32_t n,
                           int8_t* plays /*[n][4]*/);
/* apply_move, backgammon_logic.rs:176-186, with the dice roll_die would draw (dice[n][2]) */
diee_status diee_bg_apply(diee_ctx*, diee_bg_state* s /* in/out */, const int8_t* plays,
                          const uint8_t* dice, uint32_t n);
/* as_tensor, backgammon_logic.rs:198-252: out[n][144] */
diee_status diee_bg_planes(diee_ctx*, const diee_bg_state* s, uint32_t n, float* out);
/* ---- LearnableGame for TicTacToe (src/tictactoe/mod.rs:36-94), host functions, no ctx */
uint32_t diee_ttt_valid_moves(const diee_ttt_state* s, uint8_t* moves /*[9]*/);     /* get_valid_moves :36-44 -> count */
void     diee_ttt_apply_move(diee_ttt_state* s, uint8_t move);                       /* apply_move :46-49 */
int      diee_ttt_check_winner(const diee_ttt_state* s, int* winner);               /* check_winner :60-81: 1 = Some(*winner) (0 = draw), 0 = None */
void     diee_ttt_planes(const diee_ttt_state* s, float* out /*[27]*/);              /* as_tensor :83-94 */

/* Tensor::pow_(1 / temperature) as the self-play and arena drivers apply it to the visit distribution
 * (alpha_parallel.rs:165, versus.rs:283): out[i] = x[i] ^ y[i] with the engine's deterministic powf (IEEE operations only,
 * <= 1 ulp from libm, the same bits on the host oracle and on the GPU) */
diee_status diee_det_pow(diee_ctx*, const float* x, const float* y, uint32_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif
