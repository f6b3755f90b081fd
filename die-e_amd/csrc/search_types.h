// search_types.h -- kernel parameter blocks (plain device pointers) shared by mcts_kernels.hip and
// the host driver.  Layout in HBM:
//   Tree   : one fixed-stride arena per live-game slot (slot*node_cap + local index); node statistics
//            are SoA so that a wave reading the children of a node issues coalesced 4-byte loads;
//            children of a node are contiguous (first_child, n_children).
//   Slots  : per-slot search state of one move-step (slot = index into the live list).
//   Games  : per-game state of the self-play batches in flight (game = index over all batches, batch-major).
//   Segs   : the self-play batches ("segments") that share the slot space.  A batch is one call of the reference's
//            self_play_parallel: its games are coupled through the shared Dirichlet draw of a move-step, the
//            `node_selected` flag of an iteration and the stale-slot quirk (Q14), and through nothing else -- so K
//            batches can run side by side in ONE slot space (live games of batch 0, then of batch 1, ...) and share
//            every network launch, each with its own seed, flags and slot-0 bookkeeping.
#pragma once
#include <stdint.h>

#include "bg_device.h"

namespace diee {

enum Counter {
    CNT_NN_EVALS = 0, CNT_EXPANSIONS, CNT_CHILDREN, CNT_TERMINAL, CNT_DEPTH_SUM, CNT_SELECTIONS,
    CNT_ILLEGAL, CNT_MAX_CHILDREN, CNT_PLIES, CNT_GAMES, CNT_NN_ROWS, CNT_COUNT
};
constexpr uint32_t kMaxSegments = 64;

enum SlotCounter { SC_SELECTIONS = 0, SC_DEPTH_SUM, SC_TERMINAL, SC_EXPANSIONS, SC_CHILDREN, SC_MAX_CHILDREN, SC_ILLEGAL, SC_NN_ROWS, SC_COUNT };

struct Tree {
    float* visits;          // Node.visits  (f32 like the reference, node.rs:14)
    float* value;           // Node.value
    float* prior;           // Node.policy
    uint32_t* parent;       // local index, 0xFFFFFFFF = None
    uint32_t* first_child;  // local index of the first child
    uint32_t* meta;         // bit31 expanded ("expandable_moves drained"), bit30 / bit29 finished game / won by +1, bits28..16 n_children, bits15..0 action code
    BgState* state;         // Node.state (dice frozen at creation)
    uint32_t* used;         // [slots] bump allocator
    uint32_t node_cap;      // nodes per slot
};

struct Segs {
    const unsigned long long* seed;   // [segs] engine seed of the batch (keys dice, sampling, Dirichlet)
    const uint32_t* first_id;         // [segs] RNG key of the batch's game 0 (first_game_id)
    const uint32_t* game0;            // [segs] index of the batch's game 0 in the Games arrays
    uint32_t* first_slot;             // [segs] first live slot of the batch this move-step ...
    uint32_t* end_slot;               // [segs] ... and one past its last (both 0: no live game)
    uint32_t n;                       // batches in flight
    uint32_t iter_cap;                // iterations per batch in Slots::iter_flags
};

struct Slots {
    BgState* roots;         // [slots]
    BgState* eval_states;   // [slots] states evaluated by the ResNet this iteration (stale rows keep their content)
    uint32_t* game_id;      // [slots] RNG key
    uint32_t* round;        // [slots] RNG key
    uint32_t* seg;          // [slots] batch the slot's game belongs to
    uint32_t* leaf;         // [slots] selected leaf this iteration
    uint32_t* sel;          // [slots] selected_nodes_idxs (persists over iterations; 0xFFFFFFFF = initial)
    float* sel_value;       // [slots] NN value of sel
    uint8_t* leaf_term;     // [slots] leaf was terminal this iteration
    const uint32_t* slot_row;   // [slots] row of the slot's evaluation in the network outputs when the batch was compacted
                                // (k_row_map: slots with a terminal leaf get no row); null = row == slot
    const float* logits;    // [slots][1352] policy logits of this iteration's evaluation (nn_host's buffers)
    const float* hv;        // [slots][72] value features
    const float* wv;        // [73] value FC
    float* noise;           // [segs][1352] Dirichlet sample of this move-step, one per batch (noise.rs:27-34)
    float* root_value0;     // [segs] NN value of the root in the batch's first slot
    uint32_t* iter_flags;   // [segs][iter_cap][2] any_selected, stale-initial count per iteration
    unsigned long long* counters;  // [segs][CNT_COUNT] totals (written by k_reduce_counters / single lanes only)
    uint32_t* slot_cnt;     // [slots][SC_COUNT] per-slot counters of this move-step: same-address atomics from a
                            // thousand waves serialise at ~11 ns each, per-slot words cost nothing
    uint32_t* overflow;     // capacity flag (bit0 sequence table, bit1 tree arena)
    uint32_t* leaf_meta;    // [slots] Tree::meta of leaf as the selection read it (nobody else writes this game's tree)
    uint32_t* path;         // [slots][kPathCap] nodes from the root to sel (entry d = depth d): backpropagation without the parent walk
    uint8_t* path_len;      // [slots] entries of path; 0 = deeper than path_cap (or no selection yet): walk the parents
    uint32_t* grow_k;       // [slots] children the growth workgroups of a tower launch created for the slot's leaf this iteration (0xFFFFFFFF: none to create / no room)
    uint16_t* grow_code;    // [slots][kMaxPlays] their action codes, child lane + 64 q at [lane * 4 + q]
    uint32_t path_cap;      // depths recorded (<= kPathCap = 64; option path_cap lowers it so that tests reach the parent walk)
};
constexpr int kPathCap = 64;

struct Games {
    BgState* state;         // [games]
    uint32_t* rounds;       // n_rounds
    uint32_t* nfrags;
    uint8_t* alive;
    int8_t* winner;
    uint32_t* ev_a_count;   // round-limit flush: fragments flushed with outcome 0 (0xFFFFFFFF = none)
    uint32_t* ev_a_step;
    uint32_t* ev_b_count;   // win flush
    uint32_t* ev_b_step;
    uint32_t* live;         // [games] live list (ascending game index, hence batch-major)
    uint8_t* seg;           // [games] batch of the game
    float* frag_ps;         // [games][frag_cap][1352]
    float* frag_planes;     // [games][frag_cap][144]
    int8_t* frag_player;    // [games][frag_cap]
    uint32_t frag_cap;
    unsigned long long* counters;
};

// Output delivery of one move-step (alpha_parallel.rs:172-180, :215-223: a game's memory joins all_memories, relabelled, in
// the step the game is removed).  k_deliver_scan lists this step's flushes in the reference's order -- by game, round-limit
// flush before win flush, inside each batch -- with the row each starts at in its batch's output of the step; k_deliver_copy
// gathers a range of those rows (ps, planes, outcome, game id) into a staging buffer, which the host copies out on a second
// stream while the next move-step searches.
struct DeliverEvent { uint32_t g, kind_count, row0, seg; };   // kind_count = kind << 31 | rows; row0 = first row in the step's output of the batch
struct DeliverSummary {     // indices into the per-step summary the host reads with the live counts ([1 + 4 * segs] words)
    static constexpr uint32_t kLive = 0;                      // [0] games alive, [1 + b] alive in batch b   (k_compact_live)
    static __host__ __device__ uint32_t rows(uint32_t segs, uint32_t b) { return 1 + segs + b; }       // rows batch b delivers this step
    static __host__ __device__ uint32_t ev0(uint32_t segs, uint32_t b) { return 1 + 2 * segs + b; }    // its first event
    static __host__ __device__ uint32_t nev(uint32_t segs, uint32_t b) { return 1 + 3 * segs + b; }    // its events
};
struct DeliverOut { float* ps; float* planes; int8_t* outcome; uint32_t* game; };    // staging, row-major

// ---- the tail of a batch: speculative leaf evaluation (round 5) -----------------------------------------------------------------
// At <= 32 live games a network evaluation costs the same ~95 us whether its launch carries 1 or 32 boards (one cluster-tower
// launch, k_tower_cl<1, 8>; 125 us up to 64, 172 us up to 128 boards), and a search iteration is one such launch + the tree kernel:
// 147 of a batch's 364 move-steps are played with <= 32 live games, a serial chain of ~100 launches each.  But every leaf a search selects exists -- state, frozen dice -- from the moment its parent was
// expanded, and below 129 boards a row's network output is a pure function of its state (the split-K cluster family is one
// arithmetic, rows batch-independent: tests/test_nn_gpu.py).  So a launch's free rows carry unexpanded nodes the search is likely to
// select next (found by a virtual descent of the same PUCT rule on a scratch copy of the tree's statistics: "UCB selection staged in
// LDS"), every evaluated row stays in a ring of [launch][row] outputs, and an iteration whose selected leaves were all evaluated
// earlier needs NO launch: k_tail (mcts_kernels.hip) runs the iterations of all live games in lockstep, one after the other inside
// one launch, for as long as every selected leaf is in the ring, and plans the next launch's rows when one is not.  The search
// itself -- selection, expansion, backpropagation, the quirks' coupling of the games of a batch -- is the same code on the same
// numbers in the same order (expand_body): results are bit-identical to the launch-per-iteration path, which remains above spec_max_games (96) games.
// What rides in a launch beside the demanded leaves: every game's share of candidates (virtual descents), the candidates beyond the shares
// where rows are scarce (extra_rows: they take what other games leave free), and the children of a demanded leaf (child_rows: the one
// parent whose expanding iteration -- the key of its children's dice -- is known before it is expanded).
constexpr uint32_t kTailMaxSlots = 256;   // live games (all batches of the call) up to which a move-step's search may run this way: one workgroup per CU
                                          // (options spec_max_games <= 128 on the cluster family's launches, spec_fused_games for 129 ... 256 on the fused family's)
constexpr int kTailFusedRows = 512;      // rows of a tail launch at 129 ... 256 live games: the 4-board pair tower + k_policy_fc (the fused family, like those games' plain evaluations)
constexpr uint32_t kTailRowsMax = 128;    // rows of a tail launch at most: 32 (k_tower_cl<1, 8>, ~95 us), 64 (<2, 8>, ~125 us) or 128 (<4, 8>, ~172 us) -- one
                                          // arithmetic, so a row's bits do not depend on which of them evaluated it; the more games share a launch,
                                          // the more rows it carries (Tail::rows, search_host.cpp tail_rows_for)
constexpr uint32_t kTailLdsNodes = 3072;  // nodes of a game's tree whose statistics the virtual descent stages in LDS (beyond: read in place, not updated)
struct Tail {
    uint32_t* crow;         // [kTailMaxSlots][node_cap] ring row + 1 of a node's evaluation (0: none yet)
    float* cval;            // [kTailMaxSlots][node_cap] its value (for the virtual descents only: expansions recompute it from the row)
    BgState* rows_state;    // [launches][rows] states the tower launch q evaluates
    uint32_t* rows_node;    // [launches][rows] slot << 24 | node
    float* logits;          // [launches][rows][1352] the ring
    float* hv;              // [launches][rows][72]
    uint32_t* n_rows;       // [launches] rows of launch q (0: nothing to evaluate, the launch returns at once)
    uint32_t* n_dem;        // [launches] ... of which demanded (the games whose selected leaf had no evaluation): the bench's accounting
    uint32_t* state;        // [0] next iteration, [1] done, [2] launches that carried rows, [3] rows evaluated on speculation
    uint32_t* bar;          // [2 * launches + 8] one word per meeting of the games' workgroups (zeroed per move-step)
    uint32_t* bar2;         // [launches] the meeting inside the plan of launch q (extra_rows: shares taken before the free rows left over are)
    uint32_t* host;         // pinned host words: [0] next iteration, [1] done, [2] index of the last k_tail launch that finished
    uint32_t launches;      // iterations + 1
    uint32_t iterations;
    uint32_t rollout_steps; // virtual descents per game and launch at most
    uint32_t rows;          // rows of a launch in this move-step (32 / 64 / 128; 512 at 129 ... 256 live games)
    uint32_t child_rows;    // children of a demanded leaf (created ahead: their dice are keyed by the iteration that will expand it) that ride in its launch at most (0: off)
    uint32_t extra_rows;    // candidates a game may find beyond its share of a full launch's rows: they take what other games left free (0: off)
    uint32_t test_skip;     // tests: 1 + the meeting word at which the last game's workgroup does NOT arrive (0: never): every workgroup's bounded
                            // spin runs out, the starved bit is raised, the host repeats the move-step's search launch by launch
};

// ---- free-running search (round 6): 257 ... 928 live games ---------------------------------------------------------------------------
// Above the tail's reach every MCTS iteration used to cost one network launch sized by the live games -- 300 us for <= 512 boards,
// ~480 ... 520 us for 513 ... 928, 577 us for a full pass of the chip: the rows of a part-empty launch cost up to twice a full one's.  The
// reference couples the games of a call only through `node_selected` (alpha_mcts.rs:151,170) and the stale slots (:142,192-200), so
// every game keeps its OWN iteration counter: a round = one launch of up to `rows` rows (the fused family, rows gathered from the tree
// arena by index: launch_tower_compact) + k_free, where each game takes in its rows, runs iterations for as long as its selected leaf is
// a finished game or has its evaluation (and the flags it needs are final), then lists its wishes -- the leaf it waits for, then the
// unexpanded nodes its virtual PUCT descents end on -- and k_free_pack grants them: every demanded leaf, then the wishes rank by rank
// until the launch is full.  No workgroup ever waits for another one inside a launch (a game that needs a flag another game has not
// published yet simply stops for this round), so nothing has to be co-resident.  Priced before it was built: tests/tools/free_price.*,
// profiles/r06a_*.  Results are bit-identical to the lockstep search: the same operations on the same numbers in the same order per game.
constexpr uint32_t kFreeWish = 24;        // wishes a game may list per round (demanded leaf included)
constexpr uint32_t kFreeMaxSlots = 1024;  // k_free_pack: one thread per game
constexpr uint32_t kFreeBoostMax = 16;
struct Free {
    uint32_t* crow;         // [slots][node_cap] 1 + launch * rows + row of a node's evaluation (0: none); valid while launch + ring > the current launch
    float* cval;            // [slots][node_cap] its value (virtual descents only)
    uint32_t* rows_idx;     // [ring][rows] arena index (slot * node_cap + node) of every row of a launch: the tower gathers its states through it
    BgState* rows_state;    // [ring][rows] the rows' states, dense: what the cluster tower of a launch of at most 128 rows reads (null above: gathered by index)
    float* logits;          // [ring][rows][1352]
    float* hv;              // [ring][rows][72]
    uint32_t* n_rows;       // [launches] rows of launch q
    uint32_t* n_dem;        // [launches] ... of which demanded (a game's selected leaf waiting for its evaluation): the bench's accounting
    uint32_t* grant_off;    // [slots] first row of the slot in the launch planned last ...
    uint32_t* grant_cnt;    // [slots] ... and how many it got
    uint32_t* wish;         // [slots][kFreeWish] nodes, most wanted first
    uint32_t* wish_n;       // [slots] count | 0x80000000 when wish[0] is the demanded leaf | progress << 8 (the packer may favour games behind)
    uint32_t* prog;         // [slots] the iteration whose selection the game has published = its own iteration counter
    uint32_t* first_sel;    // [slots] iteration of the game's first real selection (0xFFFFFFFF: none yet): Q14's count matters only before it
    uint32_t* state;        // [0] all games done, [1] launches that carried rows, [2] rows evaluated on speculation, [4] games not done when the last round was
                            // packed (k_free sizes its budgets by them: the last games of a search have the launches to themselves)
    uint32_t* host;         // pinned: [0] done, [1] last k_free_pack that finished
    uint32_t launches;      // rounds the host sends at most: 4 x (iterations + games) + 66 (search_host.cpp free_view)
    uint32_t iterations;
    uint32_t rows;          // rows of a launch at most (512: pair tower; 1024: one pass of the chip)
    uint32_t ring;          // launches whose rows stay in the ring
    uint32_t lds_nodes;     // nodes of a game's tree staged in LDS (by the workgroups that share a CU)
    uint32_t rollout_steps; // virtual descents per game and round at most
    uint32_t cand_max;      // candidates a game lists at most (< kFreeWish)
    uint32_t lag_boost, lag_step;   // k_free_pack: a game lag_step iterations behind the leader is served one rank earlier, up to lag_boost ranks (<= kFreeBoostMax)
    uint32_t iter_cap;      // iterations a game runs in one launch at most: the launch lasts as long as its busiest game (a game whose leaves are
                            // finished games needs no rows and would run its whole search in the first one)
};

struct SearchParams {
    float dir_eps;
    uint32_t quirks;
};

struct PlayParams {
    uint32_t round_limit;
    float inv_temperature;
    uint32_t quirks;
};

}  // namespace diee
