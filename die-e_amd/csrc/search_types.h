// search_types.h -- kernel parameter blocks (plain device pointers) shared by mcts_kernels.hip and
// the host driver.  Layout in HBM:
//   Tree   : one fixed-stride arena per live-game slot (slot*node_cap + local index); node statistics
//            are SoA so that a wave reading the children of a node issues coalesced 4-byte loads;
//            children of a node are contiguous (first_child, n_children).
//   Slots  : per-slot search state of one move-step (slot = index into the live list).
//   Games  : per-game state of a self-play batch (game = index in the batch).
#pragma once
#include <stdint.h>

#include "bg_device.h"

namespace diee {

enum Counter {
    CNT_NN_EVALS = 0, CNT_EXPANSIONS, CNT_CHILDREN, CNT_TERMINAL, CNT_DEPTH_SUM, CNT_SELECTIONS,
    CNT_ILLEGAL, CNT_MAX_CHILDREN, CNT_PLIES, CNT_GAMES, CNT_COUNT
};

enum SlotCounter { SC_SELECTIONS = 0, SC_DEPTH_SUM, SC_TERMINAL, SC_EXPANSIONS, SC_CHILDREN, SC_MAX_CHILDREN, SC_ILLEGAL, SC_COUNT };

struct Tree {
    float* visits;          // Node.visits  (f32 like the reference, node.rs:14)
    float* value;           // Node.value
    float* prior;           // Node.policy
    uint32_t* parent;       // local index, 0xFFFFFFFF = None
    uint32_t* first_child;  // local index of the first child
    uint32_t* meta;         // bit31 expanded ("expandable_moves drained"), bits30..16 n_children, bits15..0 action code
    BgState* state;         // Node.state (dice frozen at creation)
    uint32_t* used;         // [slots] bump allocator
    uint32_t node_cap;      // nodes per slot
};

struct Slots {
    BgState* roots;         // [slots]
    BgState* eval_states;   // [slots] states evaluated by the ResNet this iteration (stale rows keep their content)
    uint32_t* game_id;      // [slots] RNG key
    uint32_t* round;        // [slots] RNG key
    uint32_t* leaf;         // [slots] selected leaf this iteration
    uint32_t* sel;          // [slots] selected_nodes_idxs (persists over iterations; 0xFFFFFFFF = initial)
    float* sel_value;       // [slots] NN value of sel
    uint8_t* leaf_term;     // [slots] leaf was terminal this iteration
    const float* logits;    // [slots][1352] policy logits of this iteration's evaluation (nn_host's buffers)
    const float* hv;        // [slots][72] value features
    const float* wv;        // [73] value FC
    float* noise;           // [1352] Dirichlet sample of this move-step
    float* root_value0;     // [1] NN value of slot 0's root
    uint32_t* iter_flags;   // [2*(iterations)] any_selected, stale-initial count per iteration
    unsigned long long* counters;  // [CNT_COUNT] totals (written by k_reduce_counters / single lanes only)
    uint32_t* slot_cnt;     // [slots][SC_COUNT] per-slot counters of this move-step: same-address atomics from a
                            // thousand waves serialise at ~11 ns each, per-slot words cost nothing
    uint32_t* overflow;     // capacity flag (bit0 sequence table, bit1 tree arena)
};

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
    uint32_t* live;         // [games] live list (ascending game index)
    float* frag_ps;         // [games][frag_cap][1352]
    float* frag_planes;     // [games][frag_cap][144]
    int8_t* frag_player;    // [games][frag_cap]
    uint32_t frag_cap;
    unsigned long long* counters;
};

struct SearchParams {
    uint64_t seed;
    float dir_eps;
    uint32_t quirks;
};

struct PlayParams {
    uint64_t seed;
    uint32_t first_id;
    uint32_t round_limit;
    float inv_temperature;
    uint32_t quirks;
};

}  // namespace diee
