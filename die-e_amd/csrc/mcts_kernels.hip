// mcts_kernels.hip -- batched AlphaZero-style MCTS and the self-play move step for gfx950.
//
// Semantics follow the reference's src/mcts/alpha_mcts.rs:14-33,91-202, node.rs:98-112,157-174,
// simple_mcts.rs:96-103, utils.rs:42-84, noise.rs:27-34 and src/alphazero/alpha_parallel.rs:
// 129-229; the design does not: every live game owns a fixed-stride tree arena in HBM (SoA node
// statistics, children contiguous), one wavefront walks / expands / backpropagates one game, and
// an MCTS iteration is select kernel -> ResNet -> expand kernel with no host round trip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "bg_device.h"
#include "nn_device.h"
#include "launch.h"
#include "mcts_device.h"
#include "search_types.h"
#include "wave_ops.h"

namespace diee {

// backpropagate, simple_mcts.rs:96-103: same sign at every level (one lane)
__device__ __forceinline__ void backprop(const Tree& T, size_t base, uint32_t idx, float v) {
    while (idx != kNone) {
        T.visits[base + idx] += 1.0f;
        T.value[base + idx] += v;
        idx = T.parent[base + idx];
    }
}

// the same update for a selection whose root-to-leaf nodes were recorded (entry d = depth d): lane d updates its node,
// one round of memory latency instead of one per level.  Every node is touched once, by one lane: the same bits.
__device__ __forceinline__ void backprop_path(const Tree& T, size_t base, uint32_t my_node, int lane, uint32_t len, float v) {
    if ((uint32_t)lane < len) {
        T.visits[base + my_node] += 1.0f;
        T.value[base + my_node] += v;
    }
}

// ---- roots -------------------------------------------------------------------------------------
// alpha_mcts.rs:110-112,123: one root node per state, visits = 1
__global__ void k_init_roots(Tree T, Slots S, uint32_t n) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const size_t base = (size_t)slot * T.node_cap;
    const BgState s = load_state(&S.roots[slot]);
    store_state(&T.state[base], s);
    store_state(&S.eval_states[slot], s);
    T.visits[base] = 1.0f; T.value[base] = 0.0f; T.prior[base] = 0.0f;
    const uint32_t m0 = 0xFFFFu | meta_terminal_bits(s);      // (a finished root: k_tail reads the bits where select_slot reads the state)
    T.parent[base] = kNone; T.first_child[base] = 0; T.meta[base] = m0;
    T.used[slot] = 1;
    S.sel[slot] = kNone; S.sel_value[slot] = 0.0f; S.leaf[slot] = 0; S.leaf_term[slot] = 0; S.path_len[slot] = 0;
    S.leaf_meta[slot] = m0;
#pragma unroll
    for (int c = 0; c < SC_COUNT; ++c) S.slot_cnt[slot * SC_COUNT + c] = 0;
}

// ---- selection ---------------------------------------------------------------------------------
struct Best { float s; int j; };
__device__ __forceinline__ Best better(Best a, Best b) {      // later index wins ties (Iterator::max_by)
    if (b.j < 0) return a;
    if (a.j < 0) return b;
    if (a.s > b.s || (a.s == b.s && a.j > b.j)) return a;
    return b;
}

// the best of the wave in every lane; `better` is a symmetric total order, so the pairing order is free (wave_ops.h)
__device__ __forceinline__ Best wave_best(Best b) {
    Best o;
    o.s = dpp_f32<kDppXor1>(b.s); o.j = dpp_i32<kDppXor1>(b.j); b = better(b, o);
    o.s = dpp_f32<kDppXor2>(b.s); o.j = dpp_i32<kDppXor2>(b.j); b = better(b, o);
    o.s = dpp_f32<kDppHalfMirror>(b.s); o.j = dpp_i32<kDppHalfMirror>(b.j); b = better(b, o);
    o.s = dpp_f32<kDppMirror>(b.s); o.j = dpp_i32<kDppMirror>(b.j); b = better(b, o);
    const int si = __builtin_bit_cast(int, b.s);
    Best r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        r[q].s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(si, 16 * q));
        r[q].j = __builtin_amdgcn_readlane(b.j, 16 * q);
    }
    return better(better(r[0], r[1]), better(r[2], r[3]));
}

// alpha_select_leaf_node / select_alpha (alpha_mcts.rs:14-33) with alpha_ucb (node.rs:98-112):
//   q + (c * (sqrt(N_parent) / (n + 1))) * p, f32, in this association; NaN compares Equal.
// One memory round trip per level: a level reads the statistics of all children AND their headers (meta, first child), so
// the chosen child's header is already in a register of the lane that scored it (a wave alone on its SIMD waits out every
// round trip: header -> children -> header -> ... cost two per level).  The nodes of the descent are recorded (lane d keeps
// depth d) for the backpropagation of this selection.
// cn: the slot's counters, held in registers by the caller (a read-modify-write of a counter in memory is a round trip the
// wave waits out).  hdr: the root's header when the caller knows it (it just updated the root itself), else null.
struct NodeHdr { uint32_t meta, first_child; float visits; };
// The first level of the descent without a trip to memory: the root's children always sit at nodes 1 .. k of the slot's
// arena (the root is expanded first), so the caller can request lane j's child (statistics, header, state) with its very
// first burst of loads, long before the descent, and patch in what the kernel itself changed since (this selection's
// backpropagation touches exactly one of them).  north_star: "UCB selection staged" -- in registers, one child per lane;
// the levels below are reached through one round trip each.
struct Level0 { float vis, val, pr; uint32_t cm, cf; BgState cs; bool valid; };
// on_leaf(node, meta, first_child): the descent stands on a node without children; true = the caller had an expansion of exactly
// this node pending and has committed it now (k_expand<true, true> descends while its second wave still creates the children):
// meta / first_child are the node's new header and the descent goes on.
struct NoLeafHook { __device__ __forceinline__ bool operator()(uint32_t, uint32_t&, uint32_t&) const { return false; } };
template <class OnLeaf = NoLeafHook>
__device__ __forceinline__ void select_slot(const Tree& T, const Slots& S, const Segs& G, uint32_t slot, uint32_t seg, int lane,
                                            uint32_t it, float c, uint32_t quirks, uint32_t (&cn)[SC_COUNT], const NodeHdr* hdr,
                                            const Level0* l0 = nullptr, OnLeaf&& on_leaf = OnLeaf()) {
    const size_t base = (size_t)slot * T.node_cap;
    uint32_t* iflag = S.iter_flags + 2 * ((size_t)seg * G.iter_cap + it);            // this batch's flags of iteration `it`
    uint32_t node = 0, depth = 0;
    uint32_t mine = lane == 0 ? 0u : kNone;
    uint32_t mt, fc;
    float nvis;
    if (hdr) { mt = hdr->meta; fc = hdr->first_child; nvis = hdr->visits; }
    else { mt = T.meta[base]; fc = T.first_child[base]; nvis = T.visits[base]; }
    BgState cur;                                            // state of `node` when the level above carried it along
    bool cur_ok = false;
#pragma unroll
    for (int q = 0; q < 8; ++q) cur.w[q] = 0u;
    for (;;) {
        uint32_t k = meta_nch(mt);
        if (k == 0) {
            uint32_t nm = 0, nf = 0;
            if (!on_leaf(node, nm, nf)) break;
            mt = nm; fc = nf; k = meta_nch(mt);
            if (k == 0) break;
        }
        const float sq = sqrtf(nvis);
        Best b{0.0f, -1};
        uint32_t bm = 0, bf = 0;                            // header, visits and state of this lane's best child
        float bv = 0.0f;
        BgState bs;
#pragma unroll
        for (int q = 0; q < 8; ++q) bs.w[q] = 0u;
        int lastnan = -1;
        const bool staged = l0 && l0->valid && depth == 0 && k <= 64 && fc == 1;
        for (uint32_t j = lane; j < k; j += 64) {
            const size_t ci = base + fc + j;
            float vis, val, pr; uint32_t cm, cf; BgState cs;
            if (staged) { vis = l0->vis; val = l0->val; pr = l0->pr; cm = l0->cm; cf = l0->cf; cs = l0->cs; }
            else {
                vis = T.visits[ci]; val = T.value[ci]; pr = T.prior[ci];
                cm = T.meta[ci]; cf = T.first_child[ci];
                cs = load_state(&T.state[ci]);              // rides the same round trip: the leaf's state needs none of its own
            }
            const float q = vis == 0.0f ? 0.0f : val / vis;
            const float t = sq / (vis + 1.0f);
            const float u = c * t;
            const float w = u * pr;
            const float s = q + w;
            if (s != s) lastnan = (int)j;
            else if (b.j < 0 || !(b.s > s)) { b.s = s; b.j = (int)j; bm = cm; bf = cf; bv = vis; bs = cs; }
        }
        lastnan = wave_allmax_i32(lastnan);
        if (lastnan >= 0) {
            // the sequential fold restarts after a NaN: only children after the last NaN compete
            b.s = 0.0f; b.j = -1;
            for (uint32_t j = lane; j < k; j += 64) {
                if ((int)j <= lastnan) continue;
                const size_t ci = base + fc + j;
                const float vis = T.visits[ci], val = T.value[ci], pr = T.prior[ci];
                const uint32_t cm = T.meta[ci], cf = T.first_child[ci];
                const float q = vis == 0.0f ? 0.0f : val / vis;
                const float t = sq / (vis + 1.0f);
                const float u = c * t;
                const float w = u * pr;
                const float s = q + w;
                if (b.j < 0 || !(b.s > s)) { b.s = s; b.j = (int)j; bm = cm; bf = cf; bv = vis; }
            }
        }
        b = wave_best(b);
        const int chosen = b.j >= 0 ? b.j : lastnan;       // lastnan == k-1 when nothing follows it
        node = fc + (uint32_t)chosen;
        ++depth;
        if ((uint32_t)lane == depth) mine = node;
        if (b.j >= 0 && lastnan < 0) {
            // the overall best is the best of the lane that scored it (ties go to the later index in both folds)
            const int owner = __builtin_amdgcn_readfirstlane(chosen & 63);
            mt = (uint32_t)__builtin_amdgcn_readlane((int)bm, owner);
            fc = (uint32_t)__builtin_amdgcn_readlane((int)bf, owner);
            nvis = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), owner));
#pragma unroll
            for (int q = 0; q < 8; ++q) cur.w[q] = (uint32_t)__builtin_amdgcn_readlane((int)bs.w[q], owner);
            cur_ok = true;
        } else {
            mt = T.meta[base + node]; fc = T.first_child[base + node]; nvis = T.visits[base + node];
            cur_ok = false;
        }
    }
    const BgState st = cur_ok ? cur : load_state(&T.state[base + node]);
    const int w = bg_winner_dev(st);
    const uint32_t plen = depth < S.path_cap ? depth + 1u : 0u;              // 0: deeper than the record holds
    if (w != 0) {                                           // alpha_mcts.rs:157-163: +-1 w.r.t. the ROOT player
        const BgState rs = load_state(&T.state[base]);
        const float v = w == st_player(rs) ? 1.0f : -1.0f;
        if (plen) backprop_path(T, base, mine, lane, plen, v);
        else if (lane == 0) backprop(T, base, node, v);
    } else if ((uint32_t)lane < plen) {
        S.path[(size_t)slot * kPathCap + lane] = mine;
    }
    cn[SC_SELECTIONS] += 1; cn[SC_DEPTH_SUM] += depth;
    if (w != 0) cn[SC_TERMINAL] += 1;
    if (lane == 0) {
        if (w != 0) {
            S.leaf_term[slot] = 1;
            if (quirks && S.sel[slot] == kNone) atomicAdd(&iflag[1], 1u);
        } else {
            S.leaf_term[slot] = 0; S.leaf[slot] = node; S.sel[slot] = node;
            S.leaf_meta[slot] = mt; S.path_len[slot] = (uint8_t)plen;
            iflag[0] = 1u;                                  // idempotent plain store (no same-address atomic storm)
            store_state(&S.eval_states[slot], st);
        }
    }
}

// the slot's counters to and from registers (SC_ILLEGAL is only ever bumped by atomics in memory: it is not written back)
__device__ __forceinline__ void load_counters(const Slots& S, uint32_t slot, uint32_t (&cn)[SC_COUNT]) {
    static_assert(SC_COUNT == 8, "two 16-byte loads");
    const uint4 a = ((const uint4*)(S.slot_cnt + slot * SC_COUNT))[0], b = ((const uint4*)(S.slot_cnt + slot * SC_COUNT))[1];
    cn[0] = a.x; cn[1] = a.y; cn[2] = a.z; cn[3] = a.w; cn[4] = b.x; cn[5] = b.y; cn[6] = b.z; cn[7] = b.w;
}
__device__ __forceinline__ void store_counters(const Slots& S, uint32_t slot, const uint32_t (&cn)[SC_COUNT]) {
    uint32_t* p = S.slot_cnt + slot * SC_COUNT;
    ((uint4*)p)[0] = make_uint4(cn[0], cn[1], cn[2], cn[3]);
    static_assert(SC_ILLEGAL == 6 && SC_NN_ROWS == 7, "word 6 stays in memory");
    p[4] = cn[4]; p[5] = cn[5]; p[7] = cn[7];
}
// ---- expansion + backpropagation ---------------------------------------------------------------
// dev builds (-DDIEE_EXPAND_STAMPS): shader-clock sums per phase of k_expand over all waves, read by scripts/expand_phases.py
#ifdef DIEE_EXPAND_STAMPS
__device__ unsigned long long g_expand_stamps[16];
#define EX_STAMP(i) do { const unsigned long long tn_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_expand_stamps[i], tn_ - tprev_); tprev_ = __builtin_readcyclecounter(); } while (0)
#define EX_STAMP_INIT unsigned long long tprev_ = __builtin_readcyclecounter()
extern "C" int diee_dev_expand_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_expand_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_expand_stamps), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#else
#define EX_STAMP(i) do {} while (0)
#define EX_STAMP_INIT do {} while (0)
#endif

struct ExpandScratch {
    WaveScratch ws;
    float raw[kMaxPlays];
    uint16_t code[kMaxPlays];
};
static_assert(sizeof(uint64_t) * 2 * kSeqCap >= sizeof(float) * 22 * 64, "the logits row fits over the dedup keys");
// k_expand<true> (the children exist already, see grow_slot): the logits row and the masked priors are all it stages
struct SettleScratch {
    float lgs[22 * 64];
    float raw[kMaxPlays];
};

// turn_policy_to_probs_tensor (utils.rs:74-84; root: utils.rs:60-72 on the Dirichlet-mixed policy,
// noise.rs:27-34) + alpha_expand_tensor (node.rs:157-174).  it == kRootIt expands the roots.
// After the expansion the same wave immediately selects this game's leaf for iteration `next_it` (kNoNext = none):
// one MCTS kernel per network evaluation instead of two.
constexpr uint32_t kNoNext = 0xFFFFFFFEu;
// PRE: the children of the leaf were created by grow_slot while the network ran (growth workgroups of the cluster launch) (states, dice, parent, action code; priors open):
// what is left for after the evaluation is what depends on it -- priors, the parent's link, backpropagation, the next descent.
// TWO (with PRE): two waves per slot.  Wave 1 creates the children (grow_slot) WHILE wave 0 loads, evaluates the value head, reduces
// the softmax and backpropagates; they meet once, wave 1 hands k and the codes over in LDS, wave 0 commits the children with their
// priors and descends.  Growth and backpropagation touch disjoint words of the tree (new nodes at [used, used + k) vs visits / value of
// the path), so the order between them does not show in any result.
struct TwoScratch {
    SettleScratch s;
    WaveScratch ws;
    uint32_t hand[1 + 2 * 64];
};
// TWO == 2 (with PRE, the children exist already): wave 1 is the COMMIT wave -- softmax over the logits row, masked priors of the children --
// while wave 0 evaluates the value head, backpropagates and descends; same meeting point, wave 1 hands the leaf's new header over in LDS and
// wave 0 stores it (the descent may be reading that header: nobody else may change it under its feet).
// The body of one (slot, iteration) of k_expand.  (k_tail, the tail of a batch, below, runs the same operations in the same order on an
// LDS copy of the tree: what one changes in the arithmetic the other must follow -- tests/test_tail_gpu.py holds both to the oracle.)
// k_tail requests its network row -- 22 logits per lane, the value features -- BEFORE it meets the other games: the round trip to where
// the tower launch left them passes while the wave would wait anyway
struct NetRowLoaded { ValueHeadIn vh; float lg[22]; };
template <bool PRE, int TWO>
using ExpandScratchOf = typename std::conditional<(TWO != 0), TwoScratch, typename std::conditional<PRE, SettleScratch, ExpandScratch>::type>::type;
template <bool PRE, int TWO = 0>
__device__ __forceinline__ void expand_body(const Tree& T, const Slots& S, const Segs& G, uint32_t n, uint32_t it, const SearchParams& P,
                                            uint32_t next_it, float c, uint32_t slot, ExpandScratchOf<PRE, TWO>& sc_all) {
    static_assert(PRE || !TWO, "two waves: the second one is the growth");
    auto& sc = [&]() -> auto& { if constexpr (TWO) return sc_all.s; else return sc_all; }();
    if (slot >= n) return;
    if constexpr (TWO == 1) {
        if (threadIdx.x >= 64) {
            grow_slot<false>(T, S, G, n, it, slot, sc_all.ws, sc_all.hand);
            __syncthreads();                                // the hand-over (wave 0 waits here when it has nothing left that does not need the children)
            return;
        }
    }
    // a wave's own LDS traffic: with two waves in the block a workgroup barrier would wait for the other one
    auto meet = [&] {
        if constexpr (TWO) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    EX_STAMP_INIT;
    const int lane = threadIdx.x & 63;
    const bool main_wave = threadIdx.x < 64;                // (TWO == 2: wave 1 runs the loads below too, then only the commit)
    const size_t base = (size_t)slot * T.node_cap;
    const bool root = it == kRootIt;
    const bool quirks = P.quirks != 0;
    // Loads in rounds, every round requested as one burst ahead of the integer work: (1) what only depends on the slot,
    // (2) what depends on the slot's batch and row -- for a single batch that is not compacted, nothing.  (A wave alone on its
    // SIMD waits out every round trip, and what the previous kernel wrote comes from beyond this XCD's L2: ~1 us each.)
    const uint32_t seg = G.n == 1 ? 0u : S.seg[slot];
    const uint32_t rowq = S.slot_row ? S.slot_row[slot] : slot;     // row of this slot in the network outputs (compacted batches
                                                                    // hold only the slots whose leaf was not terminal)
    const bool lterm = !root && S.leaf_term[slot] != 0;
    const uint32_t leaf = root ? 0u : S.leaf[slot];
    const uint32_t m0q = root ? T.meta[base] : S.leaf_meta[slot];   // the leaf's meta as the selection read it
    const uint32_t first = T.used[slot];
    const uint32_t gid = S.game_id[slot], rnd = S.round[slot];
    uint32_t cn[SC_COUNT];
    load_counters(S, slot, cn);
    NodeHdr rh{T.meta[base], T.first_child[base], T.visits[base]};  // the root's header, kept current below for the descent
    const uint32_t plen = root ? 0u : S.path_len[slot];             // the recorded path of the slot's selection (= sel)
    const uint32_t pnode = root ? 0u : S.path[(size_t)slot * kPathCap + lane];
    // the leaf's state: the selection left a copy in eval_states (no leaf -> node -> state chain)
    const BgState st = load_state(root ? &T.state[base] : &S.eval_states[slot]);
    const uint32_t row = (lterm && S.slot_row) ? 0u : rowq;         // a slot without a row: its map entry is stale; nothing of row 0 is used
    // the batch ("segment") of this slot: its seed, its flags, and its first slot (the reference's index 0)
    const uint32_t* iflag = root ? nullptr : S.iter_flags + 2 * ((size_t)seg * G.iter_cap + it);
    const uint2 ifl = root ? make_uint2(1u, 0u) : *(const uint2*)iflag;      // any_selected, stale-initial count
    const uint32_t if0 = ifl.x;
    const uint32_t seg_first = G.first_slot[seg], seg_end = G.end_slot[seg];
    const unsigned long long seed = G.seed[seg];
    const float rv0 = S.root_value0[seg];                           // (written by this batch's first slot at the root iteration)
    const unsigned long long evals0 = S.counters[(size_t)seg * CNT_COUNT + CNT_NN_EVALS];
    const uint32_t node = lterm ? 0u : leaf;
    const uint32_t m0 = lterm ? 0u : m0q;
    const ValueHeadIn vh = value_head_load(S.hv + (size_t)row * 72, S.wv, lane);
    float lg[22];
    if (TWO != 2 || !main_wave) softmax_load(S.logits + (size_t)row * 1352, lane, lg);     // (TWO == 2: the logits are the commit wave's business)
    else {
#pragma unroll
        for (int q = 0; q < 22; ++q) lg[q] = 0.0f;
    }
    // grow_slot's hand-over: how many children it created for this slot (kNone: none -- nothing to expand, or no room) and
    // their action codes, four per lane (child lane + 64 q in position q)
    // the root's children, one per lane, for the first level of the descent that follows (see Level0); patched below with
    // what this kernel changes in them
    Level0 l0;
#ifndef DIEE_STAGE_L0
#define DIEE_STAGE_L0 0        // measured: +0.2 us per launch (the seven extra loads per lane cost more than the round trip they save; profiles/r03i_*): off
#endif
    l0.valid = (DIEE_STAGE_L0 == 1 || (DIEE_STAGE_L0 == 2 && TWO == 2 && main_wave)) && !root && next_it != kNoNext;
    {
        const size_t c0 = base + 1 + (l0.valid ? lane : 0);
        l0.vis = T.visits[c0]; l0.val = T.value[c0]; l0.pr = T.prior[c0]; l0.cm = T.meta[c0]; l0.cf = T.first_child[c0];
        l0.cs = load_state(&T.state[c0]);
    }
    // lane of the root child on the recorded path of this slot's selection (depth 1), -1: the path ends at the root
    const int j1 = plen >= 2 ? __builtin_amdgcn_readlane((int)pnode, 1) - 1 : -1;
    uint32_t pre_k = (PRE && TWO != 1) ? S.grow_k[slot] : kNone;
    uint2 pre_codes = (PRE && TWO != 1) ? *(const uint2*)(S.grow_code + (size_t)slot * kMaxPlays + lane * 4) : make_uint2(0u, 0u);
    const bool active = if0 != 0;                           // alpha_mcts.rs:170-172 `continue`
    if constexpr (TWO) __syncthreads();                     // wave 1 has read the selection record this wave rewrites in its descent (see grow_slot)
    float v = 0.0f;
    bool do_expand = root || !lterm, do_backprop = !root;
    float smM = 0.0f, smInv = 0.0f;
    // backpropagation of this slot's selection (and the batch's stale initial slots, Q14)
    auto bp_section = [&] {
        if (do_backprop) {
            if (plen) { backprop_path(T, base, pnode, lane, plen, v); if (lane == j1) { l0.vis += 1.0f; l0.val += v; } }
            else { if (lane == 0) backprop(T, base, node, v); l0.valid = false; }
            rh.visits += 1.0f;
        }
        if (!root && quirks && slot == seg_first && ifl.y != 0) {
            // slots still holding the initial 0 index (alpha_mcts.rs:142) re-backpropagate node 0,
            // i.e. the root in the batch's first slot, with the NN value of its state: the same chain of additions, on registers
            const uint32_t cnt = ifl.y;
            for (uint32_t i = 0; i < cnt; ++i) rh.visits += 1.0f;
            if (lane == 0) {
                float rvis = T.visits[base], rval = T.value[base];
                for (uint32_t i = 0; i < cnt; ++i) { rvis += 1.0f; rval += rv0; }
                T.visits[base] = rvis; T.value[base] = rval;
            }
        }
    };
    if (active && main_wave) {
    // batch rows pushed through the ResNet for this batch (one writer per batch and launch)
    if (slot == seg_first && lane == 0) S.counters[(size_t)seg * CNT_COUNT + CNT_NN_EVALS] = evals0 + (seg_end - seg_first);
    if (root || S.slot_row == nullptr || !lterm) cn[SC_NN_ROWS] += 1;

    if (!root) {
        if (lterm) {
            // stale selected_nodes_idxs slot (alpha_mcts.rs:142,192-200): re-"expanded" (no-op) and
            // backpropagated with its own NN value again
            do_expand = false; do_backprop = false;
            if (quirks) {
                const uint32_t s = S.sel[slot];
                if (s != kNone) {
                    const float sv = S.sel_value[slot];
                    if (plen) { backprop_path(T, base, pnode, lane, plen, sv); if (lane == j1) { l0.vis += 1.0f; l0.val += sv; } }
                    else { if (lane == 0) backprop(T, base, s, sv); l0.valid = false; }      // (parent walk: which root child it passes is not recorded)
                    rh.visits += 1.0f;                      // the root is on every path
                }
            }
        } else {
            v = value_head_eval(vh, lane);                  // the value head's FC + tanh (nn_device.h)
            if (lane == 0) S.sel_value[slot] = v;
        }
    } else if (slot == seg_first) {
        const float v0 = value_head_eval(vh, lane);
        if (lane == 0) S.root_value0[seg] = v0;
    }
    EX_STAMP(0);                                            // flags, value head, leaf meta
    if constexpr (TWO) {
        // everything that does not need the children, while wave 1 creates (1) / commits (2) them
        if (TWO == 1 && do_expand && !(m0 & kDrained)) {
            softmax_reduce(lg, lane, smM, smInv);
#pragma unroll
            for (int q = 0; q < 22; ++q) sc.lgs[lane + 64 * q] = lg[q];
        }
        bp_section();
    }
    }
    // the commit of the children: with two waves it runs when the descent below stands on the leaf under expansion, or after it
    uint32_t new_meta = 0, new_first = 0;
    bool expanded = false, committed = false;
    auto commit = [&] {
    committed = true;
    if constexpr (TWO == 1) {
        __syncthreads();                                    // wave 1's children are in the tree, k and the codes in LDS
        pre_k = sc_all.hand[0];
        pre_codes = make_uint2(sc_all.hand[1 + 2 * lane], sc_all.hand[2 + 2 * lane]);
    }
    if constexpr (TWO == 2) {
        if (main_wave) {
            __syncthreads();                                // wave 1 has committed the children: the leaf's new header in LDS
            expanded = sc_all.hand[0] != 0u; new_meta = sc_all.hand[1]; new_first = sc_all.hand[2];
            if (expanded) {
                const uint32_t k = meta_nch(new_meta);
                if (lane == 0) {                            // the leaf's header: linked only now, by the wave that walks the tree
                    T.first_child[base + node] = new_first;
                    T.meta[base + node] = new_meta;
                    T.used[slot] = new_first + k;
                }
                meet();
                if (node == 0) { rh.meta = new_meta; rh.first_child = new_first; }
                cn[SC_EXPANSIONS] += 1; cn[SC_CHILDREN] += k;
                if (k > cn[SC_MAX_CHILDREN]) cn[SC_MAX_CHILDREN] = k;
            }
            return;
        }
    }
    if (active) {
    if constexpr (PRE) {
    if (do_expand && !(m0 & kDrained) && pre_k != kNone) {
        const int k = (int)pre_k;
        if constexpr (TWO != 1) {
            softmax_reduce(lg, lane, smM, smInv);            // softmax over this board's 1352 logits (nn_device.h)
#pragma unroll
            for (int q = 0; q < 22; ++q) sc.lgs[lane + 64 * q] = lg[q];
            meet();
        }
        EX_STAMP(2);                                        // softmax constants
        const float om = 1.0f - P.dir_eps;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q;
            if (j < k) {
                const uint32_t code = ((q & 2) ? pre_codes.y : pre_codes.x) >> (16 * (q & 1)) & 0xFFFFu;
                float p = softmax_prob(sc.lgs[code], smM, smInv);
                if (root) {                                  // apply_dirichlet: (1-eps)*P + eps*noise
                    const float x = om * p, y = P.dir_eps * S.noise[(size_t)seg * 1352 + code];
                    p = x + y;
                }
                sc.raw[j] = p;
            }
        }
        meet();
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[r] = lane + 64 * r < k ? sc.raw[lane + 64 * r] : 0.0f;
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kr = k - 64 * r < 64 ? k - 64 * r : 64;                    // uniform
            for (int j = 0; j < kr; ++j) sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr[r]), j));
        }
        EX_STAMP(3);                                        // priors + ordered row sum
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (lane + 64 * r < k) T.prior[base + first + lane + 64 * r] = pr[r] / sum;
        const uint32_t nmeta = kDrained | ((uint32_t)k << 16) | (m0 & kMetaKeep);
        // (TWO == 2: this is the commit wave; the main wave may be reading this very header in its descent, so the header is
        // the main wave's to store, after the meeting point, when the priors above are visible to it)
        if (lane == 0 && TWO != 2) {
            T.first_child[base + node] = first;
            T.meta[base + node] = nmeta;
            T.used[slot] = first + (uint32_t)k;
        }
        if (node == 0) { rh.meta = nmeta; rh.first_child = first; }
        if (plen == 2 && lane == j1) { l0.cm = nmeta; l0.cf = first; }          // the expanded leaf is a root child: its staged header
        cn[SC_EXPANSIONS] += 1; cn[SC_CHILDREN] += (uint32_t)k;
        if ((uint32_t)k > cn[SC_MAX_CHILDREN]) cn[SC_MAX_CHILDREN] = (uint32_t)k;
        new_meta = nmeta; new_first = first; expanded = true;
    }
    } else {
    if (do_expand && !(m0 & kDrained)) {
        int k = bg_legal_plays_wave(st, &sc.ws, lane, S.overflow);
        if (k > kMaxPlays) { if (lane == 0) atomicOr(S.overflow, 1u); k = 0; }     // never silent: DIEE_ERR_CAPACITY
        EX_STAMP(1);                                        // leaf state + legal plays
        const int r0 = st_roll(st, 0), r1 = st_roll(st, 1), player = st_player(st);
        float smM, smInv;                                    // softmax over this board's 1352 logits (nn_device.h)
        softmax_reduce(lg, lane, smM, smInv);
        // this row's logits for the gather by play code: they are in registers already, so no second trip to memory -- staged
        // over the dedup keys of the play enumeration, which is done with them
        float* lgs = (float*)&sc.ws.keyA[0];
#pragma unroll
        for (int q = 0; q < 22; ++q) lgs[lane + 64 * q] = lg[q];
        __syncthreads();
        EX_STAMP(2);                                        // softmax constants
        const float om = 1.0f - P.dir_eps;
        for (int j = lane; j < k; j += 64) {
            const uint32_t play = sc.ws.play[j];
            const uint32_t code = bg_encode_dev(r0, r1, play);
            float p = softmax_prob(lgs[code], smM, smInv);
            if (root) {                                      // apply_dirichlet: (1-eps)*P + eps*noise
                const float x = om * p, y = P.dir_eps * S.noise[(size_t)seg * 1352 + code];
                p = x + y;
                if (bg_decode_dev(r0, r1, player, code) != play) atomicAdd(&S.slot_cnt[slot * SC_COUNT + SC_ILLEGAL], 1u);
            }
            sc.raw[j] = p; sc.code[j] = (uint16_t)code;
        }
        __syncthreads();
        // row sum, sequential in play order (the oracle's order): every lane runs the same chain on values broadcast
        // from their lanes (k <= 256: four registers per lane) -- no LDS round trip per addend
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[r] = lane + 64 * r < k ? sc.raw[lane + 64 * r] : 0.0f;
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kr = k - 64 * r < 64 ? k - 64 * r : 64;                    // uniform
            for (int j = 0; j < kr; ++j) sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr[r]), j));
        }
        EX_STAMP(3);                                        // encode + priors + ordered row sum
        if (first + (uint32_t)k > T.node_cap) {
            if (lane == 0) atomicOr(S.overflow, 2u);
        } else {
            const uint32_t e = root ? 0u : it + 1u;
            for (int j = lane; j < k; j += 64) {
                const size_t ci = base + first + j;
                BgState cs = st;
                int d0, d1;
                draw_dice(seed, gid, rnd, e, (uint32_t)j, d0, d1);      // child dice frozen at creation (Q9)
                bg_apply_dev(cs, sc.ws.play[j], d0, d1);
                store_state(&T.state[ci], cs);
                T.visits[ci] = 0.0f; T.value[ci] = 0.0f; T.prior[ci] = sc.raw[j] / sum;
                T.parent[ci] = node; T.first_child[ci] = 0; T.meta[ci] = (uint32_t)sc.code[j] | meta_terminal_bits(cs);
            }
            const uint32_t nmeta = kDrained | ((uint32_t)k << 16) | (m0 & kMetaKeep);
            if (lane == 0) {
                T.first_child[base + node] = first;
                T.meta[base + node] = nmeta;
                T.used[slot] = first + (uint32_t)k;
            }
            if (node == 0) { rh.meta = nmeta; rh.first_child = first; }
        if (plen == 2 && lane == j1) { l0.cm = nmeta; l0.cf = first; }          // the expanded leaf is a root child: its staged header
            cn[SC_EXPANSIONS] += 1; cn[SC_CHILDREN] += (uint32_t)k;
            if ((uint32_t)k > cn[SC_MAX_CHILDREN]) cn[SC_MAX_CHILDREN] = (uint32_t)k;
        }
    }
    }   // !PRE
    EX_STAMP(4);                                            // child creation (stores issued)
    meet();
    EX_STAMP(5);                                            // ... stores acknowledged
    if constexpr (!TWO) bp_section();
    }   // active
    };  // commit
    if constexpr (!TWO) commit();
    if constexpr (TWO == 2) {
        if (!main_wave) {                                   // the commit wave: nothing else, then the hand-over
            commit();
            if (lane == 0) { sc_all.hand[0] = expanded ? 1u : 0u; sc_all.hand[1] = new_meta; sc_all.hand[2] = new_first; }
            __syncthreads();
            return;
        }
    }
    EX_STAMP(6);                                            // backpropagation
    if (next_it != kNoNext) {
        meet();                                             // lane 0's tree updates are visible to the whole wave
        EX_STAMP(7);
        if constexpr (TWO) {
            // (a staged first level holds the leaf's OLD header when the leaf is a root child: the descent then stands on a node
            // without children there and commits, like everywhere else)
            select_slot(T, S, G, slot, seg, lane, next_it, c, P.quirks, cn, &rh, &l0, [&](uint32_t nd, uint32_t& nm, uint32_t& nf) {
                if (committed || nd != node) return false;  // `node`: the leaf this launch expands (0 for a terminal selection: the root has children by then)
                commit();
                if (!expanded) return false;
                meet();                                     // lane 0's header stores before the wave reads on
                nm = new_meta; nf = new_first;
                return true;
            });
        } else {
            select_slot(T, S, G, slot, seg, lane, next_it, c, P.quirks, cn, &rh, &l0);
        }
        EX_STAMP(8);                                        // descent + leaf state + flags
    }
    if constexpr (TWO) if (!committed) commit();
    if (lane == 0) store_counters(S, slot, cn);
#ifdef DIEE_EXPAND_STAMPS
    if (threadIdx.x == 0) atomicAdd(&g_expand_stamps[15], 1ull);
#endif
}

template <bool PRE, int TWO = 0>
__global__ __launch_bounds__(TWO ? 128 : 64) void k_expand(Tree T, Slots S, Segs G, uint32_t n, uint32_t it, SearchParams P,
                                                           uint32_t next_it, float c) {
    __shared__ ExpandScratchOf<PRE, TWO> sc_all;
    expand_body<PRE, TWO>(T, S, G, n, it, P, next_it, c, blockIdx.x, sc_all);
}

// ---- the tail of a batch: iterations in a loop, network rows from the ring (search_types.h, Tail) ---------------------------------
constexpr int kTailSpin = 1 << 20;
// the games' workgroups meet: every one has published its selection of iteration `it` (flags, selection record) and says whether its
// leaf has its evaluation; returns the number of workgroups that said no.  One word per meeting, never reused within a move-step.
__device__ __forceinline__ bool tail_meet(uint32_t* word, uint32_t n, bool hit, int lane, uint32_t* err, uint32_t& misses, bool absent = false) {
    if (n == 1) { misses = hit ? 0u : 1u; return true; }
    __threadfence();                                        // this workgroup's stores before its arrival
    uint32_t v = 0;
    if (lane == 0) {
        if (!absent) atomicAdd(word, hit ? 1u : 0x10001u);  // (absent: the test of the time-out path -- this workgroup waits like the others but never arrives)
        int spins = 0;
        for (;;) {
            v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v & 0xffffu) >= n) break;
            if (++spins > kTailSpin) { atomicOr(err, 4u); v = 0xffffffffu; break; }      // (reported like a starved cluster hand-over)
            __builtin_amdgcn_s_sleep(2);
        }
    }
    v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the others' stores before this workgroup's loads
    misses = v >> 16;
    return v != 0xffffffffu;
}

// development builds (-DDIEE_TAIL_STAMPS): shader-clock sums per phase of k_tail, read by tests/tools/tail_phases.py
#ifdef DIEE_TAIL_STAMPS
__device__ unsigned long long g_tail_stamps[8];     // 0 take-in, 1 meeting (wait for the other games), 2 iteration body, 3 plan (virtual descents + rows), 6 iterations, 7 launches
#define TL_STAMP(i) do { const unsigned long long tn_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_tail_stamps[i], tn_ - tl_prev_); tl_prev_ = __builtin_readcyclecounter(); } while (0)
#define TL_COUNT(i) do { if (threadIdx.x == 0) atomicAdd(&g_tail_stamps[i], 1ull); } while (0)
extern "C" int diee_dev_tail_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tail_stamps), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#else
#define TL_STAMP(i) do {} while (0)
#define TL_COUNT(i) do {} while (0)
#endif
struct TailArgs { Tail L; uint32_t q; };

// ---- k_tail: the iterations of a search in a loop, the tree's statistics in LDS ("UCB selection staged in LDS") ---------------------
// The first version of this kernel ran expand_body, k_expand's code, once per iteration: every level of the descent, the
// backpropagation and a dozen words of per-slot state were round trips to the L2 (~1 us each for a wave alone on its CU: 60 % of the
// ~11 us of an iteration, profiles/r05m_tail_body_phases.txt; 111.4 -> 112.5 games/s with this one on the same box, profiles/r05o_*).
// k_tail keeps what an iteration reads again in the CU: the statistics and headers of the
// game's nodes (visits, value, prior, header words, the ring row and value of evaluated nodes: 28 bytes per node, kTailLdsNodes
// nodes) in LDS -- loaded when a launch starts, every update also written through to HBM (nodes past the capacity are read and
// written there) --, and the selection record, the path of the last selection, the counters and the arena's fill mark in registers
// from one iteration to the next.  A descent is LDS reads; the leaf's 32-byte state and its ring row are requested BEFORE the games
// meet.  The arithmetic is select_slot's / expand_body's, operation for operation, in the same order per node.
struct TailTree {
    float vis[kTailLdsNodes], val[kTailLdsNodes], pri[kTailLdsNodes], cval[kTailLdsNodes];
    uint32_t meta[kTailLdsNodes], fc[kTailLdsNodes], crow[kTailLdsNodes];
};
struct TailLds {
    TailTree t;
    float vvis[kTailLdsNodes], vval[kTailLdsNodes];          // the virtual descents' scratch copy of visits / value
    WaveScratch ws;
    float lgs[22 * 64];
    float raw[kMaxPlays];
    uint16_t code[kMaxPlays];
    uint32_t cand[64];
};
static_assert(sizeof(TailLds) <= 160 * 1024, "one workgroup per CU");

// the tree through LDS: nodes below kTailLdsNodes live there (and in HBM, written through), the others in HBM alone
struct TreeLds {
    const Tree& T; size_t base; TailTree& t;
    __device__ __forceinline__ float vis(uint32_t i) const { return i < kTailLdsNodes ? t.vis[i] : T.visits[base + i]; }
    __device__ __forceinline__ float val(uint32_t i) const { return i < kTailLdsNodes ? t.val[i] : T.value[base + i]; }
    __device__ __forceinline__ float pri(uint32_t i) const { return i < kTailLdsNodes ? t.pri[i] : T.prior[base + i]; }
    __device__ __forceinline__ uint32_t meta(uint32_t i) const { return i < kTailLdsNodes ? t.meta[i] : T.meta[base + i]; }
    __device__ __forceinline__ uint32_t fc(uint32_t i) const { return i < kTailLdsNodes ? t.fc[i] : T.first_child[base + i]; }
    __device__ __forceinline__ void add(uint32_t i, float v) const {          // visits += 1, value += v (simple_mcts.rs:96-103, one node)
        const float nv = vis(i) + 1.0f, nw = val(i) + v;
        if (i < kTailLdsNodes) { t.vis[i] = nv; t.val[i] = nw; }
        T.visits[base + i] = nv; T.value[base + i] = nw;
    }
    __device__ __forceinline__ void set_header(uint32_t i, uint32_t m, uint32_t f) const {
        if (i < kTailLdsNodes) { t.meta[i] = m; t.fc[i] = f; }
        T.meta[base + i] = m; T.first_child[base + i] = f;
    }
};

__global__ __launch_bounds__(64) void k_tail(Tree T, Slots S, Segs G, uint32_t n, SearchParams P, float c, TailArgs A) {
    extern __shared__ __attribute__((aligned(16))) char tail_smem[];
    uint32_t slot = blockIdx.x;
    if (gridDim.x == 8u * n) { if ((blockIdx.x & 7u) != 0u) return; slot = blockIdx.x >> 3; }
    if (slot >= n) return;
    const Tail& L = A.L;
    const int lane = threadIdx.x;
    if (L.state[1] != 0u) return;
#ifdef DIEE_TAIL_STAMPS
    unsigned long long tl_prev_ = __builtin_readcyclecounter();
#endif
    TL_COUNT(7);
    uint32_t it = L.state[0];
    TailLds& D = *reinterpret_cast<TailLds*>(tail_smem);
    const size_t base = (size_t)slot * T.node_cap;
    uint32_t* crow_g = L.crow + (size_t)slot * T.node_cap;
    float* cval_g = L.cval + (size_t)slot * T.node_cap;
    const TreeLds X{T, base, D.t};
    const bool quirks = P.quirks != 0;
    // ---- what the slot carries from launch to launch (the record select_slot / k_tail leave in HBM), into registers ----
    const uint32_t seg = G.n == 1 ? 0u : S.seg[slot];
    const uint32_t seg_first = G.first_slot[seg], seg_end = G.end_slot[seg];
    const unsigned long long seed = G.seed[seg];
    const uint32_t gid = S.game_id[slot], rnd = S.round[slot];
    const float rv0 = S.root_value0[seg];
    unsigned long long evals = S.counters[(size_t)seg * CNT_COUNT + CNT_NN_EVALS];
    uint32_t used = T.used[slot];
    bool lterm = S.leaf_term[slot] != 0;
    uint32_t leaf = S.leaf[slot], sel = S.sel[slot], leaf_meta = S.leaf_meta[slot];
    float sel_value = S.sel_value[slot];
    uint32_t plen = S.path_len[slot];
    uint32_t pnode = S.path[(size_t)slot * kPathCap + lane];     // lane d: the node at depth d of the last real selection's path
    uint32_t cn[SC_COUNT];
    load_counters(S, slot, cn);
    const BgState rs = load_state(&T.state[base]);
    const int root_player = st_player(rs);
    // every game of the batch held a real selection when the previous launch ended (it never loses it): no slot can still be counted
    // as a stale INITIAL one (Q14), so the batch's first slot need not look at an iteration's flag words for that
    bool all_sel = false;
    if (quirks && slot == seg_first) {
        const uint32_t g = seg_first + (uint32_t)lane;
        all_sel = __ballot(g < seg_end && S.sel[g] == kNone) == 0ull && seg_end - seg_first <= 64u;
    }
    // ---- the tree's statistics into LDS; the rows of tower launch q - 1 on top ----
    const uint32_t nl = used < kTailLdsNodes ? used : kTailLdsNodes;
    for (uint32_t i = lane; i < nl; i += 64) {
        D.t.vis[i] = T.visits[base + i]; D.t.val[i] = T.value[base + i]; D.t.pri[i] = T.prior[base + i];
        D.t.meta[i] = T.meta[base + i]; D.t.fc[i] = T.first_child[base + i];
        D.t.crow[i] = crow_g[i]; D.t.cval[i] = cval_g[i];
    }
    __syncthreads();
    // children the plan of launch q - 1 created ahead for this game's demanded leaf (child_rows) sit at [used, ahead_hi): evaluated before their
    // parent is expanded -- which the first iteration below does, creating exactly these nodes; it must not clear their rows then
    uint32_t ahead_hi = used;
    if (A.q > 0) {
        if (L.child_rows > 0) {
            for (uint32_t i = used + (uint32_t)lane; i < used + (uint32_t)kMaxPlays && i < kTailLdsNodes; i += 64) { D.t.crow[i] = 0; D.t.cval[i] = 0.0f; }
            __syncthreads();
        }
        const uint32_t pq = A.q - 1, nr = L.n_rows[pq] < L.rows ? L.n_rows[pq] : L.rows;
        for (uint32_t r = lane; r < nr; r += 64) {
            const uint32_t rn = L.rows_node[pq * L.rows + r];
            if ((rn >> 24) == slot) {
                const float* h = L.hv + (size_t)(pq * L.rows + r) * 72;
                float dot = 0.0f;
                for (int i = 0; i < 72; ++i) dot += h[i] * S.wv[i];
                const uint32_t node = rn & 0xFFFFFFu;
                const float cv = tanhf(dot + S.wv[72]);
                cval_g[node] = cv; crow_g[node] = pq * L.rows + r + 1u;
                if (node < kTailLdsNodes) { D.t.cval[node] = cv; D.t.crow[node] = pq * L.rows + r + 1u; }
                if (node >= ahead_hi) ahead_hi = node + 1u;
            }
        }
        ahead_hi = (uint32_t)wave_allmax_i32((int)ahead_hi);
        __syncthreads();
    }
    TL_STAMP(0);
    BgState lst = load_state(&S.eval_states[slot]);             // the selected leaf's state
    bool hit = false;
    uint32_t misses = 0;
    for (;;) {
        const uint32_t cr = lterm ? 0u : (leaf < kTailLdsNodes ? D.t.crow[leaf] : crow_g[leaf]);
        hit = lterm || cr != 0u;
        const uint32_t ring = (lterm || cr == 0u) ? 0u : cr - 1u;
        NetRowLoaded pre;                                       // requested now, used behind the meeting
        pre.vh = value_head_load(L.hv + (size_t)ring * 72, S.wv, lane);
        softmax_load(L.logits + (size_t)ring * 1352, lane, pre.lg);
        if (!tail_meet(L.bar + it + A.q, n, hit, lane, S.overflow, misses, L.test_skip != 0u && slot == n - 1u && it + A.q + 1u == L.test_skip)) {
            if (lane == 0) { atomicMax(&L.state[1], 2u); L.host[1] = 2u; __threadfence_system(); }
            return;
        }
        TL_STAMP(1);
        if (misses != 0u) break;
        // ================= iteration `it` (expand_body<false, 0>'s operations, in its order) =================
        uint32_t* iflag = S.iter_flags + 2 * ((size_t)seg * G.iter_cap + it);
        // the flag words matter to a game whose leaf is a finished game (`continue`, stale slot) and to the batch's first slot
        uint2 ifl = make_uint2(1u, 0u);
        if (lterm || (quirks && slot == seg_first && !all_sel)) ifl = *(const uint2*)iflag;
        const bool active = ifl.x != 0u;
        float v = 0.0f;
        bool do_expand = !lterm, do_backprop = true;
        if (active) {
            if (slot == seg_first) evals += (unsigned long long)(seg_end - seg_first);
            if (lterm) {
                do_expand = false; do_backprop = false;
                if (quirks && sel != kNone) {                   // the stale slot is backpropagated with its own value again (Q14)
                    if (plen) { if ((uint32_t)lane < plen) X.add(pnode, sel_value); }
                    else if (lane == 0) { for (uint32_t i = sel; i != kNone; i = T.parent[base + i]) X.add(i, sel_value); }
                }
            } else {
                v = value_head_eval(pre.vh, lane);
                sel_value = v;
            }
            const uint32_t m0 = lterm ? 0u : leaf_meta;
            if (do_expand && !(m0 & kDrained)) {
                int k = bg_legal_plays_wave(lst, &D.ws, lane, S.overflow);
                if (k > kMaxPlays) { if (lane == 0) atomicOr(S.overflow, 1u); k = 0; }
                const int r0 = st_roll(lst, 0), r1 = st_roll(lst, 1);
                float smM, smInv;
                softmax_reduce(pre.lg, lane, smM, smInv);
#pragma unroll
                for (int q = 0; q < 22; ++q) D.lgs[lane + 64 * q] = pre.lg[q];
                __syncthreads();
                for (int j = lane; j < k; j += 64) {
                    const uint32_t code = bg_encode_dev(r0, r1, D.ws.play[j]);
                    D.raw[j] = softmax_prob(D.lgs[code], smM, smInv); D.code[j] = (uint16_t)code;
                }
                __syncthreads();
                float pr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[r] = lane + 64 * r < k ? D.raw[lane + 64 * r] : 0.0f;
                float sum = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kr = k - 64 * r < 64 ? k - 64 * r : 64;                    // uniform
                    for (int j = 0; j < kr; ++j) sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr[r]), j));
                }
                const uint32_t first = used;
                if (first + (uint32_t)k > T.node_cap) {
                    if (lane == 0) atomicOr(S.overflow, 2u);
                } else {
                    const uint32_t e = it + 1u;
                    for (int j = lane; j < k; j += 64) {
                        const uint32_t cl = first + (uint32_t)j;
                        const size_t ci = base + cl;
                        BgState cs = lst;
                        int d0, d1;
                        draw_dice(seed, gid, rnd, e, (uint32_t)j, d0, d1);      // child dice frozen at creation (Q9)
                        bg_apply_dev(cs, D.ws.play[j], d0, d1);
                        store_state(&T.state[ci], cs);
                        const float prj = D.raw[j] / sum;
                        const uint32_t cm = (uint32_t)D.code[j] | meta_terminal_bits(cs);
                        T.visits[ci] = 0.0f; T.value[ci] = 0.0f; T.prior[ci] = prj;
                        T.parent[ci] = leaf; T.first_child[ci] = 0; T.meta[ci] = cm;
                        if (cl < kTailLdsNodes) {
                            D.t.vis[cl] = 0.0f; D.t.val[cl] = 0.0f; D.t.pri[cl] = prj; D.t.meta[cl] = cm; D.t.fc[cl] = 0;
                            if (cl >= ahead_hi) { D.t.crow[cl] = 0; D.t.cval[cl] = 0.0f; }       // (below: evaluated ahead, or cleared at the take-in)
                        }
                    }
                    const uint32_t nmeta = kDrained | ((uint32_t)k << 16) | (m0 & kMetaKeep);
                    if (lane == 0) X.set_header(leaf, nmeta, first);
                    used = first + (uint32_t)k;
                    cn[SC_EXPANSIONS] += 1; cn[SC_CHILDREN] += (uint32_t)k;
                    if ((uint32_t)k > cn[SC_MAX_CHILDREN]) cn[SC_MAX_CHILDREN] = (uint32_t)k;
                }
                __syncthreads();
            }
            if (do_backprop) {
                if (plen) { if ((uint32_t)lane < plen) X.add(pnode, v); }
                else if (lane == 0) { for (uint32_t i = leaf; i != kNone; i = T.parent[base + i]) X.add(i, v); }
            }
            if (quirks && slot == seg_first && ifl.y != 0u && lane == 0) {
                // slots still holding the initial 0 index re-backpropagate node 0 = this root with the NN value of its state (Q14)
                float rvis = X.vis(0), rval = X.val(0);
                for (uint32_t i = 0; i < ifl.y; ++i) { rvis += 1.0f; rval += rv0; }
                D.t.vis[0] = rvis; D.t.val[0] = rval; T.visits[base] = rvis; T.value[base] = rval;
            }
            __syncthreads();
        }
        // ================= selection for iteration it + 1 (select_slot on the LDS copy) =================
        if (it + 1 < L.iterations) {
            uint32_t* nflag = S.iter_flags + 2 * ((size_t)seg * G.iter_cap + it + 1);
            uint32_t node = 0, depth = 0, mine = lane == 0 ? 0u : kNone;
            uint32_t mt = X.meta(0), fcn = X.fc(0);
            float nvis = X.vis(0);
            for (;;) {
                const uint32_t k = meta_nch(mt);
                if (k == 0) break;
                const float sq = sqrtf(nvis);
                Best b{0.0f, -1};
                int lastnan = -1;
                for (uint32_t j = lane; j < k; j += 64) {
                    const uint32_t ci = fcn + j;
                    const float vis = X.vis(ci), val = X.val(ci), pr = X.pri(ci);
                    const float q = vis == 0.0f ? 0.0f : val / vis;
                    const float t = sq / (vis + 1.0f);
                    const float u = c * t;
                    const float w = u * pr;
                    const float sc_ = q + w;
                    if (sc_ != sc_) lastnan = (int)j;
                    else if (b.j < 0 || !(b.s > sc_)) { b.s = sc_; b.j = (int)j; }
                }
                lastnan = wave_allmax_i32(lastnan);
                if (lastnan >= 0) {                             // the sequential fold restarts after a NaN: only children after the last NaN compete
                    b.s = 0.0f; b.j = -1;
                    for (uint32_t j = lane; j < k; j += 64) {
                        if ((int)j <= lastnan) continue;
                        const uint32_t ci = fcn + j;
                        const float vis = X.vis(ci), val = X.val(ci), pr = X.pri(ci);
                        const float q = vis == 0.0f ? 0.0f : val / vis;
                        const float t = sq / (vis + 1.0f);
                        const float u = c * t;
                        const float w = u * pr;
                        const float sc_ = q + w;
                        if (b.j < 0 || !(b.s > sc_)) { b.s = sc_; b.j = (int)j; }
                    }
                }
                b = wave_best(b);
                const int chosen = b.j >= 0 ? b.j : lastnan;
                node = fcn + (uint32_t)chosen;
                ++depth;
                if ((uint32_t)lane == depth) mine = node;
                mt = X.meta(node); fcn = X.fc(node); nvis = X.vis(node);
            }
            const uint32_t npl = depth < S.path_cap ? depth + 1u : 0u;
            cn[SC_SELECTIONS] += 1; cn[SC_DEPTH_SUM] += depth;
            if (mt & kMetaTerminal) {                           // a finished game: +-1 for the ROOT's player (alpha_mcts.rs:157-163)
                const float tv = ((mt & kMetaWinnerPlus) ? 1 : -1) == root_player ? 1.0f : -1.0f;
                if (npl) { if ((uint32_t)lane < npl) X.add(mine, tv); }
                else if (lane == 0) { for (uint32_t i = node; i != kNone; i = T.parent[base + i]) X.add(i, tv); }
                cn[SC_TERMINAL] += 1;
                lterm = true;
                if (lane == 0 && quirks && sel == kNone) atomicAdd(&nflag[1], 1u);
            } else {
                lterm = false; leaf = node; sel = node; leaf_meta = mt; plen = npl; pnode = mine;
                if (lane == 0) nflag[0] = 1u;
                lst = load_state(&T.state[base + node]);        // in flight while the games meet
            }
        }
        ++it;
        __syncthreads();
        TL_STAMP(2); TL_COUNT(6);
        if (it >= L.iterations) break;
    }
    // ---- the record for the next launch (and for whoever reads the slot after the search) ----
    if (lane == 0) {
        S.leaf_term[slot] = lterm ? 1 : 0; S.leaf[slot] = leaf; S.sel[slot] = sel; S.leaf_meta[slot] = leaf_meta;
        S.sel_value[slot] = sel_value; S.path_len[slot] = (uint8_t)plen;
        T.used[slot] = used;
        if (slot == seg_first) S.counters[(size_t)seg * CNT_COUNT + CNT_NN_EVALS] = evals;
        store_state(&S.eval_states[slot], lst);
    }
    if ((uint32_t)lane < plen) S.path[(size_t)slot * kPathCap + lane] = pnode;
    const bool done = it >= L.iterations;
    uint32_t mine_rows = 0, ncand = 0;
    if (!done) {
        // ---- plan tower launch q: the leaf without an evaluation, and this game's share of speculative rows.  Virtual descents: the
        // search's own selection rule (q + c * sqrt(N) / (n + 1) * p, last of equal maxima) run ahead on a scratch copy of visits / value
        // (everything else of the LDS tree is read in place).  An evaluation that is not known yet counts as 0 (a random-init net's
        // values are small; the pricing run found the parent's mean no better), a known one as its value, a finished game as +-1 for
        // the root's player (alpha_mcts.rs:157-163); the unexpanded nodes the descents end on, not evaluated yet, are the candidates.
        // Heuristic only: which rows a launch carries beside the demanded ones never shows in a result. ----
        const uint32_t room = L.rows - (misses < L.rows ? misses : L.rows);
        const uint32_t want0 = room / n + (slot < room % n ? 1u : 0u), want = want0 < 63u ? want0 : 63u;
        const uint32_t nu = used < kTailLdsNodes ? used : kTailLdsNodes;
        // extra_rows: where the rows of a launch are scarce (fewer than 8 per game) a game looks for a few candidates beyond its share; they take
        // the rows that games with nothing worth evaluating (finished-game leaves, narrow bear-off trees) leave free -- after every game has taken
        // its share (one more meeting; the conditions are the same in every workgroup)
        const bool redistribute = L.extra_rows > 0 && L.rollout_steps > 0 && n > 1 && room > 0 && room / n < 8u;
        const bool second_round = redistribute || L.child_rows > 0;
        const uint32_t want_x = !redistribute ? want : want + L.extra_rows < 63u ? want + L.extra_rows : 63u;
        if (want_x > 0 && L.rollout_steps > 0) {
            for (uint32_t i = lane; i < nu; i += 64) { D.vvis[i] = D.t.vis[i]; D.vval[i] = D.t.val[i]; }
            __syncthreads();
            const uint32_t demanded = hit ? kNone : leaf;
            uint32_t fruitless = 0;
            for (uint32_t step = 0; step < L.rollout_steps && ncand < want_x && fruitless < 8; ++step) {
                uint32_t node = 0, depth = 0, mine = lane == 0 ? 0u : kNone, mt = 0;
                for (;;) {
                    mt = X.meta(node);
                    const uint32_t k = meta_nch(mt);
                    if (k == 0) break;
                    const uint32_t fcn = X.fc(node);
                    const float sq = sqrtf(node < nu ? D.vvis[node] : T.visits[base + node]);
                    Best b{0.0f, -1};
                    for (uint32_t j = lane; j < k; j += 64) {
                        const uint32_t ci = fcn + j;
                        const float vis = ci < nu ? D.vvis[ci] : T.visits[base + ci], val = ci < nu ? D.vval[ci] : T.value[base + ci];
                        const float q = vis == 0.0f ? 0.0f : val / vis;
                        const float sc_ = q + (c * (sq / (vis + 1.0f))) * X.pri(ci);
                        if (sc_ == sc_ && (b.j < 0 || !(b.s > sc_))) { b.s = sc_; b.j = (int)j; }
                    }
                    b = wave_best(b);
                    node = fcn + (uint32_t)(b.j >= 0 ? b.j : (int)k - 1);
                    ++depth;
                    if ((uint32_t)lane == depth) mine = node;
                    if (depth >= 63) break;
                }
                const uint32_t cr = node < kTailLdsNodes ? D.t.crow[node] : crow_g[node];
                float x = 0.0f;
                bool fresh = false;
                if (mt & kMetaTerminal) x = ((mt & kMetaWinnerPlus) ? 1 : -1) == root_player ? 1.0f : -1.0f;
                else if (cr != 0) x = node < kTailLdsNodes ? D.t.cval[node] : cval_g[node];
                else if (!(mt & kDrained) && node != demanded) fresh = __ballot((uint32_t)lane < ncand && D.cand[lane] == node) == 0ull;
                if (fresh) { if (lane == 0) D.cand[ncand] = node; ++ncand; fruitless = 0; } else ++fruitless;
                if ((uint32_t)lane <= depth && mine < nu) { D.vvis[mine] += 1.0f; D.vval[mine] += x; }
                __syncthreads();
            }
        }
        // child_rows: a game whose leaf waits for its evaluation will expand it in iteration `it`, the first of the next launch -- so the children's
        // dice (keyed by the expanding iteration) are known NOW, and the children can ride in the same launch as their parent: a search that
        // deepens (select the node just expanded's best child, expand it, select ITS best child ...) otherwise misses at every iteration.  Up to
        // child_rows of them, in play order (the priors that rank them are in the evaluation the leaf is waiting for), from the rows the shares leave free.
        uint32_t nchild = 0;
        if (L.child_rows > 0 && !hit && !(leaf_meta & kDrained)) {
            int k = bg_legal_plays_wave(lst, &D.ws, lane, S.overflow);
            if (k > kMaxPlays || used + (uint32_t)k > T.node_cap) k = 0;                              // (the expansion itself will report it)
            nchild = (uint32_t)k < L.child_rows ? (uint32_t)k : L.child_rows;
            __syncthreads();
        }
        const uint32_t share = ncand < want ? ncand : want, dem = hit ? 0u : 1u;
        mine_rows = dem + share;
        uint32_t start = 0;
        if (lane == 0 && mine_rows) start = atomicAdd(&L.n_rows[A.q], mine_rows);
        start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
        if ((uint32_t)lane < mine_rows && start + (uint32_t)lane < L.rows) {
            const uint32_t node = (dem && lane == 0) ? leaf : D.cand[(uint32_t)lane - dem];
            const uint32_t r = A.q * L.rows + start + (uint32_t)lane;
            store_state(&L.rows_state[r], (dem && lane == 0) ? lst : load_state(&T.state[base + node]));
            L.rows_node[r] = (slot << 24) | node;
        }
        uint32_t extra = 0;
        if (second_round) {
            uint32_t unused_;
            const bool met = tail_meet(L.bar2 + A.q, n, true, lane, S.overflow, unused_);       // every game's share is taken
            const uint32_t more = redistribute ? ncand - share : 0u;
            extra = met ? nchild + more : 0u;
            uint32_t start2 = 0;
            if (lane == 0 && extra) {
                start2 = atomicAdd(&L.n_rows[A.q], extra);
                const uint32_t fit = start2 >= L.rows ? 0u : extra < L.rows - start2 ? extra : L.rows - start2;
                if (fit < extra) atomicSub(&L.n_rows[A.q], extra - fit);                        // the count ends at the rows in use (<= L.rows): once it has
            }                                                                                   // reached L.rows it never falls below, so no two claims overlap
            start2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)start2);
            extra = start2 >= L.rows ? 0u : extra < L.rows - start2 ? extra : L.rows - start2;
            for (uint32_t j = (uint32_t)lane; j < extra; j += 64) {
                const uint32_t r = A.q * L.rows + start2 + j;
                if (j < nchild) {                                                               // the demanded leaf's children, as iteration `it` will create them
                    BgState cs = lst;
                    int d0, d1;
                    draw_dice(seed, gid, rnd, it + 1u, j, d0, d1);
                    bg_apply_dev(cs, D.ws.play[j], d0, d1);
                    store_state(&L.rows_state[r], cs);
                    L.rows_node[r] = (slot << 24) | (used + j);
                } else {
                    const uint32_t node = D.cand[share + (j - nchild)];
                    store_state(&L.rows_state[r], load_state(&T.state[base + node]));
                    L.rows_node[r] = (slot << 24) | node;
                }
            }
        }
        cn[SC_NN_ROWS] += mine_rows + extra;
        if (lane == 0 && share + extra) atomicAdd(&L.state[3], share + extra);
    }
    if (lane == 0) store_counters(S, slot, cn);
    TL_STAMP(3);
    if (slot == 0 && lane == 0 && !done) L.n_dem[A.q] = misses;
    if (slot == 0 && lane == 0) {
        // (a meeting that timed out in ANY workgroup left 2 in state[1]: it stays -- atomicMax, 0 < 1 < 2 -- and tail_run takes the starved path)
        const uint32_t was = atomicMax(&L.state[1], done ? 1u : 0u);
        const uint32_t now = was > (done ? 1u : 0u) ? was : (done ? 1u : 0u);
        L.state[0] = it;
        if (!done) L.state[2] += 1u;
        L.host[0] = it; L.host[1] = now; L.host[2] = A.q;
        __threadfence_system();
    }
}

// ---- k_free: the free-running search (search_types.h, Free) -------------------------------------------------------------------------
// One wave per live game, nobody waits for anybody.  Per launch a game (a) stages its tree's statistics in LDS (its share of the CU's
// 160 KB: `lds_nodes` nodes, the rest is read in place) and takes in the rows the previous launch evaluated for it, (b) runs ITS OWN
// iterations -- k_tail's body, i.e. expand_body's / select_slot's operations in their order -- for as long as its selected leaf is a
// finished game or has its evaluation in the ring and the flag words it needs are final, (c) lists its wishes for the next launch.
// What couples the games of a batch (Q14): a game whose leaf is a finished game needs `node_selected` of its iteration, which is final
// once it reads 1 or once every game of the batch has published that iteration's selection; the batch's first slot needs the count of
// stale initial slots while some game has never selected a node.  Both are read LIVE (agent-scope atomics; a game publishes its flags,
// then its progress) and polled for a bounded few microseconds; a game that still cannot tell stops for this round -- that changes
// when things are computed, never what.  The game furthest behind can always go on (everybody has published its iteration), so
// every launch pair completes at least one iteration of it: `iterations` pairs always suffice.
// polls of the flag words before a game gives up for the round.  A game that runs ahead of the others through finished-game leaves (they need
// no evaluation) reaches iterations nobody else has published: the others are waiting for their rows, so polling there only prolongs the
// launch (48 polls of a 600-game progress scan cost ~150 us per launch on the first build: profiles/r06b_*); it looks once while many games
// share the launch and polls only where the few games of a batch's end run side by side
constexpr int kFreeSpinFew = 48, kFreeFewGames = 64;
constexpr uint32_t kFreeRowBits = 12;                        // crow = ((launch << kFreeRowBits) | row) + 1
struct FreeArgs { Free F; uint32_t q; };
// development builds (-DDIEE_TAIL_STAMPS): shader-clock sums per phase of k_free over all workgroups, read by tests/tools/free_phases.py
#ifdef DIEE_TAIL_STAMPS
__device__ unsigned long long g_free_stamps[16];    // 0 record + tree into LDS + take-in, 1 flag words, 2 iteration body, 3 selection, 4 record out + plan; 8 workgroups, 9 iterations, 10 sum of the
                                                    // launches' busiest workgroup is not known here: 10 = the longest single workgroup, 11 = virtual descents
#define FR_STAMP(i) do { const unsigned long long tn_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_free_stamps[i], tn_ - fr_prev_); fr_prev_ = __builtin_readcyclecounter(); } while (0)
#define FR_COUNT(i, k) do { if (threadIdx.x == 0) atomicAdd(&g_free_stamps[i], (unsigned long long)(k)); } while (0)
extern "C" int diee_dev_free_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_free_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_free_stamps), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#else
#define FR_STAMP(i) do {} while (0)
#define FR_COUNT(i, k) do {} while (0)
#endif
// a node in LDS: visits, value, prior, one header word (expanded / finished-game bits, children, first child), the ring row of its evaluation, and
// its value (what a virtual descent that ends on the node adds), the virtual descents' copy of visits / value -- 32 bytes (the action code of
// Tree::meta stays in HBM: read where needed)
constexpr uint32_t kFreeLdsNodeBytes = 32;
__host__ __device__ __forceinline__ constexpr uint32_t free_hdr(uint32_t meta, uint32_t first_child) {
    return (meta & 0xE0000000u) | (((meta >> 16) & 0x1ffu) << 20) | (first_child & 0xFFFFFu);
}
__host__ __device__ __forceinline__ constexpr uint32_t hdr_nch(uint32_t h) { return (h >> 20) & 0x1ffu; }
__host__ __device__ __forceinline__ constexpr uint32_t hdr_fc(uint32_t h) { return h & 0xFFFFFu; }
__host__ __device__ constexpr size_t free_lds_bytes(uint32_t ln) {
    return (size_t)ln * kFreeLdsNodeBytes + sizeof(WaveScratch) + sizeof(float) * kMaxPlays + sizeof(uint16_t) * kMaxPlays + sizeof(uint32_t) * 64u;
}
// the tree through LDS with a run-time capacity: nodes below `ln` live there (and in HBM, written through), the others in HBM alone
struct TreeF {
    const Tree& T; size_t base; uint32_t ln;
    float *lvis, *lval, *lpri; uint32_t* lhdr;
    __device__ __forceinline__ float vis(uint32_t i) const { return i < ln ? lvis[i] : T.visits[base + i]; }
    __device__ __forceinline__ float val(uint32_t i) const { return i < ln ? lval[i] : T.value[base + i]; }
    __device__ __forceinline__ float pri(uint32_t i) const { return i < ln ? lpri[i] : T.prior[base + i]; }
    __device__ __forceinline__ uint32_t hdr(uint32_t i) const { return i < ln ? lhdr[i] : free_hdr(T.meta[base + i], T.first_child[base + i]); }
    __device__ __forceinline__ void add(uint32_t i, float v) const {          // visits += 1, value += v (simple_mcts.rs:96-103, one node)
        const float nv = vis(i) + 1.0f, nw = val(i) + v;
        if (i < ln) { lvis[i] = nv; lval[i] = nw; }
        T.visits[base + i] = nv; T.value[base + i] = nw;
    }
    __device__ __forceinline__ void set_header(uint32_t i, uint32_t m, uint32_t f) const {
        if (i < ln) lhdr[i] = free_hdr(m, f);
        T.meta[base + i] = m; T.first_child[base + i] = f;
    }
};
__device__ __forceinline__ uint32_t free_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(64) void k_free(Tree T, Slots S, Segs G, uint32_t n, SearchParams P, float c, FreeArgs A) {
    extern __shared__ __attribute__((aligned(16))) char free_smem[];
    const uint32_t slot = blockIdx.x;
    if (slot >= n) return;
    const Free& F = A.F;
    const int lane = threadIdx.x;
    if (F.state[0] != 0u) return;                               // every game was done when the previous round was packed
    uint32_t it = F.prog[slot];
    if (it >= F.iterations) { if (lane == 0) F.wish_n[slot] = it << 8; return; }
    // The games still searching when the last round was packed.  A search ends when its LAST game is done, and the last few games have the
    // launches to themselves (traced: 11 of 37 rounds at 300 games carried the rows of one to four games, profiles/r06v_*): with few games
    // left a game runs as many iterations as its rows allow and lists as many candidates as the launch has room for
    const uint32_t active0 = A.q == 0 ? n : F.state[4];
    const uint32_t active = active0 ? active0 : 1u;
    const bool few = active * 8u <= n || active <= 8u;
    const uint32_t iter_cap = few ? 64u * F.iter_cap : F.iter_cap;
#ifdef DIEE_TAIL_STAMPS
    unsigned long long fr_prev_ = __builtin_readcyclecounter();
    const unsigned long long fr_t0_ = fr_prev_;
#endif
    FR_COUNT(8, 1);
    const uint32_t ln = F.lds_nodes;
    float* lvis = (float*)free_smem; float* lval = lvis + ln; float* lpri = lval + ln;
    float* vvis = lpri + ln; float* vval = vvis + ln;
    float* lcval = vval + ln;
    uint32_t* lhdr = (uint32_t*)(lcval + ln); uint32_t* lcrow = lhdr + ln;
    WaveScratch& ws = *reinterpret_cast<WaveScratch*>(lcrow + ln);
    float* raw = reinterpret_cast<float*>(&ws + 1);
    uint16_t* code = reinterpret_cast<uint16_t*>(raw + kMaxPlays);
    uint32_t* cand = reinterpret_cast<uint32_t*>(code + kMaxPlays);
    float* lgs = (float*)&ws.keyA[0];                           // the logits row, staged over the dedup keys once the plays are enumerated
    const size_t base = (size_t)slot * T.node_cap;
    uint32_t* crow_g = F.crow + base;
    float* cval_g = F.cval + base;
    const TreeF X{T, base, ln, lvis, lval, lpri, lhdr};
    const bool quirks = P.quirks != 0;
    // ---- the record the game carries from launch to launch ----
    const uint32_t seg = G.n == 1 ? 0u : S.seg[slot];
    const uint32_t seg_first = G.first_slot[seg], seg_end = G.end_slot[seg];
    const unsigned long long seed = G.seed[seg];
    const uint32_t gid = S.game_id[slot], rnd = S.round[slot];
    const float rv0 = S.root_value0[seg];
    unsigned long long evals = S.counters[(size_t)seg * CNT_COUNT + CNT_NN_EVALS];
    uint32_t used = T.used[slot];
    bool lterm = S.leaf_term[slot] != 0;
    uint32_t leaf = S.leaf[slot], sel = S.sel[slot], leaf_meta = S.leaf_meta[slot];
    float sel_value = S.sel_value[slot];
    uint32_t plen = S.path_len[slot];
    uint32_t pnode = S.path[(size_t)slot * kPathCap + lane];
    uint32_t cn[SC_COUNT];
    load_counters(S, slot, cn);
    const BgState rs = load_state(&T.state[base]);
    const int root_player = st_player(rs);
    // Q14's count of stale INITIAL slots matters to the batch's first slot for iterations before every game of the batch has had a real
    // selection: `fsel` = the latest first selection in the batch as the previous launches left it (a conservative bound: later is safe)
    uint32_t fsel = 0;
    if (A.q == 0) { if (lane == 0) F.first_sel[slot] = sel == kNone ? kNone : 0u; }
    if (quirks && slot == seg_first) {
        uint32_t m = 0;
        for (uint32_t g = seg_first + (uint32_t)lane; g < seg_end; g += 64) {
            const uint32_t f = A.q == 0 ? (S.sel[g] == kNone ? kNone : 0u) : F.first_sel[g];
            m = f == kNone ? 0x7fffffffu : (f > m ? f : m);
            if (f == kNone) break;
        }
        fsel = (uint32_t)wave_allmax_i32((int)m);
    }
    // ---- the tree's statistics into LDS; the rows the previous launch evaluated for this game on top ----
    const uint32_t nl = used < ln ? used : ln;
    for (uint32_t i0 = 0; i0 < nl; i0 += 256) {                 // four rounds of loads in flight before the first store (a wave alone on its SIMD waits out every round trip)
        float a[4], b[4], p[4], cv4[4]; uint32_t m[4], f[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 64u * u + (uint32_t)lane;
            const size_t gi = base + (i < nl ? i : 0u);
            a[u] = T.visits[gi]; b[u] = T.value[gi]; p[u] = T.prior[gi]; m[u] = T.meta[gi]; f[u] = T.first_child[gi]; r[u] = crow_g[i < nl ? i : 0u]; cv4[u] = cval_g[i < nl ? i : 0u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 64u * u + (uint32_t)lane;
            if (i < nl) { lvis[i] = a[u]; lval[i] = b[u]; lpri[i] = p[u]; lhdr[i] = free_hdr(m[u], f[u]); lcrow[i] = r[u]; lcval[i] = cv4[u]; }
        }
    }
    __syncthreads();
    if (A.q > 0) {
        const uint32_t pq = A.q - 1, off = F.grant_off[slot], cnt = F.grant_cnt[slot];
        const size_t rb = (size_t)(pq % F.ring) * F.rows;
        for (uint32_t r = lane; r < cnt; r += 64) {
            const uint32_t row = off + r;
            const uint32_t node = F.rows_idx[rb + row] - (uint32_t)base;
            const float* h = F.hv + (rb + row) * 72;
            float dot = 0.0f;
            for (int i = 0; i < 72; ++i) dot += h[i] * S.wv[i];
            const float cv = tanhf(dot + S.wv[72]);
            const uint32_t id = ((pq << kFreeRowBits) | row) + 1u;
            cval_g[node] = cv; crow_g[node] = id;
            if (node < ln) { lcrow[node] = id; lcval[node] = cv; }
        }
        cn[SC_NN_ROWS] += cnt;
        __syncthreads();
    }
    // a ring row is there while no later launch has reused its place: launch lq's rows until launch lq + ring is evaluated
    auto at_hand = [&](uint32_t cr) { return cr != 0u && ((cr - 1u) >> kFreeRowBits) + F.ring >= A.q; };
    auto crow_of = [&](uint32_t i) { return i < ln ? lcrow[i] : crow_g[i]; };
    // the least progress of the batch's games, as published (own included)
    auto batch_progress = [&]() {
        int m = 0x7fffffff;
        for (uint32_t g = seg_first + (uint32_t)lane; g < seg_end; g += 64) { const int v = (int)free_load(&F.prog[g]); m = v < m ? v : m; }
        return (uint32_t)(-wave_allmax_i32(-m));
    };
    BgState lst = load_state(&S.eval_states[slot]);             // the selected leaf's state
    FR_STAMP(0);
    bool have = false, stalled = false;
    uint32_t minprog = 0;                                       // a lower bound of the batch's progress (every game has published iteration 0's selection)
    // the selected leaf's network row -- 22 logits per lane, the value features -- is requested as soon as the leaf is known (here, and
    // behind every selection below): the round trip to where the tower launch left it passes under the flag words and the loop's bookkeeping
    auto net_row = [&](uint32_t cr_) {
        NetRowLoaded p;
        const size_t ring = (size_t)(((cr_ - 1u) >> kFreeRowBits) % F.ring) * F.rows + ((cr_ - 1u) & ((1u << kFreeRowBits) - 1u));
        p.vh = value_head_load(F.hv + ring * 72, S.wv, lane);
        softmax_load(F.logits + ring * 1352, lane, p.lg);
        return p;
    };
    uint32_t cr = lterm ? 0u : crow_of(leaf);
    NetRowLoaded pre;
    if (!lterm && at_hand(cr)) pre = net_row(cr);
    uint32_t ran = 0;
    for (;; ++ran) {
        have = lterm || at_hand(cr);
        if (!have) break;
        if (ran >= iter_cap) { stalled = true; break; }       // (enough for this launch: the other games' workgroups are done long since)
        // ---- the flag words of this iteration, where the game needs them ----
        uint32_t* iflag = S.iter_flags + 2 * ((size_t)seg * G.iter_cap + it);
        const bool need_cnt = quirks && slot == seg_first && it < fsel;
        const bool need_any = lterm && (quirks || slot == seg_first);
        uint2 ifl = make_uint2(1u, 0u);
        if (need_any || need_cnt) {
            bool fin = false;
            const int spins = n <= (uint32_t)kFreeFewGames ? kFreeSpinFew : 1;
            for (int spin = 0; spin < spins; ++spin) {
                // (a 1 is final however old the cache line it is read from; a 0 may be stale: ask the memory side)
                if (!need_cnt && (iflag[0] != 0u || free_load(&iflag[0]) != 0u)) { ifl.x = 1u; fin = true; break; }
                if (minprog < it) minprog = batch_progress();
                if (minprog >= it) {                            // everybody has published this iteration's selection: the words are final
                    ifl.x = free_load(&iflag[0]); ifl.y = free_load(&iflag[1]);      // (memory-side loads, issued after the progress words' have returned)
                    fin = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!fin) { stalled = true; break; }
        }
        FR_STAMP(1);
        // ================= iteration `it` (expand_body<false, 0>'s operations, in its order) =================
        const bool active = ifl.x != 0u;
        float v = 0.0f;
        bool do_expand = !lterm, do_backprop = true;
        if (active) {
            if (slot == seg_first) evals += (unsigned long long)(seg_end - seg_first);
            if (lterm) {
                do_expand = false; do_backprop = false;
                if (quirks && sel != kNone) {                   // the stale slot is backpropagated with its own value again (Q14)
                    if (plen) { if ((uint32_t)lane < plen) X.add(pnode, sel_value); }
                    else if (lane == 0) { for (uint32_t i = sel; i != kNone; i = T.parent[base + i]) X.add(i, sel_value); }
                }
            } else {
                v = value_head_eval(pre.vh, lane);
                sel_value = v;
            }
            const uint32_t m0 = lterm ? 0u : leaf_meta;
            if (do_expand && !(m0 & kDrained)) {
                int k = bg_legal_plays_wave(lst, &ws, lane, S.overflow);
                if (k > kMaxPlays) { if (lane == 0) atomicOr(S.overflow, 1u); k = 0; }
                const int r0 = st_roll(lst, 0), r1 = st_roll(lst, 1);
                float smM, smInv;
                softmax_reduce(pre.lg, lane, smM, smInv);
#pragma unroll
                for (int q = 0; q < 22; ++q) lgs[lane + 64 * q] = pre.lg[q];
                __syncthreads();
                for (int j = lane; j < k; j += 64) {
                    const uint32_t cd = bg_encode_dev(r0, r1, ws.play[j]);
                    raw[j] = softmax_prob(lgs[cd], smM, smInv); code[j] = (uint16_t)cd;
                }
                __syncthreads();
                float pr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[r] = lane + 64 * r < k ? raw[lane + 64 * r] : 0.0f;
                float sum = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kr = k - 64 * r < 64 ? k - 64 * r : 64;                    // uniform
                    for (int j = 0; j < kr; ++j) sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr[r]), j));
                }
                const uint32_t first = used;
                if (first + (uint32_t)k > T.node_cap) {
                    if (lane == 0) atomicOr(S.overflow, 2u);
                } else {
                    const uint32_t e = it + 1u;
                    for (int j = lane; j < k; j += 64) {
                        const uint32_t cl = first + (uint32_t)j;
                        const size_t ci = base + cl;
                        BgState cs = lst;
                        int d0, d1;
                        draw_dice(seed, gid, rnd, e, (uint32_t)j, d0, d1);      // child dice frozen at creation (Q9), keyed by the GAME's iteration
                        bg_apply_dev(cs, ws.play[j], d0, d1);
                        store_state(&T.state[ci], cs);
                        const float prj = raw[j] / sum;
                        const uint32_t cm = (uint32_t)code[j] | meta_terminal_bits(cs);
                        T.visits[ci] = 0.0f; T.value[ci] = 0.0f; T.prior[ci] = prj;
                        T.parent[ci] = leaf; T.first_child[ci] = 0; T.meta[ci] = cm;
                        crow_g[cl] = 0u;                                         // (crow is zeroed per move-step; kept explicit: a node is born without an evaluation)
                        if (cl < ln) { lvis[cl] = 0.0f; lval[cl] = 0.0f; lpri[cl] = prj; lhdr[cl] = free_hdr(cm, 0u); lcrow[cl] = 0; }
                    }
                    const uint32_t nmeta = kDrained | ((uint32_t)k << 16) | (m0 & kMetaKeep);
                    if (lane == 0) X.set_header(leaf, nmeta, first);
                    used = first + (uint32_t)k;
                    cn[SC_EXPANSIONS] += 1; cn[SC_CHILDREN] += (uint32_t)k;
                    if ((uint32_t)k > cn[SC_MAX_CHILDREN]) cn[SC_MAX_CHILDREN] = (uint32_t)k;
                }
                __syncthreads();
            }
            if (do_backprop) {
                if (plen) { if ((uint32_t)lane < plen) X.add(pnode, v); }
                else if (lane == 0) { for (uint32_t i = leaf; i != kNone; i = T.parent[base + i]) X.add(i, v); }
            }
            if (quirks && slot == seg_first && ifl.y != 0u && lane == 0) {
                // slots still holding the initial 0 index re-backpropagate node 0 = this root with the NN value of its state (Q14)
                float rvis = X.vis(0), rval = X.val(0);
                for (uint32_t i = 0; i < ifl.y; ++i) { rvis += 1.0f; rval += rv0; }
                if (ln > 0) { lvis[0] = rvis; lval[0] = rval; }
                T.visits[base] = rvis; T.value[base] = rval;
            }
            __syncthreads();
        }
        FR_STAMP(2);
        // ================= selection for iteration it + 1 (select_slot on the LDS copy), published for the other games =================
        uint32_t published = 0;
        if (it + 1 < F.iterations) {
            uint32_t* nflag = S.iter_flags + 2 * ((size_t)seg * G.iter_cap + it + 1);
            uint32_t node = 0, depth = 0, mine = lane == 0 ? 0u : kNone;
            uint32_t mt = X.hdr(0), fcn = hdr_fc(mt);
            float nvis = X.vis(0);
            for (;;) {
                const uint32_t k = hdr_nch(mt);
                if (k == 0) break;
                const float sq = sqrtf(nvis);
                Best b{0.0f, -1};
                int lastnan = -1;
                for (uint32_t j = lane; j < k; j += 64) {
                    const uint32_t ci = fcn + j;
                    const float vis = X.vis(ci), val = X.val(ci), pr = X.pri(ci);
                    const float q = vis == 0.0f ? 0.0f : val / vis;
                    const float t = sq / (vis + 1.0f);
                    const float u = c * t;
                    const float w = u * pr;
                    const float sc_ = q + w;
                    if (sc_ != sc_) lastnan = (int)j;
                    else if (b.j < 0 || !(b.s > sc_)) { b.s = sc_; b.j = (int)j; }
                }
                lastnan = wave_allmax_i32(lastnan);
                if (lastnan >= 0) {                             // the sequential fold restarts after a NaN: only children after the last NaN compete
                    b.s = 0.0f; b.j = -1;
                    for (uint32_t j = lane; j < k; j += 64) {
                        if ((int)j <= lastnan) continue;
                        const uint32_t ci = fcn + j;
                        const float vis = X.vis(ci), val = X.val(ci), pr = X.pri(ci);
                        const float q = vis == 0.0f ? 0.0f : val / vis;
                        const float t = sq / (vis + 1.0f);
                        const float u = c * t;
                        const float w = u * pr;
                        const float sc_ = q + w;
                        if (b.j < 0 || !(b.s > sc_)) { b.s = sc_; b.j = (int)j; }
                    }
                }
                b = wave_best(b);
                const int chosen = b.j >= 0 ? b.j : lastnan;
                node = fcn + (uint32_t)chosen;
                ++depth;
                if ((uint32_t)lane == depth) mine = node;
                mt = X.hdr(node); fcn = hdr_fc(mt); nvis = X.vis(node);
            }
            const uint32_t npl = depth < S.path_cap ? depth + 1u : 0u;
            cn[SC_SELECTIONS] += 1; cn[SC_DEPTH_SUM] += depth;
            if (mt & kMetaTerminal) {                           // a finished game: +-1 for the ROOT's player (alpha_mcts.rs:157-163)
                const float tv = ((mt & kMetaWinnerPlus) ? 1 : -1) == root_player ? 1.0f : -1.0f;
                if (npl) { if ((uint32_t)lane < npl) X.add(mine, tv); }
                else if (lane == 0) { for (uint32_t i = node; i != kNone; i = T.parent[base + i]) X.add(i, tv); }
                cn[SC_TERMINAL] += 1;
                lterm = true;
                if (lane == 0 && quirks && sel == kNone) published = atomicAdd(&nflag[1], 1u);
            } else {
                lterm = false; leaf = node; plen = npl; pnode = mine;
                leaf_meta = T.meta[base + node];                // (the header word in LDS carries no action code: the expansion of this leaf keeps it, kMetaKeep)
                if (lane == 0) {
                    // (a 1 read from a cached line is final -- nobody clears the word within a move-step --, so only the first games to get here touch
                    // the word with an atomic: a thousand same-address atomics per iteration would queue up at ~11 ns each)
                    if (nflag[0] == 0u) published = atomicOr(&nflag[0], 1u);
                    if (sel == kNone) F.first_sel[slot] = it + 1u;
                }
                sel = node;
                lst = load_state(&T.state[base + node]);
            }
            cr = lterm ? 0u : crow_of(leaf);
            if (!lterm && at_hand(cr)) pre = net_row(cr);       // (in flight while the selection is published)
        }
        ++it;
        __syncthreads();
        // the flags before the progress: whoever reads `it` here finds this game's words of iteration `it` in place.  Both are atomics at the
        // memory side, and the progress is issued only when the flag's atomic has RETURNED (`published` is waited for): no release fence -- on this
        // chip that is a write-back of the XCD's whole L2, once per game and iteration (the games share nothing else: their trees are their own)
        if (lane == 0) {
            asm volatile("" :: "v"(published) : "memory");      // (the flag atomic's return value is in its register: the atomic has been performed)
            atomicExch(&F.prog[slot], it);
        }
        FR_STAMP(3); FR_COUNT(9, 1);
        if (it >= F.iterations) break;
    }
    // ---- the record for the next launch (and for whoever reads the slot after the search) ----
    if (lane == 0) {
        S.leaf_term[slot] = lterm ? 1 : 0; S.leaf[slot] = leaf; S.sel[slot] = sel; S.leaf_meta[slot] = leaf_meta;
        S.sel_value[slot] = sel_value; S.path_len[slot] = (uint8_t)plen;
        T.used[slot] = used;
        if (slot == seg_first) S.counters[(size_t)seg * CNT_COUNT + CNT_NN_EVALS] = evals;
        store_state(&S.eval_states[slot], lst);
    }
    if ((uint32_t)lane < plen) S.path[(size_t)slot * kPathCap + lane] = pnode;
    const bool done = it >= F.iterations;
    uint32_t nw = 0;
    const bool dem = !done && !stalled && !have;
    if (!done) {
        // ---- the wishes for the next launch: the leaf the game waits for, then the unexpanded, unevaluated nodes that virtual descents --
        // the search's own selection rule run ahead on a scratch copy of visits / value; an unknown evaluation counts as 0, a known one as
        // its value, a finished game as +-1 for the root's player -- end on, in the order they are found (= the order the search will want them) ----
        uint32_t* wl = F.wish + (size_t)slot * kFreeWish;
        if (dem) { if (lane == 0) wl[0] = leaf; nw = 1; }
        // (the launch lasts as long as its busiest game: one that ran its full share of iterations -- it is ahead of the others and the rows it was
        // granted served it well -- looks for fewer candidates: every iteration it ran takes two virtual descents off its budget)
        uint32_t want0 = F.cand_max < kFreeWish - 1u ? F.cand_max : kFreeWish - 1u;
        uint32_t steps_max = F.rollout_steps;
        if (few && want0) {                                     // the spare rows per game STILL SEARCHING, not per live game
            const uint32_t room = (F.rows > active ? F.rows - active : 0u) / active;
            want0 = room + 1u < kFreeWish - 1u ? (room + 1u > want0 ? room + 1u : want0) : kFreeWish - 1u;
            steps_max = 2u * F.rollout_steps + 8u;
        }
        const uint32_t want = few ? want0 : (want0 > ran / 2u + 1u ? want0 - ran / 2u : (want0 ? 1u : 0u));
        uint32_t ncand = 0;
        if (want > 0 && steps_max > 0) {
            const uint32_t nu = used < ln ? used : ln;
            for (uint32_t i = lane; i < nu; i += 64) { vvis[i] = lvis[i]; vval[i] = lval[i]; }
            __syncthreads();
            const uint32_t demanded = dem ? leaf : kNone;
            uint32_t fruitless = 0;
            for (uint32_t step = 0; step < steps_max && ncand < want && fruitless < 8; ++step) {
                uint32_t node = 0, depth = 0, mine = lane == 0 ? 0u : kNone, mt = 0;
                for (;;) {
                    mt = X.hdr(node);
                    const uint32_t k = hdr_nch(mt);
                    if (k == 0) break;
                    const uint32_t fcn = hdr_fc(mt);
                    // (a prediction, not the search: the hardware's approximate reciprocal and square root will do)
                    const float sq = __builtin_amdgcn_sqrtf(node < nu ? vvis[node] : T.visits[base + node]);
                    Best b{0.0f, -1};
                    for (uint32_t j = lane; j < k; j += 64) {
                        const uint32_t ci = fcn + j;
                        const float vis = ci < nu ? vvis[ci] : T.visits[base + ci], val = ci < nu ? vval[ci] : T.value[base + ci];
                        const float q = vis == 0.0f ? 0.0f : val * __builtin_amdgcn_rcpf(vis);
                        const float sc_ = q + (c * (sq * __builtin_amdgcn_rcpf(vis + 1.0f))) * X.pri(ci);
                        if (sc_ == sc_ && (b.j < 0 || !(b.s > sc_))) { b.s = sc_; b.j = (int)j; }
                    }
                    b = wave_best(b);
                    node = fcn + (uint32_t)(b.j >= 0 ? b.j : (int)k - 1);
                    ++depth;
                    if ((uint32_t)lane == depth) mine = node;
                    if (depth >= 63) break;
                }
                const uint32_t cr = crow_of(node);
                float x = 0.0f;
                bool fresh = false;
                if (mt & kMetaTerminal) x = ((mt & kMetaWinnerPlus) ? 1 : -1) == root_player ? 1.0f : -1.0f;
                else if (at_hand(cr)) x = node < ln ? lcval[node] : cval_g[node];
                else if (!(mt & kDrained) && node != demanded) fresh = __ballot((uint32_t)lane < ncand && cand[lane] == node) == 0ull;
                if (fresh) { if (lane == 0) { cand[ncand] = node; wl[nw + ncand] = node; } ++ncand; fruitless = 0; } else ++fruitless;
                if ((uint32_t)lane <= depth && mine < nu) { vvis[mine] += 1.0f; vval[mine] += x; }
                __syncthreads();
                FR_COUNT(11, 1);
            }
        }
        nw += ncand;
    }
    if (lane == 0) {
        F.wish_n[slot] = nw | (it << 8) | (dem ? 0x80000000u : 0u);
        store_counters(S, slot, cn);
    }
    FR_STAMP(4);
#ifdef DIEE_TAIL_STAMPS
    if (threadIdx.x == 0) atomicMax(&g_free_stamps[10], __builtin_readcyclecounter() - fr_t0_);
#endif
}

// The rows of launch q: every demanded leaf, then the other wishes rank by rank (every game's first candidate, then every game's second ...; the
// games behind the leader a few ranks earlier) until the launch is full; a game's rows are contiguous (k_free takes them in by grant_off /
// grant_cnt).  One workgroup, one thread per game, the games in an order that turns with the launch (the rank that does not fit whole goes to
// the games that come first).
__global__ __launch_bounds__(1024) void k_free_pack(Free F, uint32_t n, uint32_t node_cap, uint32_t q, const BgState* __restrict__ arena) {
    const uint32_t kBoost = F.lag_boost < kFreeBoostMax ? F.lag_boost : kFreeBoostMax, kLagStep = F.lag_step ? F.lag_step : 1u;
    __shared__ uint32_t hist[kFreeWish + kFreeBoostMax + 2];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_full, s_rem, s_dem, s_undone, s_maxprog;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (F.state[0] != 0u) return;                               // the search was complete a round ago: n_rows[q] stays 0
    if (tid < (int)(kFreeWish + kFreeBoostMax + 2)) hist[tid] = 0;
    if (tid == 0) { s_dem = 0; s_undone = 0; s_maxprog = 0; }
    __syncthreads();
    const uint32_t rot = (q * 131u) % n;
    const uint32_t g = (uint32_t)tid < n ? ((uint32_t)tid + rot) % n : 0u;
    const uint32_t w = (uint32_t)tid < n ? F.wish_n[g] : ((F.iterations) << 8);
    const uint32_t cnt = (w & 0xffu) < kFreeWish ? (w & 0xffu) : kFreeWish, dem = (w >> 31) < cnt ? (w >> 31) : cnt, spec = cnt - dem, prog = (w >> 8) & 0x7fffffu;   // (clamped: hist[] and the wish list hold kFreeWish entries)
    const bool undone = prog < F.iterations;
    if ((uint32_t)tid < n && undone) { atomicMax(&s_maxprog, prog); atomicAdd(&s_undone, 1u); if (dem) atomicAdd(&s_dem, 1u); }
    __syncthreads();
    // A search ends when its LAST game is done: the games behind are served first.  Wish number r of a game that is `lag` iterations behind
    // the leader competes as rank r + kBoost - min(kBoost, lag / kLagStep); every rank that fits whole is granted, the next one as far as the rows go.
    const uint32_t lag = undone ? s_maxprog - prog : 0u;
    const uint32_t shift = kBoost - (lag / kLagStep < kBoost ? lag / kLagStep : kBoost);      // 0 (furthest behind) ... kBoost (the leaders)
    if ((uint32_t)tid < n) for (uint32_t r = 1; r <= spec; ++r) atomicAdd(&hist[r + shift], 1u);
    __syncthreads();
    if (tid == 0) {
        const uint32_t room = F.rows > s_dem ? F.rows - s_dem : 0u;
        uint32_t full = 0, used_rows = 0;
        for (uint32_t k = 1; k <= kFreeWish + kBoost; ++k) {
            if (used_rows + hist[k] > room) break;
            used_rows += hist[k]; full = k;
        }
        s_full = full; s_rem = room - used_rows;
    }
    __syncthreads();
    const uint32_t full = s_full, rem = s_rem;
    // wishes of rank <= full: r + shift <= full, i.e. r <= full - shift; the next rank (full + 1) as far as the rows go, in this launch's order of the games
    const uint32_t whole = full > shift ? (full - shift < spec ? full - shift : spec) : 0u;
    const int more = ((uint32_t)tid < n && spec > whole && whole + 1u + shift == full + 1u) ? 1 : 0;
    int inc = wave_inclusive_scan_i32(more);
    if (lane == 63) wsum[wave] = (uint32_t)inc;
    __syncthreads();
    uint32_t before = 0;
    for (int v = 0; v < wave; ++v) before += wsum[v];
    const uint32_t my_rank = before + (uint32_t)(inc - more);
    __syncthreads();
    const uint32_t grant = (uint32_t)tid < n ? dem + whole + ((more && my_rank < rem) ? 1u : 0u) : 0u;
    inc = wave_inclusive_scan_i32((int)grant);
    if (lane == 63) wsum[wave] = (uint32_t)inc;
    __syncthreads();
    before = 0;
    uint32_t total = 0;
    for (int v = 0; v < 16; ++v) { if (v < wave) before += wsum[v]; total += wsum[v]; }
    const uint32_t off = before + (uint32_t)inc - grant;
    if ((uint32_t)tid < n) {
        const uint32_t fit = off >= F.rows ? 0u : (grant < F.rows - off ? grant : F.rows - off);     // (the host never asks for fewer rows than games: no demanded leaf is cut)
        F.grant_off[g] = off; F.grant_cnt[g] = fit;
        uint32_t* out = F.rows_idx + (size_t)(q % F.ring) * F.rows + off;
        const uint32_t* wl = F.wish + (size_t)g * kFreeWish;
        for (uint32_t j = 0; j < fit; ++j) out[j] = g * node_cap + wl[j];
        if (F.rows_state) {                                     // (launches of the cluster family read their rows densely)
            BgState* so = F.rows_state + (size_t)(q % F.ring) * F.rows + off;
            for (uint32_t j = 0; j < fit; ++j) store_state(&so[j], load_state(&arena[(size_t)g * node_cap + wl[j]]));
        }
    }
    if (tid == 0) {
        if (total > F.rows) total = F.rows;
        F.n_rows[q] = total; F.n_dem[q] = s_dem < total ? s_dem : total;
        const uint32_t all_done = s_undone ? 0u : 1u;
        F.state[0] = all_done; F.state[4] = s_undone;
        if (total) { F.state[1] += 1u; F.state[2] += total - s_dem; }
        F.host[0] = all_done; F.host[1] = q;
        __threadfence_system();
    }
}

// fold the per-slot counters of one move-step into the totals of each batch (one block per batch)
// step_log (may be null): [2 * step] += the move-step's expansions, [2 * step + 1] += the rows it evaluated -- the share of a launch's rows the
// search went on to USE is known only when its search is over; the bench's bands weigh a sampled launch by its move-step's share
__global__ __launch_bounds__(256) void k_reduce_counters(Slots S, Segs G, unsigned long long* step_log, uint32_t step) {
    __shared__ unsigned long long part[SC_COUNT][4];
    const uint32_t seg = blockIdx.x;
    const uint32_t s0 = G.first_slot[seg], s1 = G.end_slot[seg];
    if (s1 <= s0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long acc[SC_COUNT];
#pragma unroll
    for (int c = 0; c < SC_COUNT; ++c) acc[c] = 0;
    for (uint32_t s = s0 + tid; s < s1; s += 256)
#pragma unroll
        for (int c = 0; c < SC_COUNT; ++c) {
            const unsigned long long v = S.slot_cnt[s * SC_COUNT + c];
            if (c == SC_MAX_CHILDREN) acc[c] = v > acc[c] ? v : acc[c]; else acc[c] += v;
        }
#pragma unroll
    for (int c = 0; c < SC_COUNT; ++c) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(acc[c], d);
            if (c == SC_MAX_CHILDREN) acc[c] = o > acc[c] ? o : acc[c]; else acc[c] += o;
        }
        if (lane == 0) part[c][wave] = acc[c];
    }
    __syncthreads();
    if (tid == 0) {
        const int map[SC_COUNT] = {CNT_SELECTIONS, CNT_DEPTH_SUM, CNT_TERMINAL, CNT_EXPANSIONS, CNT_CHILDREN, CNT_MAX_CHILDREN, CNT_ILLEGAL, CNT_NN_ROWS};
        unsigned long long* cnt = S.counters + (size_t)seg * CNT_COUNT;
        for (int c = 0; c < SC_COUNT; ++c) {
            unsigned long long t = 0;
            for (int w = 0; w < 4; ++w) t = c == SC_MAX_CHILDREN ? (part[c][w] > t ? part[c][w] : t) : t + part[c][w];
            if (c == SC_MAX_CHILDREN) { if (t > cnt[map[c]]) cnt[map[c]] = t; }
            else cnt[map[c]] += t;
            if (step_log && c == SC_EXPANSIONS) atomicAdd(&step_log[2 * (size_t)step], t);
            if (step_log && c == SC_NN_ROWS) atomicAdd(&step_log[2 * (size_t)step + 1], t);
        }
    }
}

// ---- rows of the next network evaluation -------------------------------------------------------
// The reference pushes all N slots through the network every iteration (alpha_mcts.rs:175-183) although the rows of
// slots whose selected leaf was terminal are stale and their results are never read (the engine keeps the value such a
// slot needs, Slots::sel_value).  Above 256 live games the engine evaluates only the slots with skip[slot] == 0, in slot
// order (stable, so a slot's row depends on nothing but the flags): row_slot[row] = slot, slot_row[slot] = row.
// One block; the network launches that follow read *n_rows on the device, the host never sees it.
__global__ __launch_bounds__(1024) void k_row_map(const uint8_t* __restrict__ skip, uint32_t n, uint32_t* __restrict__ row_slot,
                                                  uint32_t* __restrict__ slot_row, uint32_t* __restrict__ n_rows,
                                                  uint32_t* __restrict__ rows_log, uint32_t log_idx) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n; c0 += 1024) {
        const uint32_t i = c0 + tid;
        const bool keep = i < n && skip[i] == 0;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = carry;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        const uint32_t pos = off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (keep) { row_slot[pos] = i; slot_row[i] = pos; }
        __syncthreads();
        if (tid == 0) { uint32_t t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; carry += t; }
        __syncthreads();
    }
    if (tid == 0) { *n_rows = carry; if (rows_log) rows_log[log_idx] = carry; }
}

// ---- get_prob_tensor_parallel, utils.rs:42-58 --------------------------------------------------
__global__ __launch_bounds__(64) void k_root_probs(Tree T, uint32_t n, float* __restrict__ probs,
                                                   uint32_t* __restrict__ nch, float* __restrict__ root_visits) {
    __shared__ float sum_s;
    const uint32_t slot = blockIdx.x;
    if (slot >= n) return;
    const int lane = threadIdx.x;
    const size_t base = (size_t)slot * T.node_cap;
    const uint32_t k = meta_nch(T.meta[base]), fc = T.first_child[base];
    float* row = probs + (size_t)slot * 1352;
    const float fill = k == 0 ? __uint_as_float(0x7fc00000u) : 0.0f;       // 0/0 row
    for (int a = lane; a < 1352; a += 64) row[a] = fill;
    if (lane == 0) {
        float s = 0.0f;
        for (uint32_t j = 0; j < k; ++j) s += T.visits[base + fc + j];
        sum_s = s;
        nch[slot] = k;
        if (root_visits) root_visits[slot] = T.visits[base];
    }
    __syncthreads();
    const float sum = sum_s;
    for (uint32_t j = lane; j < k; j += 64) row[T.meta[base + fc + j] & 0xFFFFu] = T.visits[base + fc + j] / sum;
}

// ---- self-play -----------------------------------------------------------------------------------
// alpha_parallel.rs:103-111: T::new() + roll_die for every game of every batch
__global__ void k_init_games(Games Gm, Segs G, uint32_t n) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint32_t seg = 0;
    while (seg + 1 < G.n && g >= G.game0[seg + 1]) ++seg;
    BgState s;
    // Backgammon::new, backgammon_logic.rs:80-94
    s.w[0] = 0x00000002u; s.w[1] = 0xFD00FB00u; s.w[2] = 0x05000000u; s.w[3] = 0x000000FBu;
    s.w[4] = 0x00050003u; s.w[5] = 0xFE000000u; s.w[6] = 0u;
    int d0, d1;
    draw_dice(G.seed[seg], G.first_id[seg] + (g - G.game0[seg]), 0u, kTagInitRoll, 0u, d0, d1);
    s.w[7] = (uint32_t)d0 | ((uint32_t)d1 << 8) | (0xFFu << 16);
    store_state(&Gm.state[g], s);
    Gm.rounds[g] = 0; Gm.nfrags[g] = 0; Gm.alive[g] = 1; Gm.winner[g] = 0;
    Gm.ev_a_count[g] = kNone; Gm.ev_b_count[g] = kNone; Gm.ev_a_step[g] = 0; Gm.ev_b_step[g] = 0;
    Gm.live[g] = g; Gm.seg[g] = (uint8_t)seg;
}

// live games -> slots; the live list is ascending in the game index, so the slots of a batch are contiguous: the
// slot whose predecessor belongs to another batch is the batch's first (the reference's index 0), likewise its end
__global__ void k_gather_roots(Games Gm, Slots S, Segs G, uint32_t n_live) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_live) return;
    const uint32_t g = Gm.live[slot];
    const uint32_t seg = Gm.seg[g];
    store_state(&S.roots[slot], load_state(&Gm.state[g]));
    S.game_id[slot] = G.first_id[seg] + (g - G.game0[seg]);
    S.round[slot] = Gm.rounds[g];
    S.seg[slot] = seg;
    if (slot == 0 || Gm.seg[Gm.live[slot - 1]] != seg) G.first_slot[seg] = slot;
    if (slot + 1 == n_live || Gm.seg[Gm.live[slot + 1]] != seg) G.end_slot[seg] = slot + 1;
}

// the body of the per-game loop of self_play_parallel, alpha_parallel.rs:168-224
__global__ __launch_bounds__(64) void k_play_move(Tree T, Games Gm, Segs G, uint32_t n_live, uint32_t step, PlayParams P) {
    __shared__ float row[1352];
    const uint32_t slot = blockIdx.x;
    if (slot >= n_live) return;
    const int lane = threadIdx.x;
    const uint32_t g = Gm.live[slot];
    const size_t base = (size_t)slot * T.node_cap;
    const uint32_t k = meta_nch(T.meta[base]), fc = T.first_child[base];
    BgState s = load_state(&Gm.state[g]);
    const uint32_t round = Gm.rounds[g];
    const uint32_t seg = Gm.seg[g];
    const uint32_t gid = G.first_id[seg] + (g - G.game0[seg]);
    const unsigned long long seed = G.seed[seg];
    unsigned long long* cnt = Gm.counters + (size_t)seg * CNT_COUNT;
    bool removed = false, flushed = false;
    if (round >= P.round_limit) {                               // :172-180 (no `continue`)
        if (lane == 0) { Gm.ev_a_count[g] = Gm.nfrags[g]; Gm.ev_a_step[g] = step; }
        removed = true; flushed = true;
    }
    if (k == 0) {                                               // :183-189 skip_turn
        if (lane == 0) {
            int d0, d1;
            draw_dice(seed, gid, round, kTagMoveRoll, 0u, d0, d1);
            bg_skip_dev(s, d0, d1);
            store_state(&Gm.state[g], s);
            Gm.rounds[g] = round + 1;
            if (removed) { Gm.alive[g] = 0; atomicAdd(&cnt[CNT_GAMES], 1ull); }
            atomicAdd(&cnt[CNT_PLIES], 1ull);
        }
        return;
    }
    // get_prob_tensor_parallel row (:164) and pow_(1/T) (:165), not renormalised (Q17)
    for (int a = lane; a < 1352; a += 64) row[a] = 0.0f;
    // the children's visits and codes in one round of loads (k <= 256: four per lane); the row sum in child order on values broadcast
    // from their lanes (the same chain of f32 additions in every lane)
    float vis[4]; uint32_t cod[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t j = lane + 64 * r;
        vis[r] = j < k ? T.visits[base + fc + j] : 0.0f;
        cod[r] = j < k ? T.meta[base + fc + j] & 0xFFFFu : 0u;
    }
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kr = (int)k - 64 * r < 64 ? (int)k - 64 * r : 64;                    // uniform
        for (int j = 0; j < kr; ++j) sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vis[r]), j));
    }
    __syncthreads();                                            // (the zeroed row before the scattered weights)
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if ((uint32_t)(lane + 64 * r) < k) row[cod[r]] = det_powf(vis[r] / sum, P.inv_temperature);
    __syncthreads();
    // weighted_select_tensor_idx (alphazero.rs:129-137): rand WeightedIndex over f64 weights
    // The f64 sums run over all 1352 weights in index order; at most k of them are not +0.0, and x + 0.0 == x exactly, so only those
    // are added, in index order: one ballot per 64 codes finds them (1352 dependent f64 adds from LDS, twice, were ~85 us per move-step).
    unsigned long long nzm[22];
#pragma unroll
    for (int q = 0; q < 22; ++q) nzm[q] = __ballot(64 * q + lane < 1352 && row[64 * q + lane] != 0.0f);
    double total = 0.0;
#pragma unroll
    for (int q = 0; q < 22; ++q)
        for (unsigned long long m = nzm[q]; m; m &= m - 1) total += (double)row[64 * q + __builtin_ctzll(m)];
    const double x = draw_uniform(seed, gid, round, kTagSample, 0u) * total;
    double cum = 0.0;
    int pick = -1, last_nz = 0;
#pragma unroll
    for (int q = 0; q < 22; ++q)
        for (unsigned long long m = nzm[q]; m && pick < 0; m &= m - 1) {
            const int a = 64 * q + __builtin_ctzll(m);
            last_nz = a;
            cum += (double)row[a];
            if (cum > x) pick = a;
        }
    const uint32_t code = (uint32_t)(pick >= 0 ? pick : last_nz);        // (uniform: every lane ran the same chain)
    // MemoryFragment{outcome: player, ps, state} (:195-199)
    const uint32_t nf = Gm.nfrags[g];
    const size_t fi = (size_t)g * Gm.frag_cap + nf;
    for (int a = lane; a < 1352; a += 64) Gm.frag_ps[fi * 1352 + a] = row[a];
    for (int t = lane; t < 144; t += 64) Gm.frag_planes[fi * 144 + t] = bg_plane_dev(s, t / 24, t % 24);
    if (lane == 0) {
        Gm.frag_player[fi] = (int8_t)st_player(s);
        Gm.nfrags[g] = nf + 1;
        // decode + apply_move (:202-210); legality of decode(code) is checked when the root is expanded
        const uint32_t play = bg_decode_dev(st_roll(s, 0), st_roll(s, 1), st_player(s), code);
        int d0, d1;
        draw_dice(seed, gid, round, kTagMoveRoll, 0u, d0, d1);
        bg_apply_dev(s, play, d0, d1);
        store_state(&Gm.state[g], s);
        Gm.rounds[g] = round + 1;                               // :213
        atomicAdd(&cnt[CNT_PLIES], 1ull);
        const int w = bg_winner_dev(s);
        if (w != 0) {                                           // :215-223
            if (!(flushed && !P.quirks)) { Gm.ev_b_count[g] = nf + 1; Gm.ev_b_step[g] = step; }
            Gm.winner[g] = (int8_t)w;
            removed = true;
        }
        if (removed) { Gm.alive[g] = 0; atomicAdd(&cnt[CNT_GAMES], 1ull); }
    }
}

// stable compaction of the live list (one block); n_live_out[0] = games alive, [1 + b] = alive in batch b
__global__ __launch_bounds__(1024) void k_compact_live(Games Gm, uint32_t n_live, uint32_t n_segs, uint32_t* n_live_out) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    __shared__ uint32_t per_seg[kMaxSegments];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    if (tid < (int)kMaxSegments) per_seg[tid] = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n_live; c0 += 1024) {
        const uint32_t i = c0 + tid;
        uint32_t g = 0;
        bool keep = false;
        if (i < n_live) { g = Gm.live[i]; keep = Gm.alive[g] != 0; }
        if (keep) atomicAdd(&per_seg[Gm.seg[g]], 1u);
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = carry;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        const uint32_t pos = off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();                                        // all reads of live[c0..] done before writes
        if (keep) Gm.live[pos] = g;
        __syncthreads();
        if (tid == 0) { uint32_t t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; carry += t; }
        __syncthreads();
    }
    if (tid == 0) n_live_out[0] = carry;
    if (tid < (int)n_segs) n_live_out[1 + tid] = per_seg[tid];
}

// ---- output delivery (alpha_parallel.rs:172-180, :215-223: `all_memories.append(...)` in the step a game is removed) ----
// One block lists the flushes of this move-step in the live list's order (ascending game, hence batch-major; the round-limit
// flush of a game before its win flush) and gives each the row it starts at in its batch's output of the step.  Runs behind
// k_play_move, before k_compact_live drops the removed games from the list.  summary: DeliverSummary.
__global__ __launch_bounds__(1024) void k_deliver_scan(Games Gm, Segs G, uint32_t n_live, uint32_t step, DeliverEvent* __restrict__ ev,
                                                       uint32_t* __restrict__ summary) {
    __shared__ uint32_t wrow[16], wev[16];
    __shared__ uint32_t carry_rows, carry_ev;
    __shared__ uint32_t seg_base[kMaxSegments], seg_rows[kMaxSegments], seg_ev0[kMaxSegments], seg_nev[kMaxSegments];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { carry_rows = 0; carry_ev = 0; }
    if (tid < (int)kMaxSegments) { seg_base[tid] = 0; seg_rows[tid] = 0; seg_ev0[tid] = kNone; seg_nev[tid] = 0; }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n_live; c0 += 1024) {
        const uint32_t i = c0 + tid;
        uint32_t g = 0, seg = 0, ca = 0, cb = 0;
        if (i < n_live) {
            g = Gm.live[i]; seg = Gm.seg[g];
            const uint32_t a = Gm.ev_a_count[g], b = Gm.ev_b_count[g];
            if (a != kNone && Gm.ev_a_step[g] == step) ca = a;
            if (b != kNone && Gm.ev_b_step[g] == step) cb = b;
        }
        const uint32_t nr = ca + cb, ne = (ca ? 1u : 0u) + (cb ? 1u : 0u);
        uint32_t pr = nr, pe = ne;                              // inclusive scans inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t xr = __shfl_up(pr, d, 64), xe = __shfl_up(pe, d, 64);
            if (lane >= d) { pr += xr; pe += xe; }
        }
        if (lane == 63) { wrow[wave] = pr; wev[wave] = pe; }
        __syncthreads();
        uint32_t off_r = carry_rows, off_e = carry_ev;
        for (int w = 0; w < wave; ++w) { off_r += wrow[w]; off_e += wev[w]; }
        const uint32_t row0 = off_r + pr - nr;                  // rows delivered by the slots before this one (all batches)
        uint32_t e = off_e + pe - ne;
        if (i < n_live && i == G.first_slot[seg]) seg_base[seg] = row0;
        if (ca) { ev[e] = DeliverEvent{g, ca, row0, seg}; ++e; }
        if (cb) ev[e] = DeliverEvent{g, 0x80000000u | cb, row0 + ca, seg};
        __syncthreads();
        if (tid == 0) { uint32_t tr = 0, te = 0; for (int w = 0; w < 16; ++w) { tr += wrow[w]; te += wev[w]; } carry_rows += tr; carry_ev += te; }
        __syncthreads();
    }
    const uint32_t n_ev = carry_ev;
    for (uint32_t e = tid; e < n_ev; e += 1024) {               // rows relative to the batch's first this step
        DeliverEvent d = ev[e];
        d.row0 -= seg_base[d.seg];
        ev[e] = d;
        atomicAdd(&seg_rows[d.seg], d.kind_count & 0x7FFFFFFFu);
        atomicMin(&seg_ev0[d.seg], e);
        atomicAdd(&seg_nev[d.seg], 1u);
    }
    __syncthreads();
    if (tid < (int)G.n) {
        summary[DeliverSummary::rows(G.n, tid)] = seg_rows[tid];
        summary[DeliverSummary::ev0(G.n, tid)] = seg_nev[tid] ? seg_ev0[tid] : 0u;
        summary[DeliverSummary::nev(G.n, tid)] = seg_nev[tid];
    }
}

// rows [r0, r1) of one batch's output of the step -> staging rows [0, r1 - r0): ps and planes as the game recorded them,
// outcome relabelled (:216-217: +1 where the fragment's mover won, -1 where the opponent did; 0 for a round-limit flush),
// the originating game id.  ev = the batch's n_ev events (ascending row0).
__global__ __launch_bounds__(256) void k_deliver_copy(Games Gm, Segs G, const DeliverEvent* __restrict__ ev, uint32_t n_ev, uint32_t r0,
                                                      uint32_t r1, DeliverOut out) {
    for (uint32_t r = r0 + blockIdx.x; r < r1; r += gridDim.x) {
        uint32_t lo = 0, hi = n_ev;                             // last event with row0 <= r
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ev[mid].row0 <= r) lo = mid; else hi = mid; }
        const DeliverEvent d = ev[lo];
        const uint32_t fr = r - d.row0;
        const size_t s = (size_t)d.g * Gm.frag_cap + fr, o = (size_t)(r - r0);
        for (int a = threadIdx.x; a < 1352 / 4; a += 256)
            reinterpret_cast<float4*>(out.ps + o * 1352)[a] = reinterpret_cast<const float4*>(Gm.frag_ps + s * 1352)[a];
        if (threadIdx.x < 144 / 4)
            reinterpret_cast<float4*>(out.planes + o * 144)[threadIdx.x] = reinterpret_cast<const float4*>(Gm.frag_planes + s * 144)[threadIdx.x];
        if (threadIdx.x == 0) {
            const int pl = Gm.frag_player[s], w = Gm.winner[d.g];
            out.outcome[o] = (d.kind_count >> 31) == 0 ? (int8_t)0 : (int8_t)(w == pl ? 1 : (w == -pl ? -1 : 0));
            out.game[o] = G.first_id[d.seg] + (d.g - G.game0[d.seg]);
        }
    }
}

// ---- host launchers -----------------------------------------------------------------------------
void launch_init_roots(hipStream_t st, const Tree& T, const Slots& S, uint32_t n) {
    hipLaunchKernelGGL(k_init_roots, dim3((n + 255) / 256), dim3(256), 0, st, T, S, n);
}
void launch_expand(hipStream_t st, const Tree& T, const Slots& S, const Segs& G, uint32_t n, uint32_t it, const SearchParams& P,
                   uint32_t next_it, float c, bool pre_grown, ExpandVariant v) {
    // v.two = false (option expand2 = 0): one wave per slot creates the children and then does the rest (k_expand<false>) instead of two waves side by side
    // v.two_c = false (option expand2c = 0): children grown by the tower launch are committed by the one wave that does everything else too
    // (the caller reads the switches once per search: no getenv beside a kernel launch)
    if (pre_grown && v.two_c) hipLaunchKernelGGL((k_expand<true, 2>), dim3(n), dim3(128), 0, st, T, S, G, n, it, P, next_it, c);
    else if (pre_grown) hipLaunchKernelGGL((k_expand<true, 0>), dim3(n), dim3(64), 0, st, T, S, G, n, it, P, next_it, c);
    else if (v.two) hipLaunchKernelGGL((k_expand<true, 1>), dim3(n), dim3(128), 0, st, T, S, G, n, it, P, next_it, c);
    else hipLaunchKernelGGL((k_expand<false, 0>), dim3(n), dim3(64), 0, st, T, S, G, n, it, P, next_it, c);
}
void launch_tail(hipStream_t st, const Tree& T, const Slots& S, const Segs& G, uint32_t n, const SearchParams& P, float c, const Tail& L, uint32_t q) {
    static bool attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 15]) {
        (void)hipFuncSetAttribute((const void*)k_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TailLds));
        attr_set[dev & 15] = true;
    }
    hipLaunchKernelGGL(k_tail, dim3(n <= 32 ? 8 * n : n), dim3(64), sizeof(TailLds), st, T, S, G, n, P, c, TailArgs{L, q});
}
uint32_t free_lds_nodes_for(uint32_t n, uint32_t cus) {
    // the workgroups (one per game) that share a CU split its 160 KB of LDS: what a game's scratch leaves goes to its tree
    const uint32_t per_cu = (n + cus - 1) / (cus ? cus : 1u);
    const size_t budget = (size_t)160 * 1024 / (per_cu ? per_cu : 1u);
    const size_t fixed = free_lds_bytes(0);
    if (budget <= fixed + 64 * kFreeLdsNodeBytes) return 64;
    uint32_t ln = (uint32_t)((budget - fixed) / kFreeLdsNodeBytes) / 64 * 64;
    return ln > kTailLdsNodes ? kTailLdsNodes : ln;
}
void launch_free(hipStream_t st, const Tree& T, const Slots& S, const Segs& G, uint32_t n, const SearchParams& P, float c, const Free& F, uint32_t q) {
    static bool attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 15]) {
        (void)hipFuncSetAttribute((const void*)k_free, hipFuncAttributeMaxDynamicSharedMemorySize, (int)free_lds_bytes(kTailLdsNodes));
        attr_set[dev & 15] = true;
    }
    hipLaunchKernelGGL(k_free, dim3(n), dim3(64), free_lds_bytes(F.lds_nodes), st, T, S, G, n, P, c, FreeArgs{F, q});
    hipLaunchKernelGGL(k_free_pack, dim3(1), dim3(1024), 0, st, F, n, T.node_cap, q, T.state);
}
void launch_reduce_counters(hipStream_t st, const Slots& S, const Segs& G, unsigned long long* step_log, uint32_t step) {
    hipLaunchKernelGGL(k_reduce_counters, dim3(G.n), dim3(256), 0, st, S, G, step_log, step);
}
void launch_row_map(hipStream_t st, const uint8_t* skip, uint32_t n, uint32_t* row_slot, uint32_t* slot_row, uint32_t* n_rows,
                    uint32_t* rows_log, uint32_t log_idx) {
    hipLaunchKernelGGL(k_row_map, dim3(1), dim3(1024), 0, st, skip, n, row_slot, slot_row, n_rows, rows_log, log_idx);
}
void launch_root_probs(hipStream_t st, const Tree& T, uint32_t n, float* probs, uint32_t* nch, float* root_visits) {
    hipLaunchKernelGGL(k_root_probs, dim3(n), dim3(64), 0, st, T, n, probs, nch, root_visits);
}
void launch_init_games(hipStream_t st, const Games& Gm, const Segs& G, uint32_t n) {
    hipLaunchKernelGGL(k_init_games, dim3((n + 255) / 256), dim3(256), 0, st, Gm, G, n);
}
void launch_gather_roots(hipStream_t st, const Games& Gm, const Slots& S, const Segs& G, uint32_t n_live) {
    hipLaunchKernelGGL(k_gather_roots, dim3((n_live + 255) / 256), dim3(256), 0, st, Gm, S, G, n_live);
}
void launch_play_move(hipStream_t st, const Tree& T, const Games& Gm, const Segs& G, uint32_t n_live, uint32_t step, const PlayParams& P) {
    hipLaunchKernelGGL(k_play_move, dim3(n_live), dim3(64), 0, st, T, Gm, G, n_live, step, P);
}
void launch_compact_live(hipStream_t st, const Games& Gm, uint32_t n_live, uint32_t n_segs, uint32_t* n_live_out) {
    hipLaunchKernelGGL(k_compact_live, dim3(1), dim3(1024), 0, st, Gm, n_live, n_segs, n_live_out);
}
void launch_deliver_scan(hipStream_t st, const Games& Gm, const Segs& G, uint32_t n_live, uint32_t step, DeliverEvent* ev, uint32_t* summary) {
    hipLaunchKernelGGL(k_deliver_scan, dim3(1), dim3(1024), 0, st, Gm, G, n_live, step, ev, summary);
}
void launch_deliver_copy(hipStream_t st, const Games& Gm, const Segs& G, const DeliverEvent* ev, uint32_t n_ev, uint32_t r0, uint32_t r1,
                         const DeliverOut& out) {
    if (r1 <= r0 || !n_ev) return;
    hipLaunchKernelGGL(k_deliver_copy, dim3(std::min<uint32_t>(r1 - r0, 8192u)), dim3(256), 0, st, Gm, G, ev, n_ev, r0, r1, out);
}

}  // namespace diee
