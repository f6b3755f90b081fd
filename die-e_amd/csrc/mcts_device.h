// mcts_device.h -- device code of the search shared between translation units: the tree's small helpers and grow_slot, the
// network-independent half of an expansion, which runs on k_expand<true, 1>'s second wave (mcts_kernels.hip) and as
// extra workgroups of the cluster-tower launch (nn_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bg_device.h"
#include "search_types.h"

namespace diee {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kDrained = 0x80000000u;

// Tree::meta: bit 31 expanded, bit 30 the node's state is a finished game, bit 29 ... won by player +1 (both set when the node is
// created: the virtual descents of the tail read them instead of the 32-byte state), bits 28..16 children, bits 15..0 action code
constexpr uint32_t kMetaTerminal = 0x40000000u, kMetaWinnerPlus = 0x20000000u;
constexpr uint32_t kMetaKeep = 0xFFFFu | kMetaTerminal | kMetaWinnerPlus;      // what an expansion keeps of a node's header (a finished ROOT is expanded like any root)
__device__ __forceinline__ uint32_t meta_nch(uint32_t m) { return (m >> 16) & 0x1fffu; }
__device__ __forceinline__ uint32_t meta_terminal_bits(const BgState& s) {
    const int w = bg_winner_dev(s);
    return w == 0 ? 0u : (kMetaTerminal | (w > 0 ? kMetaWinnerPlus : 0u));
}

__device__ __forceinline__ BgState load_state(const BgState* p) {
    BgState s;
    const uint4 a = ((const uint4*)p)[0], b = ((const uint4*)p)[1];
    s.w[0] = a.x; s.w[1] = a.y; s.w[2] = a.z; s.w[3] = a.w;
    s.w[4] = b.x; s.w[5] = b.y; s.w[6] = b.z; s.w[7] = b.w;
    return s;
}
__device__ __forceinline__ void store_state(BgState* p, const BgState& s) {
    ((uint4*)p)[0] = make_uint4(s.w[0], s.w[1], s.w[2], s.w[3]);
    ((uint4*)p)[1] = make_uint4(s.w[4], s.w[5], s.w[6], s.w[7]);
}

constexpr uint32_t kRootIt = 0xFFFFFFFFu;

// alpha_expand_tensor (node.rs:157-174) creates one child per legal play: state (apply_move with frozen dice), parent, action
// code -- none of which depends on the evaluation of the leaf; only the priors do.  grow_slot does that part for the leaf the
// selection chose, by ONE wave: legal plays, codes, child states and headers at [used, used + k) of the slot's arena -- not
// linked to the parent and `used` not advanced, so the tree is unchanged until k_expand<true> commits them with their priors.
// Slots::grow_k = k (kNone: nothing to expand here, or no room), Slots::grow_code = the codes, four per lane.
template <bool ONE_WAVE_BLOCK = true>
__device__ __forceinline__ void grow_slot(const Tree& T, const Slots& S, const Segs& G, uint32_t n, uint32_t it, uint32_t slot, WaveScratch& ws,
                                          uint32_t* hand = nullptr) {      // hand (LDS, 129 words): k and the codes for a wave of the SAME workgroup (k_expand<true, true>)
    if (slot >= n) return;
    const int lane = threadIdx.x & 63;
    const size_t base = (size_t)slot * T.node_cap;
    const bool root = it == kRootIt;
    const uint32_t seg = G.n == 1 ? 0u : S.seg[slot];
    const bool lterm = !root && S.leaf_term[slot] != 0;
    const uint32_t leaf = root ? 0u : S.leaf[slot];
    const uint32_t m0 = root ? T.meta[base] : S.leaf_meta[slot];
    const uint32_t first = T.used[slot];
    const uint32_t gid = S.game_id[slot], rnd = S.round[slot];
    const BgState st = load_state(root ? &T.state[base] : &S.eval_states[slot]);
    const uint32_t if0 = root ? 1u : S.iter_flags[2 * ((size_t)seg * G.iter_cap + it)];
    const unsigned long long seed = G.seed[seg];
    if (hand) __syncthreads();                               // k_expand<true, true>: wave 0 rewrites this slot's selection record after this point
    if (if0 == 0 || lterm || (m0 & kDrained)) {             // k_expand's `active`, `do_expand` and drained tests
        if (lane == 0) { if (hand) hand[0] = kNone; else S.grow_k[slot] = kNone; }
        return;
    }
    int k = bg_legal_plays_wave<ONE_WAVE_BLOCK>(st, &ws, lane, S.overflow);
    if (k > kMaxPlays) { if (lane == 0) atomicOr(S.overflow, 1u); k = 0; }     // never silent: DIEE_ERR_CAPACITY
    if (first + (uint32_t)k > T.node_cap) {
        if (lane == 0) { atomicOr(S.overflow, 2u); if (hand) hand[0] = kNone; else S.grow_k[slot] = kNone; }
        return;
    }
    const int r0 = st_roll(st, 0), r1 = st_roll(st, 1), player = st_player(st);
    const uint32_t e = root ? 0u : it + 1u;
    uint32_t codes[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lane + 64 * q;
        if (j >= k) continue;
        const uint32_t play = ws.play[j];
        const uint32_t code = bg_encode_dev(r0, r1, play);
        codes[q] = code;
        if (root && bg_decode_dev(r0, r1, player, code) != play) atomicAdd(&S.slot_cnt[slot * SC_COUNT + SC_ILLEGAL], 1u);   // alpha_parallel.rs:204
        const size_t ci = base + first + j;
        BgState cs = st;
        int d0, d1;
        draw_dice(seed, gid, rnd, e, (uint32_t)j, d0, d1);          // child dice frozen at creation (Q9)
        bg_apply_dev(cs, play, d0, d1);
        store_state(&T.state[ci], cs);
        T.visits[ci] = 0.0f; T.value[ci] = 0.0f;
        T.parent[ci] = leaf; T.first_child[ci] = 0; T.meta[ci] = code | meta_terminal_bits(cs);
    }
    const uint2 packed = make_uint2(codes[0] | (codes[1] << 16), codes[2] | (codes[3] << 16));
    if (hand) {
        hand[1 + 2 * lane] = packed.x; hand[2 + 2 * lane] = packed.y;
        if (lane == 0) hand[0] = (uint32_t)k;
        return;
    }
    *(uint2*)(S.grow_code + (size_t)slot * kMaxPlays + lane * 4) = packed;
    if (lane == 0) S.grow_k[slot] = (uint32_t)k;
}

}  // namespace diee
