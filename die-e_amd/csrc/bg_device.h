// bg_device.h -- backgammon step dynamics as wave-cooperative device code for gfx950 (wave64).
//
// What it computes follows the reference's src/backgammon/backgammon_logic.rs (lines cited per
// function); how it computes it does not: the reference builds a recursive ActionNode tree, DFS
// sequences and a HashSet<Board>; here ONE wavefront enumerates the plays of ONE state with
// 24-bit occupancy masks (ballots), one lane per first move, an LDS sequence table, and a
// first-occurrence-wins dedup on an exact 128-bit board-delta key.
#pragma once
#include "wave_ops.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace diee {

constexpr int kSeqCap = 640;     // DFS sequences per state (hard bound 34 roots x 17 children = 578; more raises the capacity flag)
constexpr int kTbl = 1024;       // dedup hash slots (load factor <= 0.56)
constexpr int kMaxPlays = 256;   // legal plays of one state after dedup (measured maximum 139; more raises the capacity flag)
constexpr int kNoMove = -2;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;

// 32-byte state, same bytes as diee_bg_state: pts[24], bar[2], off[2], roll[2], player, second
struct alignas(16) BgState {
    uint32_t w[8];
};

__device__ __forceinline__ int st_bar(const BgState& s, int i) { return (s.w[6] >> (8 * i)) & 0xff; }
__device__ __forceinline__ int st_off(const BgState& s, int i) { return (s.w[6] >> (16 + 8 * i)) & 0xff; }
__device__ __forceinline__ int st_roll(const BgState& s, int i) { return (s.w[7] >> (8 * i)) & 0xff; }
__device__ __forceinline__ int st_player(const BgState& s) { return (int)(int8_t)((s.w[7] >> 16) & 0xff); }
__device__ __forceinline__ int st_second(const BgState& s) { return (s.w[7] >> 24) & 0xff; }

// per-wave LDS scratch for legal-play enumeration (18.4 KB: with k_expand's 1.5 KB on top, 8 one-wave workgroups share a CU's
// 160 KB -- at 31 KB it was 5, and above 1280 live games k_expand ran in rounds)
struct WaveScratch {
    uint64_t keyA[kSeqCap];
    uint64_t keyB[kSeqCap];
    uint32_t play[kSeqCap];      // bytes f1,t1,f2,t2 (int8)
    uint32_t owner[kTbl];
    uint16_t slot[kSeqCap];
    uint8_t status[kSeqCap];     // 0 active, 1 survivor, 2 duplicate
    uint8_t pts[32];             // the state's bytes, for per-lane byte reads
    uint32_t flag[4];
};

__device__ __forceinline__ uint32_t pack_play(int f1, int t1, int f2, int t2) {
    return (uint32_t)(f1 & 0xff) | ((uint32_t)(t1 & 0xff) << 8) | ((uint32_t)(f2 & 0xff) << 16) |
           ((uint32_t)(t2 & 0xff) << 24);
}
__device__ __forceinline__ int play_f1(uint32_t p) { return (int)(int8_t)(p & 0xff); }
__device__ __forceinline__ int play_t1(uint32_t p) { return (int)(int8_t)((p >> 8) & 0xff); }
__device__ __forceinline__ int play_f2(uint32_t p) { return (int)(int8_t)((p >> 16) & 0xff); }
__device__ __forceinline__ int play_t2(uint32_t p) { return (int)(int8_t)((p >> 24) & 0xff); }

// Candidate first/second moves for one die as a 25-bit mask: bit 0 = entry from the bar
// (from = -1), bit f+1 = a move from point f.  Per die each `from` has at most one target, so the
// reference's sorted (die, from, to) candidate list (backgammon_logic.rs:618-620, 684-686) is
// "low die first, then ascending bit".  Follows get_entry_moves :662-703, get_normal_moves
// :555-636 incl. the arithmetic-sum bear-off guard (:571-579 player -1, :588-596 player +1).
// home6: bytes 0..5 = the mover's home points in ascending absolute index (0..5 / 18..23).
__device__ __forceinline__ uint32_t cand_mask(int m, int player, uint32_t own, uint32_t open,
                                              uint64_t home6, int bar_own, bool collectible) {
    if (bar_own > 0) {                                   // _get_action_trees :547-551
        const int e = player < 0 ? 24 - m : m - 1;
        return (open >> e) & 1u;
    }
    uint32_t mask = 0;
    if (collectible) {
        if (player < 0) {
            const int point = m - 1;
            if ((own >> point) & 1u) mask |= 1u << (point + 1);
            int run = 0;                                 // sum of home[i+1..5]
            bool found = false;
#pragma unroll
            for (int i = 5; i >= 0; --i) {
                if (!found && i < point && ((own >> i) & 1u) && run >= 0) { mask |= 1u << (i + 1); found = true; }
                run += (int)(int8_t)(home6 >> (8 * i));
            }
        } else {
            const int point = 24 - m;
            if ((own >> point) & 1u) mask |= 1u << (point + 1);
            int run = 0;                                 // sum of pts[18..i-1]
            bool found = false;
#pragma unroll
            for (int h = 0; h < 6; ++h) {
                const int i = 18 + h;
                if (!found && i >= point && ((own >> i) & 1u) && run <= 0) { mask |= 1u << (i + 1); found = true; }
                run += (int)(int8_t)(home6 >> (8 * h));
            }
        }
    }
    const uint32_t nm = player < 0 ? (own & (open << m)) : (own & (open >> m));   // :600-617
    mask |= (nm & 0xFFFFFFu) << 1;
    return mask;
}

__device__ __forceinline__ int nth_set_bit(uint32_t mask, int n) {
    for (int i = 0; i < n; ++i) mask &= mask - 1;
    return __ffs(mask) - 1;
}

__device__ __forceinline__ uint64_t home_add(uint64_t home6, int point, int delta, int player) {
    const int idx = player < 0 ? point : point - 18;
    if (idx >= 0 && idx < 6) {
        const int sh = 8 * idx;
        const uint64_t b = (uint64_t)(((int)(int8_t)(home6 >> sh) + delta) & 0xff);
        home6 = (home6 & ~(0xffull << sh)) | (b << sh);
    }
    return home6;
}

// 3-bit-per-point delta key (bias 2 per field): two plays reach the same final Board iff their
// keys are equal (remove_duplicate_states compares full boards, :753-774).
struct DKey {
    uint64_t a, b;
};
__device__ __forceinline__ void key_add(DKey& k, int point, int delta) {
    if (point < 12) k.a += (uint64_t)(int64_t)delta << (3 * point);
    else k.b += (uint64_t)(int64_t)delta << (3 * (point - 12));
}
__device__ __forceinline__ void key_move(DKey& k, int f, int t, bool hit) {
    if (f >= 0) key_add(k, f, -1);
    if (t >= 0) key_add(k, t, hit ? 2 : 1);
    else k.a += 1ull << 36;                              // one more checker collected
    if (hit) k.a += 1ull << 38;                          // one more opponent checker on the bar
}
__device__ __forceinline__ DKey key_init() {
    DKey k;
    k.a = 0x492492492ull;                                // 12 fields of value 2
    k.b = 0x492492492ull;
    return k;
}

// get_valid_moves (backgammon_logic.rs:403-414) for ONE state by ONE wave (blockDim.x == 64).
// Returns k; the plays are left in sc->play[0..k) in the reference's order.  *overflow is set when
// the sequence table would overflow (never silent).
// ONE_WAVE_BLOCK = false: the caller is one wave of a larger workgroup (the growth blocks inside the cluster-tower launch); every
// meeting point then only orders this wave's own LDS accesses -- which execute in program order anyway -- for the compiler.
template <bool ONE_WAVE_BLOCK = true>
__device__ inline int bg_legal_plays_wave(const BgState& s, WaveScratch* sc, int lane, uint32_t* overflow) {
    auto meet = [] {
        if constexpr (ONE_WAVE_BLOCK) __syncthreads();
        else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    };
    const int player = st_player(s);
    const int r0 = st_roll(s, 0), r1 = st_roll(s, 1);
    const int hi = r0 > r1 ? r0 : r1, lo = r0 > r1 ? r1 : r0;   // :406-409 dice = [max, min]
    const bool dbl = hi == lo;

    // board bytes to LDS (per-lane byte reads later) and occupancy ballots
    if (lane == 0) {                                             // constant indices: `s` stays in registers
#pragma unroll
        for (int i = 0; i < 8; ++i) ((uint32_t*)sc->pts)[i] = s.w[i];
    }
    meet();
    int v = 0;
    if (lane < 24) v = (int)(int8_t)sc->pts[lane] * player;      // own-signed count
    const uint32_t own = (uint32_t)__ballot(lane < 24 && v >= 1);
    const uint32_t open = (uint32_t)__ballot(lane < 24 && v >= -1);
    const uint32_t blot = (uint32_t)__ballot(lane < 24 && v == -1);
    const uint32_t single = (uint32_t)__ballot(lane < 24 && v == 1);
    const uint32_t home_mask = player < 0 ? 0x3Fu : 0xFC0000u;
    const int bar_own = player < 0 ? st_bar(s, 0) : st_bar(s, 1);
    const bool coll = bar_own == 0 && (own & ~home_mask) == 0;   // is_collectible :638-659
    uint64_t home6;
    if (player < 0) home6 = (uint64_t)s.w[0] | ((uint64_t)(s.w[1] & 0xffffu) << 32);
    else home6 = (uint64_t)(s.w[4] >> 16) | ((uint64_t)s.w[5] << 16);

    // root candidates: low die first, then the high die (one die only for doubles after dedup)
    const uint32_t mlo = cand_mask(lo, player, own, open, home6, bar_own, coll);
    const uint32_t mhi = dbl ? 0u : cand_mask(hi, player, own, open, home6, bar_own, coll);
    const int nlo = __popc(mlo), nroots = nlo + __popc(mhi);    // <= 50 <= 64 lanes

    // one lane per root: apply it, enumerate its children with the remaining die
    int f1 = kNoMove, t1 = kNoMove, other = 0;
    bool hit1 = false;
    uint32_t cm = 0, own1 = own, blot1 = blot;
    int nseq = 0;
    if (lane < nroots) {
        const bool use_lo = lane < nlo;
        const int die = use_lo ? lo : hi;
        other = dbl ? lo : (use_lo ? hi : lo);                   // :714-716 remove the die used
        const int bit = use_lo ? nth_set_bit(mlo, lane) : nth_set_bit(mhi, lane - nlo);
        f1 = bit - 1;
        if (f1 < 0) t1 = player < 0 ? 24 - die : die - 1;
        else { t1 = f1 + player * die; if (t1 < 0 || t1 > 23) t1 = -1; }
        hit1 = t1 >= 0 && ((blot >> t1) & 1u);
        uint64_t h1 = home6;
        int bar1 = bar_own;
        if (f1 < 0) bar1 -= 1;
        else { if ((single >> f1) & 1u) own1 &= ~(1u << f1); h1 = home_add(h1, f1, -player, player); }
        if (t1 >= 0) {
            own1 |= 1u << t1;
            blot1 &= ~(1u << t1);
            h1 = home_add(h1, t1, hit1 ? 2 * player : player, player);
        }
        const bool coll1 = bar1 == 0 && (own1 & ~home_mask) == 0;
        cm = cand_mask(other, player, own1, open, h1, bar1, coll1);
        nseq = cm ? __popc(cm) : 1;                              // :740-741 childless root = 1-move play
    }
    // exclusive prefix sum of nseq over lanes
    const int incl = wave_inclusive_scan_i32(nseq);         // DPP row scans + row broadcasts (wave_ops.h), no LDS-pipe shuffles
    const int S = __builtin_amdgcn_readlane(incl, 63);
    const int base = incl - nseq;
    if (S > kSeqCap) { if (lane == 0) atomicOr(overflow, 1u); return 0; }

    // emit sequences (DFS order = roots in order, children in order, :722-750)
    if (lane < nroots) {
        DKey k1 = key_init();
        key_move(k1, f1, t1, hit1);
        if (cm == 0) {
            sc->play[base] = pack_play(f1, t1, kNoMove, kNoMove);
            sc->keyA[base] = k1.a; sc->keyB[base] = k1.b;
        } else {
            uint32_t m = cm;
            int j = 0;
            while (m) {
                const int bit = __ffs(m) - 1;
                m &= m - 1;
                const int f2 = bit - 1;
                int t2;
                if (f2 < 0) t2 = player < 0 ? 24 - other : other - 1;
                else { t2 = f2 + player * other; if (t2 < 0 || t2 > 23) t2 = -1; }
                const bool hit2 = t2 >= 0 && ((blot1 >> t2) & 1u);
                DKey k2 = k1;
                key_move(k2, f2, t2, hit2);
                sc->play[base + j] = pack_play(f1, t1, f2, t2);
                sc->keyA[base + j] = k2.a; sc->keyB[base + j] = k2.b;
                ++j;
            }
        }
    }
#ifndef DIEE_PAIR_DEDUP
#define DIEE_PAIR_DEDUP 0      // measured: +0.9 us per k_expand launch against the hash table (profiles/r03i_*): off
#endif
    if (DIEE_PAIR_DEDUP && S <= 64) {
        // At most one sequence per lane (the usual case: 30 at the opening, 12-13 plays on average): first-occurrence-wins
        // dedup (:753-774) by comparing every lane's key with the keys of the lanes before it, broadcast one after the other
        // through v_readlane -- no hash table, no LDS atomics, one barrier instead of three per round.  The 128-bit delta key
        // has 77 significant bits (a: 12 x 3 + 5 bits of counters, b: 12 x 3): three dwords, compared exactly.
        meet();
        uint64_t ka = 0, kb = 0;
        uint32_t mine = 0u;
        if (lane < S) { ka = sc->keyA[lane]; kb = sc->keyB[lane]; mine = sc->play[lane]; }
        const uint32_t w0 = (uint32_t)ka, w1 = (uint32_t)(ka >> 32) | ((uint32_t)kb << 9), w2 = (uint32_t)(kb >> 23);
        bool dup = false;
        for (int j = 0; j + 1 < S; ++j) {                            // uniform trip count
            const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)w0, j), o1 = (uint32_t)__builtin_amdgcn_readlane((int)w1, j),
                           o2 = (uint32_t)__builtin_amdgcn_readlane((int)w2, j);
            dup = dup || (j < lane && o0 == w0 && o1 == w1 && o2 == w2);
        }
        const bool keep = lane < S && !dup;
        const unsigned long long bal = __ballot(keep);
        meet();                                             // every lane holds its play: the list is rewritten in place
        if (keep) sc->play[__popcll(bal & ((1ull << lane) - 1ull))] = mine;
        meet();
        return __popcll(bal);
    }
    for (int i = lane; i < kTbl; i += 64) sc->owner[i] = kEmpty;
    for (int i = lane; i < S; i += 64) sc->status[i] = 0;
    meet();
    for (int i = lane; i < S; i += 64) {
        const uint64_t a = sc->keyA[i], b = sc->keyB[i];
        uint64_t h = (a ^ (b * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull;
        sc->slot[i] = (uint16_t)((h >> 40) & (kTbl - 1));
    }
    meet();
    // first-occurrence-wins dedup (:753-774).  Equal keys probe the same slots in the same rounds,
    // so the minimum ordinal of a key always wins the slot in the round its key first meets it.
    const uint32_t kFinal = 0x80000000u;
    for (int round = 0; round < kSeqCap; ++round) {
        int active = 0;
        for (int i = lane; i < S; i += 64) {
            if (sc->status[i] != 0) continue;
            active = 1;
            const int sl = sc->slot[i];
            if (!(sc->owner[sl] & kFinal) || sc->owner[sl] == kEmpty) atomicMin(&sc->owner[sl], (uint32_t)i);
        }
        if (!__any(active)) break;
        meet();
        for (int i = lane; i < S; i += 64) {
            if (sc->status[i] != 0) continue;
            const int sl = sc->slot[i];
            const uint32_t o = sc->owner[sl] & ~kFinal;
            if (o == (uint32_t)i) sc->status[i] = 1;
            else if (sc->keyA[o] == sc->keyA[i] && sc->keyB[o] == sc->keyB[i]) sc->status[i] = 2;
            else sc->slot[i] = (uint16_t)((sl + 1) & (kTbl - 1));
        }
        meet();
        for (int i = lane; i < S; i += 64)
            if (sc->status[i] == 1) sc->owner[sc->slot[i]] = (uint32_t)i | kFinal;
        meet();
    }
    // compact survivors in ordinal order (in place: output index <= input index)
    int k = 0;
    for (int c0 = 0; c0 < S; c0 += 64) {
        const int i = c0 + lane;
        const bool keep = i < S && sc->status[i] == 1;
        const uint32_t p = i < S ? sc->play[i] : 0u;
        const unsigned long long bal = __ballot(keep);
        const int pos = k + __popcll(bal & ((1ull << lane) - 1ull));
        meet();
        if (keep) sc->play[pos] = p;
        k += __popcll(bal);
    }
    meet();
    return k;
}

// encode, backgammon_logic.rs:262-359 (= src/backgammon/encoding.rs:6-103)
__device__ __forceinline__ int min_roll_of(int f, int t) {
    if (f == -1 && t < 6) return t + 1;
    if (f == -1 && t > 17) return 24 - t;
    if (t == -1 && f < 6) return f + 1;
    if (t == -1 && f > 17) return 24 - f;
    const int d = f - t;
    return (d < 0 ? -d : d) & 0xff;
}
__device__ __forceinline__ uint32_t bg_encode_dev(int r0, int r1, uint32_t play) {
    const int f1 = play_f1(play), t1 = play_t1(play), f2 = play_f2(play), t2 = play_t2(play);
    if (f1 == kNoMove) return 1351u;
    const int low = r0 > r1 ? r1 : r0;
    const bool two = f2 != kNoMove;
    const int m1 = min_roll_of(f1, t1), m2 = two ? min_roll_of(f2, t2) : 0;
    bool low_first = false, low_second = false;
    uint32_t sum = 0;
    if (f1 == -1 && (t1 < 6 || t1 > 17)) { sum += 24; low_first = m1 == low; }
    else if (t1 == -1 && (f1 < 6 || f1 > 17)) { sum += (uint32_t)f1; }
    else { sum += (uint32_t)f1; low_first = m1 == low; }
    if (two) {
        if (f2 == -1 && (t2 < 6 || t2 > 17)) { sum += 26u * 24u; low_second = m2 == low; }
        else if (t2 == -1 && (f2 < 6 || f2 > 17)) { sum += 26u * (uint32_t)f2; }
        else { sum += 26u * (uint32_t)f2; low_second = m2 == low; }
    } else {
        low_first = false;
        sum += 26u * 25u;
    }
    bool high_first;
    if (low_first) high_first = false;
    else if (low_second) high_first = true;
    else if (m2 != 0) high_first = m1 >= m2;
    else high_first = m1 > low;
    return high_first ? sum : sum + 676u;
}

// decode, backgammon_logic.rs:361-401
__device__ __forceinline__ uint32_t bg_decode_dev(int r0, int r1, int player, uint32_t action) {
    if (action == 1351u) return pack_play(kNoMove, kNoMove, kNoMove, kNoMove);
    const bool high_first = action < 676u;
    const uint32_t v = high_first ? action : action - 676u;
    int f1 = (int)(v % 26u), f2 = (int)(v / 26u);
    const bool single = f2 == 25;
    const int hi = r0 > r1 ? r0 : r1, lo = r0 > r1 ? r1 : r0;
    if (f1 == 24 && player == 1) f1 = -1;
    if (f2 == 24 && player == 1) f2 = -1;
    int t1, t2;
    if (high_first) { t1 = (int)(int8_t)(f1 + hi * player); t2 = (int)(int8_t)(f2 + lo * player); }
    else { t1 = (int)(int8_t)(f1 + lo * player); t2 = (int)(int8_t)(f2 + hi * player); }
    if (t1 >= 24 || t1 <= -1) t1 = -1;
    if (t2 >= 24 || t2 <= -1) t2 = -1;
    if (f1 == 24) f1 = -1;
    if (f2 == 24) f2 = -1;
    return single ? pack_play(f1, t1, kNoMove, kNoMove) : pack_play(f1, t1, f2, t2);
}

// byte-wise add of a small delta to one point of a board held as 6 words (no carries across bytes).
// Every word is rewritten through a mask: a conditional `if (word == i) w[i] = ...` chain is turned by the compiler
// into ONE dynamically indexed read-modify-write, which puts the state into scratch memory (a global-memory round
// trip per access; measured: half of k_expand's time).
__device__ __forceinline__ void pts_add(uint32_t (&w)[8], int point, int delta) {
    const int wi = point >> 2, sh = 8 * (point & 3);
    const uint32_t d = ((uint32_t)delta & 0xffu) << sh;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint32_t m = wi == i ? 0xffu << sh : 0u;
        w[i] = (w[i] & ~m) | ((w[i] + d) & m);      // the byte's own sum; carries out of it fall outside the mask
    }
}
__device__ __forceinline__ int pts_get(const uint32_t (&w)[8], int point) {
    const int wi = point >> 2;
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) x |= wi == i ? w[i] : 0u;
    return (int)(int8_t)(x >> (8 * (point & 3)));
}

// get_next_state for one (from,to), backgammon_logic.rs:467-517
__device__ __forceinline__ void bg_move_one(BgState& s, int f, int t, int player) {
    const int me = player < 0 ? 0 : 1, opp = 1 - me;
    if (t == -1) {                                        // collect :470-480
        pts_add(s.w, f, -player);
        s.w[6] += 1u << (16 + 8 * me);
        return;
    }
    const bool hit = pts_get(s.w, t) == -player;
    if (f == -1) {                                        // from the bar :482-500
        s.w[6] -= 1u << (8 * me);
        if (hit) { pts_add(s.w, t, 2 * player); s.w[6] += 1u << (8 * opp); }
        else pts_add(s.w, t, player);
    } else {
        pts_add(s.w, f, -player);
        if (hit) { pts_add(s.w, t, 2 * player); s.w[6] += 1u << (8 * opp); }   // :501-509
        else pts_add(s.w, t, player);                                           // :510-514
    }
}

// apply_move, backgammon_logic.rs:176-186 (d0,d1 = what roll_die would draw)
__device__ __forceinline__ void bg_apply_dev(BgState& s, uint32_t play, int d0, int d1) {
    const int player = st_player(s);
    const int f1 = play_f1(play), t1 = play_t1(play), f2 = play_f2(play), t2 = play_t2(play);
    if (f1 != kNoMove) bg_move_one(s, f1, t1, player);
    if (f2 != kNoMove) bg_move_one(s, f2, t2, player);
    const int r0 = st_roll(s, 0), r1 = st_roll(s, 1);
    if (r0 == r1 && !st_second(s)) {
        s.w[7] = (s.w[7] & 0x00ffffffu) | (1u << 24);
    } else {
        s.w[7] = (uint32_t)d0 | ((uint32_t)d1 << 8) | ((uint32_t)((-player) & 0xff) << 16);
    }
}
// skip_turn, backgammon_logic.rs:192-196
__device__ __forceinline__ void bg_skip_dev(BgState& s, int d0, int d1) {
    const int player = st_player(s);
    s.w[7] = (uint32_t)d0 | ((uint32_t)d1 << 8) | ((uint32_t)((-player) & 0xff) << 16);
}
// check_winner, backgammon_logic.rs:106-108,527-534: 0 none, else -1/+1 (player -1 checked first)
__device__ __forceinline__ int bg_winner_dev(const BgState& s) {
    if (st_off(s, 0) == 15) return -1;
    if (st_off(s, 1) == 15) return 1;
    return 0;
}
// as_tensor, backgammon_logic.rs:198-252: plane c at point p (0..23)
__device__ __forceinline__ float bg_plane_dev(const BgState& s, int c, int p) {
    switch (c) {
        case 0: return (float)pts_get(s.w, p);
        case 1: return (float)st_player(s);
        case 2: return (float)st_bar(s, p < 12 ? 0 : 1);
        case 3: return (float)st_off(s, p < 12 ? 0 : 1);
        case 4: return (float)st_roll(s, p < 12 ? 0 : 1);
        default: return st_second(s) ? 1.0f : 0.0f;
    }
}

// ---- counter-based RNG (Philox4x32-10) and derived draws ---------------------------------------
__host__ __device__ inline void philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                           uint32_t c2, uint32_t c3, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
constexpr uint32_t kTagInitRoll = 0xFFFFFFFFu, kTagMoveRoll = 0xFFFFFFFEu, kTagSample = 0xFFFFFFFDu,
                   kTagDirichlet = 0xFFFFFFFCu;
// roll_die, backgammon_logic.rs:100-104: two iid uniform 1..=6
__host__ __device__ inline void draw_dice(uint64_t seed, uint32_t game, uint32_t round, uint32_t tag,
                                          uint32_t ord, int& d0, int& d1) {
    uint32_t o[4];
    philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), game, round, tag, ord, o);
    d0 = 1 + (int)(((uint64_t)o[0] * 6ull) >> 32);
    d1 = 1 + (int)(((uint64_t)o[1] * 6ull) >> 32);
}
__host__ __device__ inline double draw_uniform(uint64_t seed, uint32_t game, uint32_t round, uint32_t tag,
                                               uint32_t ord) {
    uint32_t o[4];
    philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), game, round, tag, ord, o);
    const uint64_t x = ((uint64_t)o[3] << 32) | o[2];
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

// x^y, x in [0,1], y > 0, from IEEE +,-,*,/ only (stands in for Tensor::pow_, alpha_parallel.rs:165;
// identical results on host and device; compile with -ffp-contract=off)
__host__ __device__ inline float det_powf(float xf, float yf) {
    if (xf <= 0.0f) return 0.0f;
    if (xf == 1.0f) return 1.0f;
    const double x = (double)xf, y = (double)yf;
    union { double d; uint64_t u; } cv;
    cv.d = x;
    int e = (int)((cv.u >> 52) & 0x7ff) - 1023;
    cv.u = (cv.u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = cv.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
    double s = 1.0 / 21.0;
    s = s * t2 + 1.0 / 19.0; s = s * t2 + 1.0 / 17.0; s = s * t2 + 1.0 / 15.0;
    s = s * t2 + 1.0 / 13.0; s = s * t2 + 1.0 / 11.0; s = s * t2 + 1.0 / 9.0;
    s = s * t2 + 1.0 / 7.0;  s = s * t2 + 1.0 / 5.0;  s = s * t2 + 1.0 / 3.0;
    s = s * t2 + 1.0;
    const double lg2 = (double)e + (2.0 * t * s) * 1.4426950408889634;
    const double z = y * lg2;
    if (z < -160.0) return 0.0f;
    const double kf = (double)(long long)(z - 0.5);
    const double f = (z - kf) * 0.6931471805599453;
    double p = 1.0 / 6227020800.0;
    p = p * f + 1.0 / 479001600.0; p = p * f + 1.0 / 39916800.0; p = p * f + 1.0 / 3628800.0;
    p = p * f + 1.0 / 362880.0;    p = p * f + 1.0 / 40320.0;    p = p * f + 1.0 / 5040.0;
    p = p * f + 1.0 / 720.0;       p = p * f + 1.0 / 120.0;      p = p * f + 1.0 / 24.0;
    p = p * f + 1.0 / 6.0;         p = p * f + 0.5;              p = p * f + 1.0;
    p = p * f + 1.0;
    const int k = (int)kf;
    cv.u = (uint64_t)(k + 1023) << 52;
    return (float)(p * cv.d);
}

}  // namespace diee
