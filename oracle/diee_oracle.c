/*
 * diee_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain-C restatement of alibasaran/die-e's batched self-play hot path.  Written for
 * clarity and literal fidelity to the Rust source, not speed.  Each function cites the
 * reference file:line it follows (paths relative to the reference repo root).
 *
 * Parity pin: see diee_oracle.h.  Compile with -ffp-contract=off (Makefile does).
 */
#include "diee_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>

/* ======================================================================== */
/* Backgammon rules                                                          */
/* ======================================================================== */

/* Backgammon::new, src/backgammon/backgammon_logic.rs:80-94 */
void or_bg_new(or_bg_state* s) {
    static const int8_t start[24] = {2, 0, 0, 0, 0, -5, 0, -3, 0, 0, 0, 5,
                                     -5, 0, 0, 0, 3, 0, 5, 0, 0, 0, 0, -2};
    memset(s, 0, sizeof *s);
    memcpy(s->pts, start, 24);
    s->player = -1;
}

/* get_next_state for a list of (from,to) moves, backgammon_logic.rs:467-517.
 * Operates on the Board part (pts, bar, off) of the state only. */
void or_bg_next_state(or_bg_state* st, const int8_t (*moves)[2], int n, int player) {
    for (int i = 0; i < n; ++i) {
        int from = moves[i][0], to = moves[i][1];
        if (to == -1) {                       /* :470-480 collecting a checker */
            st->pts[from] -= (int8_t)player;
            if (player == -1) st->off[0] += 1; else st->off[1] += 1;
            continue;
        }
        if (from == -1) {                     /* :482-500 playing from the bar */
            if (st->pts[to] == -player) {     /* hitting */
                st->pts[to] = (int8_t)player;
                if (player == -1) { st->bar[1] += 1; st->bar[0] -= 1; }
                else              { st->bar[0] += 1; st->bar[1] -= 1; }
            } else if (player == -1) {
                st->pts[to] -= 1; st->bar[0] -= 1;
            } else {
                st->pts[to] += 1; st->bar[1] -= 1;
            }
        } else if (st->pts[to] == -player) {  /* :501-509 hit */
            st->pts[to] = (int8_t)player;
            st->pts[from] -= (int8_t)player;
            if (player == -1) st->bar[1] += 1; else st->bar[0] += 1;
        } else {                              /* :510-514 plain move */
            st->pts[to] += (int8_t)player;
            st->pts[from] -= (int8_t)player;
        }
    }
}

/* is_collectible, backgammon_logic.rs:638-659 */
int or_bg_is_collectible(const or_bg_state* b, int player) {
    if (player == -1) {
        if (b->bar[0] != 0) return 0;
        for (int i = 6; i <= 23; ++i) if (b->pts[i] < 0) return 0;
    } else if (player == 1) {
        if (b->bar[1] != 0) return 0;
        for (int i = 0; i <= 17; ++i) if (b->pts[i] > 0) return 0;
    }
    return 1;
}

/* check_winner -> check_win_without_player, backgammon_logic.rs:106-108, 527-534 */
int or_bg_check_winner(const or_bg_state* b, int* winner) {
    if (b->off[0] == 15) { *winner = -1; return 1; }
    if (b->off[1] == 15) { *winner = 1; return 1; }
    return 0;
}

typedef struct { int8_t m, from, to; } cand_t;

static int cand_cmp(const void* a, const void* b) {      /* tuple order (m,(from,to)), :619 */
    const cand_t* x = a; const cand_t* y = b;
    if (x->m != y->m) return x->m < y->m ? -1 : 1;
    if (x->from != y->from) return x->from < y->from ? -1 : 1;
    if (x->to != y->to) return x->to < y->to ? -1 : 1;
    return 0;
}

static int cand_sort_dedup(cand_t* c, int n) {           /* sort_unstable + dedup, :619-620 */
    qsort(c, (size_t)n, sizeof *c, cand_cmp);
    int w = 0;
    for (int i = 0; i < n; ++i)
        if (w == 0 || cand_cmp(&c[w - 1], &c[i]) != 0) c[w++] = c[i];
    return w;
}

static int action_trees_rec(const uint8_t* dice, int nd, const or_bg_state* b, int player,
                            int depth, or_tree_node* out, int cap, int pos);

/* shared tail of get_normal_moves / get_entry_moves: one ActionNode per candidate, children =
 * _get_children_of_node_action (backgammon_logic.rs:622-635, 688-700, 705-720) */
static int emit_nodes(const cand_t* c, int nc, const uint8_t* dice, int nd, const or_bg_state* b,
                      int player, int depth, or_tree_node* out, int cap, int pos) {
    for (int i = 0; i < nc; ++i) {
        if (pos < cap) { out[pos].depth = (int8_t)depth; out[pos].from = c[i].from; out[pos].to = c[i].to; }
        pos++;
        /* :712 new_state = get_next_state(state, [action]) */
        or_bg_state nb = *b;
        int8_t mv[1][2] = {{c[i].from, c[i].to}};
        or_bg_next_state(&nb, mv, 1, player);
        /* :714-716 remove the FIRST occurrence of the die used */
        uint8_t nd2[8]; int k = 0, removed = 0;
        for (int j = 0; j < nd; ++j) {
            if (!removed && dice[j] == (uint8_t)c[i].m) { removed = 1; continue; }
            nd2[k++] = dice[j];
        }
        pos = action_trees_rec(nd2, k, &nb, player, depth + 1, out, cap, pos);
    }
    return pos;
}

/* get_normal_moves, backgammon_logic.rs:555-636 */
static int normal_moves_rec(const uint8_t* dice, int nd, const or_bg_state* b, int player,
                            int depth, or_tree_node* out, int cap, int pos) {
    cand_t c[256]; int nc = 0;
    const int8_t* board = b->pts;
    /* :562-580 player -1 bear-off */
    if (player == -1 && or_bg_is_collectible(b, player)) {
        for (int di = 0; di < nd; ++di) {
            int m = dice[di];
            int point = m - 1;
            if (board[point] < 0) { c[nc].m = (int8_t)m; c[nc].from = (int8_t)point; c[nc].to = -1; nc++; }
            for (int m_idx = point - 1; m_idx >= 0; --m_idx) {     /* (0..point).rev() */
                int left_sum = 0;
                for (int j = m_idx + 1; j < 6; ++j) left_sum += board[j];
                left_sum = (int8_t)left_sum;                     /* i8 sum; cannot overflow with 15 checkers */
                if (board[m_idx] < 0 && left_sum >= 0) {
                    c[nc].m = (int8_t)m; c[nc].from = (int8_t)m_idx; c[nc].to = -1; nc++;
                    break;
                }
            }
        }
    } else if (player == 1 && or_bg_is_collectible(b, player)) {   /* :581-598 */
        for (int di = 0; di < nd; ++di) {
            int m = dice[di];
            int point = 24 - m;
            if (board[point] > 0) { c[nc].m = (int8_t)m; c[nc].from = (int8_t)point; c[nc].to = -1; nc++; }
            for (int m_idx = point; m_idx <= 23; ++m_idx) {
                int left_sum = 0;
                for (int j = 18; j < m_idx; ++j) left_sum += board[j];
                if (board[m_idx] > 0 && left_sum <= 0) {
                    c[nc].m = (int8_t)m; c[nc].from = (int8_t)m_idx; c[nc].to = -1; nc++;
                    break;
                }
            }
        }
    }
    /* :600-617 normal moves (still allowed while bearing off, comment :561) */
    for (int di = 0; di < nd; ++di) {
        int m = dice[di];
        for (int point = 0; point < 24; ++point) {
            int n_pieces = board[point];
            if (player == -1 && n_pieces <= player && point - m >= 0 && board[point - m] <= 1) {
                c[nc].m = (int8_t)m; c[nc].from = (int8_t)point; c[nc].to = (int8_t)(point - m); nc++;
            } else if (player == 1 && n_pieces >= player && point + m <= 23 && board[point + m] >= -1) {
                c[nc].m = (int8_t)m; c[nc].from = (int8_t)point; c[nc].to = (int8_t)(point + m); nc++;
            }
        }
    }
    nc = cand_sort_dedup(c, nc);
    return emit_nodes(c, nc, dice, nd, b, player, depth, out, cap, pos);
}

/* get_entry_moves, backgammon_logic.rs:662-703 */
static int entry_moves_rec(const uint8_t* dice, int nd, const or_bg_state* b, int player,
                           int depth, or_tree_node* out, int cap, int pos) {
    cand_t c[16]; int nc = 0;
    const int8_t* board = b->pts;
    if (player == -1) {
        for (int di = 0; di < nd; ++di) {
            int m = dice[di], point = 24 - m;
            if (board[point] < 2) { c[nc].m = (int8_t)m; c[nc].from = -1; c[nc].to = (int8_t)point; nc++; }
        }
    } else if (player == 1) {
        for (int di = 0; di < nd; ++di) {
            int m = dice[di], point = m - 1;
            if (board[point] > -2) { c[nc].m = (int8_t)m; c[nc].from = -1; c[nc].to = (int8_t)point; nc++; }
        }
    }
    nc = cand_sort_dedup(c, nc);
    return emit_nodes(c, nc, dice, nd, b, player, depth, out, cap, pos);
}

/* _get_action_trees, backgammon_logic.rs:544-552 */
static int action_trees_rec(const uint8_t* dice, int nd, const or_bg_state* b, int player,
                            int depth, or_tree_node* out, int cap, int pos) {
    int hit = player == -1 ? b->bar[0] : b->bar[1];      /* get_pieces_hit :536-542 */
    if (hit > 0) return entry_moves_rec(dice, nd, b, player, depth, out, cap, pos);
    return normal_moves_rec(dice, nd, b, player, depth, out, cap, pos);
}

int or_bg_normal_moves(const uint8_t* dice, int nd, const or_bg_state* b, int player,
                       or_tree_node* out, int cap) {
    return normal_moves_rec(dice, nd, b, player, 0, out, cap, 0);
}
int or_bg_entry_moves(const uint8_t* dice, int nd, const or_bg_state* b, int player,
                      or_tree_node* out, int cap) {
    return entry_moves_rec(dice, nd, b, player, 0, out, cap, 0);
}
int or_bg_action_trees(const uint8_t* dice, int nd, const or_bg_state* b, int player,
                       or_tree_node* out, int cap) {
    return action_trees_rec(dice, nd, b, player, 0, out, cap, 0);
}

/* extract_sequences_list/node/helper, backgammon_logic.rs:722-750: roots in order, children in
 * order, a node without children terminates a sequence. */
int or_bg_extract_sequences(const or_tree_node* t, int n, or_seq* out, int cap) {
    int8_t path[4][2]; int ns = 0;
    for (int i = 0; i < n; ++i) {
        int d = t[i].depth;
        assert(d < 4);
        path[d][0] = t[i].from; path[d][1] = t[i].to;
        int leaf = (i + 1 >= n) || (t[i + 1].depth <= d);
        if (leaf) {
            if (ns < cap) {
                out[ns].n = (int8_t)(d + 1);
                memset(out[ns].mv, OR_NO_MOVE, sizeof out[ns].mv);
                for (int k = 0; k <= d; ++k) { out[ns].mv[k][0] = path[k][0]; out[ns].mv[k][1] = path[k][1]; }
            }
            ns++;
        }
    }
    return ns;
}

static int board_eq(const or_bg_state* a, const or_bg_state* b) {     /* Board = (pts, bar, off) */
    return memcmp(a->pts, b->pts, 24) == 0 && a->bar[0] == b->bar[0] && a->bar[1] == b->bar[1] &&
           a->off[0] == b->off[0] && a->off[1] == b->off[1];
}

/* remove_duplicate_states, backgammon_logic.rs:753-774: keep the FIRST sequence reaching each board */
int or_bg_remove_duplicate_states(const or_bg_state* initial, const or_seq* seqs, int n, int player,
                                  or_seq* out) {
    or_bg_state* seen = malloc(sizeof(or_bg_state) * (size_t)(n > 0 ? n : 1));
    int nseen = 0, nout = 0;
    for (int i = 0; i < n; ++i) {
        or_bg_state cur = *initial;
        for (int k = 0; k < seqs[i].n; ++k) {
            int8_t mv[1][2] = {{seqs[i].mv[k][0], seqs[i].mv[k][1]}};
            or_bg_next_state(&cur, mv, 1, player);
        }
        int dup = 0;
        for (int j = 0; j < nseen; ++j) if (board_eq(&seen[j], &cur)) { dup = 1; break; }
        if (!dup) { seen[nseen++] = cur; out[nout++] = seqs[i]; }
    }
    free(seen);
    return nout;
}

/* get_valid_moves, backgammon_logic.rs:403-414 */
int or_bg_valid_moves_seq(const or_bg_state* s, or_seq* out, int cap, int* n_before_dedup) {
    assert(!(s->roll[0] == 0 && s->roll[1] == 0));                  /* :404 */
    uint8_t dice[2];
    if (s->roll[0] > s->roll[1]) { dice[0] = s->roll[0]; dice[1] = s->roll[1]; }   /* :406-409 */
    else                         { dice[0] = s->roll[1]; dice[1] = s->roll[0]; }
    static const int TCAP = 8192;
    or_tree_node* tree = malloc(sizeof(or_tree_node) * (size_t)TCAP);
    int nt = or_bg_action_trees(dice, 2, s, s->player, tree, TCAP);
    assert(nt <= TCAP);
    or_seq* seqs = malloc(sizeof(or_seq) * (size_t)(nt > 0 ? nt : 1));
    int ns = or_bg_extract_sequences(tree, nt, seqs, nt);
    if (n_before_dedup) *n_before_dedup = ns;
    or_seq* uniq = malloc(sizeof(or_seq) * (size_t)(ns > 0 ? ns : 1));
    int nu = or_bg_remove_duplicate_states(s, seqs, ns, s->player, uniq);
    for (int i = 0; i < nu && i < cap; ++i) out[i] = uniq[i];
    free(tree); free(seqs); free(uniq);
    return nu;
}

int or_bg_valid_moves(const or_bg_state* s, or_play* out, int cap) {
    or_seq* seqs = malloc(sizeof(or_seq) * OR_MAX_PLAYS);
    int n = or_bg_valid_moves_seq(s, seqs, OR_MAX_PLAYS, NULL);
    assert(n <= OR_MAX_PLAYS);
    for (int i = 0; i < n && i < cap; ++i) {
        assert(seqs[i].n <= 2);
        out[i].mv[0] = seqs[i].mv[0][0]; out[i].mv[1] = seqs[i].mv[0][1];
        out[i].mv[2] = seqs[i].n > 1 ? seqs[i].mv[1][0] : OR_NO_MOVE;
        out[i].mv[3] = seqs[i].n > 1 ? seqs[i].mv[1][1] : OR_NO_MOVE;
    }
    free(seqs);
    return n;
}

/* apply_move, backgammon_logic.rs:176-186.  (d0,d1) = the dice roll_die would produce. */
void or_bg_apply_move(or_bg_state* s, const or_play* p, uint8_t d0, uint8_t d1) {
    int n = or_play_len(p);
    int8_t mv[2][2] = {{p->mv[0], p->mv[1]}, {p->mv[2], p->mv[3]}};
    or_bg_next_state(s, mv, n, s->player);
    if (s->roll[0] == s->roll[1] && !s->second) {
        s->second = 1;
    } else {
        s->second = 0;
        s->player = (int8_t)(-s->player);
        s->roll[0] = d0; s->roll[1] = d1;
    }
}

/* skip_turn, backgammon_logic.rs:192-196 */
void or_bg_skip_turn(or_bg_state* s, uint8_t d0, uint8_t d1) {
    s->second = 0;
    s->player = (int8_t)(-s->player);
    s->roll[0] = d0; s->roll[1] = d1;
}

/* encode, backgammon_logic.rs:262-359 (byte-identical copy at src/backgammon/encoding.rs:6-103) */
uint32_t or_bg_encode(const or_bg_state* s, const or_play* p) {
    int n = or_play_len(p);
    assert(n <= 2);
    if (n == 0) return 1351;                                         /* :266-268 */
    int r0 = s->roll[0], r1 = s->roll[1];
    int low_roll = r0 > r1 ? r1 : r0;                                /* :272 */
    int low_first = 0, low_second = 0;
    int minimum_rolls[2] = {0, 0};
    for (int i = 0; i < n; ++i) {                                    /* :277-285 */
        int f = p->mv[2 * i], t = p->mv[2 * i + 1], v;
        if (f == -1 && t < 6) v = t + 1;
        else if (f == -1 && t > 17) v = 24 - t;
        else if (t == -1 && f < 6) v = f + 1;
        else if (t == -1 && f > 17) v = 24 - f;
        else v = abs(f - t);
        minimum_rolls[i] = v & 0xff;                                 /* as u8 */
    }
    /* :288 single move => second minimum roll 0 (already) */
    uint32_t sum = 0;
    for (int i = 0; i < n; ++i) {                                    /* :299-349 */
        int f = p->mv[2 * i], t = p->mv[2 * i + 1];
        uint32_t w = i == 0 ? 1u : 26u;
        int* flag = i == 0 ? &low_first : &low_second;
        if (f == -1 && t < 6)       { sum += w * 24; *flag = ((t + 1) & 0xff) == low_roll; }
        else if (f == -1 && t > 17) { sum += w * 24; *flag = ((24 - t) & 0xff) == low_roll; }
        else if (t == -1 && f < 6)  { sum += w * (uint32_t)f; }
        else if (t == -1 && f > 17) { sum += w * (uint32_t)f; }
        else                        { sum += w * (uint32_t)f; *flag = minimum_rolls[i] == low_roll; }
    }
    if (n == 1) { low_first = 0; sum += 26 * 25; }                   /* :352 */
    int high_first;                                                  /* :355 */
    if (low_first) high_first = 0;
    else if (low_second) high_first = 1;
    else if (minimum_rolls[1] != 0) high_first = minimum_rolls[0] >= minimum_rolls[1];
    else high_first = minimum_rolls[0] > low_roll;
    return high_first ? sum : sum + 676;                             /* :358 */
}

/* decode, backgammon_logic.rs:361-401 */
void or_bg_decode(const or_bg_state* s, uint32_t action, or_play* out) {
    out->mv[0] = out->mv[1] = out->mv[2] = out->mv[3] = OR_NO_MOVE;
    if (action == 1351) return;
    int player = s->player;
    int high_first = action < 676;
    uint32_t v = high_first ? action : action - 676;
    int from1 = (int)(v % 26), from2 = (int)(v / 26);
    int single = from2 == 25;
    int hi = s->roll[0] > s->roll[1] ? s->roll[0] : s->roll[1];
    int lo = s->roll[0] > s->roll[1] ? s->roll[1] : s->roll[0];
    int f1 = (int8_t)from1, f2 = (int8_t)from2;
    if (f1 == 24 && player == 1) f1 = -1;                            /* :384-385 */
    if (f2 == 24 && player == 1) f2 = -1;
    int to1, to2;
    if (high_first) { to1 = (int8_t)(f1 + hi * player); to2 = (int8_t)(f2 + lo * player); }
    else            { to1 = (int8_t)(f1 + lo * player); to2 = (int8_t)(f2 + hi * player); }
    if (to1 >= 24 || to1 <= -1) to1 = -1;                            /* :395-398 */
    if (to2 >= 24 || to2 <= -1) to2 = -1;
    if (f1 == 24) f1 = -1;
    if (f2 == 24) f2 = -1;
    out->mv[0] = (int8_t)f1; out->mv[1] = (int8_t)to1;
    if (!single) { out->mv[2] = (int8_t)f2; out->mv[3] = (int8_t)to2; }
}

/* as_tensor, backgammon_logic.rs:198-252: out[c*24 + p], p = 6*row + col = point index */
void or_bg_planes(const or_bg_state* s, float* out) {
    for (int p = 0; p < 24; ++p) {
        out[0 * 24 + p] = (float)s->pts[p];
        out[1 * 24 + p] = (float)s->player;
        out[2 * 24 + p] = (float)(p < 12 ? s->bar[0] : s->bar[1]);
        out[3 * 24 + p] = (float)(p < 12 ? s->off[0] : s->off[1]);
        out[4 * 24 + p] = (float)(p < 12 ? s->roll[0] : s->roll[1]);
        out[5 * 24 + p] = s->second ? 1.0f : 0.0f;
    }
}

/* ======================================================================== */
/* Tic-tac-toe, src/tictactoe/mod.rs                                         */
/* ======================================================================== */
void or_ttt_new(or_ttt_state* s) { memset(s, 0, sizeof *s); s->player = -1; }      /* :28-30 */
int or_ttt_valid_moves(const or_ttt_state* s, uint8_t* out) {                      /* :36-44 */
    int n = 0;
    for (int i = 0; i < 9; ++i) if (s->board[i] == 0) out[n++] = (uint8_t)i;
    return n;
}
void or_ttt_apply_move(or_ttt_state* s, uint8_t a) {                               /* :46-49 */
    s->board[a] = s->player; s->player = (int8_t)(-s->player);
}
int or_ttt_check_winner(const or_ttt_state* s, int* winner) {                      /* :59-81 */
    static const int comb[8][3] = {{0,1,2},{3,4,5},{6,7,8},{0,3,6},{1,4,7},{2,5,8},{0,4,8},{2,4,6}};
    for (int i = 0; i < 8; ++i) {
        int a = s->board[comb[i][0]], b = s->board[comb[i][1]], c = s->board[comb[i][2]];
        if (a != 0 && a == b && b == c) { *winner = a; return 1; }
    }
    int full = 1;
    for (int i = 0; i < 9; ++i) if (s->board[i] == 0) full = 0;
    if (full) { *winner = 0; return 1; }
    return 0;
}
void or_ttt_planes(const or_ttt_state* s, float* out) {                            /* :83-94 */
    for (int i = 0; i < 9; ++i) {
        out[0 * 9 + i] = s->board[i] == -1 ? 1.0f : 0.0f;
        out[1 * 9 + i] = s->board[i] == 0 ? 1.0f : 0.0f;
        out[2 * 9 + i] = s->board[i] == 1 ? 1.0f : 0.0f;
    }
}

/* ======================================================================== */
/* Deterministic helpers (same algorithms, independently coded, in the HIP path) */
/* ======================================================================== */

/* Philox4x32-10 (Salmon et al., SC'11).  The reference uses unseeded thread_rng
 * (backgammon_logic.rs:100-104, noise.rs:16,30, alphazero.rs:130): distribution parity only.
 * A counter-based generator makes CPU<->GPU runs reproducible. */
void or_philox4x32(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* roll_die, backgammon_logic.rs:100-104: two iid uniform 1..=6 */
void or_dice(uint64_t seed, uint32_t game, uint32_t round, uint32_t tag, uint32_t ord,
             uint8_t* d0, uint8_t* d1) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, ctr[4] = {game, round, tag, ord}, o[4];
    or_philox4x32(key, ctr, o);
    *d0 = (uint8_t)(1 + (((uint64_t)o[0] * 6) >> 32));
    *d1 = (uint8_t)(1 + (((uint64_t)o[1] * 6) >> 32));
}

double or_uniform01(uint64_t seed, uint32_t game, uint32_t round, uint32_t tag, uint32_t ord) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, ctr[4] = {game, round, tag, ord}, o[4];
    or_philox4x32(key, ctr, o);
    uint64_t x = ((uint64_t)o[3] << 32) | o[2];
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

/* x^y for x in [0,1], y > 0 using only IEEE +,-,*,/ in double so that CPU and GPU agree bit for
 * bit (libm / ocml powf differ in the last ulp).  Stands in for Tensor::pow_ at
 * alpha_parallel.rs:165. */
float or_det_powf(float xf, float yf) {
    if (xf <= 0.0f) return 0.0f;
    if (xf == 1.0f) return 1.0f;
    double x = (double)xf, y = (double)yf;
    uint64_t bits; memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m; memcpy(&m, &bits, 8);                       /* m in [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double t = (m - 1.0) / (m + 1.0), t2 = t * t;
    double s = 1.0 / 21.0;                                /* ln(m) = 2 t (1 + t^2/3 + ... + t^20/21) */
    s = s * t2 + 1.0 / 19.0; s = s * t2 + 1.0 / 17.0; s = s * t2 + 1.0 / 15.0;
    s = s * t2 + 1.0 / 13.0; s = s * t2 + 1.0 / 11.0; s = s * t2 + 1.0 / 9.0;
    s = s * t2 + 1.0 / 7.0;  s = s * t2 + 1.0 / 5.0;  s = s * t2 + 1.0 / 3.0;
    s = s * t2 + 1.0;
    double lg2 = (double)e + (2.0 * t * s) * 1.4426950408889634;
    double z = y * lg2;                                   /* <= 0 */
    if (z < -160.0) return 0.0f;
    double kf = (double)(long long)(z - 0.5);             /* round to nearest for z <= 0 (trunc toward 0) */
    double f = (z - kf) * 0.6931471805599453;             /* |f| <= 0.35 */
    double p = 1.0 / 6227020800.0;                        /* exp(f), Taylor to f^13 */
    p = p * f + 1.0 / 479001600.0; p = p * f + 1.0 / 39916800.0; p = p * f + 1.0 / 3628800.0;
    p = p * f + 1.0 / 362880.0;    p = p * f + 1.0 / 40320.0;    p = p * f + 1.0 / 5040.0;
    p = p * f + 1.0 / 720.0;       p = p * f + 1.0 / 120.0;      p = p * f + 1.0 / 24.0;
    p = p * f + 1.0 / 6.0;         p = p * f + 0.5;              p = p * f + 1.0;
    p = p * f + 1.0;
    int k = (int)kf;
    uint64_t sb = (uint64_t)(k + 1023) << 52;             /* k >= -161: normal double */
    double sc; memcpy(&sc, &sb, 8);
    return (float)(p * sc);
}

/* Dirichlet(alpha * 1_n) restating rand_distr 0.4.3 (Cargo.toml:19, unvendored): each component
 * Gamma(alpha,1) normalised; Gamma for shape < 1 = Gamma(shape+1) * U^(1/shape); Gamma for shape
 * >= 1 by Marsaglia-Tsang.  Normals by Box-Muller (rand_distr uses a ziggurat: distribution
 * parity only).  Host-side in both oracle and product, like noise.rs:27-34. */
typedef struct { uint32_t key[2]; uint32_t step; uint32_t n; } dir_rng;
static double dir_u01(dir_rng* r) {                       /* open interval (0,1) */
    uint32_t ctr[4] = {r->n++, r->step, OR_TAG_DIRICHLET, 0}, o[4];
    or_philox4x32(r->key, ctr, o);
    uint64_t x = ((uint64_t)o[1] << 32) | o[0];
    return ((double)(x >> 12) + 0.5) * (1.0 / 4503599627370496.0);
}
static double dir_normal(dir_rng* r) {
    double u1 = dir_u01(r), u2 = dir_u01(r);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static double dir_gamma_large(dir_rng* r, double shape) {
    double d = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x = dir_normal(r), vc = 1.0 + c * x;
        if (vc <= 0.0) continue;
        double v = vc * vc * vc, u = dir_u01(r), x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2 || log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return d * v;
    }
}
void or_dirichlet(uint64_t seed, uint32_t step, float alpha, int n, float* out) {
    dir_rng r = {{(uint32_t)seed, (uint32_t)(seed >> 32)}, step, 0};
    double a = (double)alpha, sum = 0.0;
    double* g = malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        double v;
        if (a < 1.0) { double u = dir_u01(&r); v = dir_gamma_large(&r, a + 1.0) * pow(u, 1.0 / a); }
        else v = dir_gamma_large(&r, a);
        g[i] = v; sum += v;
    }
    for (int i = 0; i < n; ++i) out[i] = (float)(g[i] / sum);
    free(g);
}

/* ======================================================================== */
/* Game vtables (trait LearnableGame, src/base.rs:8-51)                      */
/* ======================================================================== */
static void bgv_new(or_state* s) { or_bg_new((or_bg_state*)s); }
static int bgv_valid(const or_state* s, or_play* o, int cap) { return or_bg_valid_moves((const or_bg_state*)s, o, cap); }
static void bgv_apply(or_state* s, const or_play* p, uint8_t a, uint8_t b) { or_bg_apply_move((or_bg_state*)s, p, a, b); }
static void bgv_skip(or_state* s, uint8_t a, uint8_t b) { or_bg_skip_turn((or_bg_state*)s, a, b); }
static int bgv_player(const or_state* s) { return ((const or_bg_state*)s)->player; }
static int bgv_winner(const or_state* s, int* w) { return or_bg_check_winner((const or_bg_state*)s, w); }
static uint32_t bgv_enc(const or_state* s, const or_play* p) { return or_bg_encode((const or_bg_state*)s, p); }
static void bgv_dec(const or_state* s, uint32_t c, or_play* p) { or_bg_decode((const or_bg_state*)s, c, p); }
static void bgv_planes(const or_state* s, float* o) { or_bg_planes((const or_bg_state*)s, o); }
static void bgv_roll(or_state* s, uint8_t a, uint8_t b) { ((or_bg_state*)s)->roll[0] = a; ((or_bg_state*)s)->roll[1] = b; }

static void tv_new(or_state* s) { or_ttt_new((or_ttt_state*)s); }
static int tv_valid(const or_state* s, or_play* o, int cap) {
    uint8_t m[9]; int n = or_ttt_valid_moves((const or_ttt_state*)s, m);
    for (int i = 0; i < n && i < cap; ++i) { o[i].mv[0] = (int8_t)m[i]; o[i].mv[1] = o[i].mv[2] = o[i].mv[3] = OR_NO_MOVE; }
    return n;
}
static void tv_apply(or_state* s, const or_play* p, uint8_t a, uint8_t b) { (void)a; (void)b; or_ttt_apply_move((or_ttt_state*)s, (uint8_t)p->mv[0]); }
static void tv_skip(or_state* s, uint8_t a, uint8_t b) { (void)a; (void)b; ((or_ttt_state*)s)->player = (int8_t)(-((or_ttt_state*)s)->player); }
static int tv_player(const or_state* s) { return ((const or_ttt_state*)s)->player; }
static int tv_winner(const or_state* s, int* w) { return or_ttt_check_winner((const or_ttt_state*)s, w); }
static uint32_t tv_enc(const or_state* s, const or_play* p) { (void)s; return (uint32_t)(uint8_t)p->mv[0]; }
static void tv_dec(const or_state* s, uint32_t c, or_play* p) { (void)s; p->mv[0] = (int8_t)c; p->mv[1] = p->mv[2] = p->mv[3] = OR_NO_MOVE; }
static void tv_planes(const or_state* s, float* o) { or_ttt_planes((const or_ttt_state*)s, o); }
static void tv_roll(or_state* s, uint8_t a, uint8_t b) { (void)s; (void)a; (void)b; }

static const or_game GAMES[2] = {
    {0, 9, 27, 1, tv_new, tv_valid, tv_apply, tv_skip, tv_player, tv_winner, tv_enc, tv_dec, tv_planes, tv_roll},
    {1, OR_BG_ACTIONS, OR_BG_PLANES, 0, bgv_new, bgv_valid, bgv_apply, bgv_skip, bgv_player, bgv_winner,
     bgv_enc, bgv_dec, bgv_planes, bgv_roll},
};
const or_game* or_game_by_id(int id) { return (id == 0 || id == 1) ? &GAMES[id] : NULL; }

/* ======================================================================== */
/* MCTS                                                                      */
/* ======================================================================== */
void or_store_init(or_store* st) { st->nodes = NULL; st->n = 0; st->cap = 0; }
void or_store_free(or_store* st) { free(st->nodes); st->nodes = NULL; st->n = st->cap = 0; }

/* NodeStore::add_node, node_store.rs:34-45 (legal plays are computed lazily at expansion; the
 * reference computes them eagerly in Node::new, node.rs:50 -- not observable) */
static int store_add(or_store* st, const or_state* s, int parent, int action, float policy) {
    if (st->n == st->cap) {
        st->cap = st->cap ? st->cap * 2 : 1024;
        st->nodes = realloc(st->nodes, sizeof(or_node) * (size_t)st->cap);
    }
    or_node* nd = &st->nodes[st->n];
    nd->state = *s; nd->parent = parent; nd->first_child = -1; nd->n_children = 0;
    nd->visits = 0.0f; nd->value = 0.0f; nd->policy = policy; nd->action = action; nd->drained = 0;
    return st->n++;
}

/* Node::alpha_ucb, node.rs:98-112: q + (c * (sqrt(N_parent) / (n + 1))) * p, f32, this association */
float or_alpha_ucb(const or_store* st, int idx, float c) {
    const or_node* nd = &st->nodes[idx];
    float q = nd->visits == 0.0f ? 0.0f : nd->value / nd->visits;
    if (nd->parent < 0) return INFINITY;
    const or_node* par = &st->nodes[nd->parent];
    float t = sqrtf(par->visits) / (nd->visits + 1.0f);
    float u = c * t;
    float w = u * nd->policy;
    return q + w;
}

/* alpha_select_leaf_node + select_alpha, alpha_mcts.rs:14-33.  Iterator::max_by keeps the LAST of
 * equal maxima and partial_cmp(..).unwrap_or(Equal) lets a NaN replace / be replaced. */
int or_select_leaf(const or_store* st, int root, float c, int* depth) {
    int idx = root, d = 0;
    for (;;) {
        const or_node* nd = &st->nodes[idx];
        if (nd->n_children == 0) break;
        int best = nd->first_child;
        float ub = or_alpha_ucb(st, best, c);
        for (int j = 1; j < nd->n_children; ++j) {
            int ch = nd->first_child + j;
            float un = or_alpha_ucb(st, ch, c);
            if (!(ub > un)) { best = ch; ub = un; }      /* Greater keeps acc, Less/Equal/NaN takes new */
        }
        idx = best; d++;
    }
    if (depth) *depth = d;
    return idx;
}

/* backpropagate, simple_mcts.rs:96-103: same sign at every level */
void or_backpropagate(or_store* st, int idx, float v) {
    while (idx >= 0) {
        or_node* nd = &st->nodes[idx];
        nd->visits += 1.0f;
        nd->value += v;
        idx = nd->parent;
    }
}

/* turn_policy_to_probs_tensor (utils.rs:74-84) + alpha_expand_tensor (node.rs:157-174).
 * probs = P[legal] / sum(P[legal]); the sum is taken sequentially over the plays in play order
 * (torch sums the 1352-vector in its own order: tolerance-level difference). */
static void expand_node(const or_game* g, or_store* st, int idx, const float* policy_row,
                        uint64_t seed, uint32_t game_id, uint32_t round, uint32_t e, or_stats* stats) {
    if (st->nodes[idx].drained) return;                   /* expandable_moves already drained */
    or_state s = st->nodes[idx].state;
    or_play* plays = malloc(sizeof(or_play) * OR_MAX_PLAYS);
    int k = g->valid_moves(&s, plays, OR_MAX_PLAYS);
    uint32_t* codes = malloc(sizeof(uint32_t) * (size_t)(k > 0 ? k : 1));
    float sum = 0.0f;
    for (int j = 0; j < k; ++j) {
        codes[j] = g->encode(&s, &plays[j]);
        sum += policy_row[codes[j]];
    }
    if (stats) {
        for (int j = 0; j < k; ++j) for (int i = 0; i < j; ++i) if (codes[i] == codes[j]) { stats->code_collisions++; break; }
        stats->expansions++; stats->children += (uint64_t)k;
        if ((uint64_t)k > stats->max_children) stats->max_children = (uint64_t)k;
    }
    int first = st->n;
    for (int j = 0; j < k; ++j) {
        float prior = policy_row[codes[j]] / sum;
        or_state ns = s;
        uint8_t d0 = 0, d1 = 0;
        if (!g->deterministic) or_dice(seed, game_id, round, e, (uint32_t)j, &d0, &d1);
        g->apply_move(&ns, &plays[j], d0, d1);
        store_add(st, &ns, idx, (int)codes[j], prior);
    }
    st->nodes[idx].first_child = k ? first : -1;
    st->nodes[idx].n_children = k;
    st->nodes[idx].drained = 1;
    free(plays); free(codes);
}

/* alpha_mcts_parallel, alpha_mcts.rs:91-202, as a resumable machine: or_mcts_next() advances to the next network
 * evaluation (OR_MCTS_EVAL: evaluate run->batch, then or_mcts_feed()), reports an iteration the reference skips
 * with `continue` (OR_MCTS_IDLE: nothing to evaluate, call next again) or the end (OR_MCTS_DONE).  The one-shot
 * or_alpha_mcts_parallel below drives it with its own evaluator; or_self_play_multi drives K of them in lockstep
 * with ONE merged evaluator call per phase (what the engine's pipelined self-play does on the GPU). */
struct or_mcts_run {
    const or_game* g; or_store* st; int n; or_mcts_cfg cfg;
    uint64_t seed; uint32_t step; const uint32_t* game_ids; const uint32_t* rounds; int ref_quirks; or_stats* stats;
    or_state* batch;          /* [n] states of the pending evaluation (stale rows keep their content)   */
    int* sel;                 /* :142 vec![0; n] */
    uint8_t* fresh;
    float* noise;
    int phase;                /* 0 = roots pending, 1 + it = iteration it pending */
    uint32_t it;
};

static or_mcts_run* g_last_run = NULL;      /* trace tools only (or_mcts_last) */
or_mcts_run* or_mcts_begin(const or_game* g, or_store* st, const or_state* states, int n, const or_mcts_cfg* cfg,
                           uint64_t seed, uint32_t step, const uint32_t* game_ids, const uint32_t* rounds,
                           int ref_quirks, or_stats* stats) {
    assert(st->n == 0);                                                    /* :94 */
    or_mcts_run* r = calloc(1, sizeof *r);
    r->g = g; r->st = st; r->n = n; r->cfg = *cfg; r->seed = seed; r->step = step;
    r->game_ids = game_ids; r->rounds = rounds; r->ref_quirks = ref_quirks; r->stats = stats;
    r->batch = malloc(sizeof(or_state) * (size_t)(n > 0 ? n : 1));
    memcpy(r->batch, states, sizeof(or_state) * (size_t)n);
    r->sel = calloc((size_t)(n > 0 ? n : 1), sizeof(int));
    r->fresh = malloc((size_t)(n > 0 ? n : 1));
    r->noise = malloc(sizeof(float) * (size_t)g->n_actions);
    r->phase = 0; r->it = 0;
    g_last_run = r;
    return r;
}
const or_state* or_mcts_batch(const or_mcts_run* r) { return r->batch; }
int or_mcts_rows(const or_mcts_run* r) { return r->n; }
/* read-only views of a run between or_mcts_next() and or_mcts_feed() (scripts/spec_price.py replays searches from them) */
or_mcts_run* or_mcts_last(void) { return g_last_run; }
const int* or_mcts_sel(const or_mcts_run* r) { return r->sel; }
const uint8_t* or_mcts_fresh(const or_mcts_run* r) { return r->fresh; }
int or_mcts_phase(const or_mcts_run* r) { return r->phase ? 1 + (int)r->it : 0; }
const or_store* or_mcts_store(const or_mcts_run* r) { return r->st; }
void or_mcts_end(or_mcts_run* r) { if (g_last_run == r) g_last_run = NULL; free(r->batch); free(r->sel); free(r->fresh); free(r->noise); free(r); }

int or_mcts_next(or_mcts_run* r) {
    if (r->phase == 0) return OR_MCTS_EVAL;                                /* :97-104 forward_policy of the roots */
    if (r->it >= r->cfg.iterations) return OR_MCTS_DONE;                   /* :149 */
    const or_game* g = r->g; or_store* st = r->st; or_stats* stats = r->stats;
    int node_selected = 0;
    for (int gi = 0; gi < r->n; ++gi) {                                    /* :153-168 */
        int depth = 0, winner = 0;
        int idx = or_select_leaf(st, gi, r->cfg.c, &depth);
        if (stats) { stats->selections++; stats->depth_sum += (uint64_t)depth; }
        r->fresh[gi] = 0;
        if (g->check_winner(&st->nodes[idx].state, &winner)) {
            int root_player = g->get_player(&st->nodes[gi].state);
            float v = winner == root_player ? 1.0f : (winner == -root_player ? -1.0f : 0.0f);
            or_backpropagate(st, idx, v);
            if (stats) stats->terminal_hits++;
        } else {
            node_selected = 1; r->sel[gi] = idx; r->fresh[gi] = 1;
        }
    }
    if (!node_selected) { r->it++; return OR_MCTS_IDLE; }                  /* :170-172 */
    for (int gi = 0; gi < r->n; ++gi) r->batch[gi] = st->nodes[r->sel[gi]].state;   /* :175-183 (stale slots too) */
    return OR_MCTS_EVAL;                                                   /* :186 */
}

void or_mcts_feed(or_mcts_run* r, float* policy /* [n][A], modified in place for the roots */, const float* value) {
    const or_game* g = r->g; or_store* st = r->st; or_stats* stats = r->stats;
    const int A = g->n_actions, n = r->n;
    if (stats) stats->nn_evals += (uint64_t)n;
    if (r->phase == 0) {
        /* apply_dirichlet, noise.rs:27-34: ONE sample shared by all rows, before masking */
        or_dirichlet(r->seed, r->step, r->cfg.dir_alpha, A, r->noise);
        const float eps = r->cfg.dir_eps, om = 1.0f - eps;
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < A; ++a) {
                float x = om * policy[(size_t)i * A + a], y = eps * r->noise[a];
                policy[(size_t)i * A + a] = x + y;
            }
        for (int i = 0; i < n; ++i) store_add(st, &r->batch[i], -1, -1, 0.0f);   /* :110-112 */
        for (int i = 0; i < n; ++i) {                                      /* :119-127 */
            st->nodes[i].visits = 1.0f;
            expand_node(g, st, i, &policy[(size_t)i * A], r->seed, r->game_ids[i], r->rounds[i], 0, stats);
        }
        r->phase = 1;
        return;
    }
    for (int slot = 0; slot < n; ++slot) {                                 /* :192-200 */
        if (!r->ref_quirks && !r->fresh[slot]) continue;                   /* clean variant: no stale re-backprop */
        int idx = r->sel[slot];
        /* child dice are keyed by the game that owns the node: a stale slot never creates children */
        expand_node(g, st, idx, &policy[(size_t)slot * A], r->seed, r->game_ids[slot], r->rounds[slot], r->it + 1, stats);
        or_backpropagate(st, idx, value[slot]);
    }
    r->it++;
}

void or_alpha_mcts_parallel(const or_game* g, or_store* st, const or_state* states, int n,
                            const or_mcts_cfg* cfg, or_eval_fn eval, void* ectx,
                            uint64_t seed, uint32_t step, const uint32_t* game_ids,
                            const uint32_t* rounds, int ref_quirks, or_stats* stats) {
    const int A = g->n_actions;
    float* policy = malloc(sizeof(float) * (size_t)n * (size_t)A);
    float* value = malloc(sizeof(float) * (size_t)n);
    or_mcts_run* r = or_mcts_begin(g, st, states, n, cfg, seed, step, game_ids, rounds, ref_quirks, stats);
    for (;;) {
        const int what = or_mcts_next(r);
        if (what == OR_MCTS_DONE) break;
        if (what == OR_MCTS_IDLE) continue;
        eval(ectx, or_mcts_batch(r), n, policy, value);
        or_mcts_feed(r, policy, value);
    }
    or_mcts_end(r);
    free(policy); free(value);
}

/* get_prob_tensor_parallel, utils.rs:42-58: root child visits scattered at encode(action) / row sum.
 * Row sum taken sequentially over children (see expand_node note). */
void or_get_prob_tensor_parallel(const or_game* g, const or_store* st, int n, float* probs) {
    const int A = g->n_actions;
    memset(probs, 0, sizeof(float) * (size_t)n * (size_t)A);
    for (int i = 0; i < n; ++i) {
        const or_node* root = &st->nodes[i];
        float sum = 0.0f;
        for (int j = 0; j < root->n_children; ++j) sum += st->nodes[root->first_child + j].visits;
        for (int j = 0; j < root->n_children; ++j) {
            const or_node* ch = &st->nodes[root->first_child + j];
            probs[(size_t)i * A + ch->action] = ch->visits / sum;          /* index_put_ accumulate=false */
        }
        if (root->n_children == 0)                                         /* 0/0 row */
            for (int a = 0; a < A; ++a) probs[(size_t)i * A + a] = NAN;
    }
}

/* ======================================================================== */
/* self_play_parallel, src/alphazero/alpha_parallel.rs:101-231              */
/* ======================================================================== */
typedef struct { int8_t player; float* ps; float* planes; } mem_frag;
typedef struct { mem_frag* f; int n, cap; } mem_list;

static void frag_push(or_fragments* out, int* cap, const or_game* g, int8_t outcome, const float* ps,
                      const float* planes, uint32_t game) {
    const size_t A = (size_t)g->n_actions, P = (size_t)g->n_planes;
    if (out->n == *cap) {
        *cap = *cap ? *cap * 2 : 1024;
        out->outcome = realloc(out->outcome, (size_t)*cap);
        out->ps = realloc(out->ps, sizeof(float) * A * (size_t)*cap);
        out->state = realloc(out->state, sizeof(float) * P * (size_t)*cap);
        out->game = realloc(out->game, sizeof(uint32_t) * (size_t)*cap);
    }
    out->outcome[out->n] = outcome;
    memcpy(out->ps + A * (size_t)out->n, ps, sizeof(float) * A);
    memcpy(out->state + P * (size_t)out->n, planes, sizeof(float) * P);
    out->game[out->n] = game;
    out->n++;
}

/* weighted_select_tensor_idx, alphazero.rs:129-137: rand 0.8 WeightedIndex over f64 weights:
 * cumulative sums in index order, u ~ U[0,total), first index whose cumulative weight > u. */
static int weighted_select(const float* w, int A, double u01) {
    double total = 0.0;
    for (int a = 0; a < A; ++a) total += (double)w[a];
    double x = u01 * total, cum = 0.0;
    int last_nz = 0;
    for (int a = 0; a < A; ++a) {
        if (w[a] != 0.0f) last_nz = a;
        cum += (double)w[a];
        if (cum > x) return a;
    }
    return last_nz;
}

/* one self-play batch (= one call of self_play_parallel) as a steppable object: sp_roots() lists the live games'
 * states for alpha_mcts_parallel (:131-146), sp_apply() is the per-game loop body after the search (:164-228) */
typedef struct {
    const or_game* g; uint32_t n_games, first_game_id; or_mcts_cfg cfg; float inv_t; uint64_t seed; int ref_quirks;
    or_state* states; uint32_t* n_rounds; uint8_t* live; mem_list* mem; int out_cap;
    or_fragments* out; or_stats* stats; uint32_t* plies; int8_t* winners;
    or_state* roots; uint32_t *ids, *gids, *rnds; float *probs, *planes;
    uint32_t n_live, m;
} sp_batch;

static void sp_init(sp_batch* b, const or_game* g, uint32_t n_games, uint32_t first_game_id, const or_mcts_cfg* cfg,
                    float temperature, uint64_t seed, int ref_quirks, or_fragments* out, or_stats* stats,
                    uint32_t* plies, int8_t* winners) {
    const int A = g->n_actions, P = g->n_planes;
    memset(b, 0, sizeof *b);
    b->g = g; b->n_games = n_games; b->first_game_id = first_game_id; b->cfg = *cfg; b->seed = seed;
    b->ref_quirks = ref_quirks; b->out = out; b->stats = stats; b->plies = plies; b->winners = winners;
    b->inv_t = (float)(1.0 / (double)temperature);                        /* :165 pow_(1.0 / temperature) */
    b->states = malloc(sizeof(or_state) * n_games);
    b->n_rounds = calloc(n_games, sizeof(uint32_t));
    b->live = malloc(n_games);
    b->mem = calloc(n_games, sizeof(mem_list));
    memset(out, 0, sizeof *out);
    for (uint32_t i = 0; i < n_games; ++i) {                               /* :103-111 */
        g->new_state(&b->states[i]);
        if (!g->deterministic) {
            uint8_t d0, d1; or_dice(seed, first_game_id + i, 0, OR_TAG_INIT_ROLL, 0, &d0, &d1);
            g->set_roll(&b->states[i], d0, d1);
        }
        b->live[i] = 1;
        if (winners) winners[i] = 0;
        if (plies) plies[i] = 0;
    }
    b->n_live = n_games;
    b->roots = malloc(sizeof(or_state) * n_games);
    b->ids = malloc(sizeof(uint32_t) * n_games);
    b->gids = malloc(sizeof(uint32_t) * n_games);
    b->rnds = malloc(sizeof(uint32_t) * n_games);
    b->probs = malloc(sizeof(float) * (size_t)n_games * (size_t)A);
    b->planes = malloc(sizeof(float) * (size_t)P);
}

static uint32_t sp_roots(sp_batch* b) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < b->n_games; ++i) if (b->live[i]) {
        b->roots[m] = b->states[i]; b->ids[m] = i; b->gids[m] = b->first_game_id + i; b->rnds[m] = b->n_rounds[i]; m++;
    }
    b->m = m;
    return m;
}

static void sp_apply(sp_batch* b, const or_store* stp) {
    const or_game* g = b->g; const or_mcts_cfg* cfg = &b->cfg; or_fragments* out = b->out; or_stats* stats = b->stats;
    const int A = g->n_actions, P = g->n_planes;
    const uint32_t m = b->m, first_game_id = b->first_game_id; const uint64_t seed = b->seed;
    const int ref_quirks = b->ref_quirks;
    or_state* states = b->states; uint32_t* n_rounds = b->n_rounds; uint8_t* live = b->live; mem_list* mem = b->mem;
    float* probs = b->probs; float* planes = b->planes; const uint32_t* ids = b->ids;
    or_get_prob_tensor_parallel(g, stp, (int)m, probs);                   /* :164 */
    for (size_t q = 0; q < (size_t)m * (size_t)A; ++q)                    /* :165 */
        probs[q] = probs[q] != probs[q] ? probs[q] : or_det_powf(probs[q], b->inv_t);
    for (uint32_t pi = 0; pi < m; ++pi) {                                 /* :168-224 */
        uint32_t gi = ids[pi];
        or_state* s = &states[gi];
        const float* row = probs + (size_t)pi * (size_t)A;
        int removed = 0, flushed = 0;
        if (n_rounds[gi] >= cfg->round_limit) {                           /* :172-180, no `continue` */
            for (int k = 0; k < mem[gi].n; ++k)
                frag_push(out, &b->out_cap, g, 0, mem[gi].f[k].ps, mem[gi].f[k].planes, first_game_id + gi);
            removed = 1; flushed = 1;
        }
        /* :183-189: sum is NaN (nonzero) for a 0/0 row, so this is "root has no children" */
        if (stp->nodes[pi].n_children == 0) {
            n_rounds[gi] += 1;
            uint8_t d0 = 0, d1 = 0;
            if (!g->deterministic) or_dice(seed, first_game_id + gi, n_rounds[gi] - 1, OR_TAG_MOVE_ROLL, 0, &d0, &d1);
            g->skip_turn(s, d0, d1);
            if (removed) live[gi] = 0;
            continue;
        }
        double u = or_uniform01(seed, first_game_id + gi, n_rounds[gi], OR_TAG_SAMPLE, 0);
        int a = weighted_select(row, A, u);                               /* :192 */
        /* :195-199 push MemoryFragment{outcome: player, ps, state} */
        if (mem[gi].n == mem[gi].cap) {
            mem[gi].cap = mem[gi].cap ? mem[gi].cap * 2 : 64;
            mem[gi].f = realloc(mem[gi].f, sizeof(mem_frag) * (size_t)mem[gi].cap);
        }
        mem_frag* mf = &mem[gi].f[mem[gi].n++];
        mf->player = (int8_t)g->get_player(s);
        mf->ps = malloc(sizeof(float) * (size_t)A); memcpy(mf->ps, row, sizeof(float) * (size_t)A);
        g->planes(s, planes);
        mf->planes = malloc(sizeof(float) * (size_t)P); memcpy(mf->planes, planes, sizeof(float) * (size_t)P);
        /* :202-210 decode, assert legal, apply */
        or_play mv; g->decode(s, (uint32_t)a, &mv);
        {
            or_play* vm = malloc(sizeof(or_play) * OR_MAX_PLAYS);
            int k = g->valid_moves(s, vm, OR_MAX_PLAYS), ok = 0;
            for (int j = 0; j < k; ++j) if (memcmp(vm[j].mv, mv.mv, 4) == 0) ok = 1;
            if (!ok && stats) stats->illegal_decodes++;
            free(vm);
        }
        uint8_t d0 = 0, d1 = 0;
        if (!g->deterministic) or_dice(seed, first_game_id + gi, n_rounds[gi], OR_TAG_MOVE_ROLL, 0, &d0, &d1);
        g->apply_move(s, &mv, d0, d1);
        n_rounds[gi] += 1;                                                /* :213 */
        int winner = 0;
        if (g->check_winner(s, &winner)) {                                /* :215-223 */
            if (!(flushed && !ref_quirks)) {                              /* Q18: the reference flushes twice */
                for (int k = 0; k < mem[gi].n; ++k) {
                    int8_t pl = mem[gi].f[k].player;
                    int8_t oc = winner == pl ? 1 : (winner == -pl ? -1 : 0);
                    frag_push(out, &b->out_cap, g, oc, mem[gi].f[k].ps, mem[gi].f[k].planes, first_game_id + gi);
                }
            }
            if (b->winners) b->winners[gi] = (int8_t)winner;
            removed = 1;
        }
        if (removed) live[gi] = 0;
    }
    for (uint32_t i = 0; i < b->n_games; ++i) if (b->plies) b->plies[i] = n_rounds[i];
    b->n_live = 0;
    for (uint32_t i = 0; i < b->n_games; ++i) b->n_live += live[i];
}

static void sp_free(sp_batch* b) {
    for (uint32_t i = 0; i < b->n_games; ++i) {
        for (int k = 0; k < b->mem[i].n; ++k) { free(b->mem[i].f[k].ps); free(b->mem[i].f[k].planes); }
        free(b->mem[i].f);
    }
    free(b->states); free(b->n_rounds); free(b->live); free(b->mem); free(b->roots); free(b->ids); free(b->gids);
    free(b->rnds); free(b->probs); free(b->planes);
}

int or_self_play_parallel(const or_game* g, uint32_t n_games, uint32_t first_game_id,
                          const or_mcts_cfg* cfg, float temperature, uint64_t seed,
                          or_eval_fn eval, void* ectx, int ref_quirks, uint32_t max_steps,
                          or_fragments* out, or_stats* stats, uint32_t* plies, int8_t* winners) {
    sp_batch b;
    sp_init(&b, g, n_games, first_game_id, cfg, temperature, seed, ref_quirks, out, stats, plies, winners);
    uint32_t step = 0;
    while (b.n_live > 0 && (max_steps == 0 || step < max_steps)) {        /* :129 */
        const uint32_t m = sp_roots(&b);
        or_store st; or_store_init(&st);                                  /* :137 fresh store every move-step */
        or_alpha_mcts_parallel(g, &st, b.roots, (int)m, cfg, eval, ectx, seed, step, b.gids, b.rnds, ref_quirks, stats);
        sp_apply(&b, &st);
        or_store_free(&st);
        step++;
    }
    sp_free(&b);
    return (int)step;
}

/* K independent self_play_parallel calls (same network, same MctsConfig; own seed, game ids and outputs -- what
 * learn_parallel issues back to back, alpha_parallel.rs:49-62) advanced in LOCKSTEP: all batches start at move-step 0
 * and every phase of the search (root evaluation, then each iteration) evaluates ONE merged batch: the live games of
 * batch 0, then of batch 1, ...  A batch whose iteration is skipped (`continue`, :170-172) still occupies its rows
 * (stale content, results ignored), a finished batch has none.  With an evaluator that is a pure function of the
 * state, batch k's records equal or_self_play_parallel(seed k) byte for byte; with the engine's ResNet as evaluator
 * they equal the engine's pipelined self-play, which merges its network batches the same way. */
int or_self_play_multi(const or_game* g, uint32_t n_batches, const uint32_t* n_games, const uint32_t* first_game_ids,
                       const uint64_t* seeds, const or_mcts_cfg* cfg, float temperature, or_eval_fn eval, void* ectx,
                       int ref_quirks, uint32_t max_steps, or_fragments* outs /* [n_batches] */,
                       or_stats* stats /* [n_batches] */, uint32_t* steps /* [n_batches] or NULL */) {
    const int A = g->n_actions;
    sp_batch* b = malloc(sizeof(sp_batch) * n_batches);
    or_store* st = malloc(sizeof(or_store) * n_batches);
    or_mcts_run** run = malloc(sizeof(or_mcts_run*) * n_batches);
    int* what = malloc(sizeof(int) * n_batches);
    size_t total = 0;
    for (uint32_t k = 0; k < n_batches; ++k) {
        sp_init(&b[k], g, n_games[k], first_game_ids[k], cfg, temperature, seeds[k], ref_quirks, &outs[k], &stats[k], NULL, NULL);
        total += n_games[k];
        if (steps) steps[k] = 0;
    }
    or_state* merged = malloc(sizeof(or_state) * (total ? total : 1));
    float* policy = malloc(sizeof(float) * (total ? total : 1) * (size_t)A);
    float* value = malloc(sizeof(float) * (total ? total : 1));
    uint32_t step = 0;
    for (;;) {
        uint32_t alive = 0;
        for (uint32_t k = 0; k < n_batches; ++k) alive += b[k].n_live;
        if (!alive || (max_steps && step >= max_steps)) break;
        for (uint32_t k = 0; k < n_batches; ++k) {
            run[k] = NULL;
            if (!b[k].n_live) continue;
            const uint32_t m = sp_roots(&b[k]);
            or_store_init(&st[k]);
            run[k] = or_mcts_begin(g, &st[k], b[k].roots, (int)m, cfg, seeds[k], step, b[k].gids, b[k].rnds, ref_quirks, &stats[k]);
            if (steps) steps[k] = step + 1;
        }
        for (;;) {                                                        /* root phase, then one phase per iteration */
            int done = 1, any_eval = 0;
            size_t rows = 0;
            for (uint32_t k = 0; k < n_batches; ++k) {
                if (!run[k]) continue;
                what[k] = or_mcts_next(run[k]);
                if (what[k] != OR_MCTS_DONE) done = 0;
                if (what[k] == OR_MCTS_EVAL) any_eval = 1;
                memcpy(merged + rows, or_mcts_batch(run[k]), sizeof(or_state) * (size_t)or_mcts_rows(run[k]));
                rows += (size_t)or_mcts_rows(run[k]);
            }
            if (done) break;
            if (!any_eval) continue;
            eval(ectx, merged, (int)rows, policy, value);
            rows = 0;
            for (uint32_t k = 0; k < n_batches; ++k) {
                if (!run[k]) continue;
                if (what[k] == OR_MCTS_EVAL) or_mcts_feed(run[k], policy + rows * (size_t)A, value + rows);
                rows += (size_t)or_mcts_rows(run[k]);
            }
        }
        for (uint32_t k = 0; k < n_batches; ++k) {
            if (!run[k]) continue;
            or_mcts_end(run[k]);
            sp_apply(&b[k], &st[k]);
            or_store_free(&st[k]);
        }
        step++;
    }
    for (uint32_t k = 0; k < n_batches; ++k) sp_free(&b[k]);
    free(b); free(st); free(run); free(what); free(merged); free(policy); free(value);
    return (int)step;
}

void or_free_fragments(or_fragments* f) {
    free(f->outcome); free(f->ps); free(f->state); free(f->game);
    memset(f, 0, sizeof *f);
}

/* A cheap deterministic stand-in evaluator for CPU-only tests: logits are an integer hash of the
 * planes, policy = softmax, value in (-1,1).  NOT a network and not part of the reference. */
void or_hash_eval(void* ctx, const or_state* states, int n, float* policy, float* value) {
    const or_game* g = (const or_game*)ctx;
    const int A = g->n_actions, P = g->n_planes;
    float* pl = malloc(sizeof(float) * (size_t)P);
    for (int i = 0; i < n; ++i) {
        g->planes(&states[i], pl);
        uint32_t h = 2166136261u;
        for (int k = 0; k < P; ++k) { h ^= (uint32_t)(int32_t)pl[k] + 0x9e3779b9u + (uint32_t)k; h *= 16777619u; }
        double sum = 0.0;
        for (int a = 0; a < A; ++a) {
            uint32_t x = h ^ ((uint32_t)a * 2654435761u); x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
            double logit = (double)(x & 0xffff) / 65536.0 * 4.0;
            double e = exp(logit);
            policy[(size_t)i * A + a] = (float)e; sum += e;
        }
        for (int a = 0; a < A; ++a) policy[(size_t)i * A + a] = (float)((double)policy[(size_t)i * A + a] / sum);
        uint32_t y = h * 3266489917u; y ^= y >> 16;
        value[i] = (float)((double)(y & 0xffff) / 32768.0 - 1.0) * 0.5f;
    }
    free(pl);
}

/* ======================================================================== */
/* Batch helpers for the parity tests (test infrastructure)                  */
/* ======================================================================== */

/* states reached by seeded uniformly-random legal play from Backgammon::new() */
int or_random_walk_states(uint64_t seed, uint32_t n_games, uint32_t max_plies, or_bg_state* out, int cap) {
    int n = 0;
    or_play* plays = malloc(sizeof(or_play) * OR_MAX_PLAYS);
    for (uint32_t g = 0; g < n_games && n < cap; ++g) {
        or_bg_state s; or_bg_new(&s);
        uint8_t d0, d1; or_dice(seed, g, 0, OR_TAG_INIT_ROLL, 0, &d0, &d1);
        s.roll[0] = d0; s.roll[1] = d1;
        for (uint32_t ply = 0; ply < max_plies && n < cap; ++ply) {
            int w;
            if (or_bg_check_winner(&s, &w)) break;
            out[n++] = s;
            int k = or_bg_valid_moves(&s, plays, OR_MAX_PLAYS);
            or_dice(seed, g, ply, OR_TAG_MOVE_ROLL, 0, &d0, &d1);
            if (k == 0) { or_bg_skip_turn(&s, d0, d1); continue; }
            double u = or_uniform01(seed, g, ply, OR_TAG_SAMPLE, 0);
            int j = (int)(u * k); if (j >= k) j = k - 1;
            or_bg_apply_move(&s, &plays[j], d0, d1);
        }
    }
    free(plays);
    return n;
}

/* plays[n][cap] (4 bytes each), counts[n] */
void or_bg_valid_moves_batch(const or_bg_state* s, int n, int8_t* plays, int cap, uint32_t* counts) {
    or_play* tmp = malloc(sizeof(or_play) * OR_MAX_PLAYS);
    for (int i = 0; i < n; ++i) {
        int k = or_bg_valid_moves(&s[i], tmp, OR_MAX_PLAYS);
        counts[i] = (uint32_t)k;
        for (int j = 0; j < k && j < cap; ++j) memcpy(plays + ((size_t)i * cap + j) * 4, tmp[j].mv, 4);
    }
    free(tmp);
}
void or_bg_encode_batch(const or_bg_state* s, const int8_t* plays, int n, uint32_t* codes) {
    for (int i = 0; i < n; ++i) { or_play p; memcpy(p.mv, plays + (size_t)i * 4, 4); codes[i] = or_bg_encode(&s[i], &p); }
}
void or_bg_decode_batch(const or_bg_state* s, const uint32_t* codes, int n, int8_t* plays) {
    for (int i = 0; i < n; ++i) { or_play p; or_bg_decode(&s[i], codes[i], &p); memcpy(plays + (size_t)i * 4, p.mv, 4); }
}
void or_bg_apply_batch(or_bg_state* s, const int8_t* plays, const uint8_t* dice, int n) {
    for (int i = 0; i < n; ++i) { or_play p; memcpy(p.mv, plays + (size_t)i * 4, 4); or_bg_apply_move(&s[i], &p, dice[2 * i], dice[2 * i + 1]); }
}
void or_bg_planes_batch(const or_bg_state* s, int n, float* out) {
    for (int i = 0; i < n; ++i) or_bg_planes(&s[i], out + (size_t)i * OR_BG_PLANES);
}
