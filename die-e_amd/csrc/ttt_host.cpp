// ttt_host.cpp -- tic-tac-toe on the host behind the C ABI (see ttt_host.h for why this game is host work).
// Reference: src/tictactoe/mod.rs:28-100 (rules), src/alphazero/nnet.rs:24-34,57-133 (ResNet, 64 filters x 4 blocks),
// src/mcts/alpha_mcts.rs:14-33,91-202 + node.rs:98-112,157-174 + simple_mcts.rs:96-103 + utils.rs:42-84 + noise.rs:27-34
// (batched search), src/alphazero/alpha_parallel.rs:101-231 + alphazero.rs:129-137 (self-play driver).
// RNG conventions are the engine's (Philox keyed by seed / game / round / purpose, bg_device.h), so that a record of this
// path can be checked against the test oracle exactly like the HIP path's.
#include "ttt_host.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "bg_device.h"
#include "engine.h"

namespace diee {
void dirichlet_host(uint64_t seed, uint32_t step, float alpha, int n, float* out);     // search_host.cpp
namespace ttt {

// ---- rules -----------------------------------------------------------------------------------------------------------
int valid_moves(const diee_ttt_state& s, uint8_t out[9]) {
    int k = 0;
    for (int i = 0; i < 9; ++i) if (s.board[i] == 0) out[k++] = (uint8_t)i;
    return k;
}
void apply_move(diee_ttt_state& s, uint8_t a) { s.board[a] = s.player; s.player = (int8_t)-s.player; }
void skip_turn(diee_ttt_state& s) { s.player = (int8_t)-s.player; }
bool check_winner(const diee_ttt_state& s, int& winner) {
    static const int L[8][3] = {{0, 1, 2}, {3, 4, 5}, {6, 7, 8}, {0, 3, 6}, {1, 4, 7}, {2, 5, 8}, {0, 4, 8}, {2, 4, 6}};
    for (const auto& l : L) {
        const int a = s.board[l[0]];
        if (a != 0 && a == s.board[l[1]] && a == s.board[l[2]]) { winner = a; return true; }
    }
    for (int i = 0; i < 9; ++i) if (s.board[i] == 0) return false;
    winner = 0;                                                         // a full board without a line: draw, Some(0)
    return true;
}
void planes(const diee_ttt_state& s, float out[27]) {                   // stack([eq(-1), eq(0), eq(1)])
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 9; ++i) out[c * 9 + i] = s.board[i] == c - 1 ? 1.0f : 0.0f;
}

// ---- weights: the blob layout of include/diee.h with this game's sizes ------------------------------------------------
namespace {
struct Cursor {
    size_t off = 0;
    size_t take(size_t n) { const size_t o = off; off += n; return o; }
};
struct Layout {
    size_t init_w, init_b, init_bn, c1w[BLOCKS], c1b[BLOCKS], c2w[BLOCKS], c2b[BLOCKS], bn1[BLOCKS], bn2[BLOCKS];
    size_t pw, pb, pbn, pfw, pfb, vw, vb, vbn, vfw, vfb, total;
};
Layout make_layout() {
    Layout L; Cursor c;
    L.init_w = c.take((size_t)F * CIN * 9); L.init_b = c.take(F); L.init_bn = c.take(4 * F);
    for (int i = 0; i < BLOCKS; ++i) {                                   // ResBlock::new creation order, nnet.rs:38-45
        L.c1w[i] = c.take((size_t)F * F * 9); L.c1b[i] = c.take(F); L.c2w[i] = c.take((size_t)F * F * 9); L.c2b[i] = c.take(F);
        L.bn1[i] = c.take(4 * F); L.bn2[i] = c.take(4 * F);
    }
    L.pw = c.take((size_t)PH * F * 9); L.pb = c.take(PH); L.pbn = c.take(4 * PH); L.pfw = c.take((size_t)A * PH * HW); L.pfb = c.take(A);
    L.vw = c.take((size_t)VH * F * 9); L.vb = c.take(VH); L.vbn = c.take(4 * VH); L.vfw = c.take((size_t)VH * HW); L.vfb = c.take(1);
    L.total = c.off;
    return L;
}
const Layout& layout() { static const Layout L = make_layout(); return L; }
}  // namespace

size_t weights_count() { return layout().total; }

// tch-default initialisation, the same rules as the backgammon blob's (nn_host.cpp random_weights_bg)
void random_weights(uint64_t seed, float* blob) {
    const Layout& L = layout();
    uint32_t tid = 0x77700000u;
    auto uni = [&](size_t off, size_t n, float lo, float hi) {
        for (size_t i = 0; i < n; i += 4) {
            uint32_t o[4];
            philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(i / 4), 0u, tid, 0x57E16u, o);
            for (size_t k = 0; k < 4 && i + k < n; ++k) blob[off + i + k] = lo + (hi - lo) * ((float)(o[k] >> 8) * (1.0f / 16777216.0f));
        }
        ++tid;
    };
    auto cst = [&](size_t off, size_t n, float v) { for (size_t i = 0; i < n; ++i) blob[off + i] = v; };
    auto conv = [&](size_t w, size_t b, int cout, int cin) {
        const float bd = 1.0f / std::sqrt((float)(cin * 9));
        uni(w, (size_t)cout * cin * 9, -bd, bd); cst(b, cout, 0.f);
    };
    auto bn = [&](size_t o, int c) { uni(o, c, 0.f, 1.f); cst(o + c, c, 0.f); cst(o + 2 * (size_t)c, c, 0.f); cst(o + 3 * (size_t)c, c, 1.f); };
    conv(L.init_w, L.init_b, F, CIN); bn(L.init_bn, F);
    for (int i = 0; i < BLOCKS; ++i) { conv(L.c1w[i], L.c1b[i], F, F); conv(L.c2w[i], L.c2b[i], F, F); bn(L.bn1[i], F); bn(L.bn2[i], F); }
    conv(L.pw, L.pb, PH, F); bn(L.pbn, PH);
    { const float bd = 1.0f / std::sqrt((float)(PH * HW)); uni(L.pfw, (size_t)A * PH * HW, -bd, bd); uni(L.pfb, A, -bd, bd); }
    conv(L.vw, L.vb, VH, F); bn(L.vbn, VH);
    { const float bd = 1.0f / std::sqrt((float)(VH * HW)); uni(L.vfw, (size_t)VH * HW, -bd, bd); uni(L.vfb, 1, -bd, bd); }
}

void Engine::load_weights(const float* blob, size_t n) {
    const Layout& L = layout();
    if (n != L.total) throw EngineError(DIEE_ERR_ARG, "weight blob has " + std::to_string(n) + " floats, expected " + std::to_string(L.total));
    // eval-mode BatchNorm (eps 1e-5) folded into the convolution: w' = w * s, b' = (b - mean) * s + beta
    auto fold = [&](Conv& c, size_t w, size_t b, size_t bnp, int cout, int cin) {
        c.cout = cout; c.cin = cin; c.w.assign((size_t)cout * cin * 9, 0.f); c.b.assign(cout, 0.f);
        for (int o = 0; o < cout; ++o) {
            const float s = blob[bnp + o] / std::sqrt(blob[bnp + 3 * (size_t)cout + o] + 1e-5f);
            for (int k = 0; k < cin * 9; ++k) c.w[(size_t)o * cin * 9 + k] = blob[w + (size_t)o * cin * 9 + k] * s;
            c.b[o] = (blob[b + o] - blob[bnp + 2 * (size_t)cout + o]) * s + blob[bnp + (size_t)cout + o];
        }
    };
    fold(init_, L.init_w, L.init_b, L.init_bn, F, CIN);
    for (int i = 0; i < BLOCKS; ++i) { fold(c1_[i], L.c1w[i], L.c1b[i], L.bn1[i], F, F); fold(c2_[i], L.c2w[i], L.c2b[i], L.bn2[i], F, F); }
    fold(pconv_, L.pw, L.pb, L.pbn, PH, F); fold(vconv_, L.vw, L.vb, L.vbn, VH, F);
    pfc_w_.assign(blob + L.pfw, blob + L.pfw + (size_t)A * PH * HW); pfc_b_.assign(blob + L.pfb, blob + L.pfb + A);
    vfc_w_.assign(blob + L.vfw, blob + L.vfw + (size_t)VH * HW); vfc_b_.assign(blob + L.vfb, blob + L.vfb + 1);
    w_.assign(blob, blob + n);
}

// ---- the network, fp32 ------------------------------------------------------------------------------------------------
namespace {
// y[cout][3][3] = conv3x3(x[cin][3][3], pad 1) + b
void conv3x3(const float* x, const std::vector<float>& w, const std::vector<float>& b, int cout, int cin, float* y) {
    for (int o = 0; o < cout; ++o) {
        float acc[9];
        for (int p = 0; p < 9; ++p) acc[p] = b[o];
        const float* wo = w.data() + (size_t)o * cin * 9;
        for (int c = 0; c < cin; ++c) {
            const float* xc = x + c * 9; const float* wc = wo + c * 9;
            for (int py = 0; py < 3; ++py)
                for (int px = 0; px < 3; ++px) {
                    float s = 0.f;
                    for (int ky = 0; ky < 3; ++ky) {
                        const int yy = py + ky - 1;
                        if (yy < 0 || yy > 2) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            const int xx = px + kx - 1;
                            if (xx < 0 || xx > 2) continue;
                            s += xc[yy * 3 + xx] * wc[ky * 3 + kx];
                        }
                    }
                    acc[py * 3 + px] += s;
                }
        }
        for (int p = 0; p < 9; ++p) y[o * 9 + p] = acc[p];
    }
}
inline void relu(float* x, int n) { for (int i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f; }
}  // namespace

void Engine::forward(const diee_ttt_state* s, uint32_t n, float* policy, float* value) const {
    if (!loaded()) throw EngineError(DIEE_ERR_NO_WEIGHTS, "diee_load_weights has not been called");
    std::vector<float> x(F * 9), h(F * 9), t(F * 9), in(PLANES), ph(PH * 9), vh(VH * 9);
    for (uint32_t i = 0; i < n; ++i) {
        planes(s[i], in.data());
        conv3x3(in.data(), init_.w, init_.b, F, CIN, x.data()); relu(x.data(), F * 9);          // nnet.rs:64-67
        for (int b = 0; b < BLOCKS; ++b) {                                                     // ResBlock::forward_t, nnet.rs:24-34
            conv3x3(x.data(), c1_[b].w, c1_[b].b, F, F, h.data()); relu(h.data(), F * 9);
            conv3x3(h.data(), c2_[b].w, c2_[b].b, F, F, t.data());
            for (int k = 0; k < F * 9; ++k) { const float v = t[k] + x[k]; x[k] = v > 0.f ? v : 0.f; }
        }
        conv3x3(x.data(), pconv_.w, pconv_.b, PH, F, ph.data()); relu(ph.data(), PH * 9);     // policy head, nnet.rs:75-85
        float logit[A], mx = -INFINITY;
        for (int a = 0; a < A; ++a) {
            float acc = pfc_b_[a];
            for (int k = 0; k < PH * HW; ++k) acc += ph[k] * pfc_w_[(size_t)a * PH * HW + k];  // flatten: c * 9 + p
            logit[a] = acc; mx = acc > mx ? acc : mx;
        }
        float sum = 0.f;
        for (int a = 0; a < A; ++a) { logit[a] = std::exp(logit[a] - mx); sum += logit[a]; }
        for (int a = 0; a < A; ++a) policy[(size_t)i * A + a] = logit[a] / sum;
        conv3x3(x.data(), vconv_.w, vconv_.b, VH, F, vh.data()); relu(vh.data(), VH * 9);     // value head, nnet.rs:87-98
        float acc = vfc_b_[0];
        for (int k = 0; k < VH * HW; ++k) acc += vh[k] * vfc_w_[k];
        value[i] = std::tanh(acc);
    }
}

// ---- the batched search: alpha_mcts_parallel on one NodeStore laid out as arrays ----------------------------------------
namespace {
struct Store {                       // NodeStore<TicTacToe>, node_store.rs:9-11 (index = creation order, roots first)
    std::vector<diee_ttt_state> state;
    std::vector<int32_t> parent, first_child, n_children, action;
    std::vector<float> visits, value, prior;
    std::vector<uint8_t> drained;
    int add(const diee_ttt_state& s, int par, int act, float pr) {
        state.push_back(s); parent.push_back(par); first_child.push_back(-1); n_children.push_back(0); action.push_back(act);
        visits.push_back(0.f); value.push_back(0.f); prior.push_back(pr); drained.push_back(0);
        return (int)state.size() - 1;
    }
    size_t size() const { return state.size(); }
};

// Node::alpha_ucb, node.rs:98-112, f32 in this association
inline float ucb(const Store& T, int idx, float c) {
    const float n = T.visits[idx];
    const float q = n == 0.f ? 0.f : T.value[idx] / n;
    const float t = std::sqrt(T.visits[T.parent[idx]]) / (n + 1.0f);
    const float u = c * t;
    const float w = u * T.prior[idx];
    return q + w;
}
// alpha_select_leaf_node + select_alpha, alpha_mcts.rs:14-33: Iterator::max_by keeps the LAST of equal maxima; NaN compares Equal
int select_leaf(const Store& T, int root, float c, int& depth) {
    int idx = root; depth = 0;
    while (T.n_children[idx] != 0) {
        int best = T.first_child[idx];
        float ub = ucb(T, best, c);
        for (int j = 1; j < T.n_children[idx]; ++j) {
            const int ch = T.first_child[idx] + j;
            const float un = ucb(T, ch, c);
            if (!(ub > un)) { best = ch; ub = un; }
        }
        idx = best; ++depth;
    }
    return idx;
}
void backpropagate(Store& T, int idx, float v) {                       // simple_mcts.rs:96-103: same sign at every level
    for (; idx >= 0; idx = T.parent[idx]) { T.visits[idx] += 1.0f; T.value[idx] += v; }
}
// turn_policy_to_probs_tensor (utils.rs:74-84) + alpha_expand_tensor (node.rs:157-174)
void expand(Store& T, int idx, const float* policy_row, diee_stats* st) {
    if (T.drained[idx]) return;
    const diee_ttt_state s = T.state[idx];
    uint8_t mv[9];
    const int k = valid_moves(s, mv);
    float sum = 0.f;
    for (int j = 0; j < k; ++j) sum += policy_row[mv[j]];              // encode(action) = action, mod.rs:100-102
    const int first = (int)T.size();
    for (int j = 0; j < k; ++j) {
        diee_ttt_state ns = s;
        apply_move(ns, mv[j]);
        T.add(ns, idx, mv[j], policy_row[mv[j]] / sum);
    }
    T.first_child[idx] = k ? first : -1; T.n_children[idx] = k; T.drained[idx] = 1;
    if (st) { st->expansions += 1; st->children += (uint64_t)k; if ((uint64_t)k > st->max_children) st->max_children = (uint64_t)k; }
}

void search(const Engine& net, Store& T, const diee_ttt_state* roots, int n, const diee_mcts_cfg& cfg, uint64_t seed, uint32_t step,
            bool quirks, diee_stats* st) {
    std::vector<float> policy((size_t)n * A), value(n), noise(A);
    std::vector<diee_ttt_state> batch(roots, roots + n);
    net.forward(batch.data(), (uint32_t)n, policy.data(), value.data());                   // forward_policy, :104
    if (st) st->nn_evals += (uint64_t)n;
    dirichlet_host(seed, step, cfg.dir_alpha, A, noise.data());                            // apply_dirichlet, noise.rs:27-34: ONE sample for all rows
    const float eps = cfg.dir_eps, om = 1.0f - eps;
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < A; ++a) {
            const float x = om * policy[(size_t)i * A + a], y = eps * noise[a];
            policy[(size_t)i * A + a] = x + y;
        }
    for (int i = 0; i < n; ++i) T.add(roots[i], -1, -1, 0.f);                              // :110-112
    for (int i = 0; i < n; ++i) { T.visits[i] = 1.0f; expand(T, i, &policy[(size_t)i * A], st); }   // :119-127
    std::vector<int> sel(n, 0);                                                            // :142 vec![0; n]
    std::vector<uint8_t> fresh(n, 0);
    for (uint32_t it = 0; it < cfg.iterations; ++it) {                                     // :149
        bool node_selected = false;
        for (int g = 0; g < n; ++g) {                                                      // :153-168
            int depth = 0, winner = 0;
            const int idx = select_leaf(T, g, cfg.c, depth);
            if (st) { st->selections += 1; st->depth_sum += (uint64_t)depth; }
            fresh[g] = 0;
            if (check_winner(T.state[idx], winner)) {
                const int rp = T.state[g].player;
                backpropagate(T, idx, winner == rp ? 1.0f : (winner == -rp ? -1.0f : 0.0f));
                if (st) st->terminal_hits += 1;
            } else { node_selected = true; sel[g] = idx; fresh[g] = 1; }
        }
        if (!node_selected) continue;                                                      // :170-172
        for (int g = 0; g < n; ++g) batch[g] = T.state[sel[g]];                            // :175-183, stale slots included
        net.forward(batch.data(), (uint32_t)n, policy.data(), value.data());               // :186
        if (st) st->nn_evals += (uint64_t)n;
        for (int slot = 0; slot < n; ++slot) {                                             // :192-200
            if (!quirks && !fresh[slot]) continue;                                         // Q14 off: no stale re-backpropagation
            expand(T, sel[slot], &policy[(size_t)slot * A], st);
            backpropagate(T, sel[slot], value[slot]);
        }
    }
}

// get_prob_tensor_parallel, utils.rs:42-58 (row sums taken over the children in order)
void root_probs(const Store& T, int i, float* row) {
    for (int a = 0; a < A; ++a) row[a] = T.n_children[i] ? 0.f : NAN;
    float sum = 0.f;
    for (int j = 0; j < T.n_children[i]; ++j) sum += T.visits[T.first_child[i] + j];
    for (int j = 0; j < T.n_children[i]; ++j) row[T.action[T.first_child[i] + j]] = T.visits[T.first_child[i] + j] / sum;
}

int weighted_select(const float* w, double u01) {                      // alphazero.rs:129-137, rand WeightedIndex over f64
    double total = 0.0;
    for (int a = 0; a < A; ++a) total += (double)w[a];
    const double x = u01 * total;
    double cum = 0.0;
    int last_nz = 0;
    for (int a = 0; a < A; ++a) {
        if (w[a] != 0.0f) last_nz = a;
        cum += (double)w[a];
        if (cum > x) return a;
    }
    return last_nz;
}
}  // namespace

void Engine::mcts_batch(const diee_ttt_state* roots, uint32_t n, const diee_mcts_cfg& cfg, uint64_t seed, uint32_t step,
                        const uint32_t*, const uint32_t*, uint32_t flags, float* visit_probs, uint32_t* n_children, float* root_visits,
                        diee_stats* stats) const {
    if (stats) memset(stats, 0, sizeof *stats);
    const auto t0 = std::chrono::steady_clock::now();
    Store T;
    search(*this, T, roots, (int)n, cfg, seed, step, (flags & DIEE_FLAG_REF_QUIRKS) != 0, stats);
    for (uint32_t i = 0; i < n; ++i) {
        root_probs(T, (int)i, visit_probs + (size_t)i * A);
        if (n_children) n_children[i] = (uint32_t)T.n_children[i];
        if (root_visits) root_visits[i] = T.visits[i];
    }
    if (stats) { stats->nn_rows = stats->nn_evals; stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
}

void Engine::self_play(uint32_t n_games, uint32_t first_game_id, const diee_mcts_cfg& cfg, float temperature, uint64_t seed,
                       uint32_t flags, uint32_t max_steps, diee_fragments* out, diee_stats* stats) const {
    if (!loaded()) throw EngineError(DIEE_ERR_NO_WEIGHTS, "diee_load_weights has not been called");
    if (out) memset(out, 0, sizeof *out);
    if (stats) memset(stats, 0, sizeof *stats);
    const bool quirks = (flags & DIEE_FLAG_REF_QUIRKS) != 0;
    const float inv_t = (float)(1.0 / (double)temperature);                                // :165 pow_(1.0 / temperature)
    const auto t0 = std::chrono::steady_clock::now();
    struct Frag { int8_t player; float ps[A]; float st[PLANES]; };
    std::vector<diee_ttt_state> states(n_games);
    std::vector<uint32_t> rounds(n_games, 0);
    std::vector<uint8_t> live(n_games, 1);
    std::vector<std::vector<Frag>> mem(n_games);
    std::vector<int8_t> o_outcome; std::vector<float> o_ps, o_state; std::vector<uint32_t> o_game;
    auto flush = [&](uint32_t g, int winner, bool zero) {
        for (const Frag& f : mem[g]) {
            o_outcome.push_back(zero ? 0 : (winner == f.player ? 1 : (winner == -f.player ? -1 : 0)));   // :216-217
            o_ps.insert(o_ps.end(), f.ps, f.ps + A); o_state.insert(o_state.end(), f.st, f.st + PLANES);
            o_game.push_back(first_game_id + g);
        }
    };
    for (uint32_t g = 0; g < n_games; ++g) { memset(&states[g], 0, sizeof states[g]); states[g].player = -1; }   // TicTacToe::new, mod.rs:28-30
    uint32_t n_live = n_games, steps = 0;
    std::vector<diee_ttt_state> roots; std::vector<uint32_t> ids;
    std::vector<float> row(A);
    while (n_live > 0 && (max_steps == 0 || steps < max_steps)) {                          // alpha_parallel.rs:129
        roots.clear(); ids.clear();
        for (uint32_t g = 0; g < n_games; ++g) if (live[g]) { roots.push_back(states[g]); ids.push_back(g); }
        Store T;                                                                           // :137 a fresh NodeStore every move-step
        search(*this, T, roots.data(), (int)roots.size(), cfg, seed, steps, quirks, stats);   // :146
        for (size_t pi = 0; pi < ids.size(); ++pi) {                                       // :168-224
            const uint32_t g = ids[pi];
            bool removed = false, flushed = false;
            if (rounds[g] >= cfg.round_limit) { flush(g, 0, true); removed = true; flushed = true; }      // :172-180, no `continue`
            if (T.n_children[pi] == 0) {                                                   // :183-189
                rounds[g] += 1; skip_turn(states[g]);
                if (stats) stats->plies += 1;
                if (removed) { live[g] = 0; --n_live; if (stats) stats->games += 1; }
                continue;
            }
            root_probs(T, (int)pi, row.data());                                            // :164
            for (int a = 0; a < A; ++a) row[a] = det_powf(row[a], inv_t);                  // :165, not renormalised (Q17)
            const int a = weighted_select(row.data(), draw_uniform(seed, first_game_id + g, rounds[g], kTagSample, 0u));   // :192
            Frag f; f.player = states[g].player;                                           // :195-199
            memcpy(f.ps, row.data(), sizeof f.ps); planes(states[g], f.st);
            mem[g].push_back(f);
            uint8_t vm[9];                                                                 // :202-210 decode, assert legal, apply
            const int k = valid_moves(states[g], vm);
            bool ok = false;
            for (int j = 0; j < k; ++j) ok = ok || vm[j] == (uint8_t)a;
            if (!ok && stats) stats->illegal_decodes += 1;
            apply_move(states[g], (uint8_t)a);
            rounds[g] += 1;                                                                // :213
            if (stats) stats->plies += 1;
            int winner = 0;
            if (check_winner(states[g], winner)) {                                         // :215-223
                if (!(flushed && !quirks)) flush(g, winner, false);                        // Q18: the reference flushes twice
                removed = true;
            }
            if (removed) { live[g] = 0; --n_live; if (stats) stats->games += 1; }
        }
        ++steps;
    }
    const size_t nf = o_outcome.size();
    if (stats) {
        stats->move_steps = steps; stats->fragments = nf; stats->nn_rows = stats->nn_evals;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (!out || nf == 0) return;
    out->outcome = (int8_t*)malloc(nf); out->ps = (float*)malloc(nf * A * sizeof(float));
    out->state = (float*)malloc(nf * PLANES * sizeof(float)); out->game = (uint32_t*)malloc(nf * sizeof(uint32_t));
    if (!out->outcome || !out->ps || !out->state || !out->game) { diee_free_fragments(out); throw std::bad_alloc(); }
    memcpy(out->outcome, o_outcome.data(), nf); memcpy(out->ps, o_ps.data(), nf * A * sizeof(float));
    memcpy(out->state, o_state.data(), nf * PLANES * sizeof(float)); memcpy(out->game, o_game.data(), nf * sizeof(uint32_t));
    out->n = (uint32_t)nf;
}

}  // namespace ttt
}  // namespace diee
