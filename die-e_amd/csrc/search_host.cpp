// search_host.cpp -- host driver of the batched search and of self-play.
// Reference call structure: alpha_mcts_parallel (src/mcts/alpha_mcts.rs:91-202) and
// self_play_parallel (src/alphazero/alpha_parallel.rs:101-231).  The host only enqueues kernels:
// per move-step there is ONE device->host read (the live-game count); the reference crosses the
// host<->device boundary twice per MCTS iteration.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include "bg_device.h"
#include "engine.h"
#include "launch.h"
#include "nn_host.h"

namespace diee {

struct SearchBufs {
    // tree
    DevBuf<float> visits, value, prior;
    DevBuf<uint32_t> parent, first_child, meta, used;
    DevBuf<BgState> nstate;
    uint32_t node_cap = 0, slot_cap = 0;
    // slots
    DevBuf<BgState> roots, eval_states;
    DevBuf<uint32_t> game_id, round, leaf, sel, iter_flags;
    DevBuf<float> sel_value, noise, root_value0;
    DevBuf<uint8_t> leaf_term;
    DevBuf<unsigned long long> counters;
    DevBuf<uint32_t> slot_cnt;
    uint32_t iter_cap = 0;
    // games
    DevBuf<BgState> gstate;
    DevBuf<uint32_t> rounds, nfrags, ev_a_count, ev_a_step, ev_b_count, ev_b_step, live, n_live_dev;
    DevBuf<uint8_t> alive;
    DevBuf<int8_t> winner, frag_player;
    DevBuf<float> frag_ps, frag_planes;
    uint32_t game_cap = 0, frag_cap = 0;
    // output staging
    DevBuf<uint32_t> out_src;
    DevBuf<float> out_ps, out_planes;
};

void free_search(SearchBufs* s) { delete s; }

namespace {

// Dirichlet(alpha * 1_n): Gamma(alpha,1) draws normalised, as rand_distr 0.4.3 does (Cargo.toml:19,
// call site noise.rs:29-30): shape < 1 via Gamma(shape+1) * U^(1/shape), shape >= 1 by
// Marsaglia-Tsang; normals by Box-Muller.  Drawn on the HOST (like the reference) from Philox.
struct DirRng {
    uint64_t seed; uint32_t step; uint32_t n;
    double u01() {
        uint32_t o[4];
        philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), n++, step, kTagDirichlet, 0u, o);
        const uint64_t x = ((uint64_t)o[1] << 32) | o[0];
        return ((double)(x >> 12) + 0.5) * (1.0 / 4503599627370496.0);
    }
    double normal() {
        const double u1 = u01(), u2 = u01();
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    double gamma_large(double shape) {
        const double d = shape - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
        for (;;) {
            const double x = normal(), vc = 1.0 + c * x;
            if (vc <= 0.0) continue;
            const double v = vc * vc * vc, u = u01(), x2 = x * x;
            if (u < 1.0 - 0.0331 * x2 * x2 || std::log(u) < 0.5 * x2 + d * (1.0 - v + std::log(v))) return d * v;
        }
    }
};

void dirichlet_host(uint64_t seed, uint32_t step, float alpha, int n, float* out) {
    DirRng r{seed, step, 0};
    const double a = (double)alpha;
    std::vector<double> g((size_t)n);
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        double v;
        if (a < 1.0) { const double u = r.u01(); v = r.gamma_large(a + 1.0) * std::pow(u, 1.0 / a); }
        else v = r.gamma_large(a);
        g[(size_t)i] = v; sum += v;
    }
    for (int i = 0; i < n; ++i) out[i] = (float)(g[(size_t)i] / sum);
}

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = getenv(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

void reserve_search(Engine& e, uint32_t slots, uint32_t iterations) {
    if (!e.search) e.search = new SearchBufs();
    SearchBufs& B = *e.search;
    const uint32_t per_exp = env_u32("DIEE_NODES_PER_EXPANSION", 128);
    const uint32_t want_cap = (iterations + 1) * per_exp + 64;
    if (slots > B.slot_cap || want_cap > B.node_cap) {
        const uint32_t sc = std::max(slots, B.slot_cap), nc = std::max(want_cap, B.node_cap);
        const size_t N = (size_t)sc * nc;
        B.visits.ensure(N); B.value.ensure(N); B.prior.ensure(N);
        B.parent.ensure(N); B.first_child.ensure(N); B.meta.ensure(N); B.nstate.ensure(N);
        B.used.ensure(sc);
        B.roots.ensure(sc); B.eval_states.ensure(sc); B.game_id.ensure(sc); B.round.ensure(sc);
        B.leaf.ensure(sc); B.sel.ensure(sc); B.sel_value.ensure(sc); B.leaf_term.ensure(sc);
        B.noise.ensure(1352); B.root_value0.ensure(4); B.counters.ensure(CNT_COUNT); B.slot_cnt.ensure((size_t)sc * SC_COUNT);
        B.slot_cap = sc; B.node_cap = nc;
    }
    if (iterations + 1 > B.iter_cap) { B.iter_flags.ensure(2 * ((size_t)iterations + 1)); B.iter_cap = iterations + 1; }
}

Tree tree_view(SearchBufs& B) {
    return Tree{B.visits.p, B.value.p, B.prior.p, B.parent.p, B.first_child.p, B.meta.p, B.nstate.p, B.used.p, B.node_cap};
}
Slots slots_view(Engine& e, SearchBufs& B) {
    const NetHeads H = nn_heads(e, (int)B.slot_cap);      // the network's output buffers, sized for every slot
    return Slots{B.roots.p, B.eval_states.p, B.game_id.p, B.round.p, B.leaf.p, B.sel.p, B.sel_value.p, B.leaf_term.p,
                 H.logits, H.hv, H.wv, B.noise.p, B.root_value0.p, B.iter_flags.p, B.counters.p, B.slot_cnt.p, e.flags_dev.p};
}

// alpha_mcts_parallel on the n slots already loaded into B.roots / game_id / round
void mcts_run(Engine& e, uint32_t n, const diee_mcts_cfg& cfg, uint64_t seed, uint32_t step, uint32_t flags) {
    SearchBufs& B = *e.search;
    const Tree T = tree_view(B);
    const Slots S = slots_view(e, B);
    hipStream_t st = e.stream;
    const uint32_t quirks = (flags & DIEE_FLAG_REF_QUIRKS) ? 1u : 0u;
    if (cfg.iterations) HIPCHK(hipMemsetAsync(B.iter_flags.p, 0, sizeof(uint32_t) * 2 * (size_t)cfg.iterations, st));
    float noise[1352];
    dirichlet_host(seed, step, cfg.dir_alpha, 1352, noise);          // noise.rs:27-34: one sample per move-step
    HIPCHK(hipMemcpyAsync(B.noise.p, noise, sizeof noise, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));                                // `noise` is a stack buffer
    launch_init_roots(st, T, S, n);
    nn_forward(e, B.eval_states.p, (int)n, nullptr, nullptr);        // forward_policy, alpha_mcts.rs:104 (softmax / tanh in k_expand)
    const SearchParams P{seed, cfg.dir_eps, quirks};
    // one MCTS kernel per network evaluation: expand + backpropagate iteration it, then select for it+1
    launch_expand(st, T, S, n, kRootIteration, P, cfg.iterations ? 0u : kNoNextIteration, cfg.c);
    for (uint32_t it = 0; it < cfg.iterations; ++it) {               // alpha_mcts.rs:149
        nn_forward(e, B.eval_states.p, (int)n, nullptr, nullptr);       // alpha_mcts.rs:186
        launch_expand(st, T, S, n, it, P, it + 1 < cfg.iterations ? it + 1 : kNoNextIteration, cfg.c);
    }
    launch_reduce_counters(st, S, n);
    HIPCHK(hipGetLastError());
}

void read_counters(Engine& e, diee_stats* stats) {
    if (!stats) return;
    unsigned long long c[CNT_COUNT];
    e.d2h(c, e.search->counters.p, (size_t)CNT_COUNT);
    e.sync();
    stats->nn_evals = c[CNT_NN_EVALS]; stats->expansions = c[CNT_EXPANSIONS]; stats->children = c[CNT_CHILDREN];
    stats->terminal_hits = c[CNT_TERMINAL]; stats->depth_sum = c[CNT_DEPTH_SUM]; stats->selections = c[CNT_SELECTIONS];
    stats->illegal_decodes = c[CNT_ILLEGAL]; stats->max_children = c[CNT_MAX_CHILDREN];
    stats->plies = c[CNT_PLIES]; stats->games = c[CNT_GAMES];
}

}  // namespace

void Engine::mcts_batch(const diee_bg_state* roots, uint32_t n, const diee_mcts_cfg* cfg, uint64_t seed, uint32_t step,
                        const uint32_t* game_ids, const uint32_t* rounds, uint32_t flags, float* visit_probs,
                        uint32_t* n_children, float* root_visits, diee_stats* stats) {
    HIPCHK(hipSetDevice(device));
    if (!net || !net->loaded) throw EngineError(DIEE_ERR_NO_WEIGHTS, "diee_load_weights has not been called");
    if (stats) memset(stats, 0, sizeof *stats);
    if (!n) return;
    for (uint32_t i = 0; i < n; ++i)
        if (roots[i].roll[0] == 0 && roots[i].roll[1] == 0) throw EngineError(DIEE_ERR_ARG, "die has not been rolled (backgammon_logic.rs:404)");
    reserve_search(*this, n, cfg->iterations);
    SearchBufs& B = *search;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> ids(n), rds(n);
    for (uint32_t i = 0; i < n; ++i) { ids[i] = game_ids ? game_ids[i] : i; rds[i] = rounds ? rounds[i] : 0; }
    h2d((uint8_t*)B.roots.p, (const uint8_t*)roots, (size_t)n * 32);
    h2d(B.game_id.p, ids.data(), (size_t)n); h2d(B.round.p, rds.data(), (size_t)n);
    HIPCHK(hipMemsetAsync(B.counters.p, 0, sizeof(unsigned long long) * CNT_COUNT, stream));
    sync();
    const int se = net->sample_every; net->sample_every = 0;
    mcts_run(*this, n, *cfg, seed, step, flags);
    net->sample_every = se;
    tmp_a.ensure((size_t)n * 1352 * 4); tmp_b.ensure((size_t)n * 4); tmp_c.ensure((size_t)n * 4);
    launch_root_probs(stream, tree_view(B), n, (float*)tmp_a.p, (uint32_t*)tmp_b.p, (float*)tmp_c.p);
    HIPCHK(hipGetLastError());
    d2h((uint8_t*)visit_probs, tmp_a.p, (size_t)n * 1352 * 4);
    if (n_children) d2h((uint8_t*)n_children, tmp_b.p, (size_t)n * 4);
    if (root_visits) d2h((uint8_t*)root_visits, tmp_c.p, (size_t)n * 4);
    sync();
    if (stats) {
        read_counters(*this, stats);
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    check_overflow();
}

void Engine::self_play(uint32_t n_games, uint32_t first_game_id, const diee_mcts_cfg* cfg, float temperature,
                       uint64_t seed, uint32_t flags, uint32_t max_steps, diee_fragments* out, diee_stats* stats) {
    HIPCHK(hipSetDevice(device));
    if (!net || !net->loaded) throw EngineError(DIEE_ERR_NO_WEIGHTS, "diee_load_weights has not been called");
    if (out) memset(out, 0, sizeof *out);
    if (stats) memset(stats, 0, sizeof *stats);
    if (cfg->iterations == 0) throw EngineError(DIEE_ERR_ARG, "iterations must be >= 1");
    reserve_search(*this, n_games, cfg->iterations);
    nn_reserve(*this, (int)n_games);
    SearchBufs& B = *search;
    const uint32_t frag_cap = cfg->round_limit + 2;
    if (n_games > B.game_cap || frag_cap > B.frag_cap) {
        const uint32_t gc = std::max(n_games, B.game_cap), fc = std::max(frag_cap, B.frag_cap);
        B.gstate.ensure(gc); B.rounds.ensure(gc); B.nfrags.ensure(gc); B.alive.ensure(gc); B.winner.ensure(gc);
        B.ev_a_count.ensure(gc); B.ev_a_step.ensure(gc); B.ev_b_count.ensure(gc); B.ev_b_step.ensure(gc);
        B.live.ensure(gc); B.n_live_dev.ensure(4);
        B.frag_ps.ensure((size_t)gc * fc * 1352); B.frag_planes.ensure((size_t)gc * fc * 144);
        B.frag_player.ensure((size_t)gc * fc);
        B.game_cap = gc; B.frag_cap = fc;
    }
    const Games Gm{B.gstate.p, B.rounds.p, B.nfrags.p, B.alive.p, B.winner.p, B.ev_a_count.p, B.ev_a_step.p,
                   B.ev_b_count.p, B.ev_b_step.p, B.live.p, B.frag_ps.p, B.frag_planes.p, B.frag_player.p,
                   B.frag_cap, B.counters.p};
    const Tree T = tree_view(B);
    const Slots S = slots_view(*this, B);
    const uint32_t quirks = (flags & DIEE_FLAG_REF_QUIRKS) ? 1u : 0u;
    const PlayParams PP{seed, first_game_id, cfg->round_limit, (float)(1.0 / (double)temperature), quirks};

    HIPCHK(hipMemsetAsync(B.counters.p, 0, sizeof(unsigned long long) * CNT_COUNT, stream));
    nn_reset_timing(*this);
    launch_init_games(stream, Gm, n_games, first_game_id, seed);
    sync();
    // ---- timed region: every input is resident in HBM ----
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t n_live = n_games, step = 0;
    const bool trace_steps = getenv("DIEE_TRACE_STEPS") != nullptr;      // development: batch size of every move-step
    while (n_live > 0 && (max_steps == 0 || step < max_steps)) {       // alpha_parallel.rs:129
        if (trace_steps) fprintf(stderr, "[diee] move-step %u: %u games alive\n", step, n_live);
        launch_gather_roots(stream, Gm, S, n_live, first_game_id);
        mcts_run(*this, n_live, *cfg, seed, step, flags);               // :146
        launch_play_move(stream, T, Gm, n_live, step, PP);              // :164-224
        launch_compact_live(stream, Gm, n_live, B.n_live_dev.p);        // :226-228
        HIPCHK(hipGetLastError());
        d2h(&n_live, B.n_live_dev.p, 1);
        sync();
        ++step;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    check_overflow();
    nn_harvest(*this, stats);
    if (stats) {
        read_counters(*this, stats);
        stats->move_steps = step; stats->seconds = secs;
    }

    // ---- outputs: order = (move-step of the flush, game), round-limit flush before win flush ----
    std::vector<uint32_t> nfr(n_games), ea(n_games), eas(n_games), eb(n_games), ebs(n_games);
    std::vector<int8_t> win(n_games);
    d2h(nfr.data(), B.nfrags.p, (size_t)n_games); d2h(ea.data(), B.ev_a_count.p, (size_t)n_games);
    d2h(eas.data(), B.ev_a_step.p, (size_t)n_games); d2h(eb.data(), B.ev_b_count.p, (size_t)n_games);
    d2h(ebs.data(), B.ev_b_step.p, (size_t)n_games); d2h(win.data(), B.winner.p, (size_t)n_games);
    sync();
    struct Ev { uint32_t step, g, kind, count; };
    std::vector<Ev> evs;
    size_t total = 0;
    for (uint32_t g = 0; g < n_games; ++g) {
        if (ea[g] != 0xFFFFFFFFu) { evs.push_back({eas[g], g, 0, ea[g]}); total += ea[g]; }
        if (eb[g] != 0xFFFFFFFFu) { evs.push_back({ebs[g], g, 1, eb[g]}); total += eb[g]; }
    }
    std::sort(evs.begin(), evs.end(), [](const Ev& a, const Ev& b) {
        if (a.step != b.step) return a.step < b.step;
        if (a.g != b.g) return a.g < b.g;
        return a.kind < b.kind;
    });
    if (stats) stats->fragments = total;
    if (!out || total == 0) return;
    std::vector<int8_t> players((size_t)n_games * B.frag_cap);
    d2h(players.data(), B.frag_player.p, players.size());
    sync();
    std::vector<uint32_t> src(total);
    out->outcome = (int8_t*)malloc(total);
    out->ps = (float*)malloc(total * 1352 * sizeof(float));
    out->state = (float*)malloc(total * 144 * sizeof(float));
    out->game = (uint32_t*)malloc(total * sizeof(uint32_t));
    if (!out->outcome || !out->ps || !out->state || !out->game) {
        free(out->outcome); free(out->ps); free(out->state); free(out->game); memset(out, 0, sizeof *out);
        throw std::bad_alloc();
    }
    size_t k = 0;
    for (const Ev& ev : evs)
        for (uint32_t r = 0; r < ev.count; ++r, ++k) {
            const size_t si = (size_t)ev.g * B.frag_cap + r;
            src[k] = (uint32_t)si;
            const int pl = players[si];
            out->outcome[k] = ev.kind == 0 ? 0 : (win[ev.g] == pl ? 1 : (win[ev.g] == -pl ? -1 : 0));   // :216-217
            out->game[k] = first_game_id + ev.g;
        }
    out->n = (uint32_t)total;
    // gather on the device in chunks, then copy out
    const size_t chunk = 65536;
    B.out_src.ensure(chunk); B.out_ps.ensure(chunk * 1352); B.out_planes.ensure(chunk * 144);
    for (size_t o = 0; o < total; o += chunk) {
        const size_t m = std::min(chunk, total - o);
        h2d(B.out_src.p, src.data() + o, m);
        launch_gather_frags(stream, Gm, B.out_src.p, (uint32_t)m, B.out_ps.p, B.out_planes.p);
        HIPCHK(hipGetLastError());
        d2h(out->ps + o * 1352, B.out_ps.p, m * 1352);
        d2h(out->state + o * 144, B.out_planes.p, m * 144);
        sync();
    }
}

}  // namespace diee
