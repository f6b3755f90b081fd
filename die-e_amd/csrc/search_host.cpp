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
#include <map>
#include <mutex>
#include <unordered_map>

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
    DevBuf<uint32_t> game_id, round, seg, leaf, sel, iter_flags, row_slot, slot_row, n_rows;
    DevBuf<float> sel_value, noise, root_value0;
    DevBuf<uint8_t> leaf_term, path_len;
    DevBuf<uint32_t> path, leaf_meta, grow_k;
    DevBuf<uint16_t> grow_code;
    DevBuf<unsigned long long> counters, counters_bak;
    DevBuf<uint32_t> slot_cnt;
    uint32_t iter_cap = 0;
    DevBuf<unsigned long long> step_log;                       // [2 * kStepLog] per move-step { expansions, rows evaluated } (k_reduce_counters)
    uint32_t cur_step = 0;
    // the tail of a batch (search_types.h, Tail)
    DevBuf<uint32_t> tl_crow, tl_rows_node, tl_words;          // tl_words = n_rows[launches] ++ state[4] ++ bar[2 * launches + 8] ++ bar2[launches] ++ n_dem[launches]
    DevBuf<float> tl_cval, tl_logits, tl_hv;
    DevBuf<BgState> tl_rows_state;
    uint32_t* tl_host = nullptr;                               // pinned, [4]
    uint32_t tl_launches = 0, tl_node_cap = 0, tl_rows_cap = 0;
    uint32_t tl_prev_need = 0;                                 // tower launches the previous move-step's search needed (sizes the first chunk)
    uint64_t tl_iterations = 0, tl_launched = 0, tl_with_rows = 0, tl_spec_rows = 0, tl_syncs = 0;      // this call's totals
    // the free-running search (search_types.h, Free)
    DevBuf<uint32_t> fr_crow, fr_rows_idx, fr_words, fr_slots, fr_wish;      // fr_words = n_rows[launches] ++ n_dem[launches] ++ state[8]; fr_slots = grant_off ++ grant_cnt ++ wish_n ++ prog ++ first_sel, [slots] each
    DevBuf<float> fr_cval, fr_logits, fr_hv;
    DevBuf<BgState> fr_rows_state;                              // dense states of the launches of at most 128 rows (cluster family)
    uint32_t* fr_host = nullptr;                                // pinned, [2]
    uint32_t fr_launches = 0, fr_ring = 0, fr_rows = 0, fr_slot_cap = 0, fr_node_cap = 0;
    uint32_t fr_prev_need = 0;
    int cus = 0;                                                // compute units of the device (how many games' workgroups share one)
    // batches ("segments") in flight
    DevBuf<unsigned long long> seg_seed;
    DevBuf<uint32_t> seg_first_id, seg_game0, seg_slots;      // seg_slots = first_slot[kMaxSegments] ++ end_slot[kMaxSegments]
    float* noise_host = nullptr;                               // pinned, [2][kMaxSegments][1352]: upload without a sync
    uint32_t* live_host = nullptr;                             // pinned, [2 + kMaxSegments]: n_live, per-batch live, flag word
    // games
    DevBuf<BgState> gstate;
    DevBuf<uint32_t> rounds, nfrags, ev_a_count, ev_a_step, ev_b_count, ev_b_step, live, n_live_dev;
    DevBuf<uint8_t> alive, gseg;
    DevBuf<int8_t> winner, frag_player;
    DevBuf<float> frag_ps, frag_planes;
    uint32_t game_cap = 0, frag_cap = 0;
    // output delivery: this step's flushes (double-buffered: the copy stream reads step s's list while step s + 1's is written),
    // one staging buffer (the copy stream runs gather -> copies out -> next gather in order), the stream and its guards
    DevBuf<DeliverEvent> dl_events[2];
    DevBuf<float> dl_ps, dl_planes;
    DevBuf<int8_t> dl_outcome;
    DevBuf<uint32_t> dl_game;
    uint32_t dl_rows = 0;                                      // staging rows
    hipStream_t copy = nullptr;
    hipEvent_t ev_copy[2] = {nullptr, nullptr};                // the copy stream is done with dl_events[i]
    bool ev_copy_armed[2] = {false, false};
    ~SearchBufs() {
        if (copy) (void)hipStreamDestroy(copy);
        for (hipEvent_t e : ev_copy) if (e) (void)hipEventDestroy(e);
        if (tl_host) (void)hipHostFree(tl_host);
        if (fr_host) (void)hipHostFree(fr_host);
        if (noise_host) (void)hipHostFree(noise_host);
        if (live_host) (void)hipHostFree(live_host);
    }
};

void free_search(SearchBufs* s) { delete s; }

// ---- pinned host memory for the delivered fragments ------------------------------------------------------------------
// The arrays of a diee_fragments are page-locked so that the copies out of HBM are real asynchronous DMA (pageable
// destinations are staged by the runtime and block the host).  Pinning costs ~0.3 ms per MB, so blocks are kept when
// diee_free_fragments hands them back (up to option pinned_pool_mb, default 8192 MiB) and the next call takes them again.
namespace {
struct PinnedPool {
    std::mutex m;
    std::multimap<size_t, void*> idle;                 // size -> block
    std::unordered_map<void*, size_t> out;             // blocks handed out
    size_t idle_bytes = 0;
    // process-wide: option pinned_pool_mb of any ctx (the environment's DIEE_PINNED_POOL_MB through it, read when a ctx is created)
    size_t cap = (size_t)8192 << 20;
    size_t cap_bytes() { std::lock_guard<std::mutex> lk(m); return cap; }
    void* acquire(size_t bytes) {
        bytes = (std::max<size_t>(bytes, 1) + 0xFFFFFull) & ~(size_t)0xFFFFFull;      // 1 MiB granules
        {
            std::lock_guard<std::mutex> lk(m);
            auto it = idle.lower_bound(bytes);
            if (it != idle.end() && it->first <= 2 * bytes + (64u << 20)) {
                void* p = it->second; const size_t sz = it->first;
                idle.erase(it); idle_bytes -= sz; out[p] = sz;
                return p;
            }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, bytes) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
        std::lock_guard<std::mutex> lk(m);
        out[p] = bytes;
        return p;
    }
    size_t size_of(void* p) { std::lock_guard<std::mutex> lk(m); auto it = out.find(p); return it == out.end() ? 0 : it->second; }
    bool release(void* p) {                              // false: not one of ours
        if (!p) return true;
        size_t sz;
        {
            std::lock_guard<std::mutex> lk(m);
            auto it = out.find(p);
            if (it == out.end()) return false;
            sz = it->second; out.erase(it);
            if (idle_bytes + sz <= cap) { idle.emplace(sz, p); idle_bytes += sz; return true; }
        }
        (void)hipHostFree(p);
        return true;
    }
};
PinnedPool& pinned_pool() { static PinnedPool* pool = new PinnedPool(); return *pool; }   // (never destroyed: the HIP runtime may be gone first)
}  // namespace
bool pinned_release(void* p) { return pinned_pool().release(p); }
void pinned_pool_set_cap_mb(size_t mb) {
    PinnedPool& P = pinned_pool();
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> lk(P.m);
        P.cap = mb << 20;
        while (P.idle_bytes > P.cap && !P.idle.empty()) {       // a lower cap gives idle blocks back at once, largest first
            auto it = std::prev(P.idle.end());
            drop.push_back(it->second); P.idle_bytes -= it->first; P.idle.erase(it);
        }
    }
    for (void* p : drop) (void)hipHostFree(p);
}
size_t pinned_pool_cap_mb() { return pinned_pool().cap_bytes() >> 20; }

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

}  // namespace
void dirichlet_host(uint64_t seed, uint32_t step, float alpha, int n, float* out) {      // (also drawn by ttt_host.cpp)
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
namespace {

constexpr uint32_t kStepLog = 2048;                          // move-steps of a call with a log entry of their own (later ones share the last)
constexpr uint32_t kLiveWords = 1 + 4 * kMaxSegments;        // DeliverSummary at its largest; live_host[kLiveWords] = the flag word

void reserve_search(Engine& e, uint32_t slots, uint32_t iterations) {
    if (!e.search) e.search = new SearchBufs();
    SearchBufs& B = *e.search;
    const uint32_t per_exp = std::max<uint32_t>(e.opt.nodes_per_expansion, 1);
    const uint32_t want_cap = (iterations + 1) * per_exp + 64;
    if (slots > B.slot_cap || want_cap > B.node_cap) {
        const uint32_t sc = std::max(slots, B.slot_cap), nc = std::max(want_cap, B.node_cap);
        const size_t N = (size_t)sc * nc;
        B.visits.ensure(N); B.value.ensure(N); B.prior.ensure(N);
        B.parent.ensure(N); B.first_child.ensure(N); B.meta.ensure(N); B.nstate.ensure(N);
        B.used.ensure(sc);
        B.roots.ensure(sc); B.eval_states.ensure(sc); B.game_id.ensure(sc); B.round.ensure(sc); B.seg.ensure(sc);
        B.leaf.ensure(sc); B.sel.ensure(sc); B.sel_value.ensure(sc); B.leaf_term.ensure(sc);
        B.path.ensure((size_t)sc * kPathCap); B.path_len.ensure(sc); B.leaf_meta.ensure(sc);
        B.grow_k.ensure(sc); B.grow_code.ensure((size_t)sc * kMaxPlays);
        B.row_slot.ensure(sc); B.slot_row.ensure(sc); B.n_rows.ensure(4);
        B.slot_cnt.ensure((size_t)sc * SC_COUNT);
        B.slot_cap = sc; B.node_cap = nc;
    }
    if (!B.noise.p) {
        B.noise.ensure((size_t)kMaxSegments * 1352); B.root_value0.ensure(kMaxSegments);
        B.counters.ensure((size_t)kMaxSegments * CNT_COUNT); B.counters_bak.ensure((size_t)kMaxSegments * CNT_COUNT);
        B.seg_seed.ensure(kMaxSegments); B.seg_first_id.ensure(kMaxSegments); B.seg_game0.ensure(kMaxSegments);
        B.seg_slots.ensure(2 * kMaxSegments); B.n_live_dev.ensure(kLiveWords);
        HIPCHK(hipHostMalloc((void**)&B.noise_host, sizeof(float) * 2 * kMaxSegments * 1352));
        HIPCHK(hipHostMalloc((void**)&B.live_host, sizeof(uint32_t) * (kLiveWords + 1)));
    }
    if (!B.step_log.p) B.step_log.ensure(2 * (size_t)kStepLog);
    if (iterations + 1 > B.iter_cap) { B.iter_flags.ensure((size_t)kMaxSegments * 2 * ((size_t)iterations + 1)); B.iter_cap = iterations + 1; }
}

Tree tree_view(SearchBufs& B) {
    return Tree{B.visits.p, B.value.p, B.prior.p, B.parent.p, B.first_child.p, B.meta.p, B.nstate.p, B.used.p, B.node_cap};
}
Slots slots_view(Engine& e, SearchBufs& B) {
    const NetHeads H = nn_heads(e, (int)B.slot_cap);      // the network's output buffers, sized for every slot
    return Slots{B.roots.p, B.eval_states.p, B.game_id.p, B.round.p, B.seg.p, B.leaf.p, B.sel.p, B.sel_value.p, B.leaf_term.p,
                 nullptr, H.logits, H.hv, H.wv, B.noise.p, B.root_value0.p, B.iter_flags.p, B.counters.p, B.slot_cnt.p, e.flags_dev.p,
                 B.leaf_meta.p, B.path.p, B.path_len.p, B.grow_k.p, B.grow_code.p, std::min<uint32_t>(e.opt.path_cap, kPathCap)};
}
Segs segs_view(SearchBufs& B, uint32_t n_segs) {
    return Segs{B.seg_seed.p, B.seg_first_id.p, B.seg_game0.p, B.seg_slots.p, B.seg_slots.p + kMaxSegments, n_segs, B.iter_cap};
}

// the batches ("segments") of a call: seeds, RNG keys of their first games, positions of their first games
void upload_segments(Engine& e, const std::vector<diee_batch>& bt) {
    SearchBufs& B = *e.search;
    std::vector<unsigned long long> seeds(bt.size());
    std::vector<uint32_t> first(bt.size()), game0(bt.size());
    uint32_t g = 0;
    for (size_t k = 0; k < bt.size(); ++k) { seeds[k] = bt[k].seed; first[k] = bt[k].first_game_id; game0[k] = g; g += bt[k].n_games; }
    e.h2d(B.seg_seed.p, seeds.data(), seeds.size());
    e.h2d(B.seg_first_id.p, first.data(), first.size());
    e.h2d(B.seg_game0.p, game0.data(), game0.size());
    HIPCHK(hipMemsetAsync(B.counters.p, 0, sizeof(unsigned long long) * kMaxSegments * CNT_COUNT, e.stream));
    e.sync();                                                          // the vectors are stack-scoped
}

// one Dirichlet sample per batch and move-step (noise.rs:27-34), drawn on the host into pinned buffer `buf`
void draw_noise(SearchBufs& B, int buf, const std::vector<diee_batch>& bt, uint32_t step, float alpha) {
    for (size_t k = 0; k < bt.size(); ++k)
        dirichlet_host(bt[k].seed, step, alpha, 1352, B.noise_host + ((size_t)buf * kMaxSegments + k) * 1352);
}

// ---- the tail of a batch (search_types.h, Tail) -------------------------------------------------------------------------------
// rows of a tail launch by live games: a launch of 32 / 64 / 128 boards costs ~95 / 125 / 172 us, and a search needs about
// 91 / (1 + s) launches when every game gets s speculative rows per launch (s saturates near 8: the rows come from nodes that exist)
uint32_t tail_rows_for(const Engine& e, uint32_t n) {
    if (n >= e.opt.spec_fused_from || n > kTailRowsMax) return (uint32_t)kTailFusedRows;          // the fused family
    return n >= e.opt.spec_rows128_from ? 128u : n >= e.opt.spec_rows64_from ? 64u : 32u;
}
bool tail_possible(Engine& e, uint32_t n, const diee_mcts_cfg& cfg) {
    if (e.opt.spec_eval == 0 || n < 1 || cfg.iterations < 1) return false;
    const uint32_t rows = tail_rows_for(e, n);
    // k_tail's workgroups (one per game, > 100 KB of LDS each: one per compute unit) meet inside the launch: all of them must be resident together.
    // A device (or a partition of one) with fewer compute units than live games searches launch by launch instead of timing out at every meeting.
    if (!e.search->cus) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, e.device) == hipSuccess) e.search->cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    if (e.search->cus && n > (uint32_t)e.search->cus) return false;
    if (rows < n) return false;      // (options spec_rows64_from / spec_rows128_from can ask for fewer rows than games: every live game needs its demanded row)
    // the ring of evaluated rows grows with the iterations ((iterations + 1) launches x rows x (1352 + 72) floats; crow / cval: 256 x node_cap words):
    // beyond option spec_ring_mb the search runs one launch per iteration instead of failing an allocation in mid-batch
    const uint32_t ring_rows = rows > kTailRowsMax ? (uint32_t)kTailFusedRows : kTailRowsMax;
    const double ring_mb = ((double)(cfg.iterations + 1) * ring_rows * (1352 + 72 + 8 + 1) * 4.0 + (double)kTailMaxSlots * ((double)(cfg.iterations + 1) * std::max<uint32_t>(e.opt.nodes_per_expansion, 1) + 64) * 8.0) / 1048576.0;
    if (ring_mb > (double)e.opt.spec_ring_mb) return false;
    const bool in_reach = rows == (uint32_t)kTailFusedRows ? n <= std::min<uint32_t>(e.opt.spec_fused_games, kTailMaxSlots)
                                                           : n <= std::min<uint32_t>(e.opt.spec_max_games, kTailRowsMax);
    return in_reach && nn_tail_available(e, (int)rows, (int)n);
}

Tail tail_view(Engine& e, SearchBufs& B, const diee_mcts_cfg& cfg, uint32_t n) {
    const uint32_t launches = cfg.iterations + 1;
    const uint32_t rows_now = tail_rows_for(e, n) > kTailRowsMax ? (uint32_t)kTailFusedRows : kTailRowsMax;
    if (launches > B.tl_launches || B.node_cap > B.tl_node_cap || rows_now > B.tl_rows_cap) {
        const uint32_t L = std::max(launches, B.tl_launches), nc = std::max(B.node_cap, B.tl_node_cap), R = std::max(rows_now, B.tl_rows_cap);
        B.tl_crow.ensure((size_t)kTailMaxSlots * nc); B.tl_cval.ensure((size_t)kTailMaxSlots * nc);
        B.tl_rows_state.ensure((size_t)L * R); B.tl_rows_node.ensure((size_t)L * R);
        B.tl_logits.ensure((size_t)L * R * 1352); B.tl_hv.ensure((size_t)L * R * 72);
        B.tl_words.ensure((size_t)L + 4 + 2 * (size_t)L + 8 + (size_t)L + (size_t)L);
        B.tl_launches = L; B.tl_node_cap = nc; B.tl_rows_cap = R;
    }
    if (!B.tl_host) { HIPCHK(hipHostMalloc((void**)&B.tl_host, sizeof(uint32_t) * 4)); memset(B.tl_host, 0, sizeof(uint32_t) * 4); }
    uint32_t* w = B.tl_words.p;
    return Tail{B.tl_crow.p, B.tl_cval.p, B.tl_rows_state.p, B.tl_rows_node.p, B.tl_logits.p, B.tl_hv.p, w, w + 4 * (size_t)B.tl_launches + 12, w + B.tl_launches, w + B.tl_launches + 4,
                w + B.tl_launches + 4 + 2 * (size_t)B.tl_launches + 8, B.tl_host, launches, cfg.iterations, e.opt.spec_rollout_steps, tail_rows_for(e, n),
                e.opt.spec_child_rows, e.opt.spec_extra_rows, (uint32_t)e.opt.test_tail_skip};
}

// The iterations of one move-step's search for n <= kTailMaxSlots (option spec_max_games) live games, behind the root expansion (k_expand has selected every
// game's leaf for iteration 0): k_tail(0), then pairs of { tower launch q over the rows k_tail(q) planned, k_tail(q + 1) }.  Every pair
// completes at least one iteration, `iterations` pairs complete the search -- and far fewer do when the free rows of the launches
// carried the right nodes.  Launches sent ahead of a search that is complete return at once (~2 us each), so the pairs go out in
// chunks: as many as the previous move-step needed, then a few at a time, the host looking at the done word in between.
void tail_run(Engine& e, uint32_t n, const Tree& T, const Slots& S, const Segs& G, const diee_mcts_cfg& cfg, const SearchParams& P) {
    SearchBufs& B = *e.search;
    const Tail L = tail_view(e, B, cfg, n);
    hipStream_t st = e.stream;
    // crow of the slots in use (the trees are rebuilt every move-step), the row counts, state words and meeting words
    HIPCHK(hipMemsetAsync(L.crow, 0, sizeof(uint32_t) * (size_t)n * T.node_cap, st));
    HIPCHK(hipMemsetAsync(B.tl_words.p, 0, sizeof(uint32_t) * (5 * (size_t)B.tl_launches + 12), st));
    B.tl_host[1] = 0; B.tl_host[2] = 0xFFFFFFFFu;
    launch_tail(st, T, S, G, n, P, cfg.c, L, 0);
    uint32_t q = 0, sent = 0;
    uint32_t chunk = B.tl_prev_need ? B.tl_prev_need + 2 : std::max<uint32_t>(8u, cfg.iterations / 6);
    bool done = false;
    while (!done && q < cfg.iterations) {
        const uint32_t end = std::min<uint32_t>(cfg.iterations, q + std::max<uint32_t>(chunk, 1u));
        for (; q < end; ++q, ++sent) {
            if (!nn_forward_tail(e, L.rows_state + (size_t)q * L.rows, (int)L.rows, L.n_rows + q, L.hv + (size_t)q * L.rows * 72,
                                 L.logits + (size_t)q * L.rows * 1352, (int)n, L.n_dem + q))
                throw EngineError(DIEE_ERR_HIP, "tail search: the cluster tower could not be launched");
            launch_tail(st, T, S, G, n, P, cfg.c, L, q + 1);
        }
        HIPCHK(hipGetLastError());
        e.sync();
        ++B.tl_syncs;
        done = B.tl_host[1] != 0;
        chunk = 4;
    }
    uint32_t words[4] = {0, 0, 0, 0}, flag_word = 0;
    e.d2h(words, L.state, 4);
    e.d2h(&flag_word, e.flags_dev.p, 1);
    e.sync();
    // a meeting that timed out raised the starved bit (and left 2 in the state word): this search is void, cluster_starved() repeats it
    // launch by launch -- whatever the done word says (a workgroup that gave up left its game's record unsaved)
    const bool starved = (flag_word & 4u) != 0u || words[1] == 2u;
    if (L.test_skip) e.opt.test_tail_skip = 0;                       // (tests: one forced time-out per request)
    if (!done && !starved) throw EngineError(DIEE_ERR_HIP, "tail search: the iterations did not complete");
    if (starved) { B.tl_prev_need = 0; return; }
    if (e.opt.trace_steps)
        fprintf(stderr, "[diee] tail: %u games, %u iterations on %u launches with rows (%u pairs sent), %u speculative rows\n", n, cfg.iterations, words[2], sent, words[3]);
    B.tl_prev_need = words[2];
    B.tl_iterations += cfg.iterations; B.tl_launched += sent; B.tl_with_rows += words[2]; B.tl_spec_rows += words[3];
}

// ---- the free-running search (search_types.h, Free) -----------------------------------------------------------------------------
// rows of a launch: up to 512 live games the 4-board pair tower's 512 rows (~300 us), beyond one pass of the chip (1024 rows, ~577 us); a
// game needs 0.91 rows per iteration and the virtual descents see about one iteration ahead, so more than ~2 rows per game and launch are not used
// (at most 128 live games: the plain evaluations are of the split-K cluster family, and so are the launches: 32 / 64 / 128 dense rows like the tail's)
// the launches of a search are of the arithmetic family of the plain evaluations of its n live games (what the oracle's evaluator runs):
// the fused 16x16x32 family where tower_table sends so many boards (from 129 by default; every size under DIEE_FLAG_INVARIANT_NN), else the split-K cluster family
bool free_fused(Engine& e, uint32_t n) { return nn_free_available(e, (int)n); }
uint32_t free_rows_for(Engine& e, uint32_t n) {
    if (!free_fused(e, n)) { const uint32_t r = tail_rows_for(e, n); return r > kTailRowsMax ? kTailRowsMax : r; }
    return n >= e.opt.free_rows1024_from ? 1024u : 512u;
}
bool free_possible(Engine& e, uint32_t n, const diee_mcts_cfg& cfg) {
    if (e.opt.free_eval == 0 || cfg.iterations < 1 || cfg.iterations >= (1u << 22)) return false;
    if (n < e.opt.free_min_games || n > std::min<uint32_t>(e.opt.free_max_games, kFreeMaxSlots)) return false;
    const uint32_t rows = free_rows_for(e, n);
    if (rows < n) return false;                                   // every live game's demanded leaf must fit the launch
    if ((uint64_t)(cfg.iterations + 1) * std::max<uint32_t>(e.opt.nodes_per_expansion, 1) + 64 > (1u << 20)) return false;      // (k_free's header word holds 20 bits of node index)
    return free_fused(e, n) || (n <= kTailRowsMax && nn_tail_available(e, (int)rows, (int)n));
}

Free free_view(Engine& e, SearchBufs& B, const diee_mcts_cfg& cfg, uint32_t n) {
    const bool fused = free_fused(e, n);
    // rounds a search may take before the host gives up: about 0.91 * iterations * n / rows are needed; a game that advances one iteration per
    // round needs iterations + 1; a round is lost where an evaluation aged out of the ring before its game's turn came (small rings, an
    // iteration cap of 1) or where a game had to wait for another's flag words -- 4 x (iterations + games) covers what the option mixes of
    // tests/tools/free_fuzz.py ever needed many times over (the words per round are 8 bytes)
    const uint32_t launches = 4 * (cfg.iterations + n) + 66, rows = free_rows_for(e, n);
    // the ring holds the rows of the last `ring` launches: a whole search at iterations = 100, option free_ring launches beyond
    const uint32_t ring = std::min<uint32_t>(cfg.iterations + 2, std::max<uint32_t>(e.opt.free_ring, 4u));
    if (n > B.fr_slot_cap || B.node_cap > B.fr_node_cap) {
        const uint32_t sc = std::max(n, B.fr_slot_cap), nc = std::max(B.node_cap, B.fr_node_cap);
        B.fr_crow.ensure((size_t)sc * nc); B.fr_cval.ensure((size_t)sc * nc);
        B.fr_slots.ensure((size_t)5 * sc); B.fr_wish.ensure((size_t)sc * kFreeWish);
        B.fr_slot_cap = sc; B.fr_node_cap = nc;
    }
    if (ring > B.fr_ring || rows > B.fr_rows) {
        const uint32_t W = std::max(ring, B.fr_ring), R = std::max(rows, B.fr_rows);
        B.fr_rows_idx.ensure((size_t)W * R); B.fr_logits.ensure((size_t)W * R * 1352); B.fr_hv.ensure((size_t)W * R * 72);
        B.fr_rows_state.ensure((size_t)W * std::min<uint32_t>(R, kTailRowsMax));
        B.fr_ring = W; B.fr_rows = R;
    }
    if (launches > B.fr_launches) { B.fr_words.ensure(2 * (size_t)launches + 8); B.fr_launches = launches; }
    if (!B.fr_host) { HIPCHK(hipHostMalloc((void**)&B.fr_host, sizeof(uint32_t) * 2)); memset(B.fr_host, 0, sizeof(uint32_t) * 2); }
    if (!B.cus) { hipDeviceProp_t pr; HIPCHK(hipGetDeviceProperties(&pr, e.device)); B.cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    uint32_t* sl = B.fr_slots.p;
    const uint32_t sc = B.fr_slot_cap;
    return Free{B.fr_crow.p, B.fr_cval.p, B.fr_rows_idx.p, fused ? nullptr : B.fr_rows_state.p, B.fr_logits.p, B.fr_hv.p, B.fr_words.p, B.fr_words.p + B.fr_launches, sl, sl + sc, B.fr_wish.p, sl + 2 * (size_t)sc, sl + 3 * (size_t)sc,
                sl + 4 * (size_t)sc, B.fr_words.p + 2 * (size_t)B.fr_launches, B.fr_host, launches, cfg.iterations, rows, ring,
                std::min<uint32_t>(free_lds_nodes_for(n, (uint32_t)B.cus), std::max<uint32_t>(e.opt.free_lds_nodes, 64u)), e.opt.free_rollout_steps,
                // candidates per game and round: about twice the spare rows a game can hope for (what is not granted is found again next round), at most the option's
                std::min<uint32_t>(e.opt.free_cand_max, 1u + (e.opt.free_cand_x4 * (rows - std::min(rows, n)) + 4u * n - 1u) / (4u * n)),
                e.opt.free_lag_boost, e.opt.free_lag_step,
                // iterations per game and launch: with few games and many spare rows a game often runs 6 ... 8 iterations on one launch's rows (profiles/r06h_*)
                std::max<uint32_t>(n <= kTailRowsMax ? 2u * e.opt.free_iter_cap : e.opt.free_iter_cap, 1u)};
}

// The iterations of one move-step's search for 257 ... 928 live games, behind the root expansion (k_expand has selected every game's leaf
// for iteration 0): round 0 = { k_free, k_free_pack }, then rounds { tower launch q over the rows granted, k_free, k_free_pack }.  The game
// furthest behind completes an iteration in nearly every round (not where its evaluation aged out of the ring or a flag word of another game is
// still missing), so about `iterations` + 1 rounds suffice at worst -- and about 0.91 * n / rows of that do; the bound is F.launches (free_view).
// Rounds sent ahead of a search that is complete return at once, so they go out in chunks, the host looking at the done word in between.
void free_run(Engine& e, uint32_t n, const Tree& T, const Slots& S, const Segs& G, const diee_mcts_cfg& cfg, const SearchParams& P) {
    SearchBufs& B = *e.search;
    const Free F = free_view(e, B, cfg, n);
    hipStream_t st = e.stream;
    // crow of the slots in use (the trees are rebuilt every move-step), the row counts and state words, the per-slot progress (first_sel is
    // written by round 0)
    HIPCHK(hipMemsetAsync(F.crow, 0, sizeof(uint32_t) * (size_t)n * T.node_cap, st));
    HIPCHK(hipMemsetAsync(B.fr_words.p, 0, sizeof(uint32_t) * (2 * (size_t)B.fr_launches + 8), st));
    HIPCHK(hipMemsetAsync(B.fr_slots.p, 0, sizeof(uint32_t) * (size_t)5 * B.fr_slot_cap, st));
    B.fr_host[0] = 0; B.fr_host[1] = 0xFFFFFFFFu;
    launch_free(st, T, S, G, n, P, cfg.c, F, 0);
    uint32_t q = 0, sent = 0;
    uint32_t chunk = B.fr_prev_need ? B.fr_prev_need + 2 : std::max<uint32_t>(8u, (uint32_t)((double)cfg.iterations * n / F.rows) + 4u);
    bool done = false;
    while (!done && q + 1 < F.launches) {
        const uint32_t end = std::min<uint32_t>(F.launches - 1, q + std::max<uint32_t>(chunk, 1u));
        for (; q < end; ++q, ++sent) {
            const size_t rb = (size_t)(q % F.ring) * F.rows;
            if (F.rows_state) {
                if (!nn_forward_tail(e, F.rows_state + rb, (int)F.rows, F.n_rows + q, F.hv + rb * 72, F.logits + rb * 1352, (int)n, F.n_dem + q))
                    throw EngineError(DIEE_ERR_HIP, "free-running search: the cluster tower could not be launched");
            } else
                nn_forward_free(e, T.state, F.rows_idx + rb, F.n_rows + q, (int)F.rows, F.hv + rb * 72, F.logits + rb * 1352, (int)n, F.n_dem + q);
            launch_free(st, T, S, G, n, P, cfg.c, F, q + 1);
        }
        HIPCHK(hipGetLastError());
        e.sync();
        ++B.tl_syncs;
        done = B.fr_host[0] != 0;
        chunk = 4;
    }
    uint32_t words[4] = {0, 0, 0, 0};
    e.d2h(words, F.state, 4);
    e.sync();
    if (!done) throw EngineError(DIEE_ERR_HIP, "free-running search: the iterations did not complete");
    if (e.opt.trace_steps)
        fprintf(stderr, "[diee] free-running: %u games, %u iterations on %u launches with rows of <= %u (%u rounds sent), %u speculative rows, %u nodes in LDS\n",
                n, cfg.iterations, words[1], F.rows, sent, words[2], F.lds_nodes);
    if (e.opt.trace_steps >= 2) {                                      // development: the rows of every round (where the stragglers' rounds begin)
        std::vector<uint32_t> nr((size_t)sent + 1), nd((size_t)sent + 1);
        e.d2h(nr.data(), F.n_rows, nr.size()); e.d2h(nd.data(), F.n_dem, nd.size()); e.sync();
        fprintf(stderr, "[diee] free-running rounds (rows/demanded):");
        for (size_t i = 0; i < nr.size(); ++i) fprintf(stderr, " %u/%u", nr[i], nd[i]);
        fprintf(stderr, "\n");
    }
    B.fr_prev_need = words[1];
    B.tl_iterations += cfg.iterations; B.tl_launched += sent; B.tl_with_rows += words[1]; B.tl_spec_rows += words[2];
}

// alpha_mcts_parallel on the n slots already loaded into B.roots / game_id / round / seg (segment table uploaded, the
// Dirichlet samples of this move-step in pinned buffer `buf`).  Above the tail's reach it enqueues only; a tail search (tail_run) looks at
// its done word between chunks of launches, i.e. synchronises a few times per move-step.
void mcts_run(Engine& e, uint32_t n, uint32_t n_segs, const diee_mcts_cfg& cfg, int buf, uint32_t flags) {
    SearchBufs& B = *e.search;
    const Tree T = tree_view(B);
    Slots S = slots_view(e, B);
    const Segs G = segs_view(B, n_segs);
    hipStream_t st = e.stream;
    const uint32_t quirks = (flags & DIEE_FLAG_REF_QUIRKS) ? 1u : 0u;
    // slots whose selected leaf was terminal need no network row: above 256 live games only the others are evaluated
    const NnRows rows{B.leaf_term.p, B.row_slot.p, B.slot_row.p, B.n_rows.p};
    HIPCHK(hipMemsetAsync(B.iter_flags.p, 0, sizeof(uint32_t) * 2 * (size_t)B.iter_cap * n_segs, st));
    HIPCHK(hipMemcpyAsync(B.noise.p, B.noise_host + (size_t)buf * kMaxSegments * 1352, sizeof(float) * 1352 * n_segs,
                          hipMemcpyHostToDevice, st));
    launch_init_roots(st, T, S, n);
    // option cl_grow (default on): below 129 boards the evaluation is ONE cluster-tower launch, latency-bound, with CUs to spare while
    // fewer than ~28 games live (the tail of a batch: 130 of its 364 move-steps): the launch takes the growth along on extra
    // workgroups (GrowReq) -- no second stream, no event --, and the k_expand behind it only has the priors, the backpropagation and
    // the next descent left (16.2 -> 9.4 us on the chain between two evaluations).
    const bool cl_grow = e.opt.cl_grow != 0;
    const ExpandVariant xv{e.opt.expand2 != 0, e.opt.expand2c != 0};      // handed down to every launch
    GrowReq greq{T, S, G, n, 0u};
    struct GrowScope { NetWeights* w; ~GrowScope() { w->grow_req = nullptr; w->grow_done = false; } } gscope{e.net};
    bool grown = false;                                              // the tower launch of this evaluation took the growth along
    auto forward = [&](uint32_t it, const NnRows* rws) {
        greq.S = S; greq.it = it;
        e.net->grow_req = cl_grow ? &greq : nullptr;
        e.net->grow_done = false;
        const bool compacted = nn_forward(e, B.eval_states.p, (int)n, nullptr, nullptr, rws);
        e.net->grow_req = nullptr;
        grown = e.net->grow_done;
        return compacted;
    };
    forward(kRootIteration, nullptr);                                // forward_policy, alpha_mcts.rs:104 (softmax / tanh in k_expand)
    const SearchParams P{cfg.dir_eps, quirks};
    // one MCTS kernel per network evaluation: expand + backpropagate iteration it, then select for it+1
    launch_expand(st, T, S, G, n, kRootIteration, P, cfg.iterations ? 0u : kNoNextIteration, cfg.c, grown, xv);
    if (free_possible(e, n, cfg)) {                                  // 129 ... 768 live games: every game on its own iteration counter
        S.slot_row = nullptr;
        free_run(e, n, T, S, G, cfg, P);
        launch_reduce_counters(st, S, G, B.step_log.p, std::min(B.cur_step, kStepLog - 1));
        HIPCHK(hipGetLastError());
        return;
    }
    if (tail_possible(e, n, cfg)) {                                  // <= 96 live games: the games in lockstep inside one launch
        S.slot_row = nullptr;
        tail_run(e, n, T, S, G, cfg, P);
        launch_reduce_counters(st, S, G, B.step_log.p, std::min(B.cur_step, kStepLog - 1));
        HIPCHK(hipGetLastError());
        return;
    }
    for (uint32_t it = 0; it < cfg.iterations; ++it) {               // alpha_mcts.rs:149
        const bool compacted = forward(it, &rows);                   // alpha_mcts.rs:186
        S.slot_row = compacted ? B.slot_row.p : nullptr;
        launch_expand(st, T, S, G, n, it, P, it + 1 < cfg.iterations ? it + 1 : kNoNextIteration, cfg.c, grown, xv);
    }
    launch_reduce_counters(st, S, G, B.step_log.p, std::min(B.cur_step, kStepLog - 1));
    HIPCHK(hipGetLastError());
}

// a starved in-launch hand-over of the cluster tower (another process on the GPU took its CUs) raises bit 2 of the
// flag word: the tower then finished "dead" and this search is garbage.  Detected before the move is played: the
// cluster table is dropped for the rest of the process and the caller repeats the search on the per-layer kernels.
bool cluster_starved(Engine& e) {
    if (!e.net || !nn_cluster_used(e)) return false;
    SearchBufs& B = *e.search;
    e.d2h(B.live_host + kLiveWords, e.flags_dev.p, 1);
    e.sync();
    // tests: option test_starve_at = k treats the k-th check of this ctx as a starved hand-over (the fallback path has
    // no other way to be exercised on a box with one process per GPU)
    const bool forced = e.opt.test_starve_at > 0 && ++e.starve_checks == e.opt.test_starve_at;
    if (!(B.live_host[kLiveWords] & 4u) && !forced) return false;
    nn_disable_cluster(e);                                            // clears the flag bit and re-arms the counters
    fprintf(stderr, "[diee] cluster tower: a workgroup hand-over starved (is another process using this GPU?); "
                    "falling back to the per-layer kernels for the rest of this process\n");
    return true;
}

void read_counters(Engine& e, uint32_t seg, diee_stats* stats) {
    if (!stats) return;
    unsigned long long c[CNT_COUNT];
    e.d2h(c, e.search->counters.p + (size_t)seg * CNT_COUNT, (size_t)CNT_COUNT);
    e.sync();
    stats->nn_evals = c[CNT_NN_EVALS]; stats->expansions = c[CNT_EXPANSIONS]; stats->children = c[CNT_CHILDREN];
    stats->terminal_hits = c[CNT_TERMINAL]; stats->depth_sum = c[CNT_DEPTH_SUM]; stats->selections = c[CNT_SELECTIONS];
    stats->illegal_decodes = c[CNT_ILLEGAL]; stats->max_children = c[CNT_MAX_CHILDREN];
    stats->plies = c[CNT_PLIES]; stats->games = c[CNT_GAMES]; stats->nn_rows = c[CNT_NN_ROWS];
}

struct InvariantScope {    // DIEE_FLAG_INVARIANT_NN for the duration of one call
    NetWeights* w; bool saved;
    InvariantScope(Engine& e, uint32_t flags) : w(e.net), saved(e.net->invariant) { if (flags & DIEE_FLAG_INVARIANT_NN) w->invariant = true; }
    ~InvariantScope() { w->invariant = saved; }
};

struct FragBufs {          // host arrays of one diee_fragments until they are handed over: pinned, grown on demand
    int8_t* outcome = nullptr; float* ps = nullptr; float* state = nullptr; uint32_t* game = nullptr;
    size_t cap = 0, n = 0;                                     // rows
    FragBufs() = default;
    FragBufs(const FragBufs&) = delete;
    FragBufs& operator=(const FragBufs&) = delete;
    // room for `rows` rows, the first n kept.  The caller has drained the copy stream if n > 0 (copies into the old blocks).
    void reserve(size_t rows) {
        if (rows <= cap) return;
        PinnedPool& P = pinned_pool();
        const size_t nc = std::max(rows, cap + cap / 2);
        // page-locked where the runtime grants it; a host that refuses to pin more (8 ranks x several GB each) gets pageable memory
        // instead -- the copies into it are staged by the runtime and block the host thread, the records are the same
        auto get = [&](size_t bytes) { void* q = P.acquire(bytes); return q ? q : malloc(std::max<size_t>(bytes, 1)); };
        int8_t* o = (int8_t*)get(nc); float* p = (float*)get(nc * 1352 * sizeof(float));
        float* st = (float*)get(nc * 144 * sizeof(float)); uint32_t* g = (uint32_t*)get(nc * sizeof(uint32_t));
        if (!o || !p || !st || !g) { give(o); give(p); give(st); give(g); throw std::bad_alloc(); }
        if (n) { memcpy(o, outcome, n); memcpy(p, ps, n * 1352 * sizeof(float)); memcpy(st, state, n * 144 * sizeof(float)); memcpy(g, game, n * sizeof(uint32_t)); }
        drop();
        outcome = o; ps = p; state = st; game = g; cap = nc;
    }
    static void give(void* q) { if (q && !pinned_pool().release(q)) free(q); }      // (a block of the pool, or the malloc fallback)
    void drop() { give(outcome); give(ps); give(state); give(game); outcome = nullptr; ps = nullptr; state = nullptr; game = nullptr; }
    void release(diee_fragments* f) { f->n = (uint32_t)n; f->outcome = outcome; f->ps = ps; f->state = state; f->game = game; outcome = nullptr; ps = nullptr; state = nullptr; game = nullptr; cap = n = 0; }
    ~FragBufs() { drop(); }
};

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
    nn_reserve(*this, (int)n);
    SearchBufs& B = *search;
    const InvariantScope inv(*this, flags);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> ids(n), rds(n);
    for (uint32_t i = 0; i < n; ++i) { ids[i] = game_ids ? game_ids[i] : i; rds[i] = rounds ? rounds[i] : 0; }
    // the roots are ONE batch (one alpha_mcts_parallel call): a single segment spanning every slot
    const std::vector<diee_batch> one{{n, 0u, seed}};
    upload_segments(*this, one);
    const uint32_t span[2] = {0u, n};
    h2d(B.seg_slots.p, &span[0], 1); h2d(B.seg_slots.p + kMaxSegments, &span[1], 1);
    HIPCHK(hipMemsetAsync(B.seg.p, 0, sizeof(uint32_t) * n, stream));
    h2d((uint8_t*)B.roots.p, (const uint8_t*)roots, (size_t)n * 32);
    h2d(B.game_id.p, ids.data(), (size_t)n); h2d(B.round.p, rds.data(), (size_t)n);
    sync();
    const int se = net->sample_every; net->sample_every = 0;
    draw_noise(B, 0, one, step, cfg->dir_alpha);
    B.tl_iterations = B.tl_launched = B.tl_with_rows = B.tl_spec_rows = B.tl_syncs = 0; B.tl_prev_need = 0; B.fr_prev_need = 0;
    HIPCHK(hipMemsetAsync(B.step_log.p, 0, sizeof(unsigned long long) * 2 * kStepLog, stream));
    B.cur_step = 0; net->cur_step = 0;
    mcts_run(*this, n, 1, *cfg, 0, flags);
    if (cluster_starved(*this)) {                                   // repeat on the per-layer kernels
        HIPCHK(hipMemsetAsync(B.counters.p, 0, sizeof(unsigned long long) * CNT_COUNT, stream));
        mcts_run(*this, n, 1, *cfg, 0, flags);
    }
    net->sample_every = se;
    tmp_a.ensure((size_t)n * 1352 * 4); tmp_b.ensure((size_t)n * 4); tmp_c.ensure((size_t)n * 4);
    launch_root_probs(stream, tree_view(B), n, (float*)tmp_a.p, (uint32_t*)tmp_b.p, (float*)tmp_c.p);
    HIPCHK(hipGetLastError());
    d2h((uint8_t*)visit_probs, tmp_a.p, (size_t)n * 1352 * 4);
    if (n_children) d2h((uint8_t*)n_children, tmp_b.p, (size_t)n * 4);
    if (root_visits) d2h((uint8_t*)root_visits, tmp_c.p, (size_t)n * 4);
    sync();
    if (stats) {
        read_counters(*this, 0, stats);
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stats->tail_iterations = B.tl_iterations; stats->tail_launches = B.tl_with_rows; stats->tail_spec_rows = B.tl_spec_rows;
    }
    check_overflow();
}

void Engine::self_play(uint32_t n_games, uint32_t first_game_id, const diee_mcts_cfg* cfg, float temperature,
                       uint64_t seed, uint32_t flags, uint32_t max_steps, diee_fragments* out, diee_stats* stats) {
    const diee_batch one{n_games, first_game_id, seed};
    self_play_multi(&one, 1, cfg, temperature, flags, max_steps, out, stats);
}

// K calls of self_play_parallel (alpha_parallel.rs:101-231) played side by side: every batch starts at move-step 0,
// the slot space holds the live games of batch 0, then of batch 1, ..., and every network launch evaluates them all.
// Each batch keeps its own seed, Dirichlet stream, `node_selected` flags and slot-0 bookkeeping, so what a batch
// produces depends on the other batches only through the network kernel its rows are evaluated by.
void Engine::self_play_multi(const diee_batch* batches, uint32_t n_batches, const diee_mcts_cfg* cfg, float temperature,
                             uint32_t flags, uint32_t max_steps, diee_fragments* outs, diee_stats* stats) {
    HIPCHK(hipSetDevice(device));
    if (!net || !net->loaded) throw EngineError(DIEE_ERR_NO_WEIGHTS, "diee_load_weights has not been called");
    if (outs) memset(outs, 0, sizeof(diee_fragments) * n_batches);
    if (stats) memset(stats, 0, sizeof(diee_stats) * n_batches);
    if (cfg->iterations == 0) throw EngineError(DIEE_ERR_ARG, "iterations must be >= 1");
    if (n_batches == 0 || n_batches > kMaxSegments) throw EngineError(DIEE_ERR_ARG, "1 .. 64 batches per call");
    const std::vector<diee_batch> bt(batches, batches + n_batches);
    std::vector<uint32_t> game0(n_batches);
    uint64_t total_games = 0;
    for (uint32_t k = 0; k < n_batches; ++k) {
        if (bt[k].n_games == 0) throw EngineError(DIEE_ERR_ARG, "a batch needs at least one game");
        game0[k] = (uint32_t)total_games; total_games += bt[k].n_games;
    }
    if (total_games > (1u << 22)) throw EngineError(DIEE_ERR_ARG, "too many games in flight");
    const uint32_t n_games = (uint32_t)total_games;
    reserve_search(*this, n_games, cfg->iterations);
    nn_reserve(*this, (int)n_games);
    SearchBufs& B = *search;
    const InvariantScope inv(*this, flags);
    const uint32_t frag_cap = cfg->round_limit + 2;
    if (n_games > B.game_cap || frag_cap > B.frag_cap) {
        const uint32_t gc = std::max(n_games, B.game_cap), fc = std::max(frag_cap, B.frag_cap);
        B.gstate.ensure(gc); B.rounds.ensure(gc); B.nfrags.ensure(gc); B.alive.ensure(gc); B.winner.ensure(gc);
        B.ev_a_count.ensure(gc); B.ev_a_step.ensure(gc); B.ev_b_count.ensure(gc); B.ev_b_step.ensure(gc);
        B.live.ensure(gc); B.gseg.ensure(gc);
        B.frag_ps.ensure((size_t)gc * fc * 1352); B.frag_planes.ensure((size_t)gc * fc * 144);
        B.frag_player.ensure((size_t)gc * fc);
        B.game_cap = gc; B.frag_cap = fc;
    }
    const Games Gm{B.gstate.p, B.rounds.p, B.nfrags.p, B.alive.p, B.winner.p, B.ev_a_count.p, B.ev_a_step.p,
                   B.ev_b_count.p, B.ev_b_step.p, B.live.p, B.gseg.p, B.frag_ps.p, B.frag_planes.p, B.frag_player.p,
                   B.frag_cap, B.counters.p};
    const Tree T = tree_view(B);
    const Slots S = slots_view(*this, B);
    const Segs G = segs_view(B, n_batches);
    const uint32_t quirks = (flags & DIEE_FLAG_REF_QUIRKS) ? 1u : 0u;
    const PlayParams PP{cfg->round_limit, (float)(1.0 / (double)temperature), quirks};

    // ---- output delivery (alpha_parallel.rs:215-230: the call returns all_memories) --------------------------------
    // A game's records are final in the move-step that removes it, and the reference's order is (move-step, game, round-limit
    // flush before win flush): the output of a batch grows by whole steps.  So each step's flushes are listed on the device
    // (k_deliver_scan, 3 words per batch read back with the live counts), gathered + relabelled into a staging buffer and
    // copied into pinned host arrays on a second stream while the next move-steps search; the long tail of a batch (147 of
    // 364 move-steps with <= 32 live games) leaves nothing to wait for at the end.  outs == NULL: listed and counted only.
    const bool deliver = outs != nullptr;
    std::vector<FragBufs> fb(deliver ? n_batches : 0);
    std::vector<uint64_t> frag_total(n_batches, 0);
    for (auto& ev : B.dl_events) ev.ensure((size_t)2 * n_games);
    B.ev_copy_armed[0] = B.ev_copy_armed[1] = false;
    const uint32_t stage_rows = std::max<uint32_t>(opt.deliver_stage_rows, 1);
    double deliver_secs = 0.0;
    uint64_t deliver_bytes = 0;
    if (deliver) {
        if (!B.copy) {
            HIPCHK(hipStreamCreateWithFlags(&B.copy, hipStreamNonBlocking));
            for (auto& e : B.ev_copy) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if (stage_rows > B.dl_rows) {
            B.dl_ps.ensure((size_t)stage_rows * 1352); B.dl_planes.ensure((size_t)stage_rows * 144);
            B.dl_outcome.ensure(stage_rows); B.dl_game.ensure(stage_rows); B.dl_rows = stage_rows;
        }
        // first guess: 128 records per game (a random-init game runs ~107 plies); grown when a batch outruns it
        for (uint32_t k = 0; k < n_batches; ++k)
            fb[k].reserve(std::min<size_t>((size_t)bt[k].n_games * frag_cap, (size_t)bt[k].n_games * opt.deliver_rows_per_game + 1024));
    }
    struct CopyDrain { hipStream_t st; ~CopyDrain() { if (st) (void)hipStreamSynchronize(st); } } drain{deliver ? B.copy : nullptr};   // (no copy may outlive its pinned target)
    const DeliverOut stage{B.dl_ps.p, B.dl_planes.p, B.dl_outcome.p, B.dl_game.p};
    std::vector<uint32_t> pending;                                      // DeliverSummary of the step whose records still have to go out
    int pending_buf = 0;
    auto deliver_pending = [&] {
        if (pending.empty()) return;
        const auto td = std::chrono::steady_clock::now();
        bool sent = false;
        for (uint32_t k = 0; k < n_batches; ++k) {
            const uint32_t rows = pending[DeliverSummary::rows(n_batches, k)], e0 = pending[DeliverSummary::ev0(n_batches, k)],
                           ne = pending[DeliverSummary::nev(n_batches, k)];
            frag_total[k] += rows;
            if (!deliver || !rows) continue;
            FragBufs& f = fb[k];
            if (f.n + rows > f.cap) { HIPCHK(hipStreamSynchronize(B.copy)); f.reserve(f.n + rows); }
            for (uint32_t o = 0; o < rows; o += stage_rows) {
                const uint32_t m = std::min(stage_rows, rows - o);
                launch_deliver_copy(B.copy, Gm, G, B.dl_events[pending_buf].p + e0, ne, o, o + m, stage);
                HIPCHK(hipGetLastError());
                const size_t at = f.n + o;
                HIPCHK(hipMemcpyAsync(f.ps + at * 1352, stage.ps, (size_t)m * 1352 * sizeof(float), hipMemcpyDeviceToHost, B.copy));
                HIPCHK(hipMemcpyAsync(f.state + at * 144, stage.planes, (size_t)m * 144 * sizeof(float), hipMemcpyDeviceToHost, B.copy));
                HIPCHK(hipMemcpyAsync(f.outcome + at, stage.outcome, (size_t)m, hipMemcpyDeviceToHost, B.copy));
                HIPCHK(hipMemcpyAsync(f.game + at, stage.game, (size_t)m * sizeof(uint32_t), hipMemcpyDeviceToHost, B.copy));
                deliver_bytes += (size_t)m * (1352 * 4 + 144 * 4 + 1 + 4);
            }
            f.n += rows;
            sent = true;
        }
        if (sent) { HIPCHK(hipEventRecord(B.ev_copy[pending_buf], B.copy)); B.ev_copy_armed[pending_buf] = true; }
        pending.clear();
        deliver_secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - td).count();
    };

    upload_segments(*this, bt);
    B.tl_iterations = B.tl_launched = B.tl_with_rows = B.tl_spec_rows = B.tl_syncs = 0; B.tl_prev_need = 0; B.fr_prev_need = 0;
    HIPCHK(hipMemsetAsync(B.step_log.p, 0, sizeof(unsigned long long) * 2 * kStepLog, stream));
    nn_reset_timing(*this);
    launch_init_games(stream, Gm, G, n_games);
    draw_noise(B, 0, bt, 0, cfg->dir_alpha);
    sync();
    // ---- timed region: every input is resident in HBM ----
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t n_live = n_games, step = 0;
    std::vector<uint32_t> steps_of(n_batches, 0);                       // move-steps each batch took part in
    std::vector<uint32_t> live_of(n_batches);
    for (uint32_t k = 0; k < n_batches; ++k) live_of[k] = bt[k].n_games;
    const bool trace_steps = opt.trace_steps != 0;                        // development: batch size of every move-step
    while (n_live > 0 && (max_steps == 0 || step < max_steps)) {       // alpha_parallel.rs:129
        if (trace_steps) fprintf(stderr, "[diee] move-step %u: %u games alive\n", step, n_live);
        for (uint32_t k = 0; k < n_batches; ++k) if (live_of[k]) steps_of[k] = step + 1;
        const int buf = (int)(step & 1u);
        HIPCHK(hipMemsetAsync(B.seg_slots.p, 0, sizeof(uint32_t) * 2 * kMaxSegments, stream));
        launch_gather_roots(stream, Gm, S, G, n_live);
        HIPCHK(hipMemcpyAsync(B.counters_bak.p, B.counters.p, sizeof(unsigned long long) * CNT_COUNT * n_batches, hipMemcpyDeviceToDevice, stream));
        B.cur_step = step; net->cur_step = (int)std::min(step, kStepLog - 1);
        mcts_run(*this, n_live, n_batches, *cfg, buf, flags);           // :146
        // while the GPU searches: the next move-step's Dirichlet samples (the only host arithmetic of a move-step) and the
        // previous move-step's records (enqueued behind this step's search so that the search never waits for the host)
        draw_noise(B, buf ^ 1, bt, step + 1, cfg->dir_alpha);
        deliver_pending();
        if (cluster_starved(*this)) {                                   // rare: repeat this move-step's search, per-layer kernels
            HIPCHK(hipMemcpyAsync(B.counters.p, B.counters_bak.p, sizeof(unsigned long long) * CNT_COUNT * n_batches, hipMemcpyDeviceToDevice, stream));
            mcts_run(*this, n_live, n_batches, *cfg, buf, flags);
        }
        launch_play_move(stream, T, Gm, G, n_live, step, PP);           // :164-224
        if (B.ev_copy_armed[buf]) { HIPCHK(hipStreamWaitEvent(stream, B.ev_copy[buf], 0)); B.ev_copy_armed[buf] = false; }   // (step - 2's list has been read)
        launch_deliver_scan(stream, Gm, G, n_live, step, B.dl_events[buf].p, B.n_live_dev.p);
        launch_compact_live(stream, Gm, n_live, n_batches, B.n_live_dev.p);   // :226-228
        HIPCHK(hipGetLastError());
        d2h(B.live_host, B.n_live_dev.p, (size_t)1 + 4 * n_batches);
        sync();
        n_live = B.live_host[0];
        for (uint32_t k = 0; k < n_batches; ++k) live_of[k] = B.live_host[1 + k];
        pending.assign(B.live_host, B.live_host + 1 + 4 * n_batches);
        pending_buf = buf;
        ++step;
    }
    deliver_pending();
    if (deliver) {                                                      // the last copies (one game's records, typically)
        const auto td = std::chrono::steady_clock::now();
        HIPCHK(hipStreamSynchronize(B.copy));
        deliver_secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - td).count();
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    check_overflow();
    std::vector<unsigned long long> step_log(2 * (size_t)kStepLog);
    d2h(step_log.data(), B.step_log.p, step_log.size());
    sync();
    net->cur_step = -1;
    if (stats) {
        nn_harvest(*this, &stats[0], &step_log);                        // sampled tower timings: whole call, reported with batch 0
        for (uint32_t k = 0; k < n_batches; ++k) {
            read_counters(*this, k, &stats[k]);
            stats[k].move_steps = steps_of[k]; stats[k].seconds = secs; stats[k].fragments = frag_total[k];
        }
        stats[0].deliver_seconds = deliver_secs; stats[0].deliver_bytes = deliver_bytes;
        stats[0].tail_iterations = B.tl_iterations; stats[0].tail_launches = B.tl_with_rows; stats[0].tail_spec_rows = B.tl_spec_rows;
    } else {
        nn_harvest(*this, nullptr, &step_log);
    }
    if (deliver)
        for (uint32_t k = 0; k < n_batches; ++k) fb[k].release(&outs[k]);
}

}  // namespace diee
