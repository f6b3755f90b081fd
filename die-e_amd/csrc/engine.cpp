// engine.cpp -- Engine construction and the pure game functions.
#include "engine.h"

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "launch.h"

namespace diee {

void free_net(NetWeights*);
void cluster_baton_register(int device, int delta);
void nn_reset_cluster(Engine& e);
void nn_disable_cluster(Engine& e);
void free_search(SearchBufs*);
void pinned_pool_set_cap_mb(size_t mb);
size_t pinned_pool_cap_mb();

// ---- options ---------------------------------------------------------------------------------------------------------
namespace {
struct OptEntry { const char* key; int Options::*i; uint32_t Options::*u; std::string Options::*s; };
const OptEntry kOptions[] = {
    {"tower_table", nullptr, nullptr, &Options::tower_table}, {"tower_cl", nullptr, nullptr, &Options::tower_cl},
    {"tower_pair", &Options::tower_pair, nullptr, nullptr}, {"fused_heads", &Options::fused_heads, nullptr, nullptr},
    {"cluster_heads", &Options::cluster_heads, nullptr, nullptr}, {"cluster_init", &Options::cluster_init, nullptr, nullptr},
    {"compact", &Options::compact, nullptr, nullptr}, {"cl_pack", &Options::cl_pack, nullptr, nullptr},
    {"shared_gpu", &Options::shared_gpu, nullptr, nullptr},
    {"cl_grow", &Options::cl_grow, nullptr, nullptr}, {"expand2", &Options::expand2, nullptr, nullptr},
    {"expand2c", &Options::expand2c, nullptr, nullptr}, {"spec_eval", &Options::spec_eval, nullptr, nullptr},
    {"spec_rollout_steps", nullptr, &Options::spec_rollout_steps, nullptr}, {"spec_max_games", nullptr, &Options::spec_max_games, nullptr},
    {"spec_extra_rows", nullptr, &Options::spec_extra_rows, nullptr}, {"spec_child_rows", nullptr, &Options::spec_child_rows, nullptr},
    {"spec_fused_games", nullptr, &Options::spec_fused_games, nullptr}, {"spec_fused_from", nullptr, &Options::spec_fused_from, nullptr},
    {"spec_rows64_from", nullptr, &Options::spec_rows64_from, nullptr}, {"spec_rows128_from", nullptr, &Options::spec_rows128_from, nullptr},
    {"spec_ring_mb", nullptr, &Options::spec_ring_mb, nullptr},
    {"free_eval", &Options::free_eval, nullptr, nullptr}, {"free_min_games", nullptr, &Options::free_min_games, nullptr},
    {"free_max_games", nullptr, &Options::free_max_games, nullptr}, {"free_rows1024_from", nullptr, &Options::free_rows1024_from, nullptr},
    {"free_rollout_steps", nullptr, &Options::free_rollout_steps, nullptr}, {"free_cand_max", nullptr, &Options::free_cand_max, nullptr},
    {"free_ring", nullptr, &Options::free_ring, nullptr}, {"free_iter_cap", nullptr, &Options::free_iter_cap, nullptr}, {"free_cand_x4", nullptr, &Options::free_cand_x4, nullptr},
    {"free_lag_boost", nullptr, &Options::free_lag_boost, nullptr}, {"free_lag_step", nullptr, &Options::free_lag_step, nullptr}, {"free_lds_nodes", nullptr, &Options::free_lds_nodes, nullptr},
    {"path_cap", nullptr, &Options::path_cap, nullptr}, {"nodes_per_expansion", nullptr, &Options::nodes_per_expansion, nullptr},
    {"deliver_stage_rows", nullptr, &Options::deliver_stage_rows, nullptr}, {"deliver_rows_per_game", nullptr, &Options::deliver_rows_per_game, nullptr},
    {"trace_steps", &Options::trace_steps, nullptr, nullptr}, {"trace_dispatch", &Options::trace_dispatch, nullptr, nullptr},
    {"test_starve_at", &Options::test_starve_at, nullptr, nullptr}, {"test_tail_skip", &Options::test_tail_skip, nullptr, nullptr},
    {"pinned_pool_mb", nullptr, nullptr, nullptr},          // process-wide: the cap of the pool of page-locked output blocks (search_host.cpp)
};
const OptEntry* find_option(const std::string& key) {
    for (const auto& e : kOptions) if (key == e.key) return &e;
    return nullptr;
}
bool parse_num(const std::string& v, long long& out) {
    if (v.empty()) return false;
    char* end = nullptr;
    out = strtoll(v.c_str(), &end, 10);
    return end && *end == 0;
}
// "a:b,c:d" | "none" | "default"
bool table_ok(const std::string& v) {
    if (v == "none" || v == "default") return true;
    size_t pos = 0;
    while (pos < v.size()) {
        const size_t c = v.find(':', pos), e = v.find(',', pos);
        long long a, b;
        if (c == std::string::npos || (e != std::string::npos && e < c)) return false;
        if (!parse_num(v.substr(pos, c - pos), a) || !parse_num(v.substr(c + 1, (e == std::string::npos ? v.size() : e) - c - 1), b)) return false;
        if (e == std::string::npos) return true;
        pos = e + 1;
    }
    return false;
}
}  // namespace

void Engine::set_option(const std::string& key, const std::string& value) {
    const OptEntry* e = find_option(key);
    if (!e) throw EngineError(DIEE_ERR_ARG, "unknown option `" + key + "`");
    if (!e->s && !e->i && !e->u) {                          // pinned_pool_mb
        long long v;
        if (!parse_num(value, v) || v < 0) throw EngineError(DIEE_ERR_ARG, "option " + key + ": `" + value + "` is not a non-negative integer");
        pinned_pool_set_cap_mb((size_t)v);
        return;
    }
    if (e->s) {
        if (!table_ok(value)) throw EngineError(DIEE_ERR_ARG, "option " + key + ": `" + value + "` is not \"a:b,c:d\", \"none\" or \"default\"");
        opt.*(e->s) = value;
    } else {
        long long v;
        if (!parse_num(value, v) || v < 0 || v > 0x7fffffffLL) throw EngineError(DIEE_ERR_ARG, "option " + key + ": `" + value + "` is not a non-negative integer");
        if (e->i) opt.*(e->i) = (int)v; else opt.*(e->u) = (uint32_t)v;
    }
    apply_options();
}

std::string Engine::get_option(const std::string& key) const {
    const OptEntry* e = find_option(key);
    if (!e) throw EngineError(DIEE_ERR_ARG, "unknown option `" + key + "`");
    if (!e->s && !e->i && !e->u) return std::to_string(pinned_pool_cap_mb());
    if (e->s) return opt.*(e->s);
    return std::to_string(e->i ? (long long)(opt.*(e->i)) : (long long)(opt.*(e->u)));
}

Engine::Engine(int dev) : device(dev) {
    err[0] = 0;
    // the environment as a development override of the option defaults: read here, once per ctx, and nowhere else
    for (const auto& e : kOptions) {
        std::string name = "DIEE_";
        for (const char* c = e.key; *c; ++c) name += (char)toupper((unsigned char)*c);
        if (const char* v = getenv(name.c_str())) {
            try { set_option(e.key, v); }
            catch (const EngineError& x) { fprintf(stderr, "[diee] %s=%s ignored: %s\n", name.c_str(), v, x.what()); }
        }
    }
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    flags_dev.ensure(16);
    HIPCHK(hipMemsetAsync(flags_dev.p, 0, 16 * sizeof(uint32_t), stream));
    sync();
    cluster_baton_register(device, +1);
}

Engine::~Engine() {
    (void)hipSetDevice(device);
    if (stream) cluster_baton_register(device, -1);
    free_net(net);
    free_search(search);
    if (stream) (void)hipStreamDestroy(stream);
}

void Engine::check_overflow() {
    uint32_t f = 0;
    d2h(&f, flags_dev.p, 1);
    sync();
    if (f) {
        HIPCHK(hipMemsetAsync(flags_dev.p, 0, sizeof(uint32_t), stream));
        sync();
        if (f & 4u) {
            // a hand-over inside the cluster tower OR the pair tower (same flag bit) timed out: both are dropped for the rest of the
            // process (as the search paths do, cluster_starved), so the next call does not spin to the same timeout again
            nn_disable_cluster(*this);                      // also re-arms the hand-over counters
            throw EngineError(DIEE_ERR_HIP, "cluster / pair tower: a workgroup hand-over timed out (grid not co-resident: another process on this GPU?); "
                                            "both are now off for this ctx -- diee_set_option(ctx, \"shared_gpu\", \"1\") starts without them");
        }
        throw EngineError(DIEE_ERR_CAPACITY, "device capacity overflow (sequence table / tree arena), flag=" + std::to_string(f));
    }
}

void Engine::legal_moves(const diee_bg_state* s, uint32_t n, int8_t* plays, uint32_t cap, uint32_t* counts) {
    HIPCHK(hipSetDevice(device));
    if (!n) return;
    tmp_a.ensure((size_t)n * 32);
    tmp_b.ensure((size_t)n * cap * 4 + 4);
    tmp_c.ensure((size_t)n * 4);
    h2d(tmp_a.p, (const uint8_t*)s, (size_t)n * 32);
    launch_legal_moves(stream, tmp_a.p, n, (uint32_t*)tmp_b.p, cap, (uint32_t*)tmp_c.p, flags_dev.p);
    HIPCHK(hipGetLastError());
    d2h((uint8_t*)plays, tmp_b.p, (size_t)n * cap * 4);
    d2h((uint8_t*)counts, tmp_c.p, (size_t)n * 4);
    sync();
    check_overflow();
}

// stand-alone timing of the get_valid_moves kernel, states resident (development probe)
void Engine::rules_bench(const diee_bg_state* s, uint32_t n, int reps, float* us_legal_moves, float* mean_plays) {
    HIPCHK(hipSetDevice(device));
    const uint32_t cap = 160;
    tmp_a.ensure((size_t)n * 32);
    tmp_b.ensure((size_t)n * cap * 4 + 4);
    tmp_c.ensure((size_t)n * 4);
    h2d(tmp_a.p, (const uint8_t*)s, (size_t)n * 32);
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    for (int r = -2; r < reps; ++r) {
        if (r == 0) HIPCHK(hipEventRecord(a, stream));
        launch_legal_moves(stream, tmp_a.p, n, (uint32_t*)tmp_b.p, cap, (uint32_t*)tmp_c.p, flags_dev.p);
    }
    HIPCHK(hipEventRecord(b, stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    std::vector<uint32_t> counts(n);
    d2h((uint8_t*)counts.data(), tmp_c.p, (size_t)n * 4);
    sync();
    check_overflow();
    double tot = 0;
    for (uint32_t c : counts) tot += c;
    *us_legal_moves = ms * 1e3f / (float)reps;
    *mean_plays = (float)(tot / n);
}

void Engine::encode(const diee_bg_state* s, const int8_t* plays, uint32_t n, uint32_t* codes) {
    HIPCHK(hipSetDevice(device));
    if (!n) return;
    tmp_a.ensure((size_t)n * 32); tmp_b.ensure((size_t)n * 4); tmp_c.ensure((size_t)n * 4);
    h2d(tmp_a.p, (const uint8_t*)s, (size_t)n * 32);
    h2d(tmp_b.p, (const uint8_t*)plays, (size_t)n * 4);
    launch_encode(stream, tmp_a.p, (const uint32_t*)tmp_b.p, n, (uint32_t*)tmp_c.p);
    HIPCHK(hipGetLastError());
    d2h((uint8_t*)codes, tmp_c.p, (size_t)n * 4);
    sync();
}

void Engine::decode(const diee_bg_state* s, const uint32_t* codes, uint32_t n, int8_t* plays) {
    HIPCHK(hipSetDevice(device));
    if (!n) return;
    tmp_a.ensure((size_t)n * 32); tmp_b.ensure((size_t)n * 4); tmp_c.ensure((size_t)n * 4);
    h2d(tmp_a.p, (const uint8_t*)s, (size_t)n * 32);
    h2d(tmp_b.p, (const uint8_t*)codes, (size_t)n * 4);
    launch_decode(stream, tmp_a.p, (const uint32_t*)tmp_b.p, n, (uint32_t*)tmp_c.p);
    HIPCHK(hipGetLastError());
    d2h((uint8_t*)plays, tmp_c.p, (size_t)n * 4);
    sync();
}

void Engine::apply(diee_bg_state* s, const int8_t* plays, const uint8_t* dice, uint32_t n) {
    HIPCHK(hipSetDevice(device));
    if (!n) return;
    tmp_a.ensure((size_t)n * 32); tmp_b.ensure((size_t)n * 4); tmp_c.ensure((size_t)n * 2);
    h2d(tmp_a.p, (const uint8_t*)s, (size_t)n * 32);
    h2d(tmp_b.p, (const uint8_t*)plays, (size_t)n * 4);
    h2d(tmp_c.p, dice, (size_t)n * 2);
    launch_apply(stream, tmp_a.p, (const uint32_t*)tmp_b.p, tmp_c.p, n);
    HIPCHK(hipGetLastError());
    d2h((uint8_t*)s, tmp_a.p, (size_t)n * 32);
    sync();
}

void Engine::planes(const diee_bg_state* s, uint32_t n, float* out) {
    HIPCHK(hipSetDevice(device));
    if (!n) return;
    tmp_a.ensure((size_t)n * 32); tmp_b.ensure((size_t)n * 144 * 4);
    h2d(tmp_a.p, (const uint8_t*)s, (size_t)n * 32);
    launch_planes(stream, tmp_a.p, n, (float*)tmp_b.p);
    HIPCHK(hipGetLastError());
    d2h((uint8_t*)out, tmp_b.p, (size_t)n * 144 * 4);
    sync();
}

void Engine::probe_f32(const float* a, const float* b, uint32_t n, float* sq, float* dv, float* pw) {
    HIPCHK(hipSetDevice(device));
    if (!n) return;
    const size_t B = (size_t)n * 4;
    tmp_a.ensure(B); tmp_b.ensure(B); tmp_c.ensure(B); tmp_d.ensure(B); tmp_e.ensure(B);
    h2d(tmp_a.p, (const uint8_t*)a, B); h2d(tmp_b.p, (const uint8_t*)b, B);
    launch_probe_f32(stream, (float*)tmp_a.p, (float*)tmp_b.p, n, (float*)tmp_c.p, (float*)tmp_d.p, (float*)tmp_e.p);
    HIPCHK(hipGetLastError());
    if (sq) d2h((uint8_t*)sq, tmp_c.p, B);
    if (dv) d2h((uint8_t*)dv, tmp_d.p, B);
    if (pw) d2h((uint8_t*)pw, tmp_e.p, B);
    sync();
}

void Engine::probe_dice(uint64_t seed, const uint32_t* ctr, uint32_t n, uint8_t* dice, double* uni) {
    HIPCHK(hipSetDevice(device));
    if (!n) return;
    tmp_a.ensure((size_t)n * 16); tmp_b.ensure((size_t)n * 2); tmp_c.ensure((size_t)n * 8);
    h2d(tmp_a.p, (const uint8_t*)ctr, (size_t)n * 16);
    launch_probe_dice(stream, seed, (const uint32_t*)tmp_a.p, n, tmp_b.p, (double*)tmp_c.p);
    HIPCHK(hipGetLastError());
    d2h(dice, tmp_b.p, (size_t)n * 2); d2h((uint8_t*)uni, tmp_c.p, (size_t)n * 8);
    sync();
}

}  // namespace diee
