// diee_api.cpp -- C ABI (include/diee.h) and host-side engine of libdiee.so.
#include "../../include/diee.h"
#include "../../include/diee_dev.h"
#include "engine.h"
#include "nn_host.h"
#include "launch.h"
#include "ttt_host.h"

#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>

using namespace diee;

#define API_BEGIN(ctx)            \
    if (!(ctx)) return DIEE_ERR_ARG; \
    (ctx)->err[0] = 0;            \
    try {
#define API_END(ctx)                                                        \
    }                                                                       \
    catch (const EngineError& e) {                                          \
        snprintf((ctx)->err, sizeof((ctx)->err), "%s", e.what());           \
        return (diee_status)e.code;                                         \
    }                                                                       \
    catch (const std::bad_alloc&) {                                         \
        snprintf((ctx)->err, sizeof((ctx)->err), "host allocation failed"); \
        return DIEE_ERR_HIP;                                                \
    }                                                                       \
    catch (const std::exception& e) { /* nothing may cross the C boundary */ \
        snprintf((ctx)->err, sizeof((ctx)->err), "%s", e.what());           \
        return DIEE_ERR_HIP;                                                \
    }                                                                       \
    catch (...) {                                                           \
        snprintf((ctx)->err, sizeof((ctx)->err), "unknown exception");      \
        return DIEE_ERR_HIP;                                                \
    }                                                                       \
    return DIEE_OK;

// a backgammon ctx is the HIP engine; a tic-tac-toe ctx (BASELINE configs[0], host path) carries its host engine instead
// and answers DIEE_ERR_UNSUPPORTED wherever a HIP kernel would be needed
struct diee_ctx {
    char err[512];
    int game;
    Engine* hip = nullptr;
    ttt::Engine* ttt = nullptr;
    ~diee_ctx() { delete hip; delete ttt; }
};
static Engine* bg(diee_ctx* c) {
    if (!c->hip) throw EngineError(DIEE_ERR_UNSUPPORTED, "this entry point needs a backgammon ctx (the tic-tac-toe ctx runs on the host)");
    return c->hip;
}

extern "C" {

const char* diee_version(void) { return "die-e_amd 0.1 (gfx950)"; }

diee_status diee_create(int device, int game_id, diee_ctx** out) {
    if (!out) return DIEE_ERR_ARG;
    *out = nullptr;
    if (game_id != DIEE_GAME_BACKGAMMON && game_id != DIEE_GAME_TTT) return DIEE_ERR_UNSUPPORTED;
    diee_ctx* c = nullptr;
    try {
        c = new diee_ctx();
        c->err[0] = 0; c->game = game_id;
        if (game_id == DIEE_GAME_TTT) {
            c->ttt = new ttt::Engine();                 // BASELINE configs[0]: host path, no GPU involved (`device` is not looked at)
        } else {
            int ndev = 0;
            if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) { delete c; return DIEE_ERR_HIP; }
            c->hip = new Engine(device);
        }
    } catch (...) {
        delete c;
        return DIEE_ERR_HIP;
    }
    *out = c;
    return DIEE_OK;
}

void diee_destroy(diee_ctx* c) { delete c; }

const char* diee_last_error(const diee_ctx* c) { return c ? c->err : "null ctx"; }

diee_status diee_device_pci_bus_id(int device, char* out, size_t cap) {
    if (!out || cap < 16) return DIEE_ERR_ARG;
    out[0] = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return DIEE_ERR_HIP;
    if (hipDeviceGetPCIBusId(out, (int)cap, device) != hipSuccess) { out[0] = 0; return DIEE_ERR_HIP; }
    return DIEE_OK;
}

diee_status diee_set_option(diee_ctx* c, const char* key, const char* value) {
    API_BEGIN(c)
    if (!key || !value) throw EngineError(DIEE_ERR_ARG, "null pointer");
    bg(c)->set_option(key, value);
    API_END(c)
}

diee_status diee_get_option(diee_ctx* c, const char* key, char* value, size_t cap) {
    API_BEGIN(c)
    if (!key || !value || !cap) throw EngineError(DIEE_ERR_ARG, "null pointer");
    const std::string v = bg(c)->get_option(key);
    if (v.size() + 1 > cap) throw EngineError(DIEE_ERR_ARG, "value buffer too small");
    memcpy(value, v.c_str(), v.size() + 1);
    API_END(c)
}

diee_status diee_bg_legal_moves(diee_ctx* c, const diee_bg_state* s, uint32_t n, int8_t* plays, uint32_t cap,
                                uint32_t* counts) {
    API_BEGIN(c)
    if (!s || !plays || !counts) throw EngineError(DIEE_ERR_ARG, "null pointer");
    bg(c)->legal_moves(s, n, plays, cap, counts);
    API_END(c)
}

diee_status diee_bg_encode(diee_ctx* c, const diee_bg_state* s, const int8_t* plays, uint32_t n, uint32_t* codes) {
    API_BEGIN(c)
    if (!s || !plays || !codes) throw EngineError(DIEE_ERR_ARG, "null pointer");
    bg(c)->encode(s, plays, n, codes);
    API_END(c)
}

diee_status diee_bg_decode(diee_ctx* c, const diee_bg_state* s, const uint32_t* codes, uint32_t n, int8_t* plays) {
    API_BEGIN(c)
    if (!s || !plays || !codes) throw EngineError(DIEE_ERR_ARG, "null pointer");
    bg(c)->decode(s, codes, n, plays);
    API_END(c)
}

diee_status diee_bg_apply(diee_ctx* c, diee_bg_state* s, const int8_t* plays, const uint8_t* dice, uint32_t n) {
    API_BEGIN(c)
    if (!s || !plays || !dice) throw EngineError(DIEE_ERR_ARG, "null pointer");
    bg(c)->apply(s, plays, dice, n);
    API_END(c)
}

diee_status diee_bg_planes(diee_ctx* c, const diee_bg_state* s, uint32_t n, float* out) {
    API_BEGIN(c)
    if (!s || !out) throw EngineError(DIEE_ERR_ARG, "null pointer");
    bg(c)->planes(s, n, out);
    API_END(c)
}

diee_status diee_det_pow(diee_ctx* c, const float* x, const float* y, uint32_t n, float* out) {
    API_BEGIN(c)
    if (!x || !y || !out) throw EngineError(DIEE_ERR_ARG, "null pointer");
    bg(c)->probe_f32(x, y, n, nullptr, nullptr, out);
    API_END(c)
}

diee_status diee_probe_f32(diee_ctx* c, const float* a, const float* b, uint32_t n, float* sq, float* dv, float* pw) {
    API_BEGIN(c)
    bg(c)->probe_f32(a, b, n, sq, dv, pw);
    API_END(c)
}

diee_status diee_probe_dice(diee_ctx* c, uint64_t seed, const uint32_t* ctr, uint32_t n, uint8_t* dice, double* uni) {
    API_BEGIN(c)
    bg(c)->probe_dice(seed, ctr, n, dice, uni);
    API_END(c)
}

}  // extern "C"

// ---- network / search entry points ---------------------------------------------------------------
namespace diee {
size_t weights_count_bg();
void random_weights_bg(uint64_t seed, float* blob);
bool pinned_release(void* p);   // search_host.cpp: back to the pool of page-locked blocks; false = not one of its blocks
}

extern "C" {

size_t diee_weights_count(int game_id) {
    return game_id == DIEE_GAME_BACKGAMMON ? diee::weights_count_bg() : game_id == DIEE_GAME_TTT ? diee::ttt::weights_count() : 0;
}

uint32_t diee_ttt_valid_moves(const diee_ttt_state* s, uint8_t* moves) { return (s && moves) ? (uint32_t)diee::ttt::valid_moves(*s, moves) : 0u; }
void diee_ttt_apply_move(diee_ttt_state* s, uint8_t move) { if (s && move < 9) diee::ttt::apply_move(*s, move); }
int diee_ttt_check_winner(const diee_ttt_state* s, int* winner) {
    int w = 0;
    const bool over = s && diee::ttt::check_winner(*s, w);
    if (winner) *winner = w;
    return over ? 1 : 0;
}
void diee_ttt_planes(const diee_ttt_state* s, float* out) { if (s && out) diee::ttt::planes(*s, out); }

diee_status diee_random_weights(int game_id, uint64_t seed, float* blob, size_t n) {
    if (game_id == DIEE_GAME_TTT) {
        if (!blob || n != diee::ttt::weights_count()) return DIEE_ERR_ARG;
        diee::ttt::random_weights(seed, blob);
        return DIEE_OK;
    }
    if (game_id != DIEE_GAME_BACKGAMMON) return DIEE_ERR_UNSUPPORTED;
    if (!blob || n != diee::weights_count_bg()) return DIEE_ERR_ARG;
    diee::random_weights_bg(seed, blob);
    return DIEE_OK;
}

diee_status diee_load_weights(diee_ctx* c, const float* blob, size_t n) {
    API_BEGIN(c)
    if (!blob) throw EngineError(DIEE_ERR_ARG, "null blob");
    if (c->ttt) c->ttt->load_weights(blob, n);
    else bg(c)->load_weights(blob, n);
    API_END(c)
}

diee_status diee_set_invariant_nn(diee_ctx* c, int on) {
    API_BEGIN(c)
    if (!bg(c)->net) throw EngineError(DIEE_ERR_NO_WEIGHTS, "diee_load_weights has not been called");
    bg(c)->net->invariant = on != 0;
    API_END(c)
}

diee_status diee_nn_forward(diee_ctx* c, const diee_bg_state* states, uint32_t n, float* policy, float* value) {
    API_BEGIN(c)
    if (!states || !policy || !value) throw EngineError(DIEE_ERR_ARG, "null pointer");
    if (c->ttt) c->ttt->forward((const diee_ttt_state*)states, n, policy, value);
    else bg(c)->nn_forward_host(states, n, policy, value);
    API_END(c)
}

diee_status diee_mcts_batch(diee_ctx* c, const diee_bg_state* roots, uint32_t n, const diee_mcts_cfg* cfg,
                            uint64_t seed, uint32_t step, const uint32_t* game_ids, const uint32_t* rounds,
                            uint32_t flags, float* visit_probs, uint32_t* n_children, float* root_visits,
                            diee_stats* stats) {
    API_BEGIN(c)
    if (!roots || !cfg || !visit_probs) throw EngineError(DIEE_ERR_ARG, "null pointer");
    if (c->ttt) c->ttt->mcts_batch((const diee_ttt_state*)roots, n, *cfg, seed, step, game_ids, rounds, flags, visit_probs, n_children, root_visits, stats);
    else bg(c)->mcts_batch(roots, n, cfg, seed, step, game_ids, rounds, flags, visit_probs, n_children, root_visits, stats);
    API_END(c)
}

diee_status diee_self_play(diee_ctx* c, uint32_t n_games, uint32_t first_game_id, const diee_mcts_cfg* cfg,
                           float temperature, uint64_t seed, uint32_t flags, uint32_t max_steps,
                           diee_fragments* out, diee_stats* stats) {
    API_BEGIN(c)
    if (!cfg || n_games == 0) throw EngineError(DIEE_ERR_ARG, "bad arguments");
    if (c->ttt) c->ttt->self_play(n_games, first_game_id, *cfg, temperature, seed, flags, max_steps, out, stats);
    else bg(c)->self_play(n_games, first_game_id, cfg, temperature, seed, flags, max_steps, out, stats);
    API_END(c)
}

diee_status diee_self_play_multi(diee_ctx* c, const diee_batch* batches, uint32_t n_batches, const diee_mcts_cfg* cfg,
                                 float temperature, uint32_t flags, uint32_t max_steps, diee_fragments* outs,
                                 diee_stats* stats) {
    API_BEGIN(c)
    if (!cfg || !batches) throw EngineError(DIEE_ERR_ARG, "bad arguments");
    bg(c)->self_play_multi(batches, n_batches, cfg, temperature, flags, max_steps, outs, stats);
    API_END(c)
}

diee_status diee_dev_conv_bench(diee_ctx* c, int G, int variant, int reps, float* us_mode0, float* us_mode1,
                                float* us_forward) {
    API_BEGIN(c)
    HIPCHK(hipSetDevice(bg(c)->device));
    nn_conv_bench(*bg(c), G, variant, reps, us_mode0, us_mode1, us_forward);
    API_END(c)
}

diee_status diee_dev_last_dispatch(diee_ctx* c, diee_dev_launch* out, uint32_t cap, uint32_t* n) {
    API_BEGIN(c)
    if (!out || !n) throw EngineError(DIEE_ERR_ARG, "null pointer");
    Engine* e = bg(c);
    *n = 0;
    if (!e->net) return DIEE_OK;
    for (const auto& d : e->net->last_dispatch) {
        if (*n >= cap) break;
        diee_dev_launch& o = out[(*n)++];
        o.family = d.family; o.geometry = d.geometry; o.boards = d.boards;
        snprintf(o.kernel, sizeof o.kernel, "%s", nn_kernel_name(d.family, d.geometry));
    }
    API_END(c)
}

diee_status diee_dev_dispatch_bands(diee_ctx* c, int upto_boards, diee_dev_band* out, uint32_t cap, uint32_t* n) {
    API_BEGIN(c)
    if (!out || !n || upto_boards < 1) throw EngineError(DIEE_ERR_ARG, "bad arguments");
    *n = 0;
    for (const auto& b : nn_dispatch_bands(*bg(c), upto_boards)) {
        if (*n >= cap) break;
        diee_dev_band& o = out[(*n)++];
        o.boards_min = b.boards_min; o.boards_max = b.boards_max; o.family = b.family; o.geometry = b.geometry;
        snprintf(o.kernel, sizeof o.kernel, "%s", nn_kernel_name(b.family, b.geometry));
    }
    API_END(c)
}

diee_status diee_dev_rules_bench(diee_ctx* c, const diee_bg_state* states, uint32_t n, int reps, float* us_legal_moves,
                                 float* mean_plays) {
    API_BEGIN(c)
    if (!states || !n || reps <= 0 || !us_legal_moves || !mean_plays) throw EngineError(DIEE_ERR_ARG, "bad arguments");
    bg(c)->rules_bench(states, n, reps, us_legal_moves, mean_plays);
    API_END(c)
}

diee_status diee_dev_wave_selftest(diee_ctx* c, uint32_t salt, uint32_t* mismatches) {
    API_BEGIN(c)
    if (!mismatches) throw EngineError(DIEE_ERR_ARG, "bad arguments");
    Engine* e = bg(c);
    HIPCHK(hipSetDevice(e->device));
    e->tmp_c.ensure(4);
    HIPCHK(hipMemsetAsync(e->tmp_c.p, 0, 4, e->stream));
    launch_wave_selftest(e->stream, (uint32_t*)e->tmp_c.p, salt);
    e->d2h((uint8_t*)mismatches, e->tmp_c.p, 4);
    e->sync();
    API_END(c)
}

// ---- training-step kernels: stateless, on the caller's stream ----
static const float* zero_bias256() {
    static float* z[16] = {nullptr};                             // one 1 KB allocation per device the process trains on
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!z[dev]) {
        if (hipMalloc((void**)&z[dev], 256 * sizeof(float)) != hipSuccess) { z[dev] = nullptr; return nullptr; }
        (void)hipMemset(z[dev], 0, 256 * sizeof(float));
    }
    return z[dev];
}
diee_status diee_train_pack_conv3x3(const float* w, void* wpack, int transpose, void* stream) {
    if (!w || !wpack) return DIEE_ERR_ARG;
    launch_pack_conv_w((hipStream_t)stream, w, (uint16_t*)wpack, transpose);
    return hipGetLastError() == hipSuccess ? DIEE_OK : DIEE_ERR_HIP;
}
diee_status diee_train_pack_conv3x3_multi(const float* const* w, int n, void* wpack, void* stream) {
    if (!w || !wpack || n <= 0 || n > kMaxPackLayers) return DIEE_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (!w[i]) return DIEE_ERR_ARG;
    launch_pack_conv_w_multi((hipStream_t)stream, w, n, (uint16_t*)wpack);
    return hipGetLastError() == hipSuccess ? DIEE_OK : DIEE_ERR_HIP;
}
diee_status diee_train_conv3x3(const void* x, const void* wpack, const float* bias, void* y, int boards, void* stream) {
    if (!x || !wpack || !y || boards <= 0) return DIEE_ERR_ARG;
    if (!bias) bias = zero_bias256();
    if (!bias) return DIEE_ERR_HIP;
    launch_conv3x3((hipStream_t)stream, 256, 3, (const uint16_t*)x, wpack, bias, nullptr, (uint16_t*)y, nullptr, boards, 256);
    return hipGetLastError() == hipSuccess ? DIEE_OK : DIEE_ERR_HIP;
}
diee_status diee_train_im2col3x3(const void* x, void* col, int boards, void* stream) {
    if (!x || !col || boards <= 0) return DIEE_ERR_ARG;
    launch_im2col3x3((hipStream_t)stream, (const uint16_t*)x, (uint16_t*)col, boards);
    return hipGetLastError() == hipSuccess ? DIEE_OK : DIEE_ERR_HIP;
}

size_t diee_train_scratch_floats(int rows) { return (size_t)train_stripes(rows) * 768 + 1280; }   // partials + the apply pass's coefficients
diee_status diee_train_bn_relu_fwd(const void* x, const void* res, const float* gamma, const float* beta, float* run_mean,
                                   float* run_var, float momentum, float eps, float* save_mean, float* save_invstd, void* y, int rows,
                                   float* scratch, void* stream) {
    if (!x || !gamma || !beta || !save_mean || !save_invstd || !y || !scratch || rows <= 0) return DIEE_ERR_ARG;
    launch_bn_relu_fwd((hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)res, gamma, beta, scratch, save_mean, save_invstd,
                       run_mean, run_var, momentum, eps, (uint16_t*)y, rows);
    return hipGetLastError() == hipSuccess ? DIEE_OK : DIEE_ERR_HIP;
}
diee_status diee_train_bn_relu_bwd(const void* dy, const void* y, const void* x, const float* gamma, const float* save_mean,
                                   const float* save_invstd, float* dgamma, float* dbeta, void* dx, void* dres, float* dx_colsum,
                                   int rows, float* scratch, void* stream) {
    if (!dy || !y || !x || !gamma || !save_mean || !save_invstd || !dgamma || !dbeta || !dx || !scratch || rows <= 0) return DIEE_ERR_ARG;
    launch_bn_relu_bwd((hipStream_t)stream, (const uint16_t*)dy, (const uint16_t*)y, (const uint16_t*)x, gamma, save_mean, save_invstd,
                       scratch, dgamma, dbeta, (uint16_t*)dx, (uint16_t*)dres, dx_colsum, rows);
    return hipGetLastError() == hipSuccess ? DIEE_OK : DIEE_ERR_HIP;
}
size_t diee_train_wgrad_scratch_floats(void) { return wgrad_scratch_floats(); }
diee_status diee_train_wgrad3x3(const void* x, const void* dy, float* dw, int boards, float* scratch, void* stream) {
    if (!x || !dy || !dw || !scratch || boards <= 0) return DIEE_ERR_ARG;
    launch_wgrad3x3((hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)dy, scratch, dw, boards);
    return hipGetLastError() == hipSuccess ? DIEE_OK : DIEE_ERR_HIP;
}
diee_status diee_train_set_bn_coop(int on) { bn_coop_set(on); return DIEE_OK; }
int diee_train_bn_coop_timeouts(int clear) { return bn_coop_poll_timeouts(clear); }
diee_status diee_train_colsum(const void* a, float* out, int rows, float* scratch, void* stream) {
    if (!a || !out || !scratch || rows <= 0) return DIEE_ERR_ARG;
    launch_colsum((hipStream_t)stream, (const uint16_t*)a, scratch, out, rows);
    return hipGetLastError() == hipSuccess ? DIEE_OK : DIEE_ERR_HIP;
}

void diee_free_fragments(diee_fragments* f) {
    if (!f) return;
    // the backgammon engine hands out page-locked blocks of its pool (search_host.cpp), the tic-tac-toe host path malloc'ed ones
    void* ps[4] = {f->outcome, f->ps, f->state, f->game};
    for (void* p : ps) if (!diee::pinned_release(p)) free(p);
    memset(f, 0, sizeof *f);
}

}  // extern "C"
