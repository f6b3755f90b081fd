// engine.cpp -- Engine construction and the pure game functions.
#include "engine.h"

#include <cstring>
#include <vector>

#include "launch.h"

namespace diee {

void free_net(NetWeights*);
void cluster_baton_register(int device, int delta);
void nn_reset_cluster(Engine& e);
void nn_disable_cluster(Engine& e);
void free_search(SearchBufs*);

Engine::Engine(int dev) : device(dev) {
    err[0] = 0;
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
                                            "both are now off for this process -- set DIEE_TOWER_CL=none DIEE_TOWER_PAIR=0 to start without them");
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
