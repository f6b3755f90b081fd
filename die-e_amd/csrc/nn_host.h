// nn_host.h -- device-resident network state shared between nn_host.cpp and the search driver.
#pragma once
#include <vector>

#include "engine.h"

namespace diee {

struct NetWeights {
    DevBuf<uint16_t> wconv[40];     // packed bf16 B fragments: 0 init, 1..38 tower, 39 heads
    DevBuf<float> bconv[40];        // folded bias
    DevBuf<uint16_t> wfc;           // policy FC fragments
    DevBuf<float> bfc, wv;          // policy FC bias [1376]; value FC weights [72] + bias
    bool loaded = false;
    // activations (bf16 NHWC rows [g*24+p][C])
    DevBuf<uint16_t> x16, actX, actH, hp;
    DevBuf<float> hv, logits;
    int cap_games = 0;
    // sampled HIP-event timing of the tower conv kernel
    struct Pending { hipEvent_t a, b; double flops; int launches; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_events;
    int sample_every = 17;
    uint64_t forward_count = 0;
    double conv_seconds = 0, conv_flops = 0;
    uint64_t conv_launches = 0;
    hipEvent_t get_event() {
        if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
        hipEvent_t e; HIPCHK(hipEventCreate(&e)); return e;
    }
    ~NetWeights() {
        for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        for (auto e : free_events) (void)hipEventDestroy(e);
    }
};

void nn_reserve(Engine& e, int G);
void nn_forward(Engine& e, const void* states_dev, int G, float* policy_dev, float* value_dev);
void nn_harvest(Engine& e, diee_stats* stats);
void nn_reset_timing(Engine& e);
void nn_conv_bench(Engine& e, int G, int variant, int reps, float* us_mode0, float* us_mode1, float* us_forward);

}  // namespace diee
