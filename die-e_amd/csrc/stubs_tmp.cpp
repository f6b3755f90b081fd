#include "engine.h"
namespace diee {
void free_net(NetWeights*) {}
void free_search(SearchBufs*) {}
size_t weights_count_bg() { return 0; }
void random_weights_bg(uint64_t, float*) {}
void Engine::load_weights(const float*, size_t) { throw EngineError(DIEE_ERR_UNSUPPORTED, "nyi"); }
void Engine::nn_forward_host(const diee_bg_state*, uint32_t, float*, float*) { throw EngineError(DIEE_ERR_UNSUPPORTED, "nyi"); }
void Engine::mcts_batch(const diee_bg_state*, uint32_t, const diee_mcts_cfg*, uint64_t, uint32_t, const uint32_t*, const uint32_t*, uint32_t, float*, uint32_t*, float*, diee_stats*) { throw EngineError(DIEE_ERR_UNSUPPORTED, "nyi"); }
void Engine::self_play(uint32_t, uint32_t, const diee_mcts_cfg*, float, uint64_t, uint32_t, uint32_t, diee_fragments*, diee_stats*) { throw EngineError(DIEE_ERR_UNSUPPORTED, "nyi"); }
}
