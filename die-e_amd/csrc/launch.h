// launch.h -- host-side launch wrappers, one group per kernel translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace diee {

// rules_kernels.hip
void launch_legal_moves(hipStream_t st, const void* states, uint32_t n, uint32_t* plays, uint32_t cap,
                        uint32_t* counts, uint32_t* overflow);
void launch_encode(hipStream_t st, const void* states, const uint32_t* plays, uint32_t n, uint32_t* codes);
void launch_decode(hipStream_t st, const void* states, const uint32_t* codes, uint32_t n, uint32_t* plays);
void launch_apply(hipStream_t st, void* states, const uint32_t* plays, const uint8_t* dice, uint32_t n);
void launch_planes(hipStream_t st, const void* states, uint32_t n, float* out);
void launch_probe_f32(hipStream_t st, const float* a, const float* b, uint32_t n, float* sq, float* dv, float* pw);
void launch_probe_dice(hipStream_t st, uint64_t seed, const uint32_t* ctr, uint32_t n, uint8_t* dice, double* uni);

}  // namespace diee
