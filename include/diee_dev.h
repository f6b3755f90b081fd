/*
 * diee_dev.h -- development entry points of libdiee.so: arithmetic probes for the bit-exactness tests and kernel timing
 * probes for scripts/.  NOT part of the drop-in boundary (include/diee.h): nothing on the product path (the Python package,
 * the CLI, bench.py's timed region) calls these, and a binding of the reference does not need them.
 */
#ifndef DIEE_DEV_H
#define DIEE_DEV_H
#include "diee.h"
#ifdef __cplusplus
extern "C" {
#endif

/* device arithmetic probes for the bit-exactness tests (f32 sqrt/div, det_pow, Philox dice) */
diee_status diee_probe_f32(diee_ctx*, const float* a, const float* b, uint32_t n,
                           float* sqrt_a, float* a_div_b, float* pow_ab);
diee_status diee_probe_dice(diee_ctx*, uint64_t seed, const uint32_t* ctr /*[n][4]*/, uint32_t n,
                            uint8_t* dice /*[n][2]*/, double* uniform /*[n]*/);

/* development probe: average device time (us, HIP events) of the 3x3 tower conv kernel without /
 * with the residual epilogue at batch G, and of a whole forward pass; variant 0 = geometry picked
 * by batch size; 1..4 per-layer (8x128ch/4 waves, 4x128/4, 2x64/2, 2x32/1 boards x channels/waves);
 * 5/6/7 and 17/18 split-K over 4 / 8 waves; 100..109 whole tower in one launch (fused geometries);
 * 201/202/204/208 cluster tower with 1/2/4/8 boards per cluster */
diee_status diee_dev_conv_bench(diee_ctx*, int G, int variant, int reps, float* us_mode0,
                                float* us_mode1, float* us_forward);
/* development probe (stand-alone step kernel): average device time (us, HIP events) of one
 * get_valid_moves launch over n states already resident in HBM (one wave per state), and the mean number of plays */
diee_status diee_dev_rules_bench(diee_ctx*, const diee_bg_state* states, uint32_t n, int reps,
                                 float* us_legal_moves, float* mean_plays);

/* which tower kernel ran: the launches of the ctx's LAST network evaluation (diee_nn_forward, or the last one of a search), in order, and
 * the dispatch of a plain evaluation by its board count as bands [boards_min, boards_max] -> kernel (what NetWeights::tower_table /
 * cluster_table say for this ctx today).  tests/test_nn_gpu.py derives one tolerance case per band from the second and asserts the
 * first, so that a re-banding moves the cases with it and shows up as a changed expectation instead of as silently moved coverage.
 * family: 0 per-layer kernels, 1 fused tower (k_tower16), 2 pair tower (k_tower16p), 3 cluster tower (k_tower_cl), 4 k_tower (32x32x16) */
typedef struct { int family, geometry, boards; char kernel[120]; } diee_dev_launch;
typedef struct { int boards_min, boards_max, family, geometry; char kernel[120]; } diee_dev_band;
diee_status diee_dev_last_dispatch(diee_ctx*, diee_dev_launch* out, uint32_t cap, uint32_t* n);
diee_status diee_dev_dispatch_bands(diee_ctx*, int upto_boards, diee_dev_band* out, uint32_t cap, uint32_t* n);

/* the DPP / permlane forms of the wave-wide operations (csrc/wave_ops.h) against the __shfl forms they replace, on
 * lane-dependent data: the number of lanes x cases that disagree (0 on a correct build) */
diee_status diee_dev_wave_selftest(diee_ctx*, uint32_t salt, uint32_t* mismatches);

#ifdef __cplusplus
}
#endif
#endif
