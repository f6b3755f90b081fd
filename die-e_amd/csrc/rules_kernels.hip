// rules_kernels.hip -- batched pure game functions (parity-test surface of the C ABI) for gfx950.
// One wavefront per state for legal-play enumeration; one lane per state for the scalar functions.
#include "bg_device.h"
#include "launch.h"
#include "wave_ops.h"

namespace diee {

// get_valid_moves for n states: plays[n][cap] packed (f1,t1,f2,t2), counts[n]
__global__ __launch_bounds__(64) void k_legal_moves(const BgState* __restrict__ states, uint32_t n,
                                                    uint32_t* __restrict__ plays, uint32_t cap,
                                                    uint32_t* __restrict__ counts, uint32_t* overflow) {
    __shared__ WaveScratch sc;
    const int lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const BgState s = states[i];
        const int k = bg_legal_plays_wave(s, &sc, lane, overflow);
        if (lane == 0) counts[i] = (uint32_t)k;
        for (int j = lane; j < k && j < (int)cap; j += 64) plays[(size_t)i * cap + j] = sc.play[j];
        __syncthreads();
    }
}

__global__ void k_encode(const BgState* __restrict__ states, const uint32_t* __restrict__ plays, uint32_t n,
                         uint32_t* __restrict__ codes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BgState s = states[i];
    codes[i] = bg_encode_dev(st_roll(s, 0), st_roll(s, 1), plays[i]);
}

__global__ void k_decode(const BgState* __restrict__ states, const uint32_t* __restrict__ codes, uint32_t n,
                         uint32_t* __restrict__ plays) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BgState s = states[i];
    plays[i] = bg_decode_dev(st_roll(s, 0), st_roll(s, 1), st_player(s), codes[i]);
}

__global__ void k_apply(BgState* __restrict__ states, const uint32_t* __restrict__ plays,
                        const uint8_t* __restrict__ dice, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BgState s = states[i];
    bg_apply_dev(s, plays[i], dice[2 * i], dice[2 * i + 1]);
    states[i] = s;
}

__global__ void k_planes(const BgState* __restrict__ states, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x;      // one block of 144 threads per state
    if (i >= n) return;
    const BgState s = states[i];
    const int t = threadIdx.x;
    if (t < 144) out[(size_t)i * 144 + t] = bg_plane_dev(s, t / 24, t % 24);
}

__global__ void k_probe_f32(const float* a, const float* b, uint32_t n, float* sq, float* dv, float* pw) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sq[i] = sqrtf(a[i]);
    dv[i] = a[i] / b[i];
    pw[i] = det_powf(a[i], b[i]);
}

__global__ void k_probe_dice(uint64_t seed, const uint32_t* ctr, uint32_t n, uint8_t* dice, double* uni) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int d0, d1;
    draw_dice(seed, ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], d0, d1);
    dice[2 * i] = (uint8_t)d0; dice[2 * i + 1] = (uint8_t)d1;
    uni[i] = draw_uniform(seed, ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3]);
}

// ---- host launchers -----------------------------------------------------------------------------
void launch_legal_moves(hipStream_t st, const void* states, uint32_t n, uint32_t* plays, uint32_t cap,
                        uint32_t* counts, uint32_t* overflow) {
    if (!n) return;
    const uint32_t grid = n < 4096u ? n : 4096u;
    hipLaunchKernelGGL(k_legal_moves, dim3(grid), dim3(64), 0, st, (const BgState*)states, n, plays, cap, counts, overflow);
}
void launch_encode(hipStream_t st, const void* states, const uint32_t* plays, uint32_t n, uint32_t* codes) {
    if (!n) return;
    hipLaunchKernelGGL(k_encode, dim3((n + 255) / 256), dim3(256), 0, st, (const BgState*)states, plays, n, codes);
}
void launch_decode(hipStream_t st, const void* states, const uint32_t* codes, uint32_t n, uint32_t* plays) {
    if (!n) return;
    hipLaunchKernelGGL(k_decode, dim3((n + 255) / 256), dim3(256), 0, st, (const BgState*)states, codes, n, plays);
}
void launch_apply(hipStream_t st, void* states, const uint32_t* plays, const uint8_t* dice, uint32_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_apply, dim3((n + 255) / 256), dim3(256), 0, st, (BgState*)states, plays, dice, n);
}
void launch_planes(hipStream_t st, const void* states, uint32_t n, float* out) {
    if (!n) return;
    hipLaunchKernelGGL(k_planes, dim3(n), dim3(192), 0, st, (const BgState*)states, n, out);
}
void launch_probe_f32(hipStream_t st, const float* a, const float* b, uint32_t n, float* sq, float* dv, float* pw) {
    if (!n) return;
    hipLaunchKernelGGL(k_probe_f32, dim3((n + 255) / 256), dim3(256), 0, st, a, b, n, sq, dv, pw);
}
void launch_probe_dice(hipStream_t st, uint64_t seed, const uint32_t* ctr, uint32_t n, uint8_t* dice, double* uni) {
    if (!n) return;
    hipLaunchKernelGGL(k_probe_dice, dim3((n + 255) / 256), dim3(256), 0, st, seed, ctr, n, dice, uni);
}

// wave_ops.h against the LDS-pipe shuffles it replaces, on lane-dependent data (diee_dev_wave_selftest): mismatching lanes
__global__ __launch_bounds__(64) void k_wave_selftest(uint32_t* mismatches, uint32_t salt) {
    const int lane = threadIdx.x;
    uint32_t bad = 0;
    for (uint32_t rep = 0; rep < 8; ++rep) {
        const int v = (int)(((uint32_t)lane * 2654435761u) ^ ((salt + rep) * 0x9E3779B9u) ^ ((uint32_t)lane << (rep + 3)));
        bad += wave_xor_i32<1>(v) != __shfl_xor(v, 1);   bad += wave_xor_i32<2>(v) != __shfl_xor(v, 2);
        bad += wave_xor_i32<4>(v) != __shfl_xor(v, 4);   bad += wave_xor_i32<8>(v) != __shfl_xor(v, 8);
        bad += wave_xor_i32<16>(v) != __shfl_xor(v, 16); bad += wave_xor_i32<32>(v) != __shfl_xor(v, 32);
        int mx = v;
        for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
        bad += wave_allmax_i32(v) != mx;
        const float f = (float)(v >> 8) * 0.37f;
        float fm = f;
        for (int d = 32; d >= 1; d >>= 1) fm = fmaxf(fm, __shfl_xor(fm, d));
        bad += wave_allmax_f32(f) != fm;
        float fs = f;
        for (int d = 32; d >= 1; d >>= 1) fs += __shfl_xor(fs, d);
        bad += __builtin_bit_cast(uint32_t, wave_butterfly_sum(f)) != __builtin_bit_cast(uint32_t, fs);
        const int x = (v >> 20) & 31;
        int incl = x;
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        bad += wave_inclusive_scan_i32(x) != incl;
    }
    atomicAdd(mismatches, bad);
}
void launch_wave_selftest(hipStream_t st, uint32_t* mismatches, uint32_t salt) {
    hipLaunchKernelGGL(k_wave_selftest, dim3(4), dim3(64), 0, st, mismatches, salt);
}

}  // namespace diee
