// wave_ops.h -- wave-wide all-reductions of order-independent operations (max, arg-max under a total order) on the DPP
// path: quad_perm / row_half_mirror / row_mirror pair up the lanes of a 16-lane row in four VALU instructions, then the four
// rows meet through v_readlane.  __shfl_xor compiles to ds_bpermute_b32 (an LDS-pipe round trip of >100 clocks per step, six
// dependent steps per reduction): a wave alone on its SIMD, as in k_expand, waits all of them out.  Sums stay on the xor
// butterfly where their rounding order is part of a parity contract (nn_device.h).
#pragma once
#include <hip/hip_runtime.h>

namespace diee {

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v))); }
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // lane i <-> 7 - i within 8
constexpr int kDppMirror = 0x140;       // lane i <-> 15 - i within 16

__device__ __forceinline__ int wave_allmax_i32(int v) {
    int o;
    o = dpp_i32<kDppXor1>(v); v = o > v ? o : v;
    o = dpp_i32<kDppXor2>(v); v = o > v ? o : v;
    o = dpp_i32<kDppHalfMirror>(v); v = o > v ? o : v;
    o = dpp_i32<kDppMirror>(v); v = o > v ? o : v;
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

__device__ __forceinline__ float wave_allmax_f32(float v) {
    v = fmaxf(v, dpp_f32<kDppXor1>(v));
    v = fmaxf(v, dpp_f32<kDppXor2>(v));
    v = fmaxf(v, dpp_f32<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_f32<kDppMirror>(v));
    const int i = __builtin_bit_cast(int, v);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)), b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)), d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// inclusive prefix sum over the lanes (integers: any association gives the same sum): a Kogge-Stone scan inside each row of
// 16 with zeros shifted in, then lane 15 of a row added to the next row, lane 31 to the upper half
__device__ __forceinline__ int wave_inclusive_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast31 into rows 2 and 3
    return v;
}

// v of lane (lane ^ D), bit for bit what __shfl_xor(v, D) returns, without the LDS pipe: quad_perm (1, 2), two bank-masked
// row shifts (4), a row rotation (8), gfx950's v_permlane16_swap / v_permlane32_swap (16, 32).  For sums whose pairing order
// is fixed (nn_device.h: the butterfly of the softmax denominator and the value head's dot product).
typedef unsigned int wo_u2 __attribute__((ext_vector_type(2)));
template <int D>
__device__ __forceinline__ int wave_xor_i32(int v) {
    static_assert(D == 1 || D == 2 || D == 4 || D == 8 || D == 16 || D == 32, "one butterfly step");
    if constexpr (D == 1) return dpp_i32<kDppXor1>(v);
    else if constexpr (D == 2) return dpp_i32<kDppXor2>(v);
    else if constexpr (D == 4) {
        int r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);      // row_shl:4 into banks 0, 2: lane i <- i + 4
        r = __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);          // row_shr:4 into banks 1, 3: lane i <- i - 4
        return r;
    } else if constexpr (D == 8) return dpp_i32<0x128>(v);                       // row_ror:8
    else if constexpr (D == 16) {
        const wo_u2 r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return (int)(((threadIdx.x >> 4) & 1) ? r[0] : r[1]);
    } else {
        const wo_u2 r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return (int)(((threadIdx.x >> 5) & 1) ? r[0] : r[1]);
    }
}
template <int D>
__device__ __forceinline__ float wave_xor_f32(float v) { return __builtin_bit_cast(float, wave_xor_i32<D>(__builtin_bit_cast(int, v))); }
// v + the butterfly: ((((v + x32) + x16) + x8) + x4) ... in the order of `for (d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d)`
__device__ __forceinline__ float wave_butterfly_sum(float v) {
    v += wave_xor_f32<32>(v); v += wave_xor_f32<16>(v); v += wave_xor_f32<8>(v);
    v += wave_xor_f32<4>(v); v += wave_xor_f32<2>(v); v += wave_xor_f32<1>(v);
    return v;
}

}  // namespace diee
