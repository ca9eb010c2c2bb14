// gs_device_utils.h -- wave64 / workgroup primitives for gfx950.  Wave width is hard-coded to 64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gs {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// number of set bits of `mask` below this lane (v_mbcnt_lo/hi)
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v; // every lane holds the sum
}

// Inclusive prefix sum over the wave with six DPP adds (row_shr 1/2/4/8, row_bcast 15/31): lane l gets
// x_0 + ... + x_l, so lane 63 holds the wave total.  Pure VALU (no LDS crossbar traffic, unlike
// __shfl_up / __shfl_xor).  Every lane of the wave must be active at the call.
__device__ __forceinline__ uint32_t wave_sum_to_lane63(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
    return v;
}

// Inclusive prefix maximum (unsigned) over the wave, same six DPP steps with max; 0 is the identity.
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false));   // row_shr:4
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false));   // row_shr:8
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31
    return v;
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {   // all 64 lanes active
    return wave_sum_to_lane63(v);
}

__device__ __forceinline__ uint64_t wave_inclusive_scan64(uint64_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t lo = __shfl_up((uint32_t)v, off, 64);
        uint32_t hi = __shfl_up((uint32_t)(v >> 32), off, 64);
        uint64_t t = ((uint64_t)hi << 32) | lo;
        if (lane >= off) v += t;
    }
    return v;
}

// GLSL clamp(x, lo, hi) = min(max(x, lo), hi) with the comparison forms the oracle uses.
__device__ __forceinline__ float clampf(float x, float lo, float hi) {
    float t = x > lo ? x : lo;
    return t < hi ? t : hi;
}
__device__ __forceinline__ float maxf(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ int clampi(int x, int lo, int hi) {
    return x < lo ? lo : (x > hi ? hi : x);
}

// GLSL int(float) / uint(float): truncate, saturate, NaN -> 0 (what v_cvt_i32_f32 / v_cvt_u32_f32 do;
// written out so the result does not depend on the compiler's treatment of out-of-range casts).
__device__ __forceinline__ int f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}
__device__ __forceinline__ uint32_t f2u_sat(float x) {
    if (x != x) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    if (x <= 0.0f) return 0u;
    return (uint32_t)x;
}

} // namespace gs
