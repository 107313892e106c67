// "S16" maps of the split-fp16 convolutions (conv2ds.hip): channel-last, 4 bytes per channel, a unit of 16 channels stored as
// [16 x hi | 16 x lo] fp16 of V = 64 * value with hi = fp16(V), lo = fp16(V - hi).  Four consecutive channels of a unit at a time.
#pragma once
#include "common.h"

namespace mv {

constexpr float CS_XSCALE = 64.0f;
constexpr float CS_XSCALE_INV = 1.0f / 64.0f;

// p = position of the hi quadruple: unit base + 4 * (channel quadruple inside the unit)
__device__ __forceinline__ float4v s16_load4(const half_t* p) {
    const half4v h = *reinterpret_cast<const half4v*>(p), l = *reinterpret_cast<const half4v*>(p + 16);
    float4v v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = ((float)h[r] + (float)l[r]) * CS_XSCALE_INV;
    return v;
}
__device__ __forceinline__ void s16_store4(half_t* p, const float4v& v) {
    half4v h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float X = fminf(fmaxf(v[r] * CS_XSCALE, -65504.0f), 65504.0f);
        h[r] = (half_t)X;
        l[r] = (half_t)(X - (float)h[r]);
    }
    *reinterpret_cast<half4v*>(p) = h;
    *reinterpret_cast<half4v*>(p + 16) = l;
}
// one channel (scalar code paths: pooling)
__device__ __forceinline__ float s16_load1(const half_t* map, int64_t pixel_ld_offset, int c) {
    const half_t* u = map + (pixel_ld_offset + (c & ~15)) * 2 + (c & 15);
    return ((float)u[0] + (float)u[16]) * CS_XSCALE_INV;
}

}  // namespace mv
