// "S16" maps of the split-fp16 convolutions (conv2ds.hip): channel-last, 4 bytes per channel, a unit of 16 channels stored as
// [16 x hi | 16 x lo] fp16 of V = 64 * value with hi = fp16(V), lo = fp16(V - hi).  Four consecutive channels of a unit at a time.
#pragma once
#include "common.h"

namespace mv {

constexpr float CS_XSCALE = 64.0f;
constexpr float CS_XSCALE_INV = 1.0f / 64.0f;

// Stored values are clamped to the fp16 range (|V| <= 65504, i.e. |value| <= 1023.5 at the scale of 64): the split SATURATES there.  A NaN is not a
// number to clamp -- fminf / fmaxf would turn it into a bound and hide it -- so it travels on.  Kernels that store S16 maps can report the largest
// |V| they wanted to store (before the clamp) to a device word (s16_peak_*: the float's bits, monotonic under an unsigned max), which is how a
// handle learns that its maps need a smaller scale (campplus.hip: the exact head's gain) and how saturation on real inputs becomes visible.
__device__ __forceinline__ float s16_clamp(float v, float lo, float hi) { return v != v ? v : fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float s16_clamp(float v) { return s16_clamp(v, -65504.0f, 65504.0f); }
__device__ __forceinline__ float s16_peak_of(float pk, const float4v& X) {   // (fmaxf drops NaNs: they are reported by the data itself)
    return fmaxf(fmaxf(pk, fmaxf(fabsf(X[0]), fabsf(X[1]))), fmaxf(fabsf(X[2]), fabsf(X[3])));
}
// every lane of the wave calls it with its running peak; one atomic per wave
__device__ __forceinline__ void s16_peak_commit(unsigned* dst, float pk) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) pk = fmaxf(pk, __shfl_xor(pk, off));
    // (a relaxed look at the word first: tens of thousands of waves of a launch report to ONE address, and same-address atomics serialise in L2 --
    //  after the first few, a wave's maximum is almost never a new one)
    if ((threadIdx.x & 63) == 0 && pk > 0.0f) {
        const unsigned bits = __builtin_bit_cast(unsigned, pk);
        if (bits > __atomic_load_n(dst, __ATOMIC_RELAXED)) atomicMax(dst, bits);
    }
}

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
        const float X = s16_clamp(v[r] * CS_XSCALE);
        h[r] = (half_t)X;
        l[r] = (half_t)(X - (float)h[r]);
    }
    *reinterpret_cast<half4v*>(p) = h;
    *reinterpret_cast<half4v*>(p + 16) = l;
}
// Wave-level forms for MFMA epilogues (lane = (pixel j16, channel quadruple q = lane / 16) of a unit): X = the four scaled values of the lane.
// s16_swap4 splits them and trades register halves between the odd and even 16-lane rows (v_permlane16_swap) so that every lane holds 16
// contiguous bytes of the unit -- q = 0: hi of channels 0-7, 1: lo of 0-7, 2: hi of 8-15, 3: lo of 8-15, i.e. byte (q & 1) * 32 + (q >> 1) * 16;
// s16_unswap4 is the way back (the exchange is its own inverse): the sum hi + lo of the lane's own four channels, still scaled.
// Every lane of the wave has to call them.
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4v s16_swap4(const float4v& X) {
    half4v h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        h[r] = (half_t)X[r];
        l[r] = (half_t)(X[r] - (float)h[r]);
    }
    unsigned h0 = __builtin_bit_cast(unsigned, half2v{h[0], h[1]}), h1 = __builtin_bit_cast(unsigned, half2v{h[2], h[3]});
    unsigned l0 = __builtin_bit_cast(unsigned, half2v{l[0], l[1]}), l1 = __builtin_bit_cast(unsigned, half2v{l[2], l[3]});
    row_swap_odd_even(h0, l0);
    row_swap_odd_even(h1, l1);
    return uint4v{h0, h1, l0, l1};
}
__device__ __forceinline__ float4v s16_unswap4(const uint4v& w) {
    unsigned h0 = w[0], h1 = w[1], l0 = w[2], l1 = w[3];
    row_swap_odd_even(h0, l0);
    row_swap_odd_even(h1, l1);
    const half2v a = __builtin_bit_cast(half2v, h0), b = __builtin_bit_cast(half2v, h1), c = __builtin_bit_cast(half2v, l0), d = __builtin_bit_cast(half2v, l1);
    return float4v{(float)a[0] + (float)c[0], (float)a[1] + (float)c[1], (float)b[0] + (float)d[0], (float)b[1] + (float)d[1]};
}

// one channel (scalar code paths: pooling)
__device__ __forceinline__ float s16_load1(const half_t* map, int64_t pixel_ld_offset, int c) {
    const half_t* u = map + (pixel_ld_offset + (c & ~15)) * 2 + (c & 15);
    return ((float)u[0] + (float)u[16]) * CS_XSCALE_INV;
}

}  // namespace mv
