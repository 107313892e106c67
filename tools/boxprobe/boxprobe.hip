// Box yard-stick for bench.py's `box` block: three numbers that say what THIS gpurun box offers, measured in the same
// process right before the timed region, so that driver numbers of different rounds / boxes can be normalised:
//   bp_copy  -- streaming copy (16 B per lane, grid-stride), GB/s over >= 1 GB: what the HBM path of this box sustains;
//   bp_mfma  -- bare v_mfma_f32_16x16x32_f16 issue (8 independent accumulators per wave, two waves per SIMD, one 512-thread workgroup per
//               CU -- the residency of the ring GEMM), no memory traffic: the matrix pipes' TFLOP/s at the clock the box SUSTAINS under
//               that load, plus that clock itself: wave 0 of every workgroup reads s_memtime (shader clock cycles) and s_memrealtime
//               (100 MHz constant reference) at both ends; clock = d(memtime) / d(memrealtime) * 100 MHz.
// Test / measurement infrastructure: built into tools/probe/libmvector_boxprobe.so by tools/boxprobe/build.py, loaded by bench.py only.
// Nothing in the product library or package refers to it.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void bp_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four loads in flight per lane
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// ticks[wg] = {memtime0, memtime1, realtime0, realtime1}; sink keeps the accumulators alive
__global__ __launch_bounds__(512) void bp_mfma_kernel(unsigned long long* ticks, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    half8v a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.001f * (float)(lane + i));
        b[i] = (_Float16)(0.002f * (float)(lane - i));
    }
    float4v c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = float4v{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[j], 0, 0, 0);
        }
    }
    if (threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        ticks[4 * blockIdx.x + 0] = t0;
        ticks[4 * blockIdx.x + 1] = t1;
        ticks[4 * blockIdx.x + 2] = r0;
        ticks[4 * blockIdx.x + 3] = r1;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    if (s == 12345.678f) sink[0] = s;  // never true: keeps the MFMAs
}

extern "C" {

// copies `bytes` (a multiple of 16) from src to dst on `stream`; returns 0 or the hipError_t
__attribute__((visibility("default"))) int bp_copy(void* dst, const void* src, size_t bytes, int workgroups, hipStream_t stream) {
    if (bytes % 16 != 0 || workgroups <= 0) return -1;
    hipLaunchKernelGGL(bp_copy_kernel, dim3(workgroups), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, bytes / 16);
    return (int)hipGetLastError();
}

// `workgroups` x 512 threads, every wave issues iters * 32 MFMAs; FLOPs of the launch = workgroups * 8 waves * iters * 32 * 16384
__attribute__((visibility("default"))) int bp_mfma(unsigned long long* ticks, float* sink, int iters, int workgroups, hipStream_t stream) {
    if (iters <= 0 || workgroups <= 0) return -1;
    hipLaunchKernelGGL(bp_mfma_kernel, dim3(workgroups), dim3(512), 0, stream, ticks, sink, iters);
    return (int)hipGetLastError();
}

}  // extern "C"
