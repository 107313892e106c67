// Shared declarations for libmvector_hip.so (gfx950 / CDNA4 only).
//
// The kernels are written for wave64, MFMA and 160 KiB LDS; there is no CUDA path and no CPU
// fallback.  The only second compilation mode is MV_EMU: the test-suite compiles these same
// sources for the host against tests/emu/hip_emu.h (a SIMT emulator) so that indexing and
// fragment layouts can be checked in a container without a GPU.  MV_EMU is never defined when
// building the product library.
#pragma once

#ifndef MV_EMU
#include <hip/hip_runtime.h>
#endif

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/mvector_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

#ifdef MV_EMU
#define MV_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emu::launch(dim3 grid, dim3 block, (shmem), [=]() { kernel(__VA_ARGS__); })
#define MV_DYN_SMEM(name) char* name = MV_EMU_DYN_SMEM()
#define MV_WAVE_FENCE() emu::wave_sync()
#define MV_SET_MAX_SMEM(kernel, bytes) hipSuccess
#else
#define MV_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3 grid, dim3 block, (shmem), (stream), __VA_ARGS__)
#define MV_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
// dynamic LDS above 64 KiB has to be requested per kernel
#define MV_SET_MAX_SMEM(kernel, bytes) \
    hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))
// LDS traffic between lanes of ONE wave: DS ops of a wave execute in program order, so only the
// compiler has to be kept from reordering across this point.
#define MV_WAVE_FENCE()                                       \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#endif

namespace mv {

// thread-local error string behind mv_last_error()
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// returns MV_OK or records "what: hip error string"
int check_launch(const char* what);

}  // namespace mv

#define MV_REQUIRE(cond, msg)                                              \
    do {                                                                   \
        if (!(cond)) return mv::fail(MV_ERR_INVALID_ARGUMENT, std::string(msg) + " [" #cond "]"); \
    } while (0)

#define MV_HIP_OK(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) return mv::fail(MV_ERR_HIP, std::string(#expr ": ") + hipGetErrorString(_e)); \
    } while (0)
