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

// DPP lane moves inside rows of 16 lanes (VALU, no LDS traffic).  CTRL is the hardware dpp_ctrl value:
//   0x00..0xFF quad_perm, 0x110+n row_shr:n, 0x120+n row_ror:n, 0x140 row_mirror, 0x141 row_half_mirror.
// Lanes without a source lane (row_shr) keep `old`.
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_ROR1 = 0x121, DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141;
#ifdef MV_EMU
template <int CTRL>
inline float dpp_mov(float old, float src) {
    const int lane = emu::flat_tid() & 63;
    int from = lane;
    bool has = true;
    if (CTRL < 0x100) {
        from = (lane & ~3) | ((CTRL >> (2 * (lane & 3))) & 3);
    } else if (CTRL > 0x100 && CTRL < 0x110) {  // row_shl:n -- lane i reads lane i + n of its row
        has = (lane & 15) + (CTRL - 0x100) <= 15;
        from = has ? lane + (CTRL - 0x100) : lane;
    } else if (CTRL > 0x110 && CTRL < 0x120) {
        has = (lane & 15) >= (CTRL - 0x110);
        from = has ? lane - (CTRL - 0x110) : lane;
    } else if (CTRL > 0x120 && CTRL < 0x130) {
        from = (lane & ~15) | ((lane - (CTRL - 0x120)) & 15);
    } else if (CTRL == 0x140) {
        from = (lane & ~15) | (15 - (lane & 15));
    } else if (CTRL == 0x141) {
        from = (lane & ~7) | (7 - (lane & 7));
    }
    const float v = emu::shfl_from(src, from);
    return has ? v : old;
}
#else
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                                 0xf, 0xf, false));
}
#endif

// Pointers with their address space stated: two branches that store the same values once to LDS and once to global memory
// are otherwise tail-merged into ONE flat store behind a selected base pointer.
#ifdef MV_EMU
#define MV_AS_LDS(T, p) (p)
#define MV_AS_GLOBAL(T, p) (p)
#else
#define MV_AS_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))
#define MV_AS_GLOBAL(T, p) ((__attribute__((address_space(1))) T*)(p))
#endif

// value the optimiser must treat as freshly computed here: keeps per-tile addresses from being hoisted out of a loop nest
// into dozens of long-lived registers
#ifdef MV_EMU
#define MV_OPAQUE(x) ((void)0)
#else
#define MV_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

// a value that is the same in every lane of the wave (wave index, loop counters derived from it), moved to a scalar
// register so that branches on it are scalar branches instead of exec-masked regions
#ifdef MV_EMU
#define MV_UNIFORM(x) (x)
#else
#define MV_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

// instruction-order hint for the machine scheduler: the next `n` instructions of class `mask` (0x008 MFMA, 0x100 DS read,
// 0x200 DS write, 0x020 VMEM read) form one group, groups are emitted in the order the hints are written
#ifdef MV_EMU
#define MV_SCHED_GROUP(mask, n) ((void)0)
#else
#define MV_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#endif

// 8-byte LDS load that the load/store merger leaves alone (volatile, with the LDS address space stated: a volatile access
// through a generic pointer would become a flat load)
__device__ __forceinline__ float2v lds_load_unmerged(const float2v* p) {
#ifdef MV_EMU
    return *p;
#else
    return *(const volatile __attribute__((address_space(3))) float2v*)(p);
#endif
}

// DPP move for patterns in which every lane has a source lane (mirror, rotate, quad_perm): no `old` operand to initialise
template <int CTRL>
__device__ __forceinline__ float dpp_mov_all(float src) {
#ifdef MV_EMU
    return dpp_mov<CTRL>(src, src);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
#endif
}

// v_permlane16_swap (gfx950): the odd 16-lane rows of x trade places with the even rows of y
//   x: row1 <- y.row0, row3 <- y.row2      y: row0 <- x.row1, row2 <- x.row3      (checked on the device: tools/permlane_probe.hip)
__device__ __forceinline__ void row_swap_odd_even(unsigned& x, unsigned& y) {
#ifdef MV_EMU
    const int lane = emu::flat_tid() & 63;
    const bool odd = (lane >> 4) & 1;
    const unsigned from_y = emu::shfl_from(y, lane - 16), from_x = emu::shfl_from(x, lane + 16);  // out-of-row sources are unused
    const unsigned nx = odd ? from_y : x, ny = odd ? y : from_x;
    x = nx;
    y = ny;
#else
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0];
    y = r[1];
#endif
}

// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<DPP_QUAD_XOR1>(0.0f, v);
    v += dpp_mov<DPP_QUAD_XOR2>(0.0f, v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(0.0f, v);
    v += dpp_mov<DPP_ROW_MIRROR>(0.0f, v);
    return v;
}

// thread-local error string behind mv_last_error()
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// returns MV_OK or records "what: hip error string"
int check_launch(const char* what);

// optional HIP-event timing of a launch (mv_profile_enable): token = prof_begin(MV_PROF_*, algorithmic work, stream)
// before the launch, prof_end(token, stream) after it; both are no-ops while profiling is off
int prof_begin(int kernel_class, double work, hipStream_t stream);
void prof_end(int token, hipStream_t stream);

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
