// Shared declarations for libmvector_hip.so (gfx950 / CDNA4 only).
//
// The kernels are written for wave64, MFMA and 160 KiB LDS; there is no CUDA path and no CPU
// fallback.  Everything that is not plain HIP C++ (DPP / permlane moves, LDS-DMA, counted waits,
// inline-assembly MFMA steps, launch macros) is spelled once, in <arch/gfx950.h>.  The test-suite
// additionally compiles these same sources for the host against tests/emu (a SIMT emulator with its
// own arch/gfx950.h in front of the include path) so that indexing and fragment layouts can be
// checked in a container without a GPU; no source here knows about that build.
#pragma once

#include <arch/gfx950.h>  // the gfx950 spellings (csrc/arch) -- or, in the test-suite's emulator build, their host twins (tests/emu/arch)

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/mvector_hip.h"

namespace mv {

constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_ROR1 = 0x121, DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141,
              DPP_ROW_BCAST15 = 0x142;  // lanes of rows 1..3 read lane 15 of the row in front of them (row 0 keeps `old`)

// sum over the 16 lanes of a DPP row, result in every lane
// (the form with old = 0 folds into one v_add_f32_dpp per step; the builtin without an `old` operand stays a v_mov_b32_dpp + v_add)
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
