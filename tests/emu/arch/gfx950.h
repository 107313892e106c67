// Host C++ spellings of csrc/arch/gfx950.h for the SIMT emulator (tests/emu/hip_emu.h, force-included before this file).
// TEST INFRASTRUCTURE ONLY: the emulator build puts tests/emu in front of csrc on the include path, so the kernels'
// #include <arch/gfx950.h> finds this file; the product build never sees it.  Same names, same semantics, lane-exact.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef _Float16 half_t;
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float3u __attribute__((ext_vector_type(3), aligned(4)));
// 12 bytes, as the device's global_load_dwordx3 (a host load of the 3-vector type would read 16)
inline float3u load_f32x3(const float* p) {
    float3u v;
    memcpy(&v, p, 12);
    return v;
}
typedef float float16v __attribute__((ext_vector_type(16)));

#define MV_LAUNCH(kernel, grid, block, shmem, stream, ...) emu::launch(dim3 grid, dim3 block, (shmem), [=]() { kernel(__VA_ARGS__); })
#define MV_DYN_SMEM(name) char* name = MV_EMU_DYN_SMEM()
#define MV_SET_MAX_SMEM(kernel, bytes) hipSuccess
#define MV_WAVE_FENCE() emu::wave_sync()
#define MV_LOCKSTEP_POINT() emu::wave_sync()
#define MV_VM_LOADS(n) emu::dma_note(n)
#define MV_AS_LDS(T, p) ((T*)(p))
#define MV_AS_GLOBAL(T, p) ((T*)(p))
#define MV_GLOBAL_PTR(T, p) reinterpret_cast<const T*>(p)
#define MV_OPAQUE(x) ((void)0)
#define MV_UNIFORM(x) (x)
#define MV_SCHED_GROUP(mask, n) ((void)0)

// hipEvent shim for the optional launch profiling (capi.cpp): events are no-ops, elapsed time is zero
typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return 0; }

namespace mv {

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
    } else if (CTRL > 0x110 && CTRL < 0x120) {  // row_shr:n -- lane i reads lane i - n
        has = (lane & 15) >= (CTRL - 0x110);
        from = has ? lane - (CTRL - 0x110) : lane;
    } else if (CTRL > 0x120 && CTRL < 0x130) {  // row_ror:n
        from = (lane & ~15) | ((lane - (CTRL - 0x120)) & 15);
    } else if (CTRL == 0x140) {
        from = (lane & ~15) | (15 - (lane & 15));
    } else if (CTRL == 0x141) {
        from = (lane & ~7) | (7 - (lane & 7));
    } else if (CTRL == 0x142) {  // row_bcast:15 -- rows 1..3 read lane 15 of the row in front of them
        has = lane >= 16;
        from = has ? ((lane >> 4) - 1) * 16 + 15 : lane;
    }
    const float v = emu::shfl_from(src, from);
    return has ? v : old;
}
template <int CTRL>
inline float dpp_mov_all(float src) { return dpp_mov<CTRL>(src, src); }
inline void row16_sum8(float4v& a, float4v& b) {   // (the device form is eight interleaved chains of v_add_f32_dpp: the same additions in the same order)
    for (int r = 0; r < 4; ++r)
        for (int which = 0; which < 2; ++which) {
            float v = which ? b[r] : a[r];
            v += dpp_mov<0xB1>(0.0f, v);
            v += dpp_mov<0x4E>(0.0f, v);
            v += dpp_mov<0x141>(0.0f, v);
            v += dpp_mov<0x140>(0.0f, v);
            if (which) b[r] = v; else a[r] = v;
        }
}
inline void row_swap_odd_even(unsigned& x, unsigned& y) {
    const int lane = emu::flat_tid() & 63;
    const bool odd = (lane >> 4) & 1;
    const unsigned from_y = emu::shfl_from(y, lane - 16), from_x = emu::shfl_from(x, lane + 16);  // out-of-row sources are unused
    const unsigned nx = odd ? from_y : x, ny = odd ? y : from_x;
    x = nx;
    y = ny;
}

inline float lane_gather(float v, int byte_index) { return emu::shfl_from(v, (byte_index >> 2) & 63); }

inline float fmed3(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }   // the median of three, whatever their order (v_med3_f32)

// MODE.FP16_OVFL stand-ins: the emulator saturates where the hardware mode would (finite overflow -> +-65504; infinities and NaNs pass)
inline void fp16_saturation_on() {}
inline half_t half_hwsat(float v) { return (half_t)((v > 65504.0f && v < INFINITY) ? 65504.0f : ((v < -65504.0f && v > -INFINITY) ? -65504.0f : v)); }
template <class V>
inline V pk_add_hwsat(V a, V b) {
    V r = a + b;
    for (unsigned i = 0; i < sizeof(V) / sizeof(half_t); ++i) {
        const float fa = (float)a[i], fb = (float)b[i], s = fa + fb;   // (an overflowed sum of finite operands saturates; inf / NaN operands pass)
        if (fa - fa == 0.0f && fb - fb == 0.0f) r[i] = (half_t)(s > 65504.0f ? 65504.0f : (s < -65504.0f ? -65504.0f : s));
    }
    return r;
}
inline float max_raw(float v, float lo) { return fmaxf(v, lo); }
inline float exp2_fast(float v) { return exp2f(v); }
inline float rcp_fast(float v) { return 1.0f / v; }
inline float sqrt_fast(float v) { return sqrtf(v); }
inline float log2_fast(float v) { return log2f(v); }
inline float4v mfma_4x4x1(float a, float b, float4v c) {  // D[lane][r] += A[4 * (lane / 4) + r] * B[lane]
    const int lane = emu::flat_tid() & 63;
    memcpy(emu::wave_slot(0, lane), &a, 4);
    emu::wave_sync();
    for (int r = 0; r < 4; ++r) {
        float av;
        memcpy(&av, emu::wave_slot(0, (lane & ~3) + r), 4);
        c[r] = fmaf(av, b, c[r]);
    }
    emu::wave_sync();
    return c;
}

inline float2v lds_load_unmerged(const float2v* p) { return *p; }
// "LDS byte address": offset from the start of the emulated block's dynamic LDS
inline unsigned lds_addr(const void* p) { return (unsigned)(reinterpret_cast<const char*>(p) - MV_EMU_DYN_SMEM()); }
inline const char* lds_ptr(unsigned addr) { return MV_EMU_DYN_SMEM() + addr; }
typedef const half_t* lds_half_ptr;
inline lds_half_ptr lds_opaque_half_ptr(const half_t* p) { return p; }
inline half8v lds_load_half8(lds_half_ptr p, int elem_off) { return *reinterpret_cast<const half8v*>(p + elem_off); }
// (the compiler-tracked transfer: cross-wave visibility is the kernel's business too -- a counted wait in front of the barrier -- so the lazy
// mode queues it like the untracked ones)
inline void glds16(const void* gsrc, char* lds_wave_base) { emu::dma_issue(lds_wave_base + (emu::flat_tid() & 63) * 16, gsrc); }
template <int N>
inline void wait_vm() { emu::dma_retire(N); }
template <int N>
inline void wait_vm_seen() { emu::dma_retire(N); }
inline void lds_barrier() {   // (no drain of transfers in flight: s_waitcnt lgkmcnt(0) + s_barrier)
    emu::lds_read_retire(0);
    emu::syncthreads();
}
inline void barrier_only() { emu::syncthreads(); }
// clock probes (MvConv1dDesc.clock_probe): the emulator has no clocks -- a monotonic count of calls (1 GHz "shader" against the 100 MHz reference)
inline unsigned long long shader_clock() { static thread_local unsigned long long t = 0; return t += 10; }
inline unsigned long long ref_clock_100mhz() { static thread_local unsigned long long t = 0; return t += 1; }
inline void glds16_untracked(const void* gsrc, unsigned lds_wave_base_addr) {
    emu::dma_issue(const_cast<char*>(lds_ptr(lds_wave_base_addr)) + (emu::flat_tid() & 63) * 16, gsrc);
}
inline void global_store8_untracked(void* gdst, const half4v& v) { *reinterpret_cast<half4v*>(gdst) = v; }
inline void glds16_untracked_so(const void* sbase, unsigned voff, unsigned lds_wave_base_addr) {
    emu::dma_issue(const_cast<char*>(lds_ptr(lds_wave_base_addr)) + (emu::flat_tid() & 63) * 16, reinterpret_cast<const char*>(sbase) + voff);
}
inline void lds_read1(half8v& d, unsigned addr) { emu::lds_read_issue(&d, lds_ptr(addr)); }
template <int N>
inline void lds_wait(half8v&, half8v&) { emu::lds_read_retire(N); }
template <int N>
inline void lds_wait(half8v&) { emu::lds_read_retire(N); }
template <int N>
inline void lds_wait(half8v&, half8v&, half8v&) { emu::lds_read_retire(N); }
template <int N>
inline void lds_wait(half8v&, half8v&, half8v&, half8v&) { emu::lds_read_retire(N); }
template <int OFF>
inline void lds_read1_off(half8v& d, unsigned addr) { emu::lds_read_issue(&d, lds_ptr(addr + OFF)); }

template <int WAIT>
inline void mfma8_step(float4v (&c0)[4], float4v (&c1)[4], const half8v& a0, const half8v& a1, const half8v (&b)[4]) {
    emu::lds_read_retire(WAIT);
    for (int i = 0; i < 4; ++i) c0[i] = emu_mfma_f32_16x16x32_f16(a0, b[i], c0[i]);
    for (int i = 0; i < 4; ++i) c1[i] = emu_mfma_f32_16x16x32_f16(a1, b[i], c1[i]);
}
template <int OFF0, int OFF1>
inline void lds_read2(half8v& d0, half8v& d1, unsigned addr) {
    emu::lds_read_issue(&d0, lds_ptr(addr + OFF0));
    emu::lds_read_issue(&d1, lds_ptr(addr + OFF1));
}
inline void lds_read4(half8v (&d)[4], unsigned addr) {
    for (int i = 0; i < 4; ++i) emu::lds_read_issue(&d[i], lds_ptr(addr + 2048 * i));
}
template <int OFF, int STRIDE>
inline void lds_read5(half8v (&d)[5], unsigned addr) {
    for (int i = 0; i < 5; ++i) emu::lds_read_issue(&d[i], lds_ptr(addr + OFF + i * STRIDE));
}
template <int WAIT>
inline void mfma10_step(float4v* c0, float4v* c1, const half8v& a0, const half8v& a1, const half8v (&b)[5]) {
    emu::lds_read_retire(WAIT);
    for (int i = 0; i < 5; ++i) {
        c0[i] = emu_mfma_f32_16x16x32_f16(a0, b[i], c0[i]);
        c1[i] = emu_mfma_f32_16x16x32_f16(a1, b[i], c1[i]);
    }
}
template <int N>
inline void lds_read_tiles(half8v (&d)[N], unsigned addr) {
    for (int i = 0; i < N; ++i) emu::lds_read_issue(&d[i], lds_ptr(addr + 1024 * i));
}
template <int N, int WAIT>
inline void mfma_tiles2(float4v (&c0)[N], float4v (&c1)[N], const half8v& a0, const half8v& a1, const half8v (&b)[N]) {
    emu::lds_read_retire(WAIT);
    for (int i = 0; i < N; ++i) {
        c0[i] = emu_mfma_f32_16x16x32_f16(a0, b[i], c0[i]);
        c1[i] = emu_mfma_f32_16x16x32_f16(a1, b[i], c1[i]);
    }
}
template <int N, int WAIT>
inline void mfma_tiles1(float4v (&c)[N], const half8v& a, const half8v (&b)[N]) {
    emu::lds_read_retire(WAIT);
    for (int i = 0; i < N; ++i) c[i] = emu_mfma_f32_16x16x32_f16(a, b[i], c[i]);
}
template <int N, int WAIT>
inline void mfma_tiles2_init(float4v (&c0)[N], float4v (&c1)[N], const half8v& a0, const half8v& a1, const half8v (&b)[N], const float4v& i0,
                             const float4v& i1) {
    emu::lds_read_retire(WAIT);
    for (int i = 0; i < N; ++i) {
        c0[i] = emu_mfma_f32_16x16x32_f16(a0, b[i], i0);
        c1[i] = emu_mfma_f32_16x16x32_f16(a1, b[i], i1);
    }
}
template <int N, int WAIT>
inline void mfma_tiles1_init(float4v (&c)[N], const half8v& a, const half8v (&b)[N], const float4v& init) {
    emu::lds_read_retire(WAIT);
    for (int i = 0; i < N; ++i) c[i] = emu_mfma_f32_16x16x32_f16(a, b[i], init);
}
inline void glds16_untracked_so_fresh(const void* sbase, unsigned voff, unsigned lds_wave_base_addr) { glds16_untracked_so(sbase, voff, lds_wave_base_addr); }
inline void mfma_hazard_pad() {}

// compute units of the emulated chip: 8 by default; MV_EMU_CUS = 256 makes every test batch a sub-chip one (the chunk / small-batch launch forms the
// real chip takes for a few utterances), MV_EMU_CUS = 1 a chip-filling one (persistent workgroups walking many tiles) -- the host side's plans follow it
inline int device_cu_count() {
    const char* e = getenv("MV_EMU_CUS");
    const int n = e != nullptr ? atoi(e) : 0;
    return n > 0 ? n : 8;
}
struct DeviceOnce { bool flag = false; };
inline bool device_once_pending(DeviceOnce& o, int* slot) { *slot = 0; return !o.flag; }
inline void device_once_done(DeviceOnce& o, int) { o.flag = true; }

}  // namespace mv
