// gfx950 (MI355X / CDNA4) spellings of everything in csrc/ that is not plain HIP C++: launch macros, DPP / permlane lane
// moves, LDS-DMA, counted waits, inline-assembly MFMA steps, address-space qualified accesses.  The kernels include this
// file as <arch/gfx950.h>; the test-suite's SIMT emulator (tests/emu) puts a header of the same name with host C++
// spellings of the same functions in front of it on the include path, so no kernel source carries an #ifdef.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
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
// three consecutive floats at any 4-byte aligned address as ONE global_load_dwordx3 (12 bytes: nothing behind p[2] is touched)
__device__ __forceinline__ float3u load_f32x3(const float* p) { return *reinterpret_cast<const float3u*>(p); }
typedef float float16v __attribute__((ext_vector_type(16)));

#define MV_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3 grid, dim3 block, (shmem), (stream), __VA_ARGS__)
#define MV_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
// dynamic LDS above 64 KiB has to be requested per kernel
#define MV_SET_MAX_SMEM(kernel, bytes) \
    hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))
// LDS traffic between lanes of ONE wave: DS ops of a wave execute in program order, so only the compiler has to be kept
// from reordering across this point.
#define MV_WAVE_FENCE()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
// Pointers with their address space stated: two branches that store the same values once to LDS and once to global memory
// are otherwise tail-merged into ONE flat store behind a selected base pointer; pointer selects between a tensor and the zero
// page lose the address space too (flat loads).
// point at which lanes of one wave hand data to each other through LDS without a barrier: the lanes of a wave run in lockstep, so
// there is nothing to do (the test-suite's SIMT emulator runs lanes as fibres and makes them meet here)
#define MV_LOCKSTEP_POINT() do { } while (0)
// "n compiler-tracked vector-memory loads of this wave were just issued, and a counted wait_vm<N>() further down includes them in its N":
// nothing to do on the device (the counter counts them by itself); the test-suite's emulator counts them beside the untracked transfers
#define MV_VM_LOADS(n) do { } while (0)
#define MV_AS_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))
#define MV_AS_GLOBAL(T, p) ((__attribute__((address_space(1))) T*)(p))
#define MV_GLOBAL_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))
// value the optimiser must treat as freshly computed here: keeps per-tile addresses from being hoisted out of a loop nest
// into dozens of long-lived registers
#define MV_OPAQUE(x) asm volatile("" : "+v"(x))
// a value that is the same in every lane of the wave, moved to a scalar register so that branches on it are scalar branches
#define MV_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// instruction-order hint for the machine scheduler: the next `n` instructions of class `mask` (0x008 MFMA, 0x100 DS read,
// 0x200 DS write, 0x020 VMEM read) form one group, groups are emitted in the order the hints are written
#define MV_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)

namespace mv {

// ---- lane moves ----------------------------------------------------------------------------------------------------------
// DPP moves inside rows of 16 lanes (VALU, no LDS traffic).  CTRL is the hardware dpp_ctrl value: 0x00..0xFF quad_perm,
// 0x100+n row_shl:n (lane i reads lane i+n), 0x110+n row_shr:n (lane i reads lane i-n), 0x120+n row_ror:n, 0x140 row_mirror,
// 0x141 row_half_mirror.  Lanes without a source lane keep `old`.  (Checked on the device: tools/dpp_probe.hip.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                                 0xf, 0xf, false));
}
// sums over the 16 lanes of a DPP row for EIGHT values at once, each result in every lane: four v_add_f32_dpp per value, interleaved so that no
// instruction reads a register the one in front of it wrote (a DPP read needs two wait states behind a VALU write).  Written out because the SLP
// vectoriser pairs the additions of neighbouring values into v_pk_add_f32, which has no DPP form: the reduction then costs a v_mov_b32_dpp AND half
// a v_pk_add_f32 per step.  Bit-identical to eight row16_sum() calls (the same additions in the same order).
#define MV_DPP_STEP8(CTRL)                                                                                                            \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void row16_sum8(float4v& a, float4v& b) {
    float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
    asm volatile("s_nop 1\n\t" MV_DPP_STEP8("quad_perm:[1,0,3,2]") MV_DPP_STEP8("quad_perm:[2,3,0,1]") MV_DPP_STEP8("row_half_mirror") MV_DPP_STEP8("row_mirror")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    a = float4v{a0, a1, a2, a3};
    b = float4v{b0, b1, b2, b3};
}
// for patterns in which every lane has a source lane (mirror, rotate, quad_perm): no `old` operand to initialise
template <int CTRL>
__device__ __forceinline__ float dpp_mov_all(float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
}
// v_permlane16_swap: the odd 16-lane rows of x trade places with the even rows of y
//   x: row1 <- y.row0, row3 <- y.row2      y: row0 <- x.row1, row2 <- x.row3      (tools/permlane_probe.hip)
__device__ __forceinline__ void row_swap_odd_even(unsigned& x, unsigned& y) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// ds_bpermute_b32: every lane reads `v` of lane byte_index / 4 (the LDS crossbar moves the data; LDS memory is not touched)
__device__ __forceinline__ float lane_gather(float v, int byte_index) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_index, __builtin_bit_cast(int, v)));
}

// ---- arithmetic with a single-instruction spelling ---------------------------------------------------------------------------
__device__ __forceinline__ float fmed3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }

// MODE.FP16_OVFL (bit 23 of the wave's mode register): an fp16 RESULT that overflows -- a conversion from fp32, a packed fp16 sum -- is clamped to
// +-65504 instead of becoming +-inf; true infinities and NaNs pass (tools/fp16_ovfl_probe.hip, profiles/r15ap: gfx950 honours it).  One s_setreg per
// wave then stands for the saturation the epilogues spelled out per value (v_med3_f32 against +-65504) and per pair (v_pk_min_f16 + v_pk_max_f16).
// A kernel that converts with half_hwsat / adds with pk_add_hwsat calls fp16_saturation_on() first -- every wave, before any fp16 arithmetic.
__device__ __forceinline__ void fp16_saturation_on() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 4" ::: "memory"); }
__device__ __forceinline__ half_t half_hwsat(float v) { return (half_t)v; }
template <class V>
__device__ __forceinline__ V pk_add_hwsat(V a, V b) { return a + b; }
// max(v, lo) as exactly one v_max_f32: fmaxf / v_med3 against +inf are lowered to a canonicalising v_max v, v, v plus the
// max itself, which doubled the activation cost of the 128-value epilogues
__device__ __forceinline__ float max_raw(float v, float lo) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(lo));
    return r;
}
__device__ __forceinline__ float rcp_fast(float v) { return __builtin_amdgcn_rcpf(v); }    // v_rcp_f32, 1 ulp
__device__ __forceinline__ float sqrt_fast(float v) { return __builtin_amdgcn_sqrtf(v); }  // v_sqrt_f32, 1 ulp
__device__ __forceinline__ float exp2_fast(float v) { return __builtin_amdgcn_exp2f(v); }  // v_exp_f32
__device__ __forceinline__ float log2_fast(float v) { return __builtin_amdgcn_logf(v); }   // v_log_f32, normal inputs only
// v_mfma_f32_4x4x1: 16 independent 4 x 4 outer products; D[lane][r] += A[4 * (lane / 4) + r] * B[lane]
__device__ __forceinline__ float4v mfma_4x4x1(float a, float b, float4v c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// ---- LDS -------------------------------------------------------------------------------------------------------------------
// 8-byte LDS load that the load/store merger leaves alone (volatile, with the LDS address space stated: a volatile access
// through a generic pointer would become a flat load); pairs would be merged into ds_read2_b64, which moves 128 B per LDS
// clock where ds_read_b64 moves 256
__device__ __forceinline__ float2v lds_load_unmerged(const float2v* p) { return *(const volatile __attribute__((address_space(3))) float2v*)(p); }
// byte address inside the workgroup's LDS allocation (what ds_* instructions take)
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
// opaque LDS pointer to fp16 fragments: keeps the compiler from parking all fragments in registers, stays LDS-qualified (an
// opaque GENERIC pointer would turn the reads into flat loads)
typedef const __attribute__((address_space(3))) half_t* lds_half_ptr;
__device__ __forceinline__ lds_half_ptr lds_opaque_half_ptr(const half_t* p) {
    lds_half_ptr q = (lds_half_ptr)p;
    asm volatile("" : "+v"(q));
    return q;
}
__device__ __forceinline__ half8v lds_load_half8(lds_half_ptr p, int elem_off) { return *(const __attribute__((address_space(3))) half8v*)(p + elem_off); }
// One 16-byte global -> LDS transfer per lane: the wave writes 1 KiB at lds_wave_base + lane * 16, the global address is per
// lane (swizzles are applied to the SOURCE address).  No VGPR staging.
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base,
                                     16, 0, 0);
}
// wait until at most N of this wave's vector-memory operations are outstanding (the N youngest may stay in flight)
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// The same wait as an instruction the compiler's wait-count insertion SEES: its scoreboard of tracked loads moves on.  (Where tracked loads stay in
// flight across a wait the compiler cannot see, it protects every register one of them might still write with an s_waitcnt vmcnt(0) of its own.)
template <int N>
__device__ __forceinline__ void wait_vm_seen() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));   // gfx9 encoding: vmcnt [3:0] + [15:14], expcnt and lgkmcnt not waited for
}
// The same transfer as inline assembly, invisible to the compiler's wait-count insertion.  With the builtin the compiler knows that
// LDS is written behind its back: it puts s_waitcnt vmcnt(0) in front of every LDS load that may alias a transfer in flight and in
// front of the first use of any global load that shares the counter with one (mixed event types are assumed to complete out of
// order) -- which drains exactly the transfers a ring is meant to keep in flight.  The caller owns the ordering: the data is in LDS
// once a later s_waitcnt vmcnt (the caller's, or the compiler's for a younger tracked load) has retired the transfer.
__device__ __forceinline__ void glds16_untracked(const void* gsrc, unsigned lds_wave_base_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_wave_base_addr) : "memory", "m0");
}
// an 8-byte global store the compiler's wait-count insertion does not see: a tracked store in flight beside tracked loads makes it treat the counter
// as completing out of order (mixed event types), and every wait it then needs becomes s_waitcnt vmcnt(0).  The caller owns the ordering.
__device__ __forceinline__ void global_store8_untracked(void* gdst, const half4v& v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(gdst), "v"(v) : "memory");
}
// the same with the address as uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset: a stream that advances by a
// constant per stage costs one scalar add per stage instead of a 64-bit vector add per transfer
__device__ __forceinline__ void glds16_untracked_so(const void* sbase, unsigned voff, unsigned lds_wave_base_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_wave_base_addr) : "memory", "m0");
}
// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain transfers still in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// counters for in-kernel clock probes (MvConv1dDesc.clock_probe): shader-clock cycles and the 100 MHz constant reference; both scalar
__device__ __forceinline__ unsigned long long shader_clock() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned long long ref_clock_100mhz() { return __builtin_amdgcn_s_memrealtime(); }

// bare workgroup barrier: no wait for this wave's outstanding LDS reads (they may stay in flight across it when nobody writes what they
// read); everything else a barrier has to order is the caller's business
__device__ __forceinline__ void barrier_only() { asm volatile("s_barrier" ::: "memory"); }

// ---- hand-scheduled pieces of the 256 x 256 GEMM stage (conv1d.hip) ---------------------------------------------------------------
#define MV_MFMA8(A0, A1)                                                                                       \
    "v_mfma_f32_16x16x32_f16 %0, " A0 ", %10, %0\n\tv_mfma_f32_16x16x32_f16 %1, " A0 ", %11, %1\n\t"          \
    "v_mfma_f32_16x16x32_f16 %2, " A0 ", %12, %2\n\tv_mfma_f32_16x16x32_f16 %3, " A0 ", %13, %3\n\t"          \
    "v_mfma_f32_16x16x32_f16 %4, " A1 ", %10, %4\n\tv_mfma_f32_16x16x32_f16 %5, " A1 ", %11, %5\n\t"          \
    "v_mfma_f32_16x16x32_f16 %6, " A1 ", %12, %6\n\tv_mfma_f32_16x16x32_f16 %7, " A1 ", %13, %7"
// wait until at most WAIT LDS reads are outstanding, then 8 MFMAs: (a0 | a1) x b[0..3] into c0[0..3] | c1[0..3]
template <int WAIT>
__device__ __forceinline__ void mfma8_step(float4v (&c0)[4], float4v (&c1)[4], const half8v& a0, const half8v& a1, const half8v (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%14)\n\t" MV_MFMA8("%8", "%9")
                 : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3])
                 : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "n"(WAIT)
                 : "memory");
}
#undef MV_MFMA8
// two / four 16-byte fragment reads from LDS byte address `addr` (+ immediate offsets), no wait
template <int OFF0, int OFF1>
__device__ __forceinline__ void lds_read2(half8v& d0, half8v& d1, unsigned addr) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(d0), "=&v"(d1) : "v"(addr), "n"(OFF0), "n"(OFF1) : "memory");
}
__device__ __forceinline__ void lds_read4(half8v (&d)[4], unsigned addr) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:4096\n\tds_read_b128 %3, %4 offset:6144"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                 : "v"(addr)
                 : "memory");
}
// one 16-byte fragment read the compiler cannot see through (it would put s_waitcnt vmcnt(0) in front of a plain LDS load that may
// alias an LDS-DMA transfer still in flight -- also the ones that are meant to stay in flight); pair with lds_wait before use
__device__ __forceinline__ void lds_read1(half8v& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=&v"(d) : "v"(addr) : "memory"); }
// wait until at most N LDS operations of this wave are outstanding; a and b are the registers the reads above deliver
template <int N>
__device__ __forceinline__ void lds_wait(half8v& a, half8v& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(half8v& a, half8v& b, half8v& c) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(half8v& a, half8v& b, half8v& c, half8v& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(half8v& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read1_off(half8v& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF) : "memory"); }
// ---- hand-scheduled pieces of the Res2Net chain stage (res2.hip): 2 channel tiles x 5 time tiles per step ----------------------------
// five 16-byte fragment reads at addr + OFF + i * STRIDE, no wait
template <int OFF, int STRIDE>
__device__ __forceinline__ void lds_read5(half8v (&d)[5], unsigned addr) {
    asm volatile("ds_read_b128 %0, %5 offset:%6\n\tds_read_b128 %1, %5 offset:%7\n\tds_read_b128 %2, %5 offset:%8\n\t"
                 "ds_read_b128 %3, %5 offset:%9\n\tds_read_b128 %4, %5 offset:%10"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4])
                 : "v"(addr), "n"(OFF), "n"(OFF + STRIDE), "n"(OFF + 2 * STRIDE), "n"(OFF + 3 * STRIDE), "n"(OFF + 4 * STRIDE)
                 : "memory");
}
// wait until at most WAIT LDS reads are outstanding, then 10 MFMAs: (a0 | a1) x b[0..4] into c0[G..G+4] | c1[G..G+4]
template <int WAIT>
__device__ __forceinline__ void mfma10_step(float4v* c0, float4v* c1, const half8v& a0, const half8v& a1, const half8v (&b)[5]) {
    asm volatile("s_waitcnt lgkmcnt(%17)\n\t"
                 "v_mfma_f32_16x16x32_f16 %0, %10, %12, %0\n\tv_mfma_f32_16x16x32_f16 %5, %11, %12, %5\n\t"
                 "v_mfma_f32_16x16x32_f16 %1, %10, %13, %1\n\tv_mfma_f32_16x16x32_f16 %6, %11, %13, %6\n\t"
                 "v_mfma_f32_16x16x32_f16 %2, %10, %14, %2\n\tv_mfma_f32_16x16x32_f16 %7, %11, %14, %7\n\t"
                 "v_mfma_f32_16x16x32_f16 %3, %10, %15, %3\n\tv_mfma_f32_16x16x32_f16 %8, %11, %15, %8\n\t"
                 "v_mfma_f32_16x16x32_f16 %4, %10, %16, %4\n\tv_mfma_f32_16x16x32_f16 %9, %11, %16, %9"
                 : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c0[4]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]),
                   "+v"(c1[4])
                 : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "n"(WAIT)
                 : "memory");
}
// MFMA results are read by VALU code only after a barrier and a round of transfers; pad the hazard anyway
__device__ __forceinline__ void mfma_hazard_pad() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }
// ---- hand-scheduled pieces of the fused FCM block kernel (fcmblock.hip) ----------------------------------------------------------
// N 16-byte fragment reads at addr + i * 1024 (16 positions x 64 B apart), no wait
template <int N>
__device__ __forceinline__ void lds_read_tiles(half8v (&d)[N], unsigned addr) {
    static_assert(N >= 1 && N <= 10, "lds_read_tiles: 1..10 tiles");
    if constexpr (N == 1)
        asm volatile("ds_read_b128 %0, %1"
                     : "=&v"(d[0])
                     : "v"(addr)
                     : "memory");
    if constexpr (N == 2)
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024"
                     : "=&v"(d[0]), "=&v"(d[1])
                     : "v"(addr)
                     : "memory");
    if constexpr (N == 3)
        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2])
                     : "v"(addr)
                     : "memory");
    if constexpr (N == 4)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                     : "v"(addr)
                     : "memory");
    if constexpr (N == 5)
        asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:1024\n\tds_read_b128 %2, %5 offset:2048\n\tds_read_b128 %3, %5 offset:3072\n\tds_read_b128 %4, %5 offset:4096"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4])
                     : "v"(addr)
                     : "memory");
    if constexpr (N == 6)
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %6 offset:2048\n\tds_read_b128 %3, %6 offset:3072\n\tds_read_b128 %4, %6 offset:4096\n\tds_read_b128 %5, %6 offset:5120"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
                     : "v"(addr)
                     : "memory");
    if constexpr (N == 7)
        asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:1024\n\tds_read_b128 %2, %7 offset:2048\n\tds_read_b128 %3, %7 offset:3072\n\tds_read_b128 %4, %7 offset:4096\n\tds_read_b128 %5, %7 offset:5120\n\tds_read_b128 %6, %7 offset:6144"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6])
                     : "v"(addr)
                     : "memory");
    if constexpr (N == 8)
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
                     : "v"(addr)
                     : "memory");
    if constexpr (N == 9)
        asm volatile("ds_read_b128 %0, %9\n\tds_read_b128 %1, %9 offset:1024\n\tds_read_b128 %2, %9 offset:2048\n\tds_read_b128 %3, %9 offset:3072\n\tds_read_b128 %4, %9 offset:4096\n\tds_read_b128 %5, %9 offset:5120\n\tds_read_b128 %6, %9 offset:6144\n\tds_read_b128 %7, %9 offset:7168\n\tds_read_b128 %8, %9 offset:8192"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(d[8])
                     : "v"(addr)
                     : "memory");
    if constexpr (N == 10)
        asm volatile("ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:1024\n\tds_read_b128 %2, %10 offset:2048\n\tds_read_b128 %3, %10 offset:3072\n\tds_read_b128 %4, %10 offset:4096\n\tds_read_b128 %5, %10 offset:5120\n\tds_read_b128 %6, %10 offset:6144\n\tds_read_b128 %7, %10 offset:7168\n\tds_read_b128 %8, %10 offset:8192\n\tds_read_b128 %9, %10 offset:9216"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9])
                     : "v"(addr)
                     : "memory");
}
// wait until at most WAIT LDS operations are outstanding, then 2 * N MFMAs: (a0 | a1) x b[0..N) into c0[] | c1[].  A following block
// accumulates into the same registers 2 * N MFMAs later: the short blocks pad that distance with s_nop (the compiler's hazard
// recogniser does not look into inline assembly).
template <int N, int WAIT>
__device__ __forceinline__ void mfma_tiles2(float4v (&c0)[N], float4v (&c1)[N], const half8v& a0, const half8v& a1, const half8v (&b)[N]) {
    static_assert(N >= 1 && N <= 5, "mfma_tiles2: 1..5 tiles");
    if constexpr (N == 1)
        asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_16x16x32_f16 %0, %2, %4, %0\n\tv_mfma_f32_16x16x32_f16 %1, %3, %4, %1\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3"
                     : "+v"(c0[0]), "+v"(c1[0])
                     : "v"(a0), "v"(a1), "v"(b[0]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 2)
        asm volatile("s_waitcnt lgkmcnt(%8)\n\tv_mfma_f32_16x16x32_f16 %0, %4, %6, %0\n\tv_mfma_f32_16x16x32_f16 %2, %5, %6, %2\n\tv_mfma_f32_16x16x32_f16 %1, %4, %7, %1\n\tv_mfma_f32_16x16x32_f16 %3, %5, %7, %3\n\ts_nop 7\n\ts_nop 3"
                     : "+v"(c0[0]), "+v"(c0[1]), "+v"(c1[0]), "+v"(c1[1])
                     : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 3)
        asm volatile("s_waitcnt lgkmcnt(%11)\n\tv_mfma_f32_16x16x32_f16 %0, %6, %8, %0\n\tv_mfma_f32_16x16x32_f16 %3, %7, %8, %3\n\tv_mfma_f32_16x16x32_f16 %1, %6, %9, %1\n\tv_mfma_f32_16x16x32_f16 %4, %7, %9, %4\n\tv_mfma_f32_16x16x32_f16 %2, %6, %10, %2\n\tv_mfma_f32_16x16x32_f16 %5, %7, %10, %5"
                     : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2])
                     : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(b[2]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%14)\n\tv_mfma_f32_16x16x32_f16 %0, %8, %10, %0\n\tv_mfma_f32_16x16x32_f16 %4, %9, %10, %4\n\tv_mfma_f32_16x16x32_f16 %1, %8, %11, %1\n\tv_mfma_f32_16x16x32_f16 %5, %9, %11, %5\n\tv_mfma_f32_16x16x32_f16 %2, %8, %12, %2\n\tv_mfma_f32_16x16x32_f16 %6, %9, %12, %6\n\tv_mfma_f32_16x16x32_f16 %3, %8, %13, %3\n\tv_mfma_f32_16x16x32_f16 %7, %9, %13, %7"
                     : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3])
                     : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 5)
        asm volatile("s_waitcnt lgkmcnt(%17)\n\tv_mfma_f32_16x16x32_f16 %0, %10, %12, %0\n\tv_mfma_f32_16x16x32_f16 %5, %11, %12, %5\n\tv_mfma_f32_16x16x32_f16 %1, %10, %13, %1\n\tv_mfma_f32_16x16x32_f16 %6, %11, %13, %6\n\tv_mfma_f32_16x16x32_f16 %2, %10, %14, %2\n\tv_mfma_f32_16x16x32_f16 %7, %11, %14, %7\n\tv_mfma_f32_16x16x32_f16 %3, %10, %15, %3\n\tv_mfma_f32_16x16x32_f16 %8, %11, %15, %8\n\tv_mfma_f32_16x16x32_f16 %4, %10, %16, %4\n\tv_mfma_f32_16x16x32_f16 %9, %11, %16, %9"
                     : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c0[4]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]), "+v"(c1[4])
                     : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "n"(WAIT)
                     : "memory");
}
// the same with one map tile: N MFMAs a x b[0..N) into c[]
template <int N, int WAIT>
__device__ __forceinline__ void mfma_tiles1(float4v (&c)[N], const half8v& a, const half8v (&b)[N]) {
    static_assert(N >= 1 && N <= 10, "mfma_tiles1: 1..10 tiles");
    if constexpr (N == 1)
        asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3"
                     : "+v"(c[0])
                     : "v"(a), "v"(b[0]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 2)
        asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %4, %1\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3"
                     : "+v"(c[0]), "+v"(c[1])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 3)
        asm volatile("s_waitcnt lgkmcnt(%7)\n\tv_mfma_f32_16x16x32_f16 %0, %3, %4, %0\n\tv_mfma_f32_16x16x32_f16 %1, %3, %5, %1\n\tv_mfma_f32_16x16x32_f16 %2, %3, %6, %2\n\ts_nop 7\n\ts_nop 3"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%9)\n\tv_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %6, %1\n\tv_mfma_f32_16x16x32_f16 %2, %4, %7, %2\n\tv_mfma_f32_16x16x32_f16 %3, %4, %8, %3\n\ts_nop 7\n\ts_nop 3"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 5)
        asm volatile("s_waitcnt lgkmcnt(%11)\n\tv_mfma_f32_16x16x32_f16 %0, %5, %6, %0\n\tv_mfma_f32_16x16x32_f16 %1, %5, %7, %1\n\tv_mfma_f32_16x16x32_f16 %2, %5, %8, %2\n\tv_mfma_f32_16x16x32_f16 %3, %5, %9, %3\n\tv_mfma_f32_16x16x32_f16 %4, %5, %10, %4"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 6)
        asm volatile("s_waitcnt lgkmcnt(%13)\n\tv_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n\tv_mfma_f32_16x16x32_f16 %1, %6, %8, %1\n\tv_mfma_f32_16x16x32_f16 %2, %6, %9, %2\n\tv_mfma_f32_16x16x32_f16 %3, %6, %10, %3\n\tv_mfma_f32_16x16x32_f16 %4, %6, %11, %4\n\tv_mfma_f32_16x16x32_f16 %5, %6, %12, %5"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 7)
        asm volatile("s_waitcnt lgkmcnt(%15)\n\tv_mfma_f32_16x16x32_f16 %0, %7, %8, %0\n\tv_mfma_f32_16x16x32_f16 %1, %7, %9, %1\n\tv_mfma_f32_16x16x32_f16 %2, %7, %10, %2\n\tv_mfma_f32_16x16x32_f16 %3, %7, %11, %3\n\tv_mfma_f32_16x16x32_f16 %4, %7, %12, %4\n\tv_mfma_f32_16x16x32_f16 %5, %7, %13, %5\n\tv_mfma_f32_16x16x32_f16 %6, %7, %14, %6"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 8)
        asm volatile("s_waitcnt lgkmcnt(%17)\n\tv_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n\tv_mfma_f32_16x16x32_f16 %1, %8, %10, %1\n\tv_mfma_f32_16x16x32_f16 %2, %8, %11, %2\n\tv_mfma_f32_16x16x32_f16 %3, %8, %12, %3\n\tv_mfma_f32_16x16x32_f16 %4, %8, %13, %4\n\tv_mfma_f32_16x16x32_f16 %5, %8, %14, %5\n\tv_mfma_f32_16x16x32_f16 %6, %8, %15, %6\n\tv_mfma_f32_16x16x32_f16 %7, %8, %16, %7"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 9)
        asm volatile("s_waitcnt lgkmcnt(%19)\n\tv_mfma_f32_16x16x32_f16 %0, %9, %10, %0\n\tv_mfma_f32_16x16x32_f16 %1, %9, %11, %1\n\tv_mfma_f32_16x16x32_f16 %2, %9, %12, %2\n\tv_mfma_f32_16x16x32_f16 %3, %9, %13, %3\n\tv_mfma_f32_16x16x32_f16 %4, %9, %14, %4\n\tv_mfma_f32_16x16x32_f16 %5, %9, %15, %5\n\tv_mfma_f32_16x16x32_f16 %6, %9, %16, %6\n\tv_mfma_f32_16x16x32_f16 %7, %9, %17, %7\n\tv_mfma_f32_16x16x32_f16 %8, %9, %18, %8"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "n"(WAIT)
                     : "memory");
    if constexpr (N == 10)
        asm volatile("s_waitcnt lgkmcnt(%21)\n\tv_mfma_f32_16x16x32_f16 %0, %10, %11, %0\n\tv_mfma_f32_16x16x32_f16 %1, %10, %12, %1\n\tv_mfma_f32_16x16x32_f16 %2, %10, %13, %2\n\tv_mfma_f32_16x16x32_f16 %3, %10, %14, %3\n\tv_mfma_f32_16x16x32_f16 %4, %10, %15, %4\n\tv_mfma_f32_16x16x32_f16 %5, %10, %16, %5\n\tv_mfma_f32_16x16x32_f16 %6, %10, %17, %6\n\tv_mfma_f32_16x16x32_f16 %7, %10, %18, %7\n\tv_mfma_f32_16x16x32_f16 %8, %10, %19, %8\n\tv_mfma_f32_16x16x32_f16 %9, %10, %20, %9"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "n"(WAIT)
                     : "memory");
}

// first tap of an accumulation: c0[] | c1[] = (a0 | a1) x b[] + (i0 | i1) -- the initial value (a bias) is the MFMA's C operand, so the
// accumulators need no initialisation instructions
template <int N, int WAIT>
__device__ __forceinline__ void mfma_tiles2_init(float4v (&c0)[N], float4v (&c1)[N], const half8v& a0, const half8v& a1, const half8v (&b)[N],
                                                 const float4v& i0, const float4v& i1) {
    static_assert(N >= 1 && N <= 5, "mfma_tiles2_init: 1..5 tiles");
    if constexpr (N == 1)
        asm volatile("s_waitcnt lgkmcnt(%7)\n\tv_mfma_f32_16x16x32_f16 %0, %2, %4, %5\n\tv_mfma_f32_16x16x32_f16 %1, %3, %4, %6\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3"
                     : "=&v"(c0[0]), "=&v"(c1[0])
                     : "v"(a0), "v"(a1), "v"(b[0]), "v"(i0), "v"(i1), "n"(WAIT)
                     : "memory");
    if constexpr (N == 2)
        asm volatile("s_waitcnt lgkmcnt(%10)\n\tv_mfma_f32_16x16x32_f16 %0, %4, %6, %8\n\tv_mfma_f32_16x16x32_f16 %2, %5, %6, %9\n\tv_mfma_f32_16x16x32_f16 %1, %4, %7, %8\n\tv_mfma_f32_16x16x32_f16 %3, %5, %7, %9\n\ts_nop 7\n\ts_nop 3"
                     : "=&v"(c0[0]), "=&v"(c0[1]), "=&v"(c1[0]), "=&v"(c1[1])
                     : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(i0), "v"(i1), "n"(WAIT)
                     : "memory");
    if constexpr (N == 3)
        asm volatile("s_waitcnt lgkmcnt(%13)\n\tv_mfma_f32_16x16x32_f16 %0, %6, %8, %11\n\tv_mfma_f32_16x16x32_f16 %3, %7, %8, %12\n\tv_mfma_f32_16x16x32_f16 %1, %6, %9, %11\n\tv_mfma_f32_16x16x32_f16 %4, %7, %9, %12\n\tv_mfma_f32_16x16x32_f16 %2, %6, %10, %11\n\tv_mfma_f32_16x16x32_f16 %5, %7, %10, %12"
                     : "=&v"(c0[0]), "=&v"(c0[1]), "=&v"(c0[2]), "=&v"(c1[0]), "=&v"(c1[1]), "=&v"(c1[2])
                     : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(i0), "v"(i1), "n"(WAIT)
                     : "memory");
    if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%16)\n\tv_mfma_f32_16x16x32_f16 %0, %8, %10, %14\n\tv_mfma_f32_16x16x32_f16 %4, %9, %10, %15\n\tv_mfma_f32_16x16x32_f16 %1, %8, %11, %14\n\tv_mfma_f32_16x16x32_f16 %5, %9, %11, %15\n\tv_mfma_f32_16x16x32_f16 %2, %8, %12, %14\n\tv_mfma_f32_16x16x32_f16 %6, %9, %12, %15\n\tv_mfma_f32_16x16x32_f16 %3, %8, %13, %14\n\tv_mfma_f32_16x16x32_f16 %7, %9, %13, %15"
                     : "=&v"(c0[0]), "=&v"(c0[1]), "=&v"(c0[2]), "=&v"(c0[3]), "=&v"(c1[0]), "=&v"(c1[1]), "=&v"(c1[2]), "=&v"(c1[3])
                     : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(i0), "v"(i1), "n"(WAIT)
                     : "memory");
    if constexpr (N == 5)
        asm volatile("s_waitcnt lgkmcnt(%19)\n\tv_mfma_f32_16x16x32_f16 %0, %10, %12, %17\n\tv_mfma_f32_16x16x32_f16 %5, %11, %12, %18\n\tv_mfma_f32_16x16x32_f16 %1, %10, %13, %17\n\tv_mfma_f32_16x16x32_f16 %6, %11, %13, %18\n\tv_mfma_f32_16x16x32_f16 %2, %10, %14, %17\n\tv_mfma_f32_16x16x32_f16 %7, %11, %14, %18\n\tv_mfma_f32_16x16x32_f16 %3, %10, %15, %17\n\tv_mfma_f32_16x16x32_f16 %8, %11, %15, %18\n\tv_mfma_f32_16x16x32_f16 %4, %10, %16, %17\n\tv_mfma_f32_16x16x32_f16 %9, %11, %16, %18"
                     : "=&v"(c0[0]), "=&v"(c0[1]), "=&v"(c0[2]), "=&v"(c0[3]), "=&v"(c0[4]), "=&v"(c1[0]), "=&v"(c1[1]), "=&v"(c1[2]), "=&v"(c1[3]), "=&v"(c1[4])
                     : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(i0), "v"(i1), "n"(WAIT)
                     : "memory");
}
template <int N, int WAIT>
__device__ __forceinline__ void mfma_tiles1_init(float4v (&c)[N], const half8v& a, const half8v (&b)[N], const float4v& init) {
    static_assert(N >= 1 && N <= 10, "mfma_tiles1_init: 1..10 tiles");
    if constexpr (N == 1)
        asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %3\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3"
                     : "=&v"(c[0])
                     : "v"(a), "v"(b[0]), "v"(init), "n"(WAIT)
                     : "memory");
    if constexpr (N == 2)
        asm volatile("s_waitcnt lgkmcnt(%6)\n\tv_mfma_f32_16x16x32_f16 %0, %2, %3, %5\n\tv_mfma_f32_16x16x32_f16 %1, %2, %4, %5\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3"
                     : "=&v"(c[0]), "=&v"(c[1])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(init), "n"(WAIT)
                     : "memory");
    if constexpr (N == 3)
        asm volatile("s_waitcnt lgkmcnt(%8)\n\tv_mfma_f32_16x16x32_f16 %0, %3, %4, %7\n\tv_mfma_f32_16x16x32_f16 %1, %3, %5, %7\n\tv_mfma_f32_16x16x32_f16 %2, %3, %6, %7\n\ts_nop 7\n\ts_nop 3"
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(init), "n"(WAIT)
                     : "memory");
    if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%10)\n\tv_mfma_f32_16x16x32_f16 %0, %4, %5, %9\n\tv_mfma_f32_16x16x32_f16 %1, %4, %6, %9\n\tv_mfma_f32_16x16x32_f16 %2, %4, %7, %9\n\tv_mfma_f32_16x16x32_f16 %3, %4, %8, %9\n\ts_nop 7\n\ts_nop 3"
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(init), "n"(WAIT)
                     : "memory");
    if constexpr (N == 5)
        asm volatile("s_waitcnt lgkmcnt(%12)\n\tv_mfma_f32_16x16x32_f16 %0, %5, %6, %11\n\tv_mfma_f32_16x16x32_f16 %1, %5, %7, %11\n\tv_mfma_f32_16x16x32_f16 %2, %5, %8, %11\n\tv_mfma_f32_16x16x32_f16 %3, %5, %9, %11\n\tv_mfma_f32_16x16x32_f16 %4, %5, %10, %11"
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(init), "n"(WAIT)
                     : "memory");
    if constexpr (N == 6)
        asm volatile("s_waitcnt lgkmcnt(%14)\n\tv_mfma_f32_16x16x32_f16 %0, %6, %7, %13\n\tv_mfma_f32_16x16x32_f16 %1, %6, %8, %13\n\tv_mfma_f32_16x16x32_f16 %2, %6, %9, %13\n\tv_mfma_f32_16x16x32_f16 %3, %6, %10, %13\n\tv_mfma_f32_16x16x32_f16 %4, %6, %11, %13\n\tv_mfma_f32_16x16x32_f16 %5, %6, %12, %13"
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(init), "n"(WAIT)
                     : "memory");
    if constexpr (N == 7)
        asm volatile("s_waitcnt lgkmcnt(%16)\n\tv_mfma_f32_16x16x32_f16 %0, %7, %8, %15\n\tv_mfma_f32_16x16x32_f16 %1, %7, %9, %15\n\tv_mfma_f32_16x16x32_f16 %2, %7, %10, %15\n\tv_mfma_f32_16x16x32_f16 %3, %7, %11, %15\n\tv_mfma_f32_16x16x32_f16 %4, %7, %12, %15\n\tv_mfma_f32_16x16x32_f16 %5, %7, %13, %15\n\tv_mfma_f32_16x16x32_f16 %6, %7, %14, %15"
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(init), "n"(WAIT)
                     : "memory");
    if constexpr (N == 8)
        asm volatile("s_waitcnt lgkmcnt(%18)\n\tv_mfma_f32_16x16x32_f16 %0, %8, %9, %17\n\tv_mfma_f32_16x16x32_f16 %1, %8, %10, %17\n\tv_mfma_f32_16x16x32_f16 %2, %8, %11, %17\n\tv_mfma_f32_16x16x32_f16 %3, %8, %12, %17\n\tv_mfma_f32_16x16x32_f16 %4, %8, %13, %17\n\tv_mfma_f32_16x16x32_f16 %5, %8, %14, %17\n\tv_mfma_f32_16x16x32_f16 %6, %8, %15, %17\n\tv_mfma_f32_16x16x32_f16 %7, %8, %16, %17"
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(init), "n"(WAIT)
                     : "memory");
    if constexpr (N == 9)
        asm volatile("s_waitcnt lgkmcnt(%20)\n\tv_mfma_f32_16x16x32_f16 %0, %9, %10, %19\n\tv_mfma_f32_16x16x32_f16 %1, %9, %11, %19\n\tv_mfma_f32_16x16x32_f16 %2, %9, %12, %19\n\tv_mfma_f32_16x16x32_f16 %3, %9, %13, %19\n\tv_mfma_f32_16x16x32_f16 %4, %9, %14, %19\n\tv_mfma_f32_16x16x32_f16 %5, %9, %15, %19\n\tv_mfma_f32_16x16x32_f16 %6, %9, %16, %19\n\tv_mfma_f32_16x16x32_f16 %7, %9, %17, %19\n\tv_mfma_f32_16x16x32_f16 %8, %9, %18, %19"
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7]), "=&v"(c[8])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(init), "n"(WAIT)
                     : "memory");
    if constexpr (N == 10)
        asm volatile("s_waitcnt lgkmcnt(%22)\n\tv_mfma_f32_16x16x32_f16 %0, %10, %11, %21\n\tv_mfma_f32_16x16x32_f16 %1, %10, %12, %21\n\tv_mfma_f32_16x16x32_f16 %2, %10, %13, %21\n\tv_mfma_f32_16x16x32_f16 %3, %10, %14, %21\n\tv_mfma_f32_16x16x32_f16 %4, %10, %15, %21\n\tv_mfma_f32_16x16x32_f16 %5, %10, %16, %21\n\tv_mfma_f32_16x16x32_f16 %6, %10, %17, %21\n\tv_mfma_f32_16x16x32_f16 %7, %10, %18, %21\n\tv_mfma_f32_16x16x32_f16 %8, %10, %19, %21\n\tv_mfma_f32_16x16x32_f16 %9, %10, %20, %21"
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7]), "=&v"(c[8]), "=&v"(c[9])
                     : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(init), "n"(WAIT)
                     : "memory");
}
// glds16_untracked_so for a base the scalar ALU has just computed: a vector-memory instruction may read a scalar register only five
// wait states after its write (s_nop 4), and m0 one state after its own (s_nop 0 inside)
__device__ __forceinline__ void glds16_untracked_so_fresh(const void* sbase, unsigned voff, unsigned lds_wave_base_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_wave_base_addr) : "memory", "m0");
}

// index of the current device, clamped to the per-device tables below
constexpr int MV_MAX_DEVICES = 64;
inline int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < MV_MAX_DEVICES ? dev : MV_MAX_DEVICES - 1;
}

// compute units of the current device (persistent kernels launch one workgroup per CU); cached per device -- one process may drive several GPUs
inline int device_cu_count() {
    static std::atomic<int> cache[MV_MAX_DEVICES];
    const int slot = current_device_slot();
    int n = cache[slot].load(std::memory_order_relaxed);
    if (n <= 0) {
        int dev = 0, v = 0;
        n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        cache[slot].store(n, std::memory_order_relaxed);
    }
    return n;
}

// "done once per device" flag for per-device function attributes (hipFuncSetAttribute of the dynamic LDS size applies to the current device's
// code object): `static DeviceOnce once; if (once.first()) { ...set attributes...; once.done(); }`.  Two threads racing through the
// first launch both set the attribute -- harmless.
struct DeviceOnce {
    std::atomic<bool> flag[MV_MAX_DEVICES];
    int slot = 0;
    DeviceOnce() {
        for (auto& f : flag) f.store(false, std::memory_order_relaxed);
    }
};
inline bool device_once_pending(DeviceOnce& o, int* slot) {
    *slot = current_device_slot();
    return !o.flag[*slot].load(std::memory_order_acquire);
}
inline void device_once_done(DeviceOnce& o, int slot) { o.flag[slot].store(true, std::memory_order_release); }

}  // namespace mv
