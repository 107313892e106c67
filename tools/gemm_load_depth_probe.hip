// Load-path probe 2 for the 256 x 256 implicit-GEMM tile: the K loop's global->LDS traffic only (same workgroup -> tile map as
// conv1d_glds_persistent_kernel), with DEPTH stages of 64 KiB in flight per workgroup (the LDS destinations alias: data is
// irrelevant here) -- does the path scale with bytes in flight (latency bound) or not (throughput bound)?
//   HALF = 0: 64-wide stages (rows of 128 B, 8 transfers per wave and stage)
//   HALF = 1: 32-wide stages (rows of 64 B, 4 transfers per wave and stage), DEPTH counted in 32 KiB half stages
// hipcc --offload-arch=gfx950 -O3 -o tools/probe/gemm_load_depth tools/gemm_load_depth_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int DEPTH, int HALF>
__global__ __launch_bounds__(512) void probe(const char* x, const char* w, int n_tiles, int co_tiles, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int BK = HALF ? 32 : 64, NT = HALF ? 2 : 4;   // transfers per operand, wave and stage
    const int nstages = K / BK;
    const int total = ((n_tiles + 7) >> 3) * 8 * co_tiles;
    const int lrow = HALF ? lane >> 2 : lane >> 3, kc = HALF ? lane & 3 : lane & 7;
    constexpr int ROWS_PER = HALF ? 16 : 8;
    int slot = 0;
    for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
        const int xcd = vb & 7, seq = vb >> 3, nx = (n_tiles + 7) >> 3;
        const int group = seq / (nx * 8), base = group * 8;
        const int gw = co_tiles - base < 8 ? co_tiles - base : 8;
        const int idx = seq - nx * base, n_local = idx / gw;
        const int co_tile = base + idx - n_local * gw, n_tile = xcd + 8 * n_local;
        if (n_tile >= n_tiles) continue;
        for (int s = 0; s < nstages; ++s) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int r = (wave * NT + i) * ROWS_PER + lrow;  // row of the tile
                const char* gx = x + ((size_t)(n_tile * 256 + r) * K + s * BK) * 2 + kc * 16;
                const char* gw_ = w + ((size_t)(co_tile * 256 + r) * K + s * BK) * 2 + kc * 16;
                char* d = smem + (slot & 1) * 65536 + (wave * NT + i) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gx,
                                                 (__attribute__((address_space(3))) void*)(d + 32768), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw_,
                                                 (__attribute__((address_space(3))) void*)d, 16, 0, 0);
            }
            wait_vm<(DEPTH - 1) * 2 * NT>();
            asm volatile("s_barrier" ::: "memory");
            ++slot;
        }
    }
    wait_vm<0>();
}
// asymmetric ring: per stage the 4 weight transfers (stage s+1) are issued before the 4 activation transfers (stage s+2), and the
// wait leaves the LEAVE youngest transfers in flight (LEAVE = 4: weights one stage ahead, activations two; 64 + 32 KiB in flight)
template <int LEAVE>
__global__ __launch_bounds__(512) void probe_asym(const char* x, const char* w, int n_tiles, int co_tiles, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nstages = K / 64;
    const int total = ((n_tiles + 7) >> 3) * 8 * co_tiles;
    const int lrow = lane >> 3, kc = lane & 7;
    int slot = 0;
    for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
        const int xcd = vb & 7, seq = vb >> 3, nx = (n_tiles + 7) >> 3;
        const int group = seq / (nx * 8), base = group * 8;
        const int gw = co_tiles - base < 8 ? co_tiles - base : 8;
        const int idx = seq - nx * base, n_local = idx / gw;
        const int co_tile = base + idx - n_local * gw, n_tile = xcd + 8 * n_local;
        if (n_tile >= n_tiles) continue;
        for (int s = 0; s < nstages; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (wave * 4 + i) * 8 + lrow;
                const char* gw_ = w + ((size_t)(co_tile * 256 + r) * K + s * 64) * 2 + kc * 16;
                char* d = smem + (slot & 1) * 65536 + (wave * 4 + i) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw_,
                                                 (__attribute__((address_space(3))) void*)d, 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (wave * 4 + i) * 8 + lrow;
                const char* gx = x + ((size_t)(n_tile * 256 + r) * K + s * 64) * 2 + kc * 16;
                char* d = smem + (slot & 1) * 65536 + (wave * 4 + i) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gx,
                                                 (__attribute__((address_space(3))) void*)(d + 32768), 16, 0, 0);
            }
            wait_vm<LEAVE>();
            asm volatile("s_barrier" ::: "memory");
            ++slot;
        }
    }
    wait_vm<0>();
}
template <int LEAVE>
void run_asym(const char* x, const char* w, int n_tiles, int co_tiles, int K, int N, int C) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_asym<LEAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        probe_asym<LEAVE><<<256, 512, 131072>>>(x, w, n_tiles, co_tiles, K);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)n_tiles * co_tiles * (K / 64) * 65536.0;
    printf("K=%d rows=%d cout=%d w-then-x order, %d youngest transfers left in flight (%d KiB in flight): %.1f us, %.2f TB/s into LDS, equivalent %.0f TFLOP/s\n", K, N, C,
           LEAVE, 64 + LEAVE * 8, best * 1e3, bytes / (best * 1e-3) / 1e12, 2.0 * N * (double)K * C / (best * 1e-3) / 1e12);
}
template <int DEPTH, int HALF>
void run(const char* x, const char* w, int n_tiles, int co_tiles, int K, int N, int C) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<DEPTH, HALF>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        probe<DEPTH, HALF><<<256, 512, 131072>>>(x, w, n_tiles, co_tiles, K);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)n_tiles * co_tiles * (K / 64) * 65536.0;
    printf("K=%d rows=%d cout=%d %s-wide stages, %d in flight (%d KiB): %.1f us, %.2f TB/s into LDS, equivalent %.0f TFLOP/s\n", K, N, C,
           HALF ? "32" : "64", DEPTH, DEPTH * (HALF ? 32 : 64), best * 1e3, bytes / (best * 1e-3) / 1e12,
           2.0 * N * (double)K * C / (best * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 3072, N = argc > 2 ? atoi(argv[2]) : 76288, C = argc > 3 ? atoi(argv[3]) : 3072;
    const int n_tiles = (N + 255) / 256, co_tiles = C / 256;
    char *x, *w;
    hipMalloc(&x, (size_t)n_tiles * 256 * K * 2);
    hipMalloc(&w, (size_t)C * K * 2);
    hipMemset(x, 0, (size_t)n_tiles * 256 * K * 2);
    hipMemset(w, 0, (size_t)C * K * 2);
    run<1, 0>(x, w, n_tiles, co_tiles, K, N, C);
    run<2, 0>(x, w, n_tiles, co_tiles, K, N, C);
    run<3, 0>(x, w, n_tiles, co_tiles, K, N, C);
    run<4, 0>(x, w, n_tiles, co_tiles, K, N, C);
    run_asym<0>(x, w, n_tiles, co_tiles, K, N, C);
    run_asym<1>(x, w, n_tiles, co_tiles, K, N, C);
    run_asym<2>(x, w, n_tiles, co_tiles, K, N, C);
    run_asym<3>(x, w, n_tiles, co_tiles, K, N, C);
    run_asym<4>(x, w, n_tiles, co_tiles, K, N, C);
    run_asym<6>(x, w, n_tiles, co_tiles, K, N, C);
    run_asym<8>(x, w, n_tiles, co_tiles, K, N, C);
    run<2, 1>(x, w, n_tiles, co_tiles, K, N, C);
    run<3, 1>(x, w, n_tiles, co_tiles, K, N, C);
    run<4, 1>(x, w, n_tiles, co_tiles, K, N, C);
    run<6, 1>(x, w, n_tiles, co_tiles, K, N, C);
    return 0;
}
