// Load-path probe for the 256 x 256 x 64 implicit-GEMM tile: the K loop's global->LDS traffic only, with the same
// workgroup -> (n-tile, co-tile) mapping as conv1d_glds_persistent_kernel, for two activation / weight layouts:
//   layout 0  row-major [rows][K] fp16: a stage slice = 256 rows x 128 B, rows K*2 bytes apart (what the kernels use)
//   layout 1  tile-blocked [row-tile][k-stage][256 rows][64] fp16: a stage slice = one contiguous 32 KiB block
// hipcc --offload-arch=gfx950 -O3 -o tools/probe/gemm_load tools/gemm_load_probe.hip ; tools/probe/gemm_load [K] [N_rows] [Cout]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int LAYOUT>
__global__ __launch_bounds__(512) void probe(const char* x, const char* w, int n_tiles, int co_tiles, int K, int rounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nstages = K / 64;
    const int total = ((n_tiles + 7) >> 3) * 8 * co_tiles;
    const int lrow = lane >> 3, kc = lane & 7;
    int buf = 0;
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
                const int r = (wave * 4 + i) * 8 + lrow;  // row of the tile
                const char* gx;
                const char* gw_;
                if (LAYOUT == 0) {
                    gx = x + ((size_t)(n_tile * 256 + r) * K + s * 64) * 2 + kc * 16;
                    gw_ = w + ((size_t)(co_tile * 256 + r) * K + s * 64) * 2 + kc * 16;
                } else {
                    gx = x + ((size_t)(n_tile * nstages + s) * 256 + r) * 128 + kc * 16;
                    gw_ = w + ((size_t)(co_tile * nstages + s) * 256 + r) * 128 + kc * 16;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gx,
                                                 (__attribute__((address_space(3))) void*)(smem + buf * 65536 + 32768 + (wave * 4 + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw_,
                                                 (__attribute__((address_space(3))) void*)(smem + buf * 65536 + (wave * 4 + i) * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            buf ^= 1;
        }
    }
}
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 3072, N = argc > 2 ? atoi(argv[2]) : 76288, C = argc > 3 ? atoi(argv[3]) : 3072;
    const int n_tiles = (N + 255) / 256, co_tiles = C / 256;
    char *x, *w;
    hipMalloc(&x, (size_t)n_tiles * 256 * K * 2);
    hipMalloc(&w, (size_t)C * K * 2);
    hipMemset(x, 0, (size_t)n_tiles * 256 * K * 2);
    hipMemset(w, 0, (size_t)C * K * 2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int layout = 0; layout < 2; ++layout)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (layout == 0) probe<0><<<256, 512, 131072>>>(x, w, n_tiles, co_tiles, K, 1);
            else probe<1><<<256, 512, 131072>>>(x, w, n_tiles, co_tiles, K, 1);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)n_tiles * co_tiles * (K / 64) * 65536.0;
            if (rep == 2)
                printf("K=%d rows=%d cout=%d layout %d (%s): %.1f us, %.2f TB/s into LDS, equivalent %.0f TFLOP/s\n", K, N, C, layout,
                       layout == 0 ? "row-major" : "tile-blocked", ms * 1e3, bytes / (ms * 1e-3) / 1e12,
                       2.0 * N * (double)K * C / (ms * 1e-3) / 1e12);
        }
    return 0;
}
