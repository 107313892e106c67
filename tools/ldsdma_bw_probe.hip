// Throughput probe: L2-resident streaming into a CU by (0) global_load_lds dwordx4, (1) global_load_dwordx4 into VGPRs,
// (2) half of the bytes by each path.  256 workgroups x 512 threads, every workgroup re-reads its own 256 KiB window
// (L2 resident after the first pass).  hipcc --offload-arch=gfx950 -O3 -o tools/probe/ldsdma_bw tools/ldsdma_bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float4v __attribute__((ext_vector_type(4)));
constexpr int STAGE = 65536;  // bytes per stage per workgroup (what the 256 x 256 conv tile moves per K stage)

template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* src, float* sink, int iters, size_t window, size_t wg_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * wg_stride;
    float4v acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const char* p = base + ((size_t)it * STAGE) % window;
        // 64 KiB per stage = 64 transfers of 1 KiB; 8 per wave
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const char* g = p + (wave * 8 + i) * 1024 + lane * 16;
            const bool dma = MODE == 0 || (MODE == 2 && (i & 1));
            if (dma) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(smem + (wave * 8 + i) * 1024), 16, 0, 0);
            } else {
                acc += *reinterpret_cast<const float4v*>(g);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0] + smem[tid];
}

int main(int argc, char** argv) {
    const size_t window = (argc > 1 ? atoi(argv[1]) : 256) * 1024;
    const bool shared = argc > 2 && atoi(argv[2]) != 0;  // every workgroup reads the same window (like the weights)
    char* src;
    float* sink;
    hipMalloc(&src, 256 * window);
    hipMemset(src, 0, 256 * window);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) probe<0><<<256, 512, 65536>>>(src, sink, iters, window, shared ? 0 : window);
            if (mode == 1) probe<1><<<256, 512, 65536>>>(src, sink, iters, window, shared ? 0 : window);
            if (mode == 2) probe<2><<<256, 512, 65536>>>(src, sink, iters, window, shared ? 0 : window);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1)
                printf("window %zu KiB %s, mode %d (%s): %.1f us per stage-set, %.2f TB/s aggregate, %.1f GB/s per CU\n", window / 1024, shared ? "shared" : "private", mode,
                       mode == 0 ? "LDS-DMA" : (mode == 1 ? "VGPR loads" : "half/half"), ms * 1e3 / iters,
                       256.0 * STAGE * iters / (ms * 1e-3) / 1e12, (double)STAGE * iters / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
