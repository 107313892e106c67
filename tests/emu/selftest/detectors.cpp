// Self-test of the emulator's checking modes (tests/emu/hip_emu.h header): two deliberately broken kernels show that the modes detect what
// they are for.  TEST INFRASTRUCTURE ONLY; built and run by tests/test_emu_detectors.py.
//   detectors race <barrier 0|1>   a cross-wave exchange through LDS; without the barrier the forward thread order happens to give the right
//                                  answer (the producer thread runs first), the reverse order does not.  Prints the number of wrong values.
//   detectors lds_oob <n>          reads one int n elements behind a 1 KiB dynamic LDS block (n = 0: the last element inside); the
//                                  AddressSanitizer build must stop at n > 0.
//   detectors global_oob <n>       the same behind a heap buffer of 256 ints.
//   detectors dma <waited 0|1>     wave 0 fetches 1 KiB into LDS with an untracked LDS-DMA transfer and every wave reads it behind an LDS-only
//                                  barrier; without the counted wait the default emulator (transfers land at issue) still gives the right
//                                  answer, MV_EMU_DMA=lazy does not.  Prints the number of wrong values.
//   detectors ldsread <waited 0|1> every thread issues a hand-scheduled LDS fragment read (lds_read1) of data it wrote itself and uses the register with or without
//                                  the counted wait (lds_wait<0>): right either way on the default emulator, NaNs without the wait under MV_EMU_LDS=lazy.
//   detectors uninit 0             reads dynamic LDS and a hipMalloc block that nobody wrote: zeros by default, -1 under MV_EMU_POISON=1.
#include <arch/gfx950.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void exchange_kernel(int* out, int with_barrier) {
    MV_DYN_SMEM(smem);
    int* v = reinterpret_cast<int*>(smem);
    const int t = threadIdx.x;
    v[t] = t + 1;
    if (with_barrier) __syncthreads();
    out[t] = t >= 64 ? v[t - 64] : v[t];   // the value of the same lane one wave below
}

__global__ void dma_kernel(const int* src, int* out, int waited) {
    MV_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        mv::glds16_untracked(reinterpret_cast<const char*>(src) + lane * 16, mv::lds_addr(smem));
        if (waited) mv::wait_vm<0>();
    }
    mv::lds_barrier();
    out[threadIdx.x] = reinterpret_cast<const int*>(smem)[threadIdx.x];
}

__global__ void lds_fragment_kernel(float* out, int waited) {
    MV_DYN_SMEM(smem);
    half8v* v = reinterpret_cast<half8v*>(smem);
    const int t = threadIdx.x;
    half8v mine;
    for (int e = 0; e < 8; ++e) mine[e] = (half_t)(t + e);
    v[t] = mine;
    __syncthreads();
    half8v got;
    mv::lds_read1(got, mv::lds_addr(v + t));
    if (waited) mv::lds_wait<0>(got);
    float sum = 0.0f;
    for (int e = 0; e < 8; ++e) sum += (float)got[e];
    out[t] = sum;
}

__global__ void lds_read_kernel(int* out, int index) {
    MV_DYN_SMEM(smem);
    int* v = reinterpret_cast<int*>(smem);
    if (threadIdx.x == 0) out[0] = v[index];
}

__global__ void global_read_kernel(const int* in, int* out, int index) {
    if (threadIdx.x == 0) out[0] = in[index];
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int n = atoi(argv[2]);
    if (strcmp(argv[1], "race") == 0) {
        std::vector<int> out(256, 0);
        int* outp = out.data();   // (the launch macro's closure copies what it names)
        MV_LAUNCH(exchange_kernel, (1, 1, 1), (256, 1, 1), 256 * sizeof(int), nullptr, outp, n);
        int wrong = 0;
        for (int t = 0; t < 256; ++t) wrong += out[t] != (t >= 64 ? t - 64 : t) + 1;
        printf("wrong=%d\n", wrong);
        return 0;
    }
    if (strcmp(argv[1], "ldsread") == 0) {
        std::vector<float> out(64, 0.0f);
        float* outp = out.data();
        MV_LAUNCH(lds_fragment_kernel, (1, 1, 1), (64, 1, 1), 1024, nullptr, outp, n);
        int wrong = 0;
        for (int t = 0; t < 64; ++t) wrong += !(out[t] == 8.0f * t + 28.0f);
        printf("wrong=%d\n", wrong);
        return 0;
    }
    if (strcmp(argv[1], "dma") == 0) {
        std::vector<int> src(256), out(256, 0);
        for (int t = 0; t < 256; ++t) src[t] = 1000 + t;
        const int* srcp = src.data();
        int* outp = out.data();
        MV_LAUNCH(dma_kernel, (1, 1, 1), (256, 1, 1), 1024, nullptr, srcp, outp, n);
        int wrong = 0;
        for (int t = 0; t < 256; ++t) wrong += out[t] != 1000 + t;
        printf("wrong=%d\n", wrong);
        return 0;
    }
    int result = 0;
    int* resp = &result;
    if (strcmp(argv[1], "uninit") == 0) {
        int* fresh = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&fresh), 1024) != hipSuccess) return 3;
        MV_LAUNCH(lds_read_kernel, (1, 1, 1), (64, 1, 1), 1024, nullptr, resp, 17);
        printf("lds=%d global=%d\n", result, fresh[17]);
        hipFree(fresh);
        return 0;
    }
    if (strcmp(argv[1], "lds_oob") == 0) {
        MV_LAUNCH(lds_read_kernel, (1, 1, 1), (64, 1, 1), 1024, nullptr, resp, 255 + n);
    } else if (strcmp(argv[1], "global_oob") == 0) {
        int* in = static_cast<int*>(malloc(256 * sizeof(int)));
        memset(in, 0, 256 * sizeof(int));
        MV_LAUNCH(global_read_kernel, (1, 1, 1), (64, 1, 1), 0, nullptr, in, resp, 255 + n);
        free(in);
    } else {
        return 2;
    }
    printf("read=%d\n", result);
    return 0;
}
