// What does one wave64 vector instruction cost on gfx950?  Issue-rate probe behind the Fbank kernel's VALU floor:
// 256 CUs x (1 | 2 | 4) waves per SIMD, each wave runs N iterations of 32 INDEPENDENT instructions of one kind (16 accumulator chains x 2),
// timed with HIP events and s_memtime / wall clock; prints cycles per wave-instruction per SIMD at the clock the run sustained.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate_probe.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void probe(float* out, int iters, unsigned long long* clk) {
    float a[16];
    float2v p[16];
    const float s = 1.0f + 1e-9f * threadIdx.x, t = 1e-9f * threadIdx.x;
    for (int i = 0; i < 16; ++i) {
        a[i] = (float)i + t;
        p[i] = float2v{(float)i, (float)i + t};
    }
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t));
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(t));
                if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(float2v{s, s}), "v"(float2v{t, t}));
                if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(float2v{t, t}));
                if (KIND == 5) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (KIND == 6) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(t));
                if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(t));
                if (KIND == 8) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 9) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
            }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float acc = 0.0f;
    for (int i = 0; i < 16; ++i) acc += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

template <int KIND>
void run(const char* name, int cus, float* d, unsigned long long* dclk) {
    const int iters = 4096;
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        probe<KIND><<<cus, 256 * wps>>>(d, 16, dclk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<KIND><<<cus, 256 * wps>>>(d, iters, dclk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long clk[2];
        hipMemcpy(clk, dclk, sizeof(clk), hipMemcpyDeviceToHost);
        const double instr_per_wave = (double)iters * 32;
        // the wave's own span in shader cycles (s_memtime) / (its instructions x the waves sharing its SIMD)
        const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);   // wall_clock64: 100 MHz
        printf("%-14s %d wave(s)/SIMD: %7.1f us, %5.2f shader cycles per wave-instruction per SIMD (wave 0: %.2f cycles per own instruction, %.2f GHz)\n", name, wps,
               ms * 1e3, (double)clk[0] / (instr_per_wave * wps), (double)clk[0] / instr_per_wave, ghz);
    }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* d;
    unsigned long long* dclk;
    hipMalloc(&d, (size_t)cus * 2048 * 4);
    hipMalloc(&dclk, 16);
    printf("%s, %d CUs\n", prop.name, cus);
    run<0>("v_fma_f32", cus, d, dclk);
    run<1>("v_add_f32", cus, d, dclk);
    run<2>("v_mul_f32", cus, d, dclk);
    run<9>("v_sub_f32", cus, d, dclk);
    run<3>("v_pk_fma_f32", cus, d, dclk);
    run<4>("v_pk_add_f32", cus, d, dclk);
    run<5>("v_mov_b32_dpp", cus, d, dclk);
    run<6>("v_add_f32_dpp", cus, d, dclk);
    run<7>("v_cndmask_b32", cus, d, dclk);
    run<8>("v_log_f32", cus, d, dclk);
    return 0;
}
